"""hipGraph capture of a whole fwd+bwd step (competesmoe_amd/graphs.py): a captured and replayed step equals the eager step bit
for bit -- outputs, aux loss, input gradient and every parameter gradient -- also on inputs that route differently from the
ones seen at capture time; and a capture is refused while an earlier step's autograd graph is still referenced (the cause of
round 1's crash, found by bisection: tools/graph_bisect*.py).  Each case runs ONCE, in a child process (tests/graph_cases.py): a
capture that goes wrong segfaults inside the HIP runtime and must not take the pytest session with it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("case", ["llava_smoe", "pretrain_smoe", "refuses_stale_graph"])
def test_graph_capture(case):
    r = subprocess.run([sys.executable, os.path.join(HERE, "graph_cases.py"), case], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH-CASE-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
