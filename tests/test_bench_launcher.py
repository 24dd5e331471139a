"""`python3 bench.py --gpus N` typed as is must start its own N ranks: the parent process spawns
`python -m torch.distributed.run` before touching the GPU, relays rank 0's ONE JSON line and exits with the children's status.
Run here at world 2 over gloo with the CPU stand-in step (`--stub`); the N>1 GPU step itself is covered by tests/test_ep_*.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_starts_two_ranks_and_prints_one_json_line():
    r = run("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["comm_world_size"] == 2
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_children_exit_status_is_passed_on():
    """Ranks that fail (here: zero timed steps, a division by zero in every rank -- the parent's own argument parsing accepts it)
    must fail the parent, not leave an rc-0 run without a line."""
    r = run("--gpus", "2", "--stub", "--steps", "0")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_mismatched_world_is_refused():
    r = run("--gpus", "4", "--stub", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "started 1 ranks" in r.stderr
