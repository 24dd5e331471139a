"""hipGraph capture of a whole training step (forward + backward) of layers on the HIP path.

At the reference's real layer shapes (4 experts, 1152 x 4304 SigLIP towers; BASELINE config 1's tiny LM) a step is ~60 launches
of 5-50 us each and the host cannot issue them fast enough: the step is launch-bound.  Replaying the step as ONE graph removes
the per-launch host cost.

What makes a step capturable here (round-1's attempt died inside the capture; the causes, all host-side):
  * the AccumulateGrad nodes of the parameters must live on the capture stream.  Found by bisection on the GPU (tools/graph_bisect.py,
    tools/graph_bisect2.py; gpurun_out r2c): every kernel of the path, the whole forward and a fwd+bwd step capture and replay
    cleanly -- UNLESS an eager step ran on the default stream first AND something kept its autograd graph alive.  `MoeLayer` did:
    `log_metrics` held `route.w` / `route.softmax` with their grad_fn, so the parameters' AccumulateGrad nodes of that eager step
    survived, the engine inserted a wait on the default stream inside the capture, and the HIP runtime segfaulted in
    hipStreamEndCapture.  Fixed at the source (the metrics are stored detached); warm-up iterations run on the SAME side stream the
    capture uses (torch's whole-network recipe), and GraphedStep refuses to capture -- with the reason -- if the warm-up still sees
    a stale node (a caller holding last step's loss / outputs);
  * no pageable host-to-device copy inside the capture: per-expert pointer tables (`ops.ptr_array`), chunk / segment offset
    tables and the host copy of the competition schedule are built by the warm-up iterations and only LOOKED UP while capturing
    (`ops.ptr_array` raises a clear error if a capture would have to build one);
  * tables cached by address are not inserted while capturing (kernels do not run during capture, so a table first "computed"
    there holds garbage until the first replay);
  * the launch geometry never depends on routing (grids are upper bounds from shapes; offsets are read on the device), so a
    replay with other inputs routes correctly.
Branches taken on the host (CompeteSMoE's competition / router step, `log_interval` statistics) are frozen into the graph: capture
one graph per branch and pick on the host, as the schedule is known ahead (`prob_flips`).
"""
from __future__ import annotations

import warnings
from typing import Callable, Sequence

import torch


class GraphedStep:
    """`loss = fn(*inputs); loss.backward()` captured once, replayed on every call.

    fn        : callable(*tensors) -> scalar loss tensor (may also return a tuple whose first element is the loss)
    inputs    : example input tensors (device); their storage becomes the graph's static input buffers
    params    : parameters whose .grad the step produces (zeroed to None before the capture so the graph owns the buffers)
    After `step(*new_inputs)` the outputs are in `step.outputs` and every gradient in `p.grad` (static buffers: copy them or
    run the optimizer before the next replay)."""

    def __init__(self, fn: Callable, inputs: Sequence[torch.Tensor], params: Sequence[torch.nn.Parameter], warmup: int = 3):
        self.fn = fn
        self.params = list(params)
        self.static_inputs = [t.detach().clone().requires_grad_(t.requires_grad) for t in inputs]
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        stale = False
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)          # the engine's stream-mismatch warning is a warn-ONCE: it must fire here even if it fired before
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(max(1, warmup)):
                    self._zero()
                    with warnings.catch_warnings(record=True) as seen:
                        warnings.simplefilter("always")
                        self._run()
                    stale = any("AccumulateGrad node's stream does not match" in str(w.message) for w in seen)
        finally:
            torch.set_warn_always(warn_always)
        torch.cuda.current_stream().wait_stream(self.stream)
        if stale:
            raise RuntimeError(
                "competesmoe_amd.graphs: a parameter's AccumulateGrad node from an earlier step (created on another stream) is still "
                "alive -- some tensor of that step with a grad_fn (its loss, outputs, logged metrics) is still referenced.  Capturing "
                "now would make the autograd engine wait on that stream inside the capture, which the HIP runtime does not survive. "
                "Drop those references (del loss / outputs) and build the GraphedStep again.")
        torch.cuda.synchronize()
        self._zero()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.outputs = self._run()
        self.input_grads = [t.grad for t in self.static_inputs]

    def _zero(self):
        for p in self.params:
            p.grad = None
        for t in self.static_inputs:
            t.grad = None

    def _run(self):
        out = self.fn(*self.static_inputs)
        loss = out[0] if isinstance(out, (tuple, list)) else out
        loss.backward()
        return out

    def __call__(self, *inputs: torch.Tensor):
        for s, t in zip(self.static_inputs, inputs):
            if t is not s:
                s.data.copy_(t)
        self.graph.replay()
        return self.outputs
