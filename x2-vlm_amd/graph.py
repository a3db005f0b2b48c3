"""hipGraph capture of a whole training step.

A step of this path is ~1300 (base) to ~2500 (large) kernel launches issued from Python through ctypes; at 15-25 us of
host time each the host, not the GPU, sets the step time (r01: host 29.2 ms of a 29.5 ms step).  The step's launch
sequence is static - same kernels, same shapes, same buffers every iteration - so it is captured ONCE into a hipGraph
(all streams: vision / text towers, weight-gradient side streams, the gradient all-reduce stream) and replayed with a
single host call per iteration.

What had to be true for that, and is:
  * no host round trip inside the step (hard negatives, CE normalisers, degenerate-box test are device-side);
  * dropout / DropPath randomness that does not live in kernel arguments: every dropout site mixes the device word
    `kernels.DROP_EPOCH` - incremented by the graph itself - into its seed (csrc/x2_common.h drop_at_epoch); DropPath
    keeps and the hard-negative uniforms come from torch's graph-safe Philox generator;
  * ROCm 7's stream capture only survives side streams that fork from and join into the capture's origin stream
    (probes/graph_capture_probe.py): the text tower keeps its own stream, the weight-gradient side stream is used by the
    origin stream's stages only (engine.SideStream.only_from);
  * static addresses: inputs are copied into the tensors the step was captured on (`GraphedStep.copy_inputs`), parameter
    gradients stay in the arenas allocated during capture (do not set .grad to None between replays);
  * the optimizer stays OUTSIDE the graph (two eager launches; its per-parameter step counters live on the host).

Usage (bench.py; INTEGRATION.md shows the same six lines inside Pretrain.run_image_iter):
    step = GraphedStep(lambda: fwd_bwd(model, static_batch))     # warm-up + capture
    for batch in loader:
        step.copy_inputs(static_batch, batch)
        losses = step()                                           # one hipGraphLaunch
        optimizer.step()
If capture is impossible (e.g. a process group whose collectives cannot be captured: gloo) the object degrades to eager
execution of `fn` and says so in `.mode`."""
import gc
import os

import torch

from . import kernels as K


class GraphedStep:
    def __init__(self, fn, warmup=2, enabled=True, verbose=False):
        self.fn, self.graph, self.out = fn, None, None
        self.stream = torch.cuda.Stream()
        self.mode = "eager"
        self.error = None
        if not enabled or os.environ.get("X2_GRAPH", "1") == "0":
            return
        from . import engine
        K.DROP_EPOCH = torch.zeros(1, dtype=torch.int32, device="cuda")
        side_rule, engine.SIDE.only_from = engine.SIDE.only_from, self.stream.cuda_stream     # see engine.SideStream.only_from
        gc.collect()                          # autograd graphs of earlier eager steps (their AccumulateGrad nodes remember the
                                              # stream they were created on) must be gone before the capture stream's own
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):           # eager, on the capture stream: caches, workspaces, AccumulateGrad streams
                K.DROP_EPOCH.add_(1)
                fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        trace = os.environ.get("X2_GRAPH_TRACE") == "1"
        try:
            if trace:
                print("GraphedStep: warm-up done, capturing", flush=True)
            with torch.cuda.graph(g, stream=self.stream):
                K.DROP_EPOCH.add_(1)
                self.out = fn()
                if trace:
                    print("GraphedStep: fn() captured, ending capture", flush=True)
            if trace:
                print("GraphedStep: graph instantiated", flush=True)
            self.graph, self.mode = g, "hipgraph"
        except Exception as e:                # noqa: BLE001 - anything that cannot be captured: run eagerly instead
            self.error = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            if verbose:
                print("GraphedStep: capture failed, running eagerly (%s)" % self.error, flush=True)
            torch.cuda.synchronize()
        engine.SIDE.only_from = side_rule
        torch.cuda.current_stream().wait_stream(self.stream)

    def __call__(self):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self.graph is not None:
                self.graph.replay()
            else:
                if K.DROP_EPOCH is not None:
                    K.DROP_EPOCH.add_(1)
                self.out = self.fn()
        cur.wait_stream(self.stream)
        return self.out

    @staticmethod
    def copy_inputs(static_batch, batch):
        """New data into the tensors the step was captured on (one multi-tensor copy)."""
        keys = [k for k in static_batch if k in batch]
        torch._foreach_copy_([static_batch[k] for k in keys], [batch[k] for k in keys])
