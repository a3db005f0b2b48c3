"""Weight-gradient (TN) GEMM: 128x128 kernel vs 256x256 kernel on the layer shapes of the step, interleaved rounds in one
process.  GPU box only:  python probes/bench_tn.py"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


LAYER = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
CASES = [("vit block", [(12608, n, k) for n, k in LAYER]),
         ("text layer", [(3840, n, k) for n, k in LAYER]),
         ("fusion layer", [(7680, n, k) for n, k in LAYER] + [(7680, 768, 768), (12608, 1536, 768), (7680, 768, 768)]),
         ("fc1 only", [(12608, 3072, 768)])]
for name, probs in CASES:
    ps = [(torch.randn(Mc, N, device=dev).bfloat16(), torch.randn(Mc, Kd, device=dev).bfloat16(), torch.zeros(N, Kd, device=dev))
          for Mc, N, Kd in probs]
    fl = sum(2.0 * Mc * N * Kd for Mc, N, Kd in probs)
    res = {}
    for rep in range(3):
        for knob, split in ((1, 1), (2, 0), (2, 1), (2, 2), (2, 3)):
            lib.x2_tune(5, knob)
            t = timeit(lambda: K.gemm_tn_grouped(ps, split=split))
            res.setdefault((knob, split), []).append(t)
    lib.x2_tune(5, 0)
    print("%-12s " % name + "  ".join("%s/s%d %6.1fus %4.0fTF" % ("128" if k == 1 else "256", sp, min(v), fl / min(v) / 1e6)
                                      for (k, sp), v in res.items()))
