"""LayerNorm forward: rows per wave (x2_tune key 13: 1 = round-1 form, 2, 4) on the step's shapes, rotating over 12 buffer sets so
that inputs / outputs are NOT cache-resident (the in-step condition); interleaved rounds in one process, minimum of three.
Also checks that every variant's outputs are bit-identical to the one-row form.  GPU box only."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"
SHAPES = [("vision", 12608, 768, True, False), ("text 2B", 3840, 768, True, True), ("fusion 4B", 7680, 768, True, True),
          ("large vision", 18464, 1024, True, False), ("large fusion", 3840, 1024, True, True)]
NSET = 12


def timeit(fn, iters=24):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % NSET)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, M, D, wb, wf in SHAPES:
    xs = [torch.randn(M, D, device=dev) for _ in range(NSET)]
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    yb = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    yf = [torch.empty(M, D, device=dev) for _ in range(NSET)] if wf else [None] * NSET
    ref = None
    res = {}
    for rnd in range(3):
        for k in (1, 2, 4, 0):
            lib.x2_tune(13, k)
            t = timeit(lambda i: K.layernorm_fwd(xs[i], w, b, 1e-6, want_bf16=wb, want_f32=wf, y_bf16=yb[i], y_f32=yf[i]))
            res[k] = min(res.get(k, 1e9), t)
            out = K.layernorm_fwd(xs[0], w, b, 1e-6, want_bf16=True, want_f32=True)
            torch.cuda.synchronize()
            if ref is None:
                ref = [o.clone() for o in out]
            else:
                assert all(torch.equal(o, r) for o, r in zip(out, ref)), (name, k)
    lib.x2_tune(13, 0)
    mb = M * D * (4 + 2 + (4 if wf else 0)) / 1e6
    print("%-14s M=%6d D=%5d %6.1f MB | " % (name, M, D, mb) + "  ".join("rpw%d %6.1f us %5.2f TB/s" % (k, res[k], mb / res[k]) for k in (1, 2, 4, 0)))
