"""Row-wise (HBM-bound) kernels of the step on cold buffers: NSET rotating buffer sets (> the 256 MB Infinity Cache in total),
so that every launch reads from and writes to HBM as inside the step.  Prints us per launch and TB/s of algorithmic bytes."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"
NSET = int(os.environ.get("NSET", "8"))


def timeit(fns, reps=6):
    """fns: one closure per buffer set; walks them round-robin."""
    for f in fns: f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for f in fns: f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * len(fns)) * 1e3


def report(name, us, nbytes):
    print("%-44s %7.1f us  %6.1f MB  %5.2f TB/s" % (name, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)


for R, D in ((12608, 768), (7680, 768), (3840, 768), (18464, 1024)):
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    xs = [torch.randn(R, D, device=dev) for _ in range(NSET)]
    ys = [torch.empty(R, D, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    st = [K.layernorm_fwd(xs[0], w, b, 1e-6, y_bf16=ys[0])[2:] for _ in range(1)][0]
    us = timeit([lambda x=x, y=y: K.layernorm_fwd(x, w, b, 1e-6, y_bf16=y) for x, y in zip(xs, ys)])
    report("layernorm_fwd [%d, %d] fp32 -> bf16" % (R, D), us, R * D * 6)
    mean, rstd = st
    dw, db, dcol = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dyb = [torch.randn(R, D, device=dev).bfloat16() for _ in range(NSET)]
    dres = [torch.randn(R, D, device=dev) for _ in range(NSET)]
    dxs = [torch.empty(R, D, device=dev) for _ in range(NSET)]
    # pre-LN block: bf16 incoming gradient, fp32 residual gradient added, fp32 out (norm1 / norm2 of a vision block)
    us = timeit([lambda a=a, x=x, r=r, o=o: K.layernorm_bwd(a, x, mean, rstd, w, dw, db, dres=r, dx=o) for a, x, r, o in zip(dyb, xs, dres, dxs)])
    report("layernorm_bwd bf16 dy + dres -> fp32 (+stage 2)", us, R * D * (2 + 4 + 4 + 4))
    # post-LN (BERT): fp32 incoming gradient, fp32 + bf16 out, column sums
    us = timeit([lambda a=a, x=x, o=o: K.layernorm_bwd(a, x, mean, rstd, w, dw, db, dcol=dcol, dx=o, want_bf16=True) for a, x, o in zip(dres, xs, dxs)])
    report("layernorm_bwd fp32 dy -> fp32 + bf16 + dcol", us, R * D * (4 + 4 + 4 + 2))
    K.DEFERRED = []
    us = timeit([lambda a=a, x=x, r=r, o=o: K.layernorm_bwd(a, x, mean, rstd, w, dw, db, dres=r, dx=o) for a, x, r, o in zip(dyb, xs, dres, dxs)])
    report("   the same, stage 1 only (deferred)", us, R * D * (2 + 4 + 4 + 4))
    items = K.DEFERRED[:NSET]
    K.DEFERRED = None
    us = timeit([lambda it=it: K.reduce_partials_multi([it]) for it in items])
    report("   its stage 2 (reduce_partials_multi, 1 item)", us, items[0][1] * 3 * D * 4)
    us = timeit([lambda i=i: K.reduce_partials_multi([items[i], items[(i + 1) % NSET], items[(i + 2) % NSET], items[(i + 3) % NSET]]) for i in range(NSET)])
    report("   stage 2, 4 items in one launch", us, 4 * items[0][1] * 3 * D * 4)
    cs = torch.zeros(D, device=dev)
    us = timeit([lambda x=x: K.rowscale_cast_colsum(x, cs) for x in dxs])
    report("rowscale_cast_colsum fp32 dx -> bf16 + colsum", us, R * D * (4 + 2))
    us = timeit([lambda a=a, x=x, r=r, o=o: K.layernorm_bwd(a, x, mean, rstd, w, dw, db, dres=r, dx=o, post=(None, cs)) for a, x, r, o in zip(dyb, xs, dres, dxs)])
    report("layernorm_bwd bf16 dy + dres -> fp32 + bf16 + colsum (post)", us, R * D * (2 + 4 + 4 + 4 + 2))
    big = [torch.randn(R, 4 * D, device=dev).bfloat16() for _ in range(max(2, NSET // 2))]
    o4 = torch.zeros(4 * D, device=dev)
    us = timeit([lambda y=y: K.colsum_bf16(y, o4) for y in big])
    report("colsum_bf16 [%d, %d]" % (R, 4 * D), us, R * 4 * D * 2)
    us = timeit([lambda y=y: K.colsum_bf16(y, dcol) for y in dyb])
    report("colsum_bf16 [%d, %d]" % (R, D), us, R * D * 2)
    del xs, ys, dyb, dres, dxs, big
    torch.cuda.empty_cache()
