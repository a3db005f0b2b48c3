"""Which NT kernel should serve which launch, now that the 160-row 256-column kernel has the three-stage ring (round 5)?  Every (shape,
feature set) of the base / large steps: the library's automatic choice against the 256-column kernel forced at 160 / 192 / 256 rows,
operands and outputs rotating over 8 buffer sets (not cache-resident: the in-step condition), minimum of 3 interleaved rounds.
    python probes/bench_nt_choice.py [base|large|all]        GPU box only."""
import importlib, importlib.util, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch
spec = importlib.util.spec_from_file_location("bench_nt256", os.path.join(HERE, "bench_nt256.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
K, lib, dev = b.K, b.lib, b.dev
NSET = 8


def case(M, N, Kd, epi):
    As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(NSET)]
    B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
    f32 = epi in ("lscale", "resid", "resid_drop")
    outs = [torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(NSET)]
    auxs = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NSET)] if epi in ("gelu", "dgelu") else None
    resid = torch.randn(M, N, device=dev) if f32 else None

    def run(i):
        kw = dict(out=outs[i])
        if epi not in ("plain", "dgelu"):
            kw["bias"] = bias
        if epi == "gelu":
            kw.update(aux=auxs[i], act=1)
        if epi == "dgelu":
            kw.update(aux=auxs[i], act=2)
        if epi == "lscale":
            kw.update(resid=resid, gamma=gamma)
        if epi in ("resid", "resid_drop"):
            kw.update(resid=resid)
        if epi == "resid_drop":
            kw.update(drop=K.dropout_spec(0.1, 99, 1))
        K.gemm_nt(As[i], B, **kw)
    return run


def timeit(fn, iters=16):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % NSET)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "base"
configs = [("auto", 0, 0), ("h160", 3, 5), ("h192", 3, 6), ("h256", 3, 8)]
tot = {c[0]: 0.0 for c in configs}
tot["best"] = 0.0
print("%-11s %6s %5s %5s %-10s | " % ("launch", "M", "N", "K", "epilogue") + " ".join("%7s" % c[0] for c in configs) + " | best    vs auto")
for name, M, N, Kd, epi, n in b.shapes(which):
    fn = case(M, N, Kd, epi)
    res = {c[0]: 1e9 for c in configs}
    for _ in range(3):
        for cname, k1, k3 in configs:
            lib.x2_tune(1, k1); lib.x2_tune(3, k3)
            res[cname] = min(res[cname], timeit(fn))
    lib.x2_tune(1, 0); lib.x2_tune(3, 0)
    best = min(res, key=res.get)
    for c in res:
        tot[c] += n * res[c]
    tot["best"] += n * res[best]
    print("%-11s %6d %5d %5d %-10s | " % (name, M, N, Kd, epi) + " ".join("%7.1f" % res[c[0]] for c in configs) +
          " | %-6s %+5.1f %%" % (best, 100 * (res[best] / res["auto"] - 1)), flush=True)
    del fn
    torch.cuda.empty_cache()
print("per step (launch counts applied), ms: " + "  ".join("%s %.2f" % (c, tot[c] / 1e3) for c in tot))
