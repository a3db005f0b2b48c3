"""GEMM micro-benchmark on the shapes of the X2VLM-base step (B=64): per-shape time and TFLOP/s,
A/B over tuning knobs in ONE process (interleaved rounds).  Run on the GPU box:  python probes/bench_gemm.py"""
import importlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


NT = [("fc1 fwd", 12608, 3072, 768, "gelu"), ("fc2 fwd", 12608, 768, 3072, "resid"), ("qkv fwd", 12608, 2304, 768, "bias"),
      ("proj fwd", 12608, 768, 768, "resid"), ("dqkv dgrad", 12608, 768, 2304, "f32"), ("text qkv", 3840, 2304, 768, "bias"),
      ("text ffn2", 3840, 768, 3072, "resid"), ("fus ffn1", 7680, 3072, 768, "gelu"), ("fus out", 7680, 768, 768, "resid"),
      ("cross kv", 12608, 1536, 768, "bias"), ("mlm dec", 768, 30528, 768, "f32"),
      ("text out", 3840, 768, 768, "resid"), ("fus ffn2", 7680, 768, 3072, "resid"), ("text dqkv", 3840, 768, 2304, "f32")]


def nt_case(M, N, Kd, epi):
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == "resid" else None
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "gelu" else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in ("resid", "f32") else torch.bfloat16)
    kw = dict(bias=bias, out=out)
    if epi == "gelu":
        kw.update(aux=aux, act=1)
    if epi == "resid":
        kw.update(resid=resid, gamma=bias)
    return lambda: K.gemm_nt(A, B, **kw)


def main():
    knobs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "3"])]
    print("NT GEMM, knob3 (1 = 128x128 tile, 2 = 192x128, 3 = 64x128; all 4 waves, 2-3 workgroups / CU) in", knobs)
    lib.x2_tune(1, 1)
    for name, M, N, Kd, epi in NT:
        fn = nt_case(M, N, Kd, epi)
        res = []
        for g in knobs:
            lib.x2_tune(3, g)
            res.append(timeit(fn))
        fl = 2.0 * M * N * Kd
        print("  %-11s M=%5d N=%5d K=%4d %-5s " % (name, M, N, Kd, epi) + "  ".join("g%-2d %6.1fus %5.0fTF" % (g, t, fl / t / 1e6) for g, t in zip(knobs, res)))
    lib.x2_tune(3, 0)
    print("write-through (sc1) output stores: knob2 = 0 plain, 16 sc1 (interleaved rounds)")
    for name, M, N, Kd, epi in NT:
        fn = nt_case(M, N, Kd, epi)
        res = {0: [], 16: []}
        for rep in range(3):
            for g in (0, 16):
                lib.x2_tune(2, g)
                res[g].append(timeit(fn, 10))
        lib.x2_tune(2, 0)
        print("  %-11s plain %6.1fus  sc1 %6.1fus" % (name, min(res[0]), min(res[16])))
    lib.x2_tune(3, 1)
    print("ablation on the 128x128 kernel (knob2: 0 full, 4 no epilogue)")
    for name, M, N, Kd, epi in NT[:5]:
        fn = nt_case(M, N, Kd, epi)
        res = []
        for g in (0, 4):
            lib.x2_tune(2, g)
            res.append((g, timeit(fn)))
        lib.x2_tune(2, 0)
        print("  %-11s " % name + "  ".join("d%d %6.1fus" % r for r in res))
    lib.x2_tune(1, 0); lib.x2_tune(3, 0)
    print("TN grouped (weight grads)")
    for name, Mc, probs in [("vit block", 12608, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                            ("text layer", 3840, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                            ("fc1 only", 12608, [(3072, 768)])]:
        ps = []
        for N, Kd in probs:
            ps.append((torch.randn(Mc, N, device=dev).bfloat16(), torch.randn(Mc, Kd, device=dev).bfloat16(), torch.empty(N, Kd, device=dev)))
        t = timeit(lambda: K.gemm_tn_grouped(ps))
        fl = sum(2.0 * Mc * N * Kd for N, Kd in probs)
        print("  %-11s Mc=%5d %d problems  %7.1fus %5.0fTF" % (name, Mc, len(probs), t, fl / t / 1e6))


if __name__ == "__main__":
    main()
