"""NT GEMM: the 128-column kernels (automatic tile choice) against the 8-wave 256-column kernel at its four tile heights,
on every (shape, epilogue feature set) the X2VLM-base and -large steps launch.  Interleaved rounds in ONE process, minimum of
three.  `noepi` = the same launch with the epilogue compiled out (x2_tune(2, 4)): main loops alone - needs the probe build
(bash probes/build_probe.sh; X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so), the shipped library refuses the switch.
    python probes/bench_nt256.py [base|large|all]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"


def timeit(fn, iters=15):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def shapes(which):
    out = []

    def tower(tag, M, D, F, n):          # pre-LN vision block: (name, M, N, K, feature set, launches per step)
        out.extend([(tag + " qkv", M, 3 * D, D, "bias", n), (tag + " proj", M, D, D, "lscale", n), (tag + " fc1", M, F, D, "gelu", n),
                    (tag + " fc2", M, D, F, "lscale", n), (tag + " dfc2", M, F, D, "dgelu", n), (tag + " dfc1", M, D, F, "plain", n),
                    (tag + " dproj", M, D, D, "plain", n), (tag + " dqkv", M, D, 3 * D, "plain", n)])

    def bert(tag, M, D, F, n, cross_M=0, Dv=0):
        out.extend([(tag + " qkv", M, 3 * D, D, "bias", n), (tag + " out", M, D, D, "resid_drop", n), (tag + " ffn1", M, F, D, "gelu", n),
                    (tag + " ffn2", M, D, F, "resid_drop", n), (tag + " dffn2", M, F, D, "dgelu", n), (tag + " dffn1", M, D, F, "resid", n),
                    (tag + " datt", M, D, D, "plain", n), (tag + " dqkv", M, D, 3 * D, "resid", n)])
        if cross_M:
            out.extend([(tag + " xq", M, D, D, "bias", n), (tag + " xkv", cross_M, 2 * D, Dv, "bias", n), (tag + " xout", M, D, D, "resid_drop", n),
                        (tag + " xdatt", M, D, D, "plain", n), (tag + " xdq", M, D, D, "resid", n), (tag + " xdkv", cross_M, Dv, 2 * D, "resid", n)])
    if which in ("base", "all"):
        tower("vit", 12608, 768, 3072, 12)
        bert("text", 3840, 768, 3072, 12)
        bert("fus", 7680, 768, 3072, 6, cross_M=12608, Dv=768)
    if which in ("large", "all"):
        tower("vitL", 18464, 1024, 4096, 24)
        bert("textL", 1920, 1024, 4096, 12)
        bert("fusL", 3840, 1024, 4096, 6, cross_M=18464, Dv=1024)
    return out


def case(M, N, Kd, epi):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    bias, gamma = torch.randn(N, device=dev), torch.randn(N, device=dev)
    f32 = epi in ("lscale", "resid", "resid_drop")
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(out=out)
    if epi != "plain" and epi != "dgelu":
        kw["bias"] = bias
    if epi == "gelu":
        kw.update(aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16), act=1)
    if epi == "dgelu":
        kw.update(aux=torch.randn(M, N, device=dev).bfloat16(), act=2)
    if epi == "lscale":
        kw.update(resid=torch.randn(M, N, device=dev), gamma=gamma)      # feature set 6: nothing saved besides the output
    if epi in ("resid", "resid_drop"):
        kw.update(resid=torch.randn(M, N, device=dev))
    if epi == "resid_drop":
        kw.update(drop=K.dropout_spec(0.1, 99, 1))
    return lambda: K.gemm_nt(A, B, **kw)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "base"
    configs = [("auto", 0, 0), ("h256", 3, 8), ("h224", 3, 7), ("h192", 3, 6), ("h160", 3, 5)]
    tot = {c[0]: 0.0 for c in configs}
    tot["best"] = 0.0
    print("%-11s %6s %5s %5s %-10s | " % ("launch", "M", "N", "K", "epilogue") + " ".join("%7s" % c[0] for c in configs) + " | best     TF   noepi(auto/best)")
    for name, M, N, Kd, epi, n in shapes(which):
        fn = case(M, N, Kd, epi)
        res = {c[0]: [] for c in configs}
        for _ in range(3):
            for cname, k1, k3 in configs:
                lib.x2_tune(1, k1); lib.x2_tune(3, k3)
                res[cname].append(timeit(fn))
        best = min(res, key=lambda c: min(res[c]))
        lib.x2_tune(2, 4)
        lib.x2_tune(1, 0); lib.x2_tune(3, 0)
        ne_auto = timeit(fn)
        k = dict((c[0], c) for c in configs)[best]
        lib.x2_tune(1, k[1]); lib.x2_tune(3, k[2])
        ne_best = timeit(fn)
        lib.x2_tune(2, 0); lib.x2_tune(1, 0); lib.x2_tune(3, 0)
        fl = 2.0 * M * N * Kd
        for c in res:
            tot[c] += n * min(res[c])
        tot["best"] += n * min(res[best])
        print("%-11s %6d %5d %5d %-10s | " % (name, M, N, Kd, epi) + " ".join("%7.1f" % min(res[c[0]]) for c in configs) +
              " | %-5s %5.0f   %6.1f / %6.1f" % (best, fl / min(res[best]) / 1e6, ne_auto, ne_best), flush=True)
    print("per step (launch counts applied), ms: " + "  ".join("%s %.2f" % (c, tot[c] / 1e3) for c in tot))


if __name__ == "__main__":
    main()
