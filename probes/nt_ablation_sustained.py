"""Loop ablations (probe build, x2_tune(2, bits)) timed as SUSTAINED back-to-back launches (events around 24 launches over rotating buffers), i.e. at the
clock the chip holds under load - probes/nt_loop_ablation.py times isolated launches between host synchronisations (cooler, higher clock).
Kernels: the two-stage 256-column kernel at 160 rows (x2_tune(1, 3) + (10, 1)), the ping-pong kernel at 160 rows (x2_tune(15, 5)).  Timing only."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
assert hasattr(lib, "x2_probe_set_buffer"), "needs the probe build"
dev = "cuda"
NSET = 8
SHAPES = [("vit dfc1", 12608, 768, 3072), ("vit qkv", 12608, 2304, 768)]
VARIANTS = [("all three", 0), ("no DMA", 32), ("no reads", 64), ("no MFMA", 128), ("MFMA only", 32 + 64), ("reads only", 32 + 128), ("DMA only", 64 + 128)]


def timeit(fn, iters=48):
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


for name, M, N, Kd in SHAPES:
    As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(NSET)]
    W = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    print("%s M=%d N=%d K=%d: sustained launch time in us (160 x 256 tiles)" % (name, M, N, Kd))
    for kname, setup, extra in (("two-stage 16x16x32", ((1, 3), (10, 1)), 0), ("ping-pong, burst", ((15, 5),), 256), ("ping-pong, spread", ((15, 5),), 0)):
        for k, v in setup:
            lib.x2_tune(k, v)
        line = "   %-20s" % kname
        for vname, bits in VARIANTS:
            assert lib.x2_tune(2, bits | extra) == 0
            t = min(timeit(lambda i: K.gemm_nt(As[i], W, out=outs[i])) for _ in range(3))
            line += " | %s %5.1f" % (vname, t)
        lib.x2_tune(2, 0)
        for k, v in setup:
            lib.x2_tune(k, 0)
        print(line, flush=True)
