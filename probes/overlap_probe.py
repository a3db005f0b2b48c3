"""Can an NT GEMM's epilogue (HBM stores + GELU VALU) overlap another workgroup's main loop on the same CU at all?
Two streams: (1) fc1-shaped GEMM with the epilogue switched off (main loops only), (2) the same output with K = 64
(one K-step: epilogue only).  Serial sum vs concurrent wall time."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = "cuda"
M, N, Kd = 4 * 12608, 3072, 768


def mk(Kd):
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev); aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    return lambda: K.gemm_nt(A, B, bias=bias, out=out, aux=aux, act=1)


def wall(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


lib.x2_tune(3, 1)
full, epi = mk(Kd), mk(64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def main_only():
    lib.x2_tune(2, 4); full(); lib.x2_tune(2, 0)


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        main_only()
    with torch.cuda.stream(s2):
        epi()
    cur.wait_stream(s1); cur.wait_stream(s2)


def serial():
    main_only(); epi()


print("full GEMM            %.1f us" % wall(full))
print("main loops only      %.1f us" % wall(main_only))
print("epilogue only (K=64) %.1f us" % wall(epi))
print("serial main + epi    %.1f us" % wall(serial))
print("concurrent main||epi %.1f us" % wall(both))
for tile in (3,):
    lib.x2_tune(3, tile)
    print("tile knob %d: full %.1f  main %.1f  epi %.1f  serial %.1f  concurrent %.1f" % (tile, wall(full), wall(main_only), wall(epi), wall(serial), wall(both)))
