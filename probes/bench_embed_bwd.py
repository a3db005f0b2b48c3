"""x2_embed_bwd at the step's shapes (token rows with [CLS] / [SEP] / [MASK] on hundreds of rows): us per call.  GPU box only."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("x2-vlm_amd.kernels")
dev = "cuda"
for name, S, L, D in (("base: 128 sequences x 30", 128, 30, 768), ("region: 256 x 30", 256, 30, 768), ("large: 64 x 30, D = 1024", 64, 30, 1024)):
    g_ = torch.Generator().manual_seed(S)
    ids = torch.randint(1000, 30000, (S, L), generator=g_)
    ids[:, 0] = 101; ids[:, -1] = 102
    half = S // 2
    pos = torch.stack([torch.randperm(L - 2, generator=g_)[:12] + 1 for _ in range(half)])
    m = ids[half:].clone(); m.scatter_(1, pos, 103); ids[half:] = m          # the masked copy of the captions: [MASK] on ~10 of 30 positions
    ids = ids.to(dev); g = torch.randn(S * L, D, device=dev)
    dw, dp, dt = torch.zeros(30522, D, device=dev), torch.zeros(512, D, device=dev), torch.zeros(2, D, device=dev)
    for _ in range(3):
        K.embed_bwd(ids, g, dw, dp, dt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        K.embed_bwd(ids, g, dw, dp, dt)
    b.record(); torch.cuda.synchronize()
    print("%-28s %7.1f us per x2_embed_bwd (word + position + type kernels)" % (name, a.elapsed_time(b) / 20 * 1e3))
