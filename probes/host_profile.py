"""Host-side cost of one training step (python + ctypes + torch dispatch): cProfile over a few steps at a small batch,
where the GPU is never the bottleneck.  GPU box only:  python probes/host_profile.py [batch]"""
import cProfile, importlib, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mp = importlib.import_module("x2-vlm_amd.model_pretrain")
cfgs = importlib.import_module("x2-vlm_amd.configs")
eng = importlib.import_module("x2-vlm_amd.engine")
syn = importlib.import_module("x2-vlm_amd.synthetic")
torch.manual_seed(0)
import tempfile
model = mp.XVLM(config=cfgs.pretrain_config(tempfile.mkdtemp(), "base", 224), load_vision_params=False, load_text_params=False, pretraining=True).to(dev)
model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(0, B, 30, 224).items()}


def step():
    eng.BANK.invalidate()
    model.zero_grad(set_to_none=True)
    loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                 masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
    (loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue %.2f ms/step, wall %.2f ms/step" % (host / 5 * 1e3, (time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()))
