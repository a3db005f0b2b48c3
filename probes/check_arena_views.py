"""Do the per-layer gradient arenas survive autograd as p.grad storage (no clone)?  Needed for the in-place early
all-reduce of accelerator.GradientBuckets to be what the optimizer sees."""
import importlib, os, sys, tempfile, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from cases import CASES, model_config
eng = importlib.import_module("x2-vlm_amd.engine")
mp = importlib.import_module("x2-vlm_amd.model_pretrain")
synthetic = importlib.import_module("x2-vlm_amd.synthetic")
c = CASES["tiny"]
model = mp.XVLM(config=model_config("tiny", tempfile.mkdtemp()), load_vision_params=False, load_text_params=False).cuda().eval()
batch = {k: v.cuda() for k, v in synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=True).items()}
arenas = []
eng.GRAD_READY_HOOK = lambda flat, key, ev=None: arenas.append((flat.data_ptr(), flat.numel() * 4, key))
warnings.simplefilter("always")
loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
sum(loss.values()).backward()
torch.cuda.synchronize()
inside = outside = 0
out_names = []
for n, p in model.named_parameters():
    if p.grad is None:
        continue
    a = p.grad.data_ptr()
    if any(lo <= a < lo + nb for lo, nb, _ in arenas):
        inside += 1
    else:
        outside += 1; out_names.append(n)
print("arenas published:", len(arenas), "params with grad inside an arena:", inside, "outside:", outside)
print("outside:", out_names)
