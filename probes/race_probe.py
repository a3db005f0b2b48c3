"""Which ingredient of the eager weight-gradient side stream makes text-tower weight gradients differ from the serial schedule?"""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from cases import CASES, model_config
synthetic = importlib.import_module("x2-vlm_amd.synthetic")
mp = importlib.import_module("x2-vlm_amd.model_pretrain")
eng = importlib.import_module("x2-vlm_amd.engine")
case = sys.argv[1] if len(sys.argv) > 1 else "base_shallow"
c = CASES[case]
torch.manual_seed(0)
model = mp.XVLM(config=model_config(case, tempfile.mkdtemp()), load_vision_params=False, load_text_params=False, pretraining=True)
synthetic.synth_state_dict(model, c["wseed"])
model = model.cuda().eval()
batch = {k: v.cuda() for k, v in synthetic.synth_batch(c["bseed"], c["batch"], c["seq_len"], c["image_res"], c["vocab"], c["max_masks"], ragged=c["ragged"]).items()}
model.injected_negatives = tuple(torch.tensor(n, dtype=torch.int32, device="cuda") for n in synthetic.synth_negatives(c["bseed"], c["batch"]))


def step():
    for p in model.parameters():
        p.grad = None
    loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"], masked_pos=batch["masked_pos"],
                 masked_ids=batch["masked_ids"])
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def compare(ref, g):
    bad = []
    for n in ref:
        e = float((ref[n].double() - g[n].double()).abs().max()) / max(float(ref[n].double().abs().max()), 1e-12)
        if e > 2e-5 and "key.bias" not in n:
            bad.append((e, n))
    return sorted(bad, reverse=True)


def trial(name, setup, n=6):
    setup()
    res = [compare(REF, step()) for _ in range(n)]
    print("%-34s mismatching tensors per iteration: %s   worst: %s" % (name, [len(r) for r in res], max((r[0] for r in res if r), default=None)), flush=True)


def cfg(side=True, overlap=True, pair=True, hold=False):
    def f():
        eng.SIDE.enabled, model.overlap_towers, eng._LayerPairs.enabled = side, overlap, pair
        eng.SIDE.hold_forever = hold
    return f


lib = importlib.import_module("x2-vlm_amd._lib").lib()
if len(sys.argv) > 2:
    lib.x2_tune(5, int(sys.argv[2]))
    print("x2_tune(5, %s)" % sys.argv[2])
cfg(False, False)()
REF = step()
trial("serial (again)", cfg(False, False))
trial("side + overlap (shipping eager)", cfg())
