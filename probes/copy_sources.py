"""Which Python lines issue the small torch copy / fill / cat kernels of a step (torch.profiler with stacks).  GPU box only."""
import collections, importlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
mp = importlib.import_module("x2-vlm_amd.model_pretrain"); cfgs = importlib.import_module("x2-vlm_amd.configs")
eng = importlib.import_module("x2-vlm_amd.engine")
torch.manual_seed(0)
model = mp.XVLM(config=cfgs.pretrain_config(tempfile.mkdtemp(), "base", 224), load_vision_params=False, load_text_params=False, pretraining=True).to(dev)
model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(0, 16, 30, 224).items()}
params = list(model.parameters())


def step():
    eng.BANK.invalidate()
    for p in params:
        p.grad = None
    loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                 masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"])
    (loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

agg = collections.Counter()


class Tally(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).split(".")[1] if "." in str(func) else str(func)
        fr = next((f for f in reversed(traceback.extract_stack()) if "x2-vlm_amd" in f.filename or f.filename.endswith("bench.py")), None)
        if fr is not None and not name.startswith(("view", "detach", "reshape", "_unsafe_view", "t.", "transpose", "slice", "select", "as_strided",
                                                   "expand", "unsqueeze", "squeeze", "alias", "empty", "permute", "split", "unbind", "sym_")):
            agg[(name, "%s:%d" % (fr.filename.split("/")[-1], fr.lineno))] += 1
        return func(*args, **(kwargs or {}))


# the dispatch mode is thread-local: it sees the forward (and the Python-side backward calls made from this thread);
# autograd's worker thread needs its own
import threading
orig = torch.autograd.Function.backward
with Tally():
    step()
torch.cuda.synchronize()
for (n, src), c in agg.most_common(70):
    print("%4d  %-26s %s" % (c, n, src))
