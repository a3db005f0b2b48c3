import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cases import CASES, model_config
synthetic = importlib.import_module("x2-vlm_amd.synthetic")
import test_graph_gpu as T
graph = importlib.import_module("x2-vlm_amd.graph")


eng = importlib.import_module("x2-vlm_amd.engine")


def grads(use_graph, nrep, side=True, overlap=True):
    model, c = T._build(synthetic, train=False)
    eng.SIDE.enabled = side
    model.overlap_towers = overlap
    data = T._batches(synthetic, c, 2)
    static = {k: v.clone() for k, v in data[0].items()}
    params = list(model.parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = model(static["image"], static["text_ids"], static["text_atts"], text_ids_masked=static["text_ids_masked"],
                     masked_pos=static["masked_pos"], masked_ids=static["masked_ids"])
        sum(loss.values()).backward()
        return loss
    step = graph.GraphedStep(fwd_bwd, enabled=use_graph)
    out = []
    for i in range(nrep):
        graph.GraphedStep.copy_inputs(static, data[i % 2])
        step()
        torch.cuda.synchronize()
        out.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    return out


runs = {"eager": grads(False, 3), "eager-noside": grads(False, 3, side=False), "eager-nooverlap": grads(False, 3, overlap=False),
        "eager-serial": grads(False, 3, side=False, overlap=False), "graph": grads(True, 3), "graph-serial": grads(True, 3, side=False, overlap=False),
        "eager2": grads(False, 3)}
ref = runs["eager-serial"]
for name, r in runs.items():
    line = []
    for i in range(3):
        bad = [(n, float((ref[i][n].double() - r[i][n].double()).abs().max()) / max(float(ref[i][n].double().abs().max()), 1e-12)) for n in ref[i]]
        bad = [b for b in bad if b[1] > 1e-5 and "key.bias" not in b[0]]
        line.append("%d(%s %.1e)" % (len(bad), max(bad, key=lambda t: t[1])[0][-40:] if bad else "-", max([b[1] for b in bad] or [0])))
    print("%-16s vs eager-serial: %s" % (name, "  ".join(line)))
