"""Same-process A/B of kernel-variant knobs on the replayed training step: the model is built once; every variant sets its
x2_tune knobs, captures a graph.SegmentedStep of its own (the knobs pick kernels at capture time) and times `--steps` replays
with HIP events; variants are interleaved over `--rounds` rounds.  Prints per variant the per-round ms / step and the minimum.
    python probes/ab_step.py [--config base|large] --variants "base:" "ln1:13=1" "ln4:13=4" [--rounds 3] [--steps 20]
GPU box only."""
import argparse, importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="base")
ap.add_argument("--variants", nargs="+", default=["default:"])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=4)
args = ap.parse_args()
conf = bench.CONFIGS[args.config]
mp = importlib.import_module("x2-vlm_amd.model_pretrain")
cfgs = importlib.import_module("x2-vlm_amd.configs")
graph = importlib.import_module("x2-vlm_amd.graph")
lib = importlib.import_module("x2-vlm_amd._lib").lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = cfgs.pretrain_config(tempfile.mkdtemp(), conf["size"], conf["res"])
if conf["frames"]:
    cfg.update(video_encoding="avgpool", frame_len=conf["frames"], add_frame_pos=True)
model = mp.XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True).to(dev).train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(0, conf["batch"], 30, conf["res"], frames=conf["frames"]).items()}
variants = []
for v in args.variants:
    name, _, kv = v.partition(":")
    env = {}
    knobs = {}
    for item in filter(None, kv.split(",")):
        k, val = item.split("=")
        if k.isdigit():
            knobs[int(k)] = int(val)
        else:
            env[k] = val.replace("+", ",")      # an environment switch read at step construction (X2_...; write a comma as +)
    variants.append((name, knobs, env))
res = {n: [] for n, _, _ in variants}
for rnd in range(args.rounds):
    for name, knobs, env in variants:
        for k, val in knobs.items():
            assert lib.x2_tune(k, val) == 0, lib.x2_last_error()
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        step = graph.SegmentedStep(model, batch, warmup=1)
        assert step.mode == "hipgraph-segments", step.error
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            step()
        b.record()
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / args.steps)
        del step
        for k in knobs:
            lib.x2_tune(k, 0)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        import gc
        gc.collect()
        torch.cuda.empty_cache()
print("config %s, %d steps per timing, ms per step by round:" % (args.config, args.steps))
for name, knobs, env in variants:
    r = res[name]
    print("  %-12s %-28s %s   min %.3f  mean %.3f" % (name, ",".join("%s=%s" % kv for kv in list(knobs.items()) + list(env.items())) or "-",
                                                       " ".join("%.3f" % x for x in r), min(r), sum(r) / len(r)))
