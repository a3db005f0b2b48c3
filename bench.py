#!/usr/bin/env python3
"""X^2-VLM pre-training step throughput on MI355X: image-text pairs/s (fwd+bwd), X2VLM-base, 224 px,
per-GPU batch 64, 30-token captions, ITC+ITM+MLM losses (BASELINE.json metric / configs[1]).

  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run (one rank per GPU, RCCL): gradients are all-reduced in
buckets overlapped with backward, the ITC features are all-gathered; weak scaling (64 pairs/GPU).
Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the bf16 MFMA GEMM) against
the 2.5 PFLOP/s dense bf16 peak from HIP-event timings of its launches; `cpu_baseline` times the
CPU oracle (oracle/, the validated restatement of the reference) on this box's host cores.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)

# BASELINE.json configs[1] / [3] / [4] on ONE GPU.  f_min = fwd+bwd GFLOP per pair (clip) actually executed, cross-attention
# K/V projected once per image (SURVEY.md 8(d), L = 30).  The default (`base`) is the configuration the metric is quoted on.
CONFIGS = {
    "base": dict(size="base", res=224, batch=64, frames=0, f_min=183.5, unit="pairs/s",
                 metric="image-text pairs/sec (fwd+bwd) X2VLM-base 224px bs=64/GPU",
                 workload="X2VLM-base (BEiT2-base + BERT-base 12+6) pre-training step fwd+bwd, ITC+ITM+MLM, 224px"),
    "large": dict(size="large", res=384, batch=32, frames=0, f_min=1315.7, unit="pairs/s",
                  metric="image-text pairs/sec (fwd+bwd) X2VLM-large 384px bs=32/GPU",
                  workload="X2VLM-large (BEiT2-large 24 blocks + BERT-large-12l 12+6, 593M) pre-training step fwd+bwd, "
                           "ITC+ITM+MLM, 384px"),
    # the region / bbox iteration (Pretrain.run_region_iter, Pretrain.py:79-111) at configs/pretrain/x2vlm_base_4m.yaml's
    # regions block: up to 26 images and 128 (region text, box) rows per GPU; masked mean pooling over 197 tokens, 5th fusion
    # pass (predict_bbox), L1 + GIoU on top of ITC + ITM + MLM.  f_min: measured - the GEMM FLOPs the step launches (bench.py
    # counts 2MNK per GEMM launch; attention and row kernels excluded, < 4 %), see `f_min_note` in the line.
    "region": dict(size="base", res=224, batch=128, images=26, frames=0, f_min=None, unit="pairs/s",
                   metric="region-text pairs/sec (fwd+bwd) X2VLM-base 224px, 128 region texts over 26 images/GPU",
                   workload="X2VLM-base region iteration (run_region_iter: 26 images, 128 region texts, masked mean pooling, "
                            "predict_bbox pass) fwd+bwd, ITC+ITM+MLM+bbox(L1+GIoU), 224px"),
    # Pretrain.run_mixed_iter (Pretrain.py:189-252), the iteration the default pre-training yaml actually runs: one image batch
    # AND one region batch per optimizer step, their gradients accumulated, one averaging.  Replayed as graph.MixedStep.
    "mixed": dict(size="base", res=224, batch=64, region_batch=128, images=26, frames=0, f_min=None, unit="pairs/s",
                  metric="(image-text + region-text) pairs/sec (fwd+bwd) X2VLM-base 224px mixed iteration: 64 image pairs + 128 region "
                         "texts over 26 images per GPU",
                  workload="X2VLM-base mixed iteration (run_mixed_iter: image batch 64 + region batch 26 images / 128 texts, gradients "
                           "accumulated, one reduction) fwd+bwd, ITC+ITM+MLM (+bbox L1+GIoU on the region part), 224px"),
    "video": dict(size="base", res=224, batch=8, frames=8, f_min=921.1, unit="clips/s",
                  metric="video-text clips/sec (fwd+bwd) X2VLM-base 8x8-frame 224px clips/GPU",
                  workload="X2VLM-base video path (avgpool over 8 frames + frame position embedding) pre-training step "
                           "fwd+bwd, ITC+ITM+MLM, 224px"),
}


def _pmc_traffic():
    """HBM traffic of the dominant kernel, measured offline with PMC counters (cannot be collected from inside the
    process): profiles/pmc_traffic.json is written from the rocprofv3 --pmc summaries committed next to it."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


PMC_TRAFFIC = _pmc_traffic()


def synthetic_batch(rank, B, L, res, vocab=30522, masks=12, frames=0):
    """per-rank generator seed 1234+rank."""
    g = torch.Generator().manual_seed(1234 + rank)
    image = torch.randn(*((B, frames) if frames else (B,)), 3, res, res, generator=g)
    ids = torch.randint(1000, 30000, (B, L), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    atts = torch.ones(B, L, dtype=torch.long)
    pos = torch.stack([torch.sort(torch.randperm(L - 2, generator=g)[:masks] + 1).values for _ in range(B)])
    return dict(image=image, text_ids=ids, text_atts=atts, masked_pos=pos, masked_ids=torch.gather(ids, 1, pos),
                text_ids_masked=ids.scatter(1, pos, 103))


def cpu_baseline(conf, seconds=20.0):
    """The oracle (CPU fp32 restatement, pinned to the reference's golden vectors) on the host cores:
    a B=4 (base) / B=2 (large, video) sample of the same workload."""
    from oracle import x2vlm_oracle as O
    synthetic = importlib.import_module("x2-vlm_amd.synthetic")
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    if conf["size"] == "large":
        cfg = O.OracleConfig(image_res=conf["res"], vision_width=1024, vision_heads=16, vision_layers=24, hidden=1024, heads=16, ffn=4096)
    else:
        cfg = O.OracleConfig(image_res=conf["res"], frames=conf["frames"])
    sd = O.make_params(cfg, 0, synthetic.synth_tensor)
    mixed = "region_batch" in conf
    B = 4 if conf is CONFIGS["base"] else 8 if "images" in conf else 2
    kw = {}
    if "images" in conf:        # 8 region texts over 2 images: the same ~4.9 texts per image as the benchmark batch
        b = synthetic.synth_region_batch(1234, 2, B, 30, conf["res"], 16, 30522, 12)
        kw = dict(ret_bbox_loss=True)
    else:
        b = synthetic_batch(0, B, 30, conf["res"], frames=conf["frames"])
    neg = synthetic.synth_negatives(0, B)
    bi, negi = (synthetic_batch(0, 4, 30, conf["res"]), synthetic.synth_negatives(0, 4)) if mixed else (None, None)
    n, t0, first = 0, time.time(), None
    while True:
        for t in sd.values():
            t.grad = None
        ts = time.time()
        losses, _ = O.xvlm_forward(sd, cfg, b, neg, **kw)
        total = sum(losses.values())
        if mixed:               # + the image part of the mixed iteration (4 pairs), gradients accumulated by one backward
            li, _ = O.xvlm_forward(sd, cfg, bi, negi)
            total = total + sum(li.values())
        total.backward()
        if first is None:
            first = time.time() - ts          # warm-up iteration, not counted
            t0 = time.time()
            continue
        n += 1
        if time.time() - t0 > seconds or n >= 8:
            break
    dt = time.time() - t0
    units = B + (4 if mixed else 0)
    return {"value": round(units * n / dt, 3), "unit": conf["unit"], "cores": threads, "kind": "port",
            "sample": "oracle fp32, %s, %s, %d timed steps after 1 warm-up (%.1fs), %d torch threads"
                      % (conf["workload"].split(" pre-training")[0], "4 image pairs + 8 region texts over 2 images" if mixed else "B=%d" % B,
                         n, dt, threads)}


def other_configs(args):
    """BASELINE.json configs[3] (X2VLM-large 384 px, batch 32) and configs[4] (8 x 8-frame video clips) on this GPU, plus the region
    / bbox iteration of the default pre-training yaml, each by a child run of this script (fresh process: its own allocator
    pools and graphs) - same timing contract, own cpu_baseline."""
    import subprocess
    res = {}
    for name, steps in (("large", 10), ("video", 15), ("region", 15), ("mixed", 10)):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", "3", "--graph", args.graph]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.no_graph:
            cmd.append("--no-graph")
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
            d = json.loads(line)
            res[name] = {k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_spread",
                                                "host_enqueue_ms_per_step", "launch_mode", "dtype", "data", "cpu_baseline")}
            res[name]["config"] = {k: d["config"][k] for k in ("workload", "per_gpu_batch", "mode")}
            res[name]["whole_step_tflops"] = d["roofline"]["also"]["whole_step_tflops"]
            res[name]["whole_step_frac"] = d["roofline"]["also"]["whole_step_frac"]
            res[name]["gemm_nt_isolated"] = {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "launches_per_step")}
        except Exception as e:      # noqa: BLE001 - the headline line must still be printed
            res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return res


def with_collectives(args):
    """The base step at ONE rank with the collectives of the N > 1 step issued for real (X2_DDP_SINGLE_RANK_COLLECTIVES=1: ITC all-gathers
    and the per-segment gradient all-reduces over a one-rank RCCL group), by a child run of this script."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--graph", args.graph,
           "--no-cpu-baseline", "--no-other-configs"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, X2_DDP_SINGLE_RANK_COLLECTIVES="1", X2_BENCH_UNPATCHED="0", X2_GRAPH_CANARY="0"))
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1])
        return {"host_enqueue_ms_per_step": d["host_enqueue_ms_per_step"], "ms_per_step": d["ms_per_step"], "launch_mode": d["launch_mode"],
                "collectives_per_step": d.get("collectives_per_step"),
                "what": "the same step at one rank with every collective of the N > 1 step issued (one-rank RCCL group, "
                        "X2_DDP_SINGLE_RANK_COLLECTIVES=1): host time to enqueue a step incl. the RCCL calls between the segments"}
    except Exception as e:      # noqa: BLE001 - the headline line must still be printed
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)          # >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="base",
                    help="base = the headline metric (BASELINE.json configs[1]); large / video = configs[3] / [4] on one GPU")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the configuration's own)")
    ap.add_argument("--seq-len", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, --config base only: do not append the large / video configurations (BASELINE.json configs[3] / [4] on "
                         "one GPU, each timed by its own child run of this script) to the JSON line")
    ap.add_argument("--eval-mode", action="store_true", help="model.eval(): dropout / DropPath off (not the headline number)")
    ap.add_argument("--tiny", action="store_true", help="debug: 2-layer towers (NOT the benchmark configuration)")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="also run clip + fused AdamW inside every step (the headline metric is fwd+bwd only)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--graph", choices=["segments", "whole"], default=os.environ.get("X2_BENCH_GRAPH", "segments"),
                    help="segments (default, any N): linear hipGraph segments joined by events, collectives between them "
                         "(graph.SegmentedStep); whole (N = 1 only): the step as one multi-stream hipGraph (graph.GraphedStep)")
    ap.add_argument("--serialize", action="store_true",
                    help="profiling aid: one HIP stream only (no concurrent text tower / weight-gradient stream), so that "
                         "per-kernel durations are not inflated by co-running kernels")
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    args.batch = args.batch or conf["batch"]

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    # X2_BENCH_BACKEND=gloo lets the N>1 code path (bucketed gradient reduction, ITC all-gather) be exercised on a
    # 1-GPU box: all ranks share GPU 0 and exchange through gloo.  The driver's multi-GPU runs use RCCL ("nccl").
    backend = os.environ.get("X2_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            # RCCL's channel kernels are resident on one CU each for most of the backward (the all-reduces run under it): cap
            # them, and let every GEMM tile plan leave that many CUs out (graph.SegmentedStep -> x2_tune(12, n)) - a plan
            # that fills "one round of 256 CUs" becomes two rounds when a few are taken.  1 GB per step under an 11 ms
            # backward needs ~100 GB/s: well within 16 channels over 7 xGMI links.  X2_RCCL_CHANNELS / NCCL_MAX_NCHANNELS override.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("X2_RCCL_CHANNELS", "16"))
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    elif os.environ.get("X2_DDP_SINGLE_RANK_COLLECTIVES", "0") == "1":
        # one rank, but every collective of the N > 1 step issued for real (a one-rank RCCL group: AVG / all-gather over one rank are the
        # identity): the only multi-GPU cost a 1-GPU box can measure is the host time of those ~40 calls between the segments
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("X2_RCCL_CHANNELS", "16"))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    K = importlib.import_module("x2-vlm_amd.kernels")
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    cfgs = importlib.import_module("x2-vlm_amd.configs")
    acc = importlib.import_module("x2-vlm_amd.accelerator")

    torch.manual_seed(0)
    if os.environ.get("X2_FAULT_DUMP"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["X2_FAULT_DUMP"]), exit=True)
    cfg = cfgs.pretrain_config(tempfile.mkdtemp(), conf["size"], conf["res"])
    if conf["frames"]:
        cfg.update(video_encoding="avgpool", frame_len=conf["frames"], add_frame_pos=True)
    if args.tiny:
        cfg.update(vision_num_hidden_layers=2, text_num_hidden_layers=3, text_fusion_start_at=2)
    model = mp.XVLM(config=cfg, load_vision_params=False, load_text_params=False, pretraining=True).to(dev)
    model.train(not args.eval_mode)
    if args.serialize:
        model.overlap_towers = False
        importlib.import_module("x2-vlm_amd.engine").SIDE.enabled = False
    ddp = acc.GradientBuckets(model, world) if world > 1 else None
    mixed = "region_batch" in conf
    region = "images" in conf and not mixed
    if region:
        synth = importlib.import_module("x2-vlm_amd.synthetic")
        batch = {k: v.to(dev) for k, v in synth.synth_region_batch(1234 + rank, conf["images"], args.batch, args.seq_len, conf["res"], 16,
                                                                   30522, 12).items()}
    else:
        batch = {k: v.to(dev) for k, v in synthetic_batch(rank, args.batch, args.seq_len, conf["res"], frames=conf["frames"]).items()}
    region_kw = dict(image_atts=batch["image_atts"], idx_to_group_img=batch["idx_to_group_img"], target_bbox=batch["target_bbox"],
                     is_image=batch["is_image"], ret_bbox_loss=True) if region else {}
    rbatch, units = None, args.batch                      # units of the metric one rank processes per step
    if mixed:
        synth = importlib.import_module("x2-vlm_amd.synthetic")
        rbatch = {k: v.to(dev) for k, v in synth.synth_region_batch(4321 + rank, conf["images"], conf["region_batch"], args.seq_len, conf["res"],
                                                                    16, 30522, 12).items()}
        units = args.batch + conf["region_batch"]

    eng = importlib.import_module("x2-vlm_amd.engine")

    opt = None
    if args.with_optimizer:
        opt = importlib.import_module("x2-vlm_amd.optim").create_optimizer(dict(lr=1e-4, weight_decay=0.01, lr_mult=2), model)

    params = list(model.parameters())

    def fwd_bwd():
        """One pass of the hot path: what Pretrain.run_image_iter does between zero_grad and optimizer.step."""
        eng.BANK.invalidate()        # as after an optimizer step: fp32 master weights are re-cast to bf16 inside the step
        with torch.no_grad():
            model.temp.clamp_(0.001, 0.5)                # Pretrain.py:327-328
        for p_ in params:                                # = model.zero_grad(set_to_none=True) without the module walk
            p_.grad = None
        loss = model(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], **region_kw)
        total = sum(loss.values())              # Pretrain.py:67-68 / 98-100: the plain sum of the returned losses
        if mixed:                               # Pretrain.py:206-251: the region forward's losses join the same backward_step
            rl = model(rbatch["image"], rbatch["text_ids"], rbatch["text_atts"], text_ids_masked=rbatch["text_ids_masked"],
                       masked_pos=rbatch["masked_pos"], masked_ids=rbatch["masked_ids"], image_atts=rbatch["image_atts"],
                       idx_to_group_img=rbatch["idx_to_group_img"], target_bbox=rbatch["target_bbox"], is_image=rbatch["is_image"],
                       ret_bbox_loss=True)
            total = total + sum(rl.values())
            loss = dict(loss, **{"region_" + k: v for k, v in rl.items()})
        total.backward()
        if ddp is not None:
            ddp.finish()
        return loss

    def optimizer_part():
        if opt is not None:
            opt.grad_norm(max_norm=1.0)                   # CLIP_GRAD_NORM 1.0, no host sync
            opt.step()

    def eager_step():
        loss = fwd_bwd()
        optimizer_part()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_gemms(n_steps):
        """HIP events around every GEMM launch of n eager steps (rank 0 times; every rank runs them: the step contains
        collectives), on the stream each launch goes to.  -> {kernel: (ms, flops, launches)}"""
        if rank == 0:
            K.GEMM_TIMER = []
        for _ in range(n_steps):
            eager_step()
        torch.cuda.synchronize()
        stat = {}
        if rank == 0:
            recs, K.GEMM_TIMER = K.GEMM_TIMER, None
            for a, b, f, name in recs:
                ms, fl, n = stat.get(name, (0.0, 0.0, 0))
                stat[name] = (ms + a.elapsed_time(b), fl + f, n + 1)
        return stat

    # ---- untimed: eager warm-up, then the dominant kernel's launch durations from eager steps.  Dominant kernel = the bf16
    # MFMA NT GEMM (7.56 of the 11.74 TFLOP of a base step).  `isolated`: one HIP stream only, so a launch's events see that
    # kernel alone - the number a rocprofv3 kernel trace of `bench.py --serialize` reproduces, and the headline roofline
    # figure; `concurrent`: the shipping three-stream schedule, where a launch also waits for CUs held by other streams.
    for _ in range(max(args.warmup, 1)):
        eager_step()
    fence()
    stat_conc = timed_gemms(2) if not args.serialize else {}
    model.overlap_towers = False
    eng.SIDE.enabled = False
    eager_step()
    stat_iso = timed_gemms(2)
    if not args.serialize:
        # A/B switches for the stream schedule of the timed steps (default: both on)
        model.overlap_towers = os.environ.get("X2_OVERLAP_TOWERS", "1") == "1"
        eng.SIDE.enabled = os.environ.get("X2_SIDE_STREAM", "1") == "1"
    fence()

    # ---- the step as ONE hipGraph (x2-vlm_amd/graph.py): the launch sequence is static, replaying it removes the
    # ~25 ms (base) / ~100 ms (large) of Python + ctypes launch time per step that bounded round 1.  The optimizer stays
    # outside the graph.  X2_GRAPH=0 / --no-graph: eager launches.
    graph = importlib.import_module("x2-vlm_amd.graph")
    fresh = None
    use_graph = not args.no_graph and (world == 1 or args.graph == "segments")
    if use_graph and os.environ.get("X2_GRAPH_CANARY", "1") == "1" and not args.tiny and world == 1:
        # Stream capture leans on ROCm behaviour found by probing (graph.py): a runtime that breaks it tends to crash
        # inside hipStreamEndCapture, which cannot be caught in-process.  A 2-layer copy of this step is captured in a child
        # process first (~10 s); if that does not come back with a hipgraph launch mode the benchmark launches eagerly.
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--tiny", "--config", args.config, "--batch", "4", "--steps", "1",
                                "--warmup", "1", "--no-cpu-baseline", "--graph", args.graph], capture_output=True, text=True, timeout=240,
                               env=dict(os.environ, X2_GRAPH_CANARY="0"))
            use_graph = r.returncode == 0 and '"launch_mode": "hipgraph' in r.stdout
        except Exception:      # noqa: BLE001
            use_graph = False
        if not use_graph and rank == 0:
            print("bench: hipGraph canary failed, launching eagerly", file=sys.stderr, flush=True)
    if args.graph == "segments" and not args.serialize:
        # Linear hipGraph segments (towers | features | tail + its backward | tower backwards) on two streams, the ITC
        # all-gather and the gradient all-reduces issued eagerly between / behind them: the same launch path for N = 1 and
        # N > 1, ~1 ms of host time per step.  Without capture (use_graph False) the same segments run eagerly.
        if ddp is not None:
            ddp.close()
            ddp = None
        if mixed:
            runner = graph.MixedStep(model, [dict(batch=batch), dict(batch=rbatch, ret_bbox_loss=True)], world=world, rank=rank, warmup=1,
                                     enabled=use_graph, verbose=(rank == 0))
        else:
            masking = None
            if not region and os.environ.get("X2_DEVICE_MASKING", "1") == "1":
                # the MLM masking of the reference's data-loader workers (dataset/pretrain_dataset.py:59-130, 242-275) as the first kernel of the text
                # segment: the step takes RAW captions and draws a new mask on every replay (X2_DEVICE_MASKING=0: the static host-made masks)
                synth = importlib.import_module("x2-vlm_amd.synthetic")
                masking = synth.masking_config({}, synth.synth_subword_flags(30522).to(dev), seed=1234 + rank)
                # ... and every timed step first receives a FRESH device batch (image, text_ids, text_atts) by GraphedStep.copy_inputs
                fresh = [{k: v.to(dev) for k, v in synthetic_batch(rank + 1000 * (i + 1), args.batch, args.seq_len, conf["res"], frames=conf["frames"]).items()
                          if k in ("image", "text_ids", "text_atts")} for i in range(3)]
            runner = graph.SegmentedStep(model, batch, world=world, rank=rank, warmup=1, enabled=use_graph, verbose=(rank == 0),
                                         ret_bbox_loss=region, masking=masking)
    else:
        # N = 1: the whole step as ONE multi-stream hipGraph (fork / join edges: ~15 us of host time per node).
        # N > 1 on this path: eager launches with bucketed all-reduces overlapped on a side stream (accelerator.GradientBuckets).
        fwd_bwd.parameters = lambda: params
        runner = graph.GraphedStep(fwd_bwd, warmup=1, enabled=use_graph and world == 1, verbose=(rank == 0))

    nstep = [0]

    def step():
        if fresh is not None:                   # new inputs into the tensors the segments were captured on (inside the timed region)
            graph.GraphedStep.copy_inputs(batch, fresh[nstep[0] % len(fresh)])
            nstep[0] += 1
        loss = runner()
        if isinstance(loss, list):              # MixedStep: one loss dict per part
            loss = dict(loss[0], **{"region_" + k: v for k, v in loss[1].items()})
        optimizer_part()
        return loss

    for _ in range(args.warmup):
        step()
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host_each = []
    for i in range(args.steps):
        th = time.perf_counter()
        loss = step()
        marks[i + 1].record()
        host_each.append(time.perf_counter() - th)
    # host time to ENQUEUE a step (no sync): the median of the first ten steps - once the runtime's queue of in-flight graph launches is full
    # (a few dozen steps ahead of the GPU) every further enqueue blocks for one GPU step, which is back-pressure, not launch cost
    first = sorted(host_each[:10])
    host_dt = first[len(first) // 2] * args.steps
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0 and getattr(runner, "times", None):
        print("segment times (ms): " + json.dumps({k: round(v, 3) for k, v in runner.segment_times().items()}), file=sys.stderr, flush=True)
    pairs_s = world * units * args.steps / dt
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))

    # ---- the same iteration through the reference's UNPATCHED call sequence (Pretrain.run_image_iter, Pretrain.py:54-76): model(...) ->
    # optimizer.zero_grad() -> accelerator.backward_step(sum of the losses) on what RocmDDPAccelerator.set_up returns (accelerator._Wrapped:
    # the second call captures the segments, later calls replay them; host-made MLM masks, as the reference's data loader delivers them)
    unpatched = None
    if world == 1 and args.graph == "segments" and use_graph and not (mixed or region or args.serialize or args.tiny) and os.environ.get("X2_BENCH_UNPATCHED", "1") == "1":
        try:
            wrapped = acc._Wrapped(model, None)
            a_ = acc.RocmDDPAccelerator(dict(), None)
            a_.ddp_model = wrapped
            full = [{k: v.to(dev) for k, v in synthetic_batch(rank + 2000 * (i + 1), args.batch, args.seq_len, conf["res"], frames=conf["frames"]).items()}
                    for i in range(3)]

            def run_image_iter(b):
                loss_ = wrapped(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                                masked_ids=b["masked_ids"], ret_match_loss=True)
                for p_ in params:                          # optimizer.zero_grad()
                    p_.grad = None
                a_.backward_step(loss_["loss_itc"] + loss_["loss_itm"] + loss_["loss_mlm"], opt)
                optimizer_part()
                return loss_
            for i in range(4):
                run_image_iter(full[i % 3])
            fence()
            t1 = time.perf_counter()
            for i in range(args.steps):
                run_image_iter(full[i % 3])
            fence()
            dt1 = time.perf_counter() - t1
            unpatched = {"value": round(units * args.steps / dt1, 1), "unit": "image-text pairs/s", "ms_per_step": round(1e3 * dt1 / args.steps, 2),
                         "launch_mode": wrapped.last_mode,
                         "what": "model(...) + optimizer.zero_grad() + accelerator.backward_step(loss_itc + loss_itm + loss_mlm) per step, the call sequence "
                                 "of Pretrain.run_image_iter, on the wrapper RocmDDPAccelerator.set_up returns (auto-captured segments); a fresh "
                                 "device batch with host-made masks every step"}
        except Exception as e:      # noqa: BLE001 - the headline line must still be printed
            unpatched = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    roof = None
    f_min, f_note = conf["f_min"], None
    if f_min is None:
        # executed GEMM FLOPs of one step (NT + TN launches of the instrumented eager steps; the MLM head's recomputed decoder
        # GEMM is not counted), per unit of the metric
        gf = sum(fl for name, (ms, fl, n) in stat_iso.items() if name in ("gemm_nt", "gemm_tn")) / 2 if rank == 0 else 0.0
        f_min = gf / 1e9 / units
        f_note = "measured: 2MNK of the GEMM launches of one step / units per step (attention and row kernels not counted)"
    if rank == 0:
        def summary(stat, name):
            ms, fl, n = stat.get(name, (0.0, 0.0, 0))
            return None if n == 0 else {"avg_launch_us": round(1e3 * ms / n, 1), "achieved": round(fl / ms / 1e9, 1),
                                        "frac": round(fl / ms / 1e9 / PEAK_TFLOPS, 4), "launches_per_step": n // 2,
                                        "ms_per_step": round(ms / 2, 2), "gflop_per_launch": round(fl / n / 1e9, 2)}
        iso = summary(stat_iso, "gemm_nt")
        roof = {"bound": "mfma", "achieved": iso["achieved"], "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": iso["frac"],
                "traffic": PMC_TRAFFIC.get("hbm_bytes_per_launch"),
                "traffic_note": ("HBM bytes per launch (fetch + write) of this kernel from separate rocprofv3 --pmc passes of "
                                 "`bench.py --serialize` (base config), corrected per MI355X_MICROARCH.md; summaries: "
                                 + ", ".join(PMC_TRAFFIC.get("files", []))) if PMC_TRAFFIC else "no PMC pass committed",
                "kernel": "gemm_nt_kernel (forward linears + input gradients)", "launches_per_step": iso["launches_per_step"],
                "avg_launch_us": iso["avg_launch_us"], "gflop_per_launch": iso["gflop_per_launch"],
                "how": "algorithmic FLOPs of the launches / HIP-event durations, one HIP stream (kernels not overlapped): what "
                       "`rocprofv3 --kernel-trace --stats -- python bench.py --serialize` shows for this kernel",
                "also": {"gemm_nt concurrent (three-stream eager schedule; a launch also waits for CUs other streams hold)":
                             summary(stat_conc, "gemm_nt"),
                         "gemm_tn256_kernel isolated (weight gradients; two layers per grouped launch)": summary(stat_iso, "gemm_tn"),
                         "whole_step_tflops": round(pairs_s / world * f_min / 1e3, 1),
                         "whole_step_frac": round(pairs_s / world * f_min / 1e3 / PEAK_TFLOPS, 4),
                         # all kernels of one base step, from the same committed PMC passes as `traffic` (an upper bound of HBM bytes: the
                         # fabric counters include Infinity-Cache hits), and the time that many bytes take at the HBM rates
                         "whole_step_hbm_gb": PMC_TRAFFIC.get("whole_step_hbm_gb") if args.config == "base" else None,
                         "whole_step_hbm_floor_ms": PMC_TRAFFIC.get("whole_step_hbm_floor_ms") if args.config == "base" else None,
                         "gflop_per_unit": round(f_min, 1), **({"f_min_note": f_note} if f_note else {})}}
    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {"metric": conf["metric"], "value": round(pairs_s, 1),
               "unit": conf["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 2),
               "ms_per_step_spread": {"min": round(per_step[0], 2), "median": round(per_step[len(per_step) // 2], 2),
                                      "max": round(per_step[-1], 2), "how": "HIP events between consecutive timed steps"},
               "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 2), "launch_mode": runner.mode,
               **({"collectives_per_step": runner.messages} if getattr(runner, "coll", False) else {}),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": conf["workload"] + ", %d-token captions, 12 masks" % args.seq_len, "name": args.config,
                          "per_gpu_batch": args.batch if not mixed else "%d image pairs + %d region texts" % (args.batch, conf["region_batch"]),
                          "global_batch": units * world,
                          "parallelism": "dp%d" % world, "streams": "single (--serialize)" if args.serialize else "concurrent",
                          "optimizer_in_step": bool(args.with_optimizer), "mode": "eval (dropout/DropPath off)" if args.eval_mode else
                          "train (BERT dropout 0.1, attention dropout 0.1, DropPath 0..0.1)",
                          "inputs": ("a fresh device batch (image, text_ids, text_atts) copied into the step's tensors inside every timed step; MLM masking "
                                     "on the device inside the text segment (x2_mask_tokens)") if fresh is not None else "static batch, host-made MLM masks",
                          "losses": {k: round(float(v), 4) for k, v in loss.items()}},
               "roofline": roof}
        # every kernel-variant knob that is not at its default (X2_TUNE / x2_tune) is part of the record: a line measured
        # with a non-default variant says so.  (Work-skipping ablation bits do not exist in the shipped library at all:
        # csrc/gemm.hip compiles them only with -DX2_PROBE, probes/build_probe.sh.)
        h_ = importlib.import_module("x2-vlm_amd._lib").lib()
        tuned = {str(k): h_.x2_tune_get(k) for k in range(16) if h_.x2_tune_get(k) > 0}
        if unpatched is not None:
            out["roofline"]["also"]["unpatched_pretrain_py"] = unpatched
        out["x2_tune_non_default"] = tuned
        out["env_switches"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("X2_") and k not in ("X2_BENCH_BACKEND",)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(conf)
        if world == 1 and args.config == "base" and not (args.no_other_configs or args.tiny or args.serialize or args.eval_mode):
            if args.graph == "segments" and use_graph:
                out["roofline"]["also"]["host_enqueue_ms_with_collectives"] = with_collectives(args)
            out["other_configs"] = other_configs(args)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
