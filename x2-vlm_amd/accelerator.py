"""Data-parallel boundary with the reference's Accelerator surface
(accelerators/accelerator.py:15-32, accelerators/apex_ddp_accelerator.py:30-102), MI355X-first:

  * one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm);
  * one flat broadcast of parameters + buffers instead of ~600 per-tensor broadcasts;
  * gradients: each layer's backward writes all its gradients into one flat fp32 arena
    (engine.Grads) and publishes it as soon as its kernels are enqueued; the arena is all-reduced
    (average) in place on a side HIP stream while the remaining backward runs on the compute
    stream (apex ran with delay_allreduce=True: no overlap).  Whatever did not arrive through an
    arena (small heads, embeddings, parameters used by more than one call in the step) is reduced
    as one flat leftover bucket at the end;
  * bf16 MFMA operands with fp32 master weights need no loss scaling: backward_step is
    loss.backward() + the gradient reduction; optimizer_step clips the global norm.
Parameters without a gradient (bbox_head on image-only steps) are skipped, like apex DDP does.
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import engine


class GradientBuckets:
    """Gradient averaging for one model replica per rank.

    Early path: a layer's backward publishes its flat arena (engine.Grads) once all kernels writing it are enqueued; if
    the layer ran exactly once in this forward AND none of its parameters holds a gradient yet (so autograd will adopt the
    arena views as `.grad` instead of adding them into an older buffer), the arena is all-reduced in place on the side
    stream while the remaining backward runs.
    Late path (`finish`, called by backward_step after loss.backward()): every other gradient - small heads, embeddings,
    parameters used by several calls, and everything in a second backward_step of the same iteration (the reference calls
    backward_step twice in run_mixed_iter, Pretrain.py:197, 247: `.grad` then already exists and autograd accumulates into
    it) - is packed into one flat buffer by a multi-tensor copy, all-reduced as one message, and `.grad` is re-pointed at
    the buffer's views (no copy back).  Re-averaging a buffer that already holds an averaged part is exact: the averaged
    part is identical on all ranks, and the mean of identical values is that value."""

    def __init__(self, model, world_size, process_group=None, comm=None):
        self.model, self.world, self.pg = model, world_size, process_group
        self.comm = comm           # comm.X2Comm: collectives through the C ABI (x2_comm_*) instead of torch.distributed
        self.side = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.done = set()          # id(parameter) whose gradient arena was reduced early in this backward
        self.pending = []          # arenas in flight on the side stream (kept alive until finish)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.messages = 0          # collectives issued (tests / diagnostics)
        self.demoted = 0           # early-reduced parameters whose .grad turned out not to live in the reduced arena
        engine.GRAD_READY_HOOK = self._on_arena
        engine.STAGE_CALLS.clear()

    def _all_reduce(self, flat):
        if self.comm is not None:
            self.comm.allreduce_bucket(flat, average=True)       # on the current stream (the side stream in _on_arena)
            self.messages += 1
            return
        nccl = dist.get_backend(self.pg) == "nccl"
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.pg)
        if not nccl:
            flat.div_(self.world)
        self.messages += 1

    def _on_arena(self, flat, key, also_after=None, params=()):
        if engine.STAGE_CALLS.get(key, 0) != 1 or not params or any(p.grad is not None for p in params):
            return
        if self.side is None:
            self._all_reduce(flat)
        else:
            ev = torch.cuda.Event()
            ev.record()
            self.side.wait_event(ev)
            if also_after is not None:
                self.side.wait_event(also_after)      # the layer's weight-gradient GEMMs run on engine.SIDE
            with torch.cuda.stream(self.side):
                self._all_reduce(flat)
        self.done.update(id(p) for p in params)
        self.pending.append((flat, list(params)))

    def finish(self):
        """Call after backward: reduce the leftovers, then make the compute stream wait for the side stream."""
        # An arena was reduced early on the assumption that autograd adopts its views as .grad.  If it did not (a hook, a
        # second contribution from outside the stage, create_graph, a layout mismatch made AccumulateGrad clone or add),
        # .grad holds a LOCAL gradient somewhere else: those parameters go out with the leftovers instead.
        for flat, params in self.pending:
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            for p in params:
                g = p.grad
                if g is not None and not (lo <= g.data_ptr() and g.data_ptr() + g.numel() * g.element_size() <= hi):
                    if self.side is not None:
                        # the copy / accumulate that produced this .grad ran on the compute stream while the side stream
                        # was all-reducing its source in place: its content cannot be trusted
                        raise RuntimeError("GradientBuckets: the gradient of a %s parameter was reduced early in its layer arena, but "
                                           "autograd did not adopt the arena view as .grad (gradient hook? create_graph?); "
                                           "remove the hook or run with engine.GRAD_READY_HOOK = None" % (tuple(p.shape),))
                    self.done.discard(id(p))
                    self.demoted += 1
        rest = [p for p in self.params if p.grad is not None and id(p) not in self.done]
        if rest:
            grads = [p.grad for p in rest]
            sizes = [g.numel() for g in grads]
            flat = torch.empty(sum(sizes), device=grads[0].device, dtype=grads[0].dtype)
            views = [v.view_as(g) for v, g in zip(flat.split(sizes), grads)]
            torch._foreach_copy_(views, grads)
            self._all_reduce(flat)
            for p, v in zip(rest, views):
                p.grad = v
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.done.clear()
        self.pending = []
        engine.STAGE_CALLS.clear()

    def close(self):
        engine.GRAD_READY_HOOK = None


class _Wrapped(torch.nn.Module):
    """What set_up returns in place of apex's DistributedDataParallel: callable like the model,
    exposes `.module` (Pretrain.py:328 clamps model.module.temp)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class Accelerator:
    def __init__(self, cfg, logger):
        self.cfg, self.logger = cfg, logger

    def set_up(self, model):
        raise NotImplementedError("Set Up method not implement in Accelerator, please check! ")

    def broadcast(self):
        raise NotImplementedError("Broadcast method not implement in Accelerator, please check! ")

    def backward_step(self, loss):
        loss.backward()

    def optimizer_step(self, optimizer, model, grad_norm):
        return float(torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm))


class RocmDDPAccelerator(Accelerator):
    """Drop-in for ApexDDPAccelerator (same constructor config keys and method signatures)."""

    def __init__(self, cfg, logger):
        super().__init__(cfg, logger)
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        self.accelerator_rng_seed = get("RNG_SEED", 42)
        self.accelerator_syncbn = get("SYNCBN", False)
        self.accelerator_fp16_opt_level = get("FP16_OPT_LEVEL", "O1")
        self.accelerator_fp16_loss_scale = get("FP16_LOSS_SCALE", "dynamic")
        self.buckets = None

    def set_up(self, model, optimizer, lr_scheduler, local_rank, world_size, rank):
        random.seed(self.accelerator_rng_seed)
        np.random.seed(self.accelerator_rng_seed)
        torch.random.manual_seed(self.accelerator_rng_seed)
        on_gpu = torch.cuda.is_available()
        if on_gpu:
            torch.cuda.manual_seed_all(self.accelerator_rng_seed)
            torch.cuda.set_device(local_rank)
            model = model.cuda()
        if not dist.is_initialized():
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = int(os.environ.get("MASTER_PORT", 34171))
            # X2_DIST_BACKEND=gloo: several ranks sharing one GPU (tests on a 1-GPU box); RCCL refuses duplicate devices
            backend = os.environ.get("X2_DIST_BACKEND", "nccl" if on_gpu else "gloo")
            if backend == "nccl" and world_size > 1:
                # cap RCCL's channel kernels (one CU each, resident under the backward); graph.SegmentedStep makes every GEMM
                # tile plan leave the same number of CUs out (x2_tune key 12).  X2_RCCL_CHANNELS / NCCL_MAX_NCHANNELS override.
                os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("X2_RCCL_CHANNELS", "16"))
            dist.init_process_group(backend=backend, init_method="tcp://%s:%d" % (addr, port), world_size=world_size, rank=rank,
                                    **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
        self.world_size = world_size
        self.broadcast(model)
        self.ddp_model = _Wrapped(model)
        # a single rank has nothing to average; X2_DDP_SINGLE_RANK_COLLECTIVES=1 keeps the collectives in (RCCL smoke test)
        if world_size > 1 or os.environ.get("X2_DDP_SINGLE_RANK_COLLECTIVES", "0") == "1":
            comm = None
            if os.environ.get("X2_COMM", "torch") == "rccl" and on_gpu:
                # gradient buckets through the C-ABI communicator (include/x2vlm_hip.h x2_comm_*); the RCCL id travels
                # through the rendezvous store torch.distributed already opened
                from .comm import X2Comm
                comm = X2Comm.from_store(dist.distributed_c10d._get_default_store(), rank, world_size)
            self.buckets = GradientBuckets(model, world_size, comm=comm)
        if optimizer is not None and hasattr(optimizer, "register_step_post_hook"):
            # optimizers that update through p.data (HF AdamW, apex FusedAdam) do not move the version counters the bf16
            # weight copies are keyed on: drop the copies after every optimizer step
            optimizer.register_step_post_hook(lambda *_a, **_k: engine.BANK.invalidate())
        return self.ddp_model, optimizer, lr_scheduler

    def broadcast(self, model, src=0):
        """All parameters and buffers in one flat message per dtype (apex: one call per tensor)."""
        by_dtype = {}
        for v in model.state_dict().values():
            by_dtype.setdefault(v.dtype, []).append(v)
        for dt, ts in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src)
            for t, v in zip(ts, flat.split([t.numel() for t in ts])):
                t.copy_(v.view_as(t))

    def backward_step(self, loss, optimizer=None):
        loss.backward()
        if self.buckets is not None:
            self.buckets.finish()

    def segmented_step(self, model, static_batch, **kw):
        """The image-text iteration (Pretrain.run_image_iter between zero_grad and optimizer.step) as replayed hipGraph
        segments with this accelerator's collectives between them: graph.SegmentedStep bound to this rank / world size /
        communicator.  Replaces model(...) + backward_step(loss) for that iteration; gradients come back averaged."""
        from .graph import SegmentedStep
        module = getattr(model, "module", model)
        rank = dist.get_rank() if dist.is_initialized() else 0
        comm = self.buckets.comm if self.buckets is not None else None
        return SegmentedStep(module, static_batch, world=getattr(self, "world_size", 1), rank=rank, comm=comm, **kw)

    def mixed_step(self, model, parts, **kw):
        """Pretrain.run_mixed_iter (Pretrain.py:189-252) - image + region (+ video) sub-iterations of one optimizer step, gradients
        accumulated, ONE averaging after the last - as replayed hipGraph segments: graph.MixedStep bound to this rank / world size /
        communicator.  Replaces the forwards and the two backward_step calls of that iteration."""
        from .graph import MixedStep
        module = getattr(model, "module", model)
        rank = dist.get_rank() if dist.is_initialized() else 0
        comm = self.buckets.comm if self.buckets is not None else None
        return MixedStep(module, parts, world=getattr(self, "world_size", 1), rank=rank, comm=comm, **kw)

    def optimizer_step(self, optimizer, model, grad_norm):
        """Clip to `grad_norm`, return the total norm as a float (one host sync, like the reference's float(...)).
        With optim.FusedAdamW the norm is one multi-tensor launch and the clip coefficient is applied inside
        optimizer.step() instead of rewriting 1 GB of gradients."""
        if hasattr(optimizer, "grad_norm"):
            return float(optimizer.grad_norm(max_norm=grad_norm)[0])
        params = [p for g in optimizer.param_groups for p in g["params"]] if optimizer is not None else list(model.parameters())
        return float(torch.nn.utils.clip_grad_norm_(params, grad_norm))


ApexDDPAccelerator = RocmDDPAccelerator   # name Pretrain.py:30 imports
