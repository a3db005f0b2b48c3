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
    def __init__(self, model, world_size, process_group=None):
        self.model, self.world, self.pg = model, world_size, process_group
        self.side = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.reduced = []          # (data_ptr, nbytes) of arenas already all-reduced this step
        self.pending = []
        engine.GRAD_READY_HOOK = self._on_arena
        engine.STAGE_CALLS.clear()

    def _all_reduce(self, flat):
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if dist.get_backend(self.pg) == "nccl" else dist.ReduceOp.SUM, group=self.pg)
        if dist.get_backend(self.pg) != "nccl":
            flat.div_(self.world)

    def _on_arena(self, flat, key, also_after=None):
        # a parameter set used by several forward calls receives several gradients that autograd sums
        # later: only single-use layers may be reduced early
        if engine.STAGE_CALLS.get(key, 0) != 1:
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
            flat.record_stream(self.side)
        self.reduced.append((flat.data_ptr(), flat.numel() * 4))
        self.pending.append(flat)

    def finish(self):
        """Call after backward: reduce the leftovers, then make the compute stream wait for the side stream."""
        rest = []
        for p in self.model.parameters():
            g = p.grad
            if g is None:
                continue
            a = g.data_ptr()
            if not any(lo <= a < lo + n for lo, n in self.reduced):
                rest.append(g)
        if rest:
            flat = torch.cat([g.reshape(-1) for g in rest])
            self._all_reduce(flat)
            torch._foreach_copy_(rest, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in rest]), rest)])
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.reduced, self.pending = [], []
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
            dist.init_process_group(backend="nccl" if on_gpu else "gloo", init_method="tcp://%s:%d" % (addr, port),
                                    world_size=world_size, rank=rank)
        self.world_size = world_size
        self.broadcast(model)
        self.ddp_model = _Wrapped(model)
        self.buckets = GradientBuckets(model, world_size)
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
        if self.buckets is not None and self.world_size > 1:
            self.buckets.finish()

    def optimizer_step(self, optimizer, model, grad_norm):
        """Clip to `grad_norm`, return the total norm as a float (one host sync, like the reference's float(...)).
        With optim.FusedAdamW the norm is one multi-tensor launch and the clip coefficient is applied inside
        optimizer.step() instead of rewriting 1 GB of gradients."""
        if hasattr(optimizer, "grad_norm"):
            return float(optimizer.grad_norm(max_norm=grad_norm)[0])
        params = [p for g in optimizer.param_groups for p in g["params"]] if optimizer is not None else list(model.parameters())
        return float(torch.nn.utils.clip_grad_norm_(params, grad_norm))


ApexDDPAccelerator = RocmDDPAccelerator   # name Pretrain.py:30 imports
