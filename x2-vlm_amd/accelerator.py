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


class _PublishGrads(torch.autograd.Function):
    """Joins the losses of a step whose backward has ALREADY run to autograd: forward hands out the loss values, backward - reached by the
    caller's accelerator.backward_step(sum of the losses) - publishes the gradients (those of the plain sum of the returned losses,
    Pretrain.py:67-68 / 98-100) into .grad instead of computing anything."""

    @staticmethod
    def forward(ctx, anchor, owner, pending, values):
        ctx.owner, ctx.pending = owner, pending
        return tuple(values.unbind(0))

    @staticmethod
    def backward(ctx, *cots):
        ctx.owner._publish(ctx.pending, cots)
        return None, None, None, None


class _Wrapped(torch.nn.Module):
    """What set_up returns in place of apex's DistributedDataParallel: callable like the model, exposes `.module`
    (Pretrain.py:328 clamps model.module.temp).

    Auto-capture (X2_AUTO_CAPTURE=0: the plain module call).  Pretrain.run_image_iter / run_region_iter do, every iteration and with the same shapes,
        loss = model(image, text_ids, ...);  optimizer.zero_grad();  accelerator.backward_step(sum of the losses, optimizer)
    (Pretrain.py:54-76, 79-107).  Launched eagerly that is ~1300 ctypes calls from Python per step (host-bound, ~29 ms for the base step); replayed as
    graph.SegmentedStep segments it is ~1 ms of host time.  A training call (grad enabled, module.training, an image and the MLM inputs, device
    tensors) therefore runs forward AND backward before it returns:
      * the first call with a shape signature: eagerly (the module, then the backward of the plain sum of its losses) - a shape seen once, like the short
        last batch of an epoch, is not worth a capture;
      * the second call builds graph.SegmentedStep for the signature (static copies of the inputs), later calls copy their inputs in and replay it;
    and returns the loss values tied to _PublishGrads.  The gradients (averaged over the ranks already) are kept out of reach of the caller's
    optimizer.zero_grad() and appear in .grad when backward_step backpropagates the losses.  No autograd graph of the model survives the call, which
    the captures need: an AccumulateGrad node kept alive by a previous iteration's loss tensors stays bound to the stream it was created on.
    Everything else - no-grad / eval calls, text-only calls (image=None), X2_AUTO_CAPTURE=0 - is the plain module call.
    Limit: the gradients are those of the PLAIN SUM of the returned losses.  A backward with unequal cotangents (a weighted sum, Pretrain.run_mixed_iter's
    iter_perc) recomputes forward and backward eagerly with the caller's weights - correct, slow, warned about once; accelerator.mixed_step is the
    fast form of that iteration."""

    def __init__(self, module, accelerator=None):
        super().__init__()
        self.module = module
        self._acc = accelerator
        self._seen, self._steps = {}, {}
        self._warned = False
        self.auto_capture = os.environ.get("X2_AUTO_CAPTURE", "1") == "1"
        self.last_mode = "eager"

    def forward(self, image=None, text_ids=None, text_atts=None, text_ids_masked=None, masked_pos=None, masked_ids=None, image_atts=None,
                idx_to_group_img=None, target_bbox=None, is_image=None, ret_bbox_loss=False, ret_match_loss=True, **extra):
        given = dict(image=image, text_ids=text_ids, text_atts=text_atts, text_ids_masked=text_ids_masked, masked_pos=masked_pos,
                     masked_ids=masked_ids, image_atts=image_atts, idx_to_group_img=idx_to_group_img, target_bbox=target_bbox, is_image=is_image)
        tensors = {k: v for k, v in given.items() if v is not None}
        call = lambda: self.module(image, text_ids, text_atts, text_ids_masked=text_ids_masked, masked_pos=masked_pos, masked_ids=masked_ids,
                                   image_atts=image_atts, idx_to_group_img=idx_to_group_img, target_bbox=target_bbox, is_image=is_image,
                                   ret_bbox_loss=ret_bbox_loss, ret_match_loss=ret_match_loss, **extra)
        ok = (self.auto_capture and not extra and torch.is_grad_enabled() and self.module.training and image is not None
              and text_ids_masked is not None and all(torch.is_tensor(v) and v.is_cuda for v in tensors.values()))
        if not ok:
            self.last_mode = "eager"
            return call()
        sig = (tuple((k, tuple(v.shape), v.dtype) for k, v in tensors.items()), bool(ret_bbox_loss), bool(ret_match_loss))
        step = self._steps.get(sig)
        params = [p for p in self.module.parameters() if p.requires_grad]
        before = [p.grad for p in params]                  # gradients of an earlier forward of the same iteration (accumulation): put back below
        if step is None:
            self._seen[sig] = self._seen.get(sig, 0) + 1
        if step is None and self._seen[sig] >= 2:
            static = {k: v.clone() for k, v in tensors.items()}
            kw = dict(ret_bbox_loss=bool(ret_bbox_loss), ret_match_loss=bool(ret_match_loss))
            if self._acc is not None:
                step = self._acc.segmented_step(self.module, static, **kw)
            else:
                from .graph import SegmentedStep
                step = SegmentedStep(self.module, static, **kw)
            step._busy = False
            self._steps[sig] = step
        elif step is not None and not step._busy:
            step.copy_inputs(step.batch, tensors)
        if step is not None and not step._busy:
            losses = step()
            step._busy = True                               # its static gradient tensors are spoken for until the caller's backward
            self.last_mode = step.mode
        else:
            # first sight of the signature (or a second forward of a step whose gradients are still unpublished): the module eagerly, backward at once
            for p in params:
                p.grad = None
            losses = call()
            sum(losses.values()).backward()
            if self._acc is not None and self._acc.buckets is not None:
                self._acc.buckets.finish()
            step = None
            self.last_mode = "eager-fused"
        dev = tensors["image"].device
        keys = list(losses)
        values = torch.stack([losses[k].detach().reshape(()).to(dev) for k in keys])     # own storage: a step's loss tensors are rewritten by its next replay
        del losses
        held = [(p, p.grad) for p in params if p.grad is not None]
        for p, g in zip(params, before):                    # out of reach of the caller's zero_grad (in place or to None) until its backward_step
            p.grad = g
        anchor = torch.zeros((), device=dev, requires_grad=True)
        outs = _PublishGrads.apply(anchor, self, dict(held=held, call=call, step=step), values)
        return dict(zip(keys, outs))

    def _publish(self, pending, cots):
        held, call, step = pending["held"], pending["call"], pending["step"]
        if held is None:
            raise RuntimeError("x2-vlm_amd: second backward through the losses of one model call (its gradients were published already)")
        pending["held"] = None
        if step is not None:
            step._busy = False
        live = [c for c in cots if c is not None]
        same = len(live) == len(cots) and all(c.data_ptr() == live[0].data_ptr() or bool(torch.equal(c, live[0])) for c in live[1:])
        if same and os.environ.get("X2_CHECK_COTANGENT", "0") == "1":
            same = float(live[0]) == 1.0        # host sync: off by default (Pretrain.py backpropagates the plain sum)
        if same:
            for p, g in held:
                p.grad = g if p.grad is None or p.grad is g else p.grad + g      # a second forward of the iteration accumulates (Pretrain.py:197, 247)
            return
        # a weighted sum: the gradients computed with the forward do not apply - one eager forward + backward with the caller's weights
        if not self._warned:
            print("x2-vlm_amd: backward with unequal loss weights through an auto-captured model call - recomputing eagerly (slow). "
                  "accelerator.mixed_step / segmented_step(total_loss=...) are the fast forms of a weighted iteration.", flush=True)
            self._warned = True
        with torch.enable_grad():
            losses = call()
            keys = list(losses)
            torch.autograd.backward([losses[k] for k in keys], [c if c is not None else torch.zeros_like(losses[k]) for k, c in zip(keys, cots)])


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
        self.ddp_model = _Wrapped(model, self)
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
        wrapped = getattr(self, "ddp_model", None)
        if wrapped is not None and wrapped.last_mode != "eager":
            return                               # a fused model call (_Wrapped): forward, backward and the gradient averaging ran inside it
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
