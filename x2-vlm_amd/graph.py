"""hipGraph capture of a whole training step.

A step of this path is ~1300 (base) to ~2500 (large) kernel launches issued from Python through ctypes; at 15-25 us of
host time each the host, not the GPU, sets the step time (r01: host 29.2 ms of a 29.5 ms step).  The step's launch
sequence is static - same kernels, same shapes, same buffers every iteration - so it is captured ONCE into a hipGraph
(all streams: vision / text towers, weight-gradient side streams, the gradient all-reduce stream) and replayed with a
single host call per iteration.

What had to be true for that, and is:
  * no host round trip inside the step (hard negatives, CE normalisers, degenerate-box test are device-side);
  * dropout / DropPath randomness that does not live in kernel arguments: every dropout site mixes the device word
    `kernels.DROP_EPOCH` - incremented by the graph itself - into its seed (csrc/x2_common.h drop_at_epoch); DropPath
    keeps and the hard-negative uniforms come from torch's graph-safe Philox generator;
  * ROCm 7's stream capture only survives side streams that fork from and join into the capture's origin stream
    (probes/graph_capture_probe.py): the text tower keeps its own stream, the weight-gradient side stream is used by the
    origin stream's stages only (engine.SideStream.only_from);
  * static addresses: inputs are copied into the tensors the step was captured on (`GraphedStep.copy_inputs`), parameter
    gradients stay in the arenas allocated during capture (do not set .grad to None between replays);
  * the optimizer stays OUTSIDE the graph (two eager launches; its per-parameter step counters live on the host).

Usage (bench.py; INTEGRATION.md shows the same six lines inside Pretrain.run_image_iter):
    step = GraphedStep(lambda: fwd_bwd(model, static_batch))     # warm-up + capture
    for batch in loader:
        step.copy_inputs(static_batch, batch)
        losses = step()                                           # one hipGraphLaunch
        optimizer.step()
If capture is impossible (e.g. a process group whose collectives cannot be captured: gloo) the object degrades to eager
execution of `fn` and says so in `.mode`."""
import gc
import os

import torch

from . import kernels as K


class GraphedStep:
    def __init__(self, fn, warmup=2, enabled=True, verbose=False):
        """fn: the step (forward + backward).  If it carries an attribute `parameters` (callable -> iterable of
        Parameters), their captured .grad tensors are re-attached after every replay."""
        self.fn, self.graph, self.out = fn, None, None
        self._grads = []
        self.stream = torch.cuda.Stream()
        self.mode = "eager"
        self.error = None
        self._epoch = None                    # disabled: eager launches; the process-wide epoch word (if any step object made one) still advances
        if not enabled or os.environ.get("X2_GRAPH", "1") == "0":
            return
        from . import engine
        # The epoch word's ADDRESS is baked into every captured dropout kernel: it is allocated once per process and never
        # replaced (a second step object - INTEGRATION 1b offers both kinds - must not free the word a live graph reads);
        # every step object increments the one it captured with.
        if K.DROP_EPOCH is None:
            K.DROP_EPOCH = torch.zeros(1, dtype=torch.int32, device="cuda")
        self._epoch = K.DROP_EPOCH
        side_rule, engine.SIDE.only_from = engine.SIDE.only_from, self.stream.cuda_stream     # see engine.SideStream.only_from
        try:
            gc.collect()                      # autograd graphs of earlier eager steps (their AccumulateGrad nodes remember the
                                              # stream they were created on) must be gone before the capture stream's own
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for _ in range(warmup):       # eager, on the capture stream: caches, workspaces, AccumulateGrad streams
                    self._epoch.add_(1)
                    fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            trace = os.environ.get("X2_GRAPH_TRACE") == "1"
            engine.BANK.invalidate()          # the bf16 weight casts must be part of the captured step, whatever warm-up left cached
            self._grads = []
            try:
                if trace:
                    print("GraphedStep: warm-up done, capturing", flush=True)
                with torch.cuda.graph(g, stream=self.stream):
                    self._epoch.add_(1)
                    self.out = fn()
                    if trace:
                        print("GraphedStep: fn() captured, ending capture", flush=True)
                if trace:
                    print("GraphedStep: graph instantiated", flush=True)
                self.graph, self.mode = g, "hipgraph"
                # the graph writes gradients into the tensors that were .grad when capture ended: remember them, so that a
                # training loop that calls optimizer.zero_grad(set_to_none=True) between replays gets them back (__call__)
                params = getattr(fn, "parameters", None)
                self._grads = [(p_, p_.grad) for p_ in (params() if callable(params) else ()) if p_.grad is not None]
            except Exception as e:            # noqa: BLE001 - anything that cannot be captured: run eagerly instead
                # the epoch word stays (other step objects may have captured it); eager launches keep mixing it in, __call__
                # keeps incrementing it
                self.error = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
                if verbose:
                    print("GraphedStep: capture failed, running eagerly (%s)" % self.error, flush=True)
                torch.cuda.synchronize()
        finally:
            engine.SIDE.only_from = side_rule
        torch.cuda.current_stream().wait_stream(self.stream)

    def __call__(self):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self.graph is not None:
                self.graph.replay()
                for p_, g_ in self._grads:
                    if p_.grad is not g_:
                        p_.grad = g_
            else:
                ep = self._epoch if self._epoch is not None else K.DROP_EPOCH
                if ep is not None:
                    ep.add_(1)
                self.out = self.fn()
        cur.wait_stream(self.stream)
        return self.out

    @staticmethod
    def copy_inputs(static_batch, batch):
        """New data into the tensors the step was captured on (one multi-tensor copy)."""
        keys = [k for k in static_batch if k in batch]
        torch._foreach_copy_([static_batch[k] for k in keys], [batch[k] for k in keys])


class SegmentedStep:
    """The X^2-VLM pre-training step as a handful of LINEAR hipGraph segments joined by events outside capture.

    Why: ROCm replays a captured graph that contains fork / join edges node by node (~15 us each: 21 ms of host time per
    base step, 65 ms per large step - `GraphedStep` above), while a single-stream capture replays in well under a
    millisecond.  And a graph that spans the whole step leaves no place to issue a collective, so multi-GPU steps had to
    be launched eagerly (host-bound).  Here every segment is captured on ONE stream; what ran on different streams inside
    one graph now runs as different graphs on different streams:

        stream A (vision / tail)                           stream B (text)                       stream C (collectives)
        ----------------------------------------------------------------------------------------------------------------
        [V   vision tower forward]                         [T  text tower forward, 2B rows]
        [F1  temp clamp, ITC features]      <- event ----
         eager: all-gather of the (B, 256) features (rank > 1: RCCL; one rank: a copy)
        [F2  ITC / hard negatives / 4B-row fusion pass / ITM / MLM, backward down to the tower outputs]
                                             -- event ->                                         all-reduce(tail gradients)
        [Vb  vision tower backward]                        [Tb text tower backward]
                                                            -- event ->                          all-reduce(text gradients)
                                                           [Fw weight gradients of the tail's fusion layers (queued by F2)]
                                                            -- event ->                          all-reduce(fusion gradients)
         -- event ------------------------------------------------------------------------->     all-reduce(vision gradients)
        join B, C

    Autograd is cut at the tower outputs (detached leaves): F2 leaves the gradients of the tower outputs in the leaves'
    .grad, Vb / Tb feed them to the towers' own graphs.  The ITC all-gather sits between F1 and F2 as an eager call; its
    backward (the local slice of the gathered gradient, xvlm.AllGather) is part of F2.  The tied word-embedding / decoder
    gradient stays on stream B (engine.TIE_WORD_GRAD), and every parameter's AccumulateGrad node is pinned to the stream of
    the segment that produces its gradient, so no segment contains an edge into another stream.

    Gradient averaging (world > 1) follows apex's delay_allreduce order at segment granularity: each segment's gradients -
    whole per-layer arenas in place, large single tensors in place, the small remainder packed into one static buffer - are
    reduced on stream C as soon as the segment that completes them has been launched.  p.grad of every parameter is a
    static tensor after capture: do not set gradients to None between replays (`reattach_grads` restores them if the
    training loop did).

    model: model_pretrain.XVLM.  batch: dict of static device tensors (image, text_ids, text_atts, text_ids_masked,
    masked_pos, masked_ids [, image_atts, idx_to_group_img, target_bbox, is_image]); new data goes in with
    GraphedStep.copy_inputs.  Falls back to eager execution of the same segments when capture is disabled or fails."""

    _STREAMS = {}

    def __init__(self, model, batch, world=1, rank=0, process_group=None, comm=None, warmup=2, enabled=True, verbose=False,
                 ret_bbox_loss=False, ret_match_loss=True, recast_weights=True, clamp_temp=True, total_loss=None, side_stream=None,
                 vision_cuts=None, reduce_grads=True, defer_reduce=False, masking=None):
        """side_stream: weight-gradient GEMMs of the stream-A segments (tail, vision backward) on engine.SIDE, forked from and
        joined into stream A inside the segment.  Those segments then replay node by node (~15 us of host time each, still
        well under the GPU time of a step) but keep the 3 % the side stream is worth on one GPU.  Default: X2_SEG_SIDE or off.
        reduce_grads=False: no gradient averaging at all (the ITC all-gather stays) - a sub-iteration whose gradients a MixedStep
        adds to another step's before ONE reduction; defer_reduce=True: the reduction plan is built but issued by `reduce_all()`
        instead of behind the segments (the accumulating step of a MixedStep)."""
        from . import engine
        self.engine = engine
        self.model, self.batch, self.world, self.rank, self.pg, self.comm = model, batch, world, rank, process_group, comm
        # masking: None (the batch carries text_ids_masked / masked_pos / masked_ids, made by the data pipeline as in the reference) or a
        # dict for kernels.mask_tokens (is_subword uint8 [vocab] on the device, plus any of mask_prob, max_masks, skipgram_prb, skipgram_size,
        # mask_whole_word, cls_id, mask_id, seed): the text segment then starts with the MLM masking of the RAW (text_ids, text_atts) of the
        # batch (dataset/pretrain_dataset.py:59-130, 242-275 on the device) and draws a new mask on every replay (the epoch word)
        self.masking = dict(masking) if masking else None
        if self.masking:
            mm = int(self.masking.get("max_masks", 12))
            ids = batch["text_ids"]
            batch.setdefault("text_ids_masked", torch.empty_like(ids))
            batch.setdefault("masked_pos", torch.zeros(ids.shape[0], mm, dtype=torch.int64, device=ids.device))
            batch.setdefault("masked_ids", torch.full((ids.shape[0], mm), -100, dtype=torch.int64, device=ids.device))
        # collectives are issued when there is more than one rank - or when X2_DDP_SINGLE_RANK_COLLECTIVES=1 asks a single rank to
        # run them anyway (the RCCL call path of the replayed step on a 1-GPU box; AVG over one rank is the identity)
        self.coll = world > 1 or os.environ.get("X2_DDP_SINGLE_RANK_COLLECTIVES", "0") == "1"
        self.ret_bbox_loss, self.ret_match_loss = ret_bbox_loss, ret_match_loss
        self.recast_weights, self.clamp_temp = recast_weights, clamp_temp
        self.reduce_grads, self.defer_reduce = reduce_grads, defer_reduce
        self._home = {}                         # id(param) -> the static tensor the captured segments leave its gradient in
        self.side_stream = (os.environ.get("X2_SEG_SIDE", "0") == "1") if side_stream is None else bool(side_stream)
        # fusion-layer weight gradients as a segment of their own on stream B (engine.WGRAD_QUEUE); X2_SEG_TAIL_WGRAD=0: in line
        # Not with ret_bbox_loss: predict_bbox runs the fusion layers a SECOND time in the same pass, and a queued weight gradient
        # is only valid with one contribution per parameter - autograd adds the second call's (still empty) arena view to the
        # first's when it receives it, long before the queue fills either; the late TN GEMM then overwrites the sum with one
        # contribution (round 4: the replayed region iteration had lost the other one; the eager path was right).
        self.defer_tail_wgrad = os.environ.get("X2_SEG_TAIL_WGRAD", "1") == "1" and not ret_bbox_loss
        # the tail segment forks a second stream for what hangs off the fusion stack's dependency chain (engine.AUX).  ROCm replays
        # a segment that contains a fork node by node and keeps ONE such replay in flight: the host thread then sits inside
        # hipGraphLaunch for a good part of the step (base 5 ms, region 25 ms per step at 20 enqueued steps).  One GPU: the step stays
        # GPU-bound and keeps the -0.5 % (base) / -4 % (X2VLM-large).  More than one rank: the same thread issues the ~40
        # collectives between the segments, and a collective issued late is an all-reduce that no longer hides under the vision
        # backward - so the fork is OFF by default when collectives are issued (X2_AUX_OVERLAP=1 forces it on, =0 off).
        env_aux = os.environ.get("X2_AUX_OVERLAP")
        self.aux_overlap = (env_aux == "1") if env_aux is not None else not self.coll
        # left to itself the fork is also skipped where the cross-attention backward is one kernel (engine.AUX.auto): the base / video /
        # region configurations; X2VLM-large (577 image tokens: two kernels, the K/V half on the second stream) keeps it
        self.aux_auto = env_aux is None
        # bf16 payload for the gradient all-reduces (0.51 GB per step instead of 1.02 GB): the arena is cast to a static
        # bf16 buffer, averaged there, and written back to the fp32 arena on arrival (the AVERAGE itself is then formed in bf16: ~3 significant
        # digits of every gradient - a precision trade, not a transport detail).  Off by default (parity-tested at 2 and
        # 8 gloo ranks; never measured on xGMI): X2_GRAD_BF16=1
        self.grad_bf16 = os.environ.get("X2_GRAD_BF16", "0") == "1"
        self._bf16_buf = {}
        # compute units every GEMM tile plan leaves to RCCL's channel kernels (x2_tune key 12) when collectives run beside
        # the backward: X2_RESERVED_CUS, default 0 at one rank, the channel cap below otherwise.  NCCL_MAX_NCHANNELS is only a
        # request to RCCL made before the communicator exists (accelerator.set_up / bench.py export it from the same number).
        self.reserved_cus = int(os.environ.get("X2_RESERVED_CUS", "-1"))
        if self.reserved_cus < 0:
            self.reserved_cus = int(os.environ.get("NCCL_MAX_NCHANNELS", "0") or 0) if world > 1 else 0
        from ._lib import lib as _x2lib
        self._lib = _x2lib() if batch["text_ids"].is_cuda else None
        # (x2_tune(12, ...) is process-wide planning state: it is set around this step's own launches - warm-up, capture, eager runs - and put
        # back afterwards (_reserve / _unreserve), so that other steps and plain eager calls keep their own tile plans)
        self._queue = None
        # The vision tower as a chain of stages cut at these block numbers (beit2.VisionTransformer.chunk_at): its backward
        # becomes one segment per stage, top first; the weight gradients of every stage but the lowest run as segments of
        # their own on stream B, under the backward of the stages below, and their all-reduce (world > 1) starts there too.
        depth = model.vision_encoder.depth
        env = os.environ.get("X2_SEG_VISION_CUT")
        if vision_cuts is None:
            # default: thirds (same box, base: no cut 25.07 ms, one cut 24.88, two cuts 24.72 - profiles/r03f_ab_segments.txt)
            # (world > 1: quarters - the last stage's arenas are the only all-reduce that nothing can overlap)
            parts = 4 if world > 1 else 3
            vision_cuts = [int(c) for c in env.split(",") if c] if env is not None else [depth * i // parts for i in range(1, parts)]
            # Round 5, one rank: where the vision blocks' attention backward is ONE kernel (64 < N <= 208 tokens) stream A's backward
            # chain lost a fifth of its time (8.9 instead of 10.9 ms per base step) and ended 1.9 ms before stream B - which carries the
            # text backward, the fusion stack's weight gradients AND the deferred vision weight gradients.  One cut at 3/4 of the depth
            # (only the top quarter's weight gradients go to stream B) when the text side is heavy enough to be the longer stream: base
            # 21.96 vs 22.27 ms per step; the video step (8 captions beside 64 frames) and X2VLM-large (N = 577) keep thirds
            # (profiles/r11b_vision_cut_ab.txt)
            if env is None and world == 1 and "image" in batch:
                pe = getattr(model.vision_encoder, "patch_embed", None)
                ntok = (pe.num_patches + 1) if pe is not None else 0
                img = batch["image"]
                nimg = img.shape[0] * (img.shape[1] if img.dim() == 5 else 1)
                text_rows = 4 * batch["text_ids"].shape[0] * batch["text_ids"].shape[1]
                # Round 6 (profiles/r12h_vision_cut_ab.txt): the balance of the two streams is within 0.1 ms of a block's weight gradients - with the first
                # fixed-order embedding backward (270 us on stream B) a cut at block 10 of 12 won by 0.17 ms, with its second form (85 us) 3/4 wins
                # again by 0.08 ms: 3/4 stays.  X2VLM-large, whose attention backward is one kernel as well since the long one-pass form
                # (208 < N <= 640), is cut at 1/2 and 3/4 (69.04 vs 69.62 ms with thirds)
                if 64 < ntok <= 208 and 2 * text_rows >= nimg * ntok:
                    vision_cuts = [depth * 3 // 4]
                elif 208 < ntok <= 640 and img.dim() == 4:
                    vision_cuts = [depth // 2, depth * 3 // 4]
        self.vcuts = sorted(c for c in vision_cuts if 0 < c < depth)
        self.defer_vision_wgrad = os.environ.get("X2_SEG_VISION_WGRAD", "1") == "1"
        self.prefetch_casts = os.environ.get("X2_SEG_PREFETCH_CASTS", "1") == "1" and recast_weights
        self._vq = {}
        self.times = {} if os.environ.get("X2_SEG_TIMES") == "1" else None
        self._held = []
        self._rorder = []                       # segment names in the order their reductions are issued (recorded at capture)
        self.total_loss = total_loss or (lambda losses: sum(losses.values()))
        # ONE set of streams per process and device, shared by every SegmentedStep: a parameter's AccumulateGrad node is
        # unique, and `_pin_accumulators` can bind it to a stream only once - a second step object with streams of its own (the
        # region part of a MixedStep, or a separately built region step) had its in-place gradient accumulations (layers that
        # run twice in a pass) executed on the FIRST object's streams, outside its captures: a race that corrupted memory at
        # random (round 4: `bench.py --config mixed --tiny` died with GPU memory faults, image part first; region first was fine).
        # Step objects never run concurrently, so sharing costs nothing.
        dev_key = (batch["text_ids"].device.index, os.environ.get("X2_SEG_ONE_STREAM", "0") == "1")
        st = SegmentedStep._STREAMS.get(dev_key)
        if st is None:
            a_ = torch.cuda.Stream()
            st = SegmentedStep._STREAMS[dev_key] = dict(A=a_, B=a_ if dev_key[1] else torch.cuda.Stream(), C=None, G=None)
        if self.coll and st["C"] is None:
            st["C"], st["G"] = torch.cuda.Stream(), torch.cuda.Stream()
        self.sA, self.sB = st["A"], st["B"]                      # B == A with X2_SEG_ONE_STREAM=1 (A/B: everything on one stream)
        self.sC = st["C"] if self.coll else None                 # gradient all-reduces
        self.sG = st["G"] if self.coll else None                 # ITC all-gathers (see _gather)
        self.t, self.graphs = {}, {}
        self.mode, self.error = "eager", None
        self.messages = 0                       # collectives issued per step (tests / diagnostics)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._plan = None                       # per segment: tensors all-reduced in place + (static flat, views, sources)
        self._arenas = []                       # (segment, flat, params) published while capturing
        self._touch = {}                        # id(param) -> segment that last wrote its gradient
        self._capturing = False
        self._pin_accumulators()
        dev = batch["text_ids"].device
        B = batch["text_ids"].shape[0]
        E = model.embed_dim
        self.t["fi_all"] = torch.zeros(world * B, E, device=dev, requires_grad=True)
        self.t["ft_all"] = torch.zeros(world * B, E, device=dev, requires_grad=True)
        if K.DROP_EPOCH is None:                  # allocated once per process, never replaced (see GraphedStep)
            K.DROP_EPOCH = torch.zeros(1, dtype=torch.int32, device=dev)
        self._epoch = K.DROP_EPOCH
        cur = torch.cuda.current_stream()
        self.sA.wait_stream(cur)
        side, tie, hook, rule = engine.SIDE.enabled, engine.TIE_WORD_GRAD, engine.GRAD_READY_HOOK, engine.SIDE.only_from
        engine.GRAD_READY_HOOK = None             # an accelerator's early all-reduce hook must not fire from these passes
        prev_cus = self._reserve()
        try:
            for _ in range(max(warmup, 1)):       # eager, on the segments' own streams: caches, workspaces, allocator pools
                self._run("eager")
            torch.cuda.synchronize()
            if enabled and os.environ.get("X2_GRAPH", "1") != "0":
                self._capture(verbose)
        finally:
            self._unreserve(prev_cus)
            engine.SIDE.enabled, engine.TIE_WORD_GRAD, engine.GRAD_READY_HOOK, engine.SIDE.only_from = side, tie, hook, rule
            # the warm-up / capture passes counted stage calls; an accelerator's GradientBuckets reads the counters of the NEXT
            # eager backward_step (a layer that "ran twice" loses its early all-reduce)
            engine.STAGE_CALLS.clear()
        if self.coll:
            self._agree_across_ranks(verbose)
        cur.wait_stream(self.sA)

    def _reserve(self):
        """GEMM tile plans of this step's launches leave `reserved_cus` compute units to RCCL's channel kernels (x2_tune key 12);
        returns the previous value for _unreserve."""
        if self._lib is None:
            return None
        prev = self._lib.x2_tune_get(12)
        if prev != self.reserved_cus and self._lib.x2_tune(12, self.reserved_cus) != 0:
            raise RuntimeError("x2_tune(12, %d): %s" % (self.reserved_cus, self._lib.x2_last_error().decode()))
        return prev

    def _unreserve(self, prev):
        if self._lib is not None and prev is not None and prev != self.reserved_cus:
            self._lib.x2_tune(12, prev)

    # ------------------------------------------------------------------ set-up
    def _plan_digest(self):
        """What this rank will send, in order: (segment, element count) of every message of a replayed step.  All ranks
        must have derived the SAME sequence from their own capture - RCCL matches collectives by issue order, a mismatch is a
        hang, not an error."""
        import hashlib
        items = []
        for seg in self._rorder:
            inplace, packed = (self._plan or {}).get(seg, ((), None))
            items.append((seg, tuple(int(t.numel()) for t in inplace), -1 if packed is None else int(packed[0].numel())))
        return int.from_bytes(hashlib.sha256(repr(items).encode()).digest()[:7], "big"), len(items)

    def _exchange_ints(self, vals):
        """[world, len(vals)] int64 on the host: every rank's values (one small all-gather through the step's own transport)."""
        dev = self.batch["text_ids"].device
        mine = torch.tensor(vals, dtype=torch.int64, device=dev)
        out = torch.empty(self.world * len(vals), dtype=torch.int64, device=dev)
        if self.comm is not None:
            self.comm.allgather(mine, out)
        else:
            import torch.distributed as dist
            dist.all_gather(list(out.chunk(self.world)), mine, group=self.pg)
        torch.cuda.synchronize()
        return out.view(self.world, len(vals)).cpu()

    def _agree_across_ranks(self, verbose=False):
        """world > 1, after capture: (a) the launch mode is a job-wide decision - a rank whose capture failed would issue ONE flat
        all-reduce per step while the others issue per-segment messages: if any rank fell back, all do; (b) the reduction plans
        must be identical (message order and sizes), checked through a digest."""
        ok = 1 if self.graphs else 0
        digest, nseg = self._plan_digest() if ok else (0, 0)
        got = self._exchange_ints([ok, digest, nseg])
        if int(got[:, 0].min()) == 0:
            if self.graphs:
                self.error = "capture failed on rank(s) %s: all ranks run the segments eagerly" % [r for r in range(self.world) if int(got[r, 0]) == 0]
                if verbose:
                    print("SegmentedStep: " + self.error, flush=True)
            self._drop_graphs()
            return
        if not bool((got[:, 1] == got[0, 1]).all()) or not bool((got[:, 2] == got[0, 2]).all()):
            raise RuntimeError("SegmentedStep: ranks derived different gradient-reduction plans (rank %d: %d segments, digest %x; all: %s) - "
                               "the collectives would not match" % (self.rank, nseg, digest, got[:, 1:].tolist()))

    def _drop_graphs(self):
        """Back to eager segments: forget the graphs and everything derived from the capture (a half-built plan would make
        the eager fallback all-reduce stale captured arenas as well)."""
        self.graphs = {}
        self.mode = "eager"
        self._plan, self._arenas, self._rorder = None, [], []
        self._grads = []
    def _pin_accumulators(self):
        """Create (and keep) every parameter's AccumulateGrad node under the stream of the segment that will produce its
        gradient: autograd accumulates on the node's stream, and a node created lazily on another stream would pull that
        stream into the segment's capture (a fork / join edge: the slow replay path)."""
        text = self.model._bert
        on_b = set(id(p) for p in text.embeddings.parameters())
        for i in range(text.config.fusion_layer):
            on_b.update(id(p) for p in text.encoder.layer[i].parameters())
        self._acc = []
        for stream, sel in ((self.sB, True), (self.sA, False)):
            with torch.cuda.stream(stream):
                for p in self.params:
                    if (id(p) in on_b) == sel:
                        self._acc.append(p.view_as(p).grad_fn.next_functions[0][0])

    def _seg(self, mode, name, stream, fn, pool):
        with torch.cuda.stream(stream):
            if mode == "replay":
                if self.times is not None:            # X2_SEG_TIMES=1: HIP events around every segment replay (diagnostics)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self.graphs[name].replay()
                    e1.record()
                    self.times.setdefault(name, []).append((e0, e1))
                    return
                self.graphs[name].replay()
            elif mode == "capture":
                g = torch.cuda.CUDAGraph()
                before = self._grad_ids()
                self._cur_seg = name
                # thread_local: a process group's watchdog thread (RCCL) keeps polling its events while this thread captures
                with torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):
                    fn()
                self.graphs[name] = g
                after = self._grad_ids()
                for p in self.params:
                    if after[id(p)] != before[id(p)]:
                        self._touch[id(p)] = name
            else:
                fn()

    def _grad_ids(self):
        return {id(p): None if p.grad is None else (p.grad.data_ptr(), p.grad._version) for p in self.params}

    # ------------------------------------------------------------------ the segments
    def _s_text(self):
        b = self.batch
        if self.masking:
            K.mask_tokens(b["text_ids"], b["text_atts"], epoch=self._epoch, out=(b["text_ids_masked"], b["masked_pos"], b["masked_ids"]), **self.masking)
        self.t["both"] = self.model.tower_text(b["text_ids"], b["text_atts"], b["text_ids_masked"])

    def _vhook(self, ci, x):
        leaf = x.detach().requires_grad_()
        self.t["vmid"].append((x, leaf))
        return leaf

    def _s_prefetch(self):
        """Stream B, behind the text tower and under the vision tower's forward: the bf16 weight copies the tail segment would
        otherwise cast at its head, on the critical stream (fusion layers, MLM head dense, the padded vocabulary)."""
        eng = self.engine
        bert = self.model._bert
        cfg = bert.config
        lo, hi = cfg.fusion_layer, cfg.num_hidden_layers
        names = eng.bert_layer_param_names(lo, hi, cfg.fusion_layer, True)
        sd = dict(bert.encoder.named_parameters())
        eng.BertLayersFn.prepare_weights({n: sd[n] for n in names}, lo, hi, cfg.fusion_layer, True)
        te = self.model.text_encoder
        if hasattr(te, "cls"):
            eng.BANK.linear(te.cls.predictions.transform.dense.weight)
            eng.BANK.vocab(bert.embeddings.word_embeddings.weight)

    def _s_vision(self):
        b = self.batch
        enc = self.model.vision_encoder
        self.t["vmid"] = []
        enc.chunk_at, enc.chunk_hook = self.vcuts, self._vhook
        try:
            self.t["vis"] = self.model.tower_vision(b["image"], b.get("image_atts"), b.get("idx_to_group_img"), self.ret_bbox_loss)
        finally:
            enc.chunk_at, enc.chunk_hook = None, None

    def _s_feat(self):
        m = self.model
        if self.clamp_temp and isinstance(m.temp, torch.nn.Parameter):
            with torch.no_grad():
                m.temp.clamp_(0.001, 0.5)                 # Pretrain.py:327-328
        ie, _, full = self.t["vis"]
        self.t["ie"] = ie.detach().requires_grad_()
        self.t["full"] = None if full is None else full.detach().requires_grad_()
        self.t["both_leaf"] = self.t["both"].detach().requires_grad_()
        self.t["feat"] = m.tail_features(self.t["ie"], self.t["both_leaf"])

    def _gather(self):
        """ITC features of all ranks into the static leaves (eager, between two segments; xvlm.py:140-160).  The collective is
        issued from a stream of its own that is never captured: torch.distributed's NCCL (= RCCL) process group records its
        completion events on the stream the call is made from, its watchdog thread keeps querying them, and ROCm refuses
        hipEventQuery on an event whose stream has meanwhile started capturing (the next segment's capture follows at once):
        `operation not permitted on an event last recorded in a capturing stream` took the process down
        (tests/test_ddp_gpu.py::test_single_rank_rccl_through_replayed_segments)."""
        fi, ft = self.t["feat"]
        if not self.coll:
            for src, dst in ((fi, self.t["fi_all"]), (ft, self.t["ft_all"])):
                dst.detach().copy_(src.detach())
            return
        self.sG.wait_stream(self.sA)
        with torch.cuda.stream(self.sG):
            for src, dst in ((fi, self.t["fi_all"]), (ft, self.t["ft_all"])):
                out = dst.detach()
                if self.comm is not None:
                    self.comm.allgather(src.detach().contiguous(), out)
                else:
                    import torch.distributed as dist
                    dist.all_gather(list(out.chunk(self.world)), src.detach().contiguous(), group=self.pg)
                self.messages += 1
        self.sA.wait_stream(self.sG)

    def _s_loss(self):
        m, b, t = self.model, self.batch, self.t
        for leaf in (t["fi_all"], t["ft_all"], t["ie"], t["both_leaf"], t["full"]):
            if leaf is not None:
                leaf.grad = None
        fi, ft = t["feat"]
        loss = m.tail_losses(t["ie"], t["vis"][1], t["both_leaf"], fi, ft, b["text_atts"], b["masked_pos"], b["masked_ids"],
                             image_embeds_fullatts=t["full"], target_bbox=b.get("target_bbox"), is_image=b.get("is_image"),
                             ret_bbox_loss=self.ret_bbox_loss, ret_match_loss=self.ret_match_loss,
                             gathered=(t["fi_all"], t["ft_all"]))
        self._backward(loss)
        # backward of the all-gather: the local rows of the gathered features' gradient (xvlm.py:156-160), on through the
        # projection heads into the tower-output leaves
        B = fi.shape[0]
        sl = slice(self.rank * B, (self.rank + 1) * B)
        if t["fi_all"].grad is not None:          # None: the ITC loss is not part of this step's backward (a part's loss_keys)
            torch.autograd.backward([fi, ft], [t["fi_all"].grad[sl], t["ft_all"].grad[sl]])
        t["loss"] = {k: v.detach() for k, v in loss.items()}

    def _backward(self, losses):
        """total_loss(losses).backward() - unless no loss of the dict takes part in the backward (a part's loss_keys)."""
        tot = self.total_loss(losses)
        if torch.is_tensor(tot) and tot.requires_grad:
            tot.backward()

    def _s_vision_bwd(self, ci=0):
        """Backward of the ci-th vision stage counted from the top."""
        # a tower output without a gradient (a part whose loss_keys leave a branch out of the backward) is simply not fed
        if ci == 0:
            ie_out, _, full_out = self.t["vis"]
            pairs = [(ie_out, self.t["ie"].grad)]
            if full_out is not None:
                pairs.append((full_out, self.t["full"].grad))
        else:
            x, leaf = self.t["vmid"][len(self.vcuts) - ci]
            pairs = [(x, leaf.grad)]
        pairs = [(o, g) for o, g in pairs if g is not None]
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])

    def _s_vision_wgrad(self, ci):
        work = self._vq.pop(ci, None)
        self._held.append(work)                # see _s_tail_wgrad
        for fn in work or ():
            fn()

    def _s_text_bwd(self):
        if self.t["both_leaf"].grad is not None:
            torch.autograd.backward([self.t["both"]], [self.t["both_leaf"].grad])

    def _s_tail_wgrad(self):
        # The closures' operands were allocated on stream A and are read here on stream B: they stay referenced (self._held)
        # until stream A has waited for B at the end of the pass - dropped earlier, the allocator would hand their blocks to
        # the next stream-A kernels (eagerly) or to the next stream-A segment's capture while B still reads them.
        work, self._queue = self._queue, None
        self._held.append(work)
        for fn in work or ():
            fn()

    # ------------------------------------------------------------------ one step
    def _run(self, mode):
        eng, A, Bs = self.engine, self.sA, self.sB
        pa = pb = None
        if mode == "capture":
            pa, pb = self._pools
        if mode != "replay":
            # linear segments: weight-gradient GEMMs in line (the other tower fills the CUs) - unless side_stream asks for the
            # fork / join form on stream A, the capture origin of its segments (engine.SideStream.only_from)
            eng.SIDE.enabled = self.side_stream
            eng.SIDE.only_from = self.sA.cuda_stream
            eng.TIE_WORD_GRAD = True
            if self.recast_weights:
                eng.BANK.invalidate()         # as after an optimizer step: fp32 master weights are re-cast inside the step
            else:
                eng.BANK.backward_seen = False    # a later sub-iteration of the same optimizer step: the copies are current
            for p in self.params:
                p.grad = None
        with torch.cuda.stream(A):
            self._epoch.add_(1)
        Bs.wait_stream(A)
        self._seg(mode, "T", Bs, self._s_text, pb)
        if self.prefetch_casts:
            self._seg(mode, "P", Bs, self._s_prefetch, pb)
        self._seg(mode, "V", A, self._s_vision, pa)
        A.wait_stream(Bs)
        self._seg(mode, "F1", A, self._s_feat, pa)
        with torch.cuda.stream(A):
            self._gather()
        if mode != "replay":
            eng.WGRAD_QUEUE = [] if self.defer_tail_wgrad else None
            eng.AUX.enabled, eng.AUX.only_from, eng.AUX.auto = self.aux_overlap, self.sA.cuda_stream, self.aux_auto
        try:
            self._seg(mode, "F2", A, self._s_loss, pa)
        finally:
            eng.AUX.enabled, eng.AUX.only_from, eng.AUX.auto = False, None, False
        if mode != "replay":
            self._queue, eng.WGRAD_QUEUE = eng.WGRAD_QUEUE, None
        Bs.wait_stream(A)
        self._reduce("F2", A)
        nstage = len(self.vcuts) + 1
        for ci in range(nstage):
            name = "Vb" if ci == 0 else "Vb%d" % ci
            defer = self.defer_vision_wgrad and ci < nstage - 1
            if mode != "replay":
                eng.WGRAD_QUEUE = [] if defer else None
            self._seg(mode, name, A, lambda ci=ci: self._s_vision_bwd(ci), pa)
            if mode != "replay":
                self._vq[ci], eng.WGRAD_QUEUE = eng.WGRAD_QUEUE, None
            if ci == 0:
                # stream B behind the tail: text tower backward, then the tail's (fusion layers') weight gradients - off
                # stream A's critical path, under the vision backward
                self._seg(mode, "Tb", Bs, self._s_text_bwd, pb)
                self._reduce("Tb", Bs)
                if self.defer_tail_wgrad:
                    self._seg(mode, "Fw", Bs, self._s_tail_wgrad, pb)
                    self._reduce("Fw", Bs)
            if defer:
                Bs.wait_stream(A)             # this stage's input-gradient chain (its (dY, X) pairs) is enqueued on A
                wname = "Vw" if ci == 0 else "Vw%d" % ci
                self._seg(mode, wname, Bs, lambda ci=ci: self._s_vision_wgrad(ci), pb)
                self._reduce(wname, Bs)
            self._reduce(name, A)
        A.wait_stream(Bs)
        if self.sC is not None:
            A.wait_stream(self.sC)
        self._held = []

    def _capture(self, verbose):
        eng = self.engine
        self._pools = (torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle())
        eng.GRAD_READY_HOOK = lambda flat, key, also_after=None, params=(): self._arenas.append((self._cur_seg, flat, list(params)))
        gc.collect()
        self._capturing = True
        try:
            self._run("capture")
            self._grads = [(p, p.grad) for p in self.params if p.grad is not None]
            self._home = {id(p): p.grad for p in self.params if p.grad is not None}
            self._make_plan()
            self.mode = "hipgraph-segments"
        except Exception as e:                # noqa: BLE001 - anything that cannot be captured: run the segments eagerly
            self.error = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            self._drop_graphs()
            if verbose:
                print("SegmentedStep: capture failed, running eagerly (%s)" % self.error, flush=True)
            torch.cuda.synchronize()
        finally:
            eng.GRAD_READY_HOOK = None
            self._capturing = False

    # ------------------------------------------------------------------ gradient averaging (world > 1)
    def _make_plan(self):
        """Per segment: [tensors to all-reduce in place], (flat, views, sources) for the packed remainder."""
        self._plan = {}
        if not self.coll or not self.reduce_grads:
            return
        inside = lambda g, flat: flat.data_ptr() <= g.data_ptr() and g.data_ptr() + g.numel() * 4 <= flat.data_ptr() + flat.numel() * 4
        done = set()
        order = {seg: i for i, seg in enumerate(self._rorder)}       # the order the reductions are issued in (_run)
        # a gradient is complete after the later of: the segment whose autograd pass last wrote .grad, the segment that
        # published its arena (deferred weight-gradient work fills the views after autograd has handed them over)
        ready = dict(self._touch)
        for s, flat, params in self._arenas:
            for p in params:
                if id(p) in ready and order[s] > order[ready[id(p)]]:
                    ready[id(p)] = s
        for seg in self._rorder:
            whole = []
            for s, flat, params in self._arenas:
                # an arena goes as one message when every gradient it was published for still lives in it (autograd adopted
                # the views) and all of them are complete after this segment
                if params and all(id(p) not in done and p.grad is not None and inside(p.grad, flat) and ready.get(id(p)) == seg
                                  for p in params):
                    whole.append(flat)
                    done.update(id(p) for p in params)
            rest = [p for p in self.params if p.grad is not None and id(p) not in done and ready.get(id(p)) == seg]
            done.update(id(p) for p in rest)
            big = [p.grad for p in rest if p.grad.numel() >= (1 << 20) and p.grad.is_contiguous()]
            small = [p for p in rest if not (p.grad.numel() >= (1 << 20) and p.grad.is_contiguous())]
            packed = None
            if small:
                sizes = [p.grad.numel() for p in small]
                flat = torch.empty(sum(sizes), device=small[0].grad.device, dtype=small[0].grad.dtype)
                views = [v.view_as(p.grad) for v, p in zip(flat.split(sizes), small)]
                packed = (flat, views, [p.grad for p in small])
                for p, v in zip(small, views):
                    p.grad = v                    # from now on the optimizer reads the averaged copy
            self._plan[seg] = (whole + big, packed)
        self._grads = [(p, p.grad) for p in self.params if p.grad is not None]
        missing = [p for p in self.params if p.grad is not None and id(p) not in done]
        assert not missing, "SegmentedStep: %d gradients were assigned to no segment" % len(missing)

    def _all_reduce(self, flat):
        self.messages += 1
        if self.grad_bf16 and flat.dtype == torch.float32:
            # bf16 on the wire, fp32 at both ends: cast -> average -> write back (the arena keeps fp32 for the optimizer)
            buf = self._bf16_buf.get(flat.data_ptr())
            if buf is None or buf.numel() != flat.numel():
                buf = self._bf16_buf[flat.data_ptr()] = torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16)
            buf.copy_(flat.view(-1))
            self._all_reduce_raw(buf)
            flat.view(-1).copy_(buf)
            return
        self._all_reduce_raw(flat)

    def _all_reduce_raw(self, flat):
        if self.comm is not None:
            self.comm.allreduce_bucket(flat, average=True)
            return
        import torch.distributed as dist
        nccl = dist.get_backend(self.pg) == "nccl"
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.pg)
        if not nccl:
            flat.div_(self.world)

    def _reduce(self, seg, stream):
        if self._capturing and seg not in self._rorder:
            self._rorder.append(seg)
        if not self.coll or self._plan is None or self.defer_reduce:      # one rank; still warming up / capturing; or reduce_all() does it
            return
        self._reduce_now(seg, stream)

    def reduce_all(self, stream=None):
        """defer_reduce=True: every segment's reduction now, in plan order, behind `stream` (default: stream A)."""
        if not self.coll or not self._plan:
            return
        for seg in self._rorder:
            self._reduce_now(seg, self.sA if stream is None else stream)
        self.sA.wait_stream(self.sC)

    def _reduce_now(self, seg, stream):
        inplace, packed = self._plan.get(seg, ((), None))
        self.sC.wait_stream(stream)
        with torch.cuda.stream(self.sC):
            for flat in inplace:
                self._all_reduce(flat)
            if packed is not None:
                flat, views, srcs = packed
                torch._foreach_copy_(views, srcs)
                self._all_reduce(flat)

    def _reduce_eager_fallback(self):
        """No graphs (capture disabled or failed): one flat message for everything, after the last segment."""
        grads = [p.grad for p in self.params if p.grad is not None]
        sizes = [g.numel() for g in grads]
        flat = torch.empty(sum(sizes), device=grads[0].device, dtype=grads[0].dtype)
        views = [v.view_as(g) for v, g in zip(flat.split(sizes), grads)]
        torch._foreach_copy_(views, grads)
        self._all_reduce(flat)
        for p, v in zip([p for p in self.params if p.grad is not None], views):
            p.grad = v

    def segment_times(self):
        """{segment: mean ms} over the replays timed so far (X2_SEG_TIMES=1), plus start offsets relative to the first segment."""
        torch.cuda.synchronize()
        out = {}
        first = None
        for name, evs in (self.times or {}).items():
            evs = evs[len(evs) // 2:]                  # later half: warm
            out[name] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        if self.times:
            base = self.times.get("T") or next(iter(self.times.values()))
            for name, evs in self.times.items():
                k = len(evs) // 2
                out["start:" + name] = sum(base[i][0].elapsed_time(evs[i][0]) for i in range(k, len(evs))) / (len(evs) - k)
        return out

    def reattach_grads(self):
        """p.grad back to the static tensors the segments write (after an optimizer.zero_grad(set_to_none=True))."""
        for p, g in self._grads:
            if p.grad is not g:
                p.grad = g

    def __call__(self):
        cur = torch.cuda.current_stream()
        self.sA.wait_stream(cur)
        self.messages = 0
        if self.graphs:
            self.reattach_grads()
            self._run("replay")
        else:
            eng = self.engine
            side, tie, hook, rule = eng.SIDE.enabled, eng.TIE_WORD_GRAD, eng.GRAD_READY_HOOK, eng.SIDE.only_from
            eng.GRAD_READY_HOOK = None
            prev_cus = self._reserve()
            try:
                self._run("eager")
                if self.coll and self.reduce_grads and not self.defer_reduce:
                    with torch.cuda.stream(self.sA):
                        self._reduce_eager_fallback()
            finally:
                self._unreserve(prev_cus)
                eng.SIDE.enabled, eng.TIE_WORD_GRAD, eng.GRAD_READY_HOOK, eng.SIDE.only_from = side, tie, hook, rule
                eng.STAGE_CALLS.clear()
        cur.wait_stream(self.sA)
        return self.t["loss"]

    copy_inputs = staticmethod(GraphedStep.copy_inputs)


class TextOnlyStep(SegmentedStep):
    """Pretrain.run_text_iter (Pretrain.py:139-157; XVLM.forward(image=None) -> forward_text, model_pretrain.py:67-72) as three
    linear hipGraph segments: the masked ids through the text layers on stream B (XT), the upper layers WITHOUT cross-attention +
    MLM head + loss + their backward on stream A (XF), the text layers' backward on stream B (XTb) - every parameter's gradient
    is produced on the stream its AccumulateGrad node is pinned to by the image step (text layers and embeddings: B; fusion
    layers and the MLM head: A), so no segment contains a fork, and the tied word-embedding / decoder gradient takes the
    same single-buffer route (engine.TIE_WORD_GRAD).  batch: text_ids_masked, text_atts, masked_pos, masked_ids (+ text_ids,
    unused by the loss, kept for copy_inputs).  A part of a MixedStep (the reference adds t_loss['loss_mlm'] * iter_perc to the
    iteration's one backward, Pretrain.py:232-235), or a step of its own."""

    def _s_xt(self):
        b = self.batch
        self.t["xt"] = self.model.get_text_embeds(b["text_ids_masked"], b["text_atts"])

    def _s_xf(self):
        m, b = self.model, self.batch
        leaf = self.t["xt_leaf"] = self.t["xt"].detach().requires_grad_()
        seq = m._bert(encoder_embeds=leaf, attention_mask=b["text_atts"], mode="fusion").last_hidden_state
        loss, m.last_mlm_lse, m.last_mlm_logits = m.text_encoder.mlm_loss_from_hidden(seq, b["masked_pos"], b["masked_ids"])
        losses = {"loss_mlm": loss}
        self._backward(losses)
        self.t["loss"] = {k: v.detach() for k, v in losses.items()}

    def _s_xtb(self):
        torch.autograd.backward([self.t["xt"]], [self.t["xt_leaf"].grad])

    def _run(self, mode):
        eng, A, Bs = self.engine, self.sA, self.sB
        pa = pb = None
        if mode == "capture":
            pa, pb = self._pools
        if mode != "replay":
            eng.SIDE.enabled = False
            eng.SIDE.only_from = A.cuda_stream
            eng.TIE_WORD_GRAD = True              # the embedding lookup of the tied parameter is part of this pass (XT)
            eng.WGRAD_QUEUE = None
            if self.recast_weights:
                eng.BANK.invalidate()
            else:
                eng.BANK.backward_seen = False
            for p in self.params:
                p.grad = None
        with torch.cuda.stream(A):
            self._epoch.add_(1)
        Bs.wait_stream(A)
        self._seg(mode, "XT", Bs, self._s_xt, pb)
        A.wait_stream(Bs)
        self._seg(mode, "XF", A, self._s_xf, pa)
        Bs.wait_stream(A)
        self._reduce("XF", A)
        self._seg(mode, "XTb", Bs, self._s_xtb, pb)
        self._reduce("XTb", Bs)
        A.wait_stream(Bs)
        if self.sC is not None:
            A.wait_stream(self.sC)
        self._held = []


class MixedStep:
    """Pretrain.run_mixed_iter (Pretrain.py:189-252) as replayed hipGraph segments: several sub-iterations of ONE optimizer step -
    the image batch, the region batch, a video batch - whose gradients accumulate (the reference sums the losses of the image and
    region forwards into one backward_step and gives the video batch a backward_step of its own, `Pretrain.py:197, 247`; either
    way the optimizer sees the sum of the sub-iterations' gradients) and are averaged over the ranks ONCE, after the last one.

    Every part is a SegmentedStep of its own (own static inputs, own graphs and gradient buffers; the bf16 weight copies are
    re-cast inside every part's graphs); after the last part ONE multi-tensor add folds the later parts'
    gradients into the first part's static buffers, and - more than one rank - the first part's reduction plan runs: one
    message per layer arena, as for a single iteration.  p.grad of every parameter is a static tensor afterwards.

    parts: list of dict(batch=static device tensors, weight=iter_perc (1.0), ret_bbox_loss=False, ret_match_loss=True,
                        negatives=None (tests: injected hard negatives, static int32 device tensors),
                        loss_keys=None (the losses of this part that enter the backward; None = all it returns - e.g.
                                        ("loss_bbox", "loss_giou") for the region part under the reference's regions_use_bbox_only,
                                        Pretrain.py:220-222), total_loss=None (or any callable losses -> scalar, instead of
                                        weight / loss_keys), text_only=False (Pretrain.run_text_iter's part: batch without an
                                        image, TextOnlyStep)).
    The first part must be an image / region / video part (it owns the reduction plan).  Not expressible here: the mtext
    sub-iteration (Pretrain.py:237-245) - XVLM.forward has no text_ids_2 arguments, that is XVLMPlus (out of scope) -
    and a video part with a backward_step of its own (Pretrain.py:193-201) is simply one more part: the optimizer sees the
    same sum of gradients either way.
    Returns the list of the parts' loss dicts (unweighted, as the reference logs them)."""

    def __init__(self, model, parts, world=1, rank=0, process_group=None, comm=None, warmup=1, enabled=True, verbose=False, **kw):
        self.model, self.world, self.rank = model, world, rank
        self.parts, self.steps = parts, []
        # what a part decides for itself must not arrive a second time through **kw (accelerator.mixed_step(**kw) passes it on)
        for k in ("total_loss", "reduce_grads", "defer_reduce", "ret_bbox_loss", "ret_match_loss"):
            if k in kw:
                raise TypeError("MixedStep: %r is a per-part setting (parts[i][%r]), not a keyword of the whole iteration" % (k, k))
        assert not parts[0].get("text_only"), "MixedStep: the first part owns the reduction plan and must not be text-only"
        kept = model.injected_negatives
        try:
            for i, part in enumerate(parts):
                w = float(part.get("weight", 1.0))
                keys = part.get("loss_keys")
                total = part.get("total_loss") or (lambda losses, w=w, keys=keys: w * sum(v for k, v in losses.items() if keys is None or k in keys))
                # a part's injected negatives (tests) are that part's only: the model gets its own back afterwards, and the eager
                # fallback of __call__ sets them per part again
                model.injected_negatives = part.get("negatives", kept)
                cls = TextOnlyStep if part.get("text_only") else SegmentedStep
                self.steps.append(cls(model, part["batch"], world=world, rank=rank, process_group=process_group, comm=comm,
                                      warmup=warmup, enabled=enabled, verbose=verbose,
                                      ret_bbox_loss=part.get("ret_bbox_loss", False), ret_match_loss=part.get("ret_match_loss", True),
                                      # every part re-casts inside its own graphs: a later part may use weights the first does
                                      # not (bbox head after an image part), and a copy cached from a part's eager warm-up would
                                      # never be refreshed after optimizer steps (0.4 ms per extra part)
                                      recast_weights=kw.get("recast_weights", True),
                                      clamp_temp=(i == 0) and kw.get("clamp_temp", True),
                                      total_loss=total, reduce_grads=(i == 0), defer_reduce=True,
                                      **{k: v for k, v in kw.items() if k not in ("recast_weights", "clamp_temp")}))
        finally:
            model.injected_negatives = kept
        first = self.steps[0]
        self.params = first.params
        self.coll = first.coll
        self.mode = "hipgraph-segments" if all(s_.graphs for s_ in self.steps) else "eager"
        if self.mode == "eager":
            for s_ in self.steps:                     # all parts or none (with more than one rank every part's mode was already agreed job-wide)
                s_._drop_graphs()
        self.error = next((s_.error for s_ in self.steps if s_.error), None)
        self.messages = 0
        self._dst, self._src, self._final, self._extra = [], [], [], None
        if self.mode != "eager":
            self._plan_accumulation()

    def _plan_accumulation(self):
        first = self.steps[0]
        home = dict(first._home)                      # where the first part leaves a gradient = what its reduction plan sends
        lonely = []                                   # parameters only later parts produce a gradient for (bbox head after an image part)
        for s_ in self.steps[1:]:
            for p in self.params:
                g = s_._home.get(id(p))
                if g is None:
                    continue
                if id(p) in home:
                    self._dst.append(home[id(p)])
                    self._src.append(g)
                else:
                    home[id(p)] = g
                    lonely.append(p)
        final = {id(p): g for p, g in first._grads}   # after the plan: packed views for the small tensors
        if lonely and self.coll:
            sizes = [home[id(p)].numel() for p in lonely]
            flat = torch.empty(sum(sizes), device=home[id(lonely[0])].device, dtype=torch.float32)
            views = [v.view_as(home[id(p)]) for v, p in zip(flat.split(sizes), lonely)]
            self._extra = (flat, views, [home[id(p)] for p in lonely])
            for p, v in zip(lonely, views):
                final[id(p)] = v
        else:
            for p in lonely:
                final[id(p)] = home[id(p)]
        self._final = [(p, final[id(p)]) for p in self.params if id(p) in final]

    def copy_inputs(self, i, batch):
        GraphedStep.copy_inputs(self.parts[i]["batch"], batch)

    def __call__(self):
        cur = torch.cuda.current_stream()
        first = self.steps[0]
        self.messages = 0
        if self.mode != "eager":
            losses = [s_() for s_ in self.steps]
            first.sA.wait_stream(cur)
            with torch.cuda.stream(first.sA):
                if self._dst:
                    torch._foreach_add_(self._dst, self._src)
                first.reduce_all()
                if self._extra is not None:
                    flat, views, srcs = self._extra
                    torch._foreach_copy_(views, srcs)
                    first._all_reduce(flat)
            cur.wait_stream(first.sA)
            for p, g in self._final:
                if p.grad is not g:
                    p.grad = g
            self.messages = sum(s_.messages for s_ in self.steps)
            return losses
        # eager fallback (capture disabled or failed): every part launches its kernels from Python and starts from p.grad = None
        total, losses = {}, []
        kept = self.model.injected_negatives
        for s_, part in zip(self.steps, self.parts):
            self.model.injected_negatives = part.get("negatives", kept)
            try:
                losses.append(s_())
            finally:
                self.model.injected_negatives = kept
            for p in self.params:
                if p.grad is not None:
                    total[id(p)] = p.grad if id(p) not in total else total[id(p)] + p.grad
        for p in self.params:
            p.grad = total.get(id(p))
        if self.coll:
            first.sA.wait_stream(cur)
            with torch.cuda.stream(first.sA):
                first._reduce_eager_fallback()
            cur.wait_stream(first.sA)
        self.messages = sum(s_.messages for s_ in self.steps)
        return losses
