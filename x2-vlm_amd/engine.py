"""Stage-level execution of the X^2-VLM step on the HIP kernels.

Each stage (vision encoder, a range of BERT layers, embeddings, MLM head) is ONE
torch.autograd.Function whose forward and hand-written backward are fixed sequences of C-ABI
kernel launches (kernels.py); autograd only connects the few stages.  Precision policy (the
reference ran apex O1 fp16: fp32 weights, fp16 GEMM operands, fp32 LN/softmax/loss): fp32 master
weights, bf16 GEMM/attention operands with fp32 MFMA accumulation, fp32 residual stream, LayerNorm
statistics, softmax, losses and every reduction.

Reference behaviour reproduced: beit2.py:125-209, 378-436 (vision), xbert.py:189-216, 322-625,
652-767 (text/fusion layers), xbert.py:785-824, 1644-1661 (MLM head).
"""
import itertools
import math
import os
import weakref

import torch

from . import kernels as K
from ._lib import raw_stream

BF16, F32 = torch.bfloat16, torch.float32


# ----------------------------------------------------------------------------- bf16 weight copies

_SERIAL = itertools.count(1)


def _tok(w):
    """Identity of a parameter OBJECT for cache keys.  id() is not one: when a model is dropped and another built (every
    parity test does), a new Parameter can land on the same address with the same storage pointer and version, and a
    cache keyed on id() then serves the previous model's weights."""
    t = w.__dict__.get("_x2_serial")
    if t is None:
        t = w.__dict__["_x2_serial"] = next(_SERIAL)
    return t


class TScale:
    """Marks the scale vector of a (weight, TScale(gamma)) group handed to WeightBank.prepare."""

    def __init__(self, g):
        self.g = g


class WeightBank:
    """bf16 (and transposed bf16) copies of the fp32 master weights.

    Staleness: a copy is rebuilt when its parameter's version counter or storage moved, AND every copy is dropped at
    the first use outside autograd's backward that follows a backward pass.  The second rule exists because optimizers that write
    through `p.data` (transformers 4.12.5 AdamW: `p.data.addcdiv_`, the reference's optim.py:102; apex FusedAdam) never
    bump the version counter: after a backward the weights must be assumed updated.  Inference (no backward in between)
    keeps its copies; a real training step re-casts once per step either way (2 x 0.5 GB of bf16, ~0.4 ms)."""

    def __init__(self):
        self._c = {}
        self.backward_seen = False

    def note_backward(self):
        """Called by every stage's backward: an optimizer step may follow."""
        self.backward_seen = True

    def _maybe_expire(self):
        if self.backward_seen and torch._C._current_graph_task_id() == -1:
            self._c.clear()
            self.backward_seen = False

    def _get(self, key, vers, build, owners=()):
        ent = self._c.get(key)
        if ent is None or ent[0] != vers:
            ent = (vers, build(), tuple(weakref.ref(w) for w in owners))
            self._c[key] = ent
        return ent[1]

    def _purge(self):
        """Drop the copies of parameters that no longer exist."""
        dead = [k for k, e in self._c.items() if any(r() is None for r in e[2])]
        for k in dead:
            del self._c[k]

    def linear(self, *ws, tscale=None):
        """(W [N,K] bf16, W^T [K,N] bf16) of one weight or of several stacked along N.  tscale (fp32 [N] parameter): the
        TRANSPOSED copy holds tscale[n] * W[n, k] - a layer scale folded into the weight its input-gradient GEMM reads
        (rowwise.hip, layer-scale backward); the plain copy is W."""
        self._maybe_expire()
        key = tuple(_tok(w) for w in ws) + ((("ts", _tok(tscale)),) if tscale is not None else ())
        own = ws + ((tscale,) if tscale is not None else ())
        vers = tuple((w._version, w.data_ptr()) for w in own)

        def build():
            w2 = [w.detach().reshape(w.shape[0], -1) for w in ws]
            src = w2[0] if len(w2) == 1 else torch.cat(w2, 0)
            plain, tr = K.cast_transpose_bf16(src.contiguous())
            if tscale is not None:
                tr = K.cast_transpose_bf16((src * tscale.detach().reshape(-1, 1)).contiguous())[1]
            return plain, tr
        return self._get(key, vers, build, own)

    def prepare(self, groups):
        """Build every stale (W, W^T) pair of `groups` (tuples of weights, as `linear` takes them) with ONE
        multi-tensor launch and one allocation, instead of one cast launch (+ a concatenation) per weight when the
        layer first asks for it: a tower's ~50-100 weights are all re-cast once per optimizer step."""
        self._maybe_expire()
        self._purge()
        todo = []
        for ws in groups:
            sc = None
            if isinstance(ws[-1], TScale):                 # (weight, TScale(gamma)): see `linear(w, tscale=gamma)`
                assert len(ws) == 2
                sc, ws = ws[1].g, ws[:1]
            key = tuple(_tok(w) for w in ws) + ((("ts", _tok(sc)),) if sc is not None else ())
            own = tuple(ws) + ((sc,) if sc is not None else ())
            vers = tuple((w._version, w.data_ptr()) for w in own)
            ent = self._c.get(key)
            if ent is not None and ent[0] == vers:
                continue
            shapes = [(w.shape[0], w.numel() // w.shape[0]) for w in ws]
            C_ = shapes[0][1]
            if any(c != C_ or (r | c) & 3 or w.dtype != F32 or not w.is_contiguous() for (r, c), w in zip(shapes, ws)):
                continue                                   # left to the one-at-a-time path of `linear`
            if sc is not None and (sc.dtype != F32 or not sc.is_contiguous() or sc.numel() != shapes[0][0]):
                continue
            todo.append((key, vers, ws, sum(r for r, _ in shapes), C_, sc, own))
        if not todo:
            return
        total = sum(2 * t[3] * t[4] for t in todo)
        flat = torch.empty(total, device=todo[0][2][0].device, dtype=BF16)
        desc, o = [], 0
        for key, vers, ws, R, C_, sc, own in todo:
            plain = flat[o:o + R * C_].view(R, C_)
            tr = flat[o + R * C_:o + 2 * R * C_].view(C_, R)
            o += 2 * R * C_
            roff = 0
            for w in ws:
                desc.append((w.data_ptr(), plain.data_ptr(), tr.data_ptr(), w.shape[0], C_, R, roff, 0 if sc is None else sc.data_ptr()))
                roff += w.shape[0]
            self._c[key] = (vers, (plain, tr), tuple(weakref.ref(w) for w in own))
        K.cast_transpose_multi(desc)

    def prepare_vectors(self, groups):
        """Stacked fp32 bias vectors (tuples of 1-D parameters; an int n stands for n zeros), all built by one launch
        into one buffer; `vector(*items)` then returns the cached stack."""
        self._maybe_expire()
        todo, total = [], 0
        for items in groups:
            key = ("vec",) + tuple(it if isinstance(it, int) else _tok(it) for it in items)
            vers = tuple(0 if isinstance(it, int) else (it._version, it.data_ptr()) for it in items)
            ent = self._c.get(key)
            if ent is not None and ent[0] == vers:
                continue
            if any(not isinstance(it, int) and (it.dtype != F32 or not it.is_contiguous()) for it in items):
                continue                                   # raw-address copy below: left to `vector`'s reshape + cat build
            n = sum(it if isinstance(it, int) else it.numel() for it in items)
            todo.append((key, vers, items, n))
            total += (n + 3) // 4 * 4
        if not todo:
            return
        dev = next(it for _, _, items, _ in todo for it in items if not isinstance(it, int)).device
        flat = torch.empty(total, device=dev, dtype=F32)
        desc, o = [], 0
        for key, vers, items, n in todo:
            out = flat[o:o + n]
            oo = o
            for it in items:
                m = it if isinstance(it, int) else it.numel()
                desc.append((0 if isinstance(it, int) else it.data_ptr(), flat.data_ptr() + 4 * oo, m))
                oo += m
            o += (n + 3) // 4 * 4
            self._c[key] = (vers, out, tuple(weakref.ref(it) for it in items if not isinstance(it, int)))
        K.copy_f32_multi(desc)

    def vector(self, *items):
        self._maybe_expire()
        key = ("vec",) + tuple(it if isinstance(it, int) else _tok(it) for it in items)
        vers = tuple(0 if isinstance(it, int) else (it._version, it.data_ptr()) for it in items)

        ent = self._c.get(key)
        if ent is None or ent[0] != vers:
            tens = [it for it in items if not isinstance(it, int)]
            # the multi-tensor launch copies numel() floats from the raw address: only fp32, contiguous items qualify
            # (anything else - a sliced or half-precision bias - takes the reshape + cat build below, which handles both)
            if tens[0].is_cuda and all(it.dtype == F32 and it.is_contiguous() for it in tens):
                self.prepare_vectors([items])          # one multi-tensor launch (zeros included) instead of zeros + cat
                ent = self._c[key]
            else:
                dev = next(it for it in items if not isinstance(it, int)).device
                val = torch.cat([torch.zeros(it, device=dev, dtype=F32) if isinstance(it, int) else it.detach().reshape(-1).to(F32) for it in items])
                ent = self._c[key] = (vers, val, tuple(weakref.ref(it) for it in items if not isinstance(it, int)))
        return ent[1]

    def vocab(self, w):
        """word embeddings [V,Hd] -> (bf16 [Vp,Hd] zero-padded rows, bf16 [Hd,Vp]), Vp = V rounded to 64."""
        self._maybe_expire()

        def build():
            V, Hd = w.shape
            Vp = K.round_up(V, 64)
            src = torch.zeros(Vp, Hd, device=w.device, dtype=F32)
            src[:V] = w.detach()
            return K.cast_transpose_bf16(src)
        return self._get(("vocab", _tok(w)), (w._version, w.data_ptr()), build, (w,))

    def invalidate(self):
        """Forget every copy (bench.py does this each step: a real training step re-casts the weights
        the optimizer just updated)."""
        self._c.clear()
        self.backward_seen = False


BANK = WeightBank()
SPLIT_DECODER_DGRAD = os.environ.get("X2_SPLIT_DECODER_DGRAD", "1") == "1"     # A/B switches (probes/run_ab3.sh)
FUSED_MLM_CE = os.environ.get("X2_FUSED_MLM_CE", "1") == "1"
FUSE_DGELU_COLSUM = os.environ.get("X2_FUSE_DGELU_COLSUM", "1") == "1"       # fc1 / intermediate bias gradient from the GELU' GEMM's epilogue
KEEP_MLM_LOGITS = False     # tests: also materialise the MLM logits (inspection only; the loss still comes from the fused path)
# Tied decoder / word-embedding gradient in ONE buffer (graph.SegmentedStep switches it on for its passes): the MLM head's
# backward parks its [V, Hd] weight gradient here instead of handing it to autograd, and the embedding backward - which
# always runs later in the same backward sweep, the embeddings being upstream of everything - scatter-adds its rows on top
# and returns the sum as THE gradient of the shared parameter.  Saves a 94 MB zero-fill and the 3 x 94 MB accumulate of
# the second contribution, and - the reason it exists - keeps the parameter's gradient on the stream of the text tower, so
# the hipGraph segment of the tail needs no edge into that stream.  Only valid when the embedding lookup of the same
# parameter is part of the same autograd graph, which the caller guarantees.
TIE_WORD_GRAD = False
_TIED_DWORD = {}


class Grads:
    """Gradient tensors of one layer (or stage), carved out of ONE flat fp32 arena so that data
    parallelism can all-reduce a layer's gradients in place, as one message, the moment its
    backward kernels are enqueued (accelerator.GradientBuckets).  Tensors the kernels accumulate into
    with atomics sit at the front of the arena and are zeroed with a single memset; weight gradients
    are written whole by the TN GEMM.

    spec: list of (name, shape, zero_init); a name may be a fused block (e.g. q/k/v weights stacked)
    that the caller later splits into views with `alias`."""

    def __init__(self, device, spec, key=None, params=(), zero=True):
        """zero=False: the caller zeroes the accumulated-into front of several arenas with ONE launch (`Grads.zero_all`) -
        a stage's backward knows all its layers' arenas up front (30 memsets per base step were 30 launches)."""
        self.params = list(params)      # the Parameter objects whose gradients live in this arena (data parallelism)
        spec = [s for s in spec if s[2]] + [s for s in spec if not s[2]]
        offs, o = [], 0
        for _, shape, _z in spec:
            offs.append(o)
            o += (int(torch.Size(shape).numel()) + 3) // 4 * 4
        self.nzero = sum((int(torch.Size(sh).numel()) + 3) // 4 * 4 for _, sh, z in spec if z)
        self.flat = torch.empty(o, device=device, dtype=F32)
        if zero and self.nzero:
            self.flat[:self.nzero].zero_()
        self.g = {n: self.flat[of:of + torch.Size(sh).numel()].view(sh) for (n, sh, _z), of in zip(spec, offs)}
        self.key = key

    @staticmethod
    def zero_all(arenas, extra=()):
        """One multi-tensor launch that zeroes the accumulated-into fronts of `arenas` (built with zero=False) and any further
        fp32 tensors in `extra`."""
        desc = [(0, g.flat.data_ptr(), g.nzero) for g in arenas if g.nzero] + [(0, t.data_ptr(), t.numel()) for t in extra]
        if not desc:
            return
        if arenas[0].flat.is_cuda if arenas else extra[0].is_cuda:
            K.copy_f32_multi(desc)
        else:
            for g in arenas:
                g.flat[:g.nzero].zero_()
            for t in extra:
                t.zero_()

    def __getitem__(self, name):
        return self.g[name]

    def alias(self, name, view):
        self.g[name] = view

    def publish(self, also_after=None):
        """Every kernel writing this arena has been enqueued: on the current stream and, for the weight-gradient
        GEMMs, on the side stream up to the event `also_after`."""
        if GRAD_READY_HOOK is not None:
            GRAD_READY_HOOK(self.flat, self.key, also_after, self.params)

    def take(self, names):
        return [self.g.pop(n) for n in names]


class SideStream:
    """Weight-gradient GEMMs need only (dY, X) of their layer, not the rest of the backward chain: they are
    launched on a second HIP stream so that their workgroups fill the CUs the dgrad / attention / LayerNorm kernels
    of the main stream leave idle at their tails (594-tile GEMMs on 512 workgroup slots, HBM-bound row kernels)."""

    def __init__(self):
        self.streams = {}          # one side stream per launching stream (vision and text stages run concurrently)
        self.held = {}             # launching stream -> tensors the side stream still reads (released by join())
        self.enabled = True
        # hipGraph capture (graph.GraphedStep): raw handle of the ONE stream allowed to fork a side stream - the capture's
        # origin stream.  ROCm 7's stream capture crashes in hipStreamEndCapture when a stream that is already part of the
        # capture waits on an event of another non-origin stream (probes/graph_capture_probe.py: nested_fork,
        # sibling_cross), so under capture every side stream may only fork from and join into the origin: launches
        # coming from any other stream (the text tower's) run inline on that stream.
        self.only_from = None

    @property
    def stream(self):
        key = raw_stream()
        if key not in self.streams:
            self.streams[key] = torch.cuda.Stream()
        return self.streams[key]

    def launch(self, fn, tensors):
        """Run fn() on the side stream after everything enqueued so far on the current stream; returns an event
        recorded behind it (None when running inline)."""
        if not self.enabled or not torch.cuda.is_available() or (self.only_from is not None and raw_stream() != self.only_from):
            fn()
            return None
        side = self.stream
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            fn()
            done.record()
        # Keep the operands alive until join() instead of Tensor.record_stream(): a step makes ~440 such registrations, each
        # a host call plus an event the caching allocator must poll before it may reuse the block.  Once the launching stream
        # has waited for the side stream, dropping the references frees the blocks for that stream in stream order.
        self.held.setdefault(raw_stream(), []).append(tensors)
        return done

    def join(self):
        """Make the current stream wait for the side stream (call before gradients leave the stage)."""
        if self.enabled and torch.cuda.is_available():
            key = raw_stream()
            if key in self.streams and (self.only_from is None or key == self.only_from):
                torch.cuda.current_stream().wait_stream(self.streams[key])
            self.held.pop(key, None)


SIDE = SideStream()
# A second stream INSIDE the fusion stack's stage, for the work that hangs off its dependency chain: the cross-attention K/V
# projections of all layers (forward: they depend on the image tokens only) and the K/V-side gradients (backward: dK / dV of every
# cross-attention and their input-gradient GEMM feed the VISION tower, not the text chain).  graph.SegmentedStep switches it on
# around its tail segment, where the GPU otherwise runs one 1.0-1.4-round launch of the M = 7680 chain at a time.
AUX = SideStream()
AUX.enabled = False
# auto: the fork is skipped where the cross-attention backward runs in ONE pass (kernels.attn_bwd_form == 2: 30 text tokens on <= 208
# image tokens) - dK / dV then come out of the kernel that forms dQ, nothing but one GEMM per layer is left for the second stream, and
# the fork-free tail measured faster (profiles/r10c_*: 22.33 vs 22.40 ms per base step) while its graph replays as ONE launch
AUX.auto = False


def _begin_layer_backward():
    """Collect the layer's stage-2 parameter-gradient reductions instead of launching them in line: they go out as ONE
    multi-tensor launch per layer (pair) in front of its weight-gradient GEMMs - on the side stream when there is one,
    in line otherwise (single-stream hipGraph segments: 170 reduction launches per base step became 15)."""
    K.DEFERRED = [] if torch.cuda.is_available() else None


class _LayerPairs:
    """Weight gradients of TWO consecutive layers per grouped launch when they fit its 8 problem slots (a vision block
    or a text layer has 4): 216 tiles of 256x256 fill the 256 CUs without splitting the contraction, so there are no
    partial tiles to write and re-add, and half the launches.  Layers with more problems (fusion: 7) go alone.
    (Measured and dropped: un-pairing the LAST two vision layers so that only two weight gradients remain after the critical
    stream has finished block 0 - the split launches cost more than the shorter tail saves: 25.48 -> 25.90 ms per base step.)"""

    enabled = os.environ.get("X2_PAIR_WGRAD", "1") == "1"

    def __init__(self):
        self.pending = []
        self.extra = []
        self.fin = []

    def add(self, G, tn, param_only=(), finish=()):
        """param_only: closures of further parameter-gradient-only kernels of this layer (bias column sums, the bias-table
        gradient), handed over only while WGRAD_QUEUE collects: they then run with the deferred work, off the input-gradient
        chain's stream.  They must write through aliases (Tensor.detach()) of the arena views, see _launch.
        finish: K.layerscale_finish items of this layer: run behind its weight-gradient GEMMs (and the stage-2 reductions that
        complete their column sums)."""
        deferred, K.DEFERRED = K.DEFERRED, None
        if len(tn) > 4 or not self.enabled:
            self.flush()
            self._launch([(G, tn, deferred)], list(param_only), list(finish))
            return
        self.pending.append((G, tn, deferred))
        self.extra += list(param_only)
        self.fin += list(finish)
        if len(self.pending) == 2:
            self.flush()

    def flush(self):
        if self.pending:
            entries, self.pending = self.pending, []
            extra, self.extra = self.extra, []
            fin, self.fin = self.fin, []
            self._launch(entries, extra, fin)

    @staticmethod
    def _launch(entries, extra=(), fin=()):
        deferred = [d for _, _, ds in entries for d in (ds or ())]
        tn = [pr for _, t, _ in entries for pr in t]

        def work():
            if deferred:
                K.reduce_partials_multi(deferred)
            K.gemm_tn_grouped(tn)
            if fin:
                K.layerscale_finish(fin)
        if WGRAD_QUEUE is not None:
            # the caller (graph.SegmentedStep) runs this layer's parameter-gradient work later, in a segment of its own on another
            # stream: the closure keeps the operands alive; the gradient views autograd receives now are filled then.  It must
            # NOT hold the view objects autograd is about to receive (AccumulateGrad adopts a gradient tensor only when nobody else
            # references it, and would otherwise store a copy of the still empty arena): fresh aliases of the same memory instead.
            tn_a = [tuple(pr[:2]) + (pr[2].detach(),) + tuple(pr[3:]) for pr in tn]
            def_a = [(ws, nblk, nk, width, tuple(None if o is None else o.detach() for o in outs)) for ws, nblk, nk, width, outs in deferred]
            fin_a = [tuple(t.detach() if (torch.is_tensor(t) and not isinstance(t, torch.nn.Parameter)) else t for t in it) for it in fin]
            pubs = [(G.flat, G.key, list(G.params)) for G, _, _ in entries]

            extra = list(extra)

            def later():
                for fn in extra:
                    fn()
                if def_a:
                    K.reduce_partials_multi(def_a)
                K.gemm_tn_grouped(tn_a)
                if fin_a:
                    K.layerscale_finish(fin_a)
                if GRAD_READY_HOOK is not None:
                    for flat, key, params in pubs:
                        GRAD_READY_HOOK(flat, key, None, params)
            # one contribution per parameter: autograd ADDS a second call's arena view to the first's the moment it receives
            # it - both still empty - and the late GEMMs would then overwrite the sum with one of them
            later.params = set(id(p_) for _, _, ps in pubs for p_ in ps)
            for fn in WGRAD_QUEUE:
                if getattr(fn, "params", set()) & later.params:
                    raise RuntimeError("engine.WGRAD_QUEUE: a layer ran twice in one pass; its weight gradients cannot be deferred "
                                       "(graph.SegmentedStep: defer_tail_wgrad / defer_vision_wgrad)")
            WGRAD_QUEUE.append(later)
            return
        assert not extra, "param_only closures are only collected while WGRAD_QUEUE is set"
        keep = [t for pr in tn for t in pr[:2]] + [G.flat for G, _, _ in entries] + [d[0] for d in deferred]
        done = SIDE.launch(work, keep)
        for G, _, _ in entries:
            G.publish(done)


def _param_only(po, fn, src, dst):
    """A kernel that only produces a parameter gradient (fn(src, dst), dst a view of the layer's arena): now, or - while
    WGRAD_QUEUE collects - with the layer's deferred work (through an alias of dst, see _LayerPairs._launch)."""
    if WGRAD_QUEUE is not None:
        dst_a = dst.detach()
        po.append(lambda: fn(src, dst_a))
    else:
        fn(src, dst)


def _finish_layer_backward(G, tn):
    """Side stream: the deferred reductions, then the layer's weight-gradient GEMMs; publish the arena."""
    deferred, K.DEFERRED = K.DEFERRED, None
    _LayerPairs._launch([(G, tn, deferred)])


# None, or a list that collects the weight-gradient work of every layer backward instead of launching it (closures: stage-2
# reductions + grouped TN GEMM + arena publication).  graph.SegmentedStep sets it around the tail segment: the fusion layers'
# weight gradients (1.3 ms of 256x256-tile GEMMs per base step) leave the step's critical path - they run as a segment of
# their own on the text stream, concurrently with the vision tower's backward.  Only valid when nothing reads or accumulates
# into those gradients before the queue has run (one contribution per parameter and step).
WGRAD_QUEUE = None
GRAD_READY_HOOK = None      # set by accelerator.GradientBuckets: f(flat_fp32_arena, key, side_stream_event, parameters)
STAGE_CALLS = {}            # key -> number of forward calls since the last reset (see GradientBuckets)


def _count_call(key):
    STAGE_CALLS[key] = STAGE_CALLS.get(key, 0) + 1


def _mask_pad(add_mask, Lk):
    """[S, Lk] additive fp32 mask -> [S, round_up(Lk,64)] contiguous (pad value irrelevant: kernels
    mask keys >= Lk by index)."""
    S = add_mask.shape[0]
    out = torch.zeros(S, K.round_up(Lk, 64), device=add_mask.device, dtype=F32)
    out[:, :Lk] = add_mask
    return out


# ----------------------------------------------------------------------------- vision encoder

def vision_param_names(depth, lo=0, hi=None):
    """Parameters of blocks [lo, hi) of a depth-block encoder, plus the stem's (lo == 0: cls token, patch embedding) and the
    head's (hi == depth: fc_norm), in the order VisionEncoderFn takes them."""
    hi = depth if hi is None else hi
    names = []
    if lo == 0:
        names += ["cls_token", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    if hi == depth:
        names += ["fc_norm.weight", "fc_norm.bias"]
    for i in range(lo, hi):
        b = "blocks.%d." % i
        names += [b + s for s in ("gamma_1", "gamma_2", "norm1.weight", "norm1.bias", "attn.q_bias", "attn.v_bias",
                                  "attn.relative_position_bias_table", "attn.qkv.weight", "attn.proj.weight",
                                  "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                                  "mlp.fc2.weight", "mlp.fc2.bias")]
    return names


class VisionEncoderFn(torch.autograd.Function):
    """Blocks [lo, hi) of the BEiT-2 encoder as ONE autograd stage.  The whole encoder (lo = 0, hi = depth, the default):
    image (B,3,R,R) fp32 -> tokens (B,1+P,D) fp32: patch embed, pre-LN blocks with rel-pos-bias attention and layer scale,
    fc_norm over patches, token 0 = (weighted) mean of patches.  A chunk that does not start at block 0 takes the residual
    stream (B,1+P,D) fp32 instead of the image; one that does not end at the last block returns the residual stream.  Chunks
    let a caller cut the tower's backward into pieces (graph.SegmentedStep: gradient all-reduce of the upper blocks behind the
    backward of the lower ones); chained they compute exactly what the single stage does.

    meta: dict(depth, [lo, hi,] heads, patch, eps, rel_index[int64 (T,T)], pool_w [B,P] fp32 or None,
               drop_path: None or list per block of the ENCODER (indexed by absolute block number) of (rs1, rs2): fp32 [B*T]
               per-row keep/(1-p) factors of the attention / MLP branch (stochastic depth, beit2.py:205-207))."""

    @staticmethod
    def forward(ctx, image, meta, *params):
        depth = meta["depth"]
        lo, hi = meta.get("lo", 0), meta.get("hi", depth)
        stem, head = lo == 0, hi == depth
        names = vision_param_names(depth, lo, hi)
        p = dict(zip(names, params))
        H = meta["heads"]
        if stem:
            B, R, ps = image.shape[0], image.shape[-1], meta["patch"]
            P_ = (R // ps) ** 2
            T = P_ + 1
            D = p["cls_token"].numel()
        else:
            B, T, D = image.shape
            P_ = T - 1
        M = B * T
        assert D == 64 * H, "vision width %d / %d heads: the attention kernels are built for head dim 64" % (D, H)
        scale = (D // H) ** -0.5
        # proj / fc2: the transposed copies (input-gradient GEMMs) carry the layer scale, see the backward
        BANK.prepare(([(p["patch_embed.proj.weight"],)] if stem else []) +
                     [(p["blocks.%d.%s.weight" % (i, n)],) for i in range(lo, hi) for n in ("attn.qkv", "mlp.fc1")] +
                     [(p["blocks.%d.%s.weight" % (i, n)], TScale(p["blocks.%d.%s" % (i, g_)])) for i in range(lo, hi)
                      for n, g_ in (("attn.proj", "gamma_1"), ("mlp.fc2", "gamma_2"))])
        BANK.prepare_vectors([(p["blocks.%d.attn.q_bias" % i], D, p["blocks.%d.attn.v_bias" % i]) for i in range(lo, hi)])
        cols = None
        if stem:
            cols = K.patchify(image.contiguous(), ps)
            wpe, _ = BANK.linear(p["patch_embed.proj.weight"])
            patch = K.gemm_nt(cols, wpe, bias=p["patch_embed.proj.bias"], out_dtype=F32)
            x = K.assemble_tokens(patch, p["cls_token"].reshape(-1), B, P_).view(M, D)
        else:
            x = image.contiguous().view(M, D)          # read only below: the caller's tensor is never written
        saved = []
        dpath = meta.get("drop_path")
        for i in range(lo, hi):
            b = "blocks.%d." % i
            rs1, rs2 = dpath[i] if dpath is not None else (None, None)
            h1, _, mean1, rstd1 = K.layernorm_fwd(x, p[b + "norm1.weight"], p[b + "norm1.bias"], meta["eps"])
            wqkv, _ = BANK.linear(p[b + "attn.qkv.weight"])
            qkv_bias = BANK.vector(p[b + "attn.q_bias"], D, p[b + "attn.v_bias"])          # no k bias: beit2.py:129
            qkv = K.gemm_nt(h1, wqkv, bias=qkv_bias)
            bias, biasT = K.relpos_bias(p[b + "attn.relative_position_bias_table"].detach(), meta["rel_index"], log2=True)
            att = torch.empty(M, D, device=x.device, dtype=BF16)
            lse = torch.empty(B * H * T, device=x.device, dtype=F32)
            K.attn_fwd(K.view3(qkv, B, T, 0), K.view3(qkv, B, T, D), K.view3(qkv, B, T, 2 * D), B, B, H, T, T, scale,
                       K.view3(att, B, T), lse, bias=bias, bias_log2=True)
            wproj, _ = BANK.linear(p[b + "attn.proj.weight"], tscale=p[b + "gamma_1"])
            x1 = K.gemm_nt(att, wproj, bias=p[b + "attn.proj.bias"], gamma=p[b + "gamma_1"], resid=x, out_dtype=F32, rowscale=rs1)
            h2, _, mean2, rstd2 = K.layernorm_fwd(x1, p[b + "norm2.weight"], p[b + "norm2.bias"], meta["eps"])
            w1, _ = BANK.linear(p[b + "mlp.fc1.weight"])
            w2, _ = BANK.linear(p[b + "mlp.fc2.weight"], tscale=p[b + "gamma_2"])
            pre = torch.empty(M, w1.shape[0], device=x.device, dtype=BF16)
            act = K.gemm_nt(h2, w1, bias=p[b + "mlp.fc1.bias"], aux=pre, act=1)
            x2 = K.gemm_nt(act, w2, bias=p[b + "mlp.fc2.bias"], gamma=p[b + "gamma_2"], resid=x1, out_dtype=F32, rowscale=rs2)
            saved.append((x, h1, mean1, rstd1, qkv, bias, biasT, att, lse, x1, h2, mean2, rstd2, pre, act))
            x = x2
        final = None
        if head:
            out = torch.empty(B, T, D, device=x.device, dtype=F32)
            _, _, meanf, rstdf = K.layernorm_fwd(x, p["fc_norm.weight"], p["fc_norm.bias"], meta["eps"], rows=B * P_, period=P_,
                                                 want_bf16=False, y_f32=out.view(M, D))
            K.pool_tokens(out, meta.get("pool_w"))
            final = (x, meanf, rstdf)
            _count_call(("vit-head", id(p["fc_norm.weight"])))
        else:
            out = x.view(B, T, D) if hi > lo else x.view(B, T, D).clone()
        for i in range(lo, hi):
            _count_call(("vit", id(p["blocks.%d.gamma_1" % i])))
        if stem:
            _count_call(("vit-stem", id(p["cls_token"])))
        ctx.meta, ctx.saved, ctx.final, ctx.cols = meta, saved, final, cols
        ctx.params = params
        ctx.dims = (B, P_, T, H, D, M, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        meta, params = ctx.meta, ctx.params
        BANK.note_backward()
        depth = meta["depth"]
        lo, hi = meta.get("lo", 0), meta.get("hi", depth)
        stem, head = lo == 0, hi == depth
        names = vision_param_names(depth, lo, hi)
        p = dict(zip(names, params))
        B, P_, T, H, D, M, scale = ctx.dims
        dev = dout.device
        out = {}
        if head:
            x_last, meanf, rstdf = ctx.final
            hn = ("fc_norm.weight", "fc_norm.bias")
            Gh = Grads(dev, [(n, p[n].shape, True) for n in hn], key=("vit-head", id(p["fc_norm.weight"])), params=[p[n] for n in hn])
            g = dout.contiguous().clone()
            K.pool_tokens(g, meta.get("pool_w"), bwd=True)
            dx, _ = K.layernorm_bwd(g.view(M, D), x_last, meanf, rstdf, p["fc_norm.weight"], Gh["fc_norm.weight"], Gh["fc_norm.bias"],
                                    period=P_)
            Gh.publish()
            out.update(Gh.g)
        else:
            dx = dout.contiguous().view(M, D)          # read only below
        dS = torch.empty(B, H, T, K.round_up(T, 64), device=dev, dtype=BF16) if hi > lo else None
        F4 = p["blocks.%d.mlp.fc1.weight" % lo].shape[0] if hi > lo else 0
        dpath = meta.get("drop_path")
        pairs = _LayerPairs()
        # every block's gradient arena up front: their accumulated-into fronts are zeroed by ONE launch
        arenas = {}
        for i in range(lo, hi):
            b = "blocks.%d." % i
            arenas[i] = Grads(dev, [("gamma_1", (D,), True), ("gamma_2", (D,), True), ("norm1.weight", (D,), True), ("norm1.bias", (D,), True),
                                    ("qkv_bias", (3 * D,), True), ("attn.relative_position_bias_table", p[b + "attn.relative_position_bias_table"].shape, True),
                                    ("attn.proj.bias", (D,), True), ("norm2.weight", (D,), True), ("norm2.bias", (D,), True),
                                    ("mlp.fc1.bias", (F4,), True), ("mlp.fc2.bias", (D,), True),
                                    ("_cs1", (D,), True), ("_cs2", (D,), True),     # column sums of the gradients entering the two layer scales
                                    ("attn.qkv.weight", (3 * D, D), False), ("attn.proj.weight", (D, D), False),
                                    ("mlp.fc1.weight", (F4, D), False), ("mlp.fc2.weight", (D, F4), False)], key=("vit", id(p[b + "gamma_1"])),
                              params=[p[n] for n in names if n.startswith(b)], zero=False)
        Gs = None
        if stem:
            sn = ("cls_token", "patch_embed.proj.weight", "patch_embed.proj.bias")
            Gs = Grads(dev, [(n, p[n].shape, True) for n in sn], key=("vit-stem", id(p["cls_token"])), params=[p[n] for n in sn], zero=False)
        Grads.zero_all(list(arenas.values()) + ([Gs] if Gs is not None else []))
        # Layer scale (x_out = x_in + r * gamma * u, u = A . W^T + b; r = DropPath factor): its backward never forms gamma * dX and
        # never reads u.  With dX' = r * dX:  dA = dX' . (diag(gamma) W) (the bank's transposed copies of proj / fc2 carry gamma),
        # G = dX'^T . A from the weight-gradient GEMM, then K.layerscale_finish: dgamma = rowdot(G, W) + b * colsum(dX'),
        # db = gamma * colsum(dX'), dW = diag(gamma) G.  bf16(dX') and colsum(dX') are by-products of the LayerNorm backward that
        # produces dX (post=...); only the gradient that enters this stage from outside needs a pass of its own.
        dxb = None
        for i in reversed(range(lo, hi)):
            b = "blocks.%d." % i
            rs1, rs2 = dpath[i] if dpath is not None else (None, None)
            (x, h1, mean1, rstd1, qkv, bias, biasT, att, lse, x1, h2, mean2, rstd2, pre, act) = ctx.saved[i - lo]
            ctx.saved[i - lo] = None
            G = arenas.pop(i)
            _begin_layer_backward()
            _, w2T = BANK.linear(p[b + "mlp.fc2.weight"], tscale=p[b + "gamma_2"])
            _, w1T = BANK.linear(p[b + "mlp.fc1.weight"])
            _, wprojT = BANK.linear(p[b + "attn.proj.weight"], tscale=p[b + "gamma_1"])
            _, wqkvT = BANK.linear(p[b + "attn.qkv.weight"])
            if dxb is None:
                dxb = K.rowscale_cast_colsum(dx, G["_cs2"], rowscale=rs2)
            if FUSE_DGELU_COLSUM:     # fc1's bias gradient as per-wave partial rows from the GEMM epilogue (no pass over dpre, no atomics)
                dpre = K.gemm_nt_dgelu_colsum(dxb, w2T, pre, G["mlp.fc1.bias"])
            else:
                dpre = K.gemm_nt(dxb, w2T, aux=pre, act=2)
                K.colsum_bf16(dpre, G["mlp.fc1.bias"])    # two-stage sums: 20 us; fused into the GEMM epilogue with atomics: 30 us
            dh2 = K.gemm_nt(dpre, w1T)            # bf16, like the fp16 grad_input of the reference's O1 linears: half the bytes
            dx1, dx1b = K.layernorm_bwd(dh2, x1, mean2, rstd2, p[b + "norm2.weight"], G["norm2.weight"], G["norm2.bias"], dres=dx,
                                        post=(rs1, G["_cs1"]))
            datt = K.gemm_nt(dx1b, wprojT)
            dqkv = torch.empty_like(qkv)
            delta = torch.empty_like(lse)
            # parameter-only gradients of the block (bias table, q / v bias): with the deferred weight-gradient work when a queue
            # collects (then the dS stream needs a buffer of its own per block), in line otherwise
            po = []
            dS_i = torch.empty_like(dS) if WGRAD_QUEUE is not None else dS
            # q / v bias gradient: where the backward is one workgroup per (sequence, head) (form 1) it leaves per-sequence column sums of dQ / dV
            # (round 6: no pass over the [M, 3D] gradient - 58 MB per block); their sum over B joins the layer's stage-2 reductions
            cs = torch.empty(B, 2, D, device=qkv.device, dtype=F32) if os.environ.get("X2_FUSE_QKV_BIAS_COLSUM", "1") == "1" else None   # (=0: A/B)
            form = K.attn_bwd(K.view3(qkv, B, T, 0), K.view3(qkv, B, T, D), K.view3(qkv, B, T, 2 * D), K.view3(att, B, T),
                              K.view3(datt, B, T), B, B, H, T, T, scale, lse, delta, K.view3(dqkv, B, T, 0), K.view3(dqkv, B, T, D),
                              K.view3(dqkv, B, T, 2 * D), dS=dS_i, bias=bias, biasT=biasT, bias_log2=True, colsum_ws=cs)
            fused_cs = cs is not None and form in (1, 3)
            if fused_cs:
                item = (cs, B, 2, D, (G["qkv_bias"][:D], G["qkv_bias"][2 * D:]))
                if K.DEFERRED is not None:
                    K.DEFERRED.append(item)
                else:
                    K.reduce_partials(*item)
            if WGRAD_QUEUE is not None:
                tbl_a, qb_a = G["attn.relative_position_bias_table"].detach(), G["qkv_bias"].detach()
                po.append(lambda dS_i=dS_i, tbl_a=tbl_a: K.relpos_bias_bwd(dS_i, meta["rel_index"], tbl_a))
                if not fused_cs:
                    po.append(lambda dqkv=dqkv, qb_a=qb_a: K.colsum_bf16(dqkv, qb_a))
            else:
                K.relpos_bias_bwd(dS_i, meta["rel_index"], G["attn.relative_position_bias_table"])
                if not fused_cs:
                    K.colsum_bf16(dqkv, G["qkv_bias"])
            G.alias("attn.q_bias", G["qkv_bias"][:D])
            G.alias("attn.v_bias", G["qkv_bias"][2 * D:])
            dh1 = K.gemm_nt(dqkv, wqkvT)
            if i > lo:      # the gradient leaving this block enters the MLP layer scale of the block below
                below = (dpath[i - 1][1] if dpath is not None else None, arenas[i - 1]["_cs2"])
                dxn, dxnb = K.layernorm_bwd(dh1, x, mean1, rstd1, p[b + "norm1.weight"], G["norm1.weight"], G["norm1.bias"], dres=dx1, post=below)
            else:
                dxn, dxnb = K.layernorm_bwd(dh1, x, mean1, rstd1, p[b + "norm1.weight"], G["norm1.weight"], G["norm1.bias"], dres=dx1)
            tn = [(dxb, act, G["mlp.fc2.weight"]), (dpre, h2, G["mlp.fc1.weight"]),
                  (dx1b, att, G["attn.proj.weight"]), (dqkv, h1, G["attn.qkv.weight"])]
            fin = [(G["mlp.fc2.weight"], p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"], p[b + "gamma_2"], G["_cs2"], G["gamma_2"], G["mlp.fc2.bias"]),
                   (G["attn.proj.weight"], p[b + "attn.proj.weight"], p[b + "attn.proj.bias"], p[b + "gamma_1"], G["_cs1"], G["gamma_1"],
                    G["attn.proj.bias"])]
            pairs.add(G, tn, po, fin)
            for n in names:
                if n.startswith(b):
                    out[n] = G.g[n[len(b):]]
            dx, dxb = dxn, dxnb
        pairs.flush()
        if stem:
            dpatch = K.assemble_tokens_bwd(dx.view(B, T, D), Gs["cls_token"].view(-1))
            K.colsum_bf16(dpatch, Gs["patch_embed.proj.bias"])
            # 9 output tiles of 256 x 256 only: the 12k-long contraction in 8 slices per tile, partial tiles added in a fixed order
            K.gemm_tn_grouped([(dpatch, ctx.cols, Gs["patch_embed.proj.weight"].view(D, -1))], accumulate=True,
                              split=8 if B * P_ >= 4096 else 1)
            SIDE.join()
            Gs.publish()
            out.update(Gs.g)
            return (None, None) + tuple(out[n] for n in names)
        SIDE.join()
        return (dx.view(B, T, D), None) + tuple(out[n] for n in names)


# ----------------------------------------------------------------------------- BERT layers

_ATT = ("self.query.weight", "self.query.bias", "self.key.weight", "self.key.bias", "self.value.weight", "self.value.bias",
        "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias")
_FFN = ("intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight", "output.dense.bias",
        "output.LayerNorm.weight", "output.LayerNorm.bias")


def bert_layer_param_names(lo, hi, fusion_at, with_cross):
    names = []
    for i in range(lo, hi):
        b = "layer.%d." % i
        names += [b + "attention." + s for s in _ATT]
        if with_cross and i >= fusion_at:
            names += [b + "crossattention." + s for s in _ATT]
        names += [b + s for s in _FFN]
    return names


class BertLayersFn(torch.autograd.Function):
    """hidden (S,L,Hd) fp32 through BERT layers [lo,hi): self-attention, cross-attention to `enc`
    (layers >= fusion_at, only when enc is given), FFN, all post-LN.

    enc: (Bi,T,Dv) fp32 image tokens or None; sequence s attends image kv_idx[s] (several text rows
    may share one image: K/V are projected once per image per layer).
    meta: dict(lo, hi, fusion_at, heads, eps, self_mask [S,Lp] fp32 additive, enc_mask [S,Tp],
               kv_idx/seq_off/seq_ids int32 or None,
               drop: None or dict(seed, p_hidden, p_attn): training-mode dropout (xbert.py:399, 429, 513), masks are
               regenerated in the backward from (seed, site = 8*layer + {0 self probs, 1 self out, 2 cross probs,
               3 cross out, 4 ffn out}, element index))."""

    @staticmethod
    def prepare_weights(p, lo, hi, fusion_at, cross):
        """bf16 (W, W^T) copies and stacked bias vectors of layers [lo, hi) in two multi-tensor launches (no-ops for what is
        fresh).  Also called ahead of time, on another stream, by graph.SegmentedStep (the fusion layers' casts run under the
        vision tower's forward instead of at the head of the tail segment)."""
        groups, vecs = [], []
        for i in range(lo, hi):
            b = "layer.%d." % i
            a, c = b + "attention.", b + "crossattention."
            groups += [(p[a + "self.query.weight"], p[a + "self.key.weight"], p[a + "self.value.weight"]),
                       (p[a + "output.dense.weight"],), (p[b + "intermediate.dense.weight"],), (p[b + "output.dense.weight"],)]
            vecs.append((p[a + "self.query.bias"], p[a + "self.key.bias"], p[a + "self.value.bias"]))
            if cross and i >= fusion_at:
                groups += [(p[c + "self.query.weight"],), (p[c + "self.key.weight"], p[c + "self.value.weight"]),
                           (p[c + "output.dense.weight"],)]
                vecs.append((p[c + "self.key.bias"], p[c + "self.value.bias"]))
        BANK.prepare(groups)
        BANK.prepare_vectors(vecs)

    @staticmethod
    def _drop(meta, layer, kind):
        d = meta.get("drop")
        if not d:
            return K.NO_DROP
        return K.dropout_spec(d["p_attn"] if kind in (0, 2) else d["p_hidden"], d["seed"], 8 * layer + kind)

    @staticmethod
    def forward(ctx, hidden, enc, meta, *params):
        cross = enc is not None
        names = bert_layer_param_names(meta["lo"], meta["hi"], meta["fusion_at"], cross)
        p = dict(zip(names, params))
        S, L, Hd = hidden.shape
        H, eps = meta["heads"], meta["eps"]
        M = S * L
        assert Hd == 64 * H, "hidden size %d / %d heads: the attention kernels are built for head dim 64" % (Hd, H)
        scale = 1.0 / math.sqrt(Hd // H)
        dev = hidden.device
        h = hidden.contiguous().view(M, Hd)
        hb = K.cast_bf16(h)
        encb = None
        if cross:
            Bi, T, Dv = enc.shape
            encb = K.cast_bf16(enc.contiguous().view(Bi * T, Dv))
        BertLayersFn.prepare_weights(p, meta["lo"], meta["hi"], meta["fusion_at"], cross)
        saved = []
        kv_ahead = {}
        use_aux = cross and AUX.enabled and not (AUX.auto and K.attn_bwd_form(L, enc.shape[1], shared_kv=meta.get("seq_off") is not None,
                                                                               dropout=BertLayersFn._drop(meta, meta["hi"] - 1, 2)[0] != 0) == 2)
        if use_aux:
            # every layer's K/V projection of the image tokens now, on the second stream: layer i waits for its event only
            for i in range(max(meta["lo"], meta["fusion_at"]), meta["hi"]):
                c = "layer.%d.crossattention." % i
                wkv, _ = BANK.linear(p[c + "self.key.weight"], p[c + "self.value.weight"])
                bkv = BANK.vector(p[c + "self.key.bias"], p[c + "self.value.bias"])
                kv = torch.empty(Bi * T, wkv.shape[0], device=dev, dtype=BF16)
                kv_ahead[i] = (kv, AUX.launch(lambda kv=kv, wkv=wkv, bkv=bkv: K.gemm_nt(encb, wkv, bias=bkv, out=kv), [encb, wkv, bkv, kv]))
        for i in range(meta["lo"], meta["hi"]):
            b = "layer.%d." % i
            a = b + "attention."
            wqkv, _ = BANK.linear(p[a + "self.query.weight"], p[a + "self.key.weight"], p[a + "self.value.weight"])
            bqkv = BANK.vector(p[a + "self.query.bias"], p[a + "self.key.bias"], p[a + "self.value.bias"])
            qkv = K.gemm_nt(hb, wqkv, bias=bqkv)
            att = torch.empty(M, Hd, device=dev, dtype=BF16)
            lse = torch.empty(S * H * L, device=dev, dtype=F32)
            dr = BertLayersFn._drop
            K.attn_fwd(K.view3(qkv, S, L, 0), K.view3(qkv, S, L, Hd), K.view3(qkv, S, L, 2 * Hd), S, S, H, L, L, scale,
                       K.view3(att, S, L), lse, mask=meta["self_mask"], drop=dr(meta, i, 0))
            wo, _ = BANK.linear(p[a + "output.dense.weight"])
            s1 = K.gemm_nt(att, wo, bias=p[a + "output.dense.bias"], resid=h, out_dtype=F32, drop=dr(meta, i, 1))
            h1b, h1, m1, r1 = K.layernorm_fwd(s1, p[a + "output.LayerNorm.weight"], p[a + "output.LayerNorm.bias"], eps, want_f32=True)
            cr = None
            h2b, h2 = h1b, h1
            if cross and i >= meta["fusion_at"]:
                c = b + "crossattention."
                wq, _ = BANK.linear(p[c + "self.query.weight"])
                wkv, _ = BANK.linear(p[c + "self.key.weight"], p[c + "self.value.weight"])
                bkv = BANK.vector(p[c + "self.key.bias"], p[c + "self.value.bias"])
                q2 = K.gemm_nt(h1b, wq, bias=p[c + "self.query.bias"])
                if i in kv_ahead:
                    kv, ev = kv_ahead.pop(i)
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                else:
                    kv = K.gemm_nt(encb, wkv, bias=bkv)
                att2 = torch.empty(M, Hd, device=dev, dtype=BF16)
                lse2 = torch.empty(S * H * L, device=dev, dtype=F32)
                K.attn_fwd(K.view3(q2, S, L), K.view3(kv, Bi, T, 0), K.view3(kv, Bi, T, Hd), S, Bi, H, L, T, scale,
                           K.view3(att2, S, L), lse2, mask=meta["enc_mask"], kv_idx=meta["kv_idx"], seq_off=meta.get("seq_off"),
                           seq_ids=meta.get("seq_ids"), drop=dr(meta, i, 2))
                wo2, _ = BANK.linear(p[c + "output.dense.weight"])
                s2 = K.gemm_nt(att2, wo2, bias=p[c + "output.dense.bias"], resid=h1, out_dtype=F32, drop=dr(meta, i, 3))
                h2b, h2, m2, r2 = K.layernorm_fwd(s2, p[c + "output.LayerNorm.weight"], p[c + "output.LayerNorm.bias"], eps, want_f32=True)
                cr = (q2, kv, att2, lse2, s2, m2, r2)
            wi, _ = BANK.linear(p[b + "intermediate.dense.weight"])
            wout, _ = BANK.linear(p[b + "output.dense.weight"])
            pre = torch.empty(M, wi.shape[0], device=dev, dtype=BF16)
            act = K.gemm_nt(h2b, wi, bias=p[b + "intermediate.dense.bias"], aux=pre, act=1)
            s3 = K.gemm_nt(act, wout, bias=p[b + "output.dense.bias"], resid=h2, out_dtype=F32, drop=dr(meta, i, 4))
            h3b, h3, m3, r3 = K.layernorm_fwd(s3, p[b + "output.LayerNorm.weight"], p[b + "output.LayerNorm.bias"], eps, want_f32=True)
            saved.append((hb, qkv, att, lse, s1, m1, r1, h1b, cr, h2b, pre, act, s3, m3, r3))
            h, hb = h3, h3b
        AUX.join()
        for i in range(meta["lo"], meta["hi"]):
            _count_call(("bert", id(p["layer.%d.attention.self.query.weight" % i])))
        ctx.meta, ctx.saved, ctx.params, ctx.encb = meta, saved, params, encb
        ctx.dims = (S, L, Hd, H, M, scale, cross, enc.shape if cross else None)
        return h.view(S, L, Hd)

    @staticmethod
    def backward(ctx, dh_out):
        meta, params, encb = ctx.meta, ctx.params, ctx.encb
        BANK.note_backward()
        S, L, Hd, H, M, scale, cross, enc_shape = ctx.dims
        names = bert_layer_param_names(meta["lo"], meta["hi"], meta["fusion_at"], cross)
        p = dict(zip(names, params))
        dev = dh_out.device
        dh = dh_out.contiguous().view(M, Hd)
        denc = None
        out = {}
        if cross:
            Bi, T, Dv = enc_shape
        pairs = _LayerPairs()
        # every layer's gradient arena up front: their accumulated-into fronts are zeroed by ONE launch
        arenas = {}
        for li, i in enumerate(range(meta["lo"], meta["hi"])):
            b = "layer.%d." % i
            Ff = p[b + "intermediate.dense.weight"].shape[0]
            spec = [("a.qkv_bias", (3 * Hd,), True), ("attention.output.dense.bias", (Hd,), True),
                    ("attention.output.LayerNorm.weight", (Hd,), True), ("attention.output.LayerNorm.bias", (Hd,), True),
                    ("intermediate.dense.bias", (Ff,), True), ("output.dense.bias", (Hd,), True),
                    ("output.LayerNorm.weight", (Hd,), True), ("output.LayerNorm.bias", (Hd,), True),
                    ("a.qkv_weight", (3 * Hd, Hd), False), ("attention.output.dense.weight", (Hd, Hd), False),
                    ("intermediate.dense.weight", (Ff, Hd), False), ("output.dense.weight", (Hd, Ff), False)]
            if ctx.saved[li][8] is not None:            # cr: the layer has a cross-attention branch in this pass
                spec += [("crossattention.self.query.bias", (Hd,), True), ("c.kv_bias", (2 * Hd,), True),
                         ("crossattention.output.dense.bias", (Hd,), True), ("crossattention.output.LayerNorm.weight", (Hd,), True),
                         ("crossattention.output.LayerNorm.bias", (Hd,), True), ("crossattention.self.query.weight", (Hd, Hd), False),
                         ("c.kv_weight", (2 * Hd, Dv), False), ("crossattention.output.dense.weight", (Hd, Hd), False)]
            arenas[i] = Grads(dev, spec, key=("bert", id(p[b + "attention.self.query.weight"])), params=[p[n] for n in names if n.startswith(b)],
                              zero=False)
        Grads.zero_all(list(arenas.values()))
        for li, i in reversed(list(enumerate(range(meta["lo"], meta["hi"])))):
            b = "layer.%d." % i
            a = b + "attention."
            hb, qkv, att, lse, s1, m1, r1, h1b, cr, h2b, pre, act, s3, m3, r3 = ctx.saved[li]
            ctx.saved[li] = None
            Ff = p[b + "intermediate.dense.weight"].shape[0]
            G = arenas.pop(i)
            tn, po = [], []
            _begin_layer_backward()
            ds3, ds3b = K.layernorm_bwd(dh, s3, m3, r3, p[b + "output.LayerNorm.weight"], G["output.LayerNorm.weight"],
                                        G["output.LayerNorm.bias"], dcol=G["output.dense.bias"], want_bf16=True,
                                        drop_out=BertLayersFn._drop(meta, i, 4))
            _, woutT = BANK.linear(p[b + "output.dense.weight"])
            _, wiT = BANK.linear(p[b + "intermediate.dense.weight"])
            if FUSE_DGELU_COLSUM:
                dpre = K.gemm_nt_dgelu_colsum(ds3b, woutT, pre, G["intermediate.dense.bias"])
            else:
                dpre = K.gemm_nt(ds3b, woutT, aux=pre, act=2)
                K.colsum_bf16(dpre, G["intermediate.dense.bias"])
            dh2 = K.gemm_nt(dpre, wiT, resid=ds3, out_dtype=F32)
            tn += [(ds3b, act, G["output.dense.weight"]), (dpre, h2b, G["intermediate.dense.weight"])]
            if cr is not None:
                c = b + "crossattention."
                q2, kv, att2, lse2, s2, m2, r2 = cr
                ds2, ds2b = K.layernorm_bwd(dh2, s2, m2, r2, p[c + "output.LayerNorm.weight"], G["crossattention.output.LayerNorm.weight"],
                                            G["crossattention.output.LayerNorm.bias"], dcol=G["crossattention.output.dense.bias"],
                                            want_bf16=True, drop_out=BertLayersFn._drop(meta, i, 3))
                _, wo2T = BANK.linear(p[c + "output.dense.weight"])
                datt2 = K.gemm_nt(ds2b, wo2T)
                dq2 = torch.empty_like(q2)
                dkv = torch.empty_like(kv)
                delta2 = torch.empty_like(lse2)
                # the K/V side of this layer (dK / dV, then their input gradient into the image tokens) feeds the vision tower only:
                # on the second stream when there is one.  Its readers inside this stage (bias sums, weight gradients) are either
                # deferred past the stage's join (WGRAD_QUEUE) or wait for its event at the end of the layer (kv_done below).
                kv_done = None

                def attn2(phase, ask_form=False):
                    return K.attn_bwd(K.view3(q2, S, L), K.view3(kv, Bi, T, 0), K.view3(kv, Bi, T, Hd), K.view3(att2, S, L),
                                      K.view3(datt2, S, L), S, Bi, H, L, T, scale, lse2, delta2, K.view3(dq2, S, L), K.view3(dkv, Bi, T, 0),
                                      K.view3(dkv, Bi, T, Hd), mask=meta["enc_mask"], kv_idx=meta["kv_idx"], seq_off=meta["seq_off"],
                                      seq_ids=meta["seq_ids"], drop=BertLayersFn._drop(meta, i, 2), phase=phase, ask_form=ask_form)
                # one workgroup per (image, head) forms dQ, dK and dV in one pass where the library has that form: no dK / dV half is left
                # for the second stream (only the K/V input gradient below)
                one_pass = attn2(0, ask_form=True) != 0
                aux = AUX.enabled and not (AUX.auto and one_pass)
                attn2(1 if aux and not one_pass else 0)
                _param_only(po, K.colsum_bf16, dq2, G["crossattention.self.query.bias"])
                G.alias("crossattention.self.key.bias", G["c.kv_bias"][:Hd])
                G.alias("crossattention.self.value.bias", G["c.kv_bias"][Hd:])
                G.alias("crossattention.self.key.weight", G["c.kv_weight"][:Hd])
                G.alias("crossattention.self.value.weight", G["c.kv_weight"][Hd:])
                _, wqT = BANK.linear(p[c + "self.query.weight"])
                _, wkvT = BANK.linear(p[c + "self.key.weight"], p[c + "self.value.weight"])
                dh1 = K.gemm_nt(dq2, wqT, resid=ds2, out_dtype=F32)
                if aux:
                    first = denc is None
                    if first:
                        denc = torch.empty(Bi * T, Dv, device=dev, dtype=F32)       # one buffer, accumulated in place layer by layer

                    def kv_side(attn2=attn2, dkv=dkv, wkvT=wkvT, first=first, denc=denc, one_pass=one_pass):
                        if not one_pass:
                            attn2(2)
                        K.gemm_nt(dkv, wkvT, resid=None if first else denc, out=denc)
                    kv_done = AUX.launch(kv_side, [q2, kv, att2, datt2, lse2, delta2, dq2, dkv, wkvT, denc])
                else:
                    denc = K.gemm_nt(dkv, wkvT, resid=denc, out_dtype=F32)
                # longest contraction (image tokens) first: its tiles start in the first round of the grouped launch
                tn = [(dkv, encb, G["c.kv_weight"])] + tn + [(ds2b, att2, G["crossattention.output.dense.weight"]),
                                                            (dq2, h1b, G["crossattention.self.query.weight"])]
            else:
                dh1 = dh2
            ds1, ds1b = K.layernorm_bwd(dh1, s1, m1, r1, p[a + "output.LayerNorm.weight"], G["attention.output.LayerNorm.weight"],
                                        G["attention.output.LayerNorm.bias"], dcol=G["attention.output.dense.bias"], want_bf16=True,
                                        drop_out=BertLayersFn._drop(meta, i, 1))
            _, woT = BANK.linear(p[a + "output.dense.weight"])
            datt = K.gemm_nt(ds1b, woT)
            dqkv = torch.empty_like(qkv)
            delta = torch.empty_like(lse)
            K.attn_bwd(K.view3(qkv, S, L, 0), K.view3(qkv, S, L, Hd), K.view3(qkv, S, L, 2 * Hd), K.view3(att, S, L), K.view3(datt, S, L),
                       S, S, H, L, L, scale, lse, delta, K.view3(dqkv, S, L, 0), K.view3(dqkv, S, L, Hd), K.view3(dqkv, S, L, 2 * Hd),
                       mask=meta["self_mask"], drop=BertLayersFn._drop(meta, i, 0))
            _param_only(po, K.colsum_bf16, dqkv, G["a.qkv_bias"])
            for k3, nm in enumerate(("query", "key", "value")):
                G.alias("attention.self.%s.bias" % nm, G["a.qkv_bias"][k3 * Hd:(k3 + 1) * Hd])
                G.alias("attention.self.%s.weight" % nm, G["a.qkv_weight"][k3 * Hd:(k3 + 1) * Hd])
            _, wqkvT = BANK.linear(p[a + "self.query.weight"], p[a + "self.key.weight"], p[a + "self.value.weight"])
            dh = K.gemm_nt(dqkv, wqkvT, resid=ds1, out_dtype=F32)
            tn += [(ds1b, att, G["attention.output.dense.weight"]), (dqkv, hb, G["a.qkv_weight"])]
            if cr is not None:
                if kv_done is not None and WGRAD_QUEUE is None:
                    torch.cuda.current_stream().wait_event(kv_done)      # dkv is read below, in this stage
                _param_only(po, K.colsum_bf16, dkv, G["c.kv_bias"])
            pairs.add(G, tn, po)
            for n in names:
                if n.startswith(b):
                    out[n] = G.g[n[len(b):]]
        pairs.flush()
        SIDE.join()
        AUX.join()
        d_enc = denc.view(enc_shape) if denc is not None else None
        return (dh.view(S, L, Hd), d_enc, None) + tuple(out[n] for n in names)


# ----------------------------------------------------------------------------- embeddings

class EmbeddingsFn(torch.autograd.Function):
    """ids (S,L) -> dropout(LayerNorm(word + position + type0)) (S,L,Hd) fp32.  xbert.py:189-216.
    drop: kernels.dropout_spec(...) triple or kernels.NO_DROP."""

    @staticmethod
    def forward(ctx, ids, eps, drop, word, pos, typ, lnw, lnb):
        S, L = ids.shape
        ids = ids.contiguous()
        e = K.embed_fwd(ids, word.detach(), pos.detach(), typ.detach())
        _, y, mean, rstd = K.layernorm_fwd(e, lnw, lnb, eps, want_bf16=False, want_f32=True, drop=drop)
        ctx.save_for_backward(ids, e, mean, rstd, word, pos, typ, lnw)
        ctx.drop = drop
        return y.view(S, L, -1)

    @staticmethod
    def backward(ctx, dy):
        ids, e, mean, rstd, word, pos, typ, lnw = ctx.saved_tensors
        n = lnw.numel()
        zeros = torch.zeros(2 * n + pos.numel() + typ.numel(), device=dy.device, dtype=F32)      # one fill for the four small gradients
        dw, db = zeros[:n], zeros[n:2 * n]
        dpos, dtyp = zeros[2 * n:2 * n + pos.numel()].view_as(pos), zeros[2 * n + pos.numel():].view_as(typ)
        de, _ = K.layernorm_bwd(dy.contiguous().view(e.shape), e, mean, rstd, lnw, dw, db, drop_in=ctx.drop)
        dword = _TIED_DWORD.pop(_tok(word), None)          # the tied decoder's gradient, if the MLM head left it (TIE_WORD_GRAD)
        if dword is None:
            dword = torch.zeros_like(word)
        K.embed_bwd(ids, de, dword, dpos, dtyp)
        return None, None, None, dword, dpos, dtyp, dw, db


# ----------------------------------------------------------------------------- MLM head + loss

class MlmLossFn(torch.autograd.Function):
    """rows (R,Hd) fp32 at the masked positions -> mean CE over labels != -100.
    transform dense + GELU + LayerNorm, decoder tied to the word embeddings + bias
    (xbert.py:785-824, 1653-1661).  Returns (loss, lse [R], logits): the per-row log-partition and - only when they were
    materialised (unfused path, keep_logits, or KEEP_MLM_LOGITS for tests; else None) - the fp32 logits [R, Vp]; neither is
    differentiable.

    Fused path (default): the decoder GEMM's epilogue reduces the logits to softmax statistics, the backward recomputes the
    GEMM and writes (softmax - onehot) * g / count straight to bf16 (csrc/gemm.hip, x2_mlm_ce_fwd / _bwd): no [R, Vp] fp32
    tensor (94 MB at R = 768) is written, saved or read."""

    @staticmethod
    def forward(ctx, rows, labels, eps, keep_logits, dw_, db_, lnw, lnb, dec_bias, word):
        R, Hd = rows.shape
        V = word.shape[0]
        rb = K.cast_bf16(rows.contiguous())
        wd, _ = BANK.linear(dw_)
        t_pre = torch.empty(R, Hd, device=rows.device, dtype=BF16)
        t_act = K.gemm_nt(rb, wd, bias=db_, aux=t_pre, act=1, out_dtype=F32)
        tb, _, mean, rstd = K.layernorm_fwd(t_act, lnw, lnb, eps)
        Eb, EbT = BANK.vocab(word)
        Vp = Eb.shape[0]
        bias_p = BANK.vector(dec_bias, Vp - V) if Vp > V else dec_bias.detach()        # zero-padded to the padded vocabulary
        labels = labels.contiguous().view(-1)
        ctx.fused = FUSED_MLM_CE
        logits = None
        if not ctx.fused or keep_logits or KEEP_MLM_LOGITS:
            logits = K.gemm_nt(tb, Eb, bias=bias_p, out_dtype=F32)
        if ctx.fused:
            stat, lse = K.mlm_ce_fwd(tb, Eb, bias_p, labels, V)
            saved = bias_p
        else:
            stat, lse = K.ce_fwd(logits, labels, C_valid=V)
            saved = logits
        ctx.save_for_backward(rb, t_pre, t_act, mean, rstd, tb, saved, labels, lse, stat, dw_, lnw, word)
        ctx.V = V
        lse_out = lse.clone()
        ctx.mark_non_differentiable(*([lse_out] + ([logits] if logits is not None else [])))
        return stat[0].clone(), lse_out, logits

    @staticmethod
    def backward(ctx, g, _glse, _gl):
        rb, t_pre, t_act, mean, rstd, tb, saved, labels, lse, stat, dw_, lnw, word = ctx.saved_tensors
        BANK.note_backward()
        V = ctx.V
        R, Hd = rb.shape
        dev = rb.device
        Eb, EbT = BANK.vocab(word)
        g1 = g.reshape(1).to(F32).contiguous()
        if ctx.fused:
            dl = K.mlm_ce_bwd(tb, Eb, saved, labels, lse, g1, stat, V)
        else:
            dl = K.ce_bwd(saved, labels, lse, g1, stat, C_valid=V, out_dtype=BF16)
        Vp = dl.shape[1]
        zeros = torch.zeros(Vp + 3 * Hd, device=dev, dtype=F32)          # one fill: decoder-bias gradient + the three head vectors below
        dbias, small = zeros[:Vp], zeros[Vp:]
        K.colsum_bf16(dl, dbias)
        # [R, Hd] from a 30528-long contraction: 36-72 output tiles, so the contraction is split (354 -> ~50 us)
        dt = K.gemm_nt_splitk(dl, EbT) if SPLIT_DECODER_DGRAD else K.gemm_nt(dl, EbT, out_dtype=F32)
        dword = torch.empty_like(word)
        dlnw, dlnb, dbd = small[:Hd], small[Hd:2 * Hd], small[2 * Hd:]
        dact, _ = K.layernorm_bwd(dt, t_act, mean, rstd, lnw, dlnw, dlnb)
        dpre = K.gelu_f32(t_pre.float(), dact)                        # small [R,Hd]: GELU' on the saved pre-activation
        dpre_b = K.cast_bf16(dpre)
        K.colsum_bf16(dpre_b, dbd)
        _, wdT = BANK.linear(dw_)
        drows = K.gemm_nt(dpre_b, wdT, out_dtype=F32)
        ddw = torch.empty_like(dw_)
        K.gemm_tn_grouped([(dl, tb, dword, Vp, Hd), (dpre_b, rb, ddw)])
        if TIE_WORD_GRAD:
            _TIED_DWORD[_tok(word)] = dword
            dword = None
        return drows, None, None, None, ddw, dbd, dlnw, dlnb, dbias[:V], dword


# ----------------------------------------------------------------------------- small differentiable ops (heads)

class LinearF32Fn(torch.autograd.Function):
    """y = x @ W^T + b in fp32 (projection heads, last layers of the MLP heads)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return K.linear_f32(x.contiguous(), w.detach().contiguous(), bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = K.linear_f32(dy, w, transB=True)                        # dy [M,N] @ W [N,K]
        dw = K.linear_f32(dy, x, transA=True, transB=True)           # dy^T [N,M] @ x [M,K]
        db = None
        if ctx.has_b:
            db = torch.zeros(w.shape[0], device=dy.device, dtype=F32)
            K.colsum_f32(dy, db)
        return dx, dw, db


class LinearBf16Fn(torch.autograd.Function):
    """y = x @ W^T + b on the bf16 MFMA GEMM (fp32 in/out rows, bf16 operands): first layers of the MLP heads
    (768 -> 1536 on 3B rows).  K % 64 == 0, N % 4 == 0."""

    @staticmethod
    def forward(ctx, x, w, b):
        xb = K.cast_bf16(x.contiguous())
        wb, _ = BANK.linear(w)
        ctx.save_for_backward(xb, w)
        return K.gemm_nt(xb, wb, bias=b, out_dtype=F32)

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        BANK.note_backward()
        dyb = K.cast_bf16(dy.contiguous())
        _, wT = BANK.linear(w)
        dx = K.gemm_nt(dyb, wT, out_dtype=F32)
        dw = torch.empty_like(w)
        K.gemm_tn_grouped([(dyb, xb, dw)])
        db = torch.zeros(w.shape[0], device=dy.device, dtype=F32)
        K.colsum_bf16(dyb, db)
        return dx, dw, db


class LayerNormF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        _, y, mean, rstd = K.layernorm_fwd(x, w, b, eps, want_bf16=False, want_f32=True)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        small = torch.zeros(2 * w.numel(), device=dy.device, dtype=F32)
        dw, db = small[:w.numel()], small[w.numel():]
        dx, _ = K.layernorm_bwd(dy.contiguous(), x, mean, rstd, w, dw, db)
        return dx, dw, db, None


class GeluF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.gelu_f32(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.gelu_f32(x, dy.contiguous())


class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.l2norm(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.l2norm(x, dy.contiguous())


class CrossEntropyFn(torch.autograd.Function):
    """mean over rows with label >= 0 of -log softmax(logits)[label]; fp32 logits [R,C]."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits, labels = logits.contiguous(), labels.contiguous()
        stat, lse = K.ce_fwd(logits, labels)
        ctx.save_for_backward(logits, labels, lse, stat)
        return stat[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, stat = ctx.saved_tensors
        return K.ce_bwd(logits, labels, lse, g.reshape(1).to(F32).contiguous(), stat), None


class GatherRowsFn(torch.autograd.Function):
    """out[r] = src[idx[r]] over the first dim (rows may be whole sequences); backward scatter-adds."""

    @staticmethod
    def forward(ctx, src, idx):
        src = src.contiguous()
        row_len = src[0].numel()
        ctx.save_for_backward(idx)
        ctx.shape = src.shape
        out, _ = K.gather_rows(src.view(src.shape[0], row_len), idx, row_len)
        return out.view((idx.numel(),) + src.shape[1:])

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        row_len = dy[0].numel()
        dst = K.scatter_rows(dy.contiguous().view(idx.numel(), row_len), idx, ctx.shape[0], row_len)
        return dst.view(ctx.shape), None
