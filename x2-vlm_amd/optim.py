"""Optimizer for the pre-training step: the reference's parameter groups (optim.py:26-104) on a fused multi-tensor
AdamW kernel with the HuggingFace AdamW update rule (transformers==4.12.5, eps 1e-8, betas (0.9, 0.98),
correct_bias=True) and the global-norm gradient clip folded in (accelerators/apex_ddp_accelerator.py:99-102).

One launch for the gradient norm, one for the update of all ~570 tensors (254.76 M parameters, 28 B/parameter of HBM
traffic) instead of HF's per-parameter Python loop.  No host sync unless the caller asks for the norm as a float.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import call, ptr

CHUNK = 16384
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight")

_REC = np.dtype([("p", np.int64), ("g", np.int64), ("m", np.int64), ("v", np.int64), ("n", np.int64), ("group", np.int32),
                 ("blk0", np.int32), ("ss", np.float32), ("pad", np.int32)])       # == struct OptTensor (csrc/optim.hip)


class FusedAdamW(torch.optim.Optimizer):
    """State per parameter as transformers' AdamW keeps it: `step` (int), `exp_avg`, `exp_avg_sq` - so a state dict saved by
    the reference's optimizer loads here and vice versa, and a parameter that receives no gradient in some iterations
    (bbox_head on image-only steps) keeps its own bias-correction step."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._reset_tables()
        self.max_grad_norm = 0.0
        self._norm = None            # device tensor [2]: (total norm, clip coefficient) of the current gradients

    def _reset_tables(self):
        assert len(self.param_groups) <= 16
        self._plist = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"]]
        self._static = None

    def add_param_group(self, group):
        super().add_param_group(group)
        if hasattr(self, "_plist"):
            self._reset_tables()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # replaces the state tensors: the pointer table must be rebuilt
        self._reset_tables()

    def _tables(self, stepping=False):
        dev = self._plist[0][1].device
        if self._static is None:
            rec = np.zeros(len(self._plist), dtype=_REC)
            blk = 0
            for i, (gi, p) in enumerate(self._plist):
                assert p.is_contiguous() and p.dtype == torch.float32
                rec[i] = (0, 0, 0, 0, p.numel(), gi, blk, 0.0, 0)
                blk += (p.numel() + CHUNK - 1) // CHUNK
            cuda = dev.type == "cuda"
            # two staging slots: the host rewrites one while the asynchronous H2D copy of the other may still be queued
            slots = [(torch.empty(rec.nbytes, dtype=torch.uint8).pin_memory() if cuda else None,
                      torch.empty(rec.nbytes, dtype=torch.uint8, device=dev), torch.cuda.Event() if cuda else None) for _ in range(2)]
            self._static = (rec, blk, slots, torch.empty(blk, dtype=torch.float32, device=dev), [0])
        rec, nblk, slots, partial, turn = self._static
        b1, b2 = self.param_groups[0]["betas"]
        P, G, M, V, SS = [], [], [], [], []
        for _gi, p in self._plist:          # every pointer is re-read: gradients move every step (per-layer arenas), and
            st = self.state[p]              # load_state_dict / .to() may have replaced parameters or moments
            if "exp_avg" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            g = p.grad
            if g is not None:
                assert g.is_contiguous() and g.dtype == torch.float32
                if stepping:
                    st["step"] = int(st.get("step", 0)) + 1
            t = max(int(st.get("step", 0)), 1)
            P.append(p.data_ptr()); G.append(g.data_ptr() if g is not None else 0)
            M.append(st["exp_avg"].data_ptr()); V.append(st["exp_avg_sq"].data_ptr())
            SS.append((1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t))
        rec["p"], rec["g"], rec["m"], rec["v"], rec["ss"] = P, G, M, V, SS
        raw = torch.from_numpy(rec.view(np.uint8))
        pinned, dtab, ev = slots[turn[0]]
        turn[0] ^= 1
        if pinned is not None:
            ev.synchronize()                # the copy that last read this pinned slot has completed (two tables ago)
            pinned.copy_(raw)
            dtab.copy_(pinned, non_blocking=True)
            ev.record()
        else:
            dtab.copy_(raw)
        return dtab, len(rec), nblk, partial

    @torch.no_grad()
    def grad_norm(self, max_norm=0.0):
        """Global L2 norm of all gradients (device tensor [norm, clip coefficient]); the coefficient is applied
        inside the next step() instead of rewriting the gradients (clip_grad_norm_ semantics)."""
        dtab, nt, nblk, partial = self._tables()
        out = torch.empty(2, dtype=torch.float32, device=dtab.device)
        call("x2_grad_norm", ptr(dtab), nt, nblk, float(max_norm), ptr(partial), ptr(out))
        self._norm = out
        self.max_grad_norm = max_norm
        return out

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        dtab, nt, nblk, _ = self._tables(stepping=True)
        ng = len(self.param_groups)
        lr = (C.c_float * ng)(*[g["lr"] for g in self.param_groups])
        wd = (C.c_float * ng)(*[g["weight_decay"] for g in self.param_groups])
        b1, b2 = self.param_groups[0]["betas"]
        call("x2_adamw_multi", ptr(dtab), nt, nblk, lr, wd, ng, b1, b2, self.param_groups[0]["eps"], ptr(self._norm))
        # the kernel updated the parameters behind torch's back: bump their version counters and drop the engine's bf16
        # weight copies
        ps = [p for _gi, p in self._plist]
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
        from . import engine
        engine.BANK.invalidate()
        self._norm = None


def create_optimizer(args, model):
    """optim.py:26-104: {decay, no-decay} x {lr, lr * lr_mult for model.init_params} (+ vision/text/cross lr pairs)."""
    get = (lambda k, d=None: args.get(k, d)) if isinstance(args, dict) else (lambda k, d=None: getattr(args, k, d))
    lr, wd, lr_mult = get("lr"), get("weight_decay"), get("lr_mult", 1)
    groups = [{"params": [], "weight_decay": wd, "lr": lr}, {"params": [], "weight_decay": 0.0, "lr": lr},
              {"params": [], "weight_decay": wd, "lr": lr * lr_mult}, {"params": [], "weight_decay": 0.0, "lr": lr * lr_mult}]
    special = get("vision_lr") is not None
    if special:
        vlr, tlr = get("vision_lr"), get("text_lr")
        clr = get("cross_lr", tlr)
        for x in (vlr, tlr, clr):
            groups += [{"params": [], "weight_decay": wd, "lr": x}, {"params": [], "weight_decay": 0.0, "lr": x}]
    large = set(getattr(model, "init_params", []))
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        nd = 1 if any(k in n for k in NO_DECAY) else 0
        if special and "vision_encoder" in n:
            gi = 4 + nd
        elif special and "text_encoder" in n:
            gi = 6 + nd
        elif special and "cross_encoder" in n:
            gi = 8 + nd
        elif n in large:
            gi = 2 + nd
        else:
            gi = nd
        groups[gi]["params"].append(p)
    return FusedAdamW(groups, lr=lr, eps=1e-8, betas=(0.9, 0.98))
