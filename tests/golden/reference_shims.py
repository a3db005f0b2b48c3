"""Import harness for the REAL reference (/root/reference) inside the build container.

Only tests/golden/make_golden.py uses this; it never runs on the GPU box (the reference is not
there).  The reference pins timm==0.4.9 / transformers==4.12.5 / torchvision / apex, none of
which exist in this image (transformers here is 5.x), so the symbols it imports are provided
as minimal stand-ins with the pinned versions' semantics.  Nothing from the reference's own
source is copied: the stand-ins cover third-party names only.
"""
import json
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("X2VLM_REFERENCE", "/root/reference")


def _timm_drop_path(x, drop_prob=0.0, training=False):
    # timm 0.4.9 drop_path: per-sample mask, scaled by 1/keep
    if drop_prob == 0.0 or not training:
        return x
    keep = 1.0 - drop_prob
    mask = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    return x.div(keep) * mask.floor_()


def install():
    """Install stand-ins into sys.modules, then make `models.*` of the reference importable."""
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be imported before the torchvision stand-in)
    import transformers.file_utils as fu
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("timm"); mod("timm.models")
    mod("timm.models.layers", drop_path=_timm_drop_path,
        to_2tuple=lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x),
        trunc_normal_=lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0:
            nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b))
    mod("timm.models.registry", register_model=lambda f: f)
    mod("torchvision"); mod("torchvision.ops")
    mod("torchvision.ops.boxes",
        box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))

    # transformers 4.12.5 names that moved or vanished in 5.x
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    mu.get_parameter_dtype = lambda m: next(m.parameters()).dtype
    mu.PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    ident = lambda *a, **k: (lambda f: f)
    for name in ("replace_return_docstrings", "add_code_sample_docstrings", "add_start_docstrings",
                 "add_start_docstrings_to_model_forward"):
        setattr(fu, name, ident)

    def init_weights_4_12_5(self):
        # 4.12.5: apply(_init_weights) then tie output embeddings to input embeddings
        self.apply(self._init_weights)
        out = self.get_output_embeddings() if hasattr(self, "get_output_embeddings") else None
        if out is not None and getattr(self.config, "tie_word_embeddings", True):
            base = getattr(self, self.base_model_prefix, self)
            out.weight = base.get_input_embeddings().weight
    mu.PreTrainedModel.init_weights = init_weights_4_12_5

    # the reference's own `utils` / `dataset` packages pull ruamel, cv2, pycocotools ... ;
    # models/xvlm.py only needs read_json and build_tokenizer(...).pad_token_id
    def _read_json(p):
        with open(p) as f:
            return json.load(f)
    mod("utils", read_json=_read_json)
    mod("dataset", build_tokenizer=lambda path, *a, **k: types.SimpleNamespace(pad_token_id=0))

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def ensure_process_group(port=29541):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("gloo", rank=0, world_size=1)
