"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/x2vlm_hip.h
declares (and nothing is bound by the Python side that the header does not declare), argument checks
fail cleanly without a GPU, the host mirror has the reference's state-dict keys."""
import ctypes
import importlib
import os
import re
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "x2vlm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(x2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = importlib.import_module("x2-vlm_amd._lib")
    assert os.path.exists(lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    h = ctypes.CDLL(lib.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(h, name), "include/x2vlm_hip.h declares %s but libx2vlm_hip.so does not export it" % name
    assert sorted(lib.EXPORTS) == declared, set(lib.EXPORTS) ^ set(declared)
    assert lib.lib().x2_abi_version() == 14


def test_attn_args_struct_matches_header_layout():
    lib = importlib.import_module("x2-vlm_amd._lib")
    # 12 pointers + 16 longs + 5 ints + float + 3 x (pointer, int, pad) + 3 pointers + int (+pad)
    # ds_ld, 3 dropout words, dbg, head_dim, epoch pointer, 4 library-owned grid words, phase (+pad)
    assert ctypes.sizeof(lib.AttnArgs) == 12 * 8 + 16 * 8 + 6 * 4 + 3 * 16 + 3 * 8 + 24 + 8 + 16 + 8 + 16 + 8
    assert lib.AttnArgs.ws.offset == 376 and lib.AttnArgs.ws_floats.offset == 384 and lib.AttnArgs.colsum_ws.offset == 392
    assert lib.AttnArgs.grid_nx.offset == 352 and lib.AttnArgs.grid_map.offset == 364 and lib.AttnArgs.phase.offset == 368
    assert lib.AttnArgs.bias.offset == 248 and lib.AttnArgs.kv_idx.offset == 296 and lib.AttnArgs.ds_ld.offset == 320
    assert lib.AttnArgs.drop_thr16.offset == 324 and lib.AttnArgs.dbg.offset == 336 and lib.AttnArgs.head_dim.offset == 340 and lib.AttnArgs.drop_epoch.offset == 344


def test_attention_backward_form_rules():
    """x2_attn_bwd_one_pass is a host-side query (no launch, pointers only tested for null): which geometries the one-pass backward
    kernels take - the BEiT-2 blocks (one sequence per K/V batch, 64 < N <= 208, no probability dropout), the cross-attention of rows
    sharing K/V (Lq <= 128 per sequence, Lk <= 208), long self-attention given a workspace (208 < N: up to 640 queries x 768 keys) - and that
    x2_tune(14, 1) switches all of them off."""
    K = importlib.import_module("x2-vlm_amd.kernels")
    h = importlib.import_module("x2-vlm_amd._lib").lib()
    own = {(197, 197): 1, (208, 208): 1, (65, 65): 1, (150, 90): 1, (64, 64): 0, (30, 30): 0, (209, 209): 0, (577, 577): 0, (197, 60): 0}
    for (lq, lk), form in own.items():
        assert K.attn_bwd_form(lq, lk) == form, (lq, lk)
        assert K.attn_bwd_form(lq, lk, dropout=True) == 0
    shared = {(30, 197): 2, (8, 5): 2, (128, 208): 2, (60, 197): 2, (129, 197): 0, (30, 577): 0, (30, 209): 0}
    for (lq, lk), form in shared.items():
        assert K.attn_bwd_form(lq, lk, shared_kv=True) == form, (lq, lk)
        assert K.attn_bwd_form(lq, lk, shared_kv=True, dropout=True) == form          # the shared-K/V kernel regenerates the masks
    # long self-attention (X2VLM-large, N = 577): one pass when the caller hands over the workspace of the dQ partials, up to 640 queries x 768 keys
    long_ = {(577, 577): 3, (209, 209): 3, (640, 768): 3, (300, 215): 3, (641, 577): 0, (577, 769): 0, (577, 208): 0, (208, 577): 0}
    for (lq, lk), form in long_.items():
        assert K.attn_bwd_form(lq, lk, workspace=True) == form, (lq, lk)
        assert K.attn_bwd_form(lq, lk) == 0 and K.attn_bwd_form(lq, lk, workspace=True, dropout=True) == 0
        assert K.attn_bwd_form(lq, lk, workspace=True, shared_kv=True) == 0
    try:
        assert h.x2_tune(14, 1) == 0
        assert all(K.attn_bwd_form(lq, lk) == 0 for lq, lk in own) and all(K.attn_bwd_form(lq, lk, shared_kv=True) == 0 for lq, lk in shared)
        assert all(K.attn_bwd_form(lq, lk, workspace=True) == 0 for lq, lk in long_)
        assert h.x2_tune(14, 3) == 0               # only the long form off
        assert K.attn_bwd_form(577, 577, workspace=True) == 0 and K.attn_bwd_form(197, 197) == 1
    finally:
        h.x2_tune(14, 0)
    assert h.x2_tune_get(14) == 0


def test_argument_checks_fail_loudly_without_launching():
    lib = importlib.import_module("x2-vlm_amd._lib")
    h = lib.lib()
    rc = h.x2_gemm_nt(None, None, None, 128, 128, 100, 100, 100, 128, None, None, None, 0, None, 0, 0, 0, 0, 0, 1.0, None, None, None, None)
    assert rc == -1 and b"multiple of 64" in h.x2_last_error()
    rc = h.x2_layernorm_fwd(None, None, None, None, None, None, None, 4, 770, 1e-6, 0, 0, 0, 1.0, None, None)
    assert rc == -1 and b"x2_layernorm_fwd" in h.x2_last_error()
    rc = h.x2_sample_negatives(None, 5000, None, None, None, None)
    assert rc == -1
    one = ctypes.c_void_p(16)                 # any non-null address: the checks below return before anything is dereferenced
    rc = h.x2_gemm_nt_splitk(one, one, one, 128, 128, 100, 104, 104, 128, 0, None, 0, None)
    assert rc == -1 and b"x2_gemm_nt_splitk" in h.x2_last_error()
    rc = h.x2_mlm_ce_fwd(one, one, None, one, 8, 1000, 1000, 128, 128, 128, one, one, None)      # vocabulary rows not padded to 64
    assert rc == -1 and b"x2_mlm_ce_fwd" in h.x2_last_error()
    rc = h.x2_mlm_ce_bwd(one, one, None, one, one, one, one, 1.0, 8, 1024, 1000, 128, 128, 128, one, 1000, None)   # ldd < Vp
    assert rc == -1 and b"x2_mlm_ce_bwd" in h.x2_last_error()
    rc = h.x2_ce_combine(None, 16, None, None, 8, None, None, None, None)
    assert rc == -1 and b"x2_ce_combine" in h.x2_last_error()


def test_communicator_entry_points_check_arguments_and_find_rccl():
    """x2_comm_*: librccl.so.1 is resolved at first use (no link-time dependency); bad handles fail cleanly."""
    lib = importlib.import_module("x2-vlm_amd._lib")
    comm = importlib.import_module("x2-vlm_amd.comm")
    h = lib.lib()
    assert h.x2_comm_destroy(None) == -1 and b"x2_comm_destroy" in h.x2_last_error()
    assert h.x2_comm_allreduce_bucket(None, None, 0, 0, 1, None, None) == -1
    assert h.x2_comm_init(None, 0, 1, None) == -1
    assert len(comm.X2Comm.unique_id()) == 128               # RCCL loaded and answered (needs no GPU)


def test_missing_library_raises(monkeypatch):
    lib = importlib.import_module("x2-vlm_amd._lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libx2vlm_hip.so")
    with pytest.raises(lib.X2HipError):
        lib.lib()


def test_state_dict_keys_match_reference(synthetic):
    """Key names + shapes of the host mirror == the reference's state dict (golden fixture lists every
    parameter name the real reference model had)."""
    from cases import CASES, model_config
    mp = importlib.import_module("x2-vlm_amd.model_pretrain")
    for case in ("tiny", "tiny_video", "base_shallow"):
        model = mp.XVLM(config=model_config(case, tempfile.mkdtemp()), load_vision_params=False, load_text_params=False)
        gold = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
        ref = sorted(k[len("gradnorm/"):] for k in gold.files if k.startswith("gradnorm/"))
        mine = sorted(n for n, _ in model.named_parameters())
        assert mine == ref
        sd = model.state_dict()
        assert "text_encoder.cls.predictions.decoder.weight" in sd            # tied, still serialised like HF does
        assert sd["text_encoder.cls.predictions.decoder.weight"].data_ptr() == \
            sd["text_encoder.bert.embeddings.word_embeddings.weight"].data_ptr()
        assert "vision_encoder.blocks.0.attn.relative_position_index" in sd and "text_encoder.bert.embeddings.position_ids" in sd
        assert sorted(model.init_params) == sorted(n for n in mine if n.split(".")[0] in
                                                    ("temp", "vision_proj", "text_proj", "itm_head", "bbox_head",
                                                     "absolute_frame_pos_embed"))


def test_relative_position_index_matches_oracle():
    from oracle import x2vlm_oracle as O
    beit2 = importlib.import_module("x2-vlm_amd.beit2")
    for g in (2, 14, 24):
        assert torch.equal(beit2.relative_position_index(g, g), O.relative_position_index(g))


def test_synthetic_generators_are_deterministic(synthetic):
    a = synthetic.synth_batch(5, 4, 12, 32, 512, 3, ragged=True)
    b = synthetic.synth_batch(5, 4, 12, 32, 512, 3, ragged=True)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert (a["masked_ids"] == -100).any() and (a["text_atts"] == 0).any()
    sel = a["masked_ids"] != -100
    assert torch.equal(torch.gather(a["text_ids"], 1, a["masked_pos"])[sel], a["masked_ids"][sel])
    ineg, tneg = synthetic.synth_negatives(3, 8)
    assert all(i != j for j, i in enumerate(ineg)) and all(i != j for j, i in enumerate(tneg))
