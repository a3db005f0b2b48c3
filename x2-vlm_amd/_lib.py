"""ctypes binding of libx2vlm_hip.so (C ABI declared in include/x2vlm_hip.h).

The product path has no fallback: if the HIP library is missing or cannot be loaded this module
raises, and every op of the package fails with it.  Build it with __graft_entry__.build() or
x2-vlm_amd/csrc/build.sh (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# X2VLM_HIP_LIB: A/B a differently built library from the probes (never a fallback: the file must exist)
LIB_PATH = os.environ.get("X2VLM_HIP_LIB") or os.path.join(_HERE, "libx2vlm_hip.so")

P, I, L, F, U = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint


class AttnArgs(C.Structure):
    """Mirror of `struct AttnArgs` in csrc/attention.hip (same order, same types)."""
    _fields_ = [(n, P) for n in ("Q", "K", "V", "O", "dO", "Out", "dQ", "dK", "dV", "dS", "LSE", "Delta")] + \
               [(n, L) for n in ("q_bs", "q_rs", "k_bs", "k_rs", "v_bs", "v_rs", "o_bs", "o_rs",
                                 "dq_bs", "dq_rs", "dk_bs", "dk_rs", "dv_bs", "dv_rs", "do_bs", "do_rs")] + \
               [(n, I) for n in ("B", "Bkv", "H", "Lq", "Lk")] + [("scale", F)] + \
               [("bias", P), ("bias_ld", I), ("biasT", P), ("biasT_ld", I), ("mask", P), ("mask_ld", I),
                ("kv_idx", P), ("seq_off", P), ("seq_ids", P), ("ds_ld", I),
                ("drop_thr16", C.c_uint), ("drop_seed", C.c_uint), ("drop_scale", F), ("dbg", I), ("head_dim", I), ("drop_epoch", P),
                ("grid_nx", I), ("grid_ny", I), ("grid_nz", I), ("grid_map", I),                     # filled by the library: leave zero
                ("phase", I), ("ws", P), ("ws_floats", L), ("colsum_ws", P)]


# name -> argtypes (all return int: 0 ok, < 0 error; message via x2_last_error)
_SIGS = {
    "x2_gemm_nt": [P, P, P, I, I, I, I, I, I, P, P, P, I, P, I, I, I, U, U, F, P, P, P, P],
    "x2_gemm_nt_splitk": [P, P, P, I, I, I, I, I, I, I, P, L, P],
    "x2_gemm_nt_dgelu_colparts": [P, P, P, I, I, I, I, I, I, P, I, P, C.POINTER(I), P],
    "x2_gemm_tn_grouped": [P, I, I, I, P, L, P],
    "x2_attn_fwd": [C.POINTER(AttnArgs), P],
    "x2_attn_bwd": [C.POINTER(AttnArgs), P],
    "x2_layernorm_fwd": [P, P, P, P, P, P, P, I, I, F, I, U, U, F, P, P],
    "x2_layernorm_bwd": [P, I, P, P, P, P, P, P, P, P, P, P, I, I, I, U, U, F, U, U, F, P, P, I, I, P, P],
    "x2_colsum_bf16": [P, P, I, I, I, P, I, P],
    "x2_reduce_partials": [P, I, I, I, P, P, P, P],
    "x2_reduce_partials_multi": [P, I, P],
    "x2_cast_transpose_multi": [P, I, P],
    "x2_copy_f32_multi": [P, I, P],
    "x2_layerscale_finish": [P, I, P],
    "x2_rowscale_cast_colsum": [P, P, P, P, I, I, P, I, P],
    "x2_cast_bf16": [P, P, L, P],
    "x2_cast_transpose_bf16": [P, P, P, I, I, I, P],
    "x2_patchify": [P, P, I, I, I, P],
    "x2_assemble_tokens": [P, P, P, I, I, I, P],
    "x2_assemble_tokens_bwd": [P, P, P, I, I, I, P],
    "x2_pool_tokens": [P, P, I, I, I, I, P],
    "x2_relpos_bias": [P, P, P, P, P, I, I, I, I, F, P],
    "x2_relpos_bias_bwd": [P, P, P, P, I, I, I, I, I, P, I, P],
    "x2_embed_fwd": [P, P, P, P, P, I, I, I, P],
    "x2_embed_bwd": [P, P, P, P, P, I, I, I, P, P],
    "x2_gather_rows": [P, P, P, P, I, L, P],
    "x2_scatter_rows": [P, P, P, I, I, L, P],
    "x2_linear_f32": [P, P, P, P, P, F, I, I, I, L, L, L, L, L, I, I, P, P],
    "x2_l2norm": [P, P, P, I, I, I, P],
    "x2_ce_fwd": [P, L, P, I, I, P, P, P, P],
    "x2_ce_bwd": [P, L, P, P, P, P, F, I, I, P, P, L, P],
    "x2_mlm_ce_fwd": [P, P, P, P, I, I, I, I, I, I, P, P, P],
    "x2_ce_combine": [P, I, P, P, I, P, P, P, P],
    "x2_mlm_ce_bwd": [P, P, P, P, P, P, P, F, I, I, I, I, I, I, P, L, P],
    "x2_sample_negatives": [P, I, P, P, P, P],
    "x2_mask_tokens": [P, P, I, I, P, I, P, I, U, P, C.c_double, I, C.c_double, I, I, L, L, L, L, P, P, P, P],
    "x2_additive_mask": [P, P, I, I, I, F, P],
    "x2_kv_csr": [P, I, I, P, P, P],
    "x2_tail_index": [P, P, P, P, I, I, I, I, P, P, P, P, P],
    "x2_droppath_rows": [P, U, P, I, I, I, P, P],
    "x2_frame_mean": [P, P, P, P, P, I, I, I, I, I, P],
    "x2_gelu_f32": [P, P, P, L, P],
    "x2_grad_norm": [P, I, I, F, P, P, P],
    "x2_adamw_multi": [P, I, I, P, P, I, F, F, F, P, P],
    "x2_colsum_f32": [P, P, I, I, P],
}
# communicator entry points (csrc/comm.hip): explicit stream / event arguments, bound without the implicit stream of call()
_COMM_SIGS = {
    "x2_comm_unique_id": [P],
    "x2_comm_init": [P, I, I, C.POINTER(P)],
    "x2_comm_info": [P, C.POINTER(I), C.POINTER(I)],
    "x2_comm_allreduce_bucket": [P, P, L, I, I, P, P],
    "x2_comm_allgather": [P, P, P, L, I, P, P],
    "x2_comm_broadcast": [P, P, L, I, I, P, P],
    "x2_comm_destroy": [P],
}
EXPORTS = sorted(list(_SIGS) + list(_COMM_SIGS) + ["x2_last_error", "x2_abi_version", "x2_device_cus", "x2_tune", "x2_tune_get",
                                                         "x2_attn_bwd_one_pass"])

_lib = None


class X2HipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises X2HipError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise X2HipError("HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
                             % LIB_PATH)
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise X2HipError("cannot load %s: %s" % (LIB_PATH, e))
        for name, sig in list(_SIGS.items()) + list(_COMM_SIGS.items()):
            fn = getattr(h, name)
            fn.argtypes, fn.restype = sig, I
        h.x2_last_error.restype = C.c_char_p
        h.x2_abi_version.restype = I
        h.x2_device_cus.restype = I
        h.x2_tune.argtypes, h.x2_tune.restype = [I, I], I
        h.x2_tune_get.argtypes, h.x2_tune_get.restype = [I], I
        h.x2_attn_bwd_one_pass.argtypes, h.x2_attn_bwd_one_pass.restype = [C.POINTER(AttnArgs)], I
        for kv in filter(None, os.environ.get("X2_TUNE", "").split(",")):      # probes: "key=value,..." kernel-variant knobs (csrc/gemm.hip)
            k, v = kv.split("=")
            if h.x2_tune(int(k), int(v)) != 0:
                raise X2HipError("X2_TUNE=%s: %s" % (kv, h.x2_last_error().decode()))
        _lib = h
    return _lib


_get_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else None
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream():
    """hipStream_t (as int) of torch's current stream on the current device.  torch.cuda.current_stream() costs
    several microseconds of python per call (device-index normalisation, env lookups); a step makes ~4500 launches."""
    return _get_raw_stream(_get_device())


def call(name, *args):
    """Invoke an entry point on torch's current HIP stream; raise on a non-zero return."""
    h = _lib or lib()
    rc = getattr(h, name)(*args, _get_raw_stream(_get_device()))
    if rc != 0:
        raise X2HipError("%s failed (%d): %s" % (name, rc, h.x2_last_error().decode()))


def ptr(t):
    return None if t is None else t.data_ptr()
