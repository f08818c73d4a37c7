"""ctypes binding of libmico_hip.so (the C-ABI in include/mico_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  `lib()` raises if the shared object has not been
built (python __graft_entry__.py / make -C mico_amd/csrc) so a missing extension can never be mistaken for a pass.
"""
import ctypes as C
import os

# PyTorch-ROCm wheels bundle their own libamdhip64; it must be the HIP runtime already resident in the process when
# libmico_hip.so is dlopen'ed, otherwise the loader binds our library to /opt/rocm's copy and the process ends up with two
# runtimes (kernel launches from ours then fail with "no ROCm-capable device is detected").
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MICO_HIP_LIB") or os.path.join(_HERE, "libmico_hip.so")   # env override: kernel ablation builds

F16, BF16, F32 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_DERIV, ACT_MUL_AUX = 0, 1, 2, 3, 4

c_i64, c_int, c_f, c_vp = C.c_int64, C.c_int, C.c_float, C.c_void_p


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", c_vp), ("aux_out", c_vp), ("aux_in", c_vp), ("ldaux", c_i64), ("act", c_int),
        ("row_scale", c_vp), ("rows_per_scale", c_int), ("resid", c_vp), ("pos", c_vp), ("pos_rows", c_int),
        ("remap_group", c_int), ("remap_skip", c_int), ("remap_offset", c_int), ("alpha", c_f), ("accumulate", c_int),
        ("nseg", c_int), ("kseg", c_int), ("a_seg_off", c_int * 3), ("b_seg_off", c_int * 3),
        ("row_map", c_vp), ("rows_per_map", c_int),
        ("drop_p", c_f), ("drop_seed", C.c_uint), ("drop_site", c_int), ("colsum_out", c_vp),
        ("splitk_ws", c_vp), ("splitk_ws_bytes", c_i64), ("aux_tiled", c_int),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("B", c_int), ("H", c_int), ("Sq", c_int), ("Sk", c_int), ("hd", c_int),
        ("q_bs", c_i64), ("q_rs", c_i64), ("k_bs", c_i64), ("k_rs", c_i64), ("v_bs", c_i64), ("v_rs", c_i64),
        ("o_bs", c_i64), ("o_rs", c_i64), ("scale", c_f), ("mask", c_vp), ("mask_mode", c_int),
        ("drop_p", c_f), ("drop_seed", C.c_uint), ("drop_site", c_int), ("kv_batch_mod", c_int), ("batch0", c_int), ("dkv_accumulate", c_int),
    ]


class LnFwdParams(C.Structure):
    _fields_ = [
        ("x", c_vp), ("x_dtype", c_int), ("x_normalized", c_int), ("gamma", c_vp), ("beta", c_vp), ("y16", c_vp), ("y32", c_vp),
        ("mean", c_vp), ("rstd", c_vp), ("rows", c_i64), ("cols", c_int), ("eps", c_f), ("post_add", c_vp),
        ("post_rows_per_group", c_int), ("post_groups", c_int), ("y16_split", c_int), ("frame_map", c_vp), ("rows_per_frame", c_int),
        ("x_copy", c_vp), ("xhat16", c_vp), ("drop_p", c_f), ("drop_seed", C.c_uint), ("drop_site", c_int), ("valid_cols", c_int),
        ("q8", c_vp), ("ldq", c_i64), ("scales", c_vp),
    ]


class LnBwdParams(C.Structure):
    _fields_ = [
        ("dy", c_vp), ("dy_dtype", c_int), ("dy_scale", c_f), ("x", c_vp), ("x_dtype", c_int), ("x_normalized", c_int), ("gamma", c_vp),
        ("mean", c_vp), ("rstd", c_vp), ("dx_add", c_vp), ("dx32", c_vp), ("dx16", c_vp), ("scale16", c_f), ("dgamma", c_vp),
        ("dbeta", c_vp), ("grad_scale", c_f), ("ws", c_vp), ("rows", c_i64), ("cols", c_int), ("frame_map", c_vp), ("rows_per_frame", c_int),
        ("valid_cols", c_int), ("dx16_dst", c_vp), ("dx16_frame_scale", c_vp), ("dx16_drop_p", c_f), ("dx16_drop_seed", C.c_uint),
        ("dx16_drop_site", c_int),
    ]


# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/mico_hip.h one to one
PROTOTYPES = {
    "mico_version": [],
    "mico_last_error_string": [],
    "mico_struct_layout": [C.POINTER(c_int), c_int],
    "mico_gemm_last_kernel": [],
    "mico_gemm_aux_tiled_elems": [c_i64, c_i64, c_i64],
    "mico_gemm": [c_int, c_int, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int,
                  C.POINTER(GemmEpilogue), c_int, c_int, c_vp],
    "mico_quant_mx8": [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp, c_f, c_int, c_vp],
    "mico_gemm_mx8": [c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_int, C.POINTER(GemmEpilogue), c_int, c_vp],
    "mico_layernorm_fwd": [C.POINTER(LnFwdParams), c_int, c_vp],
    "mico_layernorm_bwd_nblk": [c_i64],
    "mico_layernorm_bwd": [C.POINTER(LnBwdParams), c_int, c_vp],
    "mico_attn_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(AttnParams), c_int, c_vp],
    "mico_attn_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(AttnParams), c_int, c_vp],
    "mico_rope": [c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp],
    "mico_im2row": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "mico_cast_f32_to_16": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_f, c_int, c_vp],
    "mico_cast_16_to_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_f, c_int, c_int, c_vp],
    "mico_gather_rows_cast": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_int, c_f, c_vp, c_int,
                              c_vp, c_int, c_vp],
    "mico_dropout": [c_vp, c_int, c_i64, c_int, c_i64, c_f, C.c_uint, c_int, c_vp],
    "mico_colsum": [c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_f, c_int, c_vp],
    "mico_cls_rows": [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp],
    "mico_add_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f, c_int, c_vp],
    "mico_swiglu_fwd": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_swiglu_fwd_f32": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "mico_swiglu_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_bert_embed_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp],
    "mico_embed_scatter_add": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_f, c_vp],
    "mico_itm_sample": [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "mico_token_mask": [c_vp, c_int, c_int, c_f, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "mico_win_attn_fwd": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f, c_int, c_vp],
    "mico_win_attn_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f, c_f, c_int, c_vp],
    "mico_patch_merge": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "mico_ce_fwd_bwd": [c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_int, c_f, c_f, c_vp, c_vp, c_vp, c_int, c_i64,
                        c_vp, c_f, c_int, c_vp],
    "mico_sgemm_small": [c_int, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f, c_f, c_vp, c_vp],
    "mico_gelu_f32": [c_vp, c_vp, c_i64, c_vp],
    "mico_gelu_bwd_f32": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "mico_gelu_16": [c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_gelu_bwd_16": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_cls_pool_fwd": [c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp],
    "mico_cls_pool_bwd": [c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp],
    "mico_dw_colfold": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_vp],
    "mico_pool_video_fwd": [c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    "mico_pool_video_bwd": [c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    "mico_l2norm_fwd": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_l2norm_bwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_image_preprocess": [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_vp],
    "mico_fbank_windows": [c_vp, c_int, c_int, c_vp, c_int, c_int, c_f, c_f, c_vp, c_vp],
    "mico_adamw_step": [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp],
    "mico_grads_finite": [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp],
    "mico_comm_unique_id": [c_vp],
    "mico_comm_init": [C.POINTER(c_vp), c_int, c_int, c_vp],
    "mico_comm_destroy": [c_vp],
    "mico_comm_allgather": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "mico_comm_allgather_packed": [c_vp, C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_i64, c_vp, c_vp, c_vp],
    "mico_comm_alltoallv": [c_vp, c_vp, C.POINTER(c_i64), c_vp, C.POINTER(c_i64), c_vp],
    "mico_comm_allreduce_f32": [c_vp, c_vp, c_i64, c_int, c_vp],
    "mico_comm_reduce_scatter_f32": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
}
_RESTYPES = {"mico_last_error_string": C.c_char_p, "mico_gemm_aux_tiled_elems": c_i64}

_lib = None


class MicoHipError(RuntimeError):
    pass


ABI_VERSION = 115   # = mico_version() of the library this binding matches (bumped with every signature / struct change)


def _check_struct_layout(l):
    """sizeof and every field offset of the ctypes parameter structs against the layout the library was compiled with."""
    n = l.mico_struct_layout(None, 0)
    buf = (c_int * n)()
    l.mico_struct_layout(buf, n)
    table, cur = [], []
    for v in buf:
        if v == -1:
            table.append(cur)
            cur = []
        else:
            cur.append(v)
    classes = (GemmEpilogue, AttnParams, LnFwdParams, LnBwdParams)
    if len(table) != len(classes):
        raise MicoHipError(f"mico_struct_layout reports {len(table)} structs, this binding mirrors {len(classes)}")
    for cls, (size, *offs) in zip(classes, table):
        mine = [getattr(cls, name).offset for name, _ in cls._fields_]
        if C.sizeof(cls) != size or mine != offs:
            raise MicoHipError(f"ctypes {cls.__name__} does not match the compiled struct (size {C.sizeof(cls)} vs {size}, offsets {mine} vs {offs})")


def lib():
    """Loads libmico_hip.so once; raises (never falls back) when it is missing, stale (ABI version) or laid out differently."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MicoHipError(
            f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mico_amd/csrc`).  mico_amd has no CPU/PyTorch fallback path.")
    l = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(l, name)   # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    # the C ABI is versioned: a stale .so under new Python (or the reverse) would misread positional arguments silently
    if l.mico_version() != ABI_VERSION:
        raise MicoHipError(f"{LIB_PATH} reports ABI version {l.mico_version()}, this binding was written for {ABI_VERSION}: rebuild it "
                           "(`make -C mico_amd/csrc`)")
    _check_struct_layout(l)
    _lib = l
    if os.environ.get("MICO_GEMM_VARIANT"):      # A/B runs of whole test files / benches without touching their code (probe build only)
        set_gemm_variant(int(os.environ["MICO_GEMM_VARIANT"]), l)
    return l


def set_gemm_variant(v, l=None):
    """mico_gemm_set_variant of the probe build (`make -C mico_amd/csrc variants`, MICO_HIP_LIB=tools/probes/bin/libmico_variants.so): the product
    library has no routing switch.  Returns the previous value."""
    l = l or lib()
    try:
        fn = l.mico_gemm_set_variant
    except AttributeError:
        raise MicoHipError(f"{LIB_PATH} has no kernel-routing switch (the product library routes by the problem alone): build the probe library "
                           "with `make -C mico_amd/csrc variants` and point MICO_HIP_LIB at tools/probes/bin/libmico_variants.so") from None
    fn.argtypes, fn.restype = [c_int], c_int
    return fn(int(v))


def check(rc, what):
    if rc != 0:
        msg = lib().mico_last_error_string()
        raise MicoHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
