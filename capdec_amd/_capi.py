"""ctypes binding of libcapdec_hip.so (include/capdec.h).  Nothing here computes: every
call goes to the HIP library, and loading fails loudly when it is missing (there is no CPU
fallback in the product path)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch first: it brings its own libamdhip64.so.7; loading ours afterwards makes the dynamic
# linker reuse that one runtime instead of mapping /opt/rocm's copy next to it (two HIP
# runtimes in one process lose the device).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcapdec_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class Gpt2Layer(C.Structure):
    _fields_ = [(n, c_float_p) for n in (
        "ln_1_w", "ln_1_b", "c_attn_w", "c_attn_b", "c_proj_w", "c_proj_b", "ln_2_w", "ln_2_b",
        "c_fc_w", "c_fc_b", "mlp_c_proj_w", "mlp_c_proj_b")]


class Gpt2Weights(C.Structure):
    _fields_ = [("n_layer", C.c_int), ("n_head", C.c_int), ("n_embd", C.c_int), ("vocab", C.c_int),
                ("n_pos", C.c_int), ("ln_eps", C.c_float), ("wte", c_float_p), ("wpe", c_float_p),
                ("layers", C.POINTER(Gpt2Layer)), ("ln_f_w", c_float_p), ("ln_f_b", c_float_p)]


class TMapperLayer(C.Structure):
    _fields_ = [(n, c_float_p) for n in (
        "norm1_w", "norm1_b", "to_queries_w", "to_keys_values_w", "project_w", "project_b",
        "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class TMapperWeights(C.Structure):
    _fields_ = [("prefix_dim", C.c_int), ("prefix_length", C.c_int), ("clip_length", C.c_int),
                ("num_layers", C.c_int), ("num_heads", C.c_int), ("d", C.c_int), ("mlp_hidden", C.c_int),
                ("linear_w", c_float_p), ("linear_b", c_float_p), ("prefix_const", c_float_p),
                ("layers", C.POINTER(TMapperLayer))]


class ClipBlock(C.Structure):
    _fields_ = [(n, c_float_p) for n in (
        "ln_1_w", "ln_1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln_2_w", "ln_2_b",
        "c_fc_w", "c_fc_b", "c_proj_w", "c_proj_b")]


class ClipTextWeights(C.Structure):
    _fields_ = [("context_length", C.c_int), ("vocab", C.c_int), ("width", C.c_int), ("heads", C.c_int),
                ("layers", C.c_int), ("embed_dim", C.c_int), ("token_embedding", c_float_p),
                ("positional_embedding", c_float_p), ("blocks", C.POINTER(ClipBlock)), ("ln_final_w", c_float_p),
                ("ln_final_b", c_float_p), ("text_projection", c_float_p)]


class ClipVisionWeights(C.Structure):
    _fields_ = [("image_size", C.c_int), ("patch", C.c_int), ("width", C.c_int), ("heads", C.c_int),
                ("layers", C.c_int), ("embed_dim", C.c_int), ("conv1_w", c_float_p), ("class_embedding", c_float_p),
                ("positional_embedding", c_float_p), ("ln_pre_w", c_float_p), ("ln_pre_b", c_float_p),
                ("blocks", C.POINTER(ClipBlock)), ("ln_post_w", c_float_p), ("ln_post_b", c_float_p),
                ("proj", c_float_p)]


class ConvBn(C.Structure):
    _fields_ = [("w", c_float_p), ("bn_w", c_float_p), ("bn_b", c_float_p), ("bn_mean", c_float_p),
                ("bn_var", c_float_p), ("cin", C.c_int), ("cout", C.c_int), ("k", C.c_int)]


class ClipResNetWeights(C.Structure):
    _fields_ = [("image_size", C.c_int), ("width", C.c_int), ("embed_dim", C.c_int), ("layers", C.c_int * 4),
                ("stem", C.POINTER(ConvBn)), ("blocks", C.POINTER(ConvBn)), ("positional_embedding", c_float_p),
                ("q_w", c_float_p), ("q_b", c_float_p), ("k_w", c_float_p), ("k_b", c_float_p),
                ("v_w", c_float_p), ("v_b", c_float_p), ("c_w", c_float_p), ("c_b", c_float_p)]


#: every symbol include/capdec.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
ABI_VERSION = 5          # include/capdec.h: CAPDEC_ABI_VERSION
SIGNATURES = {
    "capdec_abi_version": (C.c_int, []),
    "capdec_build_id": (C.c_char_p, []),
    "capdec_last_error": (C.c_char_p, []),
    "capdec_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "capdec_destroy": (None, [_VP]),
    "capdec_set_stream": (C.c_int, [_VP, _VP]),
    "capdec_use_own_stream": (C.c_int, [_VP]),
    "capdec_synchronize": (C.c_int, [_VP]),
    "capdec_set_gemm_mode": (C.c_int, [_VP, C.c_int]),
    "capdec_get_gemm_mode": (C.c_int, [_VP]),
    "capdec_set_batch_invariant": (C.c_int, [_VP, C.c_int]),
    "capdec_cross_entropy": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "capdec_decode_counters": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "capdec_decode_second_pass_rows": (C.c_int, [_VP, C.POINTER(C.c_longlong)]),
    "capdec_train_step": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_int, C.POINTER(C.c_float)]),
    "capdec_train_get": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.c_size_t]),
    "capdec_train_reset": (C.c_int, [_VP]),
    "capdec_train_set_scope": (C.c_int, [_VP, C.c_int]),
    "capdec_train_loss": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "capdec_train_set_dropout": (C.c_int, [_VP, C.c_float, C.c_uint64]),
    "capdec_train_set_dropout_masks": (C.c_int, [_VP, _VP, C.c_size_t]),
    "capdec_train_get_dropout_masks": (C.c_int, [_VP, _VP, C.c_size_t]),
    "capdec_set_kv_budget": (C.c_int, [_VP, C.c_size_t]),
    "capdec_malloc": (C.c_int, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "capdec_free": (C.c_int, [_VP, _VP]),
    "capdec_memcpy_h2d": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "capdec_memcpy_d2h": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "capdec_load_gpt2": (C.c_int, [_VP, C.POINTER(Gpt2Weights)]),
    "capdec_load_mapper_mlp": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "capdec_load_mapper_transformer": (C.c_int, [_VP, C.POINTER(TMapperWeights)]),
    "capdec_load_clip_text": (C.c_int, [_VP, C.POINTER(ClipTextWeights)]),
    "capdec_load_clip_vision": (C.c_int, [_VP, C.POINTER(ClipVisionWeights)]),
    "capdec_load_clip_resnet": (C.c_int, [_VP, C.POINTER(ClipResNetWeights)]),
    "capdec_clip_encode_text": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "capdec_clip_encode_image": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "capdec_normalize_prefix": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "capdec_noise_inject": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, _VP, C.c_int, C.c_int, C.c_uint64,
                                      _VP, _VP, _VP]),
    "capdec_mapper_forward": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "capdec_gpt2_logits": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "capdec_wte_lookup": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "capdec_decode_greedy": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "capdec_decode_greedy_forced": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP]),
    "capdec_decode_beam": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _VP, _VP,
                                     _VP, _VP]),
    "capdec_gemm_f32": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP,
                                  _VP, C.c_int, C.c_int]),
    "capdec_preprocess_images": (C.c_int, [_VP, _VP, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, _VP]),
    "capdec_comm_unique_id": (C.c_int, [C.c_char_p]),
    "capdec_comm_init": (C.c_int, [_VP, C.c_int, C.c_int, C.c_char_p]),
    "capdec_comm_destroy": (C.c_int, [_VP]),
    "capdec_comm_info": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "capdec_shard_bounds": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "capdec_gather_rows": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "capdec_gather_ids": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP]),
    "capdec_decode_stats": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "capdec_set_compact": (C.c_int, [_VP, C.c_int]),
    "capdec_decode_step_rows": (C.c_int, [_VP, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "capdec_timer_start": (C.c_int, [_VP]),
    "capdec_timer_stop_ms": (C.c_int, [_VP, c_float_p]),
    "capdec_profile_enable": (C.c_int, [_VP, C.c_int]),
    "capdec_profile_reset": (C.c_int, [_VP]),
    "capdec_profile_get": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_char_p), c_float_p,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

#: exported by measurement builds only (libcapdec_hip_measure.so, -DCAPDEC_MEASURE): bound when present
MEASURE_SIGNATURES = {
    "capdec_set_debug_diverge": (C.c_int, [_VP, C.c_int]),
}
MEASURE_LIB_PATH = os.path.join(_HERE, "lib", "libcapdec_hip_measure.so")

_lib: Optional[C.CDLL] = None


class CapdecError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libcapdec_hip.so and bind every symbol of the header.  Raises if the library or
    any symbol is missing -- there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CAPDEC_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise CapdecError(
            f"HIP extension not built: {p} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m capdec_amd.build`). capdec_amd has no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in MEASURE_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.capdec_abi_version() != ABI_VERSION:
        raise CapdecError("libcapdec_hip.so ABI version mismatch")
    if (path is None or os.path.abspath(p) == os.path.abspath(MEASURE_LIB_PATH)) and os.environ.get("CAPDEC_SKIP_BUILD_ID_CHECK") != "1":
        # stale-library guard: the id compiled into the .so must match the sources lying next to it
        try:
            from .build import source_hash
            want = source_hash()
        except OSError:
            want = None                     # sources not shipped with the package: nothing to compare against
        got = lib.capdec_build_id().decode()
        if want is not None and got != want:
            raise CapdecError(f"stale HIP extension: {p} was built from other sources (build id {got}, tree {want}). "
                              "Run `python -m capdec_amd.build`.")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "", lib: Optional[C.CDLL] = None) -> None:
    if rc != 0:
        msg = (lib or load_library()).capdec_last_error()
        raise CapdecError(f"{what}: {msg.decode() if msg else 'unknown error'}")
