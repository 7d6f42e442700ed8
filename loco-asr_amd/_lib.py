"""ctypes binding of the C ABI declared in include/loco_asr.h.

The HIP library is the product: if ``libloco_asr.so`` is missing or does not load, importing the
encoder fails loudly here -- there is no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LOCO_ASR_LIB: load another build of the SAME library instead (A/B and regression probes under tools/); never a fallback
LIB_PATH = os.environ.get("LOCO_ASR_LIB") or os.path.join(_HERE, "libloco_asr.so")
ABI_VERSION = 1


class LocoError(RuntimeError):
    """A C-ABI call returned a negative LOCO_E_* code."""


class LocoConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32),
                ("layers", C.c_int32), ("conv_dim", C.c_int32), ("pos_conv_kernel", C.c_int32),
                ("pos_conv_groups", C.c_int32), ("rel_max", C.c_int32), ("ln_eps", C.c_float)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_vp, _i32, _i64, _sz, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float

# name -> (restype, argtypes); exactly the symbols of include/loco_asr.h
SIGNATURES = {
    "loco_abi_version": (C.c_int, []),
    "loco_last_error": (C.c_char_p, []),
    "loco_default_config": (None, [C.POINTER(LocoConfig)]),
    "loco_create": (_vp, [C.POINTER(LocoConfig)]),
    "loco_destroy": (None, [_vp]),
    "loco_set_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), C.c_int]),
    "loco_missing_weights": (C.c_int, [_vp, C.c_char_p, _sz]),
    "loco_finalize_weights": (C.c_int, [_vp, _vp]),
    "loco_output_frames": (_i64, [_i64]),
    "loco_workspace_bytes": (_sz, [_vp, _i32, _i64]),
    "loco_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp]),
    "loco_forward_status": (C.c_int, [_vp, C.c_char_p, _sz]),
    "loco_forward_range": (C.c_int, [_vp, _i32, C.POINTER(_f), C.POINTER(_i32), C.c_char_p, _sz]),
    "loco_set_range_policy": (C.c_int, [_vp, C.c_int]),
    "loco_forward_checked": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp, C.POINTER(_i32)]),
    "loco_status_bytes": (_sz, []),
    "loco_forward_async": (C.c_int, [_vp, C.c_int, _vp, _vp, _i32, _i64, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp, _vp]),
    "loco_max_pack_clips": (C.c_int, []),
    "loco_forward_packed": (C.c_int, [_vp, C.c_int, _vp, _vp, C.POINTER(_i64), _i32, _i64, C.POINTER(_i64), _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp,
                                      _vp]),
    "loco_status_check": (C.c_int, [_vp, C.c_char_p, _sz]),
    "loco_status_range": (C.c_int, [_vp, _i32, C.POINTER(_f), C.POINTER(_i32), C.c_char_p, _sz]),
    "loco_flac_last_error": (C.c_char_p, []),
    "loco_flac_info": (C.c_int, [_vp, _sz, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "loco_flac_decode": (C.c_int, [_vp, _sz, _vp, _vp, _i64, C.POINTER(_i64), _i32]),
    "loco_resample_design": (C.c_int, [_i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "loco_resample_length": (_i64, [_i64, _i32, _i32]),
    "loco_op_resample": (C.c_int, [_vp, _i32, _i64, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _i64, _vp]),
    "loco_normalize_scratch_bytes": (_sz, [_i32]),
    "loco_op_normalize_waveform": (C.c_int, [_vp, _vp, _i32, _i64, C.c_float, _vp, _vp, _sz, _vp]),
    "loco_text_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "loco_text_max_positions": (C.c_int, [_vp]),
    "loco_forward_text": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp]),
    "loco_precision_name": (C.c_char_p, [C.c_int]),
    "loco_forward_text_async": (C.c_int, [_vp, C.c_int, _vp, _vp, _i32, _i32, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp, _vp]),
    "loco_set_precision": (C.c_int, [_vp, C.c_int]),
    "loco_get_precision": (C.c_int, [_vp]),
    "loco_set_streams": (C.c_int, [_vp, C.c_int]),
    "loco_set_taps": (C.c_int, [_vp, _vp, _vp, _vp]),
    "loco_set_profiling": (C.c_int, [_vp, C.c_int]),
    "loco_set_profiling_filter": (C.c_int, [_vp, C.c_char_p]),
    "loco_profile_reset": (C.c_int, [_vp]),
    "loco_profile_read": (C.c_int, [_vp, C.POINTER(KernelStat), C.c_int]),
    "loco_op_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f, _vp]),
    "loco_op_gemm": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                               _i64, _i64, _i64, _i64, _vp]),
    "loco_conv0_scratch_bytes": (_sz, [_i32]),
    "loco_op_conv0_gn_gelu": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "loco_op_frame_counts": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "loco_op_pos_conv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "loco_op_attention": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "loco_op_split_f16": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "loco_gemm_splitk_bytes": (_sz, []),
    "loco_debug_reload_gemm_knobs": (None, []),
    "loco_op_gemm_f16x3_splitk": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                            _vp, _sz, _vp]),
    "loco_op_gemm_f16x3": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                     _i32, _i32, _i64, _i64, _i64, _i64, _vp]),
    "loco_op_permute_conv_k": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "loco_op_conv_gemm_f16x3": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _vp]),
    "loco_op_attention_f16x3": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "loco_op_attention_f16x3_pe": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i32, _i32, _vp]),
    "loco_head_last_error": (C.c_char_p, []),
    "loco_head_create": (_vp, [C.c_int]),
    "loco_head_destroy": (None, [_vp]),
    "loco_head_num_params": (_i32, []),
    "loco_head_set_params": (C.c_int, [_vp, _vp]),
    "loco_head_get_params": (C.c_int, [_vp, _vp]),
    "loco_head_workspace_bytes": (_sz, [_i32, _i32]),
    "loco_head_forward": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "loco_head_loss_grad": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "loco_head_adam_step": (C.c_int, [_vp, _vp, _f, _f, _f, _f, _f, _vp]),
}

_lib = None


def _check_foreign_build(path):
    """LOCO_ASR_LIB loads a library that `make` did not build and check (tools/ab/, tools/conv0_race/ use their own flag sets):
    inspect its gfx950 code for the banned packed-fp32 encoding (csrc/check_isa.py) before it can reach the product
    path.  LOCO_ALLOW_BANNED_ISA=1 is for the reproducer of that very hazard only."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("loco_check_isa", os.path.join(_HERE, "csrc", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        _, _, bad = mod.check(path)
    except (OSError, RuntimeError) as e:  # no llvm-objdump on this machine: say so, do not guess
        import warnings
        warnings.warn(f"LOCO_ASR_LIB={path}: could not inspect the code objects ({e})")
        return
    if bad and os.environ.get("LOCO_ALLOW_BANNED_ISA") != "1":
        raise ImportError(f"LOCO_ASR_LIB={path}: {len(bad)} packed fp32 instructions cross-select their low lane (e.g. `{bad[0]}`): "
                          "results are unreliable beside the attention kernel (DESIGN.md 5); set LOCO_ALLOW_BANNED_ISA=1 only to "
                          "reproduce that hazard")


def load():
    """Load libloco_asr.so once; raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    if os.environ.get("LOCO_ASR_LIB"):
        _check_foreign_build(LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("LOCO_ASR_LIB") and name.startswith("loco_op_") and not hasattr(lib, name):
            continue  # an A/B build of an older revision (tools/ab/): it may predate a test hook; the product entry points must exist
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.loco_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.loco_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc < 0:
        msg = load().loco_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}" if what else msg)  # HF raises ValueError for bad shapes/masks
        raise LocoError(f"{what}: [{rc}] {msg}" if what else f"[{rc}] {msg}")
    return rc
