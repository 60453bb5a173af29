"""Python binding of libsrwn.so (the C-ABI declared in include/srwn.h).

Two interchangeable bindings of the SAME library and symbol table: the pybind11 module ``_srwn_pyb`` that build.py
generates from SIGNATURES and compiles against the header's prototypes (default: the binding north_star names), and
plain ctypes (``SRWN_BINDING=ctypes``).  Callers pass the same arguments to either -- integers for device pointers,
None for null, ctypes arrays / byref() for the few host arrays.

The product path has NO fallback: if the HIP library is missing or a symbol is absent the import
fails loudly, and every call raises RuntimeError on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import numbers
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsrwn.so")
# A/B measurements of kernel variants on one box: another build of the same library (ctypes binding, no manifest check)
_ALT_LIB = os.environ.get("SRWN_LIB_PATH")
if _ALT_LIB:
    LIB_PATH = _ALT_LIB

F32, BF16 = 0, 1
PRO_NONE, PRO_GATE = 0, 1
EPI_NONE, EPI_RELU, EPI_MASK, EPI_F32 = 0, 1, 2, 4

_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol of include/srwn.h (checked by tests/test_abi.py)
SIGNATURES = {
    "srwn_version": (C.c_int, []),
    "srwn_last_error": (C.c_char_p, []),
    "srwn_mu_law_encode": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "srwn_mu_law_decode": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "srwn_pack_a_index": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_pack_gather": (C.c_int, [_p, _p, _p, _i64, _i32, _p]),
    "srwn_pack_gather_rowsum": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _i32, _i32, _p, _p]),
    "srwn_causal_conv1d_fwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_init_conv_wgrad_partials": (_i64, [_i32, _i32, _i32, _i32]),
    "srwn_init_conv_wgrad": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_residual_layer_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                          _i32, _i32, _p]),
    "srwn_residual_group_fwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _i32, _i32, _i32,
                                          _i32, _i32, _i32, _i32, _p]),
    "srwn_residual_group_bwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_group_wt_geometry": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "srwn_residual_group_fwd_wt": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p,
                                             _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_residual_group_fwd_ic": (C.c_int, [_p, _p, _p, _i32, _p, _p, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32,
                                             _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_residual_group_bwd_wt": (C.c_int, [_p, _p, _i32, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _i32, _p, _p, _p, _p,
                                             _i32, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_debug_stamp_buffer": (C.c_int, [_p]),
    "srwn_group_plan_auto": (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_log_softmax": (C.c_int, [_p, _p, _p, _i64, _i32, _p]),
    "srwn_categorical_sample": (C.c_int, [_p, _p, _i64, _i32, C.c_uint64, _p]),
    "srwn_probs_logistic": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _f32, _p]),
    "srwn_tanh_gate": (C.c_int, [_p, _p, _p, _i64, _p]),
    "srwn_gated_activation": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p]),
    "srwn_residual_combine": (C.c_int, [_p, _i32, _p, _i32, _p, _i64, _p]),
    "srwn_relu": (C.c_int, [_p, _p, _i64, _p]),
    "srwn_mol_nll_rows": (C.c_int, [_p, _i64, _p, _i32, _p, _i64, _p]),
    "srwn_group_plan": (_i32, [_p, _i32, _i32, _i32, _p]),
    "srwn_pw_linear_ychunks": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _i32, _i64, _i32, _i32, _i64, _i32, _p]),
    "srwn_pw_linear": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _i32,
                                 _i32, _i32, _p]),
    "srwn_pw_linear_ksplit": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _i32, _i32, _i64, _i32, _i32, _p]),
    "srwn_softmax_ce_partials": (_i64, [_i64]),
    "srwn_head_softmax_ce": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _f32, _i32, _p]),
    "srwn_reduce_loss": (C.c_int, [_p, _i64, _f32, _p, _p]),
    "srwn_head_chain": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i64, _f32,
                                  _i32, _p]),
    "srwn_residual_layer_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32,
                                          _i32, _i32, _i32, _p]),
    "srwn_wgrad_slabs": (_i32, [_i64]),
    "srwn_wgrad": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _p, _i64,
                             _i32, _i32, _i32, _i32, _p]),
    "srwn_reduce_partials": (C.c_int, [_p, _i32, _i64, _i32, _i32, _f32, _p, _i64, _p]),
    "srwn_reduce_partials_multi": (C.c_int, [_p, _i32, _p]),
    "srwn_add_frame_bias": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_frame_sum": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_frame_sum_batched": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _p]),
    "srwn_adam_step": (C.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _f32, _f32, _f32, _f32, _p]),
    "srwn_time_mean_slabs": (_i32, [_i32]),
    "srwn_time_mean": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "srwn_pooled_head": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "srwn_bcast_mask": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f32, _i32, _p]),
    "srwn_skip_dgrad_all": (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _i32, _i32, _i32, _p]),
    "srwn_wgrad_layers": (C.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p, _i64,
                                    _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_wgrad_skip_wt_slabs": (_i32, [_p, _p, _i32, _i32]),
    "srwn_wgrad_skip_wt": (C.c_int, [_p, _i64, _p, _p, _i32, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_generate_ring_elems": (_i64, [_p, _i32, _i32]),
    "srwn_generate": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32,
                                _i32, _i32, _i32, _i32, _i32, _i32, C.c_uint64, _i32, _p]),
    "srwn_generate16_image_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "srwn_generate16": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32,
                                  _i32, _i32, _i32, _i32, C.c_uint64, _p]),
    "srwn_generate16_mol": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32,
                                      _i32, _i32, _i32, _i32, _p, _i32, _i32, _i64, _i32, C.c_uint64, _p]),
    "srwn_generate_mol": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32,
                                    _i32, _i32, _i32, _i32, _i32, _p, _i32, _i32, _i64, _i32, C.c_uint64, _i32, _p]),
    "srwn_mol_loss": (C.c_int, [_p, _i64, _p, _i32, _p, _p, _i64, _i64, _f32, _i32, _p]),
    "srwn_wgrad256_slabs": (_i32, [_i64, _i32]),
    "srwn_wgrad256": (C.c_int, [_p, _i64, _i64, _i32, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "srwn_wgrad_wide_slabs": (_i32, [_i64, _i32, _i32]),
    "srwn_wgrad_wide_pair": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _i64, _i32, _i64, _i32, _i32,
                                       _i32, _p]),
    "srwn_wgrad_wide": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _i64, _i32, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "srwn_tap_linear": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _i32, _i64, _p, _i64, _p, _i64,
                                  _i32, _i32, _f32, _i32, _i32, _p]),
    "srwn_wgrad_nc_layers": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_nc_input_fwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_nc_layer_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "srwn_nc_mask_words": (_i64, [_i32, _i32]),
    "srwn_nc_mask_bits": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "srwn_nc_layer_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _f32, _p, _p, _i32, _i32, _i32, _i32,
                                    _i32, _p]),
    "srwn_small_gemm": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _i64, _i64, _i32, _i64, _p, _p, _i64, _i32, _i32,
                                  _i32, _i32, _i32, _p]),
    "srwn_small_wgrad": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "srwn_mol_sample": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _p]),
    "srwn_mol_loss_dx": (C.c_int, [_p, _i64, _p, _i32, _p, _p, _i64, _f32, _p]),
    "srwn_clamp": (C.c_int, [_p, _p, _i64, _f32, _f32, _p]),
    "srwn_clamp_bwd": (C.c_int, [_p, _p, _p, _i64, _f32, _f32, _p]),
    "srwn_flow_partials": (_i64, [_i64]),
    "srwn_flow_affine_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "srwn_flow_affine_bwd": (C.c_int, [_p, _p, _p, _p, _p, _f32, _p, _p, _p, _i64, _i32, _i32, _p]),
    "srwn_causal_conv1d_dgrad": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _p]),
    "srwn_stft_frames": (_i32, [_i32]),
    "srwn_stft_power": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _p]),
    "srwn_power_loss": (C.c_int, [_p, _p, _i64, _f32, _f32, _p, _p, _p]),
    "srwn_stft_power_bwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "srwn_axpy_dev": (C.c_int, [_p, _p, _p, _f32, _i64, _p]),
    "srwn_sumsq_partials": (_i64, [_i64]),
    "srwn_sumsq": (C.c_int, [_p, _i64, _p, _p]),
    "srwn_clip_scale": (C.c_int, [_p, _i64, _f32, _f32, _p, _p]),
    "srwn_adam_step_scaled": (C.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _f32, _f32, _f32, _p, _i32, _p]),
}

_lib = None
BINDING = None      # "pybind11" or "ctypes" once loaded


def signature_hash() -> str:
    """Identifies the symbol table a pybind11 module was generated from (compiled into it as SIGNATURE_HASH)."""
    import hashlib
    txt = ";".join("%s:%s:%s" % (n, r.__name__, ",".join(a.__name__ for a in args)) for n, (r, args) in SIGNATURES.items())
    return hashlib.sha256(txt.encode()).hexdigest()[:16]


def _address(a):
    """What the pybind11 functions take for an argument the ctypes binding would have converted itself."""
    if a is None:
        return 0
    if isinstance(a, (int, float)):
        return a
    if isinstance(a, numbers.Integral):          # numpy integers
        return int(a)
    if isinstance(a, numbers.Real):
        return float(a)
    if isinstance(a, C.Array):
        return C.addressof(a)
    if isinstance(a, C._SimpleCData):            # c_void_p(...) and friends
        return a.value or 0
    if hasattr(a, "_obj"):                       # ctypes.byref(x)
        return C.addressof(a._obj)
    raise TypeError("cannot pass %r through the C-ABI" % (a,))


class _PybindLib:
    """The pybind11 module behind the attribute interface of a ctypes library (lib.srwn_xxx(args))."""

    def __init__(self, mod, keepalive):
        self._keepalive = keepalive
        for name in SIGNATURES:
            fn = getattr(mod, name)              # AttributeError if the symbol is missing
            setattr(self, name, (lambda f: lambda *args: f(*[_address(a) for a in args]))(fn))


def pybind_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_srwn_pyb" + sysconfig.get_config_var("EXT_SUFFIX"))


def bind(kind: str):
    """A fresh binding of libsrwn.so of the given kind ("pybind11" / "ctypes"); load() caches the default one."""
    if kind not in ("pybind11", "ctypes"):
        raise RuntimeError("SRWN_BINDING=%s: pybind11 or ctypes" % kind)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsrwn.so not found at %s: build it with `python sr-wavenet_amd/build.py` "
            "(there is no CPU fallback for the product path)" % LIB_PATH)
    if _ALT_LIB:
        kind = "ctypes"
    else:
        _check_manifest()
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if kind == "pybind11":
        path = pybind_path()
        if not os.path.exists(path):
            raise RuntimeError("pybind11 module %s not built: run `python sr-wavenet_amd/build.py` "
                               "(or SRWN_BINDING=ctypes for the ctypes binding of the same library)" % path)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_srwn_pyb", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.SIGNATURE_HASH != signature_hash():
            raise RuntimeError("the pybind11 module was generated from a different symbol table: rebuild with "
                               "`python sr-wavenet_amd/build.py`")
        return _PybindLib(mod, lib)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Loads libsrwn.so once and binds every symbol (SRWN_BINDING: pybind11, the default, or ctypes); raises if anything
    is missing."""
    global _lib, BINDING
    if _lib is None:
        kind = "ctypes" if _ALT_LIB else os.environ.get("SRWN_BINDING", "pybind11")
        _lib, BINDING = bind(kind), kind
    return _lib


def _check_manifest():
    """build.py records the sha256 of every source next to the library; a library older than the sources it ships with
    (an edit without a rebuild) is refused instead of silently running yesterday's kernels."""
    man = os.path.join(HERE, "libsrwn.manifest.json")
    if not os.path.exists(man) or not os.path.isdir(os.path.join(HERE, "csrc")):
        return
    import hashlib
    import json
    root = os.path.dirname(HERE)
    for rel, digest in json.load(open(man)).get("sources", {}).items():
        path = os.path.join(root, rel)
        if os.path.exists(path) and hashlib.sha256(open(path, "rb").read()).hexdigest() != digest:
            raise RuntimeError("libsrwn.so was built from a different %s: rebuild with `python sr-wavenet_amd/build.py`" % rel)


def call(name, *args):
    """Calls an int-returning entry point and raises RuntimeError on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.srwn_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (name, rc, msg.decode() if msg else ""))
    return rc
