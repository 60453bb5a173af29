"""CPU tests: the C-ABI library builds, loads and exports every symbol include/srwn.h declares;
argument errors are reported without touching a GPU; the product has no CPU fallback."""
import os
import re

import pytest

from tests._pkg import ROOT, sub


def _declared():
    txt = open(os.path.join(ROOT, "include", "srwn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(srwn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = sub("_lib")
    if not os.path.exists(L.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "sr-wavenet_amd", "build.py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build()
    lib = L.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "declared in srwn.h but not exported: " + n
        assert n in L.SIGNATURES, "no ctypes signature for " + n
    assert sorted(L.SIGNATURES) == names, "signature table and header differ"
    assert lib.srwn_version() >= 100


def test_pybind11_module_is_the_default_binding():
    """The binding north_star names: a pybind11 module generated from the signature table, compiled against the prototypes
    of include/srwn.h, exporting every entry point; ctypes binds the same library on request."""
    L = sub("_lib")
    lib = L.load()
    assert L.BINDING == os.environ.get("SRWN_BINDING", "pybind11")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_srwn_pyb", L.pybind_path())
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.SIGNATURE_HASH == L.signature_hash()
    for n in _declared():
        assert callable(getattr(mod, n)), n
    assert mod.srwn_version() == L.bind("ctypes").srwn_version() == lib.srwn_version()
    src = open(os.path.join(ROOT, "sr-wavenet_amd", "csrc", "srwn_pybind.cpp")).read()
    import importlib.util as iu
    bspec = iu.spec_from_file_location("b", os.path.join(ROOT, "sr-wavenet_amd", "build.py"))
    b = iu.module_from_spec(bspec); bspec.loader.exec_module(b)
    assert src == b.pybind_source(), "csrc/srwn_pybind.cpp is not what build.py generates from _lib.SIGNATURES"


@pytest.mark.parametrize("binding", ["pybind11", "ctypes"])
def test_argument_errors_do_not_need_a_gpu(binding):
    L = sub("_lib")
    lib = L.bind(binding)
    # empty work returns 0 before any launch
    assert lib.srwn_mu_law_encode(None, None, 0, 256, None) == 0
    assert lib.srwn_pw_linear(None, 0, 0, 16, 16, None, None, None, 0, 32, 32, 0, None, 0, 0, 0, 1, None) == 0
    # null pointers / bad shapes / bad dtype -> negative codes + message, no launch
    assert lib.srwn_mu_law_encode(None, None, 10, 256, None) == -3
    assert b"null" in lib.srwn_last_error()
    assert lib.srwn_residual_layer_fwd(1, None, 1, 1, 1, 1, 1, 1, 2, 64, 64, 3, 1, 1, 1, 64, 1, None) == -4   # K=3
    assert lib.srwn_residual_layer_fwd(1, None, 1, 1, 1, 1, 1, 1, 2, 64, 48, 2, 1, 1, 1, 48, 1, None) == -4   # R=48
    assert lib.srwn_residual_layer_fwd(1, None, 1, 1, 1, 1, 1, 1, 2, 64, 64, 2, 1, 1, 1, 64, 7, None) == -1   # dtype
    assert lib.srwn_pw_linear(1, 8, 0, 8, 8, 1, None, 1, 32, 32, 32, 5, None, 0, 0, 0, 1, None) == -2          # Cin % 16
    assert lib.srwn_wgrad(1, 0, 64, 1, 0, 64, None, 0, 1, 1, 64, None, 99, 1, None, 64, 64, 1, 0, 1, None) == -2
    # the round-3 entry points: the same contract
    import ctypes as C
    i32 = lambda *v: (C.c_int32 * len(v))(*v)
    assert lib.srwn_wgrad_skip_wt(None, 0, None, None, 0, None, 256, None, None, 0, 1, 8, 100, 64, 256, 1, None) == 0    # no layers
    assert lib.srwn_wgrad_skip_wt(None, 0, i32(1), i32(100), 1, None, 256, None, None, 0, 1, 1, 100, 64, 256, 1, None) == -3
    assert lib.srwn_wgrad_skip_wt(1, 1 << 20, i32(1), i32(100), 1, 1, 256, 1, None, 0, 1, 1, 100, 64, 256, 0, None) == -4     # fp32: not built
    assert lib.srwn_wgrad_skip_wt(1, 1 << 20, i32(1), i32(100), 1, 1, 256, 1, None, 0, 1, 1, 100, 32, 128, 1, None) == -4     # widths
    assert lib.srwn_wgrad_skip_wt(1, 10, i32(1), i32(100), 1, 1, 256, 1, None, 0, 1, 1, 100, 64, 256, 1, None) == -2          # tiles exceed the layer stride
    assert lib.srwn_wgrad_skip_wt(1, 1 << 20, i32(0), i32(100), 1, 1, 256, 1, None, 0, 1, 1, 100, 64, 256, 1, None) == -2     # stride 0
    assert lib.srwn_wgrad_skip_wt_slabs(i32(1, 1, 1, 1, 32, 32), i32(500, 500, 500, 500, 500, 500), 6, 16000) >= 1
    assert lib.srwn_generate16(None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
                               2, 0, 8, 8, 64, 256, 256, 0, 0, None) == 0                                                  # no utterances
    assert lib.srwn_generate16(None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
                               2, 1, 8, 8, 64, 256, 256, 0, 0, None) == -3
    assert lib.srwn_generate16(1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, None, None, i32(1, 2), 2, 1, 8, 8, 48, 256, 256, 0, 0,
                               None) == -4                                                                                   # R=48
    assert lib.srwn_generate16_mol(1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, None, None, i32(1, 2), 2, 1, 8, 8, 64, 256, 17, None,
                                   1, 1, 0, 0, 0, None) == -2                                                               # 17 mixtures
    assert lib.srwn_generate16_image_elems(30, 0, 64, 256) == 30 * 4 * 14 * 512 and lib.srwn_generate16_image_elems(30, 0, 48, 256) == 0
    assert lib.srwn_wgrad_wide_pair(None, None, None, None, None, None, None, None, 64, 256, 4, 64, 256, 256, 0, 4, 0, 1, None) == 0
    assert lib.srwn_wgrad_wide_pair(1, 1, 1, None, None, 1, 1, None, 64, 256, 4, 64, 256, 256, 100, 4, 0, 1, None) == -3
    import ctypes
    out = [ctypes.c_int32() for _ in range(2)] + [ctypes.c_int64()] + [ctypes.c_int32()]      # byref() through either binding
    assert lib.srwn_group_wt_geometry(i32(1, 2, 4, 8, 16), 5, 8, 16000, 64, 1, 0, *[ctypes.byref(o) for o in out]) == 0
    assert out[0].value == 500 and out[3].value == 256
    with pytest.raises(RuntimeError):
        L.call("srwn_mu_law_decode", None, None, 5, 256, None)
    assert lib.srwn_wgrad_slabs(128000) == 42 and lib.srwn_softmax_ce_partials(100) == 4


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    M = sub("model")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.WaveNet(64, 8, [1, 2], output_channels=8)


def test_product_does_not_import_the_oracle():
    """Nothing under the package imports, loads or executes anything under oracle/ (only tests/, __graft_entry__.smoke()
    and bench.py's cpu_baseline may); bench.py itself touches it inside cpu_baseline only."""
    pkg = os.path.join(ROOT, "sr-wavenet_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                for line in src.splitlines():
                    if "oracle" in line and ("import" in line or "ctypes" in line or "subprocess" in line):
                        raise AssertionError("%s touches the oracle: %s" % (fn, line))
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for chunk in bench.split("\ndef ")[1:]:
        if "from oracle" in chunk or "import oracle" in chunk:
            assert chunk.startswith("cpu_baseline("), "bench.py imports the oracle outside cpu_baseline: def " + chunk[:40]
    assert "oracle" not in bench.split("\ndef ")[0].replace("oracle ii", "").replace("(oracle", ""), "bench.py: module-level oracle import"


def test_shipped_library_has_no_wrong_answer_switches():
    """The timing-ablation switch of the backward group kernel (SRWN_WT_DEBUG: skips a contraction loop or a partial
    store -- results are wrong by design) and the stamped kernel instantiations live in the -DSRWN_DIAG build only
    (libsrwn_diag.so): the shipped library neither reads that variable nor accepts a stamp buffer."""
    L = sub("_lib")
    blob = open(os.path.join(ROOT, "sr-wavenet_amd", "libsrwn.so"), "rb").read()
    for name in (b"SRWN_WT_DEBUG", b"SRWN_GW_MASK", b"SRWN_FUSE_WG"):
        assert name not in blob, name
    lib = L.bind("ctypes")
    assert lib.srwn_debug_stamp_buffer(None) == 0
    assert lib.srwn_debug_stamp_buffer(4096) == -4 and b"diagnostic" in lib.srwn_last_error()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sr-wavenet_amd")):
        for fn in files:
            if fn.endswith(".py") and fn != "build.py":      # (build.py's docstring names what the --diag build holds)
                assert "SRWN_WT_DEBUG" not in open(os.path.join(dirpath, fn)).read(), fn


def test_build_decides_staleness_from_recorded_hashes(tmp_path):
    """build.py recompiles a source when the sha256 of (source + headers + flags) differs from the one its object was
    compiled from -- not when an mtime is newer -- and records a hash per object only after compiling it."""
    import json
    B = sub("build")
    man = json.load(open(B.MANIFEST))
    assert set(man["objects"]) == {s for s in B.SOURCES}
    deps = sorted(B._deps())
    for s in B.SOURCES:
        assert man["objects"][s] == B._sha([os.path.join(B.CSRC, s)] + deps, " ".join(B.FLAGS)), s
    assert man["sources"] == B.source_hashes()
