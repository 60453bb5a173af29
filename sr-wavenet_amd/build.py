"""Builds libsrwn.so (hand-written HIP for gfx950) in-tree with hipcc.

``python sr-wavenet_amd/build.py`` or ``__graft_entry__.build()``.  hipcc cross-compiles for
gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrwn.so")
MANIFEST = os.path.join(HERE, "libsrwn.manifest.json")   # sha256 of every source the library was built from
IO_LIB = os.path.join(HERE, "libsrwn_io.so")     # host-only data path (TFRecord reader), plain g++
PYB_SRC = os.path.join(CSRC, "srwn_pybind.cpp")   # generated from _lib.SIGNATURES (the table tests/test_abi.py holds to srwn.h)
PYB_NAME = "_srwn_pyb"
IO_SOURCES = ["srwn_tfrecord.cpp"]
CXX = os.environ.get("CXX", "g++")
SOURCES = ["srwn_util.hip", "srwn_fwd.hip", "srwn_bwd.hip", "srwn_opt.hip", "srwn_pool.hip", "srwn_gemm.hip", "srwn_wgrad2.hip", "srwn_gen.hip", "srwn_gen16.hip", "srwn_flow.hip", "srwn_enc.hip", "srwn_nc.hip", "srwn_group.hip", "srwn_wgradt.hip", "srwn_ops.hip", "srwn_head.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-ffp-contract=on"]
# Kernels that fetch REGISTER operands with loads the compiler does not see (inline asm, hand-counted waits) must not
# spill: the compiler takes such a load's destination for written when the statement ends, so under register pressure
# it may spill or move it while the data is still in flight -- garbage operands, or a wild address once the register has
# been reused for a pointer (seen on a prototype: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).  The build refuses them.
NO_SPILL = {"srwn_wgradt.hip": ["wgrad_skip_wt_kernel"]}


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "srwn.h"))
    return hdrs


def source_hashes():
    """sha256 of every file the library is compiled from (sources, headers, the C-ABI header)."""
    import hashlib
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp"))]
    files.append(os.path.join(HERE, "..", "include", "srwn.h"))
    files.append(os.path.join(HERE, "..", "include", "srwn_io.h"))
    out = {}
    for f in files:
        if os.path.exists(f):
            out[os.path.relpath(f, os.path.join(HERE, ".."))] = hashlib.sha256(open(f, "rb").read()).hexdigest()
    return out


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _sha(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for f in paths:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _check_no_spill(src, obj, remarks, kernels):
    """Parses hipcc's kernel-resource-usage remarks: every kernel of `kernels` must report `VGPRs Spill: 0`."""
    import re
    seen = {}
    name = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    for k in kernels:
        hit = [(n, v) for n, v in seen.items() if k in n]
        if not hit or any(v for _, v in hit):
            if os.path.exists(obj):
                os.remove(obj)
            raise RuntimeError("%s: kernel %s must not spill registers (untracked register loads): %s" % (src, k, hit or "not found"))


def _check_inflight(src, obj, flags, kernels):
    """The structural half of the guard: the kernel's gfx950 assembly (a device-only -S compile with the same flags) is
    walked by asmcheck.check_inflight_loads -- no instruction may read or overwrite the destination of a global_load
    while that load can still be in flight under the s_waitcnt vmcnt(N) the code actually executes.  A spill-free build
    can still copy or re-coalesce such a register before the hand-counted wait; this refuses it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_srwn_asmcheck", os.path.join(HERE, "asmcheck.py"))
    ac = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ac)
    cmd = [HIPCC] + flags + ["--cuda-device-only", "-S", src, "-o", "-"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr[-4000:]))
    for k in kernels:
        bad = ac.check_inflight_loads(r.stdout, k)
        if bad:
            if os.path.exists(obj):
                os.remove(obj)
            raise RuntimeError("%s: kernel %s touches the destination of a load that may still be in flight:\n  %s"
                               % (src, k, "\n  ".join(bad[:8])))


def build(force: bool = False, verbose: bool = False, diag: bool = False, variant: str = "", extra_flags=()) -> str:
    """diag=False: libsrwn.so (+ libsrwn_io.so, the pybind11 module, the manifest).  diag=True: libsrwn_diag.so, the same
    sources with -DSRWN_DIAG (stamped kernel instantiations, SRWN_WT_DEBUG), objects under csrc/diag/; load it with
    SRWN_LIB_PATH (ctypes binding, no manifest).  variant="name" + extra_flags: an A/B build of the same sources with
    extra compiler flags -> ab/libsrwn_<name>.so (git-ignored; objects under csrc/ab_<name>/), also for SRWN_LIB_PATH."""
    import json
    per_src = {}      # "srwn_x.hip:-flag" in extra_flags: that source only (A/B builds of one translation unit's options)
    for f in list(extra_flags):
        if ".hip:" in f:
            name, fl = f.split(":", 1)
            per_src.setdefault(name, []).extend(fl.split("=", 1) if fl.startswith("-mllvm=") else [fl])
    extra_flags = [f for f in extra_flags if ".hip:" not in f]
    flags = FLAGS + (["-DSRWN_DIAG"] if diag else []) + list(extra_flags)
    if variant:
        diag = True      # (same treatment: no pybind module, no product manifest)
        odir = os.path.join(CSRC, "ab_" + variant)
        os.makedirs(os.path.join(HERE, "..", "ab"), exist_ok=True)
        lib = os.path.join(HERE, "..", "ab", "libsrwn_%s.so" % variant)
    else:
        odir = os.path.join(CSRC, "diag") if diag else CSRC
        lib = os.path.join(HERE, "libsrwn_diag.so") if diag else LIB
    os.makedirs(odir, exist_ok=True)
    man_path = os.path.join(odir, "objects.manifest.json") if diag else MANIFEST
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [os.path.join(odir, os.path.splitext(os.path.basename(s))[0] + ".o") for s in srcs]
    deps = sorted(_deps())
    # Staleness is decided from what each object was COMPILED from (recorded sha256 of source + headers + flags), not
    # from mtimes: a tree whose sources changed but carry older mtimes than the travelling .o files (rsync -t, a
    # snapshot restore) would otherwise have its stale library blessed by a fresh manifest.
    try:
        old = json.load(open(man_path)).get("objects", {})
    except (OSError, ValueError):
        old = {}
    keys = {}

    def cc(pair):
        src, obj = pair
        rel = os.path.basename(src)
        sflags = flags + per_src.get(rel, [])
        key = _sha([src] + deps, " ".join(sflags))
        if force or not os.path.exists(obj) or old.get(rel) != key:
            guarded = NO_SPILL.get(rel, [])
            cmd = [HIPCC] + sflags + (["-Rpass-analysis=kernel-resource-usage"] if guarded else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-8000:]))
            if guarded:
                _check_no_spill(src, obj, r.stderr, guarded)
                _check_inflight(src, obj, sflags, guarded)
        keys[rel] = key
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    relink = force or not os.path.exists(lib) or old != keys
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-8000:])
    if diag:
        with open(man_path, "w") as f:
            json.dump({"flags": flags, "objects": keys}, f, indent=1, sort_keys=True)
        return lib
    build_io(force)
    build_pybind(force or relink)
    with open(MANIFEST, "w") as f:      # which sources this library came from: _lib.load() refuses a stale one.  Written
        # only here, after every object has been compiled from (or verified against) the hashes it records
        json.dump({"flags": flags, "objects": keys, "sources": source_hashes()}, f, indent=1, sort_keys=True)
    return lib


def _signatures():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_srwn_lib_for_build", os.path.join(HERE, "_lib.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def pybind_source() -> str:
    """The pybind11 module over the C-ABI: one function per entry point of include/srwn.h, with the argument list of
    _lib.SIGNATURES -- pointers travel as integers (device pointers are integers on the Python side anyway), everything
    else as the header's scalar types.  The calls go through the header's prototypes, so a table entry whose argument
    count or scalar kinds disagree with srwn.h does not compile."""
    import ctypes as C
    L = _signatures()
    kinds = {C.c_void_p: ("std::uintptr_t", "P{%s}"), C.c_int32: ("int32_t", "%s"), C.c_int64: ("int64_t", "%s"),
             C.c_float: ("float", "%s"), C.c_uint64: ("uint64_t", "%s"), C.c_int: ("int", "%s")}
    out = ["// GENERATED by sr-wavenet_amd/build.py (pybind_source) from _lib.SIGNATURES -- do not edit.",
           "// The thin pybind11 module over the C-ABI of include/srwn.h: pointers as integers, scalars as declared.",
           "#include <pybind11/pybind11.h>", "#include <cstdint>", '#include "../../include/srwn.h"', "namespace py = pybind11;",
           "namespace {", "struct P {   // an address, convertible to whatever pointer type the prototype asks for",
           "  std::uintptr_t v;", "  template <class T> operator T*() const { return reinterpret_cast<T*>(v); }", "};",
           "}  // namespace", "", "PYBIND11_MODULE(%s, m) {" % PYB_NAME,
           '  m.doc() = "pybind11 binding of libsrwn.so (include/srwn.h)";',
           '  m.attr("SIGNATURE_HASH") = "%s";' % L.signature_hash()]
    for name, (res, args) in L.SIGNATURES.items():
        params = ", ".join("%s a%d" % (kinds[t][0], i) for i, t in enumerate(args))
        passed = ", ".join(kinds[t][1] % ("a%d" % i) for i, t in enumerate(args))
        if res is C.c_char_p:
            body = "const char* r = %s(%s); return py::bytes(r ? r : \"\");" % (name, passed)
        else:
            body = "return %s(%s);" % (name, passed)
        out.append('  m.def("%s", [](%s) { %s });' % (name, params, body))
    out.append("}")
    return "\n".join(out) + "\n"


def pybind_path() -> str:
    import sysconfig
    return os.path.join(HERE, PYB_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force: bool = False) -> str:
    """The pybind11 module (the binding north_star names): g++, linked against libsrwn.so next to it ($ORIGIN rpath)."""
    import sysconfig
    try:
        import pybind11
    except ImportError:
        if os.environ.get("SRWN_BINDING") == "ctypes":      # the ctypes binding of the same library needs no module
            print("pybind11 not importable: skipping the _srwn_pyb module (SRWN_BINDING=ctypes)", file=sys.stderr)
            return ""
        raise RuntimeError("pybind11 is not importable: install it, or set SRWN_BINDING=ctypes to bind libsrwn.so with ctypes")
    src = pybind_source()
    if not os.path.exists(PYB_SRC) or open(PYB_SRC).read() != src:
        with open(PYB_SRC, "w") as f:
            f.write(src)
    target = pybind_path()
    if force or _stale(target, [PYB_SRC, LIB, os.path.join(HERE, "..", "include", "srwn.h")]):
        cmd = [CXX, "-O1", "-fPIC", "-shared", "-std=c++17", "-fvisibility=hidden", "-I", pybind11.get_include(),
               "-I", sysconfig.get_paths()["include"], PYB_SRC, "-o", target, "-L", HERE, "-lsrwn", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for the pybind11 module:\n%s" % r.stderr[-8000:])
    return target


def build_io(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in IO_SOURCES]
    if force or _stale(IO_LIB, srcs + [os.path.join(HERE, "..", "include", "srwn_io.h")]):
        cmd = [CXX, "-O2", "-fPIC", "-shared", "-std=c++17", "-pthread", "-Wall", "-o", IO_LIB] + srcs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for the IO library:\n%s" % r.stderr[-8000:])
    return IO_LIB


if __name__ == "__main__":
    # python build.py [--force] [--diag] [--variant NAME -DFLAG ... -mllvm=-llvm-option=value ... srwn_x.hip:-mllvm=-option=value ...]
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv, variant=_variant,
                extra_flags=[a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-mllvm=") or ".hip:" in a
                             for a in (a.split("=", 1) if a.startswith("-mllvm=") else [a])]))      # -mllvm=-opt=val -> -mllvm -opt=val
