"""build.py's structural guard for hand-counted register loads (sr-wavenet_amd/asmcheck.py), on synthetic gfx950 assembly:
the cases a spill count cannot see -- a copy of an in-flight destination, a wait that retires too few loads, a hazard that
only exists around a loop's back edge, a wait on one path only -- and the clean forms next to them.  CPU only."""
import os
import subprocess

import pytest

from tests._pkg import sub, ROOT

AC = sub("asmcheck")


def _fn(body: str) -> str:
    return "\t.text\nkern_under_test:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"


def test_clean_counted_wait():
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	global_load_dwordx4 v[14:17], v[2:3], off offset:16
	global_load_lds_dwordx4 v[4:5], off
	s_waitcnt vmcnt(2)
	v_mfma_f32_16x16x32_bf16 v[100:103], v[10:13], v[20:23], v[100:103]
	s_waitcnt vmcnt(1)
	v_mfma_f32_16x16x32_bf16 v[100:103], v[14:17], v[20:23], v[100:103]
""")
    assert AC.check_inflight_loads(asm, "kern_under_test") == []


def test_copy_before_the_wait_is_refused():
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	v_mov_b64_e32 v[30:31], v[10:11]
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v40, v30, v31
""")
    bad = AC.check_inflight_loads(asm, "kern_under_test")
    assert len(bad) == 1 and "v_mov_b64_e32" in bad[0]


def test_destination_reused_for_an_address_is_refused():
    """The prototype's aperture violation: a register of an in-flight destination overwritten with a pointer."""
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	v_lshl_add_u64 v[12:13], v[6:7], 3, v[8:9]
	global_load_dwordx2 v[20:21], v[12:13], off
	s_waitcnt vmcnt(0)
""")
    bad = AC.check_inflight_loads(asm, "kern_under_test")
    assert any("v_lshl_add_u64" in b for b in bad) and any("global_load_dwordx2" in b for b in bad)


def test_wait_count_too_large():
    """Three younger operations behind the load: vmcnt(3) retires it, vmcnt(4) may leave it in flight."""
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	global_store_dwordx4 v[2:3], v[50:53], off
	global_load_lds_dwordx4 v[4:5], off
	global_load_lds_dwordx4 v[4:5], off offset:1024
	s_waitcnt vmcnt(4)
	v_pk_add_f32 v[60:61], v[10:11], v[12:13]
""")
    assert AC.check_inflight_loads(asm, "kern_under_test")
    assert AC.check_inflight_loads(asm.replace("vmcnt(4)", "vmcnt(3)"), "kern_under_test") == []


def test_hazard_around_the_back_edge():
    """The load issued at the bottom of the loop body is still in flight at its top on the next trip."""
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	s_waitcnt vmcnt(0)
.LBB0_1:
	v_mfma_f32_16x16x32_bf16 v[100:103], v[10:13], v[20:23], v[100:103]
	global_load_dwordx4 v[10:13], v[2:3], off
	s_add_i32 s4, s4, -1
	s_cmp_lg_u32 s4, 0
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
""")
    bad = AC.check_inflight_loads(asm, "kern_under_test")
    assert any("v_mfma" in b for b in bad)
    fixed = asm.replace("\ts_add_i32", "\ts_waitcnt vmcnt(0)\n\ts_add_i32")
    assert AC.check_inflight_loads(fixed, "kern_under_test") == []


def test_wait_on_one_path_only():
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	s_cbranch_scc1 .LBB0_2
	s_waitcnt vmcnt(0)
.LBB0_2:
	v_add_f32_e32 v40, v10, v11
""")
    assert AC.check_inflight_loads(asm, "kern_under_test")


def test_younger_operation_on_one_path_is_not_counted():
    """A store that only one path issues cannot be relied on to age the load: vmcnt(1) may retire nothing."""
    asm = _fn("""
	global_load_dwordx4 v[10:13], v[2:3], off
	s_cbranch_scc1 .LBB0_2
	global_store_dword v[2:3], v50, off
.LBB0_2:
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v40, v10, v11
""")
    assert AC.check_inflight_loads(asm, "kern_under_test")


def test_shipped_kernel_is_clean():
    """The one shipped kernel with such loads (csrc/srwn_wgradt.hip), as build.py checks it: a device-only -S compile."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    B = sub("build")
    src = os.path.join(ROOT, "sr-wavenet_amd", "csrc", "srwn_wgradt.hip")
    r = subprocess.run([hipcc] + B.FLAGS + ["--cuda-device-only", "-S", src, "-o", "-"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in B.NO_SPILL["srwn_wgradt.hip"]:
        assert AC.check_inflight_loads(r.stdout, k) == []
