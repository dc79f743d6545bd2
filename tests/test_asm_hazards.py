"""The device assembly of every HIP translation unit is free of the two gfx940+ hazards that hipcc does not cover inside
inline-asm statements (tools/check_asm_hazards.py): an asm statement reading an SGPR that a VALU instruction wrote fewer
than 2 (VMEM: 5) wait states earlier, and an asm store of more than 64 bits whose data registers a VALU instruction
overwrites fewer than 2 wait states later.  The second one corrupted rows of the GBM matrix in a round-4 build
(reproduced in isolation by tools/ubench_hazard.hip: 0.5 % of the stored values wrong with no wait
state in between, none with one or two); the asm statements now carry their own `s_nop 1`.
Compiles every .hip to assembly (`make asmcheck`, ~1 min on 8 cores): CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_the_checker_finds_both_hazards_and_accepts_the_covered_forms(tmp_path):
    import check_asm_hazards as chk
    bad_store = tmp_path / "bad_store.s"
    bad_store.write_text("""
_Z1kv:                                  ; @_Z1kv
	;;#ASMSTART
	global_store_dwordx4 v16, v[2:5], s[74:75] nt
	;;#ASMEND
	v_fmac_f64_e32 v[2:3], v[2:3], v[32:33]
""")
    ok_store = tmp_path / "ok_store.s"
    ok_store.write_text("""
_Z1kv:
	;;#ASMSTART
	global_store_dwordx4 v16, v[2:5], s[74:75] nt
	s_nop 1
	;;#ASMEND
	v_fmac_f64_e32 v[2:3], v[2:3], v[32:33]
	;;#ASMSTART
	global_store_dwordx2 v16, v[6:7], s[74:75] nt
	;;#ASMEND
	v_fmac_f64_e32 v[6:7], v[6:7], v[32:33]
""")
    bad_sgpr = tmp_path / "bad_sgpr.s"
    bad_sgpr.write_text("""
_Z1kv:
	v_cmp_lt_f64_e64 s[16:17], v[0:1], v[2:3]
	;;#ASMSTART
	v_cndmask_b32_e64 v35, v22, v43, s[16:17]
	;;#ASMEND
	v_readlane_b32 s40, v91, 0
	v_readlane_b32 s41, v91, 1
	;;#ASMSTART
	v_fma_f64 v[4:5], v[4:5], v[6:7], s[40:41]
	;;#ASMEND
	v_readfirstlane_b32 s44, v1
	s_nop 2
	;;#ASMSTART
	global_store_dwordx2 v16, v[8:9], s[44:45] nt
	;;#ASMEND
""")
    ok_sgpr = tmp_path / "ok_sgpr.s"
    ok_sgpr.write_text("""
_Z1kv:
	v_cmp_lt_f64_e64 s[16:17], v[0:1], v[2:3]
	;;#ASMSTART
	s_nop 1
	v_cndmask_b32_e64 v35, v22, v43, s[16:17]
	;;#ASMEND
	s_mov_b32 s40, 0x11111111
	;;#ASMSTART
	v_fma_f64 v[4:5], v[4:5], v[6:7], s[40:41]
	;;#ASMEND
""")
    bad_trans = tmp_path / "bad_trans.s"
    bad_trans.write_text("""
_Z1kv:
	v_rsq_f64_e32 v[10:11], v[2:3]
	;;#ASMSTART
	v_fma_f64 v[4:5], v[10:11], v[6:7], s[40:41]
	;;#ASMEND
	v_rcp_f64_e32 v[12:13], v[2:3]
	v_mul_f64 v[0:1], v[0:1], v[0:1]
	;;#ASMSTART
	v_fma_f64 v[4:5], v[12:13], v[6:7], s[40:41]
	;;#ASMEND
""")
    assert len(chk.check_trans(str(bad_trans))) == 1          # the first asm FMA only: the second has a wait state in between
    assert len(chk.check_wide_stores(str(bad_store))) == 1 and chk.check_wide_stores(str(ok_store)) == []
    found = chk.check(str(bad_sgpr))
    assert len({f[1] for f in found}) == 3, found     # three asm statements: the mask, the restored constant, the store's base (3 < 5 wait states)
    assert chk.check(str(ok_sgpr)) == [] and chk.check(str(ok_store)) == []


def test_no_translation_unit_has_an_uncovered_hazard():
    r = subprocess.run(["make", "-j", str(min(8, os.cpu_count() or 1)), "asmcheck"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert "0 hazard(s) in" in r.stdout
