"""CPU: tools/audit_async_copies.py (run by __graft_entry__.build() over k_fc_ring's ISA) does flag what it is there for -- a copy of a
register that an inline-asm load of the row loop is still writing -- and passes a loop without one (hazards)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "audit_async_copies.py")

LOOP = """_ZN3dne9k_fc_ringILb1ELi8EEEvv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
\tv_mov_b32_e32 v40, 0
.LBB0_2:
\ts_waitcnt vmcnt(6) lgkmcnt(0)
\tds_read2st64_b32 v[50:51], v39 offset0:8 offset1:9
\tv_pk_mul_f32 v[32:33], v[42:43], v[52:53]
\tglobal_load_dwordx4 v[20:23], v78, s[28:29] offset:0
\ts_waitcnt vmcnt(6) lgkmcnt(0)
%s
\ts_waitcnt vmcnt(6) lgkmcnt(0)
\tglobal_load_dwordx4 v[16:19], v78, s[28:29] offset:0x400
\ts_waitcnt vmcnt(6) lgkmcnt(0)
\ts_barrier
\ts_cbranch_scc1 .LBB0_2
\ts_branch .LBB0_1
\t.end_amdhsa_kernel
"""


def _run(extra, tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LOOP % extra)
    return subprocess.run([sys.executable, TOOL, str(p), "k_fc_ring"], capture_output=True, text=True)


def test_audit_passes_a_clean_row_loop(tmp_path):
    r = _run("\tv_pk_add_f32 v[60:61], v[20:21], v[32:33]", tmp_path)          # reading a landed register in arithmetic is what the loop is for
    assert r.returncode == 0 and "none" in r.stdout, r.stdout + r.stderr
    assert ".LBB0_2" in r.stdout                                                  # the innermost loop around the counted waits, not the outer one


def test_audit_flags_a_copy_of_a_register_a_load_is_writing(tmp_path):
    r = _run("\tv_mov_b64_e32 v[24:25], v[20:21]", tmp_path)                     # the back-edge rotation that faulted on the box (round 5)
    assert r.returncode == 1 and "v_mov_b64_e32 v[24:25], v[20:21]" in r.stdout, r.stdout + r.stderr
    r = _run("\tv_accvgpr_write_b32 a3, v51", tmp_path)                            # ... and the AGPR spill of an LDS read's destination
    assert r.returncode == 1, r.stdout + r.stderr
