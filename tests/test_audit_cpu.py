"""CPU: tools/audit_async_copies.py (csrc/Makefile runs it over the device assembly every library is assembled from) does flag what it is
there for -- any instruction that touches a register an inline-asm load is still writing, in a loop body, across a back edge or in front
of the loop -- passes code without one, and covers every kernel that uses the idiom (hazards)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "audit_async_copies.py")

KERNEL = """\t.type\t_ZN3dne9k_fc_ringILb1ELi8EEEvv,@function
_ZN3dne9k_fc_ringILb1ELi8EEEvv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\ts_waitcnt lgkmcnt(0)
\t;;#ASMSTART
\tglobal_load_dwordx4 v[20:23], v78, s[28:29] offset:0
\t;;#ASMEND
%(preheader)s
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
.LBB0_1:
\tv_mov_b32_e32 v40, 0
.LBB0_2:
\t;;#ASMSTART
\tds_read2st64_b32 v[50:51], v39 offset0:8 offset1:9
\t;;#ASMEND
\tv_pk_mul_f32 v[32:33], v[42:43], v[52:53]
\t;;#ASMSTART
\ts_waitcnt vmcnt(1) lgkmcnt(0)
\t;;#ASMEND
\tv_pk_add_f32 v[60:61], v[20:21], v[50:51]
%(body)s
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_pk_add_f32 v[64:65], v[16:17], v[50:51]
\t;;#ASMSTART
\tglobal_load_dwordx4 v[20:23], v78, s[28:29] offset:0x400
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[16:19], v78, s[28:29] offset:0x800
\t;;#ASMEND
%(tail)s
\ts_barrier
\ts_cbranch_scc1 .LBB0_2
\ts_branch .LBB0_1
.Lfunc_end0:
\t.size\t_ZN3dne9k_fc_ringILb1ELi8EEEvv, .Lfunc_end0-_ZN3dne9k_fc_ringILb1ELi8EEEvv
\t.type\t_ZN3dne5k_outILi2ELb1EEEvv,@function
_ZN3dne5k_outILi2ELb1EEEvv:
\tglobal_load_dwordx4 v[20:23], v78, s[28:29] offset:0
\tv_mov_b32_e32 v5, v20
.Lfunc_end1:
"""


def _run(tmp_path, preheader="\tv_mov_b32_e32 v41, 0", body="\tv_mov_b32_e32 v41, v60", tail="\tv_mov_b32_e32 v41, v61", names=("k_fc_ring",)):
    p = tmp_path / "k.s"
    p.write_text(KERNEL % {"preheader": preheader, "body": body, "tail": tail})
    return subprocess.run([sys.executable, TOOL, str(p), *names], capture_output=True, text=True)


def test_audit_passes_clean_code_and_skips_kernels_without_inline_asm_loads(tmp_path):
    r = _run(tmp_path)
    assert r.returncode == 0 and "none" in r.stdout and "1 kernels with inline-asm loads, 0 findings" in r.stdout, r.stdout + r.stderr
    assert "k_out" not in r.stdout          # its load is the compiler's own: the compiler waits for those itself


def test_audit_flags_a_copy_inside_the_loop(tmp_path):
    # v[16:19] was requested one instruction earlier: the back-edge rotation that faulted on the box (round 5)
    r = _run(tmp_path, tail="\tv_mov_b64_e32 v[24:25], v[16:17]")
    assert r.returncode == 1 and "v_mov_b64_e32 v[24:25], v[16:17]" in r.stdout, r.stdout + r.stderr
    r = _run(tmp_path, tail="\tv_accvgpr_write_b32 a3, v19")                       # ... and the AGPR spill
    assert r.returncode == 1, r.stdout + r.stderr


def test_audit_flags_a_use_across_the_back_edge(tmp_path):
    # at the top of the next iteration vmcnt(1) has retired v[20:23] but not v[16:19]: reading v16 behind that wait is a finding that only
    # the second walk around the loop can see
    r = _run(tmp_path, body="\tv_pk_add_f32 v[62:63], v[16:17], v[50:51]")
    assert r.returncode == 1 and "v[16:17]" in r.stdout, r.stdout + r.stderr


def test_audit_flags_a_copy_in_front_of_the_loop(tmp_path):
    # the first ring build: a register the prologue's load was still writing was copied at the loop's entry (40 % of the pairs wrong)
    r = _run(tmp_path, preheader="\tv_mov_b32_e32 v90, v21")
    assert r.returncode == 1 and "v_mov_b32_e32 v90, v21" in r.stdout, r.stdout + r.stderr


def test_audit_fails_when_a_named_kernel_is_gone(tmp_path):
    r = _run(tmp_path, names=("k_fc_ring", "k_fc_duo"))
    assert r.returncode == 1 and "k_fc_duo" in r.stdout, r.stdout + r.stderr


def test_makefile_audits_every_kernel_that_uses_the_idiom():
    """The kernels that issue loads from inline asm into C++ variables, from the sources: the Makefile names exactly those (and builds no
    library past a failed audit)."""
    csrc = os.path.join(ROOT, "deep-neuroevolution_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    named = set(re.search(r"^AUDITED\s*:=\s*(.+)$", mk, re.M).group(1).split())
    src = open(os.path.join(csrc, "forward.h")).read() + open(os.path.join(csrc, "forward_large.h")).read() + open(os.path.join(csrc, "forward_variants.h")).read()
    helpers = {"gload4", "gload4_after", "gload4_theta_after", "ring_row", "ring_x"}     # the inline-asm load helpers of forward.h
    for h in helpers:
        assert re.search(r"void %s\(" % h, src), h
    users = set()
    for m in re.finditer(r"__global__[^;{]*?void\s+(k_\w+)\s*\(", src):
        # the kernel's own body, brace to matching brace (the helpers above a kernel hold the asm; a kernel uses the idiom by calling them)
        i = src.index("{", src.index(")", m.end()))
        depth, j = 0, i
        while True:
            depth += {"{": 1, "}": -1}.get(src[j], 0)
            if depth == 0: break
            j += 1
        body = src[i:j]
        if re.search(r"\b(%s)\s*[<(]|asm volatile\(\"(global_load_dword|ds_read)" % "|".join(helpers), body):
            users.add(m.group(1))
    assert users == named, (users, named)
    assert "$(AUDIT)" in mk and "mv .build" in mk and mk.index("$(AUDIT)") < mk.index("mv .build")
