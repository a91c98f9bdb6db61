#!/usr/bin/env python3
"""hipcc does not model the loads inside an inline-asm statement: it treats their destination as written when the statement "returns",
so whenever its register allocation needs a move it may copy -- or its scheduler may read -- a register that such a load is still
writing ("hazards": wrong pairs on some launches, a memory fault).  The kernels are built so that it has no
reason to; this script checks the ISA the library was assembled from (csrc/Makefile: hipcc -save-temps, then this, then the .so appears).

For EVERY kernel of the file that issues a load from inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND: global_load_*, ds_read*) it
walks the instruction stream with the hardware's counters:
  * a vector-memory FIFO (vmcnt: loads, stores and LDS-DMAs retire in issue order) and an LDS FIFO (lgkmcnt: in order unless a scalar load
    is in flight -- then only lgkmcnt(0) proves anything);
  * every destination register of an inline-asm load is IN FLIGHT from its issue until an s_waitcnt retires it;
  * any other instruction that names an in-flight register -- a v_mov / v_accvgpr_write copy, arithmetic, another load's destination -- is a finding;
  * every loop (a branch back to an earlier label) is walked a second time with the state at its back edge, so loads in flight across the back
    edge (the row loops' rolling windows) and the loop's entry (preheader loads) are covered; at the target of a forward branch the in-flight
    sets of both paths are merged.
    python tools/audit_async_copies.py <device .s> [kernel-substring ...]        exit code 1 = a finding (or a named kernel is missing)"""
import re
import sys

LOAD = re.compile(r"^(global_load_(?!lds)\w+|flat_load_\w+|buffer_load_(?!.*\blds\b)\w+|scratch_load_\w+)\s")
VM_OTHER = re.compile(r"^(global_load_lds_\w+|global_store_\w+|flat_store_\w+|buffer_store_\w+|buffer_load_\w+|global_atomic_\w+|flat_atomic_\w+|buffer_atomic_\w+|scratch_store_\w+|buffer_wbl2|buffer_inv)\b")
DS = re.compile(r"^ds_\w+\s")
SMEM = re.compile(r"^(s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime|s_getreg_b32|s_sendmsg\w*|s_dcache_\w+|s_atc_probe\w*)\b")
REG = re.compile(r"\b([va])(?:(\d+)\b|\[(\d+):(\d+)\])")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def first_operand(text):
    """destination of a load: the first register operand"""
    m = REG.search(text)
    if not m:
        return set()
    return regs_of(m.group(0))


class State:
    def __init__(self):
        self.vm = []      # FIFO of (frozenset of async dst regs or empty, line no)
        self.lds = []
        self.smem = 0

    def copy(self):
        s = State(); s.vm = list(self.vm); s.lds = list(self.lds); s.smem = self.smem
        return s

    def inflight(self):
        out = {}
        for fifo in (self.vm, self.lds):
            for regs, ln in fifo:
                for r in regs:
                    out[r] = ln
        return out

    def merge(self, o):   # at a join: the longer FIFOs (fewer retirements proven), union of what is in flight
        if len(o.vm) > len(self.vm): self.vm, o_vm = list(o.vm), self.vm
        else: o_vm = o.vm
        if len(o.lds) > len(self.lds): self.lds, o_lds = list(o.lds), self.lds
        else: o_lds = o.lds
        have = set().union(*[r for r, _ in self.vm]) if self.vm else set()
        extra = [(r, ln) for r, ln in o_vm if r and not r <= have]
        self.vm = extra + self.vm
        have = set().union(*[r for r, _ in self.lds]) if self.lds else set()
        extra = [(r, ln) for r, ln in o_lds if r and not r <= have]
        self.lds = extra + self.lds
        self.smem = max(self.smem, o.smem)


def audit_kernel(name, lines, base_line):
    """lines: the kernel's text lines.  Returns (n_asm_loads, n_async_regs, loops walked, findings)."""
    insts = []   # (index in lines, mnemonic+operands, in_asm)
    labels = {}
    in_asm = False
    for i, l in enumerate(lines):
        s = l.strip()
        if s.startswith(";;#ASMSTART"): in_asm = True; continue
        if s.startswith(";;#ASMEND"): in_asm = False; continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m: labels[m.group(1)] = len(insts); continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if s:
            insts.append((i, s, in_asm))
    n_asm_loads = sum(1 for _, s, a in insts if a and (LOAD.match(s) or (DS.match(s) and s.startswith("ds_read"))))
    if not n_asm_loads:
        return 0, 0, 0, []
    findings, async_regs = [], set()
    pending = {}   # label index -> State saved at forward branches

    def step(k, st, report):
        i, s, a = insts[k]
        mn = s.split()[0]
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m: st.vm = st.vm[len(st.vm) - int(m.group(1)):] if int(m.group(1)) else []
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                if n == 0: st.lds, st.smem = [], 0
                elif not st.smem: st.lds = st.lds[len(st.lds) - n:]
            if re.fullmatch(r"s_waitcnt\s+0(x0)?", s): st.vm, st.lds, st.smem = [], [], 0
            return
        used = regs_of(s)
        fl = st.inflight()
        hit = used & set(fl)
        if hit and report is not None:
            report.append((base_line + i + 1, s, sorted("%s%d" % r for r in hit), min(fl[r] for r in hit) + base_line + 1))
        if LOAD.match(s):
            dst = first_operand(s) if a else set()
            if a: async_regs.update(dst)
            st.vm.append((frozenset(dst), i))
        elif VM_OTHER.match(s):
            st.vm.append((frozenset(), i))
        elif DS.match(s):
            dst = first_operand(s) if (a and s.startswith("ds_read")) else set()
            if a: async_regs.update(dst)
            st.lds.append((frozenset(dst), i))
        elif SMEM.match(s) and mn.startswith(("s_load", "s_buffer_load")):
            st.smem += 1

    st = State()
    loops = 0
    seen = set()
    for k in range(len(insts)):
        for lab, idx in labels.items():
            if idx == k and lab in pending:
                st.merge(pending.pop(lab))
        step(k, st, findings)
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", insts[k][1])
        if m and m.group(1) in labels:
            tgt = labels[m.group(1)]
            if tgt <= k:   # a loop: once more around it with what is in flight at the back edge
                loops += 1
                st2 = st.copy()
                rep = []
                for j in range(tgt, k + 1):
                    step(j, st2, rep)
                for f in rep:
                    if (f[0], f[1]) not in seen:
                        findings.append(f)
                st.merge(st2)
            else:
                p = pending.get(m.group(1))
                if p is None: pending[m.group(1)] = st.copy()
                else: p.merge(st.copy())
        for f in findings:
            seen.add((f[0], f[1]))
    # de-duplicate
    uniq, out = set(), []
    for f in findings:
        if (f[0], f[1]) not in uniq:
            uniq.add((f[0], f[1])); out.append(f)
    return n_asm_loads, len(async_regs), loops, out


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    text = open(path).read().split("\n")
    kernels = []
    for i, l in enumerate(text):
        m = re.match(r"\s*\.type\s+(\S+),@function", l)
        if m:
            kernels.append((m.group(1), i))
    bad = 0
    audited = []
    for name, st in kernels:
        en = st
        while en < len(text) and not re.match(r"\.Lfunc_end\d+:", text[en]):
            en += 1
        n_loads, n_regs, loops, findings = audit_kernel(name, text[st:en], st)
        if not n_loads:
            continue
        audited.append(name)
        print("audit %s: %d inline-asm loads, %d asynchronously written registers, %d loops re-walked, uses of a register still in flight: %s"
              % (name, n_loads, n_regs, loops, "none" if not findings else ""))
        for ln, s, regs, issued in findings[:12]:
            print("    line %d: %s   <- %s in flight since line %d" % (ln, s, ", ".join(regs), issued))
        bad += len(findings)
    for w in wanted:
        if not any(w in k for k in audited):
            print("audit: no kernel matching %r issues inline-asm loads in %s (renamed? the list in csrc/Makefile must follow)" % (w, path))
            bad += 1
    print("audit: %d kernels with inline-asm loads, %d findings" % (len(audited), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
