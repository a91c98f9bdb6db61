#!/usr/bin/env python3
"""hipcc does not model the loads inside an inline-asm statement: whenever its register allocation needs a move it copies a register
that such a load is still writing ("hazards").  k_fc_ring is built so that it has no reason to; this script checks
the ISA: inside the kernel's row loop (the innermost loop around the counted wait of the base rows) no v_mov / v_accvgpr_write may read
a register that a global_load / ds_read of that loop writes.
    make -C deep-neuroevolution_amd/csrc audit        (device-only assembly of engine.hip, then this script; exit code 1 = a copy found)"""
import re
import sys

path, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_fc_ring")
text = open(path).read().split("\n")
starts = [i for i, l in enumerate(text) if re.match(r"_ZN3dne\d+%s\w*:" % kernel, l)]
if not starts:
    sys.exit("audit: no kernel named %s in %s" % (kernel, path))
bad_total = 0
for st in starts:
    en = st
    while en < len(text) and ".end_amdhsa_kernel" not in text[en]:
        en += 1
    lines = text[st:en]
    wi = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)", l)]
    if not wi:
        sys.exit("audit: %s has no counted wait -- the row loop was not found" % lines[0].split(":")[0])
    # the row loop = the SMALLEST loop (a label and a later branch back to it) that holds at least four of the counted waits
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo_, hi_ = labels[m.group(1)], i
            if sum(1 for w in wi if lo_ <= w <= hi_) >= 4:
                loops.append((hi_ - lo_, lo_, hi_, m.group(1)))
    if not loops:
        sys.exit("audit: no loop around the counted waits of %s" % lines[0].split(":")[0])
    _, lo, hi, label = min(loops)
    body = lines[lo:hi + 1]
    dst = set()
    for l in body:
        for m in re.finditer(r"(?:global_load_dwordx4|global_load_dwordx2|ds_read\w*) v\[(\d+):(\d+)\]", l):
            dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.search(r"(?:global_load_dword|ds_read_b32) v(\d+),", l)
        if m:
            dst.add(int(m.group(1)))
    bad = []
    for i, l in enumerate(body):
        m = re.search(r"(?:v_mov_b(?:32|64)(?:_e32)?|v_accvgpr_write_b32) (?:v\[?[\d:]+\]?|a\d+), (v\[?[\d:]+\]?)\s*$", l)
        if m:
            r = re.findall(r"\d+", m.group(1))
            if any(x in dst for x in range(int(r[0]), int(r[-1]) + 1)):
                # a zero-fill of the accumulators at a sub-slice end reads a register that holds 0.0 at that point: written by a v_mov / v_accvgpr_read
                # of a constant within the previous 24 instructions, not by a load
                src = m.group(1)
                recent = "\n".join(body[max(0, i - 24):i])
                if re.search(r"(v_mov_b\d+(_e32)?|v_accvgpr_read_b32) %s, (0|a\d+|s\[?[\d:]+\]?)" % re.escape(src.split(":")[0].replace("v[", "v")), recent) or \
                   re.search(r"v_mov_b64(_e32)? %s, (0|s\[[\d:]+\])" % re.escape(src), recent):
                    continue
                bad.append((st + lo + i + 1, l.strip()))
    n_inst = len([l for l in body if l.startswith("\t") and not l.strip().startswith(";")])
    print("audit %s: row loop %s, %d instructions, %d asynchronously written registers, copies of them inside the loop: %s"
          % (lines[0].split(":")[0], label, n_inst, len(dst), bad if bad else "none"))
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
