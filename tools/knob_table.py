#!/usr/bin/env python3
"""Every DNE_* environment knob the engine reads (csrc/engine.hip: env_int(name, lo, hi, &h->field)), with its default (the field's
initialiser, or the per-kind assignment in dne_create) and the comment at the field -- printed as the markdown table of DESIGN.md's
knob appendix.   python tools/knob_table.py > /tmp/knobs.md"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "deep-neuroevolution_amd", "csrc", "engine.hip")).read()
fields = {}
for m in re.finditer(r"^\s*(?:int|bool)\s+([^;]+);\s*(?://\s*(.*))?$", src, re.M):
    comment = (m.group(2) or "").strip()
    for part in m.group(1).split(","):
        mm = re.match(r"\s*(\w+)\s*=\s*([^,]+)", part)
        if mm:
            fields.setdefault(mm.group(1), (mm.group(2).strip(), comment))
rows = []
for m in re.finditer(r'env_int\("(DNE_\w+)",\s*([^,]+),\s*([^,]+),\s*&h->(\w+)\)', src):
    name, lo, hi, field = m.groups()
    default, comment = fields.get(field, ("?", ""))
    comment = re.sub(r"^DNE_\w+(?:\s*/\s*\w+)*\s*:?\s*", "", comment)
    rows.append((name, default, "%s .. %s" % (lo.strip(), hi.strip()), comment))
print("| knob | default | range | what it selects |")
print("|---|---|---|---|")
for r in sorted(set(rows)):
    print("| `%s` | %s | %s | %s |" % (r[0], r[1], r[2].replace("|", "\\|"), r[3].replace("|", "\\|")))
