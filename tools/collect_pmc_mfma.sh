#!/bin/bash
# Matrix-core utilisation of the reference-pass kernels (north_star: "MFMA utilisation on the dense layers against gfx950 peak").
# One --pmc pass (kernel-trace only) over a workload that runs two reference passes of 5000 members and two lock-steps;
# counter collection serialises the dispatches, so every kernel has the chip to itself.  tools/summarize_pmc_mfma.py turns
# the CSV into profiles/rNN_pmc_mfma.json:  busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/pmc_mfma
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d "$O" -o kb -- python "$R/tools/kbench.py" --reps 2 --tslimit 2 > "$O/run.log" 2>&1
python "$R/tools/summarize_pmc_mfma.py" "$O" > "$O/summary.json"; cat "$O/summary.json" | head -80
find "$O" -name "*.csv" -size +20M -delete
