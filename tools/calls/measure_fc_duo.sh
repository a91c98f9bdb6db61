#!/bin/bash
# round-5 measurement call 1: what k_fc_duo waits for (tick clock + TA/TCP/TCC counters)
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
CLK=$R/deep-neuroevolution_amd/csrc/libdne_hip_clock.so
cd $R
for cfg in "X=0" "DNE_DUO_FAT=0" ; do
  env $cfg DNE_LIB_PATH=$CLK DNE_NSUB=1 timeout 200 python tools/duo_tick_clock.py > "$O/tick.$cfg.json" 2> "$O/tick.$cfg.err"
done
env DNE_LIB_PATH=$CLK DNE_NSUB=1 timeout 200 python tools/duo_tick_clock.py --pairs 1250 > "$O/tick.p1250.json" 2> "$O/tick.p1250.err"
env DNE_LIB_PATH=$CLK DNE_NSUB=4 timeout 200 python tools/duo_tick_clock.py > "$O/tick.nsub4.json" 2> "$O/tick.nsub4.err"
head -c 1500 "$O/tick.X=0.json"; tail -3 "$O/tick.X=0.err"
bash tools/collect_pmc_duo_mem.sh $TAG > $O/pmc.log 2>&1
python tools/summarize_pmc_duo_mem.py $O/pmc_duo_mem r05 > $O/pmc_summary.txt 2>&1
cp profiles/r05_pmc_fc_duo_mem.json $O/ 2>/dev/null
tail -60 $O/pmc_summary.txt
