#!/bin/bash
# BASELINE configs[1] evidence (run on the GPU box): timing, rocprofv3 kernel summary and HBM traffic counters of the single
# K = 6 layer in both forms.   bash tools/collect_config1.sh r03
set -u
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# single Chebyshev K=6 layer (BASELINE configs[1]): on-chip recurrence (default) and the materialised K-stack form, each with
# a rocprofv3 kernel summary and the HBM traffic counters (separate --pmc passes)
cd $R && python tools/bench_config2.py > $O/${TAG}_config1_fused.json 2>> $O/${TAG}_bench.err
cd $R && CAPE_FUSED_RECURRENCE=0 python tools/bench_config2.py > $O/${TAG}_config1_materialised.json 2>> $O/${TAG}_bench.err
cd /tmp
for mode in 1 0; do
  name=$([ $mode = 1 ] && echo fused || echo materialised)
  rm -rf /tmp/prof_c1 && CAPE_FUSED_RECURRENCE=$mode rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -o r -- python $R/tools/bench_config2.py > /dev/null 2>&1
  DBC=$(ls /tmp/prof_c1/*.db /tmp/prof_c1/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DBC $O/${TAG}_config1_${name}_kernel_stats.txt
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    pn=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/prof_c1p && CAPE_FUSED_RECURRENCE=$mode CAPE_CONFIG2_EAGER=1 rocprofv3 --pmc $ctrs -d /tmp/prof_c1p -o r -- python $R/tools/bench_config2.py > /dev/null 2>&1
    DBP=$(ls /tmp/prof_c1p/*.db /tmp/prof_c1p/*/*.db 2>/dev/null | head -1)
    python $R/tools/pmc_summary.py $DBP $O/${TAG}_config1_${name}_pmc_$pn.json
  done
done
cd $R
