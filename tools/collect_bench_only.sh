#!/bin/bash
# Short form of collect_profiles.sh: the bench JSON line and the rocprofv3 kernel-trace summary of the same command
# (no PMC passes, no secondary configurations).   gpurun --timeout 200 -- 'bash tools/collect_bench_only.sh r01'
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rm -rf /tmp/prof_ks && rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --steps 20 --warmup 3 > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
DB=$(ls /tmp/prof_ks/*.db /tmp/prof_ks/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/${TAG}_bench_kernel_stats.txt
python $R/tools/rocpd_step_seq.py $DB $O/${TAG}_step_sequence.txt
cut -c1-300 $O/${TAG}_bench.json
