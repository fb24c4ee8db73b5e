#!/bin/bash
# One-step kernel sequence + per-kernel stats of the replayed training step under rocprofv3 (kernel trace only).
#   gpurun -- 'bash tools/prof_step.sh <tag> [env assignments...]'   ->  gpurun_out/<tag>_kernel_stats.txt, <tag>_step_sequence.txt
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --steps 20 --warmup 3 > $O/${TAG}_under_rocprof.json 2>/dev/null
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/${TAG}_kernel_stats.txt
python $R/tools/rocpd_step_seq.py $DB $O/${TAG}_step_sequence.txt
head -45 $O/${TAG}_step_sequence.txt
