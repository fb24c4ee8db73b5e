#!/bin/bash
# kernel stats + step sequence of an arbitrary bench.py configuration under rocprofv3 (kernel trace only)
#   gpurun -- 'bash tools/prof_cfg.sh <tag> <bench.py args...>'
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --steps 10 --warmup 2 "$@" > $O/${TAG}_under_rocprof.json 2>/dev/null
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/${TAG}_kernel_stats.txt
python $R/tools/rocpd_step_seq.py $DB $O/${TAG}_step_sequence.txt
head -40 $O/${TAG}_step_sequence.txt
