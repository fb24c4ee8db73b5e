#!/bin/bash
# HBM-traffic PMC passes only (FETCH_SIZE, WRITE_SIZE; one counter per run, no trace domains) + the adversarial-step bench.
#   gpurun --timeout 120 -- 'bash tools/collect_pmc_traffic.sh r01'
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc $ctrs -d /tmp/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --no-graph --steps 2 --warmup 1 > /dev/null 2>&1
  DBP=$(ls /tmp/prof_$name/*.db /tmp/prof_$name/*/*.db 2>/dev/null | head -1)
  python $R/tools/pmc_summary.py $DBP $O/pmc_$name.json
done
if [ "${2:-}" = "gan" ]; then python $R/bench.py --gan --no-cpu-baseline > $O/${TAG}_bench_gan.json 2>> $O/${TAG}_bench.err; fi
ls -la $O | tail -5
