#!/bin/bash
# A/B validation of the kernel-selection knobs on the GPU box (run from the repo root through gpurun):
#   1. library-level sweeps (tools/ubench/gemm_bench, dw_bench) with every knob off and on; checksums compared by
#      tools/ab_compare.py (the sweeps print a checksum of every output);
#   2. the GPU parity suite with the knobs given in $KNOBS exported;
#   3. paired bench runs, alternating default / knobs, so that box-to-box clock differences cancel.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'KNOBS="CAPE_DW_BF16X6=0 CAPE_GEMM_BF16X6_DUAL=0" bash tools/ab_knobs.sh'
# (round 2 flipped both knobs to on by default after this script's run, profiles/r02_ab_*; the default KNOBS below now
#  compare the shipped defaults against the exact-fp32 kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
KNOBS=${KNOBS:-"CAPE_DW_BF16X6=0 CAPE_GEMM_BF16X6_DUAL=0"}
PAIRS=${PAIRS:-3}
cd $R/tools/ubench
env CAPE_GEMM_BF16X6=0 ./gemm_bench > $O/gemm_fp32.txt 2>&1
env CAPE_GEMM_BF16X6=1 CAPE_GEMM_BF16X6_DUAL=0 ./gemm_bench > $O/gemm_split.txt 2>&1
env CAPE_GEMM_BF16X6=1 CAPE_GEMM_BF16X6_DUAL=1 ./gemm_bench > $O/gemm_split_dual.txt 2>&1
env CAPE_DW_BF16X6=0 ./dw_bench > $O/dw_fp32.txt 2>&1
env CAPE_DW_BF16X6=1 ./dw_bench > $O/dw_split.txt 2>&1
./gemm_bf16x3 v2 > $O/ubench_split_v2.txt 2>&1        # prepared kernel variants (RN split, swizzle/occ3, 8 waves, pre-split operands)
./gemm_bf16x3 bf16 > $O/ubench_bf16_storage.txt 2>&1  # bf16-storage contraction projection
cd $R
python tools/ab_compare.py $O/gemm_fp32.txt $O/gemm_split.txt $O/gemm_split_dual.txt | tee $O/compare_gemm.txt
python tools/ab_compare.py $O/dw_fp32.txt $O/dw_split.txt | tee $O/compare_dw.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  python -m pytest tests/test_gpu_split_ragged.py -x -q 2>&1 | tail -3 | tee $O/pytest_split_ragged.txt
  env $KNOBS python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_knobs.txt
fi
for i in $(seq 1 $PAIRS); do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-ab --no-extras 2>/dev/null | tail -1 >> $O/bench_default.jsonl
  env $KNOBS python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-ab --no-extras 2>/dev/null | tail -1 >> $O/bench_knobs.jsonl
done
tail -n +1 $O/ubench_split_v2.txt $O/ubench_bf16_storage.txt | cut -c1-230
python - <<PY
import json
for name in ("default", "knobs"):
    ms = [json.loads(l)["ms_per_step"] for l in open("$O/bench_%s.jsonl" % name) if l.strip().startswith("{")]
    print("%-8s ms/step:" % name, " ".join("%.4f" % m for m in ms), " median %.4f" % sorted(ms)[len(ms) // 2])
PY
