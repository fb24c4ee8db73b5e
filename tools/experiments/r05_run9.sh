cd $GRAFT_REPO_ROOT
for sl in 512 256 512 256 512 256; do
echo "CAPE_DW_SLOTS=$sl $(CAPE_DW_SLOTS=$sl python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-ab --no-roofline 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step")')"
done > gpurun_out/r05_e7_dw_slots_ab.txt
cat gpurun_out/r05_e7_dw_slots_ab.txt
