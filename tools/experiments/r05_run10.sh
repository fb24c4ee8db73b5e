cd $GRAFT_REPO_ROOT
B="python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-extras --no-ab --no-roofline"
ms() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step")'; }
for rep in 1 2; do
for cfg in "512 0" "512 1" "256 0" "256 1"; do set -- $cfg
echo "SLOTS=$1 PAIR=$2 $(CAPE_DW_SLOTS=$1 CAPE_DW_PF=$2 $B 2>/dev/null | ms)"
done; done > gpurun_out/r05_e8_pair.txt
for sl in 512 256; do
echo "bf16 SLOTS=$sl $(CAPE_DW_SLOTS=$sl $B --dtype bf16 2>/dev/null | ms)"
echo "nz18gan32 SLOTS=$sl $(CAPE_DW_SLOTS=$sl $B --config CAPE_nz18_pose24_clotype8_male --gan --batch 32 --steps 20 --warmup 3 2>/dev/null | ms)"
echo "gan SLOTS=$sl $(CAPE_DW_SLOTS=$sl $B --gan 2>/dev/null | ms)"
done >> gpurun_out/r05_e8_pair.txt
CAPE_DW_PF=1 python -m pytest tests/test_gpu_h2.py -q -k "dw_h2" 2>&1 | tail -2 >> gpurun_out/r05_e8_pair.txt
cat gpurun_out/r05_e8_pair.txt
