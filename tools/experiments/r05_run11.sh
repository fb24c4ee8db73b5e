cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_h2.py tests/test_gpu_split_ragged.py -q 2>&1 | tail -4 > gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_ops.py -q -k "twopass" 2>&1 | tail -3 >> gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_model.py -q -k "batch16 or reproducible or operand_range or manual_update or two_phase or nz18_gan" 2>&1 | tail -3 >> gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -3 >> gpurun_out/r05_t.txt
B="python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-extras --no-ab --no-roofline"
ms() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step")'; }
echo "fp32 $($B 2>/dev/null | ms)" >> gpurun_out/r05_t.txt
echo "bf16 $($B --dtype bf16 2>/dev/null | ms)" >> gpurun_out/r05_t.txt
echo "nz18gan32 $($B --config CAPE_nz18_pose24_clotype8_male --gan --batch 32 --steps 20 --warmup 3 2>/dev/null | ms)" >> gpurun_out/r05_t.txt
cat gpurun_out/r05_t.txt
