cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_adam.py -x -q 2>&1 | tail -15 > gpurun_out/r05_adam_tests.txt
for mb in 0 48 96 160; do
  echo "CAPE_DW_FLUSH_MB=$mb $(CAPE_DW_FLUSH_MB=$mb python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step")')"
done > gpurun_out/r05_e1_dw_flush.txt
cat gpurun_out/r05_adam_tests.txt gpurun_out/r05_e1_dw_flush.txt
