cd $GRAFT_REPO_ROOT
for pf in 0 2 0 2; do
  echo "CAPE_DW_PF=$pf $(CAPE_DW_PF=$pf python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]
print(d["ms_per_step"], "ms/step;", " ".join("%s %.1f" % (n.replace("dw_h2_kernel",""), k[n]["avg_us"]) for n in sorted(k) if n.startswith("dw_h2")))')"
done > gpurun_out/r05_e3_dw_pf2.txt
CAPE_DW_PF=2 python -m pytest tests/test_gpu_h2.py -q -k "dw_h2" 2>&1 | tail -2 > gpurun_out/r05_h2_tests.txt
cat gpurun_out/r05_h2_tests.txt gpurun_out/r05_e3_dw_pf2.txt
