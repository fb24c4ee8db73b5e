cd $GRAFT_REPO_ROOT
for pf in 0 16 32 64 48 96 80 112; do
  echo "CAPE_DW_PF=$pf $(CAPE_DW_PF=$pf python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]
print(d["ms_per_step"], "ms/step;", " ".join("%s %.1f" % (n.replace("dw_h2_kernel",""), k[n]["avg_us"]) for n in sorted(k) if n.startswith("dw_h2")))')"
done > gpurun_out/r05_e4_dw_phases.txt
cat gpurun_out/r05_e4_dw_phases.txt
