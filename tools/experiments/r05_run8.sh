cd $GRAFT_REPO_ROOT
for sl in 512 256 384 320; do
echo "CAPE_DW_SLOTS=$sl $(CAPE_DW_SLOTS=$sl python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]
print(d["ms_per_step"], "ms/step;", " ".join("%s %.1f" % (n.replace("dw_h2_kernel",""), k[n]["avg_us"]) for n in sorted(k) if n.startswith("dw_h2")), "| dw_reduce %.1f" % k["dw_reduce"]["total_us"])')"
done > gpurun_out/r05_e6_dw_slots.txt
cat gpurun_out/r05_e6_dw_slots.txt
