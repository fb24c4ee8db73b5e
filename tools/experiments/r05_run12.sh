cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_h2.py tests/test_gpu_split_ragged.py -q 2>&1 | tail -4 > gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_ops.py -q -k "twopass" 2>&1 | tail -3 >> gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_model.py -q -k "batch16 or reproducible or operand_range" 2>&1 | tail -3 >> gpurun_out/r05_t.txt
python -m pytest tests/test_gpu_knobs.py -q -k "fused_activation" 2>&1 | tail -3 >> gpurun_out/r05_t.txt
for i in 1 2; do
python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]
print(d["ms_per_step"], "ms/step;", " ".join("%s %.1f" % (n.replace("dw_h2_kernel",""), k[n]["avg_us"]) for n in sorted(k) if n.startswith("dw_h2")), "| dw_reduce %.1f" % k["dw_reduce"]["total_us"], "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"])' >> gpurun_out/r05_t.txt
done
cat gpurun_out/r05_t.txt
