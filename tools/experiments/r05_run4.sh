cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_h2.py -q 2>&1 | tail -6 > gpurun_out/r05_h2_tests.txt
python -m pytest tests/test_gpu_ops.py -q -k "twopass" 2>&1 | tail -4 >> gpurun_out/r05_h2_tests.txt
python -m pytest tests/test_gpu_model.py -q -k "batch16 or reproducible or operand_range" 2>&1 | tail -4 >> gpurun_out/r05_h2_tests.txt
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | tail -1 > gpurun_out/r05_bench_quick2.json
cat gpurun_out/r05_h2_tests.txt
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_quick2.json').read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_us'])[:14]: print('%-60s %3d x %7.2f us  %6.1f TF' % (k, v['launches'], v['avg_us'], v['tflops']))"
