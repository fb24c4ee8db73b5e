cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_adam.py -q 2>&1 | tail -25 > gpurun_out/r05_adam_tests.txt
python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -25 > gpurun_out/r05_dist_tests.txt
python -m pytest tests/test_gpu_model.py -q -k "manual_update or two_phase or reproducible" 2>&1 | tail -8 > gpurun_out/r05_step_tests.txt
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab 2>/dev/null | tail -1 > gpurun_out/r05_bench_quick.json
tail -n 6 gpurun_out/r05_adam_tests.txt gpurun_out/r05_dist_tests.txt gpurun_out/r05_step_tests.txt
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_quick.json').read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['mfma_util'], d['config']['dp'])"
