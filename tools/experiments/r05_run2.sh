cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_adam.py -q 2>&1 | tail -15 > gpurun_out/r05_adam_tests.txt
rm -f gpurun_out/r05_parity_margins.tsv
export CAPE_PARITY_MARGINS=$GRAFT_REPO_ROOT/gpurun_out/r05_parity_margins.tsv CAPE_PARITY_COLLECT=1
python -m pytest tests/test_gpu_ops.py -q -k "test_cheb_conv_fwd_bwd" 2>&1 | tail -15 > gpurun_out/r05_parity_ops.txt
python -m pytest tests/test_gpu_model.py -q -k "test_model_matches_reference_golden or test_full_model_forward_backward or test_operand_range or test_batch16" 2>&1 | tail -40 > gpurun_out/r05_parity_model.txt
python -m pytest tests/test_gpu_knobs.py -q -k "test_reference_goldens_under_each_arithmetic" 2>&1 | tail -30 > gpurun_out/r05_parity_legs.txt
tail -5 gpurun_out/r05_adam_tests.txt gpurun_out/r05_parity_ops.txt gpurun_out/r05_parity_model.txt gpurun_out/r05_parity_legs.txt
