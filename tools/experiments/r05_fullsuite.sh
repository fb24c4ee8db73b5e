cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r05_parity_margins.tsv
export CAPE_PARITY_MARGINS=$GRAFT_REPO_ROOT/gpurun_out/r05_parity_margins.tsv
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r05_gputests.txt
python tools/parity_margins.py gpurun_out/r05_parity_margins.tsv > gpurun_out/r05_parity_margins.txt
tail -5 gpurun_out/r05_gputests.txt; head -6 gpurun_out/r05_parity_margins.txt
