#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root (gpurun): bench JSONs, rocprofv3 kernel stats,
# one-step dispatch sequence and the three PMC passes (separate runs, no trace domains mixed in).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python $R/bench.py --gan --no-cpu-baseline --no-ab --no-extras > $O/${TAG}_bench_gan.json 2>> $O/${TAG}_bench.err
# BASELINE configs[4] (bf16 storage, one GPU's shard of 16), configs[3] (nz18 group-norm generator + discriminator, batch 32)
python $R/bench.py --dtype bf16 --no-cpu-baseline > $O/${TAG}_bench_bf16.json 2>> $O/${TAG}_bench.err
python $R/bench.py --config CAPE_nz18_pose24_clotype8_male --gan --batch 32 --no-cpu-baseline --no-ab --no-extras > $O/${TAG}_bench_nz18_gan_b32.json 2>> $O/${TAG}_bench.err
python $R/bench.py --config CAPE_nz18_pose24_clotype8_male --batch 16 --no-cpu-baseline --no-ab --no-extras > $O/${TAG}_bench_nz18_cvae_b16.json 2>> $O/${TAG}_bench.err
python $R/bench.py --host-inputs --no-cpu-baseline --no-roofline > $O/${TAG}_bench_host_inputs.json 2>> $O/${TAG}_bench.err
rm -rf /tmp/prof_ks && rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --steps 20 --warmup 3 > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
DB=$(ls /tmp/prof_ks/*.db /tmp/prof_ks/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB $O/${TAG}_bench_kernel_stats.txt
python $R/tools/rocpd_step_seq.py $DB $O/${TAG}_step_sequence.txt
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf /tmp/prof_$name
  rocprofv3 --pmc $ctrs -d /tmp/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --no-ab --no-extras --no-graph --steps 2 --warmup 1 > /dev/null 2>&1
  DBP=$(ls /tmp/prof_$name/*.db /tmp/prof_$name/*/*.db 2>/dev/null | head -1)
  python $R/tools/pmc_summary.py $DBP $O/pmc_$name.json
done
python $R/tools/pmc_merge.py $O/pmc_fetch.json $O/pmc_write.json $O/pmc_sq.json $O/${TAG}_pmc_summary.json
cp $O/${TAG}_pmc_summary.json $O/pmc_summary.json      # the copy bench.py reads (stamped with the kernel-source fingerprint)
python $R/tools/hbm_bw_table.py $O/${TAG}_pmc_summary.json $O/${TAG}_bench_kernel_stats.txt > $O/${TAG}_hbm_bandwidth_by_kernel.txt
ls -la $O | tail -15
bash $R/tools/collect_config1.sh $TAG
cd $R
CAPE_DIST_BACKEND=gloo CAPE_FORCE_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --no-roofline > $O/${TAG}_bench_2rank_1gpu.json 2>> $O/${TAG}_bench.err
tail -c 400 $O/${TAG}_bench_2rank_1gpu.json
make -C $R/tools/ubench > /dev/null 2>&1
(cd $R/tools/ubench && ./mfma_peak) > $O/${TAG}_ubench_mfma_peak.txt 2>&1
# split data-parallel step with the real collective backend (one-rank RCCL group, collectives forced on)
python $R/tools/dp_selftest.py 2>&1 | grep -v "UserWarning\|run_backward\|amdgpu.ids\|socket.cpp\|^$" > $O/${TAG}_dp_selftest.txt
# the library's two-piece / six-product forward kernels side by side over the layer shapes of the benchmarked model
(cd $R/tools/ubench && timeout 180 ./h2_bench) > $O/${TAG}_h2_bench_final.txt 2>&1
# same-box A/B of this round's switches on the replayed step
for kv in "CAPE_H2=1" "CAPE_FUSE_PREP_SPMM=0" "CAPE_H2X=0" "CAPE_DW_V4=0" "CAPE_H2=0" "CAPE_FUSE_ACT_GRAD=0" "CAPE_FC_MFMA=0" "CAPE_H2=1"; do
  echo "$kv $(env $kv python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-ab --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms/step")')"
done > $O/${TAG}_switch_ab.txt
cd $R && timeout 120 python tools/cheb_fused_phases.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_cheb_fused_phases.txt
ls $O | grep "^${TAG}_" | wc -l
