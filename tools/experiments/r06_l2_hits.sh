#!/bin/bash
# Round 6, experiment 1: what bounds the two-piece contraction kernels -- L2 hit rate, L1 -> L2 request latency, vector-cache
# stalls -- on the library's own kernels at the widest layer shapes (tools/ubench/h2_bench, dw_h2_bench: C-ABI only, no torch).
# Counter passes only (no trace domains).   gpurun --timeout 600 -- 'bash tools/experiments/r06_l2_hits.sh'
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R/tools/ubench && ./h2_bench 20 0 0,3,5,8 && ./dw_h2_bench 20 0,1,3) > $O/r06_l2_unprofiled_times.txt 2>&1
rocprofv3 -L > $O/r06_counter_list.txt 2>&1
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/p$i
  (cd $R/tools/ubench && rocprofv3 --pmc $ctrs -d /tmp/p$i -o r -- ./h2_bench 2 0 0,3,5,8 > /dev/null 2> $O/r06_l2_pass$i.err)
  rm -rf /tmp/q$i
  (cd $R/tools/ubench && rocprofv3 --pmc $ctrs -d /tmp/q$i -o r -- ./dw_h2_bench 2 0,1,3 > /dev/null 2>> $O/r06_l2_pass$i.err)
done
python $R/tools/pmc_by_grid.py $(ls /tmp/p*/*.db /tmp/p*/*/*.db /tmp/q*/*.db /tmp/q*/*/*.db 2>/dev/null) > $O/r06_l2_hits_raw.json
python - <<PY
import json
d = json.load(open("$O/r06_l2_hits_raw.json"))
out = {}
for k, v in d.items():
    if not any(t in k for t in ("gemm_h2", "dw_h2", "gemm_split", "dw_reduce")):
        continue
    e = dict(v)
    g = v.get
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        e["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum") is not None:
        e["l1_to_l2_read_latency_cycles"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), 1)
    if g("GRBM_GUI_ACTIVE"):
        e["kernel_cycles"] = round(g("GRBM_GUI_ACTIVE") / 8)
        if g("TCP_TCC_READ_REQ_sum"):
            e["l1_to_l2_read_req_per_cu_cycle"] = round(g("TCP_TCC_READ_REQ_sum") / 256 / e["kernel_cycles"], 4)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            e["mfma_util"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * e["kernel_cycles"]), 4)
    out[k] = e
json.dump(out, open("$O/r06_l2_hits.json", "w"), indent=1, sort_keys=True)
for k, e in out.items():
    print(k, {x: e[x] for x in e if x in ("l2_hit_rate", "l1_to_l2_read_latency_cycles", "kernel_cycles", "l1_to_l2_read_req_per_cu_cycle", "mfma_util", "dispatches")})
PY
cat $O/r06_l2_unprofiled_times.txt
tail -3 $O/r06_l2_pass*.err | tail -20
