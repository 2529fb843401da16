#!/bin/bash
# stall / instruction-mix counters per kernel for one KRN step (scratch tooling): separate --pmc passes, kernel-trace only
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${OUT:-pmc_sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pq$i
  SPB_EVENT_FORKS=1 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pq$i -o q -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $OUT/err$i.txt
  cp $(find /tmp/pq$i -name "*counter_collection.csv" | head -1) $OUT/pass$i.csv 2>/dev/null
done
ls -la $OUT
