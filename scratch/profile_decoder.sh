#!/bin/bash
# decoder-only part of scratch/profile_round.sh: kernel stats, matrix-core busy counters, FETCH / WRITE passes (separate runs)
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_gh /tmp/p_ghm /tmp/p_gf /tmp/p_gw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_gh -o gh -- python $ROOT/scratch/bench_ghiasi.py > $OUT/ghiasi_bench.txt 2> $OUT/ghiasi.err
cp $(find /tmp/p_gh -name "*kernel_stats.csv" | head -1) $OUT/ghiasi_kernel_stats.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_ghm -o ghm -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_mfma.err
cp $(find /tmp/p_ghm -name "*counter_collection.csv" | head -1) $OUT/ghiasi_mfma.csv
python $ROOT/scratch/mfma_summary.py $OUT/ghiasi_mfma.csv > $OUT/ghiasi_mfma_summary.txt 2>&1
SPB_EVENT_FORKS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_gf -o f -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_fetch.err
cp $(find /tmp/p_gf -name "*counter_collection.csv" | head -1) $OUT/ghiasi_fetch.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_gw -o w -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_write.err
cp $(find /tmp/p_gw -name "*counter_collection.csv" | head -1) $OUT/ghiasi_write.csv
python $ROOT/scratch/pmc_ghiasi_summary.py $OUT/ghiasi_fetch.csv $OUT/ghiasi_write.csv $OUT/ghiasi_pmc_traffic.json > $OUT/ghiasi_pmc_summary.txt 2>&1
python $ROOT/scratch/bench_ghiasi.py > $OUT/ghiasi_bench_plain.txt 2>&1
rm -f $OUT/ghiasi_fetch.csv $OUT/ghiasi_write.csv $OUT/ghiasi_mfma.csv
cat $OUT/ghiasi_mfma_summary.txt $OUT/ghiasi_pmc_summary.txt; head -3 $OUT/ghiasi_bench_plain.txt
