#!/bin/bash
# Collect the committed profiles of a round on the GPU box (run through gpurun from the repo root):
#   KRN: kernel trace + stats of the bench command, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE), each with
#   --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes;
#   decoder / SPN: kernel stats, and one PMC pass with the matrix-core counters for the decoder.
# Outputs land in gpurun_out/prof_<tag>/.
# The --pmc passes run with SPB_EVENT_FORKS=1: counter collection lets one kernel of the device run at a time, and the plan's default
# fork (a one-wave gate kernel spinning on a device word, csrc/krn_plan.hip) would then keep the launch stream's kernel from starting.
set -u
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $OUT/fetch.err
cp $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $OUT/fetch.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $OUT/write.err
cp $(find /tmp/p_write -name "*counter_collection.csv" | head -1) $OUT/write.csv
python $ROOT/scratch/pmc_summary.py $OUT/kernel_stats.csv $OUT/fetch.csv $OUT/write.csv $OUT/pmc_traffic.json 4 > $OUT/pmc_summary.txt 2>&1
# decoder: kernel stats, then the matrix-core counters (own pass, kernel-trace only)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_gh -o gh -- python $ROOT/scratch/bench_ghiasi.py > $OUT/ghiasi_bench.txt 2> $OUT/ghiasi.err
cp $(find /tmp/p_gh -name "*kernel_stats.csv" | head -1) $OUT/ghiasi_kernel_stats.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_ghm -o ghm -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_mfma.err
cp $(find /tmp/p_ghm -name "*counter_collection.csv" | head -1) $OUT/ghiasi_mfma.csv
python $ROOT/scratch/mfma_summary.py $OUT/ghiasi_mfma.csv > $OUT/ghiasi_mfma_summary.txt 2>&1
# KRN: the same matrix-core counters (the 7x7 GEMMs)
SPB_EVENT_FORKS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_km -o km -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2> $OUT/krn_mfma.err
cp $(find /tmp/p_km -name "*counter_collection.csv" | head -1) $OUT/krn_mfma.csv
python $ROOT/scratch/mfma_summary.py $OUT/krn_mfma.csv > $OUT/krn_mfma_summary.txt 2>&1
# SPN
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_spn -o spn -- python $ROOT/bench.py --model spn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/spn_bench_under_rocprof.json 2> $OUT/spn.err
cp $(find /tmp/p_spn -name "*kernel_stats.csv" | head -1) $OUT/spn_kernel_stats.csv
# SPN HBM traffic, per dtype: two PMC passes of their own (kernel-trace only) over `--bare` runs of 3 + 1 steps, so that every launch of
# a pass belongs to one of its 4 steps (step_hbm_bytes = all bytes / 4)
for P in bf16 fp16; do
  SPB_EVENT_FORKS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_sf$P -o f -- python $ROOT/bench.py --model spn --precision $P --steps 3 --warmup 1 --bare > /dev/null 2> $OUT/spn_fetch_$P.err
  cp $(find /tmp/p_sf$P -name "*counter_collection.csv" | head -1) $OUT/spn_fetch_$P.csv
  SPB_EVENT_FORKS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_sw$P -o w -- python $ROOT/bench.py --model spn --precision $P --steps 3 --warmup 1 --bare > /dev/null 2> $OUT/spn_write_$P.err
  cp $(find /tmp/p_sw$P -name "*counter_collection.csv" | head -1) $OUT/spn_write_$P.csv
  python $ROOT/scratch/pmc_spn_summary.py $OUT/spn_fetch_$P.csv $OUT/spn_write_$P.csv $OUT/spn_${P}_pmc_traffic.json 4 > $OUT/spn_${P}_pmc_summary.txt 2>&1
done
# DANN (bs=48 source + 48 target): kernel stats of the bench command -- the round-2 domain-tail finding (a 5-workgroup kernel of
# ~1 ms on the critical path) was visible only in the per-family table of the bench line; this puts it under profiles/ too
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dann -o dann -- python $ROOT/bench.py --model dann --steps 20 --warmup 5 --no-cpu-baseline > $OUT/dann_bench_under_rocprof.json 2> $OUT/dann.err
cp $(find /tmp/p_dann -name "*kernel_stats.csv" | head -1) $OUT/dann_kernel_stats.csv
# DANN HBM traffic (bs=48+48): two PMC passes of their own
SPB_EVENT_FORKS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_df -o f -- python $ROOT/bench.py --model dann --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/dann_fetch.err
cp $(find /tmp/p_df -name "*counter_collection.csv" | head -1) $OUT/dann_fetch.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_dw -o w -- python $ROOT/bench.py --model dann --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/dann_write.err
cp $(find /tmp/p_dw -name "*counter_collection.csv" | head -1) $OUT/dann_write.csv
python $ROOT/scratch/pmc_summary.py $OUT/dann_kernel_stats.csv $OUT/dann_fetch.csv $OUT/dann_write.csv $OUT/dann_pmc_traffic.json 4 > $OUT/dann_pmc_summary.txt 2>&1
# decoder HBM traffic
SPB_EVENT_FORKS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_gf -o f -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_fetch.err
cp $(find /tmp/p_gf -name "*counter_collection.csv" | head -1) $OUT/ghiasi_fetch.csv
SPB_EVENT_FORKS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_gw -o w -- python $ROOT/scratch/bench_ghiasi.py > /dev/null 2> $OUT/ghiasi_write.err
cp $(find /tmp/p_gw -name "*counter_collection.csv" | head -1) $OUT/ghiasi_write.csv
python $ROOT/scratch/pmc_ghiasi_summary.py $OUT/ghiasi_fetch.csv $OUT/ghiasi_write.csv $OUT/ghiasi_pmc_traffic.json > $OUT/ghiasi_pmc_summary.txt 2>&1
ls -la $OUT
