#!/bin/bash
# copy the summaries of a profile round (gpurun_out/prof_<tag>, scratch/profile_round.sh) into profiles/ under the round's prefix:
#   bash scratch/collect_profiles.sh r6
T=${1:-r6}; S=gpurun_out/prof_$T; D=profiles
cp $S/kernel_stats.csv $D/${T}_rocprofv3_kernel_stats.csv
cp $S/bench_under_rocprof.json $D/${T}_bench_under_rocprof.json
cp $S/pmc_traffic.json $D/${T}_pmc_traffic.json; cp $S/pmc_summary.txt $D/${T}_pmc_summary.txt
cp $S/krn_mfma_summary.txt $D/${T}_krn_mfma_summary.txt
cp $S/krn_launches.txt $D/${T}_krn_launches.txt
cp $S/trace/chain.txt $D/${T}_krn_chain.txt
for w in ghiasi_bench.txt ghiasi_kernel_stats.csv ghiasi_mfma_summary.txt ghiasi_pmc_summary.txt ghiasi_pmc_traffic.json \
         spn_kernel_stats.csv spn_bench_under_rocprof.json spn_bf16_pmc_summary.txt spn_bf16_pmc_traffic.json spn_fp16_pmc_summary.txt spn_fp16_pmc_traffic.json \
         dann_kernel_stats.csv dann_bench_under_rocprof.json dann_pmc_summary.txt dann_pmc_traffic.json; do
  [ -f $S/$w ] && cp $S/$w $D/${T}_$w
done
ls -la $D/${T}_* | wc -l
