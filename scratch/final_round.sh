R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scratch/profile_round.sh r2c > $R/gpurun_out/profile_r2c.log 2>&1
cd $R
mkdir -p gpurun_out/final
timeout 900 python bench.py 2> gpurun_out/final/bench.err | tail -1 > gpurun_out/final/bench.json
timeout 900 python bench.py --styleaug 2> gpurun_out/final/bench_styleaug.err | tail -1 > gpurun_out/final/bench_styleaug.json
timeout 900 python bench.py --model dann 2> gpurun_out/final/bench_dann.err | tail -1 > gpurun_out/final/bench_dann.json
timeout 900 python bench.py --model spn 2> gpurun_out/final/bench_spn.err | tail -1 > gpurun_out/final/bench_spn.json
for f in bench bench_styleaug bench_dann bench_spn; do cut -c1-260 gpurun_out/final/$f.json; done
