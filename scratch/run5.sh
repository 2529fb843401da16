export SPB_ONE_DEVICE=1 SPB_DIST_BACKEND=gloo
for m in "" "--model spn" "--styleaug" "--model dann"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline $m 2>&1 | grep -E '^\{"metric"|Error|error' | cut -c1-330
done
