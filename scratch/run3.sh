for cfg in "SPB_SPN_STEM=1" "SPB_SPN_STEM=0" "SPB_SPN_STEM=0 SPB_SPN_UPDATE_BLOCKS=256" "SPB_SPN_STEM=0 SPB_SPN_UPDATE_BLOCKS=384" "SPB_SPN_STEM=0 SPB_SPN_UPDATE_BLOCKS=512" "SPB_SPN_STEM=0 SPB_SPN_UPDATE_BLOCKS=1024" "SPB_SPN_STEM=0 SPB_SPN_UPDATE_PRIORITY=0" "SPB_SPN_STEM=0"; do
  env $cfg timeout 300 python bench.py --model spn --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
