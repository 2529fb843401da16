#!/bin/bash
# build the kernels of commit $1 into the working tree, run the conditioned train-pass diagnostic on the GPU box, restore HEAD's sources
C=$1
git checkout -q $C -- speedplusbaseline_amd/csrc include speedplusbaseline_amd/_lib.py speedplusbaseline_amd/ops.py
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|built" | head -3
/usr/local/graft/bin/gpurun --timeout 900 -- 'timeout 400 python -m pytest tests/test_parity_conditioned_gpu.py -q -m gpu -s -k train_pass 2>&1 | grep -E "gradient vs|lowest per|did not settle|passed|failed|rror" | cut -c1-300 | tail -5' 2>&1 | grep -E "gradient vs|lowest|settle|status|passed|failed|rror"
git checkout -q HEAD -- speedplusbaseline_amd/csrc include speedplusbaseline_amd/_lib.py speedplusbaseline_amd/ops.py
