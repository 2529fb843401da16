#!/bin/bash
C=$1
git checkout -q $C -- speedplusbaseline_amd/csrc include speedplusbaseline_amd/_lib.py speedplusbaseline_amd/ops.py
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|built" | head -3
/usr/local/graft/bin/gpurun --timeout 900 -- "timeout 500 python scratch/grad_probe.py ${2:-300} 2>&1 | grep -E '^(after|fp32|bf16|  [0-9]+ of)' | cut -c1-400" 2>&1 | grep -E "^(after|fp32|bf16|  [0-9]+ of)|status"
git checkout -q HEAD -- speedplusbaseline_amd/csrc include speedplusbaseline_amd/_lib.py speedplusbaseline_amd/ops.py
