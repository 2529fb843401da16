for i in 1 2 3; do
  timeout 1200 python -m pytest tests/test_parity_conditioned_gpu.py -q -s 2>&1 | grep -E "^E  |AssertionError|conditioning \(|conditioned train pass|per-layer|emulated-bf16|gradient: cosine|eval|passed|failed" | head -16
  echo "--- run $i"
done
