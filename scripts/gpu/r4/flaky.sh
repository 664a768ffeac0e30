#!/bin/bash
# Round 4: how often does tests/test_golden.py's float32 GPU case fail, alone and inside its file, with small results in
# host-visible memory (default) and in device memory?
OUT=gpurun_out/${1:-r4flaky}; mkdir -p $OUT
for hv in 1 0; do
  fails=0
  for i in $(seq 1 12); do
    LYNX_HOST_VISIBLE_RECORDS=$hv python -m pytest tests/test_golden.py tests/test_gpu_grad.py -q -m gpu -p no:cacheprovider -x > $OUT/run_${hv}_$i.log 2>&1 || { fails=$((fails+1)); cp $OUT/run_${hv}_$i.log $OUT/fail_${hv}_$fails.log; }
  done
  echo "LYNX_HOST_VISIBLE_RECORDS=$hv: $fails of 12 runs of test_golden.py + test_gpu_grad.py failed"
done
