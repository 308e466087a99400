#!/bin/bash
# Per-file wall time of the GPU test suite (run on the GPU box): tools/gpu_test_times.sh > gpurun_out/test_times.log
for f in tests/test_gpu_surface.py tests/test_policies.py tests/test_encoders.py tests/test_golden_trace.py tests/test_gpu_clearance.py tests/test_gpu_parity.py; do
  s=$(date +%s)
  timeout -k 10 ${PER_FILE_LIMIT:-900} python -m pytest $f -m gpu -q -s --durations=12 2>&1 | grep -v "^$" | tail -${TAIL:-22}
  echo "=== $f: $(( $(date +%s) - s )) s"
done
