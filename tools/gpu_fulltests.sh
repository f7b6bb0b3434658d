#!/bin/bash
out=gpurun_out/${1:-tests}; mkdir -p $out
python -m pytest tests -q -m gpu -x --durations=15 > $out/pytest_gpu.log 2>&1; rc=$?
tail -25 $out/pytest_gpu.log
cp gpurun_out/b256_measured.json gpurun_out/timed_arithmetic_measured.json $out/ 2>/dev/null
exit $rc
