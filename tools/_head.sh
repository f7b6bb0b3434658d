set -o pipefail
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py tests/test_b256_gpu.py tests/test_reproducible_gpu.py -x -q 2>&1 | tail -5 || exit 1
tools/gpu_ab.sh r05_head 3 "nosplit:--tune bwd_split_max_tiles=0 --flag functional.HEAD_KSPLIT=0" "bwdsplit:--flag functional.HEAD_KSPLIT=0" "both:"
