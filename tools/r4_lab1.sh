#!/bin/bash
# round 4, first GPU lab call: fused chain kernels vs single products, SET forward with / without them, SET parity tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 tools/chain_lab.exe > gpurun_out/r4_chain_lab.txt 2>&1 || { echo "chain lab failed"; tail -5 gpurun_out/r4_chain_lab.txt; exit 1; }
cat gpurun_out/r4_chain_lab.txt
SGRL_SET_CHAIN=0 timeout -k 10 300 python tools/quick_bench_set.py > gpurun_out/r4_qb_chain0.txt 2>&1 || { tail -5 gpurun_out/r4_qb_chain0.txt; exit 1; }
timeout -k 10 300 python tools/quick_bench_set.py > gpurun_out/r4_qb_chain1.txt 2>&1 || { tail -5 gpurun_out/r4_qb_chain1.txt; exit 1; }
tail -1 gpurun_out/r4_qb_chain0.txt gpurun_out/r4_qb_chain1.txt
timeout -k 10 900 python -m pytest tests/test_set_gpu.py tests/test_split_products_gpu.py tests/test_set_critic.py -m gpu -x -q > gpurun_out/r4_set_tests.txt 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r4_set_tests.txt
