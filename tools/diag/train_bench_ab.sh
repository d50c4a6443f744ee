#!/bin/bash
# config-5 trainer bench with the fused round bookkeeping (default) and with the tensor form, alternating, same box
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out/r6verify
export SGRL_TUNE_GEMMS=0
for rep in 1 2; do
  timeout -k 10 300 python3 tools/train_bench.py 2>/dev/null | tail -n 1 | cut -c1-200 | sed 's/^/fused  /' | tee -a gpurun_out/r6verify/train_bench_ab.txt || exit 1
  timeout -k 10 300 python3 -c "
import sys, runpy
sys.path.insert(0, '.')
import sgrl_amd.rollout as r
r.FUSED_RECORD = False
sys.argv = ['tools/train_bench.py']
runpy.run_path('tools/train_bench.py', run_name='__main__')" 2>/dev/null | tail -n 1 | cut -c1-200 | sed 's/^/tensor /' | tee -a gpurun_out/r6verify/train_bench_ab.txt || exit 1
done
