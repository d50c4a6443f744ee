#!/bin/bash
# round-6 slab diet: the config sweep on the default library and on the variant whose cheetah / humanoid step kernels
# read the int tables from global memory (+ 6-double contact frames), and the parity tests of those families on the variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6slab
for rep in 1 2; do
  SWEEP_OUT=r6slab/sweep_default_$rep.json timeout -k 10 300 python tools/config_sweep.py 2>/dev/null | cut -c1-330
  SGRL_HIP_LIB=$R/sgrl_amd/libsgrl_hip_itab.so SWEEP_OUT=r6slab/sweep_itab_$rep.json timeout -k 10 300 python tools/config_sweep.py 2>/dev/null | cut -c1-330
done
SGRL_HIP_LIB=$R/sgrl_amd/libsgrl_hip_itab.so timeout -k 10 600 python -m pytest tests/test_parity_matrix_gpu.py tests/test_engine_gpu.py tests/test_policy_states_gpu.py -m gpu -x -q > gpurun_out/r6slab/pytest_itab.log 2>&1; tail -n 3 gpurun_out/r6slab/pytest_itab.log
