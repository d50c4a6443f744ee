#!/bin/bash
# round-6 slab diet, second form: the humanoid family only (int tables from global memory + 6-double contact frames + the row cut that
# gives the MOST residents, floor 19) -- humanoid_9 then fits seven times into a CU (23 008 B) instead of six
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6slab
export SWEEP_ONLY=humanoid,config5
for rep in 1 2 3; do
  SWEEP_OUT=r6slab/sweep2_default_$rep.json timeout -k 10 300 python tools/config_sweep.py 2>/dev/null | cut -c1-330
  SGRL_HIP_LIB=$R/sgrl_amd/libsgrl_hip_itab.so SWEEP_OUT=r6slab/sweep2_itab_$rep.json timeout -k 10 300 python tools/config_sweep.py 2>/dev/null | cut -c1-330
done
SGRL_HIP_LIB=$R/sgrl_amd/libsgrl_hip_itab.so timeout -k 10 600 python -m pytest tests/test_parity_matrix_gpu.py tests/test_policy_states_gpu.py "tests/test_engine_gpu.py::test_free_running_1000_steps_within_1e4" tests/test_engine_gpu.py -k "not native_library" -m gpu -q > gpurun_out/r6slab/pytest_itab2.log 2>&1; tail -n 3 gpurun_out/r6slab/pytest_itab2.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6slab/gpu_pytest_default.log 2>&1; tail -n 4 gpurun_out/r6slab/gpu_pytest_default.log
