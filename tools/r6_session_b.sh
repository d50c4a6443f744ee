#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6b
mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1; tail -n 4 $O/gpu_pytest.log
for rep in 1 2; do SGRL_GRAPH_UPDATES=1 timeout -k 10 200 python tools/update_profile.py 3d_walker_7_full 60 2>/dev/null | tail -n 1 | tee -a $O/update_graphed.txt; done
timeout -k 10 200 python tools/diag/numpy_surface_probe.py 2>/dev/null | tail -n 1 | tee $O/numpy_surface.txt
