#!/bin/bash
# round-6 GPU session A: GPU suite on the current tree, the graphed update in four arms, launch sources, the NumPy surface, one bench line
# (HISTORICAL: the switches its arms toggle -- SGRL_FAN_OUT, SGRL_TWIN_TARGETS, SGRL_W32_KT -- were removed after this measurement,
# profiles/r6_update_arms.txt; today all four arms run the same code)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/gpu_pytest.log; tail -n 3 $O/gpu_pytest.log
export SGRL_GRAPH_UPDATES=1
for arm in "shipped" "nofan:SGRL_FAN_OUT=0" "optins:SGRL_TWIN_TARGETS=1 SGRL_W32_KT=7" "optins_nofan:SGRL_TWIN_TARGETS=1 SGRL_W32_KT=7 SGRL_FAN_OUT=0"; do
  name=${arm%%:*}; vars=${arm#*:}; [ "$vars" = "$arm" ] && vars=""
  for rep in 1 2; do
    env $vars timeout -k 10 200 python tools/update_profile.py 3d_walker_7_full 60 2>/dev/null | tail -n 1 | sed "s/^/$name $rep: /" | tee -a $O/update_arms.txt
  done
done
unset SGRL_GRAPH_UPDATES
timeout -k 10 300 python tools/diag/update_launch_sources.py > $O/update_launch_sources.txt 2>&1; tail -n 3 $O/update_launch_sources.txt
timeout -k 10 200 python tools/diag/numpy_surface_probe.py 2>/dev/null | tail -n 1 | tee $O/numpy_surface.txt
timeout -k 10 400 python bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/bench.json; python - <<PY
import json
j = json.load(open("$O/bench.json"))
print("bench: %.0f env-steps/s, %.3f ms/step, k_env_step %.3f ms (in rollout %.3f), set fwd %.3f ms" % (j["value"], j["ms_per_step"], j["roofline"]["ms_per_launch"], j["roofline"]["ms_per_launch_in_rollout"], j["set_actor"]["ms_per_forward"]))
PY
