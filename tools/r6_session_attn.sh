#!/bin/bash
# round 6: the limb-attention kernels of the update as one workgroup per (environment, head) with all loads in one batch --
# the tests of everything that touches the update, then the graphed walker_7 update twice and a cheetah_14 one (L = 14)
set -o pipefail
O=gpurun_out/r6attn; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_train_ops_gpu.py tests/test_td3_update_init.py tests/test_td3_update.py tests/test_train_loop_gpu.py tests/test_wgrad_stress_gpu.py tests/test_set_critic.py -m gpu -q > $O/pytest.log 2>&1; RC=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -n 8
if [ $RC -ne 0 ]; then tail -n 40 $O/pytest.log | cut -c1-300; exit $RC; fi
for rep in 1 2; do SGRL_GRAPH_UPDATES=1 timeout -k 10 200 python tools/update_profile.py 3d_walker_7_full 60 2>/dev/null | tail -n 1 | tee -a $O/update_graphed.txt; done
SGRL_GRAPH_UPDATES=1 timeout -k 10 200 python tools/update_profile.py 3d_cheetah_14_full 40 2>/dev/null | tail -n 1 | tee -a $O/update_graphed_cheetah14.txt
