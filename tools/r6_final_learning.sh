#!/bin/bash
# learning on the FINAL code of round 6 (trainer defaults: lagged round flag, twin targets, dieted humanoid kernel): config 5 seeds 5 and 7 for
# 1000 s, hopper++ seed 3 for 420 s
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TAKEOFF_LAG=1
TAKEOFF_TAG=r6_final python tools/takeoff_table.py shipped 200 1000 5 7 &
TAKEOFF_TAG=r6_final_hopper TAKEOFF_FAMILY=hopper python tools/takeoff_table.py shipped 400 420 3 &
wait
