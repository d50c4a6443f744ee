#!/bin/bash
# learning on the FINAL code of round 6, the other two single-family configs (64 environments per morphology, trainer defaults):
# walker++ seed 7 and humanoid++ seed 3 (the dieted humanoid step kernel), 800 s each, two processes sharing the GPU
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TAKEOFF_LAG=1
TAKEOFF_TAG=r6_final_walker TAKEOFF_FAMILY=walker python tools/takeoff_table.py shipped 400 800 7 &
TAKEOFF_TAG=r6_final_humanoid TAKEOFF_FAMILY=humanoid python tools/takeoff_table.py shipped 400 800 3 &
wait
