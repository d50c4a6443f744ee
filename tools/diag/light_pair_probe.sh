#!/bin/bash
# all-light batches: the four-waves-per-SIMD light kernel (one environment per wavefront) against the family kernel with two environments
# per wavefront (SGRL_LIGHT_KERNEL=0), and against the family kernel unpaired (SGRL_LIGHT_KERNEL=0 SGRL_PAIR=0)
for names in 3d_walker_3_left_knee_right_knee 3d_walker_2_right_leg_left_knee 3d_hopper_3_shin; do
  for cfg in "1 1" "0 1" "0 0"; do
    set -- $cfg
    echo "== $names SGRL_LIGHT_KERNEL=$1 SGRL_PAIR=$2"
    QB_NAMES=$names SGRL_LIGHT_KERNEL=$1 SGRL_PAIR=$2 timeout -k 10 120 python3 tools/quick_bench.py 8192 20 2>&1 | grep -E "hip-event|lds_bytes" || exit 1
  done
done
