#!/bin/bash
# SGRL_SPLIT_CLASS=1: the LDS occupancy classes of a family as separate concurrent dispatches of the same family kernel
for fam in humanoid cheetah; do
  per=512; [ $fam = cheetah ] && per=256
  for v in 0 1 0 1; do
    echo "== $fam SGRL_SPLIT_CLASS=$v"
    QB_FAMILY=$fam SGRL_SPLIT_CLASS=$v timeout -k 10 120 python3 tools/quick_bench.py $per 20 2>&1 | grep -E "hip-event|lds_bytes" || exit 1
  done
done
echo "== config sweep"
for v in 0 1; do SGRL_SPLIT_CLASS=$v timeout -k 10 300 python3 tools/config_sweep.py > /dev/null 2>&1; python3 -c "
import json
d = json.load(open('gpurun_out/config_sweep.json'))
print('SGRL_SPLIT_CLASS=$v', {k: (v['ms_k_env_step'], v['launch_groups'], round(v['env_steps_per_s'])) for k, v in d.items()})"; done
