#!/bin/bash
# the light dimension sets as a second dispatch (SGRL_MIX_LIGHT=1) WITH stream priorities (SGRL_GROUP_PRIO=1:
# heavy dispatch high, light dispatch low), engine only, walker mix (8 x 1024) and hopper++ (3 x 1365)
for fam in walker hopper; do
  per=1024; [ $fam = hopper ] && per=1365
  for cfg in "0 0" "1 0" "1 1" "0 0" "1 1"; do
    set -- $cfg
    echo "== $fam SGRL_MIX_LIGHT=$1 SGRL_GROUP_PRIO=$2"
    QB_FAMILY=$fam SGRL_MIX_LIGHT=$1 SGRL_GROUP_PRIO=$2 timeout -k 10 120 python3 tools/quick_bench.py $per 20 2>&1 | grep -E "hip-event|ms/step" || exit 1
  done
done
