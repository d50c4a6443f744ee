#!/bin/bash
# two environments per wavefront (SGRL_PAIR) against one, engine only: walker mix 8 x 1024, hopper++ 3 x 1365, all-light batches
for fam in walker hopper; do
  per=1024; [ $fam = hopper ] && per=1365
  for p in 0 1 0 1; do
    echo "== $fam SGRL_PAIR=$p"
    QB_FAMILY=$fam SGRL_PAIR=$p timeout -k 10 120 python3 tools/quick_bench.py $per 20 2>&1 | grep -E "hip-event|ms/step|lds_bytes" || exit 1
  done
done
