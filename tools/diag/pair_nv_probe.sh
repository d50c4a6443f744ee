#!/bin/bash
# engine only, in the STATIONARY episode mix (WARM=400 steps before the clock): pairs up to nv 12 (walker_2 / walker_3 / hopper_3)
# against pairs up to nv 15 (walker_4 / hopper_4 too, on the dieted pair slabs), with the per-set count of environments that took
# the matrix-free fallback (rows beyond the LDS rows).  $1 = family (walker: 8 x 1024, hopper: 3 x 1365)
fam=${1:-walker}
per=1024; [ $fam = hopper ] && per=1365
for mx in 12 15 12 15; do
  echo "== $fam SGRL_PAIR_MAXNV=$mx"
  QB_FAMILY=$fam SGRL_PAIR_MAXNV=$mx WARM=400 timeout -k 10 120 python3 tools/quick_bench.py $per 20 2>&1 | grep -E "hip-event|ms/step|lds_bytes|_4_|_3_" || exit 1
done
