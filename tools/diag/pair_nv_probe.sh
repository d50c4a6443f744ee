#!/bin/bash
# walker mix, engine only, in the STATIONARY episode mix (WARM=400 steps before the clock): pairs up to nv 12 (walker_2 / walker_3)
# against pairs up to nv 15 (walker_4 too, on the dieted 20-row slab), with the per-set count of environments that took the
# matrix-free fallback (rows beyond the LDS rows)
for mx in 12 15 12 15; do
  echo "== walker SGRL_PAIR_MAXNV=$mx"
  SGRL_PAIR_MAXNV=$mx WARM=400 timeout -k 10 120 python3 tools/quick_bench.py 1024 20 2>&1 | grep -E "hip-event|ms/step|lds_bytes|walker_4|walker_3_left_leg" || exit 1
done
