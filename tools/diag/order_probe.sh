#!/bin/bash
# k_env_step of the walker mix under three workgroup orders (SGRL_ORDER: 0 costliest dimension set first (default), 1 sets interleaved, 2 lightest first)
for v in 0 1 2 0 1 2; do
  SGRL_ORDER=$v SGRL_BENCH_NO_CHILD=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --regions 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('SGRL_ORDER=$v', 'env-steps/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'k_env_step ms %.3f' % d['roofline']['ms_per_launch'])
"
done
