#!/bin/bash
# k_env_step of the walker mix with the light dimension sets on their own concurrent dispatch (SGRL_MIX_LIGHT=1) against the one-kernel launch
for v in 0 1 0 1; do
  SGRL_MIX_LIGHT=$v SGRL_BENCH_NO_CHILD=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --regions 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('SGRL_MIX_LIGHT=$v', 'env-steps/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'k_env_step ms %.3f' % d['roofline']['ms_per_launch'], 'dispatches', d['roofline']['dispatches_per_launch'])
"
done
