#!/bin/bash
# kernel tables of config 5's collection step with the round bookkeeping as tensor operations (0) and as one launch (1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6collect; mkdir -p $O /tmp/cprof
cd /tmp && export TMPDIR=/tmp
for arm in 0 1; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof/a$arm -o c -- python3 $R/tools/diag/collect_one.py $arm 40 > $O/arm$arm.log 2> /tmp/cprof/err$arm || { tail -n 5 /tmp/cprof/err$arm; exit 1; }
  tail -n 1 $O/arm$arm.log
  f=$(find /tmp/cprof/a$arm -name '*kernel_stats.csv' | head -n 1); cp "$f" $O/arm${arm}_kernel_stats.csv
  f=$(find /tmp/cprof/a$arm -name '*kernel_trace.csv' | head -n 1); cp "$f" $O/arm${arm}_kernel_trace.csv
done
cd $R
