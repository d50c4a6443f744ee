#!/bin/bash
# start / duration of every kernel of one SET forward (two streams), current build:  gpurun -- 'bash tools/diag/set_timeline.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/tools/quick_bench_set.py > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 $R/tools/set_timeline.py $f > $R/gpurun_out/r5_set_timeline.txt 2>&1
tail -5 $R/gpurun_out/r5_set_timeline.txt
