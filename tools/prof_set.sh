#!/bin/bash
# per-kernel times of the SET forward alone (walker mix, 8192 envs): rocprofv3 --kernel-trace --stats on tools/quick_bench_set.py
TAG=${1:-set}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o s -- python3 $R/tools/quick_bench_set.py > /tmp/prof_$TAG.log 2>&1
tail -1 /tmp/prof_$TAG.log
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:26]:
    n = r["Name"]
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", n)
    short = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    print("%-95s calls %4s avg %8.1f us total %9.1f us" % (short[:95], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
