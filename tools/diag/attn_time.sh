cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/qa; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qa -o q -- python3 $GRAFT_REPO_ROOT/tools/quick_bench_set.py 2>/dev/null | tail -1; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/qa/**/*kernel_stats.csv",recursive=True)[0])):
    if "k_attention" in r["Name"]: print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
