#!/bin/bash
# Instruction-cache counters of k_env_step (walker mix) under two workgroup orders: rocprofv3 --pmc pass of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp SGRL_BENCH_NO_CHILD=1
for v in 0 1; do
  export SGRL_ORDER=$v
  rm -rf /tmp/ic$v
  timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ic$v -o c -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --regions 1 > /tmp/ic$v.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/ic$v/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "k_env_step" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
d = {k: acc[k] / n[k] for k in acc}
print("SGRL_ORDER=$v  dispatches", max(n.values()), {k: round(v) for k, v in d.items()})
if d.get("SQC_ICACHE_REQ"):
    print("   icache hit rate %.3f, misses per request %.3f (+ duplicate %.3f); fetches in flight per busy cycle %.2f; mean fetch latency %.0f cycles; wave cycles per fetch %.1f" % (
        d["SQC_ICACHE_HITS"] / d["SQC_ICACHE_REQ"], d["SQC_ICACHE_MISSES"] / d["SQC_ICACHE_REQ"], d["SQC_ICACHE_MISSES_DUPLICATE"] / d["SQC_ICACHE_REQ"],
        d["SQ_IFETCH_LEVEL"] / max(d["SQ_BUSY_CYCLES"], 1), d["SQ_IFETCH_LEVEL"] / max(d["SQ_IFETCH"], 1), d["SQ_WAVE_CYCLES"] / max(d["SQ_IFETCH"], 1)))
PY
done
