#!/bin/bash
# Where k_env_step's waves wait: in-flight levels of LDS / scalar-memory / vector-memory instructions (rocprofv3 --pmc pass of the bench command)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp SGRL_BENCH_NO_CHILD=1
rm -rf /tmp/lat
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/lat -o c -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --regions 1 > /tmp/lat.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/lat/**/*counter_collection.csv", recursive=True)
for key in ("k_env_step", "k_chain", "k_gemm3", "k_attention"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not n: continue
    d = {k: acc[k] / n[k] for k in acc}
    wc = d["SQ_WAVE_CYCLES"]
    print(key, "dispatches", max(n.values()), {k: round(v) for k, v in d.items()})
    for nm in ("LDS", "SMEM", "VMEM"):
        ins, lvl = d["SQ_INSTS_" + nm], d["SQ_INST_LEVEL_" + nm]
        print("   %-4s %12.0f instructions, mean latency %6.0f cycles, in-flight share of wave cycles %.3f" % (nm, ins, lvl / max(ins, 1), lvl / wc))
    print("   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = %.3f" % (d["SQ_WAIT_INST_LDS"] / wc))
PY
