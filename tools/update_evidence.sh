#!/bin/bash
# TD3 update evidence: per-kernel stats of one morphology's update under rocprofv3 (eager: every launch is a
# kernel-trace record), the hipGraph-replayed time of the same update, and the config-5 trainer bench.
#   gpurun --timeout 1100 -- 'bash tools/update_evidence.sh r3'
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/update
mkdir -p $O /tmp/upd
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/upd/stats -o u -- python3 $R/tools/update_profile.py 3d_walker_7_full 20 > $O/update_eager.log 2> /tmp/upd/stats.err
cd $R
python3 - <<PY
import csv, glob, json, os
f = sorted(glob.glob("/tmp/upd/stats/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[0]))) if f else []
iters = 20 + 4
out = open("$O/${TAG}_update_kernel_stats.csv", "w")
out.write("# one TD3 update of 3d_walker_7_full, batch 256 (agent_batch_size), eager (tools/update_profile.py under rocprofv3 --kernel-trace --stats); %d updates in the run\n" % iters)
out.write("Name,Calls,CallsPerUpdate,TotalDurationNs,AverageNs,Percentage\n")
tot_calls = 0
for r in rows:
    tot_calls += int(r["Calls"])
    out.write('"%s",%s,%.1f,%s,%s,%s\n' % (r["Name"][:160], r["Calls"], int(r["Calls"]) / iters, r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
out.close()
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
json.dump({"launches_per_update": round(tot_calls / iters, 1), "gpu_ms_per_update_sum_of_kernels": round(tot_ns / iters / 1e6, 3)},
          open("$O/${TAG}_update_launches.json", "w"))
print(open("$O/${TAG}_update_launches.json").read())
PY
SGRL_GRAPH_UPDATES=1 timeout 300 python3 tools/update_profile.py 3d_walker_7_full 50 > $O/update_graphed.log 2>&1
if [ -z "$SKIP_TRAIN" ]; then SGRL_TUNE_GEMMS=0 timeout 900 python3 tools/train_bench.py > /tmp/upd/train.log 2>&1; cp gpurun_out/train_bench.json $O/${TAG}_config5_train_bench.json 2>/dev/null; fi
timeout 300 python3 tools/diag/update_launch_sources.py > $O/${TAG}_update_launch_sources.txt 2>&1
for f in $O/update_eager.log $O/update_graphed.log /tmp/upd/train.log /tmp/upd/stats.err; do tail -n 2 $f; done
