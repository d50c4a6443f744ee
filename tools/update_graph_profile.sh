#!/bin/bash
# Per-kernel decomposition of the hipGraph-replayed TD3 update (what tools/update_evidence.sh's eager pass cannot show: the graphed path
# defers the weight gradients and skips the critic's unused ones).   gpurun -- 'bash tools/update_graph_profile.sh r4'
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/update
mkdir -p $O /tmp/updg
cd /tmp && export TMPDIR=/tmp && export SGRL_GRAPH_UPDATES=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/updg/stats -o g -- python3 $R/tools/update_profile.py 3d_walker_7_full 50 > $O/update_graphed_prof.log 2> /tmp/updg/err
cd $R
python3 - <<PY
import csv, glob
f = sorted(glob.glob("/tmp/updg/stats/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[0]))) if f else []
iters = 50 + 4 + 3          # timed + untimed replays + the eager warm-up updates (close enough for shares)
out = open("$O/${TAG}_update_graphed_kernel_stats.csv", "w")
out.write("# hipGraph-replayed TD3 update of 3d_walker_7_full, batch 256 (agent_batch_size) (tools/update_graph_profile.sh): %d updates in the run (3 eager warm-ups)\n" % iters)
out.write("Name,Calls,CallsPerUpdate,TotalDurationNs,AverageNs,Percentage\n")
tc = tn = 0
for r in rows:
    tc += int(r["Calls"]); tn += float(r["TotalDurationNs"])
    out.write('"%s",%s,%.1f,%s,%s,%s\n' % (r["Name"][:140], r["Calls"], int(r["Calls"]) / iters, r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
out.close()
print("launches per update %.1f, kernel ms per update %.3f" % (tc / iters, tn / iters / 1e6))
PY
tail -n 1 $O/update_graphed_prof.log
