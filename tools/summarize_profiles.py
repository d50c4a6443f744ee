#!/usr/bin/env python3
"""Turn raw gpurun_out rocprofv3 outputs into the small tracked summaries under profiles/ (usage: tag stats_dir fetch_dir write_dir bench_log)."""
import csv, collections, glob, json, re, shutil, sys
import numpy as np
tag, stats_dir, fetch_dir, write_dir, bench_log = sys.argv[1:6]
shutil.copy(glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0], "profiles/%s_kernel_stats.csv" % tag)
means = {}
for name, d in (("fetch", fetch_dir), ("write", write_dir)):
    rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        m = re.search(r"(k_[a-z_0-9]+)[<(]", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:60]
        if k.startswith("k_env_step"):      # generic kernel and the fixed-dimension family kernels (k_env_step_spec): one row
            k = "k_env_step"
        acc[k].append(float(r["Counter_Value"]))
    with open("profiles/%s_pmc_%s_size_summary.csv" % (tag, name), "w") as f:
        f.write("kernel,counter,launches,mean_KB,min_KB,max_KB\n")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%s,%d,%.1f,%.1f,%.1f\n' % (k, rows[0]["Counter_Name"], len(v), np.mean(v), np.min(v), np.max(v)))
    means[name] = (float(np.mean(acc["k_env_step"])), len(acc["k_env_step"]))
line = open(bench_log).read().strip().split("\n")[-1]
open("profiles/%s_bench.json" % tag, "w").write(line + "\n")
groups = int(json.loads(line)["roofline"].get("dispatches_per_launch", 1))
# one launch (sgrl_step) = `groups` concurrent k_env_step dispatches: wall span per launch from the kernel trace
tr = [r for r in csv.DictReader(open(glob.glob(stats_dir + "/**/*kernel_trace.csv", recursive=True)[0])) if "k_env_step" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
spans, durs = [], []
for i in range(0, len(tr) - groups + 1, groups):
    chunk = tr[i:i + groups]
    spans.append((max(int(r["End_Timestamp"]) for r in chunk) - min(int(r["Start_Timestamp"]) for r in chunk)) / 1e6)
    durs.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in chunk) / 1e6)
half = len(spans) // 2   # the second half of the run is the stationary episode mix (after the pre-roll)
span_ms = float(np.mean(spans[half:])) if spans else None
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline (two separate passes)",
       "envs_per_gpu": json.loads(line)["config"]["envs_per_gpu"], "kernel": "k_env_step (this workload: k_env_step_spec, the walker family's fixed-dimension kernel)",
       "FETCH_SIZE_KB_per_launch_mean": round(means["fetch"][0], 1), "WRITE_SIZE_KB_per_launch_mean": round(means["write"][0], 1),
       "dispatches_per_launch": groups, "dispatches": means["fetch"][1],
       "k_env_step_bytes_per_launch": int((means["fetch"][0] + means["write"][0]) * 1024 * groups),
       "k_env_step_wall_ms_per_launch_from_trace": None if span_ms is None else round(span_ms, 4),
       "k_env_step_sum_of_dispatch_ms_per_launch": None if not durs else round(float(np.mean(durs[half:])), 4),
       "note": "raw counters x 1024; MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of the bytes of WIDE (16 B/lane) coalesced streaming "
               "reads; this kernel reads 8 B/lane records, for which the guide gives no calibration, so no correction is applied.",
       "build": "round %s (%s)" % (tag.split("_")[0].lstrip("r"), tag)}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out)[:300])
