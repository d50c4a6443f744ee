#!/usr/bin/env python3
"""HBM traffic of ONE SET forward from the two `--pmc` summaries tools/summarize_profiles.py wrote
(profiles/<tag>_pmc_{fetch,write}_size_summary.csv): per-kernel mean x launches, summed over the forward's kernels and divided by the
number of forwards in the run (= launches of k_embed).  Usage: set_traffic.py <tag> [nodes]  ->  profiles/<tag>_set_traffic.json"""
import csv, json, sys
tag = sys.argv[1]
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 35840
SET = ("k_gemm3", "k_gemm2", "k_chain", "k_attention", "k_equiv", "k_embed", "k_pack", "k_encode_rows", "k_add_ln", "k_head_out", "k_head_out2",
       "k_q_head", "k_relbias", "k_stack_proj", "k_sgemm", "k_gram576", "k_zmat_perm")
tab = {}
for name in ("fetch", "write"):
    for r in csv.DictReader(open("profiles/%s_pmc_%s_size_summary.csv" % (tag, name))):
        tab.setdefault(r["kernel"], {})[name] = (int(r["launches"]), float(r["mean_KB"]))
fwd = tab["k_embed"]["fetch"][0]
out = {"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline "
                  "(two passes); per-kernel means x launches, summed over the kernels of one SET forward",
       "forwards": fwd, "nodes": nodes, "per_forward_GB_raw": {}, "per_kernel_MB_per_forward_raw": {}}
tot = {"fetch": 0.0, "write": 0.0}
for k in SET:
    if k not in tab:
        continue
    e = {"launches_per_forward": round(tab[k]["fetch"][0] / fwd, 2)}
    for name, label in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        n, kb = tab[k][name]
        mb = n * kb * 1024.0 / 1e6 / fwd          # counters are in KiB; MB / GB below are decimal
        e[label] = round(mb, 1)
        tot[name] += mb
    out["per_kernel_MB_per_forward_raw"][k] = e
out["per_forward_GB_raw"] = {"FETCH_SIZE": round(tot["fetch"] / 1e3, 3), "WRITE_SIZE": round(tot["write"] / 1e3, 3)}
out["bytes_per_node_raw"] = int((tot["fetch"] + tot["write"]) * 1e6 / nodes)
json.dump(out, open("profiles/%s_set_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out["per_forward_GB_raw"]), out["bytes_per_node_raw"], "B/node")
