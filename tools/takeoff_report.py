#!/usr/bin/env python3
"""Markdown table of every take-off cell under profiles/r6_takeoff (tools/takeoff_table.py output): per (arm, seed) the random policy's
return, the mean train return of rounds 1-5 / 6-10 / the last five, the best round, and the first round from which the return stays
above 1.5 x random for three rounds in a row ("take-off round", - if never).   usage: takeoff_report.py [dir] [prefix]"""
import glob, json, os, sys
import numpy as np
root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_takeoff")
prefix = sys.argv[2] if len(sys.argv) > 2 else "r6_takeoff_"
rows = []
for f in sorted(glob.glob(os.path.join(root, "**", prefix + "*.jsonl"), recursive=True)):
    base = os.path.basename(f)[len(prefix):-len(".jsonl")]
    arm, seed = base.rsplit("_s", 1)
    cur = [json.loads(l) for l in open(f) if l.strip()]
    sp = f[:-len(".jsonl")] + "_summary.json"
    summ = json.load(open(sp)) if os.path.exists(sp) else {}
    rnd = summ.get("random_policy_return")
    ret = np.array([c["return"] for c in cur], dtype=float)
    if ret.size == 0 or rnd is None:
        continue
    up = ret > 1.5 * rnd
    take = next((i + 1 for i in range(len(up) - 2) if up[i] and up[i + 1] and up[i + 2]), None)
    m = lambda a: ("%.0f" % np.nanmean(a)) if len(a) else "-"
    where = os.path.relpath(os.path.dirname(f), root)
    rows.append((where if where != "." else "main", arm, int(seed), len(ret), rnd, m(ret[:5]), m(ret[5:10]), m(ret[-5:]), np.nanmax(ret), take, summ.get("updates"), summ.get("wall_s")))
print("| set | arm | seed | rounds | random | rounds 1-5 | rounds 6-10 | last 5 | best round | take-off round | updates | wall s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in sorted(rows):
    print("| %s | %s | %d | %d | %.1f | %s | %s | %s | %.0f | %s | %s | %s |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9] if r[9] else "-", r[10], ("%.0f" % r[11]) if r[11] else "-"))
