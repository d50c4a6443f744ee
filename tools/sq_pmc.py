#!/usr/bin/env python3
"""profiles/sq_pmc.json from two rocprofv3 --pmc passes (SQ counters) of the bench command: the VALU view of the step kernel
(usage: sq_pmc.py <build tag> <pass1 dir> <pass2 dir> <envs per gpu>)."""
import csv, collections, glob, json, sys
import numpy as np
tag, d1, d2, n_env = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in (d1, d2):
    for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
        if "k_env_step" in r["Kernel_Name"] and "reset" not in r["Kernel_Name"] and "refresh" not in r["Kernel_Name"]:
            acc["k_env_step"][r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {c: float(np.mean(v[len(v) // 2:])) for c, v in acc["k_env_step"].items()}       # second half: stationary episode mix
n = len(next(iter(acc["k_env_step"].values())))
cyc = m["GRBM_GUI_ACTIVE"] / 8.0
out = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline (two passes: "
                  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES | "
                  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)",
       "build": tag, "envs_per_gpu": n_env,
       "k_env_step": {"kernel": "k_env_step_spec (walker family, fixed dimensions)", "launches_averaged": n - n // 2, "waves_per_launch": int(m["SQ_WAVES"]),
                      "valu_wave_instr_per_launch": int(m["SQ_INSTS_VALU"]), "salu_instr_per_launch": int(m["SQ_INSTS_SALU"]),
                      "lds_instr_per_launch": int(m["SQ_INSTS_LDS"]), "valu_wave_instr_per_env_step": int(m["SQ_INSTS_VALU"] / n_env),
                      "kernel_cycles_under_pmc": int(cyc),
                      "valu_issue_fraction": round(4 * m["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3),
                      "active_lane_fraction": round(m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]), 3),
                      "wave_wait_fraction": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                      "wave_issue_stall_fraction": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                      "lds_pipe_busy_fraction": round(m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 3),
                      "lds_bank_conflict_fraction_of_lds_cycles": round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3),
                      "definitions": "valu_issue_fraction = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x kernel cycles, kernel cycles = "
                                     "GRBM_GUI_ACTIVE / 8 XCDs); active_lane_fraction = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                                     "wave_wait_fraction = SQ_WAIT_ANY / SQ_WAVE_CYCLES; lds_pipe_busy_fraction = SQ_LDS_IDX_ACTIVE / (256 CUs x "
                                     "kernel cycles); means over the second half of the run (stationary episode mix)",
                      "raw_means": {k: round(v, 1) for k, v in m.items()}}}
json.dump(out, open("profiles/sq_pmc.json", "w"), indent=1)
print(json.dumps(out["k_env_step"])[:400])
