#!/usr/bin/env python3
"""profiles/<tag>_set_mfma_pmc.json from one rocprofv3 --pmc pass of the bench command: how busy the matrix pipe is inside the SET
forward's product kernels (the fused chains of chain_f16.h and the single products of gemm_f32.h), per kernel.
usage: mfma_pmc.py <tag> <pass dir>"""
import csv, collections, glob, json, re, sys
import numpy as np
tag, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    n = r["Kernel_Name"]
    m = re.search(r"(k_chain<[^>]*>|k_gemm3<\d+)", n)
    if not m:
        continue
    k = m.group(1)
    if k.startswith("k_gemm3"):
        k = {"k_gemm3<0": "k_gemm3 plain (U)", "k_gemm3<1": "k_gemm3 ReLU (linear3, linear1_m)", "k_gemm3<2": "k_gemm3 row division (qkv)",
             "k_gemm3<10": "k_gemm3 equivariant epilogue (linear4)"}.get(k, k)
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 5 --warmup 3 "
                  "--no-cpu-baseline --regions 1", "build": tag, "kernels": {}}
for k, c in sorted(acc.items()):
    m = {n: float(np.mean(v)) for n, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    out["kernels"][k] = {"dispatches": len(c["GRBM_GUI_ACTIVE"]), "kernel_cycles": int(cyc),
                         "matrix_pipe_busy_fraction": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3),
                         "waves_parked_at_waitcnt_or_barrier": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                         "waves_issue_stalled": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                         "lds_bank_conflict_cycles_per_lds_instruction_cycle": round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_ACTIVE_INST_LDS"] * 4, 1), 3)}
out["definitions"] = ("matrix_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs "
                      "(means over the dispatches of the run; kernels of the two streams overlap, so a kernel's cycles include its neighbours')")
json.dump(out, open("profiles/%s_set_mfma_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out["kernels"])[:600])
