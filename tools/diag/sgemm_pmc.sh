#!/bin/bash
# one rocprofv3 --pmc pass over tools/diag/sgemm_shapes_probe.py (few repetitions): how busy the matrix pipe is inside the TD3 update's
# product kernels, where their waves wait.   gpurun -- 'bash tools/diag/sgemm_pmc.sh 1792x256x256 1792x1024x256'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sgemm_pmc
mkdir -p $O /tmp/sgpmc
cd /tmp && export TMPDIR=/tmp
REPS=20 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d /tmp/sgpmc/p1 -o p -- python3 $R/tools/diag/sgemm_shapes_probe.py "$@" > $O/probe.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections, re
import numpy as np
f = glob.glob("/tmp/sgpmc/p1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(k_sgemm[a-z_]*(<[^>]*>)?)", n)
    if not m: continue
    key = m.group(1) + " grid=" + r.get("Grid_Size", "?")
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    m = {n: float(np.mean(v)) for n, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    print("%-60s n=%3d cycles %7d  mfma_busy %.3f  wait_any %.3f  wait_inst %.3f  valu_active/wave_cycles %.3f  lds_active/wave_cycles %.3f  bank_conflict/lds %.3f" % (
        k, len(c["GRBM_GUI_ACTIVE"]), cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_LDS"] * 4 / m["SQ_WAVE_CYCLES"],
        m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_ACTIVE_INST_LDS"] * 4, 1)))
PY
