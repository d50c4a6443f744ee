#!/bin/bash
# round 6, closing call: the check of the last build (GPU suite, smoke, default bench), then the graphed update's kernel table and the
# config-5 trainer bench on it
set -o pipefail
bash tools/r6_verify.sh || exit $?
bash tools/update_graph_profile.sh r6b > gpurun_out/r6verify/update_graph_profile.log 2>&1 || { tail -n 5 gpurun_out/r6verify/update_graph_profile.log; exit 1; }
tail -n 2 gpurun_out/r6verify/update_graph_profile.log
SGRL_TUNE_GEMMS=0 timeout -k 10 420 python3 tools/train_bench.py > gpurun_out/r6verify/train_bench.log 2>&1 || { tail -n 5 gpurun_out/r6verify/train_bench.log; exit 1; }
cp gpurun_out/train_bench.json gpurun_out/r6verify/r6b_config5_train_bench.json
tail -n 2 gpurun_out/r6verify/train_bench.log | cut -c1-400
