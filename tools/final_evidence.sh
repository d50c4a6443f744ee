#!/bin/bash
# Evidence set of one build, on the MI355X box:  gpurun --timeout 2400 -- 'bash tools/final_evidence.sh r2_v3'
# Runs the default bench, the rocprofv3 stats pass and the two --pmc passes of the same command, the GPU test suite, the config
# sweep and the config-5 trainer bench; summarises into profiles/ ON THE BOX and copies the summaries to gpurun_out/final/ (the raw
# traces stay on the box: gpurun_out/ is limited to 64 MiB).
TAG=${1:-r2_v8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
RAW=/tmp/sgrl_raw
mkdir -p $O $RAW
cd $R
timeout 900 python3 bench.py > $RAW/bench_default.log 2> $RAW/bench_default.err
tail -1 $RAW/bench_default.log > $O/${TAG}_bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $RAW/bench_prof.log 2> $RAW/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -o f -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $RAW/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -o w -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $RAW/write.log 2>&1
cd $R
grep '^{"metric"' $RAW/bench_prof.log | tail -1 > $RAW/bench_prof.json
python3 tools/summarize_profiles.py $TAG $RAW/stats $RAW/fetch $RAW/write $RAW/bench_prof.json > $O/summarize.log 2>&1
python3 tools/set_traffic.py $TAG >> $O/summarize.log 2>&1
# the default bench again, now quoting the fresh pmc_traffic.json (copied AFTER the profiles/ summaries: same file name)
timeout 900 python3 bench.py > $RAW/bench_default2.log 2>> $RAW/bench_default.err
cp profiles/${TAG}_bench.json $O/${TAG}_bench_under_rocprof.json
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_*_summary.csv profiles/${TAG}_set_traffic.json profiles/pmc_traffic.json $O/ 2>/dev/null
tail -1 $RAW/bench_default2.log > $O/${TAG}_bench.json
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r2_gpu_pytest.log
timeout 600 python3 tools/config_sweep.py > $RAW/sweep.log 2>&1; cp gpurun_out/config_sweep.json $O/r2_config_sweep.json 2>/dev/null
SGRL_TUNE_GEMMS=0 timeout 900 python3 tools/train_bench.py > $RAW/train.log 2>&1; cp gpurun_out/train_bench.json $O/r2_config5_train_bench.json 2>/dev/null
tail -3 $RAW/bench_default.err $RAW/stats.err > $O/stderr_tails.log 2>&1
ls -la $O
