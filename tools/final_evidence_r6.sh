#!/bin/bash
# Evidence set of one build, on the MI355X box:  gpurun --timeout 1200 -- 'bash tools/final_evidence_r6.sh r6_v1'
# default bench, rocprofv3 stats pass, FETCH/WRITE and two SQ --pmc passes of the same command, SET traffic, GPU test suite, config sweep.
TAG=${1:-r6_v1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
RAW=/tmp/sgrl_raw
mkdir -p $O $RAW
cd $R
timeout 600 python3 bench.py > $RAW/bench_default.log 2> $RAW/bench_default.err
tail -1 $RAW/bench_default.log > $O/${TAG}_bench_default.json
echo "bench done" 
cd /tmp && export TMPDIR=/tmp && export SGRL_BENCH_NO_CHILD=1
B="python3 $R/bench.py --warmup 3 --no-cpu-baseline --regions 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o s -- $B --steps 20 > $RAW/bench_prof.log 2> $RAW/stats.err
echo "stats done"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -o f -- $B --steps 5 > $RAW/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -o w -- $B --steps 5 > $RAW/write.log 2>&1
echo "traffic done"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $RAW/sq1 -o q -- $B --steps 5 > $RAW/sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq2 -o q -- $B --steps 5 > $RAW/sq2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/sq3 -o q -- $B --steps 5 > $RAW/sq3.log 2>&1
echo "sq done"
cd $R
grep '^{"metric"' $RAW/bench_prof.log | tail -1 > $RAW/bench_prof.json
python3 tools/summarize_profiles.py $TAG $RAW/stats $RAW/fetch $RAW/write $RAW/bench_prof.json > $O/summarize.log 2>&1
python3 tools/sq_pmc.py $TAG $RAW/sq1 $RAW/sq2 8192 >> $O/summarize.log 2>&1
python3 tools/set_traffic.py $TAG >> $O/summarize.log 2>&1
python3 tools/mfma_pmc.py $TAG $RAW/sq3 >> $O/summarize.log 2>&1
unset SGRL_BENCH_NO_CHILD
timeout 600 python3 bench.py > $RAW/bench_default2.log 2>> $RAW/bench_default.err      # now quoting the fresh pmc_traffic.json / sq_pmc.json
cp profiles/${TAG}_bench.json $O/${TAG}_bench_under_rocprof.json
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_*_summary.csv profiles/${TAG}_set_traffic.json profiles/${TAG}_set_mfma_pmc.json profiles/pmc_traffic.json profiles/sq_pmc.json $O/ 2>/dev/null
tail -1 $RAW/bench_default2.log > $O/${TAG}_bench.json
echo "second bench done"
timeout 900 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r6_gpu_pytest.log
echo "pytest done"
timeout 400 python3 tools/config_sweep.py > $RAW/sweep.log 2>&1; cp gpurun_out/config_sweep.json $O/r6_config_sweep.json 2>/dev/null
for f in $RAW/bench_default.err $RAW/stats.err $O/summarize.log; do tail -n 3 $f; done > $O/stderr_tails.log 2>&1
ls -la $O
