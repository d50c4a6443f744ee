R=$(pwd)
for rep in 1 2; do
cd $R/build/ab_old && export GRAFT_REPO_ROOT=$(pwd) && python tools/quick_bench_set.py | tail -1
cd $R && export GRAFT_REPO_ROOT=$(pwd) && python tools/quick_bench_set.py | tail -1
done
cd $R/build/ab_old && export GRAFT_REPO_ROOT=$(pwd) && SGRL_SET_ONE_STREAM=1 bash tools/prof_set.sh old_serial 2>&1 | grep -E "k_gemm3|k_chain|k_encode|k_pack" | awk '{print $1, $2, $3, $4, $(NF-4), $(NF-3)}' | cut -c1-120
cd $R && export GRAFT_REPO_ROOT=$(pwd) && SGRL_SET_ONE_STREAM=1 bash tools/prof_set.sh new_serial 2>&1 | grep -E "k_gemm3|k_chain|k_encode|k_pack" | awk '{print $1, $2, $3, $4, $(NF-4), $(NF-3)}' | cut -c1-120
