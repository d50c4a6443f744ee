#!/bin/bash
# round 6: the closing check of a build on the MI355X box -- GPU suite, smoke(), default bench line
set -o pipefail
O=gpurun_out/r6verify; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1; RC=$?
tail -n 3 $O/gpu_pytest.log
if [ $RC -ne 0 ]; then exit $RC; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -n 20 $O/smoke.log; exit 1; }
tail -n 1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.log 2> $O/bench.err || { tail -n 20 $O/bench.err; exit 1; }
tail -n 1 $O/bench.log > $O/bench.json; cut -c1-600 $O/bench.json
