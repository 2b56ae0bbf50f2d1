#!/bin/bash
# the round's last library build: whole GPU suite, smoke, default bench line, kernel trace of the merged-block encoder
set -o pipefail
O=gpurun_out/r30; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 && tail -1 $O/smoke.txt &&
python bench.py > $O/bench.json 2> $O/bench.err && cut -c1-400 $O/bench.json &&
bash tools/prof_blocked.sh r04_blocked_final > $O/prof_blocked.txt 2>&1; tail -12 $O/prof_blocked.txt | cut -c1-200
bash tools/prof_blocked.sh r04_blocked_final_rg --workload random_gradient > $O/prof_blocked_rg.txt 2>&1; tail -10 $O/prof_blocked_rg.txt | cut -c1-200
