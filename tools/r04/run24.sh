#!/bin/bash
# round 4, last commit: the whole GPU suite, the smoke entry and the default bench line on one box
set -o pipefail
O=gpurun_out/r24; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 && tail -1 $O/smoke.txt &&
python bench.py > $O/bench.json 2> $O/bench.err && cat $O/bench.json | cut -c1-600
