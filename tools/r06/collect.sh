#!/usr/bin/env bash
# Build container, after tools/r06/suite_prof.sh and suite_bench.sh ran on the GPU box: copy what is to be judged from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
for t in r06_final r06_final_rg4096 r06_final_c4 r06_final_stream r06_final_accurate r06_final_blocked r06_final_blocked_rg; do
  [ -f gpurun_out/prof_$t/summary.txt ] && cp gpurun_out/prof_$t/summary.txt profiles/${t}_summary.txt
  f=$(find gpurun_out/prof_$t/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/${t}_kernel_stats.csv
done
[ -s gpurun_out/profiles/pmc_by_workload.json ] && cp gpurun_out/profiles/pmc_by_workload.json profiles/pmc_by_workload.json
for f in gpurun_out/r06_final/bench_*.json; do
  n=$(basename $f .json); n=${n#bench_}
  # (only the JSON line: the gloo rehearsals' stdout also carries gloo's own "[Gloo] Rank ... is connected" message)
  if [ "$n" = default ]; then grep '^{' $f > profiles/r06_bench_final.json; else grep '^{' $f > profiles/r06_bench_final_$n.json; fi
done
cp gpurun_out/r06_final/rc.txt profiles/r06_bench_final_rc.txt
python tools/profiles_index.py
