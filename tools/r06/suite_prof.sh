#!/usr/bin/env bash
# Round 6 evidence, part 1 (GPU box): rocprofv3 kernel trace + PMC passes of the bench modes whose lines quote counters, on the build in the tree.
# Writes gpurun_out/prof_r06_*/ and gpurun_out/profiles/pmc_by_workload.json; copy the summaries and the json into profiles/ afterwards (tools/r06/collect.sh).
set -o pipefail
rm -f gpurun_out/profiles/pmc_by_workload.json; mkdir -p gpurun_out/profiles; echo '{}' > gpurun_out/profiles/pmc_by_workload.json
bash tools/prof.sh r06_final --steps 25 | tail -1
bash tools/prof.sh r06_final_rg4096 --size 4096 --workload random_gradient | tail -1
bash tools/prof.sh r06_final_c4 --config 4 --steps 2 --warmup 1 | tail -1
PROF_KERNEL=k_stream_decode bash tools/prof.sh r06_final_stream --stream | tail -1
grep -n "k_stream" gpurun_out/prof_r06_final_stream/summary.txt | head -8
bash tools/prof.sh r06_final_accurate --accurate --steps 10 | tail -1
bash tools/prof_blocked.sh r06_final_blocked | tail -9
bash tools/prof_blocked.sh r06_final_blocked_rg --workload random_gradient | tail -9
