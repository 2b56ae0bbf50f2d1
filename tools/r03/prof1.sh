set -o pipefail
mkdir -p gpurun_out/r03
bash tools/prof.sh r03_a > gpurun_out/r03/prof_a.log 2>&1; tail -3 gpurun_out/r03/prof_a.log
PROF_KERNEL=k_fit_search,k_dither_store,k_fit_tpb bash tools/prof.sh r03_a_split --split > gpurun_out/r03/prof_a_split.log 2>&1; tail -3 gpurun_out/r03/prof_a_split.log
bash tools/prof.sh r03_a_acc --accurate > gpurun_out/r03/prof_a_acc.log 2>&1; tail -3 gpurun_out/r03/prof_a_acc.log
