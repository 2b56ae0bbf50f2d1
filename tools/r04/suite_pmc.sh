# round 4 final: rocprofv3 kernel trace + PMC passes per workload (tools/prof.sh), results merged into gpurun_out/profiles/pmc_by_workload.json.
# usage: bash tools/r04/suite_pmc.sh <part>      (three parts: a call stays below gpurun's limit)
set -o pipefail
case "$1" in
 1) bash tools/prof.sh r04_final --steps 25 | tail -1
    PROF_KERNEL=k_fit_search,k_dither_store,k_fit_tpb bash tools/prof.sh r04_final_split --split | tail -1
    bash tools/prof.sh r04_final_accurate --accurate | tail -1 ;;
 2) bash tools/prof.sh r04_final_rg4096 --size 4096 --workload random_gradient | tail -1
    bash tools/prof.sh r04_final_c4 --config 4 --steps 2 --warmup 1 | tail -1
    bash tools/prof.sh r04_final_c4_onepair --config 4 --steps 2 --warmup 1 --sub-images -1 | tail -1
    PROF_KERNEL=k_stream_decode bash tools/prof.sh r04_final_stream --stream | tail -1 ;;
 3) bash tools/prof.sh r04_final_fast --float-mode fast | tail -1
    PROF_KERNEL=k_encode_persistent bash tools/prof.sh r04_final_legacy --legacy-float-stage | tail -1
    bash tools/prof.sh r04_final_rgb --rgb | tail -1
    bash tools/prof.sh r04_final_c5 --config 5 --steps 3 --warmup 1 | tail -1 ;;
esac
