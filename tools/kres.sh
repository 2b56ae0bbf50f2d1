#!/usr/bin/env bash
# Register / LDS / scratch use of the kernels of one source file (default: limg_hip_kernels.hip), from the compiler's own remarks.
# usage: tools/kres.sh [file.hip] [extra hipcc flags...]
F=${1:-limg_hip_kernels.hip}; shift || true
cd "$(dirname "$0")/../limg_amd/csrc"
EXTRA=""
[ "$F" = limg_hip_kernels.hip ] && EXTRA="-mllvm -amdgpu-atomic-optimizer-strategy=None"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math $EXTRA "$@" --cuda-device-only -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/{name=$5} / VGPRs:/{v=$4} /TotalSGPRs:/{s=$4} /ScratchSize/{sc=$5} /SGPRs Spill:/{ss=$5} /VGPRs Spill:/{vs=$5} /Occupancy/{o=$5} /LDS Size/{print name, "vgpr", v, "sgpr", s, "sgpr_spill", ss, "vgpr_spill", vs, "scratch", sc, "occ", o, "lds", $6}' | c++filt | sed 's/limg_hip::(anonymous namespace):://; s/(limg_hip::EncodeParams)//'
