#!/usr/bin/env bash
# A build variant of ONE source file, linked with the tree's other (plain) objects:  tools/r06/variant.sh NAME limg_hip_stream.hip -DX=1 ...   -> ab/NAME/liblimg_hip.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
mkdir -p ab/$NAME
EXTRA=""
[ "$SRC" = limg_hip_kernels.hip ] && EXTRA="-mllvm -amdgpu-atomic-optimizer-strategy=None"
[ "$SRC" = limg_hip_stream.hip ] && EXTRA="-Wno-pass-failed"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA "$@" -c limg_amd/csrc/$SRC -o ab/$NAME/${SRC%.*}.o
OBJS=""
for f in limg_hip_kernels limg_hip_fit_tpb limg_hip_stream limg_hip_blocked limg_hip_synth limg_hip_noise_gpu limg_hip_api limg_hip_noise limg_hip_blocked_host; do
  if [ -f ab/$NAME/$f.o ]; then OBJS="$OBJS ab/$NAME/$f.o"; else OBJS="$OBJS limg_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/liblimg_hip.so $OBJS
echo ab/$NAME/liblimg_hip.so
