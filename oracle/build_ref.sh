#!/usr/bin/env bash
# TEST INFRASTRUCTURE -- builds the *real* reference (rainerzufalldererste/limg) hot path into
# oracle/_ref/liblimg_ref.so so the CPU restatement (oracle/limg_oracle.c) can be pinned against it and
# so bench.py can time it as cpu_baseline kind "reference".
#
# The reference sources are compiled from where they lie (/root/reference/src).  They do not build
# unmodified with g++ 11 / clang 22: five mechanical, arithmetic-neutral
# portability patches are applied to a throw-away copy in a mktemp dir which is deleted afterwards.
# No reference source is ever written into this repository; only the .so lands in oracle/_ref/
# (git-ignored, but shipped to the GPU box by gpurun).
#
#   1. limg_bit_crush_simd.h : __attribute__((target)) moved behind the `template<...>` line (2x)
#   2. limg.cpp              : -include limits.h (CHAR_BIT)
#   3./4. limg.cpp           : two `T x = ...;` initialisers crossed by `goto epilogue` split into decl + assignment
#   5. limg_simd.cpp         : local `_xgetbv` renamed (clashes with <xsaveintrin.h>)
set -euo pipefail
REF=${LIMG_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
  echo "build_ref.sh: $REF/src not present (GPU box?) -- keeping prebuilt files in $OUT" >&2
  exit 0
fi
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
cp "$REF"/src/*.h "$REF"/src/limg.cpp "$REF"/src/limg_simd.cpp "$REF"/src/limg_threading.cpp "$TMP"/
python3 - "$TMP" <<'PY'
import sys, re, os
d = sys.argv[1]
def sub(path, fn):
    p = os.path.join(d, path); s = open(p).read(); t = fn(s)
    assert t != s, "patch did not apply: " + path
    open(p, "w").write(t)
# 1
attr = '#ifndef _MSC_VER\n__attribute__((target("sse4.1")))\n#endif\n'
tmpl = 'template <bool extractPixel, bool checkBlockError>\n'
sub("limg_bit_crush_simd.h", lambda s: s.replace(attr + tmpl, tmpl + attr))
# 3/4
def p34(s):
    s = s.replace("  size_t accum_bits[3 + 3 * 9] = { 0 };\n\n  if (ctx.hasAlpha)\n    LIMG_ERROR_CHECK(",
                  "  memset(accum_bits, 0, sizeof(accum_bits));\n\n  if (ctx.hasAlpha)\n    LIMG_ERROR_CHECK(")
    # hoist the declarations to the top of limg_blocked_encode3d_test
    head = "limg_result limg_blocked_encode3d_test("
    i = s.index(head); j = s.index("{", i)
    s = s[:j+1] + "\n  size_t accum_bits[3 + 3 * 9]; size_t totalPixels;\n" + s[j+1:]
    k = s.index("const size_t totalPixels = ctx.sizeX * ctx.sizeY;", j)
    s = s[:k] + "totalPixels = ctx.sizeX * ctx.sizeY;" + s[k+len("const size_t totalPixels = ctx.sizeX * ctx.sizeY;"):]
    return s
sub("limg.cpp", p34)
# 5
sub("limg_simd.cpp", lambda s: s.replace("_xgetbv(", "limg_local_xgetbv("))
PY
CXX=${CXX:-g++}
FLAGS="-std=c++17 -O3 -msse4.1 -maes -fno-exceptions -fno-rtti -DNDEBUG -w -fPIC -include limits.h -I$TMP"
# strict IEEE build = golden source; the project's own setting is fast-math (project.lua:38) -> tolerance witness
$CXX $FLAGS -shared -o "$OUT/liblimg_ref.so" "$HERE/ref_harness.cpp" "$TMP/limg_simd.cpp" "$TMP/limg_threading.cpp" -lpthread
$CXX $FLAGS -ffast-math -shared -o "$OUT/liblimg_ref_fastmath.so" "$HERE/ref_harness.cpp" "$TMP/limg_simd.cpp" "$TMP/limg_threading.cpp" -lpthread
echo "built $OUT/liblimg_ref.so $OUT/liblimg_ref_fastmath.so"

# The reference's own CALLER on the HIP library: src/main.cpp, unmodified, compiled next to a one-line `limg.h` that forwards to include/limg_hip_shim.hpp and
# linked against limg_amd/liblimg_hip.so (what tests/test_shim_ref_main.py does in a temp dir -- here the binary is kept, in oracle/_ref like the reference's
# libraries above, so that it travels to the GPU box, which has no /root/reference: tests/test_gpu_ref_main.py runs it there).  main.cpp is copied into the
# throw-away directory only because its `#include "limg.h"` would otherwise find the reference's own header beside it.
LIB="$HERE/../limg_amd/liblimg_hip.so"
if [ -f "$LIB" ] && [ -f "$REF/src/main.cpp" ]; then
  mkdir -p "$TMP/main"
  cp "$REF/src/main.cpp" "$TMP/main/main.cpp"
  echo '#include "limg_hip_shim.hpp"' > "$TMP/main/limg.h"
  ROCM_LIB=${ROCM_LIB:-/opt/rocm/lib}
  $CXX -std=c++17 -O1 -w -I"$TMP/main" -I"$HERE/../include" -I"$REF/3rdParty/stb/include" "$TMP/main/main.cpp" -o "$OUT/limg_ref_main_on_hip" \
      -L"$HERE/../limg_amd" -llimg_hip -lpthread '-Wl,-rpath,$ORIGIN/../../limg_amd' -Wl,-rpath-link,"$ROCM_LIB" -Wl,-rpath,"$ROCM_LIB"
  echo "built $OUT/limg_ref_main_on_hip (the reference's src/main.cpp on the shim)"
fi
