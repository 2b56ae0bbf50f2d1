#!/usr/bin/env bash
# Builds the CPU oracle with AddressSanitizer + UndefinedBehaviorSanitizer and replays the golden cases through it
# (GPU sanitizers are not available on the pool; the CPU restatement is the place to catch UB such as bad shifts).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$(mktemp -d)
trap 'rm -rf "$OUT"' EXIT
gcc -std=c11 -O1 -g -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -o "$OUT/liblimg_oracle_asan.so" "$ROOT/oracle/limg_oracle.c" "$ROOT/oracle/limg_oracle_blocked.c" -lm -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 - "$ROOT" "$OUT/liblimg_oracle_asan.so" <<'PY'
import sys
root, lib = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
import numpy as np
from oracle.bind import Oracle, PLANES
import golden_util as gu
o = Oracle(lib)
idx, z = gu.cases()
for i, m in enumerate(idx):
    r = o.encode3d(z["c%02d_in" % i], m["alpha"], error_factor=m["ef"], pool_threads=m["pool"], fast=m["fast"], dither_mode=m["dither"])
    for k in PLANES:
        assert np.array_equal(r[k], z["c%02d_%s" % (i, k)]), (i, k)
rng = np.random.default_rng(0)
img = rng.integers(0, 2**32, (40, 61), dtype=np.uint32)
o.encode3d(img, True); o.encode3d(img, False, pool_threads=3, worker_threads=2)
# the merged-block encoder's restatement (limg_oracle_blocked.c): random bytes, a gradient (large rectangles), ragged sizes, accurate mode
o.blocked_encode3d(img, True); o.blocked_encode3d(img, False, fast=False)
o.blocked_encode3d(o.random_gradient(203, 61, 5, True), True); o.blocked_encode3d(o.photo_noise(96, 50, 9), True, error_factor=25)
print("oracle clean under ASan+UBSan on %d golden cases + random bytes + the merged-block encoder" % len(idx))
PY
