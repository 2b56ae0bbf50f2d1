"""Adds the accurate-bit-crush entries (`--accurate-bit-crushing`, fastBitCrushing = false: src/limg_bit_crush.h:668-830) at 1024x1024 to tests/golden/hashes.json,
from the REAL reference (oracle/_ref/liblimg_ref.so, strict build) -- the other entries come from tools/make_golden.py and are left untouched.
Run in the build container (needs /root/reference via oracle/build_ref.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bind import Oracle, Ref, PLANES  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
orc, ref = Oracle(), Ref()
path = os.path.join(G, "hashes.json")
hashes = json.load(open(path))
for name, img, alpha, kw in (("pn1024_accurate", orc.photo_noise(1024, 1024, 1), True, dict(fast=False)),
                             ("rg1024_accurate", orc.random_gradient(1024, 1024, 1, True), True, dict(fast=False))):
    out = ref.encode3d(img, alpha, **kw)
    e = {k: orc.fnv(out[k]) for k in PLANES}
    e["input"] = orc.fnv(img)
    e["psnr"], e["mse"] = ref.compare(img, out["pDecoded"], alpha)
    e["shape"] = list(img.shape)
    e["alpha"] = alpha
    e["kw"] = kw
    hashes[name] = e
    print(name, e["psnr"])
json.dump(hashes, open(path, "w"), indent=1)
