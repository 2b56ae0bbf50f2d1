#!/usr/bin/env python3
"""Writes limg_amd/BUILD_STAMP.json = {"head": <git commit, "+dirty" if product files differ from it>, "time": ...}.  Run before a gpurun call: the GPU box gets a
snapshot of the tree without .git, and bench.py puts this `head` on its JSON line there (next to lib_sha / src_sha, which it computes itself)."""
import json
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "limg_amd", "include", "bench.py"], capture_output=True, text=True).stdout.strip()
out = {"head": head + ("+dirty" if dirty else ""), "time": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
json.dump(out, open(os.path.join(ROOT, "limg_amd", "BUILD_STAMP.json"), "w"))
print(out)
