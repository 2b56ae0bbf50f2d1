#!/usr/bin/env python3
"""bench.py -- limg encode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1: one rank per GPU over RCCL.  Either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env),
      or -- WORLD_SIZE unset -- bench.py starts the N ranks itself: the parent touches no GPU, checks that the node has N devices (refuses
      loudly otherwise: it never reports a smaller n_gpus than asked), and runs N fresh child processes of this file.

A "step" is one pass of the hot path over one image: `limg_encode3d_test`-equivalent work (fit, factors, shift search,
dither, all 11 planes stored, decode) through the C ABI of liblimg_hip.so, input already resident in HBM.
Workload at every N: BASELINE.json configs[2], a synthetic 8192x8192 RGBA photo-noise image per GPU (seed = 1 + rank;
the metric is quoted on "8K RGBA"), errorFactor 100, fast bit crushing, single dither chain.  Images are independent,
so N GPUs encode N images with no data-path collective ("weak" scaling); the only torch.distributed traffic is the
barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  `roofline` prices the encode kernel (one persistent launch per image; `--split` = the
three-launch fallback path) against the HBM roofline with the algorithmic 39 B/px of SURVEY.md 8(d):
achieved = 39 B * pixels / average kernel duration (HIP events on the launch stream inside the timed region).
`cpu_baseline` times the real reference (oracle/_ref, kind "reference") -- or, where that build is absent, the CPU
oracle (kind "port") -- on a bounded band of the same image on this host's cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PX = 39          # 4 B read + 35 B written (plane-compatible mode)
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Counter-derived figures (HBM traffic, VALU instructions per block) come from profiles/pmc_by_workload.json, keyed by workload and written by
# tools/prof_summary.py from separate rocprofv3 --pmc passes of this very command; a workload without an entry prints null -- never a stale number.
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_by_workload.json")
# Measured VALU issue ceilings of the chip, wave64 instructions per second (profiles/archive/r02_valu_ceiling.md, tools/valu_ceiling.hip, 5 waves per SIMD):
# full-rate class (v_add/mul/fma_f32, v_add_u32, logic ops ...) and half-rate class (v_mad_i32_i24, v_pk_*, VOP3-only, DPP, v_cvt_* ... -- what this path is made of)
VALU_FULL_RATE_PER_S = 960e9
VALU_HALF_RATE_PER_S = 578e9


PMC_ROUND = "r06"   # the round whose kernels this file benches: a counter entry measured on another round's build is refused (lines that quoted round-2 counters for round-3 kernels)


# Every leg that fails is recorded here, printed on the line (`errors`) and turns the exit status non-zero: a bench line must not look healthy while one of its
# legs raised (a look-back timeout sat in config.two_streams.error of three committed "final" lines and the process exited 0).
ERRORS = []


def leg_failed(leg, exc):
    ERRORS.append({"leg": leg, "error": repr(exc)})
    print("bench.py: leg %r FAILED: %r" % (leg, exc), file=sys.stderr, flush=True)
    return {"error": repr(exc)}


# Auxiliary evidence that could not be gathered (not a measurement, not a correctness check): on the line as `warnings`, never fatal.
WARNINGS = []
HUNG_THREAD = []  # a helper thread that did not come back: the process then leaves through os._exit after printing its line


def leg_warned(leg, what):
    WARNINGS.append({"leg": leg, "warning": str(what)})
    print("bench.py: %s: %s" % (leg, what), file=sys.stderr, flush=True)


class known_driver_noise_filtered:
    """While the HIP runtime comes up, libdrm prints "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory" on this image's boxes (a missing marketing-name
    table: harmless, not ours).  The driver keeps bench.py's stderr as evidence that nothing went wrong -- a look-back time-out would be reported there -- so that one
    KNOWN line is taken out.  stderr (the file descriptor: the message comes from C) goes through a PIPE that a reader thread drains line by line, forwarding everything
    but that line as it arrives: a crash, abort or hang inside the window loses nothing (a temporary file replayed on exit did)."""

    def __enter__(self):
        import threading
        sys.stderr.flush()
        self.saved = os.dup(2)
        r, w = os.pipe()
        os.dup2(w, 2)
        os.close(w)
        out = self.saved

        def pump():
            with os.fdopen(r, "rb", buffering=0) as src:
                buf = b""
                while True:
                    chunk = src.read(4096)
                    if not chunk:
                        break
                    buf += chunk
                    while b"\n" in buf:
                        line, buf = buf.split(b"\n", 1)
                        if b"libdrm/amdgpu.ids: No such file or directory" not in line:
                            os.write(out, line + b"\n")
                if buf and b"libdrm/amdgpu.ids: No such file or directory" not in buf:
                    os.write(out, buf)

        self.thread = threading.Thread(target=pump, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        sys.stderr.flush()
        os.dup2(self.saved, 2)  # closes the pipe's last write end: the reader sees EOF
        self.thread.join(timeout=5.0)
        os.close(self.saved)
        return False


def _sha256(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
    return h.hexdigest()


def source_sha():
    """sha256 over the product's sources (limg_amd/csrc/*, include/*, limg_amd/build.py + isa_check.py; sorted by name): identifies the code a line was measured on wherever
    .git is absent (the GPU box gets a snapshot without it) and can be recomputed from any checkout."""
    files = []
    for d in ("limg_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".hip", ".h", ".hpp", ".cpp")):
                files.append(os.path.join(ROOT, d, f))
    files.append(os.path.join(ROOT, "limg_amd", "build.py"))
    files.append(os.path.join(ROOT, "limg_amd", "isa_check.py"))
    return _sha256(files)


def provenance():
    """What build a line was measured on: `lib_sha` = sha256 of the liblimg_hip.so that was loaded, `src_sha` = source_sha(), `head` = the git commit (from .git where
    it exists, else from limg_amd/BUILD_STAMP.json, which tools/stamp.py writes before a gpurun call; null if neither)."""
    import limg_amd
    head = None
    try:
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
        if head and subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "limg_amd", "include", "bench.py"], capture_output=True, text=True, timeout=10).stdout.strip():
            head += "+dirty"
    except Exception:
        head = None
    if not head:
        try:
            head = json.load(open(os.path.join(ROOT, "limg_amd", "BUILD_STAMP.json"))).get("head")
        except Exception:
            head = None
    return {"head": head, "lib_sha": _sha256([limg_amd.LIB_PATH]), "src_sha": source_sha(), "bench_sha": _sha256([os.path.abspath(__file__)])[:16]}


def emit(line):
    """The ONE JSON line of rank 0: with the build's provenance and the list of failed legs.  A non-empty list also fails the process (see main)."""
    line.update(provenance())
    line["errors"] = list(ERRORS)
    line["warnings"] = list(WARNINGS)
    print(json.dumps(line), flush=True)


def pmc_entry(key):
    """The counter-derived figures of one workload, or None when there are none -- or when the entry was not measured on this round's build (its `source` names the
    profile run, tools/prof.sh <tag>, and tags carry the round)."""
    try:
        e = json.load(open(PMC_FILE)).get(key)
    except Exception:
        return None
    if e is None or ("_%s" % PMC_ROUND) not in str(e.get("source", "")):
        return None
    return e


def pmc_stale_source(key):
    """Why pmc_entry(key) is None although the file has the key: the stale entry's source, for the line's `pmc_refused` field."""
    try:
        e = json.load(open(PMC_FILE)).get(key)
    except Exception:
        return None
    return None if e is None or ("_%s" % PMC_ROUND) in str(e.get("source", "")) else e.get("source")


def instruction_floor(pmc, algo_bytes):
    """The fraction of the HBM roofline this instruction count could reach if the kernels issued at the chip's measured rate for their instruction class
    (profiles/archive/r02_valu_ceiling.md): algorithmic bytes / (VALU instructions per launch / ceiling) / peak.  The distance between `frac` and this number is what
    scheduling can still recover; the distance between this number and 1 only fewer instructions can."""
    if not pmc or not pmc.get("valu_instr_per_launch"):
        return None
    t = pmc["valu_instr_per_launch"] / VALU_HALF_RATE_PER_S
    return round(algo_bytes / t / 1e9 / HBM_PEAK_GBPS, 4)


def parse_size(text):
    """`--size`: N (an N x N image) or WxH."""
    text = str(text)
    if "x" in text:
        w, h = (int(v) for v in text.split("x"))
        return w, h
    return int(text), int(text)


def workload_key(args, W, H):
    mode = "split" if args.split else "fused"
    return "%dx%d_%s_ef%d_%s%s%s%s" % (W, H, args.workload, args.error_factor, mode, "" if args.forced_shift < 0 else "_shift%d" % args.forced_shift,
                                       "_compact" if args.compact else "", ("_fastfloat" if args.float_mode == "fast" else "") + ("_legacyfit" if args.legacy_float_stage else "")
                                       + ("_accurate" if args.accurate else "") + ("_rgb" if args.rgb else ""))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(width, seed, budget_s=30.0, height=None):
    """The reference's own CPU path on this host, same image as the GPU leg (whole image, not a band): both builds of oracle/_ref (the project's own
    fast-math flags, project.lua:38, and the strict-IEEE build the parity tests pin), `_test` style (all planes + decode, src/limg.cpp:2105-2138) and
    `_perf` style (nothing stored, :2140-2173), single thread and thread pool.  `value` = the reference's own configuration: fast-math, pool, `_test` style
    (what the GPU headline computes).  Without oracle/_ref (clean checkout: the .so files are git-ignored) the scalar oracle is timed instead and the
    entry says so in `kind` -- that number is ~10x lower than the reference's and must not be read as the reference."""
    import numpy as np
    from oracle.bind import Oracle, Ref, ref_available
    orc = Oracle()
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    pool = max(1, min(avail, 16))  # (the scalar-oracle fallback below only)
    t_start = time.perf_counter()

    def best(fn, reps):
        out = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); out.append(time.perf_counter() - t)
            if time.perf_counter() - t_start > budget_s:
                break
        return min(out)

    if not ref_available():
        rows = min(width, 1024)
        band = orc.photo_noise(width, rows, seed)
        tn = best(lambda: orc.encode3d(band, True, error_factor=100, pool_threads=pool, worker_threads=pool), 2)
        return {"value": round(width * rows / tn / 1e6, 2), "unit": "Mpixels/s", "cores": pool, "kind": "port (reference build absent)", "cpu": cpu_model(),
                "value_is_not_the_reference": True,
                "sample": "first %d rows of the bench image through the scalar CPU oracle with %d worker threads; oracle/_ref is not built here, so this is NOT "
                          "the reference's SIMD path (expect ~10x below it)" % (rows, pool)}
    height = height or width
    img = orc.photo_noise(width, height, seed)
    px = width * height
    # pools: the box's share for one of its 8 GPUs (host threads / 8), all host threads (what the reference's own tool takes: limg_threading_max_threads(),
    # src/main.cpp:165), and the 16 of rounds 1-2 for continuity.  The reference makes 4 row strips per pool thread (src/limg.cpp:2114-2134).
    pools = {"pool16": max(1, min(avail, 16)), "pool_gpu_share": max(1, avail // 8), "pool_allcores": avail}
    res = {}
    for build in ("fastmath", "strict"):
        if not ref_available(fastmath=(build == "fastmath")):
            continue
        ref = Ref(fastmath=(build == "fastmath"))
        res[build] = {}
        for name, pool in pools.items():
            if build == "strict" and name != "pool_gpu_share":
                continue  # the strict-IEEE build (what the parity tests pin) at one pool size only: the two builds run at the same speed
            res[build]["test_" + name] = px / best(lambda: ref.encode3d(img, True, error_factor=100, pool_threads=pool), 2) / 1e6
            res[build]["perf_" + name] = px / best(lambda: ref.encode3d_perf(img, True, error_factor=100, pool_threads=pool), 2) / 1e6
        if time.perf_counter() - t_start < budget_s * 0.7:  # single thread: a quarter of the image is enough (linear in the rows)
            q = np.ascontiguousarray(img[: height // 4])
            res[build]["test_1thread"] = q.size / best(lambda: ref.encode3d(q, True, error_factor=100, pool_threads=0), 1) / 1e6
            res[build]["perf_1thread"] = q.size / best(lambda: ref.encode3d_perf(q, True, error_factor=100, pool_threads=0), 1) / 1e6
        del ref
    head = res.get("fastmath") or res["strict"]
    # value = the reference at the pool size that suits it best on this host (its own tool takes all host threads, which on a 256-thread box is its slowest setting:
    # 1024 one-block-row strips and a polling pool); every setting is listed
    key = max((k for k in head if k.startswith("test_pool")), key=lambda k: head[k])
    return {"value": round(head[key], 2), "unit": "Mpixels/s", "cores": pools[key[len("test_"):]], "kind": "reference",
            "cpu": cpu_model(), "host_cores_available": avail, "pools": pools,
            "builds_Mpixels_per_s": {b: {k: round(v, 2) for k, v in d.items()} for b, d in res.items()},
            "sample": "the whole %dx%d bench image (%.1f Mpx), the real reference compiled by oracle/build_ref.sh; value = its own build flags (-ffast-math, project.lua:38), "
                      "limg_encode3d_test style (all planes + decode), at the BEST of three thread-pool sizes (%s; 4 row strips per pool thread, src/limg.cpp:2114-2134), best of 2 runs; "
                      "builds_Mpixels_per_s lists fast-math / strict-IEEE x _test / _perf style x pool of 16 / host threads over 8 GPUs (%d) / all host threads (%d: what the "
                      "reference's own tool takes, src/main.cpp:165) / single thread (single thread on the first quarter of the rows)"
                      % (width, height, px / 1e6, key[len("test_"):], pools["pool_gpu_share"], pools["pool_allcores"])}


def run_sharded(args, g, dist, rank, world):
    """BASELINE configs[3] / configs[4]: the multi-GPU shapes of the path.  Same JSON contract;
    the plane reassembly on rank 0 is timed apart from the encode and reported in `config` (it is not part of `value`)."""
    import torch
    import numpy as np
    from limg_amd import shard
    if args.config == 4:
        W = H = 4096 if args.size == 8192 else args.size
        kind = "random_gradient" if args.workload == "photo_noise" and args.size == 8192 else args.workload
        mine = shard.batch_assignment(args.images, world, rank)
        units = [(g.synth_device(kind, W, H, seed=1 + i), g.alloc_planes_device(W, H)) for i in mine]
        total_px = args.images * W * H
        name = "batch of %d synthetic %dx%d RGBA %s images (seeds 1..%d), image i -> rank i %% %d" % (args.images, W, H, kind, args.images, world)
        rows = None
    else:
        W = H = 16384 if args.size == 8192 else args.size
        kind = args.workload
        strips = shard.strip_rows(H, 8)   # the reference's partition for a pool of 2 threads (2 * 4 strips), src/limg.cpp:2114-2134
        if 8 % world:
            raise SystemExit("--config 5 needs a world size that divides 8")
        per = 8 // world
        rows = strips
        units = []
        if args.single_chain:  # one chain through the whole image: a rank's consecutive strips are one taller strip (any world size that divides 8)
            strips = [(strips[r * per][0], strips[(r + 1) * per - 1][1]) for r in range(world)]
            rows = strips
            y0, y1 = strips[rank]
            units.append((g.synth_device(kind, W, y1 - y0, seed=1, y0=y0), g.alloc_planes_device(W, y1 - y0)))
        else:
            for sidx in range(rank * per, (rank + 1) * per):
                y0, y1 = strips[sidx]
                units.append((g.synth_device(kind, W, y1 - y0, seed=1, y0=y0), g.alloc_planes_device(W, y1 - y0)))
        total_px = W * H
        name = ("one synthetic %dx%d RGBA %s image as 8 strips of whole block rows with restarted dither chains (== reference with a pool of 2 threads), "
                "%d strips per rank" % (W, H, kind, per) if not args.single_chain else
                "one synthetic %dx%d RGBA %s image as %d strips of whole block rows, one per rank, ONE dither chain through all of them (== the reference with pThreadPool == nullptr; "
                "8-byte all-gather between the E and the F step)" % (W, H, kind, world))
    torch.cuda.synchronize()

    single_chain = args.config == 5 and args.single_chain
    chain_over_gloo = False
    if single_chain:
        if world == 1:  # a communicator of one rank: the exchange degenerates to a copy (what a one-GPU box can run)
            g.comm_init(g.comm_unique_id(), 0, 1)
        elif dist is None:
            raise SystemExit("--single-chain with several ranks needs torch.distributed")
        elif dist.get_backend() != "nccl":
            # REHEARSAL (--share-gpus: several ranks on one card; RCCL refuses duplicate devices): the same chain through the entry's two exchange-free halves
            # (limg_hip_encode3d_chain_device), the 8-byte counts all-gathered over gloo, the bases by limg_hip_host_chain_bases -- the arithmetic k_chain_base does on
            # the device.  Every rank is its own process with its own context: their persistent / split kernels share the one GPU.
            chain_over_gloo = True
            name += " [REHEARSAL: ranks share a GPU, the 8-byte exchange over gloo instead of RCCL]"
        else:
            g.comm_init_from_torch(dist)
        g._comm_ready = True
        before = [(y0 // 8) * (W // 8) for (y0, _) in strips]
        if chain_over_gloo:
            import limg_amd
            d_calls = torch.zeros(1, dtype=torch.int64, device="cuda")
            d_base = torch.zeros(1, dtype=torch.int64, device="cuda")

    # --contexts K (config 4): the rank's images go round-robin over K contexts, each on a HIP stream of its own -- the next image's float-stage kernel fills the
    # CUs that the previous image's persistent kernel leaves idle while its last strips drain (a 4096^2 image is only ~5 strips per workgroup)
    ctxs, streams = [g], [torch.cuda.current_stream()]
    if args.config == 4 and args.contexts > 1:
        import limg_amd
        ctxs += [limg_amd.LimgHip(torch.cuda.current_device()) for _ in range(args.contexts - 1)]
        streams += [torch.cuda.Stream() for _ in range(args.contexts - 1)]
        name += ", round-robin over %d contexts / HIP streams" % args.contexts

    batched = args.config == 4 and len(ctxs) == 1 and not args.no_batch and len(units) > 1
    if args.sub_images or args.wg_per_cu:
        g.set_options(batch_sub_images=args.sub_images, test_wg_per_cu=args.wg_per_cu, test_pipeline=args.pipeline_knobs)
    n_units = len(units)
    sub_eff = args.sub_images if args.sub_images > 0 else (0 if args.sub_images < 0 else (8 if n_units >= 32 else (4 if n_units >= 16 else 0)))  # limg_hip_options.batch_sub_images
    if batched:
        name += (", the rank's %d images in ONE launch pair (limg_hip_encode3d_batch_device)" % len(units) if not sub_eff else
                 ", the rank's %d images as a pipeline of sub-batches of %d (limg_hip_encode3d_batch_device, batch_sub_images: k_fit_tpb of sub-batch k + 1 next to the persistent "
                 "kernel of sub-batch k)" % (len(units), sub_eff))

    def step():
        if batched:  # the reference's per-file loop (src/main.cpp:278-323) as one call: one k_fit_tpb grid + one persistent launch over all images
            g.encode3d_batch_device([u[0] for u in units], True, [u[1] for u in units], error_factor=args.error_factor, pool_threads=0, fast=True)
            return
        for i, (img, planes) in enumerate(units):
            if single_chain and chain_over_gloo:
                g.encode3d_chain_device(img, True, planes, 1, calls=d_calls, blocks_before=before[rank], error_factor=args.error_factor)
                mine_calls = d_calls.cpu()  # (waits for the E step + scan)
                allc = [torch.zeros_like(mine_calls) for _ in range(world)]
                dist.all_gather(allc, mine_calls)
                bases = limg_amd.host_chain_bases(np.array([int(c.item()) for c in allc], dtype=np.uint64))
                d_base.fill_(int(bases[rank]))
                g.encode3d_chain_device(img, True, planes, 2, base=d_base, blocks_before=before[rank], error_factor=args.error_factor)
            elif single_chain:  # ONE dither chain through the 8 strips (== the reference with pThreadPool == nullptr): an 8-byte all-gather between E and F step
                g.encode3d_single_chain_device(img, True, planes, before[rank], error_factor=args.error_factor)
            elif len(ctxs) > 1:
                with torch.cuda.stream(streams[i % len(ctxs)]):
                    ctxs[i % len(ctxs)].encode3d_device(img, True, planes, error_factor=args.error_factor, pool_threads=0, fast=True)
            else:
                g.encode3d_device(img, True, planes, error_factor=args.error_factor, pool_threads=0, fast=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    g.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernels = g.profile_end(args.steps * max(len(units), 1))
    launches_per_step = 1 if batched else len(units)
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize()
    g.check()
    if single_chain and chain_over_gloo:
        t = torch.tensor([rank], dtype=torch.int64)
        views = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(views, t)
        collective = {"backend": "gloo", "world_size": world, "rccl_version": None, "comm_ranks": len({int(v.item()) for v in views}),
                      "note": "rehearsal: the ranks are processes sharing one GPU; the chain's 8-byte exchange went through gloo + limg_hip_host_chain_bases"}
    elif single_chain:
        info = g.comm_info()
        collective = {"backend": None if dist is None else dist.get_backend(), "world_size": world, "rccl_version": info["rccl_version"], "comm_ranks": info["ranks"]}
    else:
        collective = collective_evidence(g, dist, rank, world)
    golden = verify_golden(args, units, rows, rank, world, dist, single_chain, W, H, kind) if (args.verify_golden and not COLLECTIVES_OFF) else None

    # reassembly on rank 0 (config 5: the strips of the one image; config 4: every image's planes), timed apart
    gather_ms = None
    gathered_bytes = 0
    if dist is not None and not args.no_gather and COLLECTIVES_OFF:
        leg_warned("gather", "skipped: a rank's RCCL evidence helper hung, no further collective is issued")
    if dist is not None and not args.no_gather and not COLLECTIVES_OFF:
        on_gpu = dist.get_backend() == "nccl"
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.config == 5 and args.gather_stream:
            # the north-star's shape: only the bitstream crosses xGMI (limg_hip_gather_stream: RCCL behind the C ABI); rank 0 decodes every strip into its rows
            per = 1 if single_chain else 8 // world
            full = torch.empty((H, W), dtype=torch.int32, device="cuda") if rank == 0 else None
            cap = sum(g.stream_bound(W, y1 - y0) + 16 for (y0, y1) in rows[::per][:world]) if rank == 0 else 0
            gbuf = torch.empty(cap, dtype=torch.uint8, device="cuda") if rank == 0 else None
            for k in range(per):
                st, nb = g.encode_stream_device(units[k][0], True, error_factor=args.error_factor)
                res = shard.gather_streams(g, st, nb, dist, dst=0, out=gbuf)
                if rank == 0:
                    buf, offs = res
                    buf = buf if buf.is_cuda else buf.cuda()
                    for r in range(world):
                        y0, y1 = rows[r * per + k]
                        o0 = int(offs[r]); nbr = int(offs[r + 1]) - o0
                        g.decode_stream_device(buf[o0:o0 + nbr], nbr, W, y1 - y0, out=full[y0:y1])
                    gathered_bytes += int(offs[world])
            torch.cuda.synchronize()
            if rank == 0:   # decoded strips == the pDecoded planes of the same strips (this rank's own, checked here)
                for k in range(per):
                    y0, y1 = rows[k]
                    assert torch.equal(full[y0:y1], units[k][1]["pDecoded"]), "stream round trip differs from pDecoded"
        elif args.config == 5:
            per = 1 if single_chain else 8 // world
            for k in range(per):   # strip k of every rank; rows of the gathered pieces come from the strip table
                planes = units[k][1] if on_gpu else {n: v.cpu() for n, v in units[k][1].items()}
                piece_rows = [rows[r * per + k] for r in range(world)]
                full = shard.gather_planes(planes, piece_rows, W, dist, dst=0)
                if rank == 0:
                    gathered_bytes += sum(v.numel() * v.element_size() for v in full.values())
                del full
        else:
            n_round = (args.images + world - 1) // world
            for k in range(n_round):
                have = k < len(units)
                src_ranks = [r for r in range(world) if k * world + r < args.images]
                for n in limg_planes():
                    tsr = units[k][1][n] if have else None
                    if tsr is not None and not on_gpu:
                        tsr = tsr.cpu()
                    if rank == 0:
                        bufs = [torch.empty_like(tsr) for r in src_ranks if r != 0]
                        reqs = [dist.irecv(b, src=r) for b, r in zip(bufs, [r for r in src_ranks if r != 0])]
                        for q in reqs:
                            q.wait()
                        gathered_bytes += sum(b.numel() * b.element_size() for b in bufs)
                    elif rank in src_ranks:
                        dist.send(tsr.contiguous(), dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3

    if rank == 0:
        kavg = kernels.mean(axis=0) if len(kernels) else np.zeros(3)
        ksum = float(kavg.sum())  # k_fit_tpb + k_encode_persistent of one unit (single chain: the E/scan, exchange and F intervals)
        px_per_launch = (units[0][0].numel() if units else 0) * (len(units) if batched else 1)
        pmc_key = "config%d_%s%s%s%s" % (args.config, ("batched_n%d" % n_units) if batched else ("single_chain" if single_chain else "single"),
                                         "" if args.contexts == 1 else "_ctx%d" % args.contexts, "" if not (batched and sub_eff) else "_sub%d" % sub_eff,
                                         "" if (args.size == 8192 and args.workload == "photo_noise") else "_%s%d" % (args.workload, args.size))
        pmc = pmc_entry(pmc_key)
        if pmc and batched and sub_eff:
            # the counters are means per DISPATCH; a pipelined list is ceil(n / sub) dispatches of each kernel per step: scale to the whole list
            n_sub = (n_units + sub_eff - 1) // sub_eff
            pmc = dict(pmc)
            for k in ("fetch_kib", "write_kib", "valu_instr_per_launch", "salu_instr_per_launch", "lds_instr_per_launch"):
                if pmc.get(k) is not None:
                    pmc[k] = pmc[k] * n_sub
            pmc["source"] = "%s (per-dispatch means x %d sub-batches)" % (pmc.get("source"), n_sub)
        achieved = ALGO_BYTES_PER_PX * px_per_launch / (ksum * 1e-3) / 1e9 if ksum > 0 else 0.0
        line = {
            "metric": "encode Mpixels/s, RGBA (limg_encode3d_test-equivalent: all 11 planes stored), BASELINE configs[%d]" % (args.config - 1),
            "value": round(total_px * args.steps / elapsed / 1e6, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/i32 integer stage + f32 float stage (bit-exact vs the reference's strict SSE build)", "data": "synthetic",
            "config": {"workload": name + ", errorFactor %d, fast bit-crush" % args.error_factor,
                       "parallelism": "no data-path collective; plane reassembly on rank 0 timed apart",
                       "gather_ms": None if gather_ms is None else round(gather_ms, 3), "gathered_bytes_rank0": gathered_bytes,
                       "collective": collective, "golden": golden,
                       "gather_backend": None if dist is None else dist.get_backend(), "gathered": "LMG3 streams, decoded on rank 0" if args.gather_stream else "planes"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": None if not pmc or pmc.get("fetch_kib") is None else int((2 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024),
                         "valu_busy": None if not pmc else pmc.get("valu_busy"), "pmc_source": None if not pmc else pmc.get("source"), "pmc_key": pmc_key,
                         "pmc_refused_stale_source": pmc_stale_source(pmc_key), "instruction_floor": instruction_floor(pmc, ALGO_BYTES_PER_PX * px_per_launch),
                         "valu_instr_per_block": None if not pmc or not pmc.get("valu_instr_per_launch") else round(pmc["valu_instr_per_launch"] / (px_per_launch / 64.0), 1),
                         "algorithmic_bytes_per_launch": int(ALGO_BYTES_PER_PX * px_per_launch), "launch_pairs_per_step": launches_per_step,
                         "kernels_ms": ({"k_fit_tpb of the first sub-batch (alone)": round(float(kavg[0]), 4), "the pipeline: k_encode_persistent of every sub-batch, k_fit_tpb of the next one beside it": round(float(kavg[1]), 4)}
                                        if (batched and sub_eff) else {"k_fit_tpb": round(float(kavg[0]), 4), "k_encode_persistent": round(float(kavg[1]), 4)}) if not single_chain else
                                       {"E step + scan": round(float(kavg[0]), 4), "all-gather + base": round(float(kavg[1]), 4), "F step": round(float(kavg[2]), 4)},
                         "note": "per launch pair = %s on rank 0: k_fit_tpb + k_encode_persistent; HIP events on the launch stream"
                                 % ("the rank's whole image list" if batched else "one image (config 4) / one strip (config 5)")},
        }
        emit(line)


from limg_amd.shard import sum64_device  # noqa: E402  (the parallel checksum of tests/golden/fullsize.json, computed on the device)


def verify_golden(args, units, rows, rank, world, dist, single_chain, W, H, kind):
    """--verify-golden (config 5 at its real size): every rank checks the strips IT produced against the real reference's per-strip checksums of
    tests/golden/fullsize.json (pn16384_strips: one chain through the image; pn16384_pool2_strips: the chain restarted per strip) -- no plane leaves the rank.
    A mismatch is a failed leg (non-zero exit)."""
    import torch
    if args.config == 4:
        # BASELINE config 4: every image this rank encoded (image i of the batch = seed 1 + i, image i -> rank i % world) against the real reference's whole-image
        # checksums of all 11 planes (tests/golden/fullsize.json rg4096_batch64, seeds 1..64)
        try:
            from limg_amd import shard
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["rg4096_batch64"]
            if (W, H, kind) != (gold["w"], gold["h"], "random_gradient") or args.error_factor != 100 or args.images > len(gold["images"]):
                raise RuntimeError("the golden entry rg4096_batch64 is for up to %d %dx%d random_gradient images, errorFactor 100" % (len(gold["images"]), gold["w"], gold["h"]))
            mine = shard.batch_assignment(args.images, world, rank)
            bad = ["image %d %s" % (i, n) for i, (img, planes) in zip(mine, units) for n, want in gold["images"][i]["sum64"].items() if sum64_device(planes[n]) != want]
            flag = torch.tensor([len(bad)], dtype=torch.int64, device="cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu")
            allchecked = [list(mine)]
            if dist is not None:
                dist.all_reduce(flag)
                allchecked = [None] * world
                dist.all_gather_object(allchecked, list(mine))
            if int(flag.item()) != 0:
                raise RuntimeError("planes differ from the reference's: %s (this rank), %d mismatches over all ranks" % (bad[:4], int(flag.item())))
            return {"entry": "rg4096_batch64", "file": "tests/golden/fullsize.json", "images_checked_by_rank": allchecked, "planes_per_image": 11, "ok": True}
        except Exception as e:
            return leg_failed("verify_golden", e)
    entry = "pn16384_strips" if single_chain else "pn16384_pool2_strips"
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))[entry]
        if (W, H, kind) != (gold["w"], gold["h"], "photo_noise") or args.error_factor != 100:
            raise RuntimeError("the golden entry %s is for %dx%d photo_noise, errorFactor 100" % (entry, gold["w"], gold["h"]))
        sr = gold["strip_rows"]
        bad, checked = [], []
        per = len(units)
        for k, (img, planes) in enumerate(units):
            y0 = rows[rank * per + k][0] if not single_chain else rows[rank][0]
            for j in range(img.shape[0] // sr):  # the 2048-row golden strips inside this unit
                si = y0 // sr + j
                checked.append(si)
                for name, want in gold["strips"][si]["sum64"].items():
                    if sum64_device(planes[name][j * sr:(j + 1) * sr]) != want:
                        bad.append("strip %d %s" % (si, name))
        flag = torch.tensor([len(bad)], dtype=torch.int64, device="cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu")
        allchecked = [checked]
        if dist is not None:
            dist.all_reduce(flag)
            allchecked = [None] * world
            dist.all_gather_object(allchecked, checked)
        if int(flag.item()) != 0:
            raise RuntimeError("planes differ from the reference's: %s (this rank), %d mismatches over all ranks" % (bad[:4], int(flag.item())))
        return {"entry": entry, "file": "tests/golden/fullsize.json", "strips_checked_by_rank": allchecked, "planes_per_strip": 11, "ok": True}
    except Exception as e:
        return leg_failed("verify_golden", e)


BLOCKED_BYTES_PER_PIXEL = 4 + 4 * 9 + 4  # the RGBA pixel in; pDecoded, pShiftABCX, six colour planes, pBlockIndex (u32) and three factor planes + pBitsPerPixel (u8) out


def blocked_roofline(px, stage_ms, pmc_key):
    """Kernel-only rate of the merged-block encoder: its algorithmic bytes over the GPU time of its kernels alone (HIP events: pass 1, the similarity kernels, and the sums of the
    worker's fit + search and expansion + store launches) -- what the GPU side would deliver with no host stage in the way.  The end-to-end rate is `value`.
    Counters (tools/prof_blocked.sh, per image): HBM traffic of all its kernels, and for the kernel furthest below any roofline, k_blocked_fit_search, the floor its own
    instruction count sets: VALU instructions per image / the chip's measured issue rate for this instruction class."""
    gpu_ms = stage_ms["pass1_kernel"] + stage_ms["match_kernels"] + stage_ms["fit_search_kernel"] + stage_ms["expand_store_kernels"]
    ach = BLOCKED_BYTES_PER_PIXEL * px / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None
    pmc = pmc_entry(pmc_key)
    fs = ((pmc or {}).get("per_kernel") or {}).get("k_blocked_fit_search")
    fit_search = None
    if fs and fs.get("valu_instr"):
        floor_ms = fs["valu_instr"] / VALU_HALF_RATE_PER_S * 1e3
        fit_search = {"valu_instr_per_image": fs["valu_instr"], "valu_instr_per_pixel": round(fs["valu_instr"] / px, 3), "salu_instr_per_image": fs.get("salu_instr"),
                      "waves_per_image": fs.get("waves"), "dispatches_per_image": fs.get("dispatches"), "ms_under_profiler": fs.get("ms"),
                      "issue_floor_ms": round(floor_ms, 3), "valu_busy_own_time": fs.get("valu_busy"), "wait_inst_any_frac": fs.get("wait_inst_any_frac"),
                      "note": "issue_floor_ms = its wave64 VALU instructions per image / %.0f G/s (the measured half-rate-class ceiling, all SIMDs busy); the kernel runs one wave "
                              "per rectangle whose life is a serial pixel-order walk, so its time is the longest rectangles' latency chain, not this floor" % (VALU_HALF_RATE_PER_S / 1e9)}
    return {"bound": "hbm", "achieved": round(ach, 1) if ach else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4) if ach else None,
            "traffic": None if not pmc or pmc.get("fetch_kib") is None else int((2 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024),
            "pmc_key": pmc_key, "pmc_source": None if not pmc else pmc.get("source"), "pmc_refused_stale_source": pmc_stale_source(pmc_key),
            "kernels_ms": {"k_fit_tpb (pass 1)": stage_ms["pass1_kernel"], "k_blocked_match (16 bands)": stage_ms["match_kernels"], "k_blocked_fit_search": stage_ms["fit_search_kernel"],
                           "k_noise_expand_calls + k_blocked_store (+ chain-value uploads)": stage_ms["expand_store_kernels"]},
            "k_blocked_fit_search": fit_search,
            "bytes_per_pixel": BLOCKED_BYTES_PER_PIXEL, "algorithmic_bytes_per_launch": int(BLOCKED_BYTES_PER_PIXEL * px),
            "note": "kernel-only: the GPU's share of one image; the end-to-end rate (`value`) is set by the host stages, serial by construction upstream (greedy raster merge, "
                    "one AES dither chain) -- see config.stage_ms.  k_blocked_fit_search is one wave per rectangle walking its pixels in the reference's order: latency- and "
                    "issue-bound, not an HBM kernel"}


def run_blocked(args, g, dist, rank, world, W, H):
    """The reference CLI's real single-file path (src/main.cpp:255).  A step = one whole merged-block encode of one image, host stages included
    (greedy merge over GPU similarity bits, dither chain walk); input and the 13 output planes stay in HBM."""
    import torch
    import numpy as np
    if args.forced_shift >= 0:  # (kernel-time experiments: the per-rectangle search bypassed)
        g.set_options(forced_shift=(args.forced_shift,) * 3)
    blocked_opts = dict(test_blocked_no_bound=args.no_match_bound, test_blocked_no_order=args.no_order, test_blocked_no_vec_store=args.no_vec_store)
    if args.forced_shift >= 0:
        blocked_opts["forced_shift"] = (args.forced_shift,) * 3
    g.set_options(**blocked_opts)
    img = g.synth_device(args.workload, W, H, seed=1 + rank)
    planes = g.alloc_blocked_planes_device(W, H)
    for _ in range(max(args.warmup, 1)):
        g.blocked_encode3d_device(img, True, planes, error_factor=args.error_factor)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    stages = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.blocked_encode3d_device(img, True, planes, error_factor=args.error_factor)
        stages.append(dict(g.blocked_timing(), **g.blocked_kernel_timing()))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    g.check()
    # Throughput over a stream of images: the per-image critical path is host work upstream makes serial (greedy merge, one AES chain), so images are pipelined ACROSS
    # contexts -- K host threads, each with its own context and HIP stream: image i+1's pass 1 / similarity kernels and image i's host merge + chain walk overlap.
    pipe = None
    if args.contexts > 1 and rank == 0:
        import threading
        import limg_amd
        ctxs = [limg_amd.LimgHip(torch.cuda.current_device()) for _ in range(args.contexts)]
        for c in ctxs:
            c.set_options(**blocked_opts)
        imgs = [g.synth_device(args.workload, W, H, seed=101 + i) for i in range(args.contexts)]
        outs = [c.alloc_blocked_planes_device(W, H) for c in ctxs]
        streams = [torch.cuda.Stream() for _ in ctxs]

        def worker(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    ctxs[i].blocked_encode3d_device(imgs[i], True, outs[i], error_factor=args.error_factor)
                streams[i].synchronize()

        n_each = max(args.steps, 8)  # images per context in the timed round: the contexts start together, and it takes a few images before their phases have drifted apart
        for phase_n in (1, n_each):  # warm-up round, then the timed one
            ths = [threading.Thread(target=worker, args=(i, phase_n)) for i in range(args.contexts)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n_img = args.contexts * n_each
        pipe = {"contexts": args.contexts, "images": n_img, "images_per_s": round(n_img / dt, 2), "Mpixels_per_s": round(n_img * W * H / dt / 1e6, 1),
                "note": "K host threads x own context x own HIP stream on ONE GPU; every image is a different seed; all planes stay in HBM"}
        for c in ctxs:
            c.check()
            c.close()
        del imgs, outs
    if rank == 0:
        px = W * H
        mean = {k: round(float(np.mean([s[k] for s in stages])), 3) for k in stages[0]}
        nreg = len(g.blocked_regions())
        psnr = g.compare_device(img, planes["pDecoded"], True)[0]
        line = {
            "metric": "merged-block encode Mpixels/s, RGBA (limg_blocked_encode3d_test-equivalent: 13 planes stored)", "value": round(world * px * args.steps / elapsed / 1e6, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 integer stage + f32 float stage (bit-exact vs the reference)", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGBA %s (seed 1+rank) per GPU, errorFactor %d" % (W, H, args.workload, args.error_factor), "rectangles": nreg,
                       "blocks": (W // 8) * (H // 8), "psnr_db": round(psnr, 4), "stage_ms": mean, "pipelined_stream": pipe},
            "roofline": blocked_roofline(px, mean, "blocked_%dx%d_%s%s" % (W, H, args.workload, "" if args.error_factor == 100 else "_ef%d" % args.error_factor)),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle.bind import Oracle, Ref, ref_available
                orc = Oracle()
                n = min(W, 1024)
                crop = orc.photo_noise(n, n, 1) if args.workload == "photo_noise" else orc.random_gradient(n, n, 1, True)
                if ref_available():
                    ref = Ref()
                    t = time.perf_counter(); ref.blocked_encode3d(crop, True, error_factor=args.error_factor); dt = time.perf_counter() - t
                    kind = "reference"
                else:
                    t = time.perf_counter(); orc.blocked_encode3d(crop, True, error_factor=args.error_factor); dt = time.perf_counter() - t
                    kind = "port"
                line["cpu_baseline"] = {"value": round(n * n / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                                        "sample": "%dx%d of the same generator, limg_blocked_encode3d_test, single thread (upstream's merge and region stages are single-threaded)" % (n, n)}
            except Exception as e:
                leg_failed("cpu_baseline", e)
                line["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        emit(line)


def run_stream(args, g, dist, rank, world, W, H):
    """`limg_encode` / `limg_decode` over the compact stream: value = encode-to-stream throughput; the decode kernel (HBM-bound: reads
    the stream, writes 4 B/px) gets the roofline object.  Same barrier / max-over-ranks timing as the headline mode."""
    import torch
    import numpy as np
    img = g.synth_device(args.workload, W, H, seed=1 + rank)
    st, nbytes = g.encode_stream_device(img, True, error_factor=args.error_factor)
    dec = g.decode_stream_device(st, nbytes, W, H)
    planes = g.alloc_planes_device(W, H)
    g.encode3d_device(img, True, planes, error_factor=args.error_factor)
    torch.cuda.synchronize()
    same = bool(torch.equal(dec, planes["pDecoded"]))
    del planes

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        g.profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        k = g.profile_end(args.steps * 2)
        if dist is not None:
            dist.barrier()
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, k

    el_e, k_e = timed(lambda: g.encode_stream_device(img, True, out=st, error_factor=args.error_factor, want_size=False))
    el_d, k_d = timed(lambda: g.decode_stream_device(st, nbytes, W, H, out=dec))
    g.check()
    if rank == 0:
        px = W * H
        enc_ms = float(k_e[0::2].sum(axis=1).mean()) if len(k_e) else 0.0   # k_fit_tpb + k_encode_persistent (compact outputs)
        pack_ms = float(k_e[1::2, 0].mean()) if len(k_e) > 1 else 0.0
        dec_ms = float(k_d[:, 0].mean()) if len(k_d) else 0.0
        dec_bytes = nbytes + 4 * px
        achieved = dec_bytes / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        line = {
            "metric": "encode-to-stream Mpixels/s, RGBA (compact LMG3 stream: limg_encode equivalent)", "value": round(world * px * args.steps / el_e / 1e6, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el_e * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 integer stage + f32 float stage", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGBA %s (seed 1+rank) per GPU, errorFactor %d" % (W, H, args.workload, args.error_factor),
                       "stream_bytes": int(nbytes), "stream_bytes_per_px": round(nbytes / px, 4), "roundtrip_equals_pDecoded": same,
                       "decode_Mpixels_per_s": round(world * px * args.steps / el_d / 1e6, 1), "decode_ms_per_step": round(el_d * 1e3 / args.steps, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": None if not pmc_entry("stream_%dx%d_%s" % (W, H, args.workload)) else int((2 * pmc_entry("stream_%dx%d_%s" % (W, H, args.workload))["fetch_kib"] + pmc_entry("stream_%dx%d_%s" % (W, H, args.workload))["write_kib"]) * 1024),
                         "valu_busy": (pmc_entry("stream_%dx%d_%s" % (W, H, args.workload)) or {}).get("valu_busy"), "pmc_key": "stream_%dx%d_%s" % (W, H, args.workload),
                         "algorithmic_bytes_per_launch": int(dec_bytes),
                         "kernels_ms": {"k_fit_tpb+k_encode_persistent": round(enc_ms, 4), "k_stream_count+scan+pack": round(pack_ms, 4), "k_stream_decode": round(dec_ms, 4)},
                         "note": "roofline object = k_stream_decode: (stream bytes + 4 B/px written) / its average duration"},
        }
        emit(line)


def other_workloads(g, dev, args):
    """One driver-timed entry for every BASELINE config besides the headline (the `--gpus 1` line is the only independent clock this project gets).
    Each: {ms, Mpixels_per_s, frac, verified}; `verified` = every plane the leg wrote equals the REAL reference's (position-sensitive sum64 checksums of
    tests/golden/fullsize.json, computed on the device after the timed loop); a leg that raises or does not verify is a failed leg (non-zero exit).
    frac = the leg's algorithmic bytes / (wall time per step) / 8 TB/s -- wall clock around the enqueue + one synchronize, like `value` (not kernel-only)."""
    import torch
    import limg_amd
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))
    out = {}

    def wall(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    def entry(ms, px, bytes_per_px, verified, **more):
        e = {"ms": round(ms, 4), "Mpixels_per_s": round(px / ms / 1e3, 1), "frac": round(bytes_per_px * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "verified": bool(verified)}
        e.update(more)
        if not verified:
            raise RuntimeError("planes differ from the real reference's checksums: %s" % (more.get("mismatch"),))
        return e

    def check(planes, want, names):
        return [k for k in names if sum64_device(planes[k]) != want[k]]

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = leg_failed("other_workloads." + name, e)
        torch.cuda.empty_cache()

    def config2():  # BASELINE configs[1]: one 4096^2 random-gradient image
        e = gold["rg4096_batch64"]["images"][0]
        img = g.synth_device("random_gradient", 4096, 4096, seed=1)
        planes = g.alloc_planes_device(4096, 4096)
        ms = wall(lambda: g.encode3d_device(img, True, planes), 3, 20)
        g.check()
        bad = check(planes, e["sum64"], limg_amd.PLANES)
        return entry(ms, 4096 * 4096, ALGO_BYTES_PER_PX, not bad, mismatch=bad, workload="4096x4096 RGBA random_gradient seed 1, one image per call", golden="rg4096_batch64[0]")

    def config4():  # BASELINE configs[3] as one GPU holds it: all 64 images through ONE limg_hip_encode3d_batch_device call (the sub-batch pipeline)
        e = gold["rg4096_batch64"]
        imgs = [g.synth_device("random_gradient", 4096, 4096, seed=im["seed"]) for im in e["images"]]
        outs = [g.alloc_planes_device(4096, 4096) for _ in imgs]
        ms = wall(lambda: g.encode3d_batch_device(imgs, True, outs), 1, 3)
        g.check()
        bad = [(im["seed"], k) for im, pl in zip(e["images"], outs) for k in check(pl, im["sum64"], limg_amd.PLANES)]
        return entry(ms, len(imgs) * 4096 * 4096, ALGO_BYTES_PER_PX, not bad, mismatch=bad[:6], workload="64 x 4096x4096 RGBA random_gradient seeds 1..64, one batched call",
                     golden="rg4096_batch64 (all 64 images x 11 planes)")

    def config5(pool):  # BASELINE configs[4] as one of 8 GPUs sees it: its 16384 x 2048 strip (rows 0..2047), one chain / the reference's pool of 2 = 8 chains
        name = "pn16384x2048" + ("_pool2" if pool else "")
        e = gold[name]
        img = g.synth_device("photo_noise", 16384, 2048, seed=1)
        planes = g.alloc_planes_device(16384, 2048)
        ms = wall(lambda: g.encode3d_device(img, True, planes, pool_threads=pool), 3, 20)
        g.check()
        bad = check(planes, e["sum64"], limg_amd.PLANES)
        return entry(ms, 16384 * 2048, ALGO_BYTES_PER_PX, not bad, mismatch=bad, workload="16384x2048 RGBA photo_noise seed 1 (strip 0 of the 16384^2 image), poolThreads %d" % pool, golden=name)

    def stream():  # limg_encode / limg_decode: the headline image to the compact stream and back
        e = gold["pn8192"]
        img = g.synth_device("photo_noise", 8192, 8192, seed=1)
        st, nbytes = g.encode_stream_device(img, True)
        dec = g.decode_stream_device(st, nbytes, 8192, 8192)
        enc_ms = wall(lambda: g.encode_stream_device(img, True, out=st, want_size=False), 2, 10)
        g.profile_begin()
        g.encode_stream_device(img, True, out=st, want_size=False)
        torch.cuda.synchronize()
        k = g.profile_end(2)
        dec_ms = wall(lambda: g.decode_stream_device(st, nbytes, 8192, 8192, out=dec), 2, 20)
        g.check()
        bad = check({"pDecoded": dec}, e["sum64"], ("pDecoded",))
        px = 8192 * 8192
        return entry(enc_ms, px, 4 + nbytes / px, not bad, mismatch=bad, workload="8192x8192 RGBA photo_noise seed 1 -> LMG3 stream -> pDecoded", golden="pn8192.pDecoded (decode(encode) against the reference)",
                     stream_bytes=int(nbytes), pack_ms=None if len(k) < 2 else round(float(k[1, 0]), 4), decode_ms=round(dec_ms, 4), decode_Mpixels_per_s=round(px / dec_ms / 1e3, 1),
                     decode_frac=round((nbytes + 4 * px) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     note="ms / Mpixels_per_s / frac = encode to the stream (k_fit_tpb + persistent kernel with compact outputs + the packer; 4 B/px read + the stream written); decode_* = k_stream_decode")

    def blocked():  # limg_blocked_encode3d_test: what the reference's CLI runs on a single file (src/main.cpp:255)
        e = gold["blocked_pn8192"]
        img = g.synth_device("photo_noise", 8192, 8192, seed=1)
        planes = g.alloc_blocked_planes_device(8192, 8192)
        ms = wall(lambda: g.blocked_encode3d_device(img, True, planes), 1, 4)
        g.check()
        names = [k for k, _ in limg_amd.BLOCKED_PLANES if k != "pBlockError"]
        bad = check(planes, e["sum64"], names)
        if len(g.blocked_regions()) != e["regions"]:
            bad.append("rectangle count")
        return entry(ms, 8192 * 8192, BLOCKED_BYTES_PER_PIXEL, not bad, mismatch=bad, workload="8192x8192 RGBA photo_noise seed 1, merged-block encoder, host stages included", golden="blocked_pn8192 (13 planes + rectangle count)",
                     rectangles=e["regions"])

    t0 = time.perf_counter()
    leg("config2_rg4096", config2)
    leg("config4_batch64_rg4096", config4)
    leg("config5_strip_pool0", lambda: config5(0))
    leg("config5_strip_pool2", lambda: config5(2))
    leg("stream_pn8192", stream)
    leg("blocked_pn8192", blocked)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def limg_planes():
    import limg_amd
    return limg_amd.PLANES


def spawn_ranks(args):
    """`--gpus N` without a launcher: start the N ranks ourselves.  The parent never initialises a GPU (device_count only reads the topology),
    the children are fresh processes of this file with the usual torch.distributed environment; rank 0's stdout (the JSON line) passes through."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    ndev = torch.cuda.device_count()
    if ndev < n and not args.share_gpus:
        print("bench.py: --gpus %d asked but this node has %d GPU(s): refusing to run (a line with a smaller n_gpus would be a different measurement). "
              "For a logic rehearsal with several ranks on one card over gloo pass --share-gpus." % (n, ndev), file=sys.stderr, flush=True)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", default="8192", help="N (an N x N image) or WxH, e.g. 8192x8190: sizes that are not multiples of 8 take the ragged paths (config.ragged)")
    ap.add_argument("--workload", default="photo_noise", choices=["photo_noise", "random_gradient"])
    ap.add_argument("--error-factor", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-workloads", action="store_true", help="add config.other_workloads (one verified entry per BASELINE config) to a non-default line too")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the PCIe-inclusive timing of the host-pointer entry (config.host_entry)")
    ap.add_argument("--split", action="store_true", help="three-launch path instead of the fused kernel")
    ap.add_argument("--compact", action="store_true", help="compact mode: factor planes + records + shift words only (8.05 B/px)")
    ap.add_argument("--float-mode", default="exact", choices=["exact", "fast"],
                    help="exact (headline): the float stage op for op as the reference's strict SSE build; fast: native rsq / fused multiply-adds, PSNR-tolerance contract")
    ap.add_argument("--legacy-float-stage", action="store_true", help="float stage inside the E step with lane == pixel (round-1 mapping) instead of k_fit_tpb")
    ap.add_argument("--accurate", action="store_true", help="accurate bit-crush search (fastBitCrushing = false, src/limg_bit_crush.h:668-830) instead of the default guess + stepwise search")
    ap.add_argument("--rgb", action="store_true", help="encode as 3-channel (hasAlpha = false)")
    ap.add_argument("--forced-shift", type=int, default=-1, help="bypass the shift search with this shift on all three factors (bit-crush sweep)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="BASELINE.json configs, 1-based: 3 = headline (default), 4 = batch of 64 x 4096^2 images over the ranks + gather, "
                         "5 = one 16384^2 image as 8 reference strips over the ranks + gather")
    ap.add_argument("--stream", action="store_true", help="compact LMG3 stream instead of the planes: encode + pack, then decode (SURVEY 8(f) #2)")
    ap.add_argument("--blocked", action="store_true", help="merged-block encoder limg_blocked_encode3d_test (SURVEY 8(f) #1): GPU kernels + host merge / chain walk")
    ap.add_argument("--images", type=int, default=64, help="--config 4: images in the batch")
    ap.add_argument("--no-match-bound", action="store_true", help="--blocked: limg_hip_options.test_blocked_no_bound (A/B: the similarity kernel without its certain-match bound)")
    ap.add_argument("--no-vec-store", action="store_true", help="--blocked A/B (LIMG_HIP_LIB=test): k_blocked_store with one pixel per lane instead of four")
    ap.add_argument("--no-order", action="store_true", help="--blocked A/B (LIMG_HIP_LIB=test): the per-rectangle launches take the rectangles in creation order instead of large-first (k_blocked_order)")
    ap.add_argument("--contexts", type=int, default=1, help="--blocked: also time a stream of images pipelined over this many contexts / host threads on the one GPU; "
                    "--config 4: spread the rank's images round-robin over this many contexts / HIP streams")
    ap.add_argument("--no-batch", action="store_true", help="--config 4: one launch pair per image (limg_hip_encode3d_device in a loop) instead of the batched entry")
    ap.add_argument("--no-gather", action="store_true", help="--config 4/5: skip the reassembly of the planes on rank 0")
    ap.add_argument("--gather-stream", action="store_true", help="--config 5: reassemble through the compact LMG3 stream instead of the planes: every rank encodes its "
                                                                     "strips to streams, rank 0 gathers the bytes and decodes them into the full image")
    ap.add_argument("--single-chain", action="store_true", help="--config 5 on 8 ranks: one dither chain through all strips (limg_hip_encode3d_single_chain_device) instead of "
                                                                    "the reference's strip-restart semantics")
    ap.add_argument("--verify-golden", action="store_true", help="--config 4 / 5 at their real size: every rank checks the images / strips it produced against the real reference's per-image / per-strip "
                                                                     "checksums (tests/golden/fullsize.json)")
    ap.add_argument("--share-gpus", action="store_true", help="rehearsal only: allow more ranks than GPUs (ranks share cards, gloo instead of RCCL)")
    ap.add_argument("--sub-images", type=int, default=0, help="--config 4: limg_hip_options.batch_sub_images -- the list as a pipeline of sub-batches of this many images "
                                                                 "(float stage of sub-batch k + 1 next to the persistent kernel of sub-batch k); 0 = the library's rule, -1 = off")
    ap.add_argument("--graph", action="store_true", help="capture one encode into a HIP graph after the warm-up and time replays (launch-bound small images); kernel intervals "
                                                      "are not available inside a graph: roofline.achieved then divides by the wall time per replay")
    ap.add_argument("--pipeline-knobs", type=lambda v: int(v, 0), default=0, help="A/B: limg_hip_options.test_pipeline")
    ap.add_argument("--wg-per-cu", type=int, default=0, help="A/B: limg_hip_options.test_wg_per_cu (workgroups per CU of the persistent kernel, 1..6)")
    ap.add_argument("--pool-threads", type=int, default=0, help="default mode: the reference's thread-pool argument (0 = nullptr = one dither chain; T > 0 = 4 T row strips with restarted chains)")
    ap.add_argument("--ragged-bands", type=int, default=0, help="limg_hip_options.ragged_bands (images with a partial last block column: bands of the host walk's pipeline; -1 = off)")
    ap.add_argument("--walk-threads", type=int, default=0, help="limg_hip_options.ragged_walk_threads (several chains: host threads that walk them; 1 = serial)")
    ap.add_argument("--whole-image-ragged", action="store_true", help="A/B: height-ragged images through the whole-image ragged path (host walk over every dither call)")
    args = ap.parse_args()
    args.width, args.height = parse_size(args.size)
    args.size = args.width  # (the square-size uses below: configs 4 / 5, cpu_baseline)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    if not os.environ.get("LIMG_KEEP_NCCL_DEBUG"):
        os.environ["NCCL_DEBUG"] = "WARN"  # RCCL's version banner goes to stdout, where the one JSON line belongs
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")  # ... and so do its warnings (seen: "Missing iommu=pt" on a GPU box, which broke the line's parse)
    import torch
    import numpy as np
    import limg_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to run a different job than the one asked for" % (args.gpus, world), file=sys.stderr, flush=True)
        sys.exit(2)
    dist = None
    dev = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        if ndev >= world:
            dev = local_rank
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))  # RCCL over xGMI
        elif args.share_gpus:
            # rehearsal on a box with fewer GPUs than ranks (ranks share cards): RCCL refuses duplicate devices, use gloo
            dev = local_rank % max(ndev, 1)
            torch.cuda.set_device(dev)
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to 1 peer ranks"), where the one JSON line belongs: while it connects, fd 1 is fd 2
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo")
                dist.barrier()  # (the connections are made lazily: by the first collective)
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            if rank == 0:
                print("bench.py: WORLD_SIZE %d but only %d GPU(s) visible: refusing (pass --share-gpus for a gloo rehearsal)" % (world, ndev), file=sys.stderr, flush=True)
            sys.exit(2)
    else:
        with known_driver_noise_filtered():
            torch.cuda.set_device(0)
            torch.cuda.synchronize()
    n_gpus = world

    W, H = args.width, args.height
    with known_driver_noise_filtered():
        g = limg_amd.LimgHip(dev)
        torch.cuda.synchronize()
    if args.blocked:
        run_blocked(args, g, dist, rank, n_gpus, W, H)
        g.close()
        if dist is not None:
            if not COLLECTIVES_OFF:
                dist.destroy_process_group()
        return
    if args.stream:
        run_stream(args, g, dist, rank, n_gpus, W, H)
        g.close()
        if dist is not None:
            if not COLLECTIVES_OFF:
                dist.destroy_process_group()
        return
    if args.config != 3:
        run_sharded(args, g, dist, rank, n_gpus)
        g.close()
        if dist is not None:
            if not COLLECTIVES_OFF:
                dist.destroy_process_group()
        return
    g.set_options(forced_shift=(args.forced_shift,) * 3 if args.forced_shift >= 0 else None, force_split=args.split, float_fast=(args.float_mode == "fast"), legacy_float_stage=args.legacy_float_stage,
                  test_wg_per_cu=args.wg_per_cu, test_whole_image_ragged=args.whole_image_ragged, ragged_bands=args.ragged_bands, ragged_walk_threads=args.walk_threads)
    ragged = (W % 8 != 0) or (H % 8 != 0)
    ragged_fast = ragged and W % 8 == 0 and H > 8 and not (args.whole_image_ragged or args.split or args.legacy_float_stage)
    img = g.synth_device(args.workload, W, H, seed=1 + rank)
    planes = g.alloc_planes_device(W, H)
    rec = sh = None
    if args.compact:
        planes = {k: planes[k] for k in limg_amd.P8 + ("pDecoded",)}  # pDecoded only as a scratch target for the PSNR line below
        full = planes.pop("pDecoded")
        rec = torch.empty((((W + 7) // 8) * ((H + 7) // 8), 16), dtype=torch.int32, device="cuda")
        sh = torch.empty(((W + 7) // 8) * ((H + 7) // 8), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step():
        g.encode3d_device(img, not args.rgb, planes, error_factor=args.error_factor, pool_threads=args.pool_threads, fast=not args.accurate, records=rec, shifts=sh)

    # cold cost, reported apart: the first encode of a size class builds the context's dither noise table and scratch
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_encode_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        if ragged:
            raise SystemExit("--graph: images with partial edge blocks take a host step inside the call (not capturable)")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(args.warmup):
            graph.replay()
        torch.cuda.synchronize()
    g.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernels = g.profile_end(args.steps)
    if graph is not None:  # one "interval" = the whole replay
        kernels = np.array([[elapsed * 1e3 / args.steps, 0.0, 0.0]], dtype=np.float32)
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize()
    g.check()  # a look-back timeout inside the timed loop would void the line
    collective = collective_evidence(g, dist, rank, n_gpus)

    # cold cost of a SECOND, smaller size class on the warm context (the noise table is a prefix stream and the scratch only grows: nothing is rebuilt), and of a
    # fresh context whose table is built on the host the way rounds 1-2 did (limg_hip_options.host_noise_table), for comparison
    cold = {}
    if rank == 0 and n_gpus == 1 and not args.no_host_rate and W >= 4096 and W == H:
        w2 = W // 2
        img2 = g.synth_device(args.workload, w2, w2, seed=5)
        planes2 = g.alloc_planes_device(w2, w2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.encode3d_device(img2, not args.rgb, planes2, error_factor=args.error_factor, pool_threads=0, fast=not args.accurate)
        torch.cuda.synchronize()
        cold["second_smaller_size_class_first_encode_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        t0 = time.perf_counter()
        g.encode3d_device(img2, not args.rgb, planes2, error_factor=args.error_factor, pool_threads=0, fast=not args.accurate)
        torch.cuda.synchronize()
        cold["second_smaller_size_class_warm_encode_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        del img2, planes2
        gh = limg_amd.LimgHip(dev)
        gh.set_options(host_noise_table=True)
        t0 = time.perf_counter()
        gh.encode3d_device(img, not args.rgb, planes, error_factor=args.error_factor, pool_threads=0, fast=not args.accurate)
        torch.cuda.synchronize()
        cold["first_encode_ms_with_host_built_noise_table"] = round((time.perf_counter() - t0) * 1e3, 2)
        gh.close()

    px = W * H
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * px * args.steps / elapsed / 1e6
    psnr = float("nan") if args.compact else g.compare_device(img, planes["pDecoded"], not args.rgb)[0]
    bytes_per_px = (4 + 3 + 68.0 / 64) if args.compact else ALGO_BYTES_PER_PX

    if rank == 0:
        # `_perf` style on the GPU (SURVEY 8(d) last row; src/limg.cpp:2140-2173): the E step alone -- fit, factors, shift search -- nothing stored
        g.profile_begin()
        for _ in range(5):
            g.encode3d_device(img, not args.rgb, None, error_factor=args.error_factor, pool_threads=0, fast=not args.accurate)
        torch.cuda.synchronize()
        kperf = g.profile_end(5)
        perf_ms = float(kperf[1:, 0].mean()) if len(kperf) > 1 else None

        kavg = kernels.mean(axis=0) if len(kernels) else np.zeros(3)
        # the kernels listed in roofline.kernels_ms, nothing else: fused = k_fit_tpb + k_encode_persistent (the one persistent launch with --legacy-float-stage); split = the three intervals
        kms = float(kavg.sum()) if (args.split or ragged) else (float(kavg[0]) if args.legacy_float_stage else float(kavg[0] + kavg[1]))
        achieved = bytes_per_px * px / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        pmc = pmc_entry(workload_key(args, W, H))
        traffic = None if not pmc or pmc.get("fetch_kib") is None else int((2 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024)  # gfx950: FETCH_SIZE counts half the bytes
        blocks = ((W + 7) // 8) * ((H + 7) // 8)
        valu = None
        if pmc and pmc.get("valu_instr_per_launch") and kms > 0:
            rate = pmc["valu_instr_per_launch"] / (kms * 1e-3)
            valu = {"instr_per_block": round(pmc["valu_instr_per_launch"] / blocks, 1), "issued_per_s": round(rate / 1e9, 1), "unit": "G wave64 instr/s",
                    "issue_peak_half_rate_class": VALU_HALF_RATE_PER_S / 1e9, "issue_peak_full_rate_class": VALU_FULL_RATE_PER_S / 1e9,
                    "frac": round(rate / VALU_HALF_RATE_PER_S, 4), "valu_busy": pmc.get("valu_busy"), "source": pmc.get("source"),
                    "note": "frac = issued wave64 VALU instructions per second / the measured chip-wide rate of the half-rate instruction class (v_mad_i32_i24, v_pk_*, "
                            "VOP3-only, DPP, v_cvt_*: profiles/archive/r02_valu_ceiling.md); the kernel's mix holds some full-rate f32 adds, so frac can approach but not pass "
                            "the full-rate peak"}
        line = {
            "metric": "encode Mpixels/s, 8K RGBA (limg_encode3d_test-equivalent: all 11 planes stored)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("u8/i32 integer stage + f32 float stage (bit-exact vs the reference's strict SSE build)" if args.float_mode == "exact" else
                      "u8/i32 integer stage (bit-exact given the records) + f32 float stage in FAST mode (PSNR-tolerance contract)"), "data": "synthetic",
            "config": {"workload": "synthetic %dx%d %s %s (seed 1+rank) per GPU, errorFactor %d, %s bit-crush, single dither chain"
                                   % (W, H, "RGB (hasAlpha = false)" if args.rgb else "RGBA", args.workload, args.error_factor, "ACCURATE" if args.accurate else "fast")
                                   + ("" if args.pool_threads == 0 else " -- NO: pool of %d threads = %d restarted chains" % (args.pool_threads, 4 * args.pool_threads))
                                   + ("" if args.forced_shift < 0 else ", forced shift %d" % args.forced_shift)
                                   + (", COMPACT outputs (8.06 B/px)" if args.compact else "") + (", FAST float stage" if args.float_mode == "fast" else ""),
                       "ragged": None if not ragged else ("width in whole blocks, last block row partial: fast path + last row (limg_hip_api.hip encode_height_ragged)" if ragged_fast else
                                                          "whole-image ragged path: lane == pixel float stage, three launches, host chain walk over every dither call"),
                       "collective": collective,
                       "images_per_step": n_gpus, "parallelism": "independent image per GPU, no data-path collective", "psnr_db": None if psnr != psnr else round(psnr, 4),
                       "first_encode_ms": round(first_encode_ms, 2), "cold": cold,
                       "perf_style_ms": None if perf_ms is None else round(perf_ms, 4),
                       "perf_style_Mpixels_per_s": None if not perf_ms else round(px / perf_ms / 1e3, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_per_px * px),
                         "kernels_ms": ({"k_fit_tpb (block rows above the last)": round(float(kavg[0]), 4), "k_encode_persistent (block rows above the last)": round(float(kavg[1]), 4),
                                         "last block row: k_fit_search + shift words D2H + host chain walk + noise H2D + k_dither_store": round(float(kavg[2]), 4)} if ragged_fast else
                                        {"k_fit_search (lane == pixel float stage)": round(float(kavg[0]), 4), "shift words D2H + host chain walk over every dither call + noise H2D": round(float(kavg[1]), 4),
                                         "k_dither_store": round(float(kavg[2]), 4)} if ragged else
                                        {"k_fit_search": round(float(kavg[0]), 4), "k_strip_scan": round(float(kavg[1]), 4), "k_dither_store": round(float(kavg[2]), 4)}
                                        if args.split else {"HIP graph replay of one encode (k_fit_tpb + k_encode_persistent), wall": round(float(kavg[0]), 4)} if args.graph else
                                        ({"k_encode_persistent": round(float(kavg[0]), 4)} if args.legacy_float_stage else
                                                            {"k_fit_tpb": round(float(kavg[0]), 4), "k_encode_persistent": round(float(kavg[1]), 4)})),
                         "valu": valu, "pmc_key": workload_key(args, W, H), "pmc_refused_stale_source": pmc_stale_source(workload_key(args, W, H)),
                         "instruction_floor": instruction_floor(pmc, bytes_per_px * px),
                         "note": ("image with partial edge blocks: achieved = 39 B/px * pixels / the sum of the three intervals, host stage included (HIP events on the launch stream)" if ragged else
                                  "whole encode = 3 launches; achieved = 39 B/px * pixels / sum of the three average kernel durations (HIP events)" if args.split else
                                  "whole encode = k_fit_tpb (float stage, one lane per block) + one persistent launch; achieved = 39 B/px * pixels / the sum of their average durations "
                                  "(HIP events on the launch stream). "
                                  "The kernel is VALU-issue-bound, not HBM-bound: see `valu`")},
        }
        if n_gpus == 1 and not args.no_host_rate and not (args.split or args.compact or args.forced_shift >= 0 or ragged):
            try:  # what a service encoding a stream of images gets: two contexts on two HIP streams, images alternating (never `value`: the kernels overlap)
                line["config"]["two_streams"] = two_stream_rate(g, img, planes, W, H, args)
            except Exception as e:
                line["config"]["two_streams"] = leg_failed("two_streams", e)
        if n_gpus == 1 and args.contexts > 1:
            try:  # K host threads x own context x own stream: what a service gets out of one GPU for this image class (the ragged paths block their calling thread)
                line["config"]["multi_context"] = multi_context_rate(img, W, H, args, args.contexts)
            except Exception as e:
                line["config"]["multi_context"] = leg_failed("multi_context", e)
        if n_gpus == 1 and not args.no_host_rate and not (args.split or args.compact or ragged):
            try:  # what a per-request service pays that makes a context per image: init + first encode (noise table, scratch) + shutdown
                line["config"]["cold_context_per_image"] = cold_context_rate(dev, img, planes, W, H, args)
            except Exception as e:
                line["config"]["cold_context_per_image"] = leg_failed("cold_context_per_image", e)
        if n_gpus == 1 and not args.no_host_rate:
            try:
                line["config"]["host_entry"] = host_entry_rate(g, W, H, args)
            except Exception as e:
                line["config"]["host_entry"] = leg_failed("host_entry", e)
        default_line = (n_gpus == 1 and not args.no_host_rate and (W, H) == (8192, 8192) and args.workload == "photo_noise" and args.error_factor == 100 and args.pool_threads == 0 and
                        args.float_mode == "exact" and not (args.split or args.compact or args.accurate or args.rgb or args.graph or args.legacy_float_stage or args.forced_shift >= 0))
        if default_line or args.other_workloads:
            # the headline image itself against the real reference, then every other BASELINE config once (never part of `value`)
            try:
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["pn8192"]
                bad = [k for k in limg_amd.PLANES if sum64_device(planes[k]) != gold["sum64"][k]] if default_line else None
                line["config"]["verified"] = None if bad is None else not bad
                if bad:
                    raise RuntimeError("the headline encode's planes differ from the real reference's checksums: %s" % bad)
            except Exception as e:
                line["config"]["verified"] = False
                leg_failed("verify_headline", e)
            del planes, img
            torch.cuda.empty_cache()
            line["config"]["other_workloads"] = other_workloads(g, dev, args)
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(W, 1, height=H)
            except Exception as e:  # the checker does not void the measurement, but the line says so and the process fails
                leg_failed("cpu_baseline", e)
                line["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        emit(line)
    g.close()
    if dist is not None:
        if not COLLECTIVES_OFF:
            dist.destroy_process_group()


COLLECTIVES_OFF = False  # set when a rank's evidence helper hung inside RCCL: from then on nothing may issue a collective on the process group


def exchange_views(dist, rank, world, mine, timeout_s):
    """all-gather of a small list over the process group's STORE (no RCCL): [view of rank 0, ...]; a rank that does not answer in time is reported as hung"""
    import datetime
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:  # (stand-ins of tests/test_bench_helpers.py: a world of one)
        store = None
    if store is None:
        return [list(mine) for _ in range(world)]
    EXCHANGES.append(1)
    tag = "limg_evidence_%d_" % len(EXCHANGES)
    store.set(tag + str(rank), json.dumps(mine))
    views = []
    for r in range(world):
        try:
            store.wait([tag + str(r)], datetime.timedelta(seconds=timeout_s))
            views.append(json.loads(store.get(tag + str(r))))
        except Exception:
            views.append([-1, r, -1, 1])
    return views


EXCHANGES = []


def collective_evidence(g, dist, rank, world, timeout_s=60.0):
    """--gpus N > 1: what RCCL itself says about the job, so that a SCALE record shows the N ranks RCCL saw: torch.distributed's backend and world size, and -- through
    the library's own communicator (limg_hip_comm_init over the id rank 0 made; ncclCommCount / ncclGetVersion behind limg_hip_comm_info) -- every rank's view, all-gathered:
    the line is refused unless all ranks report the same `comm_ranks` == N.  Outside the timed region."""
    if dist is None:
        return None
    import torch
    ev = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": None, "comm_ranks": None}
    if dist.get_backend() != "nccl":
        ev["note"] = "gloo rehearsal: no RCCL communicator"
        return ev
    # On a helper thread with a time limit: this is the first place the library's OWN RCCL communicator spans several GPUs (the builder's boxes have one), and evidence
    # must neither hang nor fail the measurement it decorates -- a failure here is a `warning` on the line, the timed numbers stand.
    import threading
    res = {}
    have_gpu = torch.cuda.is_available()  # (false only under tests/test_bench_helpers.py, which drives this function with stand-ins)
    cur_dev = torch.cuda.current_device() if have_gpu else None  # the current device is per thread: the helper must select this rank's GPU itself, or its tensors land on device 0

    def work():
        try:
            if have_gpu:
                torch.cuda.set_device(cur_dev)
            g.comm_init_from_torch(dist)
            res["info"] = g.comm_info()
            g.comm_destroy()
        except Exception as e:  # noqa: BLE001
            res["error"] = repr(e)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout=timeout_s)
    if th.is_alive():
        res["error"] = "the library's communicator did not come up within %.0f s" % timeout_s
        HUNG_THREAD.append(th)
    info = res.get("info") or {"ranks": -1, "rank": rank, "rccl_version": -1}
    if "error" in res:
        ev["error"] = res["error"]
        leg_warned("collective_evidence", res["error"])
    # Every rank's view, WITHOUT another collective on the process group: after a time-out a helper thread may still sit inside ncclCommInitRank / the broadcast, and a
    # collective issued beside it -- by this rank or by the ranks whose helper came back -- can hang the job the time limit was meant to protect.  The
    # views travel through the process group's store (TCP); a rank that cannot be heard from within the limit counts as hung.
    mine = [info["ranks"], info["rank"], info["rccl_version"], 1 if th.is_alive() else 0]
    try:
        views = exchange_views(dist, rank, world, mine, timeout_s)
    except Exception as e:  # noqa: BLE001  (evidence must never sink the measurement)
        ev["error"] = "views could not be exchanged over the store: %r" % (e,)
        leg_warned("collective_evidence", ev["error"])
        views = [list(mine) if r == rank else [-1, r, -1, 1] for r in range(world)]
    if any(v[3] for v in views):
        global COLLECTIVES_OFF
        COLLECTIVES_OFF = True  # no RCCL call of this job is safe any more: the callers skip the gather, the process leaves without destroy_process_group
        ev["hung_ranks"] = [i for i, v in enumerate(views) if v[3]]
    if any(v[0] < 0 or v[3] for v in views):
        ev["comm_views"] = views  # (a rank could not create or query the library's communicator: reported, not fatal)
        return ev
    if any(v[0] != world for v in views) or sorted(v[1] for v in views) != list(range(world)):
        raise SystemExit("bench.py: RCCL communicator does not span the %d ranks asked for: %r" % (world, views))
    ev.update({"rccl_version": info["rccl_version"], "comm_ranks": info["ranks"], "comm_user_ranks": [v[1] for v in views]})
    return ev


def two_stream_rate(g, img, planes, W, H, args, n_images=12):
    """Throughput of a stream of images over two contexts, each on a HIP stream of its own: the second image's float-stage kernel runs on the CUs the first image's
    persistent kernel leaves idle while its last strips drain (and the other way round).  Same image and options as the timed loop; its own planes for the second context."""
    import torch
    import limg_amd
    g2 = limg_amd.LimgHip(torch.cuda.current_device())
    g2.set_options(float_fast=(args.float_mode == "fast"), legacy_float_stage=args.legacy_float_stage)
    planes2 = g2.alloc_planes_device(W, H)
    s2 = torch.cuda.Stream()
    pairs = [(g, planes, torch.cuda.current_stream()), (g2, planes2, s2)]

    def run(n):
        for i in range(n):
            c, pl, st = pairs[i & 1]
            with torch.cuda.stream(st):
                c.encode3d_device(img, not args.rgb, pl, error_factor=args.error_factor, pool_threads=0, fast=not args.accurate)
    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n_images)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.check(); g2.check()
    same = all(bool(torch.equal(planes[k], planes2[k])) for k in planes)  # every plane of the two contexts' last encodes
    g2.close()
    if not same:
        raise RuntimeError("two contexts on two streams produced different planes for the same image")
    return {"contexts": 2, "images": n_images, "ms_per_image": round(dt * 1e3 / n_images, 4), "Mpixels_per_s": round(n_images * W * H / dt / 1e6, 1), "outputs_identical": same}


def multi_context_rate(img, W, H, args, K, n_each=6):
    """Throughput of a stream of images over K contexts, each driven by its own host thread on its own HIP stream (ctypes releases the GIL inside the library).  Same
    image and options as the timed loop.  Every context's last planes must equal context 0's."""
    import threading
    import torch
    import limg_amd
    dev = torch.cuda.current_device()
    ctxs = [limg_amd.LimgHip(dev) for _ in range(K)]
    for c in ctxs:
        c.set_options(float_fast=(args.float_mode == "fast"), legacy_float_stage=args.legacy_float_stage, ragged_bands=args.ragged_bands, ragged_walk_threads=args.walk_threads)
    outs = [c.alloc_planes_device(W, H) for c in ctxs]
    streams = [torch.cuda.Stream() for _ in ctxs]
    errs = []

    def worker(i, n):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    ctxs[i].encode3d_device(img, not args.rgb, outs[i], error_factor=args.error_factor, pool_threads=args.pool_threads, fast=not args.accurate)
                streams[i].synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    dt = None
    for n in (1, n_each):  # warm-up round, then the timed one
        ths = [threading.Thread(target=worker, args=(i, n)) for i in range(K)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for c in ctxs:
        c.check()
    same = all(bool(torch.equal(outs[0][k], o[k])) for o in outs[1:] for k in outs[0])
    for c in ctxs:
        c.close()
    if errs or not same:
        raise RuntimeError("multi-context run failed: %r, outputs identical: %s" % (errs, same))
    n_img = K * n_each
    return {"contexts": K, "images": n_img, "ms_per_image": round(dt * 1e3 / n_img, 4), "Mpixels_per_s": round(n_img * W * H / dt / 1e6, 1), "outputs_identical": same}


def cold_context_rate(dev, img, planes, W, H, args, n=4):
    """limg_hip_init + ONE encode + limg_hip_shutdown per image, device-resident planes: the whole life of a context per request"""
    import torch
    import limg_amd
    ts, nbytes = [], 0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = limg_amd.LimgHip(dev)
        c.set_options(float_fast=(args.float_mode == "fast"))
        c.encode3d_device(img, not args.rgb, planes, error_factor=args.error_factor, pool_threads=args.pool_threads, fast=not args.accurate)
        torch.cuda.synchronize()
        c.check()
        nbytes = c.device_bytes()
        c.close()
        ts.append((time.perf_counter() - t0) * 1e3)
    return {"ms_per_image": round(min(ts), 3), "ms_all": [round(t, 2) for t in ts], "Mpixels_per_s": round(W * H / min(ts) / 1e3, 1), "context_device_bytes": int(nbytes),
            "note": "init + first encode (dither noise table filled on the GPU, scratch allocated) + shutdown, per image; a warm context does the same encode in ms_per_step"}


def host_entry_rate(g, W, H, args):
    """PCIe-inclusive rate of the drop-in entry itself (`limg_hip_encode3d`, host pointers in and out: what the shim's limg_encode3d_test calls).
    Never the headline `value`: reported in config."""
    import numpy as np
    import limg_amd
    import ctypes as C
    host = g.synth_device(args.workload, W, H, seed=1).cpu().numpy().view(np.uint32)
    out = {k: np.empty((H, W), dtype=np.uint32 if k in limg_amd.P32 else np.uint8) for k in limg_amd.PLANES}
    for v in out.values():
        v.fill(0)  # touch the pages: the caller's allocation cost is not the library's
    info = limg_amd.Info(*[out[k].ctypes.data for k in limg_amd.PLANES])
    times = []
    for _ in range(3):
        t = time.perf_counter()
        r = g.lib.limg_hip_encode3d(g.ctx, host.ctypes.data_as(C.c_void_p), W, H, 1, C.byref(info), args.error_factor, 0, 1)
        times.append(time.perf_counter() - t)
        if r != 0:
            raise RuntimeError("limg_hip_encode3d -> %d" % r)
    t = min(times[1:])
    perf = []
    for _ in range(4):  # limg_encode3d_test_perf's counterpart (what the reference's tool calls in its --count loop, src/main.cpp:278-323): 4 B/px up, nothing down
        tp = time.perf_counter()
        r = g.lib.limg_hip_encode3d_perf(g.ctx, host.ctypes.data_as(C.c_void_p), W, H, 1, args.error_factor, 0, 1)
        perf.append(time.perf_counter() - tp)
        if r != 0:
            raise RuntimeError("limg_hip_encode3d_perf -> %d" % r)
    tp = min(perf[1:])
    return {"entry": "limg_hip_encode3d (host pointers; pageable caller memory)", "ms": round(t * 1e3, 2), "Mpixels_per_s": round(W * H / t / 1e6, 1),
            "bytes_over_pcie": W * H * 39, "GB_per_s": round(W * H * 39 / t / 1e9, 2),
            "perf_entry": {"entry": "limg_hip_encode3d_perf (host pointer in, nothing stored)", "ms": round(tp * 1e3, 2), "Mpixels_per_s": round(W * H / tp / 1e6, 1),
                           "bytes_over_pcie": W * H * 4, "GB_per_s": round(W * H * 4 / tp / 1e9, 2)}}


if __name__ == "__main__":
    main()
    if HUNG_THREAD or COLLECTIVES_OFF:  # (a helper thread of this or another rank is stuck inside RCCL: do not wait for it, or tear the group down, at exit)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3 if ERRORS else 0)
    if ERRORS:
        sys.exit(3)
