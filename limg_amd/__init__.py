"""limg_amd -- MI355X (gfx950) implementation of limg's encode hot path.

This package is a thin ctypes view of the C ABI in include/limg_hip.h (liblimg_hip.so, built in-tree by limg_amd/build.py
from the hand-written HIP sources in limg_amd/csrc).  It is plumbing for the tests and the bench: the product is the
shared library.  There is NO CPU fallback: if the library is missing, or no HIP device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TEST_LIB_PATH = os.path.join(HERE, "liblimg_hip_test.so")  # the -DLIMG_HIP_TEST_HOOKS build (include/limg_hip_test_hooks.h): what tests/conftest.py selects
LIB_PATH = os.environ.get("LIMG_HIP_LIB") or os.path.join(HERE, "liblimg_hip.so")  # LIMG_HIP_LIB: A/B runs against another build of the same ABI ("test" = TEST_LIB_PATH)
if LIB_PATH == "test":
    LIB_PATH = TEST_LIB_PATH

P32 = ("pDecoded", "pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax")
P8 = ("pFactorsA", "pFactorsB", "pFactorsC")
PLANES = P32 + P8

# every symbol include/limg_hip.h declares (checked by tests/test_host.py without a GPU)
ABI_SYMBOLS = (
    "limg_hip_init", "limg_hip_shutdown", "limg_hip_default_options_sized", "limg_hip_set_options", "limg_hip_get_options", "limg_hip_encode3d", "limg_hip_encode3d_perf",
    "limg_hip_encode3d_stats", "limg_hip_blocked_encode3d_stats",
    "limg_hip_encode3d_device", "limg_hip_encode3d_batch_device", "limg_hip_last_stats", "limg_hip_compare", "limg_hip_compare_device", "limg_hip_synth_random_gradient_device",
    "limg_hip_synth_photo_noise_device", "limg_hip_context_device_bytes", "limg_hip_version", "limg_hip_profile_begin", "limg_hip_profile_end",
    "limg_hip_host_noise_table", "limg_hip_noise_table_device", "limg_hip_host_chain_call", "limg_hip_host_chain_checkpoints", "limg_hip_host_dense_checkpoints", "limg_hip_host_partition", "limg_hip_check_device_status",
    "limg_hip_stream_bound", "limg_hip_encode_stream_device", "limg_hip_decode_stream_device", "limg_hip_encode_stream", "limg_hip_decode_stream",
    "limg_hip_stream_info",
    "limg_hip_blocked_encode3d", "limg_hip_blocked_encode3d_device", "limg_hip_blocked_regions", "limg_hip_blocked_timing", "limg_hip_blocked_kernel_timing", "limg_hip_blocked_match_bits", "limg_hip_host_blocked_matches",
    "limg_hip_host_blocked_merge", "limg_hip_host_blocked_match_words", "limg_hip_host_blocked_match_bits",
    "limg_hip_comm_unique_id", "limg_hip_comm_init", "limg_hip_comm_destroy", "limg_hip_comm_info", "limg_hip_gather_stream", "limg_hip_encode3d_single_chain_device",
    "limg_hip_encode3d_chain_device", "limg_hip_host_gather_offsets", "limg_hip_host_chain_bases",
)
TEST_ABI_SYMBOLS = ("limg_hip_default_test_options_sized", "limg_hip_set_test_options")  # include/limg_hip_test_hooks.h: exported by liblimg_hip_test.so only
COMM_ID_BYTES = 128

# limg_blocked_encode3d_info (src/limg.h:39-44), member order
BLOCKED_PLANES = (("pDecoded", np.uint32), ("pFactorsA", np.uint8), ("pFactorsB", np.uint8), ("pFactorsC", np.uint8), ("pBlockError", np.uint8), ("pBitsPerPixel", np.uint8),
                  ("pShiftABCX", np.uint32), ("pColAMin", np.uint32), ("pColAMax", np.uint32), ("pColBMin", np.uint32), ("pColBMax", np.uint32), ("pColCMin", np.uint32),
                  ("pColCMax", np.uint32), ("pBlockIndex", np.uint32))
REGION_DTYPE = np.dtype([("ox", "<u4"), ("oy", "<u4"), ("rx", "<u4"), ("ry", "<u4")])

STREAM_HEADER_DTYPE = np.dtype([("magic", "<u4"), ("version", "<u4"), ("sizeX", "<u4"), ("sizeY", "<u4"), ("channels", "<u4"), ("errorFactor", "<u4"),
                                ("blocksX", "<u4"), ("blocksY", "<u4"), ("payloadWords", "<u8"), ("totalBytes", "<u8"), ("flags", "<u4"), ("reserved", "<u4", 3)])
STREAM_BLOCK_DTYPE = np.dtype([("dirA_min", "<i2", 4), ("dirA_max", "<i2", 4), ("dirB_offset", "<i2", 4), ("dirB_mag", "<i2", 4), ("dirC_offset", "<i2", 4),
                               ("dirC_mag", "<i2", 4), ("shift", "<u4"), ("payloadWord", "<u4")])

RECORD_DTYPE = np.dtype([("avg", "<f4", 4), ("dirA_min", "<i2", 4), ("dirA_max", "<i2", 4), ("dirB_offset", "<i2", 4),
                         ("dirB_mag", "<i2", 4), ("dirC_offset", "<i2", 4), ("dirC_mag", "<i2", 4)])


class LimgHipError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in PLANES]


class CompactOut(C.Structure):
    _fields_ = [("pRecords", C.c_void_p), ("pShifts", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("forced_shift", C.c_int32 * 3), ("force_split_kernels", C.c_int32), ("dither_pcg", C.c_int32), ("float_mode", C.c_int32),
                ("legacy_float_stage", C.c_int32), ("collect_stats", C.c_int32), ("host_noise_table", C.c_int32), ("batch_sub_images", C.c_int32), ("ragged_bands", C.c_int32),
                ("ragged_walk_threads", C.c_int32)]


class TestOptions(C.Structure):
    """include/limg_hip_test_hooks.h -- liblimg_hip_test.so only"""
    _fields_ = [("struct_size", C.c_uint32), ("record_limit", C.c_int32), ("batch_chunk", C.c_int32), ("wg_per_cu", C.c_int32), ("whole_image_ragged", C.c_int32),
                ("pipeline", C.c_int32), ("fail_chain_phase1", C.c_int32), ("blocked_no_bound", C.c_int32), ("lookback_spins", C.c_int32), ("base_error_strip", C.c_int32),
                ("skip_publish_strip", C.c_int32), ("blocked_no_order", C.c_int32), ("blocked_no_vec_store", C.c_int32)]


def load_library(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise LimgHipError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU fallback." % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; if liblimg_hip.so pulled
    # in /opt/rocm's copy first, torch's later initialisation would find "no HIP GPUs".  Importing torch first makes the
    # loader resolve our DT_NEEDED libamdhip64.so.7 to the already-loaded copy (same SONAME).  C/C++ users are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.limg_hip_version.restype = C.c_char_p
    L.limg_hip_init.restype = C.c_int
    L.limg_hip_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.limg_hip_shutdown.argtypes = [C.POINTER(C.c_void_p)]
    L.limg_hip_default_options_sized.argtypes = [C.c_void_p, C.c_size_t]
    if hasattr(L, "limg_hip_set_test_options"):
        L.limg_hip_default_test_options_sized.argtypes = [C.c_void_p, C.c_size_t]
        L.limg_hip_set_test_options.restype = C.c_int
        L.limg_hip_set_test_options.argtypes = [C.c_void_p, C.c_void_p]
    L.limg_hip_set_options.restype = C.c_int
    L.limg_hip_set_options.argtypes = [C.c_void_p, C.c_void_p]
    L.limg_hip_get_options.restype = C.c_int
    L.limg_hip_get_options.argtypes = [C.c_void_p, C.c_void_p]
    L.limg_hip_encode3d_stats.restype = C.c_int
    L.limg_hip_encode3d_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.limg_hip_blocked_encode3d_stats.restype = C.c_int
    L.limg_hip_blocked_encode3d_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.limg_hip_encode3d.restype = C.c_int
    L.limg_hip_encode3d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    L.limg_hip_encode3d_perf.restype = C.c_int
    L.limg_hip_encode3d_perf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int]
    L.limg_hip_encode3d_device.restype = C.c_int
    L.limg_hip_encode3d_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    L.limg_hip_encode3d_batch_device.restype = C.c_int
    L.limg_hip_encode3d_batch_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    L.limg_hip_compare.restype = C.c_double
    L.limg_hip_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.limg_hip_compare_device.restype = C.c_double
    L.limg_hip_compare_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.limg_hip_synth_random_gradient_device.restype = C.c_int
    L.limg_hip_synth_random_gradient_device.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int, C.c_size_t, C.c_void_p]
    L.limg_hip_synth_photo_noise_device.restype = C.c_int
    L.limg_hip_synth_photo_noise_device.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_size_t, C.c_void_p]
    L.limg_hip_profile_begin.restype = C.c_int
    L.limg_hip_profile_begin.argtypes = [C.c_void_p]
    L.limg_hip_profile_end.restype = C.c_int
    L.limg_hip_profile_end.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.limg_hip_check_device_status.restype = C.c_int
    L.limg_hip_check_device_status.argtypes = [C.c_void_p]
    L.limg_hip_host_noise_table.restype = C.c_int
    L.limg_hip_host_noise_table.argtypes = [C.c_void_p, C.c_size_t]
    L.limg_hip_last_stats.restype = C.c_int
    L.limg_hip_last_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.limg_hip_noise_table_device.restype = C.c_int
    L.limg_hip_noise_table_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.limg_hip_host_chain_call.restype = C.c_uint64
    L.limg_hip_host_chain_call.argtypes = [C.c_uint64, C.c_size_t, C.c_void_p, C.c_int]
    L.limg_hip_host_chain_checkpoints.restype = C.c_uint64
    L.limg_hip_host_chain_checkpoints.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]
    L.limg_hip_host_dense_checkpoints.restype = C.c_int
    L.limg_hip_host_dense_checkpoints.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p]
    L.limg_hip_host_partition.restype = C.c_int
    L.limg_hip_host_partition.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.limg_hip_context_device_bytes.restype = C.c_size_t
    L.limg_hip_context_device_bytes.argtypes = [C.c_void_p]
    L.limg_hip_blocked_encode3d.restype = C.c_int
    L.limg_hip_blocked_encode3d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int]
    L.limg_hip_blocked_encode3d_device.restype = C.c_int
    L.limg_hip_blocked_encode3d_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.limg_hip_blocked_regions.restype = C.c_int
    L.limg_hip_blocked_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.limg_hip_blocked_timing.restype = C.c_int
    L.limg_hip_blocked_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.limg_hip_blocked_match_bits.restype = C.c_int
    L.limg_hip_blocked_match_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.limg_hip_blocked_kernel_timing.restype = C.c_int
    L.limg_hip_blocked_kernel_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.limg_hip_host_blocked_matches.restype = C.c_int
    L.limg_hip_host_blocked_matches.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.limg_hip_host_blocked_merge.restype = C.c_int
    L.limg_hip_host_blocked_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.limg_hip_host_blocked_match_words.restype = C.c_size_t
    L.limg_hip_host_blocked_match_bits.restype = C.c_int
    L.limg_hip_host_blocked_match_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    L.limg_hip_stream_bound.restype = C.c_size_t
    L.limg_hip_stream_bound.argtypes = [C.c_size_t, C.c_size_t]
    L.limg_hip_encode_stream_device.restype = C.c_int
    L.limg_hip_encode_stream_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, C.c_int, C.c_int,
                                                C.c_void_p]
    L.limg_hip_decode_stream_device.restype = C.c_int
    L.limg_hip_decode_stream_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.limg_hip_encode_stream.restype = C.c_int
    L.limg_hip_encode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, C.c_int, C.c_int]
    L.limg_hip_decode_stream.restype = C.c_int
    L.limg_hip_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.limg_hip_stream_info.restype = C.c_int
    L.limg_hip_stream_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.limg_hip_comm_unique_id.restype = C.c_int
    L.limg_hip_comm_unique_id.argtypes = [C.c_void_p]
    L.limg_hip_comm_init.restype = C.c_int
    L.limg_hip_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.limg_hip_comm_info.restype = C.c_int
    L.limg_hip_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.limg_hip_comm_destroy.restype = C.c_int
    L.limg_hip_comm_destroy.argtypes = [C.c_void_p]
    L.limg_hip_gather_stream.restype = C.c_int
    L.limg_hip_gather_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.limg_hip_encode3d_single_chain_device.restype = C.c_int
    L.limg_hip_encode3d_single_chain_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_size_t, C.c_void_p]
    L.limg_hip_encode3d_chain_device.restype = C.c_int
    L.limg_hip_encode3d_chain_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p]
    L.limg_hip_host_gather_offsets.restype = C.c_int
    L.limg_hip_host_gather_offsets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.limg_hip_host_chain_bases.restype = C.c_int
    L.limg_hip_host_chain_bases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return L


def _check(r, what):
    if r != 0:
        raise LimgHipError("%s failed with limg_hip_result %d" % (what, r))


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def stream_info(stream, lib=None):
    """(sizeX, sizeY, hasAlpha, totalBytes) of a stream held in a numpy uint8 array (host-only, no GPU touched)."""
    lib = lib or load_library()
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    sx, sy, tb, ha = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_int(0)
    _check(lib.limg_hip_stream_info(_np_ptr(stream), stream.size, C.byref(sx), C.byref(sy), C.byref(ha), C.byref(tb)), "limg_hip_stream_info")
    return sx.value, sy.value, bool(ha.value), tb.value


def host_gather_offsets(sizes, lib=None):
    """Offsets (len + 1 entries, the last is the total) of the variable-size stream gather; host only."""
    lib = lib or load_library()
    sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
    out = np.zeros(sizes.size + 1, dtype=np.uint64)
    _check(lib.limg_hip_host_gather_offsets(_np_ptr(sizes), sizes.size, _np_ptr(out)), "limg_hip_host_gather_offsets")
    return out


def host_chain_bases(calls, lib=None):
    """Every rank's first dither-call index in a chain that runs through all strips in rank order; host only."""
    lib = lib or load_library()
    calls = np.ascontiguousarray(calls, dtype=np.uint64)
    out = np.zeros(calls.size, dtype=np.uint64)
    _check(lib.limg_hip_host_chain_bases(_np_ptr(calls), calls.size, _np_ptr(out)), "limg_hip_host_chain_bases")
    return out


def host_blocked_matches(channels, seed, cand, lib=None):
    """The similarity predicate as the host merge evaluates it (records: numpy RECORD_DTYPE items); no GPU touched."""
    lib = lib or load_library()
    a = np.ascontiguousarray(seed); b = np.ascontiguousarray(cand)
    return bool(lib.limg_hip_host_blocked_matches(channels, _np_ptr(a), _np_ptr(b)))


def host_blocked_merge(fits, channels, use_bits=True, lib=None):
    """fits: (by, bx) array of RECORD_DTYPE -> rectangles (REGION_DTYPE) in creation order; host only.  use_bits: go through the precomputed
    similarity-bit window (as the GPU pipeline does) instead of evaluating every pair on demand."""
    lib = lib or load_library()
    fits = np.ascontiguousarray(fits)
    by, bx = fits.shape
    bits = None
    if use_bits:
        bits = np.zeros(by * bx * lib.limg_hip_host_blocked_match_words(), dtype=np.uint64)
        _check(lib.limg_hip_host_blocked_match_bits(_np_ptr(fits), bx, by, channels, _np_ptr(bits)), "limg_hip_host_blocked_match_bits")
    out = np.zeros(by * bx, dtype=REGION_DTYPE)
    n = C.c_size_t(0)
    _check(lib.limg_hip_host_blocked_merge(_np_ptr(fits), _np_ptr(bits) if bits is not None else None, bx, by, channels, _np_ptr(out), out.size, C.byref(n)), "limg_hip_host_blocked_merge")
    return out[:n.value].copy()


class LimgHip:
    """One context on one GPU.  Host-array methods mirror the reference API (src/limg.h:35,37,48); *_device methods take
    torch CUDA tensors (used only as device memory) and run asynchronously on torch's current stream."""

    def __init__(self, device=-1, lib_path=None):
        self.lib = load_library(lib_path)
        self.ctx = C.c_void_p()
        self.comm_rank, self.comm_world = 0, 1
        _check(self.lib.limg_hip_init(device, C.byref(self.ctx)), "limg_hip_init")

    def close(self):
        if self.ctx:
            self.lib.limg_hip_shutdown(C.byref(self.ctx))
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, forced_shift=None, force_split=False, dither_pcg=False, float_fast=False, legacy_float_stage=False, host_noise_table=False, collect_stats=False,
                    batch_sub_images=0, ragged_bands=0, ragged_walk_threads=0, **test_hooks):
        """Every call sets ALL options (unnamed ones to their defaults).  Keywords starting with `test_` are the hooks of include/limg_hip_test_hooks.h
        (test_base_error_strip=N -> limg_hip_test_options.base_error_strip): they need liblimg_hip_test.so and raise on the product library."""
        o = Options()
        self.lib.limg_hip_default_options_sized(C.byref(o), C.sizeof(o))
        if forced_shift is not None:
            for i in range(3):
                o.forced_shift[i] = int(forced_shift[i])
        o.force_split_kernels = int(force_split)
        o.dither_pcg = int(dither_pcg)
        o.float_mode = 1 if float_fast else 0
        o.legacy_float_stage = int(legacy_float_stage)
        o.host_noise_table = int(host_noise_table)
        o.collect_stats = int(collect_stats)
        o.batch_sub_images = int(batch_sub_images)
        o.ragged_bands = int(ragged_bands)
        o.ragged_walk_threads = int(ragged_walk_threads)
        _check(self.lib.limg_hip_set_options(self.ctx, C.byref(o)), "limg_hip_set_options")
        hooks = {}
        for k, v in test_hooks.items():
            if not k.startswith("test_") or k[5:] not in dict(TestOptions._fields_):
                raise TypeError("set_options: unknown option %r" % k)
            hooks[k[5:]] = int(v)
        if self.has_test_hooks:
            t = TestOptions()
            self.lib.limg_hip_default_test_options_sized(C.byref(t), C.sizeof(t))
            for k, v in hooks.items():
                setattr(t, k, v)
            _check(self.lib.limg_hip_set_test_options(self.ctx, C.byref(t)), "limg_hip_set_test_options")
        elif any(hooks.values()):
            raise LimgHipError("test hooks %s need liblimg_hip_test.so (LIMG_HIP_LIB=test); the product library has none" % sorted(k for k, v in hooks.items() if v))

    @property
    def has_test_hooks(self):
        return hasattr(self.lib, "limg_hip_set_test_options")

    def get_options(self):
        o = Options()
        o.struct_size = C.sizeof(o)
        _check(self.lib.limg_hip_get_options(self.ctx, C.byref(o)), "limg_hip_get_options")
        return o

    def set_forced_shift(self, shift=None):
        self.set_options(forced_shift=shift)

    # ---- host pointers (drop-in for limg_encode3d_test / _perf / limg_compare) ---------------------------------------------
    def encode3d(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = {k: np.zeros((h, w), dtype=np.uint32) for k in P32}
        out.update({k: np.zeros((h, w), dtype=np.uint8) for k in P8})
        info = Info(*[out[k].ctypes.data for k in PLANES])
        _check(self.lib.limg_hip_encode3d(self.ctx, _np_ptr(img), w, h, int(has_alpha), C.byref(info), error_factor, pool_threads, int(fast)), "limg_hip_encode3d")
        return out

    def encode3d_stats(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True):
        """limg_hip_encode3d_stats: the planes AND the bit counters of this very encode -> (planes, counters[30], pixels)"""
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = {k: np.zeros((h, w), dtype=np.uint32) for k in P32}
        out.update({k: np.zeros((h, w), dtype=np.uint8) for k in P8})
        info = Info(*[out[k].ctypes.data for k in PLANES])
        cnt = np.zeros(30, dtype=np.uint64)
        px = C.c_uint64(0)
        _check(self.lib.limg_hip_encode3d_stats(self.ctx, _np_ptr(img), w, h, int(has_alpha), C.byref(info), error_factor, pool_threads, int(fast), _np_ptr(cnt), C.byref(px)),
               "limg_hip_encode3d_stats")
        return out, cnt, px.value

    def encode3d_perf(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        _check(self.lib.limg_hip_encode3d_perf(self.ctx, _np_ptr(img), w, h, int(has_alpha), error_factor, pool_threads, int(fast)), "limg_hip_encode3d_perf")

    def compare(self, a, b, has_alpha):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        mse, mx = C.c_double(), C.c_double()
        p = self.lib.limg_hip_compare(self.ctx, _np_ptr(a), _np_ptr(b), a.shape[1], a.shape[0], int(has_alpha), C.byref(mse), C.byref(mx))
        return p, mse.value

    # ---- device pointers ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def alloc_planes_device(self, w, h, device="cuda"):
        import torch
        d = {k: torch.empty((h, w), dtype=torch.int32, device=device) for k in P32}
        d.update({k: torch.empty((h, w), dtype=torch.uint8, device=device) for k in P8})
        return d

    def encode3d_device(self, img, has_alpha, planes=None, error_factor=100, pool_threads=0, fast=True, records=None, shifts=None):
        """img: torch int32 CUDA tensor (h, w); planes: dict from alloc_planes_device or None (`_perf` behaviour)."""
        h, w = img.shape
        info = None
        if planes is not None:
            info = Info(*[(planes[k].data_ptr() if k in planes else None) for k in PLANES])  # compact mode: only the factor planes
        comp = None
        if records is not None or shifts is not None:
            comp = CompactOut(records.data_ptr() if records is not None else None, shifts.data_ptr() if shifts is not None else None)
        _check(self.lib.limg_hip_encode3d_device(self.ctx, C.c_void_p(img.data_ptr()), w, h, int(has_alpha), C.byref(info) if info else None,
                                                 C.byref(comp) if comp else None, error_factor, pool_threads, int(fast), self._stream()), "limg_hip_encode3d_device")

    def encode3d_batch_device(self, imgs, has_alpha, planes_list, error_factor=100, pool_threads=0, fast=True):
        """imgs: list of torch int32 CUDA tensors of one shape; planes_list: one alloc_planes_device dict per image.  One launch pair for the whole list."""
        n = len(imgs)
        h, w = imgs[0].shape
        assert all(tuple(i.shape) == (h, w) for i in imgs) and len(planes_list) == n
        ins = (C.c_void_p * n)(*[i.data_ptr() for i in imgs])
        infos = (Info * n)(*[Info(*[(pl[k].data_ptr() if k in pl else None) for k in PLANES]) for pl in planes_list])
        _check(self.lib.limg_hip_encode3d_batch_device(self.ctx, n, ins, w, h, int(has_alpha), infos, error_factor, pool_threads, int(fast), self._stream()),
               "limg_hip_encode3d_batch_device")

    def last_stats(self):
        """(counters[30], pixels) of the last encode made with set_options(collect_stats=True): the reference's "Average Block Bits" counters (src/limg.cpp:1971-1999)."""
        out = np.zeros(30, dtype=np.uint64)
        px = C.c_uint64(0)
        _check(self.lib.limg_hip_last_stats(self.ctx, _np_ptr(out), C.byref(px)), "limg_hip_last_stats")
        return out, px.value

    def compare_device(self, a, b, has_alpha):
        mse, mx = C.c_double(), C.c_double()
        h, w = a.shape
        p = self.lib.limg_hip_compare_device(self.ctx, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), w, h, int(has_alpha), C.byref(mse), C.byref(mx), self._stream())
        return p, mse.value

    def synth_device(self, kind, w, h, seed=1, opaque=True, y0=0, device="cuda"):
        import torch
        out = torch.empty((h, w), dtype=torch.int32, device=device)
        if kind == "random_gradient":
            _check(self.lib.limg_hip_synth_random_gradient_device(C.c_void_p(out.data_ptr()), w, h, seed, int(opaque), y0, self._stream()), "synth")
        elif kind == "photo_noise":
            _check(self.lib.limg_hip_synth_photo_noise_device(C.c_void_p(out.data_ptr()), w, h, seed, y0, self._stream()), "synth")
        else:
            raise ValueError(kind)
        return out

    # ---- merged-block encoder (limg_blocked_encode3d_test) --------------------------------------------------------------------------
    def blocked_encode3d(self, img, has_alpha, error_factor=100, fast=True):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = {k: np.zeros((h, w), dtype=t) for k, t in BLOCKED_PLANES}
        info = (C.c_void_p * 14)(*[out[k].ctypes.data for k, _ in BLOCKED_PLANES])
        _check(self.lib.limg_hip_blocked_encode3d(self.ctx, _np_ptr(img), w, h, int(has_alpha), info, error_factor, int(fast)), "limg_hip_blocked_encode3d")
        out["regions"] = self.blocked_regions()
        return out

    def blocked_encode3d_device(self, img, has_alpha, planes, error_factor=100, fast=True):
        """img: torch int32 CUDA (h, w); planes: dict name -> torch CUDA tensor for every BLOCKED_PLANES name except pBlockError."""
        h, w = img.shape
        info = (C.c_void_p * 14)(*[(planes[k].data_ptr() if k in planes else None) for k, _ in BLOCKED_PLANES])
        _check(self.lib.limg_hip_blocked_encode3d_device(self.ctx, C.c_void_p(img.data_ptr()), w, h, int(has_alpha), info, error_factor, int(fast), self._stream()),
               "limg_hip_blocked_encode3d_device")

    def alloc_blocked_planes_device(self, w, h, device="cuda"):
        import torch
        return {k: torch.empty((h, w), dtype=(torch.int32 if t == np.uint32 else torch.uint8), device=device) for k, t in BLOCKED_PLANES if k != "pBlockError"}

    def blocked_regions(self):
        n = C.c_size_t(0)
        _check(self.lib.limg_hip_blocked_regions(self.ctx, None, 0, C.byref(n)), "limg_hip_blocked_regions")
        out = np.zeros(n.value, dtype=REGION_DTYPE)
        _check(self.lib.limg_hip_blocked_regions(self.ctx, _np_ptr(out), out.size, C.byref(n)), "limg_hip_blocked_regions")
        return out

    def blocked_timing(self):
        ms = np.zeros(6, dtype=np.float64)
        _check(self.lib.limg_hip_blocked_timing(self.ctx, _np_ptr(ms)), "limg_hip_blocked_timing")
        return dict(zip(("pass1_match_gpu", "merge_host", "fit_search_gpu", "chain_host", "store_gpu", "total"), ms.tolist()))

    def blocked_match_bits(self):
        """The similarity bits (uint64 words, host_blocked_merge's layout) the last merged-block encode's merge worked from."""
        n = C.c_size_t(0)
        _check(self.lib.limg_hip_blocked_match_bits(self.ctx, None, 0, C.byref(n)), "limg_hip_blocked_match_bits")
        out = np.zeros(n.value, dtype=np.uint64)
        _check(self.lib.limg_hip_blocked_match_bits(self.ctx, _np_ptr(out), out.size, C.byref(n)), "limg_hip_blocked_match_bits")
        return out

    def blocked_kernel_timing(self):
        """GPU milliseconds (HIP events) of the last merged-block encode's launches: pass 1, similarity kernels, and -- summed over the worker's batches -- the
        per-rectangle fit + search kernel and the expansion + store kernels."""
        ms = np.zeros(4, dtype=np.float64)
        _check(self.lib.limg_hip_blocked_kernel_timing(self.ctx, _np_ptr(ms)), "limg_hip_blocked_kernel_timing")
        return dict(zip(("pass1_kernel", "match_kernels", "fit_search_kernel", "expand_store_kernels"), ms.tolist()))

    # ---- compact stream ("limg_encode" / "limg_decode") ----------------------------------------------------------------------------
    def stream_bound(self, w, h):
        return self.lib.limg_hip_stream_bound(w, h)

    def encode_stream(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True):
        """host uint32 image -> stream bytes (numpy uint8)"""
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        cap = self.stream_bound(w, h)
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        _check(self.lib.limg_hip_encode_stream(self.ctx, _np_ptr(img), w, h, int(has_alpha), _np_ptr(out), cap, C.byref(n), error_factor, pool_threads, int(fast)), "limg_hip_encode_stream")
        return out[:n.value].copy()

    def decode_stream(self, stream):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        w, h, _, _ = stream_info(stream, self.lib)
        out = np.zeros((h, w), dtype=np.uint32)
        _check(self.lib.limg_hip_decode_stream(self.ctx, _np_ptr(stream), stream.size, _np_ptr(out), out.size), "limg_hip_decode_stream")
        return out

    def encode_stream_device(self, img, has_alpha, out=None, error_factor=100, pool_threads=0, fast=True, want_size=True):
        """img: torch int32 CUDA (h, w) -> (torch uint8 CUDA stream buffer of worst-case size, bytes used or None)"""
        import torch
        h, w = img.shape
        cap = self.stream_bound(w, h)
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=img.device)
        n = C.c_size_t(0)
        _check(self.lib.limg_hip_encode_stream_device(self.ctx, C.c_void_p(img.data_ptr()), w, h, int(has_alpha), C.c_void_p(out.data_ptr()), out.numel(),
                                                      C.byref(n) if want_size else None, error_factor, pool_threads, int(fast), self._stream()), "limg_hip_encode_stream_device")
        return out, (n.value if want_size else None)

    def decode_stream_device(self, stream, nbytes, w, h, out=None):
        import torch
        if out is None:
            out = torch.empty((h, w), dtype=torch.int32, device=stream.device)
        _check(self.lib.limg_hip_decode_stream_device(self.ctx, C.c_void_p(stream.data_ptr()), int(nbytes), C.c_void_p(out.data_ptr()), w, h, self._stream()),
               "limg_hip_decode_stream_device")
        return out

    def check(self):
        _check(self.lib.limg_hip_check_device_status(self.ctx), "limg_hip_check_device_status")

    # ---- multi-GPU: RCCL behind the C ABI ---------------------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        _check(self.lib.limg_hip_comm_unique_id(_np_ptr(buf)), "limg_hip_comm_unique_id")
        return buf

    def comm_init(self, comm_id, rank, world):
        comm_id = np.ascontiguousarray(comm_id, dtype=np.uint8)
        assert comm_id.size == COMM_ID_BYTES
        _check(self.lib.limg_hip_comm_init(self.ctx, _np_ptr(comm_id), rank, world), "limg_hip_comm_init")
        self.comm_rank, self.comm_world = rank, world

    def comm_init_from_torch(self, dist):
        """Create the context's RCCL communicator inside a torch.distributed job: rank 0 makes the id, the process group only carries its 128 bytes."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.from_numpy(self.comm_unique_id() if rank == 0 else np.zeros(COMM_ID_BYTES, dtype=np.uint8)).to(dev)
        dist.broadcast(t, src=0)
        self.comm_init(t.cpu().numpy(), rank, world)

    def comm_info(self):
        """What RCCL says about the context's communicator: {"rank", "ranks" (ncclCommCount), "rccl_version"}."""
        r, n, v = C.c_int(-1), C.c_int(0), C.c_int(0)
        _check(self.lib.limg_hip_comm_info(self.ctx, C.byref(r), C.byref(n), C.byref(v)), "limg_hip_comm_info")
        return {"rank": r.value, "ranks": n.value, "rccl_version": v.value}

    def comm_destroy(self):
        _check(self.lib.limg_hip_comm_destroy(self.ctx), "limg_hip_comm_destroy")

    def gather_stream(self, stream, nbytes, root=0, out=None):
        """stream: torch uint8 CUDA tensor holding this rank's LMG3 stream.  On root returns (gathered CUDA tensor, offsets[world + 1]); None elsewhere."""
        import torch
        world = self.comm_world
        offs = np.zeros(world + 1, dtype=np.uint64)
        if self.comm_rank == root and out is None:
            raise ValueError("root needs an output buffer (worst case: the sum of the ranks' stream bounds)")
        _check(self.lib.limg_hip_gather_stream(self.ctx, C.c_void_p(stream.data_ptr()), int(nbytes), root, C.c_void_p(out.data_ptr()) if out is not None else None,
                                               out.numel() if out is not None else 0, _np_ptr(offs), self._stream()), "limg_hip_gather_stream")
        return (out, offs) if self.comm_rank == root else None

    def encode3d_single_chain_device(self, strip, has_alpha, planes, blocks_before, error_factor=100, fast=True):
        h, w = strip.shape
        info = Info(*[planes[k].data_ptr() for k in PLANES])
        _check(self.lib.limg_hip_encode3d_single_chain_device(self.ctx, C.c_void_p(strip.data_ptr()), w, h, int(has_alpha), C.byref(info), error_factor, int(fast),
                                                              int(blocks_before), self._stream()), "limg_hip_encode3d_single_chain_device")

    def encode3d_chain_device(self, strip, has_alpha, planes, phase, calls=None, base=None, blocks_before=0, error_factor=100, fast=True):
        """phase 1: E step + scan, `calls` (torch int64 CUDA, 1 element) receives the strip's dither-call total; phase 2: F step from `base` (same kind of tensor)."""
        h, w = strip.shape
        info = Info(*[planes[k].data_ptr() for k in PLANES])
        _check(self.lib.limg_hip_encode3d_chain_device(self.ctx, C.c_void_p(strip.data_ptr()), w, h, int(has_alpha), C.byref(info), error_factor, int(fast), phase,
                                                       C.c_void_p(calls.data_ptr()) if calls is not None else None, C.c_void_p(base.data_ptr()) if base is not None else None,
                                                       int(blocks_before), self._stream()), "limg_hip_encode3d_chain_device")

    def profile_begin(self):
        _check(self.lib.limg_hip_profile_begin(self.ctx), "limg_hip_profile_begin")

    def profile_end(self, max_encodes=4096):
        """-> float32 array (n, 3): per profiled encode the milliseconds of k_fit_search, k_strip_scan, k_dither_store."""
        buf = np.zeros((max_encodes, 3), dtype=np.float32)
        n = self.lib.limg_hip_profile_end(self.ctx, _np_ptr(buf), max_encodes)
        if n < 0:
            raise LimgHipError("limg_hip_profile_end failed")
        return buf[:n]

    def device_bytes(self):
        return self.lib.limg_hip_context_device_bytes(self.ctx)


def format_stats(counters, pixels):
    """The text limg_encode3d_test prints for these counters (src/limg.cpp:2235-2248), character for character."""
    c = [int(v) for v in counters]
    t = float(pixels)
    out = "\nAverage Block Bits: %5.3f (A: %5.3f | B: %5.3f | C: %5.3f)\n\n" % ((c[0] + c[1] + c[2]) / t, c[0] / t, c[1] / t, c[2] / t)
    out += "".join(" %d bit   " % (8 - i) for i in range(9))
    for f in range(3):
        out += "\n" + "".join("%7.4f  " % (c[3 + f * 9 + j] * 100.0 / t) for j in range(9))
    return out + "\n\n"
