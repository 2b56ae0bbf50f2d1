"""Build liblimg_hip.so (hipcc, gfx950 only) in-tree: limg_amd/liblimg_hip.so -- the product -- and, from the same sources with -DLIMG_HIP_TEST_HOOKS,
limg_amd/liblimg_hip_test.so: the build with the fault-injection / A-B hooks of include/limg_hip_test_hooks.h that the test suite loads (tests/conftest.py).

Numerics-critical flags: -ffp-contract=off (hipcc's default is fast contraction; the float stage must round every
multiply and add separately, like the reference's SSE code) and no fast-math of any kind.  f32 division stays
correctly rounded and f32 denormals stay enabled (hipcc defaults)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblimg_hip.so")
TEST_OUT = os.path.join(HERE, "liblimg_hip_test.so")
TEST_FLAGS = ["-DLIMG_HIP_TEST_HOOKS"]
SOURCES = ["limg_hip_kernels.hip", "limg_hip_fit_tpb.hip", "limg_hip_stream.hip", "limg_hip_blocked.hip", "limg_hip_synth.hip", "limg_hip_noise_gpu.hip", "limg_hip_api.hip", "limg_hip_noise.cpp", "limg_hip_blocked_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# per-source extras.  limg_hip_kernels.hip: every atomic in it is issued by one lane (block queue, ticket, look-back descriptors); LLVM's atomic optimizer would still
# wrap each in its wave-aggregation prologue (v_mbcnt x 2, compare, s_bcnt1, broadcast, add) -- five vector instructions per 8x8 block for nothing
SOURCE_FLAGS = {"limg_hip_kernels.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
                # k_stream_decode: LLVM reports one `#pragma unroll` it did not honour (a transformation-ordering note, no particular loop: every loop over a register
                # array IS unrolled -- the kernel has no scratch, which tests/test_decode_isa.py asserts on the compiled assembly)
                "limg_hip_stream.hip": ["-Wno-pass-failed"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def compile_line(src, extra_flags=(), test_hooks=False):
    """hipcc + the flags `src` is compiled with (without -c / -o): ONE definition, used for the objects and for the ISA check"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra_flags) + (TEST_FLAGS if test_hooks else [])
    if src.endswith(".cpp"):
        # host-only translation units; the merge's similarity predicate is float code that must round like the kernels: no contraction
        return [hipcc, "-O3", "-fPIC", "-std=c++17", "-Wall", "-ffp-contract=off", "-fno-fast-math", "-x", "c++"] + [f for f in extra if f.startswith("-D")]
    return [hipcc] + FLAGS + SOURCE_FLAGS.get(src, []) + extra


def device_assembly(src, out_dir, extra_flags=(), test_hooks=False):
    """the gfx950 assembly of `src` from exactly the object's compile line -> path"""
    out = os.path.join(out_dir, src.rsplit(".", 1)[0] + (".test.s" if test_hooks else ".s"))
    subprocess.check_call(compile_line(src, extra_flags, test_hooks) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    return out


def check_isa(out_dir, extra_flags=(), test_hooks=False, verbose=False):
    """k_stream_decode waits for its inline-assembly loads with a hand-counted s_waitcnt: whether that count holds is a property of the compiler's register allocation and
    scheduling, so it is verified on every build, on the compile line that ships (limg_amd/isa_check.py); a build whose code object fails it is refused."""
    import tempfile
    try:
        from limg_amd import isa_check
    except ImportError:  # run as a script: python limg_amd/build.py
        sys.path.insert(0, os.path.dirname(HERE))
        from limg_amd import isa_check
    with tempfile.TemporaryDirectory() as tmp:
        got = isa_check.check_decode_isa(open(device_assembly("limg_hip_stream.hip", tmp, extra_flags, test_hooks)).read())
    if verbose:
        print("isa check (k_stream_decode, %s build): %s" % ("test-hooks" if test_hooks else "product", got), flush=True)
    return got


def build(force=False, verbose=False, extra_flags=(), out_dir=None, test_hooks=False):
    """out_dir: build objects and the library THERE from the sources (nothing reused, nothing in-tree touched) -- the from-scratch check of tests/test_build_from_source.py.
    test_hooks: the -DLIMG_HIP_TEST_HOOKS build (liblimg_hip_test.so; objects *.test.o)."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".o") and not f.startswith("__")] + [os.path.join(inc, "limg_hip.h"), os.path.join(inc, "limg_hip_test_hooks.h"), os.path.abspath(__file__)]
    name = "liblimg_hip_test.so" if test_hooks else "liblimg_hip.so"
    target = os.path.join(HERE, name)
    out = target if out_dir is None else os.path.join(out_dir, name)
    if out_dir is None and not force and _newer(target, deps):
        return target
    extra = list(extra_flags) + (TEST_FLAGS if test_hooks else [])
    jobs = []
    for src in SOURCES:
        obj = os.path.join(CSRC if out_dir is None else out_dir, src.rsplit(".", 1)[0] + (".test.o" if test_hooks else ".o"))
        jobs.append((compile_line(src, extra_flags, test_hooks) + ["-c", os.path.join(CSRC, src), "-o", obj], obj))

    def run(job):
        if verbose:
            print(" ".join(job[0]), flush=True)
        subprocess.check_call(job[0])
        return job[1]

    with ThreadPoolExecutor(max_workers=int(os.environ.get("LIMG_BUILD_JOBS", "4"))) as ex:
        objs = list(ex.map(run, jobs))
    check_isa(out_dir, extra_flags, test_hooks, verbose)  # before the link: no library without it
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    """both libraries: the product and the test-hooks build"""
    return build(force=force, verbose=verbose), build(force=force, verbose=verbose, test_hooks=True)


CLI_SRC = os.path.join(HERE, "..", "tools", "limg_hip_cli.cpp")
CLI_OUT = os.path.join(HERE, "limg_hip_cli")


def build_cli(force=False, verbose=False):
    """tools/limg_hip_cli.cpp -> limg_amd/limg_hip_cli: plain g++ host program on the C ABI (through include/limg_hip_shim.hpp)."""
    build(force=force, verbose=verbose)
    deps = [CLI_SRC, OUT, os.path.join(HERE, "..", "include", "limg_hip.h"), os.path.join(HERE, "..", "include", "limg_hip_shim.hpp")]
    if not force and _newer(CLI_OUT, deps):
        return CLI_OUT
    rocm_lib = os.environ.get("ROCM_LIB", "/opt/rocm/lib")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(HERE, "..", "include"), CLI_SRC, "-o", CLI_OUT, "-L", HERE, "-llimg_hip", "-lz", "-lpthread",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return CLI_OUT


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv, verbose=True))
