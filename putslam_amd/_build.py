"""Build recipe of libputslam_hip.so (one place: __graft_entry__.build(), the Makefile's `lib` target and the loader's
build-if-missing all come here).

The library is ONE device translation unit -- csrc/ps_capi.hip and the kernel headers it includes, compiled by hipcc for
gfx950 -- plus host-only translation units compiled as plain C++ against the HIP runtime API (no device code, no kernels):
a change to those recompiles in a second, not with the 54 kernels.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libputslam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")

# -ffp-contract=off: the reference is an SSE2 scalar build without FMA; every product and sum must round separately for
# inlier decisions to be bit-identical.  -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (gfx950's register file is
# unified), the epilogue's v_max3 reads them directly.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]

DEVICE_TU = "ps_capi.hip"
DEVICE_DEPS = ["ps_kernels.h", "ps_matcher_mfma.h", "ps_score_fast.h", "ps_score_euclid.h", "ps_device_math.h",
               "ps_stream_async.h", "ps_stream_push.h", "ps_diag.h", "ps_internal.h"]
HOST_TUS = ["ps_context.cpp", "ps_env.cpp", "ps_batch_queue.cpp"]
HOST_DEPS = ["ps_internal.h"]
HEADER = os.path.join(ROOT, "include", "putslam_hip.h")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def sources():
    """Every file the library is built from."""
    return [os.path.join(CSRC, f) for f in [DEVICE_TU] + DEVICE_DEPS + HOST_TUS] + [HEADER]


def build_hip(force=False, extra_hip_flags=(), out=None):
    """Compiles what is out of date and links libputslam_hip.so.  `extra_hip_flags` / `out`: A/B builds of the profiling
    scripts (another -D, another output path; objects of such a build are not cached)."""
    lib = out or LIB
    if not force and not extra_hip_flags and not _stale(lib, sources()):
        return lib
    os.makedirs(OBJ, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objs = []
    tag = "" if not (extra_hip_flags or out) else ".ab%d" % os.getpid()
    dev_o = os.path.join(OBJ, "ps_capi%s.o" % tag)
    if force or tag or _stale(dev_o, [os.path.join(CSRC, f) for f in [DEVICE_TU] + DEVICE_DEPS] + [HEADER]):
        subprocess.check_call([HIPCC] + HIP_FLAGS + list(extra_hip_flags) + inc + ["-c", os.path.join(CSRC, DEVICE_TU), "-o", dev_o])
    objs.append(dev_o)
    for f in HOST_TUS:
        o = os.path.join(OBJ, os.path.splitext(f)[0] + tag + ".o")
        if force or tag or _stale(o, [os.path.join(CSRC, x) for x in [f] + HOST_DEPS] + [HEADER]):
            subprocess.check_call([CXX] + HOST_FLAGS + inc + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    if tag:
        for o in objs:
            os.remove(o)
    return lib


DIAG_LIB = os.path.join(HERE, "libputslam_hip_diag.so")


def build_hip_diag(force=False):
    """The same library with -DPS_STREAM_DIAG: fault injection into the pipelined stream (tests/test_gpu_stream_async.py) and
    the no-upload experiment of profiles/r05h.  A test artefact: nothing loads it unless PUTSLAM_HIP_LIB names it."""
    if not force and not _stale(DIAG_LIB, sources()):
        return DIAG_LIB
    return build_hip(force=True, extra_hip_flags=["-DPS_STREAM_DIAG"], out=DIAG_LIB)
