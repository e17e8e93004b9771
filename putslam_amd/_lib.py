"""Loader of the HIP C-ABI library (putslam_amd/libputslam_hip.so).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing the import fails loudly and tells the caller how to build it.
"""
import ctypes as C
import os

from ._abi import PsDMatch, PsFrameSet, PsHostPairResults, PsPairResults, PsRansacConfig, PsRansacParams, PsRansacStats

# Hardware queues: the library's launch chains (batch queue, pipelined stream) want one each, the HIP runtime reads
# GPU_MAX_HW_QUEUES once, at its first call.  The library sets its default (16) from a constructor when it is loaded -- too late
# in a Python process whose torch has already initialised HIP, so the default is also set here, at import (csrc/ps_env.cpp).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
# PUTSLAM_HIP_LIB: A/B hook of the profiling scripts (another build of the same library, e.g. a kernel variant)
LIB_PATH = os.environ.get("PUTSLAM_HIP_LIB") or os.path.join(_HERE, "libputslam_hip.so")

# every symbol include/putslam_hip.h declares
EXPORTED = [
    "ps_context_create", "ps_context_destroy", "ps_context_set_stream", "ps_context_synchronize",
    "ps_context_stream", "ps_context_device",
    "ps_context_set_option", "ps_context_get_option",
    "ps_last_error", "ps_abi_version", "ps_device_arch",
    "ps_match_hamming256", "ps_ransac_rigid3d", "ps_umeyama_f32", "ps_kabsch_f64",
    "ps_keypoints2Dto3D", "ps_points3Dto2D", "ps_vo_pairs_device",
    "ps_batch_queue_create", "ps_batch_queue_destroy", "ps_batch_queue_submit", "ps_batch_queue_wait", "ps_batch_queue_query",
    "ps_batch_queue_wait_on_stream", "ps_batch_queue_synchronize", "ps_batch_queue_chains", "ps_batch_queue_context",
    "ps_batch_queue_last_split", "ps_pack_records_device", "ps_match_xyz", "ps_predicted_level", "ps_remove_image_distortion",
    "ps_vo_stream_create", "ps_vo_stream_destroy", "ps_vo_stream_reset", "ps_vo_stream_push",
    "ps_vo_stream_set_result_mode", "ps_vo_stream_set_frame_layout", "ps_vo_stream_packed_stride", "ps_vo_stream_push_many_packed", "ps_vo_stream_configure_async", "ps_vo_stream_push_async", "ps_vo_stream_push_many", "ps_vo_stream_flush",
    "ps_vo_stream_pop_many", "ps_vo_stream_pop", "ps_vo_stream_pending", "ps_vo_stream_graph_launches", "ps_host_alloc", "ps_host_free",
    "ps_algorithmic_bytes", "ps_kernel_names", "ps_last_kernel_times_ms", "ps_kernel_time_totals",
    "ps_context_enable_timing",
    "ps_debug_ransac_counts", "ps_debug_limits", "ps_debug_keys_clean", "ps_debug_fastdiv", "ps_debug_mathcheck", "ps_debug_score_stats", "ps_debug_score_stats_ex", "ps_debug_stage_survivors", "ps_debug_stage_order", "ps_debug_stamps",
    "ps_abi_sizeof_dmatch", "ps_abi_sizeof_params", "ps_abi_sizeof_config", "ps_abi_sizeof_stats",
    "ps_abi_sizeof_frameset", "ps_abi_sizeof_results", "ps_abi_sizeof_host_results",
]

_lib = None
_by_path = {}


class HipLibraryMissing(RuntimeError):
    pass


def _try_build():
    """The .so is a build artefact (git-ignored): if it is absent but hipcc is here, compile it in-tree (_build.py)."""
    from . import _build
    if not (os.path.exists(_build.HIPCC) and os.path.exists(os.path.join(_build.CSRC, _build.DEVICE_TU))):
        return
    import subprocess
    try:
        _build.build_hip()
    except (OSError, subprocess.CalledProcessError):
        pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        _try_build()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _lib = load_path(LIB_PATH)
    return _lib


def load_path(path):
    """Binds ONE build of the library (profiling hook: several builds side by side in one process for A/B timing on the
    same box, clock state and data -- profiles/scripts/ab_libs.py; entry points a build lacks are left unbound)."""
    path = os.path.abspath(path)
    if path in _by_path:
        return _by_path[path]
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Loading it first lets
    # the dynamic linker satisfy our DT_NEEDED libamdhip64.so.7 with the copy torch uses, so torch tensors,
    # torch streams and our kernels share one runtime (two runtimes in one process cannot both see the GPU).
    # Consumers that do not use torch (the C++ drop-in) link /opt/rocm's runtime as usual.
    if os.environ.get("PUTSLAM_HIP_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: stand-alone use
            pass
    real = C.CDLL(path)

    class _Tolerant:
        """argtypes / restype assignments on symbols an older build does not export are dropped."""
        class _Missing:
            pass

        def __getattr__(self, name):
            try:
                return getattr(real, name)
            except AttributeError:
                if path == os.path.abspath(os.path.join(_HERE, "libputslam_hip.so")):   # the product itself: every symbol must be there
                    raise
                return _Tolerant._Missing()

    L = _Tolerant()
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.ps_context_create.argtypes = [i32, C.POINTER(vp)]
    L.ps_context_destroy.argtypes = [vp]
    L.ps_context_destroy.restype = None
    L.ps_context_set_stream.argtypes = [vp, vp]
    L.ps_context_synchronize.argtypes = [vp]
    L.ps_context_stream.argtypes = [vp]
    L.ps_context_stream.restype = vp
    L.ps_context_device.argtypes = [vp]
    L.ps_context_set_option.argtypes = [vp, C.c_char_p, i32]
    L.ps_context_get_option.argtypes = [vp, C.c_char_p]
    L.ps_debug_score_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ps_debug_score_stats_ex.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ps_debug_stage_survivors.argtypes = [vp, i32, vp]
    L.ps_debug_stage_order.argtypes = [vp, i32, i32, vp, vp]
    L.ps_debug_stamps.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ps_last_error.argtypes = [vp]
    L.ps_last_error.restype = C.c_char_p
    L.ps_device_arch.argtypes = [vp]
    L.ps_device_arch.restype = C.c_char_p
    L.ps_match_hamming256.argtypes = [vp, vp, i32, sz, vp, i32, sz, vp, C.POINTER(i32)]
    L.ps_ransac_rigid3d.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp, vp, i32, vp, i32,
                                    vp, i32, vp, vp, C.POINTER(i32), vp, vp]
    L.ps_debug_ransac_counts.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp, vp, i32, vp,
                                         i32, vp, i32, vp, C.POINTER(i32)]
    L.ps_debug_keys_clean.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ps_debug_limits.argtypes = [vp, i32, C.c_double, i32, i32, vp]
    L.ps_debug_fastdiv.argtypes = [vp, C.c_uint64, i32, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ps_debug_mathcheck.argtypes = [vp, i32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ps_umeyama_f32.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.ps_kabsch_f64.argtypes = [vp, vp, vp, i32, i32, vp]
    L.ps_keypoints2Dto3D.argtypes = [vp, vp, i32, vp, i32, i32, sz, vp, C.c_double, vp]
    L.ps_points3Dto2D.argtypes = [vp, vp, i32, vp, vp]
    L.ps_match_xyz.argtypes = [vp, vp, vp, sz, vp, i32, vp, vp, sz, vp, i32, C.c_double, C.c_double, vp, i32,
                               C.POINTER(i32)]
    L.ps_predicted_level.argtypes = [i32, C.c_double, C.c_double]
    L.ps_remove_image_distortion.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ps_vo_stream_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.ps_vo_stream_destroy.argtypes = [vp]
    L.ps_vo_stream_destroy.restype = None
    L.ps_vo_stream_reset.argtypes = [vp]
    L.ps_vo_stream_push.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp, vp, sz, vp, i32, vp,
                                    C.POINTER(i32), vp, vp, vp]
    L.ps_vo_stream_configure_async.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp, i32, i32]
    L.ps_vo_stream_set_result_mode.argtypes = [vp, i32]
    L.ps_vo_stream_push_async.argtypes = [vp, vp, sz, vp, i32]
    L.ps_vo_stream_push_many.argtypes = [vp, vp, vp, vp, i32]
    L.ps_vo_stream_set_frame_layout.argtypes = [vp, i32]
    L.ps_vo_stream_packed_stride.argtypes = [vp]
    L.ps_vo_stream_packed_stride.restype = sz
    L.ps_vo_stream_push_many_packed.argtypes = [vp, vp, sz, vp, i32]
    L.ps_vo_stream_flush.argtypes = [vp]
    L.ps_vo_stream_pop_many.argtypes = [vp, i32, C.POINTER(PsHostPairResults)]
    L.ps_vo_stream_pop.argtypes = [vp, i32, vp, C.POINTER(i32), vp, vp, vp]
    L.ps_vo_stream_pending.argtypes = [vp]
    L.ps_vo_stream_graph_launches.argtypes = [vp]
    L.ps_vo_stream_graph_launches.restype = C.c_longlong
    L.ps_host_alloc.argtypes = [sz]
    L.ps_host_alloc.restype = vp
    L.ps_host_free.argtypes = [vp]
    L.ps_host_free.restype = None
    L.ps_vo_pairs_device.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp,
                                     C.POINTER(PsFrameSet), vp, i32, C.POINTER(PsPairResults)]
    L.ps_batch_queue_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.ps_batch_queue_destroy.argtypes = [vp]
    L.ps_batch_queue_destroy.restype = None
    L.ps_batch_queue_submit.argtypes = [vp, C.POINTER(PsRansacParams), C.POINTER(PsRansacConfig), vp, C.POINTER(PsFrameSet), vp, i32,
                                        C.POINTER(PsPairResults), C.POINTER(C.c_int64)]
    L.ps_batch_queue_wait.argtypes = [vp, C.c_int64]
    L.ps_batch_queue_query.argtypes = [vp, C.c_int64]
    L.ps_batch_queue_wait_on_stream.argtypes = [vp, C.c_int64, vp]
    L.ps_batch_queue_synchronize.argtypes = [vp]
    L.ps_batch_queue_chains.argtypes = [vp]
    L.ps_batch_queue_context.argtypes = [vp, i32]
    L.ps_batch_queue_context.restype = vp
    L.ps_batch_queue_last_split.argtypes = [vp, vp]
    L.ps_pack_records_device.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.ps_algorithmic_bytes.argtypes = [i32, i32, i32, i32]
    L.ps_algorithmic_bytes.restype = C.c_uint64
    L.ps_kernel_names.restype = C.POINTER(C.c_char)
    L.ps_last_kernel_times_ms.argtypes = [vp, vp]
    L.ps_kernel_time_totals.argtypes = [vp, vp, vp]
    L.ps_context_enable_timing.argtypes = [vp, i32]
    for n in ("dmatch", "params", "config", "stats", "frameset", "results", "host_results"):
        getattr(L, "ps_abi_sizeof_" + n).restype = sz
    _by_path[path] = real
    return real


def struct_sizes():
    return dict(dmatch=C.sizeof(PsDMatch), params=C.sizeof(PsRansacParams), config=C.sizeof(PsRansacConfig),
                stats=C.sizeof(PsRansacStats), frameset=C.sizeof(PsFrameSet), results=C.sizeof(PsPairResults),
                host_results=C.sizeof(PsHostPairResults))
