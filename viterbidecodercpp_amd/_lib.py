"""ctypes binding of libvit_hip.so -- the C ABI declared in include/vit_hip.h.

The HIP library is the product: there is no Python or CPU fallback.  `load()` raises if the shared object is missing
(run `python -c "import __graft_entry__ as g; g.build()"` or `make -C viterbidecodercpp_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIT_HIP_LIB_PATH: load another build of the same library (A/B timing of kernel variants; never set in tests or bench)
LIB_PATH = os.environ.get("VIT_HIP_LIB_PATH") or os.path.join(_HERE, "libvit_hip.so")

OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_RUNTIME, ERR_NO_DEVICE, ERR_WORKSPACE = -1, -2, -3, -4, -5
PLAN_AUTO, PLAN_LDS, PLAN_REG, PLAN_LDS2 = 0, 1, 2, 3
PLAN_NAMES = {PLAN_LDS: "lds", PLAN_REG: "reg", PLAN_LDS2: "lds2"}

# every symbol include/vit_hip.h declares
EXPORTS = [
    "vit_hip_last_error", "vit_hip_device_count", "vit_hip_create", "vit_hip_destroy", "vit_hip_get_info",
    "vit_hip_set_plan", "vit_hip_blob_bytes", "vit_hip_pack_blob", "vit_hip_create_from_blob",
    "vit_hip_workspace_bytes", "vit_hip_workspace_slab_bytes", "vit_hip_update_batch", "vit_hip_chainback_batch", "vit_hip_decode_batch",
    "vit_hip_export_decisions", "vit_hip_depuncture_batch", "vit_hip_update_host", "vit_hip_chainback_host",
    "vit_hip_reset_batch", "vit_hip_update_batch_resume", "vit_hip_broadcast_table", "vit_hip_synth_batch",
    "vit_hip_count_bit_errors", "vit_hip_shader_clock_mhz", "vit_hip_pipeline_create", "vit_hip_pipeline_submit", "vit_hip_pipeline_sync",
    "vit_hip_pipeline_destroy", "vit_hip_pipeline_get_schedule", "vit_hip_pipeline_last_workspace", "vit_hip_pipeline_set_timing", "vit_hip_pipeline_get_timing",
    "vit_hip_pipeline_wait_event", "vit_hip_get_kernel_resources", "vit_hip_list_kernels",
    "vit_hip_chainback_batch_ex", "vit_hip_pipeline_create_ex", "vit_hip_pipeline_get_schedule_v2", "vit_hip_plan_note",
    "vit_hip_precompile", "vit_hip_update_host_lazy", "vit_hip_chainback_host_lazy", "vit_hip_fetch_decisions_host",
]


class VitHipInfo(C.Structure):
    _fields_ = [("K", C.c_int32), ("R", C.c_int32), ("soft_bytes", C.c_int32), ("error_bytes", C.c_int32),
                ("num_states", C.c_int32), ("decision_words", C.c_int32), ("device", C.c_int32), ("plan", C.c_int32),
                ("soft_decision_high", C.c_int32), ("soft_decision_low", C.c_int32), ("polynomials", C.c_uint32 * 16),
                ("table_is_linear", C.c_int32), ("workspace_tile_frames", C.c_int32)]


class VitHipPipelineSchedule(C.Structure):
    _fields_ = [("workspaces", C.c_int32), ("update_streams", C.c_int32), ("chainback_overlapped", C.c_int32),
                ("chainback_wave_priority", C.c_int32), ("overlap_max_frames", C.c_size_t), ("two_updates_max_frames", C.c_size_t),
                ("workspace_bytes_each", C.c_size_t), ("sub_batch_frames", C.c_size_t),
                ("chainback_small_kernel", C.c_int32), ("reserved", C.c_int32)]


class VitHipPipelineOptions(C.Structure):
    """vit_hip_pipeline_options: every field at its default keeps the library's rule (include/vit_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("chainback_overlap", C.c_int32), ("update_streams", C.c_int32),
                ("sub_batches", C.c_int32), ("workspaces", C.c_int32), ("chainback_small_kernel", C.c_int32),
                ("chainback_wave_priority", C.c_int32)]

    def __init__(self, chainback_overlap=-1, update_streams=0, sub_batches=-1, workspaces=0, chainback_small_kernel=-1,
                 chainback_wave_priority=-1):
        super().__init__(C.sizeof(VitHipPipelineOptions), chainback_overlap, update_streams, sub_batches, workspaces,
                         chainback_small_kernel, chainback_wave_priority)


class VitHipKernelResources(C.Structure):
    _fields_ = [("vgpr_alloc", C.c_uint32), ("accum_offset", C.c_uint32), ("lds_static_bytes", C.c_uint32),
                ("lds_dynamic_bytes", C.c_uint32), ("scratch_bytes", C.c_uint32)]


KERNEL_UPDATE, KERNEL_CHAINBACK, KERNEL_CHAINBACK_ALT, KERNEL_RESUME = 0, 1, 2, 3


class VitHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vit_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension has not been built (there is no CPU fallback)")
    # A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; importing torch first makes
    # libvit_hip.so bind to that copy instead of pulling /opt/rocm's in beside it (two runtimes => "no device").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    L.vit_hip_last_error.restype = C.c_char_p
    L.vit_hip_device_count.restype = i32
    L.vit_hip_create.argtypes = [i32, i32, i32, i32, vp, vp, i32, C.POINTER(vp)]
    L.vit_hip_destroy.argtypes = [vp]
    L.vit_hip_get_info.argtypes = [vp, C.POINTER(VitHipInfo)]
    L.vit_hip_set_plan.argtypes = [vp, i32]
    L.vit_hip_plan_note.argtypes = [vp]
    L.vit_hip_plan_note.restype = C.c_char_p
    L.vit_hip_update_host_lazy.argtypes = [vp, vp, vp, sz, sz, sz, sz, C.POINTER(C.c_uint64)]
    L.vit_hip_chainback_host_lazy.argtypes = [vp, sz, sz, vp]
    L.vit_hip_fetch_decisions_host.argtypes = [vp, sz, sz, vp]
    L.vit_hip_precompile.argtypes = [i32, i32, C.POINTER(C.c_uint32), i32, C.c_char_p, C.c_char_p, sz]
    L.vit_hip_blob_bytes.restype = sz
    L.vit_hip_blob_bytes.argtypes = [i32, i32, i32, i32]
    L.vit_hip_pack_blob.argtypes = [i32, i32, i32, i32, vp, vp, vp, sz]
    L.vit_hip_create_from_blob.argtypes = [vp, sz, i32, C.POINTER(vp)]
    L.vit_hip_workspace_bytes.restype = sz
    L.vit_hip_workspace_bytes.argtypes = [vp, sz, sz]
    L.vit_hip_workspace_slab_bytes.restype = sz
    L.vit_hip_workspace_slab_bytes.argtypes = [vp, sz]
    L.vit_hip_update_batch.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, vp, vp, vp]
    L.vit_hip_chainback_batch.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.vit_hip_decode_batch.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, vp, vp, vp]
    L.vit_hip_export_decisions.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    L.vit_hip_depuncture_batch.argtypes = [vp, vp, sz, vp, sz, sz, vp, vp]
    L.vit_hip_reset_batch.argtypes = [vp, sz, vp, vp, vp]
    L.vit_hip_update_batch_resume.argtypes = [vp, vp, sz, sz, sz, sz, sz, vp, sz, vp, vp, vp]
    L.vit_hip_broadcast_table.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    L.vit_hip_synth_batch.argtypes = [vp, sz, sz, C.c_uint64, C.c_uint64, C.c_float, i32, vp, vp, vp]
    L.vit_hip_count_bit_errors.argtypes = [vp, vp, vp, sz, vp, vp]
    L.vit_hip_shader_clock_mhz.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.vit_hip_pipeline_create.argtypes = [vp, sz, sz, C.POINTER(vp)]
    L.vit_hip_pipeline_create_ex.argtypes = [vp, sz, sz, C.POINTER(VitHipPipelineOptions), C.POINTER(vp)]
    L.vit_hip_pipeline_get_schedule_v2.argtypes = [vp, C.POINTER(VitHipPipelineSchedule), sz]
    L.vit_hip_chainback_batch_ex.argtypes = [vp, vp, sz, sz, vp, vp, vp, i32]
    L.vit_hip_pipeline_submit.argtypes = [vp, vp, sz, vp, vp, vp]
    L.vit_hip_pipeline_sync.argtypes = [vp]
    L.vit_hip_pipeline_destroy.argtypes = [vp]
    L.vit_hip_pipeline_get_schedule.argtypes = [vp, C.POINTER(VitHipPipelineSchedule)]
    L.vit_hip_pipeline_last_workspace.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.vit_hip_pipeline_set_timing.argtypes = [vp, i32]
    L.vit_hip_pipeline_get_timing.argtypes = [vp, sz, vp, vp, vp, C.POINTER(sz)]
    L.vit_hip_pipeline_wait_event.argtypes = [vp, vp]
    L.vit_hip_get_kernel_resources.argtypes = [vp, i32, C.POINTER(VitHipKernelResources)]
    L.vit_hip_list_kernels.argtypes = [sz, C.c_char_p, sz, C.POINTER(VitHipKernelResources)]
    L.vit_hip_update_host.argtypes = [vp, vp, vp, sz, vp, C.POINTER(C.c_uint64)]
    L.vit_hip_chainback_host.argtypes = [vp, vp, sz, sz, vp]
    _lib = L
    return L


def _resources_dict(r):
    return {name: int(getattr(r, name)) for name, _ in VitHipKernelResources._fields_}


def list_kernels():
    """{kernel name: descriptor resources} of every kernel in the library's embedded gfx950 code objects (no GPU needed)"""
    L = load()
    out = {}
    name = C.create_string_buffer(512)
    r = VitHipKernelResources()
    i = 0
    while L.vit_hip_list_kernels(i, name, len(name), C.byref(r)) == OK:
        out[name.value.decode()] = _resources_dict(r)
        i += 1
    return out


def check(rc):
    if rc != OK:
        raise VitHipError(rc, load().vit_hip_last_error().decode(errors="replace"))
