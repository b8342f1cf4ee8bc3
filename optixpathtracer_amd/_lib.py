"""ctypes binding of libptamd.so (include/pt_amd.h). Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB") or os.path.join(HERE, "libptamd.so")  # PT_LIB: experiment builds (tools/variants.sh)

# every symbol include/pt_amd.h declares
EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_set_options", "pt_get_options", "pt_set_probe", "pt_build_cdf",
    "pt_resize", "pt_set_camera", "pt_uvw_frame", "pt_set_partition", "pt_render", "pt_download", "pt_upload_accum",
    "pt_device_buffer", "pt_tonemap_sqrt", "pt_owned_pixels", "pt_pack", "pt_unpack", "pt_get_stats", "pt_trace", "pt_sync",
    "pt_eval_table", "pt_version", "pt_set_probe_image", "pt_get_probe_cdf", "pt_render_regions", "pt_denoise",
    "pt_create_multi", "pt_multi_destroy", "pt_multi_last_error", "pt_multi_size", "pt_multi_ctx", "pt_multi_set_options", "pt_multi_set_probe",
    "pt_multi_set_probe_image", "pt_multi_resize", "pt_multi_set_camera", "pt_multi_render", "pt_multi_render_regions", "pt_multi_gather",
    "pt_multi_get_stats", "pt_export_bvh", "pt_render_batch", "pt_multi_render_batch",
    "pt_pack_async", "pt_pack_wait", "pt_unpack_display", "pt_display_sync", "pt_display_buffer", "pt_download_display", "pt_multi_flush",
    "pt_render_device", "pt_stream", "pt_wait_event", "pt_get_stats_n", "pt_stats_size",
    "pt_load_obj", "pt_obj_free", "pt_obj_num_meshes", "pt_obj_get_mesh", "pt_obj_num_textures", "pt_obj_texture_path", "pt_obj_last_error",
]


class DenoiseParams(C.Structure):  # pt_denoise_params
    _fields_ = [("iterations", C.c_int32), ("sigma_color", C.c_float), ("sigma_normal", C.c_float), ("sigma_albedo", C.c_float),
                ("input", C.c_int32), ("epilogue", C.c_int32)]


class Material(C.Structure):  # pt_material == Material.h:47-68
    _fields_ = [
        ("emission", C.c_float * 3), ("color", C.c_float * 3), ("absorption", C.c_float * 3),
        ("eta", C.c_float), ("metallic", C.c_float), ("subsurface", C.c_float), ("specular", C.c_float),
        ("roughness", C.c_float), ("specularTint", C.c_float), ("anisotropic", C.c_float), ("sheen", C.c_float),
        ("sheenTint", C.c_float), ("clearcoat", C.c_float), ("clearcoatGloss", C.c_float), ("transmission", C.c_float),
        ("bump", C.c_float), ("bumpTile", C.c_float * 3), ("flags", C.c_int32),
    ]


class ObjMesh(C.Structure):  # pt_obj_mesh
    _fields_ = [
        ("vertex", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)), ("texcoord", C.POINTER(C.c_float)), ("index", C.POINTER(C.c_uint32)),
        ("num_vertices", C.c_uint32), ("num_triangles", C.c_uint32), ("material", Material), ("texture_ref", C.c_int32),
    ]


class MeshDesc(C.Structure):
    _fields_ = [
        ("vertex", C.c_void_p), ("num_vertices", C.c_uint32), ("index", C.c_void_p), ("num_triangles", C.c_uint32),
        ("material", Material), ("diffuse_texture_id", C.c_int32), ("texcoord", C.c_void_p),
    ]


class TextureDesc(C.Structure):
    _fields_ = [("pixel", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("num_meshes", C.c_uint32), ("textures", C.POINTER(TextureDesc)), ("num_textures", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [
        ("max_depth", C.c_int32), ("bsdf_mode", C.c_int32), ("max_paths", C.c_uint32), ("kernel_timing", C.c_int32),
        ("bvh_kind", C.c_int32), ("trace_kernel", C.c_int32), ("streams", C.c_int32), ("split_shadow", C.c_int32),
        ("frames_in_flight", C.c_int32),
    ]


class Region(C.Structure):  # pt_region
    _fields_ = [
        ("launch_w", C.c_uint32), ("launch_h", C.c_uint32), ("factor_x", C.c_uint32), ("factor_y", C.c_uint32),
        ("fill_size", C.c_int32), ("cx", C.c_uint32), ("cy", C.c_uint32), ("r_inner", C.c_float), ("r_outer", C.c_float),
        ("offset_x", C.c_uint32), ("offset_y", C.c_uint32), ("redraw", C.c_uint32), ("spp", C.c_uint32), ("subframe_index", C.c_uint32),
    ]


class Variant(C.Structure):  # pt_variant
    _fields_ = [("radiance_tmin", C.c_float), ("cull_back_occlusion", C.c_int32), ("tonemap", C.c_int32), ("exposure", C.c_float), ("white", C.c_float),
                ("initial_depth", C.c_int32), ("write_aov", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
        ("render_ms", C.c_double), ("trace_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
        ("other_ms", C.c_double),
        ("trace_launches", C.c_uint32), ("shadow_launches", C.c_uint32), ("shade_launches", C.c_uint32),
        ("bvh_nodes", C.c_uint32), ("bvh_bytes", C.c_uint64), ("bvh_build_ms", C.c_double), ("bvh_levels", C.c_uint32), ("shaded_hits", C.c_uint64),
        ("frames", C.c_uint64), ("total_radiance_rays", C.c_uint64), ("total_shadow_rays", C.c_uint64),
        ("bvh_builder", C.c_uint32), ("fused_passes", C.c_uint32), ("path_state_allocs", C.c_uint32),
        ("bvh_challengers_skipped", C.c_uint32), ("schedule", C.c_uint32), ("sched_chain_ms", C.c_double), ("sched_fused_ms", C.c_double), ("create_ms", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MultiStats(C.Structure):  # pt_multi_stats
    _fields_ = [("sum", Stats), ("gather_ms", C.c_double), ("exchange", C.c_int32), ("ndev", C.c_int32), ("enqueue_ms", C.c_double), ("threads", C.c_int32),
                ("frames_handed_over", C.c_uint64)]


assert C.sizeof(Material) == 104

_lib = None


def build_library(force: bool = False) -> str:
    """Compile libptamd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcdir = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", srcdir, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", srcdir, "-j2"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C optixpathtracer_amd/csrc`). There is no CPU fallback for the render path."
        )
    # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  If torch is importable, load it first so that libptamd binds to the runtime torch will
    # use for device tensors / RCCL later in the same process (the other order leaves torch without devices).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"libptamd.so does not export {name}")
    vp, i, u32, f = C.c_void_p, C.c_int, C.c_uint32, C.c_float
    L.pt_create.argtypes = [C.POINTER(SceneDesc), i, C.POINTER(vp)]
    L.pt_destroy.argtypes = [vp]
    L.pt_last_error.restype = C.c_char_p
    L.pt_last_error.argtypes = [vp]
    L.pt_set_options.argtypes = [vp, C.POINTER(Options)]
    L.pt_get_options.argtypes = [vp, C.POINTER(Options)]
    L.pt_set_probe.argtypes = [vp, vp, vp, vp, vp, vp, i, i]
    L.pt_build_cdf.argtypes = [vp, i, i, vp, vp, vp, vp]
    L.pt_set_probe_image.argtypes = [vp, vp, i, i]
    L.pt_get_probe_cdf.argtypes = [vp, vp, vp, vp, vp]
    L.pt_resize.argtypes = [vp, i, i]
    L.pt_set_camera.argtypes = [vp, C.POINTER(f * 3), C.POINTER(f * 3), C.POINTER(f * 3), C.POINTER(f * 3)]
    L.pt_uvw_frame.argtypes = [C.POINTER(f * 3), C.POINTER(f * 3), C.POINTER(f * 3), f, f, C.POINTER(f * 3), C.POINTER(f * 3), C.POINTER(f * 3)]
    L.pt_set_partition.argtypes = [vp, i, i, i, i]
    L.pt_render.argtypes = [vp, u32, u32, vp]
    L.pt_render_batch.argtypes = [vp, u32, u32, u32, vp]
    L.pt_sync.argtypes = [vp]
    L.pt_render_device.argtypes = [vp, u32, u32, vp]
    L.pt_load_obj.argtypes = [C.c_char_p, i, C.POINTER(vp)]
    L.pt_obj_free.argtypes = [vp]
    L.pt_obj_free.restype = None
    L.pt_obj_num_meshes.argtypes = [vp]
    L.pt_obj_num_meshes.restype = u32
    L.pt_obj_get_mesh.argtypes = [vp, u32, C.POINTER(ObjMesh)]
    L.pt_obj_num_textures.argtypes = [vp]
    L.pt_obj_num_textures.restype = u32
    L.pt_obj_texture_path.argtypes = [vp, u32]
    L.pt_obj_texture_path.restype = C.c_char_p
    L.pt_obj_last_error.restype = C.c_char_p
    L.pt_stream.restype = vp
    L.pt_stream.argtypes = [vp]
    L.pt_wait_event.argtypes = [vp, vp]
    L.pt_get_stats_n.argtypes = [vp, vp, C.c_size_t]
    L.pt_stats_size.restype = C.c_size_t
    L.pt_render_regions.argtypes = [vp, C.POINTER(Region), u32, C.POINTER(Variant), vp]
    L.pt_download.argtypes = [vp, i, vp, C.c_size_t]
    L.pt_upload_accum.argtypes = [vp, vp, C.c_size_t]
    L.pt_device_buffer.restype = vp
    L.pt_device_buffer.argtypes = [vp, i]
    L.pt_tonemap_sqrt.argtypes = [vp, vp]
    L.pt_denoise.argtypes = [vp, C.POINTER(DenoiseParams), vp, C.POINTER(C.c_double)]
    L.pt_owned_pixels.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.pt_pack.argtypes = [vp, i, vp]
    L.pt_unpack.argtypes = [vp, i, vp]
    L.pt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.pt_pack_async.argtypes = [vp, i, vp, i]
    L.pt_pack_wait.argtypes = [vp, i]
    L.pt_unpack_display.argtypes = [vp, i, vp]
    L.pt_display_sync.argtypes = [vp]
    L.pt_display_buffer.restype = vp
    L.pt_display_buffer.argtypes = [vp, i]
    L.pt_download_display.argtypes = [vp, i, vp, C.c_size_t]
    L.pt_trace.argtypes = [vp, vp, u32, i, vp, vp, i, C.POINTER(C.c_double)]
    L.pt_eval_table.argtypes = [vp, i, vp, i, vp, u32, vp]
    L.pt_export_bvh.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(u32), C.POINTER(u32)]
    L.pt_version.restype = C.c_char_p
    f3p = C.POINTER(f * 3)
    L.pt_create_multi.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), i, C.POINTER(vp)]
    L.pt_multi_destroy.argtypes = [vp]
    L.pt_multi_last_error.restype = C.c_char_p
    L.pt_multi_last_error.argtypes = [vp]
    L.pt_multi_size.argtypes = [vp]
    L.pt_multi_ctx.restype = vp
    L.pt_multi_ctx.argtypes = [vp, i]
    L.pt_multi_set_options.argtypes = [vp, C.POINTER(Options)]
    L.pt_multi_set_probe.argtypes = [vp, vp, vp, vp, vp, vp, i, i]
    L.pt_multi_set_probe_image.argtypes = [vp, vp, i, i]
    L.pt_multi_resize.argtypes = [vp, i, i, i, i]
    L.pt_multi_set_camera.argtypes = [vp, f3p, f3p, f3p, f3p]
    L.pt_multi_render.argtypes = [vp, u32, u32, u32, vp]
    L.pt_multi_render_batch.argtypes = [vp, u32, u32, u32, u32, vp]
    L.pt_multi_render_regions.argtypes = [vp, C.POINTER(Region), u32, C.POINTER(Variant), u32, vp]
    L.pt_multi_gather.argtypes = [vp, i]
    L.pt_multi_flush.argtypes = [vp, vp]
    L.pt_multi_get_stats.argtypes = [vp, C.POINTER(MultiStats)]
    _lib = L
    return L


def build_cdf(data: np.ndarray, width: int, height: int):
    """ProbeData::BuildCDF through the native host implementation (pt_build_cdf)."""
    L = load_library()
    data = np.ascontiguousarray(data, np.float32)
    pdfX = np.empty((height, width), np.float32)
    cdfX = np.empty((height, width), np.float32)
    pdfY = np.empty(height, np.float32)
    cdfY = np.empty(height, np.float32)
    rc = L.pt_build_cdf(data.ctypes.data, width, height, pdfX.ctypes.data, cdfX.ctypes.data, pdfY.ctypes.data, cdfY.ctypes.data)
    if rc:
        raise RuntimeError(f"pt_build_cdf failed ({rc})")
    return pdfX, cdfX, pdfY, cdfY
