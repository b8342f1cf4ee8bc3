"""ctypes binding for the CPU checker (oracle/pt_oracle.c) and, when present, the reference-header
library oracle/_ref/libptref.so.  TEST INFRASTRUCTURE: importable only from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg — never from optixpathtracer_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    need = force or not all(os.path.exists(os.path.join(HERE, n)) for n in ("liborc_libm.so", "liborc_det.so"))
    if not need:
        src_m = max(os.path.getmtime(os.path.join(HERE, "pt_oracle.c")), os.path.getmtime(os.path.join(HERE, "..", "include", "pt_detmath.h")))
        need = any(os.path.getmtime(os.path.join(HERE, n)) < src_m for n in ("liborc_libm.so", "liborc_det.so"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "liborc_libm.so", "liborc_det.so"], stdout=subprocess.DEVNULL)


class Probe(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.c_void_p), ("pdfX", C.c_void_p), ("cdfX", C.c_void_p), ("pdfY", C.c_void_p), ("cdfY", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("subframe_index", C.c_uint32), ("samples_per_launch", C.c_uint32),
        ("max_depth", C.c_int), ("bsdf_mode", C.c_int),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
    ]


class Region(C.Structure):
    _fields_ = [
        ("launch_w", C.c_uint32), ("launch_h", C.c_uint32), ("factor_x", C.c_uint32), ("factor_y", C.c_uint32),
        ("fill_size", C.c_int32), ("cx", C.c_uint32), ("cy", C.c_uint32), ("r_inner", C.c_float), ("r_outer", C.c_float),
        ("offset_x", C.c_uint32), ("offset_y", C.c_uint32), ("redraw", C.c_uint32), ("spp", C.c_uint32), ("subframe_index", C.c_uint32),
    ]


class Variant(C.Structure):
    _fields_ = [("radiance_tmin", C.c_float), ("cull_back_occlusion", C.c_int), ("tonemap", C.c_int), ("exposure", C.c_float), ("white", C.c_float),
                ("initial_depth", C.c_int), ("write_aov", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64)]


BSDF_DISNEY, BSDF_LAMBERT = 0, 1


class Oracle:
    """mode: 'libm' (independent glibc transcendentals) or 'det' (pt_detmath, bit-comparable with the GPU)."""

    def __init__(self, mode: str = "det"):
        build()
        self.mode = mode
        # ORC_LIB_DIR: the sanitizer builds (make -C oracle asan -> oracle/asan/, tools/sanitize.sh)
        self.lib = L = C.CDLL(os.path.join(os.environ.get("ORC_LIB_DIR") or HERE, f"liborc_{mode}.so"))
        L.orc_tea4.restype = C.c_uint32
        L.orc_tea4.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_lcg.restype = C.c_uint32
        L.orc_lcg.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rnd.restype = C.c_float
        L.orc_rnd.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_random_init.argtypes = [u32p, C.c_uint32]
        L.orc_rand.restype = C.c_uint32
        L.orc_rand.argtypes = [u32p]
        L.orc_randf.restype = C.c_float
        L.orc_randf.argtypes = [u32p]
        L.orc_material_ior.restype = C.c_float
        L.orc_material_ior.argtypes = [C.c_void_p]
        L.orc_material_default.argtypes = [C.c_void_p]
        L.orc_build_cdf.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
        L.orc_probe_dir_to_uv.argtypes = [f32p, f32p]
        L.orc_probe_uv_to_dir.argtypes = [f32p, f32p]
        L.orc_probe_eval.argtypes = [C.POINTER(Probe), f32p, f32p]
        L.orc_probe_sample.argtypes = [C.POINTER(Probe), C.c_uint32, f32p, f32p, C.POINTER(C.c_float), u32p]
        L.orc_probe_pdf.restype = C.c_float
        L.orc_probe_pdf.argtypes = [C.POINTER(Probe), f32p]
        L.orc_make_color.restype = C.c_uint32
        L.orc_make_color.argtypes = [f32p]
        L.orc_tonemap_sqrt.argtypes = [f32p, u32p, C.c_int]
        L.orc_uvw_frame.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, f32p, f32p, f32p]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_void_p, C.c_uint32, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_bvh8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_hit_census.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_hit_census.restype = None
        L.orc_trace_closest.argtypes = [C.c_void_p, f32p, C.c_int, f32p, i32p]
        L.orc_trace_any.argtypes = [C.c_void_p, f32p, C.c_int, u8p]
        L.orc_bsdf_eval.argtypes = [C.c_int, C.c_void_p, f32p, C.c_float, C.c_float, f32p, f32p, f32p, f32p]
        L.orc_bsdf_pdf.restype = C.c_float
        L.orc_bsdf_pdf.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, f32p, f32p, f32p]
        L.orc_bsdf_sample.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, f32p, f32p, C.c_uint32, f32p, C.POINTER(C.c_float), u32p]
        L.orc_basis_from_vector.argtypes = [f32p, f32p, f32p]
        L.orc_uniform_sample_hemisphere.argtypes = [C.c_uint32, f32p]
        L.orc_cosine_sample_hemisphere.argtypes = [C.c_float, C.c_float, f32p]
        L.orc_math_table.argtypes = [C.c_int, f32p, f32p, C.c_int, f32p]
        L.orc_normalize.argtypes = [f32p, f32p]
        L.orc_faceforward.argtypes = [f32p, f32p, f32p]
        L.orc_safe_normalize.argtypes = [f32p, f32p]
        L.orc_lerp3.argtypes = [f32p, f32p, C.c_float, f32p]
        L.orc_clamp3.argtypes = [f32p, C.c_float, C.c_float, f32p]
        L.orc_to_srgb.argtypes = [f32p, f32p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Probe), C.POINTER(Params), f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_render_region.argtypes = [C.c_void_p, C.POINTER(Probe), C.POINTER(Params), C.POINTER(Region), C.POINTER(Variant), f32p, u32p, C.POINTER(Stats)]
        L.orc_render_region_aov.argtypes = [C.c_void_p, C.POINTER(Probe), C.POINTER(Params), C.POINTER(Region), C.POINTER(Variant), f32p, u32p, f32p, f32p, f32p, C.POINTER(Stats)]
        L.orc_denoise.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p, f32p, f32p]
        L.orc_tex2d.argtypes = [u32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
        L.orc_scene_set_textures.argtypes = [C.c_void_p, C.c_void_p, i32p, u8p, C.c_uint32, C.c_void_p, i32p, i32p]
        L.orc_sizeof_material.restype = C.c_size_t
        assert L.orc_sizeof_material() == 104

    # -- helpers
    def make_probe(self, pd) -> Probe:
        """pd: scenes.ProbeData with BuildCDF done; keeps references alive on the returned struct."""
        p = Probe()
        arrs = [np.ascontiguousarray(a, np.float32) for a in (pd.data, pd.pdfValuesX, pd.cdfValuesX, pd.pdfValuesY, pd.cdfValuesY)]
        p.width, p.height = pd.width, pd.height
        p.data, p.pdfX, p.cdfX, p.pdfY, p.cdfY = (a.ctypes.data for a in arrs)
        p._keep = arrs
        return p

    def build_cdf(self, data, w, h):
        data = np.ascontiguousarray(data, np.float32)
        pdfX = np.empty((h, w), np.float32); cdfX = np.empty((h, w), np.float32)
        pdfY = np.empty(h, np.float32); cdfY = np.empty(h, np.float32)
        self.lib.orc_build_cdf(data.reshape(-1), w, h, pdfX.reshape(-1), cdfX.reshape(-1), pdfY, cdfY)
        return pdfX, cdfX, pdfY, cdfY

    def make_scene(self, model, use_bvh=None):
        verts, idx, tri_mesh, mats = model.flatten()
        if use_bvh is None:
            use_bvh = len(idx) > 256
        h = self.lib.orc_scene_create(verts.reshape(-1), len(verts), idx.reshape(-1), len(idx), tri_mesh, mats.ctypes.data, len(mats), int(use_bvh))
        textures = getattr(model, "textures", []) or []
        if textures or any(m.diffuseTextureID >= 0 for m in model.meshes):
            tc, mesh_tex, has_uv = model.flatten_textures()
            pix = [np.ascontiguousarray(t.pixel, np.uint32) for t in textures]
            ptrs = (C.c_void_p * max(1, len(pix)))(*[p.ctypes.data for p in pix])
            ws = np.array([p.shape[1] for p in pix] or [0], np.int32)
            hs = np.array([p.shape[0] for p in pix] or [0], np.int32)
            self.lib.orc_scene_set_textures(h, tc.ctypes.data if tc is not None else None, mesh_tex, has_uv, len(pix), ptrs, ws, hs)
        return SceneHandle(self, h)

    def render(self, scene, probe, cam_uvw, eye, width, height, spp, max_depth=8, subframe=0, bsdf_mode=BSDF_DISNEY, accum=None, nthreads=None):
        """One optixLaunch equivalent. Returns dict(accum, frame, normal, color, albedo, radiance_rays, shadow_rays)."""
        prm = Params()
        prm.width, prm.height, prm.subframe_index, prm.samples_per_launch = width, height, subframe, spp
        prm.max_depth, prm.bsdf_mode = max_depth, bsdf_mode
        U, V, W = cam_uvw
        for dst, src in ((prm.eye, eye), (prm.U, U), (prm.V, V), (prm.W, W)):
            for k in range(3):
                dst[k] = float(src[k])
        n = width * height
        accum = np.zeros((height, width, 4), np.float32) if accum is None else np.ascontiguousarray(accum, np.float32).copy()
        frame = np.zeros((height, width), np.uint32)
        normal = np.zeros((height, width, 4), np.float32)
        color = np.zeros((height, width, 4), np.float32)
        albedo = np.zeros((height, width, 4), np.float32)
        st = Stats()
        if nthreads is None:
            nthreads = min(os.cpu_count() or 1, 64)
        self.lib.orc_render(scene.h, C.byref(probe), C.byref(prm), accum.reshape(-1), frame.ctypes.data, normal.ctypes.data, color.ctypes.data, albedo.ctypes.data, nthreads, C.byref(st))
        return dict(accum=accum, frame=frame, normal=normal, color=color, albedo=albedo, radiance_rays=int(st.radiance_rays), shadow_rays=int(st.shadow_rays), n=n)

    def render_regions(self, scene, probe, cam_uvw, eye, width, height, regions, variant, max_depth, accum, frame, bsdf_mode=BSDF_DISNEY, aov=None):
        """The foveated variants' launches, in order, in place on (accum, frame) and, for variants with write_aov, on
        aov = (normal, color, albedo). Returns total rays."""
        prm = Params()
        prm.width, prm.height, prm.max_depth, prm.bsdf_mode = width, height, max_depth, bsdf_mode
        U, V, W = cam_uvw
        for dst, src in ((prm.eye, eye), (prm.U, U), (prm.V, V), (prm.W, W)):
            for k in range(3):
                dst[k] = float(src[k])
        var = Variant(**variant)
        rays = 0
        for g in regions:
            rg = Region(**g)
            st = Stats()
            if aov is not None:
                self.lib.orc_render_region_aov(scene.h, C.byref(probe), C.byref(prm), C.byref(rg), C.byref(var), accum.reshape(-1), frame.reshape(-1),
                                               aov[0].reshape(-1), aov[1].reshape(-1), aov[2].reshape(-1), C.byref(st))
            else:
                self.lib.orc_render_region(scene.h, C.byref(probe), C.byref(prm), C.byref(rg), C.byref(var), accum.reshape(-1), frame.reshape(-1), C.byref(st))
            rays += int(st.radiance_rays) + int(st.shadow_rays)
        return rays

    def set_bvh8(self, scene, nodes, tris):
        """Make the checker traverse the PRODUCT's 8-wide tree (arrays from SampleRenderer.exportBVH(); None = its own search)."""
        if nodes is None:
            scene._bvh8 = None
            self.lib.orc_scene_set_bvh8(scene.h, None, 0, None, 0)
            return
        nodes = np.ascontiguousarray(nodes, np.uint32)
        tris = np.ascontiguousarray(tris, np.float32)
        scene._bvh8 = (nodes, tris)  # keep alive: the scene borrows them
        self.lib.orc_scene_set_bvh8(scene.h, nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris))

    def trace_closest(self, scene, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        t = np.empty(len(rays), np.float32); prim = np.empty(len(rays), np.int32)
        self.lib.orc_trace_closest(scene.h, rays.reshape(-1), len(rays), t, prim)
        return t, prim

    def hit_census(self, scene, rays):
        """orc_hit_census: the float acceptance rule against double-precision Moller-Trumbore, per ray (needs make_scene(use_bvh=True))."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(8, np.uint64)
        self.lib.orc_hit_census(scene.h, rays.ctypes.data, len(rays), out.ctypes.data)
        keys = ("rays", "same_closest_hit", "accepted_but_inexact", "rejected_but_exact", "order_only", "candidates", "candidates_classified_differently", "rays_with_such_a_candidate")
        return dict(zip(keys, (int(x) for x in out)))

    def trace_any(self, scene, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.empty(len(rays), np.uint8)
        self.lib.orc_trace_any(scene.h, rays.reshape(-1), len(rays), occ)
        return occ

    def denoise(self, color, normal, albedo, iterations=5, sigma_color=1.0, sigma_normal=0.25, sigma_albedo=0.1):
        color = np.ascontiguousarray(color, np.float32)
        h, w = color.shape[:2]
        out = np.empty_like(color)
        self.lib.orc_denoise(w, h, iterations, sigma_color, sigma_normal, sigma_albedo, color.reshape(-1),
                             np.ascontiguousarray(normal, np.float32).reshape(-1), np.ascontiguousarray(albedo, np.float32).reshape(-1), out.reshape(-1))
        return out

    def math_table(self, which, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, np.float32)
        out = np.empty_like(x)
        self.lib.orc_math_table(which, x, y, len(x), out)
        return out


class SceneHandle:
    def __init__(self, orc, h):
        self.orc, self.h = orc, h

    def __del__(self):
        try:
            self.orc.lib.orc_scene_destroy(self.h)
        except Exception:
            pass


def load_ref():
    """oracle/_ref/libptref.so (the reference's own headers) or None when it was not built."""
    path = os.path.join(HERE, "_ref", "libptref.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_tea4.restype = C.c_uint32
    R.ref_tea4.argtypes = [C.c_uint32, C.c_uint32]
    R.ref_lcg.restype = C.c_uint32
    R.ref_lcg.argtypes = [C.POINTER(C.c_uint32)]
    R.ref_rnd.restype = C.c_float
    R.ref_rnd.argtypes = [C.POINTER(C.c_uint32)]
    R.ref_random_init.argtypes = [u32p, C.c_uint32]
    R.ref_rand.restype = C.c_uint32
    R.ref_rand.argtypes = [u32p]
    R.ref_randf.restype = C.c_float
    R.ref_randf.argtypes = [u32p]
    R.ref_basis_from_vector.argtypes = [f32p, f32p, f32p]
    R.ref_uniform_sample_hemisphere.argtypes = [C.c_uint32, f32p]
    R.ref_cosine_sample_hemisphere.argtypes = [C.c_float, C.c_float, f32p]
    R.ref_probe_dir_to_uv.argtypes = [f32p, f32p]
    R.ref_probe_uv_to_dir.argtypes = [f32p, f32p]
    R.ref_probe_eval.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p]
    R.ref_probe_sample.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_uint32, f32p, f32p, C.POINTER(C.c_float), u32p]
    R.ref_luminance.restype = C.c_float
    R.ref_luminance.argtypes = [f32p]
    R.ref_make_color.restype = C.c_uint32
    R.ref_make_color.argtypes = [f32p]
    R.ref_sizeof_material.restype = C.c_size_t
    R.ref_material_default.argtypes = [C.c_void_p]
    R.ref_material_ior.restype = C.c_float
    R.ref_material_ior.argtypes = [C.c_void_p]
    R.ref_uvw_frame.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, f32p, f32p, f32p]
    R.ref_faceforward.argtypes = [f32p, f32p, f32p]
    R.ref_normalize.argtypes = [f32p, f32p]
    if hasattr(R, "ref_safe_normalize"):
        R.ref_safe_normalize.argtypes = [f32p, f32p]
        R.ref_lerp3.argtypes = [f32p, f32p, C.c_float, f32p]
        R.ref_clamp3.argtypes = [f32p, C.c_float, C.c_float, f32p]
        R.ref_to_srgb.argtypes = [f32p, f32p]
    if hasattr(R, "ref_probe_pdf"):
        R.ref_probe_pdf.restype = C.c_float
        R.ref_probe_pdf.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    if hasattr(R, "refm_load_obj"):  # oracle/ref_build/ref_model.cpp: the reference's Model.cpp
        R.refm_load_obj.restype = C.c_void_p
        R.refm_load_obj.argtypes = [C.c_char_p]
        R.refm_new_model.restype = C.c_void_p
        R.refm_add_box.argtypes = [C.c_void_p, C.c_void_p, f32p, f32p]
        for fn in (R.refm_num_meshes, R.refm_num_textures):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p]
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        R.refm_mesh_sizes.argtypes = [C.c_void_p, C.c_int, i32p]
        R.refm_mesh_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.refm_texture_size.argtypes = [C.c_void_p, C.c_int, i32p]
        R.refm_texture_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(R, "refm_loadf"):
        R.refm_loadf.restype = C.c_int
        R.refm_loadf.argtypes = [C.c_char_p, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), C.c_void_p, C.c_size_t]
    return R


def ref_loadf(R, path):
    """stbi_loadf(path, ..., 4) of the reference's vendored stb_image (main.cpp:146-156 loadProbe): (h, w, 4) float32 or None."""
    res = np.zeros(2, np.int32)
    if not R.refm_loadf(os.fsencode(path), res, None, 0):
        return None
    out = np.zeros((int(res[1]), int(res[0]), 4), np.float32)
    R.refm_loadf(os.fsencode(path), res, out.ctypes.data, out.size)
    return out


def ref_model_arrays(R, handle):
    """What the reference's Model holds (Model.h:10-42), as numpy arrays: ([mesh dict], [texture (h,w) u32])."""
    from optixpathtracer_amd import scenes

    meshes, textures = [], []
    for i in range(R.refm_num_meshes(handle)):
        sz = np.zeros(5, np.int32)
        R.refm_mesh_sizes(handle, i, sz)
        v = np.zeros((sz[0], 3), np.float32); n = np.zeros((sz[1], 3), np.float32); tc = np.zeros((sz[2], 2), np.float32)
        idx = np.zeros((sz[3], 3), np.uint32); mat = np.zeros((), scenes.MATERIAL_DTYPE)
        R.refm_mesh_copy(handle, i, v.ctypes.data, n.ctypes.data, tc.ctypes.data, idx.ctypes.data, mat.ctypes.data)
        meshes.append(dict(vertex=v, normal=n, texcoord=tc, index=idx, material=mat, diffuseTextureID=int(sz[4])))
    for i in range(R.refm_num_textures(handle)):
        res = np.zeros(2, np.int32)
        R.refm_texture_size(handle, i, res)
        px = np.zeros((res[1], res[0]), np.uint32)
        R.refm_texture_copy(handle, i, px.ctypes.data)
        textures.append(px)
    return meshes, textures


def ref_load_obj(R, path):
    """loadOBJ of the reference itself (None when it threw)."""
    h = R.refm_load_obj(os.fsencode(path))
    return None if not h else ref_model_arrays(R, h)


def ref_add_boxes(R, boxes):
    """addBox of the reference itself for a list of (material, pos, extend)."""
    h = R.refm_new_model()
    for mat, pos, ext in boxes:
        m = np.array(mat)
        R.refm_add_box(h, m.ctypes.data, np.array(pos, np.float32), np.array(ext, np.float32))
    return ref_model_arrays(R, h)[0]
