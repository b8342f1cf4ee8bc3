"""Host-side mirror of the reference's SampleRenderer (SimplePathtracer.h:38-176) over the C ABI.

Same public surface — SampleRenderer(model), render(), resize(size), downloadPixels(), setCamera(camera),
setProbe(probe), the public `launchParams` the app pokes directly (samples_per_launch, frame.subframe_index,
frame.size) — and the same error behaviour: failures raise RuntimeError (the reference throws
sutil::Exception : std::runtime_error), render() before resize() silently does nothing, resize to 0 is ignored.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import Material as CMaterial
from ._lib import MeshDesc, Options, Region, SceneDesc, Stats, TextureDesc, Variant
from .scenes import Model, ProbeData, uvw_frame

PT_BUF_ACCUM, PT_BUF_FRAME, PT_BUF_COLOR, PT_BUF_NORMAL, PT_BUF_ALBEDO, PT_BUF_DENOISED = range(6)
PT_BSDF_DISNEY, PT_BSDF_LAMBERT = 0, 1


@dataclass
class Camera:
    """sutil::Camera (eye, lookat, up, fovY in degrees, aspectRatio)."""

    eye: tuple
    lookat: tuple
    up: tuple = (0.0, 1.0, 0.0)
    fovY: float = 35.0
    aspectRatio: float = 1.0

    def UVWFrame(self):
        L = _lib.load_library()
        f3 = C.c_float * 3
        U, V, W = f3(), f3(), f3()
        rc = L.pt_uvw_frame(C.byref(f3(*self.eye)), C.byref(f3(*self.lookat)), C.byref(f3(*self.up)), self.fovY, self.aspectRatio, C.byref(U), C.byref(V), C.byref(W))
        if rc:
            raise RuntimeError("pt_uvw_frame failed")
        return np.array(U, np.float32), np.array(V, np.float32), np.array(W, np.float32)


@dataclass
class _Frame:
    size: tuple = (0, 0)
    subframe_index: int = 0


@dataclass
class LaunchParams:
    """The fields of LaunchParams (LaunchParams.h:51-79) the application sets directly."""

    frame: _Frame = field(default_factory=_Frame)
    samples_per_launch: int = 1


def _scene_desc(model: Model):
    """pt_scene_desc for a Model plus the arrays it points into (keep them alive until pt_create returns)."""
    meshes = (MeshDesc * len(model.meshes))()
    keep = []
    for k, m in enumerate(model.meshes):
        v = np.ascontiguousarray(m.vertex, np.float32)
        ix = np.ascontiguousarray(m.index, np.uint32)
        keep += [v, ix]
        meshes[k].vertex = v.ctypes.data
        meshes[k].num_vertices = len(v)
        meshes[k].index = ix.ctypes.data
        meshes[k].num_triangles = len(ix)
        C.memmove(C.byref(meshes[k].material), np.asarray(m.material).tobytes(), 104)
        meshes[k].diffuse_texture_id = m.diffuseTextureID
        if m.texcoord is not None and len(m.texcoord):
            tc = np.ascontiguousarray(m.texcoord, np.float32)
            assert tc.shape == (len(v), 2)
            keep.append(tc)
            meshes[k].texcoord = tc.ctypes.data
    textures = getattr(model, "textures", []) or []
    tdesc = (TextureDesc * max(1, len(textures)))()
    for k, t in enumerate(textures):
        px = np.ascontiguousarray(t.pixel, np.uint32)
        keep.append(px)
        tdesc[k].pixel = px.ctypes.data
        tdesc[k].height, tdesc[k].width = px.shape
    keep += [meshes, tdesc]
    return SceneDesc(meshes, len(model.meshes), tdesc, len(textures)), keep


class SampleRenderer:
    def __init__(self, model: Model, device: int = 0):
        self._L = L = _lib.load_library()
        self._ctx = C.c_void_p()
        self.launchParams = LaunchParams()
        sd, self._keep = _scene_desc(model)
        rc = L.pt_create(C.byref(sd), device, C.byref(self._ctx))
        if rc:
            self._ctx = C.c_void_p()
            raise RuntimeError(f"pt_create failed ({rc}): {L.pt_last_error(None).decode()}")
        self._keep = []  # the scene was deep-copied

    # -- internals
    def _ck(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.pt_last_error(self._ctx).decode()}")

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference API
    def render(self, out: np.ndarray | None = None):
        """render() / render(CUDAOutputBuffer&): one launch with the current launchParams."""
        ptr = None
        if out is not None:
            assert out.dtype == np.uint32 and out.flags["C_CONTIGUOUS"]
            ptr = out.ctypes.data
        self._ck(self._L.pt_render(self._ctx, int(self.launchParams.samples_per_launch), int(self.launchParams.frame.subframe_index), ptr), "pt_render")

    def renderDevice(self, dev_ptr: int):
        """render(sutil::CUDAOutputBuffer<uint32_t>&) (SimplePathtracer.cpp:99-107): the rgba8 frame lands in the caller's DEVICE buffer
        (width*height*4 bytes, e.g. tensor.data_ptr()); complete when the call returns (pt_render_device)."""
        self._ck(self._L.pt_render_device(self._ctx, int(self.launchParams.samples_per_launch), int(self.launchParams.frame.subframe_index), dev_ptr), "pt_render_device")

    @property
    def stream(self) -> int:
        """SampleRenderer::stream (SimplePathtracer.h:107): the context's hipStream_t as an integer (pt_stream); non-blocking."""
        return self._L.pt_stream(self._ctx)

    def waitEvent(self, hip_event: int):
        """Everything enqueued on the context's stream from now on waits for the caller's event (pt_wait_event; torch: event.cuda_event)."""
        self._ck(self._L.pt_wait_event(self._ctx, hip_event), "pt_wait_event")

    def renderBatch(self, count: int, out: np.ndarray | None = None):
        """`count` iterations of the reference's progressive loop (render(); subframe_index++, main.cpp:273-278) as ONE wavefront
        batch (pt_render_batch): same buffers bit for bit, count times the rays per launch.  Like render(), it leaves
        launchParams.frame.subframe_index to the application (advance it by `count`)."""
        ptr = None
        if out is not None:
            assert out.dtype == np.uint32 and out.flags["C_CONTIGUOUS"]
            ptr = out.ctypes.data
        self._ck(self._L.pt_render_batch(self._ctx, int(self.launchParams.samples_per_launch), int(self.launchParams.frame.subframe_index), int(count), ptr), "pt_render_batch")

    def resize(self, newSize):
        w, h = int(newSize[0]), int(newSize[1])
        self._ck(self._L.pt_resize(self._ctx, w, h), "pt_resize")
        if w and h:
            self.launchParams.frame.size = (w, h)

    def downloadPixels(self) -> np.ndarray:
        return self.download(PT_BUF_FRAME)

    def setCamera(self, camera: Camera):
        U, V, W = camera.UVWFrame()
        self.setCameraUVW(camera.eye, U, V, W)

    def setCameraUVW(self, eye, U, V, W):
        f3 = C.c_float * 3
        self._ck(self._L.pt_set_camera(self._ctx, C.byref(f3(*[float(x) for x in eye])), C.byref(f3(*[float(x) for x in U])), C.byref(f3(*[float(x) for x in V])), C.byref(f3(*[float(x) for x in W]))), "pt_set_camera")

    def setProbe(self, probe: ProbeData):
        if not probe.valid:
            raise RuntimeError("Probe Data is not valid")  # Probe.h:104-105
        arrs = [np.ascontiguousarray(a, np.float32) for a in (probe.data, probe.pdfValuesX, probe.cdfValuesX, probe.pdfValuesY, probe.cdfValuesY)]
        self._ck(self._L.pt_set_probe(self._ctx, *[a.ctypes.data for a in arrs], probe.width, probe.height), "pt_set_probe")
        self._probe_wh = (probe.width, probe.height)

    def setProbeImage(self, data: np.ndarray):
        """loadProbe + BuildCDF + setProbe with the CDF built on the GPU (bit-identical to the host BuildCDF)."""
        d = np.ascontiguousarray(data, np.float32)
        h, w = d.shape[0], d.shape[1]
        self._ck(self._L.pt_set_probe_image(self._ctx, d.ctypes.data, w, h), "pt_set_probe_image")
        self._probe_wh = (w, h)

    def probeCDF(self):
        w, h = self._probe_wh
        pdfX = np.empty((h, w), np.float32); cdfX = np.empty((h, w), np.float32)
        pdfY = np.empty(h, np.float32); cdfY = np.empty(h, np.float32)
        self._ck(self._L.pt_get_probe_cdf(self._ctx, pdfX.ctypes.data, cdfX.ctypes.data, pdfY.ctypes.data, cdfY.ctypes.data), "pt_get_probe_cdf")
        return pdfX, cdfX, pdfY, cdfY

    # -- the foveated variants' render() (HelloPathtracing_sv4_vmv23/SimplePathtracer.cpp:77-216)
    SV4_VARIANT = dict(radiance_tmin=0.01, cull_back_occlusion=1, tonemap=1, exposure=4.0, white=1.0)
    # HelloPathtracing_sv3: same device code except exposure 2^3 and no Reinhard in the final write; host radii 100/200, spp 2/8/64
    SV3_VARIANT = dict(radiance_tmin=0.01, cull_back_occlusion=1, tonemap=2, exposure=8.0, white=1.0)
    SV3_SCHEDULE = dict(inner_radius=100, outer_radius=200, spp=(2, 8, 64))
    # HelloPathtracing_sv and _sv2 (one device program): canonical tmin / occlusion flags / make_color, but prd.depth starts at 1 with
    # the cutoff `depth >= 3` (deviceProgram.cu:428,483 -> setOptions(max_depth=3)) and the launch also writes the AOV buffers (:553-555);
    # host schedule = sv3's radii and sample counts (HelloPathtracing_sv/SimplePathtracer.cpp:131-198)
    SV_VARIANT = dict(radiance_tmin=0.001, cull_back_occlusion=0, tonemap=0, exposure=1.0, white=1.0, initial_depth=1, write_aov=1)
    SV_MAX_DEPTH = 3

    def renderRegions(self, regions, variant=None, out: np.ndarray | None = None):
        """regions: list of dicts with pt_region's fields; variant: dict with pt_variant's fields (None = canonical)."""
        arr = (Region * len(regions))()
        for k, g in enumerate(regions):
            for name, _ in Region._fields_:
                setattr(arr[k], name, g[name])
        vp = None
        if variant is not None:
            v = Variant(**variant)
            vp = C.byref(v)
        ptr = out.ctypes.data if out is not None else None
        self._ck(self._L.pt_render_regions(self._ctx, arr, len(regions), vp, ptr), "pt_render_regions")

    @staticmethod
    def foveatedRegions(size, c, subframe_index, inner_radius=157, outer_radius=515, spp=(1, 2, 8)):
        """The three launches of sv4's FOV_ON render(): periphery at 1/4 resolution (accumulating), an annulus at 1/2
        resolution and the fovea at full resolution (both redrawn every frame with subframe_index 0)."""
        w, h = size
        cx, cy = c
        u32 = lambda v: int(v) & 0xFFFFFFFF  # uint2 arithmetic wraps like the reference's
        return [
            dict(launch_w=w // 4, launch_h=h // 4, factor_x=4, factor_y=4, fill_size=4, cx=cx, cy=cy, r_inner=float(outer_radius), r_outer=1000000000.0,
                 offset_x=0, offset_y=0, redraw=0, spp=spp[0], subframe_index=subframe_index),
            dict(launch_w=outer_radius + 2, launch_h=outer_radius + 2, factor_x=2, factor_y=2, fill_size=2, cx=cx, cy=cy, r_inner=float(inner_radius),
                 r_outer=float(outer_radius + 2), offset_x=u32(cx - (outer_radius + 2)), offset_y=u32(cy - (outer_radius + 2)), redraw=1, spp=spp[1], subframe_index=0),
            dict(launch_w=(inner_radius + 1) * 2, launch_h=(inner_radius + 1) * 2, factor_x=1, factor_y=1, fill_size=1, cx=cx, cy=cy, r_inner=0.0,
                 r_outer=float(inner_radius + 1), offset_x=u32(cx - (inner_radius + 1)), offset_y=u32(cy - (inner_radius + 1)), redraw=1, spp=spp[2], subframe_index=0),
        ]

    def renderFoveated(self, c, inner_radius=157, outer_radius=515, spp=(1, 2, 8), out=None, variant=None):
        """sv4 SampleRenderer::render() with FOV_ON: three launches around the gaze point c (launchParams.frame.c),
        then launchParams.frame.subframe_index++ (SimplePathtracer.cpp:132-216)."""
        regs = self.foveatedRegions(self.launchParams.frame.size, c, int(self.launchParams.frame.subframe_index), inner_radius, outer_radius, spp)
        self.renderRegions(regs, variant or self.SV4_VARIANT, out)
        self.launchParams.frame.subframe_index += 1

    # -- beyond the reference (runtime versions of its compile-time constants, multi-GPU, stats)
    def setOptions(self, max_depth=8, bsdf_mode=PT_BSDF_DISNEY, max_paths=0, bvh_kind=0, trace_kernel=0, streams=0, split_shadow=0, kernel_timing=0, frames_in_flight=0):
        o = Options(max_depth, bsdf_mode, max_paths, kernel_timing, bvh_kind, trace_kernel, streams, split_shadow, frames_in_flight)
        self._ck(self._L.pt_set_options(self._ctx, C.byref(o)), "pt_set_options")

    def setPartition(self, rank, world, tile_w=64, tile_h=16):
        self._ck(self._L.pt_set_partition(self._ctx, rank, world, tile_w, tile_h), "pt_set_partition")

    def sync(self):
        """Waits for the frames in flight (setOptions(frames_in_flight=2 or 3)); their errors surface here."""
        self._ck(self._L.pt_sync(self._ctx), "pt_sync")

    def download(self, which) -> np.ndarray:
        w, h = self.launchParams.frame.size
        if which == PT_BUF_FRAME:
            out = np.empty((h, w), np.uint32)
        else:
            out = np.empty((h, w, 4), np.float32)
        self._ck(self._L.pt_download(self._ctx, which, out.ctypes.data, out.nbytes), "pt_download")
        return out

    def uploadAccum(self, accum: np.ndarray):
        a = np.ascontiguousarray(accum, np.float32)
        self._ck(self._L.pt_upload_accum(self._ctx, a.ctypes.data, a.nbytes), "pt_upload_accum")

    def tonemapSqrt(self) -> np.ndarray:
        w, h = self.launchParams.frame.size
        out = np.empty((h, w), np.uint32)
        self._ck(self._L.pt_tonemap_sqrt(self._ctx, out.ctypes.data), "pt_tonemap_sqrt")
        return out

    def denoise(self, iterations=5, sigma_color=1.0, sigma_normal=0.25, sigma_albedo=0.1, input=PT_BUF_COLOR, epilogue=0):
        """OptiXDenoiser::exec() as the reference wires it (SimplePathtracer.cpp:104-105,138-146; its own body is empty): an
        a-trous filter of `input` guided by normal_buffer and albedo_buffer → PT_BUF_DENOISED; epilogue 1 = computeFinalPixelColors,
        2 = make_color into frame_buffer.  Returns (denoised float4 image, kernel ms)."""
        from ._lib import DenoiseParams

        prm = DenoiseParams(int(iterations), float(sigma_color), float(sigma_normal), float(sigma_albedo), int(input), int(epilogue))
        ms = C.c_double()
        self._ck(self._L.pt_denoise(self._ctx, C.byref(prm), None, C.byref(ms)), "pt_denoise")
        return self.download(PT_BUF_DENOISED), ms.value

    def stats(self) -> dict:
        s = Stats()
        self._ck(self._L.pt_get_stats(self._ctx, C.byref(s)), "pt_get_stats")
        return s.as_dict()

    def ownedPixels(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._ck(self._L.pt_owned_pixels(self._ctx, C.byref(a), C.byref(b)), "pt_owned_pixels")
        return a.value, b.value

    def deviceBuffer(self, which) -> int:
        return self._L.pt_device_buffer(self._ctx, which)

    def pack(self, which, dev_ptr: int):
        self._ck(self._L.pt_pack(self._ctx, which, dev_ptr), "pt_pack")

    def unpack(self, which, dev_ptr: int):
        self._ck(self._L.pt_unpack(self._ctx, which, dev_ptr), "pt_unpack")

    # -- display hand-off that overlaps the next frame (pt_pack_async ... pt_download_display, include/pt_amd.h)
    def packAsync(self, which, dev_ptr: int, slot: int):
        self._ck(self._L.pt_pack_async(self._ctx, which, dev_ptr, int(slot)), "pt_pack_async")

    def packWait(self, slot: int):
        self._ck(self._L.pt_pack_wait(self._ctx, int(slot)), "pt_pack_wait")

    def unpackDisplay(self, which, dev_ptr: int):
        self._ck(self._L.pt_unpack_display(self._ctx, which, dev_ptr), "pt_unpack_display")

    def displaySync(self):
        self._ck(self._L.pt_display_sync(self._ctx), "pt_display_sync")

    def downloadDisplay(self, which) -> np.ndarray:
        w, h = self.launchParams.frame.size
        out = np.empty((h, w), np.uint32) if which == PT_BUF_FRAME else np.empty((h, w, 4), np.float32)
        self._ck(self._L.pt_download_display(self._ctx, which, out.ctypes.data, out.nbytes), "pt_download_display")
        return out

    def exportBVH(self):
        """The traversal structure as the kernels see it (pt_export_bvh): (nodes uint32[num_nodes, 20], tris float32[num_tris, 12])."""
        nn, nt = C.c_uint32(), C.c_uint32()
        self._ck(self._L.pt_export_bvh(self._ctx, None, 0, None, 0, C.byref(nn), C.byref(nt)), "pt_export_bvh")
        nodes = np.empty((nn.value, 20), np.uint32)
        tris = np.empty((nt.value, 12), np.float32)
        self._ck(self._L.pt_export_bvh(self._ctx, nodes.ctypes.data, nodes.nbytes, tris.ctypes.data, tris.nbytes, None, None), "pt_export_bvh")
        return nodes, tris

    def trace(self, rays: np.ndarray, any_hit=False, iters=1):
        """optixTrace as a batch query: rays (n,8) = o.xyz,tmin,d.xyz,tmax → (t, prim) or occluded flags; + kernel ms."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = len(rays)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        ms = C.c_double()
        self._ck(self._L.pt_trace(self._ctx, rays.ctypes.data, n, int(any_hit), t.ctypes.data, prim.ctypes.data, iters, C.byref(ms)), "pt_trace")
        return (prim.astype(np.uint8) if any_hit else (t, prim)), ms.value

    def evalTable(self, which, inp: np.ndarray, out_width: int, material=None, bsdf_mode=PT_BSDF_DISNEY) -> np.ndarray:
        inp = np.ascontiguousarray(inp, np.float32)
        n = inp.shape[0]
        out = np.empty((n, out_width), np.float32)
        mp = None
        if material is not None:
            mbuf = CMaterial()
            C.memmove(C.byref(mbuf), np.asarray(material).tobytes(), 104)
            mp = C.cast(C.byref(mbuf), C.c_void_p)
        self._ck(self._L.pt_eval_table(self._ctx, which, mp, bsdf_mode, inp.ctypes.data, n, out.ctypes.data), "pt_eval_table")
        return out


class _RankView(SampleRenderer):
    """A rank's context of a MultiRenderer seen through the single-context facade (download, stats, deviceBuffer ...).
    It does not own the context."""

    def __init__(self, L, ctx, launchParams):
        self._L, self._ctx, self.launchParams = L, ctx, launchParams

    def close(self):
        self._ctx = C.c_void_p()


class MultiRenderer:
    """SampleRenderer over several GPUs of ONE process (pt_create_multi, include/pt_amd.h): the same call sequence as
    the reference's renderer; the frame is tile-partitioned over `devices`, rendered concurrently, and gathered on every
    rank by one RCCL all-gather (direct device-to-device copies when ranks share a device)."""

    EXCHANGE = {0: "none", 1: "rccl", 2: "peer_copy"}

    def __init__(self, model: Model, devices=(0,)):
        from ._lib import MultiStats  # noqa: F401

        self._L = L = _lib.load_library()
        self._m = C.c_void_p()
        self.launchParams = LaunchParams()
        self.devices = [int(d) for d in devices]
        sd, keep = _scene_desc(model)
        dv = (C.c_int * len(self.devices))(*self.devices)
        rc = L.pt_create_multi(C.byref(sd), dv, len(self.devices), C.byref(self._m))
        del keep
        if rc:
            self._m = C.c_void_p()
            raise RuntimeError(f"pt_create_multi failed ({rc}): {L.pt_multi_last_error(None).decode()}")
        self.gather_mask = 1 << PT_BUF_FRAME  # what render() assembles on every rank (the reference displays frame_buffer)

    def _ck(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.pt_multi_last_error(self._m).decode()}")

    def close(self):
        if getattr(self, "_m", None):
            self._L.pt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def world(self):
        return self._L.pt_multi_size(self._m)

    def rank(self, r) -> SampleRenderer:
        ctx = self._L.pt_multi_ctx(self._m, int(r))
        if not ctx:
            raise IndexError(r)
        return _RankView(self._L, C.c_void_p(ctx), self.launchParams)

    def setOptions(self, max_depth=8, bsdf_mode=PT_BSDF_DISNEY, max_paths=0, bvh_kind=0, trace_kernel=0, streams=0, split_shadow=0, kernel_timing=0, frames_in_flight=0):
        o = Options(max_depth, bsdf_mode, max_paths, kernel_timing, bvh_kind, trace_kernel, streams, split_shadow, frames_in_flight)
        self._ck(self._L.pt_multi_set_options(self._m, C.byref(o)), "pt_multi_set_options")

    def setProbe(self, probe):
        if not getattr(probe, "valid", False):
            raise RuntimeError("Probe Data is not valid")  # Probe.h:104-105
        d = np.ascontiguousarray(probe.data, np.float32)
        a = [np.ascontiguousarray(x, np.float32) for x in (probe.pdfValuesX, probe.cdfValuesX, probe.pdfValuesY, probe.cdfValuesY)]
        self._ck(self._L.pt_multi_set_probe(self._m, d.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, probe.width, probe.height), "pt_multi_set_probe")

    def resize(self, newSize, tile=(64, 16)):
        w, h = int(newSize[0]), int(newSize[1])
        self._ck(self._L.pt_multi_resize(self._m, w, h, int(tile[0]), int(tile[1])), "pt_multi_resize")
        if w and h:
            self.launchParams.frame.size = (w, h)

    def setCamera(self, camera: Camera):
        U, V, W = camera.UVWFrame()
        f3 = C.c_float * 3
        self._ck(self._L.pt_multi_set_camera(self._m, C.byref(f3(*[float(x) for x in camera.eye])), C.byref(f3(*[float(x) for x in U])), C.byref(f3(*[float(x) for x in V])), C.byref(f3(*[float(x) for x in W]))), "pt_multi_set_camera")

    def render(self, out: np.ndarray | None = None):
        ptr = out.ctypes.data if out is not None else None
        self._ck(self._L.pt_multi_render(self._m, int(self.launchParams.samples_per_launch), int(self.launchParams.frame.subframe_index), int(self.gather_mask), ptr), "pt_multi_render")

    def renderBatch(self, count: int, out: np.ndarray | None = None):
        ptr = out.ctypes.data if out is not None else None
        self._ck(self._L.pt_multi_render_batch(self._m, int(self.launchParams.samples_per_launch), int(self.launchParams.frame.subframe_index), int(count), int(self.gather_mask), ptr), "pt_multi_render_batch")

    def renderRegions(self, regions, variant=None, out: np.ndarray | None = None):
        arr = (Region * len(regions))()
        for k, g in enumerate(regions):
            for name, _ in Region._fields_:
                setattr(arr[k], name, g[name])
        vp = None
        if variant is not None:
            v = Variant(**variant)
            vp = C.byref(v)
        ptr = out.ctypes.data if out is not None else None
        self._ck(self._L.pt_multi_render_regions(self._m, arr, len(regions), vp, int(self.gather_mask), ptr), "pt_multi_render_regions")

    def renderFoveated(self, c, inner_radius=157, outer_radius=515, spp=(1, 2, 8), out=None, variant=None):
        regs = SampleRenderer.foveatedRegions(self.launchParams.frame.size, c, int(self.launchParams.frame.subframe_index), inner_radius, outer_radius, spp)
        self.renderRegions(regs, variant or SampleRenderer.SV4_VARIANT, out)
        self.launchParams.frame.subframe_index += 1

    def gather(self, which):
        self._ck(self._L.pt_multi_gather(self._m, int(which)), "pt_multi_gather")

    def flush(self, out: np.ndarray | None = None):
        """Overlapped hand-off (frames in flight + gather_mask): the newest frame goes on display now (pt_multi_flush)."""
        ptr = out.ctypes.data if out is not None else None
        self._ck(self._L.pt_multi_flush(self._m, ptr), "pt_multi_flush")

    def downloadDisplay(self, which, rank=0) -> np.ndarray:
        return self.rank(rank).downloadDisplay(which)

    def download(self, which, rank=0) -> np.ndarray:
        return self.rank(rank).download(which)

    def downloadPixels(self) -> np.ndarray:
        return self.download(PT_BUF_FRAME)

    def stats(self) -> dict:
        from ._lib import MultiStats

        s = MultiStats()
        self._ck(self._L.pt_multi_get_stats(self._m, C.byref(s)), "pt_multi_get_stats")
        d = s.sum.as_dict()
        d.update(gather_ms=s.gather_ms, exchange=self.EXCHANGE.get(s.exchange, "?"), ndev=s.ndev, enqueue_ms=s.enqueue_ms, threads=s.threads,
                 frames_handed_over=s.frames_handed_over)
        return d


def make_camera(cam: dict, aspect: float) -> Camera:
    return Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], aspect)
