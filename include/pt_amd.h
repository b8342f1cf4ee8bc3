/* pt_amd.h — C ABI of libptamd.so, the MI355X-native replacement for the reference's
 * optixLaunch hot path (bipul-mohanto/OptixPathTracer, HelloPathtracing_original/).
 *
 * The reference has no FFI: its boundary is the C++ class SampleRenderer
 * (SimplePathtracer.h:38-176) plus the POD headers LaunchParams.h / Material.h /
 * Model.h / Probe.h.  A maintainer keeps that class (see INTEGRATION.md and
 * optixpathtracer_amd/csrc/SampleRenderer.h, a header-only facade with the same
 * public methods) and forwards each method to one entry point below.  Every entry
 * point cites the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success
 * or a negative pt_status; no exception crosses the boundary; pt_last_error() gives
 * the message for the calling context (the reference throws sutil::Exception from
 * CUDA_CHECK/OPTIX_CHECK, sutil/Exception.h:93-195).  A context is single-threaded,
 * like the reference's renderer (one stream, device-synchronised render()).
 *
 * STREAM CONTRACT.  Every kernel and copy of a context runs on streams the context created with
 * hipStreamNonBlocking: they do NOT synchronise with the null stream or with any stream of the
 * caller (the current stream of a tensor framework included).  Entry points that take HOST pointers are complete when
 * they return.  Entry points that take or return DEVICE pointers — pt_pack, pt_unpack, pt_pack_async,
 * pt_unpack_display, pt_render_device, pt_device_buffer, pt_display_buffer — read and write them on
 * pt_stream(ctx), so:
 *   - a buffer the caller PRODUCED on another stream (the receive buffer of an all-gather, a buffer a
 *     memset just cleared) must be complete before the call: synchronise that stream on the host, or
 *     record an event on it and hand it to pt_wait_event(ctx, event) first (device-side ordering);
 *   - a buffer the library produced is complete when the call returned, for the synchronous entry
 *     points (pt_pack, pt_unpack, pt_render_device), and after pt_pack_wait / pt_display_sync for the
 *     asynchronous ones; to consume it on another stream without a host wait, enqueue the consumer
 *     on pt_stream(ctx) or make that stream wait for an event recorded on pt_stream(ctx).
 *
 * VERSIONING.  pt_version() = "ptamd <major>.<minor> ...".  Structs only ever grow at the end; a caller
 * that may meet a newer or older library uses pt_get_stats_n(ctx, &s, sizeof s) (copies the common
 * prefix) instead of pt_get_stats, which writes sizeof(pt_stats) of the LIBRARY's header
 * (pt_stats_size()).  0.2 -> 0.4: pt_stats grew by bvh_builder + reserved_ (8 bytes), pt_multi_stats by
 * enqueue_ms, threads, frames_handed_over.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1, /* bad argument / call order */
    PT_ERR_HIP = -2,     /* a HIP runtime call failed */
    PT_ERR_NO_DEVICE = -3,
    PT_ERR_UNSUPPORTED = -4
};

/* Material.h:47-68 — identical field order and size (104 bytes) */
typedef struct pt_material {
    float emission[3];
    float color[3];
    float absorption[3];
    float eta, metallic, subsurface, specular, roughness, specularTint, anisotropic, sheen, sheenTint, clearcoat,
        clearcoatGloss, transmission;
    float bump;
    float bumpTile[3];
    int32_t flags; /* bit0 = MATERIAL_FLAG_SHADOW_CATCHER (Material.h:9) */
} pt_material;

/* one TriangleMesh (Model.h:10-19): float3 vertices, uint3 indices, one material */
typedef struct pt_mesh_desc {
    const float* vertex; /* num_vertices * 3 */
    uint32_t num_vertices;
    const uint32_t* index; /* num_triangles * 3, local to this mesh */
    uint32_t num_triangles;
    pt_material material;
    int32_t diffuse_texture_id; /* index into pt_scene_desc.textures or -1 (TriangleMesh::diffuseTextureID) */
    const float* texcoord;      /* num_vertices * 2 (TriangleMesh::texcoord) or NULL */
} pt_mesh_desc;

/* Texture (Model.h:21-29): RGBA8 pixels, row 0 first (loadTexture already mirrored them in y, Model.cpp:112-121) */
typedef struct pt_texture_desc {
    const uint32_t* pixel;
    int32_t width, height;
} pt_texture_desc;

/* Model (Model.h:31-42) */
typedef struct pt_scene_desc {
    const pt_mesh_desc* meshes;
    uint32_t num_meshes;
    const pt_texture_desc* textures; /* may be NULL when num_textures == 0 */
    uint32_t num_textures;
} pt_scene_desc;

enum pt_bsdf_mode { PT_BSDF_DISNEY = 0, PT_BSDF_LAMBERT = 1 /* Disney.cuh:125-147 */ };

/* Everything the reference fixes at compile time (SURVEY.md §5 "config / flags") */
typedef struct pt_options {
    int32_t max_depth;      /* the literal 8 in deviceProgram.cu:429 */
    int32_t bsdf_mode;      /* pt_bsdf_mode */
    uint32_t max_paths;     /* paths in flight per wavefront batch (0 = default 8Mi; ≈280 bytes of path state and queues per path: 2.3 GB,
                             * three times that with frames_in_flight = 3, whose three batch sets each hold a whole frame) */
    int32_t kernel_timing;  /* 1: pt_stats.{trace,shadow,shade,other}_ms are measured with a pair of HIP events around every launch
                             *    (costs ≈0.3 ms of a 2 ms frame at a 1/8 share); 0 (default): only render_ms is measured */
    int32_t bvh_kind;       /* reserved, must be 0 (rounds 1-2: 1 = a binary BVH as an A/B path; removed — the 8-wide compressed tree is the structure) */
    int32_t trace_kernel;   /* reserved, must be 0 (rounds 1-2: 1 = the first grid-stride traversal kernel; removed) */
    int32_t streams;        /* pixel chunks of a frame run concurrently on this many stream pairs (0 = default 3, the measured optimum: tails of one chunk overlap the bulk of the others) */
    int32_t split_shadow;   /* 0 = default: shadow rays of bounce b share a launch with the closest-hit rays of b+1; 1 = separate kernels;
                             * 2 = asynchronous: per-bounce shadow records traced on side streams, nothing waits for them before the
                             *     resolve, which sums the visible contributions in bounce order (not with shadow-catcher materials) */
    int32_t frames_in_flight; /* 0 / 1 = default: pt_render returns when its frame is complete, like SampleRenderer::render()
                             * (SimplePathtracer.cpp:96 CUDA_SYNC_CHECK).  2 or 3: pt_render(k) enqueues frame k and returns once at most
                             * frames_in_flight - 1 frames are still running, so the kernel tails of a frame overlap the next frames (same
                             * images, bit for bit).  2: the frame is still cut into one pixel chunk per stream; every chunk stream orders
                             * its own frames and nothing else holds frame k+1 back.  3 (what bench.py times): frame k runs WHOLE on
                             * stream k mod `streams`, three frames overlap and every launch carries three times the rays; only the
                             * resolves, which blend into accum_buffer, are chained from frame to frame.  A single frame followed by a
                             * wait is slower in that mode than the default (one stream, no overlap inside the frame): it is for loops.
                             * Frame k completes — and its errors are reported — at a later pt_render, at pt_sync, or at any call that
                             * reads or changes device state (pt_download, pt_get_stats, pt_device_buffer, pt_resize, ...).
                             * pt_render_regions with 2 or 3: the frame's launches keep their schedule (passes dealt to the streams in turn) but
                             * no longer start together, and the first resolve of a frame waits for the end of the previous frame.  Ignored
                             * (synchronous) with kernel_timing and for a pt_multi_render that hands the frame over (gather_mask != 0 or
                             * host_rgba8); a pt_multi_render with neither keeps the frames in flight on every device. */
} pt_options;

enum pt_buffer {          /* LaunchParams.frame.* (LaunchParams.h:53-63) */
    PT_BUF_ACCUM = 0,     /* float4 accum_buffer */
    PT_BUF_FRAME = 1,     /* uchar4 frame_buffer (make_color) */
    PT_BUF_COLOR = 2,     /* float4 color_buffer */
    PT_BUF_NORMAL = 3,    /* float4 normal_buffer */
    PT_BUF_ALBEDO = 4,    /* float4 albedo_buffer */
    PT_BUF_DENOISED = 5   /* float4 denoisedBuffer (SimplePathtracer.h:149-150); allocated by the first pt_denoise */
};

typedef struct pt_stats {
    uint64_t radiance_rays; /* closest-hit rays traced by the last pt_render */
    uint64_t shadow_rays;   /* any-hit rays traced by the last pt_render */
    uint64_t paths;         /* camera paths started (pt_render_regions: launch indices x samples, before annulus / partition culling) */
    double render_ms;       /* device time of the last pt_render (hipEvent, stream-local) */
    double trace_ms;        /* with pt_options.kernel_timing (else 0) — of which: closest-hit traversal kernels */
    double shadow_ms;       /*           any-hit traversal kernels */
    double shade_ms;        /*           shade kernels */
    double other_ms;        /*           generate / resolve */
    uint32_t trace_launches, shadow_launches, shade_launches;
    uint32_t bvh_nodes;     /* internal nodes of the traversal structure */
    uint64_t bvh_bytes;     /* nodes + leaf triangles resident in HBM */
    double bvh_build_ms;    /* one-time on-GPU build (excluded from render_ms) */
    uint32_t bvh_levels;    /* levels of the traversal structure (must not exceed the traversal stack: pt_create checks) */
    uint64_t shaded_hits;   /* closest hits shaded by the last render whose BSDF sample was accepted (every one cost a BSDFSample, two
                             * BSDFEval and two BSDFPdf: the unit of the shade kernel's FLOP roofline) */
    uint64_t frames;        /* frames completed since pt_create, and the rays they traced: take differences around a */
    uint64_t total_radiance_rays; /* sequence of pt_render calls (pt_get_stats itself waits for the frames in flight) */
    uint64_t total_shadow_rays;
    uint32_t bvh_builder;   /* hierarchy under the wide tree: 0 = LBVH (Morton order), 1 = PLOC, 2 = imported (PT_BVH_IMPORT), 3 = binned SAH (round 5).
                             * pt_create builds the three and keeps the one through which a fixed batch of calibration rays takes fewest traversal
                             * steps; PT_BVH_BUILDER=lbvh|ploc|sah forces one, PT_BVH_SAH=0 leaves the SAH candidate out.  Images do not
                             * depend on the choice (closest hit, lowest primitive on ties, hits confined to the triangle's padded box). */
    uint32_t fused_passes;  /* passes of the last render that ran as ONE persistent kernel (generate -> trace -> shade rounds per wave, no launch
                             * chain: small synchronous frames, PT_FUSED; csrc/pt_fused.h); trace_launches counts each of them once.
                             * WHICH frames: a synchronous pt_render / pt_render_batch of a scene without shadow-catcher materials, default
                             * pt_options.streams and split_shadow, of at most PT_SCHED_MAX_PATHS (4.5 M) paths is rendered alternately as a
                             * launch chain and as one fused pass over its first 2 x (1 + PT_SCHED_TRIALS) frames (both leave the same bits)
                             * and then by whichever measured faster; `schedule`, `sched_chain_ms`, `sched_fused_ms` below report it.  The
                             * first trial frame — and every frame when PT_SCHED_TRIALS=0 — follows the rule of round 5: fused when the frame
                             * has at most PT_FUSED_MAX_PATHS (2.5 M) paths and the tree's calibration rays cost at most PT_FUSED_MAX_COST
                             * (22) traversal steps; a tree that was never calibrated (PT_BVH_BUILDER forced, PT_BVH_IMPORT, a challenger that
                             * could not be built) counts as expensive unless the scene has fewer than 4096 triangles. */
    uint32_t path_state_allocs; /* (re-)allocations of the per-path device state since pt_create.  The state only grows (sets, paths per set,
                             * pixels per set, each kept at the largest value any frame asked for), so alternating schedules — a fused-size
                             * synchronous frame, a foveated frame, frames in flight — re-allocates at most once per dimension. */
    uint32_t bvh_challengers_skipped; /* candidate hierarchies pt_create could not build (out of device memory, a failed bounds check): the standing
                             * tree then stayed without a comparison — also reported on stderr; 0 in every healthy build */
    uint32_t schedule;      /* how the last synchronous pt_render ran: 0 launch chain, 1 fused bounce loop (k_path_loop); bit 8 set while the context
                             * is still timing the two against each other for this frame configuration (sched_chain_ms / sched_fused_ms below; `fused_passes` above says which frames take part) */
    double sched_chain_ms;  /* best device time of the configuration's trial frames as a launch chain / as one fused pass (0: not measured: */
    double sched_fused_ms;  /* the configuration is not eligible for both, or PT_SCHED_TRIALS=0) */
    double create_ms;       /* host time of pt_create from the flattened scene to the finished context: uploads, the acceleration structure
                             * (bvh_build_ms is the part between its first and last kernel), the probes that pick the streams — and, in the first
                             * pt_create of a process, what loading the library's code objects costs beyond the time the scene upload hides */
} pt_stats;

/* SampleRenderer::SampleRenderer(const Model*) (SimplePathtracer.cpp:39-71): uploads the meshes
 * (buildAccel :481-489) and textures (createTextures :603-654; sampled in software: wrap, bilinear, normalised
 * float, no sRGB), builds the acceleration structure ON THE GPU (replaces optixAccelBuild +
 * optixAccelCompact :561-591) and the per-mesh material table (replaces buildSBT :390-455).
 * `device` is the HIP device ordinal.  The scene is deep-copied; the caller keeps its arrays. */
int pt_create(const pt_scene_desc* scene, int device, pt_ctx** out_ctx);

/* no reference counterpart (the reference leaks everything at exit, CUDABuffer.h:32-88) */
int pt_destroy(pt_ctx* ctx);

const char* pt_last_error(const pt_ctx* ctx); /* ctx may be NULL: last error of a failed pt_create */

int pt_set_options(pt_ctx* ctx, const pt_options* opt);
int pt_get_options(const pt_ctx* ctx, pt_options* opt);

/* SampleRenderer::setProbe(const ProbeData&) (SimplePathtracer.cpp:164-180) + CUDAProbeData::createBuffer
 * (Probe.h:102-124): copies the five host arrays to the device. data = w*h float4; pdfX,cdfX = w*h; pdfY,cdfY = h */
int pt_set_probe(pt_ctx* ctx, const float* data_rgba, const float* pdfX, const float* cdfX, const float* pdfY,
                 const float* cdfY, int width, int height);

/* main.cpp:146-156 loadProbe + ProbeData::BuildCDF (Probe.h:29-77) + setProbe in one call, with the CDF built ON THE
 * GPU: one wave per row loads 64 texels at a time and every lane runs the reference's sequential
 * left-to-right float running sum over them (so the arrays are bit-identical to the host BuildCDF; a parallel scan would
 * reassociate the sums); one wave does the same over the row totals.  data = w*h float4. */
int pt_set_probe_image(pt_ctx* ctx, const float* data_rgba, int width, int height);
/* read back the device CDF arrays (pdfX,cdfX: w*h floats; pdfY,cdfY: h floats); any pointer may be NULL */
int pt_get_probe_cdf(pt_ctx* ctx, float* pdfX, float* cdfX, float* pdfY, float* cdfY);

/* ProbeData::BuildCDF (Probe.h:29-77), host side like the reference. Pure function, no context. */
int pt_build_cdf(const float* data_rgba, int width, int height, float* pdfX, float* cdfX, float* pdfY, float* cdfY);

/* SampleRenderer::resize(const int2&) (SimplePathtracer.cpp:109-147): (re)allocates the five frame
 * buffers; a 0-sized request is ignored like the reference (:112). */
int pt_resize(pt_ctx* ctx, int width, int height);

/* SampleRenderer::setCamera (SimplePathtracer.cpp:155-162): eye + the UVW frame of sutil::Camera::UVWFrame */
int pt_set_camera(pt_ctx* ctx, const float eye[3], const float U[3], const float V[3], const float W[3]);

/* sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45), host side. Pure function. */
int pt_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovY_deg, float aspect,
                 float U[3], float V[3], float W[3]);

/* Multi-GPU image partition (no reference counterpart; pattern of sutil/WorkDistribution.h:34-91):
 * the image is cut into tile_w x tile_h tiles and this context renders only tiles with
 * (tile_x + tile_y) % world == rank.  Default is rank 0 of 1.  Must be called before pt_resize
 * or is applied at the next pt_resize. */
int pt_set_partition(pt_ctx* ctx, int rank, int world, int tile_w, int tile_h);

/* SampleRenderer::render() (SimplePathtracer.cpp:73-97): one optixLaunch(w,h,1) equivalent with
 * launchParams.samples_per_launch = spp and launchParams.frame.subframe_index = subframe_index;
 * returns after the device finished (the reference ends render() in cudaDeviceSynchronize, :96).
 * Silently does nothing before the first pt_resize (:77). If host_rgba8 is non-NULL the frame
 * buffer is copied into it (render(CUDAOutputBuffer&) + downloadPixels, :99-107,149-153). */
int pt_render(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, uint32_t* host_rgba8);
/* `count` consecutive launches of the reference's progressive loop — render() with subframe_index = first_subframe, first_subframe + 1, ...
 * (main.cpp:273-278 renders, displays and increments every frame) — as ONE wavefront batch: the generate / traversal / shade launches carry
 * the rays of all `count` subframes (seeds are tea<4>(pixel, subframe_index), deviceProgram.cu:357: subframes are independent until they
 * blend) and the resolve blends them into accum_buffer in subframe order (:460-466).  All five buffers end up bit-identical to `count`
 * calls of pt_render; what changes is the number of rays per launch, which is what a small share of a tile-partitioned frame lacks
 * (a 1/8 share of 1080p x 4 spp is 1 M paths; a persistent traversal wave wants several chunks of work).  The intermediate frames are
 * not displayed: a display loop that shows every frame keeps calling pt_render.  count in [1,4096]; pt_render == count 1.
 * pt_stats.frames advances by count. */
int pt_render_batch(pt_ctx* ctx, uint32_t spp, uint32_t first_subframe, uint32_t count, uint32_t* host_rgba8);
/* SampleRenderer::render(sutil::CUDAOutputBuffer<uint32_t>&) (SimplePathtracer.cpp:99-107): the rgba8 frame lands in a caller-owned
 * DEVICE buffer (width*height*4 bytes on the context's device; the reference aliases the caller's mapped buffer as frame_buffer for
 * the launch).  Synchronous like render(): the buffer is complete when the call returns (also with frames in flight).
 * A pointer HIP does not know (plain malloc'ed host memory) is refused with PT_ERR_INVALID before anything is rendered. */
int pt_render_device(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, void* dev_rgba8);
/* SampleRenderer::stream (SimplePathtracer.h:107, handed to the display path at main.cpp:245): the hipStream_t (as void*) on which the
 * context's packs, unpacks, epilogues and device copies run — see STREAM CONTRACT at the top. */
void* pt_stream(pt_ctx* ctx);
/* Device-side ordering instead of a host wait: everything the context enqueues on pt_stream(ctx) from now on waits for `hip_event`
 * (a hipEvent_t the caller recorded on its own stream after producing a buffer it is about to hand in).  No reference counterpart. */
int pt_wait_event(pt_ctx* ctx, void* hip_event);
/* Waits for the frames in flight (pt_options.frames_in_flight = 2 or 3) and reports their errors; a no-op otherwise.  No reference
 * counterpart: the reference's render() is synchronous. */
int pt_sync(pt_ctx* ctx);

/* Foveated variants (the HelloPathtracing_sv, _sv2, _sv3, _sv4_vmv23 directories; sv and sv2 share one device
 * program and differ from sv3/sv4 by pt_variant.initial_depth / write_aov and their host schedules).  One pt_region = one optixLaunch of the sv4 raygen
 * (HelloPathtracing_sv4_vmv23/deviceProgram.cu:388-590): LaunchParams.frame.{factor,fillSize,c,r_inner,r_outer,offset,
 * redraw} (sv4 LaunchParams.h:62-70) + the launch dimensions, samples_per_launch and subframe_index of that launch. */
typedef struct pt_region {
    uint32_t launch_w, launch_h;  /* optixLaunch width/height */
    uint32_t factor_x, factor_y;  /* frame.factor */
    int32_t fill_size;            /* frame.fillSize: the result is splatted over fill_size^2 pixels */
    uint32_t cx, cy;              /* frame.c: gaze point in pixels */
    float r_inner, r_outer;       /* pixels whose distance to c is outside [r_inner, r_outer] are skipped */
    uint32_t offset_x, offset_y;  /* frame.offset */
    uint32_t redraw;              /* 1: never blend with accum_buffer */
    uint32_t spp;                 /* samples_per_launch */
    uint32_t subframe_index;
} pt_region;

/* what the sv3/sv4 device code changed relative to the canonical variant */
typedef struct pt_variant {
    float radiance_tmin;          /* 0.001 canonical (deviceProgram.cu:420); 0.01 sv4 (global tmin, sv4 :41,485) */
    int32_t cull_back_occlusion;  /* 0 canonical (TERMINATE_ON_FIRST_HIT); 1 sv3/sv4 (CULL_BACK_FACING_TRIANGLES, sv4 :240) */
    int32_t tonemap;              /* 0: make_color(accum); 1: make_color(reinhard(accum * exposure, white)) (sv4 :555-569);
                                   * 2: make_color(accum * exposure) (sv3 :580-604, where the later plain write wins) */
    float exposure;               /* sv4: pow(2,2) = 4; sv3: pow(2,3) = 8 */
    float white;                  /* sv4: 1 */
    int32_t initial_depth;        /* prd.depth at the camera ray: 0 canonical/sv3/sv4; 1 in sv and sv2 (HelloPathtracing_sv/deviceProgram.cu:428,
                                   * with the cutoff `prd.depth >= 3` at :483 = pt_options.max_depth 3): every contribution then goes to
                                   * indirectLight and no first-hit normal/albedo is ever accumulated */
    int32_t write_aov;            /* 1: the launch also writes normal_buffer, color_buffer and albedo_buffer like sv/sv2 (:553-555); 0: accum/frame only (sv3/sv4) */
} pt_variant;

/* SampleRenderer::render() of the foveated variants (HelloPathtracing_sv4_vmv23/SimplePathtracer.cpp:77-216): the
 * given launches in order (later ones overwrite earlier pixels).  Depth cutoff = pt_options.max_depth (sv4: 4).
 * Writes accum_buffer and frame_buffer (plus the three AOV buffers with pt_variant.write_aov).  variant may be NULL (canonical settings). */
int pt_render_regions(pt_ctx* ctx, const pt_region* regions, uint32_t n, const pt_variant* variant, uint32_t* host_rgba8);

/* SampleRenderer::downloadPixels (SimplePathtracer.cpp:149-153), generalised to all five buffers.
 * bytes must equal width*height*(16 or 4). */
int pt_download(pt_ctx* ctx, int which /* pt_buffer */, void* host, size_t bytes);

/* progressive state: overwrite accum_buffer (checkpoint/resume of an accumulation) */
int pt_upload_accum(pt_ctx* ctx, const float* host_rgba, size_t bytes);

/* device pointer of a frame buffer, for zero-copy consumers (display interop in the reference:
 * render(CUDAOutputBuffer&) maps the caller's buffer, SimplePathtracer.cpp:99-107) */
void* pt_device_buffer(pt_ctx* ctx, int which);

/* toneMap.cu:41-70 computeFinalPixelColors: rgba8 = clamp(sqrt(accum))*255.9 into the frame buffer
 * (the reference's disabled alternative epilogue, SimplePathtracer.cpp:105). */
int pt_tonemap_sqrt(pt_ctx* ctx, uint32_t* host_rgba8 /* may be NULL */);

/* The denoiser pass the reference wires up but leaves empty: OptiXDenoiser::{init,exec,finish} (OptixDenoiser.h:12-31,
 * OptixDenoiser.cpp:15-42 — init() has no body) with DenoiseData{width,height,color,albedo,normal,output} set in
 * resize() (SimplePathtracer.cpp:138-146) and the disabled calls "denoiser.exec(); computeFinalPixelColors(size,
 * denoisedBuffer, result)" in render(target) (SimplePathtracer.cpp:104-105).  Here exec() is an edge-avoiding
 * a-trous wavelet filter (Dammertz et al. 2010) guided by the first-hit normal and albedo AOVs the hot path already
 * writes: `iterations` passes of a 5x5 B3-spline kernel with tap spacing 2^i; tap weight =
 * k[dx]*k[dy] * exp(-|c_p-c_q|^2 / (sigma_color*2^-i)^2) * exp(-|n_p-n_q|^2 / (4^i * sigma_normal^2)) *
 * exp(-|a_p-a_q|^2 / sigma_albedo^2); taps outside the image are skipped; alpha is carried through.
 * input: PT_BUF_COLOR (this frame's radiance, what DenoiseData.color points at) or PT_BUF_ACCUM (the progressive
 * average).  epilogue: 0 none, 1 computeFinalPixelColors (toneMap.cu:41-58, as in the disabled call), 2 make_color;
 * 1 and 2 write the rgba8 frame buffer.  Works on the full-size buffers of this context (multi-GPU: after pt_unpack
 * of the three inputs).  The reference has no behaviour to match here: the CPU checker defines the semantics. */
typedef struct pt_denoise_params {
    int32_t iterations;   /* 0..8; 0 copies the input */
    float sigma_color;    /* > 0 */
    float sigma_normal;   /* > 0 */
    float sigma_albedo;   /* > 0 */
    int32_t input;        /* PT_BUF_COLOR or PT_BUF_ACCUM */
    int32_t epilogue;     /* 0, 1, 2 */
} pt_denoise_params;
int pt_denoise(pt_ctx* ctx, const pt_denoise_params* params, uint32_t* host_rgba8 /* may be NULL */, double* kernel_ms /* may be NULL */);

/* Multi-GPU exchange helpers. pt_owned_pixels = number of pixels this rank renders (padded count is
 * the same on every rank: pt_owned_pixels_padded). pt_pack packs this rank's pixels of buffer `which`
 * into dev_dst (padded count * elem bytes); pt_unpack scatters the concatenation of all ranks' packs
 * (world * padded * elem bytes, rank-major — exactly what an RCCL all-gather produces) into the
 * full-size local buffer `which`. */
int pt_owned_pixels(const pt_ctx* ctx, uint32_t* owned, uint32_t* padded);
int pt_pack(pt_ctx* ctx, int which, void* dev_dst);
int pt_unpack(pt_ctx* ctx, int which, const void* dev_src_all);

/* Display hand-off that overlaps the next frame (the reference's loop renders, then displays, every frame: main.cpp:273-278; on several
 * GPUs "display" is preceded by the exchange of the ranks' strips).  With pt_options.frames_in_flight = 2 or 3 the loop is
 *     pt_render(k);  pt_pack_async(FRAME, send[k & 1], k & 1);
 *     if (k > 0) { pt_pack_wait((k - 1) & 1);  all-gather(recv, send[(k - 1) & 1]);  pt_unpack_display(FRAME, recv);  show pt_display_buffer(FRAME); }
 * pt_pack_async enqueues the pack of buffer `which` behind the newest frame in flight WITHOUT waiting for it (pt_pack waits) and makes
 * the resolves of later frames wait for it, so the strip is frame k's whatever is enqueued next; pt_pack_wait blocks the host until that
 * strip is complete (frames enqueued after it keep running); pt_unpack_display scatters the all-gathered strips into the DISPLAY copy
 * of the buffer — a second, full-size buffer no render writes (allocated on first use) — so the exchange and display of frame k-1
 * overlap the rendering of frame k and never show pixels of two frames.  pt_display_sync waits for the newest pt_unpack_display,
 * pt_download_display copies a display buffer to the host.  Same bits as pt_pack / pt_unpack. */
int pt_pack_async(pt_ctx* ctx, int which, void* dev_dst, int slot /* 0 or 1: which of the caller's two send buffers */);
int pt_pack_wait(pt_ctx* ctx, int slot);
int pt_unpack_display(pt_ctx* ctx, int which, const void* dev_src_all);
int pt_display_sync(pt_ctx* ctx);
void* pt_display_buffer(pt_ctx* ctx, int which); /* NULL until the first pt_unpack_display of that buffer */
int pt_download_display(pt_ctx* ctx, int which, void* host, size_t bytes);

int pt_get_stats(const pt_ctx* ctx, pt_stats* out);
/* size-checked variant (see VERSIONING): copies min(out_bytes, pt_stats_size()) bytes, zero-fills the rest */
int pt_get_stats_n(const pt_ctx* ctx, void* out, size_t out_bytes);
size_t pt_stats_size(void);

/* ---------------------------------------------------------------------------------------------------------------------
 * Several GPUs from ONE process (SURVEY.md 8b "pt_create_multi"; the reference renders on whatever single device is
 * current, SimplePathtracer.cpp:203-212).  A pt_multi is ndev contexts — rank r on HIP device devices[r] — that share
 * one flattened scene (uploaded to every device, acceleration structure built on each), the same probe / camera /
 * options, and one frame cut into interleaved tile_w x tile_h tiles (pt_set_partition with world = ndev, the pattern of
 * sutil/WorkDistribution.h:34-91).  pt_multi_render enqueues the frame on every device before it waits for any, so
 * the devices work concurrently from one host thread; no collective is on the data path.  The display hand-off,
 * pt_multi_gather, makes a buffer complete on EVERY rank: pack of the owned pixels -> one all-gather of the packed
 * strips -> unpack.  The all-gather is RCCL's ncclAllGather, one communicator per rank inside one ncclGroup
 * (librccl is opened on first use); when two ranks share a device (rehearsal on a one-GPU box: RCCL refuses duplicate
 * devices) or librccl is absent, the strips travel as direct device-to-device copies (hipMemcpyPeerAsync over xGMI).
 * devices may repeat.  Every call returns a pt_status; pt_multi_last_error gives the message. */
typedef struct pt_multi pt_multi;
enum pt_exchange { PT_EXCHANGE_NONE = 0, PT_EXCHANGE_RCCL = 1, PT_EXCHANGE_PEER_COPY = 2 };
typedef struct pt_multi_stats {
    pt_stats sum;            /* rays / paths / frames summed over the ranks (frames = frames x ranks); the *_ms fields are the MAXIMUM over the ranks */
    double gather_ms;        /* wall time of the last pt_multi_gather (pack + exchange + unpack, all ranks); overlapped hand-off: host time of the exchange of the previous frame */
    int32_t exchange;        /* pt_exchange used by the last gather */
    int32_t ndev;
    double enqueue_ms;       /* host time of the enqueue phase of the last render (all launches of the frame on every device): the slowest
                              * rank's when the ranks have their own threads, the sum over the ranks otherwise */
    int32_t threads;         /* host threads that enqueue the ranks (0: the calling thread does it for all — one rank, or PT_MULTI_THREADS=0) */
    uint64_t frames_handed_over; /* frames that went through the overlapped hand-off (frames in flight + gather_mask) */
} pt_multi_stats;
int pt_create_multi(const pt_scene_desc* scene, const int* devices, int ndev, pt_multi** out);
int pt_multi_destroy(pt_multi* m);
const char* pt_multi_last_error(const pt_multi* m); /* m may be NULL: last error of a failed pt_create_multi */
int pt_multi_size(const pt_multi* m);
pt_ctx* pt_multi_ctx(pt_multi* m, int rank); /* the rank's context, for per-rank calls (pt_download, pt_get_stats, pt_device_buffer) */
int pt_multi_set_options(pt_multi* m, const pt_options* opt);
int pt_multi_set_probe(pt_multi* m, const float* data_rgba, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY, int width, int height);
int pt_multi_set_probe_image(pt_multi* m, const float* data_rgba, int width, int height);
int pt_multi_resize(pt_multi* m, int width, int height, int tile_w, int tile_h); /* tile sizes: multiples of 8; 0 = 64 x 16 */
int pt_multi_set_camera(pt_multi* m, const float eye[3], const float U[3], const float V[3], const float W[3]);
/* gather_mask: bit (1 << pt_buffer) for every buffer to assemble on all ranks after the frame (0 = none: pure throughput);
 * host_rgba8 (may be NULL) receives rank 0's frame buffer and implies gathering PT_BUF_FRAME.
 * Every rank's launches are enqueued by a host thread of its own (pt_multi_stats.enqueue_ms, .threads), so the host time of a frame does
 * not grow with the number of devices.
 * With pt_options.frames_in_flight = 2 or 3 AND something to hand over, the hand-over overlaps the next frame: the call enqueues frame k,
 * then exchanges the strips frame k-1 packed behind its last kernel and scatters them into the ranks' DISPLAY buffers (pt_display_buffer /
 * pt_download_display of pt_multi_ctx(m, r); the ordinary buffers keep only the rank's own pixels) while frame k renders, and returns
 * when frame k-1 is on display: host_rgba8 receives frame k-1 (nothing on the first call), pt_multi_flush hands over the last frame.
 * Same bits as the synchronous hand-over. */
int pt_multi_render(pt_multi* m, uint32_t spp, uint32_t subframe_index, uint32_t gather_mask, uint32_t* host_rgba8);
/* pt_render_batch on every rank: `count` subframes in one wavefront batch per device, then the hand-over of the last one */
int pt_multi_render_batch(pt_multi* m, uint32_t spp, uint32_t first_subframe, uint32_t count, uint32_t gather_mask, uint32_t* host_rgba8);
int pt_multi_render_regions(pt_multi* m, const pt_region* regions, uint32_t n, const pt_variant* variant, uint32_t gather_mask, uint32_t* host_rgba8);
int pt_multi_gather(pt_multi* m, int which /* pt_buffer */);
int pt_multi_flush(pt_multi* m, uint32_t* host_rgba8 /* may be NULL */); /* overlapped hand-off: the newest frame goes on display now */
int pt_multi_get_stats(const pt_multi* m, pt_multi_stats* out);

/* Ray-search entry (what optixTrace did: deviceProgram.cu:165,190).  rays = n * 8 floats
 * (o.xyz, tmin, d.xyz, tmax) in HOST memory.  any_hit=0: t_out[n], prim_out[n] (global triangle
 * index in mesh order, -1 = miss; t_out = tmax on miss).  any_hit=1: prim_out[n] = 1 occluded / 0.
 * iters>1 repeats the kernel for timing; kernel_ms (may be NULL) receives the mean kernel time. */
int pt_trace(pt_ctx* ctx, const float* rays, uint32_t n, int any_hit, float* t_out, int32_t* prim_out, int iters,
             double* kernel_ms);

/* The acceleration structure as the traversal kernels see it, copied to host memory — for inspection, for a host-side
 * traversal of the SAME tree (bench.py's CPU baseline, tests) or for serialisation.  Call with nodes == tris == NULL to get the
 * counts.  nodes: num_nodes x 80 bytes, node 0 = root, breadth-first by level (20 little-endian 32-bit words each):
 *   [0..2] origin.xyz (f32) | [3] upper 16 bits of the f32 grid steps sx (low half) and sy (high half) | [4] child_base |
 *   [5] tri_base | [6] leafbits | [7] upper 16 bits of sz (low half), imask (high half) |
 *   [8,9] qlo.x[8] | [10,11] qlo.y[8] | [12,13] qlo.z[8] | [14,15] qhi.x[8] | [16,17] qhi.y[8] | [18,19] qhi.z[8]  (one byte per child slot)
 * child box s = origin + q * step per axis (rounded outward at build; an unused slot has qlo 255 > qhi 0);
 * internal child s = node child_base + popcount(imask & ((1 << s) - 1)); leafbits bit 3s+k: slot s holds more than k
 * triangles, triangle (s,k) = tris[tri_base + popcount(leafbits & ((1 << (3s+k)) - 1))].
 * tris: num_tris x 48 bytes = 12 f32: v0.xyz, v1.xyz, v2.xyz, the global primitive index (i32 bits), the mesh index (i32 bits; = the material
 * record: k_shade reads these very triangles through the leaf index a closest-hit record holds), 1 unused. */
int pt_export_bvh(pt_ctx* ctx, void* nodes, size_t nodes_bytes, void* tris, size_t tris_bytes, uint32_t* num_nodes, uint32_t* num_tris);

/* Scene ingestion, host only (no GPU needed): loadOBJ (HelloPathtracing_original/Model.cpp:137-212 — tinyobjloader 2.0.0's LoadObj with
 * triangulation, then one TriangleMesh per (shape, material id)) as native code.  The arrays are the reference's bit for bit
 * (tests/test_objloader.py: the reference's own Model.cpp, compiled from where it lies, on committed fixtures and random files):
 * vertices, normals (null when the mesh has none), texcoords (null when none), indices, the Material bytes (Kd -> color, Ke ->
 * emission, defaults otherwise), mesh order.
 * per_mesh_vertex_map = 0 reproduces the reference's ONE knownVertices map per shape shared by its materials (Model.cpp:176): a
 * second material's mesh then indexes vertices it does not own — out of bounds for the renderer (pt_create refuses such a mesh);
 * per_mesh_vertex_map = 1 gives every mesh its own map: what a renderer needs, and the default of the Python facade's
 * load_obj.  Images are not decoded here (the reference uses stb_image): texture_ref numbers a mesh's texture REFERENCE
 * — (shape, file name) pairs in loadTexture's order of first use (Model.cpp:88-135, :177), path = model directory + "/" + name with
 * '\\' -> '/' — or is -1; the caller decodes pt_obj_texture_path(i) (RGBA8, rows mirrored in y), gives unreadable files the id -1
 * and numbers the others in order, as loadTexture does.  Errors: PT_ERR_INVALID, text in pt_obj_last_error() (thread-local). */
typedef struct pt_obj pt_obj;
typedef struct pt_obj_mesh {
    const float* vertex;   /* num_vertices * 3 */
    const float* normal;   /* num_vertices * 3 or NULL */
    const float* texcoord; /* num_vertices * 2 or NULL */
    const uint32_t* index; /* num_triangles * 3 */
    uint32_t num_vertices, num_triangles;
    pt_material material;
    int32_t texture_ref; /* index for pt_obj_texture_path, or -1 */
} pt_obj_mesh;
int pt_load_obj(const char* obj_path, int per_mesh_vertex_map, pt_obj** out);
void pt_obj_free(pt_obj* obj);
uint32_t pt_obj_num_meshes(const pt_obj* obj);
int pt_obj_get_mesh(const pt_obj* obj, uint32_t i, pt_obj_mesh* out); /* the pointers live until pt_obj_free */
uint32_t pt_obj_num_textures(const pt_obj* obj);
const char* pt_obj_texture_path(const pt_obj* obj, uint32_t i);
const char* pt_obj_last_error(void);

/* Device-function tables for function-level parity tests (the reference's commented-out BSDFTest /
 * ProbeCreateTest, Disney.cuh:430-503, Probe.cuh:207-269, turned into entry points).
 *  which = 0: BSDFEval+BSDFPdf  in: n x {N[3],V[3],L[3],etaI,etaO} (11 floats)  out: n x {f[3],pdf}
 *  which = 1: BasisFromVector+BSDFSample  in: n x {N[3],V[3],etaI,etaO,seed(as u32 bits)} (9)  out: n x {L[3],pdf,seed1,seed2 (bits)}
 *  which = 2: ProbeSample  in: n x {seed bits} (1)  out: n x {dir[3],color[3],pdf,seed1,seed2} (9)
 *  which = 3: ProbeEval(ProbeDirToUV(dir)) in: n x dir[3] out: n x {u,v,r,g,b,a} (6)
 *  which = 4: make_color in: n x rgb[3] out: n x {packed bits} (1)
 *  which = 5: detmath in: n x {fn, x, y} (3) out: n x 1   fn: 0 sin 1 cos 2 acos 3 atan2(x,y) 4 log 5 pow(x,y) 6 x/y 7 sqrt(x)
 *  which = 6: tea4/lcg/Random in: n x {a bits, b bits} out: n x {tea4(a,b), lcg state, rnd, Randf bits...} (8)
 *  which = 7: tex2D of scene texture 0 in: n x {s,t} (2) out: n x rgba (4)
 *  which = 8: ProbePdf (Probe.cuh:69-93; unused by the reference's device code) in: n x dir[3] out: n x pdf (1)
 * material applies to which 0,1; the context's probe to 2,3,8. All arrays are host memory. */
int pt_eval_table(pt_ctx* ctx, int which, const pt_material* material, int bsdf_mode, const float* in, uint32_t n,
                  float* out);

const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
