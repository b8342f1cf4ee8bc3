// pt_api.hip — the C ABI of libptamd.so (include/pt_amd.h): context, device memory, and the
// per-frame wavefront schedule that replaces SampleRenderer::render()'s single optixLaunch
// (SimplePathtracer.cpp:73-97).
#include <hip/hip_runtime.h>
#include <math.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "pt_host.h"
#include "pt_kernels.h"
#include "pt_bvh8.h"
#include "pt_fused.h"

static thread_local std::string g_create_error;

struct LaunchCounts {
    uint32_t trace = 0, shadow = 0, shade = 0, fused = 0;
};

// A host thread that enqueues on behalf of a context (pt_multi: one per rank; a synchronous frame: one per pixel chunk).  A frame is 21
// launches per chunk; from one thread the second and third chunk's chains start a third and two thirds of the enqueue time late.
struct EnqueueWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    double ms = 0; // host time of the last job
};
static void enqueue_worker_main(EnqueueWorker* w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = job();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lk.lock();
        w->rc = rc;
        w->ms = ms;
        w->done = true;
        w->cv.notify_all();
    }
}
static void enqueue_worker_post(EnqueueWorker* w, std::function<int()> job) {
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = std::move(job);
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
}
static int enqueue_worker_wait(EnqueueWorker* w) {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    return w->rc;
}
static void enqueue_worker_stop(EnqueueWorker* w) {
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->quit = true;
        w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();
}

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::string launch_err; // first failed kernel launch of the frame being enqueued (SpanGuard), reported when the frame is waited for
    std::mutex launch_err_mu; // (the chunks of a synchronous frame are enqueued by several threads)
    std::vector<std::unique_ptr<EnqueueWorker>> chunk_workers; // created on first use: PT_ENQUEUE_THREADS=0 keeps one thread
    pt_options opt{};
    // scene
    uint32_t ntri = 0, nmesh = 0;
    float* d_verts = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t* d_tri_mesh = nullptr;
    pt_material* d_mats = nullptr;
    bool has_catcher = false;
    // textures
    int32_t* d_mesh_tex = nullptr;
    PrimUV* d_uvs = nullptr;   // per primitive, only until the leaf-ordered records are emitted
    TexTri* d_textris = nullptr;
    DevTex* d_textures = nullptr;
    std::vector<uint32_t*> d_tex_pixels;
    DevTex tex0{};
    PtBvh bvh;
    double bvh_build_ms = 0;
    double create_ms = 0; // host time of the whole pt_create after the scene was flattened: uploads, code-object loads of a first use, the build, the stream probes
    // probe
    DevProbe probe{};
    float4* d_probe_data = nullptr;
    float *d_pdfX = nullptr, *d_cdfX = nullptr, *d_pdfY = nullptr, *d_cdfY = nullptr, *d_c64Y = nullptr, *d_c8Y = nullptr;
    float4* d_tri_nrm = nullptr; // per leaf triangle of the traversal structure: (geometric normal, mesh) for k_shade
    ProbeLine* d_lines = nullptr; // ProbeSample's column tables: six columns' cdf + (rgb, pdfX) per 128-byte line
    uint16_t* d_guide = nullptr;
    // frame
    int width = 0, height = 0;
    float4 *accum = nullptr, *color = nullptr, *normal = nullptr, *albedo = nullptr;
    uint32_t* frame = nullptr;
    float4 *denoised = nullptr, *denoise_tmp = nullptr; // allocated by the first pt_denoise at the current size
    // Display hand-off that overlaps the next frame (pt_pack_async / pt_unpack_display): the second set of frame buffers.  A frame's
    // strips are packed behind its last kernel on the context's stream, later frames' resolves wait for that pack before they
    // overwrite the frame buffers, and the gathered strips of all ranks are scattered into `display[which]`, which no render writes.
    void* display[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_packed[2] = {nullptr, nullptr}; // behind the pack of hand-off slot 0 / 1
    hipEvent_t ev_displayed = nullptr;            // behind the newest pt_unpack_display
    hipEvent_t ev_pack_guard = nullptr;           // the newest pack still to be waited for by later resolves (null: none)
    v3 eye{0, 0, 0}, U{1, 0, 0}, V{0, 1, 0}, W{0, 0, 1};
    // partition
    int rank = 0, world = 1, tile_w = 64, tile_h = 16;
    uint32_t owned = 0, padded = 0;
    uint32_t* d_pixels = nullptr;     // this rank's pixel list
    uint32_t* d_all_pixels = nullptr; // world * padded, rank-major (0xffffffff = pad)
    // path state: NSETS independent batch sets; pixel chunks of one frame are dealt round-robin to the sets and each
    // set runs on its own pair of streams, so one chunk's kernel tails overlap with the other chunks' bulk work
    struct BatchSet {
        hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
        hipStream_t shade_stream = nullptr; // PT_SHADE_CUS experiment: k_shade runs here (every CU) while `stream` is confined to the other CUs
        PathState st{}; // the arrays indexed by path slot; the stream pointers are filled per launch (stream_view)
        // what a path carries from bounce to bounce, in queue order (pt_kernels.h PathState): X[0] travels with queueA, X[1] with queueB
        // (the generate kernels write X[1]); the shadow records travel with squeue
        struct StreamBuf {
            float4 *rayO = nullptr, *rayD = nullptr, *thr = nullptr;
            uint4* rf = nullptr;
            float2* hit = nullptr;
        } X[2];
        float4 *shO = nullptr, *shD = nullptr, *shPend = nullptr;
        uint32_t *queueA = nullptr, *queueB = nullptr, *squeue = nullptr;
        uint32_t* squeueB = nullptr; // asynchronous shadow rays: one shadow queue per bounce
        uint32_t* ovf3 = nullptr;
        uint32_t* counters = nullptr; // [nq][PT_NSUB*PT_CSTRIDE] radiance sub-queue counts, same for shadow, then 2*nq work counters of PT_WSTRIDE words
        uint32_t *ovf = nullptr, *ovf2 = nullptr;
        float4 *pixResult = nullptr, *pixAlpha = nullptr, *pixNormal = nullptr, *pixAlbedo = nullptr;
        unsigned long long* totals = nullptr; // this set's ray counters of the frame being enqueued (a slice of d_totals)
    };
    std::vector<BatchSet> sets;
    // The streams of the batch sets live as long as the context: path-state re-allocations keep them.  HIP maps streams onto four
    // hardware queues in creation order, and which set streams come to share a queue decides whether frames overlap — measured: after a
    // re-allocation had destroyed and re-created the streams, three frames in flight ran at 9.1 instead of 8.2 ms; with a creation order
    // that put two set streams on one queue a synchronous frame took 10.1 instead of 9.0 ms and a 1/8 share 2.57 instead of 1.77.  So the
    // four streams that carry a frame are chosen by probing (pick_streams) and never re-created.
    std::vector<hipStream_t> set_streams, side_streams; // [set], [2 * set + {0,1}]
    bool streams_probed = false; // pick_streams ran: the context's stream and the first three set streams sit on different hardware queues
    uint32_t set_cap = 0, set_pix_cap = 0, sub_cap = 0;
    uint32_t path_state_allocs = 0; // (re-)allocations of the path state so far (pt_stats.path_state_allocs)
    bool cap_catcher = false, cap_async = false;
    int nq = 0;
    // [frame slot 0..2][batch set 0..15][4]: [0] radiance rays, [1] shadow rays, [2] low word = traversal fault bits (pt_bvh8.h push), [3] shaded hits
    unsigned long long* d_totals = nullptr;
    unsigned long long* h_totals = nullptr; // pinned host copy, same shape: filled by an asynchronous copy behind each frame (a blocking hipMemcpy would wait for the NEXT frame too)
    uint32_t* ovf = nullptr; // spill stacks for pt_trace queries
    unsigned long long* dbg = nullptr; // PT_DEBUG_COUNTS: traversal step counters of the last frame
    int trace_grid = 0;
    int num_cus = 0;
    bool cam_packets = true; // camera rays as packets (PT_CAM_PACKETS=0 turns it off)
    uint64_t cam_min_paths = 1u << 19; // ... for launches of at least this many camera rays (PT_CAM_MIN_PATHS): a packet is one wave's work from start to end,
                                       // so a launch of a few thousand packets is as long as its longest packet (chunks of a 1/8 share of C3, 345 k rays: 1.93 against 1.84 ms per frame; 1/4 share, 690 k: 2.88 against 2.92)
    int shade_cus = 0;       // PT_SHADE_CUS (experiment): CUs the chunk chains' streams may NOT use; k_shade launches go to unmasked streams
    std::vector<hipStream_t> masked_streams; // [set]: the set's stream when shade_cus > 0 (set_streams[set] then carries its k_shade launches)
    int cam_grid = 0;        // PT_CAM_GRID (tuning hook): waves of a packet launch, 0 = the policy of launch_closest
    // the bounce loop of a small frame as ONE persistent kernel (pt_fused.h): PT_FUSED=0 never / 1 (default) synchronous frames of at most
    // fused_max_paths paths (PT_FUSED_MAX_PATHS), as one pass / 2 every pass the kernel covers, chunked as usual (tests); fused_cap: window entries per wave (PT_FUSED_CAP, a multiple of 64; default: 128 or 64 by frame size)
    int fused = 1;
    uint64_t fused_max_paths = 2500000;
    uint32_t fused_cap = 0; // 0: by frame size (enqueue_chunk)
    float fused_max_cost = 22.f; // PT_FUSED_MAX_COST: fused_one_pass only for trees whose calibration rays cost at most this many steps (see render_enqueue)
    bool fused_frame = false; // the frame being enqueued is one fused pass (render_enqueue)
    // Which of the two schedules a synchronous frame takes is MEASURED (round 6): both leave the same bits, so for every frame configuration
    // (pixels owned x samples, depth limit, BSDF mode, size, partition) the context renders the first frames alternately as a launch chain and as
    // one fused pass — the first frame of each is warm-up, then `sched_trials` timed frames each (device time between the frame's begin and end
    // events) — keeps the faster and re-decides when the configuration changes (pt_resize, pt_set_partition, samples, options).  Every
    // `sched_probe` frames the loser gets one frame; if that beats the winner's running mean by more than 5 % (the camera moved into a part of
    // the scene with different rays) the trial starts over.  The thresholds of round 5 (fused_max_paths, fused_max_cost) only pick which schedule
    // the first trial frame uses — and what a context with PT_SCHED_TRIALS=0 does.  Frames of up to sched_max_paths paths take part
    // (above, a fused pass never won: DESIGN.md §6).  pt_stats.schedule / sched_chain_ms / sched_fused_ms report it.
    int sched_trials = 3;               // PT_SCHED_TRIALS (0: no measurement, thresholds only)
    int sched_probe = 64;               // PT_SCHED_PROBE (0: never look at the loser again)
    uint64_t sched_max_paths = 4500000; // PT_SCHED_MAX_PATHS
    int sched_initial = -1;             // PT_SCHED_INITIAL=chain|fused (test hook): the schedule of the first trial frame
    double sched_fake[2] = {0, 0};      // PT_SCHED_FAKE="chain_ms,fused_ms" (test hook): these times instead of the measured ones
    struct SchedKey {
        uint32_t owned = 0, vspp = 0, max_paths = 0;
        int max_depth = 0, bsdf = 0, w = 0, h = 0, rank = 0, world = 0;
        bool operator==(const SchedKey& o) const {
            return owned == o.owned && vspp == o.vspp && max_paths == o.max_paths && max_depth == o.max_depth && bsdf == o.bsdf && w == o.w && h == o.h && rank == o.rank && world == o.world;
        }
    };
    struct Sched {
        SchedKey key;
        int n[2] = {0, 0};         // frames rendered as [0] launch chain / [1] fused pass during the trial (the first of each is warm-up)
        double best[2] = {0, 0};   // fastest timed trial frame of each
        int choice = -1;           // -1 while the trial runs
        uint64_t since = 0;        // frames since the choice
        double mean = 0;           // running mean of the chosen schedule's frame time
    } sched;
    struct { bool valid = false; int which = 0; bool probe = false; } sched_pending; // the synchronous frame in flight, if it takes part
    uint32_t sched_flags = 0;      // pt_stats.schedule of the last synchronous frame
    int fused_grid = 0; // PT_FUSED_GRID: waves of the fused kernel (0: the traversal grid)
    int enqueue_threads = 1; // PT_ENQUEUE_THREADS: 0 one enqueue thread, 1 one thread per pixel chunk for small synchronous frames (default), 2 at every size
    bool adapt_grid = false; // set around the enqueue of a whole frame (frames_in_flight = 3)
    int trace_grid_min = 2048, grid_chunks = 6; // PT_GRID_MIN / PT_GRID_CHUNKS (tuning hooks): smallest persistent grid, chunks of 64 paths per wave aimed at
    int lds_skip = 0; // PT_STACK_LDS_SKIP (test hook, pt_bvh8.h)
    int ovf_depth = 0; // spill levels of the traversal stack (PT8_OVF_DEPTH; test hook PT_STACK_CAP lowers the total capacity)
    int stack_check = 1; // PT_STACK_NOCHECK=1 (test hook): skip the build-time depth check so that the in-kernel fault flag is reached
    // stats + timing
    pt_stats stats{};
    std::vector<hipEvent_t> ev_pools[3]; // one event pool per frame slot
    size_t ev_used = 0;
    struct Span { size_t a, b; int cls; };
    std::vector<Span> spans;
    // Frames that are enqueued but not yet waited for (render_enqueue / render_finish).  Synchronous rendering uses slot 0
    // only.  pt_options.frames_in_flight = 2: consecutive pt_render calls alternate between two slots and each call waits for
    // the PREVIOUS frame, so the kernel tails of frame k overlap the start of frame k+1 (every batch set's stream orders its own
    // chunks; the sets share nothing else but the frame's counters, kept per slot and per set).  frames_in_flight = 3: a frame is
    // no longer cut into one pixel chunk per stream — frame k runs whole on stream k mod 3, three frames overlap, and a launch
    // carries three times the rays (only the resolves, which blend into accum_buffer, are chained from frame to frame).
    struct Inflight {
        int active = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        uint64_t paths = 0, seq = 0;
        uint32_t subframes = 1; // subframes the slot's launch chain completes (pt_render_batch)
        LaunchCounts lc;
    };
    Inflight fr[3];
    hipEvent_t ev_resolved = nullptr; // behind the last k_resolve of the newest frame in flight (the next frame's resolves wait for it where stream order alone does not order them)
    int resolved_kind = 0;            // what recorded ev_resolved: 0 pt_render with pixel chunks (mode 2), 1 whole frames (mode 3), 2 pt_render_regions
    int last_slot = -1;
    int cur_slot = 0;      // slot whose events / counters the enqueue functions are filling
    uint64_t frame_seq = 0;
    uint64_t cum_radiance = 0, cum_shadow = 0, cum_frames = 0;
    int env_timing = -1; // PT_TIMING=0/1 overrides pt_options.kernel_timing
    bool span_timing() const { return env_timing >= 0 ? env_timing != 0 : opt.kernel_timing != 0; }
};

#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return PT_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

static int fail(pt_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

template <typename T>
static hipError_t dalloc(T** p, size_t n) {
    return hipMalloc((void**)p, sizeof(T) * (n ? n : 1));
}
// The library's streams are created hipStreamNonBlocking: they neither wait for nor hold up the null stream — an application (or torch)
// working on the default stream next to the renderer would otherwise serialise with every frame in flight, and an exchange of frame
// k-1 could not overlap the rendering of frame k.  So nothing here may rely on the null stream's implicit ordering: clears go through
// the context's stream and are waited for.
static hipError_t dclear(hipStream_t stream, void* p, size_t bytes) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
    return e == hipSuccess ? hipStreamSynchronize(stream) : e;
}
static hipError_t stream_create(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
template <typename T>
static void dfree(T*& p) {
    if (p) hipFree((void*)p);
    p = nullptr;
}

// frees temporaries of the query entry points on every exit path
struct DevScope {
    std::vector<void*> ptrs;
    std::vector<hipEvent_t> events;
    template <typename T>
    hipError_t alloc(T** p, size_t n) {
        hipError_t e = dalloc(p, n);
        if (e == hipSuccess) ptrs.push_back((void*)*p);
        return e;
    }
    hipError_t event(hipEvent_t* ev) {
        hipError_t e = hipEventCreate(ev);
        if (e == hipSuccess) events.push_back(*ev);
        return e;
    }
    ~DevScope() {
        for (void* p : ptrs) hipFree(p);
        for (hipEvent_t e : events) hipEventDestroy(e);
    }
};

extern "C" const char* pt_version(void) { return "ptamd 0.4 (gfx950 wavefront path tracer)"; }

extern "C" const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static void default_options(pt_options* o) {
    memset(o, 0, sizeof(*o));
    o->max_depth = 8;
    o->bsdf_mode = PT_BSDF_DISNEY;
    o->max_paths = 8u << 20;
}

// words of a batch set's counter block, zeroed at the start of every pass: [nq][PT_NSUB * PT_CSTRIDE] radiance sub-queue counts, the same
// for the shadow queues, 2 nq chunk counters of the traversal launches (PT_WSTRIDE words apart)
static size_t counter_words(int nq) {
    return (size_t)2 * nq * PT_NSUB * PT_CSTRIDE + (size_t)2 * nq * PT_WSTRIDE;
}
static size_t ovf_words(const pt_ctx* ctx);
static const int PT_MAX_SETS = 16;
static const int PT_MAX_FRAMES = 3;
static unsigned long long* totals_of(pt_ctx* ctx, int slot, int set) { return ctx->d_totals + ((size_t)slot * PT_MAX_SETS + set) * 4; }
static uint32_t* fault_word(pt_ctx* ctx) { return reinterpret_cast<uint32_t*>(totals_of(ctx, 0, 0) + 2); } // pt_trace queries
static uint32_t* fault_word(pt_ctx::BatchSet& bs) { return reinterpret_cast<uint32_t*>(bs.totals + 2); }
static int drain(pt_ctx* ctx);
static int stack_capacity8(const pt_ctx* ctx) { return (PT8_LDS_DEPTH - ctx->lds_skip) + ctx->ovf_depth; }

// The traversal stack is finite (k_trace8: one pushed group per level of the wide tree).  A tree deeper than the stack is refused
// here, loudly, instead of being traversed with dropped entries.
static int check_tree_depth(pt_ctx* ctx, char* msg, size_t msg_len) {
    if (!ctx->stack_check) return PT_OK;
    if (ctx->bvh.levels8 > stack_capacity8(ctx)) {
        snprintf(msg, msg_len, "acceleration structure has %d levels, the traversal stack holds %d", ctx->bvh.levels8, stack_capacity8(ctx));
        return PT_ERR_UNSUPPORTED;
    }
    return PT_OK;
}

// The scene in one global vertex / index space, flattened once on the host like buildAccel (SimplePathtracer.cpp:481-489);
// pt_create_multi uploads the same flat scene to every device.
struct FlatScene {
    std::vector<float> verts, tc;
    std::vector<uint32_t> idx, tri_mesh;
    std::vector<pt_material> mats;
    std::vector<int32_t> mesh_tex;
    bool any_tex = false, has_catcher = false;
    size_t nv = 0, nt = 0;
    const pt_scene_desc* scene = nullptr; // textures are read from the caller's arrays
};

static int flatten_scene(const pt_scene_desc* scene, FlatScene& fs) {
    if (!scene || scene->num_meshes == 0 || !scene->meshes) return fail(nullptr, PT_ERR_INVALID, "pt_create: null or empty scene");
    size_t nv = 0, nt = 0;
    for (uint32_t m = 0; m < scene->num_meshes; ++m) {
        const pt_mesh_desc& md = scene->meshes[m];
        if (!md.vertex || !md.index || md.num_triangles == 0) return fail(nullptr, PT_ERR_INVALID, "pt_create: empty mesh");
        if (md.diffuse_texture_id >= (int32_t)scene->num_textures) return fail(nullptr, PT_ERR_INVALID, "pt_create: diffuse_texture_id out of range");
        for (size_t k = 0; k < 3 * (size_t)md.num_triangles; ++k)
            if (md.index[k] >= md.num_vertices) return fail(nullptr, PT_ERR_INVALID, "pt_create: vertex index out of range");
        nv += md.num_vertices;
        nt += md.num_triangles;
    }
    if (nt >= (1u << 28)) return fail(nullptr, PT_ERR_UNSUPPORTED, "pt_create: more than 2^28 triangles");
    for (uint32_t t = 0; t < scene->num_textures; ++t)
        if (!scene->textures || !scene->textures[t].pixel || scene->textures[t].width <= 0 || scene->textures[t].height <= 0)
            return fail(nullptr, PT_ERR_INVALID, "pt_create: bad texture");
    fs.scene = scene;
    fs.nv = nv;
    fs.nt = nt;
    fs.verts.resize(3 * nv);
    fs.idx.resize(3 * nt);
    fs.tri_mesh.resize(nt);
    fs.mats.resize(scene->num_meshes);
    fs.mesh_tex.assign(scene->num_meshes, -1);
    size_t vb = 0, tb = 0;
    for (uint32_t m = 0; m < scene->num_meshes; ++m) {
        const pt_mesh_desc& md = scene->meshes[m];
        memcpy(&fs.verts[3 * vb], md.vertex, sizeof(float) * 3 * md.num_vertices);
        for (size_t k = 0; k < 3 * (size_t)md.num_triangles; ++k) fs.idx[3 * tb + k] = md.index[k] + (uint32_t)vb;
        for (size_t k = 0; k < md.num_triangles; ++k) fs.tri_mesh[tb + k] = m;
        fs.mats[m] = md.material;
        if (md.material.flags & 1) fs.has_catcher = true;
        if (md.diffuse_texture_id >= 0 && md.texcoord) { // hasTexture && sbtData.texcoord (deviceProgram.cu:512)
            fs.mesh_tex[m] = md.diffuse_texture_id;
            fs.any_tex = true;
        }
        vb += md.num_vertices;
        tb += md.num_triangles;
    }
    if (fs.any_tex) { // per-vertex texcoords in the global vertex space (buildSBT, SimplePathtracer.cpp:430-447)
        fs.tc.assign(2 * nv, 0.f);
        size_t vb2 = 0;
        for (uint32_t m = 0; m < scene->num_meshes; ++m) {
            const pt_mesh_desc& md = scene->meshes[m];
            if (md.texcoord) memcpy(&fs.tc[2 * vb2], md.texcoord, sizeof(float) * 2 * md.num_vertices);
            vb2 += md.num_vertices;
        }
    }
    return PT_OK;
}

__global__ void k_warm_api() {}
static int create_from_flat(const FlatScene& fs, int device, pt_ctx** out_ctx) {
    const auto t_create0 = std::chrono::steady_clock::now();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, PT_ERR_NO_DEVICE, "pt_create: no HIP device");
    if (device < 0 || device >= ndev) return fail(nullptr, PT_ERR_INVALID, "pt_create: bad device ordinal");
    const pt_scene_desc* scene = fs.scene;
    const size_t nv = fs.nv, nt = fs.nt;
    pt_ctx* ctx = new pt_ctx();
    default_options(&ctx->opt);
    ctx->device = device;
    ctx->has_catcher = fs.has_catcher;
    if (const char* e = getenv("PT_TIMING")) ctx->env_timing = atoi(e) != 0;
    std::thread warm_thread; // (first pt_create of a process: loads the library's code objects while the scene is uploaded, below)
    auto bail = [&](int code) {
        if (warm_thread.joinable()) warm_thread.join(); // it launches on the context's stream, which pt_destroy is about to destroy
        g_create_error = ctx->err;
        pt_destroy(ctx);
        return code;
    };
#define CKC(call)                                                           \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);   \
            return bail(PT_ERR_HIP);                                        \
        }                                                                   \
    } while (0)
    DevScope tmp;
    CKC(hipSetDevice(device));
    if (const char* e = getenv("PT_STREAM_SHIFT")) { // test hook: streams created (and kept) ahead of the context's own shift the hardware-queue phase
        static std::vector<hipStream_t> ahead;
        for (int k = atoi(e); k > 0; --k) {
            hipStream_t st = nullptr;
            if (stream_create(&st) == hipSuccess) ahead.push_back(st);
        }
    }
    CKC(stream_create(&ctx->stream));
    // The first pt_create of a process pays for loading the two code objects of the library (the builder with its rocPRIM templates, and this
    // file) at their first kernel launch: about 8 + 2 ms that the first-build figure of round 5 (35 ms against 10 ms warm) contained.  A helper
    // thread launches an empty kernel of each now, while this thread uploads the scene (the copies below are synchronous and leave the CPU idle).
    static std::atomic<uint64_t> g_warmed{0}; // one bit per device
    if (device < 64 && !((g_warmed.fetch_or(1ull << device) >> device) & 1ull))
        warm_thread = std::thread([device, stream = ctx->stream]() {
            if (hipSetDevice(device) != hipSuccess) return;
            pt_bvh_warm(stream);
            hipLaunchKernelGGL(k_warm_api, dim3(1), dim3(64), 0, stream);
            (void)hipGetLastError();
        });
    struct JoinWarm { std::thread& t; ~JoinWarm() { if (t.joinable()) t.join(); } } join_warm{warm_thread};
    ctx->ntri = (uint32_t)nt;
    ctx->nmesh = scene->num_meshes;
    CKC(dalloc(&ctx->d_verts, 3 * nv));
    CKC(dalloc(&ctx->d_idx, 3 * nt));
    CKC(dalloc(&ctx->d_tri_mesh, nt));
    CKC(dalloc(&ctx->d_mats, fs.mats.size()));
    CKC(hipMemcpy(ctx->d_verts, fs.verts.data(), sizeof(float) * 3 * nv, hipMemcpyHostToDevice));
    CKC(hipMemcpy(ctx->d_idx, fs.idx.data(), sizeof(uint32_t) * 3 * nt, hipMemcpyHostToDevice));
    CKC(hipMemcpy(ctx->d_tri_mesh, fs.tri_mesh.data(), sizeof(uint32_t) * nt, hipMemcpyHostToDevice));
    CKC(hipMemcpy(ctx->d_mats, fs.mats.data(), sizeof(pt_material) * fs.mats.size(), hipMemcpyHostToDevice));
    {   // textures (createTextures, SimplePathtracer.cpp:603-654) and per-primitive texcoords (buildSBT :430-447)
        if (fs.any_tex) {
            float* d_tc = nullptr;
            CKC(tmp.alloc(&d_tc, 2 * nv));
            CKC(hipMemcpy(d_tc, fs.tc.data(), sizeof(float) * 2 * nv, hipMemcpyHostToDevice));
            CKC(dalloc(&ctx->d_uvs, nt));
            hipLaunchKernelGGL(k_emit_uvs, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ctx->stream, d_tc, ctx->d_idx, (uint32_t)nt, ctx->d_uvs);
            CKC(hipStreamSynchronize(ctx->stream));
            CKC(dalloc(&ctx->d_mesh_tex, fs.mesh_tex.size()));
            CKC(hipMemcpy(ctx->d_mesh_tex, fs.mesh_tex.data(), sizeof(int32_t) * fs.mesh_tex.size(), hipMemcpyHostToDevice));
        }
        std::vector<DevTex> tex(scene->num_textures);
        for (uint32_t t = 0; t < scene->num_textures; ++t) {
            const pt_texture_desc& td = scene->textures[t];
            uint32_t* px = nullptr;
            const int tiles_x = (td.width + 7) / 8, tiles_y = (td.height + 3) / 4; // 8 x 4-texel tiles of 128 bytes (pt_device.h DevTex)
            std::vector<uint32_t> tiled((size_t)tiles_x * tiles_y * 32, 0u);
            for (int y = 0; y < td.height; ++y)
                for (int x = 0; x < td.width; ++x) tiled[tex_tiled_index(x, y, tiles_x)] = td.pixel[(size_t)y * td.width + x];
            CKC(dalloc(&px, tiled.size()));
            ctx->d_tex_pixels.push_back(px);
            CKC(hipMemcpy(px, tiled.data(), sizeof(uint32_t) * tiled.size(), hipMemcpyHostToDevice));
            tex[t] = DevTex{px, td.width, td.height, tiles_x};
        }
        if (!tex.empty()) {
            CKC(dalloc(&ctx->d_textures, tex.size()));
            CKC(hipMemcpy(ctx->d_textures, tex.data(), sizeof(DevTex) * tex.size(), hipMemcpyHostToDevice));
            ctx->tex0 = tex[0];
        }
    }
    hipEvent_t e0, e1;
    CKC(tmp.event(&e0));
    CKC(tmp.event(&e1));
    if (warm_thread.joinable()) warm_thread.join(); // (whatever of the code-object load the upload did not cover is waited for here, outside bvh_build_ms but inside create_ms)
    CKC(hipEventRecord(e0, ctx->stream));
    CKC(pt_bvh_build(ctx->d_verts, ctx->d_idx, ctx->d_tri_mesh, (uint32_t)nt, ctx->stream, &ctx->bvh));
    CKC(dalloc(&ctx->d_tri_nrm, (size_t)ctx->bvh.num_tris8));
    hipLaunchKernelGGL(k_shade_normals, dim3((ctx->bvh.num_tris8 + 255) / 256), dim3(256), 0, ctx->stream, ctx->bvh.tris8, ctx->bvh.num_tris8, ctx->d_tri_nrm);
    if (ctx->d_uvs) { // textured scene: the 64-byte records a textured hit reads, in leaf order
        CKC(dalloc(&ctx->d_textris, (size_t)ctx->bvh.num_tris8));
        hipLaunchKernelGGL(k_emit_textris, dim3((ctx->bvh.num_tris8 + 255) / 256), dim3(256), 0, ctx->stream, ctx->bvh.tris8, ctx->d_uvs, ctx->bvh.num_tris8, ctx->d_textris);
    }
    CKC(hipEventRecord(e1, ctx->stream));
    CKC(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ctx->bvh_build_ms = ms;
    dfree(ctx->d_uvs);
    CKC(dalloc(&ctx->d_totals, (size_t)PT_MAX_FRAMES * PT_MAX_SETS * 4));
    CKC(dclear(ctx->stream, ctx->d_totals, sizeof(unsigned long long) * PT_MAX_FRAMES * PT_MAX_SETS * 4));
    CKC(hipHostMalloc((void**)&ctx->h_totals, sizeof(unsigned long long) * PT_MAX_FRAMES * PT_MAX_SETS * 4));
    {
        hipDeviceProp_t prop;
        CKC(hipGetDeviceProperties(&prop, device));
        int wpe = PT8_WAVES_PER_EU; // persistent waves per SIMD = the occupancy the traversal kernels are compiled for
        if (const char* e = getenv("PT_TRACE_WAVES")) wpe = atoi(e);
        if (const char* e = getenv("PT_GRID_MIN")) ctx->trace_grid_min = std::max(64, atoi(e));
        if (const char* e = getenv("PT_GRID_CHUNKS")) ctx->grid_chunks = std::max(1, atoi(e));
        if (const char* e = getenv("PT_STACK_LDS_SKIP")) ctx->lds_skip = std::max(0, std::min(PT8_LDS_DEPTH, atoi(e)));
        ctx->trace_grid = prop.multiProcessorCount * 4 * wpe;
        if (const char* e = getenv("PT_CAM_PACKETS")) ctx->cam_packets = atoi(e) != 0;
        if (const char* e = getenv("PT_CAM_MIN_PATHS")) ctx->cam_min_paths = (uint64_t)atoll(e);
        if (const char* e = getenv("PT_CAM_GRID")) ctx->cam_grid = atoi(e);
        if (const char* e = getenv("PT_SHADE_CUS")) ctx->shade_cus = std::max(0, std::min(prop.multiProcessorCount - 8, atoi(e)));
        ctx->num_cus = prop.multiProcessorCount;
        if (const char* e = getenv("PT_ENQUEUE_THREADS")) ctx->enqueue_threads = atoi(e);
        if (const char* e = getenv("PT_FUSED")) ctx->fused = atoi(e);
        if (const char* e = getenv("PT_FUSED_MAX_PATHS")) ctx->fused_max_paths = strtoull(e, nullptr, 10);
        if (const char* e = getenv("PT_FUSED_GRID")) ctx->fused_grid = atoi(e);
        if (const char* e = getenv("PT_FUSED_MAX_COST")) ctx->fused_max_cost = (float)atof(e);
        if (const char* e = getenv("PT_SCHED_TRIALS")) ctx->sched_trials = std::max(0, std::min(64, atoi(e)));
        if (const char* e = getenv("PT_SCHED_PROBE")) ctx->sched_probe = std::max(0, atoi(e));
        if (const char* e = getenv("PT_SCHED_MAX_PATHS")) ctx->sched_max_paths = strtoull(e, nullptr, 10);
        if (const char* e = getenv("PT_SCHED_INITIAL")) ctx->sched_initial = !strcmp(e, "fused") ? 1 : (!strcmp(e, "chain") ? 0 : -1);
        if (const char* e = getenv("PT_SCHED_FAKE")) {
            double c = 0, f = 0;
            if (sscanf(e, "%lf,%lf", &c, &f) == 2 && c > 0 && f > 0) { ctx->sched_fake[0] = c; ctx->sched_fake[1] = f; }
        }
        if (const char* e = getenv("PT_FUSED_CAP")) ctx->fused_cap = std::min(4096u, std::max(64u, ((uint32_t)atoi(e) + 63u) & ~63u));
        CKC(dalloc(&ctx->ovf, ovf_words(ctx)));
        ctx->ovf_depth = PT8_OVF_DEPTH;
        if (const char* e = getenv("PT_STACK_CAP")) ctx->ovf_depth = std::max(0, std::min(PT8_OVF_DEPTH, atoi(e) - (PT8_LDS_DEPTH - ctx->lds_skip)));
        if (const char* e = getenv("PT_STACK_NOCHECK")) ctx->stack_check = atoi(e) == 0;
        char msg[160];
        if (check_tree_depth(ctx, msg, sizeof(msg)) != PT_OK) {
            ctx->err = std::string("pt_create: ") + msg;
            return bail(PT_ERR_UNSUPPORTED);
        }
    }
    ctx->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count();
    *out_ctx = ctx;
    return PT_OK;
#undef CKC
}

extern "C" int pt_create(const pt_scene_desc* scene, int device, pt_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, PT_ERR_INVALID, "pt_create: null output pointer");
    FlatScene fs;
    int rc = flatten_scene(scene, fs);
    if (rc != PT_OK) return rc;
    return create_from_flat(fs, device, out_ctx);
}

static void free_path_state(pt_ctx* ctx) {
    for (auto& b : ctx->sets) {
        if (b.stream) hipStreamSynchronize(b.stream);
        if (b.stream2) hipStreamSynchronize(b.stream2);
        if (b.stream3) hipStreamSynchronize(b.stream3);
        PathState& s = b.st;
        for (auto& x : b.X) { dfree(x.rayO); dfree(x.rayD); dfree(x.thr); dfree(x.rf); dfree(x.hit); }
        dfree(b.shO); dfree(b.shD); dfree(b.shPend); dfree(s.pflags);
        dfree(s.direct); dfree(s.indirect); dfree(s.alpha); dfree(s.nrm); dfree(s.alb); dfree(s.prdN); dfree(s.prdA);
        dfree(b.queueA); dfree(b.queueB); dfree(b.squeue); dfree(b.counters); dfree(b.ovf); dfree(b.ovf2);
        dfree(b.squeueB); dfree(b.ovf3); dfree(s.sO); dfree(s.sD); dfree(s.pendB); dfree(s.vis);
        dfree(b.pixResult); dfree(b.pixAlpha); dfree(b.pixNormal); dfree(b.pixAlbedo);
    }
    ctx->sets.clear();
    ctx->set_cap = ctx->set_pix_cap = 0;
}
static void free_frame(pt_ctx* ctx) {
    for (void*& d : ctx->display) {
        if (d) hipFree(d);
        d = nullptr;
    }
    ctx->ev_pack_guard = nullptr;
    dfree(ctx->accum); dfree(ctx->color); dfree(ctx->normal); dfree(ctx->albedo); dfree(ctx->frame);
    dfree(ctx->denoised); dfree(ctx->denoise_tmp);
    dfree(ctx->d_pixels); dfree(ctx->d_all_pixels);
}

extern "C" int pt_destroy(pt_ctx* ctx) {
    if (!ctx) return PT_OK;
    hipSetDevice(ctx->device);
    drain(ctx);
    for (auto& w : ctx->chunk_workers) enqueue_worker_stop(w.get());
    ctx->chunk_workers.clear();
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    free_path_state(ctx);
    for (hipStream_t st : ctx->set_streams) if (st) hipStreamDestroy(st);
    for (hipStream_t st : ctx->side_streams) if (st) hipStreamDestroy(st);
    for (hipStream_t st : ctx->masked_streams) if (st) hipStreamDestroy(st);
    free_frame(ctx);
    dfree(ctx->d_verts); dfree(ctx->d_idx); dfree(ctx->d_tri_mesh); dfree(ctx->d_mats);
    dfree(ctx->d_mesh_tex); dfree(ctx->d_uvs); dfree(ctx->d_textris); dfree(ctx->d_textures);
    for (uint32_t*& px : ctx->d_tex_pixels) dfree(px);
    pt_bvh_free(&ctx->bvh);
    dfree(ctx->d_tri_nrm);
    dfree(ctx->d_probe_data); dfree(ctx->d_pdfX); dfree(ctx->d_cdfX); dfree(ctx->d_pdfY); dfree(ctx->d_cdfY);
    dfree(ctx->d_c64Y); dfree(ctx->d_c8Y); dfree(ctx->d_lines); dfree(ctx->d_guide);
    dfree(ctx->d_totals);
    if (ctx->h_totals) hipHostFree(ctx->h_totals);
    dfree(ctx->ovf);
    dfree(ctx->dbg);
    for (auto& pool : ctx->ev_pools)
        for (hipEvent_t e : pool) hipEventDestroy(e);
    for (hipEvent_t e : {ctx->ev_packed[0], ctx->ev_packed[1], ctx->ev_displayed})
        if (e) hipEventDestroy(e);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PT_OK;
}

extern "C" int pt_set_options(pt_ctx* ctx, const pt_options* opt) {
    if (!ctx || !opt) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (opt->max_depth < 0 || opt->max_depth > 250) return fail(ctx, PT_ERR_INVALID, "pt_set_options: max_depth out of range [0,250]");
    if (opt->bsdf_mode != PT_BSDF_DISNEY && opt->bsdf_mode != PT_BSDF_LAMBERT) return fail(ctx, PT_ERR_INVALID, "pt_set_options: bad bsdf_mode");
    if (opt->frames_in_flight < 0 || opt->frames_in_flight > PT_MAX_FRAMES) return fail(ctx, PT_ERR_INVALID, "pt_set_options: frames_in_flight must be 0 ... 3");
    if (opt->bvh_kind != 0 || opt->trace_kernel != 0)
        return fail(ctx, PT_ERR_UNSUPPORTED, "pt_set_options: bvh_kind and trace_kernel are reserved (the binary-BVH A/B paths of rounds 1-2 were removed): must be 0");
    ctx->opt = *opt;
    if (ctx->opt.max_paths == 0) ctx->opt.max_paths = 8u << 20;
    return PT_OK;
}
extern "C" int pt_get_options(const pt_ctx* ctx, pt_options* opt) {
    if (!ctx || !opt) return PT_ERR_INVALID;
    *opt = ctx->opt;
    return PT_OK;
}

extern "C" int pt_build_cdf(const float* data, int width, int height, float* pdfX, float* cdfX, float* pdfY, float* cdfY) {
    if (!data || width <= 0 || height <= 0 || !pdfX || !cdfX || !pdfY || !cdfY) return PT_ERR_INVALID;
    // ProbeData::BuildCDF, Probe.h:29-77: sequential float running sums (this TU is built with -ffp-contract=off)
    float totalWeightY = 0.0f;
    for (int j = 0; j < height; ++j) {
        float totalWeightX = 0.0f;
        for (int i = 0; i < width; ++i) {
            const float* c = &data[4 * ((size_t)j * width + i)];
            float weight = c[0] * 0.3f + c[1] * 0.6f + c[2] * 0.1f; // Luminance, maths.h:165-168
            totalWeightX += weight;
            pdfX[(size_t)j * width + i] = weight;
            cdfX[(size_t)j * width + i] = totalWeightX;
        }
        float invTotalWeightX = 1.0f / totalWeightX;
        for (int i = 0; i < width; ++i) {
            pdfX[(size_t)j * width + i] *= invTotalWeightX;
            cdfX[(size_t)j * width + i] *= invTotalWeightX;
        }
        totalWeightY += totalWeightX;
        pdfY[j] = totalWeightX;
        cdfY[j] = totalWeightY;
    }
    for (int j = 0; j < height; ++j) {
        cdfY[j] /= totalWeightY;
        pdfY[j] /= totalWeightY;
    }
    return PT_OK;
}

extern "C" int pt_uvw_frame(const float eye[3], const float lookat[3], const float up[3], float fovY, float aspect,
                            float U[3], float V[3], float W[3]) {
    if (!eye || !lookat || !up || !U || !V || !W) return PT_ERR_INVALID;
    // sutil::Camera::UVWFrame, sutil/Camera.cpp:34-45
    auto dot = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto cross = [](const float* a, const float* b, float* r) {
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto normalize = [&](float* v) {
        float inv = 1.0f / sqrtf(dot(v, v));
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    };
    float w[3] = {lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]};
    float wlen = sqrtf(dot(w, w));
    float u[3], v[3];
    cross(w, up, u);
    normalize(u);
    cross(u, w, v);
    normalize(v);
    float vlen = wlen * tanf(0.5f * fovY * 3.14159265358979323846f / 180.0f);
    float ulen = vlen * aspect;
    for (int k = 0; k < 3; ++k) {
        V[k] = v[k] * vlen;
        U[k] = u[k] * ulen;
        W[k] = w[k];
    }
    return PT_OK;
}

// finish a probe upload: build the block-search accelerators and publish the device view
static int finish_probe(pt_ctx* ctx, int w, int h) {
    dfree(ctx->d_c64Y); dfree(ctx->d_c8Y); dfree(ctx->d_lines); dfree(ctx->d_guide);
    const int ncy = h / PT_CDF_BLOCK, ncy_pad = (ncy + 7) & ~7;
    const bool oky = (h % PT_CDF_BLOCK) == 0 && h >= PT_CDF_BLOCK;
    const bool enabled = getenv("PT_NO_BLOCKED_SEARCH") == nullptr;
    // column tables: any width; the guide has one cell per ~2 columns (a power of two, so that r2 * gk is exact), PT_GUIDE_K overrides (A/B)
    const int lpr = (w + PT_LINE_COLS - 1) / PT_LINE_COLS;
    int gk = 64;
    while (gk < 4096 && gk * 2 < w) gk *= 2;
    if (const char* e = getenv("PT_GUIDE_K")) {
        const int v = atoi(e);
        if (v >= 2 && v <= 65536 && (v & (v - 1)) == 0) gk = v;
    }
    const int gpitch = gk + 2; // k = 0..gk, padded to an even count
    {
        CK(dalloc(&ctx->d_lines, (size_t)h * lpr + 1)); // + 1: a non-monotone row handed to pt_set_probe can index one texel past its line
        CK(dalloc(&ctx->d_guide, (size_t)h * gpitch));
        CK(hipMemsetAsync(ctx->d_lines + (size_t)h * lpr, 0, sizeof(ProbeLine), ctx->stream));
        hipLaunchKernelGGL(k_probe_lines, dim3((unsigned)(((size_t)h * lpr + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_probe_data, ctx->d_pdfX,
                           ctx->d_cdfX, h, w, lpr, ctx->d_lines);
        hipLaunchKernelGGL(k_probe_guide, dim3((unsigned)(((size_t)h * gpitch + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_lines, h, lpr, gk,
                           gpitch, ctx->d_guide);
    }
    if (oky && enabled) {
        CK(dalloc(&ctx->d_c64Y, (size_t)ncy_pad));
        CK(dalloc(&ctx->d_c8Y, (size_t)(h / 8)));
        hipLaunchKernelGGL(k_probe_coarse, dim3((ncy_pad + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_cdfY, 1, h, PT_CDF_BLOCK, ncy_pad, ctx->d_c64Y);
        hipLaunchKernelGGL(k_probe_coarse, dim3((h / 8 + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_cdfY, 1, h, 8, h / 8, ctx->d_c8Y);
    }
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipGetLastError());
    ctx->probe = DevProbe{w, h, ctx->d_probe_data, ctx->d_pdfX, ctx->d_cdfX, ctx->d_pdfY, ctx->d_cdfY, ctx->d_c64Y, ctx->d_c8Y, ncy,
                          ctx->d_lines, ctx->d_guide, lpr, gk, gpitch};
    return PT_OK;
}

extern "C" int pt_set_probe(pt_ctx* ctx, const float* data, const float* pdfX, const float* cdfX, const float* pdfY,
                            const float* cdfY, int w, int h) {
    if (!ctx) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (!data || !pdfX || !cdfX || !pdfY || !cdfY || w <= 0 || h <= 0) return fail(ctx, PT_ERR_INVALID, "pt_set_probe: Probe Data is not valid");
    if ((long long)w * h >= (1ll << 31)) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_set_probe: probe too large");
    if (w > 65535 * PT_LINE_COLS) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_set_probe: probe wider than 393210 columns"); // u16 line counts in the guide
    CK(hipSetDevice(ctx->device));
    dfree(ctx->d_probe_data); dfree(ctx->d_pdfX); dfree(ctx->d_cdfX); dfree(ctx->d_pdfY); dfree(ctx->d_cdfY);
    const size_t n = (size_t)w * h;
    CK(dalloc(&ctx->d_probe_data, n));
    CK(dalloc(&ctx->d_pdfX, n));
    CK(dalloc(&ctx->d_cdfX, n));
    CK(dalloc(&ctx->d_pdfY, (size_t)h));
    CK(dalloc(&ctx->d_cdfY, (size_t)h));
    CK(hipMemcpy(ctx->d_probe_data, data, sizeof(float4) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_pdfX, pdfX, sizeof(float) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_cdfX, cdfX, sizeof(float) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_pdfY, pdfY, sizeof(float) * h, hipMemcpyHostToDevice));
    CK(hipMemcpy(ctx->d_cdfY, cdfY, sizeof(float) * h, hipMemcpyHostToDevice));
    return finish_probe(ctx, w, h);
}

extern "C" int pt_set_probe_image(pt_ctx* ctx, const float* data, int w, int h) {
    if (!ctx) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (!data || w <= 0 || h <= 0) return fail(ctx, PT_ERR_INVALID, "pt_set_probe_image: Probe Data is not valid");
    if ((long long)w * h >= (1ll << 31)) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_set_probe_image: probe too large");
    if (w > 65535 * PT_LINE_COLS) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_set_probe_image: probe wider than 393210 columns"); // u16 line counts in the guide
    CK(hipSetDevice(ctx->device));
    CK(hipStreamSynchronize(ctx->stream));
    dfree(ctx->d_probe_data); dfree(ctx->d_pdfX); dfree(ctx->d_cdfX); dfree(ctx->d_pdfY); dfree(ctx->d_cdfY);
    const size_t n = (size_t)w * h;
    DevScope tmp;
    float* rowTotal = nullptr;
    CK(dalloc(&ctx->d_probe_data, n));
    CK(dalloc(&ctx->d_pdfX, n));
    CK(dalloc(&ctx->d_cdfX, n));
    CK(dalloc(&ctx->d_pdfY, (size_t)h));
    CK(dalloc(&ctx->d_cdfY, (size_t)h));
    CK(tmp.alloc(&rowTotal, (size_t)h));
    CK(hipMemcpy(ctx->d_probe_data, data, sizeof(float4) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cdf_rows, dim3(h), dim3(64), 0, ctx->stream, ctx->d_probe_data, w, h, ctx->d_pdfX, ctx->d_cdfX, rowTotal);
    hipLaunchKernelGGL(k_cdf_marginal, dim3(1), dim3(64), 0, ctx->stream, rowTotal, h, ctx->d_pdfY, ctx->d_cdfY);
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipGetLastError());
    return finish_probe(ctx, w, h);
}

extern "C" int pt_get_probe_cdf(pt_ctx* ctx, float* pdfX, float* cdfX, float* pdfY, float* cdfY) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->probe.data) return fail(ctx, PT_ERR_INVALID, "pt_get_probe_cdf: no probe set");
    CK(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->probe.width * ctx->probe.height;
    if (pdfX) CK(hipMemcpy(pdfX, ctx->d_pdfX, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (cdfX) CK(hipMemcpy(cdfX, ctx->d_cdfX, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (pdfY) CK(hipMemcpy(pdfY, ctx->d_pdfY, sizeof(float) * ctx->probe.height, hipMemcpyDeviceToHost));
    if (cdfY) CK(hipMemcpy(cdfY, ctx->d_cdfY, sizeof(float) * ctx->probe.height, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_set_camera(pt_ctx* ctx, const float eye[3], const float U[3], const float V[3], const float W[3]) {
    if (!ctx || !eye || !U || !V || !W) return PT_ERR_INVALID;
    ctx->eye = v3{eye[0], eye[1], eye[2]};
    ctx->U = v3{U[0], U[1], U[2]};
    ctx->V = v3{V[0], V[1], V[2]};
    ctx->W = v3{W[0], W[1], W[2]};
    return PT_OK;
}

extern "C" int pt_set_partition(pt_ctx* ctx, int rank, int world, int tile_w, int tile_h) {
    if (!ctx) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (world < 1 || rank < 0 || rank >= world || tile_w < 8 || tile_h < 8 || (tile_w % 8) || (tile_h % 8))
        return fail(ctx, PT_ERR_INVALID, "pt_set_partition: need 0<=rank<world and tile sizes that are multiples of 8");
    ctx->rank = rank; ctx->world = world; ctx->tile_w = tile_w; ctx->tile_h = tile_h;
    if (ctx->width > 0) { // re-apply to the current frame size
        int w = ctx->width, h = ctx->height;
        ctx->width = ctx->height = 0;
        return pt_resize(ctx, w, h);
    }
    return PT_OK;
}

// pixel lists in 8x8-block order: lanes of a wave cover one 8x8 block (coherent primary rays)
static void build_pixel_lists(int w, int h, int world, int tile_w, int tile_h, std::vector<std::vector<uint32_t>>& lists) {
    lists.assign(world, {});
    for (int by = 0; by < (h + 7) / 8; ++by)
        for (int bx = 0; bx < (w + 7) / 8; ++bx) {
            const int owner = ((bx * 8) / tile_w + (by * 8) / tile_h) % world;
            for (int iy = 0; iy < 8; ++iy)
                for (int ix = 0; ix < 8; ++ix) {
                    const int x = bx * 8 + ix, y = by * 8 + iy;
                    if (x < w && y < h) lists[owner].push_back((uint32_t)x | ((uint32_t)y << 16));
                }
        }
}

extern "C" int pt_resize(pt_ctx* ctx, int width, int height) {
    if (!ctx) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (width == 0 || height == 0) return PT_OK; // SimplePathtracer.cpp:112
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(ctx, PT_ERR_INVALID, "pt_resize: size out of range");
    CK(hipSetDevice(ctx->device));
    CK(hipStreamSynchronize(ctx->stream));
    free_frame(ctx);
    const size_t n = (size_t)width * height;
    CK(dalloc(&ctx->accum, n));
    CK(dalloc(&ctx->color, n));
    CK(dalloc(&ctx->normal, n));
    CK(dalloc(&ctx->albedo, n));
    CK(dalloc(&ctx->frame, n));
    CK(hipMemsetAsync(ctx->accum, 0, sizeof(float4) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->color, 0, sizeof(float4) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->normal, 0, sizeof(float4) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->albedo, 0, sizeof(float4) * n, ctx->stream));
    CK(dclear(ctx->stream, ctx->frame, sizeof(uint32_t) * n));
    std::vector<std::vector<uint32_t>> lists;
    build_pixel_lists(width, height, ctx->world, ctx->tile_w, ctx->tile_h, lists);
    size_t padded = 0;
    for (auto& l : lists) padded = std::max(padded, l.size());
    ctx->owned = (uint32_t)lists[ctx->rank].size();
    ctx->padded = (uint32_t)padded;
    CK(dalloc(&ctx->d_pixels, (size_t)ctx->owned));
    if (ctx->owned) CK(hipMemcpy(ctx->d_pixels, lists[ctx->rank].data(), sizeof(uint32_t) * ctx->owned, hipMemcpyHostToDevice));
    std::vector<uint32_t> all((size_t)ctx->world * padded, 0xffffffffu);
    for (int r = 0; r < ctx->world; ++r) std::copy(lists[r].begin(), lists[r].end(), all.begin() + (size_t)r * padded);
    CK(dalloc(&ctx->d_all_pixels, all.size()));
    if (!all.empty()) CK(hipMemcpy(ctx->d_all_pixels, all.data(), sizeof(uint32_t) * all.size(), hipMemcpyHostToDevice));
    ctx->width = width;
    ctx->height = height;
    return PT_OK;
}

static size_t ovf_words(const pt_ctx* ctx) {
    return (size_t)ctx->trace_grid * 64 * (PT8_OVF_DEPTH * 2);
}

// asynchronous shadow rays (split_shadow = 2): per-bounce shadow records and queues; not for shadow-catcher scenes, whose
// alpha accumulation interleaves assignments and sums, and only with the default traversal
static bool async_shadows(const pt_ctx* ctx) {
    return ctx->opt.split_shadow == 2 && !ctx->has_catcher && ctx->opt.max_depth < 31;
}

// Do two streams run concurrently, or do they share a hardware queue (and serialise)?  HIP deals its streams onto four hardware queues
// in creation order, back and forth (tools/micro/queue_map.hip on MI355X / ROCm 7.2: streams 0..7 -> queues 0 1 2 3 3 2 1 0), counting
// every stream of the process — torch's included — so which of a context's streams collide depends on what was created before it.
// Measured consequences: a synchronous frame 10.1 instead of 9.0 ms, three frames in flight 9.1 instead of 8.2 ms.  So the context asks:
// a ~200 us spin kernel on `a`, a kernel on `b` that stamps the device clock: if `b` was stamped before the spin ended, the two ran
// concurrently.  Both times are read on the device, so a host thread that is preempted between the launches (or the rank threads of a
// pt_multi contending for cores) cannot turn a concurrent pair into a serial one or vice versa.
__global__ void k_spin(long long ticks, long long* end_out) {
    const long long t0 = wall_clock64(); // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) *end_out = wall_clock64();
}
__global__ void k_stamp(long long* out) {
    if (threadIdx.x == 0) *out = wall_clock64();
}
static bool streams_concurrent(hipStream_t a, hipStream_t b, long long* d_stamps) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 20000LL, d_stamps);
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, b, d_stamps + 1);
    long long h[2] = {0, 0};
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
        hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return true; // inconclusive: keep creation order
    }
    return h[1] < h[0];
}
// The context's stream and the streams of the first three batch sets — the four that carry a frame — are picked so that no two share a
// hardware queue: streams are created until enough mutually concurrent ones are found (at most 12; among any eight consecutive
// creations every queue appears twice), the others become the side streams of the split / asynchronous shadow schedules.
// PT_STREAM_PROBE=0 keeps plain creation order.
static std::mutex g_probe_mu; // one probe at a time per process: contexts that share a device (pt_create_multi rehearsals, several renderers of
                              // one application) would otherwise time each other's spin kernels and misjudge which streams run concurrently
static int pick_streams(pt_ctx* ctx, int nsets) {
    const char* pe = getenv("PT_STREAM_PROBE");
    if (pe && atoi(pe) == 0) return PT_OK;
    std::lock_guard<std::mutex> probe_lock(g_probe_mu);
    long long* d_stamps = nullptr;
    CK(hipMalloc(&d_stamps, 2 * sizeof(long long)));
    CK(hipStreamSynchronize(ctx->stream));
    // the streams that already carry frames stay; new ones must be concurrent with all of them (re-probed when the number of sets grows)
    std::vector<hipStream_t> chosen{ctx->stream}, spare;
    for (int i = 0; i < nsets && i < 3; ++i)
        if (ctx->set_streams[i]) chosen.push_back(ctx->set_streams[i]);
    const int want = 1 + std::min(nsets, 3);
    int attempts = 0;
    for (; attempts < 12 && (int)chosen.size() < want; ++attempts) {
        hipStream_t st = nullptr;
        if (stream_create(&st) != hipSuccess) break;
        bool ok = true;
        for (hipStream_t c : chosen)
            if (!streams_concurrent(c, st, d_stamps)) { ok = false; break; }
        (ok ? chosen : spare).push_back(st);
    }
    hipFree(d_stamps);
    (void)hipGetLastError();
    if (getenv("PT_DEBUG"))
        fprintf(stderr, "[ptamd] pick_streams: %d sets, %d streams created, %zu mutually concurrent (want %d), %zu to the side\n", nsets, attempts, chosen.size(), want, spare.size());
    size_t next_chosen = 1, next_spare = 0;
    for (int i = 0; i < nsets && i < 3; ++i)
        if (ctx->set_streams[i]) ++next_chosen; // those were put at the front of `chosen`
    for (int i = 0; i < nsets && i < 3; ++i) {
        if (ctx->set_streams[i]) continue;
        if (next_chosen < chosen.size()) ctx->set_streams[i] = chosen[next_chosen++];
        else if (next_spare < spare.size()) ctx->set_streams[i] = spare[next_spare++]; // probing was inconclusive: any stream will do
    }
    for (size_t k = 0; k < ctx->side_streams.size() && next_spare < spare.size(); ++k)
        if (!ctx->side_streams[k]) ctx->side_streams[k] = spare[next_spare++];
    for (; next_spare < spare.size(); ++next_spare) ctx->side_streams.push_back(spare[next_spare]); // kept (and destroyed with the context)
    ctx->streams_probed = true;
    return PT_OK;
}

// the streams of batch set i (created on first use, kept until pt_destroy)
static int assign_streams(pt_ctx* ctx, int nsets) {
    const bool async = async_shadows(ctx);
    if ((int)ctx->set_streams.size() < nsets) ctx->set_streams.resize(nsets, nullptr);
    if ((int)ctx->side_streams.size() < 2 * nsets) ctx->side_streams.resize((size_t)2 * nsets, nullptr);
    bool missing = false;
    for (int i = 0; i < nsets && i < 3; ++i) missing |= !ctx->set_streams[i];
    if (!ctx->streams_probed || missing) { // first use, or more sets than were probed for
        int rc = pick_streams(ctx, nsets);
        if (rc) return rc;
    }
    for (int i = 0; i < nsets; ++i) {
        if (!ctx->set_streams[i]) CK(stream_create(&ctx->set_streams[i]));
        if (!ctx->side_streams[2 * i]) CK(stream_create(&ctx->side_streams[2 * i]));
        if (async && !ctx->side_streams[2 * i + 1]) CK(stream_create(&ctx->side_streams[2 * i + 1]));
        if (ctx->shade_cus > 0) { // experiment: the chain's stream is a queue of its own with a CU mask, the probed stream carries k_shade
            if ((int)ctx->masked_streams.size() <= i) ctx->masked_streams.resize(i + 1, nullptr);
            if (!ctx->masked_streams[i]) {
                std::vector<uint32_t> mask((size_t)(ctx->num_cus + 31) / 32, 0u);
                for (int c = 0; c < ctx->num_cus - ctx->shade_cus; ++c) mask[c / 32] |= 1u << (c % 32);
                CK(hipExtStreamCreateWithCUMask(&ctx->masked_streams[i], (uint32_t)mask.size(), mask.data()));
            }
        }
        if (i < (int)ctx->sets.size()) {
            ctx->sets[i].stream = ctx->set_streams[i];
            if (ctx->shade_cus > 0) {
                ctx->sets[i].stream = ctx->masked_streams[i];
                ctx->sets[i].shade_stream = ctx->set_streams[i];
            }
            ctx->sets[i].stream2 = ctx->side_streams[2 * i];
            ctx->sets[i].stream3 = ctx->side_streams[2 * i + 1];
        }
    }
    return PT_OK;
}

// Path state for a frame cut into `nsets` chunks of at most `cap` paths over `pix_cap` pixels: the first `nsets` batch sets are the frame's.
// The state only ever GROWS — in the number of sets, in paths per set and in pixels per set, each kept at the largest value any frame asked
// for — so an application that alternates schedules (a fused-size synchronous frame = one set holding the whole frame; a foveated or
// pipelined frame = three sets) re-allocates at most once per dimension and never again (the one-set / three-set switch used
// to drain, free and re-allocate every buffer on every alternation).  pt_stats.path_state_allocs counts the (re-)allocations.
static int ensure_path_state(pt_ctx* ctx, int nsets, uint32_t cap, uint32_t pix_cap) {
    // one queue-counter slot per bounce plus the "would continue" slot; shadow-catcher scenes get two more for the extra
    // iterations of paths that passed through catcher surfaces (enqueue_chunk)
    int nq = ctx->opt.max_depth + 2 + (ctx->has_catcher ? 2 : 0);
    const bool async = async_shadows(ctx);
    if ((int)ctx->sets.size() >= nsets && cap <= ctx->set_cap && pix_cap <= ctx->set_pix_cap && nq <= ctx->nq &&
        (!ctx->has_catcher || ctx->cap_catcher) && async == ctx->cap_async)
        return assign_streams(ctx, (int)ctx->sets.size());
    {
        int rc = drain(ctx); // frames in flight still use the old buffers
        if (rc) return rc;
    }
    nsets = std::max(nsets, (int)ctx->sets.size());
    cap = std::max(cap, ctx->set_cap);
    pix_cap = std::max(pix_cap, ctx->set_pix_cap);
    nq = std::max(nq, ctx->nq);
    ++ctx->path_state_allocs;
    free_path_state(ctx);
    CK(dclear(ctx->stream, ctx->d_totals, sizeof(unsigned long long) * PT_MAX_FRAMES * PT_MAX_SETS * 4)); // slices of sets that no longer exist
    ctx->sets.resize(nsets);
    ctx->sub_cap = cap / PT_NSUB + 8192 + 1; // a sub-queue receives at most cap/64 + 32 workgroups * 256 entries
    const size_t qsize = (size_t)PT_NSUB * ctx->sub_cap;
    {
        int rc = assign_streams(ctx, nsets);
        if (rc) return rc;
    }
    for (auto& b : ctx->sets) {
        PathState& s = b.st;
        for (auto& x : b.X) { CK(dalloc(&x.rayO, qsize)); CK(dalloc(&x.rayD, qsize)); CK(dalloc(&x.thr, qsize)); CK(dalloc(&x.rf, qsize)); CK(dalloc(&x.hit, qsize)); }
        CK(dalloc(&b.shO, qsize)); CK(dalloc(&b.shD, qsize)); CK(dalloc(&b.shPend, qsize)); CK(dalloc(&s.pflags, cap));
        CK(dalloc(&s.direct, cap)); CK(dalloc(&s.indirect, cap)); CK(dalloc(&s.nrm, cap)); CK(dalloc(&s.alb, cap));
        if (ctx->has_catcher) { CK(dalloc(&s.alpha, cap)); CK(dalloc(&s.prdN, cap)); CK(dalloc(&s.prdA, cap)); }
        CK(dalloc(&b.queueA, qsize)); CK(dalloc(&b.queueB, qsize)); CK(dalloc(&b.squeue, qsize));
        CK(dalloc(&b.counters, counter_words(nq)));
        CK(dalloc(&b.ovf, ovf_words(ctx))); CK(dalloc(&b.ovf2, ovf_words(ctx)));
        CK(dalloc(&b.pixResult, pix_cap)); CK(dalloc(&b.pixAlpha, pix_cap)); CK(dalloc(&b.pixNormal, pix_cap)); CK(dalloc(&b.pixAlbedo, pix_cap));
        if (async) {
            CK(dalloc(&s.sO, (size_t)nq * cap)); CK(dalloc(&s.sD, (size_t)nq * cap)); CK(dalloc(&s.pendB, (size_t)nq * cap));
            CK(dalloc(&s.vis, cap));
            s.bstride = cap;
            CK(dalloc(&b.squeueB, (size_t)nq * qsize));
            CK(dalloc(&b.ovf3, ovf_words(ctx)));
        }
    }
    ctx->cap_async = async;
    ctx->cap_catcher = ctx->has_catcher;
    ctx->nq = nq;
    ctx->set_cap = cap;
    ctx->set_pix_cap = pix_cap;
    return PT_OK;
}

enum { CLS_TRACE = 0, CLS_SHADOW = 1, CLS_SHADE = 2, CLS_OTHER = 3 };

static hipEvent_t next_event(pt_ctx* ctx) {
    auto& pool = ctx->ev_pools[ctx->cur_slot];
    if (ctx->ev_used == pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        pool.push_back(e);
    }
    return pool[ctx->ev_used++];
}
struct SpanGuard {
    pt_ctx* ctx;
    size_t a;
    int cls;
    hipStream_t s;
    SpanGuard(pt_ctx* c, int cl, hipStream_t st = nullptr) : ctx(c), cls(cl), s(st ? st : c->stream) {
        if (!ctx->span_timing()) return;
        a = ctx->ev_used;
        hipEventRecord(next_event(ctx), s);
    }
    ~SpanGuard() {
        // every launch group is checked where it is enqueued, so that a failed launch is reported by name (the frame's final
        // hipGetLastError would only say that something failed); costs a thread-local read per group
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            std::lock_guard<std::mutex> lk(ctx->launch_err_mu);
            if (ctx->launch_err.empty()) {
                static const char* const names[4] = {"closest-hit traversal (k_trace8)", "shadow traversal (k_trace8)", "k_shade", "generate / resolve"};
                ctx->launch_err = std::string("launch of ") + names[cls & 3] + " failed: " + hipGetErrorString(le);
            }
        }
        if (!ctx->span_timing()) return;
        size_t b = ctx->ev_used;
        hipEventRecord(next_event(ctx), s);
        ctx->spans.push_back({a, b, cls});
    }
};

static const int GRID = 256 * 8;

// dynamic LDS of k_shade: its copy of the probe's marginal arrays (cdfY, pdfY, c8Y, c64Y), or nothing when they stay in global memory
static size_t shade_lds_bytes(const DevProbe& p) {
    if (!p.c64Y || p.height > PT_LDS_PROBE_ROWS) return 0;
    return sizeof(float) * ((size_t)2 * p.height + p.height / 8 + ((p.ncy + 7) & ~7));
}
// the path state a launch sees: the slot-indexed arrays plus the radiance stream of its input queue (X[k]) and the shadow stream
static PathState stream_view(const pt_ctx::BatchSet& bs, int k) {
    PathState s = bs.st;
    s.rayO = bs.X[k].rayO; s.rayD = bs.X[k].rayD; s.thr = bs.X[k].thr; s.rf = bs.X[k].rf; s.hit = bs.X[k].hit;
    s.shO = bs.shO; s.shD = bs.shD; s.shPend = bs.shPend;
    return s;
}
template <int MODE>
static void launch_shade(pt_ctx* ctx, pt_ctx::BatchSet& bs, const PathState& st, const ShadeParams& sp) {
    const unsigned lds = (unsigned)shade_lds_bytes(sp.probe);
    hipStream_t s = bs.stream;
    if (bs.shade_stream) { // PT_SHADE_CUS experiment: behind the chain's last launch, on the stream that may use every CU
        hipEvent_t e = next_event(ctx);
        hipEventRecord(e, bs.stream);
        s = bs.shade_stream;
        hipStreamWaitEvent(s, e, 0);
    }
    if (ctx->has_catcher)
        hipLaunchKernelGGL((k_shade<MODE, true>), dim3(GRID), dim3(256), lds, s, st, sp);
    else
        hipLaunchKernelGGL((k_shade<MODE, false>), dim3(GRID), dim3(256), lds, s, st, sp);
    if (bs.shade_stream) {
        hipEvent_t e = next_event(ctx);
        hipEventRecord(e, s);
        hipStreamWaitEvent(bs.stream, e, 0);
    }
}

// Closest-hit launch of a bounce chain.  The identity queue of bounce 0 holds camera rays in pixel-block order: they are traversed as
// packets, one wave per 64 consecutive rays (k_trace8_cam, pt_bvh8.h: 55 VGPRs, so eight waves per SIMD instead of five); every other queue
// (later bounces, foveated launches, whose paths arrive through the sub-queues) takes the per-ray kernel.  PT_CAM_PACKETS=0: per-ray always.
template <int MODE>
static void launch_trace8(hipStream_t stream, unsigned tgrid, const Trace8Args& ta) {
    hipLaunchKernelGGL((k_trace8<MODE>), dim3(tgrid), dim3(64), 0, stream, ta);
}
static Bvh8Dev bvh_dev(const pt_ctx* ctx) {
#if PT8_NODE64
    return Bvh8Dev{ctx->bvh.nodes8, ctx->bvh.tris8, 0.5f * ctx->bvh.pad, ctx->bvh.grid};
#else
    return Bvh8Dev{ctx->bvh.nodes8, ctx->bvh.tris8, 0.5f * ctx->bvh.pad};
#endif
}
static void launch_closest(pt_ctx* ctx, hipStream_t stream, const Trace8Args& ta, unsigned tgrid, uint64_t paths) {
    if (ta.queue.base == nullptr && ctx->cam_packets && paths >= ctx->cam_min_paths) {
        const int env_grid = ctx->cam_grid; // (read at pt_create like every other switch, so a context keeps what it was created with)
        // as many waves as the per-ray kernel gets (five per SIMD), two per SIMD for a tree of a few nodes whose packets cost next to nothing
        // (measured: C3 8.00 / stadium 12.51 ms at 5120 waves, 8.20 / 12.81 at 2048; Cornell 2.98 ms at 2048, 3.14 at 5120 = the per-ray kernel's)
        const unsigned g = env_grid > 0 ? (unsigned)env_grid : std::max(1u, ctx->bvh.num_nodes8 < 256u ? tgrid * 2u / (unsigned)PT8_WAVES_PER_EU : tgrid);
        hipLaunchKernelGGL(k_trace8_cam, dim3(g), dim3(64), 0, stream, ta);
    } else {
        launch_trace8<TR_CLOSEST>(stream, tgrid, ta);
    }
}

// enqueue every kernel of one pixel chunk (all its samples) on the streams of one batch set
struct RegionJob { // non-null: one launch-index range of a foveated launch instead of a pixel chunk
    RegionParams rg;
    VariantParams var;
    uint32_t l0, nl;
};

// spp: samples of the whole chunk = samples_per_launch x subframes of the batch (pt_render_batch; a foveated launch: its own spp),
// S: samples per pass
static void enqueue_chunk(pt_ctx* ctx, pt_ctx::BatchSet& bs, const FrameParams& fp, uint32_t pix0, uint32_t npix, uint32_t spp, uint32_t S,
                          LaunchCounts& lc, const RegionJob* job = nullptr, const std::vector<hipEvent_t>* before_resolve = nullptr) {
    const int nq = ctx->nq;
    const float tmin_rad = job ? job->var.radiance_tmin : 0.001f;
    const int cull = job ? job->var.cull_back_occlusion : 0;
    const Bvh8Dev bvh8 = bvh_dev(ctx);
    const LeafTri* shade_tris = ctx->bvh.tris8; // the hit records index the leaf triangles of the structure that was traversed
    const size_t CS = (size_t)PT_NSUB * PT_CSTRIDE;
    for (uint32_t s0 = 0; s0 < spp; s0 += S) {
        const uint32_t Sc = std::min(S, spp - s0);
        // Persistent traversal waves of this pass's launches.  A synchronous frame wants the full grid (5 waves per SIMD): its latency is what
        // counts.  With whole frames in flight (frames_in_flight = 3) throughput counts and the other streams take every wave slot left free:
        // a small batch then runs faster on fewer waves (refill and drain phases amortised over more rays per wave) — measured optimum
        // ≈ 6 chunks of 64 of the pass's paths per wave
        const uint64_t pass_paths = job ? (uint64_t)job->nl * spp : (uint64_t)npix * Sc;
        // (never above trace_grid: the spill stacks are sized for trace_grid waves and the kernels are compiled for that occupancy)
        const unsigned tgrid = !ctx->adapt_grid ? (unsigned)ctx->trace_grid
            : (unsigned)std::min<uint64_t>((uint64_t)ctx->trace_grid, std::max<uint64_t>((uint64_t)ctx->trace_grid_min, pass_paths / (64ull * (uint64_t)ctx->grid_chunks)));
        BatchParams bp{ctx->d_pixels + pix0, npix, s0, Sc, ctx->has_catcher ? 1 : 0, bs.pixResult, bs.pixAlpha, bs.pixNormal, bs.pixAlbedo};
        hipMemsetAsync(bs.counters, 0, sizeof(uint32_t) * counter_words(nq), bs.stream);
        uint32_t* cntA = bs.counters;                       // radiance queue counters, per bounce
        uint32_t* cntS = bs.counters + (size_t)nq * CS;     // shadow queue counters, per bounce
        uint32_t* work = bs.counters + (size_t)2 * nq * CS; // work counters of the persistent traversal
        QView qcur{nullptr, cntA, ctx->sub_cap};            // identity for bounce 0 (k_generate wrote the count)
        if (job) qcur.base = bs.queueB;                     // foveated launch: only the paths inside the annulus are queued
        const bool fused = !job && ctx->opt.split_shadow == 0 && !ctx->cap_async && !ctx->has_catcher && (ctx->fused == 2 || ctx->fused_frame);
        if (!fused) {
            SpanGuard g(ctx, CLS_OTHER, bs.stream);
            if (job)
                hipLaunchKernelGGL(k_generate_region, dim3(GRID), dim3(256), 0, bs.stream, stream_view(bs, 1), fp, job->rg, PartParams{ctx->rank, ctx->world, ctx->tile_w, ctx->tile_h}, tmin_rad, (uint32_t)job->var.initial_depth, job->l0, job->nl, qcur);
            else
                hipLaunchKernelGGL(k_generate, dim3(GRID), dim3(256), 0, bs.stream, stream_view(bs, 1), fp, bp, bs.counters + 0);
        }
        uint32_t* qnext_base = bs.queueA;
        int sin = 1; // X[sin] holds the state of the queue being traced / shaded, X[sin ^ 1] receives the next queue's
        // depth d = 0..max_depth traces in the reference (the trace at depth == max_depth can only matter
        // through a shadow-catcher pass-through or alpha; without catcher materials it is provably dead and skipped)
        // (a foveated launch of the sv / sv2 variants starts its paths at depth 1: that many fewer bounces are live)
        const int depth0 = job ? job->var.initial_depth : 0;
        const int last_bounce = (ctx->has_catcher ? ctx->opt.max_depth : ctx->opt.max_depth - 1) - depth0;
        hipEvent_t ev_shadow_done = nullptr;
        const bool unified = ctx->opt.split_shadow == 0;
        const bool async = ctx->cap_async;
        if (fused) {
            // one persistent kernel instead of the chain below: every wave runs generate -> trace -> shade rounds on a private window of the
            // queue arrays (pt_fused.h); the windows of all waves fit the arrays (grid x cap <= the pass's paths, rounded up to whole waves)
            SpanGuard g(ctx, CLS_TRACE, bs.stream);
            // Window entries per wave.  The pool of unstarted paths is what balances the waves, so the windows of all waves together should hold well
            // under the pass: 128 entries (two lane-fulls per round: fewer, fuller rounds) when at least 30 % of the paths stay in the pool at the
            // start, else 64 (measured, C3: 2.07 M paths 2.77 ms with 128 / 2.97 with 64; 1.04 M 1.71 / 1.77; 0.69 M 1.67 / 1.41; 0.52 M 1.47 / 1.13;
            // 0.26 M 1.19 / 0.97 — profiles/r5_14_fused_bounce_loop.md)
            const uint32_t cap_auto = pass_paths * 10ull >= 13ull * (uint64_t)tgrid * 128ull ? 128u : 64u;
            const uint32_t cap = std::min<uint32_t>(ctx->fused_cap ? ctx->fused_cap : cap_auto, ((uint32_t)pass_paths + 63u) & ~63u);
            const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ctx->fused_grid > 0 ? std::min<unsigned>(tgrid, (unsigned)ctx->fused_grid) : tgrid, pass_paths / cap));
            PathLoopArgs pa{};
            pa.ta = Trace8Args{stream_view(bs, 0), bvh8, QView{bs.queueA, nullptr, 0u}, QView{bs.squeue, nullptr, 0u}, work, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
            pa.rayO1 = bs.X[1].rayO; pa.rayD1 = bs.X[1].rayD; pa.thr1 = bs.X[1].thr; pa.hit1 = bs.X[1].hit; pa.rf1 = bs.X[1].rf;
            pa.qbase1 = bs.queueB;
            pa.sp = ShadeParams{shade_tris, ctx->d_tri_nrm, ctx->d_mats, ctx->d_mesh_tex, ctx->d_textris, ctx->d_textures, ctx->probe, ctx->opt.max_depth, tmin_rad, QView{}, QView{}, QView{}, nullptr, nullptr, nullptr, nullptr, 1, 0};
            pa.fp = fp;
            pa.bp = bp;
            pa.pool = work; // the first traversal launch's chunk counter, unused here (zeroed with the pass's counters)
            pa.cap = cap;
            pa.totals = bs.totals;
            if (ctx->opt.bsdf_mode == PT_BSDF_LAMBERT) hipLaunchKernelGGL((k_path_loop<PT_BSDF_LAMBERT>), dim3(grid), dim3(64), 0, bs.stream, pa);
            else hipLaunchKernelGGL((k_path_loop<PT_BSDF_DISNEY>), dim3(grid), dim3(64), 0, bs.stream, pa);
            ++lc.trace;
            ++lc.fused;
        } else if (async) {
            // The bounce chain holds closest-hit launches only; the shadow rays of bounce b are traced from their own records
            // on one of two side streams as soon as k_shade(b) has written them, and nothing waits for them before k_resolve.
            const size_t qsize = (size_t)PT_NSUB * ctx->sub_cap;
            std::vector<hipEvent_t> shadow_done;
            {
                SpanGuard g(ctx, CLS_TRACE, bs.stream);
                Trace8Args ta{stream_view(bs, sin), bvh8, qcur, QView{}, work, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                launch_closest(ctx, bs.stream, ta, tgrid, pass_paths);
                ++lc.trace;
            }
            for (int b = 0; b <= last_bounce; ++b) {
                QView qnext{qnext_base, cntA + (size_t)(b + 1) * CS, ctx->sub_cap};
                QView qshadow{bs.squeueB + (size_t)b * qsize, cntS + (size_t)b * CS, ctx->sub_cap};
                ShadeParams sp{shade_tris, ctx->d_tri_nrm, ctx->d_mats, ctx->d_mesh_tex, ctx->d_textris, ctx->d_textures, ctx->probe, ctx->opt.max_depth, tmin_rad, qcur, qnext, qshadow, bs.X[sin ^ 1].rayO, bs.X[sin ^ 1].rayD, bs.X[sin ^ 1].thr, bs.X[sin ^ 1].rf, job ? (job->var.write_aov && job->var.initial_depth == 0 ? 1 : 0) : 1, (b == 0) ? 1 : 0};
                {
                    SpanGuard g(ctx, CLS_SHADE, bs.stream);
                    if (ctx->opt.bsdf_mode == PT_BSDF_LAMBERT) launch_shade<PT_BSDF_LAMBERT>(ctx, bs, stream_view(bs, sin), sp);
                    else launch_shade<PT_BSDF_DISNEY>(ctx, bs, stream_view(bs, sin), sp);
                    ++lc.shade;
                }
                {
                    hipEvent_t ev_shaded = next_event(ctx);
                    hipEventRecord(ev_shaded, bs.stream);
                    hipStream_t ss = (b & 1) ? bs.stream3 : bs.stream2;
                    hipStreamWaitEvent(ss, ev_shaded, 0);
                    {
                        SpanGuard g(ctx, CLS_SHADOW, ss);
                        Trace8Args ta{stream_view(bs, sin), bvh8, qshadow, QView{}, work + (size_t)(nq + b) * PT_WSTRIDE, (b & 1) ? bs.ovf3 : bs.ovf2, cull, nullptr, b, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                        hipLaunchKernelGGL((k_trace8<TR_SHADOW_APPLY>), dim3(tgrid), dim3(64), 0, ss, ta);
                        ++lc.shadow;
                    }
                    hipEvent_t ev = next_event(ctx);
                    hipEventRecord(ev, ss);
                    shadow_done.push_back(ev);
                }
                if (b < last_bounce) {
                    SpanGuard g(ctx, CLS_TRACE, bs.stream);
                    Trace8Args ta{stream_view(bs, sin ^ 1), bvh8, qnext, QView{}, work + (size_t)(b + 1) * PT_WSTRIDE, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                    launch_closest(ctx, bs.stream, ta, tgrid, pass_paths);
                    ++lc.trace;
                }
                qcur = qnext;
                qnext_base = (qnext_base == bs.queueA) ? bs.queueB : bs.queueA;
                sin ^= 1;
            }
            for (hipEvent_t e : shadow_done) hipStreamWaitEvent(bs.stream, e, 0);
        } else if (unified) {
            // One traversal launch per bounce: the closest-hit rays of bounce b+1 and the shadow rays of bounce b share
            // a persistent kernel (per-lane ray type), so the long-ray tail of one kind is filled with rays of the other
            // and a frame has max_depth+1 traversal launches instead of 2*max_depth.
            {
                SpanGuard g(ctx, CLS_TRACE, bs.stream);
                Trace8Args ta{stream_view(bs, sin), bvh8, qcur, QView{}, work, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                launch_closest(ctx, bs.stream, ta, tgrid, pass_paths);
                ++lc.trace;
            }
            for (int b = 0; b <= last_bounce; ++b) {
                QView qnext{qnext_base, cntA + (size_t)(b + 1) * CS, ctx->sub_cap};
                QView qshadow{bs.squeue, cntS + (size_t)b * CS, ctx->sub_cap};
                ShadeParams sp{shade_tris, ctx->d_tri_nrm, ctx->d_mats, ctx->d_mesh_tex, ctx->d_textris, ctx->d_textures, ctx->probe, ctx->opt.max_depth, tmin_rad, qcur, qnext, qshadow, bs.X[sin ^ 1].rayO, bs.X[sin ^ 1].rayD, bs.X[sin ^ 1].thr, bs.X[sin ^ 1].rf, job ? (job->var.write_aov && job->var.initial_depth == 0 ? 1 : 0) : 1, (b == 0) ? 1 : 0};
                {
                    SpanGuard g(ctx, CLS_SHADE, bs.stream);
                    if (ctx->opt.bsdf_mode == PT_BSDF_LAMBERT) launch_shade<PT_BSDF_LAMBERT>(ctx, bs, stream_view(bs, sin), sp);
                    else launch_shade<PT_BSDF_DISNEY>(ctx, bs, stream_view(bs, sin), sp);
                    ++lc.shade;
                }
                if (b < last_bounce) {
                    SpanGuard g(ctx, CLS_TRACE, bs.stream);
                    Trace8Args ta{stream_view(bs, sin ^ 1), bvh8, qnext, qshadow, work + (size_t)(b + 1) * PT_WSTRIDE, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                    launch_trace8<TR_UNIFIED>(bs.stream, tgrid, ta);
                    ++lc.trace;
                } else {
                    SpanGuard g(ctx, CLS_SHADOW, bs.stream);
                    Trace8Args ta{stream_view(bs, sin), bvh8, qshadow, QView{}, work + (size_t)(nq + b) * PT_WSTRIDE, bs.ovf, cull, ctx->dbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                    launch_trace8<TR_SHADOW_APPLY>(bs.stream, tgrid, ta);
                    ++lc.shadow;
                }
                qcur = qnext;
                qnext_base = (qnext_base == bs.queueA) ? bs.queueB : bs.queueA;
                sin ^= 1;
            }
        } else
        for (int b = 0; b <= last_bounce; ++b) {
            QView qnext{qnext_base, cntA + (size_t)(b + 1) * CS, ctx->sub_cap};
            QView qshadow{bs.squeue, cntS + (size_t)b * CS, ctx->sub_cap};
            {
                SpanGuard g(ctx, CLS_TRACE, bs.stream);
                Trace8Args ta{stream_view(bs, sin), bvh8, qcur, QView{}, work + (size_t)b * PT_WSTRIDE, bs.ovf, cull, nullptr, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                launch_closest(ctx, bs.stream, ta, tgrid, pass_paths);
                ++lc.trace;
            }
            ShadeParams sp{shade_tris, ctx->d_tri_nrm, ctx->d_mats, ctx->d_mesh_tex, ctx->d_textris, ctx->d_textures, ctx->probe, ctx->opt.max_depth, tmin_rad, qcur, qnext, qshadow, bs.X[sin ^ 1].rayO, bs.X[sin ^ 1].rayD, bs.X[sin ^ 1].thr, bs.X[sin ^ 1].rf, job ? (job->var.write_aov && job->var.initial_depth == 0 ? 1 : 0) : 1, (b == 0) ? 1 : 0};
            if (ev_shadow_done) hipStreamWaitEvent(bs.stream, ev_shadow_done, 0); // shade overwrites what shadow(b-1) reads
            {
                SpanGuard g(ctx, CLS_SHADE, bs.stream);
                if (ctx->opt.bsdf_mode == PT_BSDF_LAMBERT) launch_shade<PT_BSDF_LAMBERT>(ctx, bs, stream_view(bs, sin), sp);
                else launch_shade<PT_BSDF_DISNEY>(ctx, bs, stream_view(bs, sin), sp);
                ++lc.shade;
            }
            {
                // shadow rays of this bounce run on the set's second stream, concurrently with the next bounce's
                // closest-hit traversal (both only read the ray arrays); the next shade waits for both
                hipEvent_t ev_shaded = next_event(ctx);
                hipEventRecord(ev_shaded, bs.stream);
                hipStreamWaitEvent(bs.stream2, ev_shaded, 0);
                SpanGuard g(ctx, CLS_SHADOW, bs.stream2);
                Trace8Args ta{stream_view(bs, sin), bvh8, qshadow, QView{}, work + (size_t)(nq + b) * PT_WSTRIDE, bs.ovf2, cull, nullptr, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                hipLaunchKernelGGL((k_trace8<TR_SHADOW_APPLY>), dim3(tgrid), dim3(64), 0, bs.stream2, ta);
                ++lc.shadow;
            }
            ev_shadow_done = next_event(ctx);
            hipEventRecord(ev_shadow_done, bs.stream2);
            qcur = qnext;
            qnext_base = (qnext_base == bs.queueA) ? bs.queueB : bs.queueA;
                sin ^= 1;
        }
        if (ev_shadow_done) hipStreamWaitEvent(bs.stream, ev_shadow_done, 0);
        if (ctx->has_catcher) {
            // A secondary hit on a shadow-catcher surface passes through WITHOUT consuming depth (--prd->depth, deviceProgram.cu:503-508),
            // so a path with k pass-throughs is traced max_depth+1+k times by the reference's raygen loop (:411-443).  The chain
            // above ran max_depth+1 iterations; the paths still queued are exactly those, and they are iterated until none is left
            // (a host read of one queue size per extra iteration — shadow-catcher scenes only).
            const int e0 = last_bounce + 1;
            int cur = e0;
            for (int guard = 0; guard < 4096; ++guard) {
                uint32_t hc[PT_NSUB * PT_CSTRIDE];
                if (hipMemcpyAsync(hc, cntA + (size_t)cur * CS, sizeof(hc), hipMemcpyDeviceToHost, bs.stream) != hipSuccess) break;
                if (hipStreamSynchronize(bs.stream) != hipSuccess) break;
                unsigned long long left = 0;
                for (int q = 0; q < PT_NSUB; ++q) left += hc[q * PT_CSTRIDE];
                if (left == 0) break;
                const int nxt = (cur == e0 + 1) ? e0 + 2 : e0 + 1;
                hipMemsetAsync(cntA + (size_t)nxt * CS, 0, sizeof(uint32_t) * CS, bs.stream);
                hipMemsetAsync(cntS + (size_t)cur * CS, 0, sizeof(uint32_t) * CS, bs.stream);
                hipMemsetAsync(work + (size_t)cur * PT_WSTRIDE, 0, sizeof(uint32_t) * PT_WSTRIDE, bs.stream);
                hipMemsetAsync(work + (size_t)(nq + cur) * PT_WSTRIDE, 0, sizeof(uint32_t) * PT_WSTRIDE, bs.stream);
                QView qnext{qnext_base, cntA + (size_t)nxt * CS, ctx->sub_cap};
                QView qshadow{bs.squeue, cntS + (size_t)cur * CS, ctx->sub_cap};
                {
                    SpanGuard g(ctx, CLS_TRACE, bs.stream);
                    Trace8Args ta{stream_view(bs, sin), bvh8, qcur, QView{}, work + (size_t)cur * PT_WSTRIDE, bs.ovf, cull, nullptr, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                    launch_closest(ctx, bs.stream, ta, tgrid, pass_paths);
                    ++lc.trace;
                }
                ShadeParams sp{shade_tris, ctx->d_tri_nrm, ctx->d_mats, ctx->d_mesh_tex, ctx->d_textris, ctx->d_textures, ctx->probe, ctx->opt.max_depth, tmin_rad, qcur, qnext, qshadow, bs.X[sin ^ 1].rayO, bs.X[sin ^ 1].rayD, bs.X[sin ^ 1].thr, bs.X[sin ^ 1].rf, 1, 0};
                {
                    SpanGuard g(ctx, CLS_SHADE, bs.stream);
                    if (ctx->opt.bsdf_mode == PT_BSDF_LAMBERT) launch_shade<PT_BSDF_LAMBERT>(ctx, bs, stream_view(bs, sin), sp);
                    else launch_shade<PT_BSDF_DISNEY>(ctx, bs, stream_view(bs, sin), sp);
                    ++lc.shade;
                }
                {
                    SpanGuard g(ctx, CLS_SHADOW, bs.stream);
                    Trace8Args ta{stream_view(bs, sin), bvh8, qshadow, QView{}, work + (size_t)(nq + cur) * PT_WSTRIDE, bs.ovf, cull, nullptr, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(bs), ctx->bvh.num_nodes8};
                    hipLaunchKernelGGL((k_trace8<TR_SHADOW_APPLY>), dim3(tgrid), dim3(64), 0, bs.stream, ta);
                    ++lc.shadow;
                }
                hipLaunchKernelGGL(k_accum_stats, dim3(1), dim3(64), 0, bs.stream, bs.counters + (size_t)cur * CS, nq, 1, 0, bs.totals);
                qcur = qnext;
                qnext_base = (qnext_base == bs.queueA) ? bs.queueB : bs.queueA;
                sin ^= 1;
                cur = nxt;
            }
        }
        // foveated launches: this launch's pixels are written after the previous launch's (they overlap); whole frames in flight: after the
        // previous frame's.  Only the pass that writes the frame buffers has to wait — earlier sample passes of a pixel chunk keep their sums in the
        // set's own pixResult arrays
        if (before_resolve && (job || (s0 % fp.spp) + Sc >= fp.spp)) // the pass completes a subframe
            for (hipEvent_t e : *before_resolve) hipStreamWaitEvent(bs.stream, e, 0);
        {
            SpanGuard g(ctx, CLS_OTHER, bs.stream);
            // counters[last_bounce+1] holds paths that would have continued: not traced, not counted
            if (!fused) hipLaunchKernelGGL(k_accum_stats, dim3(1), dim3(64), 0, bs.stream, bs.counters, nq, last_bounce + 1, 1, bs.totals);
            if (job)
                hipLaunchKernelGGL(k_resolve_region, dim3((job->nl + 255) / 256), dim3(256), 0, bs.stream, bs.st, fp, job->rg, PartParams{ctx->rank, ctx->world, ctx->tile_w, ctx->tile_h}, job->var, job->l0, job->nl);
            else
                hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, bs.stream, bs.st, fp, bp);
        }
    }
}

// pt_render in two halves, so that pt_multi_render can have every device working before it waits for any of them and so
// that consecutive frames can overlap (pt_options.frames_in_flight): render_enqueue launches the whole frame asynchronously
// into a frame slot, render_finish waits for that slot's frame and collects its statistics.
static void begin_slot(pt_ctx* ctx, int slot) {
    ctx->cur_slot = slot;
    ctx->ev_used = 0;
    ctx->spans.clear();
    for (size_t i = 0; i < ctx->sets.size(); ++i) ctx->sets[i].totals = totals_of(ctx, slot, (int)i);
}

// mode 0: synchronous frame (pixel chunks on all streams); 2: the same chunks without a frame-wide start; 3: the whole frame on one stream
// count > 1 (pt_render_batch): the launch chain carries the rays of `count` consecutive subframes — generate / trace / shade launches are
// count times as large, the resolve blends the subframes in order — and leaves the buffers `count` frames would have left.
static int render_enqueue(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, int slot = 0, int mode = 0, uint32_t count = 1) {
    const bool pipelined = mode != 0, whole = mode == 3;
    if (!pipelined) {
        int rc = drain(ctx);
        if (rc) return rc;
    }
    ctx->fr[slot].active = 0;
    if (ctx->width == 0) return PT_OK; // not resized yet (SimplePathtracer.cpp:77)
    if (spp == 0 || spp > 4096) return fail(ctx, PT_ERR_INVALID, "pt_render: samples_per_launch must be in [1,4096]");
    if (count == 0 || count > 4096 || (uint64_t)subframe_index + count > 0xffffffffull) return fail(ctx, PT_ERR_INVALID, "pt_render_batch: count must be in [1,4096] and the subframe indices must fit 32 bits");
    const uint32_t vspp = spp * count; // samples of a pixel over the whole batch
    if (!ctx->probe.data) return fail(ctx, PT_ERR_INVALID, "pt_render: no probe set (setProbe)");
    CK(hipSetDevice(ctx->device));
    const uint32_t owned = ctx->owned;
    // Chunking.  All samples of a pixel stay in one pixel chunk.  The frame is cut into (at least) `streams` pixel
    // chunks that run concurrently on separate stream pairs; a chunk holds at most max_paths/streams paths, so
    // samples are split when spp*pixels exceed that; shadow-catcher scenes run one sample per pass so that the
    // per-pixel normal/albedo sums keep the reference order.  None of this changes a bit of the result.
    // PT_FUSED=1: a frame small enough for the fused bounce loop (pt_fused.h) is ONE pass on one stream — the persistent waves of one fused
    // kernel fill the chip, so three chunk kernels would only run one after the other
    // (pt_options.streams set by the caller is respected: the chain on that many chunk streams).  Scenes whose rays are expensive and heavy-tailed keep
    // the chain: every round of every fused wave ends with its own slowest ray, and on the stadium (27 steps per calibration ray; terrain 14) a
    // 1/8 share runs 7-13 % slower fused while the terrain, the textured terrain and the Cornell box run 3-11 % faster (profiles/r5_14_fused_bounce_loop.md).
    // One scene family on either side of the threshold: a rule of thumb, PT_FUSED_MAX_COST moves it.
    const uint64_t frame_paths = (uint64_t)owned * vspp;
    // what the fused pass needs at all: the default schedule of a scene without shadow-catcher materials, the whole frame in one batch set
    const bool fused_ok = ctx->fused == 1 && !pipelined && owned > 0 && ctx->opt.streams <= 0 && !ctx->has_catcher && ctx->opt.split_shadow == 0 && frame_paths <= ctx->opt.max_paths;
    // round 5's rule of thumb, now the first guess: small frames of scenes whose calibration rays are cheap; a tree that was never calibrated
    // (builder forced or imported, challenger not built) keeps the chain unless the scene is tiny
    const bool cost_ok = ctx->bvh.calib_cost > 0.f ? ctx->bvh.calib_cost <= ctx->fused_max_cost : ctx->ntri < 4096u;
    const bool guess = fused_ok && cost_ok && frame_paths <= ctx->fused_max_paths;
    bool fused_one_pass = guess;
    ctx->sched_pending.valid = false;
    if (fused_ok && ctx->sched_trials > 0 && frame_paths <= ctx->sched_max_paths && !ctx->span_timing()) {
        const pt_ctx::SchedKey key{owned, vspp, ctx->opt.max_paths, ctx->opt.max_depth, ctx->opt.bsdf_mode, ctx->width, ctx->height, ctx->rank, ctx->world};
        pt_ctx::Sched& sc = ctx->sched;
        if (!(sc.key == key)) { sc = pt_ctx::Sched{}; sc.key = key; }
        const int first = ctx->sched_initial >= 0 ? ctx->sched_initial : (guess ? 1 : 0);
        int use = sc.choice;
        bool probe = false;
        if (sc.choice < 0) use = sc.n[first] <= sc.n[1 - first] ? first : 1 - first; // alternate, starting with the guess
        else if (ctx->sched_probe > 0 && sc.since % (uint64_t)ctx->sched_probe == (uint64_t)ctx->sched_probe - 1) { use = 1 - sc.choice; probe = true; }
        fused_one_pass = use == 1;
        ctx->sched_pending.valid = true;
        ctx->sched_pending.which = use;
        ctx->sched_pending.probe = probe;
    }
    if (!pipelined) ctx->sched_flags = (fused_one_pass ? 1u : 0u) | ((ctx->sched_pending.valid && ctx->sched.choice < 0) ? 0x100u : 0u);
    ctx->fused_frame = fused_one_pass;
    const int nsets = fused_one_pass ? 1 : std::max(1, std::min(PT_MAX_SETS, ctx->opt.streams > 0 ? ctx->opt.streams : 3));
    const uint32_t max_paths = std::max<uint32_t>(ctx->opt.max_paths, 64u);
    const uint32_t cap = std::max<uint32_t>(64u, whole ? max_paths : max_paths / nsets);
    uint32_t Np = whole ? owned : (owned + nsets - 1) / nsets; // pixels per chunk ...
    Np = std::min(cap, std::max(64u, (Np + 63u) & ~63u)); // ... whole 8x8 blocks, within the set capacity
    const uint32_t S = ctx->has_catcher ? 1u : std::max(1u, std::min(vspp, cap / Np));
    if (owned) {
        int rc = ensure_path_state(ctx, nsets, Np * S, Np); // waits for the frames in flight before it re-allocates
        if (rc) return rc;
    }
    if (getenv("PT_DEBUG_COUNTS")) {
        if (!ctx->dbg) CK(dalloc(&ctx->dbg, 64 + (size_t)8 * PT_WAVELOG_CAP)); // 64 counters + the per-wave log of PT_DEBUG_STATS builds
        CK(dclear(ctx->stream, ctx->dbg, 512));
    }
    begin_slot(ctx, slot);
    hipEvent_t ev_begin = next_event(ctx);
    if (!pipelined) {
        CK(hipMemsetAsync(totals_of(ctx, slot, 0), 0, sizeof(unsigned long long) * PT_MAX_SETS * 4, ctx->stream));
        CK(hipEventRecord(ev_begin, ctx->stream));
    }
    FrameParams fp{ctx->accum, ctx->frame, ctx->color, ctx->normal, ctx->albedo, ctx->width, ctx->height, subframe_index,
                   ctx->eye, ctx->U, ctx->V, ctx->W, spp, ctx->probe};
    LaunchCounts lc;
    hipStream_t end_stream = ctx->stream; // where the frame's counters are copied out and its end event is recorded
    if (owned && whole) {
        // the frame's chunks (one, unless spp x pixels exceed max_paths) all go to the stream of this frame; its resolve blends into
        // accum_buffer after the previous frame's
        pt_ctx::BatchSet& bs = ctx->sets[ctx->frame_seq % (uint64_t)nsets];
        CK(hipMemsetAsync(totals_of(ctx, slot, 0), 0, sizeof(unsigned long long) * PT_MAX_SETS * 4, bs.stream));
        CK(hipEventRecord(ev_begin, bs.stream));
        std::vector<hipEvent_t> before;
        if (ctx->ev_resolved) before.push_back(ctx->ev_resolved);
        if (ctx->ev_pack_guard) before.push_back(ctx->ev_pack_guard); // the hand-off of the previous frame reads the buffers this frame's resolve overwrites
        ctx->adapt_grid = true;
        for (uint32_t pix0 = 0; pix0 < owned; pix0 += Np)
            enqueue_chunk(ctx, bs, fp, pix0, std::min(Np, owned - pix0), vspp, S, lc, nullptr, before.empty() ? nullptr : &before);
        ctx->adapt_grid = false;
        hipEvent_t e = next_event(ctx);
        hipEventRecord(e, bs.stream);
        ctx->ev_resolved = e;
        ctx->resolved_kind = 1;
        end_stream = bs.stream; // the whole frame is on this stream: the context's own stream stays out of the way (one hardware queue fewer)
    } else if (owned) {
        if (!pipelined) {
            for (int i = 0; i < nsets; ++i) hipStreamWaitEvent(ctx->sets[i].stream, ev_begin, 0); // (the context may hold more sets than this frame uses)
        } else {
            // no frame-wide start: every set clears its own counters behind its own previous chunk and goes on
            for (int i = 0; i < nsets; ++i) CK(hipMemsetAsync(ctx->sets[i].totals, 0, sizeof(unsigned long long) * 4, ctx->sets[i].stream));
            CK(hipEventRecord(ev_begin, ctx->sets[0].stream));
        }
        // Chunk c of consecutive pt_render frames runs on the same stream, which orders their resolves; a foveated frame (pt_render_regions)
        // deals its launches to the streams differently, so after one of those every resolve waits for that frame's end
        std::vector<hipEvent_t> before;
        if (pipelined && ctx->ev_resolved && ctx->resolved_kind != 0) before.push_back(ctx->ev_resolved);
        if (pipelined && ctx->ev_pack_guard) before.push_back(ctx->ev_pack_guard); // (a synchronous frame starts behind the context's stream, which carries the pack)
        // A SMALL synchronous frame in the default schedule (a share of a partitioned image): chunk c's chain is enqueued by thread c, so that
        // all chains start together instead of one third / two thirds of the enqueue time apart.  Measured: a 1/8 share of C3 1.84 -> 1.78 ms;
        // 1/4 and 1/2 shares unchanged; the full frame 8.03 -> 8.16 ms (its chains are long, and the staggered start is what makes one chunk's
        // shade launches overlap another's traversal) — hence the size limit.  Each chunk runs on its own batch set and stream; nothing the
        // threads touch is shared but the error string (locked).  PT_ENQUEUE_THREADS=0 keeps one thread, =2 uses threads at every size.
        const uint32_t nchunks = (owned + Np - 1) / Np;
        const int threads_env = ctx->enqueue_threads;
        const bool small_frame = (uint64_t)owned * vspp <= (3u << 19); // 1.5 M paths
        const bool parallel = ctx->shade_cus == 0 && threads_env != 0 && (small_frame || threads_env == 2) && !pipelined && nchunks > 1 && nchunks <= (uint32_t)nsets && !ctx->span_timing() &&
                              ctx->opt.split_shadow == 0 && before.empty();
        if (parallel) {
            while (ctx->chunk_workers.size() + 1 < nchunks) {
                ctx->chunk_workers.emplace_back(new EnqueueWorker());
                EnqueueWorker* w = ctx->chunk_workers.back().get();
                w->th = std::thread(enqueue_worker_main, w, ctx->device);
            }
            std::vector<LaunchCounts> lcs(nchunks);
            for (uint32_t c = 1; c < nchunks; ++c) {
                const uint32_t pix0 = c * Np;
                enqueue_worker_post(ctx->chunk_workers[c - 1].get(), [&, c, pix0]() {
                    enqueue_chunk(ctx, ctx->sets[c], fp, pix0, std::min(Np, owned - pix0), vspp, S, lcs[c]);
                    return 0;
                });
            }
            enqueue_chunk(ctx, ctx->sets[0], fp, 0, std::min(Np, owned), vspp, S, lcs[0]);
            int wrc = 0;
            for (uint32_t c = 1; c < nchunks; ++c) wrc |= enqueue_worker_wait(ctx->chunk_workers[c - 1].get());
            if (wrc != 0) return fail(ctx, PT_ERR_HIP, "pt_render: a chunk's enqueue thread failed");
            for (const LaunchCounts& l : lcs) { lc.trace += l.trace; lc.shadow += l.shadow; lc.shade += l.shade; lc.fused += l.fused; }
        } else {
            uint32_t k = 0;
            for (uint32_t pix0 = 0; pix0 < owned; pix0 += Np, ++k)
                enqueue_chunk(ctx, ctx->sets[k % nsets], fp, pix0, std::min(Np, owned - pix0), vspp, S, lc, nullptr, before.empty() ? nullptr : &before);
        }
        for (int i = 0; i < nsets; ++i) {
            hipEvent_t e = next_event(ctx);
            hipEventRecord(e, ctx->sets[i].stream);
            hipStreamWaitEvent(ctx->stream, e, 0);
        }
    } else if (pipelined) {
        CK(hipEventRecord(ev_begin, ctx->stream));
    }
    CK(hipMemcpyAsync(ctx->h_totals + (size_t)slot * PT_MAX_SETS * 4, totals_of(ctx, slot, 0), sizeof(unsigned long long) * PT_MAX_SETS * 4, hipMemcpyDeviceToHost, end_stream));
    hipEvent_t ev_end = next_event(ctx);
    CK(hipEventRecord(ev_end, end_stream));
    if (pipelined && !whole) { // a later foveated frame (or a whole frame) orders its resolves behind this frame's end
        ctx->ev_resolved = ev_end;
        ctx->resolved_kind = 0;
    }
    pt_ctx::Inflight& fr = ctx->fr[slot];
    fr.ev_begin = ev_begin;
    fr.ev_end = ev_end;
    fr.active = 1;
    fr.paths = (uint64_t)owned * vspp;
    fr.subframes = count;
    fr.lc = lc;
    fr.seq = ++ctx->frame_seq;
    return PT_OK;
}

static int render_finish(pt_ctx* ctx, int slot = 0) {
    pt_ctx::Inflight& fr = ctx->fr[slot];
    if (!fr.active) return PT_OK;
    fr.active = 0;
    CK(hipSetDevice(ctx->device));
    const uint32_t owned = ctx->owned;
    const LaunchCounts lc = fr.lc;
    hipEvent_t ev_begin = fr.ev_begin, ev_end = fr.ev_end;
    CK(hipEventSynchronize(ev_end)); // SimplePathtracer.cpp:96 CUDA_SYNC_CHECK (the frame's last event on the context's stream)
    if (!ctx->launch_err.empty()) {
        ctx->err = ctx->launch_err;
        ctx->launch_err.clear();
        return PT_ERR_HIP;
    }
    CK(hipGetLastError());
    const unsigned long long* per_set = ctx->h_totals + (size_t)slot * PT_MAX_SETS * 4; // copied behind the frame's last kernel, before ev_end
    unsigned long long totals[4] = {0, 0, 0, 0};
    for (int i = 0; i < PT_MAX_SETS; ++i) {
        totals[0] += per_set[i * 4 + 0];
        totals[1] += per_set[i * 4 + 1];
        totals[2] |= per_set[i * 4 + 2];
        totals[3] += per_set[i * 4 + 3];
    }
    if (totals[2] & 1ull) return fail(ctx, PT_ERR_UNSUPPORTED, "traversal stack overflow: the acceleration structure is deeper than the traversal stack; the frame is invalid");
    if (ctx->dbg) {
        unsigned long long h[64];
        CK(hipMemcpy(h, ctx->dbg, 512, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pt_render] rays by traversal steps (2^k..): closest");
        for (int k = 0; k < 14; ++k) fprintf(stderr, " %llu", h[16 + k]);
        fprintf(stderr, " | shadow");
        for (int k = 0; k < 14; ++k) fprintf(stderr, " %llu", h[32 + k]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[pt_render] traversal: node steps %llu, tri tests %llu, max steps of one ray %llu, max wave loop iterations %llu, mean %.1f\n", h[0], h[1], h[4], h[5], h[7] ? (double)h[6] / h[7] : 0.0);
        fprintf(stderr, "[pt_render] per loop iteration: lanes holding a ray %.1f / 64, lanes executing the chosen step %.1f / 64, node-step iterations %.1f %%\n",
                h[6] ? (double)h[8] / h[6] : 0.0, h[6] ? (double)h[9] / h[6] : 0.0, h[6] ? 100.0 * h[10] / h[6] : 0.0);
        if (const char* wl = getenv("PT_WAVELOG")) { // PT_DEBUG_STATS builds: per-wave log of the frame's traversal launches
            const size_t nlog = (size_t)std::min<unsigned long long>(h[63], PT_WAVELOG_CAP);
            std::vector<unsigned long long> log(8 * nlog);
            if (nlog) CK(hipMemcpy(log.data(), ctx->dbg + 64, log.size() * 8, hipMemcpyDeviceToHost));
            if (FILE* f = fopen(wl, "wb")) {
                fwrite(log.data(), 8, log.size(), f);
                fclose(f);
            }
        }
        if (h[53] + h[54] + h[55] + h[56]) // -DPT_DEBUG_WAVELOG=3 builds
            fprintf(stderr, "[pt_render] traversal iterations by lanes active at their start (<=8 | <=16 | <=32 | more): %llu %llu %llu %llu; cycles: %llu %llu %llu %llu\n",
                    h[53], h[54], h[55], h[56], h[57], h[58], h[59], h[60]);
        if (h[48])
            fprintf(stderr, "[pt_render] camera packets %llu: node steps %.1f per packet (%.1f lanes hit something), triangle tests %.1f per packet (%.1f lanes inside the leaf's box)\n",
                    h[48], (double)h[49] / h[48], h[49] ? (double)h[52] / h[49] : 0.0, (double)h[50] / h[48], h[50] ? (double)h[51] / h[50] : 0.0);
    }
    if (getenv("PT_DEBUG_COUNTS") && owned) { // per-bounce queue sizes of the last chunk of set 0
        const size_t CS = (size_t)PT_NSUB * PT_CSTRIDE;
        std::vector<uint32_t> hc((size_t)2 * ctx->nq * CS);
        CK(hipMemcpy(hc.data(), ctx->sets[0].counters, sizeof(uint32_t) * hc.size(), hipMemcpyDeviceToHost));
        fprintf(stderr, "[pt_render] rays per bounce (radiance/shadow):");
        for (int b = 0; b < ctx->nq; ++b) {
            unsigned long long r = 0, sh = 0;
            for (int q = 0; q < PT_NSUB; ++q) {
                r += hc[(size_t)b * CS + q * PT_CSTRIDE];
                sh += hc[(size_t)(ctx->nq + b) * CS + q * PT_CSTRIDE];
            }
            fprintf(stderr, " %llu/%llu", r, sh);
        }
        fprintf(stderr, "\n");
    }
    pt_stats& st = ctx->stats;
    st.radiance_rays = totals[0];
    st.shadow_rays = totals[1];
    st.shaded_hits = totals[3];
    st.paths = fr.paths;
    ctx->cum_radiance += totals[0];
    ctx->cum_shadow += totals[1];
    ctx->cum_frames += fr.subframes;
    float ms = 0;
    hipEventElapsedTime(&ms, ev_begin, ev_end);
    st.render_ms = ms;
    if (ctx->sched_pending.valid && slot == 0) { // the frame took part in the chain-against-fused measurement (render_enqueue)
        ctx->sched_pending.valid = false;
        pt_ctx::Sched& sc = ctx->sched;
        const int w = ctx->sched_pending.which;
        const double t = ctx->sched_fake[0] > 0 ? ctx->sched_fake[w] : (double)ms;
        if (sc.choice < 0) {
            if (sc.n[w] > 0) sc.best[w] = sc.best[w] > 0 ? std::min(sc.best[w], t) : t; // (the first frame of each schedule is warm-up: code objects, cold caches, first-touch of the path state)
            ++sc.n[w];
            if (sc.n[0] > ctx->sched_trials && sc.n[1] > ctx->sched_trials) {
                sc.choice = sc.best[1] < sc.best[0] ? 1 : 0;
                sc.since = 0;
                sc.mean = sc.best[sc.choice];
            }
        } else if (ctx->sched_pending.probe) {
            ++sc.since;
            if (t < 0.95 * sc.mean) { // the loser is clearly ahead now: time both again (they are warm: no warm-up frame)
                sc.choice = -1;
                sc.n[0] = sc.n[1] = 1;
                sc.best[0] = sc.best[1] = 0;
            }
        } else {
            ++sc.since;
            sc.mean = 0.9 * sc.mean + 0.1 * t;
        }
    }
    st.schedule = ctx->sched_flags;
    st.sched_chain_ms = ctx->sched.best[0];
    st.sched_fused_ms = ctx->sched.best[1];
    double cls_ms[4] = {0, 0, 0, 0};
    for (auto& sp : ctx->spans) { // kernel timing implies synchronous frames: the spans are those of this frame
        float m = 0;
        hipEventElapsedTime(&m, ctx->ev_pools[slot][sp.a], ctx->ev_pools[slot][sp.b]);
        cls_ms[sp.cls] += m;
    }
    st.trace_ms = cls_ms[CLS_TRACE]; // sums over concurrent streams: they overlap, so they can exceed render_ms
    st.shadow_ms = cls_ms[CLS_SHADOW];
    st.shade_ms = cls_ms[CLS_SHADE];
    st.other_ms = cls_ms[CLS_OTHER];
    st.trace_launches = lc.trace;
    st.shadow_launches = lc.shadow;
    st.shade_launches = lc.shade;
    st.fused_passes = lc.fused;
    return PT_OK;
}

static int oldest_active(pt_ctx* ctx) {
    int o = -1;
    for (int i = 0; i < PT_MAX_FRAMES; ++i)
        if (ctx->fr[i].active && (o < 0 || ctx->fr[i].seq < ctx->fr[o].seq)) o = i;
    return o;
}

// waits for every frame in flight, oldest first
static int drain(pt_ctx* ctx) {
    int rc = PT_OK;
    for (int o; (o = oldest_active(ctx)) >= 0;) {
        const int r = render_finish(ctx, o);
        if (rc == PT_OK) rc = r;
    }
    ctx->ev_resolved = nullptr; // nothing in flight: nothing to order the next resolve behind
    ctx->last_slot = -1;
    if (ctx->ev_pack_guard) {
        if (hipEventQuery(ctx->ev_pack_guard) == hipSuccess) ctx->ev_pack_guard = nullptr;
        else (void)hipGetLastError(); // "not ready" is an answer, not an error to be found by the next hipGetLastError
    }
    return rc;
}

extern "C" int pt_sync(pt_ctx* ctx) { return ctx ? drain(ctx) : PT_ERR_INVALID; }

// frames in flight asked for and possible (kernel timing and the debug counters read per-frame state that is not kept per slot)
static int frames_mode(pt_ctx* ctx) {
    return (ctx->opt.frames_in_flight >= 2 && !ctx->span_timing() && !getenv("PT_DEBUG_COUNTS")) ? std::min(ctx->opt.frames_in_flight, PT_MAX_FRAMES) : 1;
}
// frame k goes into the slot the oldest finished frame has left ...
static int pipelined_enqueue(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, int F, uint32_t count = 1) {
    const int slot = (ctx->last_slot + 1) % F;
    int rc = render_finish(ctx, slot);
    if (rc == PT_OK) rc = render_enqueue(ctx, spp, subframe_index, slot, F, count);
    if (rc == PT_OK) ctx->last_slot = slot;
    return rc;
}
// ... then wait until at most F-1 frames are in flight (a frame's errors are reported by the call that waits for it)
static int pipelined_wait(pt_ctx* ctx, int F) {
    int rc = PT_OK;
    for (;;) {
        int n = 0;
        for (int i = 0; i < PT_MAX_FRAMES; ++i) n += ctx->fr[i].active;
        if (n < F) break;
        const int r = render_finish(ctx, oldest_active(ctx));
        if (rc == PT_OK) rc = r;
    }
    return rc;
}

extern "C" int pt_render(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, uint32_t* host_rgba8) {
    return pt_render_batch(ctx, spp, subframe_index, 1, host_rgba8);
}

extern "C" int pt_render_batch(pt_ctx* ctx, uint32_t spp, uint32_t first_subframe, uint32_t count, uint32_t* host_rgba8) {
    if (!ctx) return PT_ERR_INVALID;
    const int F = frames_mode(ctx);
    const uint32_t subframe_index = first_subframe;
    int rc;
    if (F == 1) {
        rc = drain(ctx);
        if (rc == PT_OK) rc = render_enqueue(ctx, spp, subframe_index, 0, 0, count);
        if (rc == PT_OK) rc = render_finish(ctx);
    } else {
        rc = pipelined_enqueue(ctx, spp, subframe_index, F, count);
        const int rw = pipelined_wait(ctx, F);
        if (rc == PT_OK) rc = rw;
    }
    if (rc != PT_OK) return rc;
    if (host_rgba8 && ctx->width) return pt_download(ctx, PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)ctx->width * ctx->height);
    return PT_OK;
}


// render(sutil::CUDAOutputBuffer<uint32_t>&) (SimplePathtracer.cpp:99-107): the reference points frame_buffer at the caller's mapped DEVICE
// buffer for the launch.  Here the frame is rendered into the context's own frame buffer and copied device-to-device (8 MB at 1080p: a few
// microseconds), so pt_download / the display hand-off keep seeing the frame too; the call returns when the copy is complete.
extern "C" int pt_render_device(pt_ctx* ctx, uint32_t spp, uint32_t subframe_index, void* dev_rgba8) {
    if (!ctx || !dev_rgba8) return PT_ERR_INVALID;
    {
        // the pointer must be memory the GPU copy engine can write: device or managed memory, or host memory registered / allocated through
        // HIP.  A plain malloc'ed buffer (what SampleRenderer::render(uint32_t*) took before round 4) is refused BEFORE the frame is rendered.
        CK(hipSetDevice(ctx->device));
        hipPointerAttribute_t at;
        const hipError_t pe = hipPointerGetAttributes(&at, dev_rgba8);
        if (pe != hipSuccess || at.type == hipMemoryTypeUnregistered) {
            (void)hipGetLastError(); // an unknown pointer leaves hipErrorInvalidValue behind
            return fail(ctx, PT_ERR_INVALID, "pt_render_device: dev_rgba8 is not device memory (for a host buffer use pt_render / renderToHost)");
        }
    }
    int rc = pt_render_batch(ctx, spp, subframe_index, 1, nullptr);
    if (rc == PT_OK) rc = drain(ctx); // frames in flight: this frame has to be complete
    if (rc != PT_OK) return rc;
    if (ctx->width == 0) return PT_OK; // not resized yet: render() silently returns (:77)
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(dev_rgba8, ctx->frame, sizeof(uint32_t) * (size_t)ctx->width * ctx->height, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// SampleRenderer::stream (SimplePathtracer.h:107; main.cpp:245 hands it to the display path): the stream the context's epilogues, packs and
// unpacks run on.  hipStreamNonBlocking: it does not synchronise with the null stream.
extern "C" void* pt_stream(pt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int pt_wait_event(pt_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return PT_ERR_INVALID;
    CK(hipSetDevice(ctx->device));
    CK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return PT_OK;
}

// The foveated variants' render() (HelloPathtracing_sv4_vmv23/SimplePathtracer.cpp:132-216) issues up to three
// optixLaunch calls per frame with different LaunchParams.frame.{factor,fillSize,c,r_inner,r_outer,offset,redraw},
// samples_per_launch and subframe_index; later launches overwrite the pixels of earlier ones, so the launches run
// in order on one stream.
// pipelined = false: a synchronous frame (slot 0); true (frames_in_flight >= 2): the same schedule without a frame-wide start — the
// passes of the frame's launches go to the batch sets in turn, every stream carries on behind its own previous work, and the first
// resolve of the frame waits for the end of the previous frame (later frames overwrite earlier ones)
static int regions_enqueue(pt_ctx* ctx, const pt_region* regions, uint32_t n, const pt_variant* variant, int slot = 0, bool pipelined = false) {
    if (!pipelined) {
        int rc = drain(ctx);
        if (rc) return rc;
    }
    ctx->fr[slot].active = 0;
    if (ctx->width == 0) return PT_OK;
    if (!ctx->probe.data) return fail(ctx, PT_ERR_INVALID, "pt_render_regions: no probe set (setProbe)");
    if (ctx->has_catcher) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_regions: shadow-catcher materials are not supported in foveated launches");
    CK(hipSetDevice(ctx->device));
    VariantParams var{0.001f, 0, 0, 1.0f, 1.0f, 0, 0};
    if (variant) var = VariantParams{variant->radiance_tmin, variant->cull_back_occlusion, variant->tonemap, variant->exposure, variant->white, variant->initial_depth, variant->write_aov};
    if (var.initial_depth < 0 || (var.initial_depth > 0 && var.initial_depth >= ctx->opt.max_depth)) return fail(ctx, PT_ERR_INVALID, "pt_render_regions: initial_depth must be 0 or below max_depth");
    const uint32_t max_paths = std::max<uint32_t>(ctx->opt.max_paths, 64u);
    // The launches of one frame are independent until they write pixels: their passes are dealt round-robin to the batch
    // sets (separate streams, so one launch's long-ray tails overlap the others' work) and only the resolves are ordered —
    // a launch's pixels are written after all pixels of the previous launch, as later launches overwrite earlier ones.
    const int nsets = std::max(1, std::min(16, ctx->opt.streams > 0 ? ctx->opt.streams : 3));
    const uint32_t cap = std::max<uint32_t>(64u, max_paths / nsets);
    uint32_t need = 64;
    for (uint32_t r = 0; r < n; ++r) {
        const pt_region& g = regions[r];
        if (g.spp == 0 || g.spp > 4096 || g.launch_w == 0 || g.launch_h == 0 || g.fill_size < 0 || g.fill_size > 64 ||
            (unsigned long long)g.launch_w * g.launch_h >= (1ull << 31))
            return fail(ctx, PT_ERR_INVALID, "pt_render_regions: bad region");
        if (g.spp > cap) return fail(ctx, PT_ERR_INVALID, "pt_render_regions: samples_per_launch exceeds max_paths / streams");
        const unsigned long long total = (unsigned long long)g.launch_w * g.launch_h * g.spp;
        need = (uint32_t)std::max<unsigned long long>(need, std::min<unsigned long long>(total, cap));
    }
    {
        int rc = ensure_path_state(ctx, nsets, need, 64);
        if (rc) return rc;
    }
    begin_slot(ctx, slot);
    hipEvent_t ev_begin = next_event(ctx);
    if (!pipelined) {
        CK(hipMemsetAsync(totals_of(ctx, slot, 0), 0, sizeof(unsigned long long) * PT_MAX_SETS * 4, ctx->stream));
        CK(hipEventRecord(ev_begin, ctx->stream));
        for (int i = 0; i < nsets; ++i) hipStreamWaitEvent(ctx->sets[i].stream, ev_begin, 0);
    } else {
        for (int i = 0; i < nsets; ++i) CK(hipMemsetAsync(ctx->sets[i].totals, 0, sizeof(unsigned long long) * 4, ctx->sets[i].stream));
        CK(hipEventRecord(ev_begin, ctx->sets[0].stream));
    }
    LaunchCounts lc;
    uint64_t paths = 0;
    uint32_t next_set = 0;
    std::vector<hipEvent_t> prev_done, cur_done;
    if (pipelined && ctx->ev_resolved) prev_done.push_back(ctx->ev_resolved); // the previous frame's pixels are written first
    if (pipelined && ctx->ev_pack_guard) prev_done.push_back(ctx->ev_pack_guard);
    for (uint32_t r = 0; r < n; ++r) {
        const pt_region& g = regions[r];
        FrameParams fp{ctx->accum, ctx->frame, ctx->color, ctx->normal, ctx->albedo, ctx->width, ctx->height, g.subframe_index,
                       ctx->eye, ctx->U, ctx->V, ctx->W, g.spp, ctx->probe};
        RegionJob job;
        job.rg = RegionParams{g.launch_w, g.launch_h, g.factor_x, g.factor_y, g.fill_size, g.cx, g.cy, g.r_inner, g.r_outer, g.offset_x, g.offset_y, g.redraw, g.spp, g.subframe_index};
        job.var = var;
        const uint32_t nlaunch = g.launch_w * g.launch_h;
        const uint32_t per = std::max(1u, ctx->set_cap / g.spp); // launch indices per pass (all their samples together)
        std::vector<bool> used(ctx->sets.size(), false);
        for (uint32_t l0 = 0; l0 < nlaunch; l0 += per) {
            job.l0 = l0;
            job.nl = std::min(per, nlaunch - l0);
            const uint32_t si = next_set++ % (uint32_t)nsets;
            enqueue_chunk(ctx, ctx->sets[si], fp, 0, 0, g.spp, g.spp, lc, &job, prev_done.empty() ? nullptr : &prev_done);
            used[si] = true;
        }
        cur_done.clear();
        for (size_t si = 0; si < ctx->sets.size(); ++si)
            if (used[si]) {
                hipEvent_t e = next_event(ctx);
                hipEventRecord(e, ctx->sets[si].stream);
                cur_done.push_back(e);
            }
        prev_done = cur_done; // these resolves waited for the older launches themselves: the order is transitive
        paths += (uint64_t)nlaunch * g.spp;
    }
    for (int i = 0; i < nsets; ++i) {
        hipEvent_t e = next_event(ctx);
        hipEventRecord(e, ctx->sets[i].stream);
        hipStreamWaitEvent(ctx->stream, e, 0);
    }
    CK(hipMemcpyAsync(ctx->h_totals + (size_t)slot * PT_MAX_SETS * 4, totals_of(ctx, slot, 0), sizeof(unsigned long long) * PT_MAX_SETS * 4, hipMemcpyDeviceToHost, ctx->stream));
    hipEvent_t ev_end = next_event(ctx);
    CK(hipEventRecord(ev_end, ctx->stream));
    if (pipelined) { // behind every resolve of this frame
        ctx->ev_resolved = ev_end;
        ctx->resolved_kind = 2;
    }
    pt_ctx::Inflight& fr = ctx->fr[slot];
    fr.subframes = 1;
    fr.ev_begin = ev_begin;
    fr.ev_end = ev_end;
    fr.active = 1;
    fr.paths = paths;
    fr.lc = lc;
    fr.seq = ++ctx->frame_seq;
    return PT_OK;
}

extern "C" int pt_render_regions(pt_ctx* ctx, const pt_region* regions, uint32_t n, const pt_variant* variant, uint32_t* host_rgba8) {
    if (!ctx || (!regions && n)) return PT_ERR_INVALID;
    const int F = frames_mode(ctx);
    int rc;
    if (F < 2) {
        rc = regions_enqueue(ctx, regions, n, variant);
        if (rc == PT_OK) rc = render_finish(ctx);
    } else {
        const int slot = (ctx->last_slot + 1) % F;
        rc = render_finish(ctx, slot);
        if (rc == PT_OK) rc = regions_enqueue(ctx, regions, n, variant, slot, true);
        if (rc == PT_OK) ctx->last_slot = slot;
        const int rw = pipelined_wait(ctx, F);
        if (rc == PT_OK) rc = rw;
    }
    if (rc != PT_OK) return rc;
    if (host_rgba8 && ctx->width) return pt_download(ctx, PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)ctx->width * ctx->height);
    return PT_OK;
}

static void* buffer_ptr(pt_ctx* ctx, int which, size_t* elem) {
    switch (which) {
        case PT_BUF_ACCUM: *elem = 16; return ctx->accum;
        case PT_BUF_FRAME: *elem = 4; return ctx->frame;
        case PT_BUF_COLOR: *elem = 16; return ctx->color;
        case PT_BUF_NORMAL: *elem = 16; return ctx->normal;
        case PT_BUF_ALBEDO: *elem = 16; return ctx->albedo;
        case PT_BUF_DENOISED: *elem = 16; return ctx->denoised;
    }
    *elem = 0;
    return nullptr;
}

extern "C" void* pt_device_buffer(pt_ctx* ctx, int which) {
    if (!ctx) return nullptr;
    if (drain(ctx) != PT_OK) return nullptr; // the caller is about to read or write the buffer
    size_t e;
    return buffer_ptr(ctx, which, &e);
}

extern "C" int pt_download(pt_ctx* ctx, int which, void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    size_t elem;
    void* p = buffer_ptr(ctx, which, &elem);
    if (!p) return fail(ctx, PT_ERR_INVALID, "pt_download: unknown buffer or not resized");
    if (bytes != elem * (size_t)ctx->width * ctx->height) return fail(ctx, PT_ERR_INVALID, "pt_download: byte count does not match the frame size");
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpy(host, p, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_upload_accum(pt_ctx* ctx, const float* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (!ctx->accum || bytes != 16 * (size_t)ctx->width * ctx->height) return fail(ctx, PT_ERR_INVALID, "pt_upload_accum: byte count does not match the frame size");
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpy(ctx->accum, host, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}

extern "C" int pt_tonemap_sqrt(pt_ctx* ctx, uint32_t* host_rgba8) {
    if (!ctx) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (ctx->width == 0) return PT_OK;
    CK(hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t)ctx->width * ctx->height;
    hipLaunchKernelGGL(k_tonemap_sqrt, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->accum, ctx->frame, n);
    CK(hipStreamSynchronize(ctx->stream));
    if (host_rgba8) return pt_download(ctx, PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)n);
    return PT_OK;
}

// OptiXDenoiser::exec() + the computeFinalPixelColors call of render(target) (SimplePathtracer.cpp:104-105)
extern "C" int pt_denoise(pt_ctx* ctx, const pt_denoise_params* prm, uint32_t* host_rgba8, double* kernel_ms) {
    if (!ctx || !prm) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (ctx->width == 0) return PT_OK;
    if (prm->iterations < 0 || prm->iterations > 8) return fail(ctx, PT_ERR_INVALID, "pt_denoise: iterations must be in [0,8]");
    if (!(prm->sigma_color > 0.f) || !(prm->sigma_normal > 0.f) || !(prm->sigma_albedo > 0.f)) return fail(ctx, PT_ERR_INVALID, "pt_denoise: sigmas must be positive");
    if (prm->input != PT_BUF_COLOR && prm->input != PT_BUF_ACCUM) return fail(ctx, PT_ERR_INVALID, "pt_denoise: input must be PT_BUF_COLOR or PT_BUF_ACCUM");
    if (prm->epilogue < 0 || prm->epilogue > 2) return fail(ctx, PT_ERR_INVALID, "pt_denoise: unknown epilogue");
    CK(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->width * ctx->height;
    if (!ctx->denoised) {
        CK(dalloc(&ctx->denoised, n));
        CK(dalloc(&ctx->denoise_tmp, n));
    }
    DevScope tmp;
    hipEvent_t e0, e1;
    CK(tmp.event(&e0));
    CK(tmp.event(&e1));
    const float4* src = prm->input == PT_BUF_COLOR ? ctx->color : ctx->accum;
    CK(hipEventRecord(e0, ctx->stream));
    if (prm->iterations == 0) CK(hipMemcpyAsync(ctx->denoised, src, sizeof(float4) * n, hipMemcpyDeviceToDevice, ctx->stream));
    // ping-pong so that the last pass lands in `denoised`
    float4* bufs[2] = {ctx->denoised, ctx->denoise_tmp};
    int cur = (prm->iterations & 1) ? 0 : 1;
    const dim3 grid((ctx->width + 31) / 32, (ctx->height + 7) / 8), block(256);
    for (int i = 0; i < prm->iterations; ++i) {
        const float sc = prm->sigma_color / (float)(1 << i), sn = prm->sigma_normal * (float)(1 << i);
        AtrousParams ap{ctx->width, ctx->height, 1 << i, 1.0f / (sc * sc), 1.0f / (sn * sn), 1.0f / (prm->sigma_albedo * prm->sigma_albedo)};
        hipLaunchKernelGGL(k_atrous, grid, block, 0, ctx->stream, src, ctx->normal, ctx->albedo, bufs[cur], ap);
        src = bufs[cur];
        cur ^= 1;
    }
    if (prm->epilogue == 1)
        hipLaunchKernelGGL(k_tonemap_sqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->denoised, ctx->frame, (uint32_t)n);
    else if (prm->epilogue == 2)
        hipLaunchKernelGGL(k_make_color, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->denoised, ctx->frame, (uint32_t)n);
    CK(hipEventRecord(e1, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipGetLastError());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (kernel_ms) *kernel_ms = ms;
    if (host_rgba8 && prm->epilogue) return pt_download(ctx, PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * n);
    return PT_OK;
}

extern "C" int pt_owned_pixels(const pt_ctx* ctx, uint32_t* owned, uint32_t* padded) {
    if (!ctx) return PT_ERR_INVALID;
    if (owned) *owned = ctx->owned;
    if (padded) *padded = ctx->padded;
    return PT_OK;
}

static int pack_launch(pt_ctx* ctx, int which, void* dev_dst);
static int unpack_launch(pt_ctx* ctx, int which, const void* dev_src_all);
extern "C" int pt_pack(pt_ctx* ctx, int which, void* dev_dst) {
    if (!ctx || !dev_dst) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    int rc = pack_launch(ctx, which, dev_dst);
    if (rc != PT_OK) return rc;
    CK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}
extern "C" int pt_unpack(pt_ctx* ctx, int which, const void* dev_src_all) {
    if (!ctx || !dev_src_all) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    int rc = unpack_launch(ctx, which, dev_src_all);
    if (rc != PT_OK) return rc;
    CK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}
static int newest_active(pt_ctx* ctx) {
    int o = -1;
    for (int i = 0; i < PT_MAX_FRAMES; ++i)
        if (ctx->fr[i].active && (o < 0 || ctx->fr[i].seq > ctx->fr[o].seq)) o = i;
    return o;
}
static size_t buffer_elem(int which) { return which == PT_BUF_FRAME ? 4 : ((which >= 0 && which <= PT_BUF_DENOISED) ? 16 : 0); }

// ---- overlapped display hand-off (include/pt_amd.h): nothing here waits for a frame on the host
static int pack_async_enqueue(pt_ctx* ctx, int which, void* dev_dst, int slot) {
    if (slot < 0 || slot > 1) return fail(ctx, PT_ERR_INVALID, "pt_pack_async: slot must be 0 or 1");
    CK(hipSetDevice(ctx->device));
    const int nf = newest_active(ctx);
    if (nf >= 0) CK(hipStreamWaitEvent(ctx->stream, ctx->fr[nf].ev_end, 0)); // the newest frame enqueued; earlier ones end before it writes the buffers
    int rc = pack_launch(ctx, which, dev_dst);
    if (rc != PT_OK) return rc;
    if (!ctx->ev_packed[slot]) CK(hipEventCreateWithFlags(&ctx->ev_packed[slot], hipEventDisableTiming));
    CK(hipEventRecord(ctx->ev_packed[slot], ctx->stream));
    ctx->ev_pack_guard = ctx->ev_packed[slot];
    return PT_OK;
}
extern "C" int pt_pack_async(pt_ctx* ctx, int which, void* dev_dst, int slot) {
    if (!ctx || !dev_dst) return PT_ERR_INVALID;
    return pack_async_enqueue(ctx, which, dev_dst, slot);
}
extern "C" int pt_pack_wait(pt_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot > 1) return PT_ERR_INVALID;
    if (!ctx->ev_packed[slot]) return PT_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipEventSynchronize(ctx->ev_packed[slot]));
    return PT_OK;
}
static int unpack_display_enqueue(pt_ctx* ctx, int which, const void* dev_src_all) {
    const size_t elem = buffer_elem(which);
    if (!elem || ctx->width == 0) return fail(ctx, PT_ERR_INVALID, "pt_unpack_display: unknown buffer or not resized");
    CK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)ctx->width * ctx->height;
    if (!ctx->display[which]) {
        CK(hipMalloc(&ctx->display[which], npx * elem));
        CK(hipMemsetAsync(ctx->display[which], 0, npx * elem, ctx->stream));
    }
    const uint32_t n = ctx->padded * (uint32_t)ctx->world;
    if (n) {
        if (elem == 16)
            hipLaunchKernelGGL((k_unpack<float4>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (float4*)ctx->display[which], ctx->d_all_pixels, n, ctx->width, (const float4*)dev_src_all);
        else
            hipLaunchKernelGGL((k_unpack<uint32_t>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t*)ctx->display[which], ctx->d_all_pixels, n, ctx->width, (const uint32_t*)dev_src_all);
    }
    if (!ctx->ev_displayed) CK(hipEventCreateWithFlags(&ctx->ev_displayed, hipEventDisableTiming));
    CK(hipEventRecord(ctx->ev_displayed, ctx->stream));
    return PT_OK;
}
extern "C" int pt_unpack_display(pt_ctx* ctx, int which, const void* dev_src_all) {
    if (!ctx || !dev_src_all) return PT_ERR_INVALID;
    return unpack_display_enqueue(ctx, which, dev_src_all);
}
extern "C" int pt_display_sync(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->ev_displayed) return PT_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipEventSynchronize(ctx->ev_displayed));
    return PT_OK;
}
extern "C" void* pt_display_buffer(pt_ctx* ctx, int which) {
    if (!ctx || which < 0 || which > PT_BUF_DENOISED) return nullptr;
    return ctx->display[which];
}
extern "C" int pt_download_display(pt_ctx* ctx, int which, void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID;
    const size_t elem = buffer_elem(which);
    if (!elem || !ctx->display[which]) return fail(ctx, PT_ERR_INVALID, "pt_download_display: nothing has been handed over into this buffer");
    if (bytes != elem * (size_t)ctx->width * ctx->height) return fail(ctx, PT_ERR_INVALID, "pt_download_display: byte count does not match the frame size");
    int rc = pt_display_sync(ctx);
    if (rc != PT_OK) return rc;
    CK(hipMemcpy(host, ctx->display[which], bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

// the kernels alone, on the context's stream (pt_multi_gather chains pack -> exchange -> unpack without host waits)
static int pack_launch(pt_ctx* ctx, int which, void* dev_dst) {
    size_t elem;
    void* p = buffer_ptr(ctx, which, &elem);
    if (!p) return fail(ctx, PT_ERR_INVALID, "pt_pack: unknown buffer or not resized");
    CK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->owned;
    if (n) {
        if (elem == 16)
            hipLaunchKernelGGL((k_pack<float4>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)p, ctx->d_pixels, n, ctx->width, (float4*)dev_dst);
        else
            hipLaunchKernelGGL((k_pack<uint32_t>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t*)p, ctx->d_pixels, n, ctx->width, (uint32_t*)dev_dst);
    }
    return PT_OK;
}

static int unpack_launch(pt_ctx* ctx, int which, const void* dev_src_all) {
    size_t elem;
    void* p = buffer_ptr(ctx, which, &elem);
    if (!p) return fail(ctx, PT_ERR_INVALID, "pt_unpack: unknown buffer or not resized");
    CK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->padded * (uint32_t)ctx->world;
    if (n) {
        if (elem == 16)
            hipLaunchKernelGGL((k_unpack<float4>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (float4*)p, ctx->d_all_pixels, n, ctx->width, (const float4*)dev_src_all);
        else
            hipLaunchKernelGGL((k_unpack<uint32_t>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t*)p, ctx->d_all_pixels, n, ctx->width, (const uint32_t*)dev_src_all);
    }
    return PT_OK;
}

extern "C" int pt_get_stats(const pt_ctx* ctx, pt_stats* out) {
    if (!ctx || !out) return PT_ERR_INVALID;
    {
        int rc = drain(const_cast<pt_ctx*>(ctx)); // the statistics of the frame still in flight are wanted
        if (rc != PT_OK) return rc;
    }
    *out = ctx->stats;
    out->frames = ctx->cum_frames;
    out->total_radiance_rays = ctx->cum_radiance;
    out->total_shadow_rays = ctx->cum_shadow;
    out->bvh_nodes = ctx->bvh.num_nodes8;
    out->bvh_bytes = (uint64_t)ctx->bvh.num_nodes8 * sizeof(Node8) + (uint64_t)ctx->bvh.num_tris8 * sizeof(LeafTri);
    out->bvh_build_ms = ctx->bvh_build_ms;
    out->bvh_levels = (uint32_t)ctx->bvh.levels8;
    out->bvh_builder = (uint32_t)ctx->bvh.builder;
    out->path_state_allocs = ctx->path_state_allocs;
    out->bvh_challengers_skipped = (uint32_t)ctx->bvh.challengers_skipped;
    out->create_ms = ctx->create_ms;
    return PT_OK;
}

extern "C" size_t pt_stats_size(void) { return sizeof(pt_stats); }
extern "C" int pt_get_stats_n(const pt_ctx* ctx, void* out, size_t out_bytes) {
    if (!ctx || !out) return PT_ERR_INVALID;
    pt_stats s;
    int rc = pt_get_stats(ctx, &s);
    if (rc != PT_OK) return rc;
    memcpy(out, &s, out_bytes < sizeof(s) ? out_bytes : sizeof(s)); // fields are only ever appended: a caller built against an older header gets its prefix
    if (out_bytes > sizeof(s)) memset((char*)out + sizeof(s), 0, out_bytes - sizeof(s));
    return PT_OK;
}

extern "C" int pt_trace(pt_ctx* ctx, const float* rays, uint32_t n, int any_hit, float* t_out, int32_t* prim_out, int iters,
                        double* kernel_ms) {
    if (!ctx || !rays || !prim_out || (!any_hit && !t_out)) return PT_ERR_INVALID;
    { int rc_ = drain(ctx); if (rc_ != PT_OK) return rc_; } // frames in flight (pt_options.frames_in_flight) finish first
    if (n == 0) return PT_OK;
    CK(hipSetDevice(ctx->device));
    if (iters < 1) iters = 1;
    // temporary state just for the query
    DevScope tmp;
    float4 *dO = nullptr, *dD = nullptr;
    float2* dHit = nullptr;
    uint32_t *dCount = nullptr, *dWork = nullptr;
    unsigned long long* dDbg = nullptr;
    hipEvent_t e0, e1;
    CK(tmp.alloc(&dO, n)); CK(tmp.alloc(&dD, n)); CK(tmp.alloc(&dHit, n)); CK(tmp.alloc(&dCount, 1));
    CK(tmp.alloc(&dWork, (size_t)iters));
    CK(tmp.event(&e0));
    CK(tmp.event(&e1));
    std::vector<float4> hO(n), hD(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = &rays[8 * (size_t)i];
        hO[i] = make_float4(r[0], r[1], r[2], r[3]);
        hD[i] = make_float4(r[4], r[5], r[6], r[7]);
    }
    CK(hipMemcpy(dO, hO.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dD, hD.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dCount, &n, 4, hipMemcpyHostToDevice));
    PathState st{};
    st.rayO = dO;
    st.rayD = dD;
    st.hit = dHit;
    if (getenv("PT_DEBUG_COUNTS")) {
        CK(tmp.alloc(&dDbg, 64));
        CK(dclear(ctx->stream, dDbg, 512));
    }
    CK(hipMemsetAsync(dWork, 0, sizeof(uint32_t) * iters, ctx->stream));
    CK(hipMemsetAsync(ctx->d_totals + 2, 0, sizeof(unsigned long long), ctx->stream));
    CK(hipEventRecord(e0, ctx->stream));
    for (int it = 0; it < iters; ++it) {
        {
            Trace8Args ta{st, bvh_dev(ctx), QView{nullptr, dCount, 0}, QView{}, dWork + it, ctx->ovf, 0, dDbg, 0, ctx->lds_skip, ctx->ovf_depth, fault_word(ctx), ctx->bvh.num_nodes8};
            if (any_hit) hipLaunchKernelGGL((k_trace8<TR_ANY_QUERY>), dim3(ctx->trace_grid), dim3(64), 0, ctx->stream, ta);
            else hipLaunchKernelGGL((k_trace8<TR_CLOSEST>), dim3(ctx->trace_grid), dim3(64), 0, ctx->stream, ta);
        }
    }
    CK(hipEventRecord(e1, ctx->stream));
    if (!any_hit) { // hit records name leaf triangles; the caller wants optixGetPrimitiveIndex
        hipLaunchKernelGGL(k_hits_to_prims, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dHit, ctx->bvh.tris8, n);
    }
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipGetLastError());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (kernel_ms) *kernel_ms = ms / iters;
    {
        unsigned long long fault = 0;
        CK(hipMemcpy(&fault, ctx->d_totals + 2, sizeof(fault), hipMemcpyDeviceToHost));
        if (fault & 1ull) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_trace: traversal stack overflow: the acceleration structure is deeper than the traversal stack");
    }
    if (dDbg) {
        unsigned long long h[8];
        CK(hipMemcpy(h, dDbg, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pt_trace] rays %u x %d: node steps/ray %.2f, tri tests/ray %.2f, pushes/ray %.2f, max stack %llu, max steps of one ray %llu, "
                "wave loop iterations max %llu mean %.1f (waves %llu)\n", n, iters,
                (double)h[0] / n / iters, (double)h[1] / n / iters, (double)h[3] / n / iters, h[2], h[4], h[5], h[7] ? (double)h[6] / h[7] : 0.0, h[7] / iters);
    }
    if (any_hit) {
        std::vector<float2> hh(n);
        CK(hipMemcpy(hh.data(), dHit, sizeof(float2) * n, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) memcpy(&prim_out[i], &hh[i].y, 4);
    } else {
        std::vector<float2> hh(n);
        CK(hipMemcpy(hh.data(), dHit, sizeof(float2) * n, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) {
            t_out[i] = hh[i].x;
            memcpy(&prim_out[i], &hh[i].y, 4);
        }
    }
    return PT_OK;
}

extern "C" int pt_export_bvh(pt_ctx* ctx, void* nodes, size_t nodes_bytes, void* tris, size_t tris_bytes, uint32_t* num_nodes, uint32_t* num_tris) {
    if (!ctx) return PT_ERR_INVALID;
    // the exported layout is the documented 80-byte node whatever the kernels traverse (PT8_NODE64: converted on the device, box for box)
    static_assert(sizeof(Node80) == 80 && sizeof(LeafTri) == 48, "exported layout");
    if (num_nodes) *num_nodes = ctx->bvh.num_nodes8;
    if (num_tris) *num_tris = ctx->bvh.num_tris8;
    if (!nodes && !tris) return PT_OK;
    if (!nodes || !tris || nodes_bytes != sizeof(Node80) * (size_t)ctx->bvh.num_nodes8 || tris_bytes != sizeof(LeafTri) * (size_t)ctx->bvh.num_tris8)
        return fail(ctx, PT_ERR_INVALID, "pt_export_bvh: buffer sizes must be num_nodes * 80 and num_tris * 48 bytes");
    CK(hipSetDevice(ctx->device));
#if PT8_NODE64
    {
        int rc = drain(ctx);
        if (rc) return rc;
        Node80* tmp80 = nullptr;
        CK(hipMalloc((void**)&tmp80, nodes_bytes));
        hipLaunchKernelGGL(k_nodes_to80, dim3((ctx->bvh.num_nodes8 + 255) / 256), dim3(256), 0, ctx->stream, ctx->bvh.nodes8, ctx->bvh.num_nodes8, ctx->bvh.grid, tmp80);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(nodes, tmp80, nodes_bytes, hipMemcpyDeviceToHost);
        hipFree(tmp80);
        CK(e);
    }
#else
    CK(hipMemcpy(nodes, ctx->bvh.nodes8, nodes_bytes, hipMemcpyDeviceToHost));
#endif
    CK(hipMemcpy(tris, ctx->bvh.tris8, tris_bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

extern "C" int pt_eval_table(pt_ctx* ctx, int which, const pt_material* material, int bsdf_mode, const float* in, uint32_t n, float* out) {
    if (!ctx || !in || !out) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    static const int in_w[9] = {11, 9, 1, 3, 3, 3, 2, 2, 3}, out_w[9] = {4, 6, 9, 6, 1, 1, 8, 4, 1};
    if (which < 0 || which > 8) return fail(ctx, PT_ERR_INVALID, "pt_eval_table: unknown table");
    if (which <= 1 && !material) return fail(ctx, PT_ERR_INVALID, "pt_eval_table: material required");
    if ((which == 2 || which == 3 || which == 8) && !ctx->probe.data) return fail(ctx, PT_ERR_INVALID, "pt_eval_table: no probe set");
    if (which == 7 && !ctx->tex0.pixel) return fail(ctx, PT_ERR_INVALID, "pt_eval_table: the scene has no texture");
    CK(hipSetDevice(ctx->device));
    DevScope tmp;
    float *dIn = nullptr, *dOut = nullptr;
    CK(tmp.alloc(&dIn, (size_t)n * in_w[which]));
    CK(tmp.alloc(&dOut, (size_t)n * out_w[which]));
    CK(hipMemcpy(dIn, in, sizeof(float) * (size_t)n * in_w[which], hipMemcpyHostToDevice));
    const dim3 g((n + 127) / 128), b(128);
    pt_material mat{};
    if (material) mat = *material;
    switch (which) {
        case 0:
            if (bsdf_mode == PT_BSDF_LAMBERT) hipLaunchKernelGGL((k_table_bsdf<PT_BSDF_LAMBERT>), g, b, 0, ctx->stream, mat, dIn, n, dOut);
            else hipLaunchKernelGGL((k_table_bsdf<PT_BSDF_DISNEY>), g, b, 0, ctx->stream, mat, dIn, n, dOut);
            break;
        case 1:
            if (bsdf_mode == PT_BSDF_LAMBERT) hipLaunchKernelGGL((k_table_sample<PT_BSDF_LAMBERT>), g, b, 0, ctx->stream, mat, dIn, n, dOut);
            else hipLaunchKernelGGL((k_table_sample<PT_BSDF_DISNEY>), g, b, 0, ctx->stream, mat, dIn, n, dOut);
            break;
        case 2: hipLaunchKernelGGL(k_table_probe_sample, g, b, 0, ctx->stream, ctx->probe, dIn, n, dOut); break;
        case 3: hipLaunchKernelGGL(k_table_probe_eval, g, b, 0, ctx->stream, ctx->probe, dIn, n, dOut); break;
        case 4: hipLaunchKernelGGL(k_table_color, g, b, 0, ctx->stream, dIn, n, dOut); break;
        case 5: hipLaunchKernelGGL(k_table_math, g, b, 0, ctx->stream, dIn, n, dOut); break;
        case 6: hipLaunchKernelGGL(k_table_rng, g, b, 0, ctx->stream, dIn, n, dOut); break;
        case 7: hipLaunchKernelGGL(k_table_tex, g, b, 0, ctx->stream, ctx->tex0, dIn, n, dOut); break;
        case 8: hipLaunchKernelGGL(k_table_probe_pdf, g, b, 0, ctx->stream, ctx->probe, dIn, n, dOut); break;
    }
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipGetLastError());
    CK(hipMemcpy(out, dOut, sizeof(float) * (size_t)n * out_w[which], hipMemcpyDeviceToHost));
    return PT_OK;
}

// ===================================================================================================================
// pt_multi: N contexts in one process (include/pt_amd.h).  No reference counterpart: the reference is single-GPU.
#include <dlfcn.h>
// declarations only: the pointer types below are RCCL's own, the library itself is opened at run time
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else // a box without RCCL's headers: the six prototypes as rccl.h (2.x) declares them, so that libptamd still builds (single-GPU path, peer copies)
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace {
// the six RCCL entry points the exchange needs, resolved at run time: libptamd must load (and the single-GPU path must
// work) on a box without librccl, and a process that already holds torch's bundled librccl must not get a second copy.
// The pointer types are decltype of rccl.h's declarations, so a signature change breaks the build instead of the call.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && GetErrorString; }
};
const ncclDataType_t kNcclUint8 = ncclUint8;

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        }
    }
    return r;
}
thread_local std::string g_multi_error;
} // namespace

namespace {
// One host thread per rank (pt_multi): a frame is ~60 launches and events per device, so enqueueing the devices one after the other
// from one thread costs ndev x (60 x 5-8 us) — at a 1/8 share of a 1080p frame (1.3-1.8 ms of GPU work) more than the frame itself.
// Every rank's thread enqueues its own device; the calling thread posts a job to each and waits for all.
struct RankWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = PT_OK;
    double ms = 0; // host time of the last job
};
void rank_worker_main(RankWorker* w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = job();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lk.lock();
        w->rc = rc;
        w->ms = ms;
        w->done = true;
        w->cv.notify_all();
    }
}
} // namespace

struct pt_multi {
    std::vector<pt_ctx*> ctx;
    std::vector<int> devices;
    std::string err;
    bool distinct = true;           // no device appears twice (RCCL's requirement)
    std::vector<ncclComm_t> comms;  // per rank (empty until the first RCCL gather)
    int exchange_pref = 0;          // PT_MULTI_EXCHANGE: 0 auto, 1 rccl, 2 peer copies
    std::vector<void*> send, recv;  // per rank: padded * 16 bytes, world * padded * 16 bytes (the synchronous pt_multi_gather)
    uint32_t padded = 0;
    double gather_ms = 0;
    int last_exchange = PT_EXCHANGE_NONE;
    std::vector<std::unique_ptr<RankWorker>> workers; // empty: single rank or PT_MULTI_THREADS=0 (everything on the calling thread)
    double enqueue_ms = 0;          // host time of the enqueue phase of the last render: the slowest rank's
    // Overlapped hand-off (frames in flight + a gather mask): frame k's strips are packed behind frame k (pt_pack_async) into the
    // buffers of slot k & 1 and exchanged + scattered into the ranks' DISPLAY buffers by the NEXT render call, while frame k+1 renders.
    struct Strips { std::vector<void*> send, recv; };
    Strips hand[2][6];              // [slot][pt_buffer], allocated on first use
    std::vector<hipEvent_t> xfer_done[2]; // peer-copy exchange: rank r's copies of slot s are complete
    int pending_slot = -1;          // slot whose strips are packed (or being packed) but not yet exchanged
    uint32_t pending_mask = 0;
    uint64_t handed = 0;            // frames handed over through the overlapped path
    uint64_t packs = 0;             // pack rounds enqueued (slot = packs & 1)
};

static int mfail(pt_multi* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_multi_error = msg;
    return code;
}
// forward a per-context failure
static int mctx(pt_multi* m, int r, int rc, const char* what) {
    if (rc == PT_OK) return PT_OK;
    return mfail(m, rc, std::string(what) + " (rank " + std::to_string(r) + ", device " + std::to_string(m->devices[r]) + "): " + m->ctx[r]->err);
}
#define MCK(m, call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return mfail(m, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* pt_multi_last_error(const pt_multi* m) { return m ? m->err.c_str() : g_multi_error.c_str(); }
extern "C" int pt_multi_size(const pt_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" pt_ctx* pt_multi_ctx(pt_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[rank] : nullptr; }

static int multi_present_pending(pt_multi* m, bool wait);
static void multi_free_exchange(pt_multi* m) {
    for (size_t r = 0; r < m->ctx.size(); ++r) { // pending packs / exchanges run on the contexts' streams
        hipSetDevice(m->devices[r]);
        if (m->ctx[r] && m->ctx[r]->stream) hipStreamSynchronize(m->ctx[r]->stream);
    }
    for (size_t r = 0; r < m->send.size(); ++r) {
        hipSetDevice(m->devices[r]);
        if (m->send[r]) hipFree(m->send[r]);
        if (m->recv[r]) hipFree(m->recv[r]);
    }
    m->send.clear();
    m->recv.clear();
    for (int sl = 0; sl < 2; ++sl) {
        for (auto& st : m->hand[sl]) {
            for (size_t r = 0; r < st.send.size(); ++r) {
                hipSetDevice(m->devices[r]);
                if (st.send[r]) hipFree(st.send[r]);
                if (st.recv[r]) hipFree(st.recv[r]);
            }
            st.send.clear();
            st.recv.clear();
        }
        for (hipEvent_t e : m->xfer_done[sl])
            if (e) hipEventDestroy(e);
        m->xfer_done[sl].clear();
    }
    m->pending_slot = -1;
    m->pending_mask = 0;
    m->padded = 0;
}

// fn(rank) for every rank, concurrently on the ranks' threads (or one after the other on the calling thread); returns the first
// failing rank's status, forwarded with its message.  max_ms: host time of the slowest rank.
static int multi_run(pt_multi* m, const std::function<int(int)>& fn, const char* what, double* max_ms = nullptr) {
    const int world = (int)m->ctx.size();
    std::vector<int> rcs(world, PT_OK);
    double slow = 0;
    if (m->workers.empty()) {
        for (int r = 0; r < world; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            rcs[r] = fn(r);
            slow += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); // serial: the times add up
        }
    } else {
        for (int r = 0; r < world; ++r) {
            RankWorker& w = *m->workers[r];
            std::lock_guard<std::mutex> lk(w.mu);
            w.job = [&fn, r] { return fn(r); };
            w.has_job = true;
            w.done = false;
            w.cv.notify_all();
        }
        for (int r = 0; r < world; ++r) {
            RankWorker& w = *m->workers[r];
            std::unique_lock<std::mutex> lk(w.mu);
            w.cv.wait(lk, [&] { return w.done; });
            rcs[r] = w.rc;
            slow = std::max(slow, w.ms);
        }
    }
    if (max_ms) *max_ms = slow;
    for (int r = 0; r < world; ++r)
        if (rcs[r] != PT_OK) return mctx(m, r, rcs[r], what);
    return PT_OK;
}

extern "C" int pt_multi_destroy(pt_multi* m) {
    if (!m) return PT_OK;
    for (auto& w : m->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
    }
    m->workers.clear();
    multi_free_exchange(m);
    if (!m->comms.empty() && rccl().ok())
        for (ncclComm_t c : m->comms)
            if (c) rccl().CommDestroy(c);
    for (pt_ctx* c : m->ctx) pt_destroy(c);
    delete m;
    return PT_OK;
}

extern "C" int pt_create_multi(const pt_scene_desc* scene, const int* devices, int ndev, pt_multi** out) {
    if (!out || !devices || ndev < 1 || ndev > 64) return mfail(nullptr, PT_ERR_INVALID, "pt_create_multi: need 1..64 devices and an output pointer");
    FlatScene fs; // ONE host-side flatten + validation, shared by all ranks
    int rc = flatten_scene(scene, fs);
    if (rc != PT_OK) return mfail(nullptr, rc, g_create_error);
    pt_multi* m = new pt_multi();
    m->devices.assign(devices, devices + ndev);
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) m->distinct = false;
    if (const char* e = getenv("PT_MULTI_EXCHANGE")) m->exchange_pref = !strcmp(e, "rccl") ? 1 : (!strcmp(e, "peer") ? 2 : 0);
    for (int r = 0; r < ndev; ++r) {
        pt_ctx* c = nullptr;
        rc = create_from_flat(fs, devices[r], &c);
        if (rc != PT_OK) {
            const std::string msg = "pt_create_multi: rank " + std::to_string(r) + ": " + g_create_error;
            pt_multi_destroy(m);
            return mfail(nullptr, rc, msg);
        }
        m->ctx.push_back(c);
    }
    // direct device-to-device copies between distinct devices need peer access (xGMI); failure is not fatal — hipMemcpyPeerAsync
    // then stages through the host
    if (m->distinct && ndev > 1)
        for (int a = 0; a < ndev; ++a) {
            hipSetDevice(devices[a]);
            for (int b = 0; b < ndev; ++b) {
                int can = 0;
                if (a != b && hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        }
    const char* te = getenv("PT_MULTI_THREADS");
    if (ndev > 1 && !(te && atoi(te) == 0))
        for (int r = 0; r < ndev; ++r) {
            m->workers.emplace_back(new RankWorker());
            m->workers.back()->th = std::thread(rank_worker_main, m->workers.back().get(), devices[r]);
        }
    *out = m;
    return PT_OK;
}

extern "C" int pt_multi_set_options(pt_multi* m, const pt_options* opt) {
    if (!m || !opt) return PT_ERR_INVALID;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        int rc = mctx(m, (int)r, pt_set_options(m->ctx[r], opt), "pt_multi_set_options");
        if (rc) return rc;
    }
    return PT_OK;
}
extern "C" int pt_multi_set_probe(pt_multi* m, const float* data, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY, int w, int h) {
    if (!m) return PT_ERR_INVALID;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        int rc = mctx(m, (int)r, pt_set_probe(m->ctx[r], data, pdfX, cdfX, pdfY, cdfY, w, h), "pt_multi_set_probe");
        if (rc) return rc;
    }
    return PT_OK;
}
extern "C" int pt_multi_set_probe_image(pt_multi* m, const float* data, int w, int h) {
    if (!m) return PT_ERR_INVALID;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        int rc = mctx(m, (int)r, pt_set_probe_image(m->ctx[r], data, w, h), "pt_multi_set_probe_image");
        if (rc) return rc;
    }
    return PT_OK;
}
extern "C" int pt_multi_set_camera(pt_multi* m, const float eye[3], const float U[3], const float V[3], const float W[3]) {
    if (!m) return PT_ERR_INVALID;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        int rc = mctx(m, (int)r, pt_set_camera(m->ctx[r], eye, U, V, W), "pt_multi_set_camera");
        if (rc) return rc;
    }
    return PT_OK;
}

extern "C" int pt_multi_resize(pt_multi* m, int width, int height, int tile_w, int tile_h) {
    if (!m) return PT_ERR_INVALID;
    if (width == 0 || height == 0) return PT_OK;
    if (tile_w == 0) tile_w = 64;
    if (tile_h == 0) tile_h = 16;
    const int world = (int)m->ctx.size();
    multi_free_exchange(m);
    for (int r = 0; r < world; ++r) {
        pt_ctx* c = m->ctx[r];
        c->width = c->height = 0; // pt_set_partition then only records the partition; pt_resize applies it
        int rc = mctx(m, r, pt_set_partition(c, r, world, tile_w, tile_h), "pt_multi_resize");
        if (rc == PT_OK) rc = mctx(m, r, pt_resize(c, width, height), "pt_multi_resize");
        if (rc) return rc;
    }
    m->padded = m->ctx[0]->padded;
    m->send.assign(world, nullptr);
    m->recv.assign(world, nullptr);
    for (int r = 0; r < world; ++r) {
        MCK(m, hipSetDevice(m->devices[r]));
        MCK(m, hipMalloc(&m->send[r], std::max<size_t>(16, (size_t)m->padded * 16)));
        MCK(m, hipMalloc(&m->recv[r], std::max<size_t>(16, (size_t)world * m->padded * 16)));
    }
    return PT_OK;
}

extern "C" int pt_multi_gather(pt_multi* m, int which) {
    if (!m) return PT_ERR_INVALID;
    const int world = (int)m->ctx.size();
    if (m->send.empty()) return mfail(m, PT_ERR_INVALID, "pt_multi_gather: not resized");
    size_t elem = 0;
    if (!buffer_ptr(m->ctx[0], which, &elem)) return mfail(m, PT_ERR_INVALID, "pt_multi_gather: unknown buffer");
    const size_t bytes = (size_t)m->padded * elem; // one rank's packed strip (padded to the largest share: same on every rank)
    {
        int rc = multi_present_pending(m, true); // a frame of the overlapped path that was still waiting for its hand-over
        if (rc) return rc;
    }
    for (int r = 0; r < world; ++r) { // frames in flight (pt_options.frames_in_flight) are not ordered before the contexts' own streams
        int rc = mctx(m, r, drain(m->ctx[r]), "pt_multi_gather");
        if (rc) return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < world; ++r) {
        int rc = mctx(m, r, pack_launch(m->ctx[r], which, m->send[r]), "pt_multi_gather(pack)");
        if (rc) return rc;
    }
    bool use_rccl = m->exchange_pref == 1 || (m->exchange_pref == 0 && m->distinct && world > 1);
    if (use_rccl && (!rccl().ok() || !m->distinct)) {
        if (m->exchange_pref == 1) return mfail(m, PT_ERR_UNSUPPORTED, !m->distinct ? "pt_multi_gather: RCCL needs distinct devices" : "pt_multi_gather: librccl not found");
        use_rccl = false;
    }
    if (use_rccl) {
        Rccl& R = rccl();
        if (m->comms.empty()) {
            m->comms.assign(world, nullptr);
            const ncclResult_t e = R.CommInitAll(m->comms.data(), world, m->devices.data());
            if (e != ncclSuccess) {
                m->comms.clear();
                return mfail(m, PT_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(e));
            }
        }
        // one all-gather of `bytes` per rank: with 8 GPUs each of the 7 xGMI links of a GPU carries one strip
        ncclResult_t e = R.GroupStart();
        for (int r = 0; r < world && e == ncclSuccess; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            e = R.AllGather(m->send[r], m->recv[r], bytes, kNcclUint8, m->comms[r], m->ctx[r]->stream);
        }
        const ncclResult_t e2 = R.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) return mfail(m, PT_ERR_HIP, std::string("ncclAllGather: ") + R.GetErrorString(e != ncclSuccess ? e : e2));
        m->last_exchange = PT_EXCHANGE_RCCL;
    } else {
        // every rank writes its strip into slot r of every rank's receive buffer (direct peer writes over xGMI; a plain
        // device-to-device copy when both ranks live on one device), then every rank waits for all writers
        std::vector<hipEvent_t> done(world);
        for (int r = 0; r < world; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            for (int p = 0; p < world; ++p) {
                char* dst = (char*)m->recv[p] + (size_t)r * bytes;
                if (m->devices[p] == m->devices[r]) MCK(m, hipMemcpyAsync(dst, m->send[r], bytes, hipMemcpyDeviceToDevice, m->ctx[r]->stream));
                else MCK(m, hipMemcpyPeerAsync(dst, m->devices[p], m->send[r], m->devices[r], bytes, m->ctx[r]->stream));
            }
            MCK(m, hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
            MCK(m, hipEventRecord(done[r], m->ctx[r]->stream));
        }
        for (int p = 0; p < world; ++p) {
            MCK(m, hipSetDevice(m->devices[p]));
            for (int r = 0; r < world; ++r)
                if (r != p) MCK(m, hipStreamWaitEvent(m->ctx[p]->stream, done[r], 0));
        }
        // the events must outlive the waits: destroyed after the streams are synchronised below
        for (int r = 0; r < world; ++r) {
            int rc = mctx(m, r, unpack_launch(m->ctx[r], which, m->recv[r]), "pt_multi_gather(unpack)");
            if (rc) return rc;
        }
        for (int r = 0; r < world; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            MCK(m, hipStreamSynchronize(m->ctx[r]->stream));
        }
        for (int r = 0; r < world; ++r) hipEventDestroy(done[r]);
        m->last_exchange = PT_EXCHANGE_PEER_COPY;
        m->gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return PT_OK;
    }
    for (int r = 0; r < world; ++r) {
        int rc = mctx(m, r, unpack_launch(m->ctx[r], which, m->recv[r]), "pt_multi_gather(unpack)");
        if (rc) return rc;
    }
    for (int r = 0; r < world; ++r) {
        MCK(m, hipSetDevice(m->devices[r]));
        MCK(m, hipStreamSynchronize(m->ctx[r]->stream));
    }
    m->gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

static int multi_after_render(pt_multi* m, uint32_t gather_mask, uint32_t* host_rgba8) {
    if (host_rgba8) gather_mask |= 1u << PT_BUF_FRAME;
    if (m->ctx[0]->width == 0) return PT_OK;
    for (int which = 0; which <= PT_BUF_DENOISED; ++which)
        if (gather_mask & (1u << which)) {
            int rc = pt_multi_gather(m, which);
            if (rc) return rc;
        }
    if (host_rgba8) return mctx(m, 0, pt_download(m->ctx[0], PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)m->ctx[0]->width * m->ctx[0]->height), "pt_multi_render(download)");
    return PT_OK;
}

// ---- overlapped hand-off (frames in flight + something to hand over): everything below only ENQUEUES on the ranks' own streams
static int multi_ensure_strips(pt_multi* m, int slot, int which) {
    pt_multi::Strips& st = m->hand[slot][which];
    const int world = (int)m->ctx.size();
    if ((int)st.send.size() == world) return PT_OK;
    const size_t elem = buffer_elem(which);
    st.send.assign(world, nullptr);
    st.recv.assign(world, nullptr);
    for (int r = 0; r < world; ++r) {
        MCK(m, hipSetDevice(m->devices[r]));
        MCK(m, hipMalloc(&st.send[r], std::max<size_t>(16, (size_t)m->padded * elem)));
        MCK(m, hipMalloc(&st.recv[r], std::max<size_t>(16, (size_t)world * m->padded * elem)));
    }
    if (m->xfer_done[slot].empty()) {
        m->xfer_done[slot].assign(world, nullptr);
        for (int r = 0; r < world; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            MCK(m, hipEventCreateWithFlags(&m->xfer_done[slot][r], hipEventDisableTiming));
        }
    }
    return PT_OK;
}
static bool multi_use_rccl(pt_multi* m, int* err) {
    const int world = (int)m->ctx.size();
    bool use_rccl = m->exchange_pref == 1 || (m->exchange_pref == 0 && m->distinct && world > 1);
    *err = PT_OK;
    if (use_rccl && (!rccl().ok() || !m->distinct)) {
        if (m->exchange_pref == 1) *err = mfail(m, PT_ERR_UNSUPPORTED, !m->distinct ? "pt_multi: RCCL needs distinct devices" : "pt_multi: librccl not found");
        use_rccl = false;
    }
    return use_rccl;
}
// the strips of `slot` (packed on every rank's stream) -> all ranks' receive buffers -> the ranks' display buffers
static int multi_exchange_display(pt_multi* m, int slot, int which) {
    const int world = (int)m->ctx.size();
    pt_multi::Strips& st = m->hand[slot][which];
    const size_t bytes = (size_t)m->padded * buffer_elem(which);
    int err;
    const bool use_rccl = multi_use_rccl(m, &err);
    if (err) return err;
    if (use_rccl) {
        Rccl& R = rccl();
        if (m->comms.empty()) {
            m->comms.assign(world, nullptr);
            const ncclResult_t e = R.CommInitAll(m->comms.data(), world, m->devices.data());
            if (e != ncclSuccess) {
                m->comms.clear();
                return mfail(m, PT_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(e));
            }
        }
        ncclResult_t e = R.GroupStart();
        for (int r = 0; r < world && e == ncclSuccess; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            e = R.AllGather(st.send[r], st.recv[r], bytes, kNcclUint8, m->comms[r], m->ctx[r]->stream);
        }
        const ncclResult_t e2 = R.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) return mfail(m, PT_ERR_HIP, std::string("ncclAllGather: ") + R.GetErrorString(e != ncclSuccess ? e : e2));
        m->last_exchange = PT_EXCHANGE_RCCL;
    } else {
        for (int r = 0; r < world; ++r) {
            MCK(m, hipSetDevice(m->devices[r]));
            for (int p = 0; p < world; ++p) {
                char* dst = (char*)st.recv[p] + (size_t)r * bytes;
                if (m->devices[p] == m->devices[r]) MCK(m, hipMemcpyAsync(dst, st.send[r], bytes, hipMemcpyDeviceToDevice, m->ctx[r]->stream));
                else MCK(m, hipMemcpyPeerAsync(dst, m->devices[p], st.send[r], m->devices[r], bytes, m->ctx[r]->stream));
            }
            MCK(m, hipEventRecord(m->xfer_done[slot][r], m->ctx[r]->stream));
        }
        for (int p = 0; p < world; ++p) {
            MCK(m, hipSetDevice(m->devices[p]));
            for (int r = 0; r < world; ++r)
                if (r != p) MCK(m, hipStreamWaitEvent(m->ctx[p]->stream, m->xfer_done[slot][r], 0));
        }
        m->last_exchange = PT_EXCHANGE_PEER_COPY;
    }
    for (int r = 0; r < world; ++r) {
        int rc = mctx(m, r, unpack_display_enqueue(m->ctx[r], which, st.recv[r]), "pt_multi(unpack_display)");
        if (rc) return rc;
    }
    return PT_OK;
}
// exchange what the previous call packed (if anything) and wait until it is on display
static int multi_present_pending(pt_multi* m, bool wait) {
    if (m->pending_slot < 0) return PT_OK;
    const int slot = m->pending_slot;
    const uint32_t mask = m->pending_mask;
    m->pending_slot = -1;
    m->pending_mask = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int which = 0; which <= PT_BUF_DENOISED; ++which)
        if (mask & (1u << which)) {
            int rc = multi_exchange_display(m, slot, which);
            if (rc) return rc;
        }
    ++m->handed;
    if (wait)
        for (size_t r = 0; r < m->ctx.size(); ++r) {
            int rc = mctx(m, (int)r, pt_display_sync(m->ctx[r]), "pt_multi(display)");
            if (rc) return rc;
        }
    m->gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}
static int multi_pack_newest(pt_multi* m, uint32_t mask) {
    // the slots alternate: the other slot's strips were exchanged by this call a moment ago (they may still be in flight), this slot's
    // were put on display — and waited for — one call earlier
    const int use = (int)(m->packs++ & 1);
    for (int which = 0; which <= PT_BUF_DENOISED; ++which)
        if (mask & (1u << which)) {
            int rc = multi_ensure_strips(m, use, which);
            if (rc) return rc;
            for (size_t r = 0; r < m->ctx.size(); ++r) {
                rc = mctx(m, (int)r, pack_async_enqueue(m->ctx[r], which, m->hand[use][which].send[r], use), "pt_multi(pack)");
                if (rc) return rc;
            }
        }
    m->pending_slot = use;
    m->pending_mask = mask;
    return PT_OK;
}

// One frame (or launch chain of `count` subframes, or — regions != null — one foveated frame) on every rank.
static int multi_render_common(pt_multi* m, uint32_t spp, uint32_t subframe_index, uint32_t count, const pt_region* regions, uint32_t nreg,
                               const pt_variant* variant, uint32_t gather_mask, uint32_t* host_rgba8, const char* what) {
    const int world = (int)m->ctx.size();
    if (world == 0) return PT_OK;
    if (host_rgba8) gather_mask |= 1u << PT_BUF_FRAME;
    if (gather_mask >> (PT_BUF_DENOISED + 1)) return mfail(m, PT_ERR_INVALID, std::string(what) + ": unknown buffer in gather_mask");
    const int F = frames_mode(m->ctx[0]);
    auto enqueue = [&](int r, int slot, int mode) {
        pt_ctx* c = m->ctx[r];
        return regions ? regions_enqueue(c, regions, nreg, variant, slot, mode != 0) : render_enqueue(c, spp, subframe_index, slot, mode, count);
    };
    auto enqueue_pipelined = [&](int r) {
        pt_ctx* c = m->ctx[r];
        const int slot = (c->last_slot + 1) % F;
        int rc = render_finish(c, slot);
        if (rc == PT_OK) rc = enqueue(r, slot, F);
        if (rc == PT_OK) c->last_slot = slot;
        return rc;
    };
    if (F > 1 && gather_mask == 0) {
        // pure throughput (nothing is handed over after this frame): pt_options.frames_in_flight applies on every device
        int rc = multi_present_pending(m, true); // a frame of an earlier call that was waiting for its hand-over
        int rc1 = multi_run(m, enqueue_pipelined, what, &m->enqueue_ms);
        int rc2 = multi_run(m, [&](int r) { return pipelined_wait(m->ctx[r], F); }, what);
        return rc ? rc : (rc1 ? rc1 : rc2);
    }
    if (F > 1) {
        // Frames in flight AND a hand-over: frame k is enqueued, then the strips frame k-1 packed behind itself are exchanged and
        // scattered into the display buffers while frame k renders, then frame k's pack is enqueued behind frame k.  The call
        // returns when frame k-1 is on display (host_rgba8 receives it) and at most F-1 frames are still running.
        int rc = multi_run(m, enqueue_pipelined, what, &m->enqueue_ms);
        if (rc) return rc;
        const bool had = m->pending_slot >= 0;
        rc = multi_present_pending(m, true);
        if (rc) return rc;
        if (m->ctx[0]->width) {
            rc = multi_pack_newest(m, gather_mask);
            if (rc) return rc;
        }
        rc = multi_run(m, [&](int r) { return pipelined_wait(m->ctx[r], F); }, what);
        if (rc) return rc;
        if (host_rgba8 && had && m->ctx[0]->width)
            return mctx(m, 0, pt_download_display(m->ctx[0], PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)m->ctx[0]->width * m->ctx[0]->height), what);
        return PT_OK;
    }
    // synchronous frames: every rank enqueues and finishes its own frame on its own thread, then the hand-over
    int rc = multi_present_pending(m, true);
    if (rc) return rc;
    rc = multi_run(m, [&](int r) { return enqueue(r, 0, 0); }, what, &m->enqueue_ms);
    int rc2 = multi_run(m, [&](int r) { return render_finish(m->ctx[r]); }, what); // every rank's frame is waited for, also after a failure
    if (rc || rc2) return rc ? rc : rc2;
    return multi_after_render(m, gather_mask, host_rgba8);
}

extern "C" int pt_multi_render(pt_multi* m, uint32_t spp, uint32_t subframe_index, uint32_t gather_mask, uint32_t* host_rgba8) {
    return pt_multi_render_batch(m, spp, subframe_index, 1, gather_mask, host_rgba8);
}

extern "C" int pt_multi_render_batch(pt_multi* m, uint32_t spp, uint32_t subframe_index, uint32_t count, uint32_t gather_mask, uint32_t* host_rgba8) {
    if (!m) return PT_ERR_INVALID;
    return multi_render_common(m, spp, subframe_index, count, nullptr, 0, nullptr, gather_mask, host_rgba8, "pt_multi_render");
}

extern "C" int pt_multi_render_regions(pt_multi* m, const pt_region* regions, uint32_t n, const pt_variant* variant, uint32_t gather_mask, uint32_t* host_rgba8) {
    if (!m || (!regions && n)) return PT_ERR_INVALID;
    return multi_render_common(m, 0, 0, 1, regions, n, variant, gather_mask, host_rgba8, "pt_multi_render_regions");
}

// the frame that is still waiting for its hand-over (overlapped path) is exchanged and put on display
extern "C" int pt_multi_flush(pt_multi* m, uint32_t* host_rgba8) {
    if (!m) return PT_ERR_INVALID;
    const bool had = m->pending_slot >= 0;
    const bool frame = had && (m->pending_mask & (1u << PT_BUF_FRAME));
    int rc = multi_present_pending(m, true);
    if (rc) return rc;
    if (host_rgba8 && frame && !m->ctx.empty() && m->ctx[0]->width)
        return mctx(m, 0, pt_download_display(m->ctx[0], PT_BUF_FRAME, host_rgba8, sizeof(uint32_t) * (size_t)m->ctx[0]->width * m->ctx[0]->height), "pt_multi_flush");
    return PT_OK;
}

extern "C" int pt_multi_get_stats(const pt_multi* m, pt_multi_stats* out) {
    if (!m || !out) return PT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        pt_stats s;
        pt_get_stats(m->ctx[r], &s);
        out->sum.radiance_rays += s.radiance_rays;
        out->sum.shadow_rays += s.shadow_rays;
        out->sum.shaded_hits += s.shaded_hits;
        out->sum.paths += s.paths;
        out->sum.frames += s.frames; // frames x ranks
        out->sum.total_radiance_rays += s.total_radiance_rays;
        out->sum.total_shadow_rays += s.total_shadow_rays;
        out->sum.render_ms = std::max(out->sum.render_ms, s.render_ms);
        out->sum.trace_ms = std::max(out->sum.trace_ms, s.trace_ms);
        out->sum.shadow_ms = std::max(out->sum.shadow_ms, s.shadow_ms);
        out->sum.shade_ms = std::max(out->sum.shade_ms, s.shade_ms);
        out->sum.other_ms = std::max(out->sum.other_ms, s.other_ms);
        out->sum.trace_launches += s.trace_launches;
        out->sum.shadow_launches += s.shadow_launches;
        out->sum.shade_launches += s.shade_launches;
        out->sum.fused_passes += s.fused_passes;
        out->sum.bvh_nodes = s.bvh_nodes;
        out->sum.bvh_bytes = s.bvh_bytes;
        out->sum.bvh_levels = s.bvh_levels;
        out->sum.bvh_builder = s.bvh_builder;
        out->sum.bvh_build_ms = std::max(out->sum.bvh_build_ms, s.bvh_build_ms);
    }
    out->gather_ms = m->gather_ms;
    out->exchange = m->last_exchange;
    out->ndev = (int)m->ctx.size();
    out->enqueue_ms = m->enqueue_ms;
    out->threads = (int32_t)m->workers.size();
    out->frames_handed_over = m->handed;
    return PT_OK;
}
