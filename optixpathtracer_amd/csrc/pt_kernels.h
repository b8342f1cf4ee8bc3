// pt_kernels.h — the wavefront kernels that replace the optixLaunch megakernel
// (__raygen__renderFrame / __closesthit__radiance / __miss__radiance / occlusion programs,
// deviceProgram.cu:206-594).  One path = one (pixel, sample) of the reference's raygen loops.
//
//   k_generate   raygen prologue per sample (deviceProgram.cu:357,370-410): seed, jitter, camera ray
//   k_trace8<0>  (pt_bvh8.h) traceRadiance (:152-178, 416-422): closest hit for every queued path
//   k_shade      __closesthit__radiance + __miss__radiance + the raygen loop body (:424-442)
//   k_trace8<1>  traceOcclusion + __anyhit__/__miss__occlusion (:181-204,237-250) and the deferred
//                "if (!occluded) sum += val" of SampleLights / "if (occluded)" of SampleShadow (:272-289,314-331)
//   k_resolve    raygen epilogue (:445-474): per-pixel sample sums in sample order, backplate,
//                progressive blend, make_color, the five buffer writes
//
// HBM layout: structures of arrays (float4 / float2 / uint4, 16-byte aligned, one coalesced access per array per wave).  Queues are
// arrays of path slots; the state a path carries from bounce to bounce sits in queue order beside its entry, the per-path sums are
// indexed by slot (PathState below).
#pragma once
#include "pt_bvh.h"

enum { FLAG_DONE = 1, FLAG_SECONDARY = 2, FLAG_CULLED = 4 /* outside the foveation annulus */,
       FLAG_HIT0 = 8 /* scenes without shadow catchers: prd.alpha = 1 (deviceProgram.cu:547) as one bit, materialised by the resolve kernels */ }; // RAY_STATE_FLAGS_* deviceProgram.cu:46-48
enum { PEND_DIRECT = 1, PEND_INDIRECT = 2, PEND_ALPHA = 3 };
enum { TR_CLOSEST = 0, TR_SHADOW_APPLY = 1, TR_ANY_QUERY = 2, TR_UNIFIED = 3 /* closest-hit queue + shadow queue in one launch */ }; // modes of the persistent traversal kernels

// What a path carries from bounce to bounce travels WITH ITS QUEUE ENTRY: element `pos` of these arrays belongs to entry `pos` of the queue the
// launch reads (k_trace8: the rays it traces and the hit records it writes; k_shade: its input), and k_shade writes the continuing path's
// state at the position its push into the next queue returned (ShadeParams::o*).  So every launch reads and writes the state densely however
// few of the frame's paths are still alive — indexed by path slot, bounce 3 found 2.4 live paths per 128-byte line and bounce 6 one
// (k_shade cost 0.078 ns per path at bounce 0 and 0.21 at bounce 7).  Two copies (ping-pong); the generate kernels write the first
// in launch order (position = slot).  What is summed per path stays indexed by slot.
struct PathState {
    float4* rayO;  // ray origin xyz, tmin
    float4* rayD;  // ray direction xyz, tmax
    float2* hit;   // t, bits of the leaf triangle's index (pt_bvh.h LeafTri; negative = miss)
    float4* thr;   // pathThroughput xyz, rayEta
    uint4* rf;     // Random seed1, seed2 | depth | flags << 8 | unused
    // the shadow rays of the bounce, in the order of the shadow queue: origin, direction (tmin .01, tmax 1e16), pending NEE contribution xyz | kind
    float4 *shO, *shD, *shPend;
    // ---- by path slot
    uint32_t* pflags; // FLAG_CULLED (generate) | FLAG_HIT0 (first hit): what the resolve kernels need of a path's flags
    float4 *direct, *indirect, *nrm, *alb;
    // Scenes without shadow-catcher materials (prdN == null): prd.alpha is the FLAG_HIT0 bit of fd, and nrm/alb are written once,
    // by the depth-0 closest-hit/miss, never read back by k_shade.  Shadow-catcher scenes keep alpha as a float sum (SampleShadow
    // accumulates into it, a later ordinary hit overwrites it) and nrm/alb as running sums (pass-throughs re-add prd.normal).
    float4* alpha;
    float4 *prdN, *prdA; // prd.normal / prd.albedo, shadow-catcher scenes only (else null)
    // asynchronous shadow rays (pt_options.split_shadow = 2; null otherwise): every bounce b has its own shadow records
    // [b * bstride + slot] — origin, direction, pending contribution — so that the shadow rays of bounce b no longer have to
    // finish before k_shade(b+1) overwrites the bounce's shadow stream, and a visibility bit per bounce instead of the immediate
    // `direct/indirect += contribution`: k_resolve adds the visible contributions in bounce order, which keeps the
    // reference's float sums whatever order the shadow launches finish in.
    float4 *sO, *sD, *pendB;
    uint32_t* vis;
    uint32_t bstride;
};

struct FrameParams { // LaunchParams (LaunchParams.h:51-79) minus the OptiX handle and the dead light
    float4* accum;
    uint32_t* frame;
    float4 *color, *normal, *albedo;
    int width, height;
    uint32_t subframe_index;
    v3 eye, U, V, W;
    uint32_t spp; // samples_per_launch
    DevProbe probe;
};

struct BatchParams {
    const uint32_t* pixels; // x | y << 16 for the pixels of this chunk
    uint32_t npix;          // pixels in the chunk
    // Samples of the pass, counted over the whole batch of subframes (pt_render_batch): "virtual sample" v = j * spp + s is sample s of
    // subframe fp.subframe_index + j (seeds are tea<4>(pixel, subframe), deviceProgram.cu:357, so subframes are independent until
    // they blend).  A single frame is the batch of one subframe: v = s.
    uint32_t s0, S;         // first virtual sample and number of virtual samples in the pass
    int carry;              // nrm/alb carried through pixNormal/pixAlbedo (sequential-sample mode)
    float4 *pixResult, *pixAlpha, *pixNormal, *pixAlbedo; // per pixel of the chunk
};

// ------------------------------------------------------------------ queues
// A queue is PT_NSUB independent sub-queues, each with its own append counter 64 B apart: producers of
// workgroup b append to sub-queue b % PT_NSUB, so same-address atomic traffic (≈10 ns each on MI355X,
// measured: it was 60 % of k_shade's time with a single counter) is spread over 64 addresses.
// Consumers see one index space [0,total) through a 64-entry prefix sum kept in LDS.
// base == nullptr means the identity queue [0, counts[0]).
#define PT_NSUB 64
#define PT_CSTRIDE 16
struct QView {
    uint32_t* base;
    uint32_t* counts;
    uint32_t sub_cap;
};

// all threads of the workgroup call this; s_prefix has PT_NSUB + 1 entries; returns the total
PT_DEV uint32_t qreader_init(const QView& q, uint32_t* s_prefix) {
    if (q.base == nullptr) return q.counts[0];
    if (threadIdx.x < 64) {
        uint32_t v = q.counts[threadIdx.x * PT_CSTRIDE];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off);
            if ((int)threadIdx.x >= off) v += o;
        }
        s_prefix[threadIdx.x + 1] = v;
        if (threadIdx.x == 0) s_prefix[0] = 0;
    }
    __syncthreads();
    return s_prefix[PT_NSUB];
}
// same, trying the sub-queue of the lane's previous lookup first (consecutive work indices mostly share it)
PT_DEV uint32_t qreader_get_hint(const QView& q, const uint32_t* s_prefix, uint32_t i, uint32_t& hint) {
    if (q.base == nullptr) return i;
    uint32_t lo = hint;
    if (!(s_prefix[lo] <= i && i < s_prefix[lo + 1])) {
        lo = 0;
        uint32_t hi = PT_NSUB;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_prefix[mid] <= i) lo = mid; else hi = mid;
        }
        hint = lo;
    }
    return q.base[(size_t)lo * q.sub_cap + (i - s_prefix[lo])];
}
// position of entry i in the queue's arrays (the slot array q.base and the state that travels with it)
PT_DEV uint32_t qreader_pos_hint(const QView& q, const uint32_t* s_prefix, uint32_t i, uint32_t& hint) {
    if (q.base == nullptr) return i;
    uint32_t lo = hint;
    if (!(s_prefix[lo] <= i && i < s_prefix[lo + 1])) {
        lo = 0;
        uint32_t hi = PT_NSUB;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_prefix[mid] <= i) lo = mid; else hi = mid;
        }
        hint = lo;
    }
    return lo * q.sub_cap + (i - s_prefix[lo]);
}
PT_DEV uint32_t qreader_pos(const QView& q, const uint32_t* s_prefix, uint32_t i) {
    if (q.base == nullptr) return i;
    uint32_t lo = 0, hi = PT_NSUB; // find lo with s_prefix[lo] <= i < s_prefix[lo+1]
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_prefix[mid] <= i) lo = mid; else hi = mid;
    }
    return lo * q.sub_cap + (i - s_prefix[lo]);
}
PT_DEV uint32_t qslot(const QView& q, uint32_t pos) { return q.base ? q.base[pos] : pos; } // the path slot of the entry at `pos`
PT_DEV uint32_t qreader_get(const QView& q, const uint32_t* s_prefix, uint32_t i) {
    if (q.base == nullptr) return i;
    uint32_t lo = 0, hi = PT_NSUB; // find lo with s_prefix[lo] <= i < s_prefix[lo+1]
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_prefix[mid] <= i) lo = mid; else hi = mid;
    }
    return q.base[(size_t)lo * q.sub_cap + (i - s_prefix[lo])];
}

// ------------------------------------------------------------------ generate
// raygen prologue of path i (sample-major over the chunk's pixel list): the camera ray and random state go to position `pos` of the
// queue-order arrays (k_generate: pos = i, the identity queue), what is summed per path to slot i
PT_DEV void generate_path(const PathState& st, const FrameParams& fp, const BatchParams& bp, uint32_t i, uint32_t pos) {
    const uint32_t vl = i / bp.npix, pix = i - vl * bp.npix;
    const uint32_t xy = bp.pixels[pix];
    const uint32_t x = xy & 0xffffu, y = xy >> 16;
    const uint32_t v = bp.s0 + vl, sub = v / fp.spp, sl = v - sub * fp.spp; // sample sl of subframe subframe_index + sub
    uint32_t seed = tea4(y * (uint32_t)fp.width + x, fp.subframe_index + sub);
    for (uint32_t k = 0; k < 2u * sl; ++k) lcg(seed); // earlier samples drew 2 rnd() each (:388)
    Rng r;
    r.init(seed); // prd.rand = Random(seed) BEFORE the jitter draws (:375-376)
    const float jx = rnd(seed), jy = rnd(seed);
    const float dx = 2.0f * (((float)x + jx) / (float)fp.width) - 1.0f;
    const float dy = 2.0f * (((float)y + jy) / (float)fp.height) - 1.0f;
    const v3 dir = normalize3(add3(add3(scl3(fp.U, dx), scl3(fp.V, dy)), fp.W));
    st.rayO[pos] = make_float4(fp.eye.x, fp.eye.y, fp.eye.z, 0.001f);
    st.rayD[pos] = make_float4(dir.x, dir.y, dir.z, 1e16f);
    st.rf[pos] = make_uint4(r.seed1, r.seed2, 0u, 0u);
    st.pflags[i] = 0u;
    if (st.vis) st.vis[i] = 0u;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    st.direct[i] = z;
    st.indirect[i] = z;
    if (st.prdN) { // shadow-catcher scenes
        st.alpha[i] = z;
        st.nrm[i] = (bp.carry && sl > 0) ? bp.pixNormal[pix] : z; // carry-in of the running per-pixel sum of this subframe
        st.alb[i] = (bp.carry && sl > 0) ? bp.pixAlbedo[pix] : z;
        st.prdN[i] = z;
        st.prdA[i] = z;
    }
}
__global__ void __launch_bounds__(256) k_generate(PathState st, FrameParams fp, BatchParams bp, uint32_t* qcount0) {
    const uint32_t total = bp.npix * bp.S;
    if (blockIdx.x == 0 && threadIdx.x == 0) *qcount0 = total;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) generate_path(st, fp, bp, i, i);
}

// ------------------------------------------------------------------ shade
struct PrimUV { // texcoords of the three vertices of a primitive (sbtData.texcoord[index.*])
    float2 c0, c1, c2;
};
// What a TEXTURED hit reads, as one aligned 64-byte record per leaf triangle (in leaf order, like tri_nrm): the vertices for the
// barycentrics and the three texcoords.  Before: 48 bytes of LeafTri (1.4 lines) + 24 bytes of PrimUV indexed by primitive (a random line).
struct TexTri {
    float4 a; // v0.xyz, v1.x
    float4 b; // v1.yz, v2.xy
    float4 c; // v2.z, uv0.xy, uv1.x
    float4 d; // uv1.y, uv2.xy, -
};
struct ShadeParams {
    const LeafTri* tris; // the traversal structure's leaf triangles: vertices, primitive and mesh of a hit (what sbtData.vertex[index[prim]] and the SBT record gave, :481-489)
    const float4* tri_nrm; // per leaf triangle: (normalize(cross(v1 - v0, v2 - v0)), mesh) — all the closest-hit program needs of an untextured hit, in ONE 16-byte load (k_shade_normals)
    const pt_material* mats;
    const int32_t* mesh_tex; // per mesh: texture id when the mesh has a texture AND texcoords, else -1 (null: no textures)
    const TexTri* textris; // scenes with textures only (k_emit_textris)
    const DevTex* textures;
    DevProbe probe;
    int max_depth;
    float tmin_radiance; // 0.001 (deviceProgram.cu:420); 0.01 in the sv4 variant
    QView queue;        // paths to shade (identity at bounce 0)
    QView next_queue;   // paths that continue
    QView shadow_queue; // paths with a live shadow ray
    float4 *oRayO, *oRayD, *oThr; // the continuing paths' state, at the positions of their entries in next_queue (PathState)
    uint4* oRf;
    int aov;            // write the first-hit normal/albedo (the foveated variants only keep accum/frame)
    int first;          // the queue holds camera rays: throughput (1,1,1) and eta 1 are not read (the generate kernels do not write them)
};

// wave-aggregated queue append: one atomic per wave (over the lanes that are active at the call: it may sit in divergent code).  Returns the
// position of the lane's entry in the queue's arrays (meaningful where pred holds); the slot is written there.
// LOCAL (pt_fused.h): the queue is a wave's private window of the arrays — q.counts is the wave's own count (LDS), q.sub_cap the window's first
// position; no atomic, the wave is the only producer.
template <bool LOCAL = false>
PT_DEV uint32_t queue_push(bool pred, uint32_t value, const QView& q) {
    if (LOCAL) {
        const unsigned long long lmask = __ballot(pred);
        if (lmask == 0ull) return 0u;
        const uint32_t llane = __lane_id();
        const uint32_t lbase = __hip_atomic_load(q.counts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (llane == (uint32_t)__ffsll((long long)lmask) - 1u) __hip_atomic_store(q.counts, lbase + (uint32_t)__popcll(lmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t lpos = q.sub_cap + lbase + (uint32_t)__popcll(lmask & ((1ull << llane) - 1ull));
        if (pred) q.base[lpos] = value;
        return lpos;
    }
    const uint32_t sub = blockIdx.x & (PT_NSUB - 1);
    uint32_t* counter = q.counts + sub * PT_CSTRIDE;
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t lane = __lane_id();
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    const uint32_t pos = sub * q.sub_cap + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (pred) q.base[pos] = value;
    return pos;
}

// __miss__radiance (:209-235): prd.normal = prd.albedo = 0 (adds nothing at depth 0), DONE
template <bool CATCHER>
PT_DEV void shade_miss(const PathState& st, const ShadeParams& sp, uint32_t pos, uint32_t p) {
    if (CATCHER) {
        st.prdN[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        st.prdA[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (sp.aov && (reinterpret_cast<const uint32_t*>(st.rf + pos)[2] & 0xffu) == 0u) { // primary miss: normal += 0, albedo += 0 (:424-427)
        st.nrm[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        st.alb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // DONE: the path is simply not queued again
}

// One closest-hit / miss invocation for the queue entry at `pos` = path slot p (the body of the reference's __closesthit__radiance /
// __miss__radiance plus the raygen loop's bookkeeping).  A path that continues is pushed into next_queue and its state written at the
// position of that entry; a pending shadow ray is pushed into shadow_queue with its record (or, with asynchronous shadow rays, written
// to the per-bounce record of the slot).
template <int MODE, bool CATCHER, bool LOCAL = false>
PT_DEV void shade_path(const PathState& st, const ShadeParams& sp, const ProbeMarg& pm, const float* u8lut, uint32_t pos, uint32_t p, float2 h) {
    const int32_t leaf = __float_as_int(h.y);
    if (leaf < 0) {
        shade_miss<CATCHER>(st, sp, pos, p);
        return;
    }
    bool push_next = false;
    const uint4 rf = st_ld<PT_NT_SHADE_LD>(&st.rf[pos]);
    int depth = (int)(rf.z & 0xffu);
    uint32_t flags = rf.z >> 8;
    uint2 rng_out = make_uint2(rf.x, rf.y);
    // what a continuing path carries on (the pass-through of a catcher keeps direction, throughput and random state)
    v3 P_out = mk3(0.f), dir_out = mk3(0.f);
    float4 thr_out = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // SampleLights' probe search (:252-334 → Probe.cuh:138-169) depends on the path's random state only: the first of its dependent loads
        // (guide → lines → texel) is started here, next to the hit's own chain (triangle → material); the rest runs where the reference calls it.
        // A pass-through hit of the shadow catcher draws nothing: its search is discarded with this copy of the state.
        Rng rand;
        rand.seed1 = rf.x;
        rand.seed2 = rf.y;
        int ps_row, ps_lo, ps_hi;
        float ps_r2;
        // PT_ABL_*: traffic-attribution builds (round 6, profiles/r6_05_shade_traffic.md) — what k_shade's measured bytes are made of, by leaving one
        // class of accesses out at a time; the images are wrong by construction, only the counters of the shade dispatches are read
#ifdef PT_ABL_NO_PROBE
        sample2d(rand, ps_r2, ps_r2); // (the two random numbers are still drawn: the paths stay the ones of the full build)
        ps_row = ps_lo = ps_hi = 0;
#else
        probe_search_begin(sp.probe, pm, rand, ps_row, ps_lo, ps_hi, ps_r2);
#endif
#ifdef PT_ABL_NO_TRINRM
        const float4 tn = make_float4(0.f, 1.f, 0.f, 0.f);
#else
        const float4 tn = sp.tri_nrm[leaf];
#endif
        const int32_t mesh = __float_as_int(tn.w);
        // the SBT record's material (:481-484): when every hit of the wave is on the same mesh the record comes through the scalar cache (k_shade
        // is bound by the address rate of its vector loads: six fewer per hit, +1.5 %).  (An LDS copy of the table was measured: no difference.)
        const int32_t mesh0 = __builtin_amdgcn_readfirstlane(mesh);
        pt_material mat;
        if (__all(mesh == mesh0)) {
#if __HIP_DEVICE_COMPILE__
            typedef const __attribute__((address_space(4))) pt_material* ConstMat; // constant address space: s_load
            mat = ((ConstMat)(uintptr_t)sp.mats)[mesh0];
#endif
        } else {
            mat = sp.mats[mesh];
        }
        const float4 o4 = st_ld<PT_NT_SHADE_LD>(&st.rayO[pos]), d4 = st_ld<PT_NT_SHADE_LD>(&st.rayD[pos]);
        const float4 th = sp.first ? make_float4(1.f, 1.f, 1.f, 1.f) : st_ld<PT_NT_SHADE_LD>(&st.thr[pos]); // pathThroughput = 1, rayEta = 1 (deviceProgram.cu:379-380)
        dir_out = mk3(d4.x, d4.y, d4.z);
        thr_out = th;
        const v3 ray_o = mk3(o4.x, o4.y, o4.z), ray_dir = mk3(d4.x, d4.y, d4.z);
        const v3 N_0 = mk3(tn.x, tn.y, tn.z); // normalize(cross(v1 - v0, v2 - v0)) (:491), evaluated once per triangle by the same device code
        const v3 N = faceforward3(N_0, neg3(ray_dir), N_0);
        const v3 P = add3(ray_o, scl3(ray_dir, h.x));
        P_out = P;
        const bool is_catcher = (mat.flags & 1) != 0;
        if (CATCHER && is_catcher && (flags & FLAG_SECONDARY)) {
            // pass-through (:503-508): origin = P, direction unchanged, --depth; then raygen (:424-439)
            --depth;
            if (depth == 0) {
                const float4 a = st.nrm[p], b = st.alb[p], pn = st.prdN[p], pa = st.prdA[p];
                st.nrm[p] = make_float4(a.x + pn.x, a.y + pn.y, a.z + pn.z, 0.f);
                st.alb[p] = make_float4(b.x + pa.x, b.y + pa.y, b.z + pa.z, 0.f);
            }
            ++depth;
            push_next = true;
        } else {
            v3 albedo = mk3(mat.color[0], mat.color[1], mat.color[2]);
            if (sp.mesh_tex) { // deviceProgram.cu:512-523: a textured mesh's albedo is REPLACED by tex2D at the hit's texcoord
                const int tid = sp.mesh_tex[mesh];
                if (tid >= 0) {
                    const TexTri tt = sp.textris[leaf];
                    const v3 v0 = mk3(tt.a.x, tt.a.y, tt.a.z), v1 = mk3(tt.a.w, tt.b.x, tt.b.y), v2 = mk3(tt.b.z, tt.b.w, tt.c.x);
                    // optixGetTriangleBarycentrics = (weight of vertex 1, weight of vertex 2): the hit test's own weights
                    const v3 A = sub3(v0, ray_o), B = sub3(v1, ray_o), C = sub3(v2, ray_o);
                    const v3 CxB = cross3(C, B), AxC = cross3(A, C), BxA = cross3(B, A);
                    const float Uw = dot3(ray_dir, CxB), Vw = dot3(ray_dir, AxC), Ww = dot3(ray_dir, BxA);
                    const float det = Uw + Vw + Ww;
                    const float bu = Vw / det, bv = Ww / det;
                    const float w0 = 1.f - bu - bv; // texcoords of optixGetPrimitiveIndex()'s vertices, (1 - u - v, u, v) weighted (:515-518)
                    const float tcx = w0 * tt.c.y + bu * tt.c.w + bv * tt.d.y;
                    const float tcy = w0 * tt.c.z + bu * tt.d.x + bv * tt.d.z;
                    const float4 tx = tex2d_wrap_linear(sp.textures[tid], tcx, tcy, u8lut);
                    albedo = mk3(tx.x, tx.y, tx.z);
                }
            }
            const v3 T_old = mk3(th.x, th.y, th.z);
            float rayEta = th.w;
            const float outEta = (rayEta == 1.0f) ? material_ior(mat) : 1.0f;
            const v3 wo = neg3(ray_dir);
            // SampleLights / SampleShadow (:252-334) up to the visibility test
            v3 wi, skyColor;
            float skyPdf;
            // (started at the reference's position: −0.3 % frame; the candidate lines' loads started early as well: 18 more live registers, spills, slower)
#ifdef PT_ABL_NO_PROBE
            wi = mk3(0.f, 1.f, 0.f); skyColor = mk3(1.f, 1.f, 1.f); skyPdf = 1.f + ps_r2;
#else
            probe_search_end(sp.probe, pm, ps_row, ps_lo, ps_hi, ps_r2, wi, skyColor, skyPdf);
#endif
            bool has_val = false;
            v3 val = mk3(0.f);
            {
                const float bsdfPdf = bsdf_pdf<MODE>(mat, rayEta, outEta, N, wo, wi);
                const v3 f = bsdf_eval<MODE>(mat, albedo, rayEta, outEta, N, wo, wi);
                if (bsdfPdf > 0.0f) {
                    const float weight = 0.5f * skyPdf / (0.5f * bsdfPdf + 0.5f * skyPdf);
                    if (weight > 0.0f) {
                        val = scl3(div3s(scl3(mul3(scl3(skyColor, weight), f), fabsf(dot3(wi, N))), skyPdf), 1.0f);
                        has_val = true;
                    }
                }
            }
            if (CATCHER) {
                if (!is_catcher) st.alpha[p] = make_float4(1.f, 1.f, 1.f, 0.f); // :547
            } else {
                if (!(flags & FLAG_HIT0)) st.pflags[p] = FLAG_HIT0; // :547 — every hit assigns the same value: one bit, written by the path's first hit
                flags |= FLAG_HIT0;
            }
            const bool primary = (flags & FLAG_SECONDARY) == 0;
            v3 u, v, bsdfDir = mk3(0.f);
            float bsdfPdf;
            basis_from_vector(N, u, v);
            bsdf_sample<MODE>(mat, rayEta, outEta, u, v, N, wo, bsdfDir, bsdfPdf, rand);
            v3 T_new = T_old;
            if (bsdfPdf <= 0.0f) {
                flags |= FLAG_DONE; // :570-573
            } else {
                const v3 f = bsdf_eval<MODE>(mat, albedo, rayEta, outEta, N, wo, bsdfDir);
                if (dot3(bsdfDir, N) <= 0.0f) rayEta = outEta;
                T_new = mul3(T_old, div3s(scl3(f, fabsf(dot3(N, bsdfDir))), bsdfPdf));
                dir_out = bsdfDir;
                flags |= FLAG_SECONDARY;
            }
            thr_out = make_float4(T_new.x, T_new.y, T_new.z, rayEta);
            rng_out = make_uint2(rand.seed1, rand.seed2);
            if (CATCHER) {
                st.prdN[p] = make_float4(N.x, N.y, N.z, 0.f);
                st.prdA[p] = make_float4(albedo.x, albedo.y, albedo.z, 0.f);
            }
            // raygen loop body (:424-439)
            if (depth == 0) {
                if (CATCHER) {
                    const float4 a = st.nrm[p], b = st.alb[p];
                    st.nrm[p] = make_float4(a.x + N.x, a.y + N.y, a.z + N.z, 0.f);
                    st.alb[p] = make_float4(b.x + albedo.x, b.y + albedo.y, b.z + albedo.z, 0.f);
                } else if (sp.aov) { // the only depth-0 closest hit of this path: 0 + x, written once (0 + -0 = +0 as in the sum)
                    st.nrm[p] = make_float4(0.f + N.x, 0.f + N.y, 0.f + N.z, 0.f);
                    st.alb[p] = make_float4(0.f + albedo.x, 0.f + albedo.y, 0.f + albedo.z, 0.f);
                }
            }
            const bool term = (flags & FLAG_DONE) || depth >= sp.max_depth;
            // direct is still the +0 k_generate wrote when the camera ray's hit adds the emission: +0 + (±0) = +0, so a material without emission
            // leaves the sum as it is, bit for bit, and the read-modify-write of 32 bytes per primary hit is skipped
            const bool emissive = mat.emission[0] != 0.f || mat.emission[1] != 0.f || mat.emission[2] != 0.f;
            const v3 contrib = mul3(T_old, val);
            if (CATCHER && is_catcher) {
                // SampleShadow: alpha += T * shadowSample when OCCLUDED (:550-551), whatever happens next
                if (has_val) {
                    const uint32_t sq = queue_push<LOCAL>(true, p, sp.shadow_queue);
                    st_st<PT_NT_SHADE_ST>(&st.shO[sq], make_float4(P.x, P.y, P.z, 0.f));
                    st_st<PT_NT_SHADE_ST>(&st.shD[sq], make_float4(wi.x, wi.y, wi.z, 0.f));
                    st_st<PT_NT_SHADE_ST>(&st.shPend[sq], make_float4(contrib.x, contrib.y, contrib.z, __int_as_float(PEND_ALPHA)));
                }
                if (!term && primary && emissive) { // radiance = emission (:558-560)
                    const float4 dd = st.direct[p];
                    st.direct[p] = make_float4(dd.x + mat.emission[0], dd.y + mat.emission[1], dd.z + mat.emission[2], 0.f);
                }
            } else if (!term) {
                // radiance = T*lightSample (+ emission on primary hits) is added to direct/indirect (:432-437)
                // only when the path goes on; the visibility-dependent part is deferred to the traversal kernel's write-back (k_trace8).
                if (primary && emissive) {
                    const float4 dd = st.direct[p];
                    st.direct[p] = make_float4(dd.x + mat.emission[0], dd.y + mat.emission[1], dd.z + mat.emission[2], 0.f);
                }
                if (has_val) {
                    const float4 pe = make_float4(contrib.x, contrib.y, contrib.z, __int_as_float(depth == 0 ? PEND_DIRECT : PEND_INDIRECT));
                    const uint32_t sq = queue_push<LOCAL>(true, p, sp.shadow_queue);
                    if (st.vis) { // asynchronous shadow rays: a self-contained record of this bounce, by slot (the queue entry names the slot)
                        const size_t bi = (size_t)depth * st.bstride + p;
                        st.sO[bi] = make_float4(P.x, P.y, P.z, 0.f);
                        st.sD[bi] = make_float4(wi.x, wi.y, wi.z, 0.f);
                        st.pendB[bi] = pe;
                    } else {
#ifndef PT_ABL_NO_SHADOW_ST
                        st_st<PT_NT_SHADE_ST>(&st.shO[sq], make_float4(P.x, P.y, P.z, 0.f));
                        st_st<PT_NT_SHADE_ST>(&st.shD[sq], make_float4(wi.x, wi.y, wi.z, 0.f));
                        st_st<PT_NT_SHADE_ST>(&st.shPend[sq], pe);
#else
                        if (sq == 0xffffffffu) st.shPend[0] = pe; // (keeps the values live)
#endif
                    }
                }
            }
            if (!term) {
                ++depth;
                push_next = true;
            }
        }
    }
    if (LOCAL) {
        // the fused loop has no launch that simply is not made: a path that "would continue" past the depth cutoff (the provably dead trace at
        // depth == max_depth, pt_api.hip enqueue_chunk) is counted as a shaded hit (sp.next_queue.counts[1]) and not queued
        const unsigned long long cm = __ballot(push_next);
        if (cm != 0ull && __lane_id() == (uint32_t)__ffsll((long long)cm) - 1u) {
            uint32_t* hc = sp.next_queue.counts + 1;
            __hip_atomic_store(hc, __hip_atomic_load(hc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + (uint32_t)__popcll(cm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        if (!CATCHER && depth >= sp.max_depth) push_next = false;
    }
    const uint32_t nq = queue_push<LOCAL>(push_next, p, sp.next_queue);
#ifdef PT_ABL_NO_STATE_ST
    if (push_next && nq == 0xffffffffu) { // (never true: keeps the values live without the stores)
#else
    if (push_next) {
#endif
        st_st<PT_NT_SHADE_ST>(&sp.oRayO[nq], make_float4(P_out.x, P_out.y, P_out.z, sp.tmin_radiance));
        st_st<PT_NT_SHADE_ST>(&sp.oRayD[nq], make_float4(dir_out.x, dir_out.y, dir_out.z, 1e16f));
        st_st<PT_NT_SHADE_ST>(&sp.oThr[nq], thr_out);
        st_st<PT_NT_SHADE_ST>(&sp.oRf[nq], make_uint4(rng_out.x, rng_out.y, (uint32_t)depth | (flags << 8), 0u));
    }
}

// k_shade waits on memory half of the time (dense state, then the triangle record and the probe search's scattered lines): 5 waves per SIMD
// at 96 VGPRs measured +1.5 % on C2 and C3 over the compiler's 4 waves at 109 in round 2 and again with the round-3 kernel (4: +0.7 % frame
// time, 6: ±0 with 8 spills, 7: slower)
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 5
#endif
#define PT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(PT_SHADE_WAVES, PT_SHADE_WAVES)))

// rows of the probe whose marginal arrays fit k_shade's LDS copy: cdfY + pdfY (rows each), c8Y (rows/8), c64Y (rows/64 padded to 8)
#ifndef PT_LDS_PROBE_ROWS
#define PT_LDS_PROBE_ROWS 2048
#endif

template <int MODE, bool CATCHER>
__global__ void __launch_bounds__(256) PT_SHADE_ATTR k_shade(PathState st, ShadeParams sp) {
    __shared__ uint32_t s_prefix[PT_NSUB + 1];
    __shared__ float s_u8[256]; // textured scenes: (float)b / 255.0f for the 256 byte values (tex2d_wrap_linear)
    extern __shared__ __attribute__((aligned(16))) float s_marg[]; // sized at launch for the probe in use (shade_lds_bytes): 8.8 KB for 1024 rows
    const float* u8lut = nullptr;
    if (sp.mesh_tex) { // ordered before its readers by the barrier of qreader_init below
        s_u8[threadIdx.x & 255u] = (float)(threadIdx.x & 255u) / 255.0f;
        u8lut = s_u8;
    }
    ProbeMarg pm = probe_marg_global(sp.probe);
    if (sp.probe.c64Y && sp.probe.height <= PT_LDS_PROBE_ROWS) {
        // ProbeSample's row search (cdfY through its 64- and 8-entry count tables) and pdfY from LDS: 3 of the 6 dependent
        // loads of the lookup stay on the CU (hard part 3)
        const int h = sp.probe.height, h8 = h / 8, n64 = (sp.probe.ncy + 7) & ~7;
        float* s_cdfY = s_marg;
        float* s_pdfY = s_cdfY + h;
        float* s_c8Y = s_pdfY + h;
        float* s_c64Y = s_c8Y + h8;
        for (int k = threadIdx.x; k < h; k += blockDim.x) {
            s_cdfY[k] = sp.probe.cdfY[k];
            s_pdfY[k] = sp.probe.pdfY[k];
        }
        for (int k = threadIdx.x; k < h8; k += blockDim.x) s_c8Y[k] = sp.probe.c8Y[k];
        for (int k = threadIdx.x; k < n64; k += blockDim.x) s_c64Y[k] = sp.probe.c64Y[k];
        pm = ProbeMarg{s_cdfY, s_pdfY, s_c64Y, s_c8Y};
    }
    const uint32_t n = qreader_init(sp.queue, s_prefix); // ends in a workgroup barrier (identity queue: see below)
    if (sp.queue.base == nullptr) __syncthreads();
    const uint32_t nround = (n + 63u) & ~63u; // whole waves stay in the loop so ballots see every lane
    // (Measured and rejected: each wave parking its hits in LDS until 64 are waiting, so that the closest-hit program always runs on full waves
    // and the misses — a third of the queue — only cost their stores: 31 % fewer wave passes through the hit path, k_shade 6 % SLOWER.  The
    // kernel is bound by the address rate of its scattered per-lane loads, which a fuller wave does not lower, and gathered lanes coalesce worse.)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += gridDim.x * blockDim.x) {
        if (i < n) {
            const uint32_t pos = qreader_pos(sp.queue, s_prefix, i);
            shade_path<MODE, CATCHER>(st, sp, pm, u8lut, pos, qslot(sp.queue, pos), st_ld<PT_NT_SHADE_LD>(&st.hit[pos]));
        }
    }
}

// ------------------------------------------------------------------ resolve
// asynchronous shadow rays: the deferred `sum += val` of SampleLights, in bounce order (bounce 0 → direct, later → indirect)
PT_DEV void apply_visible_contributions(const PathState& st, uint32_t i, float4& d, float4& in) {
    if (!st.vis) return;
    uint32_t v = st.vis[i];
    if (v & 1u) {
        const float4 pe = st.pendB[i];
        d = make_float4(d.x + pe.x, d.y + pe.y, d.z + pe.z, 0.f);
    }
    for (uint32_t b = 1; (v >> b) != 0u; ++b)
        if ((v >> b) & 1u) {
            const float4 pe = st.pendB[(size_t)b * st.bstride + i];
            in = make_float4(in.x + pe.x, in.y + pe.y, in.z + pe.z, 0.f);
        }
}
// One pass over bp.S virtual samples of every pixel of the chunk.  Samples are summed in order; whenever a subframe's last sample
// has been added the raygen epilogue runs for that subframe (:445-474) — blending with the PREVIOUS subframe's accum value, which is
// the register of this loop when the previous subframe was completed in the same pass and accum_buffer otherwise — so that a batch
// of `count` subframes leaves exactly what `count` launches would have left: the stored float4 is the value the next launch re-reads.
// A pass that ends inside a subframe parks the partial sums in the chunk's pix* arrays; the frame buffers are written by passes that
// completed at least one subframe (all five, with the values of the last completed one).
__global__ void __launch_bounds__(256) k_resolve(PathState st, FrameParams fp, BatchParams bp) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= bp.npix) return;
    v3 result = mk3(0.f), alpha = mk3(0.f), normal = mk3(0.f), albedo = mk3(0.f);
    uint32_t sub = bp.s0 / fp.spp, sl = bp.s0 - sub * fp.spp; // position of the pass's first sample
    if (sl != 0) { // the subframe was begun by an earlier pass
        const float4 r = bp.pixResult[pix], a = bp.pixAlpha[pix], nn = bp.pixNormal[pix], al = bp.pixAlbedo[pix];
        result = mk3(r.x, r.y, r.z);
        alpha = mk3(a.x, a.y, a.z);
        normal = mk3(nn.x, nn.y, nn.z);
        albedo = mk3(al.x, al.y, al.z);
    }
    const uint32_t xy = bp.pixels[pix];
    const uint32_t x = xy & 0xffffu, y = xy >> 16;
    const size_t image_index = (size_t)y * fp.width + x;
    const float spp = (float)fp.spp;
    bool have = false; // a subframe was completed in this pass
    v3 accum_cur = mk3(0.f), normal_fin = mk3(0.f), albedo_fin = mk3(0.f);
    for (uint32_t vl = 0; vl < bp.S; ++vl) {
        const uint32_t i = vl * bp.npix + pix;
        float4 d = st.direct[i], in = st.indirect[i];
        const float4 nn = st.nrm[i], al = st.alb[i];
        float4 a;
        if (st.prdN) {
            a = st.alpha[i];
        } else {
            const float hit0 = (st.pflags[i] & FLAG_HIT0) ? 1.0f : 0.0f;
            a = make_float4(hit0, hit0, hit0, 0.f);
        }
        apply_visible_contributions(st, i, d, in);
        // result += directLight + indirectLight; alpha += prd.alpha (:445-446)
        result = add3(result, add3(mk3(d.x, d.y, d.z), mk3(in.x, in.y, in.z)));
        alpha = add3(alpha, mk3(a.x, a.y, a.z));
        if (bp.carry) {
            normal = mk3(nn.x, nn.y, nn.z);
            albedo = mk3(al.x, al.y, al.z);
        } else {
            normal = add3(normal, mk3(nn.x, nn.y, nn.z));
            albedo = add3(albedo, mk3(al.x, al.y, al.z));
        }
        if (++sl < fp.spp) continue;
        // ---- subframe complete
        const uint32_t subframe_index = fp.subframe_index + sub;
        normal_fin = div3s(normal, spp);
        albedo_fin = div3s(albedo, spp);
        alpha = div3s(alpha, spp);
        // backplate of the LAST sample's camera ray (:410)
        uint32_t seed = tea4(y * (uint32_t)fp.width + x, subframe_index);
        for (uint32_t k = 0; k < 2u * (fp.spp - 1u); ++k) lcg(seed);
        const float jx = rnd(seed), jy = rnd(seed);
        const float dx = 2.0f * (((float)x + jx) / (float)fp.width) - 1.0f;
        const float dy = 2.0f * (((float)y + jy) / (float)fp.height) - 1.0f;
        const v3 dir = normalize3(add3(add3(scl3(fp.U, dx), scl3(fp.V, dy)), fp.W));
        float pu, pv;
        probe_dir_to_uv(dir, pu, pv);
        const float4 bpx = probe_eval(fp.probe, pu, pv);
        const v3 backplate = mk3(bpx.x, bpx.y, bpx.z);
        const v3 color = add3(mul3(scl3(backplate, spp), sub3(mk3(1.0f), alpha)), result);
        v3 accum_color = div3s(color, spp);
        if (subframe_index > 0) {
            accum_color = mk3(clampf(accum_color.x, 0.0f, 10.0f), clampf(accum_color.y, 0.0f, 10.0f), clampf(accum_color.z, 0.0f, 10.0f));
            const float w = 1.0f / (float)(subframe_index + 1);
            v3 prev = accum_cur;
            if (!have) {
                const float4 p4 = fp.accum[image_index];
                prev = mk3(p4.x, p4.y, p4.z);
            }
            accum_color = lerp3(prev, accum_color, w);
        }
        accum_cur = accum_color;
        have = true;
        result = alpha = normal = albedo = mk3(0.f);
        sl = 0;
        ++sub;
    }
    if (sl != 0) {
        bp.pixResult[pix] = make_float4(result.x, result.y, result.z, 0.f);
        bp.pixAlpha[pix] = make_float4(alpha.x, alpha.y, alpha.z, 0.f);
        bp.pixNormal[pix] = make_float4(normal.x, normal.y, normal.z, 0.f);
        bp.pixAlbedo[pix] = make_float4(albedo.x, albedo.y, albedo.z, 0.f);
    }
    if (!have) return;
    fp.accum[image_index] = make_float4(accum_cur.x, accum_cur.y, accum_cur.z, 1.0f);
    fp.frame[image_index] = make_color(accum_cur);
    fp.normal[image_index] = make_float4(normal_fin.x, normal_fin.y, normal_fin.z, 1.0f);
    fp.color[image_index] = make_float4(accum_cur.x, accum_cur.y, accum_cur.z, 1.0f);
    fp.albedo[image_index] = make_float4(albedo_fin.x, albedo_fin.y, albedo_fin.z, 1.0f);
}

// ------------------------------------------------------------------ foveated variant (HelloPathtracing_sv4_vmv23/)
// One optixLaunch of the sv4 raygen (deviceProgram.cu:388-590): launch index → pixel = index*factor + offset, early
// return outside the annulus [r_inner,r_outer] around c, splat over fillSize^2 pixels, blend only when
// subframe_index > 0 && !redraw, write accum_buffer + frame_buffer (exposure, Reinhard, make_color) only.
struct RegionParams { // LaunchParams.frame.{factor,fillSize,c,r_inner,r_outer,offset,redraw} (sv4 LaunchParams.h:62-70)
    uint32_t launch_w, launch_h;
    uint32_t factor_x, factor_y;
    int32_t fill_size;
    uint32_t cx, cy;
    float r_inner, r_outer;
    uint32_t offset_x, offset_y;
    uint32_t redraw;
    uint32_t spp;
    uint32_t subframe_index;
};
struct PartParams { // image partition of the context (pt_set_partition): pixel (x,y) belongs to rank (x / tile_w + y / tile_h) % world
    int rank, world, tile_w, tile_h;
};
PT_DEV bool part_owns(const PartParams& pp, uint32_t px, uint32_t py) {
    return pp.world <= 1 || (int)((px / (uint32_t)pp.tile_w + py / (uint32_t)pp.tile_h) % (uint32_t)pp.world) == pp.rank;
}
struct VariantParams {
    float radiance_tmin;
    int cull_back_occlusion;
    int tonemap;
    float exposure, white;
    int initial_depth, write_aov;
};

// paths: i = sample * nl + (launch index - l0); culled launch indices are flagged and never queued
__global__ void __launch_bounds__(256) k_generate_region(PathState st, FrameParams fp, RegionParams rg, PartParams pp, float tmin, uint32_t depth0, uint32_t l0, uint32_t nl, QView qgen) {
    const uint32_t total = nl * rg.spp;
    const uint32_t nround = (total + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += gridDim.x * blockDim.x) {
        bool live = false;
        v3 dir = mk3(0.f);
        uint32_t seed1 = 0, seed2 = 0;
        if (i < total) {
            const uint32_t sl = i / nl, li = l0 + (i - sl * nl);
            const uint32_t lx = li % rg.launch_w, ly = li / rg.launch_w;
            uint32_t seed = tea4(ly * (uint32_t)fp.width + lx, rg.subframe_index); // the LAUNCH index seeds (:406)
            const uint32_t ix = lx * rg.factor_x + rg.offset_x, iy = ly * rg.factor_y + rg.offset_y;
            const v3 dv = sub3(mk3((float)ix, (float)iy, 0.0f), mk3((float)rg.cx, (float)rg.cy, 0.0f));
            const float range = sqrtf(dot3(dv, dv));
            // On a partitioned image (multi-GPU) a launch index is traced by every rank that owns at least one pixel of its
            // fillSize^2 splat (the pixel coordinates are the resolve kernel's: u32 arithmetic, clamped to the image); a splat
            // that straddles a tile border is traced by both neighbours — same seeds, same bits — and each writes only its pixels.
            bool mine = pp.world <= 1;
            for (int fi = 0; fi < rg.fill_size && !mine; ++fi)
                for (int fj = 0; fj < rg.fill_size && !mine; ++fj) {
                    uint32_t px = lx * rg.factor_x + (uint32_t)fi + rg.offset_x, py = ly * rg.factor_y + (uint32_t)fj + rg.offset_y;
                    px = px > (uint32_t)(fp.width - 1) ? (uint32_t)(fp.width - 1) : px;
                    py = py > (uint32_t)(fp.height - 1) ? (uint32_t)(fp.height - 1) : py;
                    mine = part_owns(pp, px, py);
                }
            if (range < rg.r_inner || range > rg.r_outer || !mine) {
                st.pflags[i] = FLAG_CULLED;
            } else {
                for (uint32_t k = 0; k < 2u * sl; ++k) lcg(seed);
                Rng r;
                r.init(seed);
                const float jx = rnd(seed), jy = rnd(seed);
                const float dx = 2.0f * (((float)ix + jx) / (float)fp.width) - 1.0f;
                const float dy = 2.0f * (((float)iy + jy) / (float)fp.height) - 1.0f;
                dir = normalize3(add3(add3(scl3(fp.U, dx), scl3(fp.V, dy)), fp.W));
                seed1 = r.seed1;
                seed2 = r.seed2;
                st.pflags[i] = 0u;
                if (st.vis) st.vis[i] = 0u;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                st.direct[i] = z;
                st.indirect[i] = z;
                if (st.prdN) {
                    st.alpha[i] = z;
                    st.nrm[i] = z;
                    st.alb[i] = z;
                    st.prdN[i] = z;
                    st.prdA[i] = z;
                }
                live = true;
            }
        }
        const uint32_t pos = queue_push(live, i, qgen); // the path's state goes where its queue entry went
        if (live) {
            st.rayO[pos] = make_float4(fp.eye.x, fp.eye.y, fp.eye.z, tmin);
            st.rayD[pos] = make_float4(dir.x, dir.y, dir.z, 1e16f);
            st.rf[pos] = make_uint4(seed1, seed2, depth0, 0u); // prd.depth = 0 (1 in the sv / sv2 variants)
        }
    }
}

PT_DEV v3 reinhard_tonemap(v3 color, float white) { // sv4 deviceProgram.cu:124-128
    const float luminance = 0.2126f * color.x + 0.7152f * color.y + 0.0722f * color.z;
    return div3s(scl3(color, 1.0f), 1.0f + luminance / white);
}

__global__ void __launch_bounds__(256) k_resolve_region(PathState st, FrameParams fp, RegionParams rg, PartParams pp, VariantParams var, uint32_t l0, uint32_t nl) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nl) return;
    if (st.pflags[k] & FLAG_CULLED) return;
    const uint32_t li = l0 + k;
    const uint32_t lx = li % rg.launch_w, ly = li / rg.launch_w;
    v3 result = mk3(0.f), alpha = mk3(0.f), normal = mk3(0.f), albedo = mk3(0.f);
    const bool aov_sums = var.write_aov && var.initial_depth == 0; // with prd.depth starting at 1 nothing is ever added (:474-477)
    for (uint32_t sl = 0; sl < rg.spp; ++sl) {
        const uint32_t i = sl * nl + k;
        float4 d = st.direct[i], in = st.indirect[i];
        const float hit0 = (st.pflags[i] & FLAG_HIT0) ? 1.0f : 0.0f; // foveated launches never run with shadow catchers
        apply_visible_contributions(st, i, d, in);
        result = add3(result, add3(mk3(d.x, d.y, d.z), mk3(in.x, in.y, in.z)));
        alpha = add3(alpha, mk3(hit0, hit0, hit0));
        if (aov_sums) {
            const float4 nn = st.nrm[i], al = st.alb[i];
            normal = add3(normal, mk3(nn.x, nn.y, nn.z));
            albedo = add3(albedo, mk3(al.x, al.y, al.z));
        }
    }
    const float spp = (float)rg.spp;
    alpha = div3s(alpha, spp);
    normal = div3s(normal, spp);
    albedo = div3s(albedo, spp);
    // backplate of the last sample's camera ray
    uint32_t seed = tea4(ly * (uint32_t)fp.width + lx, rg.subframe_index);
    for (uint32_t q = 0; q < 2u * (rg.spp - 1u); ++q) lcg(seed);
    const uint32_t ix = lx * rg.factor_x + rg.offset_x, iy = ly * rg.factor_y + rg.offset_y;
    const float jx = rnd(seed), jy = rnd(seed);
    const float dx = 2.0f * (((float)ix + jx) / (float)fp.width) - 1.0f;
    const float dy = 2.0f * (((float)iy + jy) / (float)fp.height) - 1.0f;
    const v3 dir = normalize3(add3(add3(scl3(fp.U, dx), scl3(fp.V, dy)), fp.W));
    float pu, pv;
    probe_dir_to_uv(dir, pu, pv);
    const float4 bpx = probe_eval(fp.probe, pu, pv);
    const v3 backplate = mk3(bpx.x, bpx.y, bpx.z);
    for (int fi = 0; fi < rg.fill_size; ++fi) {
        for (int fj = 0; fj < rg.fill_size; ++fj) {
            uint32_t px = lx * rg.factor_x + (uint32_t)fi + rg.offset_x, py = ly * rg.factor_y + (uint32_t)fj + rg.offset_y;
            px = px > (uint32_t)(fp.width - 1) ? (uint32_t)(fp.width - 1) : px;
            py = py > (uint32_t)(fp.height - 1) ? (uint32_t)(fp.height - 1) : py;
            if (!part_owns(pp, px, py)) continue;
            const size_t image_index = (size_t)py * fp.width + px;
            const v3 color = add3(mul3(scl3(backplate, spp), sub3(mk3(1.0f), alpha)), result);
            v3 accum_color = div3s(color, spp);
            if (rg.subframe_index > 0 && !rg.redraw) {
                accum_color = mk3(clampf(accum_color.x, 0.0f, 10.0f), clampf(accum_color.y, 0.0f, 10.0f), clampf(accum_color.z, 0.0f, 10.0f));
                const float a = 1.0f / (float)(rg.subframe_index + 1);
                const float4 prev = fp.accum[image_index];
                accum_color = lerp3(mk3(prev.x, prev.y, prev.z), accum_color, a);
            }
            fp.accum[image_index] = make_float4(accum_color.x, accum_color.y, accum_color.z, 1.0f);
            v3 shown = accum_color;
            if (var.tonemap == 1) shown = reinhard_tonemap(scl3(accum_color, var.exposure), var.white);
            else if (var.tonemap == 2) shown = scl3(accum_color, var.exposure); // sv3: its Reinhard write is overwritten (sv3 :591-604)
            fp.frame[image_index] = make_color(shown);
            if (var.write_aov) { // sv / sv2 (HelloPathtracing_sv/deviceProgram.cu:553-555)
                fp.normal[image_index] = make_float4(normal.x, normal.y, normal.z, 1.0f);
                fp.color[image_index] = make_float4(accum_color.x, accum_color.y, accum_color.z, 1.0f);
                fp.albedo[image_index] = make_float4(albedo.x, albedo.y, albedo.z, 1.0f);
            }
        }
    }
}

// toneMap.cu:41-58 computeFinalPixelColorsKernel
__global__ void k_tonemap_sqrt(const float4* __restrict__ accum, uint32_t* __restrict__ frame, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = accum[i];
    const uint32_t r = (uint32_t)(255.9f * fminf(1.0f, fmaxf(0.0f, sqrtf(f.x))));
    const uint32_t g = (uint32_t)(255.9f * fminf(1.0f, fmaxf(0.0f, sqrtf(f.y))));
    const uint32_t b = (uint32_t)(255.9f * fminf(1.0f, fmaxf(0.0f, sqrtf(f.z))));
    const uint32_t a = (uint32_t)(255.9f * fminf(1.0f, fmaxf(0.0f, sqrtf(f.w))));
    frame[i] = r | (g << 8) | (b << 16) | (a << 24);
}

// One a-trous pass (pt_amd.h pt_denoise): 25 taps at spacing `step`, fixed row-major summation order, expf = pt_expf
// (include/pt_detmath.h) so that the CPU checker reproduces every bit.  inv_* are 1/sigma^2 of this pass.
struct AtrousParams {
    int width, height, step;
    float inv_color, inv_normal, inv_albedo;
};
__global__ void __launch_bounds__(256) k_atrous(const float4* __restrict__ src, const float4* __restrict__ nrm, const float4* __restrict__ alb,
                                                float4* __restrict__ dst, AtrousParams p) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= p.width || y >= p.height) return;
    const size_t ip = (size_t)y * p.width + x;
    const float4 cp = src[ip], np = nrm[ip], ap = alb[ip];
    const float kern[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {
        const int qy = y + dy * p.step;
        if (qy < 0 || qy >= p.height) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int qx = x + dx * p.step;
            if (qx < 0 || qx >= p.width) continue;
            const size_t iq = (size_t)qy * p.width + qx;
            const float4 cq = src[iq], nq = nrm[iq], aq = alb[iq];
            const float dcx = cp.x - cq.x, dcy = cp.y - cq.y, dcz = cp.z - cq.z;
            const float dnx = np.x - nq.x, dny = np.y - nq.y, dnz = np.z - nq.z;
            const float dax = ap.x - aq.x, day = ap.y - aq.y, daz = ap.z - aq.z;
            const float ec = (dcx * dcx + dcy * dcy + dcz * dcz) * p.inv_color;
            const float en = (dnx * dnx + dny * dny + dnz * dnz) * p.inv_normal;
            const float ea = (dax * dax + day * day + daz * daz) * p.inv_albedo;
            const float w = pt_expf(-fminf(ec + en + ea, 80.0f)) * (kern[dy + 2] * kern[dx + 2]); // pt_expf is defined on |x| < 80
            sx += cq.x * w;
            sy += cq.y * w;
            sz += cq.z * w;
            sw += w;
        }
    }
    dst[ip] = make_float4(sx / sw, sy / sw, sz / sw, cp.w);
}
__global__ void k_make_color(const float4* __restrict__ src, uint32_t* __restrict__ frame, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = src[i];
    frame[i] = make_color(mk3(f.x, f.y, f.z));
}

// multi-GPU exchange: gather owned pixels of a buffer into a packed array and back
template <typename T>
__global__ void k_pack(const T* __restrict__ buf, const uint32_t* __restrict__ pixels, uint32_t n, int width, T* __restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t xy = pixels[i];
    dst[i] = buf[(size_t)(xy >> 16) * width + (xy & 0xffffu)];
}
template <typename T>
__global__ void k_unpack(T* __restrict__ buf, const uint32_t* __restrict__ all_pixels, uint32_t n, int width, const T* __restrict__ src) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t xy = all_pixels[i];
    if (xy == 0xffffffffu) return;
    buf[(size_t)(xy >> 16) * width + (xy & 0xffffu)] = src[i];
}

// counters: [nq][PT_NSUB*PT_CSTRIDE] radiance sub-queue counts, then the same for shadow queues
__global__ void k_accum_stats(const uint32_t* __restrict__ counters, int nq, int nb, int count_hits, unsigned long long* __restrict__ totals) {
    unsigned long long r = 0, s = 0, h = 0; // one wave: lane q sums sub-queue q over the traced bounces
    const uint32_t q = threadIdx.x;
    for (int b = 0; b < nb; ++b) {
        r += counters[(size_t)b * PT_NSUB * PT_CSTRIDE + q * PT_CSTRIDE];
        s += counters[(size_t)(nq + b) * PT_NSUB * PT_CSTRIDE + q * PT_CSTRIDE];
        // closest hits whose BSDF sample was accepted = the paths k_shade(b) queued for bounce b+1 (the last slot holds those that
        // "would continue" past the depth cutoff: shaded, not traced)
        if (count_hits) h += counters[(size_t)(b + 1) * PT_NSUB * PT_CSTRIDE + q * PT_CSTRIDE];
    }
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_xor(r, off);
        s += __shfl_xor(s, off);
        h += __shfl_xor(h, off);
    }
    if (q == 0) { // several batch sets finish concurrently
        atomicAdd(&totals[0], r);
        atomicAdd(&totals[1], s);
        atomicAdd(&totals[3], h);
    }
}

// per leaf triangle: the geometric normal of __closesthit__radiance (:491) and the mesh, for k_shade
__global__ void k_shade_normals(const LeafTri* __restrict__ tris, uint32_t n, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LeafTri tri = tris[i];
    const v3 v0 = mk3(tri.t0.x, tri.t0.y, tri.t0.z), v1 = mk3(tri.t0.w, tri.t1.x, tri.t1.y), v2 = mk3(tri.t1.z, tri.t1.w, tri.t2.x);
    const v3 N = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
    out[i] = make_float4(N.x, N.y, N.z, tri.t2.z);
}
// pt_trace (the query entry point): closest-hit records hold leaf-triangle indices, the caller is given primitive indices
__global__ void k_hits_to_prims(float2* __restrict__ hit, const LeafTri* __restrict__ tris, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t leaf = __float_as_int(hit[i].y);
    if (leaf >= 0) hit[i].y = tris[leaf].t2.y;
}

// ------------------------------------------------------------------ probe CDF on the GPU (Probe.h:29-77)
// BuildCDF (Probe.h:29-77) on the GPU with the reference's sequential float accumulation order, so the arrays equal the host
// loop bit for bit (a parallel scan would reassociate the sums).  One wave per row: 64 pixels are loaded coalesced and their
// weights parked in LDS; every lane then runs the same left-to-right chain of 64 adds (broadcast LDS reads) and keeps the
// partial sum of its own column — the dependent chain is the only serial part, and a 4096-row probe has 4096 of them in flight.
PT_DEV float seq_prefix64(const float* s_w, uint32_t lane, uint32_t count, float& carry) {
    float acc = carry, mine = 0.0f;
    for (uint32_t m = 0; m < count; ++m) {
        acc += s_w[m];
        mine = (m == lane) ? acc : mine;
    }
    carry = acc;
    return mine;
}
__global__ void __launch_bounds__(64) k_cdf_rows(const float4* __restrict__ data, int width, int height, float* __restrict__ pdfX,
                                                 float* __restrict__ cdfX, float* __restrict__ rowTotal) {
    __shared__ float s_w[64];
    const int j = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const size_t row = (size_t)j * width;
    float totalWeightX = 0.0f;
    for (int i0 = 0; i0 < width; i0 += 64) {
        const uint32_t count = (uint32_t)min(64, width - i0);
        float weight = 0.0f;
        if (lane < count) {
            const float4 c = data[row + i0 + lane];
            weight = c.x * 0.3f + c.y * 0.6f + c.z * 0.1f; // Luminance, maths.h:165-168
        }
        s_w[lane] = weight;
        __syncthreads();
        const float mine = seq_prefix64(s_w, lane, count, totalWeightX);
        __syncthreads();
        if (lane < count) {
            pdfX[row + i0 + lane] = weight;
            cdfX[row + i0 + lane] = mine;
        }
    }
    const float invTotalWeightX = 1.0f / totalWeightX;
    for (int i = (int)lane; i < width; i += 64) {
        pdfX[row + i] *= invTotalWeightX;
        cdfX[row + i] *= invTotalWeightX;
    }
    if (lane == 0) rowTotal[j] = totalWeightX;
}
__global__ void __launch_bounds__(64) k_cdf_marginal(const float* __restrict__ rowTotal, int height, float* __restrict__ pdfY, float* __restrict__ cdfY) {
    __shared__ float s_w[64];
    if (blockIdx.x) return;
    const uint32_t lane = threadIdx.x;
    float totalWeightY = 0.0f;
    for (int j0 = 0; j0 < height; j0 += 64) {
        const uint32_t count = (uint32_t)min(64, height - j0);
        const float weight = lane < count ? rowTotal[j0 + lane] : 0.0f;
        s_w[lane] = weight;
        __syncthreads();
        const float mine = seq_prefix64(s_w, lane, count, totalWeightY);
        __syncthreads();
        if (lane < count) {
            pdfY[j0 + lane] = weight;
            cdfY[j0 + lane] = mine;
        }
    }
    for (int j = (int)lane; j < height; j += 64) {
        cdfY[j] /= totalWeightY;
        pdfY[j] /= totalWeightY;
    }
}

// ProbeSample's column tables (pt_device.h ProbeLine): one thread per line gathers six columns' cdf values and (rgb, pdfX) texels
__global__ void k_probe_lines(const float4* __restrict__ data, const float* __restrict__ pdfX, const float* __restrict__ cdfX, int rows, int width,
                              int lpr, ProbeLine* __restrict__ lines) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * lpr) return;
    const int r = i / lpr, l = i - r * lpr;
    ProbeLine out;
    for (int c = 0; c < PT_LINE_COLS; ++c) {
        const int col = l * PT_LINE_COLS + c;
        const size_t at = (size_t)r * width + col;
        const bool in = col < width;
        const float4 d = in ? data[at] : make_float4(0.f, 0.f, 0.f, 0.f);
        out.cdf[c] = in ? cdfX[at] : INFINITY;
        out.px[c] = make_float4(d.x, d.y, d.z, in ? pdfX[at] : 0.0f);
    }
    out.pad_[0] = out.pad_[1] = 0u;
    lines[i] = out;
}
// guide of the column search (pt_device.h probe_lower_bound_lines): entry (row, k), k = 0..gk, = the number of lines whose last cdf entry is < k/gk
__global__ void k_probe_guide(const ProbeLine* __restrict__ lines, int rows, int lpr, int gk, int gpitch, uint16_t* __restrict__ guide) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * gpitch) return;
    const int r = i / gpitch, k = i - r * gpitch;
    const ProbeLine* row = lines + (size_t)r * lpr;
    const float v = (float)(k < gk ? k : gk) / (float)gk; // exact: gk is a power of two
    int lo = 0, hi = lpr; // number of entries < v (NaN rows: every comparison is false -> 0, like the searches)
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (row[mid].cdf[PT_LINE_COLS - 1] < v) lo = mid + 1; else hi = mid;
    }
    guide[i] = (uint16_t)lo;
}
__global__ void k_probe_coarse(const float* __restrict__ cdf, int rows, int n, int stride, int row_pitch, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * row_pitch) return;
    const int r = i / row_pitch, k = i - r * row_pitch;
    out[i] = (k < n / stride) ? cdf[(size_t)r * n + (size_t)(k + 1) * stride - 1] : INFINITY;
}

// ------------------------------------------------------------------ function tables (tests)
template <int MODE>
__global__ void k_table_bsdf(pt_material mat, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = &in[11 * (size_t)i];
    const v3 N = mk3(a[0], a[1], a[2]), V = mk3(a[3], a[4], a[5]), L = mk3(a[6], a[7], a[8]);
    const v3 albedo = mk3(mat.color[0], mat.color[1], mat.color[2]);
    const v3 f = bsdf_eval<MODE>(mat, albedo, a[9], a[10], N, V, L);
    out[4 * (size_t)i + 0] = f.x;
    out[4 * (size_t)i + 1] = f.y;
    out[4 * (size_t)i + 2] = f.z;
    out[4 * (size_t)i + 3] = bsdf_pdf<MODE>(mat, a[9], a[10], N, V, L);
}
template <int MODE>
__global__ void k_table_sample(pt_material mat, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = &in[9 * (size_t)i];
    const v3 N = mk3(a[0], a[1], a[2]), V = mk3(a[3], a[4], a[5]);
    Rng r;
    r.init(__float_as_uint(a[8]));
    v3 u, v, L = mk3(0.f);
    float pdf;
    basis_from_vector(N, u, v);
    bsdf_sample<MODE>(mat, a[6], a[7], u, v, N, V, L, pdf, r);
    float* o = &out[6 * (size_t)i];
    o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = pdf;
    o[4] = __uint_as_float(r.seed1);
    o[5] = __uint_as_float(r.seed2);
}
__global__ void k_table_probe_sample(DevProbe probe, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r;
    r.init(__float_as_uint(in[i]));
    v3 dir, color;
    float pdf;
    probe_sample(probe, probe_marg_global(probe), dir, color, pdf, r);
    float* o = &out[9 * (size_t)i];
    o[0] = dir.x; o[1] = dir.y; o[2] = dir.z;
    o[3] = color.x; o[4] = color.y; o[5] = color.z;
    o[6] = pdf;
    o[7] = __uint_as_float(r.seed1);
    o[8] = __uint_as_float(r.seed2);
}
__global__ void k_table_probe_eval(DevProbe probe, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float u, v;
    probe_dir_to_uv(mk3(in[3 * (size_t)i], in[3 * (size_t)i + 1], in[3 * (size_t)i + 2]), u, v);
    const float4 c = probe_eval(probe, u, v);
    float* o = &out[6 * (size_t)i];
    o[0] = u; o[1] = v; o[2] = c.x; o[3] = c.y; o[4] = c.z; o[5] = c.w;
}
__global__ void k_table_probe_pdf(DevProbe probe, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = probe_pdf(probe, mk3(in[3 * (size_t)i], in[3 * (size_t)i + 1], in[3 * (size_t)i + 2]));
}
__global__ void k_table_color(const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __uint_as_float(make_color(mk3(in[3 * (size_t)i], in[3 * (size_t)i + 1], in[3 * (size_t)i + 2])));
}
__global__ void k_table_math(const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int fn = (int)in[3 * (size_t)i];
    const float x = in[3 * (size_t)i + 1], y = in[3 * (size_t)i + 2];
    float r;
    switch (fn) {
        case 0: r = pt_sinf(x); break;
        case 1: r = pt_cosf(x); break;
        case 2: r = pt_acosf(x); break;
        case 3: r = pt_atan2f(x, y); break;
        case 4: r = pt_logf(x); break;
        case 5: r = pt_powf(x, y); break;
        case 6: r = x / y; break;
        default: r = sqrtf(x); break;
    }
    out[i] = r;
}
__global__ void k_table_tex(DevTex tex, const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = tex2d_wrap_linear(tex, in[2 * (size_t)i], in[2 * (size_t)i + 1]);
    out[4 * (size_t)i] = c.x; out[4 * (size_t)i + 1] = c.y; out[4 * (size_t)i + 2] = c.z; out[4 * (size_t)i + 3] = c.w;
}
__global__ void k_emit_uvs(const float* __restrict__ texcoord, const uint32_t* __restrict__ idx, uint32_t ntri, PrimUV* __restrict__ uvs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ntri) return;
    const uint32_t a = idx[3 * (size_t)p], b = idx[3 * (size_t)p + 1], c = idx[3 * (size_t)p + 2];
    PrimUV u;
    u.c0 = make_float2(texcoord[2 * (size_t)a], texcoord[2 * (size_t)a + 1]);
    u.c1 = make_float2(texcoord[2 * (size_t)b], texcoord[2 * (size_t)b + 1]);
    u.c2 = make_float2(texcoord[2 * (size_t)c], texcoord[2 * (size_t)c + 1]);
    uvs[p] = u;
}
__global__ void k_emit_textris(const LeafTri* __restrict__ tris, const PrimUV* __restrict__ uvs, uint32_t n, TexTri* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LeafTri t = tris[i];
    const PrimUV u = uvs[__float_as_int(t.t2.y)];
    TexTri o;
    o.a = t.t0;
    o.b = t.t1;
    o.c = make_float4(t.t2.x, u.c0.x, u.c0.y, u.c1.x);
    o.d = make_float4(u.c1.y, u.c2.x, u.c2.y, 0.f);
    out[i] = o;
}
__global__ void k_table_rng(const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = __float_as_uint(in[2 * (size_t)i]), b = __float_as_uint(in[2 * (size_t)i + 1]);
    float* o = &out[8 * (size_t)i];
    uint32_t s = tea4(a, b);
    o[0] = __uint_as_float(s);
    uint32_t l = s;
    o[1] = rnd(l);
    o[2] = __uint_as_float(l);
    Rng r;
    r.init(s);
    o[3] = r.randf();
    o[4] = r.randf();
    o[5] = __uint_as_float(r.seed1);
    o[6] = __uint_as_float(r.seed2);
    o[7] = __uint_as_float(r.rand());
}
