// pt_bvh8.h — 8-wide compressed BVH ("CWBVH", after Ylitie, Karras, Laine 2017) and its persistent-wave
// traversal kernel for gfx950.  Why: with the binary tree the traversal kernel is bound by the per-CU
// texture-address/L1 path — every lane's 16-byte load of a different cache line costs that path a cycle,
// and a ray touches ≈205 such loads (≈40 nodes x 4 + ≈15 triangles x 3, rocprofv3 PMC: 9.2 TB/s of L1
// requests against a ≈9.8 TB/s ceiling for fully divergent 16-byte loads).  An 80-byte node holding 8
// quantised child boxes needs 5 loads and a third of the tree depth.
//
// Node (80 B = 5 x 16 B):
//   n0: origin.xyz (f32)                       | upper halves (sign, exponent, 7 mantissa bits) of the grid steps sx, sy
//   n1: child_base | tri_base | leafbits | upper half of sz, imask << 16
//       leafbits bit 3*s+k: slot s is a leaf child holding more than k triangles
//   n2: qlo.x[8] (2 dwords) | qlo.y[8]   n3: qlo.z[8] | qhi.x[8]   n4: qhi.y[8] | qhi.z[8]
// Child box s = origin + q * step per axis (step = extent/255 rounded up to 8 significant bits), rounded outward at build from the padded float box, so
// the box test stays conservative; empty slots hold an inverted box (qlo=255, qhi=0).  Internal children
// are contiguous (child_base + rank of the slot among the set bits of imask), the triangles of all leaf
// children are contiguous from tri_base in (slot, k) order, i.e. triangle (s,k) is tri_base + rank of bit 3*s+k in leafbits.  Children sit in the slot whose octant (sign bits of
// centroid - node centre) best matches them (greedy assignment at build), so visiting hit slots in
// increasing (slot XOR ray-octant) order is an approximate front-to-back order without sorting; the next slot in
// that order is found by three mask-and-select steps (z half, y quarter, x slot on the side the ray enters first).
// The triangle test is pt_bvh.h's tri_test, bit-identical to the CPU checker; closest hit + lowest
// primitive tie-break make the result independent of the tree.
// The closest-hit record is (t, index of the leaf triangle): k_shade re-reads those 48 bytes (vertices, primitive, mesh).  A lane
// therefore tracks the leaf index of its best hit beside the primitive id (which decides exact ties in t).  In the stealing phase
// the shared 64-bit key carries the primitive (lexicographic minimum = the tie-break rule) and the leaf index of the key's holder
// sits beside the key in LDS (s_leaf).
#pragma once
#ifdef PT_BVH8_NODE_ONLY
#include "pt_bvh.h"
#else
#include "pt_kernels.h"
#endif
#include "pt_host.h" // Grid8

// The documented 80-byte node (above): what pt_export_bvh hands out whatever the kernels traverse
struct Node80 {
    float4 n0, n1, n2, n3, n4;
};
// One-line node (PT8_NODE64, round 5): the same eight quantised child boxes in 64 bytes, 64-byte aligned — four
// 16-byte loads from ONE line instead of five from a record that straddles two 128-byte lines four times in eight; the node array
// shrinks by 20 %.  What pays for it: the grid origin is no longer three floats but three integers on a grid over the scene's (padded)
// bounds (14 / 13 / 14 bits, rounded DOWN so the box only grows, by less than 1 / 8192 of the scene per axis), the grid steps are powers
// of two (a 5-bit exponent per axis above the tree's base exponent: up to twice as coarse as the 80-byte node's 8-significant-bit
// steps), child and triangle bases have 24 bits (16.7 M nodes / leaf triangles), and the triangle counts of the leaf slots are two bit
// planes (count = b0 + 2 b1) instead of three unary bits per slot.
//   h.x: child_base | imask << 24
//   h.y: tri_base | ex << 24 | oy[0:3] << 29
//   h.z: b0 | b1 << 8 | ey << 16 | ez << 21 | oy[3:9] << 26
//   h.w: ox | oz << 14 | oy[9:13] << 28
//   q0, q1, q2: qlo.x[8] qlo.y[8] | qlo.z[8] qhi.x[8] | qhi.y[8] qhi.z[8] — the 80-byte node's n2, n3, n4
// origin.a = fma(o.a, gstep.a, glo.a), step.a = 2^(ebase + e.a - 127): Bvh8Dev carries glo, gstep, ebase.
struct Node64 {
    uint4 h, q0, q1, q2;
};
#ifndef PT8_NODE64
#define PT8_NODE64 0
#endif
#if PT8_NODE64
struct Node8 : Node64 {};
#else
struct Node8 : Node80 {};
#endif
#define PT8_GRID_BITS_XZ 14
#define PT8_GRID_BITS_Y 13
#ifndef PT8_LEAF_MAX
#define PT8_LEAF_MAX 3
#endif
static_assert(PT8_LEAF_MAX >= 1 && PT8_LEAF_MAX <= 3, "leafbits holds 3 bits per slot");

PT_DEV float u8f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xffu); }

struct Bvh8Dev {
    const Node8* nodes;
    const LeafTri* tris;
    float hit_pad; // half the builder's box padding (pt_bvh.h tri_test_det)
#if PT8_NODE64
    Grid8 grid;
#endif
};

// What a traversal step needs of a node's first 16 / 32 bytes, in either layout
struct NodeHdr {
    float ox, oy, oz, sx, sy, sz;
    uint32_t imask, child_base, tri_base;
    uint32_t lbits; // triangle counts of the leaf slots: 80-byte node: bit 3 s + k = slot s holds more than k; one-line node: b0 | b1 << 8
};
#if PT8_NODE64
PT_DEV NodeHdr node_hdr(const uint4 h, const Grid8& b) {
    NodeHdr n;
    n.child_base = h.x & 0xffffffu;
    n.imask = h.x >> 24;
    n.tri_base = h.y & 0xffffffu;
    n.lbits = h.z & 0xffffu;
    n.sx = __uint_as_float((b.ebase + ((h.y >> 24) & 31u)) << 23);
    n.sy = __uint_as_float((b.ebase + ((h.z >> 16) & 31u)) << 23);
    n.sz = __uint_as_float((b.ebase + ((h.z >> 21) & 31u)) << 23);
    const uint32_t oyq = (h.y >> 29) | ((h.z >> 26) << 3) | ((h.w >> 28) << 9);
    n.ox = __builtin_fmaf((float)(h.w & 0x3fffu), b.gstep[0], b.glo[0]);
    n.oy = __builtin_fmaf((float)oyq, b.gstep[1], b.glo[1]);
    n.oz = __builtin_fmaf((float)((h.w >> 14) & 0x3fffu), b.gstep[2], b.glo[2]);
    return n;
}
// pending (slot, k) pairs of the leaf slots in `hm`: bit 8 k + s
PT_DEV uint32_t leaf_pending(uint32_t hm, uint32_t lbits) {
    const uint32_t b0 = lbits & 0xffu, b1 = lbits >> 8;
    return ((b0 | b1) & hm) | ((b1 & hm) << 8) | ((b0 & b1 & hm) << 16);
}
PT_DEV uint32_t leaf_first(uint32_t tri_base, uint32_t lbits, uint32_t s) { // index of the first triangle of leaf slot s
    const uint32_t low = (1u << s) - 1u;
    return tri_base + (uint32_t)__popc(lbits & low) + 2u * (uint32_t)__popc((lbits >> 8) & low);
}
PT_DEV uint32_t leaf_of(uint32_t tri_base, uint32_t lbits, uint32_t bit) { return leaf_first(tri_base, lbits, bit & 7u) + (bit >> 3); }
PT_DEV uint32_t leaf_count(uint32_t lbits, uint32_t s) { return ((lbits >> s) & 1u) + 2u * ((lbits >> (8u + s)) & 1u); }
#else
PT_DEV NodeHdr node_hdr(const float4 n0, const float4 n1) {
    NodeHdr n;
    const uint32_t e01 = __float_as_uint(n0.w), e2m = __float_as_uint(n1.w);
    n.ox = n0.x; n.oy = n0.y; n.oz = n0.z;
    n.sx = __uint_as_float(e01 << 16); n.sy = __uint_as_float(e01 & 0xffff0000u); n.sz = __uint_as_float(e2m << 16);
    n.imask = e2m >> 16;
    n.child_base = __float_as_uint(n1.x);
    n.tri_base = __float_as_uint(n1.y);
    n.lbits = __float_as_uint(n1.z);
    return n;
}
// pending (slot, k) pairs of the leaf slots in `hm`: bit 3 s + k
PT_DEV uint32_t leaf_pending(uint32_t hm, uint32_t lbits) {
    uint32_t sp3 = (hm | (hm << 8)) & 0x00F00Fu; // every hit bit s -> bits 3s..3s+2, masked by the node's leafbits
    sp3 = (sp3 | (sp3 << 4)) & 0x0C30C3u;
    sp3 = (sp3 | (sp3 << 2)) & 0x249249u;
    return (sp3 * 7u) & lbits;
}
PT_DEV uint32_t leaf_of(uint32_t tri_base, uint32_t lbits, uint32_t bit) { return tri_base + (uint32_t)__popc(lbits & ((1u << bit) - 1u)); }
PT_DEV uint32_t leaf_first(uint32_t tri_base, uint32_t lbits, uint32_t s) { return leaf_of(tri_base, lbits, 3u * s); }
PT_DEV uint32_t leaf_count(uint32_t lbits, uint32_t s) { return (uint32_t)__popc((lbits >> (3u * s)) & 7u); }
#endif

#if PT8_NODE64
// pt_export_bvh / PT_DEBUG_BVH: one-line nodes as the documented 80-byte records — the same origin and (power-of-two) steps as floats,
// the same planes, the leaf counts as three unary bits per slot: the tree the kernels traverse, box for box
static __global__ void k_nodes_to80(const Node64* __restrict__ in, uint32_t n, Grid8 b, Node80* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Node64 nd = in[i];
    const NodeHdr h = node_hdr(nd.h, b);
    uint32_t leafbits = 0u;
    for (uint32_t s = 0; s < 8u; ++s) leafbits |= ((1u << leaf_count(h.lbits, s)) - 1u) << (3u * s);
    Node80 o;
    o.n0 = make_float4(h.ox, h.oy, h.oz, __uint_as_float((__float_as_uint(h.sx) >> 16) | (__float_as_uint(h.sy) & 0xffff0000u)));
    o.n1 = make_float4(__uint_as_float(h.child_base), __uint_as_float(h.tri_base), __uint_as_float(leafbits), __uint_as_float((__float_as_uint(h.sz) >> 16) | (h.imask << 16)));
    o.n2 = make_float4(__uint_as_float(nd.q0.x), __uint_as_float(nd.q0.y), __uint_as_float(nd.q0.z), __uint_as_float(nd.q0.w));
    o.n3 = make_float4(__uint_as_float(nd.q1.x), __uint_as_float(nd.q1.y), __uint_as_float(nd.q1.z), __uint_as_float(nd.q1.w));
    o.n4 = make_float4(__uint_as_float(nd.q2.x), __uint_as_float(nd.q2.y), __uint_as_float(nd.q2.z), __uint_as_float(nd.q2.w));
    out[i] = o;
}
#endif

#ifndef PT_BVH8_NODE_ONLY
#ifndef PT8_LDS_DEPTH
#define PT8_LDS_DEPTH 10 // levels of the traversal stack kept in LDS (5 KB per wave at 5 waves per SIMD = 100 KB per CU: room is left for k_shade workgroups of the other chunk streams; 12: +2 % frame time, 8: +1 %)
#endif
#define PT8_OVF_DEPTH 52
#ifndef PT8_REFILL
#define PT8_REFILL 40 // refill when fewer than this many lanes hold a ray (dense, one-step refill loads since the state travels with the queue: 24 → 40 −0.7 % frame; 16: +2.6 %, 56: +3 %)
#endif
#ifndef PT8_TRI_BIAS
#define PT8_TRI_BIAS 2 // a triangle step runs when tri-waiting lanes * bias > node-waiting lanes
#endif
#ifndef PT8_MIN_CHUNK
#define PT8_MIN_CHUNK 64
#endif
#ifndef PT8_INTERLEAVE
#define PT8_INTERLEAVE 0 // > 0: launches of one static chunk per wave and at least this many rays deal their rays to the waves round robin
#endif
#ifndef PT8_WIDE_MIN
#define PT8_WIDE_MIN 64 // rays per wave at least in a launch smaller than the grid (64: one full chunk per wave, as before)
#endif
#define PT8_CHUNK 512
#ifndef PT8_WAVES_PER_EU
#define PT8_WAVES_PER_EU 5
#endif
// In-wave work stealing (tail of a launch).  Once a wave has taken its last chunk of the queue, a lane that runs dry no
// longer idles until the wave's longest ray is done: it takes over the BOTTOM entry of a busy lane's traversal stack (the
// far end of that ray: the not yet visited children of the node closest to the root) together with a copy of the ray, and
// traverses it as a co-worker.  Co-workers of a ray share one record in LDS — the best (t, primitive) found so far as a
// 64-bit key merged with an atomic minimum (closest hit = lexicographic minimum, exactly the kernel's tie-break rule; a
// shadow ray's key drops to 0 when any co-worker finds an occluder) and a count of workers; whoever brings the count to 0
// writes the ray's result.  The answer is the minimum over the same set of accepted triangles whatever the split, so
// images do not change by a bit; what changes is that a 500-step ray that used to hold its wave (and the launch: 9 launches
// per frame, each as long as its longest ray) for 500 iterations is traversed by up to 64 lanes.
#ifndef PT8_STEAL
#define PT8_STEAL 1
#endif
// (a lane may be robbed from its first step: requiring 24 / 64 prior steps of the victim cost +13 % / +26 % frame time at a 1/8 share)
#ifndef PT8_DEFER_WRITE
#define PT8_DEFER_WRITE 1
#endif
#ifndef PT8_STEAL_PERIOD
#define PT8_STEAL_PERIOD 3 // traversal iterations between two steal rounds while lanes are idle
#endif

// The stealing phase is where a small launch spends its whole life, one wave per SIMD, and there every stall is paid in full (per-wave cycle
// log, profiles/r5_15_tail_cycles.md: of 4500 cycles per iteration only 650 wait for the step's own loads).
// PT8_STEAL_WSYNC: the lanes of a steal round talk through LDS only and the workgroup IS one wave, whose LDS operations execute in order — a
//   compiler-level barrier orders them; __syncthreads() also drains every outstanding global load and store (vmcnt(0)), twice per round:
//   1/8 share 1.803 -> 1.781 ms, full frame 7.881 -> 7.844.
// (Measured and rejected with it: writing back the rays that finish in the stealing phase together, at the top of the outer loop, like those
// that finish before it, instead of one by one where they finish: 1/8 share 1.78 -> 1.90 ms, full frame 7.84 -> 8.16.)
#ifndef PT8_STEAL_WSYNC
#define PT8_STEAL_WSYNC 1
#endif
#if PT8_STEAL_WSYNC
#define PT8_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define PT8_WAVE_SYNC() __syncthreads()
#endif

#define PT_WAVELOG_CAP (1u << 20)
#ifdef PT_DEBUG_STATS
#define PT_STAT(x) x
#else
#define PT_STAT(x)
#endif
// PT_DEBUG_WAVELOG builds (tools/r5_wavelog.sh): every traversal wave logs its start, the start of its stealing phase, its end and its loop
// iterations — one atomic per WAVE, so launch timing stays what it is (PT_DEBUG_STATS pays same-address atomics per ray: 10 x the frame time)
#if defined(PT_DEBUG_STATS) || defined(PT_DEBUG_WAVELOG)
#define PT_WLOG(x) x
#else
#define PT_WLOG(x)
#endif

#define PT_WSTRIDE 16 // words between the chunk counters of two launches (a line each)
struct Trace8Args {
    PathState st;
    Bvh8Dev bvh;
    QView queue;
    QView queue2; // TR_UNIFIED only: the shadow queue of the previous bounce, traced in the same launch
    uint32_t* work;
    uint32_t* ovf; // spill stack: [PT8_OVF_DEPTH][2][gridDim.x * 64]
    int cull_back; // shadow rays ignore back-facing triangles (sv3/sv4 occlusion ray flag)
    unsigned long long* dbg; // optional, 64 words: [0] node steps, [1] triangle tests, [2] max stack depth, [3] pushes, ...; [16+k] closest-hit rays and
                             // [32+k] shadow rays that took 2^k..2^(k+1)-1 steps (PT_DEBUG_STATS builds + PT_DEBUG_COUNTS)
    int bounce;   // shadow modes with asynchronous shadow records (st.vis != null): the bounce whose records this launch traces
    int lds_skip; // test hook (PT_STACK_LDS_SKIP): keep this many fewer stack levels in LDS, so shallow trees exercise the global spill path
    int ovf_depth;  // spill levels available (PT8_OVF_DEPTH; the test hook PT_STACK_CAP lowers it)
    uint32_t* fault; // device word: bit 0 set when a push found the stack full — the host turns it into PT_ERR_UNSUPPORTED
    uint32_t num_nodes;
#if PT_DEBUG_WAVELOG + 0 == 3
    unsigned long long* hist; // dbg + 53 (set from dbg by the kernels' entry points; survives the fused loop's `dbg = nullptr`)
#endif
};

// The traversal of one persistent wave.  LOCAL = false: wave `wid` of `nw` of a launch over the queues of `a` (k_trace8 below).
// LOCAL = true (pt_fused.h): the wave traverses a private window of the queue arrays — entries [woff, woff + ln1) of `a.queue`'s arrays and, in
// TR_UNIFIED, the shadow entries [woff, woff + ln2) of `a.queue2`'s — on its own: no sub-queue prefix, no chunk counter, every ray taken by
// this wave (its spare lanes steal from the first iteration it has fewer rays than lanes).  Spill stack and fault word are the launch's.
template <int MODE, bool LOCAL>
PT_DEV void trace8_wave(const Trace8Args& a, const uint32_t wid, const uint32_t nw, const uint32_t ln1, const uint32_t ln2, const uint32_t woff) {
    __shared__ uint32_t s_stack[PT8_LDS_DEPTH * 2 * 64];
    __shared__ uint32_t s_prefix[PT_NSUB + 1];
    __shared__ uint32_t s_prefix2[PT_NSUB + 1];
#if PT8_STEAL
    __shared__ unsigned long long s_key[64]; // per owner lane: merged result of the ray that lane loaded
    __shared__ uint32_t s_cnt[64];           // per owner lane: co-workers still traversing that ray
    __shared__ int32_t s_leaf[64];           // per owner lane: leaf triangle of the hit s_key holds (written by whoever lowered the key — one wave per
                                             // workgroup, so after every lane's atomic minimum exactly the lanes whose key IS the minimum write it)
    __shared__ uint32_t s_vlane[64];         // steal round: victim lane by rank
#endif
    const uint32_t lane = threadIdx.x;
    const uint32_t gtid = blockIdx.x * 64u + lane;
    const uint32_t gstride = gridDim.x * 64u;
    const int lds_depth = PT8_LDS_DEPTH - a.lds_skip;
    // index space of the launch: TR_UNIFIED: [0,n2) = shadow rays of `queue2`, [n2, n) = rays of `queue`; else [0,n1) = rays of `queue`
    const uint32_t n1 = LOCAL ? ln1 : qreader_init(a.queue, s_prefix);
    const uint32_t n2 = (MODE == TR_UNIFIED) ? (LOCAL ? ln2 : qreader_init(a.queue2, s_prefix2)) : 0u;
    const uint32_t n = n1 + n2;
    uint32_t chunk = (n / (nw * 2u)) & ~(uint32_t)(PT8_MIN_CHUNK < 64 ? PT8_MIN_CHUNK - 1 : 63);
    chunk = chunk < (uint32_t)PT8_MIN_CHUNK ? (uint32_t)PT8_MIN_CHUNK : (chunk > (uint32_t)PT8_CHUNK ? (uint32_t)PT8_CHUNK : chunk);
#if PT8_WIDE_MIN < 64
    // Wide start (round 5): a launch with fewer rays than the grid has lanes is not throughput-bound but as long as its slowest ray's chain of
    // dependent steps (1.5 us each with a lone wave on a SIMD).  Such a launch is spread over more waves — ceil(n / waves) rays each, at least
    // PT8_WIDE_MIN — whose spare lanes take stack entries of the wave's rays from the first iteration on (the stealing phase below), so a
    // ray's subtrees are traversed side by side.  The split is static: no chunk counter is touched.
    if (n < nw * 64u) {
        chunk = ((n + nw - 1u) / nw + 15u) & ~15u;
        chunk = chunk < (uint32_t)PT8_WIDE_MIN ? (uint32_t)PT8_WIDE_MIN : chunk;
    }
#endif
    if (LOCAL) chunk = n; // one chunk: this wave's
    const bool no_share = (unsigned long long)wid * chunk >= n; // this wave has no chunk of its own
    const uint32_t nchunks = (n + chunk - 1u) / chunk;
    if (no_share) return;
    uint32_t chunk_next = wid * chunk;
    uint32_t chunk_end = (chunk_next + chunk < n) ? chunk_next + chunk : n;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool shadow_lane = (MODE == TR_SHADOW_APPLY); // TR_UNIFIED: per lane, set at refill

    bool active = false, exhausted = false;
    RaySetup r;
    r.o = r.d = r.idir = r.dn = mk3(0.f);
    float tmin = 0.f, tmax = 0.f, best = 0.f;
    int32_t bprim = -1;                        // closest-hit lanes: primitive of the best hit; shadow lanes: 1 = occluded
    int32_t bleaf = -1;                        // closest-hit lanes: leaf triangle of the best hit (before the stealing phase; then s_leaf)
    uint32_t pm = 0;                           // near-side slot masks of the ray's octant: z half | y quarters << 8 | x slots << 16
    uint32_t g_base = 0, g_imask = 0, g_hits = 0; // current node group: children still to visit (slot positions)
    uint32_t t_base = 0, t_mask = 0, t_bits = 0; // current triangle group: pending bits of the node's leafbits
    int sp = 0;
    bool stealing = false;   // wave-uniform: the queue is exhausted and the shared records are in use
    // four small per-lane fields share one register (the kernel sits exactly at the 96 VGPRs of 5 waves per SIMD):
    //   owner (bits 0-7): the lane whose shared record this lane's ray belongs to | hint1, hint2 (8-15, 16-23): sub-queue of the lane's last
    //   queue lookups | sb (24-31): stack bottom — levels [sb, sp) are live, entries below sb were taken by co-workers
    uint32_t misc = lane;
#define OWNER (misc & 0xffu)
#define SB ((int)(misc >> 24))
#define SET_OWNER(v) (misc = (misc & ~0xffu) | (uint32_t)(v))
#define SET_SB(v) (misc = (misc & 0x00ffffffu) | ((uint32_t)(v) << 24))
    uint32_t slot = 0; // position of the lane's ray in its queue's arrays (PathState)
#if PT8_DEFER_WRITE
    bool unwritten = false; // this lane holds a finished ray whose result is not written yet
#endif
    PT_STAT(uint32_t c_nodes = 0; uint32_t c_tris = 0; uint32_t c_maxsp = 0; uint32_t c_push = 0; uint32_t c_ray = 0; uint32_t c_raymax = 0; uint32_t c_iters = 0;
            uint32_t c_act = 0; uint32_t c_exec = 0; uint32_t c_nodeit = 0;)
    PT_WLOG(const unsigned long long w_t0 = wall_clock64(); unsigned long long w_tex = 0; uint32_t w_iters = 0;)
    PT_WLOG(unsigned long long w_cw = 0; unsigned long long w_cr = 0; unsigned long long w_cs = 0; uint32_t w_np = 0;) // -DPT_DEBUG_WAVELOG=2: cycles in the loop-top write-back | refill | steal round, outer passes
    PT_WLOG(unsigned long long w_c01 = 0; unsigned long long w_c12 = 0; unsigned long long w_c23 = 0; const unsigned long long w_c0 = clock64();) // shader-clock cycles: vote + pop + addresses | waiting for the loads | arithmetic
#if PT_DEBUG_WAVELOG + 0 == 3
    // round 6 (sizing a several-lanes-per-ray mode): iterations and cycles of this wave by the number of lanes that held a ray (or a stolen share of one)
    // at the top of the iteration: <= 8 | <= 16 | <= 32 | more — summed over all waves and launches of a frame in dbg[53..60]
    unsigned long long h_i0 = 0, h_i1 = 0, h_i2 = 0, h_i3 = 0, h_c0 = 0, h_c1 = 0, h_c2 = 0, h_c3 = 0;
#endif

    auto push = [&](uint32_t v0, uint32_t v1) {
        PT_STAT(++c_push; if ((uint32_t)sp + 1 > c_maxsp) c_maxsp = sp + 1;)
        if (sp < lds_depth) {
            s_stack[(sp * 2) * 64 + lane] = v0;
            s_stack[(sp * 2 + 1) * 64 + lane] = v1;
        } else if (sp < lds_depth + a.ovf_depth) {
            a.ovf[(size_t)((sp - lds_depth) * 2) * gstride + gtid] = v0;
            a.ovf[(size_t)((sp - lds_depth) * 2 + 1) * gstride + gtid] = v1;
        } else {
            // Stack full: pt_create refuses trees deeper than the stack, so this is unreachable unless that check is
            // bypassed; the entry is dropped WITHOUT advancing sp (pops stay matched, no stray read) and the launch is
            // reported as failed instead of returning a silently wrong image.
            atomicOr(a.fault, 1u);
            return;
        }
        ++sp;
    };
    auto pop = [&](uint32_t& v0, uint32_t& v1) { // caller checked sp > sb
        --sp;
        if (sp < lds_depth) {
            v0 = s_stack[(sp * 2) * 64 + lane];
            v1 = s_stack[(sp * 2 + 1) * 64 + lane];
        } else {
            v0 = a.ovf[(size_t)((sp - lds_depth) * 2) * gstride + gtid];
            v1 = a.ovf[(size_t)((sp - lds_depth) * 2 + 1) * gstride + gtid];
        }
    };
    // write the ray's result: (best, bprim) = closest hit / occlusion flag
    auto write_result = [&](float rbest, int32_t rprim) {
        if (MODE == TR_SHADOW_APPLY || (MODE == TR_UNIFIED && shadow_lane)) {
            const bool occluded = rprim != 0;
            const uint32_t ps = (MODE == TR_UNIFIED ? a.queue2 : a.queue).base[slot]; // the path slot of this shadow-queue entry
            if (a.st.vis) { // asynchronous shadow records: only publish the visibility, k_resolve sums in bounce order
                if (!occluded) atomicOr(&a.st.vis[ps], 1u << a.bounce);
                return;
            }
            const float4 pe = st_ld<PT_NT_TRACE_LD>(&a.st.shPend[slot]);
            const int kind = __float_as_int(pe.w);
            if (kind == PEND_ALPHA) {
                if (occluded) {
                    const float4 x = a.st.alpha[ps];
                    a.st.alpha[ps] = make_float4(x.x + pe.x, x.y + pe.y, x.z + pe.z, 0.f);
                }
            } else if (!occluded) {
                float4* acc = (kind == PEND_DIRECT) ? a.st.direct : a.st.indirect;
                const float4 x = st_ld<PT_NT_TRACE_LD>(&acc[ps]);
                st_st<PT_NT_TRACE_ST>(&acc[ps], make_float4(x.x + pe.x, x.y + pe.y, x.z + pe.z, 0.f));
            }
        } else {
            st_st<PT_NT_TRACE_ST>(&a.st.hit[slot], make_float2(rbest, __int_as_float(rprim)));
        }
    };
    const auto is_shadow = [&]() { return MODE == TR_SHADOW_APPLY || MODE == TR_ANY_QUERY || (MODE == TR_UNIFIED && shadow_lane); };
#if PT8_STEAL
    // 64-bit result key: closest hit = (bits of t) << 32 | primitive (t > 0, so the bits order like the floats; "no hit" is
    // (tmax, -1) = the largest key of the ray); shadow ray = 0 occluded / ~0 not occluded
    auto local_key = [&]() -> unsigned long long {
        if (is_shadow()) return bprim != 0 ? 0ull : ~0ull;
        return ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(uint32_t)bprim;
    };
#endif
    // this lane is done with its (share of the) ray
    auto finish = [&]() {
#if PT8_STEAL
        if (stealing) {
            atomicMin(&s_key[OWNER], local_key());
            if (atomicSub(&s_cnt[OWNER], 1u) == 1u) { // the last co-worker publishes the merged result
                const unsigned long long k = s_key[OWNER];
                if (is_shadow()) write_result(0.f, k == 0ull ? 1 : 0);
                else write_result(__uint_as_float((uint32_t)(k >> 32)), (int32_t)(uint32_t)k < 0 ? -1 : s_leaf[OWNER]);
            }
            active = false;
            return;
        }
#endif
#if PT8_DEFER_WRITE
        // Before the stealing phase the result is not written here: a finish reached by one or two lanes at a time would stall the whole
        // wave on the write-back's dependent loads (pending contribution -> accumulator).  The lane keeps (slot, best, bprim) — it is
        // idle until the next refill — and all finished lanes write together at the top of the loop.
        unwritten = true;
#else
        write_result(best, is_shadow() ? bprim : bleaf);
#endif
        active = false;
        PT_STAT(if (c_ray > c_raymax) c_raymax = c_ray;
                if (a.dbg) atomicAdd(&a.dbg[(is_shadow() ? 32 : 16) + (31 - __clz((int)(c_ray | 1u)))], 1ull);
                c_ray = 0;)
    };

    // a lane takes up a ray
    auto start_ray = [&](float ox, float oy, float oz, float dx, float dy, float dz) {
        r.o = mk3(ox, oy, oz);
        r.d = mk3(dx, dy, dz);
#ifdef PT8_IEEE_IDIR
        r.idir = mk3(1.0f / dx, 1.0f / dy, 1.0f / dz);
#else
        r.idir = mk3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
#endif
        r.dn = scl3(r.d, 1.0f / dot3(r.d, r.d));
        if (!(fabsf(dx) > 1e-30f)) r.idir.x = copysignf(1e30f, dx);
        if (!(fabsf(dy) > 1e-30f)) r.idir.y = copysignf(1e30f, dy);
        if (!(fabsf(dz) > 1e-30f)) r.idir.z = copysignf(1e30f, dz);
        // sign BITS (so that -0.0 picks the same near/far planes as its -1e30 reciprocal)
        pm = ((__float_as_uint(dz) >> 31) ? 0xF0u : 0x0Fu) | (((__float_as_uint(dy) >> 31) ? 0xCCu : 0x33u) << 8) | (((__float_as_uint(dx) >> 31) ? 0xAAu : 0x55u) << 16);
        sp = 0;
        SET_SB(0);
        t_mask = 0;
    };

    for (;;) {
        PT_WLOG(const unsigned long long w_p0 = clock64(); ++w_np;)
#if PT8_DEFER_WRITE
        if (unwritten) { // results of the rays that finished since the last pass: all finished lanes at once
            write_result(best, is_shadow() ? bprim : bleaf);
            unwritten = false;
        }
#endif
        PT_WLOG(__builtin_amdgcn_s_waitcnt(0); const unsigned long long w_p1 = clock64(); w_cw += w_p1 - w_p0;)
        // ---------------- refill idle lanes
        const unsigned long long idle = __ballot(!active);
        if (idle != 0ull && !exhausted) {
            const uint32_t cnt = (uint32_t)__popcll(idle);
            if (chunk_next == chunk_end) {
                if (nchunks <= nw) { // every chunk belongs to a wave by its block index: nothing to fetch (and no same-address atomic: 10 ns each, serialised)
                    exhausted = true;
                } else {
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(a.work, 1u);
                    c = __shfl(c, 0) + nw;
                    const unsigned long long b0 = (unsigned long long)c * chunk;
                    chunk_next = b0 < n ? (uint32_t)b0 : n;
                    chunk_end = (b0 + chunk < n) ? (uint32_t)(b0 + chunk) : n;
                    if (chunk_next >= chunk_end) exhausted = true;
                }
            }
            const uint32_t take = (chunk_end - chunk_next) < cnt ? (chunk_end - chunk_next) : cnt;
            const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
            const uint32_t first = chunk_next;
            chunk_next += take;
            if (chunk_next == chunk_end && nchunks <= nw) exhausted = true; // the wave's only chunk is taken: its spare lanes start stealing at once
#if PT8_INTERLEAVE
            // A launch of one static chunk per wave: wave w's lane l takes ray l * (waves with a share) + w instead of ray 64 w + l, so every wave
            // holds rays from all over the queue — which is in image order: a wave of 64 neighbouring camera rays that all graze the terrain has
            // no idle lane to help (max / mean wave time 2.7-3.4 in camera launches of a 1/8 share, profiles/r5_01_wavelog_w8.txt).
            const bool spread = nchunks <= nw && chunk == 64u && n >= (uint32_t)PT8_INTERLEAVE;
            const uint32_t gi_spread = lane * (nchunks < nw ? nchunks : nw) + wid;
            if (!active && (spread ? gi_spread < n : rank < take)) {
                const uint32_t gi = spread ? gi_spread : first + rank;
#else
            if (!active && rank < take) {
                const uint32_t gi = first + rank;
#endif
                if (LOCAL) {
                    shadow_lane = (MODE == TR_SHADOW_APPLY) || (MODE == TR_UNIFIED && gi < n2);
                    slot = woff + ((MODE == TR_UNIFIED && !shadow_lane) ? gi - n2 : gi);
                } else if (MODE == TR_UNIFIED) {
                    // shadow rays first: the longest rays of a launch are probe shadow rays that graze the terrain and hit nothing;
                    // started early, their tails overlap the closest-hit bulk instead of trailing it
                    shadow_lane = gi < n2;
                    uint32_t hint = shadow_lane ? (misc >> 16) & 0xffu : (misc >> 8) & 0xffu;
                    slot = shadow_lane ? qreader_pos_hint(a.queue2, s_prefix2, gi, hint) : qreader_pos_hint(a.queue, s_prefix, gi - n2, hint);
                    misc = shadow_lane ? (misc & 0xff00ffffu) | (hint << 16) : (misc & 0xffff00ffu) | (hint << 8);
                } else {
                    uint32_t hint = (misc >> 8) & 0xffu;
                    slot = qreader_pos_hint(a.queue, s_prefix, gi, hint);
                    misc = (misc & 0xffff00ffu) | (hint << 8);
                }
                float4 o4, d4;
                if (MODE == TR_SHADOW_APPLY || (MODE == TR_UNIFIED && shadow_lane)) {
                    if (a.st.vis) {
                        const size_t bi = (size_t)a.bounce * a.st.bstride + (MODE == TR_UNIFIED ? a.queue2 : a.queue).base[slot];
                        o4 = a.st.sO[bi];
                        d4 = a.st.sD[bi];
                    } else {
                        o4 = st_ld<PT_NT_TRACE_LD>(&a.st.shO[slot]);
                        d4 = st_ld<PT_NT_TRACE_LD>(&a.st.shD[slot]);
                    }
                    tmin = 0.01f;
                    tmax = 1e16f;
                } else {
                    o4 = st_ld<PT_NT_TRACE_LD>(&a.st.rayO[slot]);
                    d4 = st_ld<PT_NT_TRACE_LD>(&a.st.rayD[slot]);
                    tmin = o4.w;
                    tmax = d4.w;
                }
                // Refilling a lane runs with few lanes active, so it is kept short: the reciprocal direction only feeds the
                // conservative box tests and uses v_rcp_f32 (1 ulp); dn = d / dot(d,d) feeds t and keeps the IEEE division.
                // With idir = ±inf (a direction component that is exactly ±0) the box test's q*(step*idir) + (origin-o)*idir
                // is inf-inf = NaN, the axis constraint is dropped and the ray visits every node of a whole slab (measured:
                // 0.05 % of the shadow rays — probe column 0 — cost milliseconds of tail): a huge finite reciprocal keeps the
                // slab test meaningful and conservative.
                start_ray(o4.x, o4.y, o4.z, d4.x, d4.y, d4.z);
                best = tmax;
                bprim = (MODE == TR_CLOSEST || (MODE == TR_UNIFIED && !shadow_lane)) ? -1 : 0;
                bleaf = -1;
                        // the root is node 0: a group whose only internal child is slot 0 of a virtual parent
                g_base = 0;
                g_imask = 1u;
                g_hits = 1u;
                active = true;
            }
        }
        PT_WLOG(__builtin_amdgcn_s_waitcnt(0); const unsigned long long w_p2 = clock64(); w_cr += w_p2 - w_p1;)
#if PT8_STEAL
        if (exhausted) {
            if (!stealing) { // the wave took its last chunk: from here on its rays are shared work
                stealing = true;
                PT_WLOG(w_tex = wall_clock64();)
                SET_OWNER(lane);
                if (active) {
                    s_leaf[lane] = bleaf; // from here on the leaf index of a ray's best hit lives beside its shared key
                    s_cnt[lane] = 1u;
                    s_key[lane] = local_key();
                }
                PT8_WAVE_SYNC(); // one wave per workgroup: orders the LDS writes above before other lanes' reads
            }
            // ---------------- steal round: idle lane k takes the bottom stack entry of victim k
            const unsigned long long idle2 = __ballot(!active);
            const bool victim = active && sp > SB && SB < lds_depth;
            const unsigned long long vmask = __ballot(victim);
            if (idle2 != 0ull && vmask != 0ull) {
                const uint32_t ni = (uint32_t)__popcll(idle2), nv = (uint32_t)__popcll(vmask);
                const uint32_t m = ni < nv ? ni : nv;
                const uint32_t vrank = (uint32_t)__popcll(vmask & lt_mask), irank = (uint32_t)__popcll(idle2 & lt_mask);
                const bool give = victim && vrank < m, take = !active && irank < m;
                if (give) s_vlane[vrank] = lane;
                PT8_WAVE_SYNC();
                const uint32_t v = take ? s_vlane[irank] : lane; // every lane runs the shuffles; only takers keep what they read
                const float ox = __shfl(r.o.x, (int)v), oy = __shfl(r.o.y, (int)v), oz = __shfl(r.o.z, (int)v);
                const float dx = __shfl(r.d.x, (int)v), dy = __shfl(r.d.y, (int)v), dz = __shfl(r.d.z, (int)v);
                const float ix = __shfl(r.idir.x, (int)v), iy = __shfl(r.idir.y, (int)v), iz = __shfl(r.idir.z, (int)v);
                const float nx_ = __shfl(r.dn.x, (int)v), ny_ = __shfl(r.dn.y, (int)v), nz_ = __shfl(r.dn.z, (int)v);
                const float vtmin = __shfl(tmin, (int)v), vtmax = __shfl(tmax, (int)v);
                const uint32_t vpm = __shfl(pm, (int)v), vslot = __shfl(slot, (int)v), vmisc = __shfl(misc, (int)v);
                const uint32_t vowner = vmisc & 0xffu;
                const int vsb = (int)(vmisc >> 24);
                const int vshadow = __shfl((int)shadow_lane, (int)v);
                if (take) {
                    r.o = mk3(ox, oy, oz);
                    r.d = mk3(dx, dy, dz);
                    r.idir = mk3(ix, iy, iz);
                    r.dn = mk3(nx_, ny_, nz_);
                    tmin = vtmin;
                    tmax = vtmax;
                    pm = vpm;
                    slot = vslot;
                    SET_OWNER(vowner);
                    shadow_lane = vshadow != 0;
                    const uint32_t e0 = s_stack[(vsb * 2) * 64 + v], e1 = s_stack[(vsb * 2 + 1) * 64 + v];
                    g_base = e0;
                    g_imask = e1 & 0xffu;
                    g_hits = e1 >> 8;
                    t_mask = 0;
                    sp = 0;
                    SET_SB(0);
                    // start from the ray's merged state (so that the tie-break "equal t, lower primitive" sees the current holder)
                    const unsigned long long k = s_key[OWNER];
                    if (is_shadow()) {
                        best = tmax;
                        bprim = 0; // k == 0 (already occluded) is caught by the refresh below
                    } else {
                        best = __uint_as_float((uint32_t)(k >> 32));
                        bprim = (int32_t)(uint32_t)k;
                    }
                    atomicAdd(&s_cnt[OWNER], 1u);
                    active = true;
                }
                PT8_WAVE_SYNC(); // takers have read the entries before their victims may overwrite those levels
                if (give) {
                    const int nsb = SB + 1;
                    SET_SB(nsb == sp ? 0 : nsb);
                    if (nsb == sp) sp = 0;
                }
            }
        }
#endif
        PT_WLOG(w_cs += clock64() - w_p2;)
        unsigned long long act = __ballot(active);
        if (act == 0ull) break;
        const uint32_t thresh = exhausted ? 1u : (uint32_t)PT8_REFILL;
        uint32_t it = 0;
        // ---------------- traverse
        do {
#if PT8_STEAL
            if (stealing && active) { // pick up what the ray's other workers found
                const unsigned long long k = s_key[OWNER];
                if (is_shadow()) {
                    if (k == 0ull) finish(); // occluded elsewhere: nothing left to do for this ray
                } else if (k < local_key()) {
                    best = __uint_as_float((uint32_t)(k >> 32));
                    bprim = (int32_t)(uint32_t)k;
                }
            }
#endif
            PT_WLOG(const unsigned long long w_a = clock64();)
            const bool want_tri = active && t_mask != 0u;
            const bool want_node = active && t_mask == 0u; // node step also covers "group empty → pop"
            const unsigned long long m_tri = __ballot(want_tri), m_node = __ballot(want_node);
            // One step type per iteration, the one more lanes wait for (biased 2:1 towards triangle steps), so that the two code paths never
            // run with complementary half-empty masks.  (Running BOTH types per iteration in the tail phase, their loads in flight together,
            // was measured: no gain — a lone wave of 64 rays takes 40 us either way.)
            const bool node_turn = __popcll(m_node) >= PT8_TRI_BIAS * __popcll(m_tri);
            const bool tri_turn = !node_turn;
            // ---- phase 1: addresses and loads (r0..r4 hold a node for node lanes, r0..r2 a triangle for triangle lanes)
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0;
            bool do_node = false;
            if (want_node && node_turn) {
                if (g_hits == 0u) {
                    if (sp == SB) {
                        finish();
                    } else {
                        uint32_t v0, v1;
                        pop(v0, v1);
                        g_base = v0;
                        g_imask = v1 & 0xffu;
                        g_hits = v1 >> 8;
                    }
                }
                if (active) {
                    // next child of the group in (slot ^ octant) order
                    uint32_t h = g_hits, t = h & pm;
                    h = t ? t : h;
                    t = h & (pm >> 8);
                    h = t ? t : h;
                    t = h & (pm >> 16);
                    h = t ? t : h; // a single bit now
                    g_hits ^= h;
                    const uint32_t idx = g_base + (uint32_t)__popc(g_imask & (h - 1u));
                    if (g_hits != 0u) push(g_base, g_imask | (g_hits << 8));
                    PT_STAT(++c_nodes; ++c_ray;)
#if PT8_NODE64
                    const float4* nd = reinterpret_cast<const float4*>(&a.bvh.nodes[idx]); // one 64-byte line: header, then the planes
                    r0 = nd[0]; r2 = nd[1]; r3 = nd[2]; r4 = nd[3];
#else
                    const Node8* nd = &a.bvh.nodes[idx];
                    r0 = nd->n0; r1 = nd->n1; r2 = nd->n2; r3 = nd->n3; r4 = nd->n4;
#endif
                    do_node = true;
                }
            }
            const bool do_tri = want_tri && tri_turn;
            if (do_tri) {
                const uint32_t bit = (uint32_t)__ffs((int)t_mask) - 1u;
                t_mask &= t_mask - 1u;
                PT_STAT(++c_tris; ++c_ray;)
                const uint32_t leaf = leaf_of(t_base, t_bits, bit);
                const LeafTri* tp = &a.bvh.tris[leaf];
                r0 = tp->t0; r1 = tp->t1; r2 = tp->t2;
                r3.x = __uint_as_float(leaf); // rides to phase 2 in a register the node lanes keep live anyway
            }
            PT_WLOG(const unsigned long long w_b = clock64(); __builtin_amdgcn_s_waitcnt(0); const unsigned long long w_c = clock64();)
            // ---- phase 2: the arithmetic
            if (do_node) {
                const float4 n2 = r2, n3 = r3, n4 = r4;
#if PT8_NODE64
                const NodeHdr nh = node_hdr(make_uint4(__float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r0.z), __float_as_uint(r0.w)), a.bvh.grid);
#else
                const NodeHdr nh = node_hdr(r0, r1);
#endif
                const uint32_t imask = nh.imask;
                const float ax = nh.sx * r.idir.x, ay = nh.sy * r.idir.y, az = nh.sz * r.idir.z;
                const float bx = (nh.ox - r.o.x) * r.idir.x, by = (nh.oy - r.o.y) * r.idir.y, bz = (nh.oz - r.o.z) * r.idir.z;
                // near/far planes per axis follow the direction sign
                const bool nx = r.idir.x < 0.0f, ny = r.idir.y < 0.0f, nz = r.idir.z < 0.0f;
                const uint32_t lox0 = __float_as_uint(n2.x), lox1 = __float_as_uint(n2.y), loy0 = __float_as_uint(n2.z), loy1 = __float_as_uint(n2.w);
                const uint32_t loz0 = __float_as_uint(n3.x), loz1 = __float_as_uint(n3.y), hix0 = __float_as_uint(n3.z), hix1 = __float_as_uint(n3.w);
                const uint32_t hiy0 = __float_as_uint(n4.x), hiy1 = __float_as_uint(n4.y), hiz0 = __float_as_uint(n4.z), hiz1 = __float_as_uint(n4.w);
                const uint32_t nearx[2] = {nx ? hix0 : lox0, nx ? hix1 : lox1}, farx[2] = {nx ? lox0 : hix0, nx ? lox1 : hix1};
                const uint32_t neary[2] = {ny ? hiy0 : loy0, ny ? hiy1 : loy1}, fary[2] = {ny ? loy0 : hiy0, ny ? loy1 : hiy1};
                const uint32_t nearz[2] = {nz ? hiz0 : loz0, nz ? hiz1 : loz1}, farz[2] = {nz ? loz0 : hiz0, nz ? loz1 : hiz1};
                // No extra widening of the far plane here: the 8-bit grid (rounded outward from boxes already padded
                // by 2^-16 of the scene size) is orders of magnitude coarser than the rounding of these products.
                // (Measured and rejected, twice: two children per v_pk_fma_f32 — 24 packed instead of 48 scalar FMAs, conversions landing in adjacent
                // registers, no packing moves: traversal 5 % SLOWER in round 3 (0.847 vs 0.807 ms per launch).  The packed FP32 FMA does not issue
                // at twice the scalar rate here, and the register pairs add 7 spills.)
                uint32_t miss = 0u; // sign bits of tfar - tnear, slot 7 first (one subtract + one funnel shift per child)
#pragma unroll
                for (int s = 7; s >= 0; --s) {
                    const int w = s >> 2, k = s & 3;
                    const float tnx = __builtin_fmaf(u8f(nearx[w], k), ax, bx), tfx = __builtin_fmaf(u8f(farx[w], k), ax, bx);
                    const float tny = __builtin_fmaf(u8f(neary[w], k), ay, by), tfy = __builtin_fmaf(u8f(fary[w], k), ay, by);
                    const float tnz = __builtin_fmaf(u8f(nearz[w], k), az, bz), tfz = __builtin_fmaf(u8f(farz[w], k), az, bz);
                    const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                    const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best));
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);
                }
                const uint32_t hm = miss ^ 0xffu; // hit mask in slot positions
                const uint32_t hits = hm & imask;
                // triangles of the leaf children that were hit
                g_base = nh.child_base;
                g_imask = imask;
                g_hits = hits;
                t_base = nh.tri_base;
                t_bits = nh.lbits;
                t_mask = leaf_pending(hm, nh.lbits);
            }
            if (do_tri) {
                const float4 ta = r0, tb = r1, tc = r2;
                float t, det;
                const v3 v0 = mk3(ta.x, ta.y, ta.z), v1 = mk3(ta.w, tb.x, tb.y), v2 = mk3(tb.z, tb.w, tc.x);
                if (tri_test_det(r, v0, v1, v2, t, det)) {
                    const int32_t prim = __float_as_int(tc.y);
                    if (MODE == TR_SHADOW_APPLY || MODE == TR_ANY_QUERY || (MODE == TR_UNIFIED && shadow_lane)) {
                        if (t > tmin && t < tmax && (!a.cull_back || det > 0.0f) && hit_in_box(r, v0, v1, v2, a.bvh.hit_pad, t)) {
                            bprim = 1;
                            best = t;
                            finish();
                        }
                    } else if (t > tmin && (t < best || (t == best && bprim >= 0 && prim < bprim)) && hit_in_box(r, v0, v1, v2, a.bvh.hit_pad, t)) {
                        best = t;
                        bprim = prim;
                        bleaf = (int32_t)__float_as_uint(r3.x);
#if PT8_STEAL
                        if (stealing) {
                            const unsigned long long key = local_key();
                            atomicMin(&s_key[OWNER], key);
                            if (s_key[OWNER] == key) s_leaf[OWNER] = bleaf;
                        }
#endif
                    }
                }
            }
            PT_STAT(++c_iters; if (a.dbg && lane == 0) {
                const bool node_step = __popcll(m_node) >= PT8_TRI_BIAS * __popcll(m_tri);
                c_act += (uint32_t)__popcll(m_node | m_tri);
                c_exec += (uint32_t)__popcll(node_step ? m_node : m_tri);
                c_nodeit += node_step ? 1u : 0u;
            })
            PT_WLOG(++w_iters; w_c01 += w_b - w_a; w_c12 += w_c - w_b; w_c23 += clock64() - w_c;)
#if PT_DEBUG_WAVELOG + 0 == 3
            {
                const uint32_t nact = (uint32_t)__popcll(act); // lanes that were active when the iteration began
                const unsigned long long dt = clock64() - w_a;
                if (nact <= 8u) { ++h_i0; h_c0 += dt; } else if (nact <= 16u) { ++h_i1; h_c1 += dt; } else if (nact <= 32u) { ++h_i2; h_c2 += dt; } else { ++h_i3; h_c3 += dt; }
            }
#endif
            act = __ballot(active);
            ++it;
#if PT8_STEAL
            // with idle lanes around, return to the steal round every PT8_STEAL_PERIOD iterations
            if (stealing && act != ~0ull && it >= (uint32_t)PT8_STEAL_PERIOD) break;
#endif
        } while ((uint32_t)__popcll(act) >= thresh);
    }
    PT_STAT(if (a.dbg) {
        atomicAdd(&a.dbg[0], (unsigned long long)c_nodes);
        atomicAdd(&a.dbg[1], (unsigned long long)c_tris);
        atomicMax(&a.dbg[2], (unsigned long long)c_maxsp);
        atomicAdd(&a.dbg[3], (unsigned long long)c_push);
        atomicMax(&a.dbg[4], (unsigned long long)c_raymax);
        atomicMax(&a.dbg[5], (unsigned long long)c_iters);
        if (lane == 0) {
            atomicAdd(&a.dbg[6], (unsigned long long)c_iters);
            atomicAdd(&a.dbg[7], 1ull);
            atomicAdd(&a.dbg[8], (unsigned long long)c_act);
            atomicAdd(&a.dbg[9], (unsigned long long)c_exec);
            atomicAdd(&a.dbg[10], (unsigned long long)c_nodeit);
        }
    })
#if PT_DEBUG_WAVELOG + 0 == 3
    if (a.hist && lane == 0) {
        atomicAdd(&a.hist[0], h_i0); atomicAdd(&a.hist[1], h_i1); atomicAdd(&a.hist[2], h_i2); atomicAdd(&a.hist[3], h_i3);
        atomicAdd(&a.hist[4], h_c0); atomicAdd(&a.hist[5], h_c1); atomicAdd(&a.hist[6], h_c2); atomicAdd(&a.hist[7], h_c3);
    }
#endif
    PT_WLOG(if (a.dbg && lane == 0) {
        // per-wave log (tools/r5_wavelog.py): [64 + 8 k ...] = work pointer (names the launch), mode and rays of the launch, start, end, start
        // of the stealing phase (100 MHz clock), loop iterations, ticks spent hungry, cells taken | cells donated << 32
        const unsigned long long k = atomicAdd(&a.dbg[63], 1ull);
        if (k < (unsigned long long)PT_WAVELOG_CAP) {
            unsigned long long* w = a.dbg + 64 + 8 * k;
#if PT_DEBUG_WAVELOG + 0 == 2
            w[6] = (w_cw << 32) | (w_cr & 0xffffffffull); // cycles in the loop-top write-back | in the refill
            w[7] = (w_cs << 32) | w_np;                   // cycles in the steal round | outer passes
#else
            w[6] = (w_c01 << 32) | (w_c12 & 0xffffffffull); // cycles in vote + pop + addresses | cycles waiting for the step's loads
            w[7] = (w_c23 << 32) | ((clock64() - w_c0 - w_c01 - w_c12 - w_c23) & 0xffffffffull); // cycles in the arithmetic | outside the steps (refill, steal rounds, write-back)
#endif
            w[0] = (unsigned long long)(uintptr_t)a.work;
            w[1] = ((unsigned long long)MODE << 32) | n;
            w[2] = w_t0;
            w[3] = wall_clock64();
            w[4] = w_tex;
            w[5] = w_iters;
        }
    })
}
#undef OWNER
#undef SB
#undef SET_OWNER
#undef SET_SB

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT8_WAVES_PER_EU, PT8_WAVES_PER_EU)))
k_trace8(Trace8Args a) {
#if PT_DEBUG_WAVELOG + 0 == 3
    a.hist = a.dbg ? a.dbg + 53 : nullptr;
#endif
    trace8_wave<MODE, false>(a, blockIdx.x, gridDim.x, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Camera rays as PACKETS: k_generate writes the paths of a pass sample-major in the order of the pixel list, which is made of 8 x 8-pixel
// blocks — 64 consecutive queue entries are the same sample of one 8 x 8 block, a frustum a few triangles wide.  One wave traverses the
// tree ONCE for its 64 rays: a single stack of node groups per wave (LDS, 512 bytes), node and leaf-triangle records fetched by scalar
// loads (one request per wave instead of five 16-byte loads per lane from up to 64 different lines: the traversal kernels are bound by
// that address rate), every lane testing its own ray against the eight child boxes with its own interval (tmin, best), a child visited
// when ANY lane hits it, a leaf's triangles tested by the lanes that hit the leaf's box.  Control flow is wave-uniform: no votes, no
// per-lane stacks, no idle lanes waiting for the other step type.
// Same answer as k_trace8<TR_CLOSEST>, bit for bit: a ray's result is the minimum (t, then primitive) over the triangles its lane tested,
// every triangle whose padded box the ray enters within (tmin, best] is tested (box tests conservative as in k_trace8, same arithmetic),
// and a triangle tested "too often" cannot add a hit brute force would not find (hit_in_box confines accepted hits to the triangle's box).
// Domain: that confinement holds while hit_in_box's tolerance hp + 2^-21 t |d|_1 stays inside the builders' padding 2 hp, i.e.
// for hits within about 16 scene sizes of the ray's origin; beyond, a rounding-NOISE hit (edge functions that are pure rounding error) accepted
// by a lane under a node its own ray never entered could differ from the per-ray kernel's answer — genuine hits cannot, they lie inside every
// box of their triangle.  tests/test_gpu_packets.py renders from 30 scene sizes away with both kernels (equal).
// Used for the identity queue of bounce 0 only (pt_api.hip); foveated launches queue their paths through sub-queues and keep k_trace8.
#ifndef PT8_CAM_STACK
#define PT8_CAM_STACK 64 // groups with children still to visit: at most one per level (pt_create refuses trees deeper than 62 levels)
#endif
#if __HIP_DEVICE_COMPILE__
typedef const __attribute__((address_space(4))) Node8* ConstNode8;     // constant address space: s_load
typedef const __attribute__((address_space(4))) LeafTri* ConstLeafTri;
typedef const __attribute__((address_space(4))) float4* ConstF4;
#endif
__global__ void __launch_bounds__(64) k_trace8_cam(Trace8Args a) {
#if __HIP_DEVICE_COMPILE__
    __shared__ uint2 s_grp[PT8_CAM_STACK];
    const uint32_t lane = threadIdx.x;
    const uint32_t n = a.queue.counts[0]; // identity queue: entries [0, n), state in launch order
    const uint32_t npk = (n + 63u) >> 6;
    // Work distribution: the first grab of a wave is static (blockIdx), later ones come from ONE work counter, `per` packets at a time, in
    // image order — the waves in flight then work on one band of the image and share the tree's lines in L2.  (Measured and rejected: 64
    // sub-range counters to spread the atomic traffic — the waves spread over 64 bands of the image: C3 8.17 instead of 8.00 ms; eight
    // waves per SIMD (the kernel needs 55 VGPRs) — they crowd out the other chunk streams' kernels: C3 8.6 ms, Cornell 4.1 instead of 3.1.)
    uint32_t per = npk / (gridDim.x * 4u);
    per = per < 4u ? 4u : (per > 16u ? 16u : per);
    uint32_t pk = blockIdx.x * per, pk_end = pk + per;
#if !PT8_NODE64
    const ConstNode8 nodes = (ConstNode8)(uintptr_t)a.bvh.nodes;
#endif
    const ConstLeafTri tris = (ConstLeafTri)(uintptr_t)a.bvh.tris;
    for (;;) {
        if (pk == pk_end) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(a.work, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c) + gridDim.x;
            pk = c * per;
            pk_end = pk + per;
        }
        if (pk >= npk) break;
        const uint32_t pos = (pk << 6) + lane;
        ++pk;
        const bool valid = pos < n;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 1.f), d4 = make_float4(0.f, 0.f, 1.f, -1.f);
        if (valid) {
            o4 = st_ld<PT_NT_TRACE_LD>(&a.st.rayO[pos]);
            d4 = st_ld<PT_NT_TRACE_LD>(&a.st.rayD[pos]);
        }
        RaySetup r;
        r.o = mk3(o4.x, o4.y, o4.z);
        r.d = mk3(d4.x, d4.y, d4.z);
        r.idir = mk3(__builtin_amdgcn_rcpf(d4.x), __builtin_amdgcn_rcpf(d4.y), __builtin_amdgcn_rcpf(d4.z)); // as the refill of k_trace8
        r.dn = scl3(r.d, 1.0f / dot3(r.d, r.d));
        if (!(fabsf(d4.x) > 1e-30f)) r.idir.x = copysignf(1e30f, d4.x);
        if (!(fabsf(d4.y) > 1e-30f)) r.idir.y = copysignf(1e30f, d4.y);
        if (!(fabsf(d4.z) > 1e-30f)) r.idir.z = copysignf(1e30f, d4.z);
        const float tmin = o4.w;
        float best = d4.w; // tmax; a lane past the end holds -1: every box test fails (tf <= -1 < tmin <= tn)
        int32_t bprim = -1, bleaf = -1;
        // visiting order of a group's children: the octant of the packet's first ray (any order gives the same answer)
        const uint32_t pm_lane = ((__float_as_uint(d4.z) >> 31) ? 0xF0u : 0x0Fu) | (((__float_as_uint(d4.y) >> 31) ? 0xCCu : 0x33u) << 8) |
                                 (((__float_as_uint(d4.x) >> 31) ? 0xAAu : 0x55u) << 16);
        const uint32_t pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pm_lane);
        const bool nx = r.idir.x < 0.0f, ny = r.idir.y < 0.0f, nz = r.idir.z < 0.0f;
        uint32_t g_base = 0u, g_imask = 1u, g_hits = 1u; // the root is slot 0 of a virtual parent
        int sp = 0;
        PT_STAT(uint32_t c_nodes = 0; uint32_t c_tris = 0; uint32_t c_mine = 0; uint32_t c_hitl = 0;)
        for (;;) {
            if (g_hits == 0u) {
                if (sp == 0) break;
                --sp;
                const uint2 e = s_grp[sp];
                g_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                g_imask = e1 & 0xffu;
                g_hits = e1 >> 8;
            }
            uint32_t h = g_hits, t = h & pm;
            h = t ? t : h;
            t = h & (pm >> 8);
            h = t ? t : h;
            t = h & (pm >> 16);
            h = t ? t : h; // a single bit
            g_hits ^= h;
            const uint32_t idx = g_base + (uint32_t)__popc(g_imask & (h - 1u));
            if (g_hits != 0u) {
                if (sp < PT8_CAM_STACK) s_grp[sp] = make_uint2(g_base, g_imask | (g_hits << 8)); // every lane writes the same value
                else atomicOr(a.fault, 1u);
                sp = sp < PT8_CAM_STACK ? sp + 1 : sp;
            }
#if PT8_NODE64
            const ConstF4 np = (ConstF4)(uintptr_t)(a.bvh.nodes + idx);
            const float4 hq = np[0], n2 = np[1], n3 = np[2], n4 = np[3];
            const NodeHdr nh = node_hdr(make_uint4(__float_as_uint(hq.x), __float_as_uint(hq.y), __float_as_uint(hq.z), __float_as_uint(hq.w)), a.bvh.grid);
#else
            const float4 n0 = nodes[idx].n0, n1 = nodes[idx].n1, n2 = nodes[idx].n2, n3 = nodes[idx].n3, n4 = nodes[idx].n4;
            const NodeHdr nh = node_hdr(n0, n1);
#endif
            const uint32_t imask = nh.imask;
            const float ax = nh.sx * r.idir.x, ay = nh.sy * r.idir.y, az = nh.sz * r.idir.z;
            const float bx = (nh.ox - r.o.x) * r.idir.x, by = (nh.oy - r.o.y) * r.idir.y, bz = (nh.oz - r.o.z) * r.idir.z;
            const uint32_t lox0 = __float_as_uint(n2.x), lox1 = __float_as_uint(n2.y), loy0 = __float_as_uint(n2.z), loy1 = __float_as_uint(n2.w);
            const uint32_t loz0 = __float_as_uint(n3.x), loz1 = __float_as_uint(n3.y), hix0 = __float_as_uint(n3.z), hix1 = __float_as_uint(n3.w);
            const uint32_t hiy0 = __float_as_uint(n4.x), hiy1 = __float_as_uint(n4.y), hiz0 = __float_as_uint(n4.z), hiz1 = __float_as_uint(n4.w);
            const uint32_t nearx[2] = {nx ? hix0 : lox0, nx ? hix1 : lox1}, farx[2] = {nx ? lox0 : hix0, nx ? lox1 : hix1};
            const uint32_t neary[2] = {ny ? hiy0 : loy0, ny ? hiy1 : loy1}, fary[2] = {ny ? loy0 : hiy0, ny ? loy1 : hiy1};
            const uint32_t nearz[2] = {nz ? hiz0 : loz0, nz ? hiz1 : loz1}, farz[2] = {nz ? loz0 : hiz0, nz ? loz1 : hiz1};
            uint32_t miss = 0u;
#pragma unroll
            for (int s = 7; s >= 0; --s) {
                const int w = s >> 2, k = s & 3;
                const float tnx = __builtin_fmaf(u8f(nearx[w], k), ax, bx), tfx = __builtin_fmaf(u8f(farx[w], k), ax, bx);
                const float tny = __builtin_fmaf(u8f(neary[w], k), ay, by), tfy = __builtin_fmaf(u8f(fary[w], k), ay, by);
                const float tnz = __builtin_fmaf(u8f(nearz[w], k), az, bz), tfz = __builtin_fmaf(u8f(farz[w], k), az, bz);
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best));
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);
            }
            const uint32_t hm = miss ^ 0xffu; // this lane's hit mask in slot positions
            PT_STAT(++c_nodes; c_hitl += (uint32_t)__popcll(__ballot(hm != 0u));)
            uint32_t whm = 0u;                // slots hit by any lane of the packet
#pragma unroll
            for (int s = 0; s < 8; ++s) whm |= __ballot((hm >> s) & 1u) != 0ull ? (1u << s) : 0u;
            // the node's leaf triangles first (they shrink the lanes' intervals before the packet descends) ...
            uint32_t lm = whm & ~imask;
            while (lm != 0u) {
                const uint32_t s = (uint32_t)__ffs((int)lm) - 1u;
                lm &= lm - 1u;
                const bool mine = (hm >> s) & 1u;
                const uint32_t cnt = leaf_count(nh.lbits, s), first_leaf = leaf_first(nh.tri_base, nh.lbits, s);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t leaf = first_leaf + k;
                    const float4 ta = tris[leaf].t0, tb = tris[leaf].t1, tc = tris[leaf].t2;
                    PT_STAT(++c_tris; c_mine += (uint32_t)__popcll(__ballot(mine));)
                    if (mine) {
                        float tt, det;
                        const v3 v0 = mk3(ta.x, ta.y, ta.z), v1 = mk3(ta.w, tb.x, tb.y), v2 = mk3(tb.z, tb.w, tc.x);
                        if (tri_test_det(r, v0, v1, v2, tt, det)) {
                            const int32_t prim = __float_as_int(tc.y);
                            if (tt > tmin && (tt < best || (tt == best && bprim >= 0 && prim < bprim)) && hit_in_box(r, v0, v1, v2, a.bvh.hit_pad, tt)) {
                                best = tt;
                                bprim = prim;
                                bleaf = (int32_t)leaf;
                            }
                        }
                    }
                }
            }
            // ... then its internal children
            g_base = nh.child_base;
            g_imask = imask;
            g_hits = whm & imask;
        }
        if (valid) st_st<PT_NT_TRACE_ST>(&a.st.hit[pos], make_float2(best, __int_as_float(bleaf)));
        PT_STAT(if (a.dbg && lane == 0) {
            atomicAdd(&a.dbg[48], 1ull);
            atomicAdd(&a.dbg[49], (unsigned long long)c_nodes);
            atomicAdd(&a.dbg[50], (unsigned long long)c_tris);
            atomicAdd(&a.dbg[51], (unsigned long long)c_mine);
            atomicAdd(&a.dbg[52], (unsigned long long)c_hitl);
        })
    }
#endif
}
#endif // PT_BVH8_NODE_ONLY
