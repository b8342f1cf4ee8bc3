// pt_trace2.h — persistent-wave BVH2 traversal for gfx950 (wave64).
//
// What the first kernel (k_trace, pt_kernels.h) left on the table, from rocprofv3 PMC on MI355X:
// VALU lane utilisation 26 % (rays of one wave finish at very different times and node/leaf work
// serialises), 60 % of wave-cycles in s_waitcnt at 2.5 waves/SIMD (the 64-deep LDS stack capped
// occupancy).  This kernel:
//   * one wave per workgroup, waves are persistent: a wave pulls rays from a global work counter
//     and REFILLS idle lanes as soon as fewer than PT2_REFILL lanes are busy (one atomic per refill);
//   * majority-vote scheduling: each iteration the wave executes either one inner-node step or one
//     leaf step, whichever has more lanes waiting for it, so the two code paths never run with
//     complementary half-empty masks;
//   * traversal stack: PT2_LDS_DEPTH entries in LDS ([level][lane], conflict-free) and the rare
//     deeper levels in a per-lane global spill area → 6 KB LDS per wave instead of 16 KB.
// Results are independent of traversal order (closest hit + lowest-primitive tie-break), so this
// kernel is bit-identical to k_trace and to the CPU checker.
#pragma once
#include "pt_kernels.h"

#define PT2_LDS_DEPTH 24
#define PT2_OVF_DEPTH 48
#define PT2_REFILL 40
#define PT2_CHUNK 512


struct Trace2Args {
    PathState st;
    BvhDev bvh;
    QView queue;
    uint32_t* work;  // global work counter, zero before launch (waves take PT2_CHUNK entries per atomic)
    uint32_t* ovf;   // spill stack: [PT2_OVF_DEPTH][gridDim.x * 64]
    unsigned long long* dbg; // optional debug counters (see pt_bvh8.h)
};

#ifndef PT2_WAVES_PER_EU
#define PT2_WAVES_PER_EU 5
#endif
template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT2_WAVES_PER_EU, PT2_WAVES_PER_EU))) k_trace2(Trace2Args a) {
    __shared__ uint32_t s_stack[PT2_LDS_DEPTH * 64];
    __shared__ uint32_t s_prefix[PT_NSUB + 1];
    const uint32_t lane = threadIdx.x;
    const uint32_t gtid = blockIdx.x * 64u + lane;
    const uint32_t gstride = gridDim.x * 64u;
    const uint32_t n = qreader_init(a.queue, s_prefix);
    // work distribution: the queue is cut into chunks of `chunk` rays (64..PT2_CHUNK, sized so that every wave
    // gets about two); chunk number blockIdx.x is this wave's first one (no atomic), later ones come from the
    // global counter — same-address atomics cost ≈10 ns each on MI355X, so they are kept to one per chunk.
    uint32_t chunk = (n / (gridDim.x * 2u)) & ~63u;
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)PT2_CHUNK ? (uint32_t)PT2_CHUNK : chunk);
    if ((unsigned long long)blockIdx.x * chunk >= n) return;
    uint32_t chunk_next = blockIdx.x * chunk; // wave-uniform: the part of the work range this wave still owns
    uint32_t chunk_end = (chunk_next + chunk < n) ? chunk_next + chunk : n;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    bool active = false, exhausted = false;
    RaySetup r;
    r.o = r.d = r.idir = r.dn = mk3(0.f);
    float tmin = 0.f, tmax = 0.f, best = 0.f;
    int32_t bprim = -1, bleaf = -1, node = 0;
    int sp = 0;
    uint32_t slot = 0;
    uint32_t c_nodes = 0, c_tris = 0, c_maxsp = 0, c_push = 0; // step counters, reported only when a.dbg is set (pt_trace + PT_DEBUG_COUNTS)

    auto push = [&](uint32_t v) {
        ++c_push;
        if ((uint32_t)sp + 1 > c_maxsp) c_maxsp = sp + 1;
        if (sp < PT2_LDS_DEPTH) s_stack[sp * 64 + lane] = v;
        else if (sp < PT2_LDS_DEPTH + PT2_OVF_DEPTH) a.ovf[(size_t)(sp - PT2_LDS_DEPTH) * gstride + gtid] = v;
        else return; // unreachable: binary trees deeper than PT_STACK_DEPTH are refused when they are built (pt_api.hip)
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        return sp < PT2_LDS_DEPTH ? s_stack[sp * 64 + lane] : a.ovf[(size_t)(sp - PT2_LDS_DEPTH) * gstride + gtid];
    };
    auto finish = [&]() {
        if (MODE == TR_CLOSEST) {
            a.st.hit[slot] = make_float2(best, __int_as_float(bleaf)); // the leaf triangle's index (pt_bvh.h)
        } else if (MODE == TR_ANY_QUERY) {
            a.st.hit[slot] = make_float2(best, __int_as_float(bprim)); // bprim = 1 occluded / 0
        } else {
            const float4 pe = a.st.pend[slot];
            const int kind = __float_as_int(pe.w);
            const bool occluded = bprim != 0;
            if (kind == PEND_ALPHA) {
                if (occluded) {
                    const float4 x = a.st.alpha[slot];
                    a.st.alpha[slot] = make_float4(x.x + pe.x, x.y + pe.y, x.z + pe.z, 0.f);
                }
            } else if (!occluded) {
                float4* acc = (kind == PEND_DIRECT) ? a.st.direct : a.st.indirect;
                const float4 x = acc[slot];
                acc[slot] = make_float4(x.x + pe.x, x.y + pe.y, x.z + pe.z, 0.f);
            }
        }
        active = false;
    };

    for (;;) {
        // ---------------- refill idle lanes
        const unsigned long long idle = __ballot(!active);
        if (idle != 0ull && !exhausted) {
            const uint32_t cnt = (uint32_t)__popcll(idle);
            if (chunk_next == chunk_end) { // take the next chunk of the queue: one atomic per PT2_CHUNK rays
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(a.work, 1u);
                c = __shfl(c, 0) + gridDim.x;
                const unsigned long long b0 = (unsigned long long)c * chunk;
                chunk_next = b0 < n ? (uint32_t)b0 : n;
                chunk_end = (b0 + chunk < n) ? (uint32_t)(b0 + chunk) : n;
                if (chunk_next >= chunk_end) exhausted = true;
            }
            const uint32_t take = (chunk_end - chunk_next) < cnt ? (chunk_end - chunk_next) : cnt;
            const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
            const uint32_t first = chunk_next;
            chunk_next += take;
            if (!active && rank < take) {
                const uint32_t i = first + rank;
                {
                    slot = qreader_get(a.queue, s_prefix, i);

                    const float4 o4 = a.st.rayO[slot];
                    float4 d4;
                    if (MODE == TR_SHADOW_APPLY) {
                        d4 = a.st.srayD[slot];
                        tmin = 0.01f;
                        tmax = 1e16f;
                    } else {
                        d4 = a.st.rayD[slot];
                        tmin = o4.w;
                        tmax = d4.w;
                    }
                    r = ray_setup(mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z));
                    best = tmax;
                    bprim = (MODE == TR_CLOSEST) ? -1 : 0;
                    bleaf = -1;
                    sp = 0;
                    node = a.bvh.root;
                    active = true;
                }
            }
        }
        unsigned long long act = __ballot(active);
        if (act == 0ull) break;
        const uint32_t thresh = exhausted ? 1u : (uint32_t)PT2_REFILL;
        // ---------------- traverse until too few lanes are busy
        do {
            const bool inner = active && node >= 0;
            const bool leaf = active && node < 0;
            const unsigned long long m_in = __ballot(inner), m_lf = __ballot(leaf);
            if (__popcll(m_in) >= __popcll(m_lf)) {
                if (inner) {
                    bool need_pop = true;
                    if (node != PT_REF_EMPTY) {
                        ++c_nodes;
                        const Node2* nd = &a.bvh.nodes[node];
                        const float4 na = nd->a, nb = nd->b, nc = nd->c, nx = nd->d;
                        float t0, t1;
                        const bool h0 = box_test(na.x, na.y, na.z, na.w, nb.x, nb.y, r, tmin, best, t0);
                        const bool h1 = box_test(nb.z, nb.w, nc.x, nc.y, nc.z, nc.w, r, tmin, best, t1);
                        const int32_t c0 = __float_as_int(nx.x), c1 = __float_as_int(nx.y);
                        if (h0 && h1) {
                            const bool sw = t1 < t0;
                            push((uint32_t)(sw ? c0 : c1));
                            node = sw ? c1 : c0;
                            need_pop = false;
                        } else if (h0 || h1) {
                            node = h0 ? c0 : c1;
                            need_pop = false;
                        }
                    }
                    if (need_pop) {
                        if (sp == 0) finish();
                        else node = (int32_t)pop();
                    }
                }
            } else {
                if (leaf) {
                    const uint32_t code = ~(uint32_t)node;
                    const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
                    bool done = false;
                    for (uint32_t k = 0; k < cnt; ++k) {
                        ++c_tris;
                        const LeafTri* tp = &a.bvh.tris[first + k];
                        const float4 ta = tp->t0, tb = tp->t1, tc = tp->t2;
                        float t, det;
                        const v3 v0 = mk3(ta.x, ta.y, ta.z), v1 = mk3(ta.w, tb.x, tb.y), v2 = mk3(tb.z, tb.w, tc.x);
                        if (tri_test_det(r, v0, v1, v2, t, det)) {
                            const int32_t prim = __float_as_int(tc.y);
                            if (MODE != TR_CLOSEST) {
                                if (t > tmin && t < tmax && hit_in_box(r, v0, v1, v2, a.bvh.hit_pad, t)) {
                                    bprim = 1;
                                    best = t;
                                    done = true;
                                    break;
                                }
                            } else if (t > tmin && (t < best || (t == best && bprim >= 0 && prim < bprim)) && hit_in_box(r, v0, v1, v2, a.bvh.hit_pad, t)) {
                                best = t;
                                bprim = prim;
                                bleaf = (int32_t)(first + k);
                            }
                        }
                    }
                    if (done || sp == 0) finish();
                    else node = (int32_t)pop();
                }
            }
            act = __ballot(active);
        } while ((uint32_t)__popcll(act) >= thresh);
    }
    if (a.dbg) {
        atomicAdd(&a.dbg[0], (unsigned long long)c_nodes);
        atomicAdd(&a.dbg[1], (unsigned long long)c_tris);
        atomicMax(&a.dbg[2], (unsigned long long)c_maxsp);
        atomicAdd(&a.dbg[3], (unsigned long long)c_push);
    }
}
