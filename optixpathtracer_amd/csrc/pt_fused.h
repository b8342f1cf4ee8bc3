// pt_fused.h — the bounce loop of a SMALL pass as one persistent kernel (round 5).
//
// A pass of the wavefront pipeline is a chain of dependent launches — generate, camera traversal, then shade / traversal per bounce — and each
// launch ends when its slowest wave does.  A full frame (8 M paths) hides that: every launch runs at full occupancy for most of its time.  A
// 1/8 share of the frame (what each of 8 GPUs renders; 1 M paths in three pixel chunks) does not: its twenty launches of 30–350 k rays are
// each as long as the slowest ray's chain of dependent traversal steps (1.5 us each with a lone wave on a SIMD), the mean wave is busy 58 % of
// its launch's span (profiles/r5_01_wavelog_w8.txt), and nothing inside a launch can fix that (cross-wave stealing, wide start, interleaving:
// profiles/r5_02_cross_wave_stealing.md).  What can: no barrier between the bounces at all.
//
// k_path_loop gives every persistent wave a private WINDOW of the pass's queue arrays (PT_FUSED_CAP entries of the two ping-pong radiance
// streams and of the shadow stream, at positions [wave x cap, (wave + 1) x cap)) and lets it run the SAME pipeline on that window by itself:
//     refill the window from the pass's pool of unstarted paths (generate_path = k_generate's body)
//     -> trace8_wave<TR_UNIFIED, LOCAL> over the window's closest-hit rays and the previous round's shadow rays (k_trace8's body, with its
//        in-wave work stealing; the shadow write-back adds the visible contributions)
//     -> shade_path<.., LOCAL> for every entry (k_shade's body; queue appends are wave-local, no atomics)
//     -> swap the windows, next round,
// until the pool is empty and the window has drained.  Waves are coupled only through the pool counter (one atomic per refill), so a wave
// whose rays are slow delays nobody, and the chip stays full until the pool runs out.  Per path the arithmetic, the order of its
// contributions (shadow ray of bounce b applied before bounce b + 1 is shaded) and its random numbers are those of the launch chain: every
// buffer is bit-identical (tests/test_gpu_schedule.py::test_fused_bounce_loop_is_bit_identical), and so are the ray counts.
// Scope: the default schedule of scenes without shadow-catcher materials, whole-chunk passes (not foveated launches); pt_api.hip decides.
#pragma once
#include "pt_bvh8.h"

struct PathLoopArgs {
    Trace8Args ta;        // st: the slot-indexed arrays + radiance stream X[0] + the shadow stream; queue.base = queueA, queue2.base = squeue; ovf, fault, ...
    float4 *rayO1, *rayD1, *thr1; // radiance stream X[1] (travels with queueB)
    float2* hit1;
    uint4* rf1;
    uint32_t* qbase1;     // queueB
    ShadeParams sp;       // scene tables, probe, depth limit (the queue views are filled per round)
    FrameParams fp;
    BatchParams bp;
    uint32_t* pool;       // next unstarted path of the pass (zeroed with the pass's counters)
    uint32_t cap;         // window entries per wave, a multiple of 64
    unsigned long long* totals; // [0] closest-hit rays, [1] shadow rays, [3] shaded hits (k_accum_stats' sums)
};

#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES PT8_WAVES_PER_EU
#endif
template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_FUSED_WAVES, PT_FUSED_WAVES)))
k_path_loop(PathLoopArgs A) {
    __shared__ uint32_t s_q[4]; // [0] entries of the next window, [1] shaded hits that continue (incl. those past the depth cutoff), [2] shadow entries
    __shared__ float s_u8[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t woff = blockIdx.x * A.cap;
    const uint32_t total = A.bp.npix * A.bp.S;
    const float* u8lut = nullptr;
    if (A.sp.mesh_tex) {
        for (uint32_t k = lane; k < 256u; k += 64u) s_u8[k] = (float)k / 255.0f;
        u8lut = s_u8;
    }
    const ProbeMarg pm = probe_marg_global(A.sp.probe);
    uint32_t cur = 0, n_cur = 0, n_sh = 0;
    bool pool_open = true;
    unsigned long long c_r = 0, c_s = 0, c_h = 0;
    PT_WLOG(const unsigned long long w_t0 = wall_clock64(); unsigned long long w_tpool = 0; unsigned long long w_trace = 0; unsigned long long w_shade = 0; uint32_t w_rounds = 0;)
    for (;;) {
        Trace8Args t = A.ta;
#if PT_DEBUG_WAVELOG + 0 == 3
        t.hist = A.ta.dbg ? A.ta.dbg + 53 : nullptr;
#endif
        if (cur) { t.st.rayO = A.rayO1; t.st.rayD = A.rayD1; t.st.thr = A.thr1; t.st.hit = A.hit1; t.st.rf = A.rf1; t.queue.base = A.qbase1; }
        // ---- refill the window from the pool
        if (pool_open && n_cur < A.cap) {
            const uint32_t want = A.cap - n_cur;
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(A.pool, want);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            const uint32_t avail = first < total ? (total - first < want ? total - first : want) : 0u;
            if (avail < want) {
                pool_open = false;
                PT_WLOG(w_tpool = wall_clock64();)
            }
            for (uint32_t k = lane; k < avail; k += 64u) {
                const uint32_t i = first + k, pos = woff + n_cur + k;
                generate_path(t.st, A.fp, A.bp, i, pos);
                t.st.thr[pos] = make_float4(1.f, 1.f, 1.f, 1.f); // pathThroughput = 1, rayEta = 1 (deviceProgram.cu:379-380): k_shade's `first` launch does not read it
                t.queue.base[pos] = i;
            }
            n_cur += avail;
        }
        if (n_cur == 0u && n_sh == 0u) break;
        __syncthreads(); // (one wave per workgroup: a fence) the window's entries are written before other lanes trace them
        // ---- closest-hit rays of the window + shadow rays of the previous round
        PT_WLOG(t.dbg = nullptr; const unsigned long long w_a = wall_clock64();) // (the traversal's own per-launch log would get an entry per round)
        trace8_wave<TR_UNIFIED, true>(t, 0u, 1u, n_cur, n_sh, woff);
        PT_WLOG(const unsigned long long w_b = wall_clock64(); w_trace += w_b - w_a; ++w_rounds;)
        c_r += n_cur;
        c_s += n_sh;
        if (lane < 4u) s_q[lane] = 0u;
        __syncthreads();
        // ---- closest-hit / miss programs
        ShadeParams sp = A.sp;
        sp.queue = QView{t.queue.base, nullptr, 0u};
        sp.next_queue = QView{cur ? A.ta.queue.base : A.qbase1, &s_q[0], woff};
        sp.shadow_queue = QView{A.ta.queue2.base, &s_q[2], woff};
        sp.oRayO = cur ? A.ta.st.rayO : A.rayO1;
        sp.oRayD = cur ? A.ta.st.rayD : A.rayD1;
        sp.oThr = cur ? A.ta.st.thr : A.thr1;
        sp.oRf = cur ? A.ta.st.rf : A.rf1;
        sp.first = 0;
        for (uint32_t j = 0; j < n_cur; j += 64u) {
            const uint32_t i = j + lane;
            if (i < n_cur) {
                const uint32_t pos = woff + i;
                shade_path<MODE, false, true>(t.st, sp, pm, u8lut, pos, t.queue.base[pos], t.st.hit[pos]);
            }
        }
        __syncthreads();
        PT_WLOG(w_shade += wall_clock64() - w_b;)
        n_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q[0]);
        c_h += (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q[1]);
        n_sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q[2]);
        cur ^= 1u;
    }
    PT_WLOG(if (A.ta.dbg && lane == 0) { // tools/r5_wavelog.py: mode 7 = a fused wave: rounds, ticks in traversal, ticks in shading (incl. the window swap)
        const unsigned long long k = atomicAdd(&A.ta.dbg[63], 1ull);
        if (k < (unsigned long long)PT_WAVELOG_CAP) {
            unsigned long long* w = A.ta.dbg + 64 + 8 * k;
            w[0] = (unsigned long long)(uintptr_t)A.pool;
            w[1] = (7ull << 32) | total;
            w[2] = w_t0;
            w[3] = wall_clock64();
            w[4] = w_tpool;
            w[5] = w_rounds;
            w[6] = w_trace;
            w[7] = w_shade;
        }
    })
    if (lane == 0) {
        atomicAdd(&A.totals[0], c_r);
        atomicAdd(&A.totals[1], c_s);
        atomicAdd(&A.totals[3], c_h);
    }
}
