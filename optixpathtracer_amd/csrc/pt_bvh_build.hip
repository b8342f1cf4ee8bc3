// pt_bvh_build.hip — on-GPU construction of the acceleration structure (replaces optixAccelBuild/optixAccelCompact,
// SimplePathtracer.cpp:561-591).  Morton codes of triangle centroids → radix sort → Karras 2012 hierarchy (LBVH) → bottom-up refit →
// SAH-optimal collapse into the 8-wide compressed tree of pt_bvh8.h (80-byte nodes, 48-byte leaf triangles in leaf order); a PLOC
// hierarchy over the same leaves is collapsed as well and calibration rays choose between the two (k_calibrate8).
// Deterministic: keys are (morton30 << 32 | primitive) so they are unique.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

#include <cstring>
#include <rocprim/rocprim.hpp>

#define PT_BVH8_NODE_ONLY
#include "pt_bvh8.h"
#include "pt_host.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// scene bounds over all triangle vertices that are referenced: bounds[0..2]=min, [3..5]=max (ordered uints)
__global__ void k_bounds(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t ntri,
                         uint32_t* __restrict__ bounds) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < ntri; p += gridDim.x * blockDim.x) {
        for (int k = 0; k < 3; ++k) {
            const float* v = &verts[3 * (size_t)idx[3 * (size_t)p + k]];
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], v[a]);
                hi[a] = fmaxf(hi[a], v[a]);
            }
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bounds[a], f2ord(lo[a]));
            atomicMax(&bounds[3 + a], f2ord(hi[a]));
        }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void k_morton(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t ntri,
                         const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ntri) return;
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = ord2f(bounds[a]);
        ext[a] = ord2f(bounds[3 + a]) - lo[a];
    }
    float c[3];
    for (int a = 0; a < 3; ++a) {
        float v0 = verts[3 * (size_t)idx[3 * (size_t)p + 0] + a];
        float v1 = verts[3 * (size_t)idx[3 * (size_t)p + 1] + a];
        float v2 = verts[3 * (size_t)idx[3 * (size_t)p + 2] + a];
        float mn = fminf(v0, fminf(v1, v2)), mx = fmaxf(v0, fmaxf(v1, v2));
        float cc = 0.5f * (mn + mx);
        float n = ext[a] > 0.0f ? (cc - lo[a]) / ext[a] : 0.0f;
        c[a] = fminf(fmaxf(n * 1024.0f, 0.0f), 1023.0f);
    }
    uint32_t code = (expand10((uint32_t)c[0]) << 2) | (expand10((uint32_t)c[1]) << 1) | expand10((uint32_t)c[2]);
    keys[p] = ((uint64_t)code << 32) | p;
}

// Karras 2012.  Internal nodes 0..n-2, leaves are n-1+i.  child arrays hold node ids in that space.
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ void k_karras(const uint64_t* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                         int* __restrict__ parent, int* __restrict__ rfirst, int* __restrict__ rlast) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc = (lo == gamma) ? (n - 1 + gamma) : gamma;
    int rc = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
    left[i] = lc;
    right[i] = rc;
    parent[lc] = i;
    parent[rc] = i;
    rfirst[i] = lo;
    rlast[i] = hi;
    if (i == 0) parent[0] = -1;
}

// bottom-up boxes: box[node*6..] for all 2n-1 nodes
__global__ void k_refit(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ keys,
                        int n, const int* __restrict__ left, const int* __restrict__ right,
                        const int* __restrict__ parent, float* __restrict__ box, int* __restrict__ visits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = (uint32_t)(keys[i] & 0xffffffffu);
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k) {
        const float* v = &verts[3 * (size_t)idx[3 * (size_t)p + k]];
        for (int a = 0; a < 3; ++a) {
            b[a] = fminf(b[a], v[a]);
            b[3 + a] = fmaxf(b[3 + a], v[a]);
        }
    }
    int node = n - 1 + i;
    for (int a = 0; a < 6; ++a) box[(size_t)node * 6 + a] = b[a];
    __threadfence();
    int cur = parent[node];
    while (cur >= 0) {
        if (atomicAdd(&visits[cur], 1) == 0) return; // first arrival: the sibling finishes this node
        __threadfence();
        const int l = left[cur], r = right[cur];
        for (int a = 0; a < 3; ++a) {
            float lo = fminf(__builtin_nontemporal_load(&box[(size_t)l * 6 + a]), __builtin_nontemporal_load(&box[(size_t)r * 6 + a]));
            float hi = fmaxf(__builtin_nontemporal_load(&box[(size_t)l * 6 + 3 + a]), __builtin_nontemporal_load(&box[(size_t)r * 6 + 3 + a]));
            box[(size_t)cur * 6 + a] = lo;
            box[(size_t)cur * 6 + 3 + a] = hi;
        }
        __threadfence();
        cur = parent[cur];
    }
}

// ---- level-synchronous bottom-up passes.  The climb above pays an agent-scope fence (buffer_wbl2 + buffer_inv, microseconds each) per
// thread and level: 6.3 ms for a million leaves, twice that for the collapse costs.  Instead: the depth of every internal node by pointer
// jumping over the parent array (log2(depth) rounds), nodes bucketed by depth, then one small launch per level from the deepest up — a
// kernel boundary is the only ordering needed, nothing fences.  Same values as the climb (min/max and the DP are order-independent).
__global__ void k_leaf_boxes(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ keys, int n, float* __restrict__ box) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = (uint32_t)(keys[i] & 0xffffffffu);
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k) {
        const float* v = &verts[3 * (size_t)idx[3 * (size_t)p + k]];
        for (int a = 0; a < 3; ++a) {
            b[a] = fminf(b[a], v[a]);
            b[3 + a] = fmaxf(b[3 + a], v[a]);
        }
    }
    for (int a = 0; a < 6; ++a) box[(size_t)(n - 1 + i) * 6 + a] = b[a];
}
__global__ void k_depth_init(int ni, const int* __restrict__ parent, int root, int* __restrict__ anc, int* __restrict__ depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ni) return;
    const int p = i == root ? -1 : parent[i];
    anc[i] = p;
    depth[i] = p >= 0 ? 1 : 0;
}
__global__ void k_depth_jump(int ni, const int* __restrict__ anc_in, const int* __restrict__ d_in, int* __restrict__ anc_out, int* __restrict__ d_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ni) return;
    const int a = anc_in[i];
    if (a >= 0) {
        d_out[i] = d_in[i] + d_in[a];
        anc_out[i] = anc_in[a];
    } else {
        d_out[i] = d_in[i];
        anc_out[i] = -1;
    }
}
#define PT_MAX_TREE_LEVELS 512
// hist[0..PT_MAX_TREE_LEVELS): nodes per depth; hist[PT_MAX_TREE_LEVELS]: nodes whose depth is not final or does not fit
__global__ void k_depth_hist(int ni, const int* __restrict__ anc, const int* __restrict__ depth, int* __restrict__ hist) {
    __shared__ int h[PT_MAX_TREE_LEVELS + 1];
    for (int k = threadIdx.x; k <= PT_MAX_TREE_LEVELS; k += blockDim.x) h[k] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ni; i += gridDim.x * blockDim.x) {
        const int d = depth[i];
        atomicAdd(&h[(anc[i] >= 0 || d >= PT_MAX_TREE_LEVELS) ? PT_MAX_TREE_LEVELS : d], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= PT_MAX_TREE_LEVELS; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], h[k]);
}
// one global atomic per (workgroup, depth present in it): the workgroup counts its nodes per depth in LDS, reserves a range per depth and
// places its nodes inside (a global atomic per NODE on ~40 cursors serialised: 5.3 ms for a million nodes)
__global__ void __launch_bounds__(1024) k_depth_scatter(int ni, const int* __restrict__ depth, int* __restrict__ cursor, int* __restrict__ order) {
    __shared__ int h[PT_MAX_TREE_LEVELS];
    for (int k = threadIdx.x; k < PT_MAX_TREE_LEVELS; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int d = 0, local = 0;
    if (i < ni) {
        d = depth[i];
        local = atomicAdd(&h[d], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PT_MAX_TREE_LEVELS; k += blockDim.x) {
        const int c = h[k];
        if (c) h[k] = atomicAdd(&cursor[k], c); // count -> base of this workgroup's range
    }
    __syncthreads();
    if (i < ni) order[h[d] + local] = i;
}
__global__ void k_refit_level(const int* __restrict__ order, int count, const int* __restrict__ left, const int* __restrict__ right, float* __restrict__ box) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int cur = order[t];
    const int l = left[cur], r = right[cur];
    for (int a = 0; a < 3; ++a) {
        box[(size_t)cur * 6 + a] = fminf(box[(size_t)l * 6 + a], box[(size_t)r * 6 + a]);
        box[(size_t)cur * 6 + 3 + a] = fmaxf(box[(size_t)l * 6 + 3 + a], box[(size_t)r * 6 + 3 + a]);
    }
}

__global__ void k_emit_tris(const float* __restrict__ verts, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ tri_mesh,
                            const uint64_t* __restrict__ keys, int n, LeafTri* __restrict__ tris) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = (uint32_t)(keys[i] & 0xffffffffu);
    const float* v0 = &verts[3 * (size_t)idx[3 * (size_t)p + 0]];
    const float* v1 = &verts[3 * (size_t)idx[3 * (size_t)p + 1]];
    const float* v2 = &verts[3 * (size_t)idx[3 * (size_t)p + 2]];
    LeafTri t;
    t.t0 = make_float4(v0[0], v0[1], v0[2], v1[0]);
    t.t1 = make_float4(v1[1], v1[2], v2[0], v2[1]);
    t.t2 = make_float4(v2[2], __int_as_float((int)p), __int_as_float(tri_mesh ? (int)tri_mesh[p] : 0), 0.f); // (v2.z, primitive, mesh = material record, -)
    tris[i] = t;
}

// ------------------------------------------------------------------ binary hierarchy → compressed 8-wide tree (pt_bvh8.h)
struct Task8 {
    int bnode;     // binary node (Karras numbering: internal 0..n-2, leaf n-1+i)
    uint32_t widx; // index of the wide node to emit
};

__device__ __forceinline__ float box_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
// leaves (sorted-order triangle indices) of a small binary subtree, left to right
__device__ __forceinline__ int gather_leaves(int c, int n, const int* left, const int* right, int* out, int cap) {
    int stack[8], sp = 0, k = 0;
    stack[sp++] = c;
    while (sp && k < cap) {
        const int x = stack[--sp];
        if (x >= n - 1) {
            out[k++] = x - (n - 1);
        } else if (sp + 2 <= 8) {
            stack[sp++] = right[x];
            stack[sp++] = left[x];
        }
    }
    return k;
}
__global__ void k_counts_from_ranges(int n, const int* __restrict__ rfirst, const int* __restrict__ rlast, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n - 1) return;
    cnt[i] = i < n - 1 ? (rlast[i] - rfirst[i] + 1) : 1;
}

// One thread per wide node: open the binary subtree (largest surface area first) until 8 children,
// place children in octant-matching slots, quantise, emit the node, its leaf triangles and the next tasks.

// ---- optimal collapse (Ylitie, Karras, Laine 2017, §3.2): for every binary node n and slot budget i = 1..7,
// C(n,i) = least SAH cost of representing n's subtree with at most i slots of the parent wide node:
//   C(n,1) = min(leaf: A_n * cnt * c_p (cnt <= PT8_LEAF_MAX),  internal: A_n * c_n + D(n,8))
//   C(n,i) = min(D(n,i), C(n,i-1)),   D(n,j) = min_{0<k<j} C(left,k) + C(right,j-k)
// dec[n][i-1] records the choice: i = 1: 0 leaf / 1 internal; i = 2..7: split k or 0 = "as for i-1"; dec[n][7] = split of D(n,8).
__global__ void k_parents(int n, const int* __restrict__ left, const int* __restrict__ right, int root, int* __restrict__ parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    parent[left[i]] = i;
    parent[right[i]] = i;
    if (i == root) parent[i] = -1;
}
__device__ __forceinline__ float padded_area(const float* b, float pad) {
    const float dx = b[3] - b[0] + 2.f * pad, dy = b[4] - b[1] + 2.f * pad, dz = b[5] - b[2] + 2.f * pad;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void collapse_cost_node(int cur, int n, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ cnt_of,
                                                   const float* __restrict__ box, float pad, float cp, float* __restrict__ cost, uint8_t* __restrict__ dec, bool coherent) {
    const int l = left[cur], r = right[cur];
    float Cl[8], Cr[8];
    const float al = padded_area(&box[(size_t)l * 6], pad) * cp, ar = padded_area(&box[(size_t)r * 6], pad) * cp;
    for (int k = 1; k <= 7; ++k) {
        Cl[k] = l >= n - 1 ? al : (coherent ? cost[(size_t)l * 7 + (k - 1)] : __builtin_nontemporal_load(&cost[(size_t)l * 7 + (k - 1)]));
        Cr[k] = r >= n - 1 ? ar : (coherent ? cost[(size_t)r * 7 + (k - 1)] : __builtin_nontemporal_load(&cost[(size_t)r * 7 + (k - 1)]));
    }
    float dist[9];
    int ks[9];
    for (int j = 2; j <= 8; ++j) {
        float best = INFINITY;
        int bk = 1;
        for (int k = 1; k < j; ++k) {
            if (k > 7 || j - k > 7) continue;
            const float c = Cl[k] + Cr[j - k];
            if (c < best) { best = c; bk = k; }
        }
        dist[j] = best;
        ks[j] = bk;
    }
    const float A = padded_area(&box[(size_t)cur * 6], pad);
    const int cnt = cnt_of[cur];
    const float c_leaf = cnt <= PT8_LEAF_MAX ? A * (float)cnt * cp : INFINITY;
    const float c_int = A + dist[8];
    float prev = fminf(c_leaf, c_int);
    uint8_t* d = &dec[(size_t)cur * 8];
    d[0] = c_leaf <= c_int ? 0 : 1;
    cost[(size_t)cur * 7 + 0] = prev;
    for (int b = 2; b <= 7; ++b) {
        if (dist[b] < prev) { prev = dist[b]; d[b - 1] = (uint8_t)ks[b]; }
        else d[b - 1] = 0;
        cost[(size_t)cur * 7 + (b - 1)] = prev;
    }
    d[7] = (uint8_t)ks[8];
}
// fallback for hierarchies deeper than PT_MAX_TREE_LEVELS: the atomic climb (one agent-scope fence per thread and level)
__global__ void k_collapse_cost(int n, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parent,
                                const int* __restrict__ cnt_of, const float* __restrict__ box, float pad, float cp,
                                float* __restrict__ cost, uint8_t* __restrict__ dec, int* __restrict__ visits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent[n - 1 + i];
    while (cur >= 0) {
        if (atomicAdd(&visits[cur], 1) == 0) return; // the sibling subtree's thread finishes this node
        __threadfence();
        collapse_cost_node(cur, n, left, right, cnt_of, box, pad, cp, cost, dec, false);
        __threadfence();
        cur = parent[cur];
    }
}
__global__ void k_collapse_cost_level(const int* __restrict__ order, int count, int n, const int* __restrict__ left, const int* __restrict__ right,
                                      const int* __restrict__ cnt_of, const float* __restrict__ box, float pad, float cp,
                                      float* __restrict__ cost, uint8_t* __restrict__ dec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    collapse_cost_node(order[t], n, left, right, cnt_of, box, pad, cp, cost, dec, true);
}

__global__ void k_collapse8(const Task8* __restrict__ tin, uint32_t nin, Task8* __restrict__ tout, uint32_t* __restrict__ counters /*0 next tasks,1 nodes,2 tris*/,
                            int n, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ cnt_of,
                            const float* __restrict__ box, float pad, const uint8_t* __restrict__ dec /* null: greedy by area */,
                            const LeafTri* __restrict__ tris_sorted, Node8* __restrict__ nodes, LeafTri* __restrict__ tris_out, Grid8 grid) {
    const uint32_t ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= nin) return;
    const Task8 task = tin[ti];
    int ch[8];
    bool internal[8];
    int nch = 0;
    if (dec) { // follow the optimal-collapse decisions: distribute the 8 slots down the binary tree
        int sn[16], sb[16], sp = 0;
        {
            const int k = dec[(size_t)task.bnode * 8 + 7];
            sn[sp] = right[task.bnode]; sb[sp++] = 8 - k;
            sn[sp] = left[task.bnode]; sb[sp++] = k;
        }
        while (sp) {
            const int x = sn[--sp];
            int b = sb[sp];
            if (x >= n - 1) { ch[nch] = x; internal[nch++] = false; continue; }
            while (b > 1 && dec[(size_t)x * 8 + (b - 1)] == 0) --b;
            if (b == 1) { ch[nch] = x; internal[nch++] = dec[(size_t)x * 8] != 0; continue; }
            const int k = dec[(size_t)x * 8 + (b - 1)];
            sn[sp] = right[x]; sb[sp++] = b - k;
            sn[sp] = left[x]; sb[sp++] = k;
        }
    } else {
    nch = 2;
    ch[0] = left[task.bnode];
    ch[1] = right[task.bnode];
    for (;;) {
        if (nch == 8) break;
        int bestj = -1;
        float besta = -1.f;
        for (int j = 0; j < nch; ++j) {
            const int c = ch[j];
            if (c >= n - 1 || cnt_of[c] <= PT8_LEAF_MAX) continue; // stays a leaf child
            const float ar = box_area(&box[(size_t)c * 6]);
            if (ar > besta) {
                besta = ar;
                bestj = j;
            }
        }
        if (bestj < 0) break;
        const int c = ch[bestj];
        ch[bestj] = left[c];
        ch[nch++] = right[c];
    }
    for (int j = 0; j < nch; ++j) internal[j] = ch[j] < n - 1 && cnt_of[ch[j]] > PT8_LEAF_MAX;
    }
    // node box (padded)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < nch; ++j)
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], box[(size_t)ch[j] * 6 + a] - pad);
            hi[a] = fmaxf(hi[a], box[(size_t)ch[j] * 6 + 3 + a] + pad);
        }
    // greedy octant slot assignment: cost(child j, slot s) = dot(centroid_j - centre, signs(s))
    float cen[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    int slot_of[8], child_in[8];
    for (int s = 0; s < 8; ++s) child_in[s] = -1;
    for (int j = 0; j < nch; ++j) slot_of[j] = -1;
    for (int it = 0; it < nch; ++it) {
        float bc = -INFINITY;
        int bj = -1, bs = -1;
        for (int j = 0; j < nch; ++j) {
            if (slot_of[j] >= 0) continue;
            const float* b = &box[(size_t)ch[j] * 6];
            const float dx = 0.5f * (b[0] + b[3]) - cen[0], dy = 0.5f * (b[1] + b[4]) - cen[1], dz = 0.5f * (b[2] + b[5]) - cen[2];
            for (int s = 0; s < 8; ++s) {
                if (child_in[s] >= 0) continue;
                const float c = ((s & 1) ? dx : -dx) + ((s & 2) ? dy : -dy) + ((s & 4) ? dz : -dz);
                if (c > bc) {
                    bc = c;
                    bj = j;
                    bs = s;
                }
            }
        }
        slot_of[bj] = bs;
        child_in[bs] = bj;
    }
    // classify, count
    uint32_t imask = 0, nint = 0, ntri = 0;
    for (int s = 0; s < 8; ++s) {
        const int j = child_in[s];
        if (j < 0) continue;
        const int c = ch[j];
        const int cnt = cnt_of[c];
        if (internal[j]) {
            imask |= 1u << s;
            ++nint;
        } else {
            ntri += (uint32_t)cnt;
        }
    }
    const uint32_t child_base = nint ? atomicAdd(&counters[1], nint) : 0u;
    const uint32_t tri_base = ntri ? atomicAdd(&counters[2], ntri) : 0u;
    const uint32_t task_base = nint ? atomicAdd(&counters[0], nint) : 0u;
#if PT8_NODE64
    // one-line node: the origin on the scene grid, rounded down (never above the node's padded lower corner), and the smallest power-of-two
    // step whose 255 multiples reach the upper corner from there; both exactly as the traversal kernels decode them (node_hdr)
    uint32_t oq[3], e5[3];
    float step[3];
    for (int a = 0; a < 3; ++a) {
        const uint32_t cells = 1u << (a == 1 ? PT8_GRID_BITS_Y : PT8_GRID_BITS_XZ);
        float fq = floorf((lo[a] - grid.glo[a]) / grid.gstep[a]);
        fq = fminf(fmaxf(fq, 0.f), (float)(cells - 1u));
        uint32_t q0 = (uint32_t)fq;
        while (q0 > 0u && __builtin_fmaf((float)q0, grid.gstep[a], grid.glo[a]) > lo[a]) --q0;
        oq[a] = q0;
        lo[a] = __builtin_fmaf((float)q0, grid.gstep[a], grid.glo[a]); // the grid's origin from here on
        uint32_t e = 0u;
        for (; e < 31u; ++e) {
            const float st = __uint_as_float((grid.ebase + e) << 23);
            if (lo[a] + 255.0f * st >= hi[a]) break;
        }
        e5[a] = e;
        step[a] = __uint_as_float((grid.ebase + e) << 23);
    }
#else
    // grid step per axis: ext/255 rounded UP to a float with an 8-bit significand (the 16 bits the node stores), so the
    // 255 steps cover the extent with < 1 % slack (a power-of-two step wasted up to 2x of the 8-bit resolution)
    uint32_t eb[3]; // upper 16 bits of the step
    float step[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = hi[a] - lo[a];
        float st = ext * (1.0f / 255.0f) * 1.015625f;
        if (!(st >= 2.3509887e-38f)) st = 2.3509887e-38f; // 2^-125
        if (st > 1.0e38f) st = 1.0e38f;
        eb[a] = (__float_as_uint(st) + 0xffffu) >> 16;
        step[a] = __uint_as_float(eb[a] << 16);
    }
#endif
    uint32_t q[6][2] = {{0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
    uint32_t leafbits = 0u;
    uint32_t toff = 0, irank = 0;
    for (int s = 0; s < 8; ++s) {
        const int j = child_in[s];
        if (j < 0) continue; // empty slot keeps the inverted box
        const int c = ch[j];
        const float* b = &box[(size_t)c * 6];
        for (int a = 0; a < 3; ++a) {
            float ql = floorf(((b[a] - pad) - lo[a]) / step[a]);
            float qh = ceilf(((b[3 + a] + pad) - lo[a]) / step[a]);
            ql = fminf(fmaxf(ql, 0.f), 255.f);
            qh = fminf(fmaxf(qh, 0.f), 255.f);
            // the division by a non-power-of-two step rounds: make sure the grid planes still enclose the padded box
            while (ql > 0.f && lo[a] + ql * step[a] > b[a] - pad) ql -= 1.f;
            while (qh < 255.f && lo[a] + qh * step[a] < b[3 + a] + pad) qh += 1.f;
            const int w = s >> 2, k = s & 3;
            q[a][w] = (q[a][w] & ~(0xffu << (8 * k))) | ((uint32_t)ql << (8 * k));
            q[3 + a][w] = (q[3 + a][w] & ~(0xffu << (8 * k))) | ((uint32_t)qh << (8 * k));
        }
        if (imask & (1u << s)) {
            tout[task_base + irank] = Task8{c, child_base + irank};
            ++irank;
        } else {
            int ids[PT8_LEAF_MAX > 4 ? PT8_LEAF_MAX : 4];
            const int cnt = gather_leaves(c, n, left, right, ids, PT8_LEAF_MAX > 4 ? PT8_LEAF_MAX : 4);
            for (int k = 0; k < cnt; ++k) tris_out[tri_base + toff + k] = tris_sorted[ids[k]];
            leafbits |= ((1u << cnt) - 1u) << (3 * s);
            toff += (uint32_t)cnt;
        }
    }
    Node8 nd;
#if PT8_NODE64
    uint32_t b0 = 0u, b1 = 0u; // count planes of the leaf slots
    for (int s = 0; s < 8; ++s) {
        const uint32_t c = (uint32_t)__popc((leafbits >> (3 * s)) & 7u);
        b0 |= (c & 1u) << s;
        b1 |= (c >> 1) << s;
    }
    nd.h = make_uint4(child_base | (imask << 24), tri_base | (e5[0] << 24) | ((oq[1] & 7u) << 29),
                      b0 | (b1 << 8) | (e5[1] << 16) | (e5[2] << 21) | (((oq[1] >> 3) & 63u) << 26), oq[0] | (oq[2] << 14) | ((oq[1] >> 9) << 28));
    nd.q0 = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
    nd.q1 = make_uint4(q[2][0], q[2][1], q[3][0], q[3][1]);
    nd.q2 = make_uint4(q[4][0], q[4][1], q[5][0], q[5][1]);
#else
    nd.n0 = make_float4(lo[0], lo[1], lo[2], __uint_as_float(eb[0] | (eb[1] << 16)));
    nd.n1 = make_float4(__uint_as_float(child_base), __uint_as_float(tri_base), __uint_as_float(leafbits), __uint_as_float(eb[2] | (imask << 16)));
    nd.n2 = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[1][0]), __uint_as_float(q[1][1]));
    nd.n3 = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[3][0]), __uint_as_float(q[3][1]));
    nd.n4 = make_float4(__uint_as_float(q[4][0]), __uint_as_float(q[4][1]), __uint_as_float(q[5][0]), __uint_as_float(q[5][1]));
#endif
    nodes[task.widx] = nd;
}

// ------------------------------------------------------------------ PLOC (Meister & Bittner 2018)
// Bottom-up agglomerative clustering over the Morton-sorted leaves: every cluster looks PLOC_R positions left and
// right for the partner that minimises the surface area of the merged box; mutual nearest neighbours merge; repeat.
// Much closer to a SAH tree than the LBVH split-at-Morton-bit hierarchy, still fully parallel.
#define PLOC_R 25
__device__ __forceinline__ int ploc_search(const int* cl, int N, const float* __restrict__ box, int tie_partner, int i) {
    const float* bi = &box[(size_t)cl[i] * 6];
    const float l0 = bi[0], l1 = bi[1], l2 = bi[2], h0 = bi[3], h1 = bi[4], h2 = bi[5];
    float best = INFINITY;
    int bj = -1;
    const int j0 = i - PLOC_R < 0 ? 0 : i - PLOC_R, j1 = i + PLOC_R > N - 1 ? N - 1 : i + PLOC_R;
    auto dist = [&](int j) {
        const float* bjp = &box[(size_t)cl[j] * 6];
        const float dx = fmaxf(h0, bjp[3]) - fminf(l0, bjp[0]), dy = fmaxf(h1, bjp[4]) - fminf(l1, bjp[1]), dz = fmaxf(h2, bjp[5]) - fminf(l2, bjp[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    // Ties go to the lowest index (deterministic).  With that rule a neighbourhood of identical boxes (thousands of copies of one
    // triangle) has a single mutual pair per round — clusters 0 and 1 — and the build would need as many rounds as there are clusters:
    // a round that merges next to nothing is therefore repeated with tie_partner = 1, where a tie goes to the parity partner i ^ 1 first
    // and every tie-only neighbourhood pairs up at once.  (The partner rule for every round was measured on the stadium scene: a 9 % worse
    // tree — there ties are common and "lowest index" = the far end of the Morton window happens to pick better partners.)
    const int jp = tie_partner ? (i ^ 1) : -1;
    if (jp >= j0 && jp <= j1) {
        best = dist(jp);
        bj = jp;
    }
    for (int j = j0; j <= j1; ++j) {
        if (j == i || j == jp) continue;
        const float a = dist(j);
        if (a < best) {
            best = a;
            bj = j;
        }
    }
    return bj;
}
__global__ void k_ploc_nn(const int* __restrict__ cl, int N, const float* __restrict__ box, int tie_partner, int* __restrict__ nn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    nn[i] = ploc_search(cl, N, box, tie_partner, i);
}
// flags: merge[i] = 1 if i starts a merged pair (i < nn[i], mutual); valid[i] = 0 if i is the absorbed partner
__global__ void k_ploc_flags(const int* __restrict__ nn, int N, uint32_t* __restrict__ merge, uint32_t* __restrict__ valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    merge[i] = (mutual && i < j) ? 1u : 0u;
    valid[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ void k_ploc_merge(const int* __restrict__ cl, const int* __restrict__ nn, int N, const uint32_t* __restrict__ merge,
                             const uint32_t* __restrict__ merge_rank, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ valid_rank,
                             int next_id, int* __restrict__ left, int* __restrict__ right, float* __restrict__ box, int* __restrict__ cnt,
                             int* __restrict__ cl_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !valid[i]) return;
    int node = cl[i];
    if (merge[i]) {
        const int a = cl[i], b = cl[nn[i]];
        node = next_id + (int)merge_rank[i];
        left[node] = a;
        right[node] = b;
        const float *ba = &box[(size_t)a * 6], *bb = &box[(size_t)b * 6];
        float* bo = &box[(size_t)node * 6];
        for (int k = 0; k < 3; ++k) {
            bo[k] = fminf(ba[k], bb[k]);
            bo[3 + k] = fmaxf(ba[3 + k], bb[3 + k]);
        }
        cnt[node] = cnt[a] + cnt[b];
    }
    cl_out[valid_rank[i]] = node;
}
__global__ void k_ploc_init(int n, int* __restrict__ cl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cl[i] = n - 1 + i; // leaf node ids, Morton order
}

// the two numbers the host needs of a round, in one 8-byte read-back: pairs merged, clusters left
__global__ void k_ploc_totals(const uint32_t* __restrict__ merge, const uint32_t* __restrict__ merge_rank, const uint32_t* __restrict__ valid,
                              const uint32_t* __restrict__ valid_rank, int N, uint32_t* __restrict__ out2) {
    if (threadIdx.x || blockIdx.x) return;
    out2[0] = merge[N - 1] + merge_rank[N - 1];
    out2[1] = valid[N - 1] + valid_rank[N - 1];
}
// The last rounds of the clustering (N <= PLOC_TAIL clusters: about a dozen rounds, each five launches, a read-back and a host wait when run
// from the host — 3-4 ms of a 7-11 ms hierarchy) in ONE workgroup: the same search, the same acceptance rule for a round, the same
// numbering of the merged nodes, hence the same tree.  result[0] = root, result[1] = 0 ok / 1 a round merged nothing.
#define PLOC_TAIL 2048
__global__ void __launch_bounds__(1024) k_ploc_tail(const int* __restrict__ cl_in, int N, int next_id, int* __restrict__ left, int* __restrict__ right,
                                                    float* box, int* cnt, int* __restrict__ result) {
    __shared__ int cl[2][PLOC_TAIL];
    __shared__ int nn[PLOC_TAIL];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t totals;
    const int tid = threadIdx.x;
    for (int e = tid; e < N; e += 1024) cl[0][e] = cl_in[e];
    __syncthreads();
    int cur = 0, status = 0;
    while (N > 1) {
        int tp = 0;
        uint32_t m[2], v[2], excl = 0, merged = 0, nleft = 0;
        for (;;) {
            for (int k = 0; k < 2; ++k) {
                const int e = 2 * tid + k;
                if (e < N) nn[e] = ploc_search(cl[cur], N, box, tp, e);
            }
            __syncthreads();
            for (int k = 0; k < 2; ++k) {
                const int e = 2 * tid + k;
                m[k] = 0; v[k] = 0;
                if (e < N) {
                    const int j = nn[e];
                    const bool mutual = j >= 0 && nn[j] == e;
                    m[k] = (mutual && e < j) ? 1u : 0u;
                    v[k] = (mutual && e > j) ? 0u : 1u;
                }
            }
            // exclusive scan of (m, v) packed as m | v << 16 (N <= 2048 < 65536): two elements per thread, then 1024 partial sums
            const uint32_t mine = (m[0] + m[1]) | ((v[0] + v[1]) << 16);
            part[tid] = mine;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const uint32_t add = tid >= off ? part[tid - off] : 0u;
                __syncthreads();
                part[tid] += add;
                __syncthreads();
            }
            excl = part[tid] - mine;
            if (tid == 1023) totals = part[1023];
            __syncthreads();
            merged = totals & 0xffffu;
            nleft = totals >> 16;
            if (tp == 0 && !((long long)merged * 64 >= N || N <= 64)) { // a starved round: again with the partner rule (build_ploc)
                tp = 1;
                __syncthreads();
                continue;
            }
            break;
        }
        if (merged == 0) { status = 1; break; }
        uint32_t mr = excl & 0xffffu, vr = excl >> 16;
        for (int k = 0; k < 2; ++k) {
            const int e = 2 * tid + k;
            if (e < N && v[k]) {
                int node = cl[cur][e];
                if (m[k]) {
                    const int a = cl[cur][e], b = cl[cur][nn[e]];
                    node = next_id + (int)mr;
                    left[node] = a;
                    right[node] = b;
                    const float *ba = &box[(size_t)a * 6], *bb = &box[(size_t)b * 6];
                    float* bo = &box[(size_t)node * 6];
                    for (int q = 0; q < 3; ++q) {
                        bo[q] = fminf(ba[q], bb[q]);
                        bo[3 + q] = fmaxf(ba[3 + q], bb[3 + q]);
                    }
                    cnt[node] = cnt[a] + cnt[b];
                }
                cl[cur ^ 1][vr] = node;
            }
            mr += m[k];
            vr += v[k];
        }
        __threadfence_block(); // the merged nodes' boxes and counts are read by other waves of this workgroup in the next round
        __syncthreads();
        next_id += (int)merged;
        N = (int)nleft;
        cur ^= 1;
    }
    if (tid == 0) {
        result[0] = cl[cur][0];
        result[1] = status;
    }
}

// scenes with <= PT8_LEAF_MAX triangles: one node, one leaf child
__global__ void k_single_node8(int n, const float* __restrict__ bounds6, float pad, Node8* __restrict__ nodes, Grid8 grid) {
    if (threadIdx.x || blockIdx.x) return;
    float lo[3], hi[3];
#if PT8_NODE64
    uint32_t e5[3];
    for (int a = 0; a < 3; ++a) { // origin = the grid's own origin (cell 0), slot 0 = the whole 255-step range
        hi[a] = bounds6[3 + a] + pad;
        uint32_t e = 0u;
        for (; e < 31u; ++e)
            if (grid.glo[a] + 255.0f * __uint_as_float((grid.ebase + e) << 23) >= hi[a]) break;
        e5[a] = e;
    }
    const uint32_t c = (uint32_t)n; // 1..3 triangles in slot 0
    Node8 nd;
    nd.h = make_uint4(0u, e5[0] << 24, (c & 1u) | ((c >> 1) << 8) | (e5[1] << 16) | (e5[2] << 21), 0u);
    nd.q0 = make_uint4(0xffffff00u, 0xffffffffu, 0xffffff00u, 0xffffffffu);
    nd.q1 = make_uint4(0xffffff00u, 0xffffffffu, 0x000000ffu, 0u);
    nd.q2 = make_uint4(0x000000ffu, 0u, 0x000000ffu, 0u);
    nodes[0] = nd;
    (void)lo;
#else
    uint32_t eb[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = bounds6[a] - pad;
        hi[a] = bounds6[3 + a] + pad;
        int e;
        frexpf((hi[a] - lo[a]) / 255.0f, &e);
        if (e < -125) e = -125;
        if (e > 127) e = 127;
        eb[a] = (uint32_t)(e + 127);
    }
    Node8 nd;
    nd.n0 = make_float4(lo[0], lo[1], lo[2], __uint_as_float((eb[0] << 7) | (eb[1] << 23)));
    nd.n1 = make_float4(__uint_as_float(0u), __uint_as_float(0u), __uint_as_float((1u << n) - 1u), __uint_as_float(eb[2] << 7));
    const uint32_t qlo = 0xffffff00u, qhi = 0x000000ffu; // slot 0 = whole grid, others inverted
    nd.n2 = make_float4(__uint_as_float(qlo), __uint_as_float(0xffffffffu), __uint_as_float(qlo), __uint_as_float(0xffffffffu));
    nd.n3 = make_float4(__uint_as_float(qlo), __uint_as_float(0xffffffffu), __uint_as_float(qhi), __uint_as_float(0u));
    nd.n4 = make_float4(__uint_as_float(qhi), __uint_as_float(0u), __uint_as_float(qhi), __uint_as_float(0u));
    nodes[0] = nd;
#endif
}

// ---- choosing between two hierarchies by measurement (pt_bvh_build, PT_BVH_BUILDER unset).  The SAH cost of the collapsed tree
// mispredicts: on the voxel terrain PLOC has the LOWER cost (17.19 vs 17.67) and 12 % MORE node steps per ray, on the stadium scene the
// lower cost (3.73 vs 6.88) and 15 % fewer.  So both wide trees are built and a fixed, deterministic batch of calibration rays — from the
// centroid of one triangle towards the centroid of another, both drawn by a hash of the ray index: the segments along which a path
// tracer's camera, bounce and shadow rays travel — is traced through each by this plain per-thread traversal, which visits what
// k_trace8 visits (a node step per internal child hit when its parent was tested, the leaf triangles of a node before its children,
// children in slot ^ octant order, everything culled by the closest hit so far) and only counts.
// One launch traces the batch through BOTH trees (the second half of the grid takes nodes_b / tris_b and adds to counts + 2): a single ray per
// lane and 1024 waves leave the chip latency-bound, so two launches cost twice one.
__global__ void __launch_bounds__(64) k_calibrate8(const Node8* __restrict__ nodes_a, const LeafTri* __restrict__ tris_a, const Node8* __restrict__ nodes_b,
                                                   const LeafTri* __restrict__ tris_b, const LeafTri* __restrict__ src,
                                                   uint32_t ntri, uint32_t nrays, float hp, unsigned long long* __restrict__ counts, Grid8 grid, int mode, float4 sphere) {
    const uint32_t half = (nrays + 63u) / 64u;
    const bool second = blockIdx.x >= half;
    const Node8* __restrict__ nodes = second ? nodes_b : nodes_a;
    const LeafTri* __restrict__ tris = second ? tris_b : tris_a;
    if (second) counts += 2;
    const uint32_t i = (blockIdx.x - (second ? half : 0u)) * blockDim.x + threadIdx.x;
    unsigned long long nsteps = 0, ntests = 0;
    if (i < nrays) {
        uint32_t h = i * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t ia = h % ntri;
        h = h * 3266489917u + 668265263u; h ^= h >> 16;
        const uint32_t ib = h % ntri;
        const LeafTri ta = src[ia], tb = src[ib];
        const v3 ca = mk3((ta.t0.x + ta.t0.w + ta.t1.z) * (1.f / 3.f), (ta.t0.y + ta.t1.x + ta.t1.w) * (1.f / 3.f), (ta.t0.z + ta.t1.y + ta.t2.x) * (1.f / 3.f));
        const v3 cb = mk3((tb.t0.x + tb.t0.w + tb.t1.z) * (1.f / 3.f), (tb.t0.y + tb.t1.x + tb.t1.w) * (1.f / 3.f), (tb.t0.z + tb.t1.y + tb.t2.x) * (1.f / 3.f));
        v3 org = ca;
        v3 d = sub3(cb, ca);
        float dl = sqrtf(dot3(d, d));
        if (mode == 1) {
            // the rays of a frame rather than segments between triangles (round 5: the segments preferred the LBVH on the voxel terrain, where the
            // SAH tree renders 1.5 % faster): odd rays leave a triangle in a uniform direction of the hemisphere about its geometric normal (bounce and
            // shadow rays), even rays come from a point outside the scene towards a triangle (camera rays); everything by hashes of the ray index
            h = h * 2891336453u + 1013904223u; h ^= h >> 15;
            const float u1 = (float)(h & 0xffffu) * (1.0f / 65536.0f), u2 = (float)(h >> 16) * (1.0f / 65536.0f);
            if (i & 1u) {
                const v3 v0 = mk3(ta.t0.x, ta.t0.y, ta.t0.z), v1 = mk3(ta.t0.w, ta.t1.x, ta.t1.y), v2 = mk3(ta.t1.z, ta.t1.w, ta.t2.x);
                v3 nrm = cross3(sub3(v1, v0), sub3(v2, v0));
                const float nl = sqrtf(dot3(nrm, nrm));
                nrm = nl > 0.f ? scl3(nrm, ((ib & 1u) ? -1.0f : 1.0f) / nl) : mk3(0.f, 1.f, 0.f);
                const v3 ax = fabsf(nrm.x) > 0.9f ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
                v3 tt = cross3(ax, nrm);
                tt = scl3(tt, 1.0f / sqrtf(dot3(tt, tt)));
                const v3 bb = cross3(nrm, tt);
                const float z = u1, rr = sqrtf(fmaxf(0.f, 1.f - z * z)), phi = 6.2831853f * u2;
                d = add3(add3(scl3(tt, rr * cosf(phi)), scl3(bb, rr * sinf(phi))), scl3(nrm, z));
                dl = sphere.w;
            } else {
                const float z = 1.f - 2.f * u1, rr = sqrtf(fmaxf(0.f, 1.f - z * z)), phi = 6.2831853f * u2;
                org = add3(mk3(sphere.x, sphere.y, sphere.z), scl3(mk3(rr * cosf(phi), z, rr * sinf(phi)), 1.5f * sphere.w));
                d = sub3(ca, org);
                dl = sqrtf(dot3(d, d));
                d = dl > 0.f ? scl3(d, 1.0f / dl) : mk3(0.f, 1.f, 0.f);
            }
        } else {
            d = dl > 0.f ? scl3(d, 1.0f / dl) : mk3(0.f, 1.f, 0.f);
        }
        RaySetup r = ray_setup(org, d);
        if (!(fabsf(d.x) > 1e-30f)) r.idir.x = copysignf(1e30f, d.x);
        if (!(fabsf(d.y) > 1e-30f)) r.idir.y = copysignf(1e30f, d.y);
        if (!(fabsf(d.z) > 1e-30f)) r.idir.z = copysignf(1e30f, d.z);
        const uint32_t oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
        const float tmin = 1e-3f * fmaxf(dl, 1e-3f);
        float best = 1e16f;
        uint32_t stack[192];
        int sp = 0;
        stack[sp++] = 0u;
        while (sp) {
            const Node8 nd = nodes[stack[--sp]];
            ++nsteps;
#if PT8_NODE64
            const NodeHdr nh = node_hdr(nd.h, grid);
            const uint32_t q[12] = {nd.q0.x, nd.q0.y, nd.q0.z, nd.q0.w, nd.q1.x, nd.q1.y, nd.q1.z, nd.q1.w, nd.q2.x, nd.q2.y, nd.q2.z, nd.q2.w};
#else
            const NodeHdr nh = node_hdr(nd.n0, nd.n1);
            const uint32_t q[12] = {__float_as_uint(nd.n2.x), __float_as_uint(nd.n2.y), __float_as_uint(nd.n2.z), __float_as_uint(nd.n2.w),
                                    __float_as_uint(nd.n3.x), __float_as_uint(nd.n3.y), __float_as_uint(nd.n3.z), __float_as_uint(nd.n3.w),
                                    __float_as_uint(nd.n4.x), __float_as_uint(nd.n4.y), __float_as_uint(nd.n4.z), __float_as_uint(nd.n4.w)};
#endif
            const float sx = nh.sx, sy = nh.sy, sz = nh.sz;
            const uint32_t imask = nh.imask;
            uint32_t hits = 0;
            for (int sl = 0; sl < 8; ++sl) {
                const int w = sl >> 2, k = sl & 3;
                const float lx = nh.ox + u8f(q[0 + w], k) * sx, ly = nh.oy + u8f(q[2 + w], k) * sy, lz = nh.oz + u8f(q[4 + w], k) * sz;
                const float hx = nh.ox + u8f(q[6 + w], k) * sx, hy = nh.oy + u8f(q[8 + w], k) * sy, hz = nh.oz + u8f(q[10 + w], k) * sz;
                float tn;
                if (lx <= hx && box_test(lx, ly, lz, hx, hy, hz, r, tmin, best, tn)) hits |= 1u << sl;
            }
            // the leaf triangles of this node first ...
            for (int sl = 0; sl < 8; ++sl) {
                if (!(hits & (1u << sl)) || (imask & (1u << sl))) continue;
                const uint32_t cnt = leaf_count(nh.lbits, (uint32_t)sl), first_leaf = leaf_first(nh.tri_base, nh.lbits, (uint32_t)sl);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const LeafTri t = tris[first_leaf + k];
                    ++ntests;
                    float tt, det;
                    const v3 v0 = mk3(t.t0.x, t.t0.y, t.t0.z), v1 = mk3(t.t0.w, t.t1.x, t.t1.y), v2 = mk3(t.t1.z, t.t1.w, t.t2.x);
                    if (tri_test_det(r, v0, v1, v2, tt, det) && tt > tmin && tt < best && hit_in_box(r, v0, v1, v2, hp, tt)) best = tt;
                }
            }
            // ... then its internal children, the one the ray enters first on top of the stack
            for (int k = 7; k >= 0; --k) {
                const uint32_t sl = (uint32_t)k ^ oct;
                if ((hits & imask & (1u << sl)) && sp < 192) stack[sp++] = nh.child_base + (uint32_t)__popc(imask & ((1u << sl) - 1u));
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        nsteps += __shfl_xor(nsteps, off);
        ntests += __shfl_xor(ntests, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], nsteps);
        atomicAdd(&counts[1], ntests);
    }
}

} // namespace

#define HIPCHK(x)                         \
    do {                                  \
        hipError_t e_ = (x);              \
        if (e_ != hipSuccess) return e_;  \
    } while (0)

// The build's temporaries (≈270 bytes per triangle over ~40 arrays) come out of ONE device allocation: hipMalloc / hipFree cost tens to
// hundreds of microseconds each and a free synchronises the device (3 ms of "frees" at the end of a 1 M-triangle build, measured).
// Phases release what they took by resetting the mark (ArenaMark).  A request that does not fit falls back to hipMalloc.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    // PT_BVH_GUARD=1 (tests/test_gpu_builder.py, tools): every slice is followed by a 256-byte band of 0xA5 that nothing may write; the bands of a
    // phase are read back when its mark is released (ArenaMark) and at the end of the build, and one changed byte fails the build — an overflow
    // INSIDE the arena lands in a neighbouring slice and would otherwise go unnoticed (or corrupt a tree silently), unlike one past a hipMalloc.
    bool guard = false;
    struct Band { size_t off, slice_off, slice_bytes; };
    std::vector<Band> bands;
    int guard_bad = 0;
    static const size_t kBand = 256;
    hipError_t init(size_t bytes) {
        guard = getenv("PT_BVH_GUARD") != nullptr && atoi(getenv("PT_BVH_GUARD")) != 0;
        cap = bytes + (guard ? 256 * kBand * 2 : 0); // room for the bands of a few hundred slices
        used = 0;
        return hipMalloc((void**)&base, cap);
    }
    void* take(size_t bytes) {
        const size_t a = (used + 255) & ~(size_t)255;
        const size_t end = guard ? ((a + bytes + 255) & ~(size_t)255) + kBand : a + bytes;
        if (!base || end > cap) return nullptr;
        if (guard) {
            if (hipMemset(base + end - kBand, 0xA5, kBand) != hipSuccess) return nullptr;
            bands.push_back({end - kBand, a, bytes});
        }
        used = end;
        return base + a;
    }
    // read back the bands at or above `from` (the device must be idle: the callers synchronise) and forget them
    void verify(size_t from) {
        unsigned char h[kBand];
        while (!bands.empty() && bands.back().off >= from) {
            const Band b = bands.back();
            bands.pop_back();
            if (hipMemcpy(h, base + b.off, kBand, hipMemcpyDeviceToHost) != hipSuccess) { ++guard_bad; continue; }
            for (size_t k = 0; k < kBand; ++k)
                if (h[k] != 0xA5) {
                    fprintf(stderr, "[pt_bvh] ARENA GUARD: byte %zu after the slice at offset %zu (%zu bytes) was overwritten (0x%02x)\n", k, b.slice_off, b.slice_bytes, h[k]);
                    ++guard_bad;
                    break;
                }
        }
    }
    bool owns(const void* q) const { return base && (const char*)q >= base && (const char*)q < base + cap; }
    ~Arena() { if (base) hipFree(base); }
};
thread_local Arena* t_arena = nullptr; // pt_multi builds one tree per rank thread
template <typename T> static hipError_t tmalloc(T** q, size_t bytes) {
    if (t_arena) {
        if (void* r = t_arena->take(bytes ? bytes : 16)) { *q = (T*)r; return hipSuccess; }
    }
    return hipMalloc((void**)q, bytes ? bytes : 16);
}
static void tfree(const void* q) {
    if (q && !(t_arena && t_arena->owns(q))) hipFree(const_cast<void*>(q));
}
struct ArenaMark { // arena memory taken after the mark is handed back when the mark goes out of scope
    size_t used;
    ArenaMark() : used(t_arena ? t_arena->used : 0) {}
    ~ArenaMark() {
        if (!t_arena) return;
        if (t_arena->guard) {
            (void)hipDeviceSynchronize();
            t_arena->verify(used);
        }
        t_arena->used = used;
    }
};
// frees its device allocations on every exit path of the function that owns it
struct DevFrees {
    std::vector<void*> p;
    template <typename T> hipError_t alloc(T** q, size_t bytes) { hipError_t e = tmalloc(q, bytes); if (e == hipSuccess) p.push_back((void*)*q); return e; }
    ~DevFrees() { for (void* q : p) tfree(q); }
};

// internal nodes of a binary hierarchy bucketed by depth (k_depth_* above): order[off[d] .. off[d+1]) are the nodes at depth d
struct LevelOrder {
    int* order = nullptr; // device, n-1 entries
    std::vector<int> off; // host, levels + 1 entries; empty = the hierarchy is deeper than PT_MAX_TREE_LEVELS (callers fall back to the climb)
    void release() { tfree(order); order = nullptr; off.clear(); }
};
static hipError_t build_levels(int n, const int* parent, int root, hipStream_t stream, LevelOrder* lv) {
    const int ni = n - 1, B = 256;
    lv->release();
    if (getenv("PT_BVH_CLIMB")) return hipSuccess; // test hook: the atomic climb instead (the trees must come out the same)
    HIPCHK(tmalloc(&lv->order, sizeof(int) * (size_t)std::max(ni, 1))); // outlives this function's mark
    ArenaMark mark;
    DevFrees mem;
    int *anc[2] = {nullptr, nullptr}, *dep[2] = {nullptr, nullptr}, *hist = nullptr;
    for (int k = 0; k < 2; ++k) { HIPCHK(mem.alloc(&anc[k], sizeof(int) * (size_t)ni)); HIPCHK(mem.alloc(&dep[k], sizeof(int) * (size_t)ni)); }
    HIPCHK(mem.alloc(&hist, sizeof(int) * (PT_MAX_TREE_LEVELS + 1)));
    hipLaunchKernelGGL(k_depth_init, dim3((ni + B - 1) / B), dim3(B), 0, stream, ni, parent, root, anc[0], dep[0]);
    int cur = 0;
    std::vector<int> h(PT_MAX_TREE_LEVELS + 1);
    for (int rounds = 6;; rounds = 3) { // 2^6 = 64 levels cover a Morton hierarchy; deeper ones take more rounds
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(k_depth_jump, dim3((ni + B - 1) / B), dim3(B), 0, stream, ni, anc[cur], dep[cur], anc[cur ^ 1], dep[cur ^ 1]);
            cur ^= 1;
        }
        HIPCHK(hipMemsetAsync(hist, 0, sizeof(int) * (PT_MAX_TREE_LEVELS + 1), stream));
        hipLaunchKernelGGL(k_depth_hist, dim3(std::min((ni + B - 1) / B, 1024)), dim3(B), 0, stream, ni, anc[cur], dep[cur], hist);
        HIPCHK(hipMemcpyAsync(h.data(), hist, sizeof(int) * (PT_MAX_TREE_LEVELS + 1), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (h[PT_MAX_TREE_LEVELS] == 0) break;
        int deepest = 0;
        for (int d = 0; d < PT_MAX_TREE_LEVELS; ++d) if (h[d]) deepest = d;
        if (deepest >= PT_MAX_TREE_LEVELS - 1) return hipSuccess; // too deep: lv->off stays empty
    }
    int levels = 0;
    for (int d = 0; d < PT_MAX_TREE_LEVELS; ++d) if (h[d]) levels = d + 1;
    std::vector<int> off(levels + 1, 0);
    for (int d = 0; d < levels; ++d) off[d + 1] = off[d] + h[d];
    if (off[levels] != ni) return hipErrorUnknown;
    HIPCHK(hipMemcpyAsync(hist, off.data(), sizeof(int) * (size_t)levels, hipMemcpyHostToDevice, stream)); // cursors
    hipLaunchKernelGGL(k_depth_scatter, dim3((ni + 1023) / 1024), dim3(1024), 0, stream, ni, dep[cur], hist, lv->order);
    HIPCHK(hipStreamSynchronize(stream)); // off.data() was read by the copy; mem is freed on return
    lv->off = off;
    return hipSuccess;
}

// wide BVH from the LBVH temporaries; nodes/tris are sized for the worst case and trimmed logically
static hipError_t build_bvh8(int n, int root, const int* left, const int* right, const int* cnt, const float* box, float pad,
                             const LeafTri* tris_sorted, hipStream_t stream, PtBvh* out, const LevelOrder* levels_in = nullptr) {
    Node8* nodes = nullptr;
    LeafTri* tris8 = nullptr;
    Task8 *ta = nullptr, *tb = nullptr;
    uint32_t* counters = nullptr;
    ArenaMark mark;
    const size_t max_nodes = (size_t)n + 1; // every wide node has >= 2 children, so < n nodes
    HIPCHK(tmalloc(&nodes, sizeof(Node8) * max_nodes)); // worst-case size while the tree is emitted; the context keeps an exact-size copy
    HIPCHK(hipMalloc(&tris8, sizeof(LeafTri) * (size_t)n));
    HIPCHK(tmalloc(&ta, sizeof(Task8) * max_nodes));
    HIPCHK(tmalloc(&tb, sizeof(Task8) * max_nodes));
    HIPCHK(tmalloc(&counters, sizeof(uint32_t) * 4));
    uint32_t hc[4] = {0u, 1u, 0u, 0u}; // node 0 = root is taken
    HIPCHK(hipMemcpyAsync(counters, hc, sizeof(hc), hipMemcpyHostToDevice, stream));
    // which binary nodes become wide nodes / leaf slots: SAH-optimal collapse (default) or greedy opening of the
    // largest child (PT_BVH_COLLAPSE=greedy)
    uint8_t* dec = nullptr;
    const char* collapse = getenv("PT_BVH_COLLAPSE");
    if (!(collapse && strcmp(collapse, "greedy") == 0)) {
        float cp = 1.0f; // cost of a triangle test relative to a wide-node step (PT_BVH_CP).  Rounds 1-4: 0.3; re-swept on the SAH hierarchy (round 5, profiles/r5_09_collapse_cp.log):
                         // stadium 12.14 ms at 0.3, 11.84-11.92 from 1.2 to 10, C3 and the textured terrain flat within 0.3 %
        if (const char* e = getenv("PT_BVH_CP")) cp = (float)atof(e);
        int *parent = nullptr, *visits = nullptr;
        float* cost = nullptr;
        HIPCHK(tmalloc(&parent, sizeof(int) * (size_t)(2 * n)));
        HIPCHK(tmalloc(&visits, sizeof(int) * (size_t)n));
        HIPCHK(tmalloc(&cost, sizeof(float) * 7 * (size_t)n));
        HIPCHK(tmalloc(&dec, 8 * (size_t)n));
        hipLaunchKernelGGL(k_parents, dim3((n + 255) / 256), dim3(256), 0, stream, n, left, right, root, parent);
        LevelOrder own;
        const LevelOrder* lv = levels_in;
        if (!lv) {
            HIPCHK(build_levels(n, parent, root, stream, &own));
            lv = &own;
        }
        if (!lv->off.empty()) { // one launch per level, deepest first
            for (int d = (int)lv->off.size() - 2; d >= 0; --d) {
                const int count = lv->off[d + 1] - lv->off[d];
                hipLaunchKernelGGL(k_collapse_cost_level, dim3((count + 127) / 128), dim3(128), 0, stream, lv->order + lv->off[d], count, n, left, right, cnt, box, pad, cp, cost, dec);
            }
        } else {
            HIPCHK(hipMemsetAsync(visits, 0, sizeof(int) * (size_t)n, stream));
            hipLaunchKernelGGL(k_collapse_cost, dim3((n + 255) / 256), dim3(256), 0, stream, n, left, right, parent, cnt, box, pad, cp, cost, dec, visits);
        }
        HIPCHK(hipStreamSynchronize(stream));
        own.release();
        if (getenv("PT_DEBUG_BVH")) {
            float rc = 0, rb[6];
            HIPCHK(hipMemcpy(&rc, cost + (size_t)root * 7, 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(rb, box + (size_t)root * 6, 24, hipMemcpyDeviceToHost));
            const float dx = rb[3] - rb[0] + 2 * pad, dy = rb[4] - rb[1] + 2 * pad, dz = rb[5] - rb[2] + 2 * pad;
            fprintf(stderr, "[pt_bvh] SAH cost of the wide tree (node steps + %.2f x triangle tests per random ray through the root box): %.3f\n", cp, rc / (dx * dy + dy * dz + dz * dx));
        }
        tfree(parent); tfree(visits); tfree(cost);
    }
    Task8 root_task{root, 0u};
    HIPCHK(hipMemcpyAsync(ta, &root_task, sizeof(root_task), hipMemcpyHostToDevice, stream));
    uint32_t nin = 1;
    int levels = 0;
    while (nin) {
        hipLaunchKernelGGL(k_collapse8, dim3((nin + 63) / 64), dim3(64), 0, stream, ta, nin, tb, counters, n, left, right, cnt, box, pad, dec,
                           tris_sorted, nodes, tris8, out->grid);
        HIPCHK(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        nin = hc[0];
        const uint32_t zero = 0;
        HIPCHK(hipMemcpyAsync(counters, &zero, 4, hipMemcpyHostToDevice, stream));
        Task8* t = ta; ta = tb; tb = t;
        if (++levels > 128) return hipErrorUnknown;
    }
    HIPCHK(hipGetLastError());
    if (getenv("PT_DEBUG_BVH")) { // slot occupancy of the wide tree
        auto fbits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        std::vector<Node80> h(hc[1]);
        HIPCHK(hipStreamSynchronize(stream)); // the library's streams do not synchronise with the null stream
#if PT8_NODE64
        {
            Node80* tmp80 = nullptr;
            HIPCHK(hipMalloc(&tmp80, sizeof(Node80) * h.size()));
            hipLaunchKernelGGL(k_nodes_to80, dim3((hc[1] + 255) / 256), dim3(256), 0, stream, nodes, hc[1], out->grid, tmp80);
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpy(h.data(), tmp80, sizeof(Node80) * h.size(), hipMemcpyDeviceToHost));
            hipFree(tmp80);
        }
#else
        HIPCHK(hipMemcpy(h.data(), nodes, sizeof(Node8) * h.size(), hipMemcpyDeviceToHost));
#endif
        unsigned long long fill[9] = {0}, leafsz[9] = {0}, nint = 0, nleaf = 0;
        for (const Node80& nd : h) {
            const uint32_t imask = fbits(nd.n1.w) >> 16;
            const uint32_t qlo[2] = {fbits(nd.n2.x), fbits(nd.n2.y)}, qhi[2] = {fbits(nd.n3.z), fbits(nd.n3.w)};
            const uint32_t leafbits = fbits(nd.n1.z);
            int used = 0;
            for (int s = 0; s < 8; ++s) {
                const uint32_t lo = (qlo[s >> 2] >> (8 * (s & 3))) & 0xffu, hi = (qhi[s >> 2] >> (8 * (s & 3))) & 0xffu;
                if (lo == 255u && hi == 0u) continue; // inverted = empty
                ++used;
                if (imask & (1u << s)) ++nint;
                else { ++nleaf; ++leafsz[__builtin_popcount((leafbits >> (3 * s)) & 7u)]; }
            }
            ++fill[used];
        }
        fprintf(stderr, "[pt_bvh] wide nodes %u, levels %d, internal children %llu, leaf slots %llu; slots used per node:", hc[1], levels, nint, nleaf);
        for (int k = 0; k <= 8; ++k) fprintf(stderr, " %d:%llu", k, fill[k]);
        fprintf(stderr, "; triangles per leaf slot:");
        for (int k = 1; k <= 8; ++k) if (leafsz[k]) fprintf(stderr, " %d:%llu", k, leafsz[k]);
        fprintf(stderr, "\n");
    }
    {
        Node8* exact = nullptr;
        HIPCHK(hipMalloc(&exact, sizeof(Node8) * (size_t)std::max(hc[1], 1u)));
        HIPCHK(hipMemcpyAsync(exact, nodes, sizeof(Node8) * (size_t)hc[1], hipMemcpyDeviceToDevice, stream));
        tfree(nodes);
        nodes = exact;
    }
    out->nodes8 = nodes;
    out->tris8 = tris8;
    out->num_nodes8 = hc[1];
    out->num_tris8 = hc[2];
    out->levels8 = levels;
    tfree(ta); tfree(tb); tfree(counters);
    if (dec) tfree(dec);
    return hipSuccess;
}

// PLOC hierarchy over the sorted leaves (leaf boxes at box[n-1+i] come from the refit pass); overwrites the
// internal-node arrays left/right/box/cnt (ids 0..n-2, root = the last one created)
static hipError_t build_ploc(int n, int* left, int* right, float* box, int* cnt, hipStream_t stream, int* root_out) {
    ArenaMark mark;
    DevFrees mem;
    int *cl_a = nullptr, *cl_b = nullptr, *nn = nullptr;
    uint32_t *merge = nullptr, *valid = nullptr, *merge_rank = nullptr, *valid_rank = nullptr;
    HIPCHK(mem.alloc(&cl_a, sizeof(int) * (size_t)n)); HIPCHK(mem.alloc(&cl_b, sizeof(int) * (size_t)n)); HIPCHK(mem.alloc(&nn, sizeof(int) * (size_t)n));
    HIPCHK(mem.alloc(&merge, 4 * (size_t)n)); HIPCHK(mem.alloc(&valid, 4 * (size_t)n));
    HIPCHK(mem.alloc(&merge_rank, 4 * (size_t)n)); HIPCHK(mem.alloc(&valid_rank, 4 * (size_t)n));
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, merge, merge_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    void* tmp = nullptr;
    HIPCHK(mem.alloc(&tmp, tmp_bytes));
    const int B = 256;
    uint32_t* d2 = nullptr;
    HIPCHK(mem.alloc(&d2, 16));
    hipLaunchKernelGGL(k_ploc_init, dim3((n + B - 1) / B), dim3(B), 0, stream, n, cl_a);
    int N = n, next_id = 0, iters = 0;
    const bool tail = getenv("PT_PLOC_TAIL") == nullptr || atoi(getenv("PT_PLOC_TAIL")) != 0;
    while (N > 1) {
        if (tail && N <= PLOC_TAIL) { // the rest in one workgroup
            hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(1024), 0, stream, cl_a, N, next_id, left, right, box, cnt, (int*)d2);
            int res[2] = {0, 1};
            HIPCHK(hipMemcpyAsync(res, d2, sizeof(res), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (res[1] != 0) return hipErrorUnknown;
            *root_out = res[0];
            return hipSuccess;
        }
        uint32_t tot[2] = {0, 0};
        int merged = 0;
        for (int tie_partner = 0; tie_partner < 2; ++tie_partner) {
            hipLaunchKernelGGL(k_ploc_nn, dim3((N + B - 1) / B), dim3(B), 0, stream, cl_a, N, box, tie_partner, nn);
            hipLaunchKernelGGL(k_ploc_flags, dim3((N + B - 1) / B), dim3(B), 0, stream, nn, N, merge, valid);
            HIPCHK(rocprim::exclusive_scan(tmp, tmp_bytes, merge, merge_rank, 0u, (size_t)N, rocprim::plus<uint32_t>(), stream));
            HIPCHK(rocprim::exclusive_scan(tmp, tmp_bytes, valid, valid_rank, 0u, (size_t)N, rocprim::plus<uint32_t>(), stream));
            hipLaunchKernelGGL(k_ploc_totals, dim3(1), dim3(64), 0, stream, merge, merge_rank, valid, valid_rank, N, d2);
            HIPCHK(hipMemcpyAsync(tot, d2, sizeof(tot), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            merged = (int)tot[0];
            if ((int64_t)merged * 64 >= N || N <= 64) break; // a healthy round merges a good part of the clusters; a starved one is repeated with the partner rule
        }
        hipLaunchKernelGGL(k_ploc_merge, dim3((N + B - 1) / B), dim3(B), 0, stream, cl_a, nn, N, merge, merge_rank, valid, valid_rank, next_id, left, right, box, cnt, cl_b);
        if (merged == 0 || ++iters > 4096) return hipErrorUnknown; // cannot happen: the global closest pair is always mutual
        next_id += merged;
        N = (int)tot[1];
        int* t = cl_a; cl_a = cl_b; cl_b = t;
    }
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(root_out, cl_a, sizeof(int), hipMemcpyDeviceToHost));
    return hipSuccess;
}

// ------------------------------------------------------------------ binned SAH, top-down (round 5: the third hierarchy the calibration chooses from)
// Wald 2007's binned surface-area heuristic, built breadth-first on the GPU.  The leaves are the Morton-sorted triangles (leaf i = node
// n - 1 + i, its box in box[]), prim[] holds them partitioned by node: a node owns the positions [first, first + count).
//   * Nodes with more than SAH_SMALL leaves ("large") are split level by level.  Every leaf drops its box into one of SAH_BINS bins per axis
//     (bins over the node's box, by centroid), always in LDS — a same-address global atomic costs 10.5 ns on this chip
//     (tools/micro/xw_probe.hip): a node of at most SAH_BIG leaves is binned by a wave (or the four waves of a workgroup) of its own
//     (k_sah_node); a bigger one by position windows of SAH_WG leaves (k_sah_bin), merged per node (k_sah_reduce) into one of the few global
//     bin sets there can be (at most n / SAH_BIG nodes are that big) which k_sah_node then loads.  The same wave then sweeps the
//     3 x (SAH_BINS - 1) candidate planes — one lane per plane, from the bins in LDS — for the smallest A_L N_L + A_R N_R (no candidate with
//     an empty side: the leaves are halved by position instead) and its first lane writes the decision and the sides' exact boxes (unions of
//     bins), 56 bytes, from which one thread per node creates the children (k_sah_children).  (Round 5 stored every node's bins to global memory — 1344 bytes per node, 150 MB for a million triangles — for a
//     kernel with one THREAD per node to sweep: 42 launches of 49 us, 2 of the hierarchy's 5.6 ms.)  The leaves are partitioned by one
//     exclusive scan of the "goes left" flags and a scatter.  One 20-byte read-back per level tells the host how many large nodes are left.
//   * A child with at most SAH_SMALL leaves is finished later by ONE thread: exact sweep SAH (every axis, every position of the sorted
//     centroids) down to single leaves.
// Node numbering is handed out by atomic counters (like k_collapse8's): the TREE is deterministic, the numbers are not.
// (Bins over the node's CENTROID bounds instead of its box — the textbook choice — and 32 bins were measured in round 5,
// profiles/r5_08_sah.md: the same frame times; that variant's code left with round 6, commit 910de7f has it.)
#ifndef SAH_BINS
#define SAH_BINS 16
#endif
#define SAH_SMALL 8
#define SAH_WG 1024
#define SAH_BIG 8192 // nodes with more leaves are binned by position windows (k_sah_bin + k_sah_reduce), smaller ones by a wave of their own
#define SAH_W 7
#define SAH_NODE_WORDS (3 * SAH_BINS * SAH_W)
struct SahBin { // per axis and bin: box of the leaves (ordered-uint floats) and their number
    uint32_t lo[3], hi[3], count;
};
// word w of a bin: 0-2 minima, 3-5 maxima, 6 count
__device__ __forceinline__ int sah_word_kind(int w) { return w < 3 ? 0 : (w < 6 ? 1 : 2); } // 0 min, 1 max, 2 sum
__device__ __forceinline__ uint32_t sah_word_init(int w) { return sah_word_kind(w) == 0 ? 0xffffffffu : 0u; }
struct SahCtl { // device counters of a build
    uint32_t next_id;     // next internal node id
    uint32_t nlarge_next; // large nodes created for the next level
    uint32_t nsmall;      // small subtrees waiting for k_sah_small
    uint32_t nbig_next;   // of the large nodes of the next level: those with more than SAH_BIG leaves (= global bin sets handed out)
    uint32_t fault;       // -DPT_BVH_CHECK=1 builds: bit 4 k + w = kernel k found index w outside its array (SAH_OK below); build_sah fails the build on any bit
};
// Bounds-checked builds of the SAH kernels (tools/variants.sh chk "-DPT_BVH_CHECK=1"; tests/test_gpu_builder.py runs the builder tests against that
// library): every index a kernel derives from device data — node ids out of active[] / node_of[] / the id counter, bin slots, leaf ranges
// first + count, scatter positions, the small-subtree sizes that index thread-local arrays — is compared with the size of the array it goes into
// BEFORE the access; a violation sets a bit of SahCtl::fault and the thread drops the access instead of making it.  Why (round 5 -> 6): a GPU
// memory access fault was seen once while these kernels were being rewritten (profiles/r6_01_sah_fault.md); the shipped kernels were audited
// index by index (the table there), and this build is what turns any future overflow into a reported error instead of a fault or silent
// corruption inside the arena.  K: 0 node, 1 bin (windows), 2 reduce, 3 children, 4 flag, 5 scatter, 6 small.
// W: 0 node id, 1 leaf range, 2 bin slot / scatter position, 3 leaf id / list position.
#ifndef PT_BVH_CHECK
#define PT_BVH_CHECK 0
#endif
#if PT_BVH_CHECK
#define SAH_OK(cond, K, W, ctl) ((cond) ? true : (atomicOr(&(ctl)->fault, 1u << (4 * (K) + (W))), false))
// fault injection, check builds only (PT_BVH_INJECT, tests/test_gpu_builder.py: the checks and the guard bands must be seen to fire):
//   1 = the root's second child is given a leaf count beyond the array (caught by k_sah_node's range check on the next level);
//   2 = k_sah_flag writes 96 words past the end of flag[] (lands in the arena's guard band behind that slice, PT_BVH_GUARD=1)
__device__ int g_sah_inject = 0;
#else
#define SAH_OK(cond, K, W, ctl) (true)
#endif
__device__ __forceinline__ int sah_bin_of(float c, float lo, float hi) {
    const float ext = hi - lo;
    if (!(ext > 0.f)) return 0;
    int b = (int)((c - lo) * ((float)SAH_BINS / ext));
    return b < 0 ? 0 : (b > SAH_BINS - 1 ? SAH_BINS - 1 : b);
}
__global__ void k_sah_init(int n, int* __restrict__ prim, int* __restrict__ node_of, int large_root) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    prim[i] = i;
    node_of[i] = large_root ? 0 : -1;
}
__global__ void k_sah_clear_bins(uint32_t* __restrict__ bins, uint32_t nwords) { // lo = +inf, hi = -inf (ordered), count = 0
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    bins[i] = sah_word_init((int)(i % (uint32_t)SAH_W));
}
// a leaf into one bin (LDS words): its box, its count
__device__ __forceinline__ void sah_accumulate(uint32_t* w, const float* b) {
    for (int k = 0; k < 3; ++k) {
        atomicMin(&w[k], f2ord(b[k]));
        atomicMax(&w[3 + k], f2ord(b[3 + k]));
    }
    atomicAdd(&w[6], 1u);
}
// The same for a whole wave whose lanes mostly fall into the SAME bin (the position windows of the top levels: the leaves are in Morton order,
// 64 neighbours share a bin, and 64 lanes on one LDS word serialise: 286 us per level for a million leaves): the lanes are grouped by bin with
// ballots, every group's boxes are reduced by a butterfly, and the group's first lane updates the bin with plain loads and stores — `w_base`
// must be bins only this wave touches.
__device__ __forceinline__ void sah_accumulate_wave(uint32_t* w_base, int bin, const float* b, bool valid) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int bsel = __shfl(bin, leader);
        const bool mine = valid && bin == bsel;
        const unsigned long long grp = __ballot(mine);
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = mine ? b[k] : INFINITY; hi[k] = mine ? b[3 + k] : -INFINITY; }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            }
        if (lane == leader) {
            uint32_t* w = w_base + bsel * SAH_W; // bsel is a leader's bin: sah_bin_of's result, in [0, SAH_BINS)
            for (int k = 0; k < 3; ++k) {
                w[k] = min(w[k], f2ord(lo[k]));
                w[3 + k] = max(w[3 + k], f2ord(hi[k]));
            }
            w[6] += (uint32_t)__popcll(grp);
        }
        todo &= ~grp;
    }
}
__device__ __forceinline__ float sah_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
// union of the bins [j0, j1] of one axis (bins without a leaf hold the sentinels and are skipped)
__device__ __forceinline__ int sah_union(const uint32_t* axis_bins, int j0, int j1, float* lo, float* hi) {
    int cn = 0;
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (int j = j0; j <= j1; ++j) {
        const uint32_t* b = axis_bins + j * SAH_W;
        if (!b[6]) continue;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], ord2f(b[k])); hi[k] = fmaxf(hi[k], ord2f(b[3 + k])); }
        cn += (int)b[6];
    }
    return cn;
}
// The split of a node whose bins sit in LDS (`s_bins`, written before a barrier by the caller): the wave's lanes take the 3 x (SAH_BINS - 1)
// candidate planes (axis a, leaves of bins 0..j left), each computes A_L N_L + A_R N_R from the unions of the bins on either side, and a
// butterfly picks the smallest cost, the lowest (axis, bin) among equal ones — what a sweep in (axis, bin) order with `c < best` picks.
// Returns axis | bin << 2 and the number of leaves that go left, or -1 when every plane has an empty side.  Every lane gets the result.
__device__ __forceinline__ int sah_sweep_wave(const uint32_t* s_bins, int lane, int* nl_out) {
    float best = INFINITY;
    int best_l = 0x7fffffff, best_nl = 0;
    for (int L = lane; L < 3 * (SAH_BINS - 1); L += 64) {
        const int a = L / (SAH_BINS - 1), j = L % (SAH_BINS - 1);
        float llo[3], lhi[3], rlo[3], rhi[3];
        const int cn = sah_union(s_bins + a * SAH_BINS * SAH_W, 0, j, llo, lhi);
        const int rn = sah_union(s_bins + a * SAH_BINS * SAH_W, j + 1, SAH_BINS - 1, rlo, rhi);
        if (cn == 0 || rn == 0) continue;
        const float c = sah_area(llo, lhi) * (float)cn + sah_area(rlo, rhi) * (float)rn;
        if (c < best) { best = c; best_l = L; best_nl = cn; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(best, off);
        const int ol = __shfl_xor(best_l, off), on = __shfl_xor(best_nl, off);
        if (oc < best || (oc == best && ol < best_l)) { best = oc; best_l = ol; best_nl = on; }
    }
    *nl_out = best_nl;
    if (best_l == 0x7fffffff) return -1;
    return (best_l / (SAH_BINS - 1)) | ((best_l % (SAH_BINS - 1)) << 2);
}
// What the sweep decided for a large node, written by the node's wave and read by k_sah_children (one THREAD per node): 56 bytes instead of the
// 1344 bytes of bins round 5's split kernel read per thread.  The children are NOT created by the wave itself: handing out node ids and list
// positions takes same-address atomics, the compiler folds those of a wave's lanes into one per wave, and one lane per wave doing them alone is
// 64 times as many (measured: the hierarchy 5.6 -> 10.7 ms with 680 k lone atomics at 10.5 ns).
struct SahDec {
    int sp, nl;           // axis | bin << 2 and the leaves going left; sp < 0: halved by position
    float blo[2][3], bhi[2][3]; // the sides' exact boxes (unions of bins)
};
__device__ __forceinline__ void sah_decide(int sp, int nl_best, const uint32_t* s_bins, SahDec* __restrict__ d) {
    d->sp = sp;
    d->nl = nl_best;
    if (sp < 0) return; // (halved by position: the children take the node's box, k_sah_children reads it)
    const uint32_t* ab = s_bins + (sp & 3) * SAH_BINS * SAH_W;
    float lo[3], hi[3];
    sah_union(ab, 0, sp >> 2, lo, hi);
    for (int k = 0; k < 3; ++k) { d->blo[0][k] = lo[k]; d->bhi[0][k] = hi[k]; }
    sah_union(ab, (sp >> 2) + 1, SAH_BINS - 1, lo, hi);
    for (int k = 0; k < 3; ++k) { d->blo[1][k] = lo[k]; d->bhi[1][k] = hi[k]; }
}
// one thread per large node of this level: the two children (ids, ranges, exact boxes), who is large next, who gets a global bin set
__global__ void k_sah_children(const int* __restrict__ active, int nactive, int n, const SahDec* __restrict__ dec, int* __restrict__ first, int* __restrict__ cnt,
                               float* __restrict__ box, int* __restrict__ left, int* __restrict__ right, int* __restrict__ split, int* __restrict__ slot_next,
                               int* __restrict__ active_next, int* __restrict__ small_list, SahCtl* __restrict__ ctl) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nactive) return;
    const int node = active[t];
    if (!SAH_OK(node >= 0 && node < n - 1, 3, 0, ctl)) return;
    const SahDec d = dec[t];
    const int m = cnt[node], f = first[node];
    if (!SAH_OK(f >= 0 && m > SAH_SMALL && (long long)f + m <= n && (d.sp < 0 || (d.nl > 0 && d.nl < m)), 3, 1, ctl)) return;
    int nl;
    float blo[2][3], bhi[2][3];
    if (d.sp < 0) { // every centroid in one bin on every axis: halve by position; the children's boxes are the node's (a superset: conservative)
        nl = m / 2;
        split[node] = -1 - nl;
        for (int k = 0; k < 3; ++k) { blo[0][k] = blo[1][k] = box[(size_t)node * 6 + k]; bhi[0][k] = bhi[1][k] = box[(size_t)node * 6 + 3 + k]; }
    } else {
        nl = d.nl;
        split[node] = d.sp;
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < 3; ++k) { blo[s][k] = d.blo[s][k]; bhi[s][k] = d.bhi[s][k]; }
    }
    for (int s = 0; s < 2; ++s) {
        const int cf = s ? f + nl : f, cm = s ? m - nl : nl;
        int child;
        if (cm == 1) {
            child = -2 - cf; // a single leaf: the scatter writes n - 1 + (the leaf that lands on position cf)
        } else {
            child = (int)atomicAdd(&ctl->next_id, 1u);
            if (!SAH_OK(child < n - 1, 3, 0, ctl)) return; // a binary tree over n leaves has n - 1 internal nodes: the id counter cannot pass it
            first[child] = cf;
            cnt[child] = cm;
#if PT_BVH_CHECK
            if (g_sah_inject == 1 && node == 0 && s == 1) cnt[child] = cm + n;
#endif
            for (int k = 0; k < 3; ++k) { box[(size_t)child * 6 + k] = blo[s][k]; box[(size_t)child * 6 + 3 + k] = bhi[s][k]; }
            if (cm > SAH_SMALL) {
                const int al = (int)atomicAdd(&ctl->nlarge_next, 1u);
                if (!SAH_OK(al < n / (SAH_SMALL + 1) + 2, 3, 3, ctl)) return; // active[]: a large node has more than SAH_SMALL leaves
                active_next[al] = child;
                if (cm > SAH_BIG) {
                    const int sl = (int)atomicAdd(&ctl->nbig_next, 1u);
                    if (!SAH_OK(sl < n / SAH_BIG + 2, 3, 2, ctl)) return; // global bin sets: big nodes are disjoint and hold more than SAH_BIG leaves each
                    slot_next[child] = sl;
                }
            } else {
                const uint32_t si = atomicAdd(&ctl->nsmall, 1u);
                if (!SAH_OK(si < (uint32_t)(n / 2 + 2), 3, 3, ctl)) return; // small_list: a small subtree has at least two leaves
                small_list[si] = child;
            }
        }
        (s ? right : left)[node] = child;
    }
}
// One wave per large node of this level (per_node = 1: levels with many nodes) or the four waves of a workgroup (per_node = 4: levels with few
// nodes: a lone wave looping over 8192 leaves is 128 dependent iterations, 0.3 ms): the node's bins in LDS that only these waves touch —
// binned from its leaves, or, for a node of more than SAH_BIG leaves, loaded from the global set the window kernels filled — then the plane
// sweep, from LDS, and the decision record for k_sah_children.  No global atomic on the bins and no bins in global memory for the 99.9 % of
// the nodes that are not big.
__global__ void __launch_bounds__(256) k_sah_node(const int* __restrict__ active, int nactive, int n, int nleaf_base, const int* __restrict__ prim, const int* __restrict__ first,
                                                  const int* __restrict__ cnt, const int* __restrict__ slot_of, const float* __restrict__ box, const uint32_t* __restrict__ bins,
                                                  SahDec* __restrict__ dec, SahCtl* __restrict__ ctl, int per_node) {
    __shared__ uint32_t s_all[4][SAH_NODE_WORDS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = per_node == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int part = per_node == 4 ? wave : 0;
    uint32_t* s_bins = s_all[wave];
    for (int k = lane; k < SAH_NODE_WORDS; k += 64) s_bins[k] = sah_word_init(k % SAH_W);
    __syncthreads();
    int node = t < nactive ? active[t] : -1;
    // (a dropped node keeps the workgroup's barriers matched: its waves bin nothing and create nothing)
    if (node >= 0 && !SAH_OK(node < n - 1, 0, 0, ctl)) node = -1;
    if (node >= 0 && !SAH_OK(first[node] >= 0 && cnt[node] > SAH_SMALL && (long long)first[node] + cnt[node] <= n, 0, 1, ctl)) node = -1;
    const int m = node >= 0 ? cnt[node] : 0;
    if (node >= 0 && m > SAH_BIG && !SAH_OK(slot_of[node] >= 0 && slot_of[node] < n / SAH_BIG + 2, 0, 2, ctl)) node = -1;
    if (node >= 0) {
        const int f = first[node];
        if (m > SAH_BIG) { // binned by the window kernels: the first wave fetches the node's global set (the others' LDS stays neutral)
            if (part == 0) {
                const uint32_t* g = bins + (size_t)slot_of[node] * SAH_NODE_WORDS;
                for (int k = lane; k < SAH_NODE_WORDS; k += 64) s_bins[k] = g[k];
            }
        } else {
            const float* nb = &box[(size_t)node * 6]; // what the bins span
            const float nlo[3] = {nb[0], nb[1], nb[2]}, nhi[3] = {nb[3], nb[4], nb[5]};
            for (int i = f + part * 64 + lane; i < f + m; i += 64 * per_node) {
                if (!SAH_OK(prim[i] >= 0 && prim[i] < n, 0, 3, ctl)) continue;
                const float* bp = &box[(size_t)(nleaf_base + prim[i]) * 6];
                const float b[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[5]};
                for (int a = 0; a < 3; ++a) sah_accumulate(&s_bins[(a * SAH_BINS + sah_bin_of(0.5f * (b[a] + b[3 + a]), nlo[a], nhi[a])) * SAH_W], b);
            }
        }
    }
    __syncthreads();
    if (per_node == 4) { // the four partial sets into the first (every thread reads and writes its own words only)
        for (int k = threadIdx.x; k < SAH_NODE_WORDS; k += 256) {
            const int kind = sah_word_kind(k % SAH_W);
            uint32_t acc = s_all[0][k];
            for (int v = 1; v < 4; ++v) acc = kind == 0 ? min(acc, s_all[v][k]) : (kind == 1 ? max(acc, s_all[v][k]) : acc + s_all[v][k]);
            s_all[0][k] = acc;
        }
        __syncthreads();
        if (wave != 0) return;
    }
    if (t >= nactive) return;
    int nl = 0;
    const int sp = node < 0 ? -1 : sah_sweep_wave(s_bins, lane, &nl); // (per_node = 4: wave 0's set is s_all[0], the merged one; a dropped node: halved, and k_sah_children drops it too)
    if (lane == 0) sah_decide(sp, nl, s_bins, &dec[t]);
}
// Big nodes (more than SAH_BIG leaves) by position windows of SAH_WG leaves.  A big node spans at least eight windows, so a window meets at most TWO of them
// (one ending, one starting): each gets a set of bins per wave in LDS (grouped accumulation, sah_accumulate_wave), the sets are merged per window and
// stored to partial[2 window + j] with plain stores, and k_sah_reduce merges the windows of a node.  No global atomic: the first versions sent the second
// node's leaves of a straddling window — a thousand leaves, 21 000 atomics on eleven cache lines, 10.5 ns each — to the global bins, and that one window
// made every level take 260 us whatever the other 1023 windows did (level 0, which has no straddling window: 42 us).
// Index ranges: window = blockIdx.x < ceil(n / SAH_WG) = the windows partial[] / partial_node[] are sized for (two entries each); s_first, s_second
// are thread indices < SAH_WG of threads with i < n.
__global__ void __launch_bounds__(SAH_WG) k_sah_bin(int n, int nleaf_base, const int* __restrict__ prim, const int* __restrict__ node_of, const int* __restrict__ cnt,
                                                    const float* __restrict__ box, uint32_t* __restrict__ partial, int* __restrict__ partial_node, SahCtl* __restrict__ ctl) {
    __shared__ uint32_t s_wbins[SAH_WG / 64][2][SAH_NODE_WORDS];
    __shared__ int s_first, s_second;
    const int i = blockIdx.x * SAH_WG + threadIdx.x;
    for (int k = threadIdx.x; k < (SAH_WG / 64) * 2 * SAH_NODE_WORDS; k += SAH_WG) (&s_wbins[0][0][0])[k] = sah_word_init(k % SAH_W);
    if (threadIdx.x == 0) { s_first = 0x7fffffff; s_second = 0x7fffffff; }
    __syncthreads();
    int node = i < n ? node_of[i] : -1;
    if (node >= 0 && !SAH_OK(node < n - 1, 1, 0, ctl)) node = -1;
    if (node >= 0 && !SAH_OK(prim[i] >= 0 && prim[i] < n, 1, 3, ctl)) node = -1;
    if (node >= 0 && cnt[node] <= SAH_BIG) node = -1; // a wave of its own bins that node (k_sah_node)
    if (node >= 0) atomicMin(&s_first, (int)threadIdx.x);
    __syncthreads();
    if (s_first == 0x7fffffff) { // no big node in this window
        if (threadIdx.x < 2) partial_node[2 * blockIdx.x + threadIdx.x] = -1;
        return;
    }
    const int node0 = node_of[blockIdx.x * SAH_WG + s_first];
    if (node >= 0 && node != node0) atomicMin(&s_second, (int)threadIdx.x);
    __syncthreads();
    const int node1 = s_second == 0x7fffffff ? -1 : node_of[blockIdx.x * SAH_WG + s_second];
    {
        float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float nlo[3] = {0.f, 0.f, 0.f}, nhi[3] = {0.f, 0.f, 0.f};
        if (node >= 0) {
            const float* bp = &box[(size_t)(nleaf_base + prim[i]) * 6];
            const float* nb = &box[(size_t)node * 6];
            for (int k = 0; k < 6; ++k) b[k] = bp[k];
            for (int k = 0; k < 3; ++k) { nlo[k] = nb[k]; nhi[k] = nb[3 + k]; }
        }
        const int wv = threadIdx.x >> 6;
        for (int a = 0; a < 3; ++a) { // (every lane of the wave runs the loop: the wave-level accumulation uses ballots and shuffles)
            const int bi = node >= 0 ? sah_bin_of(0.5f * (b[a] + b[3 + a]), nlo[a], nhi[a]) : 0;
            sah_accumulate_wave(&s_wbins[wv][0][a * SAH_BINS * SAH_W], bi, b, node >= 0 && node == node0);
            if (node1 >= 0) sah_accumulate_wave(&s_wbins[wv][1][a * SAH_BINS * SAH_W], bi, b, node >= 0 && node == node1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { partial_node[2 * blockIdx.x] = node0; partial_node[2 * blockIdx.x + 1] = node1; }
    for (int k = threadIdx.x; k < 2 * SAH_NODE_WORDS; k += SAH_WG) {
        const int j = k / SAH_NODE_WORDS, kk = k % SAH_NODE_WORDS;
        if (j == 1 && node1 < 0) continue;
        const int kind = sah_word_kind(kk % SAH_W);
        uint32_t acc = kind == 0 ? 0xffffffffu : 0u;
        for (int v = 0; v < SAH_WG / 64; ++v) {
            const uint32_t x = s_wbins[v][j][kk];
            acc = kind == 0 ? min(acc, x) : (kind == 1 ? max(acc, x) : acc + x);
        }
        partial[((size_t)2 * blockIdx.x + j) * SAH_NODE_WORDS + kk] = acc;
    }
}
// gridDim.y workgroups per large node (those of at most SAH_BIG leaves leave at once): merge the partial bins of the windows a big node covers
// into its (cleared) global set
__global__ void __launch_bounds__(256) k_sah_reduce(const int* __restrict__ active, int nactive, int n, const int* __restrict__ first, const int* __restrict__ cnt,
                                                    const int* __restrict__ slot_of, const uint32_t* __restrict__ partial, const int* __restrict__ partial_node,
                                                    uint32_t* __restrict__ bins, SahCtl* __restrict__ ctl) {
    const int node = active[blockIdx.x];
    if (!SAH_OK(node >= 0 && node < n - 1, 2, 0, ctl)) return; // (no barrier in this kernel)
    const int m = cnt[node];
    if (m <= SAH_BIG) return;
    const int f = first[node], w0 = f / SAH_WG, w1 = (f + m - 1) / SAH_WG;
    if (!SAH_OK(f >= 0 && (long long)f + m <= n, 2, 1, ctl)) return; // so w1 < ceil(n / SAH_WG), the windows partial[] holds
    if (!SAH_OK(slot_of[node] >= 0 && slot_of[node] < n / SAH_BIG + 2, 2, 2, ctl)) return;
    uint32_t* g = bins + (size_t)slot_of[node] * SAH_NODE_WORDS;
    for (int k = threadIdx.x; k < SAH_NODE_WORDS; k += 256) {
        const int kind = sah_word_kind(k % SAH_W);
        uint32_t acc = kind == 0 ? 0xffffffffu : 0u;
        for (int win = w0 + (int)blockIdx.y; win <= w1; win += (int)gridDim.y) // (the windows of a node are dealt to gridDim.y workgroups: one looping over a thousand windows took 0.6 ms)
            for (int j = 0; j < 2; ++j) {
                if (partial_node[2 * win + j] != node) continue;
                const uint32_t v = partial[((size_t)2 * win + j) * SAH_NODE_WORDS + k];
                acc = kind == 0 ? min(acc, v) : (kind == 1 ? max(acc, v) : acc + v);
            }
        if (kind == 0) { if (acc != 0xffffffffu) atomicMin(&g[k], acc); }
        else if (kind == 1) { if (acc != 0u) atomicMax(&g[k], acc); }
        else if (acc != 0u) atomicAdd(&g[k], acc);
    }
}
__global__ void k_sah_flag(int n, int nleaf_base, const int* __restrict__ prim, const int* __restrict__ node_of, const int* __restrict__ first, const int* __restrict__ split,
                           const float* __restrict__ box, uint32_t* __restrict__ flag, SahCtl* __restrict__ ctl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t f = 0u;
    if (i < n) {
        const int node = node_of[i];
        if (node >= 0 && SAH_OK(node < n - 1, 4, 0, ctl) && SAH_OK(prim[i] >= 0 && prim[i] < n, 4, 3, ctl)) {
            const int sp = split[node];
            if (sp < 0) {
                f = (i - first[node]) < (-1 - sp) ? 1u : 0u;
            } else {
                const int a = sp & 3;
                const float* b = &box[(size_t)(nleaf_base + prim[i]) * 6];
                const float* nb = &box[(size_t)node * 6];
                f = sah_bin_of(0.5f * (b[a] + b[3 + a]), nb[a], nb[3 + a]) <= (sp >> 2) ? 1u : 0u;
            }
        }
    }
    flag[i] = f; // flag[n] = 0: the scan's last entry is the total
#if PT_BVH_CHECK
    if (g_sah_inject == 2 && i == n)
        for (int k = 1; k <= 96; ++k) flag[n + k] = 0xdeadbeefu;
#endif
}
__global__ void k_sah_scatter(int n, int nleaf_base, const int* __restrict__ prim, const int* __restrict__ node_of, const int* __restrict__ first, const int* __restrict__ cnt_of,
                              const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scan, int* __restrict__ left, int* __restrict__ right,
                              int* __restrict__ prim_out, int* __restrict__ node_out, SahCtl* __restrict__ ctl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int node = node_of[i];
    if (node < 0) { // finished or small: stays where it is
        prim_out[i] = prim[i];
        node_out[i] = -1;
        return;
    }
    if (!SAH_OK(node < n - 1, 5, 0, ctl)) return;
    if (!SAH_OK(first[node] >= 0 && cnt_of[node] > 0 && (long long)first[node] + cnt_of[node] <= n && i >= first[node] && i < first[node] + cnt_of[node], 5, 1, ctl)) return;
    const int f = first[node], m = cnt_of[node];
    const int nl = (int)(scan[f + m] - scan[f]);
    const int rl = (int)(scan[i] - scan[f]);
    const bool goes_left = flag[i] != 0u;
    const int pos = goes_left ? f + rl : f + nl + ((i - f) - rl);
    const int p = prim[i];
    if (!SAH_OK(pos >= f && pos < f + m, 5, 2, ctl)) return; // the partition stays inside the node's own range
    prim_out[pos] = p;
    const int child = goes_left ? left[node] : right[node];
    if (!SAH_OK(child <= -2 || (child >= 0 && child < n - 1), 5, 3, ctl)) return;
    if (child <= -2) { // the child is this single leaf
        (goes_left ? left : right)[node] = nleaf_base + p;
        node_out[pos] = -1;
    } else {
        node_out[pos] = cnt_of[child] > SAH_SMALL ? child : -1;
    }
}
// one thread per small subtree (2 .. SAH_SMALL leaves): exact sweep SAH down to single leaves
__global__ void k_sah_small(const int* __restrict__ small_list, int nsmall, int nleaf_base, const int* __restrict__ prim, const int* __restrict__ first, int* __restrict__ cnt,
                            float* __restrict__ box, int* __restrict__ left, int* __restrict__ right, SahCtl* __restrict__ ctl) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsmall) return;
    const int root = small_list[t];
    if (!SAH_OK(root >= 0 && root < nleaf_base, 6, 0, ctl)) return; // nleaf_base = n - 1
    if (!SAH_OK(cnt[root] >= 2 && cnt[root] <= SAH_SMALL && first[root] >= 0 && first[root] + cnt[root] <= nleaf_base + 1, 6, 1, ctl)) return; // id[], lo[], hi[] hold SAH_SMALL leaves
    const int m0 = cnt[root], f0 = first[root];
    int id[SAH_SMALL];       // leaves of the subtree, reordered as the splits go
    float lo[SAH_SMALL][3], hi[SAH_SMALL][3];
    for (int k = 0; k < m0; ++k) {
        id[k] = prim[f0 + k];
        if (!SAH_OK(id[k] >= 0 && id[k] <= nleaf_base, 6, 3, ctl)) return;
        const float* b = &box[(size_t)(nleaf_base + id[k]) * 6];
        for (int a = 0; a < 3; ++a) { lo[k][a] = b[a]; hi[k][a] = b[3 + a]; }
    }
    int next = m0 > 2 ? (int)atomicAdd(&ctl->next_id, (uint32_t)(m0 - 2)) : 0; // the subtree has m0 - 1 internal nodes, `root` is one of them
    if (!SAH_OK(m0 <= 2 || next + (m0 - 2) <= nleaf_base, 6, 2, ctl)) return;
    int st_node[SAH_SMALL], st_f[SAH_SMALL], st_m[SAH_SMALL], sp = 0;
    st_node[0] = root; st_f[0] = 0; st_m[0] = m0; sp = 1;
    while (sp) {
        --sp;
        const int node = st_node[sp], f = st_f[sp], m = st_m[sp];
        // node box
        float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = f; k < f + m; ++k)
            for (int a = 0; a < 3; ++a) { nlo[a] = fminf(nlo[a], lo[k][a]); nhi[a] = fmaxf(nhi[a], hi[k][a]); }
        for (int a = 0; a < 3; ++a) { box[(size_t)node * 6 + a] = nlo[a]; box[(size_t)node * 6 + 3 + a] = nhi[a]; }
        cnt[node] = m;
        int nl = m / 2;
        if (m > 2) {
            float best = INFINITY;
            int best_axis = 0;
            for (int a = 0; a < 3; ++a) {
                // insertion sort of [f, f+m) by centroid along a (keeps id / lo / hi together)
                for (int i = f + 1; i < f + m; ++i)
                    for (int j = i; j > f && (lo[j][a] + hi[j][a]) < (lo[j - 1][a] + hi[j - 1][a]); --j) {
                        const int ti = id[j]; id[j] = id[j - 1]; id[j - 1] = ti;
                        for (int k = 0; k < 3; ++k) {
                            const float tl = lo[j][k]; lo[j][k] = lo[j - 1][k]; lo[j - 1][k] = tl;
                            const float th = hi[j][k]; hi[j][k] = hi[j - 1][k]; hi[j - 1][k] = th;
                        }
                    }
                float ra[SAH_SMALL];
                float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int k = f + m - 1; k > f; --k) {
                    for (int c = 0; c < 3; ++c) { slo[c] = fminf(slo[c], lo[k][c]); shi[c] = fmaxf(shi[c], hi[k][c]); }
                    ra[k - f] = sah_area(slo, shi);
                }
                for (int c = 0; c < 3; ++c) { slo[c] = INFINITY; shi[c] = -INFINITY; }
                for (int k = f; k < f + m - 1; ++k) {
                    for (int c = 0; c < 3; ++c) { slo[c] = fminf(slo[c], lo[k][c]); shi[c] = fmaxf(shi[c], hi[k][c]); }
                    const int cl = k - f + 1;
                    const float c = sah_area(slo, shi) * (float)cl + ra[cl] * (float)(m - cl);
                    if (c < best) { best = c; best_axis = a; nl = cl; }
                }
            }
            if (best_axis != 2) // the leaves are sorted along axis 2 now: put them back in the best axis' order
                for (int i = f + 1; i < f + m; ++i)
                    for (int j = i; j > f && (lo[j][best_axis] + hi[j][best_axis]) < (lo[j - 1][best_axis] + hi[j - 1][best_axis]); --j) {
                        const int ti = id[j]; id[j] = id[j - 1]; id[j - 1] = ti;
                        for (int k = 0; k < 3; ++k) {
                            const float tl = lo[j][k]; lo[j][k] = lo[j - 1][k]; lo[j - 1][k] = tl;
                            const float th = hi[j][k]; hi[j][k] = hi[j - 1][k]; hi[j - 1][k] = th;
                        }
                    }
        }
        for (int s = 0; s < 2; ++s) {
            const int cf = s ? f + nl : f, cm = s ? m - nl : nl;
            int child;
            if (cm == 1) {
                child = nleaf_base + id[cf];
            } else {
                child = next++;
                st_node[sp] = child; st_f[sp] = cf; st_m[sp] = cm; ++sp;
            }
            (s ? right : left)[node] = child;
        }
    }
}

static hipError_t build_sah(int n, int* left, int* right, float* box, int* cnt, const float bounds[6], hipStream_t stream, int* root_out) {
    ArenaMark mark;
    DevFrees mem;
    const int nleaf_base = n - 1, B = 256;
    int *prim[2] = {nullptr, nullptr}, *node_of[2] = {nullptr, nullptr}, *first = nullptr, *split = nullptr, *slot[2] = {nullptr, nullptr}, *active[2] = {nullptr, nullptr}, *small_list = nullptr;
    uint32_t *flag = nullptr, *scan = nullptr;
    SahCtl* ctl = nullptr;
    // Sizes, and the largest index each array is accessed with (what -DPT_BVH_CHECK=1 verifies on the device; profiles/r6_01_sah_fault.md):
    //   prim, node_of: n, by leaf position < n | slot, first, split: n, by internal node id < n - 1 | active: n / (SAH_SMALL + 1) + 2, by the
    //   level's large-node counter (large nodes are disjoint and hold more than SAH_SMALL leaves) | small_list: n / 2 + 2, by the small-subtree
    //   counter (at least two leaves each) | flag, scan: n + 1, by position <= n | partial: 2 sets per window of SAH_WG positions |
    //   bins: n / SAH_BIG + 2 sets, by the big-node counter (big nodes are disjoint and hold more than SAH_BIG leaves)
    const size_t max_large = (size_t)n / (SAH_SMALL + 1) + 2, max_big = (size_t)n / SAH_BIG + 2;
    for (int k = 0; k < 2; ++k) {
        HIPCHK(mem.alloc(&prim[k], sizeof(int) * (size_t)n));
        HIPCHK(mem.alloc(&node_of[k], sizeof(int) * (size_t)n));
        HIPCHK(mem.alloc(&slot[k], sizeof(int) * (size_t)n));
        HIPCHK(mem.alloc(&active[k], sizeof(int) * max_large));
    }
    HIPCHK(mem.alloc(&first, sizeof(int) * (size_t)n));
    HIPCHK(mem.alloc(&split, sizeof(int) * (size_t)n));
    HIPCHK(mem.alloc(&small_list, sizeof(int) * (size_t)(n / 2 + 2)));
    HIPCHK(mem.alloc(&flag, sizeof(uint32_t) * ((size_t)n + 1)));
    HIPCHK(mem.alloc(&scan, sizeof(uint32_t) * ((size_t)n + 1)));
    HIPCHK(mem.alloc(&ctl, sizeof(SahCtl)));
    uint32_t* partial = nullptr; // per position window: the merged LDS bins of the (at most two) big nodes the window meets, and which nodes those were
    int* partial_node = nullptr;
    const size_t nwin = ((size_t)n + SAH_WG - 1) / SAH_WG;
    HIPCHK(mem.alloc(&partial, sizeof(uint32_t) * 2 * SAH_NODE_WORDS * nwin));
    HIPCHK(mem.alloc(&partial_node, sizeof(int) * 2 * nwin));
    uint32_t* bins = nullptr; // global bin sets of the big nodes of one level (1344 bytes each; every other node's bins never leave LDS)
    HIPCHK(mem.alloc(&bins, sizeof(uint32_t) * SAH_NODE_WORDS * max_big));
    SahDec* dec = nullptr; // one decision per large node of the level, by its position in active[]
    HIPCHK(mem.alloc(&dec, sizeof(SahDec) * max_large));
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, scan, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
    void* tmp = nullptr;
    HIPCHK(mem.alloc(&tmp, tmp_bytes ? tmp_bytes : 16));
#if PT_BVH_CHECK
    {
        const char* ie = getenv("PT_BVH_INJECT");
        const int inject = ie ? atoi(ie) : 0;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_sah_inject), &inject, sizeof(int)));
    }
#endif
    // root = internal node 0 over all leaves, its box = the scene's bounds
    const bool large_root = n > SAH_SMALL;
    SahCtl h{1u, 0u, large_root ? 0u : 1u, 0u, 0u};
    HIPCHK(hipMemcpyAsync(ctl, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    const int zero = 0;
    HIPCHK(hipMemcpyAsync(first, &zero, sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(cnt, &n, sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(box, bounds, sizeof(float) * 6, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(slot[0], &zero, sizeof(int), hipMemcpyHostToDevice, stream)); // (the root's bin set, if it is big)
    HIPCHK(hipMemcpyAsync(large_root ? active[0] : small_list, &zero, sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_sah_init, dim3((n + B - 1) / B), dim3(B), 0, stream, n, prim[0], node_of[0], large_root ? 1 : 0);
    int cur = 0, nactive = large_root ? 1 : 0, nbig = n > SAH_BIG ? 1 : 0, levels = 0;
    while (nactive > 0) {
        if (nbig > 0) {
            const uint32_t nwords = (uint32_t)((size_t)nbig * SAH_NODE_WORDS); // (nbig <= max_big, checked below before the next level is launched)
            hipLaunchKernelGGL(k_sah_clear_bins, dim3((nwords + B - 1) / B), dim3(B), 0, stream, bins, nwords);
            hipLaunchKernelGGL(k_sah_bin, dim3((n + SAH_WG - 1) / SAH_WG), dim3(SAH_WG), 0, stream, n, nleaf_base, prim[cur], node_of[cur], cnt, box, partial, partial_node, ctl);
            hipLaunchKernelGGL(k_sah_reduce, dim3(nactive, 32), dim3(256), 0, stream, active[cur], nactive, n, first, cnt, slot[cur], partial, partial_node, bins, ctl);
        }
        {
            const int per_node = nactive <= 4096 ? 4 : 1;
            hipLaunchKernelGGL(k_sah_node, dim3(per_node == 4 ? nactive : (nactive + 3) / 4), dim3(256), 0, stream, active[cur], nactive, n, nleaf_base, prim[cur], first, cnt, slot[cur], box,
                               bins, dec, ctl, per_node);
            hipLaunchKernelGGL(k_sah_children, dim3((nactive + 63) / 64), dim3(64), 0, stream, active[cur], nactive, n, dec, first, cnt, box, left, right, split, slot[cur ^ 1], active[cur ^ 1],
                               small_list, ctl);
        }
        hipLaunchKernelGGL(k_sah_flag, dim3((n + 1 + B - 1) / B), dim3(B), 0, stream, n, nleaf_base, prim[cur], node_of[cur], first, split, box, flag, ctl);
        HIPCHK(rocprim::exclusive_scan(tmp, tmp_bytes, flag, scan, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_sah_scatter, dim3((n + B - 1) / B), dim3(B), 0, stream, n, nleaf_base, prim[cur], node_of[cur], first, cnt, flag, scan, left, right,
                           prim[cur ^ 1], node_of[cur ^ 1], ctl);
        HIPCHK(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        nactive = (int)h.nlarge_next;
        nbig = (int)h.nbig_next;
        // host-side bounds of what the next level indexes with these counters (the kernels that incremented them wrote at most that far:
        // k_sah_children's capacity arguments; in ordinary builds this is where a broken invariant surfaces, before anything reads the lists)
        if (h.fault) {
            fprintf(stderr, "[pt_bvh] binned SAH: bounds check failed, fault bits 0x%x (kernel = bit / 4, index = bit %% 4; pt_bvh_build.hip SAH_OK)\n", h.fault);
            return hipErrorUnknown;
        }
        if ((size_t)nactive > max_large || (size_t)nbig > max_big || (size_t)h.nsmall > (size_t)n / 2 + 2 || h.next_id > (uint32_t)(n - 1) || ++levels > 4096) return hipErrorUnknown;
        const uint32_t z = 0;
        HIPCHK(hipMemcpyAsync(&ctl->nlarge_next, &z, sizeof(z), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(&ctl->nbig_next, &z, sizeof(z), hipMemcpyHostToDevice, stream));
        cur ^= 1;
    }
    if (h.nsmall) hipLaunchKernelGGL(k_sah_small, dim3((h.nsmall + 63) / 64), dim3(64), 0, stream, small_list, (int)h.nsmall, nleaf_base, prim[cur], first, cnt, box, left, right, ctl);
    HIPCHK(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    if (h.fault) {
        fprintf(stderr, "[pt_bvh] binned SAH: bounds check failed in the small-subtree kernel, fault bits 0x%x\n", h.fault);
        return hipErrorUnknown;
    }
    if (h.next_id != (uint32_t)(n - 1)) return hipErrorUnknown; // a binary tree over n leaves has n - 1 internal nodes
    if (getenv("PT_DEBUG_BVH")) fprintf(stderr, "[pt_bvh] binned SAH: %d levels of large nodes, %u small subtrees\n", levels, h.nsmall);
    *root_out = 0;
    return hipSuccess;
}

// Experiment hook (PT_BVH_IMPORT=file): a binary hierarchy built elsewhere over the same triangles replaces the internal nodes.  File: int32 n,
// then n-1 pairs (left, right) of int32, node 0 = root, a child >= 0 is an internal node, a child < 0 is primitive ~child.
static hipError_t import_hierarchy(const char* path, int n, const uint64_t* keys_sorted, int* left, int* right, float* box, int* cnt, hipStream_t stream, int* root_out) {
    FILE* f = fopen(path, "rb");
    if (!f) return hipErrorInvalidValue;
    int nf = 0;
    if (fread(&nf, 4, 1, f) != 1 || nf != n) { fclose(f); return hipErrorInvalidValue; }
    std::vector<int> lr((size_t)2 * (n - 1));
    if (fread(lr.data(), 4, lr.size(), f) != lr.size()) { fclose(f); return hipErrorInvalidValue; }
    fclose(f);
    std::vector<uint64_t> keys((size_t)n);
    std::vector<float> hbox((size_t)12 * n);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(keys.data(), keys_sorted, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hbox.data(), box, sizeof(float) * 12 * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<int> pos((size_t)n), hl((size_t)n), hr((size_t)n), hc((size_t)2 * n, 1);
    for (int i = 0; i < n; ++i) {
        if ((uint32_t)keys[i] >= (uint32_t)n) return hipErrorInvalidValue;
        pos[(uint32_t)keys[i]] = i;
    }
    for (int c : lr) // the file is untrusted: an internal child must be a node id, a leaf child a primitive
        if (c >= 0 ? c >= n - 1 : ~c >= n) return hipErrorInvalidValue;
    auto tr = [&](int c) { return c >= 0 ? c : n - 1 + pos[~c]; };
    for (int j = 0; j < n - 1; ++j) { hl[j] = tr(lr[2 * j]); hr[j] = tr(lr[2 * j + 1]); }
    // ... and it must be a TREE over exactly these nodes: every internal node but the root and every leaf referenced once, the root never.
    // (The level-synchronous passes read parent[] of every internal node: a node no parent names would leave uninitialised arena memory
    // there, and one named twice would be climbed twice.)
    {
        std::vector<uint8_t> refs((size_t)2 * n - 1, 0);
        for (int j = 0; j < n - 1; ++j)
            for (int c : {hl[j], hr[j]})
                if (c == 0 || refs[(size_t)c]++ != 0) return hipErrorInvalidValue;
        for (size_t c = 1; c < refs.size(); ++c)
            if (refs[c] != 1) return hipErrorInvalidValue;
    }
    // children have larger ids than their parent (preorder): one backward sweep computes boxes and counts
    for (int j = n - 2; j >= 0; --j) {
        const int a = hl[j], b = hr[j];
        if ((a < n - 1 && a <= j) || (b < n - 1 && b <= j)) return hipErrorInvalidValue;
        for (int k = 0; k < 3; ++k) {
            hbox[(size_t)j * 6 + k] = fminf(hbox[(size_t)a * 6 + k], hbox[(size_t)b * 6 + k]);
            hbox[(size_t)j * 6 + 3 + k] = fmaxf(hbox[(size_t)a * 6 + 3 + k], hbox[(size_t)b * 6 + 3 + k]);
        }
        hc[j] = hc[a] + hc[b];
    }
    HIPCHK(hipMemcpy(left, hl.data(), sizeof(int) * (size_t)(n - 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(right, hr.data(), sizeof(int) * (size_t)(n - 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(box, hbox.data(), sizeof(float) * 6 * (size_t)(n - 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(cnt, hc.data(), sizeof(int) * (size_t)(2 * n - 1), hipMemcpyHostToDevice));
    *root_out = 0;
    return hipSuccess;
}

// Builds the traversal structure for (verts, idx) already resident on the device.
// PT_DEBUG_BVH: host time of every phase of the build (each closed by a stream synchronisation, so the sum is above the unprofiled total)
struct PhaseClock {
    hipStream_t stream;
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseClock(hipStream_t s) : stream(s), on(getenv("PT_DEBUG_BVH") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pt_bvh] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

hipError_t pt_bvh_build(const float* d_verts, const uint32_t* d_idx, const uint32_t* d_tri_mesh, uint32_t ntri, hipStream_t stream, PtBvh* out) {
    if (PT8_NODE64 && ntri > (1u << 24)) return hipErrorInvalidValue; // one-line nodes address 2^24 nodes and leaf triangles
    const int n = (int)ntri;
    const int B = 256;
    PhaseClock pc(stream);
    Arena arena;
    struct ArenaScope { ~ArenaScope() { t_arena = nullptr; } } arena_scope;
    if (n >= 4096 && arena.init((size_t)n * 300 + (8u << 20)) == hipSuccess) t_arena = &arena; // else every temporary is its own allocation
    else (void)hipGetLastError();
    pc.mark("arena");
    // leaf triangles + keys
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    uint32_t* bounds = nullptr;
    HIPCHK(tmalloc(&keys, sizeof(uint64_t) * (size_t)n));
    HIPCHK(tmalloc(&keys_sorted, sizeof(uint64_t) * (size_t)n));
    HIPCHK(tmalloc(&bounds, sizeof(uint32_t) * 6));
    uint32_t binit[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIPCHK(hipMemcpyAsync(bounds, binit, sizeof(binit), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_bounds, dim3(min((n + B - 1) / B, 2048)), dim3(B), 0, stream, d_verts, d_idx, ntri, bounds);
    hipLaunchKernelGGL(k_morton, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, ntri, bounds, keys);
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_sorted, (size_t)n, 0, 64, stream));
    void* tmp = nullptr;
    HIPCHK(tmalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    HIPCHK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys_sorted, (size_t)n, 0, 64, stream));
    uint32_t hb[6];
    HIPCHK(hipMemcpyAsync(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float maxabs = 0.f;
    for (int a = 0; a < 6; ++a) {
        uint32_t u = hb[a];
        uint32_t bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        float f;
        memcpy(&f, &bits, 4);
        out->bounds[a] = f;
        maxabs = fmaxf(maxabs, fabsf(f));
    }
    const float pad = maxabs * (1.0f / 65536.0f);
    out->pad = pad;
#if PT8_NODE64
    {
        // the one-line nodes' origin grid: over the scene's bounds widened by twice the padding (every node box lies inside), 2^14 / 2^13 / 2^14
        // cells; base exponent: 255 steps of 2^(ebase + 31 - 127) span twice the largest extent, so the root always finds its step
        float maxext = 0.f;
        for (int a = 0; a < 3; ++a) {
            const float glo = out->bounds[a] - 2.0f * pad, ghi = out->bounds[3 + a] + 2.0f * pad;
            const float cells = (float)(1u << (a == 1 ? PT8_GRID_BITS_Y : PT8_GRID_BITS_XZ));
            float gs = (ghi - glo) / cells * (1.0f + 1.0f / 262144.0f);
            if (!(gs > 1e-37f)) gs = 1e-37f;
            out->grid.glo[a] = glo;
            out->grid.gstep[a] = gs;
            maxext = fmaxf(maxext, ghi - glo);
        }
        int e = 0;
        if (maxext > 0.f) frexpf(2.0f * maxext / 255.0f, &e); // 2^(e-1) <= x < 2^e
        int eb = e + 127 - 31;
        out->grid.ebase = (uint32_t)std::max(1, std::min(223, eb));
    }
#endif
    pc.mark("bounds, morton, sort");

    LeafTri* tris = nullptr;
    HIPCHK(tmalloc(&tris, sizeof(LeafTri) * (size_t)n));
    hipLaunchKernelGGL(k_emit_tris, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, d_tri_mesh, keys_sorted, n, tris);

    if (n <= PT8_LEAF_MAX) { // the whole scene is one leaf child of one node
        Node8* nodes8 = nullptr;
        LeafTri* tris8 = nullptr;
        float* dbounds = nullptr;
        HIPCHK(hipMalloc(&nodes8, sizeof(Node8)));
        HIPCHK(hipMalloc(&tris8, sizeof(LeafTri) * (size_t)n));
        HIPCHK(tmalloc(&dbounds, sizeof(float) * 6));
        HIPCHK(hipMemcpyAsync(dbounds, out->bounds, sizeof(float) * 6, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(tris8, tris, sizeof(LeafTri) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_single_node8, dim3(1), dim3(64), 0, stream, n, dbounds, pad, nodes8, out->grid);
        out->nodes8 = nodes8; out->tris8 = tris8; out->num_nodes8 = 1; out->num_tris8 = (uint32_t)n; out->levels8 = 1;
        HIPCHK(hipStreamSynchronize(stream));
        tfree(dbounds);
        tfree(tris);
        tfree(keys); tfree(keys_sorted); tfree(bounds); tfree(tmp);
        return hipSuccess;
    }

    int *left, *right, *parent, *rfirst, *rlast, *visits;
    float* box;
    HIPCHK(tmalloc(&left, sizeof(int) * (size_t)n));
    HIPCHK(tmalloc(&right, sizeof(int) * (size_t)n));
    HIPCHK(tmalloc(&parent, sizeof(int) * (size_t)(2 * n)));
    HIPCHK(tmalloc(&rfirst, sizeof(int) * (size_t)n));
    HIPCHK(tmalloc(&rlast, sizeof(int) * (size_t)n));
    HIPCHK(tmalloc(&visits, sizeof(int) * (size_t)n));
    HIPCHK(tmalloc(&box, sizeof(float) * 6 * (size_t)(2 * n)));
    hipLaunchKernelGGL(k_karras, dim3((n + B - 1) / B), dim3(B), 0, stream, keys_sorted, n, left, right, parent, rfirst, rlast);
    pc.mark("emit tris, allocs");
    LevelOrder lbvh_levels; // of the Karras hierarchy: the refit and the LBVH's collapse costs both run level by level over it
    HIPCHK(build_levels(n, parent, 0, stream, &lbvh_levels));
    if (!lbvh_levels.off.empty()) {
        hipLaunchKernelGGL(k_leaf_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, keys_sorted, n, box);
        for (int d = (int)lbvh_levels.off.size() - 2; d >= 0; --d) {
            const int count = lbvh_levels.off[d + 1] - lbvh_levels.off[d];
            hipLaunchKernelGGL(k_refit_level, dim3((count + B - 1) / B), dim3(B), 0, stream, lbvh_levels.order + lbvh_levels.off[d], count, left, right, box);
        }
    } else {
        HIPCHK(hipMemsetAsync(visits, 0, sizeof(int) * (size_t)n, stream));
        hipLaunchKernelGGL(k_refit, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, keys_sorted, n, left, right, parent, box, visits);
    }
    pc.mark("karras, levels, refit");
    {
        // hierarchy for the wide tree: the LBVH itself (default) or PLOC (PT_BVH_BUILDER=ploc).  Measured on the C3 voxel
        // terrain: PLOC gives MORE node visits per ray (13.3 vs 12.5 primary, 15.9 vs 13.6 diffuse bounce) and 5 % lower
        // Mrays/s — on a scene of uniform, evenly spread quads the Morton split is already near the SAH optimum and
        // balances the 8-wide collapse better; PLOC stays selectable for irregular scenes.
        int* cnt = nullptr;
        HIPCHK(tmalloc(&cnt, sizeof(int) * (size_t)(2 * n)));
        hipLaunchKernelGGL(k_counts_from_ranges, dim3((2 * n + B - 1) / B), dim3(B), 0, stream, n, rfirst, rlast, cnt);
        int root = 0;
        const char* builder = getenv("PT_BVH_BUILDER");
        const bool force_lbvh = builder && strcmp(builder, "lbvh") == 0, force_ploc = builder && strcmp(builder, "ploc") == 0, force_sah = builder && strcmp(builder, "sah") == 0;
        if (force_ploc) HIPCHK(build_ploc(n, left, right, box, cnt, stream, &root));
        if (force_sah) HIPCHK(build_sah(n, left, right, box, cnt, out->bounds, stream, &root));
        const char* import = getenv("PT_BVH_IMPORT");
        if (import) HIPCHK(import_hierarchy(import, n, keys_sorted, left, right, box, cnt, stream, &root));
        HIPCHK(build_bvh8(n, root, left, right, cnt, box, pad, tris, stream, out, (force_ploc || force_sah || import) ? nullptr : &lbvh_levels));
        lbvh_levels.release();
        pc.mark("wide tree 1");
        out->builder = import ? 2 : force_sah ? 3 : force_ploc ? 1 : 0;
        if (import) { tfree(cnt); goto done; }
        if (!force_lbvh && !force_ploc && !force_sah && n >= 4096) {
            // The LBVH (already emitted) against the binned-SAH hierarchy, the one that costs the calibration rays less (see k_calibrate8); small scenes
            // keep the LBVH.  A challenger overwrites the binary hierarchy's internal nodes (the standing wide tree is already emitted), is collapsed into
            // a wide tree of its own and traced against the standing tree by one calibration launch.  PLOC — the challenger of rounds 3-4 — never
            // rendered faster than the SAH tree (stadium 12.40 against 12.09 ms, terrain 8.74 against 7.92) and the calibration cannot tell two
            // trees 4 % apart, so it is a candidate only on request (PT_BVH_PLOC=1; PT_BVH_SAH=0 gives the round-4 pair LBVH | PLOC).
            static const char* const names[4] = {"LBVH", "PLOC", "imported", "SAH"};
            const char* cm = getenv("PT_BVH_CALIB"); // segments: centroid-to-centroid segments (rounds 3-4); default: rays like a frame's (k_calibrate8)
            const int calib_mode = (cm && strcmp(cm, "segments") == 0) ? 0 : 1;
            const float* bd = out->bounds;
            const float4 sphere = make_float4(0.5f * (bd[0] + bd[3]), 0.5f * (bd[1] + bd[4]), 0.5f * (bd[2] + bd[5]),
                                              0.5f * sqrtf((bd[3] - bd[0]) * (bd[3] - bd[0]) + (bd[4] - bd[1]) * (bd[4] - bd[1]) + (bd[5] - bd[2]) * (bd[5] - bd[2])));
            const char *sah_env = getenv("PT_BVH_SAH"), *ploc_env = getenv("PT_BVH_PLOC");
            const bool with_sah = !(sah_env && atoi(sah_env) == 0), with_ploc = !with_sah || (ploc_env && atoi(ploc_env) != 0);
            int kinds[2], ncand = 0;
            if (with_ploc) kinds[ncand++] = 1;
            if (with_sah) kinds[ncand++] = 3;
            for (int cand = 0; cand < ncand; ++cand) {
                const int kind = kinds[cand];
                PtBvh alt;
                alt.grid = out->grid; // same scene, same origin grid
                hipError_t pe = kind == 1 ? build_ploc(n, left, right, box, cnt, stream, &root) : build_sah(n, left, right, box, cnt, out->bounds, stream, &root);
                pc.mark(kind == 1 ? "ploc hierarchy" : "sah hierarchy");
                if (pe == hipSuccess) pe = build_bvh8(n, root, left, right, cnt, box, pad, tris, stream, &alt);
                pc.mark(kind == 1 ? "wide tree 2" : "wide tree 3");
                if (pe != hipSuccess) { // a challenger is optional: the standing tree stays — but never silently (the choice of tree then depended on free memory)
                    (void)hipGetLastError();
                    pt_bvh_free(&alt);
                    ++out->challengers_skipped;
                    fprintf(stderr, "[pt_bvh] warning: the %s hierarchy could not be built (%s): the %s tree stays without a comparison\n", names[kind], hipGetErrorString(pe), names[out->builder]);
                    continue;
                }
                unsigned long long* counts = nullptr;
                unsigned long long hcnt[4] = {0, 0, 0, 0};
                HIPCHK(tmalloc(&counts, sizeof(hcnt)));
                HIPCHK(hipMemsetAsync(counts, 0, sizeof(hcnt), stream));
                const uint32_t nrays = 1u << 16;
                hipLaunchKernelGGL(k_calibrate8, dim3(2 * (nrays / 64)), dim3(64), 0, stream, out->nodes8, out->tris8, alt.nodes8, alt.tris8, tris, (uint32_t)n, nrays, 0.5f * pad, counts, out->grid, calib_mode, sphere);
                HIPCHK(hipMemcpyAsync(hcnt, counts, sizeof(hcnt), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                tfree(counts);
                pc.mark("calibration");
                // a triangle step costs the traversal kernel about 0.6 node steps (≈110 against ≈185 instructions)
                const double cost_cur = (double)hcnt[0] + 0.6 * (double)hcnt[1], cost_alt = (double)hcnt[2] + 0.6 * (double)hcnt[3];
                if (getenv("PT_DEBUG_BVH"))
                    fprintf(stderr, "[pt_bvh] calibration (%u rays): %s %.2f node steps + %.2f triangle tests per ray, %s %.2f + %.2f -> %s\n", nrays, names[out->builder],
                            (double)hcnt[0] / nrays, (double)hcnt[1] / nrays, names[kind], (double)hcnt[2] / nrays, (double)hcnt[3] / nrays, cost_alt < cost_cur ? names[kind] : names[out->builder]);
                out->calib_cost = (float)(std::min(cost_cur, cost_alt) / (double)nrays);
                if (cost_alt < cost_cur) {
                    hipFree((void*)out->nodes8); hipFree((void*)out->tris8);
                    out->nodes8 = alt.nodes8; out->tris8 = alt.tris8; out->num_nodes8 = alt.num_nodes8; out->num_tris8 = alt.num_tris8; out->levels8 = alt.levels8;
                    out->builder = kind;
                } else {
                    hipFree((void*)alt.nodes8); hipFree((void*)alt.tris8);
                }
            }
        }
        tfree(cnt);
    }
done:
    tfree(keys); tfree(keys_sorted); tfree(bounds); tfree(tmp);
    tfree(left); tfree(right); tfree(parent); tfree(rfirst); tfree(rlast); tfree(visits);
    tfree(box);
    tfree(tris); // the Morton-ordered triangles only fed the wide tree's leaf arrays
    pc.mark("frees");
    if (arena.guard) { // PT_BVH_GUARD=1: the bands behind the build's standing slices (the phases' own were checked when their marks were released)
        HIPCHK(hipDeviceSynchronize());
        arena.verify(0);
        if (arena.guard_bad) {
            fprintf(stderr, "[pt_bvh] %d arena guard band(s) overwritten: the build is refused\n", arena.guard_bad);
            pt_bvh_free(out);
            return hipErrorUnknown;
        }
    }
    return hipSuccess;
}

// First use of this translation unit's device code in a process: the runtime loads the code object (the rocPRIM sort and scan templates make it
// a few megabytes) when the first of its kernels is launched — 8 ms that used to sit in the first build's "bounds, morton, sort" phase.
// pt_create launches this empty kernel from a helper thread while the scene is being uploaded (pt_api.hip create_from_flat).
__global__ void k_warm_bvh() {}
void pt_bvh_warm(hipStream_t stream) {
    hipLaunchKernelGGL(k_warm_bvh, dim3(1), dim3(64), 0, stream);
    (void)hipGetLastError();
}

void pt_bvh_free(PtBvh* b) {
    if (b->nodes8) hipFree((void*)b->nodes8);
    if (b->tris8) hipFree((void*)b->tris8);
    b->nodes8 = nullptr;
    b->tris8 = nullptr;
}
