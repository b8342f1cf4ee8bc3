// pt_bvh_build.hip — on-GPU LBVH construction (replaces optixAccelBuild/optixAccelCompact,
// SimplePathtracer.cpp:561-591).  Morton codes of triangle centroids → radix sort → Karras 2012
// hierarchy → bottom-up refit → subtrees of <= PT_LEAF_MAX triangles collapsed to leaves →
// compact 64-byte traversal nodes with padded child boxes and 48-byte leaf triangles in leaf order.
// Deterministic: keys are (morton30 << 32 | primitive) so they are unique.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "pt_bvh.h"
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

// keep[i] = 1 if internal node i stays internal (its range holds more than PT_LEAF_MAX triangles)
__global__ void k_mark(int n, const int* __restrict__ rfirst, const int* __restrict__ rlast, uint32_t* __restrict__ keep) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    keep[i] = (rlast[i] - rfirst[i] + 1) > PT_LEAF_MAX ? 1u : 0u;
}

__device__ __forceinline__ int32_t child_ref(int c, int n, const int* rfirst, const int* rlast, const uint32_t* keep,
                                             const uint32_t* remap) {
    if (c >= n - 1) { // single-triangle leaf
        uint32_t first = (uint32_t)(c - (n - 1));
        return (int32_t)~((first << 3) | 0u);
    }
    if (keep[c]) return (int32_t)remap[c];
    uint32_t first = (uint32_t)rfirst[c], cnt = (uint32_t)(rlast[c] - rfirst[c] + 1);
    return (int32_t)~((first << 3) | (cnt - 1u));
}

__global__ void k_emit_nodes(int n, const int* __restrict__ left, const int* __restrict__ right,
                             const int* __restrict__ rfirst, const int* __restrict__ rlast,
                             const uint32_t* __restrict__ keep, const uint32_t* __restrict__ remap,
                             const float* __restrict__ box, float pad, Node2* __restrict__ nodes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || !keep[i]) return;
    const int l = left[i], r = right[i];
    const float* bl = &box[(size_t)l * 6];
    const float* br = &box[(size_t)r * 6];
    Node2 nd;
    nd.a = make_float4(bl[0] - pad, bl[1] - pad, bl[2] - pad, bl[3] + pad);
    nd.b = make_float4(bl[4] + pad, bl[5] + pad, br[0] - pad, br[1] - pad);
    nd.c = make_float4(br[2] - pad, br[3] + pad, br[4] + pad, br[5] + pad);
    nd.d = make_float4(__int_as_float(child_ref(l, n, rfirst, rlast, keep, remap)),
                       __int_as_float(child_ref(r, n, rfirst, rlast, keep, remap)), 0.f, 0.f);
    nodes[remap[i]] = nd;
}

__global__ void k_emit_tris(const float* __restrict__ verts, const uint32_t* __restrict__ idx,
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
    t.t2 = make_float4(v2[2], __int_as_float((int)p), 0.f, 0.f);
    tris[i] = t;
}

} // namespace

#define HIPCHK(x)                         \
    do {                                  \
        hipError_t e_ = (x);              \
        if (e_ != hipSuccess) return e_;  \
    } while (0)

// Builds the traversal structure for (verts, idx) already resident on the device.
hipError_t pt_bvh_build(const float* d_verts, const uint32_t* d_idx, uint32_t ntri, hipStream_t stream, PtBvh* out) {
    out->nodes = nullptr;
    out->tris = nullptr;
    out->num_nodes = 0;
    out->num_tris = ntri;
    const int n = (int)ntri;
    const int B = 256;
    // leaf triangles + keys
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    uint32_t* bounds = nullptr;
    HIPCHK(hipMalloc(&keys, sizeof(uint64_t) * (size_t)n));
    HIPCHK(hipMalloc(&keys_sorted, sizeof(uint64_t) * (size_t)n));
    HIPCHK(hipMalloc(&bounds, sizeof(uint32_t) * 6));
    uint32_t binit[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIPCHK(hipMemcpyAsync(bounds, binit, sizeof(binit), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_bounds, dim3(min((n + B - 1) / B, 2048)), dim3(B), 0, stream, d_verts, d_idx, ntri, bounds);
    hipLaunchKernelGGL(k_morton, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, ntri, bounds, keys);
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_sorted, (size_t)n, 0, 64, stream));
    void* tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
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

    LeafTri* tris = nullptr;
    HIPCHK(hipMalloc(&tris, sizeof(LeafTri) * (size_t)n));
    hipLaunchKernelGGL(k_emit_tris, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, keys_sorted, n, tris);
    out->tris = tris;

    if (n <= PT_LEAF_MAX) { // the whole scene is one leaf
        out->root = (int32_t)~((0u << 3) | (uint32_t)(n - 1));
        HIPCHK(hipStreamSynchronize(stream));
        hipFree(keys); hipFree(keys_sorted); hipFree(bounds); hipFree(tmp);
        return hipSuccess;
    }

    int *left, *right, *parent, *rfirst, *rlast, *visits;
    float* box;
    uint32_t *keep, *remap;
    HIPCHK(hipMalloc(&left, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&right, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&parent, sizeof(int) * (size_t)(2 * n)));
    HIPCHK(hipMalloc(&rfirst, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&rlast, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&visits, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&box, sizeof(float) * 6 * (size_t)(2 * n)));
    HIPCHK(hipMalloc(&keep, sizeof(uint32_t) * (size_t)n));
    HIPCHK(hipMalloc(&remap, sizeof(uint32_t) * (size_t)n));
    HIPCHK(hipMemsetAsync(visits, 0, sizeof(int) * (size_t)n, stream));
    hipLaunchKernelGGL(k_karras, dim3((n + B - 1) / B), dim3(B), 0, stream, keys_sorted, n, left, right, parent, rfirst, rlast);
    hipLaunchKernelGGL(k_refit, dim3((n + B - 1) / B), dim3(B), 0, stream, d_verts, d_idx, keys_sorted, n, left, right, parent, box, visits);
    hipLaunchKernelGGL(k_mark, dim3((n + B - 1) / B), dim3(B), 0, stream, n, rfirst, rlast, keep);
    size_t tmp2_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp2_bytes, keep, remap, 0u, (size_t)(n - 1), rocprim::plus<uint32_t>(), stream));
    void* tmp2 = nullptr;
    HIPCHK(hipMalloc(&tmp2, tmp2_bytes ? tmp2_bytes : 16));
    HIPCHK(rocprim::exclusive_scan(tmp2, tmp2_bytes, keep, remap, 0u, (size_t)(n - 1), rocprim::plus<uint32_t>(), stream));
    uint32_t last_keep = 0, last_remap = 0;
    HIPCHK(hipMemcpyAsync(&last_keep, keep + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&last_remap, remap + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const uint32_t nnodes = last_keep + last_remap;
    Node2* nodes = nullptr;
    HIPCHK(hipMalloc(&nodes, sizeof(Node2) * (size_t)(nnodes ? nnodes : 1)));
    hipLaunchKernelGGL(k_emit_nodes, dim3((n + B - 1) / B), dim3(B), 0, stream, n, left, right, rfirst, rlast, keep, remap, box, pad, nodes);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    out->nodes = nodes;
    out->num_nodes = nnodes;
    out->root = 0; // node 0 (range = everything) always stays internal when n > PT_LEAF_MAX, and remap[0] = 0
    hipFree(keys); hipFree(keys_sorted); hipFree(bounds); hipFree(tmp); hipFree(tmp2);
    hipFree(left); hipFree(right); hipFree(parent); hipFree(rfirst); hipFree(rlast); hipFree(visits);
    hipFree(box); hipFree(keep); hipFree(remap);
    return hipSuccess;
}

void pt_bvh_free(PtBvh* b) {
    if (b->nodes) hipFree((void*)b->nodes);
    if (b->tris) hipFree((void*)b->tris);
    b->nodes = nullptr;
    b->tris = nullptr;
}
