// tools/micro/atomic_f32.hip: is the hardware FP32 atomic add (global_atomic_add_f32, no return) the IEEE round-to-nearest add, denormals
// included?  Adds pairs (a, b) three ways — plain a + b in a register, hardware atomic into memory holding a, CPU — and compares bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const float* a, const float* b, float* plain, float* atom, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    plain[i] = a[i] + b[i];
    atom[i] = a[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsafeAtomicAdd(&atom[i], b[i]);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), p(n), q(n);
    std::mt19937 rng(7);
    for (int i = 0; i < n; ++i) {
        uint32_t ua = rng(), ub = rng();
        const int kind = i & 7;
        if (kind == 0) { ua &= 0x807fffffu; ub &= 0x807fffffu; }            // both denormal
        if (kind == 1) { ua &= 0x80ffffffu; ub &= 0x807fffffu; }            // tiny normal + denormal
        if (kind == 2) { ub = ua ^ 0x80000000u ^ (rng() & 0xffu); }          // near-cancellation
        if (kind == 3) { ua = (ua & 0x807fffffu) | 0x3f800000u; ub = (ub & 0x807fffffu) | 0x33800000u; } // rounding at 2^-24 relative
        memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4);
        if (kind != 4 && (a[i] != a[i] || b[i] != b[i] || std::isinf(a[i]) || std::isinf(b[i]))) { a[i] = 1.f; b[i] = 2.f; }
        if (kind == 4) { const uint32_t sp[4] = {0x7f800000u, 0xff800000u, 0x7fc00000u, 0x7f800001u}; ua = sp[(i >> 3) & 3]; if (i & 64) ub = sp[(i >> 8) & 3]; memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4); } // inf / NaN operands
    }
    float *da, *db, *dp, *dq;
    hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dp, 4 * n); hipMalloc(&dq, 4 * n);
    hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dp, dq, n);
    hipMemcpy(p.data(), dp, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, 4 * n, hipMemcpyDeviceToHost);
    long bad_plain = 0, bad_atom = 0, bad_atom_den = 0, bad_vs_gpu = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&p[i], &q[i], 4)) ++bad_vs_gpu; // atomic against the GPU's own register add (also for inf / NaN operands)
        if ((i & 7) == 4) continue;                // CPU comparison: finite operands only (NaN payloads are the host's business)
        const float c = a[i] + b[i];
        if (memcmp(&c, &p[i], 4)) ++bad_plain;
        if (memcmp(&c, &q[i], 4)) { ++bad_atom; if ((i & 7) <= 1) ++bad_atom_den; if (bad_atom <= 5) printf("a %a b %a cpu %a atomic %a\n", a[i], b[i], c, q[i]); }
    }
    printf("pairs %d: plain add differs from CPU %ld, hardware atomic differs from CPU %ld (denormal classes: %ld), hardware atomic differs from the GPU's register add %ld (inf / NaN operands included)\n", n, bad_plain, bad_atom, bad_atom_den, bad_vs_gpu);
    return 0;
}
