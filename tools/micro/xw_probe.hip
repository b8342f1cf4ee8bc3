// tools/micro/xw_probe.hip — what a flag shared by the waves of a launch costs on MI355X (round 5, cross-wave stealing):
//   1. round-trip latency of an agent-scope VMEM load, of s_dcache_inv + s_load (scalar path), of an atomic add with return,
//      on memory from hipMalloc and from hipExtMallocWithFlags(hipDeviceMallocUncached);
//   2. visibility: does a wave polling through the scalar path / the VMEM path see a flag another wave sets with an atomic? after how long?
//   3. same-address throughput: N waves polling one word (VMEM agent-scope loads / scalar loads / atomics), ns per access in aggregate.
// hipcc -O3 --offload-arch=gfx950 tools/micro/xw_probe.hip -o tools/micro/xw_probe && ./tools/micro/xw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef const volatile __attribute__((address_space(4))) unsigned int* CVU;

__device__ unsigned int ld_agent(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ unsigned int ld_scalar(const unsigned int* p) {
    __builtin_amdgcn_s_dcache_inv();
    return *(CVU)(uintptr_t)p;
}
// mode 0: VMEM agent load, 1: scalar load, 2: atomic add 0 with return; one wave, `iters` dependent accesses
__global__ void k_latency(unsigned int* word, int mode, int iters, long long* out) {
    unsigned int acc = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        unsigned int v;
        if (mode == 0) v = ld_agent(word + (acc & 1u));
        else if (mode == 1) v = ld_scalar(word + (acc & 1u));
        else v = atomicAdd(word + (acc & 1u), 0u);
        acc += v; // the next address depends on the value: accesses are serialised
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}
// block 0 sets the flag after `delay` ticks; every other block polls (mode 0 / 1) and records when it saw it (or -1 after `limit`)
__global__ void k_visibility(unsigned int* flag, int mode, long long delay, long long limit, long long* seen, long long* set_at) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0) {
        while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(8);
        if (threadIdx.x == 0) { atomicAdd(flag, 1u); set_at[0] = wall_clock64(); }
        return;
    }
    long long when = -1;
    for (;;) {
        const unsigned int v = mode == 0 ? ld_agent(flag) : ld_scalar(flag);
        const long long now = wall_clock64();
        if (v != 0u) { when = now; break; }
        if (now - t0 > limit) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) seen[blockIdx.x] = when;
}
// N waves hammer one word for `iters` accesses each (independent: not serialised inside a wave)
__global__ void k_throughput(unsigned int* word, int mode, int iters, unsigned int* sink, long long* span) {
    unsigned int acc = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) acc += ld_agent(word);
        else if (mode == 1) acc += ld_scalar(word);
        else if (mode == 2) acc += atomicAdd(word, 0u);
        else atomicAdd(word + 1, 1u); // no return
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { sink[blockIdx.x] = acc; atomicMin((unsigned long long*)&span[0], (unsigned long long)t0); atomicMax((unsigned long long*)&span[1], (unsigned long long)t1); }
}

int main() {
    unsigned int* mem[2];
    CK(hipMalloc((void**)&mem[0], 4096));
    CK(hipExtMallocWithFlags((void**)&mem[1], 4096, hipDeviceMallocUncached));
    long long* out; CK(hipMalloc((void**)&out, 8 * 8192));
    unsigned int* sink; CK(hipMalloc((void**)&sink, 4 * 8192));
    const char* mname[2] = {"hipMalloc", "uncached"};
    const char* aname[4] = {"VMEM agent-scope load", "s_dcache_inv + s_load", "atomicAdd with return", "atomicAdd without return"};
    for (int m = 0; m < 2; ++m) {
        CK(hipMemset(mem[m], 0, 4096));
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(k_latency, dim3(1), dim3(64), 0, 0, mem[m], mode, 200, out);  // warm
            hipLaunchKernelGGL(k_latency, dim3(1), dim3(64), 0, 0, mem[m], mode, 2000, out);
            long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            printf("latency  %-9s %-26s %7.1f ns per dependent access\n", mname[m], aname[mode], h[0] * 10.0 / 2000);
        }
    }
    for (int m = 0; m < 2; ++m)
        for (int mode = 0; mode < 2; ++mode) {
            const int nb = 512;
            std::vector<long long> seen(nb);
            CK(hipMemset(mem[m], 0, 4096));
            // pollers first read the flag while it is 0 (so a cached copy exists), the setter fires after 20 us; limit 2 ms
            hipLaunchKernelGGL(k_visibility, dim3(nb), dim3(64), 0, 0, mem[m], mode, 2000ll, 200000ll, out, out + nb);
            CK(hipMemcpy(seen.data(), out, 8 * nb, hipMemcpyDeviceToHost));
            long long set_at; CK(hipMemcpy(&set_at, out + nb, 8, hipMemcpyDeviceToHost));
            int never = 0; double sum = 0, mx = 0;
            for (int b = 1; b < nb; ++b) { if (seen[b] < 0) ++never; else { double d = (seen[b] - set_at) * 0.01; sum += d; if (d > mx) mx = d; } }
            printf("visible  %-9s %-26s %3d of %d pollers never saw the flag; the others after %.2f us on average, %.2f us at most\n", mname[m], aname[mode], never, nb - 1,
                   (nb - 1 - never) ? sum / (nb - 1 - never) : 0.0, mx);
        }
    for (int m = 0; m < 2; ++m)
        for (int mode = 0; mode < 4; ++mode)
            for (int nb : {64, 512, 4096}) {
                long long init[2] = {0x7fffffffffffffffll, 0};
                CK(hipMemcpy(out, init, 16, hipMemcpyHostToDevice));
                const int iters = 200;
                hipLaunchKernelGGL(k_throughput, dim3(nb), dim3(64), 0, 0, mem[m], mode, iters, sink, out);
                long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
                printf("thruput  %-9s %-26s %5d waves: %7.2f ns per access in aggregate (%.1f us for %d accesses)\n", mname[m], aname[mode], nb, (h[1] - h[0]) * 10.0 / ((double)nb * iters),
                       (h[1] - h[0]) * 0.01, nb * iters);
            }
    return 0;
}
