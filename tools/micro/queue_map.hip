// Which HIP streams share a hardware queue (and serialise)?  N streams created in order; for every ordered pair (i, j) a ~2 ms spin
// kernel goes to stream i and a trivial kernel to stream j; if j's kernel finishes long before i's the two streams run concurrently.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/queue_map.hip -o /tmp/queue_map ; run: /tmp/queue_map [nstreams]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
__global__ void tiny(int* sink) { if (sink && threadIdx.x == 9999) *sink = 1; }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; ++i) hipStreamCreate(&s[i]);
    hipEvent_t e0, ei, ej;
    hipEventCreate(&e0); hipEventCreate(&ei); hipEventCreate(&ej);
    const long long cyc = 200000; // wall_clock64 ticks at 100 MHz: 2 ms
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i], nullptr); }
    hipDeviceSynchronize();
    printf("rows: stream of the long kernel, columns: stream of the short one; S = serialised behind it, . = concurrent\n");
    for (int i = 0; i < n; ++i) {
        printf("%2d: ", i);
        for (int j = 0; j < n; ++j) {
            if (i == j) { printf("- "); continue; }
            hipDeviceSynchronize();
            hipEventRecord(e0, s[i]);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cyc, nullptr);
            hipEventRecord(ei, s[i]);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[j], nullptr);
            hipEventRecord(ej, s[j]);
            hipEventSynchronize(ej);
            const bool i_done = hipEventQuery(ei) == hipSuccess;
            hipDeviceSynchronize();
            printf("%s ", i_done ? "S" : ".");
        }
        printf("\n");
    }
    return 0;
}
