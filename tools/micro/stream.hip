// Micro-benchmark: streaming 1 M 20-byte records (20 MB) the way k_fuse's phase A does (each lane: 4 consecutive records = 80 contiguous
// bytes as five 16-byte loads, i.e. a lane stride of 80 B per instruction) vs fully coalesced 16-byte loads (lane stride 16 B).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(128) void k(const uint4 *p, unsigned *sink, long long nChunks /* 16-byte chunks */) {
    const long long wave = (long long)blockIdx.x * 2 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long base = wave * 320;   // 320 chunks = 5120 B per wave
    if (base + 320 > nChunks) return;
    uint4 q[5];
#pragma unroll
    for (int j = 0; j < 5; j++) q[j] = MODE == 0 ? p[base + 5 * lane + j] : p[base + 64 * j + lane];
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) a += q[j].x ^ q[j].y ^ q[j].z ^ q[j].w;
    if (a == 0x12345678u) sink[0] = a;
}
int main() {
    const long long bytes = 20LL * 1000 * 1024;   // ~20 MB
    const long long nChunks = bytes / 16;
    uint4 *d; unsigned *s;
    hipMalloc(&d, bytes); hipMalloc(&s, 64); hipMemset(d, 1, bytes);
    const int grid = (int)(nChunks / 320 / 2);
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e9;
            for (int it = 0; it < 10; it++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(128), 0, 0, d, s, nChunks); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(128), 0, 0, d, s, nChunks);
                hipEventRecord(b); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
            }
            if (rep == 1) printf("%s: best %.2f us for %.1f MB -> %.2f TB/s\n", mode == 0 ? "80-byte lane stride (phase A today)" : "coalesced 16-byte lanes             ", best * 1e3,
                                 bytes / 1e6, bytes / (best * 1e-3) / 1e12);
        }
    return 0;
}
