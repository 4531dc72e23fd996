// FETCH_SIZE / WRITE_SIZE calibration: kernels that read (or write) a KNOWN number of bytes exactly once, in the access
// shapes of this library's kernels.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/calibrate_counters.sh): the ratio
// known bytes / (counter KiB x 1024) is the per-pattern correction tools/summarize_profile.py applies.
//   calib_x16   16 B per lane, consecutive lanes consecutive (k_pyramid / k_blur interior tiles, kb_commit_px, the hot stream when coalesced)
//   calib_x4     4 B per lane, coalesced                      (k_describe patch rows, kb_assign's float reads)
//   calib_x1     1 B per lane, coalesced                      (k_fast's byte-wise tile staging, gray reads of kb_assign)
//   calib_s80   five 16-B loads per lane at an 80-B lane stride (k_fuse phase A: four 20-byte hot records per lane)
//   calib_g8     one 8-B gather per lane at random 8-B aligned positions of a 2.4 MB table (k_fuse texel lookups)
//   calib_w16 / calib_w4 / calib_w1   the same widths as stores (WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void calib_x16(const uint4 *p, unsigned *sink, long long n16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_x4(const unsigned *p, unsigned *sink, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (p[i] == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_x1(const unsigned char *p, unsigned *sink, long long n1) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    if (p[i] == 0x7Fu) sink[0] = 1;
}
__global__ __launch_bounds__(64) void calib_s80(const uint4 *p, unsigned *sink, long long nChunks) {
    const long long base = (long long)blockIdx.x * 320;   // 320 16-byte chunks = 256 records of 20 B per wave
    if (base + 320 > nChunks) return;
    uint4 q[5];
#pragma unroll
    for (int j = 0; j < 5; j++) q[j] = p[base + 5 * threadIdx.x + j];
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) a += q[j].x ^ q[j].y ^ q[j].z ^ q[j].w;
    if (a == 0x12345678u) sink[0] = a;
}
__global__ __launch_bounds__(64) void calib_g8(const uint2 *table, const unsigned *pos, unsigned *sink, long long n) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint2 v = table[pos[i]];
    if ((v.x ^ v.y) == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_w16(uint4 *p, long long n16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void calib_w4(unsigned *p, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = (unsigned)i;
}
__global__ __launch_bounds__(256) void calib_w1(unsigned char *p, long long n1) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n1) p[i] = (unsigned char)i;
}

int main() {
    const long long bytes = 64LL << 20;   // 64 MiB per pattern (x1: 16 MiB)
    unsigned char *d; unsigned *s;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&s, 64);
    (void)hipMemset(d, 1, bytes);
    (void)hipDeviceSynchronize();
    const long long n16 = bytes / 16, n4 = bytes / 4, n1 = bytes / 4;
    // texel gathers: 350 000 random positions in a 2.4 MB table (640 x 480 x 8 B), like one k_fuse launch with 35 % of 1 M surfels in view
    const long long nG = 350000, tableN = 640 * 480;
    std::vector<unsigned> hp(nG);
    srand(7);
    for (auto &v : hp) v = (unsigned)(((long long)rand() * 32768 + rand()) % tableN);
    unsigned *dpos; (void)hipMalloc(&dpos, sizeof(unsigned) * nG);
    (void)hipMemcpy(dpos, hp.data(), sizeof(unsigned) * nG, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_x16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)d, s, n16);
        hipLaunchKernelGGL(calib_x4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const unsigned *)d, s, n4);
        hipLaunchKernelGGL(calib_x1, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, 0, (const unsigned char *)d, s, n1);
        hipLaunchKernelGGL(calib_s80, dim3((unsigned)(n16 / 320)), dim3(64), 0, 0, (const uint4 *)d, s, n16);
        hipLaunchKernelGGL(calib_g8, dim3((unsigned)((nG + 63) / 64)), dim3(64), 0, 0, (const uint2 *)d, dpos, s, nG);
        hipLaunchKernelGGL(calib_w16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4 *)d, n16);
        hipLaunchKernelGGL(calib_w4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (unsigned *)d, n4);
        hipLaunchKernelGGL(calib_w1, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, 0, d, n1);
        (void)hipDeviceSynchronize();
    }
    printf("known bytes: calib_x16 %lld calib_x4 %lld calib_x1 %lld calib_s80 %lld calib_g8 %lld (useful) calib_w16 %lld calib_w4 %lld calib_w1 %lld\n", bytes, bytes, n1,
           (n16 / 320) * 320 * 16, nG * 8, bytes, bytes, n1);
    return 0;
}
