// Latency micro-benchmark: dependent chains of plain / agent-scope loads and returning agent-scope atomics on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
template <int MODE>
__global__ void chase(unsigned *buf, int n, unsigned *out) {
    unsigned idx = 0;
    for (int i = 0; i < n; i++) {
        unsigned *p = buf + (size_t)idx * 16;
        if (MODE == 0) idx = *p;
        else if (MODE == 1) idx = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) idx = __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 3) idx = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 4) idx = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    out[0] = idx;
}
// many workgroups each doing one agent-scope load of the SAME line vs different lines, then exit: throughput of hot-line coherent reads
template <int MODE>
__global__ void hot(unsigned *buf, unsigned *out, int spread) {
    unsigned *p = buf + (size_t)(spread ? (blockIdx.x % spread) : 0) * 16;
    unsigned v;
    if (MODE == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) { __hip_atomic_fetch_or(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0; }
    else if (MODE == 2) { __hip_atomic_store(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0; }
    else v = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0xDEADBEEF) out[1] = v;
}
int main() {
    const int N = 1 << 20;   // 1 M lines of 64 B = 64 MB
    std::vector<unsigned> h((size_t)N * 16);
    // random permutation cycle
    std::vector<unsigned> perm(N);
    for (int i = 0; i < N; i++) perm[i] = i;
    unsigned s = 12345;
    for (int i = N - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; int j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < N; i++) h[(size_t)perm[i] * 16] = perm[(i + 1) % N];
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int n = 2000;
    auto run = [&](auto kern, const char *name) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, n, o); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, n, o); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-32s %.3f us per dependent op\n", name, ms * 1e3 / n);
    };
    run(chase<0>, "plain load (HBM miss)");
    run(chase<1>, "agent-scope load");
    run(chase<2>, "agent-scope fetch_or returning");
    run(chase<3>, "system-scope load");
    run(chase<4>, "workgroup-scope load");
    auto runhot = [&](auto kern, const char *name, int spread) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(64), 0, 0, d, o, spread); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(2048), dim3(64), 0, 0, d, o, spread); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s spread %4d: %.2f us for 2048 workgroups x 64 lanes\n", name, spread, ms * 1e3);
    };
    for (int sp : {0, 150, 2048}) {
        runhot(hot<0>, "agent load", sp);
        runhot(hot<1>, "agent fetch_or (result unused)", sp);
        runhot(hot<2>, "agent store", sp);
        runhot(hot<3>, "agent fetch_add returning", sp);
    }
    return 0;
}
