// Micro-benchmark: cost of reading, inside the same launch, lines that other workgroups (other XCDs) have just hit with
// agent-scope atomics / write-through stores -- the hand-over pattern of k_fuse's continuation.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int METHOD>
__global__ __launch_bounds__(128) void k(unsigned *bits, unsigned long long *stamps, unsigned seq, int nWg, long long *t, unsigned *sink, int ops) {
    const int b = blockIdx.x;
    // every workgroup: `ops` atomics onto replica (b % 16) lines, then its stamp
    for (int w = threadIdx.x; w < ops; w += 128) __hip_atomic_fetch_or(&bits[((size_t)(b % 16) * 150 + (w * 7 + b) % 150) * 16], 1u << (b & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&stamps[b], (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b != 0) return;
    if (threadIdx.x == 0) t[0] = (long long)__builtin_readcyclecounter();
    for (;;) {
        int ok = 1;
        unsigned long long st[16];
#pragma unroll
        for (int e = 0; e < 16; e++) { const int c = threadIdx.x * 16 + e; st[e] = c < nWg ? __hip_atomic_load(&stamps[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq; }
#pragma unroll
        for (int e = 0; e < 16; e++) ok &= st[e] == seq;
        if (__syncthreads_and(ok)) break;
    }
    if (threadIdx.x == 0) t[1] = (long long)__builtin_readcyclecounter();
    unsigned v[32], r = 0;
#pragma unroll
    for (int e = 0; e < 32; e++) {
        unsigned *p = &bits[((size_t)(e % 16) * 150 + (threadIdx.x + (e / 16) * 20) % 150) * 16];
        if (METHOD == 0) v[e] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (METHOD == 1) v[e] = __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (METHOD == 2) v[e] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (METHOD == 3) v[e] = *p;
        else v[e] = __builtin_nontemporal_load(p);
    }
#pragma unroll
    for (int e = 0; e < 32; e++) r |= v[e];
    sink[threadIdx.x] = r;
    __syncthreads();
    if (threadIdx.x == 0) t[2] = (long long)__builtin_readcyclecounter();
}
int main() {
    unsigned *bits, *sink; unsigned long long *stamps; long long *t, ht[3];
    hipMalloc(&bits, 16 * 150 * 64); hipMalloc(&sink, 4096); hipMalloc(&stamps, 8 * 4096); hipMalloc(&t, 64);
    hipMemset(bits, 0, 16 * 150 * 64); hipMemset(stamps, 0, 8 * 4096);
    const int nWg = 2000;
    unsigned seq = 1;
    auto run = [&](auto kern, const char *name, int ops) {
        for (int it = 0; it < 3; it++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(nWg), dim3(128), 0, 0, bits, stamps, seq++, nWg, t, sink, ops); hipEventRecord(b);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost);
            if (it == 2) printf("%-28s ops/wg %3d: kernel %.1f us; wg0 poll %lld cycles, read of 32 freshly-ORed lines per thread %lld cycles\n", name, ops, ms * 1e3,
                                ht[1] - ht[0], ht[2] - ht[1]);
        }
    };
    for (int ops : {0, 30}) {
        run(k<0>, "agent-scope load", ops);
        run(k<1>, "agent fetch_or(0) returning", ops);
        run(k<2>, "system-scope load", ops);
        run(k<3>, "plain load", ops);
        run(k<4>, "nontemporal load", ops);
    }
    return 0;
}
