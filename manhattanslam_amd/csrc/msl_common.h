// msl_common.h -- shared host/device helpers for the gfx950 front-end library (internal).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

#include "../../include/msl.h"
#include "../../include/msl_debug.h"

namespace msl {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define MSL_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::msl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                             __LINE__);                                                           \
            return MSL_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

int bind_device(int device);  // hipSetDevice + gfx950 check; returns msl_status

// Exception barrier of the C ABI: every extern "C" entry point is `noexcept { try { ... } MSL_ABI_CATCH_* }` -- the library uses std::vector,
// new and std::thread behind it, and nothing may unwind into a C (cgo / JNI / ctypes) caller.
#define MSL_ABI_CATCH_(fail)                                                                                                       \
    catch (const std::bad_alloc &) { ::msl::set_error("out of host memory inside the library"); fail; }                             \
    catch (const std::exception &e_) { ::msl::set_error("internal error: %s", e_.what()); fail; }                                   \
    catch (...) { ::msl::set_error("internal error (unknown exception)"); fail; }
#define MSL_ABI_CATCH_INT  catch (const std::bad_alloc &) { ::msl::set_error("out of host memory inside the library"); return MSL_ERR_NOMEM; } \
    catch (const std::exception &e_) { ::msl::set_error("internal error: %s", e_.what()); return MSL_ERR_INTERNAL; }                 \
    catch (...) { ::msl::set_error("internal error (unknown exception)"); return MSL_ERR_INTERNAL; }
#define MSL_ABI_CATCH_PTR  MSL_ABI_CATCH_(return nullptr)
#define MSL_ABI_CATCH_VOID MSL_ABI_CATCH_(return)

// Event-pair profiler: one (start, stop) pair per launch of the selected kernels, drained at sync points.
// set_mode(0) = off, set_mode(-1) = every kernel, otherwise a bit mask of kernel ids.
struct KernelProfiler {
    bool on = false;
    unsigned mask = 0xFFFFFFFFu;  // bit k set = kernel id k is timed
    int stride = 1;               // kernel_pair(): every stride-th launch of a kernel id carries events (sampling keeps the instrumented run close to the plain one)
    unsigned tick[16] = {0};
    bool open_ = false;
    int nk = 0;
    float ms[16] = {0};
    int32_t launches[16] = {0};
    struct Pair { hipEvent_t a, b; int k; };
    Pair *pairs = nullptr;
    int npairs = 0, cap = 0;
    void begin(int k, hipStream_t s);
    void end(hipStream_t s);
    // For single-kernel timing without extra stream packets: returns an event pair to hand to hipExtLaunchKernelGGL
    // (the events then carry the dispatch's own start/end timestamps), or false when kernel k is not being timed.
    bool kernel_pair(int k, hipEvent_t *a, hipEvent_t *b);
    void drain();  // requires the stream to be idle
    void set_mode(int m) { on = m != 0; mask = (unsigned)m; for (int i = 0; i < 16; i++) { ms[i] = 0; launches[i] = 0; tick[i] = 0; } }
    void destroy();
};

// ---- device helpers --------------------------------------------------------------------------
#ifdef __HIPCC__

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    // DPP scan (VALU only, no LDS crossbar round trips): Hillis-Steele inside each row of 16 lanes, then the row totals
    // are carried over with row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3).  Lanes without a source add 0.
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return (unsigned)x;
}

// Block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024).
// s_wave must hold >= 17 unsigned. Returns the exclusive prefix; *total gets the block sum.
// Two barriers, no serial section: every thread folds the <= 16 wave totals itself (LDS broadcast reads).
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *s_wave, unsigned *total) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned inc = wave_incl_scan(v);
    __syncthreads();  // protect s_wave reuse
    if (lane_id() == 63) s_wave[w] = inc;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const unsigned t = i < nw ? s_wave[i] : 0u;
        all += t;
        before += i < w ? t : 0u;
    }
    *total = all;
    return before + inc - v;
}
// Two independent exclusive scans sharing one pair of barriers (s_wave: 2 x 16 entries).
__device__ __forceinline__ void block_excl_scan_pair(unsigned a, unsigned b, unsigned *s_wave, unsigned *totalA, unsigned *totalB,
                                                     unsigned &exA, unsigned &exB) {
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned ia = wave_incl_scan(a), ib = wave_incl_scan(b);
    __syncthreads();
    if (lane_id() == 63) { s_wave[w] = ia; s_wave[16 + w] = ib; }
    __syncthreads();
    unsigned beforeA = 0, allA = 0, beforeB = 0, allB = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const unsigned ta = i < nw ? s_wave[i] : 0u, tb = i < nw ? s_wave[16 + i] : 0u;
        allA += ta; beforeA += i < w ? ta : 0u;
        allB += tb; beforeB += i < w ? tb : 0u;
    }
    *totalA = allA; *totalB = allB;
    exA = beforeA + ia - a; exB = beforeB + ib - b;
}

// In-place inclusive scan of an LDS array (len elements) by the whole block.
template <typename T>
__device__ __forceinline__ void block_scan_array_incl(T *arr, int len, unsigned *s_wave) {
    const int nt = blockDim.x;
    const int per = (len + nt - 1) / nt;
    const int b = threadIdx.x * per, e = min(b + per, len);
    unsigned sum = 0;
    for (int i = b; i < e; i++) sum += arr[i];
    unsigned tot;
    unsigned base = block_excl_scan(sum, s_wave, &tot);
    for (int i = b; i < e; i++) { base += arr[i]; arr[i] = (T)base; }
    __syncthreads();
}

#endif  // __HIPCC__

}  // namespace msl
