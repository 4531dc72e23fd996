// msl_common.hip -- error state, device binding, event profiler (internal).
#include "msl_common.h"

#include <cstdlib>

namespace msl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int bind_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available: this library has no CPU fallback");
        return MSL_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (have %d)", device, n); return MSL_ERR_INVALID; }
    if (device >= 16) { set_error("device %d: the library keeps per-device state (default handles, scratch) for devices 0 .. 15 only", device); return MSL_ERR_INVALID; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return MSL_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MSL_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return MSL_ERR_HIP; }
    return MSL_OK;
}

void KernelProfiler::begin(int k, hipStream_t s) {
    open_ = false;
    if (!on || !((mask >> k) & 1u)) return;
    if (npairs == cap) {
        int ncap = cap ? cap * 2 : 256;
        Pair *np = (Pair *)realloc(pairs, sizeof(Pair) * ncap);
        if (!np) return;
        for (int i = cap; i < ncap; i++) { (void)hipEventCreate(&np[i].a); (void)hipEventCreate(&np[i].b); }
        pairs = np; cap = ncap;
    }
    pairs[npairs].k = k;
    (void)hipEventRecord(pairs[npairs].a, s);
    open_ = true;
}
void KernelProfiler::end(hipStream_t s) {
    if (!open_ || npairs >= cap) return;
    open_ = false;
    (void)hipEventRecord(pairs[npairs].b, s);
    npairs++;
}
bool KernelProfiler::kernel_pair(int k, hipEvent_t *a, hipEvent_t *b) {
    if (!on || !((mask >> k) & 1u)) return false;
    if (stride > 1 && (tick[k & 15]++ % (unsigned)stride) != 0) return false;
    if (npairs == cap) {
        int ncap = cap ? cap * 2 : 256;
        Pair *np = (Pair *)realloc(pairs, sizeof(Pair) * ncap);
        if (!np) return false;
        for (int i = cap; i < ncap; i++) { (void)hipEventCreate(&np[i].a); (void)hipEventCreate(&np[i].b); }
        pairs = np; cap = ncap;
    }
    pairs[npairs].k = k;
    *a = pairs[npairs].a; *b = pairs[npairs].b;
    npairs++;
    return true;
}
void KernelProfiler::drain() {
    for (int i = 0; i < npairs; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, pairs[i].a, pairs[i].b) == hipSuccess) { ms[pairs[i].k] += t; launches[pairs[i].k]++; }
    }
    npairs = 0;
}
void KernelProfiler::destroy() {
    for (int i = 0; i < cap; i++) { (void)hipEventDestroy(pairs[i].a); (void)hipEventDestroy(pairs[i].b); }
    free(pairs); pairs = nullptr; cap = npairs = 0;
}

}  // namespace msl

extern "C" {
const char *msl_last_error(void) noexcept { try { return msl::g_err; } MSL_ABI_CATCH_PTR }
const char *msl_version(void) noexcept { try { return "manhattanslam_amd 0.1 (gfx950)"; } MSL_ABI_CATCH_PTR }
int msl_device_count(void) noexcept {
    try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
    } MSL_ABI_CATCH_(return 0)
}
}
