/*
 * msl_debug.h -- test / measurement accessors of libmsl.so (NOT part of the drop-in boundary, which is msl.h).
 *
 * Used by tests/ (intermediate stages against the CPU oracle), bench.py (device counters, HIP-event timing of single kernels) and the
 * experiment tools under tools/.  Same conventions as msl.h: extern "C", plain pointers, MSL_OK or a negative msl_status.
 */
#ifndef MSL_DEBUG_H
#define MSL_DEBUG_H

#include "msl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ORB extractor ---- */
/* Debug accessors for the parity tests (host output, synchronous, after an extract call):
 * pyramid level image of frame f (unpadded, tightly packed w*h), its blurred version, and the
 * FAST candidates handed to the quadtree (x,y in level pixel coords, response), in order. */
/* Experiment builds (-DMSL_OCT_STAMPS): n <= 200 64-bit words = (100 MHz device clock, shader clock) pairs the level-0 quadtree workgroup of frame 0
 * parked at its phase boundaries; zeros otherwise. */
MSL_API int msl_orb_debug_stamps(msl_orb *h, uint64_t *out, int n) MSL_NOEXCEPT;
MSL_API int msl_orb_debug_level_size(const msl_orb *h, int level, int *w, int *h_out) MSL_NOEXCEPT;
MSL_API int msl_orb_debug_level(msl_orb *h, int frame, int level, int blurred, uint8_t *out) MSL_NOEXCEPT;
MSL_API int msl_orb_debug_candidates(msl_orb *h, int frame, int level, int32_t *xys /*3 ints each*/,
                                     int cap, int *n_out) MSL_NOEXCEPT;
/* Per-kernel timing with HIP events on the handle's stream.  mode 0 = off, -1 = every kernel,
 * otherwise a bit mask of kernel ids (bit k = time kernel k only, so a timed region can carry a
 * single kernel's events).  msl_orb_profile_read returns accumulated milliseconds and launch
 * counts per kernel since the last enable call. */
#define MSL_ORB_NKERNELS 6
MSL_API int msl_orb_profile_enable(msl_orb *h, int mode) MSL_NOEXCEPT;
MSL_API int msl_orb_profile_read(msl_orb *h, float *ms /*[MSL_ORB_NKERNELS]*/,
                                 int32_t *launches /*[MSL_ORB_NKERNELS]*/) MSL_NOEXCEPT;
MSL_API const char *msl_orb_kernel_name(int k) MSL_NOEXCEPT;

/* ---- surfel fusion ---- */
/* Debug accessors (host output, synchronous): superpixel seeds and the pixel->seed index map
 * as left by the last fuse call. */
MSL_API int msl_sf_debug_seeds(msl_sf *h, msl_seed *out /*(w/8)*(h/8)*/) MSL_NOEXCEPT;
MSL_API int msl_sf_debug_index(msl_sf *h, int32_t *out /*w*h*/) MSL_NOEXCEPT;
/* The handle's 16 device counters after a sync (0: live surfels, 1-4/6: last keyframe's new / deleted / updated / before / after,
 * 5: deferred error code; 8-12: running totals over all keyframes since creation -- new, deleted, updated surfels, keyframes, live
 * surfels before each keyframe; 13: some record keeps wide r, g, b; 14-15: spare). */
MSL_API int msl_sf_debug_ctr(msl_sf *h, int64_t out[16]) MSL_NOEXCEPT;
/* n_words 32-bit words from offset_words of one of the map stage's scratch arrays (which = 0: tail-move sources, 1: deleted-slot list, 2: the screen
 * keys the last k_fuse launch left per sub-block, 3: the wave -> sub-block dealing table, XCD-major; 4: out[0] = the grid that table is a permutation
 * for, 0 if none; 5: out[0], out[1] = keyframes this handle has sent through the classic chain (k_fuse + k_compact) and through deferred windows; host
 * output, synchronous).  Instrumented experiment builds (-DMSL_FUSE_STAMPS=<keyframe>, tools/fuse_stamps.py) park device-clock stamps of
 * k_fuse / k_compact / kb_seed_plane there; otherwise the content is meaningless. */
/* Mean time (us) an event pair carried by a dispatch reports for an EMPTY kernel of `grid` single-wave workgroups on the map stream (n launches):
 * the measurement overhead contained in msl_sf_profile_read's per-kernel times (rocprofv3's kernel durations do not contain it). */
MSL_API int msl_sf_debug_event_overhead(msl_sf *h, int grid, int n, float *mean_us) MSL_NOEXCEPT;
MSL_API int msl_sf_debug_scratch(msl_sf *h, int which, size_t offset_words, uint32_t *out, size_t n_words) MSL_NOEXCEPT;

/* Test hook (host only): mse_out[i] = the MSE ahc::PlaneSeg::Stats::compute (AHCPlaneSeg.hpp:148-183) reports for stats[i], evaluated by the
 * scalar code (lanes = 0) or by the clustering's lock-step SIMD form with `lanes` (2, 4, 8, 16) candidates per group; MSL_ERR_INVALID if the CPU
 * lacks the instruction set that width is built for (4 and 8: AVX2, 16: AVX-512F). */
/* Test hook: 1 if msl_peac_membership_batch / msl_peac_extract_batch would cluster a call of n_frames keyframes on the device, 0 if on the host workers
 * (the automatic rule: more than eight frames per worker this process may use, the worker count being divided by LOCAL_WORLD_SIZE; MSL_PEAC_CLUSTER
 * = host / device overrides).  Frames whose node data does not fit the LDS take the host path regardless. */
MSL_API int msl_debug_peac_cluster_on_device(int n_frames) MSL_NOEXCEPT;
/* Worker threads of the plane extractor's host pool that could NOT be started since the process began (std::system_error from thread creation: the
 * calls went on with fewer workers; the first occurrence also leaves a line in msl_last_error() and on stderr).  0 in a healthy process. */
MSL_API long long msl_debug_peac_thread_shortfall(void) MSL_NOEXCEPT;
MSL_API int msl_debug_peac_mse(const msl_peac_stats *stats, size_t n, int lanes, double *mse_out) MSL_NOEXCEPT;
/* Test hook: the dealing of n_subblocks (a multiple of 8) k_fuse sub-blocks to the eight XCDs by screen key (0 .. 254: mean image row of the
 * sub-block's in-view surfels, >= 255: nothing in view), as k_compact / k_deal build it (msl_sf_map.hip, deal_subblocks): deal[x * n / 8 + j] = the
 * sub-block wave 8 j + x takes.  Host arrays; synchronous. */
MSL_API int msl_debug_deal(const uint32_t *keys_host, int n_subblocks, uint32_t *deal_host) MSL_NOEXCEPT;
/* Test hook: out[i] = the kernels' division-free evaluation of (double)(x[i]*x[i]) / 100.0 (host arrays). */
MSL_API int msl_debug_div100(const float *x_host, double *out_host, size_t n) MSL_NOEXCEPT;
/* Test hook: out[q] = the strictly sequential (left-to-right) float sum of the first n[q] <= 256 entries of list q (256 floats each) as the superpixel
 * kernels evaluate it -- a rotating chain over 16 lanes (msl_sf_superpixel.hip); huber != 0: entries +-inf stand for the Huber tail's DOUBLE constant
 * +-0.4 (src/SurfelFusion.cpp:494-503).  Host arrays; synchronous. */
MSL_API int msl_debug_chain_sum(const float *x_host, const int32_t *n_host, int lists, int huber, float *out_host) MSL_NOEXCEPT;

#define MSL_SF_NKERNELS 12
MSL_API int msl_sf_profile_enable(msl_sf *h, int mode) MSL_NOEXCEPT;
/* Sampling for the per-dispatch event pairs: only every stride-th launch of a timed kernel carries events (default 1 = every launch).  A
 * dispatch that carries events costs the stream ~0.3 us; bench.py times every 5th k_fuse launch of its timed region (a stride coprime with the
 * 32 keyframes of a call, so that every position of the chain is sampled equally). */
MSL_API int msl_sf_profile_stride(msl_sf *h, int stride) MSL_NOEXCEPT;
MSL_API int msl_sf_profile_read(msl_sf *h, float *ms, int32_t *launches) MSL_NOEXCEPT;
MSL_API const char *msl_sf_kernel_name(int k) MSL_NOEXCEPT;

/* Test hook of the C ABI's exception barrier: raises a failure inside the library (0: std::bad_alloc, 1: std::runtime_error, 2: a non-standard
 * exception, 3: std::bad_alloc in a worker thread of the plane extractor's pool) and returns what the boundary makes of it: MSL_ERR_NOMEM /
 * MSL_ERR_INTERNAL with msl_last_error() set; never an abort.  No device needed. */
MSL_API int msl_debug_throw(int kind) MSL_NOEXCEPT;

#ifdef __cplusplus
}
#endif
#endif /* MSL_DEBUG_H */
