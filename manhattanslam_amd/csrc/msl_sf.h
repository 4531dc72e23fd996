// msl_sf.h -- types and device helpers shared by the three translation units of the surfel-fusion path (internal):
//   msl_sf_superpixel.hip  frame-batched superpixel stage (generateSuperPixels, src/SurfelFusion.cpp:333-773)
//   msl_sf_map.hip         map stage: fusion (:167-283), new surfels (:285-331), compaction (src/SurfelMapping.cpp:366-391)
//   msl_surfel.hip         handle, streams, batching, the C ABI
#pragma once

#include "msl_common.h"

#include <hip/hip_ext.h>

namespace msl {
namespace sf {

constexpr int SP = 8;
constexpr int NCHUNK = 10;  // THREAD_NUM, include/SurfelFusion.h:34
constexpr double MAX_ANGLE_COS = 0.1, HUBER_RANGE = 0.4, MIN_TOLERATE_DIFF = 0.1;   // (BASELINE 0.5 and DISPARITY_ERROR 4.0 appear as exact float factors in k_fuse)
constexpr unsigned T_INF = 0xFFFFFFFFu;
constexpr int PROP_ROUNDS = 6;          // worklist relaxation rounds before the single-workgroup finisher
constexpr unsigned short IDX_NONE = 0xFFFF, IDX_PLANE = 0xFFFE;
constexpr int LIST_D = 256;             // fastest compaction path: k_fuse hands over the few deleted slots directly
constexpr int SCAN_ITEMS = 1024;        // surfels per workgroup and pass in the map-maintenance kernels (k_select_*)
// Sub-block: the surfels one k_fuse wave owns = the granularity of the deleted / updated partials.  128 (two records per lane) since round 5:
// the wave with the most survivors fuses them in two rounds of gathers instead of four, which is what bounds the kernel -- on the dense map, whole
// front end, same box: 256 -> 128 -> 64 surfels per wave: k_fuse 16.2 -> 14.8 -> 14.7 us in the timed region (13.2 alone either way), frames/s
// 22 300 -> 22 080 -> 21 380 (every wave carries ~130 instructions of fixed cost, VALU time the frame-batched kernels lose).
#ifndef MSL_SUB_ITEMS
#define MSL_SUB_ITEMS 128
#endif
constexpr int SUB_ITEMS = MSL_SUB_ITEMS;
constexpr int DEFER_WIN = 32;           // keyframes per deferred-compaction window (one replay per window)
// FuseRec (what k_fuse reads of a seed: three 16-byte words) as one 48-byte record per seed, or as three planes of 16-byte words
#ifndef MSL_FUSEREC_PLANES
#define MSL_FUSEREC_PLANES 0
#endif
__host__ __device__ inline size_t fuserec_index(int nseeds, unsigned seed, int word) {
#if MSL_FUSEREC_PLANES
    return (size_t)word * (size_t)nseeds + seed;
#else
    (void)nseeds; return 3 * (size_t)seed + (size_t)word;
#endif
}

// Device-resident surfel map, split hot/cold.  The fuse kernel streams only the hot records (what decides a surfel's fate for the ones that
// leave early) and touches the cold record of those it updates.
// Hot record, 16 bytes (round 5; 20 bytes before): position + ONE word for updateTimes / lastUpdate, so that a record is one aligned
// dwordx4 access and an updated record is written back as a whole 16-byte piece of a sector.
//   tl bit 31 clear: bits 30..20 = updateTimes (0 .. 2047), bits 19..0 = lastUpdate as a 20-bit two's complement number (-524288 .. 524287:
//                    maps seeded "a few keyframes before keyframe 0", like bench.py's, carry small negative indices)
//   tl == HOT_WIDE : the exact ints live in utlWide[2 i], utlWide[2 i + 1] (values outside those ranges; only maps uploaded by the caller or
//                    sequences beyond 2047 fusions of one surfel / half a million keyframes) -- every accessor honours it
//   tl == HOT_HOLE : deferred compaction only: a slot deleted earlier in the current window, already in the deletion log (never survives a
//                    window: the replay fills or truncates every hole)
struct alignas(16) HotPk { float px, py, pz; unsigned tl; };
struct HotRec { float px, py, pz; int updateTimes, lastUpdate; };   // the unpacked form kernels compute with
constexpr unsigned HOT_WIDE = 0x80000000u, HOT_HOLE = 0xFFFFFFFFu;
__host__ __device__ inline bool tl_fits(int ut, int lu) { return (unsigned)ut < 2048u && ((unsigned)lu + (1u << 19)) < (1u << 20); }   // (unsigned sum: no signed overflow for lu near INT_MAX)
__host__ __device__ inline unsigned tl_pack(int ut, int lu) { return ((unsigned)ut << 20) | ((unsigned)lu & 0xFFFFFu); }
__host__ __device__ inline int tl_ut(unsigned tl) { return (int)(tl >> 20); }                 // (of a record with bit 31 clear)
__host__ __device__ inline int tl_lu(unsigned tl) { return (int)(tl << 12) >> 12; }
// 32 bytes, 32-byte aligned: a fused surfel touches exactly one 32-byte sector of its cold record.  r, g, b always come from a cv::Vec3b
// (src/SurfelFusion.cpp:484, 551), so they travel as three bytes; a record whose ints do not fit a byte (only possible for maps uploaded by
// the caller) sets COLD_WIDE and keeps the exact ints in rgbWide[3 i ..].
struct alignas(32) ColdRec { float nx, ny, nz, size, color, weight; unsigned rgbf; unsigned _spare; };
constexpr unsigned COLD_WIDE = 1u << 24;
struct MapSoA {
    HotPk *hot;            // [cap]
    ColdRec *cold;         // [cap]
    int *rgbWide;          // [cap][3]
    int *utlWide;          // [cap][2]
    long long *wideFlag;   // ctr[13]: bit 0 set once any COLD_WIDE record has been stored, bit 1 once any HOT_WIDE one (map copies then carry the side arrays along)
};
__host__ __device__ inline bool rgb_fits(int r, int g, int b) { return ((unsigned)r | (unsigned)g | (unsigned)b) < 256u; }
__host__ __device__ inline unsigned rgb_pack(int r, int g, int b) { return (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16); }

// Per-keyframe parameters of one slot (device memory, uploaded per batch).
// Image pointers travel through memory, so the compiler only knows them as generic pointers and would emit FLAT loads
// (which also count against lgkmcnt and so serialise with LDS / scalar traffic); the accessors restore the global
// address space.
template <typename T> using gptr = const T __attribute__((address_space(1))) *;
struct FrameDev {
    const uint8_t *gray; const float *depth; const int32_t *member;
    float pose[16], invPose[16];
    int ref, _pad;
#ifdef __HIPCC__
    __device__ __forceinline__ gptr<uint8_t> grayG() const { return (gptr<uint8_t>)gray; }
    __device__ __forceinline__ gptr<float> depthG() const { return (gptr<float>)depth; }
    __device__ __forceinline__ gptr<int32_t> memberG() const { return (gptr<int32_t>)member; }
#endif
};

// What the pixel pass needs of a seed, 32 bytes, so that a wave fetches a candidate with one scalar load (and the two candidates of a lattice row,
// neighbours in memory, with a single 64-byte one): written wherever a seed's x / y / meanDepth / meanIntensity change.
struct alignas(32) AssignRec {
    float x, y, meanIntensity;
    unsigned stable;               // the seed's stable flag as the next pixel pass finds it (t(s) == T_INF)
    double invDepth;               // meanDepth > 0: 1.0 / (double)meanDepth, the value calculateCost's divide gives (:349) -- never negative;
                                   // otherwise -1.0, i.e. the sign doubles as the seed's "has depth" test (:348) and the record needs no meanDepth
    unsigned long long _pad;
};

// What only a few waves of a k_fuse launch need (rare paths): kept in device memory, not in kernel arguments.
struct FuseAux {
    MapSoA map;                     // (the side arrays of wide records, the wide-record flags)
    unsigned long long cap;
    unsigned *delU, *delUCount;     // classic: the hand-over list of k_compact and its length
    unsigned *delList;              // deferred: the window's deletion log
};
// Deferred compaction (round 5): what the fuse launches of one window leave for the replay.
struct DeferCtl {
    long long ext[DEFER_WIN + 1];   // ext[f]: physical extent of the array the fuse launch of keyframe f works on (= ext[f - 1] + new surfels of f - 1)
    unsigned delCnt[DEFER_WIN];     // deleted slots keyframe f logged
    unsigned logBase[DEFER_WIN];    // ... behind the entries of the keyframes before it: logBase[f] = sum of delCnt[0 .. f - 1] (written by launch f)
    unsigned nMoves;                // replay -> gather / scatter
    int flagStride;                 // = SfDev::flagStride
    // bases of the superpixel stage's candidate arrays (all slots; static per allocation): the launch of keyframe kf materialises the new
    // surfels of keyframe kf - 1 from them
    const uint8_t *candOk, *fused;
    const msl_surfel *cand;
    FuseAux aux;                    // written whenever the map is (re)allocated
};

struct SfDev {
    int W, H, spW, spH, nseeds, npx;   // npx = W * H (the flat pixel index range of the reference); spW = W / 8, spH = H / 8 (truncated, :29-38)
    float fx, fy, cx, cy, fuseFar, fuseNear;
    unsigned long long gstride, gbytes, dstride, mstride;   // gray bytes, depth floats, member ints
    unsigned gsB, dsB, msB;      // the same row strides in BYTES as 32-bit numbers (run_batch refuses images whose rows span 4 GB or more): byte_off() below
    const FrameDev *frames;      // [slots]
    msl_seed *seeds, *seedsTmp;  // [slots][nseeds]
    msl_surfel *cand;            // [slots][nseeds] world-frame surfel a seed would spawn
    uint8_t *candOk;             // [slots][flagStride]
    uint8_t *fused;              // [slots][flagStride] 1: seed consumed by a fusion; 2: invalid candidate (kb_seed_plane); 0: the seed spawns a surfel
    uint2 *tex;                  // [slots][npx] {depth bits, final superpixel index} of every pixel: k_fuse's ONE gather per in-view surfel
    float4 *fuseRec;             // [slots][3][nseeds] what k_fuse needs of a seed (FuseRec, msl_sf_superpixel.hip), three planes of 16-byte words
    unsigned short *index, *amap;  // [slots][npx]
    unsigned *tmin;              // [slots][nseeds]
    AssignRec *arec;             // [slots][nseeds] (+ one record of padding at either end) what kb_assign reads of a seed
    float *pxInv;                // [slots][npx] (float)(1.0 / (double)depth) of every pixel (0 when depth <= 0.01): pass 0 writes, passes 1-2 read
    unsigned *wl;                // [slots][npx] relaxation worklist: pixels on a stable seed that pick a different seed
    unsigned *wlCount;           // [slots]
    int *chunkAbort;             // [slots][2][16]
    int *changed;                // [slots][8]
    MapSoA map;
    unsigned long long cap;
    // ctr[0]=n_live  [1]=K new (last)  [2]=D deleted (last)  [3]=updated (last)  [4]=n before (last)  [5]=err  [6]=n after (last)  [7]=tail fallback flag
    // ctr[8..12] = running totals: new, deleted, updated, keyframes, live-before   ctr[13] = wide-record flags
    long long *ctr;
    msl_surfel *newSurfels;
    unsigned *blockSums, *blockUpd, *delList, *srcOf;
    // Dealing of the sub-blocks to the XCDs by SCREEN position (round 6; msl_sf_map.hip, deal_subblocks): every k_fuse wave leaves the screen key of
    // its sub-block (mean image row of its in-view surfels, 0 .. 254; 255 = nothing in view) in sbKeys[]; the launch that follows k_fuse on the map
    // stream (k_compact's second workgroup; k_deal behind a deferred window) turns them into deal[]: wave w of the next k_fuse launch takes the
    // sub-block deal[(w & 7) * (G / 8) + (w >> 3)], so that XCD x works on the sub-blocks that project into the x-th band of image rows and its L2
    // fetches that band of the texel map and of the seed records instead of all of them.  Hints only: any permutation of 0 .. G - 1 is correct.
    unsigned *sbKeys, *deal;     // [blkStride] each
    int dealG;                   // k_compact / k_deal: the grid (sub-blocks, a multiple of 8) to build deal[] for; 0 = leave it alone
    unsigned *tickets;           // [0..1] hand-off counters, [3] change-list length, [4] delUCount
    unsigned *delU;              // [LIST_D] unordered list of the slots k_fuse found deleted (fast path of k_compact)
    unsigned *delUCount;         // number of slots appended (may exceed LIST_D: then the list is incomplete and unused)
    const float *colX, *rowY;    // [W+1], [H+1]: (u - cx) / fx and (v - cy) / fy of the integer pixel coordinates (back_project)
    int pxStride;                // per-slot stride of the per-pixel arrays: npx rounded up to 64 (16-byte vector accesses stay aligned)
    int flagStride;              // per-slot stride of candOk / fused: 64 lanes x a multiple of 16 seeds (zero padding: the flag scans of the map stage read whole 16-byte words)
    // ---- deferred compaction (fuse_body<true>, k_defer_tail, k_replay, k_gather / k_scatter) ----
    int kf;                      // keyframe number inside the window (0 .. DEFER_WIN - 1); the launch of keyframe kf > 0 first materialises the new surfels of kf - 1
    int prevSlotAbs;             // superpixel slot of keyframe kf - 1, counted from the handle's first slot
    DeferCtl *dc;
    unsigned *moveDst;           // [cap] replay: destination of move j (its source: srcOf[j])
    unsigned long long *loc64;   // [cap] replay, dense tables: virtual position -> (element + 1) | stamp << 32
    unsigned *vposD;             // [cap] element -> virtual position + 1
    unsigned *locKeys, *vposKeys;   // [cap] keys written (to build the move list and to clean the tables)
    unsigned *dBig;              // [cap] a keyframe's deleted positions in ascending order when they do not fit the LDS
    unsigned *bitmap;            // [cap / 32 + 1] the same, marking pass
    HotPk *stageHot; ColdRec *stageCold; int *stageRgb, *stageUtl;   // [cap] sources of the window's moves (gathered, then scattered)
};

#ifdef __HIPCC__
__device__ __forceinline__ AssignRec assign_rec(const msl_seed &s) {
    AssignRec a;
    a.x = s.x; a.y = s.y; a.meanIntensity = s.meanIntensity; a.stable = s.stable ? 1u : 0u;
    a.invDepth = s.meanDepth > 0 ? 1.0 / (double)s.meanDepth : -1.0; a._pad = 0;
    return a;
}
__device__ __forceinline__ int seed_chunk(int seedI, int nseeds) {   // THREAD_NUM partition of :430-434
    const int step = nseeds / NCHUNK;
    if (step == 0) return NCHUNK - 1;
    const int c = seedI / step;
    return c > NCHUNK - 1 ? NCHUNK - 1 : c;
}
// The element `off` BYTES behind a global pointer.  With a 32-bit unsigned offset a load takes the scalar-base + vector-offset form: one 32-bit multiply-add per
// address instead of a 64-bit multiply, add, shift and add (five to sixteen instructions per load in the frame-batched kernels, which issue 8 - 17 loads per lane).
template <typename T> __device__ __forceinline__ gptr<T> byte_off(gptr<T> p, unsigned off) {
    return reinterpret_cast<gptr<T>>(reinterpret_cast<const char __attribute__((address_space(1))) *>(p) + off);
}
template <typename T> __device__ __forceinline__ gptr<T> byte_off(const T *p, unsigned off) { return byte_off((gptr<T>)p, off); }   // (a pointer into global memory)
template <typename T> using gptr_w = T __attribute__((address_space(1))) *;
template <typename T> __device__ __forceinline__ gptr_w<T> byte_off_w(T *p, unsigned off) {   // the same for stores
    return reinterpret_cast<gptr_w<T>>(reinterpret_cast<char __attribute__((address_space(1))) *>((gptr_w<T>)p) + off);
}
// (non-negative in-image coordinates)
__device__ __forceinline__ uint8_t gray_at(const SfDev &P, const FrameDev &F, int y, int x) { return *byte_off(F.grayG(), (unsigned)y * P.gsB + (unsigned)x); }
__device__ __forceinline__ float depth_at(const SfDev &P, const FrameDev &F, int y, int x) { return *byte_off(F.depthG(), (unsigned)y * P.dsB + 4u * (unsigned)x); }
__device__ __forceinline__ int32_t member_at(const SfDev &P, const FrameDev &F, int y, int x) {   // membershipImg(y / 2, x / 2)
    return *byte_off(F.memberG(), ((unsigned)y >> 1) * P.msB + 4u * ((unsigned)x >> 1));
}
__device__ __forceinline__ void vec3b(const SfDev &P, const FrameDev &F, float row, float col, int &r, int &g, int &b) {
    const unsigned long long off = (unsigned long long)(int)row * P.gstride + 3ull * (unsigned long long)(int)col;
    r = off < P.gbytes ? F.grayG()[off] : 0;
    g = off + 1 < P.gbytes ? F.grayG()[off + 1] : 0;
    b = off + 2 < P.gbytes ? F.grayG()[off + 2] : 0;
}
__device__ __forceinline__ void back_project(const SfDev &P, float u, float v, float d, float &x, float &y, float &z) {
    x = (u - P.cx) / P.fx * d;   // src/SurfelFusion.cpp:80-85 (float expression, stored to double there)
    y = (v - P.cy) / P.fy * d;
    z = d;
}
// std::min(1.0 / depth / depth, 1.0) (:87-89) is `(1.0 < a) ? 1.0 : a`: a NaN depth (a seed whose plane fit produced NaN) gives NaN, where
// fmin() would give 1.0 -- found by the furnished-room parity tests of round 4.
__device__ __forceinline__ float get_weight(float d) { const double a = 1.0 / (double)d / (double)d; return (float)(1.0 < a ? 1.0 : a); }
__device__ __forceinline__ void mul4(const float *m, float v0, float v1, float v2, float v3, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2) + m[12 + r] * v3;
}
__device__ __forceinline__ void mul3(const float *m, float v0, float v1, float v2, float out[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2;
}
#endif  // __HIPCC__

// adjugate / determinant inverse of a 4x4 (column-major); pins Eigen's Matrix4::inverse()
template <typename T>
__host__ __device__ inline void inverse4(const T *m, T *inv) {
#define M_(r, c) m[(c) * 4 + (r)]
#define DET3(r0, r1, r2, c0, c1, c2)                                                                     \
    (M_(r0, c0) * (M_(r1, c1) * M_(r2, c2) - M_(r1, c2) * M_(r2, c1)) -                                 \
     M_(r0, c1) * (M_(r1, c0) * M_(r2, c2) - M_(r1, c2) * M_(r2, c0)) +                                 \
     M_(r0, c2) * (M_(r1, c0) * M_(r2, c1) - M_(r1, c1) * M_(r2, c0)))
    T cof[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            int rr[3], cc[3], k = 0;
            for (int i = 0; i < 4; i++) if (i != r) rr[k++] = i;
            k = 0;
            for (int i = 0; i < 4; i++) if (i != c) cc[k++] = i;
            const T d = DET3(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
            cof[r][c] = ((r + c) & 1) ? -d : d;
        }
    const T det = ((M_(0, 0) * cof[0][0] + M_(0, 1) * cof[0][1]) + M_(0, 2) * cof[0][2]) + M_(0, 3) * cof[0][3];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[c * 4 + r] = cof[c][r] / det;
#undef DET3
#undef M_
}

// XCD-aware block -> (keyframe slot, block-in-frame) mapping for the frame-batched kernels.  Workgroup `lin` runs on
// XCD lin % 8 (observed dispatch order; used for speed only), and each XCD has its own 4 MB L2: give every XCD whole
// keyframes (slot = xcd, xcd + 8, ...) so that one keyframe's images, index map and seeds (~3 MB) stay L2 resident
// while its workgroups stream through, instead of all 8 L2s thrashing over the whole batch.
// Launch with a 1-D grid of 8 * ceil(n_slots / 8) * blocksPerFrame workgroups.
#ifdef __HIPCC__
__device__ __forceinline__ bool xcd_slot(int blocksPerFrame, int nSlots, int &slot, int &blk) {
    const unsigned lin = blockIdx.x;
    const unsigned j = lin >> 3;
    slot = (int)(lin & 7u) + 8 * (int)(j / (unsigned)blocksPerFrame);
    blk = (int)(j % (unsigned)blocksPerFrame);
    return slot < nSlots;
}
#endif
inline unsigned xcd_grid(int blocksPerFrame, int nSlots) { return 8u * (unsigned)((nSlots + 7) / 8) * (unsigned)blocksPerFrame; }

// Kernel ids of the event profiler (msl_sf_profile_*; names in msl_surfel.hip)
enum { SK_SEED_INIT = 0, SK_ASSIGN, SK_PROP, SK_COMMIT_PX, SK_UPDATE_SEEDS, SK_COMMIT_SEEDS, SK_SEED_PLANE, SK_FUSE, SK_NEW, SK_COMPACT,
       SK_CONVERT, SK_COPY };

// When a kernel is being timed its dispatch carries its own start/stop events (hipExtLaunchKernelGGL), so the
// measurement adds no extra packets to the stream.  (Cross-checked once against in-kernel 100 MHz device-clock stamps:
// 64.3 us by events vs 62.1 us by stamps for the same launches.)
#define MSL_SF_LAUNCH_LDS(prof, kid, st, kern, grid, block, lds, ...)                                  \
    do {                                                                                               \
        hipEvent_t _ea, _eb;                                                                           \
        if ((prof).kernel_pair(kid, &_ea, &_eb))                                                       \
            hipExtLaunchKernelGGL(kern, grid, block, lds, st, _ea, _eb, 0, __VA_ARGS__);               \
        else                                                                                           \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                               \
    } while (0)
#define MSL_SF_LAUNCH(prof, kid, st, kern, grid, block, ...) MSL_SF_LAUNCH_LDS(prof, kid, st, kern, grid, block, 0, __VA_ARGS__)

// ---- host entry points of the two kernel translation units (all asynchronous on the given stream) ----
// msl_sf_superpixel.hip
bool sp_init_attributes(int nseeds);   // true: one keyframe's t(s) fits the LDS (single-launch relaxation)
void sp_launch_stage(KernelProfiler &prof, hipStream_t st, const SfDev &P, int nFrames, bool propLds);
// raw 16-bit depth -> float metres (src/Frame.cc:96-97) for nFrames images: src rows srcStride bytes apart, frames srcFrameStride bytes apart; dst tightly
// packed rows of W floats, frames dstFrameStride floats apart
void sp_launch_depth_u16(hipStream_t st, const void *src, size_t srcStride, size_t srcFrameStride, float *dst, size_t dstFrameStride, int W, int H, int nFrames,
                         float factor);
// msl_sf_map.hip
void map_launch_fuse(KernelProfiler &prof, hipStream_t st, const SfDev &P, int slot, const FrameDev &F, int nSubGrid, int nSubHint, bool deferred, bool dealt);
void map_launch_compact(KernelProfiler &prof, hipStream_t st, const SfDev &P, int slot, bool resident);
void map_launch_replay(KernelProfiler &prof, hipStream_t st, const SfDev &P, int nFrames, unsigned blkStride);   // closes a deferred window of nFrames keyframes
void map_launch_deal(hipStream_t st, const SfDev &P);   // builds P.deal for a grid of P.dealG sub-blocks from the screen keys the last k_fuse launch left
void map_launch_empty_pair(KernelProfiler &prof, hipStream_t st);                   // the profiler's empty-kernel event pair (SK_NEW)
void map_launch_set_ctr(hipStream_t st, const SfDev &P, long long n, int wide);
void map_launch_add_ctr(hipStream_t st, const SfDev &P, long long add);
void map_launch_aos_to_soa(KernelProfiler &prof, hipStream_t st, const SfDev &P, const msl_surfel *src, long long n, bool atEnd);
void map_launch_soa_to_aos(KernelProfiler &prof, hipStream_t st, const SfDev &P, msl_surfel *dst, long long n);
void map_launch_select_count(hipStream_t st, const SfDev &P, int mode, int arg);
void map_launch_select_write(hipStream_t st, const SfDev &P, int mode, int arg, msl_surfel *out, int markDeleted);
void map_launch_collect_changed(hipStream_t st, const SfDev &P, int ref, long long n, unsigned *count, unsigned *idxOut, msl_surfel *recOut, unsigned capOut);
void map_launch_empty(hipStream_t st, int grid, hipEvent_t a, hipEvent_t b);
int sp_debug_div100(const float *x_host, double *out_host, size_t n);
int map_debug_deal(const uint32_t *keys_host, int G, uint32_t *deal_host);
int sp_debug_chain(const float *x_host, const int32_t *n_host, int lists, int huber, float *out_host);

}  // namespace sf
}  // namespace msl
