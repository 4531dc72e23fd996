// msl_sf_map.hip -- map stage of the surfel fusion for gfx950 (MI355X): fusion, new surfels, compaction.
//
// Replaces fuseSurfelsKernel (reference src/SurfelFusion.cpp:167-283), initializeSurfels (:285-331) and the slot refill / tail
// compaction of SurfelMapping::fuseMap (src/SurfelMapping.cpp:366-391) on a device-resident map of 16-byte hot + 32-byte cold records.
//
// Two ways through a keyframe (msl_surfel.hip decides):
//   classic  : k_fuse<false> -> k_compact           two dependent launches per keyframe; the array is in the reference's order after
//                                                   every keyframe (single keyframes, the host-vector drop-in, the first keyframe after
//                                                   the map was replaced from outside)
//   deferred : k_fuse<true> x F -> k_defer_tail -> k_replay -> k_gather -> k_scatter      (round 5) ONE launch per keyframe.
//              fuseSurfelsKernel treats every surfel independently of its array position, so inside a window of F <= 32 keyframes nothing
//              is moved: a keyframe's new surfels are appended physically behind the array (by the "spawn wave" of the NEXT keyframe's fuse
//              launch, which fuses them right away), deleted slots stay as holes and are logged.  The window's placements and tail moves
//              (new surfel k -> k-th largest hole else appended; back-to-front refill, SurfelMapping.cpp:372-390) are then replayed
//              SYMBOLICALLY by one wave over the logs -- virtual position <-> element, only for the few positions that differ from the
//              identity -- and applied as one gather + scatter, which leaves the array exactly as F classic keyframes would have.
//
// HBM-bound integer/float streaming; no MFMA.  Every float expression keeps the reference's evaluation order and float/double
// promotions; compiled with -ffp-contract=off.

#include "msl_sf.h"

using namespace msl;
using namespace msl::sf;

namespace {

// ---- record accessors ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void set_wide_flag_ptr(long long *flag, unsigned long long bit) { atomicOr(reinterpret_cast<unsigned long long *>(flag), bit); }
__device__ __forceinline__ void set_wide_flag(const MapSoA &M, unsigned long long bit) { set_wide_flag_ptr(M.wideFlag, bit); }
// updateTimes / lastUpdate of a record whose packed word is tl (the side array only for HOT_WIDE: rare)
__device__ __forceinline__ void tl_unpack(const MapSoA &M, long long i, unsigned tl, int &ut, int &lu) {
    ut = tl_ut(tl); lu = tl_lu(tl);
    if (tl & 0x80000000u) {
        if (tl == HOT_WIDE) { ut = M.utlWide[2 * i]; lu = M.utlWide[2 * i + 1]; }
        else { ut = 0; lu = 0; }   // HOT_HOLE
    }
}
__device__ __forceinline__ HotRec hot_load(const MapSoA &M, long long i) {
    const HotPk p = M.hot[i];
    HotRec h; h.px = p.px; h.py = p.py; h.pz = p.pz;
    tl_unpack(M, i, p.tl, h.updateTimes, h.lastUpdate);
    return h;
}
__device__ __forceinline__ unsigned tl_store_word(const MapSoA &M, long long i, int ut, int lu) {   // the packed word; writes the side array when it does not fit
    if (tl_fits(ut, lu)) return tl_pack(ut, lu);
    M.utlWide[2 * i] = ut; M.utlWide[2 * i + 1] = lu;
    set_wide_flag(M, 2ull);
    return HOT_WIDE;
}
__device__ __forceinline__ void hot_store(const MapSoA &M, long long i, const HotRec &h) {
    HotPk p; p.px = h.px; p.py = h.py; p.pz = h.pz; p.tl = tl_store_word(M, i, h.updateTimes, h.lastUpdate);
    M.hot[i] = p;
}
// updateTimes = 0 (:201, :229): lastUpdate stays what it was (the host-vector drop-in hands the record back)
__device__ __forceinline__ void hot_mark_deleted(const MapSoA &M, long long i, unsigned tl) {
    if (tl == HOT_WIDE) M.utlWide[2 * i] = 0;
    else M.hot[i].tl = tl & 0xFFFFFu;
}
__device__ __forceinline__ bool hot_is_deleted(const MapSoA &M, long long i) {
    const unsigned tl = M.hot[i].tl;
    return tl == HOT_WIDE ? M.utlWide[2 * i] == 0 : (tl == HOT_HOLE || (tl >> 20) == 0);
}

// Cold records travel as two 16-byte words: a plain struct copy of the 32-byte-aligned ColdRec goes through a private
// temporary that the compiler parks in LDS (12 KB per workgroup in k_compact before this).
struct ColdBits { uint4 a, b; };
__device__ __forceinline__ ColdRec cold_load(const ColdRec *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    ColdBits v; v.a = q[0]; v.b = q[1];
    ColdRec c;
    c.nx = __uint_as_float(v.a.x); c.ny = __uint_as_float(v.a.y); c.nz = __uint_as_float(v.a.z); c.size = __uint_as_float(v.a.w);
    c.color = __uint_as_float(v.b.x); c.weight = __uint_as_float(v.b.y); c.rgbf = v.b.z; c._spare = v.b.w;
    return c;
}
__device__ __forceinline__ void cold_store(ColdRec *p, const ColdRec &c) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(__float_as_uint(c.nx), __float_as_uint(c.ny), __float_as_uint(c.nz), __float_as_uint(c.size));
    q[1] = make_uint4(__float_as_uint(c.color), __float_as_uint(c.weight), c.rgbf, c._spare);
}

__device__ __forceinline__ void store_surfel(const MapSoA &M, long long i, const msl_surfel &e) {
    HotRec h; h.px = e.px; h.py = e.py; h.pz = e.pz; h.updateTimes = e.updateTimes; h.lastUpdate = e.lastUpdate;
    ColdRec c; c.nx = e.nx; c.ny = e.ny; c.nz = e.nz; c.size = e.size; c.color = e.color; c.weight = e.weight; c._spare = 0;
    if (rgb_fits(e.r, e.g, e.b)) c.rgbf = rgb_pack(e.r, e.g, e.b);
    else { c.rgbf = COLD_WIDE; set_wide_flag(M, 1ull); M.rgbWide[3 * i] = e.r; M.rgbWide[3 * i + 1] = e.g; M.rgbWide[3 * i + 2] = e.b; }
    hot_store(M, i, h); cold_store(M.cold + i, c);
}
__device__ __forceinline__ void load_surfel(const MapSoA &M, long long i, const HotRec &h, msl_surfel &e) {
    const ColdRec c = cold_load(M.cold + i);
    e.px = h.px; e.py = h.py; e.pz = h.pz; e.nx = c.nx; e.ny = c.ny; e.nz = c.nz; e.size = c.size; e.color = c.color;
    if (c.rgbf & COLD_WIDE) { e.r = M.rgbWide[3 * i]; e.g = M.rgbWide[3 * i + 1]; e.b = M.rgbWide[3 * i + 2]; }
    else { e.r = (int)(c.rgbf & 255u); e.g = (int)((c.rgbf >> 8) & 255u); e.b = (int)((c.rgbf >> 16) & 255u); }
    e.weight = c.weight; e.updateTimes = h.updateTimes; e.lastUpdate = h.lastUpdate;
}
__device__ __forceinline__ void move_surfel(const MapSoA &M, long long dst, long long src) {
    const ColdRec c = cold_load(M.cold + src);
    const HotPk p = M.hot[src];
    M.hot[dst] = p; cold_store(M.cold + dst, c);
    if (p.tl == HOT_WIDE) { M.utlWide[2 * dst] = M.utlWide[2 * src]; M.utlWide[2 * dst + 1] = M.utlWide[2 * src + 1]; }
    if (c.rgbf & COLD_WIDE) { M.rgbWide[3 * dst] = M.rgbWide[3 * src]; M.rgbWide[3 * dst + 1] = M.rgbWide[3 * src + 1]; M.rgbWide[3 * dst + 2] = M.rgbWide[3 * src + 2]; }
}

// "Last workgroup continues" hand-off (cdna_hip_programming.md G16): every workgroup publishes its global stores with an
// agent-scope release, then takes a ticket; the one that draws the last ticket acquires and carries on with the next
// stage inside the same launch, saving a dependent kernel boundary (~5 us each on this latency-critical chain).
__device__ __forceinline__ bool last_workgroup(unsigned *ticket, unsigned *s_flag) {
    // Everything the continuing workgroup reads from this launch is stored write-through (agent-scope atomic stores /
    // RMW atomics) and read back with agent-scope loads, so no L2 write-back fence is needed.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        *s_flag = (t == gridDim.x - 1) ? 1u : 0u;
        if (*s_flag) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // reset for the next launch
    }
    __syncthreads();
    return *s_flag != 0;
}

// int(projectU + 0.5) of :204-205 (a double addition, truncation towards zero) without double arithmetic: for u >= 1/2 it equals
// floor(u) + (u - floor(u) >= 1/2) -- floor and the difference are exact in float --, and for smaller u (or NaN) both expressions are
// <= 0, which the image test (pUInt < 1) rejects whatever the exact value is; the clamp keeps the conversion defined for huge / infinite u.
__device__ __forceinline__ int round_half_up_pixel(float u) {
    const float c = fminf(fmaxf(u, -4.0f), 1.0e6f);   // NaN -> -4
    const float f = floorf(c);
    return (int)f + ((c - f) >= 0.5f ? 1 : 0);
}
__device__ __forceinline__ unsigned lane_rank(unsigned long long m) {   // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// 16-byte stores of the records phase B rewrites: plain stores (the lines stay dirty in the XCD's L2 until the kernel ends).  Measured and dropped in
// round 5 (A/B on one box): sc1 = write-through (+2.5 us per launch), nt (+0.3 us).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void *p, u32x4 v) { *reinterpret_cast<u32x4 *>(p) = v; }

// What k_fuse reads of the handle and of the keyframe: slim copies of SfDev / FrameDev with the slot offsets folded in on the host.  The
// whole structs are ~150 dwords of kernel arguments = scalar registers the compiler loads up front and then spills around the hot loop;
// what only the spawn wave of a deferred launch needs (the previous keyframe's candidate arrays) and the side arrays of wide records stay in
// memory (DeferCtl).
struct FuseFrame {
    float inv[12];   // rows 0..2 of pose.inverse(), inv[3 c + r] = invPose[4 c + r] (the fourth row is never used)
    int ref;
    const FrameDev *frame;   // the keyframe's device record: the pose itself (only the update path of phase B rotates a normal back into the world)
};
struct FuseArgs {
    int W, H, nseeds, kf;          // kf: keyframe number inside a deferred window (its launch materialises the new surfels of kf - 1 first)
    int prevSlot;                  // superpixel slot of keyframe kf - 1, counted from the first slot of the handle (DeferCtl holds the array bases)
    int rowScale;                  // (254 << 16) / H: image row -> screen key 0 .. 253 of the dealing (SfDev::sbKeys)
    float fx, fy, cx, cy, fuseFar, fuseNear;
    const uint2 *tex; const float4 *fuseRec; uint8_t *fused;   // this keyframe's slot
    HotPk *hot; ColdRec *cold;
    long long *ctr;
    unsigned *blockSums, *blockUpd;   // per-sub-block deleted (classic) / updated counts (deferred: the keyframe's slice)
    unsigned *sbKeys;              // per-sub-block screen key this launch leaves for the next dealing
    const unsigned *deal;          // wave -> sub-block table of THIS launch (XCD-major: [w & 7][w >> 3]); nullptr: array order in runs of FUSE_CHUNK per XCD
    unsigned *delOut;              // where deleted slots go: classic delU[LIST_D] (k_compact's hand-over list), deferred the window's deletion log
    unsigned *delCount;            // ... and their count: classic delUCount, deferred DeferCtl::delCnt[kf]
    DeferCtl *dc;                  // extents and deletion counts of a deferred window; and what only a few waves per launch need (DeferCtl::aux):
                                   // side arrays of wide records, deletion lists, capacity -- loaded where they are used instead of living in scalar
                                   // registers through the whole kernel
};
__host__ inline FuseArgs fuse_args(const SfDev &P, int slot, bool deferred, bool dealt = false) {
    FuseArgs A;
    A.W = P.W; A.H = P.H; A.nseeds = P.nseeds; A.kf = P.kf; A.prevSlot = P.prevSlotAbs; A.rowScale = (254 << 16) / P.H;
    A.sbKeys = P.sbKeys; A.deal = dealt ? P.deal : nullptr;
    A.fx = P.fx; A.fy = P.fy; A.cx = P.cx; A.cy = P.cy; A.fuseFar = P.fuseFar; A.fuseNear = P.fuseNear;
    A.tex = P.tex + (size_t)slot * P.pxStride; A.fuseRec = P.fuseRec + (size_t)slot * P.nseeds * 3; A.fused = P.fused + (size_t)slot * P.flagStride;
    A.hot = P.map.hot; A.cold = P.map.cold; A.ctr = P.ctr;
    A.dc = P.dc;
    A.blockSums = P.blockSums; A.blockUpd = P.blockUpd + (size_t)(deferred ? P.kf : 0) * (P.cap / SUB_ITEMS + 8200);   // (= blkStride of map_realloc)
    A.delOut = deferred ? P.delList : P.delU;
    A.delCount = deferred ? &P.dc->delCnt[P.kf < DEFER_WIN ? P.kf : 0] : P.delUCount;
    return A;
}
__host__ inline FuseFrame fuse_frame(const FrameDev &F, const FrameDev *dev) {
    FuseFrame f;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) f.inv[3 * c + r] = F.invPose[4 * c + r];
    f.ref = F.ref; f.frame = dev;
    return f;
}
// mul4 / mul3 of msl_sf.h on the packed rows: the same products and the same association
__device__ __forceinline__ void mul4r(const float *m, float v0, float v1, float v2, float v3, float out[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = ((m[r] * v0 + m[3 + r] * v1) + m[6 + r] * v2) + m[9 + r] * v3;
}
__device__ __forceinline__ void mul3r(const float *m, float v0, float v1, float v2, float out[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (m[r] * v0 + m[3 + r] * v1) + m[6 + r] * v2;
}

// ---- new surfels of the previous keyframe, materialised by the fuse launch that follows it (deferred compaction) ----------------------
// initializeSurfels (:285-331): every seed whose candidate is valid and that no fusion consumed spawns a surfel, in seed order.  New surfel
// k of the keyframe before (slot P.prevSlot) goes to the physical slot E0 + k (E0 = the extent that keyframe's fuse launch worked on).  The
// spawn wave of the next launch (workgroup 0; k_defer_tail for a window's last keyframe) scans the `fused` bytes of the whole lattice (lane l
// owns the `per` consecutive seeds from l * per on).  A seed spawns iff its byte is 0: kb_seed_init clears it, kb_seed_plane sets 2 where the
// candidate is invalid (candOk = 0), a fusion sets 1; the padding behind the lattice holds 1.  It publishes the new extent, writes the records
// 128 at a time and fuses them like any others.  All 64 lanes must call these.
__device__ __forceinline__ unsigned spawn_word(unsigned fw) { return ~(fw | (fw >> 1)) & 0x01010101u; }   // one bit per byte that is 0
// Pass 1: this lane's number of spawning seeds; the wave-wide exclusive prefix and the total K come from one scan.
__device__ __forceinline__ unsigned spawn_count(const FuseArgs &P, unsigned lane, unsigned &excl) {
    const DeferCtl *dc = P.dc;
    const int fs = dc->flagStride;
    const uint8_t *fusedP = dc->fused + (size_t)P.prevSlot * fs;   // of keyframe kf - 1
    const int per = fs >> 6, nch = per >> 4;   // seeds per lane (a multiple of 16), 16-byte words per lane
    unsigned cnt = 0;
    const uint4 *fq = reinterpret_cast<const uint4 *>(fusedP + (size_t)lane * per);
    for (int c = 0; c < nch; c += 5) {   // five words per trip (640 x 480: the whole lattice in ONE round trip, beside the wave's hot records)
        uint4 b[5];
#pragma unroll
        for (int q = 0; q < 5; q++) b[q] = fq[min(c + q, nch - 1)];
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (c + q < nch) cnt += (unsigned)(__popc(spawn_word(b[q].x)) + __popc(spawn_word(b[q].y)) + __popc(spawn_word(b[q].z)) + __popc(spawn_word(b[q].w)));
    }
    const unsigned incl = wave_incl_scan(cnt);
    excl = incl - cnt;
    return (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
}
// Pass 2 (only when the keyframe spawned something that lands in this sub-block): slot i of the sub-block takes new surfel k = i - E0; its seed is
// the (k - excl[owner])-th spawning seed of the lane whose range contains it.
__device__ __forceinline__ void emit_records(const FuseArgs &P, long long E0, long long c0, int nj, unsigned lane, unsigned K, unsigned excl) {
    const DeferCtl *dc = P.dc;
    const int fs = dc->flagStride;
    const uint8_t *fusedP = dc->fused + (size_t)P.prevSlot * fs;
    const int per = fs >> 6, nch = per >> 4;
    const msl_surfel *cand = dc->cand + (size_t)P.prevSlot * P.nseeds;
#pragma unroll 1
    for (int j = 0; j < nj; j++) {
        const long long i = c0 + 64 * j + lane, kS = i - E0;
        const bool on = kS >= 0 && kS < (long long)K;
        const unsigned k = on ? (unsigned)kS : 0u;
        // owner: the last lane whose exclusive prefix is <= k (its inclusive prefix then exceeds k)
        unsigned lo = 0, hi = 63, eLo = 0;
#pragma unroll
        for (int s = 0; s < 6; s++) {
            const unsigned mid = (lo + hi + 1) >> 1;
            const unsigned e = (unsigned)__builtin_amdgcn_ds_bpermute((int)(mid * 4u), (int)excl);
            if (e <= k) { lo = mid; eLo = e; } else hi = mid - 1;
        }
        unsigned r = k - eLo;   // the r-th spawning seed of lane `lo`'s range
        int seed = -1;
        const uint4 *of = reinterpret_cast<const uint4 *>(fusedP + (size_t)lo * per);
        for (int c = 0; c < nch; c++) {
            const uint4 b = of[c];
            const unsigned w[4] = {spawn_word(b.x), spawn_word(b.y), spawn_word(b.z), spawn_word(b.w)};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned pc = (unsigned)__popc(w[q]);
                if (seed < 0) {
                    if (r < pc) {
                        unsigned m = w[q];
                        for (unsigned t = 0; t < r; t++) m &= m - 1;
                        seed = (int)lo * per + 16 * c + 4 * q + (__builtin_ctz(m) >> 3);
                    } else r -= pc;
                }
            }
        }
        if (on && seed >= 0) {
            if ((unsigned long long)i < dc->aux.cap) store_surfel(dc->aux.map, i, cand[seed]);
            else { long long code = 20; asm volatile("" : "+v"(code)); P.ctr[5] = code; }   // capacity exceeded (the host reserves nseeds slots per keyframe: never
                                                                                          // happens; the constant is kept out of the loop-invariant registers)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave reads these records back right away
}

// k_fuse (:167-283): ONE WAVE per sub-block of SUB_ITEMS = 128 consecutive surfels, no LDS and no workgroup barrier, so a wave starts wherever
// a SIMD has a free slot and 64 registers -- next to the LDS-heavy frame-batched kernels workgroups with LDS waited for it.
//   Phase A (streaming): lane l owns the surfels l and 64 + l of the sub-block (16-byte hot records; a load instruction covers 64
//     consecutive records = 1 KB).  Stale / deleted / out of range / out of image surfels finish here; the in-view ones need ONE 8-byte
//     gather each ({depth, superpixel index} texel written by kb_seed_plane) for the occlusion test.  The gathers of a lane leave together
//     (branch-free, clamped addresses).
//   Hand-over inside the wave: survivor number s (rank by (k, lane) = array order) goes to lane s % 64, round s / 64, with one
//     ds_permute_b32 per k -- a push through the LDS crossbar that allocates no LDS.  Non-survivors push an empty word to the remaining
//     lanes, so every k is a permutation of the 64 lanes and no two lanes ever target the same destination.
//   Phase B (gathers): per round one survivor per lane, neighbouring lanes = neighbouring surfels; its hot record (just streamed: cache
//     hit), 32-byte cold record, the 48-byte record of its seed and the pose's rotation are requested together, so <= 64 survivors cost
//     one round trip and a sub-block wholly in view two.
// DEFER = false (classic): deleted slots are handed to k_compact in delU (one atomic per wave that deleted something), per-sub-block deleted /
//   updated counts go to blockSums / blockUpd with plain stores.
// DEFER = true: deleted slots become HOT_HOLE and go to the window's deletion log; the regular waves of keyframe kf > 0 work on the slots below
//   E0 (the extent keyframe kf - 1 worked on), the launch's spawn wave (spawnWave = true, its own instantiation) on the new surfels of kf - 1.
template <bool DEFER, bool spawnWave>
__device__ __forceinline__ void fuse_body(const FuseArgs &P, const FuseFrame &F, int nSubHint, unsigned waveIdx, int G) {
    constexpr int KPL = SUB_ITEMS / 64;        // records per lane; a wave owns WSPAN = SUB_ITEMS consecutive surfels (measurements: msl_sf.h)
    constexpr long long WSPAN = 64 * KPL;
    struct { HotPk *hot; ColdRec *cold; } M = {P.hot, P.cold};
    const FuseAux *aux = &P.dc->aux;
    const unsigned lane0 = threadIdx.x;
    const uint2 *tex = P.tex;
    const float4 *fuseRec = P.fuseRec;
    uint8_t *fused = P.fused;
    const int ref = F.ref;
    const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
    const float halfF = 0.5f * cameraF;   // BASELINE * cameraF (:220), exact
    // deferred, keyframe kf > 0: E0 = the extent keyframe kf - 1 worked on; its new surfels follow from there
    const bool pending = DEFER && P.kf > 0;
    long long E0v = 0;   // (requested here, first used behind the hot records of the wave's first sub-block: the two travel together)
    if (pending) E0v = P.dc->ext[P.kf - 1];
    long long E0 = 0;
    if (DEFER && spawnWave) E0 = ((long long)__builtin_amdgcn_readfirstlane((int)(E0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)E0v);
    if (spawnWave && !pending) return;   // (the first keyframe of a window has nothing to materialise)
    if (DEFER && !pending && waveIdx == 0 && lane0 == 0) P.dc->ext[0] = P.ctr[0];
    if (DEFER && !spawnWave && waveIdx == 0 && lane0 == 0) P.dc->logBase[P.kf] = P.kf > 0 ? P.dc->logBase[P.kf - 1] + P.dc->delCnt[P.kf - 1] : 0u;   // where this keyframe's log entries start
    // The spawn wave (deferred, one per launch, dispatched first): the new surfels of keyframe kf - 1 go to the physical slots E0, E0 + 1, ...; this
    // wave counts them (one trip over the lattice's flag words), publishes the extent for the next launch, and -- only if there are any -- writes
    // them and fuses them itself, 256 at a time.  No other wave of the launch ever waits for the count: they work on the slots below E0.
    unsigned spK = 0;
    if (DEFER && spawnWave) {
        unsigned excl;
        spK = spawn_count(P, lane0, excl);
        if (lane0 == 0) P.dc->ext[P.kf] = E0 + (long long)spK;
        if (spK == 0) return;
    }
    // Wave g owns sub-block G - 1 - g (the newest surfels -- nearly all in view: most phase-B work -- are dispatched first) and, should the
    // map have outgrown the grid, G - 1 - g + G, ... (grid-stride; normally one iteration).  The grid covers the host's last KNOWN live count
    // plus a margin, not its upper bound.  Sub-blocks below nSubHint load at once; above it the wave reads the live count first and leaves if
    // there is nothing for it.  Capacity is a multiple of 4096 and every sub-block that loads speculatively lies below it.
    // Workgroups are dispatched round-robin over the 8 XCDs: give each XCD runs of FUSE_CHUNK consecutive sub-blocks (neighbouring surfels
    // project to neighbouring pixels, so an XCD's L2 fetches a part of the texel map instead of all of it; small enough runs keep the XCDs
    // balanced -- whole eighths of the map were 2 x slower).
#ifndef MSL_FUSE_CHUNK
#define MSL_FUSE_CHUNK 16
#endif
    constexpr unsigned FUSE_CHUNK = MSL_FUSE_CHUNK;
    // Round 6: when the launch before left screen keys, the sub-blocks are DEALT by screen position instead (P.deal, built by deal_subblocks below):
    // XCD x gets the sub-blocks whose in-view surfels project into the x-th band of image rows, top to bottom, then its share of the sub-blocks
    // with nothing in view -- its L2 then fetches one band of the texel map and of the seed records, not the whole screen (every XCD fetching the
    // whole 2.46 MB texel map was a third of the kernel's fabric traffic).  One scalar load on the head of the wave's chain.
    long long sb0;
    if (!spawnWave && P.deal != nullptr) {
        const unsigned gs = (unsigned)G >> 3;   // (G is a multiple of 8 whenever a table is handed over)
        // (a scalar load by hand: the compiler cannot prove that no store of the kernel aliases the table and would fetch the wave-uniform word
        // through the vector cache; the launch before wrote it, and the scalar cache is invalidated at every kernel start)
        const unsigned *dp = P.deal + ((waveIdx & 7u) * gs + (waveIdx >> 3));
        unsigned dv;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dv) : "s"(dp) : "memory");
        sb0 = (long long)dv;
    } else {
        long long lin = waveIdx;
        constexpr unsigned T = 8u * FUSE_CHUNK;
        const unsigned full = ((unsigned)G / T) * T;
        if (waveIdx < full) { const unsigned grp = waveIdx / T, r = waveIdx % T; lin = (long long)grp * T + (r & 7u) * FUSE_CHUNK + (r >> 3); }
        sb0 = (long long)G - 1 - lin;
    }
    for (long long it = 0;; it++) {
        // (the lane number is re-materialised per iteration: values derived from it are then not hoisted out of this -- normally single-trip --
        // loop and kept in registers / scratch for its whole body)
        unsigned lane = lane0;
        asm volatile("" : "+v"(lane));
#define REC_LOCAL(k) (64u * (unsigned)(k) + lane)
        const long long sb = sb0 + it * G;   // regular waves: the sub-block; grid-stride should the map have outgrown the grid
        long long c0, n = 0, cntIdx;
        if (DEFER && spawnWave) {
            c0 = E0 + it * WSPAN;
            n = E0 + (long long)spK;
            if (c0 >= n) return;
            unsigned excl;   // (the per-lane prefix again rather than a register kept through the whole body: this path runs when a keyframe spawned something)
            (void)spawn_count(P, lane, excl);
            emit_records(P, E0, c0, KPL, lane, spK, excl);
            cntIdx = E0 / SUB_ITEMS + 1 + it;   // its updated counts sit behind those of the sub-blocks below E0 (k_defer_tail adds them up)
        } else {
            c0 = sb * WSPAN; cntIdx = sb;
            if (pending) {
                // (the grid lies inside the capacity, so a wave's FIRST sub-block is requested before the extent has arrived; the slots from E0 on belong
                // to the spawn wave: sub-blocks wholly beyond E0 leave below, records beyond it inside a sub-block fail the `i < n` test)
                if (it > 0) {
                    E0 = ((long long)__builtin_amdgcn_readfirstlane((int)(E0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)E0v);
                    if (c0 >= E0) return;
                }
            } else if (sb >= nSubHint && c0 >= __hip_atomic_load(&P.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
        // lane l owns records l, 64 + l, 128 + l, 192 + l of the sub-block: the survivors' rank order (k, lane) is then the array order, so
        // neighbouring lanes of phase B work on neighbouring records and their gathers and stores share cache lines
        HotPk hq[KPL];
#pragma unroll
        for (int k = 0; k < KPL; k++) hq[k] = M.hot[c0 + REC_LOCAL(k)];
        if (!pending) n = P.ctr[0];
        else if (!spawnWave) {
            E0 = ((long long)__builtin_amdgcn_readfirstlane((int)(E0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)E0v);
            if (c0 >= E0) return;
            n = E0;
        }
        unsigned stp = 0;  // two bits per record: 0: nothing to do, 1: stale -> delete, 2: already deleted, 3: in view
        unsigned keyAcc = 0;   // bits 0..15: sum of the screen keys (image row scaled to 0 .. 253) of the lane's in-view records, bits 16..: their number
        float pzv[KPL];
        unsigned offT[KPL];
        // rare: a record with exact ints in the side array, or a slot the window has logged already -- ONE test for the lane's four records
        unsigned anyTl = 0;
#pragma unroll
        for (int k = 0; k < KPL; k++) anyTl |= hq[k].tl;
        const bool anyHi = __builtin_expect(__ballot(anyTl >> 31) != 0ull, 0);
#pragma unroll
        for (int k = 0; k < KPL; k++) {
            const long long i = c0 + REC_LOCAL(k);
            const float x = hq[k].px, y = hq[k].py, z = hq[k].pz;
            const unsigned tl = hq[k].tl;
            int ut = tl_ut(tl), lu = tl_lu(tl);
            bool hole = false;
            if (anyHi) {
                if (tl == HOT_WIDE) { const int *w = aux->map.utlWide; ut = w[2 * i]; lu = w[2 * i + 1]; }
                else if (tl & 0x80000000u) hole = true;
            }
            float pc[3];
            mul4r(F.inv, x, y, z, 1.0f, pc);
            const bool inRange = !(pc[2] < P.fuseNear || pc[2] > P.fuseFar);
            const bool live = i < n && !hole, stale = ref - lu > 5 && ut < 5;
            // what does not need the projection: stale -> delete (1), already deleted (2), out of range (0)
            int st = 0;
            if (live) st = stale ? (ut != 0 ? 1 : 2) : (ut == 0 ? 2 : 0);
            const bool cand = live && !stale && ut != 0 && inRange;
            unsigned off = 0;
            // (branch-free on purpose.  Skipping the two divisions, the roundings and the image test for 64-record groups that lie outside the frustum
            // as a whole -- `if (__ballot(cand))` -- measured 0.5 us SLOWER alone and no faster beside the frame-batched kernels: the kernel is
            // bound by its chain of memory round trips, not by these instructions)
            {
                const float zq = inRange ? pc[2] : 1.0f;   // keeps the (unused) quotients of skipped surfels finite
                const float projectU = pc[0] * P.fx / zq + P.cx, projectV = pc[1] * P.fy / zq + P.cy;  // :75-78
                const int pUInt = round_half_up_pixel(projectU), pVInt = round_half_up_pixel(projectV);   // int(projectU + 0.5) wherever it matters
                const bool inImage = !(pUInt < 1 || pUInt > P.W - 2 || pVInt < 1 || pVInt > P.H - 2);
                if (cand && inImage) { st = 3; keyAcc += (((unsigned)pVInt * (unsigned)P.rowScale) >> 16) | 0x10000u; }
                // a record that is not in view needs no texel: all such lanes read texel 0 (ONE line for the whole wave) instead of up to 64 scattered
                // border texels -- two thirds of the dense map's records, each a separate request to the vector cache (round 4 clamped the address
                // to the border texel nearest to the projection)
                off = st == 3 ? (unsigned)(pVInt * P.W + pUInt) : 0u;
            }
            stp |= (unsigned)st << (2 * k); pzv[k] = pc[2];
            offT[k] = off;
        }
        uint2 tx[KPL];
        {
#pragma unroll
            for (int k = 0; k < KPL; k++) tx[k] = tex[offT[k]];
            // a common use of all four results: keeps the compiler from sinking each load into its (conditional) consumer, which would turn one
            // round trip back into up to four dependent ones
            if constexpr (KPL == 4) asm volatile("" ::"v"(tx[0].x), "v"(tx[1].x), "v"(tx[2].x), "v"(tx[3].x), "v"(tx[0].y), "v"(tx[1].y), "v"(tx[2].y), "v"(tx[3].y));
            else asm volatile("" ::"v"(tx[0].x), "v"(tx[KPL - 1].x), "v"(tx[0].y), "v"(tx[KPL - 1].y));
        }
        // the sub-block's screen key for the next dealing: mean row of its in-view records (255: nothing in view) -- stored here, before phase B,
        // so that nothing of it stays live through the gathers (a hint: the approximate reciprocal is good enough)
        if (!(DEFER && spawnWave)) {
            const unsigned ks = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(keyAcc), 63);
            const unsigned kc = ks >> 16;
            const unsigned key = kc ? min((unsigned)((float)(ks & 0xFFFFu) * __builtin_amdgcn_rcpf((float)kc)), 254u) : 255u;
            if (lane == 0) P.sbKeys[cntIdx] = key;
        }
        // ---- classification: deletions of phase A, survivors ----
        // (one bit field per lane instead of eight lane masks: the masks would live in scalar registers, which this kernel is short of)
        unsigned fl = 0;   // bit k: record k deleted in phase A; bit 4 + k: record k survives into phase B
        unsigned cntDel = 0;
#pragma unroll
        for (int k = 0; k < KPL; k++) {
            const unsigned st = (stp >> (2 * k)) & 3u;
            const bool occluded = st == 3u && (double)pzv[k] < (double)__uint_as_float(tx[k].x) - 1.0;
            const bool del = st == 1u || st == 2u || occluded;
            if (DEFER) { if (del) M.hot[c0 + REC_LOCAL(k)].tl = HOT_HOLE; }
            else if (st == 1u || occluded) {   // updateTimes = 0, lastUpdate stays (:201; the host-vector drop-in hands the record back)
                if (__builtin_expect(hq[k].tl == HOT_WIDE, 0)) aux->map.utlWide[2 * (c0 + REC_LOCAL(k))] = 0;
                else M.hot[c0 + REC_LOCAL(k)].tl = hq[k].tl & 0xFFFFFu;
            }
            fl |= del ? (1u << k) : 0u;
            fl |= (st == 3u && !occluded) ? (16u << k) : 0u;
            cntDel += (unsigned)__popcll(__ballot(del));
        }
        // deleted slots: classic -> delU (k_compact's fast path), deferred -> the window's log behind the entries of the keyframes before
        // (a wave that deletes is rare but often among the last to finish: everything it needs travels in ONE round trip -- the count's atomic and, for
        // a deferred keyframe, the log position the keyframes before left, dc->logBase[kf - 1] + dc->delCnt[kf - 1])
        auto list_base = [&](unsigned c) -> unsigned {
            unsigned base = 0, prior = 0;
            if (DEFER && P.kf > 0) prior = P.dc->logBase[P.kf - 1] + P.dc->delCnt[P.kf - 1];
            if (lane == 0) base = atomicAdd(P.delCount, c);
            return (unsigned)__builtin_amdgcn_readfirstlane((int)base) + prior;
        };
        auto hand_over = [&](bool d, unsigned long long m, unsigned base, long long i) {
            if (d) {
                const unsigned j = base + lane_rank(m);
                if (DEFER || j < (unsigned)LIST_D) P.delOut[j] = (unsigned)i;   // (the log holds one entry per physical slot at most: it cannot overflow the capacity)
            }
        };
        if (cntDel) {   // rare: a handful of slots per keyframe
            unsigned base = list_base(cntDel);
#pragma unroll
            for (int k = 0; k < KPL; k++) {
                const bool d = (fl >> k) & 1u;
                const unsigned long long m = __ballot(d);
                hand_over(d, m, base, c0 + REC_LOCAL(k)); base += (unsigned)__popcll(m);
            }
        }
        // ---- survivors -> (round, lane): one push per k.  word = local index, valid bit, superpixel << 16 ----
        unsigned rcv[KPL], bk[KPL];
        unsigned total = 0;
#pragma unroll
        for (int k = 0; k < KPL; k++) {
            const bool sv = (fl >> (4 + k)) & 1u;
            const unsigned long long m = __ballot(sv);
            const unsigned c = (unsigned)__popcll(m), rs = lane_rank(m);
            const unsigned dest = (sv ? total + rs : total + c + (lane - rs)) & 63u;
            const unsigned payload = sv ? (REC_LOCAL(k) | 0x100u | (tx[k].y << 16)) : 0u;
            rcv[k] = (unsigned)__builtin_amdgcn_ds_permute((int)(dest * 4u), (int)payload);
            bk[k] = total;
            total += c;
        }
        const unsigned rounds = (total + 63u) >> 6;
        unsigned nupd = 0, cntDelB = 0;
        for (unsigned r = 0; r < rounds; r++) {   // one round for <= 64 survivors
            unsigned item = 0u;
#pragma unroll
            for (int k = 0; k < KPL; k++) {
                const unsigned rk = (bk[k] + ((lane - bk[k]) & 63u)) >> 6;   // round of the survivor this lane received from k (if any)
                if ((rcv[k] & 0x100u) && rk == r) item = rcv[k];
            }
            // branch-free loads: a lane without a survivor in this round reads record c0 / seed 0 (valid addresses, one line for all such
            // lanes) -- conditional loads made the compiler sink the first uses into the load block and wait there
            const long long i = c0 + (item & 0xFFu);
            const unsigned sp = item >> 16;
            const HotPk h = M.hot[i];
            // the update reads normal, size and weight of the cold record and overwrites the rest: two loads (a whole-struct copy became three)
            ColdRec c;
            {
                const float4 cn = *reinterpret_cast<const float4 *>(M.cold + i);
                c.nx = cn.x; c.ny = cn.y; c.nz = cn.z; c.size = cn.w; c.weight = M.cold[i].weight;
            }
            const float4 f0 = fuseRec[fuserec_index(P.nseeds, sp, 0)], f1 = fuseRec[fuserec_index(P.nseeds, sp, 1)], f2 = fuseRec[fuserec_index(P.nseeds, sp, 2)];
            // the rotation of the pose (only the update path needs it, to turn the fused normal back into the world): three 12-byte loads from the
            // keyframe's device record, requested HERE with the records -- left to the compiler they sat behind the tests, one more dependent round
            // trip in every round (k_fuse 18.1 against 16.5 us under rocprofv3); as kernel arguments they cost nine scalar registers this kernel lacks
            const float *poseM = F.frame->pose;
            const float r00 = poseM[0], r10 = poseM[1], r20 = poseM[2], r01 = poseM[4], r11 = poseM[5], r21 = poseM[6], r02 = poseM[8], r12 = poseM[9], r22 = poseM[10];
            // common use of one field per load instruction: all records are in flight together
            asm volatile("" ::"v"(h.px), "v"(h.tl), "v"(c.nx), "v"(c.weight), "v"(f0.x), "v"(f1.x), "v"(f2.x), "v"(r00), "v"(r01), "v"(r02));
            bool upd = false, delB = false;
            if (item && __float_as_uint(f2.w) != 0u) {   // seed tests of :214-219 (norm != 0, viewCos >= MAX_ANGLE_COS)
                const float seedDepth = f0.w;
                const float pz = ((F.inv[2] * h.px + F.inv[5] * h.py) + F.inv[8] * h.pz) + F.inv[11] * 1.0f;   // row 2 of mul4: as in phase A
                // :220-221 is (float)((double)(pz pz) / (0.5 (double)cameraF) * 4.0).  Both operands of the division are float values (0.5 cameraF
                // exactly), the multiplication by 4 is exact, and rounding a correctly rounded binary64 quotient of two binary32 numbers to
                // binary32 gives the correctly rounded binary32 quotient (53 >= 2 * 24 + 2: double rounding is innocuous for division), so one
                // IEEE float division yields the same bits as the double expression at a third of the instructions.
                float tolerateDiff = (pz * pz) / halfF * 4.0f;
                tolerateDiff = tolerateDiff < MIN_TOLERATE_DIFF ? (float)MIN_TOLERATE_DIFF : tolerateDiff;
                if (!(pz < seedDepth - tolerateDiff) && !(pz > seedDepth + tolerateDiff)) {
                    float nc[3];
                    mul3r(F.inv, c.nx, c.ny, c.nz, nc);
                    const float normDiffCos = nc[0] * f0.x + nc[1] * f0.y + nc[2] * f0.z;
                    if (normDiffCos < MAX_ANGLE_COS) {
                        if (DEFER) M.hot[i].tl = HOT_HOLE;
                        else if (__builtin_expect(h.tl == HOT_WIDE, 0)) aux->map.utlWide[2 * i] = 0;
                        else M.hot[i].tl = h.tl & 0xFFFFFu;
                        delB = true;
                    } else {
                        const float oldWeight = c.weight;
                        const float newWeight = f1.w;                      // getWeight(seed.meanDepth)
                        const float sumWeight = oldWeight + newWeight;
                        const float fusedPx = (h.px * oldWeight + newWeight * f1.x) / sumWeight;   // f1.xyz = pose * seed.pos
                        const float fusedPy = (h.py * oldWeight + newWeight * f1.y) / sumWeight;
                        const float fusedPz = (h.pz * oldWeight + newWeight * f1.z) / sumWeight;
                        float fusedNx = nc[0] * oldWeight + newWeight * f0.x;
                        float fusedNy = nc[1] * oldWeight + newWeight * f0.y;
                        float fusedNz = nc[2] * oldWeight + newWeight * f0.z;
                        // :254-257: newNormLength is a double that holds a float (std::sqrt(float)); float /= double is a binary64 division
                        // of two float values rounded to float = the IEEE float division (same argument as above)
                        const float newNormLength = sqrtf(fusedNx * fusedNx + fusedNy * fusedNy + fusedNz * fusedNz);
                        fusedNx = fusedNx / newNormLength; fusedNy = fusedNy / newNormLength; fusedNz = fusedNz / newNormLength;
                        float newNormW[3];
                        newNormW[0] = (r00 * fusedNx + r01 * fusedNy) + r02 * fusedNz;   // mul3(pose, ...): the same products, the same association
                        newNormW[1] = (r10 * fusedNx + r11 * fusedNy) + r12 * fusedNz;
                        newNormW[2] = (r20 * fusedNx + r21 * fusedNy) + r22 * fusedNz;
                        int ut = (int)(h.tl >> 20);   // (a survivor is never a hole; HOT_WIDE: the side array)
                        if (__builtin_expect(h.tl == HOT_WIDE, 0)) ut = aux->map.utlWide[2 * i];
                        unsigned tlNew = tl_pack(ut + 1, ref);             // updateTimes + 1, lastUpdate = reference index (:275-276)
                        if (__builtin_expect(!tl_fits(ut + 1, ref), 0)) {   // rare: exact ints to the side array (pointers fetched one at a time: no register tuples in a cold path)
                            int *w = aux->map.utlWide;
                            w[2 * i] = ut + 1; w[2 * i + 1] = ref;
                            asm volatile("" ::: "memory");
                            set_wide_flag_ptr(aux->map.wideFlag, 2ull);
                            tlNew = HOT_WIDE;
                        }
                        c.rgbf = __float_as_uint(f2.z);                    // r, g, b of the seed (bytes: never COLD_WIDE)
                        c.nx = newNormW[0]; c.ny = newNormW[1]; c.nz = newNormW[2];
                        c.weight = sumWeight;
                        c.color = f2.y;                                    // seed.meanIntensity
                        const float newSize = f2.x;                        // seed.size * fabs(meanDepth / (cameraF * viewCos))
                        if (newSize < c.size) c.size = newSize;
                        u32x4 hv = {__float_as_uint(fusedPx), __float_as_uint(fusedPy), __float_as_uint(fusedPz), tlNew};
                        u32x4 c0v = {__float_as_uint(c.nx), __float_as_uint(c.ny), __float_as_uint(c.nz), __float_as_uint(c.size)};
                        u32x4 c1v = {__float_as_uint(c.color), __float_as_uint(c.weight), c.rgbf, 0u};   // (_spare is 0 in every record: store_surfel)
                        st16(M.hot + i, hv);
                        st16(M.cold + i, c0v);
                        st16(reinterpret_cast<u32x4 *>(M.cold + i) + 1, c1v);
                        fused[sp] = 1;
                        upd = true;
                    }
                }
            }
            nupd += (unsigned)__popcll(__ballot(upd));
            const unsigned long long mb = __ballot(delB);
            if (mb) {   // rare
                const unsigned cb = (unsigned)__popcll(mb);
                hand_over(delB, mb, list_base(cb), i);
                cntDelB += cb;
            }
        }
        if (lane == 0) {   // per-sub-block counts: deleted (classic: the slow paths of k_compact, the host-vector download), updated (deferred: the keyframe's slice)
            if (!DEFER) P.blockSums[cntIdx] = cntDel + cntDelB;
            P.blockUpd[cntIdx] = nupd;
        }
        // (normally) nothing beyond the grid; a deferred launch decides at the head of the loop (the new surfels may reach into the next sub-block)
        if (!(DEFER && spawnWave) && (sb + G) * WSPAN >= n) return;   // (n = E0 for a pending launch's regular waves)
    }
#undef REC_LOCAL
}

template <bool DEFER>
// Register budget (round 6): at 8 waves per SIMD a wave has 80 scalar registers (800 per SIMD / 8 less the trap handler's 16) and the kernel spilled 41 of them to
// lanes of a VGPR: 147 v_readlane / v_writelane instructions, a fifth of its VALU count.  A minimum of 6 waves lets the compiler use 106 SGPRs: no spills, 59 VGPRs
// (the wave still fits the 64-register holes the frame-batched kernels leave), 7 waves per SIMD by the scalar file.  15.5 -> 15.0 us by rocprofv3 beside the
// (faster, round 6) superpixel stage, config 3 +1.2 %, front end +- 0; measured before the superpixel stage was trimmed: +- 0 everywhere.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_fuse(FuseArgs P, FuseFrame F, int nSubHint) {   // by value: kernarg -> SGPRs
    __builtin_amdgcn_s_setprio(3);   // the map chain is sequential per keyframe: issue ahead of the batched kernels' waves
    if (DEFER) {
        if (blockIdx.x == 0) fuse_body<DEFER, true>(P, F, nSubHint, 0u, (int)gridDim.x - 1);   // workgroup 0: the spawn wave (its own instantiation: what it
        else fuse_body<DEFER, false>(P, F, nSubHint, blockIdx.x - 1u, (int)gridDim.x - 1);     // carries through the loop costs the other waves no register)
    } else {
        fuse_body<false, false>(P, F, nSubHint, blockIdx.x, (int)gridDim.x);
    }
}

// ---- dealing the sub-blocks to the XCDs by screen position (round 6) ----------------------------------------------------------------------
// Workgroup g of a launch runs on XCD g % 8, and every XCD has its own L2.  With the sub-blocks handed out in ARRAY order every XCD's waves
// project all over the screen: each of the eight L2s fetched the whole texel map (2.46 MB) and all seed records of the keyframe -- about 17 of
// the 52 MB k_fuse read per launch (round 5 counters).  Array neighbours do project to neighbouring pixels (creation order = superpixel
// raster order of the source keyframe), so a sub-block's in-view surfels cover a narrow band of image rows; k_fuse leaves that band's mean row
// as the sub-block's screen key, and this pass -- one workgroup beside the compaction, one launch behind the fusion that measured the keys --
// sorts the sub-blocks by key and cuts the list into eight equal runs: XCD x gets the x-th run (adaptive bands: equal numbers of in-view
// sub-blocks whatever the distribution of the rows), in row order, followed by its share of the sub-blocks with nothing in view, so that
// every XCD runs exactly G / 8 waves and the heavy ones are dispatched first.  Counting sort on the 255 key values in the LDS.
// The table is a hint: whatever the keys are, deal[] is a permutation of 0 .. G - 1 (G a multiple of 8).
template <int NT>
__device__ __forceinline__ void deal_subblocks(const unsigned *keys, int G, unsigned *deal, unsigned *s_hist, unsigned *s_off, unsigned *s_wave, unsigned *s_aux) {
    static_assert(NT == 256, "one histogram bin per thread");
    // Thread t owns the 32 consecutive sub-blocks [base + 32 t, base + 32 t + 32) of a chunk of 8192 (a map of 1 M surfels is one chunk): their keys
    // arrive as eight 16-byte loads issued together and are packed to one byte each.  The two passes walk the eight registers in ROLLED loops (the
    // group is rotated by one register per step and is itself again after eight) -- four waves that run alone on their SIMDs pay every dependent LDS
    // round trip and every instruction (4 cycles each) in full, so: no returning atomic in pass 1, four in flight per step in pass 2 together with
    // the per-key table word that says where the key's ranks go, and the sub-blocks with nothing in view (a third to two thirds of the map, all
    // in bin 255) are ranked by prefix sums instead of atomics.  (Round 6 history: straight-line code for 32 keys per thread was 40 KB of
    // instructions executed once -- 15 us beside the compaction's 8; one key per loop trip with two dependent LDS round trips each -- 18 us; a
    // seven-compare search for the XCD of every rank -- 12 us.)
    const unsigned t = threadIdx.x;
    constexpr int CH = 32 * NT;
    auto load_pack = [&](int base, unsigned (&kp)[8]) {
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const uint4 *>(keys + base + 32 * (int)t + 4 * q);   // (the key plane is padded by > 8192 entries)
#pragma unroll
        for (int q = 0; q < 8; q++) kp[q] = min(v[q].x, 255u) | (min(v[q].y, 255u) << 8) | (min(v[q].z, 255u) << 16) | (min(v[q].w, 255u) << 24);
    };
    auto next_word = [&](unsigned (&kp)[8]) -> unsigned {   // the group's first register; the group rotated by one
        const unsigned w = kp[0];
#pragma unroll
        for (int q = 0; q < 7; q++) kp[q] = kp[q + 1];
        kp[7] = w;
        return w;
    };
    // pass 1 over a chunk: histogram of the in-view keys; returns the thread's number of sub-blocks with nothing in view
    auto count_chunk = [&](int base, unsigned (&kp)[8], bool hist) -> unsigned {
        unsigned fc = 0;
#pragma unroll 1
        for (int d = 0; d < 8; d++) {
            const unsigned w = next_word(kp);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned key = (w >> (8 * b)) & 255u;
                const bool in = base + 32 * (int)t + 4 * d + b < G;
                if (hist && in && key != 255u) atomicAdd(&s_hist[key], 1u);
                fc += in && key == 255u ? 1u : 0u;
            }
        }
        return fc;
    };
    unsigned k0[8];
    load_pack(0, k0);
    s_hist[t] = 0;
    __syncthreads();
    const unsigned fc0 = count_chunk(0, k0, true);
    for (int base = CH; base < G; base += CH) { unsigned kc[8]; load_pack(base, kc); (void)count_chunk(base, kc, true); }
    __syncthreads();
    // in-view rank r -> XCD x = floor(8 r / NI): the ranks [inS(x), inS(x + 1)); the fillers take what is left of each XCD's G / 8 waves: XCD x the
    // filler ranks [outS(x), outS(x + 1)), outS(x) = x G / 8 - inS(x)
    unsigned NI, fTot0, ex, fb0;   // NI: sub-blocks with something in view
    block_excl_scan_pair(t < 255u ? s_hist[t] : 0u, fc0, s_wave, &NI, &fTot0, ex, fb0);
    const unsigned gs = (unsigned)G >> 3;
    auto inS = [&](unsigned x) { return (x * NI + 7u) >> 3; };
    auto outS = [&](unsigned x) { return x * gs - inS(x); };
    {   // per key: the XCD its first rank falls into and how many more ranks fit there (nearly always all of the bin's); the running rank of the
        // bin counts from that XCD's start, so an atomic's return value IS the place in the XCD's run
        unsigned x0 = 0;
#pragma unroll
        for (unsigned y = 1; y < 8; y++) x0 += ex >= inS(y) ? 1u : 0u;
        s_off[t] = ex - inS(x0);
        s_hist[t] = x0 | ((inS(x0 + 1) - inS(x0)) << 3);
        if (t < 9) s_aux[t] = inS(t);   // (a rank that crosses into the next XCD's run looks its bounds up here)
    }
    __syncthreads();
    unsigned fillBase = 0;
    auto place_chunk = [&](int base, unsigned (&kp)[8], unsigned fb, unsigned ftot) {   // fb: the rank of the thread's first filler inside the chunk (array order)
        fb += fillBase;
        fillBase += ftot;
        unsigned xf = 0;
#pragma unroll
        for (unsigned y = 1; y < 8; y++) xf += fb >= outS(y) ? 1u : 0u;
        unsigned jf = (inS(xf + 1) - inS(xf)) + (fb - outS(xf));
#pragma unroll 1
        for (int d = 0; d < 8; d++) {
            const unsigned w = next_word(kp);
            unsigned rr[4], tb[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned key = (w >> (8 * b)) & 255u;
                rr[b] = 0; tb[b] = 0;
                if (base + 32 * (int)t + 4 * d + b < G && key != 255u) { rr[b] = atomicAdd(&s_off[key], 1u); tb[b] = s_hist[key]; }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                // branch-free: the place of an in-view sub-block (its rank inside the XCD the key's table word names) or of a filler (the thread's
                // running filler place) selected per lane; only a rank that crosses into the next XCD's run -- rare -- takes a (wave-level) slow path.
                // (With a divergent branch per kind every step ran both sides one after the other: 6 us for this pass.)
                const unsigned key = (w >> (8 * b)) & 255u;
                const int sb = base + 32 * (int)t + 4 * d + b;
                const bool in = sb < G, iv = in && key != 255u, fl = in && key == 255u;
                unsigned x = tb[b] & 7u, room = tb[b] >> 3, j = rr[b];
                if (__builtin_expect(__ballot((iv && j >= room) || (fl && jf >= gs)) != 0ull, 0)) {
                    if (iv) while (j >= room) { j -= room; x++; room = s_aux[x + 1] - s_aux[x]; }      // the bin straddles two XCDs' runs
                    if (fl) while (jf >= gs) { xf++; jf = s_aux[xf + 1] - s_aux[xf]; }                  // this XCD's run is full: on to the next one with room for fillers
                }
                const unsigned at = (iv ? x : xf) * gs + (iv ? j : jf);
                if (in) deal[at] = (unsigned)sb;
                jf += fl ? 1u : 0u;
            }
        }
    };
    place_chunk(0, k0, fb0, fTot0);
    for (int base = CH; base < G; base += CH) {
        unsigned kc[8];
        load_pack(base, kc);
        const unsigned fc = count_chunk(base, kc, false);
        unsigned ftot;
        const unsigned fb = block_excl_scan(fc, s_wave, &ftot);
        place_chunk(base, kc, fb, ftot);
    }
    __syncthreads();   // (the caller reuses the LDS arrays)
}
__global__ __launch_bounds__(256) void k_deal(const unsigned *keys, int G, unsigned *deal) {
    __shared__ unsigned s_hist[256], s_off[256], s_wave[33], s_aux[16];
    deal_subblocks<256>(keys, G, deal, s_hist, s_off, s_wave, s_aux);
}

constexpr int TAIL_MAX_HOPS = 64;   // relay hops resolved per hole before the literal loop takes over (k_compact)

// =============================================================================================
// Classic compaction (one launch per keyframe, behind k_fuse<false>)
// =============================================================================================
// Resident-map compaction (SurfelMapping.cpp:366-391) with prefix sums.  Deleted slots ascending d_0..d_{D-1};
// new surfel k -> d_{D-1-k} while any remain, else appended.  If D > K the literal `while` loop (:386-390) moves,
// at step i = 1..R (R = D-K), the element at position n-i into the i-th largest leftover hole; a hole inside the
// tail [nFinal, n) only relays what lands in it.  So the a-th smallest leftover hole (< nFinal) finally receives
// resolve(nFinal + a), resolve(p) = p if p is live, else resolve(n - rank_desc(p)): a short upward chain.

// k_compact: everything after k_fuse in ONE launch.
//   every workgroup : exclusive scan of the per-chunk deleted counts (each workgroup scans the <= cap/1024 partials itself,
//                     so there is no inter-workgroup dependency), then lists the deleted slots of its own chunks in
//                     ascending order (write-through stores);
//   last workgroup  : initializeSurfels (:285-331) = ordered emission of the seed candidates the fuse step did not consume,
//                     counters, new surfel k -> k-th largest deleted slot else appended, tail sources resolved and moved.
// mode 1 (host-vector drop-in, one workgroup): emission and counters only; the caller compacts (SurfelMapping.cpp:366-391).
constexpr int SMALL_D = 512, SMALL_CHUNKS = 48;   // single-workgroup path: few deletions in few chunks

// What k_compact reads of the handle: a slim copy of SfDev with the keyframe's slot folded in (round 6; 36 instead of ~150 dwords of kernel
// arguments).  Unlike k_fuse in round 5 the kernel's register count did not follow (182 VGPRs in the 8-word form: the prefetched counts, flag
// words and candidate records of the steady-state path are live together by design -- every load of the chain leaves before the first use; a
// 128-register cap spills 78 of them to scratch), so what the slim arguments buy is the shorter scalar prologue only.
struct CompactArgs {
    int nseeds, flagStride, dealG, _pad;
    unsigned long long cap;
    MapSoA map;
    long long *ctr;
    const uint8_t *candOk, *fused;     // this keyframe's slot
    const msl_surfel *cand;            // ...
    msl_surfel *newSurfels;
    unsigned *blockSums, *blockUpd, *delList, *srcOf, *tickets, *delU, *delUCount;
    const unsigned *sbKeys; unsigned *deal;
};
__host__ inline CompactArgs compact_args(const SfDev &P, int slot) {
    CompactArgs A;
    A.nseeds = P.nseeds; A.flagStride = P.flagStride; A.dealG = P.dealG; A._pad = 0; A.cap = P.cap; A.map = P.map; A.ctr = P.ctr;
    A.candOk = P.candOk + (size_t)slot * P.flagStride; A.fused = P.fused + (size_t)slot * P.flagStride; A.cand = P.cand + (size_t)slot * P.nseeds;
    A.newSurfels = P.newSurfels; A.blockSums = P.blockSums; A.blockUpd = P.blockUpd; A.delList = P.delList; A.srcOf = P.srcOf; A.tickets = P.tickets;
    A.delU = P.delU; A.delUCount = P.delUCount; A.sbKeys = P.sbKeys; A.deal = P.deal;
    return A;
}

// LDS is kept to ~3.5 KB: on a GPU saturated by the LDS-heavy batched kernels a larger workgroup waits for a CU to drain.
// NQW: the seed flags of a thread arrive in ONE round trip as NQW 32-bit words per array (8: <= 32 seeds per thread, 640 x 480 has 19; 24: <= 96,
// 1280 x 960 has 76); 0: the generic loop (any size or alignment).  Separate instantiations: the 24-word form costs 45 registers more (227
// against 182), which the common geometry need not carry.
template <int NQW>
__global__ __launch_bounds__(256) void k_compact(CompactArgs P, int mode) {
    constexpr int NT = 256, TILE = 4 * NT;
    __shared__ unsigned s_wave[33];
    __shared__ unsigned s_dl[SMALL_D];          // single-workgroup paths: the ascending deleted-slot list stays in LDS
    __shared__ unsigned s_raw[LIST_D];          // fastest path: k_fuse's unordered hand-over list
    __shared__ unsigned s_last, s_upd, s_nzChunks, s_base, s_cntChunk;
    __shared__ unsigned s_nzIdx[SMALL_CHUNKS], s_nzCnt[SMALL_CHUNKS], s_nzSortIdx[SMALL_CHUNKS], s_nzSortCnt[SMALL_CHUNKS];   // sub-blocks with deletions
    __shared__ int s_fallback;
    __builtin_amdgcn_s_setprio(3);   // latency-critical serial chain next to the throughput-oriented batched kernels
    // Steady state (k_fuse handed over <= LIST_D deleted slots): workgroup 0 does everything alone; the others leave after one load
    // instead of fetching the partials and flags as well.
    // (round 6) the second workgroup first deals the sub-blocks for the next fuse launch from the screen keys this keyframe's launch left
    // (deal_subblocks above) -- beside workgroup 0's compaction, not behind it
    if (mode == 0 && blockIdx.x == 1 && P.dealG > 0) deal_subblocks<NT>(P.sbKeys, P.dealG, P.deal, s_raw, s_dl, s_wave, s_nzIdx);
    if (mode == 0 && blockIdx.x != 0 && *P.delUCount <= LIST_D) return;
    // Loads that do not depend on anything are issued first; in particular every workgroup already fetches the seed flags
    // the continuation needs, so the continuing workgroup does not start its dependent chain with a cold memory round trip.
    const uint4 bs0 = *reinterpret_cast<const uint4 *>(P.blockSums + 4 * threadIdx.x);   // first tile of chunk partials
    constexpr int NBU = 8;
    uint4 bu[NBU];   // the first 8192 per-sub-block updated counts = a map of 1 M surfels in one trip (arrays are padded by >= 8192 zeroed entries)
#pragma unroll
    for (int q = 0; q < NBU; q++) bu[q] = *reinterpret_cast<const uint4 *>(P.blockUpd + TILE * q + 4 * threadIdx.x);
    static_assert(LIST_D == NT, "one hand-over entry per thread");
    const unsigned du = P.delU[threadIdx.x];
    const unsigned dHand = *P.delUCount;   // k_fuse's running total of deleted slots = D of this keyframe
    const long long n = P.ctr[0];
    const bool bad = P.ctr[5] == 20;
    const uint8_t *candOk = P.candOk, *fused = P.fused;
    const int per = (((P.nseeds + NT - 1) / NT) + 3) & ~3;      // seeds per thread, multiple of 4: aligned 32-bit flag loads
    const int s0 = threadIdx.x * per, s1 = min(s0 + per, P.nseeds);
    unsigned cnt = 0;
    unsigned long long emit = 0, emitHi = 0;   // bit j: seed s0 + j spawns a surfel (emit: j < 64; emitHi: 64 <= j < 128)
    const msl_surfel *cand = P.cand;
    const bool aligned4 = (P.nseeds & 3) == 0 && ((reinterpret_cast<size_t>(candOk) | reinterpret_cast<size_t>(fused)) & 3) == 0;
    // all flag words of the thread in ONE round trip: 8 words each for <= 32 seeds per thread (640 x 480: 19), 24 words for <= 96 (1280 x 960: 76 --
    // round 3 walked the seeds beyond the 64th one by one, two dependent byte loads each, and the kernel took 30 us at that size)
    auto flags_in_one_trip = [&](auto nqTag) {
        constexpr int NQ = decltype(nqTag)::value;
        unsigned cw[NQ], fw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int i = s0 + 4 * q;
            const bool in = 4 * q < per && i < s1;
            cw[q] = in ? *reinterpret_cast<const unsigned *>(candOk + i) : 0u;
            fw[q] = in ? *reinterpret_cast<const unsigned *>(fused + i) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NQ; q += 8)   // (a common use per group of loads keeps them from being sunk into their consumers)
            asm volatile("" ::"v"(cw[q]), "v"(cw[q + 1]), "v"(cw[q + 2]), "v"(cw[q + 3]), "v"(cw[q + 4]), "v"(cw[q + 5]), "v"(cw[q + 6]), "v"(cw[q + 7]),
                         "v"(fw[q]), "v"(fw[q + 1]), "v"(fw[q + 2]), "v"(fw[q + 3]), "v"(fw[q + 4]), "v"(fw[q + 5]), "v"(fw[q + 6]), "v"(fw[q + 7]));
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned e = (s0 + 4 * q + j < s1 && ((cw[q] >> (8 * j)) & 0xFF) && !((fw[q] >> (8 * j)) & 0xFF)) ? 1u : 0u;
                cnt += e;
                if (4 * q + j < 64) emit |= (unsigned long long)e << ((4 * q + j) & 63);
                else emitHi |= (unsigned long long)e << ((4 * q + j - 64) & 63);
            }
    };
    if constexpr (NQW > 0) {
        (void)aligned4;   // (the host picked this instantiation: per <= 4 NQW and aligned flag arrays)
        flags_in_one_trip(std::integral_constant<int, NQW>{});
    } else {
        for (int i = s0; i < s1; i += 4) {
            unsigned c4, f4;
            if (i + 4 <= P.nseeds && ((reinterpret_cast<size_t>(candOk + i) | reinterpret_cast<size_t>(fused + i)) & 3) == 0) {
                c4 = *reinterpret_cast<const unsigned *>(candOk + i); f4 = *reinterpret_cast<const unsigned *>(fused + i);
            } else {
                c4 = f4 = 0;
                for (int j = 0; j < 4 && i + j < P.nseeds; j++) { c4 |= (unsigned)candOk[i + j] << (8 * j); f4 |= (unsigned)fused[i + j] << (8 * j); }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned e = (i + j < s1 && ((c4 >> (8 * j)) & 0xFF) && !((f4 >> (8 * j)) & 0xFF)) ? 1u : 0u;
                cnt += e;
                if (i + j - s0 < 64) emit |= (unsigned long long)e << (i + j - s0);
                else if (i + j - s0 < 128) emitHi |= (unsigned long long)e << (i + j - s0 - 64);
            }
        }
    }
    const long long nblk = (n + SUB_ITEMS - 1) / SUB_ITEMS;   // sub-block partials written by k_fuse
    const long long nWg = nblk;   // k_fuse waves (blockUpd entries): one per sub-block
    // The prefetched updated counts are folded into ONE register here, as soon as the flag words have been consumed (they were requested before
    // them, so they have arrived): 32 registers that stayed live down to the continuation otherwise -- the kernel's register count decides how soon a
    // workgroup of this latency-critical launch finds room on a CU that the frame-batched kernels fill.  Round 6: 182 -> 87 VGPRs with this and without
    // the prefetch of the thread's first two candidate surfels into registers (51 registers, for a round trip that only keyframes with new surfels
    // pay): k_compact 12.2 -> 9.1 us in the timed region (its time alone is unchanged), config 3 +2 %, moving camera 14.8 -> 16.1 k frames/s.
    unsigned updPart = 0;
#pragma unroll
    for (int q = 0; q < NBU; q++) {
        const long long c = TILE * q + 4 * threadIdx.x;
        updPart += (c < nWg ? bu[q].x : 0u) + (c + 1 < nWg ? bu[q].y : 0u) + (c + 2 < nWg ? bu[q].z : 0u) + (c + 3 < nWg ? bu[q].w : 0u);
    }
    s_raw[threadIdx.x] = du;
    if (threadIdx.x == 0) { s_upd = 0; s_fallback = 0; s_nzChunks = 0; }
    __syncthreads();
    // k_fuse already counted the deleted slots; when they all fit its hand-over list (the steady state) the per-sub-block
    // counts are not needed at all.  Otherwise one pass over them (4 consecutive per thread and tile) lists the sub-blocks
    // that contain deletions.
    const bool fastest = mode == 0 && dHand <= LIST_D;
    unsigned vsum = 0;
    if (!fastest)
        for (long long t0 = 0; t0 < nblk; t0 += TILE) {
            const long long c = t0 + 4 * threadIdx.x;
            const uint4 v4 = t0 == 0 ? bs0 : *reinterpret_cast<const uint4 *>(P.blockSums + c);
            const unsigned x[4] = {c < nblk ? v4.x : 0u, c + 1 < nblk ? v4.y : 0u, c + 2 < nblk ? v4.z : 0u, c + 3 < nblk ? v4.w : 0u};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x[j] > 0) {
                    vsum += x[j];
                    const unsigned q = atomicAdd(&s_nzChunks, 1u);
                    if (q < SMALL_CHUNKS) { s_nzIdx[q] = (unsigned)(c + j); s_nzCnt[q] = x[j]; }
                }
        }
    unsigned Dtot, Ku, exUnused, pos;
    block_excl_scan_pair(vsum, cnt, s_wave, &Dtot, &Ku, exUnused, pos);   // total deletions + emission scan
    const long long D = fastest ? (long long)dHand : (long long)Dtot;
    // single-workgroup paths: workgroup 0 does everything alone -- no ticket, no write-through list
    const bool small = mode == 0 && !fastest && D <= SMALL_D && s_nzChunks <= SMALL_CHUNKS;
    const bool single = fastest || small;
    if (single && blockIdx.x != 0) return;
    if (mode == 0 && !bad) {
        if (fastest) {
            if (threadIdx.x < D) {   // rank-sort in LDS
                unsigned r = 0;
                for (unsigned j = 0; j < (unsigned)D; j++) r += s_raw[j] < du ? 1u : 0u;
                s_dl[r] = du;
            }
        } else if (small) {
            // few sub-blocks hold all deletions: order them by index (rank sort); a sub-block's offset in the ascending
            // list is the sum of the counts before it -- no scan over the (thousands of) empty sub-blocks
            const unsigned nz = s_nzChunks;
            if (threadIdx.x < nz) {
                const unsigned me = s_nzIdx[threadIdx.x];
                unsigned r = 0;
                for (unsigned j = 0; j < nz; j++) r += s_nzIdx[j] < me ? 1u : 0u;
                s_nzSortIdx[r] = me; s_nzSortCnt[r] = s_nzCnt[threadIdx.x];
            }
            __syncthreads();
            unsigned base = 0;
            for (unsigned it = 0; it < nz; it++) {
                const long long i0 = (long long)s_nzSortIdx[it] * SUB_ITEMS + threadIdx.x;   // one slot per thread: ascending
                const unsigned f = (threadIdx.x < (unsigned)SUB_ITEMS && i0 < n && hot_is_deleted(P.map, i0)) ? 1u : 0u;
                unsigned tt;
                const unsigned w = base + block_excl_scan(f, s_wave, &tt);
                if (f) s_dl[w] = (unsigned)i0;
                base += s_nzSortCnt[it];
            }
        } else {
            // every workgroup lists the deleted slots of its own sub-blocks in ascending order; a sub-block's base offset
            // lives in the registers of the thread that scanned it and is broadcast through one LDS word
            unsigned carry = 0;
            for (long long t0 = 0; t0 < nblk; t0 += TILE) {
                const long long c = t0 + 4 * threadIdx.x;
                const uint4 v4 = t0 == 0 ? bs0 : *reinterpret_cast<const uint4 *>(P.blockSums + c);
                const unsigned v[4] = {c < nblk ? v4.x : 0u, c + 1 < nblk ? v4.y : 0u, c + 2 < nblk ? v4.z : 0u, c + 3 < nblk ? v4.w : 0u};
                unsigned tot;
                const unsigned ex = carry + block_excl_scan(v[0] + v[1] + v[2] + v[3], s_wave, &tot);
                const long long nIter = (min(t0 + TILE, nblk) - t0 - blockIdx.x + gridDim.x - 1) / gridDim.x;
                for (long long it = 0; it < nIter; it++) {
                    const long long b = t0 + blockIdx.x + it * gridDim.x;
                    const int q = (int)(b - t0);
                    if ((int)threadIdx.x == (q >> 2)) {
                        const int comp = q & 3;
                        s_base = ex + (comp > 0 ? v[0] : 0u) + (comp > 1 ? v[1] : 0u) + (comp > 2 ? v[2] : 0u);
                        s_cntChunk = v[comp];
                    }
                    __syncthreads();
                    const unsigned base = s_base, cntChunk = s_cntChunk;
                    if (cntChunk == 0) { __syncthreads(); continue; }   // nothing deleted in this sub-block
                    const long long i0 = b * SUB_ITEMS + threadIdx.x;       // one slot per thread keeps the list ascending
                    const unsigned f = (threadIdx.x < (unsigned)SUB_ITEMS && i0 < n && hot_is_deleted(P.map, i0)) ? 1u : 0u;
                    unsigned tt;
                    const unsigned w = base + block_excl_scan(f, s_wave, &tt);   // (its barriers also protect s_base)
                    if (f) st_agent(&P.delList[w], (unsigned)i0);
                }
                carry += tot;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (mode == 0 && !single && !last_workgroup(&P.tickets[1], &s_last)) return;
    // ================= continuation: one workgroup =================
    // updated count
    {
        unsigned u = updPart;
        for (long long c2 = (long long)NBU * TILE + threadIdx.x; c2 < nWg; c2 += blockDim.x) u += P.blockUpd[c2];
        u = wave_incl_scan(u);                                   // one LDS atomic per wave instead of 256 on one address
        if ((threadIdx.x & 63) == 63 && u) atomicAdd(&s_upd, u);
    }
    // initializeSurfels (:285-331): thread t owns the contiguous seeds [t*per, (t+1)*per); emission order = seed index order
    const long long K = Ku;
    const long long nAfter = mode == 1 ? n : (D >= K ? n - (D - K) : n + (K - D));
    const bool place = mode == 0 && !bad && (unsigned long long)nAfter <= P.cap;
    auto DL = [&](long long j) -> unsigned { return single ? s_dl[j] : ld_agent(&P.delList[j]); };
    if (cnt) {
        auto emit_one = [&](const msl_surfel &e) {
            const long long k = pos++;
            P.newSurfels[k] = e;                    // host-vector mode and debugging read this list
            if (place)                              // new surfel k -> k-th largest deleted slot while any remain, else appended
                store_surfel(P.map, k < D ? (long long)DL(D - 1 - k) : n + (k - D), e);   // (SurfelMapping.cpp:372-384)
        };
        for (unsigned long long m = emit; m; m &= m - 1) emit_one(cand[s0 + __builtin_ctzll(m)]);
        for (unsigned long long mh = emitHi; mh; mh &= mh - 1) emit_one(cand[s0 + 64 + __builtin_ctzll(mh)]);
        for (int i = s0 + 128; i < s1; i++)
            if (candOk[i] && !fused[i]) emit_one(cand[i]);
    }
    __syncthreads();   // s_upd complete; new-surfel stores ordered before the tail moves below (same workgroup)
    if (threadIdx.x == 0) {
        P.ctr[1] = K; P.ctr[2] = D; P.ctr[3] = s_upd; P.ctr[4] = n; P.ctr[6] = nAfter;
        // running totals over all keyframes of this handle (one writer per launch, launches are ordered): bench.py derives the
        // per-keyframe averages of a timed region from their differences
        P.ctr[8] += K; P.ctr[9] += D; P.ctr[10] += s_upd; P.ctr[11] += 1; P.ctr[12] += n;
        if ((unsigned long long)nAfter > P.cap) P.ctr[5] = 20;  // capacity exceeded
    }
    if (!place) { if (threadIdx.x == 0) *P.delUCount = 0; return; }   // (host-vector mode, or the deferred capacity error: the live count stays)
    const long long t0 = threadIdx.x, stride = blockDim.x;
    if (D > K) {
        const long long R = D - K, nFinal = n - R;
        auto lower = [&](long long x) -> long long {   // first index in delList[0..R) with value >= x
            long long lo = 0, hi = R;
            while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)DL(mid) < x) lo = mid + 1; else hi = mid; }
            return lo;
        };
        const long long cntLow = lower(nFinal);
        for (long long a = t0; a < cntLow; a += stride) {
            long long p = nFinal + a;
            int hop = 0;
            for (; hop < TAIL_MAX_HOPS; hop++) {
                const long long lb = lower(p);
                if (lb < R && (long long)DL(lb) == p) p = n - (R - lb);   // relay hole: follow to where its content came from
                else break;
            }
            if (hop == TAIL_MAX_HOPS) s_fallback = 1;   // pathological chain: fall back to the literal loop
            P.srcOf[a] = (unsigned)p;
        }
        __syncthreads();   // also orders the new-surfel stores above before the moves below (same workgroup)
        if (s_fallback) {
            if (threadIdx.x == 0)   // literal back-to-front loop (SurfelMapping.cpp:386-390), pathological delete patterns only
                for (long long i = 1; i <= R; i++) {
                    const long long hole = DL(R - i), src = n - i;
                    if (src != hole) move_surfel(P.map, hole, src);
                }
        } else {
            for (long long a = t0; a < cntLow; a += stride) move_surfel(P.map, (long long)DL(a), (long long)P.srcOf[a]);
        }
    }
    if (threadIdx.x == 0) { P.ctr[0] = nAfter; *P.delUCount = 0; }   // publish the new live count, re-arm the hand-over list
}

// =============================================================================================
// Deferred compaction: the end of a window
// =============================================================================================
// k_defer_tail: what is left to do per keyframe once the window's F fuse launches are through.
//   workgroups 0 .. F - 1          : the updated-surfel count of keyframe f (sum of its per-sub-block counts) -> running total, ctr[3] for the last
//   workgroups F .. F + NFRONT - 1 : the new surfels of the LAST keyframe (there is no next fuse launch to materialise them)
__global__ __launch_bounds__(64) void k_defer_tail(SfDev P, FuseArgs A, int F, unsigned blkStride) {   // A.kf = F
    __builtin_amdgcn_s_setprio(3);
    const unsigned lane = threadIdx.x;
    if ((int)blockIdx.x < F) {
        const int f = (int)blockIdx.x;
        // keyframe f's counts: one per sub-block below the extent its regular waves worked on, and behind them (from index E0 / SUB_ITEMS + 1 on) one per
        // 256 new surfels of keyframe f - 1 that its spawn wave wrote and fused
        const long long E0 = f > 0 ? P.dc->ext[f - 1] : P.dc->ext[0], Kp = f > 0 ? P.dc->ext[f] - E0 : 0;
        const long long nblk = (E0 + SUB_ITEMS - 1) / SUB_ITEMS, x0 = E0 / SUB_ITEMS + 1, x1 = x0 + (Kp + SUB_ITEMS - 1) / SUB_ITEMS;
        const unsigned *bu = P.blockUpd + (size_t)f * blkStride;
        unsigned u = 0;
        for (long long b = lane; b < nblk; b += 64) u += bu[b];
        for (long long b = x0 + lane; b < x1; b += 64) u += bu[b];
        u = wave_incl_scan(u);
        if (lane == 63) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&P.ctr[10]), (unsigned long long)u);
            if (f == F - 1) P.ctr[3] = u;
        }
        return;
    }
    const long long E0 = P.dc->ext[F - 1];
    const long long q = (long long)blockIdx.x - F, sb = E0 / SUB_ITEMS + q;
    if (sb * SUB_ITEMS >= E0 + P.nseeds) return;
    unsigned excl;
    const unsigned K = spawn_count(A, lane, excl);
    if (q == 0 && lane == 0) P.dc->ext[F] = E0 + (long long)K;
    if (K && sb * SUB_ITEMS < E0 + (long long)K) emit_records(A, E0, sb * SUB_ITEMS, SUB_ITEMS / 64, lane, K, excl);
}

// k_replay: the window's F compactions, replayed symbolically by ONE wave.
// Elements are named by their PHYSICAL slot (nothing moved during the window): base elements 0 .. n0 - 1, the k-th new surfel of keyframe
// f = ext[f] + k.  The reference's array ("virtual" order) differs from the identity only where a compaction put something:
//   loc64[p]  = element at virtual position p, with the keyframe (stamp) that put it there      -- only for explicit placements
//   vposD[e]  = virtual position of element e                                                   -- only for elements placed explicitly
//   run f     = the new surfels of keyframe f that were APPENDED: elements ext[f] + k0 + q at virtual positions runV + q, q < runCnt
// (a run is clipped when a later keyframe shortens the array; where a run and an explicit entry both cover a position the later stamp wins).
// Per keyframe: virtual positions of the logged slots -> ascending (LDS rank sort; a bitmap over the positions beyond RP_SORT entries) ->
// new surfel k to the k-th largest hole, else appended (SurfelMapping.cpp:372-384) -> if holes remain, the back-to-front loop (:386-390) as
// k_compact resolves it: the a-th smallest leftover hole below the new end receives resolve(nFinal + a).  At the end every virtual position
// whose element is not already in that physical slot becomes one move (source, destination); k_gather / k_scatter apply them.
// All table traffic is agent-scope (L2): one wave, but its own stores must be what its later loads see.
constexpr int RP_SORT = 1024;          // deleted positions of one keyframe ordered in the LDS up to here
constexpr int RP_HASH = 2048;          // slots of the LDS tables (explicit placements of a window with <= RP_HASH / 2 deletions in all)
constexpr unsigned RP_EMPTY = 0xFFFFFFFFu;
struct ReplayLds {
    unsigned v[RP_SORT + 4], d[RP_SORT];                      // a keyframe's deleted positions: as logged, ascending
    long long ext[DEFER_WIN + 1], runV[DEFER_WIN];
    unsigned runK0[DEFER_WIN], runCnt[DEFER_WIN];
    unsigned dcnt[DEFER_WIN];                                 // deletions per keyframe
    unsigned log[RP_HASH / 2];                                // LDS mode: the whole window's deletion log (fetched in one trip)
    unsigned locK[RP_HASH], locV[RP_HASH], vposK[RP_HASH], vposV[RP_HASH];   // LDS tables (open addressing; locV = element + 1 | stamp << 26)
};
__device__ __forceinline__ unsigned rp_hash(unsigned key) { return (key * 2654435761u) >> 21; }   // 11 bits
static_assert(RP_HASH == 2048, "rp_hash yields 11 bits");
static_assert(SUB_ITEMS == 256 || SUB_ITEMS == 128 || SUB_ITEMS == 64, "k_fuse: four or two records per lane; k_compact lists a sub-block with one thread per slot");

// LDS = true: the window's explicit placements live in two LDS hash tables (few deletions: the steady state; no global round trips inside the
// keyframe loop).  LDS = false: dense global tables indexed by position / element (any number of deletions; agent-scope accesses).
template <bool LDS>
__device__ __forceinline__ void replay_body(const SfDev &P, int F, ReplayLds &S) {
    const unsigned lane = threadIdx.x;
    DeferCtl *dc = P.dc;
    const long long n0 = S.ext[0];
    long long n = n0;
    unsigned nLocKeys = 0, nVposKeys = 0, logBase = 0;
    long long totK = 0, totD = 0, totNb = 0, lastK = 0, lastD = 0, lastNb = 0;
    // ---- the two tables: virtual position -> (element, stamp), element -> virtual position ----
    auto loc_get = [&](unsigned p, unsigned &elem, unsigned &stampOut) -> bool {
        if constexpr (LDS) {
            for (unsigned s = rp_hash(p);; s = (s + 1) & (RP_HASH - 1)) {
                const unsigned k = S.locK[s];
                if (k == RP_EMPTY) return false;
                if (k == p) { const unsigned v = S.locV[s]; elem = (v & 0x3FFFFFFu) - 1u; stampOut = v >> 26; return true; }
            }
        } else {
            const unsigned long long v = ld_agent64(&P.loc64[p]);
            if (!v) return false;
            elem = (unsigned)v - 1u; stampOut = (unsigned)(v >> 32);
            return true;
        }
    };
    auto vpos_get = [&](unsigned id, unsigned &pos) -> bool {
        if constexpr (LDS) {
            for (unsigned s = rp_hash(id);; s = (s + 1) & (RP_HASH - 1)) {
                const unsigned k = S.vposK[s];
                if (k == RP_EMPTY) return false;
                if (k == id) { pos = S.vposV[s]; return true; }
            }
        } else {
            const unsigned v = ld_agent(&P.vposD[id]);
            if (!v) return false;
            pos = v - 1u;
            return true;
        }
    };
    // explicit placement (all lanes call; `on` lanes place): element `elem` now sits at virtual position `pos`
    auto put = [&](bool on, unsigned pos, unsigned elem, unsigned stampNo) {
        if constexpr (LDS) {
            if (on) {
                unsigned s = rp_hash(pos);
                for (;; s = (s + 1) & (RP_HASH - 1)) { const unsigned old = atomicCAS(&S.locK[s], RP_EMPTY, pos); if (old == RP_EMPTY || old == pos) break; }
                S.locV[s] = (elem + 1u) | (stampNo << 26);
                s = rp_hash(elem);
                for (;; s = (s + 1) & (RP_HASH - 1)) { const unsigned old = atomicCAS(&S.vposK[s], RP_EMPTY, elem); if (old == RP_EMPTY || old == elem) break; }
                S.vposV[s] = pos;
            }
        } else {
            if (on) { st_agent64(&P.loc64[pos], (unsigned long long)(elem + 1u) | ((unsigned long long)stampNo << 32)); st_agent(&P.vposD[elem], pos + 1u); }
            const unsigned long long m = __ballot(on);
            if (on) { const unsigned r = lane_rank(m); st_agent(&P.locKeys[nLocKeys + r], pos); st_agent(&P.vposKeys[nVposKeys + r], elem); }
            nLocKeys += (unsigned)__popcll(m); nVposKeys += (unsigned)__popcll(m);
        }
    };
    auto tables_sync = [&]() {   // a keyframe's (or phase's) table stores are complete before anything reads them
        if constexpr (LDS) __syncthreads();
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    };
    unsigned runMask = 0;   // bit g: keyframe g appended a run that still has entries (uniform)
    auto vpos_of = [&](unsigned id) -> unsigned {
        unsigned pos;
        if (vpos_get(id, pos)) return pos;
        if ((long long)id < n0) return id;
        int g = 0;
        for (int q = 1; q < F; q++) if ((long long)id >= S.ext[q]) g = q;   // the keyframe that spawned it
        return (unsigned)(S.runV[g] + ((long long)id - S.ext[g] - (long long)S.runK0[g]));
    };
    // newest run covering p: its keyframe (-1: none) and element
    auto run_of = [&](long long p, int upto, unsigned &elem) -> int {
        for (unsigned m = upto >= 31 ? runMask : (runMask & ((2u << upto) - 1u)); m;) {   // (newest first; the steady state has no runs at all)
            const int g = 31 - __builtin_clz(m);
            m &= ~(1u << g);
            const long long v0 = S.runV[g];
            if (p >= v0 && p < v0 + (long long)S.runCnt[g]) { elem = (unsigned)(S.ext[g] + (long long)S.runK0[g] + (p - v0)); return g; }
        }
        return -1;
    };
    auto loc_of = [&](long long p, int upto) -> unsigned {
        unsigned ee = 0, st = 0, er = 0;
        const bool have = loc_get((unsigned)p, ee, st);
        const int g = run_of(p, upto, er);
        if (have && (g < 0 || st > (unsigned)(g + 1))) return ee;
        return g >= 0 ? er : (unsigned)p;
    };
    for (int f = 0; f < F; f++) {
        const unsigned D = S.dcnt[f];
        const long long K = S.ext[f + 1] - S.ext[f];
        const unsigned stampNo = (unsigned)(f + 1);
        const bool inLds = D <= (unsigned)RP_SORT;
        lastK = K; lastD = D; lastNb = n; totK += K; totD += D; totNb += n;
        // ---- 1. virtual positions of the logged slots ----
        for (unsigned j0 = 0; j0 < D; j0 += 64) {
            const unsigned j = j0 + lane;
            if (j < D) {
                const unsigned vp = vpos_of(LDS ? S.log[logBase + j] : ld_agent(&P.delList[logBase + j]));
                if (inLds) S.v[j] = vp;
                else atomicOr(&P.bitmap[vp >> 5], 1u << (vp & 31u));
            }
        }
        if (inLds && lane < 4) S.v[D + lane] = 0xFFFFFFFFu;   // padding of the last 16-byte read
        __syncthreads();
        // ---- 2. ascending order ----
        if (inLds) {
            for (unsigned j0 = 0; j0 < D; j0 += 64) {
                const unsigned j = j0 + lane;
                const unsigned v = j < D ? S.v[j] : 0u;
                unsigned r = 0;
                for (unsigned q = 0; q < D; q += 4) {   // (the positions are distinct: the ranks are a permutation)
                    const uint4 x = *reinterpret_cast<const uint4 *>(&S.v[q]);
                    r += (x.x < v ? 1u : 0u) + (x.y < v ? 1u : 0u) + (x.z < v ? 1u : 0u) + (x.w < v ? 1u : 0u);
                }
                if (j < D) S.d[r] = v;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long nw = (n + 31) >> 5;
            unsigned base = 0;
            for (long long w0 = 0; w0 < nw; w0 += 64) {
                const long long w = w0 + lane;
                unsigned bits = w < nw ? ld_agent(&P.bitmap[w]) : 0u;
                const unsigned c = (unsigned)__popc(bits);
                const unsigned incl = wave_incl_scan(c);
                unsigned o = base + incl - c;
                if (bits) st_agent(&P.bitmap[w], 0u);   // clean for the next use
                for (; bits; bits &= bits - 1) st_agent(&P.dBig[o++], (unsigned)(w * 32 + __builtin_ctz(bits)));
                base += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        auto DL = [&](long long j) -> unsigned { return inLds ? S.d[j] : ld_agent(&P.dBig[j]); };
        // ---- 3. new surfel k -> k-th largest hole (SurfelMapping.cpp:372-384) ----
        const long long nPl = K < (long long)D ? K : (long long)D;
        for (long long k0 = 0; k0 < nPl; k0 += 64) {
            const long long k = k0 + lane;
            const bool on = k < nPl;
            put(on, on ? DL((long long)D - 1 - k) : 0u, (unsigned)(S.ext[f] + k), stampNo);
        }
        if (K > (long long)D) {   // the others are appended: a run
            if (lane == 0) { S.runV[f] = n; S.runK0[f] = D; S.runCnt[f] = (unsigned)(K - (long long)D); }
            runMask |= 1u << f;
            n += K - (long long)D;
        } else if ((long long)D > K) {
            // ---- 4. leftover holes: the back-to-front loop of :386-390, per hole ----
            const long long R = (long long)D - K, nFinal = n - R;
            tables_sync();   // (a tail source may be a surfel placed just above)
            auto lower = [&](long long x) -> long long {   // first index in the R smallest holes with value >= x
                long long lo = 0, hi = R;
                while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)DL(mid) < x) lo = mid + 1; else hi = mid; }
                return lo;
            };
            const long long cntLow = lower(nFinal);   // holes below the new end: each receives a tail element
            for (long long a0 = 0; a0 < cntLow; a0 += 64) {
                const long long a = a0 + lane;
                const bool on = a < cntLow;
                long long p = nFinal + (on ? a : 0);
                bool chain = on;
                while (__ballot(chain)) {
                    if (chain) {
                        const long long lb = lower(p);
                        if (lb < R && (long long)DL(lb) == p) p = n - (R - lb);   // a hole inside the tail only relays: follow to where its content comes from
                        else chain = false;
                    }
                }
                const unsigned e = on ? loc_of(p, f) : 0u;
                if constexpr (LDS) __syncthreads();   // (every lane has read the tables before this chunk's placements go in: a destination < nFinal is never a source, but slots move)
                put(on, on ? DL(a) : 0u, e, stampNo);
            }
            __syncthreads();
            if (runMask) {   // runs that reach beyond the new end are clipped
                bool gone = false;
                if ((int)lane <= f && ((runMask >> lane) & 1u) && S.runV[lane] + (long long)S.runCnt[lane] > nFinal) {
                    S.runCnt[lane] = S.runV[lane] >= nFinal ? 0u : (unsigned)(nFinal - S.runV[lane]);
                    gone = S.runCnt[lane] == 0;
                }
                runMask &= ~(unsigned)__ballot(gone);
            }
            n = nFinal;
        }
        tables_sync();
        logBase += D;
    }
    // ---- the moves: every virtual position whose element is not already in that physical slot ----
    const long long nF = n;
    unsigned nMoves = 0;
    auto add_move = [&](bool on, unsigned dst, unsigned src) {
        const unsigned long long m = __ballot(on);
        if (on) { const unsigned r = nMoves + lane_rank(m); P.moveDst[r] = dst; P.srcOf[r] = src; }
        nMoves += (unsigned)__popcll(m);
    };
    for (int g = 0; g < F; g++) {   // appended runs first, while the explicit table is intact
        const unsigned cnt = ((runMask >> g) & 1u) ? S.runCnt[g] : 0u;
        for (unsigned q0 = 0; q0 < cnt; q0 += 64) {
            const unsigned q = q0 + lane;
            bool on = q < cnt;
            const long long p = S.runV[g] + q;
            const unsigned id = (unsigned)(S.ext[g] + (long long)S.runK0[g] + q);
            unsigned ee = 0, st = 0;
            if (on && loc_get((unsigned)p, ee, st) && st > (unsigned)(g + 1)) on = false;   // a later explicit placement owns p
            if (on && (long long)id == p) on = false;
            add_move(on, (unsigned)p, id);
        }
    }
    // explicit placements: a stale one (a later run covers its position) or one beyond the final end is dropped
    auto explicit_move = [&](bool on, unsigned p, unsigned id, unsigned st) {
        unsigned er = 0;
        if (on) { const int g = run_of((long long)p, F - 1, er); if (g >= 0 && (unsigned)(g + 1) > st) on = false; }
        if (on && (long long)p >= nF) on = false;
        if (on && id == p) on = false;
        add_move(on, p, id);
    };
    if constexpr (LDS) {
        for (unsigned s0 = 0; s0 < (unsigned)RP_HASH; s0 += 64) {
            const unsigned k = S.locK[s0 + lane], v = S.locV[s0 + lane];
            explicit_move(k != RP_EMPTY, k, (v & 0x3FFFFFFu) - 1u, v >> 26);
        }
    } else {
        for (unsigned j0 = 0; j0 < nLocKeys; j0 += 64) {   // (a position may be listed more than once: cleared at its first visit)
            const unsigned j = j0 + lane;
            bool on = j < nLocKeys;
            const unsigned p = on ? ld_agent(&P.locKeys[j]) : 0u;
            const unsigned long long v = on ? ld_agent64(&P.loc64[p]) : 0ull;
            on = on && v != 0ull;
            if (on) st_agent64(&P.loc64[p], 0ull);
            explicit_move(on, p, (unsigned)v - 1u, (unsigned)(v >> 32));
        }
        for (unsigned j0 = 0; j0 < nVposKeys; j0 += 64) { const unsigned j = j0 + lane; if (j < nVposKeys) st_agent(&P.vposD[ld_agent(&P.vposKeys[j])], 0u); }
    }
    if (lane < DEFER_WIN) dc->delCnt[lane] = 0;   // the next window starts with empty logs
    if (lane == 0) {
        dc->nMoves = nMoves;
        P.ctr[0] = nF; P.ctr[1] = lastK; P.ctr[2] = lastD; P.ctr[4] = lastNb; P.ctr[6] = nF;
        P.ctr[8] += totK; P.ctr[9] += totD; P.ctr[11] += F; P.ctr[12] += totNb;
    }
}

// k_replay: the window's F compactions, replayed symbolically by ONE wave.
// Elements are named by their PHYSICAL slot (nothing moved during the window): base elements 0 .. n0 - 1, the k-th new surfel of keyframe
// f = ext[f] + k.  The reference's array ("virtual" order) differs from the identity only where a compaction put something:
//   loc   : virtual position -> element, with the keyframe (stamp) that put it there      -- only explicit placements
//   vpos  : element -> virtual position                                                   -- only elements placed explicitly
//   run f : the new surfels of keyframe f that were APPENDED: elements ext[f] + k0 + q at virtual positions runV + q, q < runCnt
// (a run is clipped when a later keyframe shortens the array; where a run and an explicit entry both cover a position the later stamp wins).
// Per keyframe: virtual positions of the logged slots -> ascending (LDS rank sort; a bitmap over the positions beyond RP_SORT entries) ->
// new surfel k to the k-th largest hole, else appended (SurfelMapping.cpp:372-384) -> if holes remain, the back-to-front loop (:386-390) as
// k_compact resolves it: the a-th smallest leftover hole below the new end receives resolve(nFinal + a).  At the end every virtual position
// whose element is not already in that physical slot becomes one move (source, destination); k_gather / k_scatter apply them.
// Checked against the literal loop by a host model of exactly this scheme (tests/test_replay_model.py) and by the GPU parity tests.
__global__ __launch_bounds__(64) void k_replay(SfDev P, int F) {
    __shared__ __attribute__((aligned(16))) ReplayLds S;
    __builtin_amdgcn_s_setprio(3);   // one wave on the latency-critical map stream, next to the throughput-oriented batched kernels
    const unsigned lane = threadIdx.x;
    DeferCtl *dc = P.dc;
    if ((int)lane <= F) S.ext[lane] = dc->ext[lane];
    if (lane < DEFER_WIN) { S.runCnt[lane] = 0; S.runV[lane] = 0; S.runK0[lane] = 0; }
    const unsigned dmine = (int)lane < F ? ld_agent(&dc->delCnt[lane]) : 0u;
    if (lane < DEFER_WIN) S.dcnt[lane] = dmine;
    const unsigned dsum = wave_incl_scan(dmine);
    const unsigned totalD = (unsigned)__builtin_amdgcn_readlane((int)dsum, 63);
    const bool useLds = totalD <= (unsigned)RP_HASH / 2 && P.cap < (1ull << 26) - 1;
    if (useLds) {
        for (unsigned j = lane; j < totalD; j += 64) S.log[j] = ld_agent(&P.delList[j]);   // (all requests leave together)
        for (unsigned s = lane; s < (unsigned)RP_HASH; s += 64) { S.locK[s] = RP_EMPTY; S.vposK[s] = RP_EMPTY; }
    }
    __syncthreads();
    if (useLds) replay_body<true>(P, F, S);
    else replay_body<false>(P, F, S);
}

// The window's moves: all sources first (a destination may be another move's source), then all destinations.
__global__ __launch_bounds__(256) void k_gather(SfDev P) {
    const MapSoA &M = P.map;
    const unsigned nM = P.dc->nMoves;
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < nM; j += gridDim.x * 256) {
        const unsigned s = P.srcOf[j];
        const HotPk h = M.hot[s];
        const ColdRec c = cold_load(M.cold + s);
        P.stageHot[j] = h; cold_store(P.stageCold + j, c);
        if (h.tl == HOT_WIDE) { P.stageUtl[2 * (size_t)j] = M.utlWide[2 * (size_t)s]; P.stageUtl[2 * (size_t)j + 1] = M.utlWide[2 * (size_t)s + 1]; }
        if (c.rgbf & COLD_WIDE) for (int q = 0; q < 3; q++) P.stageRgb[3 * (size_t)j + q] = M.rgbWide[3 * (size_t)s + q];
    }
}
__global__ __launch_bounds__(256) void k_scatter(SfDev P) {
    const MapSoA &M = P.map;
    const unsigned nM = P.dc->nMoves;
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < nM; j += gridDim.x * 256) {
        const unsigned d = P.moveDst[j];
        const HotPk h = P.stageHot[j];
        const ColdRec c = cold_load(P.stageCold + j);
        M.hot[d] = h; cold_store(M.cold + d, c);
        if (h.tl == HOT_WIDE) { M.utlWide[2 * (size_t)d] = P.stageUtl[2 * (size_t)j]; M.utlWide[2 * (size_t)d + 1] = P.stageUtl[2 * (size_t)j + 1]; }
        if (c.rgbf & COLD_WIDE) for (int q = 0; q < 3; q++) M.rgbWide[3 * (size_t)d + q] = P.stageRgb[3 * (size_t)j + q];
    }
}

// ---- map maintenance: ordered selection of surfels by a predicate -------------------------------
// mode 0: updateTimes > 0 && lastUpdate == arg (moveAddSurfels, src/SurfelMapping.cpp:213)   mode 1: updateTimes >= arg (Stop, :68)
__device__ __forceinline__ bool select_pred(const HotRec &h, int mode, int arg) {
    return mode == 0 ? (h.updateTimes > 0 && h.lastUpdate == arg) : (h.updateTimes >= arg);
}
__global__ __launch_bounds__(256) void k_select_count(SfDev P, int mode, int arg) {
    __shared__ unsigned s_c;
    const long long n = P.ctr[0];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        if (threadIdx.x == 0) s_c = 0;
        __syncthreads();
        unsigned c = 0;
        for (int k = 0; k < SCAN_ITEMS / 256; k++) {
            const long long i = b * SCAN_ITEMS + k * 256 + threadIdx.x;
            if (i < n && select_pred(hot_load(P.map, i), mode, arg)) c++;
        }
        if (c) atomicAdd(&s_c, c);
        __syncthreads();
        if (threadIdx.x == 0) P.blockSums[b] = s_c;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_select_scan(SfDev P) {   // one workgroup: exclusive scan of the chunk counts, total -> ctr[7]
    __shared__ unsigned s_wave[17];
    const long long n = P.ctr[0];
    const int nblk = (int)((n + SCAN_ITEMS - 1) / SCAN_ITEMS);
    unsigned carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const unsigned v = b < nblk ? P.blockSums[b] : 0;
        unsigned tot;
        const unsigned ex = carry + block_excl_scan(v, s_wave, &tot);
        if (b < nblk) P.blockSums[b] = ex;
        carry += tot;
    }
    if (threadIdx.x == 0) P.ctr[7] = carry;
}
__global__ __launch_bounds__(256) void k_select_write(SfDev P, int mode, int arg, msl_surfel *out, int markDeleted) {
    __shared__ unsigned s_wave[17];
    const long long n = P.ctr[0];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        unsigned base = P.blockSums[b];
        for (int k = 0; k < SCAN_ITEMS / 256; k++) {           // 256 consecutive surfels per round keep the map order
            const long long i = b * SCAN_ITEMS + k * 256 + threadIdx.x;
            HotRec h{};
            unsigned tl = 0;
            if (i < n) { tl = P.map.hot[i].tl; h = hot_load(P.map, i); }
            const bool sel = i < n && select_pred(h, mode, arg);
            unsigned tot;
            const unsigned pos = base + block_excl_scan(sel ? 1u : 0u, s_wave, &tot);
            if (sel) {
                msl_surfel e;
                load_surfel(P.map, i, h, e);
                out[pos] = e;
                if (markDeleted) hot_mark_deleted(P.map, i, tl);   // "Delete the surfel from the local point" (:224)
            }
            base += tot;
        }
    }
}
__global__ void k_add_ctr(long long *ctr, long long add) {
    if (threadIdx.x == 0) { ctr[0] += add; ctr[4] = ctr[0]; ctr[6] = ctr[0]; }
}
__global__ __launch_bounds__(256) void k_aos_to_soa_at(MapSoA M, const msl_surfel *src, long long n, const long long *ctr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_surfel(M, ctr[0] + i, src[i]);
}

// AoS <-> SoA conversion for upload / download / host-vector mode
__global__ __launch_bounds__(256) void k_aos_to_soa(MapSoA M, const msl_surfel *src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_surfel(M, i, src[i]);
}
__global__ __launch_bounds__(256) void k_soa_to_aos(MapSoA M, msl_surfel *dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HotRec h = hot_load(M, i);
    msl_surfel e;
    load_surfel(M, i, h, e);
    dst[i] = e;
}
// wide: -1 = leave the wide-record flags ctr[13] alone (upload: k_aos_to_soa has just set them if needed), otherwise the restored snapshot's flags
__global__ void k_set_ctr(long long *ctr, long long n, unsigned *delUCount, int wide) {
    if (threadIdx.x == 0) {
        delUCount[0] = 0;
        ctr[0] = n; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; ctr[4] = n; ctr[6] = n; ctr[7] = 0;
        if (wide >= 0) ctr[13] = wide;
    }
}

// Host-vector mode, sparse case: the records this keyframe touched (updated: lastUpdate == ref; deleted: updateTimes == 0) of the sub-blocks that
// report any, as a compact list {index, reference-layout record}.  One wave per sub-block; slots by one atomic per wave.
__global__ __launch_bounds__(64) void k_collect_changed(SfDev P, int ref, long long n, unsigned *count, unsigned *idxOut, msl_surfel *recOut, unsigned capOut) {
    const long long sb = blockIdx.x;
    if (!(P.blockSums[sb] | P.blockUpd[sb])) return;
    const unsigned lane = threadIdx.x;
    for (int k = 0; k < SUB_ITEMS / 64; k++) {
        const long long i = sb * SUB_ITEMS + k * 64 + lane;
        HotRec h; h.px = h.py = h.pz = 0; h.updateTimes = 1; h.lastUpdate = ref - 1;
        if (i < n) h = hot_load(P.map, i);
        const bool ch = i < n && (h.updateTimes == 0 || h.lastUpdate == ref);
        const unsigned long long m = __ballot(ch);
        if (!m) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (ch) {
            const unsigned j = base + lane_rank(m);
            if (j < capOut) { msl_surfel e; load_surfel(P.map, i, h, e); recOut[j] = e; idxOut[j] = (unsigned)i; }
        }
    }
}

__global__ void k_empty(int grid_dummy) { (void)grid_dummy; }

}  // namespace

namespace msl {
namespace sf {

void map_launch_deal(hipStream_t st, const SfDev &P) { hipLaunchKernelGGL(k_deal, dim3(1), dim3(256), 0, st, P.sbKeys, P.dealG, P.deal); }
// Test hook: the dealing of G sub-blocks (a multiple of 8) with the given screen keys, host arrays, synchronous.
int map_debug_deal(const uint32_t *keys_host, int G, uint32_t *deal_host) {
    if (!keys_host || !deal_host || G < 8 || (G & 7)) return MSL_ERR_INVALID;
    unsigned *dk = nullptr, *dd = nullptr;
    const size_t pad = (size_t)G + 8192;   // the key plane of a handle is padded the same way (whole chunks are loaded)
    MSL_HIP_TRY(hipMalloc(&dk, sizeof(unsigned) * pad)); MSL_HIP_TRY(hipMalloc(&dd, sizeof(unsigned) * (G + 64)));
    MSL_HIP_TRY(hipMemset(dk, 0xFF, sizeof(unsigned) * pad)); MSL_HIP_TRY(hipMemset(dd, 0xFF, sizeof(unsigned) * G));
    MSL_HIP_TRY(hipMemcpy(dk, keys_host, sizeof(unsigned) * G, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_deal, dim3(1), dim3(256), 0, 0, dk, G, dd);
    MSL_HIP_TRY(hipMemcpy(deal_host, dd, sizeof(unsigned) * G, hipMemcpyDeviceToHost));
    if (const char *reps = getenv("MSL_DEAL_REPS")) {   // experiment: mean time of the dealing kernel alone (HIP events around back-to-back launches)
        const int n = atoi(reps);
        hipEvent_t a, b;
        MSL_HIP_TRY(hipEventCreate(&a)); MSL_HIP_TRY(hipEventCreate(&b));
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_deal, dim3(1), dim3(256), 0, 0, dk, G, dd);
        MSL_HIP_TRY(hipEventRecord(a, 0));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_deal, dim3(1), dim3(256), 0, 0, dk, G, dd);
        MSL_HIP_TRY(hipEventRecord(b, 0));
        MSL_HIP_TRY(hipEventSynchronize(b));
        float ms = 0; MSL_HIP_TRY(hipEventElapsedTime(&ms, a, b));
        fprintf(stderr, "k_deal G=%d: %.2f us per launch (back to back, %d launches)\n", G, ms * 1e3f / n, n);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    (void)hipFree(dk); (void)hipFree(dd);
    return MSL_OK;
}
void map_launch_fuse(KernelProfiler &prof, hipStream_t st, const SfDev &P, int slot, const FrameDev &F, int nSubGrid, int nSubHint, bool deferred, bool dealt) {
    const FuseArgs A = fuse_args(P, slot, deferred, dealt);
    const FuseFrame FF = fuse_frame(F, P.frames + slot);
    if (deferred) MSL_SF_LAUNCH(prof, SK_FUSE, st, k_fuse<true>, dim3((unsigned)nSubGrid + 1u), dim3(64), A, FF, nSubHint);
    else MSL_SF_LAUNCH(prof, SK_FUSE, st, k_fuse<false>, dim3((unsigned)nSubGrid), dim3(64), A, FF, nSubHint);
}
void map_launch_compact(KernelProfiler &prof, hipStream_t st, const SfDev &P, int slot, bool resident) {
    const CompactArgs A = compact_args(P, slot);
    const int per = (((P.nseeds + 255) / 256) + 3) & ~3;   // seeds per thread (k_compact)
    const bool aligned4 = (P.nseeds & 3) == 0 && ((reinterpret_cast<size_t>(A.candOk) | reinterpret_cast<size_t>(A.fused)) & 3) == 0;
    const dim3 grid(resident ? 128 : 1);
    if (per <= 32 && aligned4) MSL_SF_LAUNCH(prof, SK_COMPACT, st, k_compact<8>, grid, dim3(256), A, resident ? 0 : 1);
    else if (per <= 96 && aligned4) MSL_SF_LAUNCH(prof, SK_COMPACT, st, k_compact<24>, grid, dim3(256), A, resident ? 0 : 1);
    else MSL_SF_LAUNCH(prof, SK_COMPACT, st, k_compact<0>, grid, dim3(256), A, resident ? 0 : 1);   // scan + new surfels + refill + tail compaction
}
// Closes a deferred window of F keyframes (P.prevSlotAbs = the slot of its last keyframe, P.blockUpd = the window's first per-sub-block slice).
void map_launch_replay(KernelProfiler &prof, hipStream_t st, const SfDev &P, int F, unsigned blkStride) {
    const unsigned nFront = (unsigned)((P.nseeds + SUB_ITEMS - 1) / SUB_ITEMS) + 1u;
    hipLaunchKernelGGL(k_defer_tail, dim3((unsigned)F + nFront), dim3(64), 0, st, P, fuse_args(P, 0, true), F, blkStride);
    MSL_SF_LAUNCH(prof, SK_COMPACT, st, k_replay, dim3(1), dim3(64), P, F);
    hipLaunchKernelGGL(k_gather, dim3(128), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_scatter, dim3(128), dim3(256), 0, st, P);
}
void map_launch_empty_pair(KernelProfiler &prof, hipStream_t st) {   // what an event pair reports for an EMPTY dispatch at this place of the chain
    hipEvent_t ea, eb;   // (the pair's first event completes with the previous command, so every event time contains the dependent-launch gap)
    if (prof.kernel_pair(SK_NEW, &ea, &eb)) hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, ea, eb, 0, 0);
}
void map_launch_set_ctr(hipStream_t st, const SfDev &P, long long n, int wide) { hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(64), 0, st, P.ctr, n, P.delUCount, wide); }
void map_launch_add_ctr(hipStream_t st, const SfDev &P, long long add) { hipLaunchKernelGGL(k_add_ctr, dim3(1), dim3(64), 0, st, P.ctr, add); }
void map_launch_aos_to_soa(KernelProfiler &prof, hipStream_t st, const SfDev &P, const msl_surfel *src, long long n, bool atEnd) {
    if (atEnd) hipLaunchKernelGGL(k_aos_to_soa_at, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P.map, src, n, P.ctr);
    else MSL_SF_LAUNCH(prof, SK_CONVERT, st, k_aos_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), P.map, src, n);
}
void map_launch_soa_to_aos(KernelProfiler &prof, hipStream_t st, const SfDev &P, msl_surfel *dst, long long n) {
    MSL_SF_LAUNCH(prof, SK_CONVERT, st, k_soa_to_aos, dim3((unsigned)((n + 255) / 256)), dim3(256), P.map, dst, n);
}
void map_launch_select_count(hipStream_t st, const SfDev &P, int mode, int arg) {
    hipLaunchKernelGGL(k_select_count, dim3(512), dim3(256), 0, st, P, mode, arg);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, st, P);
}
void map_launch_select_write(hipStream_t st, const SfDev &P, int mode, int arg, msl_surfel *out, int markDeleted) {
    hipLaunchKernelGGL(k_select_write, dim3(512), dim3(256), 0, st, P, mode, arg, out, markDeleted);
}
void map_launch_collect_changed(hipStream_t st, const SfDev &P, int ref, long long n, unsigned *count, unsigned *idxOut, msl_surfel *recOut, unsigned capOut) {
    const unsigned nblk = (unsigned)((n + SUB_ITEMS - 1) / SUB_ITEMS);
    hipLaunchKernelGGL(k_collect_changed, dim3(nblk), dim3(64), 0, st, P, ref, n, count, idxOut, recOut, capOut);
}
void map_launch_empty(hipStream_t st, int grid, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3((unsigned)grid), dim3(64), 0, st, a, b, 0, grid); }

}  // namespace sf
}  // namespace msl
