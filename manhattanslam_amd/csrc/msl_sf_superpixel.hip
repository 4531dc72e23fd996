// msl_sf_superpixel.hip -- frame-batched superpixel stage of the surfel fusion for gfx950 (MI355X).
//
// generateSuperPixels() of a keyframe (reference src/SurfelFusion.cpp:333-773) depends only on that keyframe's images, never on the
// map, so it is FRAME-BATCHED (one launch sequence per batch of F keyframes, XCD-aware 1-D grids) on the handle's "pre" stream:
//     kb_seed_init                        one thread per 8x8 superpixel seed                  (:528-584)
//     3 x { kb_assign                     one wave per two dual cells: argmin over <= 4 seeds (:333-415)
//           [kb_prop_lds,                 raster-order `stable` semantics as a min-fixpoint   (App. B.7.1)
//            kb_commit_px]                  over a compact worklist of the only pixels that can extend a chain (one launch, LDS)
//           kb_update_seeds               16 lanes per seed: ordered window gather, Huber mean (:428-515)
//           kb_commit_seeds }             chunk-abort (`return`) semantics: restore-only      (App. B.7.2)
//     kb_seed_plane                       16 lanes per seed: back-projection, pixel normals, Huber plane
//                                         fit with FP64 4x4 normal equations                  (:91-165, :597-773)
// What the map stage (msl_sf_map.hip) reads of a keyframe is written here: tex (one 8-byte texel per pixel), fuseRec (three 16-byte
// words per seed, one plane per word), cand / candOk (the surfel a seed would spawn).
//
// Every float expression keeps the reference's evaluation order and float/double promotions; compiled with -ffp-contract=off.

#include "msl_sf.h"

using namespace msl;
using namespace msl::sf;

namespace {

// Correctly rounded x / 100.0 (x >= 0 finite) without the ~35-instruction f64 divide: two Markstein steps with
// y = RN(1/100).  q1 is a faithful quotient (error < 1 ulp), so the final fused correction rounds to RN(x/100)
// (Markstein's theorem; 100 = 1.5625 * 2^6 is not an all-ones significand).  Checked against true division on
// the GPU by tests/test_surfel_gpu.py::test_div100_exact.
__device__ __forceinline__ double div100_exact(double x) {
    const double y = 0.01;                       // RN(1/100)
    const double q0 = x * y;
    const double q1 = fma(fma(-q0, 100.0, x), y, q0);
    return fma(fma(-q1, 100.0, x), y, q1);
}

// Comparisons of a FLOAT x with one of the reference's DOUBLE constants c (HUBER_RANGE 0.4, MAX_ANGLE_COS 0.1, the 0.05 / 0.1 depth limits), which
// C++ evaluates as (double)x OP c: none of these constants is a float, and for each of them the float nearest to it, cf = (float)c, is the float
// next ABOVE it (asserted below), so there is no float in [c, cf) and the sets of floats on either side of c and of cf are the same:
//   (double)x <  c  <=>  x <  cf        (double)x >  -c  <=>  x >  -cf
//   (double)x >= c  <=>  x >= cf        (double)x <= -c  <=>  x <= -cf        (double)x > c  <=>  x >= cf
// (NaN: false on both sides.)  One v_cmp_f32 instead of v_cvt_f64_f32 + v_cmp_f64 per test.
constexpr float float_below(float f) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) - 1u); }   // (positive finite f)
constexpr float HUBER_RANGE_F = (float)HUBER_RANGE, MAX_ANGLE_COS_F = (float)MAX_ANGLE_COS, DEPTH_005_F = 0.05f, DEPTH_01_F = 0.1f;
static_assert((double)HUBER_RANGE_F > HUBER_RANGE && (double)float_below(HUBER_RANGE_F) < HUBER_RANGE, "0.4f is the float next above 0.4");
static_assert((double)MAX_ANGLE_COS_F > MAX_ANGLE_COS && (double)float_below(MAX_ANGLE_COS_F) < MAX_ANGLE_COS, "0.1f is the float next above 0.1");
static_assert((double)DEPTH_005_F > 0.05 && (double)float_below(DEPTH_005_F) < 0.05, "0.05f is the float next above 0.05");
static_assert((double)DEPTH_01_F > 0.1 && (double)float_below(DEPTH_01_F) < 0.1, "0.1f is the float next above 0.1");
__device__ __forceinline__ bool in_huber_band(float r) { return r < HUBER_RANGE_F && r > -HUBER_RANGE_F; }   // residual < HUBER_RANGE && residual > -HUBER_RANGE

// Strictly sequential (left-to-right) float sums over 16-byte aligned LDS arrays; wide LDS reads are issued
// ahead of the dependent add chain so the chain runs at VALU latency instead of LDS latency.  (kb_seed_plane; kb_update_seeds uses the chains below.)
__device__ __forceinline__ float seq_sum_f32(const float *a, int n, float s) {
    int p = 0;
    for (; p + 8 <= n; p += 8) {
        const float4 u = *reinterpret_cast<const float4 *>(a + p), v = *reinterpret_cast<const float4 *>(a + p + 4);
        s += u.x; s += u.y; s += u.z; s += u.w; s += v.x; s += v.y; s += v.z; s += v.w;
    }
    for (; p < n; p++) s += a[p];
    return s;
}
// Huber/Newton numerator (:494-503) in list order: finite terms are 2*residual (a float add; identical to the double
// add rounded to float), +-inf marks a tail element whose contribution is the DOUBLE constant +-HUBER_RANGE.
__device__ __forceinline__ float huber_term_add(float s, float t) {
    return __builtin_isinf(t) ? (float)((double)s + (t > 0 ? HUBER_RANGE : -1 * HUBER_RANGE)) : s + t;
}

// The same strictly sequential sums without the LDS round trips: a ROTATING chain over the 16 lanes of a DPP row.  Lane i of the row holds the
// elements i, 16 + i, 32 + i, ... of the list (one per block of 16); step k of the chain lets every lane compute (value of its left neighbour) +
// (its element of block k / 16), row_ror:1 making lane 0 the neighbour of lane 15.  Lane k mod 16 then holds exactly s_k = s_(k-1) + e_k -- its
// neighbour held s_(k-1) after the step before -- while the lanes behind the front hold garbage nobody reads.  Elements beyond the end of a list
// are +0.0f: s + (+0.0f) == s for every s the chain can hold (it starts at +0.0f, and a float sum is -0.0f only if both operands are), so after
// any number of whole blocks lane 15 holds the sum of the list in list order, bit for bit what a left-to-right walk with `s += e` (seq_sum_f32) or huber_term_add returns.  One
// v_add_f32 with a DPP operand per element, for the four seeds of a wave at once.
__device__ __forceinline__ float row_ror1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));   // row_ror:1
}
__device__ __forceinline__ float chain_block_f32(float s, float t) {
#pragma unroll
    for (int q = 0; q < 16; q++) s = row_ror1(s) + t;
    return s;
}
__device__ __forceinline__ float chain_block_huber(float s, float t) {
#pragma unroll
    for (int q = 0; q < 16; q++) s = huber_term_add(row_ror1(s), t);
    return s;
}

// =============================================================================================
// Frame-batched superpixel stage
// =============================================================================================
__global__ __launch_bounds__(256) void kb_seed_init(SfDev P) {
    const int slot = blockIdx.y;
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    if (seedI == 0) P.wlCount[slot] = 0;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    int imageX = spX * SP + SP / 2, imageY = spY * SP + SP / 2;
    imageX = imageX < (P.W - 1) ? imageX : (P.W - 1);
    imageY = imageY < (P.H - 1) ? imageY : (P.H - 1);
    msl_seed s;
    memset(&s, 0, sizeof(s));
    P.fused[(size_t)slot * P.flagStride + seedI] = 0;
    if (member_at(P, F, imageY, imageX) != -1) {
        P.seeds[(size_t)slot * P.nseeds + seedI] = s; P.arec[(size_t)slot * P.nseeds + seedI] = assign_rec(s);
        return;
    }
    s.use = 1;
    s.x = (float)imageX; s.y = (float)imageY;
    vec3b(P, F, (float)imageY, (float)imageX, s.r, s.g, s.b);
    s.meanIntensity = gray_at(P, F, imageY, imageX);
    s.meanDepth = depth_at(P, F, imageY, imageX);
    if (s.meanDepth < 0.01) {
        int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
        int xe = xb + SP * 2, ye = yb + SP * 2;
        xb = xb > 0 ? xb : 0; yb = yb > 0 ? yb : 0;
        xe = xe < P.W - 1 ? xe : P.W - 1; ye = ye < P.H - 1 ? ye : P.H - 1;
        bool found = false;
        for (int j = yb; j < ye && !found; j++)
            for (int i = xb; i < xe; i++) {
                const float d = depth_at(P, F, j, i);
                if (d > 0.01) { s.meanDepth = d; found = true; break; }
            }
    }
    P.seeds[(size_t)slot * P.nseeds + seedI] = s;
    P.arec[(size_t)slot * P.nseeds + seedI] = assign_rec(s);
}

// kb_assign: a(p) = argmin seed of pixel p (:357-415 without the `stable` gate).  it == 0: every seed is
// unstable, so every free pixel is processed: write the index map directly.  it > 0: store a(p) and run
// relaxation round 0 (pixels whose current seed is unstable at pass start are processed for sure).
//
// One wave per "dual cell" [8 bx + 4, 8 bx + 12) x [8 by + 4, 8 by + 12), bx / by from -1.  Of the 3x3 neighbourhood only the seeds with
// |8c + 4 - x| < 8 on both axes are candidates (:384-389): per axis the pixel's own cell plus the left / upper neighbour when (x mod 8) < 4 or
// the right / lower one when (x mod 8) > 4 -- so ALL pixels of a dual cell have the same candidates {bx, bx + 1} x {by, by + 1} (its first
// column / row, x mod 8 == 4, only the first of each pair).  The candidates are therefore wave-uniform: their fields are scalar operands, and
// the per-pixel work is the four cost evaluations and nothing else.  Enumeration in the reference's order (checkI outer, checkJ inner, ascending).
constexpr int ASSIGN_NY = 2;   // dual cells (one below the other) per wave.  Everything the wave reads -- the NY + 1 lattice rows of candidate records
                               // (scalar loads) and the pixels' member / gray / depth / index words -- is requested before the first use: with one
                               // pixel per lane and loads that wait for one another the kernel had too few bytes in flight to keep HBM busy while
                               // other waves computed (35 us of memory time and 43 us of cost arithmetic per pass simply added up).
__global__ __launch_bounds__(256) void kb_assign(SfDev P, int it, int nSlots, int nbx, int nby) {
    const int bpr = (nbx + 3) >> 2;   // workgroups per row of dual cells (four waves = four dual cells along x)
    const int nbyG = (nby + ASSIGN_NY - 1) / ASSIGN_NY;
    int slot, blk;
    if (!xcd_slot(bpr * nbyG, nSlots, slot, blk)) return;
    if (blk == 0) {
        if (it > 0 && threadIdx.x < 8) P.changed[slot * 8 + threadIdx.x] = threadIdx.x == 0 ? 1 : 0;
        if (threadIdx.x >= 64 && threadIdx.x < 64 + NCHUNK) P.chunkAbort[(slot * 2 + (it & 1)) * 16 + threadIdx.x - 64] = 0x7FFFFFFF;
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int byg = blk / bpr, bxi = (blk - byg * bpr) * 4 + wv;
    if (bxi >= nbx) return;
    const int bx = bxi - 1, by0 = byg * ASSIGN_NY - 1;
    // The candidates (wave-uniform): cell k uses lattice rows by0 + k and by0 + k + 1, in each the neighbours bx and bx + 1 -- two records that
    // are adjacent in memory.  Rows / columns outside the lattice are clamped for the address (the array has a record of padding either side)
    // and never evaluated (the range test of :384-389).
    const bool okx0 = bx >= 0 && bx < P.spW, okx1 = bx + 1 < P.spW;
    const int bxc = min(bx, P.spW - 1);
    const AssignRec *arec = P.arec + (unsigned)slot * (unsigned)P.nseeds;
    AssignRec cr[ASSIGN_NY + 1][2];
    int rowIdx[ASSIGN_NY + 1];
#pragma unroll
    for (int r = 0; r <= ASSIGN_NY; r++) {
        const int rc = min(max(by0 + r, 0), P.spH - 1);
        rowIdx[r] = rc * P.spW + bx;                       // seed index of (bx, by0 + r) when valid
        const AssignRec *rp = arec + (rc * P.spW + bxc);
        cr[r][0] = rp[0]; cr[r][1] = rp[1];
    }
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    float *pxInv = P.pxInv + (size_t)slot * P.pxStride;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const int lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
    const int colI = 8 * bx + 4 + lx;
    const float colF = (float)colI;
    const bool colIn = colI >= 0 && colI < P.W;
    // ---- all loads of the wave's pixels ----
    bool inImg[ASSIGN_NY];
    int mem[ASSIGN_NY], cur[ASSIGN_NY];
    float gI[ASSIGN_NY], dIn[ASSIGN_NY];
    unsigned tCur[ASSIGN_NY];
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++) {
        const int rowI = 8 * (by0 + k) + 4 + ly;
        inImg[k] = colIn && rowI >= 0 && rowI < P.H && by0 + k + 1 < nby;
        const int rowC = min(max(rowI, 0), P.H - 1), colC = min(max(colI, 0), P.W - 1), pc = rowC * P.W + colC;   // (a clamped address: loaded, never used)
        mem[k] = member_at(P, F, rowC, colC);
        gI[k] = gray_at(P, F, rowC, colC);
        dIn[k] = it == 0 ? depth_at(P, F, rowC, colC) : *byte_off(pxInv, 4u * (unsigned)pc);
        cur[k] = it == 0 ? 0 : (int)*byte_off(index, 2u * (unsigned)pc);
    }
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++)
        tCur[k] = it == 0 ? 0u : *byte_off(tmin, 4u * (unsigned)cur[k]);   // (a plain load: 0 stays 0 and non-zero stays non-zero during the pass, so a stale line answers the same)
    // ---- per cell: the four cost evaluations ----
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++) {
        const int by = by0 + k;
        if (__ballot(inImg[k]) == 0) continue;
        const bool oky0 = by >= 0 && by < P.spH, oky1 = by + 1 < P.spH;
        const int rowI = 8 * by + 4 + ly;
        const int p = rowI * P.W + colI;
        const bool isPlane = mem[k] != -1;
        const float myIntensity = gI[k];
        // (float)(1.0 / (double)depth) is the same in all three passes: computed (one f64 divide) in pass 0, read back afterwards
        float myInvDepth = dIn[k];
        if (it == 0) {
            myInvDepth = 0.0f;
            if (dIn[k] > 0.01) myInvDepth = (float)(1.0 / (double)dIn[k]);
            if (inImg[k] && !isPlane) *byte_off_w(pxInv, 4u * (unsigned)p) = myInvDepth;
        }
        const bool pxHasDepth = myInvDepth > 0;
        const double myInvD = (double)myInvDepth;
        const float rowF = (float)rowI;
        float minDistDepth = 1e6f, minDistNodepth = 1e6f;
        int minSpIndexDepth = -1, minSpIndexNodepth = -1;
        bool allHasDepth = true;
        // calculateCost (:333-355) + the two running minima (:398-410) for one candidate; `use` = this pixel has the candidate (x mod 8 == 4:
        // the pixel's own cell only).  Selects instead of branches.
        auto consider = [&](const AssignRec &C, int spIndex, bool use) {
            float nodepthCost = 0;
            const float dist = (C.x - colF) * (C.x - colF) + (C.y - rowF) * (C.y - rowF);
            nodepthCost += dist / ((SP / 2) * (SP / 2));
            const float intensityDiff = C.meanIntensity - myIntensity;
            nodepthCost = (float)((double)nodepthCost + div100_exact((double)(intensityDiff * intensityDiff)));
            const bool has = C.invDepth >= 0 && pxHasDepth;
            const float inverseDepthDiff = (float)(C.invDepth - myInvD);
            const float withDepth = (float)((double)nodepthCost + (double)(inverseDepthDiff * inverseDepthDiff) * 400.0);
            const float depthCost = has ? withDepth : nodepthCost;
            allHasDepth = allHasDepth && (has || !use);
            const bool bd = use && depthCost < minDistDepth, bn = use && nodepthCost < minDistNodepth;
            minDistDepth = bd ? depthCost : minDistDepth; minSpIndexDepth = bd ? spIndex : minSpIndexDepth;
            minDistNodepth = bn ? nodepthCost : minDistNodepth; minSpIndexNodepth = bn ? spIndex : minSpIndexNodepth;
        };
        const bool anyStable = (cr[k][0].stable | cr[k][1].stable | cr[k + 1][0].stable | cr[k + 1][1].stable) != 0;   // (wave-uniform; rare)
        // the reference's order: checkI (x) outer, checkJ (y) inner, ascending
        if (okx0 && oky0) consider(cr[k][0], rowIdx[k], true);
        if (okx0 && oky1) consider(cr[k + 1][0], rowIdx[k + 1], ly != 0);
        if (okx1 && oky0) consider(cr[k][1], rowIdx[k] + 1, lx != 0);
        if (okx1 && oky1) consider(cr[k + 1][1], rowIdx[k + 1] + 1, lx != 0 && ly != 0);
        const int pick = allHasDepth ? minSpIndexDepth : minSpIndexNodepth;
        if (!inImg[k]) continue;
        if (it == 0) { *byte_off_w(index, 2u * (unsigned)p) = isPlane ? (unsigned short)0 : (unsigned short)(pick >= 0 ? pick : 0); continue; }
        *byte_off_w(amap, 2u * (unsigned)p) = isPlane ? IDX_PLANE : (pick >= 0 ? (unsigned short)pick : IDX_NONE);
        if (!isPlane && pick >= 0) {
            // the current seed is unstable at pass start <=> t(cur) == 0 (kb_update_seeds / kb_commit_seeds left 0 or T_INF, and this pass
            // only ever lowers a t to p + 1 >= 1, so a value read at any time during the pass answers the same)
            if (tCur[k] == 0) {
                // processed for sure (round 0): t(pick) = min(t(pick), p + 1) -- only a candidate that entered the pass stable has a t above 0
                if (anyStable && tmin[pick] > (unsigned)p + 1u) atomicMin(&tmin[pick], (unsigned)p + 1u);
            } else if (pick != cur[k]) {
                // Only these pixels can extend a chain: p is processed iff its (stable) seed gets unstabilised before p, and it
                // then unstabilises a DIFFERENT seed.  (pick == cur would only re-lower t(cur) above its current value.)
                P.wl[(size_t)slot * P.pxStride + atomicAdd(&P.wlCount[slot], 1u)] = (unsigned)p;
            }
        }
    }
}

// t(s) = raster position from which seed s counts as unstable: 0 if unstable at pass start, else
// 1 + the first processed pixel that picked it (min-fixpoint).
__device__ __forceinline__ bool relax_pixel(unsigned *tmin, const unsigned short *index, const unsigned short *amap, int p) {
    const unsigned short a = amap[p];
    if (a >= IDX_PLANE) return false;
    const unsigned tc = __hip_atomic_load(&tmin[index[p]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tc == 0 || tc > (unsigned)p) return false;     // tc == 0: handled in round 0; tc > p: not processed (yet)
    if (__hip_atomic_load(&tmin[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)p + 1u) return false;
    return atomicMin(&tmin[a], (unsigned)p + 1u) > (unsigned)p + 1u;
}

constexpr int PROP_BLOCKS = 16;   // workgroups per keyframe over the (small) worklist
__global__ __launch_bounds__(256) void kb_prop(SfDev P, int round, int nSlots) {
    int slot, blk;
    if (!xcd_slot(PROP_BLOCKS, nSlots, slot, blk)) return;
    if (!P.changed[slot * 8 + round]) return;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    const unsigned nwl = P.wlCount[slot];
    bool any = false;
    for (unsigned e = blk * 256 + threadIdx.x; e < nwl; e += PROP_BLOCKS * 256) any |= relax_pixel(tmin, index, amap, (int)wl[e]);
    if (any) P.changed[slot * 8 + round + 1] = 1;
}

// Finisher: one workgroup per keyframe iterates the relaxation to its fixpoint (normally zero rounds).
__global__ __launch_bounds__(1024) void kb_prop_finish(SfDev P) {
    __shared__ int s_ch;
    const int slot = blockIdx.x;
    if (threadIdx.x == 0) s_ch = P.changed[slot * 8 + PROP_ROUNDS];
    __syncthreads();
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    const unsigned nwl = P.wlCount[slot];
    while (s_ch) {
        __syncthreads();
        if (threadIdx.x == 0) s_ch = 0;
        __syncthreads();
        bool any = false;
        for (unsigned e = threadIdx.x; e < nwl; e += 1024) any |= relax_pixel(tmin, index, amap, (int)wl[e]);
        if (any) s_ch = 1;
        __syncthreads();
    }
}

// The whole relaxation in ONE launch: one workgroup per keyframe keeps t(s) in LDS (4 B per seed) and its share of the
// worklist in registers, so a round costs a few LDS operations instead of a kernel boundary plus agent-scope round trips.
// The min-fixpoint is unique, so the evaluation order does not matter.  (kb_prop / kb_prop_finish remain as the fallback
// for seed counts whose t(s) does not fit the LDS.)
constexpr int PROP_LDS_MAX_SEEDS = 36 * 1024;   // 144 KB
__global__ __launch_bounds__(256) void kb_prop_lds(SfDev P) {
    extern __shared__ unsigned s_t[];
    const int slot = blockIdx.x;
    const unsigned nwl = P.wlCount[slot];
    if (nwl == 0) return;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    constexpr int NT = 256, R = 16;   // a 256-thread workgroup finds room on a busy GPU; a 16-wave one waits for a whole CU
    unsigned ep[R];
    unsigned short ec[R], ea[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned e = threadIdx.x + r * NT;
        ep[r] = 0xFFFFFFFFu; ec[r] = 0; ea[r] = 0;
        if (e < nwl) { const unsigned p = wl[e]; ep[r] = p; ec[r] = index[p]; ea[r] = amap[p]; }
    }
    for (int i = threadIdx.x; i < P.nseeds; i += NT) s_t[i] = tmin[i];
    __syncthreads();
    auto relax = [&](unsigned p, unsigned short cur, unsigned short a) -> bool {
        if (a >= IDX_PLANE) return false;
        const unsigned tc = s_t[cur];
        if (tc == 0 || tc > p) return false;            // tc == 0: handled in round 0; tc > p: not processed (yet)
        if (s_t[a] <= p + 1u) return false;
        return atomicMin(&s_t[a], p + 1u) > p + 1u;
    };
    int any;
    do {
        bool ch = false;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (ep[r] != 0xFFFFFFFFu) ch |= relax(ep[r], ec[r], ea[r]);
        for (unsigned e = threadIdx.x + R * NT; e < nwl; e += NT) { const unsigned p = wl[e]; ch |= relax(p, index[p], amap[p]); }
        any = __syncthreads_or(ch ? 1 : 0);
    } while (any);
    for (int i = threadIdx.x; i < P.nseeds; i += NT) {
        const unsigned t = s_t[i];
        if (t != tmin[i]) tmin[i] = t;
    }
}

__global__ __launch_bounds__(256) void kb_commit_px(SfDev P, int nSlots) {
    // 8 consecutive pixels per thread (16-byte loads of both maps; the slot stride is a multiple of 64).  A pixel whose pick equals its
    // current seed cannot change, so t(s) is only looked up for the few pixels that picked a different seed.
    int slot, blk;
    if (!xcd_slot(((P.npx + 7) / 8 + 255) / 256, nSlots, slot, blk)) return;
    const int p0 = (blk * 256 + threadIdx.x) * 8;
    if (p0 >= P.npx) return;
    unsigned short *index = P.index + (size_t)slot * P.pxStride;
    const uint4 a4 = *reinterpret_cast<const uint4 *>(P.amap + (size_t)slot * P.pxStride + p0);
    uint4 i4 = *reinterpret_cast<const uint4 *>(index + p0);
    const unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    unsigned aw[4] = {a4.x, a4.y, a4.z, a4.w}, iw[4] = {i4.x, i4.y, i4.z, i4.w};
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const unsigned a = (aw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, cur = (iw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if (a >= IDX_PLANE || a == cur || p0 + k >= P.npx) continue;   // (the last group may reach into the slot's padding)
        if (tmin[cur] <= (unsigned)(p0 + k)) {
            iw[k >> 1] = (iw[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | (a << (16 * (k & 1)));
            changed = true;
        }
    }
    if (changed) { i4.x = iw[0]; i4.y = iw[1]; i4.z = iw[2]; i4.w = iw[3]; *reinterpret_cast<uint4 *>(index + p0) = i4; }
}

// Four consecutive elements loaded as one access of whatever alignment the element type guarantees (global memory
// tolerates dword-/byte-aligned wide loads).
template <typename T> struct Quad { T v[4]; };
template <typename T> __device__ __forceinline__ Quad<T> load_quad(const T *p) { Quad<T> q; __builtin_memcpy(&q, p, sizeof(q)); return q; }
template <typename T> __device__ __forceinline__ Quad<T> load_quad(gptr<T> p) {
    Quad<T> q;
#pragma unroll
    for (int e = 0; e < 4; e++) q.v[e] = p[e];
    return q;
}
// Inclusive prefix sum over the 16 lanes of a DPP row (= one seed group); lanes without a source read 0.
__device__ __forceinline__ int row_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    return v;
}

// 16-lane (DPP row = seed group) exchanges in the VALU instead of __shfl / __shfl_xor, which compile to ds_bpermute: a trip through the LDS crossbar per
// value (plus the address arithmetic in front of it and a dependent wait behind it) where a DPP move or operand costs one VALU slot and a few cycles.
// Only controls that give EVERY lane a source lane (row rotations, quad permutes, row_newbcast), so `old` is never read (bound_ctrl).
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
template <int N> __device__ __forceinline__ int row_lane_i32(int v) { return dpp_i32<0x150 + N>(v); }      // lane N of the row, to all its lanes (row_newbcast)
template <int N> __device__ __forceinline__ float row_lane_f32(float v) { return dpp_f32<0x150 + N>(v); }
// sum / maximum over the 16 lanes of a row, every lane receives it: row_ror:8, row_ror:4, quad_perm [2,3,0,1], quad_perm [1,0,3,2].  For integers, and for
// floats whose partial sums are all exact (integer-valued sums below 2^24), any order gives the same result as the xor butterfly this replaces.
__device__ __forceinline__ int row_sum_i32(int v) { v += dpp_i32<0x128>(v); v += dpp_i32<0x124>(v); v += dpp_i32<0x4E>(v); v += dpp_i32<0xB1>(v); return v; }
__device__ __forceinline__ float row_sum_exact_f32(float v) { v += dpp_f32<0x128>(v); v += dpp_f32<0x124>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0xB1>(v); return v; }
__device__ __forceinline__ float row_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0x128>(v)); v = fmaxf(v, dpp_f32<0x124>(v)); v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0xB1>(v));
    return v;
}
// maximum over the four rows of a wave of a value that is uniform inside each row: a scalar
__device__ __forceinline__ int rows_max_i32(int v) {
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// kb_update_seeds (:428-515): 16 lanes per seed (lane = window row), 16 seeds per workgroup.
// Integer-valued sums are exact in any order; the float depth sum and the Huber/Newton sums run in window raster order as rotating DPP chains
// over the seed's 16 lanes (chain_block_f32 above), fed by terms each lane computes from its own elements of the ordered depth list.
// Round 5: 56 VGPRs and 17 KB of LDS (rounds 2-4: 80 and 35 KB -- the term list, the per-seed LDS scalars and their atomics are gone): 8 instead
// of 4 workgroups per CU for a kernel whose waves mostly wait (13.5 -> 6.9 us per frame beside the other stages, front end +3.7 %).
// What kb_update_seeds leaves for kb_commit_seeds in the seed's record of seedsTmp[] (round 6): the per-seed scalar rest of updateSeedsKernel -- three
// divisions, the colour fetch, the stability test, the new seed record and its AssignRec with an FP64 division: ~130 instructions that ONE lane of a
// seed's sixteen executed -- runs there, one thread per seed.
struct SeedUpd {
    int state;            // 0: skipped (unused or stable: kb_update_seeds wrote what changes); 1: no pixel owned, the seed ends its chunk (:473-474); 2: update
    int cnt, sumI, sumX, sumY;   // the integer-valued sums (:461-464)
    int depthLoop;        // the seed has valid depths: meanDepth below is the refined mean (:486-512), otherwise 0 goes to the record (:489-490)
    float meanDepth;
};
static_assert(sizeof(SeedUpd) <= sizeof(msl_seed), "the hand-over record fits a seedsTmp record");
template <bool STRADDLE>   // STRADDLE: W mod 8 in {1, 2, 3} (a window quad can stick out over the right edge)
__global__ __launch_bounds__(256) void kb_update_seeds(SfDev P, int it, int nSlots) {
    // rows of 256 + 16 words: the two seeds of a 32-lane half read / write entry l + 16 t of their own row together (ds_*_b32: bank = word address mod
    // 32); with a row stride of 256 words both rows started on the same bank (round 4: 32 % of the kernel's LDS cycles were bank conflicts)
    __shared__ __attribute__((aligned(16))) float s_depth[16][272];   // the ordered depth lists (the only LDS of the kernel since round 5: 17 KB)
    int slot, blk;
    if (!xcd_slot((P.nseeds + 15) / 16, nSlots, slot, blk)) return;
#ifdef MSL_FUSE_STAMPS   // section cycle counts of the waves of slot 0, summed into delList[96 ..] (tools/fuse_stamps.py)
    unsigned long long ust[8]; int usn = 0;
#define USTAMP() ust[usn++] = __builtin_amdgcn_s_memtime()
#else
#define USTAMP()
#endif
    USTAMP();
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int seedI = blk * 16 + g;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    const unsigned short *index = P.index + (size_t)slot * P.pxStride;
    msl_seed S;
    memset(&S, 0, sizeof(S));
    bool active = seedI < P.nseeds;
    bool stable = false;
    if (active) {
        S = P.seeds[(size_t)slot * P.nseeds + seedI];
        stable = it > 0 ? (P.tmin[(size_t)slot * P.nseeds + seedI] == T_INF) : (S.stable != 0);
        // The seed records themselves are written by kb_commit_seeds (from the SeedUpd this kernel leaves in seedsTmp[]), which also knows by then
        // whether the seed's chunk had ended earlier.
        if (!S.use || stable) {
            if (l == 0) {   // skipped: only the stable flag (as left by the pixel pass) and t(s) change
                P.seeds[(size_t)slot * P.nseeds + seedI].stable = stable;
                P.arec[(size_t)slot * P.nseeds + seedI].stable = stable ? 1u : 0u;
                reinterpret_cast<SeedUpd *>(P.seedsTmp + ((size_t)slot * P.nseeds + seedI))->state = 0;
                P.tmin[(size_t)slot * P.nseeds + seedI] = stable ? T_INF : 0u;
            }
            active = false;
        }
    }
    if (!__ballot(active)) return;   // all four seeds of the wave are skipped (stable or unused): nothing to gather
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    const int xb0 = spX * SP + SP / 2 - SP, yb0 = spY * SP + SP / 2 - SP;
    const int xb = xb0 > 0 ? xb0 : 0, yb = yb0 > 0 ? yb0 : 0;
    const int xe = (xb0 + SP * 2) < P.W - 1 ? (xb0 + SP * 2) : P.W - 1, ye = (yb0 + SP * 2) < P.H - 1 ? (yb0 + SP * 2) : P.H - 1;
    int sumX = 0, sumY = 0, sumI = 0, cnt = 0, nd = 0;
    {
        // Lane = (row r of a group of four window rows, quad q of four window columns): 12 wide loads per lane (8 B of
        // index, 16 B of depth, 4 B of gray, four times) instead of 48 scalar ones.  Window columns start at a multiple
        // of 4: a quad lies left of the image as a whole (first lattice column) or starts inside it.  When W is not a multiple of 4 the last
        // quad of a window may stick out over the right edge: it is then loaded from W - 4 (inside the row) and its first elements, which
        // belong to the neighbouring lane's quad, are masked (col >= col0) -- no element-wise path, the window order is unchanged.  Raster
        // order of the window = (iteration, lane, element), which the ordered depth list below follows.
        const int rq = l >> 2, cq = l & 3;
        const int col0 = xb0 + 4 * cq;
        const bool quadIn = col0 >= 0 && (STRADDLE ? col0 < P.W : col0 + 3 < P.W);
        const int colc = quadIn ? (STRADDLE ? min(col0, P.W - 4) : col0) : 0;
        Quad<unsigned short> idq[4];
        Quad<float> dq[4];
        Quad<uint8_t> gq[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int jc = min(max(yb0 + 4 * m + rq, 0), P.H - 1);
            idq[m] = load_quad(byte_off(index, 2u * (unsigned)(jc * P.W + colc)));
            dq[m] = load_quad(byte_off(F.depthG(), (unsigned)jc * P.dsB + 4u * (unsigned)colc));
            gq[m] = load_quad(byte_off(F.grayG(), (unsigned)jc * P.gsB + (unsigned)colc));
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = yb0 + 4 * m + rq;
            const bool rowOk = active && quadIn && j >= yb && j < ye;
            bool hd[4];
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = colc + e;
                const bool own = rowOk && (!STRADDLE || col >= col0) && col >= xb && col < xe && idq[m].v[e] == seedI;
                hd[e] = own && dq[m].v[e] >= DEPTH_01_F;   // `> 0.1` (:452), float form (float_below)
                if (own) { sumX += col; sumY += j; sumI += gq[m].v[e]; cnt++; }
                c += hd[e] ? 1 : 0;
            }
            const int incl = row_incl_scan(c);
            int o = nd + incl - c;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (hd[e]) s_depth[g][o++] = dq[m].v[e];
            nd += row_lane_i32<15>(incl);
        }
    }
    USTAMP();   // 1: seed record + window gather + ordered depth list
    sumX = row_sum_i32(sumX); sumY = row_sum_i32(sumY); sumI = row_sum_i32(sumI); cnt = row_sum_i32(cnt);
    __builtin_amdgcn_wave_barrier();
    const bool depthLoop = active && cnt != 0 && nd > 0;   // (uniform over the seed's 16 lanes)
    if (l == 0 && active && cnt == 0)   // `return`: ends the chunk (:473-474); the seed itself stays as it is, unstable (kb_commit_seeds)
        atomicMin(&P.chunkAbort[(slot * 2 + (it & 1)) * 16 + seed_chunk(seedI, P.nseeds)], seedI);
    // ---- mean depth and its Huber refinement (:486-512): the sequential sums as rotating chains (above); everything per seed is uniform over its
    // 16 lanes and lives in registers -- no LDS, no atomics, no barriers in the Newton loop ----
    const int ndL = depthLoop ? nd : 0;                 // a seed without a depth loop contributes empty lists
    const int nblk = rows_max_i32((ndL + 15) >> 4);    // blocks of the longest list of the wave's four seeds
    float meanDepth = 0.0f;
    {
        float sd = 0.0f;
        for (int bq = 0; bq < nblk; bq++) {
            const int e = l + 16 * bq;
            sd = chain_block_f32(sd, e < ndL ? s_depth[g][e] : 0.0f);
        }
        const float sumDepth = row_lane_f32<15>(sd);
        if (depthLoop) meanDepth = sumDepth / (float)nd;
    }
    USTAMP();   // 2: means, colour fetch, sequential depth sum
    bool open = depthLoop;
    for (int newtonI = 0; newtonI < 5; newtonI++) {
        if (!__ballot(open)) break;
        // pass 1: in-range count (sumB) and whether any list of the wave has a Huber tail this step
        int inr = 0;
        bool tail = false;
        for (int bq = 0; bq < nblk; bq++) {
            const int e = l + 16 * bq;
            if (open && e < ndL) {
                const float residual = meanDepth - s_depth[g][e];
                if (in_huber_band(residual)) inr++; else tail = true;
            }
        }
        inr = row_sum_i32(inr);
        const bool anyTail = __ballot(tail) != 0ull;
        // pass 2: the chain over the terms -- in range: 2*residual; a tail element: the +-inf marker huber_term_add() turns into +-HUBER_RANGE
        float sa = 0.0f;
        for (int bq = 0; bq < nblk; bq++) {
            const int e = l + 16 * bq;
            float t = 0.0f;
            if (open && e < ndL) {
                const float residual = meanDepth - s_depth[g][e];
                if (in_huber_band(residual)) t = 2 * residual;
                else t = residual > 0 ? __builtin_inff() : -__builtin_inff();
            }
            // no Huber tails anywhere in the wave (the common case): a plain float chain, 1 VALU op per element instead of ~8
            sa = anyTail ? chain_block_huber(sa, t) : chain_block_f32(sa, t);
        }
        const float sumA = row_lane_f32<15>(sa);
        const float sumB = (float)(2 * inr);
        const float deltaDepth = (float)((double)(-sumA) / ((double)sumB + 10.0));
        if (open) {
            meanDepth = meanDepth + deltaDepth;
            if ((deltaDepth < 0.01 && deltaDepth > -0.01) || newtonI == 4) open = false;
        }
    }
    USTAMP();   // 3: Newton steps
    if (active && l == 0) {
        SeedUpd U;
        U.state = cnt == 0 ? 1 : 2; U.cnt = cnt; U.sumI = sumI; U.sumX = sumX; U.sumY = sumY; U.depthLoop = depthLoop ? 1 : 0; U.meanDepth = meanDepth;
        *reinterpret_cast<SeedUpd *>(P.seedsTmp + ((size_t)slot * P.nseeds + seedI)) = U;
    }
#ifdef MSL_FUSE_STAMPS
    USTAMP();   // 4: stores
    if (slot == 0 && (threadIdx.x & 63) == 0) {
        for (int q = 1; q < usn; q++) atomicAdd(&P.delList[96 + q], (unsigned)(ust[q] - ust[q - 1]));
        atomicAdd(&P.delList[96], 1u);
    }
#endif
}


// kb_commit_seeds: the per-seed end of updateSeedsKernel (:475-515), one thread per seed: means, colour fetch, stability test, the new seed record, t(s)
// and the AssignRec, from the sums kb_update_seeds left (SeedUpd) -- and the chunk-abort rule: a seed without a single owned pixel ends its chunk
// (`return`, :473-474), so the seeds BEHIND it in the chunk stay as they are, unstable.  That rule can never fire: a used seed (lattice position
// spX < W / 8, spY < H / 8) always owns the pixel at its lattice centre (8 spX + 4, 8 spY + 4).  That pixel is free (what `use` means, :541-545); its
// ONLY updatePixels candidate is this seed (|8 c + 4 - x| < 8 holds for c = spX alone when x mod 8 == 4, :384-389); pass 0 assigns it with cost
// 0 < 1e6 whatever intensity / depth are; no later pass can move it; and it lies inside the clipped window updateSeeds counts.  So the owned-pixel
// count is >= 1 and the abort path is kept for fidelity only (property-tested on adversarial inputs in the CPU suite).
__global__ __launch_bounds__(256) void kb_commit_seeds(SfDev P, int it) {
    const int slot = blockIdx.y;
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    if (seedI == 0) P.wlCount[slot] = 0;   // the next pixel pass rebuilds the relaxation worklist
    const size_t si = (size_t)slot * P.nseeds + seedI;
    const SeedUpd U = *reinterpret_cast<const SeedUpd *>(P.seedsTmp + si);
    if (U.state == 0) return;   // skipped (unused or stable): already as it should be
    const int abortAt = P.chunkAbort[(slot * 2 + (it & 1)) * 16 + seed_chunk(seedI, P.nseeds)];   // first seed of the chunk without a pixel (0x7FFFFFFF: none)
    if (U.state == 1 || seedI > abortAt) {   // the seed that ended the chunk, or one behind it: values untouched, unstable
        P.seeds[si].stable = 0; P.arec[si].stable = 0u; P.tmin[si] = 0u;
        return;
    }
    const FrameDev &F = P.frames[slot];
    // the 64-byte record as four 16-byte words (nobody else writes it): words 0-1 x, y; 10-11 meanDepth, meanIntensity; 12-14 r, g, b;
    // 15 the bytes fused | stable << 8 | use << 16 | _pad << 24
    static_assert(sizeof(msl_seed) == 64 && offsetof(msl_seed, meanDepth) == 40 && offsetof(msl_seed, r) == 48 && offsetof(msl_seed, stable) == 61, "msl_seed layout");
    uint4 *rec = reinterpret_cast<uint4 *>(P.seeds + si);
    const uint4 w0o = rec[0], w2o = rec[2], w3o = rec[3];
    const float sumIntensityNum = (float)U.cnt;
    const float sumIntensity = (float)U.sumI / sumIntensityNum, mX = (float)U.sumX / sumIntensityNum, mY = (float)U.sumY / sumIntensityNum;
    const float preIntensity = __uint_as_float(w2o.w), preX = __uint_as_float(w0o.x), preY = __uint_as_float(w0o.y);
    int tR = 0, tG = 0, tB = 0;
    vec3b(P, F, mY, mX, tR, tG, tB);
    const float updateDiff = fabsf(preIntensity - sumIntensity) + fabsf(preX - mX) + fabsf(preY - mY);
    const bool tStable = updateDiff < 0.2;
    const float tDepth = U.depthLoop ? U.meanDepth : 0.0f;   // no valid depth among the seed's pixels: 0 (:489-490)
    uint4 w0 = w0o, w2 = w2o, w3 = w3o;
    w0.x = __float_as_uint(mX); w0.y = __float_as_uint(mY);
    w2.z = __float_as_uint(tDepth); w2.w = __float_as_uint(sumIntensity);
    w3.x = (unsigned)tR; w3.y = (unsigned)tG; w3.z = (unsigned)tB;
    w3.w = (w3.w & 0x00FF00FFu) | (tStable ? 0x100u : 0u);
    rec[0] = w0; rec[2] = w2; rec[3] = w3;
    P.tmin[si] = tStable ? T_INF : 0u;
    AssignRec a;
    a.x = mX; a.y = mY; a.meanIntensity = sumIntensity; a.stable = tStable ? 1u : 0u;
    a.invDepth = tDepth > 0 ? 1.0 / (double)tDepth : -1.0; a._pad = 0;
    P.arec[si] = a;
}

// kb_seed_plane: calculateNorms (:775-803) fused per seed, 16 lanes per seed, 4 seeds per wave/workgroup.
// Pixel positions and cross-product normals are recomputed from depth instead of materialising spaceMap
// (7.4 MB f64) / normMap.  Also prepares the surfel the seed would spawn (initializeSurfels, :285-331).
__device__ __forceinline__ void pixel_normal(const SfDev &P, int row, int col, float myX, float myY, float myZ, float rightDepth,
                                             float downDepth, float cxr, float cx1, float ryr, float ry1,
                                             float &nX, float &nY, float &nZ) {
    nX = nY = nZ = 0.0f;
    if (row < 1 || row > P.H - 2 || col < 1 || col > P.W - 2) return;  // never written (:620-625)
    // back_project of the right / down neighbours with the tabulated quotients: (col+1, row) and (col, row+1)
    float rightX = cx1 * rightDepth, rightY = ryr * rightDepth, rightZ = rightDepth;
    float downX = cxr * downDepth, downY = ry1 * downDepth, downZ = downDepth;
    if (myZ < DEPTH_01_F || rightZ < DEPTH_01_F || downZ < DEPTH_01_F) return;   // `< 0.1` (:628): float form, see float_below
    rightX = rightX - myX; rightY = rightY - myY; rightZ = rightZ - myZ;
    downX = downX - myX; downY = downY - myY; downZ = downZ - myZ;
    float normX = rightY * downZ - rightZ * downY;
    float normY = rightZ * downX - rightX * downZ;
    float normZ = rightX * downY - rightY * downX;
    const float normLength = sqrtf(normX * normX + normY * normY + normZ * normZ);
    normX /= normLength; normY /= normLength; normZ /= normLength;
    const float viewAngle = (normX * myX + normY * myY + normZ * myZ) / sqrtf(myX * myX + myY * myY + myZ * myZ);
    if (viewAngle > -MAX_ANGLE_COS_F && viewAngle < MAX_ANGLE_COS_F) return;
    nX = normX; nY = normY; nZ = normZ;
}

// Sum over the 16 lanes of a DPP row (= one seed group); every lane receives the total.  Row rotations by 8 and 4
// and quad permutes run in the VALU (a few cycles) instead of ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
    const unsigned long long u = __double_as_longlong(v);
    // (bound_ctrl: the row rotations and quad permutes used here give every lane a source lane, so the `old` operand is never read -- without it the
    // compiler materialises a zero for it in front of every move: 2 of 5 instructions per value and step of group_sum_d)
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double group_sum_d(double v) {
    v += dpp_mov_d<0x128>(v);   // row_ror:8
    v += dpp_mov_d<0x124>(v);   // row_ror:4
    v += dpp_mov_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_d<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}

// FuseRec: the 48 bytes of a seed that fuseSurfelsKernel reads (three 16-byte loads instead of the 64-byte msl_seed), with the terms
// that depend on the seed alone evaluated once per seed instead of once per fused surfel -- same expressions, same operands:
//   [0] normX, normY, normZ (camera frame), meanDepth
//   [1] pose * (posX, posY, posZ, 1) (:240-245), getWeight(meanDepth) (:236)
//   [2] size * fabs(meanDepth / (cameraF * viewCos)) (:270-271), meanIntensity, r | g << 8 | b << 16, valid
// valid = !(norm == 0) && !(viewCos < MAX_ANGLE_COS), the two seed tests of :214-219.
//
// LDS: one pool per wave.  The four seeds of a wave form a 2x2 block of the seed lattice, so their 16x16 windows cover
// 24x24 = 576 distinct pixels; every pixel belongs to one seed, hence the four ordered lists hold <= 576 entries in total
// (+ 3 x 3 for 16-byte alignment of each list) instead of 4 x 256.  14 KB per wave: 11 waves per CU instead of 5.
struct PlaneFit { int active; float nx, ny, nz, nb, sumX, sumY, sumZ, maxDist; };   // kb_seed_plane -> kb_seed_finish, in the seed's slot of SfDev::cand
static_assert(sizeof(PlaneFit) <= sizeof(msl_surfel), "the hand-over record fits a candidate slot");
constexpr int PLANE_POOL = 24 * 24 + 12 + 2 * 28;   // + the bank-phase gaps in front of the second list of each half
template <bool STRADDLE>   // STRADDLE: W mod 8 in {1, 2, 3} -- a window quad can stick out over the right edge (instantiated separately: the common
                           // geometry carries none of that code)
__global__ __launch_bounds__(64) void kb_seed_plane(SfDev P, int nSlots) {
    __shared__ __attribute__((aligned(16))) float s_pool[6][PLANE_POOL];   // position x y z, normal x y z
    __shared__ __attribute__((aligned(16))) double s_h[4][16];
    int slot, blk;
    const int bW = (P.spW + 1) / 2, bH = (P.spH + 1) / 2;
    if (!xcd_slot(bW * bH, nSlots, slot, blk)) return;
#ifdef MSL_FUSE_STAMPS   // section cycle counts of the waves of slot 0, summed into delList[64 ..] (tools/fuse_stamps.py)
    unsigned long long sst[14]; int ssn = 0;
#define SECTION_STAMP() sst[ssn++] = __builtin_amdgcn_s_memtime()
#else
#define SECTION_STAMP()
#endif
    SECTION_STAMP();
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15, lane = threadIdx.x;
    const int spX = (blk % bW) * 2 + (g & 1), spY = (blk / bW) * 2 + (g >> 1);
    const bool inRange = spX < P.spW && spY < P.spH;
    const int seedI = inRange ? spY * P.spW + spX : 0;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
#ifdef MSL_FUSE_STAMPS
    { unsigned long long a = (unsigned long long)F.depth; asm volatile("" :: "s"(a)); }
    SECTION_STAMP();   // 0a: kernel arguments + frame record
#endif
    const unsigned short *index = P.index + (size_t)slot * P.pxStride;
    // (unconditionally: a group outside the lattice reads seed 0 -- seedI = 0 above -- and never uses or stores it; the zero-filled record the
    // conditional load needed cost 84 select instructions)
    const msl_seed S = P.seeds[(size_t)slot * P.nseeds + seedI];
#ifdef MSL_FUSE_STAMPS
    asm volatile("" :: "v"(S.x), "v"(S.meanDepth));
    SECTION_STAMP();   // 0b: seed record
#endif
    const int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
    // ---- gather: lane = (row r of a group of four window rows, quad q of four window columns), four iterations; the
    // unclipped window is guarded by the flat index range (:680-684).  16 wide loads per lane: 8 B of index, 16 B of depth,
    // 16 B of the row below, 4 B right of the quad (the other right neighbours are the quad's own elements). ----
    float maxDist = 0;
    int nvalid = 0, base = 0, poolUsed = 0;
    {
        const int rq = l >> 2, cq = l & 3;
        // wrapped pixels (App. B.6) without an integer division: a quad left / right of the image (window columns start at a multiple
        // of 4) belongs to the previous / next row of the flat index.  When W is not a multiple of 4 the last quad of a window in the last
        // lattice column can straddle the right edge: its elements beyond W - 1 are the first pixels of the next row (`straddle`, rare:
        // element-wise loads).
        const int cx0 = xb + 4 * cq;
        const int wrapRow = cx0 < 0 ? -1 : (cx0 >= P.W ? 1 : 0), wcol0 = cx0 - wrapRow * P.W;
        const bool straddle = STRADDLE && cx0 < P.W && cx0 + 3 >= P.W;
        auto elem_wrap = [&](int e) -> int { return (straddle && cx0 + e >= P.W) ? 1 : 0; };   // extra row wrap of element e of a straddling quad
        Quad<unsigned short> idq[4];
        Quad<float> dq[4], ddq[4];
        float dr3[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int wr = yb + 4 * m + rq + wrapRow;
            const int row = min(max(wr, 0), P.H - 1);     // rows outside the image fail the flat-index test below
            if (!straddle) {
                idq[m] = load_quad(byte_off(index, 2u * (unsigned)(row * P.W + wcol0)));
                dq[m] = load_quad(byte_off(F.depthG(), (unsigned)row * P.dsB + 4u * (unsigned)wcol0));
                ddq[m] = load_quad(byte_off(F.depthG(), (unsigned)min(row + 1, P.H - 1) * P.dsB + 4u * (unsigned)wcol0));
                dr3[m] = *byte_off(F.depthG(), (unsigned)row * P.dsB + 4u * (unsigned)min(wcol0 + 4, P.W - 1));
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int we = elem_wrap(e), col = cx0 + e - we * P.W, rowe = min(max(wr + we, 0), P.H - 1);
                    idq[m].v[e] = index[(size_t)rowe * P.W + col];
                    dq[m].v[e] = F.depthG()[(size_t)rowe * P.dstride + col];
                    ddq[m].v[e] = F.depthG()[(size_t)min(rowe + 1, P.H - 1) * P.dstride + col];
                }
                dr3[m] = 0.0f;
            }
        }
        // Texel map for k_fuse: every pixel's {depth, final index} as one 8-byte word.  The seed's own 8x8 cell is rows / columns
        // [4, 12) of its window (iterations 1, 2; column quads 1, 2), and the cells tile the image, so each pixel is written exactly
        // once from values this lane holds anyway: two 16-byte stores per iteration for half of the lanes.
        if (inRange && (cq == 1 || cq == 2)) {
            uint2 *tex = P.tex + (size_t)slot * P.pxStride;
#pragma unroll
            for (int m = 1; m <= 2; m++) {
                uint4 *t4 = reinterpret_cast<uint4 *>(tex + (size_t)(yb + 4 * m + rq) * P.W + cx0);
                t4[0] = make_uint4(__float_as_uint(dq[m].v[0]), idq[m].v[0], __float_as_uint(dq[m].v[1]), idq[m].v[1]);
                t4[1] = make_uint4(__float_as_uint(dq[m].v[2]), idq[m].v[2], __float_as_uint(dq[m].v[3]), idq[m].v[3]);
            }
        }
        unsigned vm = 0;   // bit 4 m + e: pixel e of the quad in iteration m is a valid-depth pixel of the seed
        // Branch-free (sixteen per-lane branches per wave otherwise; one or two waves per SIMD cannot hide their bubbles): the two squares of `dist` are the
        // same products whichever pixel of a column / row they are computed for, so each is evaluated once per column and once per row of the lane's quads.
        float xd2[4], yd2[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { const float xDiff = (xb + 4 * cq + e) - S.x; xd2[e] = xDiff * xDiff; }
#pragma unroll
        for (int m = 0; m < 4; m++) { const float yDiff = (yb + 4 * m + rq) - S.y; yd2[m] = yDiff * yDiff; }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = xb + 4 * cq + e, jrow = yb + 4 * m + rq;
                const int pixelIndex = jrow * P.W + i;
                const bool own = inRange & (pixelIndex >= 0) & (pixelIndex < P.npx) & (idq[m].v[e] == seedI);   // (`&`: no short-circuit branch around the compare of the loaded index)
                const float dist = xd2[e] + yd2[m];   // xDiff * xDiff + yDiff * yDiff (:680-683)
                maxDist = (own & (dist > maxDist)) ? dist : maxDist;
                vm |= ((own & (dq[m].v[e] >= DEPTH_005_F)) ? 1u : 0u) << (4 * m + e);   // `> 0.05` (:686)
            }
        nvalid = __popc(vm);
        SECTION_STAMP();   // 1a: window loads arrived, ownership tests
        nvalid = row_sum_i32(nvalid);
        {   // list bases inside the pool: multiples of 4 entries (16-byte reads of the sequential sums), and the second list of each 32-lane half
            // 16 banks away from the first one (mod 32) -- the loops below read entry base + l + 16 t with ds_read_b32, whose lane groups are the two
            // halves of the wave and whose bank is the word address mod 32: with arbitrary bases the two seeds of a half collided on every access
            // (round 4: 27 % of the kernel's LDS cycles were bank conflicts)
            const int pad = (nvalid + 3) & ~3;
            const int n0 = __builtin_amdgcn_readlane(pad, 0), n1 = __builtin_amdgcn_readlane(pad, 16), n2 = __builtin_amdgcn_readlane(pad, 32), n3 = __builtin_amdgcn_readlane(pad, 48);
            const int b1 = n0 + ((16 - n0) & 31), b2 = b1 + n1, b3 = b2 + n2 + ((16 - n2) & 31);   // b1 = 16 (mod 32) relative to b0 = 0; b3 likewise to b2
            base = g == 0 ? 0 : g == 1 ? b1 : g == 2 ? b2 : b3;
            poolUsed = b3 + n3;
            // the padding entries behind a list (<= 3 + 28) take part in the wave-wide pass below: give them a valid pixel (row 0, column 0)
            const int padEnd = g == 0 ? b1 : g == 1 ? b2 : g == 2 ? b3 : b3 + n3;
            for (int q = base + nvalid + l; q < padEnd; q += 16) { s_pool[2][q] = 0.0f; s_pool[3][q] = 0.0f; s_pool[4][q] = 0.0f; s_pool[5][q] = 0.0f; }
        }
        int run = base;
#pragma unroll
        for (int m = 0; m < 4; m++) {   // ordered compaction in window raster order = (iteration, lane, element)
            const unsigned q = (vm >> (4 * m)) & 0xFu;
            const int c = __popc(q);
            const int incl = row_incl_scan(c);
            int o = run + incl - c;
            const int rc = ((yb + 4 * m + rq + wrapRow) << 16) | wcol0;   // a valid pixel lies inside the image
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (q & (1u << e)) {   // depth, right depth, down depth, (row, col)
                    float right = e < 3 ? dq[m].v[e < 3 ? e + 1 : 3] : dr3[m];
                    int rce = rc + e;
                    if (straddle) {   // (row, col) and the right neighbour of an element of a straddling quad, fetched here (rare)
                        const int we = elem_wrap(e), col = cx0 + e - we * P.W, rowe = yb + 4 * m + rq + we;
                        rce = (rowe << 16) | col;
                        right = F.depthG()[(size_t)rowe * P.dstride + min(col + 1, P.W - 1)];
                    }
                    s_pool[2][o] = dq[m].v[e]; s_pool[3][o] = right;
                    s_pool[4][o] = ddq[m].v[e]; s_pool[5][o] = __int_as_float(rce);
                    o++;
                }
            run += row_lane_i32<15>(incl);
        }
    }
    SECTION_STAMP();   // 1: gather + ordered lists
    float *const pX = s_pool[0] + base, *const pY = s_pool[1] + base, *const pZ = s_pool[2] + base;
    float *const qX = s_pool[3] + base, *const qY = s_pool[4] + base, *const qZ = s_pool[5] + base;
    maxDist = row_max_f32(maxDist);
    __builtin_amdgcn_wave_barrier();
    // entry e -> position + cross-product normal, written back in place (order preserved).  The work per entry does not depend on the seed, so
    // the 64 lanes walk the whole pool together: ceil(pool / 64) rounds instead of ceil(longest list / 16) -- the four superpixels of a wave
    // rarely have the same size.
    for (int e = lane; e < poolUsed; e += 64) {
        const int rc = __float_as_int(s_pool[5][e]);
        const int row = rc >> 16, col = rc & 0xFFFF;     // a valid pixel lies inside the image: (row, col) of its flat index
        const float myDepth = s_pool[2][e], rightD = s_pool[3][e], downD = s_pool[4][e];
        const float cxr = P.colX[col], cx1 = P.colX[col + 1], ryr = P.rowY[row], ry1 = P.rowY[row + 1];
        const float x = cxr * myDepth, y = ryr * myDepth;   // back_project(col, row, myDepth)
        float nX, nY, nZ;
        pixel_normal(P, row, col, x, y, myDepth, rightD, downD, cxr, cx1, ryr, ry1, nX, nY, nZ);
        s_pool[0][e] = x; s_pool[1][e] = y;
        s_pool[3][e] = nX; s_pool[4][e] = nY; s_pool[5][e] = nZ;
    }
    __builtin_amdgcn_wave_barrier();
    SECTION_STAMP();   // 2: positions + pixel normals
    bool active = inRange && nvalid >= 16;   // validDepthNum < 16 -> continue (:702)
    float meanDepth = S.meanDepth;
    // ---- inliers, kept in order (:707-720).  Count first: when every valid pixel is an inlier (the common case)
    // the list is already in place; otherwise in-place ordered compaction, 16 entries per round. ----
    int ninl = 0;
    {
        int c = 0;
        if (active)
            for (int o = l; o < nvalid; o += 16) {
                const float residual = meanDepth - pZ[o];
                c += in_huber_band(residual) ? 1 : 0;
            }
        ninl = row_sum_i32(c);
    }
    const bool needCompact = active && ninl != nvalid;
    if (__ballot(needCompact)) {
        int w0 = 0;
        for (int t = 0; t < 16; t++) {
            const int o = t * 16 + l;
            bool inl = false;
            float a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
            if (needCompact && o < nvalid) {
                const float residual = meanDepth - pZ[o];
                inl = in_huber_band(residual);
                a0 = pX[o]; a1 = pY[o]; a2 = pZ[o];
                b0 = qX[o]; b1 = qY[o]; b2 = qZ[o];
            }
            const unsigned gm = (unsigned)((__ballot(inl) >> (g * 16)) & 0xFFFFull);
            __builtin_amdgcn_wave_barrier();   // every lane has read its slot before anyone overwrites (w <= o)
            if (inl) {
                const int w = w0 + __popc(gm & ((1u << l) - 1u));
                pX[w] = a0; pY[w] = a1; pZ[w] = a2;
                qX[w] = b0; qY[w] = b1; qZ[w] = b2;
            }
            w0 += __popc(gm);
            __builtin_amdgcn_wave_barrier();
        }
    }
    SECTION_STAMP();   // 3: inlier count / compaction
    if (active && (float)ninl / (float)nvalid < 0.8) active = false;
    // Six strictly sequential f32 sums (inlier normals x,y,z and positions x,y,z, :709-713 and :95-99) run side by side:
    // lane q < 6 of the group walks array q in list order, so the serial latency is one chain instead of six.
    float normX, normY, normZ, sumX, sumY, sumZ;
    {
        float acc = 0.0f;
        if (active && l < 6) acc = seq_sum_f32(s_pool[l < 3 ? 3 + l : l - 3] + base, ninl, 0.0f);
        normX = row_lane_f32<0>(acc); normY = row_lane_f32<1>(acc); normZ = row_lane_f32<2>(acc);
        sumX = row_lane_f32<3>(acc); sumY = row_lane_f32<4>(acc); sumZ = row_lane_f32<5>(acc);
        const float normLength = sqrtf(normX * normX + normY * normY + normZ * normZ);
        normX = normX / normLength; normY = normY / normLength; normZ = normZ / normLength;
        sumX /= ninl; sumY /= ninl; sumZ /= ninl;
    }
    SECTION_STAMP();   // 4: six sequential sums
    // ---- getHuberNorm (:91-165): 5 Gauss-Newton steps, FP64 normal equations reduced over the 16 lanes ----
    float nx = normX, ny = normY, nz = normZ, nb = 0.0f;
    // The Hessian depends only on WHICH points lie inside the Huber band; while that set is unchanged between
    // iterations (the common case: all of them) its sums -- and the inverse -- are bit-identical and are reused.
    unsigned prevMask = 0xFFFFFFFFu;   // impossible mask: forces the first evaluation
    // Cooperative 4x4 inverse: lane l = 4a+b of the group evaluates cofactor (a,b) with exactly the DET3 expression of
    // inverse4(), so lane l ends up holding inv[l] (column-major) -- 1/16 of the work and 2 instead of 32 registers.
    double invl = 0;
    const int ca = l >> 2, cb = l & 3;
    const int r0 = ca == 0 ? 1 : 0, r1 = ca <= 1 ? 2 : 1, r2 = ca <= 2 ? 3 : 2;
    const int c0 = cb == 0 ? 1 : 0, c1 = cb <= 1 ? 2 : 1, c2 = cb <= 2 ? 3 : 2;
    const int tRounds = rows_max_i32(active ? (ninl + 15) >> 4 : 0);   // (ninl and active are uniform inside a group of 16 lanes)
    // The centred points (`points[i] -= sum`, :107-111, once in the reference) of the first RREG rounds stay in registers through the five steps: the
    // loops re-read and re-centred every point from the LDS in every step (three reads and three subtractions per point and step, and an LDS round trip
    // per round that one or two waves per SIMD do not hide); rounds beyond RREG (lists longer than 96 points) still do.
    constexpr int RREG = 6;
    float cpx[RREG], cpy[RREG], cpz[RREG];
#pragma unroll
    for (int t = 0; t < RREG; t++) {
        const int o = l + 16 * t, oc = (active && o < ninl) ? o : 0;
        cpx[t] = cpy[t] = cpz[t] = 0.0f;
        if (t < tRounds) { cpx[t] = pX[oc] - sumX; cpy[t] = pY[oc] - sumY; cpz[t] = pZ[oc] - sumZ; }
    }
    for (int gnI = 0; gnI < 5; gnI++) {
        double J0 = 0, J1 = 0, J2 = 0, J3 = 0;
        unsigned mask = 0;
        // One round of the Jacobian.  Branch-free for the common case (every point inside the Huber band): a lane without a point in this round, or whose
        // point is outside the band, adds +0.0 to its four sums -- exact: a sum that starts at +0.0 never becomes -0.0 -- with the coordinates replaced by
        // zeros BEFORE the products (a NaN / infinite coordinate of a point outside the band must not reach them).  The nested per-lane branches of the
        // literal form cost four taken branches per round, which one or two waves per SIMD cannot hide.  Points outside the band take the reference's two
        // tail cases behind ONE wave-uniform test.
        auto jstep = [&](int t, float px, float py, float pz) {
            const bool has = l + 16 * t < ninl;
            const float residual = px * nx + py * ny + pz * nz + nb;
            const bool inb = has && in_huber_band(residual);
            mask |= (inb ? 1u : 0u) << t;
            const float r2 = inb ? 2 * residual : 0.0f;
            const float qx = inb ? px : 0.0f, qy = inb ? py : 0.0f, qz = inb ? pz : 0.0f;
            J0 += r2 * qx; J1 += r2 * qy; J2 += r2 * qz; J3 += r2;
            if (__builtin_expect(__ballot(has && !inb) != 0ull, 0)) {
                if (has && residual >= HUBER_RANGE_F) {
                    J0 += HUBER_RANGE * px; J1 += HUBER_RANGE * py; J2 += HUBER_RANGE * pz; J3 += HUBER_RANGE;
                } else if (has && residual <= -HUBER_RANGE_F) {
                    J0 += -1 * HUBER_RANGE * px; J1 += -1 * HUBER_RANGE * py; J2 += -1 * HUBER_RANGE * pz; J3 += -1 * HUBER_RANGE;
                }
            }
        };
        if (active) {
#pragma unroll
            for (int t = 0; t < RREG; t++)
                if (t < tRounds) jstep(t, cpx[t], cpy[t], cpz[t]);   // (a wave-uniform bound: the longest inlier list of the four seeds, typically 4-6 of the 16 rounds)
#pragma unroll 1
            for (int t = RREG; t < tRounds; t++) {
                const int o = l + 16 * t, oc = o < ninl ? o : 0;
                jstep(t, pX[oc] - sumX, pY[oc] - sumY, pZ[oc] - sumZ);
            }
        }
        J0 = group_sum_d(J0); J1 = group_sum_d(J1); J2 = group_sum_d(J2); J3 = group_sum_d(J3);
        const bool sameSet = mask == prevMask;
        const unsigned diffGroups = (unsigned)((__ballot(!sameSet) >> (g * 16)) & 0xFFFFull);   // uniform per group
        prevMask = mask;
        if (__ballot(diffGroups != 0)) {
            double H00 = 0, H01 = 0, H02 = 0, H03 = 0, H11 = 0, H12 = 0, H13 = 0, H22 = 0, H23 = 0, H33 = 0;
            if (active && diffGroups) {
                auto hstep = [&](int t, float rx, float ry, float rz) {   // (branch-free like the Jacobian: a lane without an in-band point in this round adds zeros)
                    const bool inb = (mask >> t) & 1u;
                    const float px = inb ? rx : 0.0f, py = inb ? ry : 0.0f, pz = inb ? rz : 0.0f;
                    H00 += 2 * px * px; H01 += 2 * px * py; H02 += 2 * px * pz; H03 += 2 * px;
                    H11 += 2 * py * py; H12 += 2 * py * pz; H13 += 2 * py;
                    H22 += 2 * pz * pz; H23 += 2 * pz; H33 += inb ? 2.0 : 0.0;
                };
#pragma unroll
                for (int t = 0; t < RREG; t++)
                    if (t < tRounds) hstep(t, cpx[t], cpy[t], cpz[t]);
#pragma unroll 1
                for (int t = RREG; t < tRounds; t++) {
                    const int oc = ((mask >> t) & 1u) ? l + 16 * t : 0;
                    hstep(t, pX[oc] - sumX, pY[oc] - sumY, pZ[oc] - sumZ);
                }
            }
            H00 = group_sum_d(H00); H01 = group_sum_d(H01); H02 = group_sum_d(H02); H03 = group_sum_d(H03);
            H11 = group_sum_d(H11); H12 = group_sum_d(H12); H13 = group_sum_d(H13);
            H22 = group_sum_d(H22); H23 = group_sum_d(H23); H33 = group_sum_d(H33);
            if (l == 0) {   // the (symmetric) Hessian + 5 I, column-major
                double *m = s_h[g];
                m[0] = H00 + 5; m[1] = H01; m[2] = H02; m[3] = H03; m[4] = H01; m[5] = H11 + 5; m[6] = H12; m[7] = H13;
                m[8] = H02; m[9] = H12; m[10] = H22 + 5; m[11] = H23; m[12] = H03; m[13] = H13; m[14] = H23; m[15] = H33 + 5;
            }
            __builtin_amdgcn_wave_barrier();
            {
                const double *m = s_h[g];
#define M_(r, c) m[(c) * 4 + (r)]
                const double d3 = M_(r0, c0) * (M_(r1, c1) * M_(r2, c2) - M_(r1, c2) * M_(r2, c1)) -
                                  M_(r0, c1) * (M_(r1, c0) * M_(r2, c2) - M_(r1, c2) * M_(r2, c0)) +
                                  M_(r0, c2) * (M_(r1, c0) * M_(r2, c1) - M_(r1, c1) * M_(r2, c0));
                const double cof = ((ca + cb) & 1) ? -d3 : d3;
                const double f0 = dpp_mov_d<0x150>(cof), f1 = dpp_mov_d<0x151>(cof), f2 = dpp_mov_d<0x152>(cof), f3 = dpp_mov_d<0x153>(cof);   // lanes 0..3 of the group
                const double det = ((M_(0, 0) * f0 + M_(0, 1) * f1) + M_(0, 2) * f2) + M_(0, 3) * f3;
#undef M_
                if (diffGroups) invl = cof / det;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // upd[r] = ((inv[0*4+r] J0 + inv[1*4+r] J1) + inv[2*4+r] J2) + inv[3*4+r] J3; lane l holds inv[l], its column is l >> 2
        const double prod = invl * (ca == 0 ? J0 : ca == 1 ? J1 : ca == 2 ? J2 : J3);
        // lanes r = 0..3 of the group (column 0, row r) collect their row: lane r + 4 a holds the term of column a -- row_ror:n hands lane i the value of
        // lane i - n (mod 16), so n = 12, 8, 4 fetch the lanes 4, 8, 12 ahead; the other lanes compute sums nobody reads
        const double q0 = prod, q1 = dpp_mov_d<0x12C>(prod), q2 = dpp_mov_d<0x128>(prod), q3 = dpp_mov_d<0x124>(prod);
        const double updr = ((q0 + q1) + q2) + q3;            // lane r < 4 of the group now holds upd[r]
        const double u0 = dpp_mov_d<0x150>(updr), u1 = dpp_mov_d<0x151>(updr), u2 = dpp_mov_d<0x152>(updr), u3 = dpp_mov_d<0x153>(updr);
        nx = (float)((double)nx - u0); ny = (float)((double)ny - u1); nz = (float)((double)nz - u2); nb = (float)((double)nb - u3);
        SECTION_STAMP();   // 5-9: Gauss-Newton steps
    }
#ifdef MSL_FUSE_STAMPS
    if (slot == 0 && lane == 0) {
        for (int q = 1; q < ssn; q++) atomicAdd(&P.delList[64 + q], (unsigned)(sst[q] - sst[q - 1]));
        atomicAdd(&P.delList[64], 1u);
    }
#endif
    // The per-seed rest -- the plane's normalisation, the seed record, FuseRec and the candidate surfel: ~310 instructions that only ONE lane of a
    // seed's sixteen would execute here (4 of 64 lanes busy) -- runs in kb_seed_finish, one thread per seed.  What it needs of this kernel travels in the
    // seed's slot of the candidate array, which kb_seed_finish itself overwrites afterwards.
    if (!inRange || l != 0) return;
    PlaneFit T;
    T.active = active ? 1 : 0; T.nx = nx; T.ny = ny; T.nz = nz; T.nb = nb; T.sumX = sumX; T.sumY = sumY; T.sumZ = sumZ; T.maxDist = maxDist;
    __builtin_memcpy(reinterpret_cast<char *>(P.cand + ((size_t)slot * P.nseeds + seedI)), &T, sizeof(T));
}

// kb_seed_finish: the end of calculateNorms for one seed (:744-773: plane normalisation, the seed's position on the plane, viewCos, size), then what the
// map stage reads of the seed (FuseRec) and the surfel it would spawn (initializeSurfels, :291-329).  One thread per seed; same expressions, same
// operands as the reference, fed by the fit kb_seed_plane left in the seed's candidate slot.
__global__ __launch_bounds__(256) void kb_seed_finish(SfDev P) {
    const int slot = blockIdx.y;
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    const FrameDev &F = P.frames[slot];
    msl_seed S = P.seeds[(size_t)slot * P.nseeds + seedI];
    PlaneFit T;
    __builtin_memcpy(&T, reinterpret_cast<const char *>(P.cand + ((size_t)slot * P.nseeds + seedI)), sizeof(T));
    const bool active = T.active != 0;
    float nx = T.nx, ny = T.ny, nz = T.nz, nb = T.nb;
    const float sumX = T.sumX, sumY = T.sumY, sumZ = T.sumZ, maxDist = T.maxDist;
    float normX, normY, normZ, meanDepth = S.meanDepth;
    if (active) {
        nb = nb - (nx * sumX + ny * sumY + nz * sumZ);
        {
            const float normLength = sqrtf(nx * nx + ny * ny + nz * nz);
            nx /= normLength; ny /= normLength; nz /= normLength; nb /= normLength;
        }
        normX = nx; normY = ny; normZ = nz;
        const float normB = nb;
        float ax, ay, az;
        back_project(P, S.x, S.y, meanDepth, ax, ay, az);
        double avgX = ax, avgY = ay, avgZ = az;
        {
            const float k = (float)(-1 * (avgX * (double)normX + avgY * (double)normY + avgZ * (double)normZ) - (double)normB);
            avgX += (double)(k * normX); avgY += (double)(k * normY); avgZ += (double)(k * normZ);
            meanDepth = (float)avgZ;
        }
        float viewCos = (float)(-1.0 * ((double)normX * avgX + (double)normY * avgY + (double)normZ * avgZ) / sqrt(avgX * avgX + avgY * avgY + avgZ * avgZ));
        if (viewCos < 0) { viewCos = -viewCos; normX = -normX; normY = -normY; normZ = -normZ; }
        S.normX = normX; S.normY = normY; S.normZ = normZ;
        S.posX = (float)avgX; S.posY = (float)avgY; S.posZ = (float)avgZ;
        S.meanDepth = meanDepth; S.viewCos = viewCos; S.size = sqrtf(maxDist);
        P.seeds[(size_t)slot * P.nseeds + seedI] = S;
    }
    // what the map stage reads of this seed (FuseRec) and the candidate new surfel (:291-329, everything except the `fused` test,
    // which needs the map stage); both use the same per-seed terms
    const bool valid = !(S.viewCos < MAX_ANGLE_COS) && !(S.normX == 0 && S.normY == 0 && S.normZ == 0);
    const bool ok = valid && !(S.meanDepth == 0);
    P.candOk[(size_t)slot * P.flagStride + seedI] = ok ? 1 : 0;
    if (!ok) P.fused[(size_t)slot * P.flagStride + seedI] = 2;   // "spawns nothing" for the deferred map stage's one-array scan (kb_seed_init cleared the byte; a fusion writes 1)
    float pw[4] = {0, 0, 0, 0};
    float seedWeight = 0, seedSize = 0;
    if (valid) {
        mul4(F.pose, S.posX, S.posY, S.posZ, 1.0f, pw);
        const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
        seedSize = S.size * fabsf(S.meanDepth / (cameraF * S.viewCos));
        seedWeight = get_weight(S.meanDepth);
    }
    {
        float4 *fr = P.fuseRec + (size_t)slot * P.nseeds * 3;
        fr[fuserec_index(P.nseeds, seedI, 0)] = make_float4(S.normX, S.normY, S.normZ, S.meanDepth);
        fr[fuserec_index(P.nseeds, seedI, 1)] = make_float4(pw[0], pw[1], pw[2], seedWeight);
        fr[fuserec_index(P.nseeds, seedI, 2)] = make_float4(seedSize, S.meanIntensity, __uint_as_float(rgb_pack(S.r, S.g, S.b)), __uint_as_float(valid ? 1u : 0u));
    }
    if (ok) {
        float nw[3];
        mul3(F.pose, S.normX, S.normY, S.normZ, nw);
        msl_surfel e;
        e.px = pw[0]; e.py = pw[1]; e.pz = pw[2];
        e.r = S.r; e.g = S.g; e.b = S.b;
        e.nx = nw[0]; e.ny = nw[1]; e.nz = nw[2];
        e.size = seedSize;
        e.color = S.meanIntensity;
        e.weight = seedWeight;
        e.updateTimes = 1;
        e.lastUpdate = F.ref;
        P.cand[(size_t)slot * P.nseeds + seedI] = e;
    }
}

// Raw 16-bit depth (what the sensor / the data set's PNG holds) -> metres, on the device: imDepth.convertTo(imDepthScaled, CV_32F, depthMapFactor) of
// src/Frame.cc:96-97.  OpenCV's 16U -> 32F conversion with a scale works in float: dst = (float)src * (float)alpha + 0.0f (one rounding: the float
// product), which is what this kernel evaluates; a third of the bytes cross PCIe (2 instead of 4 per pixel) and the host loop disappears.
__global__ __launch_bounds__(256) void kb_depth_u16(const uint8_t *src, size_t srcStride, size_t srcFrameStride, float *dst, size_t dstFrameStride, int W, int H,
                                                    float factor) {
    const int frame = blockIdx.y;
    const int npx = W * H;
    const uint8_t *sf = src + (size_t)frame * srcFrameStride;
    float *df = dst + (size_t)frame * dstFrameStride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int row = i / W, col = i - row * W;
        const unsigned raw = *reinterpret_cast<const uint16_t *>(sf + (size_t)row * srcStride + 2 * (size_t)col);
        df[i] = (float)raw * factor;
    }
}

// Image sizes that are not multiples of 8: the strips right of / below the last whole 8x8 cell belong to no cell, so kb_seed_plane does not
// write their texels; this (tiny, rarely launched) kernel does.
__global__ __launch_bounds__(256) void kb_tex_strips(SfDev P) {
    const int slot = blockIdx.y;
    const int wStrip = P.W - P.spW * SP, hStrip = P.H - P.spH * SP;
    const int nRight = wStrip * P.spH * SP, nBottom = P.W * hStrip;   // right strip over the cell rows, bottom strip over the full width
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nRight + nBottom) return;
    int x, y;
    if (i < nRight) { y = i / wStrip; x = P.spW * SP + i % wStrip; }
    else { const int j = i - nRight; y = P.spH * SP + j / P.W; x = j % P.W; }
    const FrameDev F = P.frames[slot];
    const size_t p = (size_t)y * P.W + x;
    P.tex[(size_t)slot * P.pxStride + p] = make_uint2(__float_as_uint(F.depthG()[(size_t)y * P.dstride + x]), P.index[(size_t)slot * P.pxStride + p]);
}

// Test hook of the rotating chains: list q (<= 256 floats at x + 256 q, n[q] of them valid) is summed by the 16 lanes of row q & 3 of wave q >> 2 exactly
// the way kb_update_seeds sums a seed's depth list (huber = 0) or its Huber terms (huber = 1: +-inf entries mark tail elements).
__global__ __launch_bounds__(64) void k_debug_chain(const float *x, const int *n, float *out, int lists, int huber) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    const int nq = q < lists ? n[q] : 0;
    int nblk = (nq + 15) >> 4;
#pragma unroll
    for (int d = 32; d >= 16; d >>= 1) nblk = max(nblk, __shfl_xor(nblk, d, 64));
    nblk = __builtin_amdgcn_readfirstlane(nblk);
    float s = 0.0f;
    for (int bq = 0; bq < nblk; bq++) {
        const int e = l + 16 * bq;
        const float t = e < nq ? x[(size_t)q * 256 + e] : 0.0f;
        s = huber ? chain_block_huber(s, t) : chain_block_f32(s, t);
    }
    if (q < lists && l == 15) out[q] = s;
}

__global__ void k_debug_div100(const float *x, double *out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = div100_exact((double)(x[i] * x[i]));
}

}  // namespace

namespace msl {
namespace sf {

void sp_launch_depth_u16(hipStream_t st, const void *src, size_t srcStride, size_t srcFrameStride, float *dst, size_t dstFrameStride, int W, int H, int nFrames,
                         float factor) {
    const unsigned bx = (unsigned)std::min(((long long)W * H + 1023) / 1024, 1024ll);   // four pixels per thread
    hipLaunchKernelGGL(kb_depth_u16, dim3(bx, (unsigned)nFrames), dim3(256), 0, st, (const uint8_t *)src, srcStride, srcFrameStride, dst, dstFrameStride, W, H, factor);
}

bool sp_init_attributes(int nseeds) {
    // the attribute belongs to the function, not to a handle: always ask for the largest size any handle may use
    if (nseeds > PROP_LDS_MAX_SEEDS) return false;
    return hipFuncSetAttribute((const void *)kb_prop_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned) * PROP_LDS_MAX_SEEDS)) == hipSuccess;
}

// Superpixel stage of nFrames keyframes; P's per-slot pointers address the first of them (blockIdx.y / xcd_slot() == 0).
void sp_launch_stage(KernelProfiler &prof, hipStream_t sp, const SfDev &P, int n, bool propLds) {
    const int W = P.W, H = P.H;
    const unsigned un = (unsigned)n;
    const dim3 seedGrid((P.nseeds + 255) / 256, un);
    const int nbx = ((W - 5) >> 3) + 2, nby = ((H - 5) >> 3) + 2;   // dual cells [8 b + 4, 8 b + 12), b from -1, that meet the image
    const dim3 pxGrid(xcd_grid(((nbx + 3) / 4) * ((nby + ASSIGN_NY - 1) / ASSIGN_NY), n)), flatPx(xcd_grid(((P.npx + 7) / 8 + 255) / 256, n));
    MSL_SF_LAUNCH(prof, SK_SEED_INIT, sp, kb_seed_init, seedGrid, dim3(256), P);
    for (int it = 0; it < 3; it++) {
        MSL_SF_LAUNCH(prof, SK_ASSIGN, sp, kb_assign, pxGrid, dim3(256), P, it, n, nbx, nby);
        if (it > 0) {
            prof.begin(SK_PROP, sp);
            if (propLds) {
                hipLaunchKernelGGL(kb_prop_lds, dim3(un), dim3(256), sizeof(unsigned) * P.nseeds, sp, P);
            } else {
                for (int r = 0; r < PROP_ROUNDS; r++) hipLaunchKernelGGL(kb_prop, dim3(xcd_grid(PROP_BLOCKS, n)), dim3(256), 0, sp, P, r, n);
                hipLaunchKernelGGL(kb_prop_finish, dim3(un), dim3(1024), 0, sp, P);
            }
            prof.end(sp);
            MSL_SF_LAUNCH(prof, SK_COMMIT_PX, sp, kb_commit_px, flatPx, dim3(256), P, n);
        }
        // (8 workgroups per CU: 17 KB of LDS, 56 VGPRs.  Capped at 7 / 6 / 5 by unused dynamic LDS: 23.1 / 23.2 / 22.7 k frames/s against 23.1 k -- no
        // sweet spot below the maximum, unlike kb_seed_plane; at the 4 of rounds 2-4 the kernel took twice as long)
        if ((W % SP) >= 1 && (W % SP) <= 3) MSL_SF_LAUNCH(prof, SK_UPDATE_SEEDS, sp, kb_update_seeds<true>, dim3(xcd_grid((P.nseeds + 15) / 16, n)), dim3(256), P, it, n);
        else MSL_SF_LAUNCH(prof, SK_UPDATE_SEEDS, sp, kb_update_seeds<false>, dim3(xcd_grid((P.nseeds + 15) / 16, n)), dim3(256), P, it, n);
        MSL_SF_LAUNCH(prof, SK_COMMIT_SEEDS, sp, kb_commit_seeds, seedGrid, dim3(256), P, it);
    }
    // 4 KB of (unused) dynamic LDS cap the kernel at 8 waves per CU (it could run 10; 6 KB = 7 waves until the end of round 6).  Measured on the whole front end, alternating runs on one
    // box -- round 3: 11 waves 19.6 k frames/s, 10 waves 20.6-20.9 k, 9 waves 21.0-21.1 k, 8 waves 21.3-21.5 k, 7 waves 20.3-20.9 k; round 5,
    // beside the slimmer kb_update_seeds: 10 waves 21.9 k, 9 waves 23.1-23.3 k, 8 waves 22.8-23.1 k, 7 waves 23.3-23.5 k, 6 waves 22.8 k, 5 waves
    // 22.5 k.  The kernel alone is no slower with fewer waves (its waves are VALU-latency bound), and the wave slots, registers and LDS it leaves
    // go to the ORB kernels and to the map stage that run beside it.
    // round 6, after the per-seed epilogue moved to kb_seed_finish (88 VGPRs, 339 us per launch at this cap): 10 waves per CU 281 us / 21.9 k frames/s, 8 waves
    // 306 us / 23.2-23.4 k, 7 waves 339 us / 24.0-24.2 k, 6 waves 384 us / 23.9 k -- the sweet spot did not move
    // ... and again with the centred points of the Gauss-Newton steps in registers (156 VGPRs: two waves fill a SIMD's budget more evenly than 7 per CU did): 10 waves
    // per CU 268 us / 26.2 k frames/s but config 3 30.5 k, **8 waves 291 us / 25.8 k / config 3 31.5 k**, 7 waves 321 us / 25.1 k / 30.9 k, 6 waves 361 us / 24.5 k / 29.9 k
    constexpr unsigned planePad = 4096;
    if ((W % SP) >= 1 && (W % SP) <= 3) MSL_SF_LAUNCH_LDS(prof, SK_SEED_PLANE, sp, kb_seed_plane<true>, dim3(xcd_grid(((P.spW + 1) / 2) * ((P.spH + 1) / 2), n)), dim3(64), planePad, P, n);
    else MSL_SF_LAUNCH_LDS(prof, SK_SEED_PLANE, sp, kb_seed_plane<false>, dim3(xcd_grid(((P.spW + 1) / 2) * ((P.spH + 1) / 2), n)), dim3(64), planePad, P, n);
    hipLaunchKernelGGL(kb_seed_finish, seedGrid, dim3(256), 0, sp, P);
    if ((W % SP) || (H % SP)) {   // pixels outside the whole cells (sizes that are not multiples of 8)
        const int nStrip = (W - P.spW * SP) * P.spH * SP + W * (H - P.spH * SP);
        hipLaunchKernelGGL(kb_tex_strips, dim3((unsigned)((nStrip + 255) / 256), un), dim3(256), 0, sp, P);
    }
}

int sp_debug_chain(const float *x_host, const int32_t *n_host, int lists, int huber, float *out_host) {
    if (lists <= 0) return MSL_OK;
    if (!x_host || !n_host || !out_host) return MSL_ERR_INVALID;
    for (int q = 0; q < lists; q++) if (n_host[q] < 0 || n_host[q] > 256) return MSL_ERR_INVALID;
    float *dx = nullptr, *dout = nullptr; int *dn = nullptr;
    MSL_HIP_TRY(hipMalloc(&dx, sizeof(float) * 256 * (size_t)lists)); MSL_HIP_TRY(hipMalloc(&dn, sizeof(int) * (size_t)lists)); MSL_HIP_TRY(hipMalloc(&dout, sizeof(float) * (size_t)lists));
    MSL_HIP_TRY(hipMemcpy(dx, x_host, sizeof(float) * 256 * (size_t)lists, hipMemcpyHostToDevice));
    MSL_HIP_TRY(hipMemcpy(dn, n_host, sizeof(int) * (size_t)lists, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_chain, dim3((unsigned)((lists + 3) / 4)), dim3(64), 0, 0, dx, dn, dout, lists, huber);
    MSL_HIP_TRY(hipMemcpy(out_host, dout, sizeof(float) * (size_t)lists, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dn); (void)hipFree(dout);
    return MSL_OK;
}

int sp_debug_div100(const float *x_host, double *out_host, size_t n) {
    if (n == 0) return MSL_OK;
    if (!x_host || !out_host) return MSL_ERR_INVALID;
    float *dx = nullptr; double *dout = nullptr;
    MSL_HIP_TRY(hipMalloc(&dx, sizeof(float) * n)); MSL_HIP_TRY(hipMalloc(&dout, sizeof(double) * n));
    MSL_HIP_TRY(hipMemcpy(dx, x_host, sizeof(float) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_div100, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dout, (long long)n);
    MSL_HIP_TRY(hipMemcpy(out_host, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout);
    return MSL_OK;
}

}  // namespace sf
}  // namespace msl
