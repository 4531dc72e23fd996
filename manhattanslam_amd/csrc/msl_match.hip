// msl_match.hip -- batched Hamming matching by projection for gfx950.
//
// Replaces ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th) (reference src/ORBmatcher.cc:547-678)
// with Frame::GetFeaturesInArea (src/Frame.cc:332-381), ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:835-849) and the
// rotation-consistency histogram (ComputeThreeMaxima, :799-830), for many independent frame pairs per call.
//
// MI355X-first structure (frame-batched: blockIdx.y = pair), three launches per call:
//   k_match_grid        one workgroup per pair: the 64 x 48 feature grid of the current frame as a cell-sorted item list
//                       (counting sort in LDS; ascending cell id, ascending keypoint index inside a cell = mGrid insertion order)
//                       + the forward / backward search mode of the pair (:560-571);
//   k_match_candidates  one wave per last-frame point: projection (:577-593), window cells spread over the lanes, level /
//                       distance / mvuRight filters (:596-628, Frame.cc:353-376), 256-bit Hamming distance by v_bcnt on 8 dwords;
//                       candidates are stored as (distance << 16 | position in the item list).  The window is walked in
//                       ascending cell id and a cell's items in insertion order, so "first candidate wins ties" (:635) is the
//                       minimum of that packed key and the candidates need not be stored in order;
//   k_match_assign      one workgroup per pair: the reference's greedy, order dependent assignment (:621-623: a candidate held by
//                       a point with Observations() > 0 is skipped; :638: later points overwrite earlier ones) as a min-fixpoint:
//                       t(i2) = first point with observations that picks i2; point i skips candidates with t < i.  By induction
//                       over i the fixpoint is unique and equals the sequential result; it is reached in a handful of rounds
//                       (bounded by n_last).  Then holder = last picker, rotation histogram, three maxima, NULLing (:657-674).
//
// Integer / byte work, L2-resident gathers; no MFMA.  The float expressions keep the reference's order; the 3x3 cv::Mat products
// follow cv::gemm's float kernel (double accumulation, one rounding) -- pinned in DESIGN.md section 3.
#include "msl_common.h"

#include <mutex>
#include <new>

using namespace msl;

namespace {

constexpr int GRID_ROWS = MSL_FRAME_GRID_ROWS, GRID_COLS = MSL_FRAME_GRID_COLS, NCELLS = GRID_ROWS * GRID_COLS;
constexpr int TH_HIGH = 100, HISTO_LENGTH = 30;     // src/ORBmatcher.cc:33-35
constexpr int CMAX = 32;                            // stored candidates per point; more are re-enumerated by k_match_assign
constexpr int MAX_CAP = 8192;
constexpr unsigned T_NONE = 0xFFFFFFFFu;

struct MatchDev {
    int nPairs, cap;
    msl_match_params prm;
    float gridWInv, gridHInv, mb;
    const msl_keypoint *curKps; const float *curUn; const float *curUright; const int32_t *curCell; const uint8_t *curDesc; const int32_t *nCur;
    const float *lastXyz; const uint8_t *lastDesc; const uint8_t *lastFlags; const int32_t *lastOctave; const float *lastAngle; const int32_t *nLast;
    const float *TcwCur, *TcwLast;
    // scratch
    unsigned short *items;     // [nPairs][cap]      keypoint indices sorted by (cell, index)
    unsigned *cellStart;       // [nPairs][NCELLS+1]
    int *mode;                 // [nPairs]           0: octave +-1, 1: forward, 2: backward
    unsigned *cand;            // [nPairs][cap][CMAX] dist << 16 | item position
    unsigned *candCnt;         // [nPairs][cap]      total candidates of the point (may exceed CMAX)
    int32_t *matchOut, *nmatches;
};

// d[r] = (float)(alpha * sum_k A(r, k) b[k] + c[r]) with double accumulation: cv::gemm's CV_32F kernel
__device__ __forceinline__ void gemm3(const float *A, bool transA, double alpha, const float b[3], const float *c, float d[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += (double)(transA ? A[k * 4 + r] : A[r * 4 + k]) * (double)b[k];
        d[r] = (float)(s * alpha + (c ? (double)c[r] : 0.0));
    }
}

// ---- k_match_grid ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_grid(MatchDev P) {
    __shared__ unsigned s_start[NCELLS + 1];
    __shared__ unsigned s_fill[NCELLS];
    __shared__ unsigned s_wave[17];
    extern __shared__ unsigned short s_items[];   // [cap]
    const int pair = blockIdx.x;
    const int n = min(P.nCur[pair], P.cap);
    const int32_t *cell = P.curCell + (size_t)pair * P.cap;
    for (int c = threadIdx.x; c <= NCELLS; c += 256) { s_start[c] = 0; if (c < NCELLS) s_fill[c] = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = cell[i];
        if (c >= 0 && c < NCELLS) atomicAdd(&s_start[c + 1], 1u);
    }
    __syncthreads();
    block_scan_array_incl(s_start + 1, NCELLS, s_wave);     // s_start[c] = first item of cell c, s_start[NCELLS] = total
    for (int i = threadIdx.x; i < n; i += 256) {            // unordered placement inside each cell ...
        const int c = cell[i];
        if (c >= 0 && c < NCELLS) s_items[s_start[c] + atomicAdd(&s_fill[c], 1u)] = (unsigned short)i;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < NCELLS; c += 256) {       // ... then ascending keypoint index per cell (cells hold a few items)
        const unsigned b = s_start[c], e = s_start[c + 1];
        for (unsigned a = b + 1; a < e; a++) {
            const unsigned short v = s_items[a];
            unsigned j = a;
            while (j > b && s_items[j - 1] > v) { s_items[j] = s_items[j - 1]; j--; }
            s_items[j] = v;
        }
    }
    __syncthreads();
    const unsigned total = s_start[NCELLS];
    for (unsigned i = threadIdx.x; i < total; i += 256) P.items[(size_t)pair * P.cap + i] = s_items[i];
    for (int c = threadIdx.x; c <= NCELLS; c += 256) P.cellStart[(size_t)pair * (NCELLS + 1) + c] = s_start[c];
    if (threadIdx.x == 0) {   // bForward / bBackward (:560-571)
        const float *Tc = P.TcwCur + (size_t)pair * 12, *Tl = P.TcwLast + (size_t)pair * 12;
        const float tcw[3] = {Tc[3], Tc[7], Tc[11]}, tlw[3] = {Tl[3], Tl[7], Tl[11]};
        float twc[3], tlc[3];
        gemm3(Tc, true, -1.0, tcw, nullptr, twc);      // twc = -Rcw.t() * tcw
        gemm3(Tl, false, 1.0, twc, tlw, tlc);          // tlc = Rlw * twc + tlw
        P.mode[pair] = tlc[2] > P.mb ? 1 : (-tlc[2] > P.mb ? 2 : 0);
    }
}

// ---- shared pieces of the candidate test ---------------------------------------------------------------------------------
struct Query {
    float u, v, invzc, radius;
    int minLevel, maxLevel;
    int minCX, maxCX, minCY, maxCY;   // window in grid cells (empty when minCX > maxCX)
};

__device__ __forceinline__ bool project_query(const MatchDev &P, int pair, int q, int mode, Query &Q) {
    const float *Tc = P.TcwCur + (size_t)pair * 12;
    const float tcw[3] = {Tc[3], Tc[7], Tc[11]};
    const float *xw = P.lastXyz + ((size_t)pair * P.cap + q) * 3;
    const float x3Dw[3] = {xw[0], xw[1], xw[2]};
    float x3Dc[3];
    gemm3(Tc, false, 1.0, x3Dw, tcw, x3Dc);            // x3Dc = Rcw * x3Dw + tcw (:577)
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (invzc < 0) return false;
    const float u = P.prm.fx * xc * invzc + P.prm.cx;
    const float v = P.prm.fy * yc * invzc + P.prm.cy;
    if (!(u >= P.prm.minX && u <= P.prm.maxX)) return false;   // NaN: GetFeaturesInArea would find no feature
    if (!(v >= P.prm.minY && v <= P.prm.maxY)) return false;
    const int nLastOctave = P.lastOctave[(size_t)pair * P.cap + q];
    if (nLastOctave < 0 || nLastOctave >= P.prm.nlevels) return false;   // not an octave of this pyramid (the reference would index out of bounds): no candidates
    const float radius = P.prm.th * P.prm.scale_factors[nLastOctave];
    Q.u = u; Q.v = v; Q.invzc = invzc; Q.radius = radius;
    if (mode == 1) { Q.minLevel = nLastOctave; Q.maxLevel = -1; }
    else if (mode == 2) { Q.minLevel = 0; Q.maxLevel = nLastOctave; }
    else { Q.minLevel = nLastOctave - 1; Q.maxLevel = nLastOctave + 1; }
    // Frame::GetFeaturesInArea window (src/Frame.cc:337-351); float -> int conversions clamped so that they stay defined
    const float fx0 = floorf((u - P.prm.minX - radius) * P.gridWInv), fx1 = ceilf((u - P.prm.minX + radius) * P.gridWInv);
    const float fy0 = floorf((v - P.prm.minY - radius) * P.gridHInv), fy1 = ceilf((v - P.prm.minY + radius) * P.gridHInv);
    const int nMinCellX = max(0, (int)fminf(fmaxf(fx0, -1.0e6f), 1.0e6f));
    const int nMaxCellX = min(GRID_COLS - 1, (int)fminf(fmaxf(fx1, -1.0e6f), 1.0e6f));
    const int nMinCellY = max(0, (int)fminf(fmaxf(fy0, -1.0e6f), 1.0e6f));
    const int nMaxCellY = min(GRID_ROWS - 1, (int)fminf(fmaxf(fy1, -1.0e6f), 1.0e6f));
    if (nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0) return false;
    Q.minCX = nMinCellX; Q.maxCX = nMaxCellX; Q.minCY = nMinCellY; Q.maxCY = nMaxCellY;
    return nMinCellX <= nMaxCellX && nMinCellY <= nMaxCellY;
}

// filters of GetFeaturesInArea (:353-376) + the mvuRight test (:625-630) for item position p; returns the Hamming distance or -1
__device__ __forceinline__ int eval_item(const MatchDev &P, int pair, const Query &Q, unsigned i2, const uint4 &d0, const uint4 &d1) {
    const size_t base = (size_t)pair * P.cap + i2;
    const int octave = P.curKps[base].octave;
    const bool bCheckLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
    if (bCheckLevels) {
        if (octave < Q.minLevel) return -1;
        if (Q.maxLevel >= 0 && octave > Q.maxLevel) return -1;
    }
    const float2 pt = *reinterpret_cast<const float2 *>(P.curUn + 2 * base);
    const float distx = pt.x - Q.u, disty = pt.y - Q.v;
    if (!(fabsf(distx) < Q.radius && fabsf(disty) < Q.radius)) return -1;
    const float uRight = P.curUright[base];
    if (uRight > 0) {
        const float ur = Q.u - P.prm.bf * Q.invzc;
        const float er = fabsf(ur - uRight);
        if (er > Q.radius) return -1;
    }
    const uint4 *dp = reinterpret_cast<const uint4 *>(P.curDesc + base * 32);
    const uint4 e0 = dp[0], e1 = dp[1];
    return __popc(d0.x ^ e0.x) + __popc(d0.y ^ e0.y) + __popc(d0.z ^ e0.z) + __popc(d0.w ^ e0.w) + __popc(d1.x ^ e1.x) +
           __popc(d1.y ^ e1.y) + __popc(d1.z ^ e1.z) + __popc(d1.w ^ e1.w);
}

// ---- k_match_candidates: one wave per last-frame point ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_candidates(MatchDev P) {
    __shared__ unsigned s_cnt[4];
    const int pair = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wv;
    if (q >= min(P.nLast[pair], P.cap)) return;
    const size_t qi = (size_t)pair * P.cap + q;
    if (lane == 0) s_cnt[wv] = 0;
    __builtin_amdgcn_wave_barrier();
    Query Q;
    const bool live = (P.lastFlags[qi] & 1) && project_query(P, pair, q, P.mode[pair], Q);
    if (!live) {
        if (lane == 0) P.candCnt[qi] = 0;
        return;
    }
    const uint4 *dq = reinterpret_cast<const uint4 *>(P.lastDesc + qi * 32);
    const uint4 d0 = dq[0], d1 = dq[1];
    const unsigned *cellStart = P.cellStart + (size_t)pair * (NCELLS + 1);
    const unsigned short *items = P.items + (size_t)pair * P.cap;
    unsigned *cand = P.cand + qi * CMAX;
    const int ny = Q.maxCY - Q.minCY + 1, C = (Q.maxCX - Q.minCX + 1) * ny;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        if (c >= C) continue;
        const int cellId = (Q.minCX + c / ny) * GRID_ROWS + Q.minCY + c % ny;
        const unsigned b = cellStart[cellId], e = cellStart[cellId + 1];
        for (unsigned p = b; p < e; p++) {
            const int dist = eval_item(P, pair, Q, items[p], d0, d1);
            if (dist < 0) continue;
            const unsigned slot = atomicAdd(&s_cnt[wv], 1u);
            if (slot < CMAX) cand[slot] = ((unsigned)dist << 16) | p;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) P.candCnt[qi] = s_cnt[wv];
}

// ---- k_match_assign: one workgroup per pair ------------------------------------------------------------------------------------
constexpr int ASSIGN_NT = 1024;

__global__ __launch_bounds__(ASSIGN_NT) void k_match_assign(MatchDev P) {
    extern __shared__ unsigned s_dyn[];     // t[cap] | holder[cap] | pick[cap]
    __shared__ int s_hist[HISTO_LENGTH], s_keep[3], s_nm;
    const int pair = blockIdx.x;
    const int nLast = min(P.nLast[pair], P.cap), nCur = min(P.nCur[pair], P.cap), mode = P.mode[pair];
    unsigned *s_t = s_dyn;
    int *s_holder = reinterpret_cast<int *>(s_dyn + P.cap);
    int *s_pick = reinterpret_cast<int *>(s_dyn + 2 * P.cap);   // current-frame keypoint picked by each point, -1 = none
    const unsigned short *items = P.items + (size_t)pair * P.cap;
    for (int i = threadIdx.x; i < P.cap; i += ASSIGN_NT) { s_t[i] = T_NONE; s_holder[i] = -1; }
    if (threadIdx.x < HISTO_LENGTH) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nm = 0;
    __syncthreads();

    // best unblocked candidate of point q: minimum of (dist << 16 | item position); 0xFFFFFFFF = none
    auto best_of = [&](int q) -> unsigned {
        const size_t qi = (size_t)pair * P.cap + q;
        const unsigned cnt = P.candCnt[qi];
        unsigned best = T_NONE;
        if (cnt == 0) return best;
        if (cnt <= CMAX) {
            const unsigned *cand = P.cand + qi * CMAX;
            for (unsigned k = 0; k < cnt; k++) {
                const unsigned key = cand[k];
                if (s_t[items[key & 0xFFFFu]] < (unsigned)q) continue;   // held by an earlier point with observations (:621-623)
                best = min(best, key);
            }
            return best;
        }
        // more candidates than stored: walk the window again (rare; any count stays exact)
        Query Q;
        if (!project_query(P, pair, q, mode, Q)) return best;
        const uint4 *dq = reinterpret_cast<const uint4 *>(P.lastDesc + qi * 32);
        const uint4 d0 = dq[0], d1 = dq[1];
        const unsigned *cellStart = P.cellStart + (size_t)pair * (NCELLS + 1);
        for (int ix = Q.minCX; ix <= Q.maxCX; ix++)
            for (int iy = Q.minCY; iy <= Q.maxCY; iy++) {
                const unsigned b = cellStart[ix * GRID_ROWS + iy], e = cellStart[ix * GRID_ROWS + iy + 1];
                for (unsigned p = b; p < e; p++) {
                    const unsigned i2 = items[p];
                    if (s_t[i2] < (unsigned)q) continue;
                    const int dist = eval_item(P, pair, Q, i2, d0, d1);
                    if (dist >= 0) best = min(best, ((unsigned)dist << 16) | p);
                }
            }
        return best;
    };

    for (int q = threadIdx.x; q < nLast; q += ASSIGN_NT) s_pick[q] = -2;      // -2: not evaluated yet (forces a first round)
    for (int round = 0; round <= nLast; round++) {
        bool changed = false;
        for (int q = threadIdx.x; q < nLast; q += ASSIGN_NT) {
            const unsigned key = best_of(q);
            const int np = (key != T_NONE && (int)(key >> 16) <= TH_HIGH) ? (int)items[key & 0xFFFFu] : -1;   // bestDist <= TH_HIGH (:637)
            changed |= np != s_pick[q];
            s_pick[q] = np;
        }
        if (!__syncthreads_or(changed ? 1 : 0)) break;
        for (int i = threadIdx.x; i < nCur; i += ASSIGN_NT) s_t[i] = T_NONE;
        __syncthreads();
        for (int q = threadIdx.x; q < nLast; q += ASSIGN_NT)
            if (s_pick[q] >= 0 && (P.lastFlags[(size_t)pair * P.cap + q] & 2)) atomicMin(&s_t[s_pick[q]], (unsigned)q);
        __syncthreads();
    }
    // holder = the last point that picked the keypoint (:638 overwrites); nmatches counts every assignment (:639).
    // t(.) is no longer needed: its storage now holds each point's rotation bin (-1 = no match).
    __syncthreads();
    int *s_bin = reinterpret_cast<int *>(s_t);
    for (int q = threadIdx.x; q < nLast; q += ASSIGN_NT) {
        int b = -1;
        const int pk = s_pick[q];
        if (pk >= 0) {
            atomicMax(&s_holder[pk], q);
            atomicAdd(&s_nm, 1);
            if (P.prm.check_orientation) {
                float rot = P.lastAngle[(size_t)pair * P.cap + q] - P.curKps[(size_t)pair * P.cap + pk].angle;   // :643-649
                if (rot < 0.0) rot += 360.0f;
                b = (int)roundf(rot * (1.0f / HISTO_LENGTH));
                if (b == HISTO_LENGTH) b = 0;
                if (b >= 0 && b < HISTO_LENGTH) atomicAdd(&s_hist[b], 1); else b = -1;
            }
        }
        s_bin[q] = b;
    }
    __syncthreads();
    if (P.prm.check_orientation) {
        if (threadIdx.x == 0) {   // ComputeThreeMaxima (:799-830)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nLast; q += ASSIGN_NT) {
            const int b = s_bin[q];
            if (b >= 0 && b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {   // :664-672
                s_holder[s_pick[q]] = -1;
                atomicSub(&s_nm, 1);
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < P.cap; i += ASSIGN_NT) P.matchOut[(size_t)pair * P.cap + i] = i < nCur ? s_holder[i] : -1;
    if (threadIdx.x == 0) P.nmatches[pair] = s_nm;
}

__global__ void k_descriptor_distance(const uint8_t *a, const uint8_t *b, int n, int32_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 *pa = reinterpret_cast<const uint4 *>(a + (size_t)i * 32), *pb = reinterpret_cast<const uint4 *>(b + (size_t)i * 32);
    const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    out[i] = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
             __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// grow-only device buffers owned by a handle (a per-frame caller pays no hipMalloc)
struct Buf { void *p = nullptr; size_t cap = 0; };

hipError_t grow(Buf &b, size_t need) {
    if (need <= b.cap) return hipSuccess;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    const hipError_t e = hipMalloc(&b.p, need);
    if (e == hipSuccess) b.cap = need;
    return e;
}

}  // namespace

// One matcher object = one ORBmatcher of the reference (src/ORBmatcher.cc:41): its own stream and its own scratch, used by one thread at a time;
// the device is re-bound at every entry like the other handles.
struct msl_match {
    int device = 0;
    hipStream_t stream = nullptr; bool ownStream = true;
    Buf in[14], items, cellStart, mode, cand, candCnt, outMatch, outN;   // staged inputs (host-memory calls), scratch, staged outputs
    Buf da, db, dout;                                                     // msl_match_descriptor_distance
    bool attrSet = false;
};

namespace {

msl_match *g_default[16];      // the device-indexed convenience entry points share one lazily created handle per device
std::mutex g_mutex;

#define M_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("msl_match: %s failed: %s", #expr, hipGetErrorString(e_)); return MSL_ERR_HIP; } } while (0)

void free_handle(msl_match *h) {
    Buf *all[] = {&h->items, &h->cellStart, &h->mode, &h->cand, &h->candCnt, &h->outMatch, &h->outN, &h->da, &h->db, &h->dout};
    for (Buf *b : all) if (b->p) (void)hipFree(b->p);
    for (Buf &b : h->in) if (b.p) (void)hipFree(b.p);
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

msl_match *default_handle(int device) {   // g_mutex held
    msl_match *&h = g_default[device & 15];
    if (!h) h = msl_match_create(device);
    return h;
}

int run_projection(msl_match *h, int n_pairs, int cap, const msl_match_params *params, const msl_keypoint *cur_kps, const float *cur_un_xy,
                   const float *cur_uright, const int32_t *cur_grid_cell, const uint8_t *cur_desc, const int32_t *n_cur, const float *last_xyz,
                   const uint8_t *last_desc, const uint8_t *last_flags, const int32_t *last_octave, const float *last_angle, const int32_t *n_last,
                   const float *Tcw_cur, const float *Tcw_last, msl_mem mem, int32_t *match_out, int32_t *nmatches, msl_mem out_mem) {
    if (!h || n_pairs < 1 || cap < 1 || cap > MAX_CAP || !params || !cur_kps || !cur_un_xy || !cur_uright || !cur_grid_cell || !cur_desc || !n_cur ||
        !last_xyz || !last_desc || !last_flags || !last_octave || !last_angle || !n_last || !Tcw_cur || !Tcw_last || !match_out || !nmatches ||
        params->nlevels < 1 || params->nlevels > MSL_MATCH_MAX_LEVELS || !(params->maxX > params->minX) || !(params->maxY > params->minY) ||
        params->fx == 0) {
        set_error("msl_match_by_projection: invalid argument (cap <= %d, nlevels <= %d)", MAX_CAP, MSL_MATCH_MAX_LEVELS);
        return MSL_ERR_INVALID;
    }
    int rc = bind_device(h->device);
    if (rc != MSL_OK) return rc;
    hipStream_t st = h->stream;
    const size_t n = (size_t)n_pairs * cap;
    MatchDev P{};
    P.nPairs = n_pairs; P.cap = cap; P.prm = *params;
    P.gridWInv = static_cast<float>(GRID_COLS) / static_cast<float>(params->maxX - params->minX);   // src/Frame.cc:137-138
    P.gridHInv = static_cast<float>(GRID_ROWS) / static_cast<float>(params->maxY - params->minY);
    P.mb = params->bf / params->fx;                                                                  // :150
    const void *src[14] = {cur_kps, cur_un_xy, cur_uright, cur_grid_cell, cur_desc, n_cur, last_xyz, last_desc, last_flags, last_octave,
                           last_angle, n_last, Tcw_cur, Tcw_last};
    const size_t bytes[14] = {sizeof(msl_keypoint) * n, 8 * n, 4 * n, 4 * n, 32 * n, 4 * (size_t)n_pairs, 12 * n, 32 * n, n, 4 * n, 4 * n,
                              4 * (size_t)n_pairs, 48 * (size_t)n_pairs, 48 * (size_t)n_pairs};
    const void *dev[14];
    for (int i = 0; i < 14; i++) {
        if (mem == MSL_MEM_HOST) {
            if (bytes[i] > h->in[i].cap) M_TRY(hipStreamSynchronize(st));   // an earlier asynchronous call may still read the buffer about to be replaced
            M_TRY(grow(h->in[i], bytes[i]));
            M_TRY(hipMemcpyAsync(h->in[i].p, src[i], bytes[i], hipMemcpyHostToDevice, st));
            dev[i] = h->in[i].p;
        } else {
            dev[i] = src[i];
        }
    }
    P.curKps = (const msl_keypoint *)dev[0]; P.curUn = (const float *)dev[1]; P.curUright = (const float *)dev[2]; P.curCell = (const int32_t *)dev[3];
    P.curDesc = (const uint8_t *)dev[4]; P.nCur = (const int32_t *)dev[5]; P.lastXyz = (const float *)dev[6]; P.lastDesc = (const uint8_t *)dev[7];
    P.lastFlags = (const uint8_t *)dev[8]; P.lastOctave = (const int32_t *)dev[9]; P.lastAngle = (const float *)dev[10];
    P.nLast = (const int32_t *)dev[11]; P.TcwCur = (const float *)dev[12]; P.TcwLast = (const float *)dev[13];
    const size_t need[5] = {sizeof(unsigned short) * n, sizeof(unsigned) * (NCELLS + 1) * n_pairs, sizeof(int) * n_pairs, sizeof(unsigned) * CMAX * n, sizeof(unsigned) * n};
    Buf *scr[5] = {&h->items, &h->cellStart, &h->mode, &h->cand, &h->candCnt};
    for (int i = 0; i < 5; i++) {
        if (need[i] > scr[i]->cap) M_TRY(hipStreamSynchronize(st));
        M_TRY(grow(*scr[i], need[i]));
    }
    P.items = (unsigned short *)h->items.p; P.cellStart = (unsigned *)h->cellStart.p; P.mode = (int *)h->mode.p; P.cand = (unsigned *)h->cand.p;
    P.candCnt = (unsigned *)h->candCnt.p;
    if (out_mem == MSL_MEM_HOST) {
        M_TRY(grow(h->outMatch, sizeof(int32_t) * n)); M_TRY(grow(h->outN, sizeof(int32_t) * n_pairs));   // (host-output calls end with a sync: nothing in flight reads these)
        P.matchOut = (int32_t *)h->outMatch.p; P.nmatches = (int32_t *)h->outN.p;
    } else {
        P.matchOut = match_out; P.nmatches = nmatches;
    }
    if (!h->attrSet) {
        M_TRY(hipFuncSetAttribute((const void *)k_match_assign, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * sizeof(unsigned) * MAX_CAP)));
        h->attrSet = true;
    }
    hipLaunchKernelGGL(k_match_grid, dim3((unsigned)n_pairs), dim3(256), sizeof(unsigned short) * cap, st, P);
    hipLaunchKernelGGL(k_match_candidates, dim3((unsigned)((cap + 3) / 4), (unsigned)n_pairs), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_match_assign, dim3((unsigned)n_pairs), dim3(ASSIGN_NT), 3 * sizeof(unsigned) * cap, st, P);
    M_TRY(hipGetLastError());
    if (out_mem == MSL_MEM_HOST) {
        M_TRY(hipMemcpyAsync(match_out, P.matchOut, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        M_TRY(hipMemcpyAsync(nmatches, P.nmatches, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToHost, st));
    }
    if (out_mem == MSL_MEM_HOST || mem == MSL_MEM_HOST) M_TRY(hipStreamSynchronize(st));   // host buffers are the caller's again on return
    return MSL_OK;
}

int run_distance(msl_match *h, const uint8_t *a32, const uint8_t *b32, int n, int32_t *dist_out) {
    if (!h || n < 0 || (n && (!a32 || !b32 || !dist_out))) { set_error("msl_match_descriptor_distance: invalid argument"); return MSL_ERR_INVALID; }
    if (n == 0) return MSL_OK;
    int rc = bind_device(h->device);
    if (rc != MSL_OK) return rc;
    hipStream_t st = h->stream;
    M_TRY(grow(h->da, (size_t)n * 32)); M_TRY(grow(h->db, (size_t)n * 32)); M_TRY(grow(h->dout, sizeof(int32_t) * n));   // synchronous call: nothing in flight
    M_TRY(hipMemcpyAsync(h->da.p, a32, (size_t)n * 32, hipMemcpyHostToDevice, st)); M_TRY(hipMemcpyAsync(h->db.p, b32, (size_t)n * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_descriptor_distance, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)h->da.p, (const uint8_t *)h->db.p, n, (int32_t *)h->dout.p);
    M_TRY(hipGetLastError());
    M_TRY(hipMemcpyAsync(dist_out, h->dout.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    M_TRY(hipStreamSynchronize(st));
    return MSL_OK;
}

}  // namespace

extern "C" {

msl_match *msl_match_create(int device) noexcept {
    try {
    if (bind_device(device) != MSL_OK) return nullptr;
    msl_match *h = new (std::nothrow) msl_match();
    if (!h) { set_error("msl_match_create: out of memory"); return nullptr; }
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { set_error("msl_match_create: hipStreamCreate failed"); delete h; return nullptr; }
    return h;
    } MSL_ABI_CATCH_PTR
}

void msl_match_destroy(msl_match *h) noexcept {
    try {
    if (!h) return;
    (void)bind_device(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_handle(h);
    } MSL_ABI_CATCH_VOID
}

int msl_match_set_stream(msl_match *h, void *hip_stream) noexcept {
    try {
    if (!h) { set_error("msl_match_set_stream: null handle"); return MSL_ERR_INVALID; }
    int rc = bind_device(h->device);
    if (rc != MSL_OK) return rc;
    M_TRY(hipStreamSynchronize(h->stream));
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream; h->ownStream = false;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_match_sync(msl_match *h) noexcept {
    try {
    if (!h) { set_error("msl_match_sync: null handle"); return MSL_ERR_INVALID; }
    int rc = bind_device(h->device);
    if (rc != MSL_OK) return rc;
    M_TRY(hipStreamSynchronize(h->stream));
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_match_by_projection(msl_match *h, int n_pairs, int cap, const msl_match_params *params, const msl_keypoint *cur_kps,
                            const float *cur_un_xy, const float *cur_uright, const int32_t *cur_grid_cell, const uint8_t *cur_desc,
                            const int32_t *n_cur, const float *last_xyz, const uint8_t *last_desc, const uint8_t *last_flags,
                            const int32_t *last_octave, const float *last_angle, const int32_t *n_last, const float *Tcw_cur,
                            const float *Tcw_last, msl_mem mem, int32_t *match_out, int32_t *nmatches, msl_mem out_mem) noexcept {
    try {
    return run_projection(h, n_pairs, cap, params, cur_kps, cur_un_xy, cur_uright, cur_grid_cell, cur_desc, n_cur, last_xyz, last_desc, last_flags,
                          last_octave, last_angle, n_last, Tcw_cur, Tcw_last, mem, match_out, nmatches, out_mem);
    } MSL_ABI_CATCH_INT
}

int msl_match_descriptor_distances(msl_match *h, const uint8_t *a32, const uint8_t *b32, int n, int32_t *dist_out) noexcept { try { return run_distance(h, a32, b32, n, dist_out); } MSL_ABI_CATCH_INT }

// Device-indexed convenience forms: one lazily created handle per device, serialised by a mutex, always synchronous.
int msl_match_by_projection_batch(int device, int n_pairs, int cap, const msl_match_params *params, const msl_keypoint *cur_kps,
                                  const float *cur_un_xy, const float *cur_uright, const int32_t *cur_grid_cell, const uint8_t *cur_desc,
                                  const int32_t *n_cur, const float *last_xyz, const uint8_t *last_desc, const uint8_t *last_flags,
                                  const int32_t *last_octave, const float *last_angle, const int32_t *n_last, const float *Tcw_cur,
                                  const float *Tcw_last, msl_mem mem, int32_t *match_out, int32_t *nmatches, msl_mem out_mem) noexcept {
    try {
    std::lock_guard<std::mutex> lock(g_mutex);
    msl_match *h = default_handle(device);
    if (!h) return MSL_ERR_NO_DEVICE;
    // device-resident inputs of this form are complete, or enqueued on the legacy default stream, when the call is made (as before the handle existed)
    if (mem == MSL_MEM_DEVICE) { if (bind_device(device) == MSL_OK) (void)hipStreamSynchronize(0); }
    int rc = run_projection(h, n_pairs, cap, params, cur_kps, cur_un_xy, cur_uright, cur_grid_cell, cur_desc, n_cur, last_xyz, last_desc, last_flags,
                            last_octave, last_angle, n_last, Tcw_cur, Tcw_last, mem, match_out, nmatches, out_mem);
    if (rc == MSL_OK) rc = msl_match_sync(h);
    return rc;
    } MSL_ABI_CATCH_INT
}

int msl_match_descriptor_distance(int device, const uint8_t *a32, const uint8_t *b32, int n, int32_t *dist_out) noexcept {
    try {
    std::lock_guard<std::mutex> lock(g_mutex);
    msl_match *h = default_handle(device);
    if (!h) return MSL_ERR_NO_DEVICE;
    return run_distance(h, a32, b32, n, dist_out);
    } MSL_ABI_CATCH_INT
}
#undef M_TRY

}  // extern "C"
