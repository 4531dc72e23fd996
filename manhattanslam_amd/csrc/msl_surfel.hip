// msl_surfel.hip -- surfel fusion for gfx950 (MI355X): the handle, streams and batching, and the C ABI.
//
// Replaces SurfelFusion (reference src/SurfelFusion.cpp) and the slot refill / tail compaction of
// SurfelMapping::fuseMap (src/SurfelMapping.cpp:353-392).  The kernels live in two other translation units:
//   msl_sf_superpixel.hip   generateSuperPixels() of a keyframe depends only on that keyframe's images, never on the map, so it is
//                           FRAME-BATCHED on the "pre" stream (kb_* kernels, one launch sequence per batch of F keyframes);
//   msl_sf_map.hip          the map stage (fusion -> new surfels -> compaction) is sequential per keyframe, on the "map" stream:
//                           k_fuse + k_compact per keyframe (classic), or ONE k_fuse launch per keyframe with the compactions of a
//                           window of <= 32 keyframes replayed at its end (deferred); run_batch picks per batch: deferred for a handle on ONE
//                           caller-provided stream under low churn, classic otherwise (a handle with its own two streams: always classic);
// the two stages overlap across batches (double-buffered slot sets).

#include "msl_sf.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <new>
#include <vector>

using namespace msl;
using namespace msl::sf;

namespace {
const char *kSfNames[MSL_SF_NKERNELS] = {"kb_seed_init", "kb_assign", "kb_prop", "kb_commit_px", "kb_update_seeds", "kb_commit_seeds",
                                         "kb_seed_plane", "k_fuse", "k_empty", "k_compact", "k_convert", "copy"};
}  // namespace

struct msl_sf {
    int device = 0;
    SfDev dev{};
    int maxBatch = 1;              // keyframes per batch; slots = 2 * maxBatch (double-buffered sets)
    hipStream_t preStream = nullptr, mapStream = nullptr; bool ownStreams = true;
    size_t blkStride = 0;          // entries per per-sub-block count slice (blockSums: one slice; blockUpd: DEFER_WIN slices, one per keyframe of a window)
    unsigned long long kfClassic = 0, kfDeferred = 0;   // keyframes that went through the classic pair of launches / through deferred windows (msl_sf_debug_scratch, which = 5)
    int dealG = 0;                 // the k_fuse grid SfDev::deal currently is a permutation for (0: none yet) -- screen-position dealing, msl_sf_map.hip
    hipStream_t copyStream = nullptr;   // host-image mode: the H2D copies of slot set i + 1 run beside the superpixel kernels of set i
    hipEvent_t evH2D[2] = {nullptr, nullptr};
    hipEvent_t evPre[2] = {nullptr, nullptr}, evMap[2] = {nullptr, nullptr}, evCopy[2] = {nullptr, nullptr};
    bool evMapValid[2] = {false, false}, evCopyValid[2] = {false, false}, evPreValid[2] = {false, false};
    unsigned long long batchNo = 0;
    int stagedSet = -1; size_t stagedGs = 0;   // slot set / row stride of the gray images the last host-image batch staged (msl_sf_staged_gray); -1: none
    int lastSlot = 0;
    // per-slot device buffers
    FrameDev *d_frames = nullptr; FrameDev *h_frames = nullptr;  // pinned host staging [slots]
    msl_seed *d_seeds = nullptr, *d_seedsTmp = nullptr; msl_surfel *d_cand = nullptr; uint8_t *d_candOk = nullptr, *d_fused = nullptr; uint2 *d_tex = nullptr; float4 *d_fuseRec = nullptr;
    unsigned short *d_index = nullptr, *d_amap = nullptr; unsigned *d_tmin = nullptr; int *d_chunkAbort = nullptr, *d_changed = nullptr;
    AssignRec *d_arec = nullptr; unsigned *d_wl = nullptr, *d_wlCount = nullptr;
    float *d_pxInv = nullptr;
    // staged images (host input mode), per slot
    uint8_t *d_gray = nullptr; float *d_depth = nullptr; int32_t *d_member = nullptr;
    size_t grayCap = 0, depthCap = 0, memberCap = 0;  // bytes per slot
    uint8_t *d_depth16 = nullptr; size_t depth16Cap = 0;   // raw 16-bit depth of host-image calls (msl_sf_fuse_resident_batch_d16), bytes per slot
    long long *d_ctr = nullptr; long long *h_ctr = nullptr;
    unsigned *d_tickets = nullptr, *d_delU = nullptr;
    DeferCtl *d_dc = nullptr;
    float *d_projTab = nullptr;
    bool propLds = false;        // t(s) of one keyframe fits the LDS: single-launch relaxation
    bool classicNext = true;     // the map was replaced from outside the keyframe chain (upload / restore / append / detach): its first keyframe takes the classic
                                 // pair of launches, whose compaction handles any number of stale or deleted slots at full speed
    msl_surfel *d_new = nullptr;
    float *d_mapStore = nullptr; size_t mapCap = 0;
    unsigned *d_rpStore = nullptr;   // deferred compaction: move lists, dense replay tables, staging (see set_map_ptrs)
    size_t liveBound = 0;        // host-side upper bound of the live count: last known count + nseeds per keyframe enqueued since
    size_t liveKnown = 0;        // the most recent live count the host has seen (exact at that time; only a hint for k_fuse's speculative loads)
    unsigned long long liveKnownKf = 0;   // ... and the number of keyframes that had been enqueued when it was exact: an older snapshot never replaces a newer one
    // asynchronous refresh of that bound: after every batch the live count is copied to pinned memory behind an event; a later call picks
    // up whatever has arrived, so the bound follows the real count a couple of batches late instead of forcing a pipeline drain
    // every capacity / nseeds keyframes
    static constexpr int NSNAP = 4;
    static constexpr int SNAPW = 16;   // counters per snapshot: ctr[0 .. 15]
    long long *h_snap = nullptr; hipEvent_t snapEv[NSNAP] = {}; unsigned long long snapKf[NSNAP] = {}; bool snapBusy[NSNAP] = {};
    bool snapLive[NSNAP] = {};   // the snapshot's live count still describes the resident map (no upload / restore since it was taken)
    unsigned long long kfEnq = 0; int snapNext = 0;
    // churn = surfels spawned + deleted per keyframe over the most recent batch the host has seen (from the running totals of two snapshots):
    // the deferred compaction is built for the steady state (a replay by ONE wave per window); under heavy churn the classic chain, whose
    // compaction works with all its workgroups, is faster
    long long churnNew = -1, churnDel = 0, churnKf = 0; unsigned long long churnAt = 0; double churn = 0.0;
    unsigned *d_blockSums = nullptr, *d_blockUpd = nullptr, *d_delList = nullptr, *d_srcOf = nullptr;
    msl_surfel *d_aos = nullptr; size_t aosCap = 0;
    float *d_snapStore = nullptr; size_t snapCap = 0, snapN = 0; bool snapValid = false; long long snapWide = 0;   // msl_sf_map_snapshot / _restore
    // host-vector mode (msl_sf_fuse_ex): the device map equals the caller's vector as the last call left it
    bool mirrorValid = false; size_t mirrorN = 0;
    unsigned *h_blk = nullptr; size_t blkCap = 0;   // pinned: per-sub-block deleted / updated counts of the call's k_fuse launch
    uint8_t *h_list = nullptr; size_t listCap = 0;  // pinned: {count | indices | records} of the sparse download
    KernelProfiler prof;
};

namespace {

// Map storage, in 4-byte words per surfel of capacity c (c is a multiple of 4096, so every array starts 32-byte aligned):
//   d_mapStore: hot 4 | cold 8 | rgbWide 3 | utlWide 2                                                               = 17 c
//   d_rpStore : moveDst 1 | loc64 2 | vposD 1 | locKeys 1 | vposKeys 1 | dBig 1 | stageHot 4 | stageCold 8 | stageRgb 3 | stageUtl 2 | bitmap c / 32 + 64
constexpr size_t MAP_WORDS = 17, RP_WORDS = 24;
void set_map_ptrs(msl_sf *h) {
    const size_t c = h->mapCap;
    MapSoA &M = h->dev.map;
    M.hot = reinterpret_cast<HotPk *>(h->d_mapStore);
    M.cold = reinterpret_cast<ColdRec *>(h->d_mapStore + 4 * c);
    M.rgbWide = reinterpret_cast<int *>(h->d_mapStore + 12 * c);        // exact ints of the COLD_WIDE records (untouched otherwise)
    M.utlWide = reinterpret_cast<int *>(h->d_mapStore + 15 * c);        // exact ints of the HOT_WIDE records
    M.wideFlag = h->d_ctr + 13;
    SfDev &D = h->dev;
    D.cap = c;
    D.blockSums = h->d_blockSums; D.blockUpd = h->d_blockUpd; D.delList = h->d_delList; D.srcOf = h->d_srcOf;
    D.sbKeys = h->d_blockSums + h->blkStride; D.deal = h->d_blockSums + 2 * h->blkStride; D.dealG = 0;   // (three planes of one allocation: counts, screen keys, dealing table)
    unsigned *r = h->d_rpStore;
    D.loc64 = reinterpret_cast<unsigned long long *>(r);            // (first: 8-byte aligned)
    D.stageCold = reinterpret_cast<ColdRec *>(r + 2 * c);
    D.stageHot = reinterpret_cast<HotPk *>(r + 10 * c);
    D.moveDst = r + 14 * c; D.vposD = r + 15 * c; D.locKeys = r + 16 * c; D.vposKeys = r + 17 * c; D.dBig = r + 18 * c;
    D.stageRgb = reinterpret_cast<int *>(r + 19 * c); D.stageUtl = reinterpret_cast<int *>(r + 22 * c);
    D.bitmap = r + 24 * c;
}

// The asynchronous live-count snapshots only ever LOWER liveBound; whenever the map is replaced from outside the keyframe chain
// (upload, restore) the ones still pending describe the old map and must be ignored.
void drop_live_snapshots(msl_sf *h) {
    for (int i = 0; i < msl_sf::NSNAP; i++) h->snapLive[i] = false;   // (their running totals -- the churn estimate -- stay valid)
}

int sync_all(msl_sf *h) {
    if (h->ownStreams && h->copyStream) MSL_HIP_TRY(hipStreamSynchronize(h->copyStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->preStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    return MSL_OK;
}

// The device-side control block of the map stage: bases of the slot arrays and of the map's side arrays / lists (what only a few waves of a
// k_fuse launch read).  Rewritten whenever one of them is reallocated; the streams are idle then, and the window state it also holds
// (extents, deletion counts) is all zero between windows.
int write_ctl(msl_sf *h) {
    DeferCtl dc;
    memset(&dc, 0, sizeof(dc));
    const SfDev &D = h->dev;
    dc.flagStride = D.flagStride; dc.candOk = h->d_candOk; dc.fused = h->d_fused; dc.cand = h->d_cand;
    dc.aux.map = D.map; dc.aux.cap = D.cap; dc.aux.delU = D.delU; dc.aux.delUCount = D.delUCount; dc.aux.delList = D.delList;
    MSL_HIP_TRY(hipMemcpy(h->d_dc, &dc, sizeof(dc), hipMemcpyHostToDevice));
    return MSL_OK;
}

// (Re)allocate the resident map for `cap` surfels, preserving the first `keep` entries.
int map_realloc(msl_sf *h, size_t cap, size_t keep) {
    cap = (cap + 4095) & ~(size_t)4095;
    float *nstore = nullptr; unsigned *nbs = nullptr, *nbu = nullptr, *ndl = nullptr, *nso = nullptr, *nrp = nullptr;
    const size_t bst = cap / SUB_ITEMS + 8200;   // per slice; >= 1024 / 8192 padding entries: the compaction reads its first tiles unconditionally
    auto attempt = [&]() -> int {
        MSL_HIP_TRY(hipMalloc(&nstore, sizeof(float) * MAP_WORDS * cap));
        MSL_HIP_TRY(hipMalloc(&nbs, sizeof(unsigned) * 3 * bst));           // deleted counts | screen keys | dealing table (SfDev::sbKeys, ::deal)
        MSL_HIP_TRY(hipMalloc(&nbu, sizeof(unsigned) * DEFER_WIN * bst));   // one slice per keyframe of a deferred window (classic: the first)
        MSL_HIP_TRY(hipMemset(nbs, 0, sizeof(unsigned) * bst));
        MSL_HIP_TRY(hipMemset(nbs + bst, 0xFF, sizeof(unsigned) * bst));    // no key yet: "nothing in view"
        MSL_HIP_TRY(hipMemset(nbu, 0, sizeof(unsigned) * DEFER_WIN * bst));
        MSL_HIP_TRY(hipMalloc(&ndl, sizeof(unsigned) * cap));
        MSL_HIP_TRY(hipMalloc(&nso, sizeof(unsigned) * cap));
        const size_t rpWords = RP_WORDS * cap + cap / 32 + 64;
        MSL_HIP_TRY(hipMalloc(&nrp, sizeof(unsigned) * rpWords));
        MSL_HIP_TRY(hipMemset(nrp, 0, sizeof(unsigned) * 2 * cap));                              // loc64: empty (the replay leaves it clean)
        MSL_HIP_TRY(hipMemset(nrp + 15 * cap, 0, sizeof(unsigned) * cap));                       // vposD
        MSL_HIP_TRY(hipMemset(nrp + 24 * cap, 0, sizeof(unsigned) * (cap / 32 + 64)));           // bitmap
        if (keep && h->d_mapStore) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            const size_t oc = h->mapCap;
            MSL_HIP_TRY(hipMemcpy(nstore, h->d_mapStore, sizeof(HotPk) * keep, hipMemcpyDeviceToDevice));
            MSL_HIP_TRY(hipMemcpy(nstore + 4 * cap, h->d_mapStore + 4 * oc, sizeof(ColdRec) * keep, hipMemcpyDeviceToDevice));
            MSL_HIP_TRY(hipMemcpy(nstore + 12 * cap, h->d_mapStore + 12 * oc, sizeof(int) * 3 * keep, hipMemcpyDeviceToDevice));   // (rare path: no need to know whether any record is wide)
            MSL_HIP_TRY(hipMemcpy(nstore + 15 * cap, h->d_mapStore + 15 * oc, sizeof(int) * 2 * keep, hipMemcpyDeviceToDevice));
        }
        return MSL_OK;
    };
    const int arc = attempt();
    if (arc != MSL_OK) {   // nothing of a failed attempt stays allocated; the old map is untouched
        if (nstore) (void)hipFree(nstore);
        if (nbs) (void)hipFree(nbs);
        if (nbu) (void)hipFree(nbu);
        if (ndl) (void)hipFree(ndl);
        if (nso) (void)hipFree(nso);
        if (nrp) (void)hipFree(nrp);
        return arc;
    }
    if (h->d_mapStore) {
        (void)hipFree(h->d_mapStore); (void)hipFree(h->d_blockSums); (void)hipFree(h->d_blockUpd); (void)hipFree(h->d_delList); (void)hipFree(h->d_srcOf); (void)hipFree(h->d_rpStore);
    }
    h->d_mapStore = nstore; h->d_blockSums = nbs; h->d_blockUpd = nbu; h->d_delList = ndl; h->d_srcOf = nso; h->d_rpStore = nrp; h->mapCap = cap;
    h->blkStride = bst;
    h->dealG = 0;              // (the dealing table went with the old allocation)
    set_map_ptrs(h);
    return write_ctl(h);
}

void free_slots(msl_sf *h) {
    auto F = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->d_frames); F(h->d_seeds); F(h->d_seedsTmp); F(h->d_cand); F(h->d_candOk); F(h->d_fused); F(h->d_tex); F(h->d_fuseRec); F(h->d_index); F(h->d_amap); F(h->d_tmin);
    F(h->d_chunkAbort); F(h->d_changed); F(h->d_arec); F(h->d_pxInv); F(h->d_wl); F(h->d_wlCount); F(h->d_gray); F(h->d_depth); F(h->d_member); F(h->d_depth16);
    if (h->h_frames) { (void)hipHostFree(h->h_frames); h->h_frames = nullptr; }
    h->grayCap = h->depthCap = h->memberCap = 0; h->depth16Cap = 0;
}

int alloc_slots(msl_sf *h, int maxBatch) {
    free_slots(h);
    SfDev &D = h->dev;
    const size_t slots = 2 * (size_t)maxBatch, ns = D.nseeds, npx = D.pxStride, fs = D.flagStride;
    MSL_HIP_TRY(hipMalloc(&h->d_frames, sizeof(FrameDev) * slots));
    MSL_HIP_TRY(hipHostMalloc(&h->h_frames, sizeof(FrameDev) * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_seeds, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_seedsTmp, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_cand, sizeof(msl_surfel) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_candOk, fs * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_fused, fs * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_tex, sizeof(uint2) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_fuseRec, sizeof(float4) * 3 * ns * slots));
    MSL_HIP_TRY(hipMemset(h->d_tex, 0, sizeof(uint2) * npx * slots));
    MSL_HIP_TRY(hipMemset(h->d_fuseRec, 0, sizeof(float4) * 3 * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_index, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_amap, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_tmin, sizeof(unsigned) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_arec, sizeof(AssignRec) * (ns * slots + 2)));   // + a record either side: kb_assign loads row pairs that may start one before / end one after
    MSL_HIP_TRY(hipMalloc(&h->d_pxInv, sizeof(float) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_wl, sizeof(unsigned) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_wlCount, sizeof(unsigned) * slots));
    MSL_HIP_TRY(hipMemset(h->d_wlCount, 0, sizeof(unsigned) * slots));
    MSL_HIP_TRY(hipMemset(h->d_arec, 0, sizeof(AssignRec) * (ns * slots + 2)));
    MSL_HIP_TRY(hipMalloc(&h->d_chunkAbort, sizeof(int) * 32 * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_changed, sizeof(int) * 8 * slots));
    MSL_HIP_TRY(hipMemset(h->d_seeds, 0, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMemset(h->d_index, 0, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMemset(h->d_fused, 1, fs * slots));     // (the bytes behind the lattice stay 1 = "spawns nothing": the map stage scans whole 16-byte words)
    MSL_HIP_TRY(hipMemset(h->d_candOk, 0, fs * slots));
    D.frames = h->d_frames; D.seeds = h->d_seeds; D.seedsTmp = h->d_seedsTmp; D.cand = h->d_cand; D.candOk = h->d_candOk; D.fused = h->d_fused; D.tex = h->d_tex; D.fuseRec = h->d_fuseRec;
    D.index = h->d_index; D.amap = h->d_amap; D.tmin = h->d_tmin; D.chunkAbort = h->d_chunkAbort; D.changed = h->d_changed;
    D.arec = h->d_arec + 1; D.pxInv = h->d_pxInv; D.wl = h->d_wl; D.wlCount = h->d_wlCount;
    { const int rc = write_ctl(h); if (rc != MSL_OK) return rc; }
    h->maxBatch = maxBatch;
    h->lastSlot = 0;            // the debug accessors must never index beyond the reallocated slot buffers
    h->evMapValid[0] = h->evMapValid[1] = false;
    h->evPreValid[0] = h->evPreValid[1] = false;
    h->evCopyValid[0] = h->evCopyValid[1] = false;
    return MSL_OK;
}

// Counter snapshots that have arrived (their event has fired): a tighter bound / a fresher value of the live count, and the churn estimate -- spawned +
// deleted surfels per keyframe between two snapshots, from the running totals.
void consume_snapshots(msl_sf *h, size_t nseeds) {
    for (int i = 0; i < msl_sf::NSNAP; i++)
        if (h->snapBusy[i] && hipEventQuery(h->snapEv[i]) == hipSuccess) {
            h->snapBusy[i] = false;
            const long long *sn = h->h_snap + (size_t)i * msl_sf::SNAPW;
            if (h->snapLive[i]) {
                const size_t cand = (size_t)sn[0] + (size_t)(h->kfEnq - h->snapKf[i]) * nseeds;   // count then + what was enqueued since
                if (cand < h->liveBound) h->liveBound = cand;
                if (h->snapKf[i] >= h->liveKnownKf) { h->liveKnown = (size_t)sn[0]; h->liveKnownKf = h->snapKf[i]; }   // completed snapshots are visited in array order, not age order
            }
            if (h->snapKf[i] > h->churnAt) {   // running totals: new ctr[8], deleted ctr[9], keyframes ctr[11]
                if (h->churnNew >= 0 && sn[11] > h->churnKf) h->churn = (double)((sn[8] - h->churnNew) + (sn[9] - h->churnDel)) / (double)(sn[11] - h->churnKf);
                h->churnNew = sn[8]; h->churnDel = sn[9]; h->churnKf = sn[11]; h->churnAt = h->snapKf[i];
            }
        }
}

int read_ctr(msl_sf *h) {
    if (h->ownStreams && h->copyStream) MSL_HIP_TRY(hipStreamSynchronize(h->copyStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->preStream));
    MSL_HIP_TRY(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(long long) * 16, hipMemcpyDeviceToHost, h->mapStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    h->prof.drain();
    consume_snapshots(h, (size_t)h->dev.nseeds);   // (all pending snapshots have arrived: their churn information is kept, the live count below is newer)
    h->liveBound = (size_t)h->h_ctr[0];   // both streams are idle: the count is exact
    h->liveKnown = h->liveBound; h->liveKnownKf = h->kfEnq;
    for (int i = 0; i < msl_sf::NSNAP; i++) h->snapBusy[i] = false;   // (their events have fired: the stream is idle)
    return MSL_OK;
}

int check_err(msl_sf *h) {
    if (h->h_ctr[5]) {
        const long long e = h->h_ctr[5];
        (void)hipMemsetAsync(h->d_ctr + 5, 0, sizeof(long long), h->mapStream);
        if (e == 20) set_error("resident surfel map capacity exceeded (reserve more with msl_sf_map_reserve)");
        else set_error("surfel pipeline device-side bound exceeded (code %lld)", e);
        return MSL_ERR_OVERFLOW;
    }
    return MSL_OK;
}
// Superpixel stage for slots [slot0, slot0+n) on the pre stream, then the map stage per keyframe on the map stream.
// depth16 != nullptr: the depth images are raw 16-bit values (rows d16s bytes apart, frames d16fs bytes apart) that become metres on the device,
// (float)raw * depthFactor (src/Frame.cc:96-97); `depth` / ds / dfs are ignored then.
int run_batch(msl_sf *h, int n, const int32_t *refs, const uint8_t *gray, size_t gs, size_t gfs, const float *depth, size_t ds, size_t dfs,
              const int32_t *member, size_t ms, size_t mfs, msl_mem mem, const float *poses, bool compact, const uint16_t *depth16 = nullptr,
              size_t d16s = 0, size_t d16fs = 0, float depthFactor = 1.0f) {
    SfDev &D = h->dev;
    const int W = D.W, H = D.H;
    if (n < 1 || n > h->maxBatch) { set_error("msl_sf: batch of %d keyframes exceeds the batch capacity %d", n, h->maxBatch); return MSL_ERR_INVALID; }
    const bool d16 = depth16 != nullptr;
    if (d16) {
        if (d16s < (size_t)W * 2 || (d16s & 1) || (d16fs & 1) || ((uintptr_t)depth16 & 1)) { set_error("msl_sf: bad 16-bit depth pointer or strides"); return MSL_ERR_INVALID; }
        ds = (size_t)W * 4; dfs = ds * (size_t)H;   // the converted images are tightly packed
    }
    if (!gray || (!depth && !d16) || !member || !poses || !refs || gs < (size_t)W || ds < (size_t)W * 4 || ms < (size_t)((W + 1) / 2) * 4 || (ds & 3) || (ms & 3)) {
        set_error("msl_sf: bad image pointers or strides");
        return MSL_ERR_INVALID;
    }
    if (gs * (size_t)H >= (1ull << 32) || ds * (size_t)H >= (1ull << 32) || ms * (size_t)((H + 1) / 2) >= (1ull << 32)) {   // (the kernels address an image with 32-bit byte offsets)
        set_error("msl_sf: image rows span 4 GB or more");
        return MSL_ERR_INVALID;
    }
    if (compact) {
        h->mirrorValid = false;   // the resident map moves on without the host-vector caller
        // The reference's mvLocalSurfels is an unbounded std::vector (include/Map.h:130): grow the resident map before a batch could
        // overflow it.  Every keyframe adds at most nseeds surfels, so the host only needs an upper bound of the live count; the
        // exact count is read back (one sync) only when that bound reaches the capacity.
        const size_t need = (size_t)n * (size_t)D.nseeds;
        consume_snapshots(h, (size_t)D.nseeds);
        if (h->liveBound + need > h->mapCap) {
            int rc = read_ctr(h);
            if (rc != MSL_OK) return rc;
            rc = check_err(h);
            if (rc != MSL_OK) return rc;
            if (h->liveBound + need > h->mapCap) {
                rc = map_realloc(h, 2 * h->liveBound + 2 * need + 65536, h->liveBound);
                if (rc != MSL_OK) return rc;
            }
        }
        h->liveBound += need;
        h->kfEnq += (unsigned long long)n;
    }
    const int set = (int)(h->batchNo & 1), slot0 = set * h->maxBatch;
    if (mem != MSL_MEM_HOST) h->stagedSet = -1;
    hipStream_t sp = h->preStream, sm = h->mapStream;
    if (h->evMapValid[set] && sp != sm) MSL_HIP_TRY(hipStreamWaitEvent(sp, h->evMap[set], 0));   // the set's previous user is done
    D.gstride = gs; D.gbytes = gs * (size_t)(H - 1) + W; D.dstride = ds / 4; D.mstride = ms / 4;
    D.gsB = (unsigned)gs; D.dsB = (unsigned)ds; D.msB = (unsigned)ms;
    // bytes actually present in the caller's buffers: the last row carries no stride padding
    const size_t d16b = d16 ? d16s * (size_t)(H - 1) + (size_t)W * 2 : 0;
    const size_t gb = gs * (size_t)(H - 1) + W, db = ds * (size_t)(H - 1) + (size_t)W * 4, mb = ms * (size_t)((H + 1) / 2 - 1) + (size_t)((W + 1) / 2) * 4;   // the membership image is ceil(H / 2) x ceil(W / 2) (PlaneDetection's cloud size)
    if (mem == MSL_MEM_HOST) {
        const size_t slots = 2 * (size_t)h->maxBatch;
        if (gb > h->grayCap || db > h->depthCap || mb > h->memberCap) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            if (h->d_gray) (void)hipFree(h->d_gray);
            if (h->d_depth) (void)hipFree(h->d_depth);
            if (h->d_member) (void)hipFree(h->d_member);
            h->d_gray = nullptr; h->d_depth = nullptr; h->d_member = nullptr;
            h->grayCap = h->depthCap = h->memberCap = 0;
            MSL_HIP_TRY(hipMalloc(&h->d_gray, gb * slots)); MSL_HIP_TRY(hipMalloc(&h->d_depth, db * slots)); MSL_HIP_TRY(hipMalloc(&h->d_member, mb * slots));
            h->grayCap = gb; h->depthCap = db; h->memberCap = mb;
        }
        if (d16 && d16b > h->depth16Cap) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            if (h->d_depth16) (void)hipFree(h->d_depth16);
            h->d_depth16 = nullptr; h->depth16Cap = 0;
            MSL_HIP_TRY(hipMalloc(&h->d_depth16, d16b * slots));
            h->depth16Cap = d16b;
        }
        // The images travel on their own stream so that they overlap the superpixel kernels of the previous call (the other slot set);
        // with caller-provided streams (msl_sf_set_stream) everything stays on that one stream.
        hipStream_t sc = (h->ownStreams && h->copyStream) ? h->copyStream : sp;
        // (round 6) the staged images of a set are read by the SUPERPIXEL stage only -- the map stage works on the slot arrays (texels, seed records,
        // candidates) -- so the copies of call k wait for the superpixel stage of call k - 2 (evPre), not for its map stage (evMap, which the
        // superpixel stage of call k still waits for): the link runs up to two calls ahead of the map chain instead of in step with it
        // (A/B on one box, bench.py --io host: 17.6 k -> 19.0 k frames/s with f32 depth, 18.8 k -> 20.1 k with raw 16-bit depth)
        if (sc != sp && h->evPreValid[set]) MSL_HIP_TRY(hipStreamWaitEvent(sc, h->evPre[set], 0));
        h->prof.begin(SK_COPY, sc);
        // Tightly packed frame arrays (the streaming case) travel as ONE copy per image kind instead of one per frame; a membership image
        // shared by all keyframes of the call (member_frame_stride == 0) is staged once.
        const bool packedG = n > 1 && gfs == gb && h->grayCap == gb, packedD = !d16 && n > 1 && dfs == db && h->depthCap == db;
        const bool packed16 = d16 && n > 1 && d16fs == d16b && h->depth16Cap == d16b;
        const bool packedM = n > 1 && mfs == mb && h->memberCap == mb;
        if (packedG) MSL_HIP_TRY(hipMemcpyAsync(h->d_gray + (size_t)slot0 * h->grayCap, gray, gb * (size_t)n, hipMemcpyHostToDevice, sc));
        if (packedD) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_depth + (size_t)slot0 * h->depthCap, depth, db * (size_t)n, hipMemcpyHostToDevice, sc));
        if (packed16) MSL_HIP_TRY(hipMemcpyAsync(h->d_depth16 + (size_t)slot0 * h->depth16Cap, depth16, d16b * (size_t)n, hipMemcpyHostToDevice, sc));
        if (packedM) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + (size_t)slot0 * h->memberCap, member, mb * (size_t)n, hipMemcpyHostToDevice, sc));
        if (mfs == 0) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + (size_t)slot0 * h->memberCap, member, mb, hipMemcpyHostToDevice, sc));
        for (int f = 0; f < n; f++) {
            const size_t s = slot0 + f;
            if (!packedG) MSL_HIP_TRY(hipMemcpyAsync(h->d_gray + s * h->grayCap, gray + f * gfs, gb, hipMemcpyHostToDevice, sc));
            if (!d16 && !packedD) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_depth + s * h->depthCap, (const uint8_t *)depth + f * dfs, db, hipMemcpyHostToDevice, sc));
            if (d16 && !packed16) MSL_HIP_TRY(hipMemcpyAsync(h->d_depth16 + s * h->depth16Cap, (const uint8_t *)depth16 + f * d16fs, d16b, hipMemcpyHostToDevice, sc));
            if (!packedM && mfs != 0) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + s * h->memberCap, (const uint8_t *)member + f * mfs, mb, hipMemcpyHostToDevice, sc));
        }
        if (d16)   // raw -> metres behind the copies, on their stream (same rows of stride d16s in the staging slots; frames depth16Cap bytes apart)
            sp_launch_depth_u16(sc, h->d_depth16 + (size_t)slot0 * h->depth16Cap, d16s, h->depth16Cap, (float *)((uint8_t *)h->d_depth + (size_t)slot0 * h->depthCap),
                                h->depthCap / 4, W, H, n, depthFactor);
        h->prof.end(sc);
        MSL_HIP_TRY(hipEventRecord(h->evH2D[set], sc));   // (always: msl_sf_staged_gray hands it to other handles)
        if (sc != sp) MSL_HIP_TRY(hipStreamWaitEvent(sp, h->evH2D[set], 0));
        h->stagedSet = set; h->stagedGs = gs;
    }
    if (d16 && mem != MSL_MEM_HOST) {   // device-resident raw depth: converted into the handle's float slots on the superpixel stream
        const size_t slots = 2 * (size_t)h->maxBatch;
        if (db > h->depthCap || !h->d_depth) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            if (h->d_gray) (void)hipFree(h->d_gray);
            if (h->d_depth) (void)hipFree(h->d_depth);
            if (h->d_member) (void)hipFree(h->d_member);
            h->d_gray = nullptr; h->d_depth = nullptr; h->d_member = nullptr;
            h->grayCap = h->depthCap = h->memberCap = 0;     // (a later host-image call allocates all three anew)
            MSL_HIP_TRY(hipMalloc(&h->d_depth, db * slots));
            h->depthCap = db;
        }
        sp_launch_depth_u16(sp, depth16, d16s, d16fs, (float *)((uint8_t *)h->d_depth + (size_t)slot0 * h->depthCap), h->depthCap / 4, W, H, n, depthFactor);
    }
    if (h->evCopyValid[set]) MSL_HIP_TRY(hipEventSynchronize(h->evCopy[set]));   // pinned staging of this set is free again
    for (int f = 0; f < n; f++) {
        FrameDev &F = h->h_frames[slot0 + f];
        if (mem == MSL_MEM_HOST) {
            const size_t s = slot0 + f;
            F.gray = h->d_gray + s * h->grayCap; F.depth = (const float *)((uint8_t *)h->d_depth + s * h->depthCap);
            F.member = (const int32_t *)((uint8_t *)h->d_member + (mfs == 0 ? (size_t)slot0 : s) * h->memberCap);
        } else {
            F.gray = gray + f * gfs; F.member = (const int32_t *)((const uint8_t *)member + f * mfs);
            F.depth = d16 ? (const float *)((uint8_t *)h->d_depth + (size_t)(slot0 + f) * h->depthCap) : (const float *)((const uint8_t *)depth + f * dfs);
        }
        memcpy(F.pose, poses + 16 * f, sizeof(float) * 16);
        inverse4<float>(F.pose, F.invPose);   // pose.inverse() (:59), adjugate/determinant in float
        F.ref = refs[f]; F._pad = 0;
    }
    MSL_HIP_TRY(hipMemcpyAsync(h->d_frames + slot0, h->h_frames + slot0, sizeof(FrameDev) * n, hipMemcpyHostToDevice, sp));
    MSL_HIP_TRY(hipEventRecord(h->evCopy[set], sp));
    h->evCopyValid[set] = true;

    SfDev P = D;
    // shift every per-slot base so that blockIdx.y/z == 0 addresses slot0
    P.frames = D.frames + slot0; P.seeds = D.seeds + (size_t)slot0 * D.nseeds; P.seedsTmp = D.seedsTmp + (size_t)slot0 * D.nseeds;
    P.cand = D.cand + (size_t)slot0 * D.nseeds; P.candOk = D.candOk + (size_t)slot0 * D.flagStride; P.fused = D.fused + (size_t)slot0 * D.flagStride;
    P.tex = D.tex + (size_t)slot0 * D.pxStride; P.fuseRec = D.fuseRec + (size_t)slot0 * D.nseeds * 3;
    P.index = D.index + (size_t)slot0 * D.pxStride; P.amap = D.amap + (size_t)slot0 * D.pxStride; P.tmin = D.tmin + (size_t)slot0 * D.nseeds;
    P.arec = D.arec + (size_t)slot0 * D.nseeds; P.pxInv = D.pxInv + (size_t)slot0 * D.pxStride; P.wl = D.wl + (size_t)slot0 * D.pxStride; P.wlCount = D.wlCount + slot0;
    P.chunkAbort = D.chunkAbort + slot0 * 32; P.changed = D.changed + slot0 * 8;
    sp_launch_stage(h->prof, sp, P, n, h->propLds);
    if (sp != sm) {
        MSL_HIP_TRY(hipEventRecord(h->evPre[set], sp));
        h->evPreValid[set] = true;
        MSL_HIP_TRY(hipStreamWaitEvent(sm, h->evPre[set], 0));
    }
    const size_t boundLive = compact ? h->liveBound : h->mapCap;
    // grid: the last known live count plus a margin (k_fuse is grid-stride, so a map that outgrew it is still covered), never beyond the upper
    // bound; hint: the sub-blocks that were full at the last known count load without waiting for the live count
    const size_t known = std::min(h->liveKnown, boundLive);
    // (round 6) ... rounded up to a multiple of 64 sub-blocks inside the capacity (a multiple of 32 sub-blocks): the dealing table of the launch before is
    // a permutation for ONE grid size, so the grid should change rarely -- a wave beyond the live count costs one load
    const size_t subWant = (std::min(known + 2 * (size_t)D.nseeds, boundLive) + SUB_ITEMS - 1) / SUB_ITEMS;
    const int nSubGrid = (int)std::max<size_t>(8, std::min((subWant + 63) & ~(size_t)63, (size_t)D.cap / SUB_ITEMS));
    const int nSubHint = (int)(known / SUB_ITEMS);
    static const int DEAL_EVERY = getenv("MSL_SF_DEAL_EVERY") ? std::max(1, atoi(getenv("MSL_SF_DEAL_EVERY"))) : 4;
    static const char *dealEnv = getenv("MSL_SF_DEAL");   // "0": sub-blocks in array order (rounds 1-5), for A/B measurements
    // ... and only while the map (48 bytes per surfel) fits the 256 MB Infinity Cache: a larger map is streamed from HBM, where waves that walk the array in
    // order keep DRAM pages open -- 8 M surfels: k_fuse 76.9 us in array order, 80.5 us dealt (bench.py --surfels 8000000, A/B on one box)
    constexpr int DEAL_MAX_GRID = (4 << 20) / SUB_ITEMS;
    const bool dealOn = !(dealEnv && !strcmp(dealEnv, "0")) && (nSubGrid & 7) == 0 && (size_t)nSubGrid <= h->blkStride && nSubGrid <= DEAL_MAX_GRID;
    // Map stage.  Deferred compaction (MSL_SF_DEFER=0 turns it off, =1 forces it; unset: the policy below): windows of <= DEFER_WIN keyframes, ONE
    // launch per keyframe, the window's compactions replayed at its end (msl_sf_map.hip).  Classic (k_fuse + k_compact per keyframe): single
    // keyframes, the host-vector drop-in, the first keyframe after the map was replaced from outside, and batches enqueued while the recent
    // churn (spawned + deleted surfels per keyframe, from the asynchronous counter snapshots) is high -- k_compact takes any number of stale or
    // deleted slots with all its workgroups, the replay's single wave is built for the steady state.  Both leave identical maps.
    static const char *deferEnv = getenv("MSL_SF_DEFER");   // "0": never; "1": always (the parity tests); unset: the policy below
    static const bool deferOff = deferEnv && !strcmp(deferEnv, "0"), deferForce = deferEnv && !strcmp(deferEnv, "1");
    constexpr double CHURN_MAX = 96.0;   // spawned + deleted surfels per keyframe up to which the one-wave replay beats k_compact (bench.py --map moving: 670)
    // Policy.  The deferred chain is 8 us per keyframe shorter (22.6 against 31 us alone), which pays exactly when the map chain is the critical
    // path: a handle on ONE caller-provided stream (superpixel stage and map stage back to back: 20.8 k against 19.0 k keyframes/s).  With the
    // handle's own two streams the frame-batched superpixel stage is the longer one; k_fuse launches that follow each other without the idle
    // stretch of k_compact in between only take issue slots from it (front end 22 010 against 22 330 frames/s, k_fuse 18.8 against 16.3 us in the
    // timed region), so that shape keeps the classic pair.
    const bool churny = !deferForce && (h->churn > CHURN_MAX || sp != sm);
    auto classic = [&](int f) {
        P.kf = 0;
        map_launch_fuse(h->prof, sm, P, f, h->h_frames[slot0 + f], nSubGrid, nSubHint, false, dealOn && h->dealG == nSubGrid);
        // k_compact's second workgroup deals the sub-blocks for the launches that follow (resident mode: 128 workgroups) -- on every DEAL_EVERY-th
        // keyframe of a call and whenever the table does not fit the grid: the pass takes one workgroup ~10 us against the compaction's ~6 beside it
        // (32 keys per thread through LDS atomics and scattered stores), and a table a few keyframes old still has nearly every sub-block in the
        // right band (the view moves a fraction of a band per keyframe; a misplaced sub-block only costs its XCD some extra lines)
        P.dealG = dealOn && compact && (f % DEAL_EVERY == 0 || h->dealG != nSubGrid) ? nSubGrid : 0;
        map_launch_compact(h->prof, sm, P, f, compact);
        if (P.dealG) h->dealG = nSubGrid;
        h->kfClassic++;
    };
    int f = 0;
    const int fProbe = n / 2;   // only when its profiler slot is enabled: what an event pair reports for an EMPTY dispatch at this place of the chain
    if (!compact || deferOff || churny || n < 2) {
        for (; f < n; f++) { classic(f); if (f == fProbe) map_launch_empty_pair(h->prof, sm); }
    } else {
        if (h->classicNext) { classic(0); f = 1; if (fProbe == 0) map_launch_empty_pair(h->prof, sm); }
        while (f < n) {
            const int w = std::min(DEFER_WIN, n - f);
            for (int q = 0; q < w; q++) {
                P.kf = q; P.prevSlotAbs = slot0 + f + q - 1;
                map_launch_fuse(h->prof, sm, P, f + q, h->h_frames[slot0 + f + q], nSubGrid, nSubHint, true, dealOn && h->dealG == nSubGrid);
                if (f + q == fProbe) map_launch_empty_pair(h->prof, sm);
            }
            P.kf = w; P.prevSlotAbs = slot0 + f + w - 1;
            map_launch_replay(h->prof, sm, P, w, (unsigned)h->blkStride);
            h->kfDeferred += (unsigned long long)w;
            if (dealOn) { P.dealG = nSubGrid; map_launch_deal(sm, P); h->dealG = nSubGrid; }   // one dealing per window, from the keys of its last keyframe
            f += w;
        }
    }
    if (compact) h->classicNext = false;
    if (sp != sm) { MSL_HIP_TRY(hipEventRecord(h->evMap[set], sm)); h->evMapValid[set] = true; }
    if (compact && h->h_snap) {   // snapshot of the live count after this batch (picked up by a later call, never waited for)
        const int i = h->snapNext;
        if (!h->snapBusy[i]) {
            MSL_HIP_TRY(hipMemcpyAsync(h->h_snap + (size_t)i * msl_sf::SNAPW, h->d_ctr, sizeof(long long) * msl_sf::SNAPW, hipMemcpyDeviceToHost, sm));
            MSL_HIP_TRY(hipEventRecord(h->snapEv[i], sm));
            h->snapKf[i] = h->kfEnq; h->snapBusy[i] = true; h->snapLive[i] = true; h->snapNext = (i + 1) % msl_sf::NSNAP;
        }
    }
    MSL_HIP_TRY(hipGetLastError());
    h->lastSlot = slot0 + n - 1;
    h->batchNo++;
    return MSL_OK;
}

}  // namespace

extern "C" {

msl_sf *msl_sf_create(int width, int height, float fx, float fy, float cx, float cy, float fuseFar, float fuseNear, int device) noexcept {
    try {
    if (width < 16 || height < 16 || fx == 0 || fy == 0 || (width / SP) * (height / SP) >= IDX_PLANE || (long long)width * height >= (1ll << 31)) {
        set_error("msl_sf_create: width/height must be >= 16 with fewer than 65534 superpixels, fx and fy non-zero");
        return nullptr;
    }
    if (bind_device(device) != MSL_OK) return nullptr;
    // (owned by a guard until the handle is complete: an exception below -- std::bad_alloc from the table vector -- lands in the catch barrier, and
    // the streams, events and device buffers created so far must go with it)
    std::unique_ptr<msl_sf, void (*)(msl_sf *)> guard(new msl_sf, [](msl_sf *p) { msl_sf_destroy(p); });
    msl_sf *h = guard.get();
    h->device = device;
    SfDev &D = h->dev;
    D.W = width; D.H = height; D.spW = width / SP; D.spH = height / SP; D.nseeds = D.spW * D.spH; D.npx = width * height;   // spWidth = width / SP_SIZE: truncation (:29-38)
    D.pxStride = (D.npx + 63) & ~63;
    D.flagStride = 64 * ((((D.nseeds + 63) / 64) + 15) & ~15);   // 64 lanes x a multiple of 16 seeds each
    D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy; D.fuseFar = fuseFar; D.fuseNear = fuseNear;
    bool ok = true;
    {   // the per-keyframe map stage is the latency-critical chain: highest priority for its stream, lowest for the
        // throughput-oriented frame-batched superpixel stage
        int lo = 0, hi = 0;
        ok = ok && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&h->preStream, hipStreamNonBlocking, lo) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&h->mapStream, hipStreamNonBlocking, hi) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking) == hipSuccess;
    }
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreateWithFlags(&h->evPre[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->evMap[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->evCopy[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->evH2D[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&h->d_ctr, sizeof(long long) * 32) == hipSuccess;   // 16 counters (read_ctr) + [16..18] the published live counts
    ok = ok && hipMemset(h->d_ctr, 0, sizeof(long long) * 32) == hipSuccess;
    ok = ok && hipHostMalloc(&h->h_ctr, sizeof(long long) * 16) == hipSuccess;
    ok = ok && hipHostMalloc(&h->h_snap, sizeof(long long) * msl_sf::NSNAP * msl_sf::SNAPW) == hipSuccess;
    for (int i = 0; i < msl_sf::NSNAP && ok; i++) ok = hipEventCreateWithFlags(&h->snapEv[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&h->d_new, sizeof(msl_surfel) * D.nseeds) == hipSuccess;
    ok = ok && hipMalloc(&h->d_tickets, sizeof(unsigned) * 8) == hipSuccess && hipMemset(h->d_tickets, 0, sizeof(unsigned) * 8) == hipSuccess;   // [0..1] tickets, [3] change-list length, [4..6] the rotating hand-over counts
    ok = ok && hipMalloc(&h->d_delU, sizeof(unsigned) * LIST_D) == hipSuccess;
    ok = ok && hipMalloc(&h->d_dc, sizeof(DeferCtl)) == hipSuccess && hipMemset(h->d_dc, 0, sizeof(DeferCtl)) == hipSuccess;
    {   // (u - cx) / fx and (v - cy) / fy of every integer pixel coordinate: the float expression of back_project
        // (src/SurfelFusion.cpp:80-85) evaluated once here instead of six divisions per pixel in kb_seed_plane
        std::vector<float> tab((size_t)width + 1 + height + 1);
        for (int u = 0; u <= width; u++) tab[u] = ((float)u - cx) / fx;
        for (int v = 0; v <= height; v++) tab[(size_t)width + 1 + v] = ((float)v - cy) / fy;
        ok = ok && hipMalloc(&h->d_projTab, sizeof(float) * tab.size()) == hipSuccess;
        ok = ok && hipMemcpy(h->d_projTab, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) == hipSuccess;
        D.colX = h->d_projTab; D.rowY = h->d_projTab + width + 1;
    }
    if (ok) h->propLds = sp_init_attributes(D.nseeds);
    if (!ok) { set_error("msl_sf_create: HIP allocation failed"); return nullptr; }
    memset(h->h_ctr, 0, sizeof(long long) * 16);
    D.ctr = h->d_ctr; D.newSurfels = h->d_new; D.tickets = h->d_tickets; D.delU = h->d_delU; D.delUCount = h->d_tickets + 4;
    D.dc = h->d_dc; D.kf = 0; D.prevSlotAbs = 0;
    h->prof.nk = MSL_SF_NKERNELS;
    if (alloc_slots(h, 1) != MSL_OK || map_realloc(h, 1 << 16, 0) != MSL_OK) return nullptr;
    return guard.release();
    } MSL_ABI_CATCH_PTR
}

void msl_sf_destroy(msl_sf *h) noexcept {
    try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->copyStream) (void)hipStreamSynchronize(h->copyStream);
    if (h->preStream) (void)hipStreamSynchronize(h->preStream);
    if (h->mapStream) (void)hipStreamSynchronize(h->mapStream);
    h->prof.destroy();
    free_slots(h);
    auto F = [](auto *p) { if (p) (void)hipFree(p); };
    F(h->d_ctr); F(h->d_tickets); F(h->d_delU); F(h->d_dc); F(h->d_rpStore); F(h->d_projTab); F(h->d_new); F(h->d_mapStore); F(h->d_blockSums); F(h->d_blockUpd); F(h->d_delList); F(h->d_srcOf); F(h->d_aos); F(h->d_snapStore);
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
    if (h->h_snap) (void)hipHostFree(h->h_snap);
    if (h->h_blk) (void)hipHostFree(h->h_blk);
    if (h->h_list) (void)hipHostFree(h->h_list);
    for (int i = 0; i < msl_sf::NSNAP; i++) if (h->snapEv[i]) (void)hipEventDestroy(h->snapEv[i]);
    for (int i = 0; i < 2; i++) { if (h->evPre[i]) (void)hipEventDestroy(h->evPre[i]); if (h->evMap[i]) (void)hipEventDestroy(h->evMap[i]); if (h->evCopy[i]) (void)hipEventDestroy(h->evCopy[i]); if (h->evH2D[i]) (void)hipEventDestroy(h->evH2D[i]); }
    if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
    if (h->ownStreams) { if (h->preStream) (void)hipStreamDestroy(h->preStream); if (h->mapStream) (void)hipStreamDestroy(h->mapStream); }
    delete h;
    } MSL_ABI_CATCH_VOID
}

int msl_sf_staged_gray(msl_sf *h, const uint8_t **gray_dev, size_t *row_stride, size_t *frame_stride, void **uploaded_event) noexcept {
    try {
    if (!h || !gray_dev || !row_stride || !frame_stride || !uploaded_event) return MSL_ERR_INVALID;
    if (h->stagedSet < 0 || !h->d_gray) { set_error("msl_sf_staged_gray: the last batch had no host images"); return MSL_ERR_INVALID; }
    *gray_dev = h->d_gray + (size_t)h->stagedSet * (size_t)h->maxBatch * h->grayCap;
    *row_stride = h->stagedGs; *frame_stride = h->grayCap; *uploaded_event = (void *)h->evH2D[h->stagedSet];
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_set_stream(msl_sf *h, void *hip_stream) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (h->ownStreams) { (void)hipStreamDestroy(h->preStream); (void)hipStreamDestroy(h->mapStream); }
    h->preStream = h->mapStream = (hipStream_t)hip_stream; h->ownStreams = false;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_set_batch_capacity(msl_sf *h, int max_frames) noexcept {
    try {
    if (!h || max_frames < 1 || max_frames > 4096) { set_error("msl_sf_set_batch_capacity: invalid argument"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (max_frames == h->maxBatch) return MSL_OK;
    return alloc_slots(h, max_frames);
    } MSL_ABI_CATCH_INT
}

int msl_sf_sync(msl_sf *h) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    return check_err(h);
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_reserve(msl_sf *h, size_t capacity) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    if (capacity <= h->mapCap) return MSL_OK;
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    return map_realloc(h, capacity, (size_t)h->h_ctr[0]);
    } MSL_ABI_CATCH_INT
}

static int ensure_aos(msl_sf *h, size_t n) {
    if (n > h->aosCap) {
        if (h->d_aos) (void)hipFree(h->d_aos);
        h->d_aos = nullptr; h->aosCap = 0;
        MSL_HIP_TRY(hipMalloc(&h->d_aos, sizeof(msl_surfel) * n));
        h->aosCap = n;
    }
    return MSL_OK;
}

int msl_sf_map_upload(msl_sf *h, const msl_surfel *host, size_t n) noexcept {
    try {
    if (!h || (n && !host)) return MSL_ERR_INVALID;
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (n + (size_t)h->dev.nseeds > h->mapCap) {
        rc = map_realloc(h, n + n / 4 + 4 * (size_t)h->dev.nseeds, 0);
        if (rc != MSL_OK) return rc;
    }
    hipStream_t s = h->mapStream;
    MSL_HIP_TRY(hipMemsetAsync(h->d_ctr + 13, 0, sizeof(long long), s));   // a fresh map: no wide r, g, b records yet
    if (n) {
        rc = ensure_aos(h, n);
        if (rc != MSL_OK) return rc;
        MSL_HIP_TRY(hipMemcpyAsync(h->d_aos, host, sizeof(msl_surfel) * n, hipMemcpyHostToDevice, s));
        map_launch_aos_to_soa(h->prof, s, h->dev, h->d_aos, (long long)n, false);
    }
    map_launch_set_ctr(s, h->dev, (long long)n, -1);
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->liveBound = n; h->liveKnown = n; h->liveKnownKf = h->kfEnq; h->classicNext = true;
    drop_live_snapshots(h);   // a count recorded before the upload would otherwise lower the bound below n
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_snapshot(msl_sf *h) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    if (n > h->snapCap) {
        if (h->d_snapStore) (void)hipFree(h->d_snapStore);
        h->d_snapStore = nullptr; h->snapCap = 0; h->snapValid = false;
        const size_t c = (n + 4095) & ~(size_t)4095;
        MSL_HIP_TRY(hipMalloc(&h->d_snapStore, sizeof(float) * MAP_WORDS * c));
        h->snapCap = c;
    }
    if (n) {
        MSL_HIP_TRY(hipMemcpy(h->d_snapStore, h->dev.map.hot, sizeof(HotPk) * n, hipMemcpyDeviceToDevice));
        MSL_HIP_TRY(hipMemcpy(h->d_snapStore + 4 * h->snapCap, h->dev.map.cold, sizeof(ColdRec) * n, hipMemcpyDeviceToDevice));
        if (h->h_ctr[13] & 1) MSL_HIP_TRY(hipMemcpy(h->d_snapStore + 12 * h->snapCap, h->dev.map.rgbWide, sizeof(int) * 3 * n, hipMemcpyDeviceToDevice));
        if (h->h_ctr[13] & 2) MSL_HIP_TRY(hipMemcpy(h->d_snapStore + 15 * h->snapCap, h->dev.map.utlWide, sizeof(int) * 2 * n, hipMemcpyDeviceToDevice));
    }
    h->snapN = n; h->snapValid = true; h->snapWide = h->h_ctr[13];
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_restore(msl_sf *h) noexcept {
    try {
    if (!h || !h->snapValid) { set_error("msl_sf_map_restore: no snapshot"); return MSL_ERR_INVALID; }
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    const size_t n = h->snapN;
    if (n + (size_t)h->dev.nseeds > h->mapCap) {   // the map was reallocated smaller than the snapshot (upload of a small map): grow again
        int rc = read_ctr(h);
        if (rc != MSL_OK) return rc;
        rc = map_realloc(h, n + n / 4 + 4 * (size_t)h->dev.nseeds, 0);
        if (rc != MSL_OK) return rc;
    }
    hipStream_t s = h->mapStream;   // ordered after every keyframe enqueued so far; the superpixel stream never touches the map
    if (n) {
        MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.hot, h->d_snapStore, sizeof(HotPk) * n, hipMemcpyDeviceToDevice, s));
        MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.cold, h->d_snapStore + 4 * h->snapCap, sizeof(ColdRec) * n, hipMemcpyDeviceToDevice, s));
        if (h->snapWide & 1) MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.rgbWide, h->d_snapStore + 12 * h->snapCap, sizeof(int) * 3 * n, hipMemcpyDeviceToDevice, s));
        if (h->snapWide & 2) MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.utlWide, h->d_snapStore + 15 * h->snapCap, sizeof(int) * 2 * n, hipMemcpyDeviceToDevice, s));
    }
    // the restored map has exactly the snapshot's wide-rgb state: without the flag a later snapshot would skip rgbWide and a restore of THAT
    // one would bring COLD_WIDE records back without their exact ints
    map_launch_set_ctr(s, h->dev, (long long)n, (int)h->snapWide);
    MSL_HIP_TRY(hipGetLastError());
    h->liveBound = n; h->liveKnown = n; h->liveKnownKf = h->kfEnq; h->classicNext = true;
    drop_live_snapshots(h);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_size(msl_sf *h, size_t *n_out) noexcept {
    try {
    if (!h || !n_out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    *n_out = (size_t)h->h_ctr[0];
    return check_err(h);
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_download(msl_sf *h, msl_surfel *host, size_t cap, size_t *n_out) noexcept {
    try {
    if (!h || !n_out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    *n_out = n;
    if (n > cap || (n && !host)) { set_error("msl_sf_map_download: capacity %zu < map size %zu", cap, n); return MSL_ERR_CAPACITY; }
    if (n) {
        rc = ensure_aos(h, n);
        if (rc != MSL_OK) return rc;
        hipStream_t s = h->mapStream;
        map_launch_soa_to_aos(h->prof, s, h->dev, h->d_aos, (long long)n);
        MSL_HIP_TRY(hipMemcpyAsync(host, h->d_aos, sizeof(msl_surfel) * n, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipStreamSynchronize(s));
    }
    return check_err(h);
    } MSL_ABI_CATCH_INT
}

static int map_select(msl_sf *h, int mode, int arg, bool mark, msl_surfel *out, size_t cap, size_t *n_out, const char *what) {
    if (!h || !n_out) { set_error("%s: invalid argument", what); return MSL_ERR_INVALID; }
    if (mark) { h->mirrorValid = false; h->classicNext = true; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);                       // waits for both streams
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    *n_out = 0;
    if (n == 0) return MSL_OK;
    hipStream_t s = h->mapStream;
    const SfDev P = h->dev;
    map_launch_select_count(s, P, mode, arg);
    rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    const size_t m = (size_t)h->h_ctr[7];
    *n_out = m;
    (void)hipMemsetAsync(h->d_ctr + 7, 0, sizeof(long long), s);
    if (m > cap || (m && !out)) { set_error("%s: %zu surfels selected, capacity %zu", what, m, cap); return MSL_ERR_CAPACITY; }
    if (m == 0) return MSL_OK;
    rc = ensure_aos(h, m);
    if (rc != MSL_OK) return rc;
    map_launch_select_write(s, P, mode, arg, h->d_aos, mark ? 1 : 0);
    MSL_HIP_TRY(hipMemcpyAsync(out, h->d_aos, sizeof(msl_surfel) * m, hipMemcpyDeviceToHost, s));
    MSL_HIP_TRY(hipStreamSynchronize(s));
    return MSL_OK;
}

int msl_sf_map_detach(msl_sf *h, int pose_index, msl_surfel *out, size_t cap, size_t *n_out) noexcept { try {
    return map_select(h, 0, pose_index, true, out, cap, n_out, "msl_sf_map_detach"); } MSL_ABI_CATCH_INT }
int msl_sf_map_export(msl_sf *h, int min_update_times, msl_surfel *out, size_t cap, size_t *n_out) noexcept { try {
    return map_select(h, 1, min_update_times, false, out, cap, n_out, "msl_sf_map_export"); } MSL_ABI_CATCH_INT }
// System::saveSurfels (src/System.cc:296-382) for the cloud SurfelMapping::Stop builds (src/SurfelMapping.cpp:62-104): the local surfels
// seen at least min_update_times times (filtered on the device, map order), then the caller's inactive surfels.  ASCII PLY with the
// element / property layout the reference hands to tinyply; NaN positions are skipped (:311-312); alpha = 1, quality = weight,
// radius = size * 1000 (SurfelMapping.cpp:80).  Number formatting is that of a default std::ostream (tinyply itself is a third party).
int msl_sf_export_ply(msl_sf *h, int min_update_times, const msl_surfel *inactive, size_t n_inactive, const char *path) noexcept {
    try {
    if (!h || !path || (n_inactive && !inactive)) { set_error("msl_sf_export_ply: invalid argument"); return MSL_ERR_INVALID; }
    size_t n = 0;
    int rc = msl_sf_map_export(h, min_update_times, nullptr, 0, &n);
    if (rc != MSL_OK && rc != MSL_ERR_CAPACITY) return rc;
    std::vector<msl_surfel> pts(n + n_inactive);
    if (n) { rc = msl_sf_map_export(h, min_update_times, pts.data(), n, &n); if (rc != MSL_OK) return rc; }
    for (size_t i = 0; i < n_inactive; i++) pts[n + i] = inactive[i];
    size_t count = 0;
    for (const msl_surfel &e : pts) count += std::isnan(e.px) ? 0 : 1;
    std::ofstream os(path, std::ios::out);
    if (os.fail()) { set_error("msl_sf_export_ply: cannot open %s", path); return MSL_ERR_INVALID; }
    os << "ply\nformat ascii 1.0\nelement vertex " << count << "\n";
    for (const char *p : {"x", "y", "z", "nx", "ny", "nz"}) os << "property float " << p << "\n";
    for (const char *p : {"red", "green", "blue", "alpha"}) os << "property uchar " << p << "\n";
    for (const char *p : {"quality", "radius"}) os << "property float " << p << "\n";
    os << "element camera 1\n";
    for (const char *p : {"view_px", "view_py", "view_pz", "x_axisx", "x_axisy", "x_axisz", "y_axisx", "y_axisy", "y_axisz", "z_axisx", "z_axisy", "z_axisz",
                          "focal", "scalex", "scaley", "centerx", "centery"})
        os << "property float " << p << "\n";
    os << "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
    for (const msl_surfel &e : pts) {
        if (std::isnan(e.px)) continue;
        os << e.px << " " << e.py << " " << e.pz << " " << e.nx << " " << e.ny << " " << e.nz << " " << (unsigned)(uint8_t)e.r << " " << (unsigned)(uint8_t)e.g << " "
           << (unsigned)(uint8_t)e.b << " 1 " << e.weight << " " << e.size * 1000 << "\n";
    }
    os << "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 " << (int)count << " 1 0 0\n";
    return os.fail() ? MSL_ERR_INVALID : MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_map_append(msl_sf *h, const msl_surfel *surfels, size_t n) noexcept {
    try {
    if (!h || (n && !surfels)) { set_error("msl_sf_map_append: invalid argument"); return MSL_ERR_INVALID; }
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    if (n == 0) return MSL_OK;
    const size_t cur = (size_t)h->h_ctr[0];
    if (cur + n + (size_t)h->dev.nseeds > h->mapCap) {
        rc = map_realloc(h, cur + n + (cur + n) / 4 + 4 * (size_t)h->dev.nseeds, cur);
        if (rc != MSL_OK) return rc;
    }
    rc = ensure_aos(h, n);
    if (rc != MSL_OK) return rc;
    hipStream_t s = h->mapStream;
    MSL_HIP_TRY(hipMemcpyAsync(h->d_aos, surfels, sizeof(msl_surfel) * n, hipMemcpyHostToDevice, s));
    map_launch_aos_to_soa(h->prof, s, h->dev, h->d_aos, (long long)n, true);
    map_launch_add_ctr(s, h->dev, (long long)n);
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->liveBound = cur + n; h->liveKnown = cur + n; h->liveKnownKf = h->kfEnq; h->classicNext = true;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_fuse_resident_batch(msl_sf *h, int n_frames, const int32_t *refs, const uint8_t *gray, size_t gray_stride,
                               size_t gray_frame_stride, const float *depth, size_t depth_stride, size_t depth_frame_stride,
                               const int32_t *member, size_t member_stride, size_t member_frame_stride, msl_mem img_mem,
                               const float *poses_colmajor) noexcept {
    try {
    if (!h) { set_error("msl_sf_fuse_resident_batch: NULL handle"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    return run_batch(h, n_frames, refs, gray, gray_stride, gray_frame_stride, depth, depth_stride, depth_frame_stride, member, member_stride,
                     member_frame_stride, img_mem, poses_colmajor, true);
    } MSL_ABI_CATCH_INT
}

int msl_sf_fuse_resident_batch_d16(msl_sf *h, int n_frames, const int32_t *refs, const uint8_t *gray, size_t gray_stride, size_t gray_frame_stride,
                                   const uint16_t *depth16, size_t depth16_stride, size_t depth16_frame_stride, float depth_factor,
                                   const int32_t *member, size_t member_stride, size_t member_frame_stride, msl_mem img_mem,
                                   const float *poses_colmajor) noexcept {
    try {
    if (!h) { set_error("msl_sf_fuse_resident_batch_d16: NULL handle"); return MSL_ERR_INVALID; }
    if (!depth16) { set_error("msl_sf_fuse_resident_batch_d16: NULL depth"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    return run_batch(h, n_frames, refs, gray, gray_stride, gray_frame_stride, nullptr, 0, 0, member, member_stride, member_frame_stride, img_mem, poses_colmajor,
                     true, depth16, depth16_stride, depth16_frame_stride, depth_factor);
    } MSL_ABI_CATCH_INT
}

int msl_sf_fuse_resident(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth,
                         size_t depth_stride, const int32_t *member, size_t member_stride, msl_mem img_mem,
                         const float pose_colmajor[16]) noexcept {
    try {
    if (!h) { set_error("msl_sf_fuse_resident: NULL handle"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    const int32_t ref = referenceFrameIndex;
    return run_batch(h, 1, &ref, gray, gray_stride, 0, depth, depth_stride, 0, member, member_stride, 0, img_mem, pose_colmajor, true);
    } MSL_ABI_CATCH_INT
}

int msl_sf_last_counters(msl_sf *h, int64_t counters[5]) noexcept {
    try {
    if (!h || !counters) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    counters[0] = h->h_ctr[4]; counters[1] = h->h_ctr[1]; counters[2] = h->h_ctr[2]; counters[3] = h->h_ctr[3]; counters[4] = h->h_ctr[0];
    return check_err(h);
    } MSL_ABI_CATCH_INT
}

// Host-vector mode.  The caller's vector is the map for this call; what travels is kept to what has to:
//   in : the whole vector (56 B per surfel) -- unless MSL_SF_LOCAL_UNCHANGED says it still is what the previous call on this handle left there, in
//        which case the device copy of that call is used as it stands (checked: same length, no other map operation on the handle in between);
//   out: only the stretches of the vector that hold surfels this keyframe touched.  k_fuse leaves a deleted and an updated count per SUB_ITEMS-surfel
//        sub-block; sub-blocks with neither are byte-identical to the caller's copy and are not sent back (runs of touched sub-blocks travel as
//        one copy each, small gaps bridged; more than 64 runs collapse into fewer by bridging larger gaps).
int msl_sf_fuse_ex(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth, size_t depth_stride,
                   const int32_t *member, size_t member_stride, const float pose_colmajor[16], msl_surfel *local, size_t n_local,
                   msl_surfel *new_out, size_t new_cap, size_t *n_new, unsigned flags) noexcept {
    try {
    if (!h || !pose_colmajor || !n_new || (n_local && !local)) { set_error("msl_sf_fuse: invalid argument"); return MSL_ERR_INVALID; }
    if (new_cap < (size_t)h->dev.nseeds || !new_out) { set_error("msl_sf_fuse: new_cap must be >= (w/8)*(h/8) = %d", h->dev.nseeds); return MSL_ERR_CAPACITY; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc;
    const bool reuse = (flags & MSL_SF_LOCAL_UNCHANGED) && h->mirrorValid && h->mirrorN == n_local;
    if (reuse) {
        // the device map is the caller's vector already: only the per-call counters start over
        map_launch_set_ctr(h->mapStream, h->dev, (long long)n_local, -1);
        h->liveBound = n_local; h->liveKnown = n_local; h->liveKnownKf = h->kfEnq;
    } else {
        rc = msl_sf_map_upload(h, local, n_local);
        if (rc != MSL_OK) return rc;
    }
    h->mirrorValid = false;   // (until this call has completed)
    const int32_t ref = referenceFrameIndex;
    rc = run_batch(h, 1, &ref, gray, gray_stride, 0, depth, depth_stride, 0, member, member_stride, 0, MSL_MEM_HOST, pose_colmajor, false);
    if (rc != MSL_OK) return rc;
    hipStream_t s = h->mapStream;
    const size_t nblk = (n_local + SUB_ITEMS - 1) / SUB_ITEMS;
    if (nblk > h->blkCap) {
        if (h->h_blk) (void)hipHostFree(h->h_blk);
        h->h_blk = nullptr; h->blkCap = 0;
        MSL_HIP_TRY(hipHostMalloc(&h->h_blk, sizeof(unsigned) * 2 * (nblk + 1024)));
        h->blkCap = nblk + 1024;
    }
    if (nblk) {
        MSL_HIP_TRY(hipMemcpyAsync(h->h_blk, h->dev.blockSums, sizeof(unsigned) * nblk, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipMemcpyAsync(h->h_blk + h->blkCap, h->dev.blockUpd, sizeof(unsigned) * nblk, hipMemcpyDeviceToHost, s));
    }
    rc = read_ctr(h);   // the call's first synchronisation: counters and the per-sub-block counts are on the host
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t K = (size_t)h->h_ctr[1];
    *n_new = K;
    // what was touched, and where
    struct Run { size_t b0, b1; };
    std::vector<Run> runs;
    size_t touched = 0, runSurfels = 0;
    for (size_t b = 0; b < nblk; b++) touched += (size_t)h->h_blk[b] + h->h_blk[h->blkCap + b];
    for (size_t gapMax = 4; ; gapMax *= 4) {
        runs.clear();
        for (size_t b = 0; b < nblk; b++) {
            if (!(h->h_blk[b] | h->h_blk[h->blkCap + b])) continue;
            if (!runs.empty() && b - runs.back().b1 <= gapMax) runs.back().b1 = b + 1;
            else runs.push_back({b, b + 1});
        }
        if (runs.size() <= 64) break;
    }
    for (const Run &r : runs) runSurfels += std::min(r.b1 * SUB_ITEMS, n_local) - r.b0 * SUB_ITEMS;
    // Two ways back.  Runs of touched sub-blocks copied straight into the caller's vector (~45 GB/s), or -- when few surfels in many sub-blocks
    // changed (a map in no particular order) -- a compact {index, record} list scattered by the CPU (~6 ns per record on top of its 60 bytes).
    const size_t listLimit = n_local / 8;
    const double costRuns = 56.0 * (double)runSurfels / 45e9, costList = (double)touched * (60.0 / 45e9 + 6e-9);
    if (touched && touched <= listLimit && costList < costRuns) {
        const size_t need = 256 + (sizeof(unsigned) + sizeof(msl_surfel)) * listLimit;
        if (need > h->listCap) {
            if (h->h_list) (void)hipHostFree(h->h_list);
            h->h_list = nullptr; h->listCap = 0;
            MSL_HIP_TRY(hipHostMalloc(&h->h_list, need));
            h->listCap = need;
        }
        // device side: the count sits in tickets[3], indices in delList, records in the AoS buffer (both >= n_local entries)
        unsigned *d_count = h->d_tickets + 3;
        MSL_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned), s));
        map_launch_collect_changed(s, h->dev, (int)ref, (long long)n_local, d_count, h->dev.delList, h->d_aos, (unsigned)listLimit);
        unsigned *hc = reinterpret_cast<unsigned *>(h->h_list);
        unsigned *hi = reinterpret_cast<unsigned *>(h->h_list + 256);
        msl_surfel *hr = reinterpret_cast<msl_surfel *>(h->h_list + 256 + sizeof(unsigned) * listLimit);
        MSL_HIP_TRY(hipMemcpyAsync(hc, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipStreamSynchronize(s));
        // (the list may be longer than `touched`: a surfel that carried lastUpdate == ref before the call is listed as well -- harmless, its
        // record is unchanged -- so the length is read first; a list beyond the staging size falls back to the runs)
        const size_t cnt = *hc;
        if (cnt <= listLimit) {
            MSL_HIP_TRY(hipMemcpyAsync(hi, h->dev.delList, sizeof(unsigned) * cnt, hipMemcpyDeviceToHost, s));
            MSL_HIP_TRY(hipMemcpyAsync(hr, h->d_aos, sizeof(msl_surfel) * cnt, hipMemcpyDeviceToHost, s));
            if (K) MSL_HIP_TRY(hipMemcpyAsync(new_out, h->d_new, sizeof(msl_surfel) * K, hipMemcpyDeviceToHost, s));
            MSL_HIP_TRY(hipStreamSynchronize(s));
            for (size_t j = 0; j < cnt; j++) local[hi[j]] = hr[j];
            h->mirrorValid = true; h->mirrorN = n_local;
            return MSL_OK;
        }
    }
    if (nblk && !runs.empty())
        map_launch_soa_to_aos(h->prof, s, h->dev, h->d_aos, (long long)n_local);
    for (const Run &r : runs) {
        const size_t i0 = r.b0 * SUB_ITEMS, i1 = std::min(r.b1 * SUB_ITEMS, n_local);
        MSL_HIP_TRY(hipMemcpyAsync(local + i0, h->d_aos + i0, sizeof(msl_surfel) * (i1 - i0), hipMemcpyDeviceToHost, s));
    }
    if (K) MSL_HIP_TRY(hipMemcpyAsync(new_out, h->d_new, sizeof(msl_surfel) * K, hipMemcpyDeviceToHost, s));
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->mirrorValid = true; h->mirrorN = n_local;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_fuse(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth, size_t depth_stride,
                const int32_t *member, size_t member_stride, const float pose_colmajor[16], msl_surfel *local, size_t n_local,
                msl_surfel *new_out, size_t new_cap, size_t *n_new) noexcept {
    try {
    return msl_sf_fuse_ex(h, referenceFrameIndex, gray, gray_stride, depth, depth_stride, member, member_stride, pose_colmajor, local, n_local, new_out,
                          new_cap, n_new, 0u);
    } MSL_ABI_CATCH_INT
}

int msl_sf_debug_seeds(msl_sf *h, msl_seed *out) noexcept {
    try {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    const size_t ns = h->dev.nseeds;
    MSL_HIP_TRY(hipMemcpy(out, h->d_seeds + ns * h->lastSlot, sizeof(msl_seed) * ns, hipMemcpyDeviceToHost));
    std::vector<uint8_t> fused(ns);
    MSL_HIP_TRY(hipMemcpy(fused.data(), h->d_fused + (size_t)h->dev.flagStride * h->lastSlot, ns, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ns; i++) out[i].fused = fused[i] & 1;   // (2 = invalid candidate, not a fusion)
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_sf_debug_ctr(msl_sf *h, int64_t out[16]) noexcept {
    try {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    for (int i = 0; i < 16; i++) out[i] = h->h_ctr[i];
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_sf_debug_scratch(msl_sf *h, int which, size_t offset_words, uint32_t *out, size_t n_words) noexcept {
    try {
    if (!h || !out || which < 0 || which > 5) return MSL_ERR_INVALID;
    if (which == 5) {   // keyframes this handle sent through the classic chain / through deferred windows (host state)
        if (n_words < 2) return MSL_ERR_INVALID;
        out[0] = (uint32_t)h->kfClassic; out[1] = (uint32_t)h->kfDeferred;
        return MSL_OK;
    }
    if (which == 4) {   // the grid the dealing table currently is a permutation for (host state; 0: none)
        if (n_words < 1) return MSL_ERR_INVALID;
        out[0] = (uint32_t)h->dealG;
        return MSL_OK;
    }
    if (offset_words + n_words > (which < 2 ? h->mapCap : h->blkStride)) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    const uint32_t *src = which == 0 ? h->d_srcOf : which == 1 ? h->d_delList : which == 2 ? h->dev.sbKeys : h->dev.deal;
    MSL_HIP_TRY(hipMemcpy(out, src + offset_words, sizeof(uint32_t) * n_words, hipMemcpyDeviceToHost));
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
// What an event pair carried by a dispatch (hipExtLaunchKernelGGL) reports for a kernel that does nothing: n launches of an empty kernel with
// `grid` single-wave workgroups on the map stream.  bench.py quotes it next to the roofline kernel's event time: rocprofv3's kernel duration
// (first wave start to last wave end) is shorter than the event time by about this much.
int msl_sf_debug_event_overhead(msl_sf *h, int grid, int n, float *mean_us) noexcept {
    try {
    if (!h || !mean_us || n < 1 || grid < 1) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    std::vector<hipEvent_t> ev(2 * (size_t)n);
    for (auto &e : ev) MSL_HIP_TRY(hipEventCreate(&e));
    for (int i = 0; i < n; i++) map_launch_empty(h->mapStream, grid, ev[2 * i], ev[2 * i + 1]);
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    double tot = 0;
    for (int i = 0; i < n; i++) { float ms = 0; MSL_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
    for (auto &e : ev) (void)hipEventDestroy(e);
    *mean_us = (float)(tot * 1e3 / n);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_sf_debug_index(msl_sf *h, int32_t *out) noexcept {
    try {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    const size_t npx = h->dev.npx;
    std::vector<unsigned short> tmp(npx);
    MSL_HIP_TRY(hipMemcpy(tmp.data(), h->d_index + (size_t)h->dev.pxStride * h->lastSlot, sizeof(unsigned short) * npx, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < npx; i++) out[i] = tmp[i];
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_sf_profile_enable(msl_sf *h, int on) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    h->prof.drain();
    h->prof.set_mode(on);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_sf_profile_stride(msl_sf *h, int stride) noexcept {
    try {
    if (!h || stride < 1) return MSL_ERR_INVALID;
    h->prof.stride = stride;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_sf_profile_read(msl_sf *h, float *ms, int32_t *launches) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    h->prof.drain();
    for (int i = 0; i < MSL_SF_NKERNELS; i++) { if (ms) ms[i] = h->prof.ms[i]; if (launches) launches[i] = h->prof.launches[i]; }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_debug_deal(const uint32_t *keys_host, int n_subblocks, uint32_t *deal_host) noexcept { try { return map_debug_deal(keys_host, n_subblocks, deal_host); } MSL_ABI_CATCH_INT }
int msl_debug_div100(const float *x_host, double *out_host, size_t n) noexcept { try { return sp_debug_div100(x_host, out_host, n); } MSL_ABI_CATCH_INT }
int msl_debug_chain_sum(const float *x_host, const int32_t *n_host, int lists, int huber, float *out_host) noexcept { try { return sp_debug_chain(x_host, n_host, lists, huber, out_host); } MSL_ABI_CATCH_INT }
const char *msl_sf_kernel_name(int k) noexcept { try { return (k >= 0 && k < MSL_SF_NKERNELS) ? kSfNames[k] : ""; } MSL_ABI_CATCH_PTR }

}  // extern "C"