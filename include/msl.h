/*
 * msl.h -- C ABI of the MI355X-native RGB-D front end (ORB extractor + surfel fusion).
 *
 * This is the drop-in boundary for ManhattanSLAM's two hot-path classes.  The reference has no
 * FFI layer; the interfaces replaced are C++ class members, cited per entry point below
 * (paths relative to the reference repository):
 *
 *   ORB_SLAM2::ORBextractor::ORBextractor(...)        include/ORBextractor.h:46-47, src/ORBextractor.cc:412-468
 *   ORB_SLAM2::ORBextractor::operator()(...)          include/ORBextractor.h:54-56, src/ORBextractor.cc:813-870
 *   ORB_SLAM2::ORBextractor::Get*Scale*()             include/ORBextractor.h:58-80
 *   SurfelFusion::SurfelFusion(...)                   include/SurfelFusion.h:127-129, src/SurfelFusion.cpp:29-38
 *   SurfelFusion::fuseInitializeMap(...)              include/SurfelFusion.h:131-138, src/SurfelFusion.cpp:40-73
 *   ORB_SLAM2::SurfelMapping::fuseMap(...)            src/SurfelMapping.cpp:353-392   (slot refill / tail compaction)
 *
 * All entry points are extern "C", take plain pointers and sizes, never throw, and return
 * MSL_OK (0) or a negative msl_status; msl_last_error() gives a thread-local message.
 * Every handle owns its HIP stream and scratch; a handle is used by one thread at a time
 * (same rule as the reference objects) but the calling thread may change between calls
 * (src/Frame.cc:100 spawns a fresh std::thread per frame) -- the device is re-bound on entry.
 *
 * There is NO CPU fallback: creation fails with MSL_ERR_NO_DEVICE when no gfx950 device is
 * usable.  The CPU oracle under oracle/ is test infrastructure and is not linked here.
 *
 * This header is the drop-in surface only.  The accessors the parity tests and bench.py use to look inside a handle (intermediate
 * stages, device counters, per-kernel HIP-event timing) are declared in msl_debug.h: exported by the same library, not part of the
 * boundary a maintainer of the reference binds.
 */
#ifndef MSL_H
#define MSL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSL_API __attribute__((visibility("default")))
/* No exception ever crosses this boundary: every entry point is noexcept and turns a failure inside the library (std::bad_alloc, a failed
 * thread spawn, ...) into a status code ("all noexcept, int return"). */
#ifdef __cplusplus
#define MSL_NOEXCEPT noexcept
#else
#define MSL_NOEXCEPT
#endif

typedef enum msl_status {
    MSL_OK = 0,
    MSL_ERR_INVALID = -1,      /* bad argument / unsupported geometry            */
    MSL_ERR_NO_DEVICE = -2,    /* no usable HIP device (no CPU fallback exists)  */
    MSL_ERR_HIP = -3,          /* a HIP runtime call failed                      */
    MSL_ERR_CAPACITY = -4,     /* caller-provided output capacity too small      */
    MSL_ERR_OVERFLOW = -5,     /* an internal device-side bound was exceeded     */
    MSL_ERR_NOMEM = -6,        /* host memory exhausted inside the library       */
    MSL_ERR_INTERNAL = -7      /* any other exception caught at the boundary     */
} msl_status;

typedef enum msl_mem {
    MSL_MEM_HOST = 0,          /* pointer is ordinary host memory                */
    MSL_MEM_DEVICE = 1         /* pointer is device memory on the handle's GPU   */
} msl_mem;

/* Same layout as cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct msl_keypoint {
    float x, y;
    float size;
    float angle;      /* degrees, [0,360) */
    float response;   /* FAST-9/16 corner score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} msl_keypoint;

/* Same layout as the reference `struct Surfel` (include/Surfel.h:28-37), 56 bytes. */
typedef struct msl_surfel {
    float px, py, pz;
    float nx, ny, nz;
    float size;
    float color;
    int32_t r, g, b;
    float weight;
    int32_t updateTimes;
    int32_t lastUpdate;
} msl_surfel;

/* Same layout as SurfelFusion::SuperpixelSeed (include/SurfelFusion.h:46-58), 64 bytes.
 * Only used by the debug accessors of msl_debug.h that let the parity tests look at intermediate stages. */
typedef struct msl_seed {
    float x, y;
    float size;
    float normX, normY, normZ;
    float posX, posY, posZ;
    float viewCos;
    float meanDepth;
    float meanIntensity;
    int32_t r, g, b;
    uint8_t fused, stable, use, _pad;
} msl_seed;

MSL_API const char *msl_last_error(void) MSL_NOEXCEPT;
MSL_API const char *msl_version(void) MSL_NOEXCEPT;
/* Number of usable gfx950 devices (0 if none / HIP unavailable). */
MSL_API int msl_device_count(void) MSL_NOEXCEPT;

/* ------------------------------------------------------------------------------------------
 * ORB extractor
 * ---------------------------------------------------------------------------------------- */
typedef struct msl_orb msl_orb;

/* Replaces ORBextractor::ORBextractor (src/ORBextractor.cc:412-468).  max_width/max_height bound
 * the frame size, max_batch the number of frames one msl_orb_extract_batch call may carry. */
MSL_API msl_orb *msl_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                                int minThFAST, int max_width, int max_height, int max_batch,
                                int device) MSL_NOEXCEPT;
MSL_API void msl_orb_destroy(msl_orb *h) MSL_NOEXCEPT;

/* Scale tables of include/ORBextractor.h:58-80; each out array holds nlevels floats (NULL = skip). */
MSL_API int msl_orb_scale_tables(const msl_orb *h, float *scaleFactors, float *invScaleFactors,
                                 float *levelSigma2, float *invLevelSigma2) MSL_NOEXCEPT;
/* mnFeaturesPerLevel (src/ORBextractor.cc:433-445). */
MSL_API int msl_orb_features_per_level(const msl_orb *h, int32_t *out) MSL_NOEXCEPT;
/* Upper bound on keypoints per frame: nfeatures + 2*nlevels (the one-by-one phase of DistributeOctTree overshoots a level's quota by at most 2,
 * src/ORBextractor.cc:691-696); for frames at least ~4 times as wide as high with a small budget the first quadtree round alone returns up to
 * 4 * round(width / height) nodes per level (:536-552, 575-640), and the capacity of an extractor created for such a frame size includes them. */
MSL_API int msl_orb_capacity(const msl_orb *h) MSL_NOEXCEPT;
MSL_API int msl_orb_levels(const msl_orb *h) MSL_NOEXCEPT;

/* Replaces ORBextractor::operator() (src/ORBextractor.cc:813-870) for one CV_8UC1 frame held in
 * host memory.  stride is in bytes.  On return *n_out keypoints (level order 0..L-1, in-level
 * order = quadtree list order) and n_out*32 descriptor bytes are in the caller's host buffers.
 * An empty image (width==0||height==0||gray==NULL) returns MSL_OK with *n_out = 0 (:815-816). */
MSL_API int msl_orb_extract(msl_orb *h, const uint8_t *gray, int width, int height, size_t stride,
                            msl_keypoint *kps, uint8_t *desc32, int cap, int *n_out) MSL_NOEXCEPT;

/* Frame-batched variant (throughput path).  Frame f starts at gray + f*frame_stride.  Outputs for
 * frame f are written at kps + f*cap, desc32 + f*cap*32, n_out[f].  in_mem/out_mem say whether
 * the input / the three output pointers are host or device memory.  With device outputs the
 * call is asynchronous on the handle's stream: use msl_orb_sync() before reading. */
/* The extractor's stream waits for a hipEvent_t (e.g. the one msl_sf_staged_gray returns) before anything enqueued after this call. */
MSL_API int msl_orb_wait_event(msl_orb *h, void *hip_event) MSL_NOEXCEPT;
MSL_API int msl_orb_extract_batch(msl_orb *h, const uint8_t *gray, int n_frames, int width,
                                  int height, size_t row_stride, size_t frame_stride,
                                  msl_mem in_mem, msl_keypoint *kps, uint8_t *desc32, int cap,
                                  int32_t *n_out, msl_mem out_mem) MSL_NOEXCEPT;
/* ---- widening, SURVEY.md 8(f) rank 1: the Frame steps that consume the ORB output right after the join
 * (src/Frame.cc:107-153): UndistortKeyPoints (:437-463), ComputeStereoFromRGBD (:495-513), AssignFeaturesToGrid
 * (:155-168, PosInGrid :418-427), fused behind the extraction so the keypoints never leave HBM in between. ---- */
typedef struct msl_frame_params {
    float fx, fy, cx, cy;          /* mK (CV_32F) */
    float k1, k2, p1, p2, k3;      /* mDistCoef; k1 == 0 => mvKeysUn = mvKeys (src/Frame.cc:438-441) */
    float bf;                      /* mbf */
    float minX, maxX, minY, maxY;  /* mnMinX.. (ComputeImageBounds, src/Frame.cc:465-494): see msl_frame_image_bounds */
} msl_frame_params;
#define MSL_FRAME_GRID_ROWS 48     /* include/Frame.h:53-54 */
#define MSL_FRAME_GRID_COLS 64
/* ComputeImageBounds: fills minX..maxY from fx..k3 and the image size (host arithmetic only). */
MSL_API int msl_frame_image_bounds(msl_frame_params *p, int width, int height) MSL_NOEXCEPT;
/* msl_orb_extract_batch plus, per keypoint i of frame f (outputs at index f*cap + i, same memory space as kps):
 *   kps_un_xy[2i..2i+1] = mvKeysUn[i].pt      depth_out[i] = mvDepth[i] (-1 if the depth pixel is <= 0)
 *   uright_out[i] = mvuRight[i]               grid_cell[i] = posX * 48 + posY of mGrid[posX][posY], or -1 (PosInGrid false)
 * depth: CV_32FC1 metres (imDepthScaled), strides in bytes. */
MSL_API int msl_orb_extract_frame_batch(msl_orb *h, const uint8_t *gray, const float *depth, int n_frames, int width, int height,
                                        size_t gray_row_stride, size_t gray_frame_stride, size_t depth_row_stride,
                                        size_t depth_frame_stride, msl_mem in_mem, const msl_frame_params *params,
                                        msl_keypoint *kps, uint8_t *desc32, float *kps_un_xy, float *depth_out, float *uright_out,
                                        int32_t *grid_cell, int cap, int32_t *n_out, msl_mem out_mem) MSL_NOEXCEPT;
MSL_API int msl_orb_sync(msl_orb *h) MSL_NOEXCEPT;
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the handle's own. */
MSL_API int msl_orb_set_stream(msl_orb *h, void *hip_stream) MSL_NOEXCEPT;


/* ------------------------------------------------------------------------------------------
 * Surfel fusion
 * ---------------------------------------------------------------------------------------- */
typedef struct msl_sf msl_sf;

/* Replaces SurfelFusion::SurfelFusion (src/SurfelFusion.cpp:29-38).  Any width, height >= 16: like the reference, the superpixel lattice is
 * (width / 8) x (height / 8), truncated; the pixels right of / below the last whole cell still take part in every per-pixel step. */
MSL_API msl_sf *msl_sf_create(int width, int height, float fx, float fy, float cx, float cy,
                              float fuseFar, float fuseNear, int device) MSL_NOEXCEPT;
MSL_API void msl_sf_destroy(msl_sf *h) MSL_NOEXCEPT;

/* Host-vector mode == SurfelFusion::fuseInitializeMap (src/SurfelFusion.cpp:40-73).
 * gray: CV_8UC1 w*h; depth: CV_32FC1 metres; member: CV_32SC1 ceil(w/2)*ceil(h/2), -1 = no plane; strides
 * in bytes; pose = Twc as column-major 4x4 (Eigen::Matrix4f storage).  `local` (n_local surfels)
 * is updated in place; new surfels are written to new_out (<= (w/8)*(h/8)), count in *n_new. */
MSL_API int msl_sf_fuse(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride,
                        const float *depth, size_t depth_stride, const int32_t *member,
                        size_t member_stride, const float pose_colmajor[16], msl_surfel *local,
                        size_t n_local, msl_surfel *new_out, size_t new_cap, size_t *n_new) MSL_NOEXCEPT;

/* The same with hints.  MSL_SF_LOCAL_UNCHANGED: local[0 .. n_local) is byte for byte what the previous msl_sf_fuse / msl_sf_fuse_ex call on this
 * handle left there (a caller that keeps the new surfels in a list of their own, or that has not run SurfelMapping::fuseMap's refill yet); the
 * library then fuses into the device copy of that call instead of uploading 56 bytes per surfel again.  The hint is ignored -- a full upload
 * happens -- when the length differs or any other map operation touched the handle in between.  Both forms send back only the stretches of
 * `local` that hold surfels this keyframe updated or deleted (per 256-surfel sub-block), not the whole vector. */
#define MSL_SF_LOCAL_UNCHANGED 1u
MSL_API int msl_sf_fuse_ex(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride,
                           const float *depth, size_t depth_stride, const int32_t *member,
                           size_t member_stride, const float pose_colmajor[16], msl_surfel *local,
                           size_t n_local, msl_surfel *new_out, size_t new_cap, size_t *n_new, unsigned flags) MSL_NOEXCEPT;

/* Device-resident map mode: the live surfel map stays in HBM between keyframes. */
MSL_API int msl_sf_map_reserve(msl_sf *h, size_t capacity) MSL_NOEXCEPT;
MSL_API int msl_sf_map_upload(msl_sf *h, const msl_surfel *host, size_t n) MSL_NOEXCEPT;
MSL_API int msl_sf_map_download(msl_sf *h, msl_surfel *host, size_t cap, size_t *n_out) MSL_NOEXCEPT;
MSL_API int msl_sf_map_size(msl_sf *h, size_t *n_out) MSL_NOEXCEPT;
/* Replay support (bench.py's stationary sequence, tests): msl_sf_map_snapshot keeps a device-side copy of the resident map and its
 * live count (synchronous); msl_sf_map_restore puts that copy back, asynchronously on the map stream, ordered after every keyframe
 * enqueued so far -- a device-to-device copy of the records, no host traffic.  (No reference counterpart: Tracking::Reset does not
 * touch the surfel vectors, SURVEY.md App. D.) */
MSL_API int msl_sf_map_snapshot(msl_sf *h) MSL_NOEXCEPT;
MSL_API int msl_sf_map_restore(msl_sf *h) MSL_NOEXCEPT;

/* fuseInitializeMap + the SurfelMapping::fuseMap slot refill / tail compaction
 * (src/SurfelMapping.cpp:353-392) on the resident map.  Image pointers may be host or device
 * (img_mem).  Asynchronous on the handle's streams: only argument errors are reported by the call itself;
 * device-side errors are DEFERRED to the next msl_sf_sync / map_size / download / detach / append / export
 * / last_counters, which return MSL_ERR_OVERFLOW once.  The resident map grows on demand (like the
 * reference's std::vector): when the host-side upper bound of the live count reaches the capacity the call
 * syncs once and reallocates; msl_sf_map_reserve avoids that pause.  msl_sf_last_counters gives
 * {n_live_before, n_new, n_deleted, n_updated, n_live_after} of the last keyframe. */
MSL_API int msl_sf_fuse_resident(msl_sf *h, int referenceFrameIndex, const uint8_t *gray,
                                 size_t gray_stride, const float *depth, size_t depth_stride,
                                 const int32_t *member, size_t member_stride, msl_mem img_mem,
                                 const float pose_colmajor[16]) MSL_NOEXCEPT;
/* ---- widening, SURVEY.md 8(f) rank 4: map maintenance on the resident map (so SurfelMapping::moveAddSurfels and Stop() need
 * no full download/upload).  All three are synchronous and keep the reference's element order. ----
 * msl_sf_map_detach: the inner loop of moveAddSurfels (src/SurfelMapping.cpp:207-224) for one leaving pose: every live surfel
 *   (updateTimes > 0) whose lastUpdate == pose_index is copied, in map order, to `out` (host) and marked deleted in the map
 *   (updateTimes = 0).  *n_out = number found; MSL_ERR_CAPACITY (nothing modified) if it exceeds cap.
 * msl_sf_map_append: mvLocalSurfels.insert(end, ...) of re-entering poses (src/SurfelMapping.cpp:291-296).
 * msl_sf_map_export: the local-surfel filter of SurfelMapping::Stop (src/SurfelMapping.cpp:67-84): surfels with
 *   updateTimes >= min_update_times, in map order. */
MSL_API int msl_sf_map_detach(msl_sf *h, int pose_index, msl_surfel *out, size_t cap, size_t *n_out) MSL_NOEXCEPT;
MSL_API int msl_sf_map_append(msl_sf *h, const msl_surfel *surfels, size_t n) MSL_NOEXCEPT;
MSL_API int msl_sf_map_export(msl_sf *h, int min_update_times, msl_surfel *out, size_t cap, size_t *n_out) MSL_NOEXCEPT;
/* System::saveSurfels (src/System.cc:296-382) on the cloud of SurfelMapping::Stop (src/SurfelMapping.cpp:62-104): msl_sf_map_export(min_update_times)
 * followed by the caller's inactive surfels, written as the reference's ASCII PLY (vertex: x y z nx ny nz red green blue alpha quality radius; one
 * camera element).  (The map-plane points Stop() appends are Map data outside this library; pass them through `inactive` if wanted.) */
MSL_API int msl_sf_export_ply(msl_sf *h, int min_update_times, const msl_surfel *inactive, size_t n_inactive, const char *path) MSL_NOEXCEPT;

/* ---- widening, SURVEY.md 8(f) rank 2: the data-parallel front of the PEAC plane extractor (producer of membershipImg) ----
 * msl_peac_block_stats: for n_frames raw 16-bit depth images
 *   (a) the organised half-resolution point cloud of PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:44-76):
 *       vertex (i/2, j/2) = (((double)j - cx) * z / fx, ((double)i - cy) * z / fy, z), z = (double)depth(i, j) * depthMapFactor,
 *       for even i, j; cloud_out[frame][(h/2... ceil) * (w/2 ... ceil)][3] doubles (may be NULL);
 *   (b) the initial node of every window_w x window_h block, i.e. the body of ahc::PlaneSeg::PlaneSeg
 *       (include/peac/AHCPlaneSeg.hpp:237-285) as ahc::PlaneFitter::initGraph calls it (include/peac/AHCPlaneFitter.hpp:756-776):
 *       points with z == 0 are missing data (include/PlaneExtractor.h:47-55), a block with missing data (INIT_STRICT; more
 *       than half missing for init_loose != 0) or a depth discontinuity |z - z_nb| > depth_alpha * |z| + depth_change_tol
 *       towards the right / lower neighbour (AHCPlaneSeg.hpp:41-43, AHCParamSet.hpp:140-142) is rejected (nouse = 1, N = 0,
 *       sums 0); otherwise the nine FP64 sums of ahc::PlaneSeg::Stats::push (AHCPlaneSeg.hpp:81-92) in window raster order.
 *       stats_out[frame][(H / window_h) * (W / window_w)], H = ceil(height / 2), W = ceil(width / 2), block (i, j) at i * Nw + j.
 * The PCA plane fit of each block (a 3x3 symmetric eigen-solve through Eigen), the graph and the agglomerative clustering stay
 * host code.  Synchronous; depth / outputs in host or device memory as `mem` / `out_mem` say. */
typedef struct msl_peac_stats {
    double sx, sy, sz, sxx, syy, szz, sxy, syz, sxz;   /* ahc::PlaneSeg::Stats (AHCPlaneSeg.hpp:59-63) */
    int32_t N;
    int32_t nouse;
} msl_peac_stats;
MSL_API int msl_peac_block_stats(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height,
                                 int n_frames, msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, int window_w,
                                 int window_h, double depth_alpha, double depth_change_tol, int init_loose, double *cloud_out,
                                 msl_peac_stats *stats_out, msl_mem out_mem) MSL_NOEXCEPT;

/* The rest of the plane extractor: the producer of SurfelFusion's inputPlaneMembershipImg (BASELINE config 4).
 * msl_peac_params = the members of ahc::PlaneFitter (include/peac/AHCPlaneFitter.hpp:122-131, 157-161) and ahc::ParamSet
 * (include/peac/AHCParamSet.hpp:36-76); msl_peac_default_params fills the reference's defaults (PlaneDetection overrides none).
 * msl_peac_block = the initial graph node of one window: its Stats plus the PCA plane fit of ahc::PlaneSeg::Stats::compute
 * (AHCPlaneSeg.hpp:148-183: centre of mass, unit normal towards the camera, MSE, curvature; NaN MSE / curvature for N < 4), the
 * 3x3 symmetric eigen-solve being LA::eig33sym = Eigen::SelfAdjointEigenSolver<Matrix3d> (include/peac/eig33sym.hpp:71-75).
 * msl_peac_block_fit: cloud + block statistics + PCA on the GPU (one wave per window, FP64, the reference's summation order).
 * msl_peac_membership_batch: PlaneDetection::readDepthImage + runPlaneDetection (src/PlaneExtractor.cpp:44-81) for n_frames
 *   depth images: block fit on the GPU; graph initialisation (AHCPlaneFitter.hpp:756-928) on the host; agglomerative clustering (:939-1143)
 *   on the GPU, one wave per frame, for calls of more than about three frames per usable CPU (on the host workers for smaller calls -- a lone
 *   frame is clustered faster by one core --, if a frame's node data does not fit the LDS, or as MSL_PEAC_CLUSTER=host / device says); block
 *   erosion (:490-596), region growing (:422-471) and the final merge / relabelling (:296-372) on the host (order-dependent pixel work: a
 *   FIFO flood fill), one frame per worker thread at a time (as many workers as the process may use CPUs: affinity mask and cgroup quota,
 *   at most 64).  Device-resident input must be complete, or enqueued on the legacy default stream, when the call is made.
 *   membership_out (HOST, [n_frames][ceil(h/2)][ceil(w/2)]) = plane_filter.membershipImg as SurfelMapping receives it
 *   (src/Tracking.cc:228): plane id >= 0, -1 = no plane, and -- exactly like the reference -- the region-growing visit counters
 *   -2..-6 on pixels that were tried and rejected.  n_planes_out (HOST, may be NULL) = extractedPlanes.size() per frame. */
typedef struct msl_peac_block {
    msl_peac_stats stats;
    double center[3], normal[3], mse, curvature;
} msl_peac_block;
typedef struct msl_peac_params {
    int32_t window_w, window_h;       /* windowWidth, windowHeight (10, 10) */
    int32_t min_support;              /* minSupport (3000) */
    int32_t max_step;                 /* maxStep (100000) */
    int32_t do_refine;                /* doRefine (true) */
    int32_t erode_type;               /* ErodeType: 0 none, 1 segment borders, 2 all borders (ERODE_ALL_BORDER) */
    int32_t init_loose;               /* initType == INIT_LOOSE (INIT_STRICT) */
    int32_t _pad;
    double depth_sigma, std_tol_init, std_tol_merge;              /* depthSigma, stdTol_init, stdTol_merge */
    double z_near, z_far, angle_near, angle_far;                  /* T_ang(P_INIT) */
    double similarity_th_merge, similarity_th_refine;             /* cos 60 deg, cos 30 deg */
    double depth_alpha, depth_change_tol;                         /* T_dz */
} msl_peac_params;
MSL_API void msl_peac_default_params(msl_peac_params *p) MSL_NOEXCEPT;
MSL_API int msl_peac_block_fit(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height,
                               int n_frames, msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor,
                               const msl_peac_params *params, msl_peac_block *blocks_out, msl_mem out_mem) MSL_NOEXCEPT;
MSL_API int msl_peac_membership_batch(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width,
                                      int height, int n_frames, msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor,
                                      const msl_peac_params *params, int32_t *membership_out, int32_t *n_planes_out) MSL_NOEXCEPT;
/* The host stage of msl_peac_membership_batch alone (graph initialisation, clustering, erosion, region growing; persistent worker threads, one
 * frame per thread at a time), on block fits the caller already has (blocks: HOST, [n_frames][Nh * Nw] as msl_peac_block_fit returns them) and the
 * HOST depth images they came from.  No device is touched: this is the part of the extractor that is sequential by construction. */
MSL_API int msl_peac_membership_from_blocks(const msl_peac_block *blocks, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes,
                                            int width, int height, int n_frames, float fx, float fy, float cx, float cy, float depth_map_factor,
                                            const msl_peac_params *params, int32_t *membership_out, int32_t *n_planes_out) MSL_NOEXCEPT;
/* Everything the reference's PlaneDetection hands on after runPlaneDetection (include/PlaneExtractor.h:57-62, src/PlaneExtractor.cpp:77-80):
 * msl_peac_membership_batch's outputs plus, per frame,
 *   planes_out         HOST [n_frames][max_planes]      plane_filter.extractedPlanes[i]: normal, centre, MSE, N (src/Frame.cc:626-632 reads them);
 *                                                       the call fails with MSL_ERR_CAPACITY if a frame has more than max_planes planes
 *   vertex_offsets_out HOST [n_frames][max_planes + 1]  plane_vertices_[i] = vertex_indices_out[f][offsets[i] .. offsets[i + 1])
 *   vertex_indices_out HOST [n_frames][ceil(h/2) * ceil(w/2)]  cloud vertex indices of every plane, raster order inside a plane (the pMembership
 *                                                       argument of PlaneFitter::run, AHCPlaneFitter.hpp:341-361); needs params->do_refine
 *   cloud_out          HOST [n_frames][ceil(h/2) * ceil(w/2)][3] doubles or NULL: PlaneDetection::cloud.vertices, the organised cloud of
 *                                                       readDepthImage (src/PlaneExtractor.cpp:60-74) as the device computed it for the block fit
 * Host worker threads: as many as the process may use CPUs (affinity mask, cgroup quota), divided by LOCAL_WORLD_SIZE when one process per GPU
 * shares the node (torch.distributed.run sets it); MSL_PEAC_THREADS overrides.
 * msl_peac_extract_from_blocks: the same from block fits the caller already has (host stage only, no device). */
typedef struct msl_peac_plane { double normal[3], center[3], mse; int32_t N, _pad; } msl_peac_plane;
MSL_API int msl_peac_extract_batch(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height,
                                   int n_frames, msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor,
                                   const msl_peac_params *params, int32_t *membership_out, int32_t *n_planes_out, int max_planes,
                                   msl_peac_plane *planes_out, int32_t *vertex_offsets_out, int32_t *vertex_indices_out, double *cloud_out) MSL_NOEXCEPT;
MSL_API int msl_peac_extract_from_blocks(const msl_peac_block *blocks, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes,
                                         int width, int height, int n_frames, float fx, float fy, float cx, float cy, float depth_map_factor,
                                         const msl_peac_params *params, int32_t *membership_out, int32_t *n_planes_out, int max_planes,
                                         msl_peac_plane *planes_out, int32_t *vertex_offsets_out, int32_t *vertex_indices_out) MSL_NOEXCEPT;

/* ---- widening, SURVEY.md 8(f) rank 3: Hamming matching by projection, the next consumer of the ORB descriptors ----
 * msl_match_by_projection_batch: n_pairs independent calls of
 *     int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th)   (src/ORBmatcher.cc:547-678)
 * with Frame::GetFeaturesInArea (src/Frame.cc:332-381), ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:835-849) and the rotation
 * histogram / ComputeThreeMaxima (:799-830), for an ORBmatcher with mbCheckOrientation = check_orientation.
 * All per-keypoint arrays hold `cap` entries per pair (pair f at index f * cap); the current-frame arrays are exactly the
 * outputs of msl_orb_extract_frame_batch (mvKeys for octave / angle, mvKeysUn.pt, mvuRight, grid cell, descriptors), so in a
 * batched pipeline they never leave HBM.  Last frame, per keypoint i < n_last[f]:
 *   last_xyz[3 i..]  LastFrame.mvpMapPoints[i]->GetWorldPos()          last_desc[32 i..]  ->GetDescriptor()
 *   last_flags[i]    bit 0: mvpMapPoints[i] != NULL && !mvbOutlier[i]; bit 1: ->Observations() > 0
 *   last_octave[i]   LastFrame.mvKeys[i].octave                        last_angle[i]      LastFrame.mvKeysUn[i].angle
 * Tcw_cur / Tcw_last: rows 0-2 of the CV_32F 4x4 mTcw of the two frames, row-major (12 floats per pair).
 * CurrentFrame.mvpMapPoints is all NULL on entry (src/Tracking.cc:1252); on return match_out[f * cap + i2] is the index i of the
 * last-frame keypoint whose MapPoint current keypoint i2 holds, or -1 (NULL); nmatches[f] is the function's return value.
 * The greedy, order-dependent assignment of the reference (a candidate already held by a point with Observations() > 0 is
 * skipped, later points overwrite earlier ones) is reproduced exactly.  Limits: cap <= 8192, nlevels <= MSL_MATCH_MAX_LEVELS.
 * `mem` / `out_mem` say where the input / the two output arrays live. */
#define MSL_MATCH_MAX_LEVELS 16
typedef struct msl_match_params {
    float fx, fy, cx, cy;             /* CurrentFrame.fx .. cy */
    float bf;                         /* mbf; mb = mbf / fx (src/Frame.cc:150) */
    float minX, maxX, minY, maxY;     /* mnMinX .. (msl_frame_image_bounds) */
    float th;                         /* search window: radius = th * mvScaleFactors[octave] */
    int32_t check_orientation;        /* ORBmatcher::mbCheckOrientation */
    int32_t nlevels;
    float scale_factors[MSL_MATCH_MAX_LEVELS];   /* CurrentFrame.mvScaleFactors (msl_orb_scale_tables) */
} msl_match_params;
/* One matcher handle = one ORBmatcher object of the reference (src/ORBmatcher.cc:41): its own HIP stream and its own grow-only scratch and staging
 * buffers, used by one thread at a time, device re-bound at every entry.  msl_match_by_projection is asynchronous on the handle's stream when inputs
 * AND outputs are device memory (msl_match_sync / msl_match_set_stream as for the other handles); with host memory on either side it returns when the
 * caller's buffers are its own again. */
typedef struct msl_match msl_match;
MSL_API msl_match *msl_match_create(int device) MSL_NOEXCEPT;
MSL_API void msl_match_destroy(msl_match *h) MSL_NOEXCEPT;
MSL_API int msl_match_sync(msl_match *h) MSL_NOEXCEPT;
MSL_API int msl_match_set_stream(msl_match *h, void *hip_stream) MSL_NOEXCEPT;
MSL_API int msl_match_by_projection(msl_match *h, int n_pairs, int cap, const msl_match_params *params,
                                    const msl_keypoint *cur_kps, const float *cur_un_xy, const float *cur_uright,
                                    const int32_t *cur_grid_cell, const uint8_t *cur_desc, const int32_t *n_cur,
                                    const float *last_xyz, const uint8_t *last_desc, const uint8_t *last_flags,
                                    const int32_t *last_octave, const float *last_angle, const int32_t *n_last,
                                    const float *Tcw_cur, const float *Tcw_last, msl_mem mem, int32_t *match_out,
                                    int32_t *nmatches, msl_mem out_mem) MSL_NOEXCEPT;
/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:835-849) for n descriptor pairs (host arrays, synchronous; parity hook for the popcount path). */
MSL_API int msl_match_descriptor_distances(msl_match *h, const uint8_t *a32, const uint8_t *b32, int n, int32_t *dist_out) MSL_NOEXCEPT;
/* Device-indexed convenience forms of the two calls above: a lazily created handle per device shared by all callers (serialised), always
 * synchronous; device-resident inputs must be complete, or enqueued on the legacy default stream, when the call is made. */
MSL_API int msl_match_by_projection_batch(int device, int n_pairs, int cap, const msl_match_params *params,
                                          const msl_keypoint *cur_kps, const float *cur_un_xy, const float *cur_uright,
                                          const int32_t *cur_grid_cell, const uint8_t *cur_desc, const int32_t *n_cur,
                                          const float *last_xyz, const uint8_t *last_desc, const uint8_t *last_flags,
                                          const int32_t *last_octave, const float *last_angle, const int32_t *n_last,
                                          const float *Tcw_cur, const float *Tcw_last, msl_mem mem, int32_t *match_out,
                                          int32_t *nmatches, msl_mem out_mem) MSL_NOEXCEPT;
MSL_API int msl_match_descriptor_distance(int device, const uint8_t *a32, const uint8_t *b32, int n, int32_t *dist_out) MSL_NOEXCEPT;

/* Batched form: n_frames keyframes in order, semantically n_frames consecutive msl_sf_fuse_resident calls.
 * Keyframe f's images start at base + f * <frame_stride> bytes (member_frame_stride may be 0: one shared
 * membership image); refs[n_frames] and poses (16 * n_frames floats, column-major Twc each) are host arrays.
 * generateSuperPixels() of all keyframes of a batch runs frame-batched on a second stream and overlaps the
 * per-keyframe map stage of the previous batch.  n_frames <= the capacity set below (default 1). */
MSL_API int msl_sf_set_batch_capacity(msl_sf *h, int max_frames) MSL_NOEXCEPT;
/* (All surfel entry points: the rows of ONE image must span less than 4 GB -- stride * rows < 2^32 bytes for gray, depth and membership images alike;
 * otherwise MSL_ERR_INVALID.  The kernels address an image with 32-bit byte offsets.) */
MSL_API int msl_sf_fuse_resident_batch(msl_sf *h, int n_frames, const int32_t *refs, const uint8_t *gray,
                                       size_t gray_stride, size_t gray_frame_stride, const float *depth,
                                       size_t depth_stride, size_t depth_frame_stride, const int32_t *member,
                                       size_t member_stride, size_t member_frame_stride, msl_mem img_mem,
                                       const float *poses_colmajor) MSL_NOEXCEPT;
/* The same call for RAW 16-bit depth images (the sensor's / the data set's format): keyframe f's depth is depth16 + f * depth16_frame_stride bytes,
 * rows depth16_stride bytes apart, and becomes metres on the device as (float)raw * depth_factor -- what Frame::Frame does on the host with
 * imDepth.convertTo(imDepthScaled, CV_32F, depthMapFactor) (src/Frame.cc:96-97, depthMapFactor = 1 / DepthMapFactor of the settings file,
 * src/Tracking.cc:133-137; OpenCV evaluates that conversion in float).  Half the depth bytes cross PCIe and the host loop disappears; the results
 * are bit-identical to msl_sf_fuse_resident_batch on the converted images.  img_mem applies to gray, depth16 and member alike. */
MSL_API int msl_sf_fuse_resident_batch_d16(msl_sf *h, int n_frames, const int32_t *refs, const uint8_t *gray,
                                           size_t gray_stride, size_t gray_frame_stride, const uint16_t *depth16,
                                           size_t depth16_stride, size_t depth16_frame_stride, float depth_factor,
                                           const int32_t *member, size_t member_stride, size_t member_frame_stride,
                                           msl_mem img_mem, const float *poses_colmajor) MSL_NOEXCEPT;
MSL_API int msl_sf_last_counters(msl_sf *h, int64_t counters[5]) MSL_NOEXCEPT;
MSL_API int msl_sf_sync(msl_sf *h) MSL_NOEXCEPT;
MSL_API int msl_sf_set_stream(msl_sf *h, void *hip_stream) MSL_NOEXCEPT;
/* ONE upload of the gray image for both consumers (round 6).  Tracking::GrabImageRGBD hands the same gray frame to the ORB extractor (Frame
 * constructor, src/Frame.cc:103) and, for keyframes, to the surfel fusion (src/SurfelMapping.cpp:160-166); with host images each handle would copy it
 * over PCIe.  After msl_sf_fuse_resident_batch[_d16] with MSL_MEM_HOST images, msl_sf_staged_gray returns the device address of the gray images that
 * call staged (frame f at *gray_dev + f * *frame_stride, rows *row_stride bytes apart) and a hipEvent_t, owned by the handle, that completes when
 * their copy has.  msl_orb_wait_event makes the extractor's stream wait for such an event; msl_orb_extract_batch(..., MSL_MEM_DEVICE, ...) then reads
 * the images in place.  The staged images stay valid until the SECOND next host-image batch is enqueued on the surfel handle: the caller lets the
 * extraction of batch k return before it enqueues batch k + 2 (msl_orb_extract_batch with host outputs is synchronous).  MSL_ERR_INVALID when the
 * handle's last batch had no host images. */
MSL_API int msl_sf_staged_gray(msl_sf *h, const uint8_t **gray_dev, size_t *row_stride, size_t *frame_stride, void **uploaded_event) MSL_NOEXCEPT;


#ifdef __cplusplus
}
#endif
#endif /* MSL_H */
