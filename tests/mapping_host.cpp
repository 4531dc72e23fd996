// Test host for adapter/SurfelMapping.cpp + adapter/SurfelFusion.cpp (tests/test_mapping_gpu.py): the two adapter translation units are
// compiled against tests/stubs/, linked with libmsl.so and driven keyframe by keyframe through the reference's own entry points
// (InsertKeyFrame -> CheckNewKeyFrames -> ProcessNewKeyFrame, src/SurfelMapping.cpp:137-192); after every keyframe the test reads the
// complete host bookkeeping back and compares it with the oracle's restatement of src/SurfelMapping.cpp:148-392.
#include <cstring>
#include <exception>
#include <string>

#include "SurfelMapping.h"

namespace {
struct Mapper : ORB_SLAM2::SurfelMapping {   // the bookkeeping members are protected in the reference's class: a test subclass looks at them
    Mapper(ORB_SLAM2::Map *m, const std::string &settings) : ORB_SLAM2::SurfelMapping(m, settings), map(m) {}
    bool step() {
        if (!CheckNewKeyFrames()) return false;
        ProcessNewKeyFrame();
        return true;
    }
    const std::vector<ORB_SLAM2::PoseElement> &poses() const { return posesDatabase; }
    const std::vector<int> &cloudIndex() const { return pointcloudPoseIndex; }
    const std::set<int> &localIndexs() const { return localSurfelsIndexs; }
    ORB_SLAM2::Map *map;
};
struct Host {
    ORB_SLAM2::Map map;
    Mapper *mapper = nullptr;
    int w = 0, h = 0;
    std::string error;
};
}  // namespace

extern "C" {
#define MH_API __attribute__((visibility("default")))

MH_API void *mh_create(int w, int h, float fx, float fy, float cx, float cy, float far_, float near_) {
    Host *H = new Host;
    H->w = w; H->h = h;
    auto &S = cv::FileStorage::registry()["settings.yaml"];   // the keys src/SurfelMapping.cpp:30-41 reads
    S["Camera.fx"] = fx; S["Camera.fy"] = fy; S["Camera.cx"] = cx; S["Camera.cy"] = cy;
    S["Camera.width"] = w; S["Camera.height"] = h; S["Surfel.distanceFar"] = far_; S["Surfel.distanceNear"] = near_;
    try {
        H->mapper = new Mapper(&H->map, "settings.yaml");
    } catch (const std::exception &e) {
        H->error = e.what();
    }
    return H;
}
MH_API const char *mh_error(void *p) { return static_cast<Host *>(p)->error.c_str(); }
MH_API void mh_destroy(void *p) {
    Host *H = static_cast<Host *>(p);
    delete H->mapper;
    delete H;
}
// one keyframe through InsertKeyFrame + ProcessNewKeyFrame; pose = the CV_32F 4x4 Twc (row-major cv::Mat); 0 = ok
MH_API int mh_keyframe(void *p, const unsigned char *gray, const float *depth, const int *member, const float *pose16, int referenceIndex) {
    Host *H = static_cast<Host *>(p);
    try {
        cv::Mat im(H->h, H->w, CV_8UC1, (void *)gray), dp(H->h, H->w, CV_32FC1, (void *)depth), mb(H->h / 2, H->w / 2, CV_32SC1, (void *)member);
        cv::Mat pose(4, 4, CV_32F);
        std::memcpy(pose.data, pose16, sizeof(float) * 16);
        H->mapper->InsertKeyFrame(im, dp, mb, pose, referenceIndex);
        return H->mapper->step() ? 0 : -2;
    } catch (const std::exception &e) {
        H->error = e.what();
        return -1;
    }
}
MH_API size_t mh_local(void *p, Surfel *out, size_t cap) {   // SyncLocalSurfelsToHost, then mMap->mvLocalSurfels
    Host *H = static_cast<Host *>(p);
    H->mapper->SyncLocalSurfelsToHost();
    const size_t n = H->map.mvLocalSurfels.size();
    if (out && n <= cap && n) std::memcpy(out, H->map.mvLocalSurfels.data(), n * sizeof(Surfel));
    return n;
}
MH_API size_t mh_inactive(void *p, Surfel *out, size_t cap) {
    Host *H = static_cast<Host *>(p);
    const size_t n = H->map.mvInactiveSurfels.size();
    if (out && n <= cap && n) std::memcpy(out, H->map.mvInactiveSurfels.data(), n * sizeof(Surfel));
    return n;
}
MH_API int mh_poses(void *p) { return (int)static_cast<Host *>(p)->mapper->poses().size(); }
MH_API void mh_pose(void *p, int i, int info[4]) {
    const ORB_SLAM2::PoseElement &e = static_cast<Host *>(p)->mapper->poses()[i];
    info[0] = e.pointsBeginIndex; info[1] = e.pointsPoseIndex; info[2] = (int)e.attachedSurfels.size(); info[3] = (int)e.linkedPoseIndex.size();
}
MH_API void mh_pose_data(void *p, int i, Surfel *attached, int *links) {
    const ORB_SLAM2::PoseElement &e = static_cast<Host *>(p)->mapper->poses()[i];
    if (attached && !e.attachedSurfels.empty()) std::memcpy(attached, e.attachedSurfels.data(), e.attachedSurfels.size() * sizeof(Surfel));
    if (links) for (size_t k = 0; k < e.linkedPoseIndex.size(); k++) links[k] = e.linkedPoseIndex[k];
}
MH_API size_t mh_cloud_index(void *p, int *out, size_t cap) {
    const std::vector<int> &v = static_cast<Host *>(p)->mapper->cloudIndex();
    if (out && v.size() <= cap) for (size_t k = 0; k < v.size(); k++) out[k] = v[k];
    return v.size();
}
MH_API size_t mh_local_indexs(void *p, int *out, size_t cap) {
    const std::set<int> &v = static_cast<Host *>(p)->mapper->localIndexs();
    size_t k = 0;
    if (out && v.size() <= cap) for (int x : v) out[k++] = x;
    return v.size();
}
// SurfelMapping::Stop(): the exported cloud as (x, y, z, nx, ny, nz, r, g, b, radius, confidence) rows of 11 floats
MH_API size_t mh_stop(void *p, float *out, size_t cap_points) {
    Host *H = static_cast<Host *>(p);
    auto cloud = H->mapper->Stop();
    const size_t n = cloud->points.size();
    if (out && n <= cap_points)
        for (size_t k = 0; k < n; k++) {
            const pcl::PointSurfel &q = cloud->points[k];
            const float row[11] = {q.x, q.y, q.z, q.normal_x, q.normal_y, q.normal_z, (float)q.r, (float)q.g, (float)q.b, q.radius, q.confidence};
            std::memcpy(out + 11 * k, row, sizeof(row));
        }
    return n;
}
}  // extern "C"
