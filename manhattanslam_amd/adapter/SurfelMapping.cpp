// Adapter: ORB_SLAM2::SurfelMapping with the local surfel map resident in HBM (replaces the reference's
// src/SurfelMapping.cpp).  Host bookkeeping (pose graph, inactive cloud) follows the reference's behaviour; every walk over
// the local surfels is a C-ABI call on the device-resident map -- see SurfelMapping.h.
#include "SurfelMapping.h"

#include <algorithm>
#include <stdexcept>

namespace ORB_SLAM2 {

namespace {
// Device-side errors (capacity, internal bounds) are deferred to the next sync; poll them often enough that the
// keyframe that caused one can still be named.
const int kSyncEveryKeyframes = 16;

void throwOn(int rc, const char *what) {
    if (rc != MSL_OK) throw std::runtime_error(std::string(what) + ": " + msl_last_error());
}

pcl::PointSurfel toPointSurfel(const Surfel &s) {   // the per-surfel conversion of Stop() (src/SurfelMapping.cpp:69-83, 89-103)
    pcl::PointSurfel p;
    p.x = s.px; p.y = s.py; p.z = s.pz;
    p.r = s.r; p.g = s.g; p.b = s.b;
    p.normal_x = s.nx; p.normal_y = s.ny; p.normal_z = s.nz;
    p.radius = s.size * 1000;
    p.confidence = s.weight;
    return p;
}
}  // namespace

SurfelMapping::SurfelMapping(Map *map, const std::string &strSettingPath)
    : mbStop(false), mMap(map), mSurfelFusion(nullptr), driftFreePoses(10), mKeyframesSinceSync(0) {
    cv::FileStorage fSettings(strSettingPath, cv::FileStorage::READ);   // same keys as src/SurfelMapping.cpp:30-41
    const float fx = fSettings["Camera.fx"], fy = fSettings["Camera.fy"], cx = fSettings["Camera.cx"], cy = fSettings["Camera.cy"];
    const int imgWidth = fSettings["Camera.width"], imgHeight = fSettings["Camera.height"];
    const float distanceFar = fSettings["Surfel.distanceFar"], distanceNear = fSettings["Surfel.distanceNear"];
    mSurfelFusion = new SurfelFusion(imgWidth, imgHeight, fx, fy, cx, cy, distanceFar, distanceNear);
    // a map that already holds local surfels (System reloaded into an existing Map) becomes the resident map
    if (!mMap->mvLocalSurfels.empty()) mSurfelFusion->uploadMap(mMap->mvLocalSurfels);
}

void SurfelMapping::Run() {
    // same loop as the reference (src/SurfelMapping.cpp:46-60): drain the queue until Stop() raises the flag
    for (;;) {
        if (CheckNewKeyFrames()) ProcessNewKeyFrame();
        std::unique_lock<std::mutex> lock(mMutexStop);
        if (mbStop) { mbStop = false; return; }
    }
}

pcl::PointCloud<pcl::PointSurfel>::Ptr SurfelMapping::Stop() {
    std::unique_lock<std::mutex> lock(mMutexStop);
    mbStop = true;
    pcl::PointCloud<pcl::PointSurfel>::Ptr pointCloud(new pcl::PointCloud<pcl::PointSurfel>());
    // local surfels seen at least five times, in map order (:67-84): filtered on the device, only the survivors cross PCIe
    size_t n = 0;
    int rc = msl_sf_map_export(mSurfelFusion->handle(), 5, nullptr, 0, &n);
    if (rc != MSL_OK && rc != MSL_ERR_CAPACITY) throwOn(rc, "msl_sf_map_export");
    mScratch.resize(n);
    if (n) throwOn(msl_sf_map_export(mSurfelFusion->handle(), 5, reinterpret_cast<msl_surfel *>(mScratch.data()), n, &n), "msl_sf_map_export");
    for (size_t i = 0; i < n; i++) pointCloud->push_back(toPointSurfel(mScratch[i]));
    // every inactive surfel (:86-104)
    for (const Surfel &s : mMap->mvInactiveSurfels) pointCloud->push_back(toPointSurfel(s));
    // plane points of the map planes with the plane normal (:106-130)
    const double radius = 0.1414 * 1000;
    for (auto pMP : mMap->GetAllMapPlanes()) {
        const cv::Mat P3Dw = pMP->GetWorldPos();
        for (auto &planePoint : pMP->mvPlanePoints->points) {
            pcl::PointSurfel p;
            p.x = planePoint.x; p.y = planePoint.y; p.z = planePoint.z;
            p.r = planePoint.r; p.g = planePoint.g; p.b = planePoint.b;
            p.normal_x = P3Dw.at<float>(0); p.normal_y = P3Dw.at<float>(1); p.normal_z = P3Dw.at<float>(2);
            p.radius = radius;
            p.confidence = 1;
            pointCloud->push_back(p);
        }
    }
    return pointCloud;
}

void SurfelMapping::InsertKeyFrame(const cv::Mat &imRGB, const cv::Mat &imDepth, const cv::Mat planeMembershipImg,
                                   const cv::Mat &pose, const int referenceIndex) {
    std::unique_lock<std::mutex> lock(mMutexNewKFs);
    mlNewKeyFrames.emplace_back(imRGB, imDepth, planeMembershipImg, pose, referenceIndex);
}

bool SurfelMapping::CheckNewKeyFrames() {
    std::unique_lock<std::mutex> lock(mMutexNewKFs);
    return !mlNewKeyFrames.empty();
}

void SurfelMapping::ProcessNewKeyFrame() {
    std::tuple<cv::Mat, cv::Mat, cv::Mat, cv::Mat, int> frame;
    {
        std::unique_lock<std::mutex> lock(mMutexNewKFs);
        frame = mlNewKeyFrames.front();
        mlNewKeyFrames.pop_front();
    }
    const cv::Mat &pose = std::get<3>(frame);
    const int relativeIndex = std::get<4>(frame);

    // register the keyframe in the pose graph (:161-169)
    const int index = (int)posesDatabase.size();
    PoseElement poseElement;
    if (!posesDatabase.empty()) {
        poseElement.linkedPoseIndex.push_back(relativeIndex);
        posesDatabase[relativeIndex].linkedPoseIndex.push_back(index);
    }
    posesDatabase.push_back(poseElement);
    localSurfelsIndexs.insert(index);

    moveAddSurfels(relativeIndex);

    Eigen::Matrix4f poseEigen = Eigen::Matrix4f::Zero();   // element-wise CV_32F 4x4 -> column-major Eigen (:173-189)
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) poseEigen(r, c) = pose.at<float>(r, c);
    fuseMap(std::get<0>(frame), std::get<1>(frame), std::get<2>(frame), poseEigen, relativeIndex);
}

// One pose leaves the local window (:204-226): its live surfels move, in map order, from the resident map to the pose's
// parking vector and to the end of the inactive cloud; in the map they are marked deleted (the next fuseMap refills the slots).
void SurfelMapping::detachPose(int poseIndex) {
    PoseElement &pe = posesDatabase[poseIndex];
    pe.pointsBeginIndex = (int)mMap->mvInactiveSurfels.size();
    pe.pointsPoseIndex = (int)pointcloudPoseIndex.size();
    pointcloudPoseIndex.push_back(poseIndex);
    size_t n = 0;
    int rc = msl_sf_map_detach(mSurfelFusion->handle(), poseIndex, nullptr, 0, &n);   // count only: nothing is modified on MSL_ERR_CAPACITY
    if (rc != MSL_OK && rc != MSL_ERR_CAPACITY) throwOn(rc, "msl_sf_map_detach");
    if (n) {
        mScratch.resize(n);
        throwOn(msl_sf_map_detach(mSurfelFusion->handle(), poseIndex, reinterpret_cast<msl_surfel *>(mScratch.data()), n, &n), "msl_sf_map_detach");
        pe.attachedSurfels.insert(pe.attachedSurfels.end(), mScratch.begin(), mScratch.begin() + n);
        mMap->mvInactiveSurfels.insert(mMap->mvInactiveSurfels.end(), mScratch.begin(), mScratch.begin() + n);
    }
    localSurfelsIndexs.erase(poseIndex);
}

// Poses re-enter the local window (:233-289): their parked surfels leave the inactive cloud.  Runs of poses that are
// adjacent in pointcloudPoseIndex are erased as one range; every later pose's offsets shift down.
void SurfelMapping::unparkPoses(const std::vector<int> &poses) {
    std::vector<std::pair<int, int>> byCloudPos;   // (position in pointcloudPoseIndex, pose)
    for (int p : poses) byCloudPos.emplace_back(posesDatabase[p].pointsPoseIndex, p);
    std::sort(byCloudPos.begin(), byCloudPos.end(),
              [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    size_t runBegin = 0;
    while (runBegin < byCloudPos.size()) {
        size_t runEnd = runBegin + 1;   // one past the last pose of this run of consecutive cloud positions
        while (runEnd < byCloudPos.size() && byCloudPos[runEnd].first == byCloudPos[runEnd - 1].first + 1) runEnd++;
        const int firstPose = byCloudPos[runBegin].second, lastPose = byCloudPos[runEnd - 1].second;
        int points = 0;
        for (size_t i = runBegin; i < runEnd; i++) points += (int)posesDatabase[byCloudPos[i].second].attachedSurfels.size();
        const int poseCount = (int)(runEnd - runBegin);
        auto first = mMap->mvInactiveSurfels.begin() + posesDatabase[firstPose].pointsBeginIndex;
        mMap->mvInactiveSurfels.erase(first, first + points);
        for (size_t pi = posesDatabase[lastPose].pointsPoseIndex + 1; pi < pointcloudPoseIndex.size(); pi++) {
            posesDatabase[pointcloudPoseIndex[pi]].pointsBeginIndex -= points;
            posesDatabase[pointcloudPoseIndex[pi]].pointsPoseIndex -= poseCount;
        }
        pointcloudPoseIndex.erase(pointcloudPoseIndex.begin() + posesDatabase[firstPose].pointsPoseIndex,
                                  pointcloudPoseIndex.begin() + posesDatabase[lastPose].pointsPoseIndex + 1);
        runBegin = runEnd;
    }
}

void SurfelMapping::moveAddSurfels(int referenceIndex) {
    std::vector<int> posesToAdd, posesToRemove;
    getAddRemovePoses(referenceIndex, posesToAdd, posesToRemove);
    for (int inactiveIndex : posesToRemove) detachPose(inactiveIndex);
    if (posesToAdd.empty()) return;
    localSurfelsIndexs.insert(posesToAdd.begin(), posesToAdd.end());
    unparkPoses(posesToAdd);
    for (int poseIndex : posesToAdd) {   // parked surfels are appended to the resident map in posesToAdd order (:291-302)
        PoseElement &pe = posesDatabase[poseIndex];
        if (!pe.attachedSurfels.empty())
            throwOn(msl_sf_map_append(mSurfelFusion->handle(), reinterpret_cast<const msl_surfel *>(pe.attachedSurfels.data()),
                                      pe.attachedSurfels.size()), "msl_sf_map_append");
        pe.attachedSurfels.clear();
        pe.pointsBeginIndex = -1;
        pe.pointsPoseIndex = -1;
    }
}

void SurfelMapping::getAddRemovePoses(int rootIndex, std::vector<int> &poseToAdd, std::vector<int> &poseToRemove) {
    std::vector<int> window;
    getDriftfreePoses(rootIndex, window, driftFreePoses);
    poseToAdd.clear();
    poseToRemove.clear();
    for (int p : window)                 // in BFS order (:311-315)
        if (!localSurfelsIndexs.count(p)) poseToAdd.push_back(p);
    for (int p : localSurfelsIndexs)     // ascending pose index (:317-322)
        if (std::find(window.begin(), window.end(), p) == window.end()) poseToRemove.push_back(p);
}

// breadth-first over the pose links, root included, driftfreeRange - 1 hops (:326-351)
void SurfelMapping::getDriftfreePoses(int rootIndex, std::vector<int> &driftfreePoses, int driftfreeRange) {
    if ((int)posesDatabase.size() < rootIndex + 1) return;
    std::vector<int> frontier(1, rootIndex), next;
    driftfreePoses.push_back(rootIndex);
    for (int hop = 1; hop < driftfreeRange; hop++) {
        for (int p : frontier)
            for (int q : posesDatabase[p].linkedPoseIndex)
                if (std::find(driftfreePoses.begin(), driftfreePoses.end(), q) == driftfreePoses.end()) {
                    next.push_back(q);
                    driftfreePoses.push_back(q);
                }
        frontier.swap(next);
        next.clear();
    }
}

void SurfelMapping::fuseMap(cv::Mat image, cv::Mat depth, cv::Mat planeMembershipImg, Eigen::Matrix4f poseInput, int referenceIndex) {
    // fuseInitializeMap + deleted-slot refill + tail compaction (:353-392), all on the resident map, asynchronous
    mSurfelFusion->fuseMapResident(referenceIndex, image, depth, planeMembershipImg, poseInput);
    if (++mKeyframesSinceSync >= kSyncEveryKeyframes) {   // surface deferred device-side errors while the keyframe is still known
        mKeyframesSinceSync = 0;
        throwOn(msl_sf_sync(mSurfelFusion->handle()), "msl_sf_sync");
    }
}

void SurfelMapping::SyncLocalSurfelsToHost() { mSurfelFusion->downloadMap(mMap->mvLocalSurfels); }

}  // namespace ORB_SLAM2
