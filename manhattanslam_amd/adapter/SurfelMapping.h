/**
 * Resident-map replacement for ManhattanSLAM's include/SurfelMapping.h (reference include/SurfelMapping.h:34-95).
 *
 * Same namespace, class name and public interface (constructor, Run, Stop, InsertKeyFrame -- the members Tracking and
 * System call, src/Tracking.cc:227,496, src/System.cc:100-103,296-382), so the rest of the reference compiles unchanged.
 * What differs is where the local surfels live: in HBM, inside the msl_sf handle of SurfelFusion (adapter/SurfelFusion.h).
 * Per keyframe nothing but the images (1.8 MB) crosses PCIe; the three places the reference walks mMap->mvLocalSurfels
 * become C-ABI calls on the resident map:
 *
 *   fuseMap             (src/SurfelMapping.cpp:353-392)   -> SurfelFusion::fuseMapResident  (fuse + slot refill + tail compaction)
 *   moveAddSurfels      (:194-304) per leaving pose       -> msl_sf_map_detach              (ordered copy-out + updateTimes = 0)
 *                                  per re-entering pose   -> msl_sf_map_append
 *   Stop                (:62-84)   updateTimes >= 5       -> msl_sf_map_export
 *
 * mMap->mvInactiveSurfels, the pose graph and the inactive-cloud index bookkeeping stay host data, as in the reference.
 * mMap->mvLocalSurfels is refreshed from HBM only on request (SyncLocalSurfelsToHost, e.g. before MapDrawer::DrawSurfels).
 * Build with SurfelMapping.cpp INSTEAD of the reference's src/SurfelMapping.cpp.
 */
#ifndef SURFELMAPPING_H
#define SURFELMAPPING_H

#include <list>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "System.h"
#include "Map.h"
#include "Surfel.h"
#include "SurfelFusion.h"
#include <pcl/point_types.h>

namespace ORB_SLAM2 {
    typedef pcl::PointXYZRGB PointType;
    typedef pcl::PointCloud<PointType> PointCloud;

    // one entry per keyframe handed to the mapper (reference include/SurfelMapping.h:39-46)
    struct PoseElement {
        std::vector<Surfel> attachedSurfels;   // surfels parked with this pose while it is outside the local window
        std::vector<int> linkedPoseIndex;      // undirected pose-graph edges
        int pointsBeginIndex;                  // offset of attachedSurfels inside Map::mvInactiveSurfels, -1 while local
        int pointsPoseIndex;                   // position in pointcloudPoseIndex, -1 while local

        PoseElement() : pointsBeginIndex(-1), pointsPoseIndex(-1) {}
    };

    class SurfelMapping {
    public:
        SurfelMapping(Map *map, const std::string &strSettingsFile);

        void Run();

        pcl::PointCloud<pcl::PointSurfel>::Ptr Stop();

        void InsertKeyFrame(const cv::Mat &imRGB, const cv::Mat &imDepth, const cv::Mat planeMembershipImg,
                            const cv::Mat &pose, const int referenceIndex);

        // Not in the reference: copy the resident local surfels into mMap->mvLocalSurfels (viewer / debugging).
        void SyncLocalSurfelsToHost();

    protected:
        bool CheckNewKeyFrames();

        void ProcessNewKeyFrame();

        void moveAddSurfels(int referenceIndex);

        void getAddRemovePoses(int rootIndex, std::vector<int> &poseToAdd, std::vector<int> &poseToRemove);

        void getDriftfreePoses(int rootIndex, std::vector<int> &driftfreePoses, int driftfreeRange);

        void fuseMap(cv::Mat image, cv::Mat depth, cv::Mat planeMembershipImg, Eigen::Matrix4f poseInput,
                     int referenceIndex);

        std::list<std::tuple<cv::Mat, cv::Mat, cv::Mat, cv::Mat, int>> mlNewKeyFrames;

        std::mutex mMutexNewKFs;
        bool mbStop;
        std::mutex mMutexStop;

        Map *mMap;

        SurfelFusion *mSurfelFusion;

        std::vector<PoseElement> posesDatabase;
        std::set<int> localSurfelsIndexs;
        int driftFreePoses;

        std::vector<int> pointcloudPoseIndex;

    private:
        void detachPose(int poseIndex);                               // one leaving pose: resident map -> host parking
        void unparkPoses(const std::vector<int> &poses);              // re-entering poses: inactive-cloud bookkeeping
        std::vector<Surfel> mScratch;                                 // staging for detach / export
        int mKeyframesSinceSync;
    };
}

#endif //SURFELMAPPING_H
