// Drop-in replacement for the reference's src/PlaneExtractor.cpp on top of libmsl.so (see PlaneExtractor.h).
// readDepthImage only records its inputs and sizes the public cloud; runPlaneDetection hands the raw depth image to msl_peac_extract_batch, which
// returns plane_filter.run's outputs -- the membership image, the extracted planes, the per-plane vertex lists -- AND the organised
// half-resolution cloud (cloud.vertices, src/PlaneExtractor.cpp:60-74) the device computes for the block fit anyway; only the vertex colours,
// a strided copy of the colour image, are gathered on the host (Frame::ExtractPlanes reads both, src/Frame.cc:612-622).
#include "PlaneExtractor.h"

#include <cmath>
#include <iostream>
#include <stdexcept>
#include <string>

PlaneDetection::PlaneDetection() : plane_num_(0) {}

PlaneDetection::~PlaneDetection() {
    cloud.vertices.clear();
    seg_img_.release();
    color_img_.release();
}

bool PlaneDetection::readColorImage(cv::Mat RGBImg) {   // src/PlaneExtractor.cpp:33-41
    color_img_ = RGBImg;
    if (color_img_.empty() || color_img_.depth() != CV_8U) {
        std::cout << "ERROR: cannot read color image. No such a file, or the image format is not 8UC3" << std::endl;
        return false;
    }
    return true;
}

bool PlaneDetection::readDepthImage(const cv::Mat depthImg, const cv::Mat &K, const float &depthMapFactor) {   // src/PlaneExtractor.cpp:44-75
    if (depthImg.empty() || depthImg.depth() != CV_16U) {
        std::cout << "WARNING: cannot read depth image. No such a file, or the image format is not 16UC1" << std::endl;
        return false;
    }
    const int width = (int)std::ceil(depthImg.cols / 2.0), height = (int)std::ceil(depthImg.rows / 2.0);
    cloud.vertices.resize((size_t)height * width);
    cloud.verticesColour.resize((size_t)height * width);
    cloud.w = width;
    cloud.h = height;
    seg_img_ = cv::Mat(height, width, CV_8UC3);
    fx_ = K.at<float>(0, 0); fy_ = K.at<float>(1, 1); cx_ = K.at<float>(0, 2); cy_ = K.at<float>(1, 2);
    depthMapFactor_ = depthMapFactor;
    depth16_ = depthImg;
    // The public cloud is filled here as in the reference (:60-74), so a caller that reads cloud.vertices right after readDepthImage -- or never
    // calls runPlaneDetection -- sees the vertices of THIS depth image.  80 k vertices of three double operations each: ~0.1 ms
    // of host time; runPlaneDetection overwrites them with the device's (bit-identical) values.
    int vertex_idx = 0;
    for (int i = 0; i < depthImg.rows; i += 2)
        for (int j = 0; j < depthImg.cols; j += 2) {
            const double z = (double)(depthImg.at<unsigned short>(i, j)) * depthMapFactor;
            if (z != z) { cloud.vertices[vertex_idx++] = VertexType(0, 0, z); continue; }     // _isnan(z), :63-66
            const double x = ((double)j - K.at<float>(0, 2)) * z / K.at<float>(0, 0);
            const double y = ((double)i - K.at<float>(1, 2)) * z / K.at<float>(1, 1);
            const cv::Vec3b c = color_img_.at<cv::Vec3b>(i, j);
            cloud.verticesColour[vertex_idx] = VertexColour(c[0], c[1], c[2]);
            cloud.vertices[vertex_idx++] = VertexType(x, y, z);
        }
    return true;
}

void PlaneDetection::runPlaneDetection() {   // src/PlaneExtractor.cpp:77-80: plane_filter.run(&cloud, &plane_vertices_, &seg_img_)
    const int cw = cloud.w, ch = cloud.h, maxPlanes = 256;
    plane_filter.membershipImg = cv::Mat(ch, cw, CV_32SC1);
    std::vector<msl_peac_plane> planes(maxPlanes);
    std::vector<int32_t> offsets(maxPlanes + 1), indices((size_t)cw * ch);
    std::vector<double> xyz((size_t)cw * ch * 3);   // cloud.vertices as the device computed them
    int32_t nPlanes = 0;
    const int rc = msl_peac_extract_batch(device_, depth16_.ptr<uint16_t>(), depth16_.step, 0, depth16_.cols, depth16_.rows, 1, MSL_MEM_HOST, fx_, fy_, cx_, cy_,
                                          depthMapFactor_, &plane_filter.params, plane_filter.membershipImg.ptr<int32_t>(), &nPlanes, maxPlanes, planes.data(),
                                          offsets.data(), indices.data(), xyz.data());
    if (rc != MSL_OK) throw std::runtime_error(std::string("msl_peac_extract_batch: ") + msl_last_error());
    for (size_t v = 0; v < (size_t)cw * ch; v++) cloud.vertices[v] = VertexType(xyz[3 * v], xyz[3 * v + 1], xyz[3 * v + 2]);
    plane_filter.extractedPlanes.clear();
    plane_vertices_.assign((size_t)nPlanes, std::vector<int>());
    for (int i = 0; i < nPlanes; i++) {
        ahc::PlaneSeg::shared_ptr ps(new ahc::PlaneSeg);
        for (int c = 0; c < 3; c++) { ps->normal[c] = planes[i].normal[c]; ps->center[c] = planes[i].center[c]; }
        ps->mse = planes[i].mse; ps->N = planes[i].N;
        plane_filter.extractedPlanes.push_back(ps);
        plane_vertices_[i].assign(indices.begin() + offsets[i], indices.begin() + offsets[i + 1]);
    }
    plane_num_ = (int)plane_vertices_.size();
}
