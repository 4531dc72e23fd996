// CPU oracle (TEST INFRASTRUCTURE ONLY -- never linked into libmsl.so) for SURVEY.md 8(f) rank 2: literal restatement of
//   PlaneDetection::readDepthImage   /root/reference/src/PlaneExtractor.cpp:44-76
//   ImagePointCloud::get             /root/reference/include/PlaneExtractor.h:47-55
//   ahc::depthDisContinuous          /root/reference/include/peac/AHCPlaneSeg.hpp:41-43, ParamSet::T_dz AHCParamSet.hpp:140-142
//   ahc::PlaneSeg::PlaneSeg (init)   /root/reference/include/peac/AHCPlaneSeg.hpp:237-285, Stats::push :81-92
//   block loop of initGraph          /root/reference/include/peac/AHCPlaneFitter.hpp:756-776
// Parity unpinned: PEAC needs OpenCV / Eigen / boost and cannot be built in this image.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/msl.h"

#define MSLO_API extern "C" __attribute__((visibility("default")))

namespace {
struct Cloud {
    std::vector<double> v;   // x y z per vertex
    int w, h;
    bool get(int row, int col, double &x, double &y, double &z) const {
        const int pixIdx = row * w + col;
        z = v[3 * (size_t)pixIdx + 2];
        if (z == 0 || std::isnan(z)) return false;
        x = v[3 * (size_t)pixIdx]; y = v[3 * (size_t)pixIdx + 1];
        return true;
    }
};
}  // namespace

MSLO_API void mslo_peac_block_stats(const uint16_t *depth, size_t strideBytes, int cols, int rows, float fx, float fy, float cx, float cy,
                                    float depthMapFactor, int winWidth, int winHeight, double depthAlpha, double depthChangeTol, int initLoose,
                                    double *cloudOut, msl_peac_stats *statsOut) {
    Cloud cloud;
    const double width = std::ceil(cols / 2.0), height = std::ceil(rows / 2.0);
    cloud.w = (int)width; cloud.h = (int)height;
    cloud.v.assign((size_t)(height * width) * 3, 0.0);
    int vertex_idx = 0;
    for (int i = 0; i < rows; i += 2)
        for (int j = 0; j < cols; j += 2) {
            const uint16_t d = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(depth) + (size_t)i * strideBytes + 2 * (size_t)j);
            double z = (double)d * depthMapFactor;
            double x = ((double)j - cx) * z / fx;
            double y = ((double)i - cy) * z / fy;
            cloud.v[3 * (size_t)vertex_idx] = x; cloud.v[3 * (size_t)vertex_idx + 1] = y; cloud.v[3 * (size_t)vertex_idx + 2] = z;
            vertex_idx++;
        }
    if (cloudOut) for (size_t k = 0; k < cloud.v.size(); k++) cloudOut[k] = cloud.v[k];
    const int imgWidth = cloud.w, imgHeight = cloud.h;
    const int Nh = imgHeight / winHeight, Nw = imgWidth / winWidth;
    auto discontinuous = [&](double d0, double d1) { return std::fabs(d0 - d1) > depthAlpha * std::fabs(d0) + depthChangeTol; };
    for (int bi = 0; bi < Nh; ++bi)
        for (int bj = 0; bj < Nw; ++bj) {
            const int seed_row = bi * winHeight, seed_col = bj * winWidth;
            msl_peac_stats S{};
            bool windowValid = true;
            int nanCnt = 0, nanCntTh = winHeight * winWidth / 2;
            for (int i = seed_row, icnt = 0; icnt < winHeight && i < imgHeight; ++i, ++icnt) {
                for (int j = seed_col, jcnt = 0; jcnt < winWidth && j < imgWidth; ++j, ++jcnt) {
                    double x = 0, y = 0, z = 10000;
                    if (!cloud.get(i, j, x, y, z)) {
                        if (initLoose) {
                            ++nanCnt;
                            if (nanCnt < nanCntTh) continue;
                        }
                        windowValid = false;
                        break;
                    }
                    double xn = 0, yn = 0, zn = 10000;
                    if (j + 1 < imgWidth && (cloud.get(i, j + 1, xn, yn, zn) && discontinuous(z, zn))) { windowValid = false; break; }
                    if (i + 1 < imgHeight && (cloud.get(i + 1, j, xn, yn, zn) && discontinuous(z, zn))) { windowValid = false; break; }
                    S.sx += x; S.sy += y; S.sz += z;
                    S.sxx += x * x; S.syy += y * y; S.szz += z * z;
                    S.sxy += x * y; S.syz += y * z; S.sxz += x * z;
                    ++S.N;
                }
                if (!windowValid) break;
            }
            if (!windowValid) { S = msl_peac_stats{}; S.nouse = 1; }
            statsOut[bi * Nw + bj] = S;
        }
}

// =====================================================================================================================
// The rest of the PEAC plane extractor: PCA plane fit per block, graph initialisation, agglomerative
// hierarchical clustering, block erosion + region growing, final merge, membership image.  Restates
//   ahc::PlaneSeg (Stats::compute, merge constructor, connect / disconnectAllNbs / mergeNbsFrom)   include/peac/AHCPlaneSeg.hpp:148-183, 299-437
//   ahc::ParamSet::T_mse / T_ang                                                                    include/peac/AHCParamSet.hpp:87-131
//   DisjointSet                                                                                     include/peac/DisjointSet.hpp
//   ahc::PlaneFitter::run / initGraph / ahCluster / findBlockMembership / floodFill / refineDetails include/peac/AHCPlaneFitter.hpp:218-262,
//                                                                                                   :756-928, :939-1143, :490-596, :422-471, :296-372
// with the defaults PlaneDetection uses (nothing in src/ overrides them): minSupport 3000, 10x10 windows, doRefine, ERODE_ALL_BORDER.
// Pinned where the reference depends on third parties or on heap addresses:
//   * LA::eig33sym = Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (include/peac/eig33sym.hpp:71-75): restated from the published
//     Eigen 3.3 algorithm -- scale by the largest coefficient, 3x3 Householder tridiagonalisation, implicit symmetric QR steps
//     with Wilkinson shift, eigenvalues sorted increasingly.  Eigen versions differ in the deflation test, so this is a pin.
//   * std::set<PlaneSeg *> iterates neighbours by heap address; pinned to creation order (only exact MSE ties can see it).
//   * std::priority_queue / std::sort are the C++ library's (same algorithms on both sides of the parity test).
#include <algorithm>
#include <map>
#include <memory>
#include <queue>
#include <set>

namespace {

double hypot_eigen(double x, double y) {   // Eigen::numext::hypot (positive_real_hypot)
    const double ax = std::fabs(x), ay = std::fabs(y);
    double p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0) return 0;
    return p * std::sqrt(1.0 + qp * qp);
}

// eigenvalues s[0] <= s[1] <= s[2], V[:][i] the eigenvector of s[i]
void eig33sym(const double K[3][3], double s[3], double V[3][3]) {
    // lower triangle, scaled into [-1, 1]
    double m[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[r][c] = c <= r ? K[r][c] : 0.0;
    double scale = 0;
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) scale = std::max(scale, std::fabs(m[r][c]));
    if (scale == 0) scale = 1;
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) m[r][c] /= scale;
    double diag[3], sub[2], Q[3][3];
    {   // tridiagonalization_inplace_selector<Matrix3d, 3, false>
        const double tol = std::numeric_limits<double>::min();
        diag[0] = m[0][0];
        const double v1norm2 = m[2][0] * m[2][0];
        if (v1norm2 <= tol) {
            diag[1] = m[1][1]; diag[2] = m[2][2]; sub[0] = m[1][0]; sub[1] = m[2][1];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Q[r][c] = r == c ? 1.0 : 0.0;
        } else {
            const double beta = std::sqrt(m[1][0] * m[1][0] + v1norm2);
            const double invBeta = 1.0 / beta;
            const double m01 = m[1][0] * invBeta, m02 = m[2][0] * invBeta;
            const double q = 2.0 * m01 * m[2][1] + m02 * (m[2][2] - m[1][1]);
            diag[1] = m[1][1] + m02 * q; diag[2] = m[2][2] - m02 * q;
            sub[0] = beta; sub[1] = m[2][1] - m01 * q;
            Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0; Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02; Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
        }
    }
    {   // computeFromTridiagonal_impl
        const int n = 3, maxIterations = 30;
        int end = n - 1, start = 0, iter = 0;
        const double considerAsZero = std::numeric_limits<double>::min();
        const double precision = 2.0 * std::numeric_limits<double>::epsilon();
        while (end > 0) {
            for (int i = start; i < end; ++i)
                if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(sub[i]) <= considerAsZero) sub[i] = 0;
            while (end > 0 && sub[end - 1] == 0.0) end--;
            if (end <= 0) break;
            iter++;
            if (iter > maxIterations * n) break;
            start = end - 1;
            while (start > 0 && sub[start - 1] != 0) start--;
            // tridiagonal_qr_step
            const double td = (diag[end - 1] - diag[end]) * 0.5;
            const double e = sub[end - 1];
            double mu = diag[end];
            if (td == 0.0) mu -= std::fabs(e);
            else if (e != 0.0) {
                const double e2 = e * e, h = hypot_eigen(td, e);
                if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
                else mu -= e2 / (td + (td > 0.0 ? h : -h));
            }
            double x = diag[start] - mu, z = sub[start];
            for (int k = start; k < end && z != 0.0; ++k) {
                double c, sn;   // JacobiRotation::makeGivens(x, z)
                if (z == 0.0) { c = x < 0.0 ? -1.0 : 1.0; sn = 0.0; }
                else if (x == 0.0) { c = 0.0; sn = z < 0.0 ? 1.0 : -1.0; }
                else if (std::fabs(x) > std::fabs(z)) { const double t = z / x; double u = std::sqrt(1.0 + t * t); if (x < 0.0) u = -u; c = 1.0 / u; sn = -t * c; }
                else { const double t = x / z; double u = std::sqrt(1.0 + t * t); if (z < 0.0) u = -u; sn = -1.0 / u; c = -t * sn; }
                const double sdk = sn * diag[k] + c * sub[k];
                const double dkp1 = sn * sub[k] + c * diag[k + 1];
                diag[k] = c * (c * diag[k] - sn * sub[k]) - sn * (c * sub[k] - sn * diag[k + 1]);
                diag[k + 1] = sn * sdk + c * dkp1;
                sub[k] = c * sdk - sn * dkp1;
                if (k > start) sub[k - 1] = c * sub[k - 1] - sn * z;
                x = sub[k];
                if (k < end - 1) { z = -sn * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
                for (int r = 0; r < 3; r++) {   // Q = Q * G
                    const double xi = Q[r][k], yi = Q[r][k + 1];
                    Q[r][k] = c * xi - sn * yi;
                    Q[r][k + 1] = sn * xi + c * yi;
                }
            }
        }
        for (int i = 0; i < n - 1; ++i) {   // sort increasingly
            int k = 0;
            for (int j = 1; j < n - i; j++) if (diag[i + j] < diag[i + k]) k = j;
            if (k > 0) { std::swap(diag[i], diag[k + i]); for (int r = 0; r < 3; r++) std::swap(Q[r][i], Q[r][k + i]); }
        }
    }
    for (int i = 0; i < 3; i++) s[i] = diag[i] * scale;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) V[r][c] = Q[r][c];
}

struct ParamSet {
    double depthSigma, stdTol_init, stdTol_merge, z_near, z_far, angle_near, angle_far, similarityTh_merge, similarityTh_refine, depthAlpha, depthChangeTol;
    int initLoose;
    enum Phase { P_INIT = 0, P_MERGING = 1, P_REFINE = 2 };
    double T_mse(Phase phase, double z = 0) const {
        switch (phase) {
            case P_INIT: return std::pow(depthSigma * z * z + stdTol_init, 2);
            default: return std::pow(depthSigma * z * z + stdTol_merge, 2);
        }
    }
    double T_ang(Phase phase, double z = 0) const {
        switch (phase) {
            case P_INIT: {
                double clipped_z = z;
                clipped_z = std::max(clipped_z, z_near);
                clipped_z = std::min(clipped_z, z_far);
                const double factor = (angle_far - angle_near) / (z_far - z_near);
                return std::cos(factor * clipped_z + angle_near - factor * z_near);
            }
            case P_MERGING: return similarityTh_merge;
            default: return similarityTh_refine;
        }
    }
};

struct PlaneSeg;
struct ById { bool operator()(const PlaneSeg *a, const PlaneSeg *b) const; };
struct PlaneSeg {
    typedef PlaneSeg *Ptr;
    typedef std::shared_ptr<PlaneSeg> shared_ptr;
    msl_peac_stats stats;
    int id;            // creation order: stands in for the heap address in NbSet
    int rid;
    double mse, center[3], normal[3];
    int N;
    double curvature;
    bool nouse;
    typedef std::set<Ptr, ById> NbSet;
    NbSet nbs;

    static void compute(const msl_peac_stats &st, double center[3], double normal[3], double &mse, double &curvature) {   // Stats::compute
        const double sc = ((double)1.0) / st.N;
        center[0] = st.sx * sc; center[1] = st.sy * sc; center[2] = st.sz * sc;
        double K[3][3] = {{st.sxx - st.sx * st.sx * sc, st.sxy - st.sx * st.sy * sc, st.sxz - st.sx * st.sz * sc},
                          {0, st.syy - st.sy * st.sy * sc, st.syz - st.sy * st.sz * sc},
                          {0, 0, st.szz - st.sz * st.sz * sc}};
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double sv[3] = {0, 0, 0}, V[3][3] = {{0}};
        eig33sym(K, sv, V);
        if (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) { normal[0] = V[0][0]; normal[1] = V[1][0]; normal[2] = V[2][0]; }
        else { normal[0] = -V[0][0]; normal[1] = -V[1][0]; normal[2] = -V[2][0]; }
        mse = sv[0] * sc;
        curvature = sv[0] / (sv[0] + sv[1] + sv[2]);
    }
    PlaneSeg(int id_, int root_block_id, const msl_peac_stats &blockStats) : stats(blockStats), id(id_), rid(root_block_id) {   // init constructor, after the window loop
        center[0] = center[1] = center[2] = normal[0] = normal[1] = normal[2] = 0;
        nouse = blockStats.nouse != 0;
        N = nouse ? 0 : stats.N;
        if (N < 4) mse = curvature = std::numeric_limits<double>::quiet_NaN();
        else compute(stats, center, normal, mse, curvature);
    }
    PlaneSeg(int id_, const PlaneSeg &pa, const PlaneSeg &pb) : id(id_) {   // merge constructor
        stats.sx = pa.stats.sx + pb.stats.sx; stats.sy = pa.stats.sy + pb.stats.sy; stats.sz = pa.stats.sz + pb.stats.sz;
        stats.sxx = pa.stats.sxx + pb.stats.sxx; stats.syy = pa.stats.syy + pb.stats.syy; stats.szz = pa.stats.szz + pb.stats.szz;
        stats.sxy = pa.stats.sxy + pb.stats.sxy; stats.syz = pa.stats.syz + pb.stats.syz; stats.sxz = pa.stats.sxz + pb.stats.sxz;
        stats.N = pa.stats.N + pb.stats.N; stats.nouse = 0;
        nouse = false;
        rid = pa.N >= pb.N ? pa.rid : pb.rid;
        N = stats.N;
        compute(stats, center, normal, mse, curvature);
    }
    double normalSimilarity(const PlaneSeg &p) const { return std::abs(normal[0] * p.normal[0] + normal[1] * p.normal[1] + normal[2] * p.normal[2]); }
    double signedDist(const double pt[3]) const { return normal[0] * (pt[0] - center[0]) + normal[1] * (pt[1] - center[1]) + normal[2] * (pt[2] - center[2]); }
    void connect(Ptr p) { if (p) { nbs.insert(p); p->nbs.insert(this); } }
    void disconnectAllNbs() {
        for (NbSet::iterator itr = nbs.begin(); itr != nbs.end(); ++itr) (*itr)->nbs.erase(this);
        nbs.clear();
    }
};
bool ById::operator()(const PlaneSeg *a, const PlaneSeg *b) const { return a->id < b->id; }

struct DisjointSet {
    std::vector<int> parent_, size_;
    explicit DisjointSet(int n) { for (int i = 0; i < n; ++i) { parent_.push_back(i); size_.push_back(1); } }
    int getSetSize(int x) { return size_[Find(x)]; }
    int Union(int x, int y) {
        const int xRoot = Find(x), yRoot = Find(y);
        if (xRoot == yRoot) return xRoot;
        const int xRootSize = size_[xRoot], yRootSize = size_[yRoot];
        if (xRootSize < yRootSize) { parent_[xRoot] = yRoot; size_[yRoot] += size_[xRoot]; return yRoot; }
        parent_[yRoot] = xRoot; size_[xRoot] += size_[yRoot]; return xRoot;
    }
    int Find(int x) { if (parent_[x] != x) parent_[x] = Find(parent_[x]); return parent_[x]; }
};

struct PlaneFitter {
    struct PlaneSegSizeCmp { bool operator()(const PlaneSeg::shared_ptr &a, const PlaneSeg::shared_ptr &b) const { return b->N < a->N; } };
    struct PlaneSegMinMSECmp { bool operator()(const PlaneSeg::shared_ptr &a, const PlaneSeg::shared_ptr &b) const { return b->mse < a->mse; } };
    typedef std::priority_queue<PlaneSeg::shared_ptr, std::vector<PlaneSeg::shared_ptr>, PlaneSegMinMSECmp> PlaneSegMinMSEQueue;
    const Cloud *points;
    int width, height, maxStep, minSupport, windowWidth, windowHeight, erodeType;   // erodeType: 0 none, 1 segment borders, 2 all borders
    bool doRefine;
    ParamSet params;
    std::unique_ptr<DisjointSet> ds;
    std::vector<PlaneSeg::shared_ptr> extractedPlanes;
    std::vector<int> membershipImg;
    std::vector<std::vector<int>> pMembership;   // run(points, pMembership, pSeg): vertex indices per plane (doRefine only)
    std::map<int, int> rid2plid;
    std::vector<int> blkMap;
    std::vector<std::pair<int, int>> rfQueue;
    int nextId = 0;

    static int getValid4Neighbor(int i, int j, int H, int W, int nbs[4]) {
        const int id = i * W + j;
        int cnt = 0;
        if (j > 0) nbs[cnt++] = (id - 1);
        if (j < W - 1) nbs[cnt++] = (id + 1);
        if (i > 0) nbs[cnt++] = (id - W);
        if (i < H - 1) nbs[cnt++] = (id + W);
        return cnt;
    }
    int getBlockIdx(int pixX, int pixY) const {
        const int Nw = width / windowWidth, Nh = height / windowHeight;
        const int by = pixY / windowHeight, bx = pixX / windowWidth;
        return (by < Nh && bx < Nw) ? (by * Nw + bx) : -1;
    }
    void mergeNbsFrom(PlaneSeg &self, PlaneSeg &pa, PlaneSeg &pb) {
        ds->Union(pa.rid, pb.rid);
        self.nbs.insert(pa.nbs.begin(), pa.nbs.end());
        self.nbs.insert(pb.nbs.begin(), pb.nbs.end());
        self.nbs.erase(&pa);
        self.nbs.erase(&pb);
        pa.disconnectAllNbs();
        pb.disconnectAllNbs();
        for (PlaneSeg::NbSet::iterator itr = self.nbs.begin(); itr != self.nbs.end(); ++itr) (*itr)->nbs.insert(&self);
        pa.nouse = pb.nouse = true;
    }

    void initGraph(PlaneSegMinMSEQueue &minQ, const msl_peac_stats *blockStats, msl_peac_block *blocksOut) {
        const int Nh = height / windowHeight, Nw = width / windowWidth;
        std::vector<PlaneSeg::Ptr> G(Nh * Nw, 0);
        for (int i = 0; i < Nh; ++i)
            for (int j = 0; j < Nw; ++j) {
                PlaneSeg::shared_ptr p(new PlaneSeg(nextId++, i * Nw + j, blockStats[i * Nw + j]));
                if (blocksOut) {
                    msl_peac_block &B = blocksOut[i * Nw + j];
                    B.stats = blockStats[i * Nw + j];
                    for (int k = 0; k < 3; k++) { B.center[k] = p->center[k]; B.normal[k] = p->normal[k]; }
                    B.mse = p->mse; B.curvature = p->curvature;
                }
                if (p->mse < params.T_mse(ParamSet::P_INIT, p->center[2]) && !p->nouse) { G[i * Nw + j] = p.get(); minQ.push(p); }
                else G[i * Nw + j] = 0;
            }
        for (int i = 0; i < Nh; ++i)
            for (int j = 1; j < Nw; j += 2) {
                const int cidx = i * Nw + j;
                if (G[cidx - 1] == 0) { --j; continue; }
                if (G[cidx] == 0) continue;
                if (j < Nw - 1 && G[cidx + 1] == 0) { ++j; continue; }
                const double similarityTh = params.T_ang(ParamSet::P_INIT, G[cidx]->center[2]);
                if ((j < Nw - 1 && G[cidx - 1]->normalSimilarity(*G[cidx + 1]) >= similarityTh) ||
                    (j == Nw - 1 && G[cidx]->normalSimilarity(*G[cidx - 1]) >= similarityTh)) {
                    G[cidx]->connect(G[cidx - 1]);
                    if (j < Nw - 1) G[cidx]->connect(G[cidx + 1]);
                } else {
                    --j;
                }
            }
        for (int j = 0; j < Nw; ++j)
            for (int i = 1; i < Nh; i += 2) {
                const int cidx = i * Nw + j;
                if (G[cidx - Nw] == 0) { --i; continue; }
                if (G[cidx] == 0) continue;
                if (i < Nh - 1 && G[cidx + Nw] == 0) { ++i; continue; }
                const double similarityTh = params.T_ang(ParamSet::P_INIT, G[cidx]->center[2]);
                if ((i < Nh - 1 && G[cidx - Nw]->normalSimilarity(*G[cidx + Nw]) >= similarityTh) ||
                    (i == Nh - 1 && G[cidx]->normalSimilarity(*G[cidx - Nw]) >= similarityTh)) {
                    G[cidx]->connect(G[cidx - Nw]);
                    if (i < Nh - 1) G[cidx]->connect(G[cidx + Nw]);
                } else {
                    --i;
                }
            }
    }

    int ahCluster(PlaneSegMinMSEQueue &minQ) {
        int step = 0;
        while (!minQ.empty() && step <= maxStep) {
            PlaneSeg::shared_ptr p = minQ.top();
            minQ.pop();
            if (p->nouse) continue;
            PlaneSeg::shared_ptr cand_merge;
            PlaneSeg::Ptr cand_nb(0);
            for (PlaneSeg::NbSet::iterator itr = p->nbs.begin(); itr != p->nbs.end(); itr++) {
                PlaneSeg::Ptr nb = (*itr);
                if (p->normalSimilarity(*nb) < params.T_ang(ParamSet::P_MERGING, p->center[2])) continue;
                PlaneSeg::shared_ptr merge(new PlaneSeg(nextId++, *p, *nb));
                if (cand_merge == 0 || cand_merge->mse > merge->mse || (cand_merge->mse == merge->mse && cand_merge->N < merge->mse)) {
                    cand_merge = merge;
                    cand_nb = nb;
                }
            }
            if (cand_merge != 0 && cand_merge->mse < params.T_mse(ParamSet::P_MERGING, cand_merge->center[2])) {
                minQ.push(cand_merge);
                mergeNbsFrom(*cand_merge, *p, *cand_nb);
            } else {
                if (p->N >= minSupport) extractedPlanes.push_back(p);
                p->disconnectAllNbs();
            }
            ++step;
        }
        while (!minQ.empty()) {
            const PlaneSeg::shared_ptr p = minQ.top();
            minQ.pop();
            if (p->N >= minSupport) extractedPlanes.push_back(p);
            p->disconnectAllNbs();
        }
        static PlaneSegSizeCmp sizecmp;
        std::sort(extractedPlanes.begin(), extractedPlanes.end(), sizecmp);
        return step;
    }

    void findBlockMembership(std::vector<bool> &isValidExtractedPlane) {
        rid2plid.clear();
        for (int plid = 0; plid < (int)extractedPlanes.size(); ++plid) rid2plid.insert(std::pair<int, int>(extractedPlanes[plid]->rid, plid));
        const int Nh = height / windowHeight, Nw = width / windowWidth, NptsPerBlk = windowHeight * windowWidth;
        membershipImg.assign((size_t)height * width, -1);
        blkMap.resize(Nh * Nw);
        isValidExtractedPlane.resize(extractedPlanes.size(), false);
        for (int i = 0, blkid = 0; i < Nh; ++i)
            for (int j = 0; j < Nw; ++j, ++blkid) {
                const int setid = ds->Find(blkid);
                const int setSize = ds->getSetSize(setid) * NptsPerBlk;
                if (setSize >= minSupport) {
                    int nbs[4] = {-1};
                    const int nNbs = getValid4Neighbor(i, j, Nh, Nw, nbs);
                    bool nbClsAllTheSame = true;
                    for (int k = 0; k < nNbs && erodeType != 0; ++k)
                        if (ds->Find(nbs[k]) != setid && (erodeType == 2 || ds->getSetSize(nbs[k]) * NptsPerBlk >= minSupport)) { nbClsAllTheSame = false; break; }
                    const int plid = rid2plid[setid];   // operator[]: an unknown set id inserts (and yields) 0, as in the reference
                    if (nbClsAllTheSame) {
                        blkMap[blkid] = plid;
                        const int by = blkid / Nw, bx = blkid - by * Nw;
                        for (int y = by * windowHeight; y < (by + 1) * windowHeight; y++)
                            for (int x = bx * windowWidth; x < (bx + 1) * windowWidth; x++) membershipImg[(size_t)y * width + x] = plid;
                        isValidExtractedPlane[plid] = true;
                    } else {
                        blkMap[blkid] = -1;
                    }
                } else {
                    blkMap[blkid] = -1;
                }
                if (blkMap[blkid] < 0) {
                    if (i > 0) {
                        const int u_blkid = blkid - Nw;
                        if (blkMap[u_blkid] >= 0) {
                            const int u_plid = blkMap[u_blkid];
                            const int spixidx = (i * windowHeight - 1) * width + j * windowWidth;
                            for (int k = 1; k < windowWidth; ++k) rfQueue.push_back(std::pair<int, int>(spixidx + k, u_plid));
                        }
                    }
                    if (j > 0) {
                        const int l_blkid = blkid - 1;
                        if (blkMap[l_blkid] >= 0) {
                            const int l_plid = blkMap[l_blkid];
                            const int spixidx = (i * windowHeight) * width + j * windowWidth - 1;
                            for (int k = 0; k < windowHeight - 1; ++k) rfQueue.push_back(std::pair<int, int>(spixidx + k * width, l_plid));
                        }
                    }
                } else {
                    const int plid = blkMap[blkid];
                    if (i > 0) {
                        const int u_blkid = blkid - Nw;
                        if (blkMap[u_blkid] != plid) {
                            const int spixidx = (i * windowHeight) * width + j * windowWidth;
                            for (int k = 0; k < windowWidth - 1; ++k) rfQueue.push_back(std::pair<int, int>(spixidx + k, plid));
                        }
                    }
                    if (j > 0) {
                        const int l_blkid = blkid - 1;
                        if (blkMap[l_blkid] != plid) {
                            const int spixidx = (i * windowHeight) * width + j * windowWidth;
                            for (int k = 1; k < windowHeight; ++k) rfQueue.push_back(std::pair<int, int>(spixidx + k * width, plid));
                        }
                    }
                }
            }
    }

    void floodFill() {
        std::vector<float> distMap((size_t)height * width, std::numeric_limits<float>::max());
        for (int k = 0; k < (int)rfQueue.size(); ++k) {
            const int sIdx = rfQueue[k].first;
            const int seedy = sIdx / width, seedx = sIdx - seedy * width;
            const int plid = rfQueue[k].second;
            const PlaneSeg &pl = *extractedPlanes[plid];
            int nbs[4] = {-1};
            const int Nnbs = getValid4Neighbor(seedy, seedx, height, width, nbs);
            for (int itr = 0; itr < Nnbs; ++itr) {
                const int cIdx = nbs[itr];
                int &trail = membershipImg[cIdx];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = cIdx / width, cx = cIdx - cy * width;
                const int blkid = getBlockIdx(cx, cy);
                if (blkid >= 0 && blkMap[blkid] >= 0) continue;
                double pt[3] = {0};
                float cdist = -1;
                if (points->get(cy, cx, pt[0], pt[1], pt[2]) && std::pow(cdist = (float)std::abs(pl.signedDist(pt)), 2) < 9 * pl.mse + 1e-5) {
                    if (trail >= 0) {
                        PlaneSeg &n_pl = *extractedPlanes[trail];
                        if (pl.normalSimilarity(n_pl) >= params.T_ang(ParamSet::P_REFINE, pl.center[2])) n_pl.connect(extractedPlanes[plid].get());
                    }
                    float &old_dist = distMap[cIdx];
                    if (cdist < old_dist) {
                        trail = plid;
                        old_dist = cdist;
                        rfQueue.push_back(std::pair<int, int>(cIdx, plid));
                    } else if (trail < 0) {
                        trail -= 1;
                    }
                } else {
                    if (trail < 0) trail -= 1;
                }
            }
        }
    }

    void refineDetails() {
        std::vector<bool> isValidExtractedPlane;
        findBlockMembership(isValidExtractedPlane);
        floodFill();
        std::vector<PlaneSeg::shared_ptr> oldExtractedPlanes;
        extractedPlanes.swap(oldExtractedPlanes);
        PlaneSegMinMSEQueue minQ;
        for (int i = 0; i < (int)oldExtractedPlanes.size(); ++i)
            if (isValidExtractedPlane[i]) minQ.push(oldExtractedPlanes[i]);
        ahCluster(minQ);
        std::vector<int> plidmap(oldExtractedPlanes.size(), -1);
        for (int i = 0; i < (int)oldExtractedPlanes.size(); ++i) {
            const PlaneSeg &op = *oldExtractedPlanes[i];
            if (!isValidExtractedPlane[i]) { plidmap[i] = -1; continue; }
            const int np_rid = ds->Find(op.rid);
            for (size_t j = 0; j < extractedPlanes.size(); ++j)
                if (np_rid == extractedPlanes[j]->rid) { plidmap[i] = (int)j; break; }
        }
        // pMembership (:341-361): the pixels of every final plane, in raster order -- what PlaneDetection keeps as plane_vertices_
        pMembership.assign(extractedPlanes.size(), std::vector<int>());
        const int nPixels = width * height;
        for (int i = 0; i < nPixels; ++i) {
            int &plid = membershipImg[i];
            if (plid >= 0 && plidmap[plid] >= 0) { plid = plidmap[plid]; pMembership[plid].push_back(i); }   // every other value (incl. the negative visit counters) stays as it is
        }
    }

    void run(const Cloud *pointsIn, const msl_peac_stats *blockStats, msl_peac_block *blocksOut) {
        points = pointsIn;
        height = points->h; width = points->w;
        ds.reset(new DisjointSet((height / windowHeight) * (width / windowWidth)));
        PlaneSegMinMSEQueue minQ;
        initGraph(minQ, blockStats, blocksOut);
        ahCluster(minQ);
        if (doRefine) refineDetails();
        else membershipImg.assign((size_t)height * width, -1);   // (run() without refinement never builds membershipImg unless asked for pMembership)
    }
};

}  // namespace

// Whole plane extractor for one depth image.  membershipOut: ceil(rows / 2) x ceil(cols / 2) ints = plane_filter.membershipImg after
// runPlaneDetection(); *nPlanes = extractedPlanes.size(); blocksOut (may be NULL): the initial node of every block incl. its PCA.
static thread_local std::vector<msl_peac_plane> g_lastPlanes;
static thread_local std::vector<int32_t> g_lastOffsets, g_lastIndices;
MSLO_API void mslo_peac_run(const uint16_t *depth, size_t strideBytes, int cols, int rows, float fx, float fy, float cx, float cy, float depthMapFactor,
                            const msl_peac_params *prm, int32_t *membershipOut, int32_t *nPlanes, msl_peac_block *blocksOut) {
    Cloud cloud;
    cloud.w = (int)std::ceil(cols / 2.0); cloud.h = (int)std::ceil(rows / 2.0);
    cloud.v.assign((size_t)cloud.w * cloud.h * 3, 0.0);
    const int Nh = cloud.h / prm->window_h, Nw = cloud.w / prm->window_w;
    std::vector<msl_peac_stats> stats((size_t)Nh * Nw);
    mslo_peac_block_stats(depth, strideBytes, cols, rows, fx, fy, cx, cy, depthMapFactor, prm->window_w, prm->window_h, prm->depth_alpha, prm->depth_change_tol,
                          prm->init_loose, cloud.v.data(), stats.data());
    PlaneFitter pf;
    pf.maxStep = prm->max_step; pf.minSupport = prm->min_support; pf.windowWidth = prm->window_w; pf.windowHeight = prm->window_h;
    pf.doRefine = prm->do_refine != 0; pf.erodeType = prm->erode_type;
    pf.params.depthSigma = prm->depth_sigma; pf.params.stdTol_init = prm->std_tol_init; pf.params.stdTol_merge = prm->std_tol_merge;
    pf.params.z_near = prm->z_near; pf.params.z_far = prm->z_far; pf.params.angle_near = prm->angle_near; pf.params.angle_far = prm->angle_far;
    pf.params.similarityTh_merge = prm->similarity_th_merge; pf.params.similarityTh_refine = prm->similarity_th_refine;
    pf.params.depthAlpha = prm->depth_alpha; pf.params.depthChangeTol = prm->depth_change_tol; pf.params.initLoose = prm->init_loose;
    pf.run(&cloud, stats.data(), blocksOut);
    for (size_t i = 0; i < pf.membershipImg.size(); i++) membershipOut[i] = pf.membershipImg[i];
    *nPlanes = (int32_t)pf.extractedPlanes.size();
    g_lastPlanes.clear(); g_lastOffsets.assign(1, 0); g_lastIndices.clear();
    for (size_t j = 0; j < pf.extractedPlanes.size(); j++) {
        const PlaneSeg &ps = *pf.extractedPlanes[j];
        msl_peac_plane o;
        for (int c = 0; c < 3; c++) { o.normal[c] = ps.normal[c]; o.center[c] = ps.center[c]; }
        o.mse = ps.mse; o.N = ps.N; o._pad = 0;
        g_lastPlanes.push_back(o);
        if (j < pf.pMembership.size()) g_lastIndices.insert(g_lastIndices.end(), pf.pMembership[j].begin(), pf.pMembership[j].end());
        g_lastOffsets.push_back((int32_t)g_lastIndices.size());
    }
}
// What PlaneDetection exposes besides the membership image after the last mslo_peac_run of this thread: extractedPlanes (normal, centre, MSE, N:
// src/Frame.cc:626-632) and plane_vertices_ (vertex indices per plane, concatenated; offsets has n_planes + 1 entries)
MSLO_API int32_t mslo_peac_last_planes(msl_peac_plane *planes, int32_t *offsets, int32_t *indices) {
    for (size_t j = 0; j < g_lastPlanes.size(); j++) planes[j] = g_lastPlanes[j];
    for (size_t j = 0; j < g_lastOffsets.size(); j++) offsets[j] = g_lastOffsets[j];
    for (size_t j = 0; j < g_lastIndices.size(); j++) indices[j] = g_lastIndices[j];
    return (int32_t)g_lastPlanes.size();
}

// ahc::ParamSet / ahc::PlaneFitter defaults (include/peac/AHCParamSet.hpp:68-76, AHCPlaneFitter.hpp:157-161)
MSLO_API void mslo_peac_default_params(msl_peac_params *p) {
    p->window_w = 10; p->window_h = 10; p->min_support = 3000; p->max_step = 100000; p->do_refine = 1; p->erode_type = 2; p->init_loose = 0; p->_pad = 0;
    p->depth_sigma = 1.6e-6; p->std_tol_init = 5; p->std_tol_merge = 8; p->z_near = 500; p->z_far = 4000;
    p->angle_near = ((15.0) * M_PI / 180.0); p->angle_far = ((90.0) * M_PI / 180.0);
    p->similarity_th_merge = std::cos(((60.0) * M_PI / 180.0)); p->similarity_th_refine = std::cos(((30.0) * M_PI / 180.0));
    p->depth_alpha = 0.04; p->depth_change_tol = 0.02;
}
MSLO_API void mslo_eig33sym(const double *K9, double *s3, double *V9) {
    double K[3][3], s[3], V[3][3];
    for (int i = 0; i < 9; i++) K[i / 3][i % 3] = K9[i];
    eig33sym(K, s, V);
    for (int i = 0; i < 3; i++) s3[i] = s[i];
    for (int i = 0; i < 9; i++) V9[i] = V[i / 3][i % 3];
}
