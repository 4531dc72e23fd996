// orb_oracle.cpp -- CPU restatement of ManhattanSLAM's ORB extractor.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (manhattanslam_amd/) never links or calls it.
//
// PARITY UNPINNED: the reference ships no tests / golden vectors, and its pixel arithmetic lives in
// OpenCV (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound), which is neither vendored
// under /root/reference nor installed in this image, so the reference cannot be built or run here.
// This file restates (a) the reference's own control flow, citing src/ORBextractor.cc line ranges,
// and (b) the published OpenCV 3.x plain-C++ algorithms for the five primitives.
// Pinned choices where the reference itself is ambiguous or platform dependent:
//   * GaussianBlur 8-bit path: the "classic" fixed-point separable filter of OpenCV <= 3.4.0
//     (kernel round(k*256) = {18,34,49,55,49,34,18}, (acc + 2^15) >> 16).
//   * cosf/sinf (src/ORBextractor.cc:108): a fixed double-precision evaluation rounded once to f32
//     (orb_sincos_pinned below) instead of the platform libm, so host and GPU agree bit for bit.
//   * std::sort of (size, node*) pairs (src/ORBextractor.cc:654): ties on size are broken by node
//     creation order (later created = larger "address").
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -march=native, no -ffast-math).

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "../include/msl.h"

namespace {

// ---------------------------------------------------------------------------------------------
// OpenCV rounding helpers: cvRound = round-half-to-even, cvFloor/cvCeil as named.
// ---------------------------------------------------------------------------------------------
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // tightly packed rows
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

const int8_t kPattern[256 * 4] = {
#include "../include/msl_orb_pattern.inc"
};

const int PATCH_SIZE = 31;       // src/ORBextractor.cc:70
const int HALF_PATCH_SIZE = 15;  // :71
const int EDGE_THRESHOLD = 19;   // :72

// ---------------------------------------------------------------------------------------------
// Extractor parameters (src/ORBextractor.cc:412-468)
// ---------------------------------------------------------------------------------------------
struct Params {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;  // include/ORBextractor.h:97 stores the float ctor argument in a double
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel;
    std::vector<int> umax;
};

Params make_params(int nfeatures, float scaleFactorF, int nlevels, int iniTh, int minTh) {
    Params p;
    p.nfeatures = nfeatures; p.nlevels = nlevels; p.iniTh = iniTh; p.minTh = minTh;
    p.scaleFactor = scaleFactorF;
    p.scale.resize(nlevels); p.sigma2.resize(nlevels);
    p.scale[0] = 1.0f; p.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {                       // :420-423
        p.scale[i] = (float)(p.scale[i - 1] * p.scaleFactor);
        p.sigma2[i] = p.scale[i] * p.scale[i];
    }
    p.invScale.resize(nlevels); p.invSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {                       // :427-430
        p.invScale[i] = 1.0f / p.scale[i];
        p.invSigma2[i] = 1.0f / p.sigma2[i];
    }
    p.perLevel.resize(nlevels);
    float factor = (float)(1.0f / p.scaleFactor);             // :435
    float nDesired = nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nlevels));  // :436-437
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {       // :440-444
        p.perLevel[level] = cv_round(nDesired);
        sum += p.perLevel[level];
        nDesired *= factor;
    }
    p.perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);   // :445
    // circular patch row ends, :453-467
    p.umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) p.umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (p.umax[v0] == p.umax[v0 + 1]) ++v0;
        p.umax[v] = v0;
        ++v0;
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
// cv::resize, INTER_LINEAR, CV_8UC1, plain C++ path (OpenCV 3.x imgproc/resize).
// Called from src/ORBextractor.cc:882.
// ---------------------------------------------------------------------------------------------
const int RESIZE_COEF_BITS = 11, RESIZE_COEF_SCALE = 1 << RESIZE_COEF_BITS;

inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void resize_linear_u8(const Image &src, Image &dst, int dw, int dh) {
    const int sw = src.w, sh = src.h;
    dst.w = dw; dst.h = dh; dst.px.assign((size_t)dw * dh, 0);
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[dx * 2] = sat_short(cv_round(c0 * RESIZE_COEF_SCALE));
        ialpha[dx * 2 + 1] = sat_short(cv_round(c1 * RESIZE_COEF_SCALE));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = sat_short(cv_round(c0 * RESIZE_COEF_SCALE));
        ibeta[dy * 2 + 1] = sat_short(cv_round(c1 * RESIZE_COEF_SCALE));
    }
    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](int sy, std::vector<int> &out) {
        sy = std::min(std::max(sy, 0), sh - 1);  // clip(sy, 0, ssize.height)
        const uint8_t *S = &src.px[(size_t)sy * sw];
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (sx + 1 < sw)
                out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
            else
                out[dx] = S[sx] * RESIZE_COEF_SCALE;  // dx >= xmax branch of HResizeLinear
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        hresize(yofs[dy], row0);
        hresize(yofs[dy] + 1, row1);
        short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        for (int dx = 0; dx < dw; dx++) {
            // VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> 8-bit specialisation
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            dst.px[(size_t)dy * dw + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8UC1 -- "classic" fixed-point separable
// filter.  Called from src/ORBextractor.cc:851-852 on a clone of the level.
// ---------------------------------------------------------------------------------------------
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

// MSL_BLUR_VARIANT selects which OpenCV generation's 8-bit kernel is pinned (README.md:43 lists 3.3.0 and 3.4.3 as tested):
//   0 (default)  every coefficient rounded on its own, round(k * 256) = {18,34,49,55,49,34,18} (sum 257): OpenCV <= 3.4.0
//                (getGaussianKernel -> convertTo(CV_32S, 256)); the Q8.8 ufixedpoint16 kernel of the 3.4.x fixed-point path rounds
//                the same coefficients the same way and its saturating u16 / u32 arithmetic cannot saturate for this kernel
//                (255 * 257 = 65535), so both generations give the same bytes;
//   1            the "bit-exact" Gaussian of later releases (4.x): the rounding error of each coefficient is carried to the next
//                one and the centre takes the remainder, {18,34,48,56,48,34,18} (sum 256).
// Both use (acc + 2^15) >> 16 with reflect-101 borders.  The same macro switches k_blur in the product.
#ifndef MSL_BLUR_VARIANT
#define MSL_BLUR_VARIANT 0
#endif
void gaussian_kernel_q8(int k[7]) {
#if MSL_BLUR_VARIANT == 1
    {   // getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (error diffusion from the border towards the centre)
        double kd[7], sigmaX = 2.0, scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; kd[i] = std::exp(scale2X * x * x); sum += kd[i]; }
        double err = 0;
        long long tot = 0;
        for (int i = 0; i < 3; i++) {
            const double adj = kd[i] / sum * 256.0 + err;
            const int v0 = cv_round(adj);
            err = adj - (double)v0;
            k[i] = k[6 - i] = v0;
            tot += v0;
        }
        k[3] = (int)(256 - 2 * tot);
        return;
    }
#endif
    // getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256)
    float cf[7];
    double sigmaX = 2.0, scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) cf[i] = (float)(cf[i] * sum);
    for (int i = 0; i < 7; i++) k[i] = cv_round((double)cf[i] * 256.0);
}

void gaussian_blur7(const Image &src, Image &dst) {
    int K[7];
    gaussian_kernel_q8(K);
    const int w = src.w, h = src.h;
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += K[k] * src.at(y, reflect101(x + k - 3, w));
            tmp[(size_t)y * w + x] = acc;
        }
    dst.w = w; dst.h = h; dst.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += K[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            dst.px[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(view, keypoints, threshold, nonmaxSuppression=true), FAST-9/16 (OpenCV 3.x
// features2d/fast.cpp + fast_score.cpp, plain C++ path).
// Called from src/ORBextractor.cc:763 and :767 on rowRange/colRange views.
// ---------------------------------------------------------------------------------------------
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const int ring[25], int v, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int d[N];
    for (int k = 0; k < N; k++) d[k] = v - ring[k];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

struct Cand { int x, y, score; };  // view coordinates

// view = rows [y0,y1) x cols [x0,x1) of img
void fast_view(const Image &img, int x0, int y0, int x1, int y1, int threshold, std::vector<Cand> &out) {
    out.clear();
    const int cols = x1 - x0, rows = y1 - y0;
    if (cols < 7 || rows < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> score((size_t)rows * cols, 0);  // 0 = not a corner (OpenCV's zeroed row buffers)
    std::vector<uint8_t> iscorner((size_t)rows * cols, 0);
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            int v = img.at(y0 + i, x0 + j);
            int ring[25];
            for (int k = 0; k < 25; k++) ring[k] = img.at(y0 + i + kRing[k & 15][1], x0 + j + kRing[k & 15][0]);
            bool corner = false;
            {   // darker arc: more than 8 contiguous ring pixels < v - t
                int vt = v - threshold, count = 0;
                for (int k = 0; k < 25; k++) {
                    if (ring[k] < vt) { if (++count > 8) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (!corner) {  // brighter arc
                int vt = v + threshold, count = 0;
                for (int k = 0; k < 25; k++) {
                    if (ring[k] > vt) { if (++count > 8) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (corner) {
                iscorner[(size_t)i * cols + j] = 1;
                score[(size_t)i * cols + j] = (uint8_t)corner_score16(ring, v, threshold);
            }
        }
    // 3x3 non-max suppression against the zero-initialised score rows
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            if (!iscorner[(size_t)i * cols + j]) continue;
            int s = score[(size_t)i * cols + j];
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (!(s > score[(size_t)(i + dy) * cols + (j + dx)])) { keep = false; break; }
                }
            if (keep) out.push_back({j, i, s});
        }
}

// ---------------------------------------------------------------------------------------------
// Quadtree distribution (src/ORBextractor.cc:477-529 DivideNode, :531-721 DistributeOctTree)
// ---------------------------------------------------------------------------------------------
struct Key { float x, y, response; };

struct Node {
    std::vector<Key> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;  // creation order: stands in for the heap address in the (size,pointer) sort
};

void divide_node(const Node &n, Node &n1, Node &n2, Node &n3, Node &n4) {
    const int halfX = (int)ceilf((float)(n.URx - n.ULx) / 2);
    const int halfY = (int)ceilf((float)(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy;
    n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY;
    n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy;
    n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (const Key &kp : n.keys) {
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.keys.push_back(kp);
            else n3.keys.push_back(kp);
        } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

std::vector<Key> distribute_octree(const std::vector<Key> &toDistribute, int minX, int maxX, int minY,
                                   int maxY, int N) {
    long seqCounter = 0;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node *> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * (float)i); ni.ULy = 0;
        ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seqCounter++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (const Key &kp : toDistribute) ini[(int)(kp.x / hX)]->keys.push_back(kp);

    auto lit = nodes.begin();
    while (lit != nodes.end()) {
        if (lit->keys.size() == 1) { lit->noMore = true; lit++; }
        else if (lit->keys.empty()) lit = nodes.erase(lit);
        else lit++;
    }

    bool finish = false;
    typedef std::pair<int, Node *> SizeNode;
    std::vector<SizeNode> sizeAndNode;
    auto by_size_then_creation = [](const SizeNode &a, const SizeNode &b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;
    };
    auto push_children = [&](Node *ch[4], std::vector<SizeNode> &rec, int *nToExpand) {
        for (int c = 0; c < 4; c++) {
            if (ch[c]->keys.size() > 0) {
                ch[c]->seq = seqCounter++;
                nodes.push_front(*ch[c]);
                if (ch[c]->keys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    rec.push_back(std::make_pair((int)ch[c]->keys.size(), &nodes.front()));
                    nodes.front().lit = nodes.begin();
                }
            }
        }
    };

    while (!finish) {
        int prevSize = (int)nodes.size();
        lit = nodes.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (lit != nodes.end()) {
            if (lit->noMore) { lit++; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            Node *ch[4] = {&n1, &n2, &n3, &n4};
            push_children(ch, sizeAndNode, &nToExpand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if (((int)nodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SizeNode> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end(), by_size_then_creation);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, n1, n2, n3, n4);
                    Node *ch[4] = {&n1, &n2, &n3, &n4};
                    push_children(ch, sizeAndNode, nullptr);
                    nodes.erase(prev[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }

    std::vector<Key> result;
    for (auto &n : nodes) {
        const Key *best = &n.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > maxResponse) { best = &n.keys[k]; maxResponse = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2 (OpenCV 3.x core/mathfuncs_core, scalar path), degrees.
// ---------------------------------------------------------------------------------------------
float fast_atan2_deg(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// IC_Angle, src/ORBextractor.cc:75-99
float ic_angle(const Image &img, int px, int py, const std::vector<int> &umax) {
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * img.at(py, px + u);
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = img.at(py + v, px + u), val_minus = img.at(py - v, px + u);
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float)m_01, (float)m_10);
}

// ---------------------------------------------------------------------------------------------
// Pinned sin/cos for src/ORBextractor.cc:108 (`cos(angle)`, `sin(angle)` on a float).
// Spec: x = (double)angle; k = floor(x*2/pi + 0.5); r = (x - k*PIO2_HI) - k*PIO2_LO;
// fdlibm kernel polynomials in double, evaluated left to right without FMA; quadrant select;
// result rounded once to f32.  The device code implements the same spec.
// ---------------------------------------------------------------------------------------------
void orb_sincos_pinned(float angle, float *s_out, float *c_out) {
    const double x = (double)angle;
    const double kd = std::floor(x * 6.36619772367581382433e-01 + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double v = z * r;
    const double sr = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + v * (S1 + z * sr);
    const double cr = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cs = 1.0 - (0.5 * z - z * cr);
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// computeOrbDescriptor, src/ORBextractor.cc:104-149
void orb_descriptor(const Image &blurred, int px, int py, float angle_deg, uint8_t desc[32]) {
    const float factorPI = (float)(M_PI / 180.f);  // :102
    float angle = angle_deg * factorPI;
    float a, b;
    orb_sincos_pinned(angle, &b, &a);  // a = cos, b = sin
    auto value = [&](int idx) -> int {
        const int x = kPattern[idx * 2], y = kPattern[idx * 2 + 1];
        int row = cv_round(x * b + y * a);
        int col = cv_round(x * a - y * b);
        return blurred.at(py + row, px + col);
    };
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = value(i * 16 + 2 * j), t1 = value(i * 16 + 2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// ---------------------------------------------------------------------------------------------
// Whole extractor (src/ORBextractor.cc:813-870 operator(), :872-893 ComputePyramid,
// :723-803 ComputeKeyPointsOctTree)
// ---------------------------------------------------------------------------------------------
struct Extractor {
    Params p;
    std::vector<Image> pyramid, blurred;
    std::vector<std::vector<Key>> candidates;  // per level, border-frame coords, pre-quadtree

    void compute_pyramid(const Image &image) {
        pyramid.assign(p.nlevels, Image());
        for (int level = 0; level < p.nlevels; ++level) {
            float scale = p.invScale[level];
            int w = cv_round((float)image.w * scale), h = cv_round((float)image.h * scale);
            // The reference also fills a 19-px reflect-101 border (:884-889); nothing downstream
            // reads it (FAST views start 16 px inside, the blur works on an unpadded clone), so
            // the restatement keeps unpadded levels.
            if (level != 0) resize_linear_u8(pyramid[level - 1], pyramid[level], w, h);
            else pyramid[0] = image;
        }
    }

    void level_candidates(int level, std::vector<Key> &toDistribute) {
        const Image &img = pyramid[level];
        const float W = 30;
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = img.w - EDGE_THRESHOLD + 3, maxBorderY = img.h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
        std::vector<Cand> cell;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                fast_view(img, (int)iniX, (int)iniY, (int)maxX, (int)maxY, p.iniTh, cell);
                if (cell.empty()) fast_view(img, (int)iniX, (int)iniY, (int)maxX, (int)maxY, p.minTh, cell);
                for (const Cand &c : cell)
                    toDistribute.push_back({(float)(c.x + j * wCell), (float)(c.y + i * hCell), (float)c.score});
            }
        }
    }

    int run(const Image &image, std::vector<msl_keypoint> &kps, std::vector<uint8_t> &desc) {
        kps.clear(); desc.clear();
        compute_pyramid(image);
        candidates.assign(p.nlevels, {});
        blurred.assign(p.nlevels, Image());
        std::vector<std::vector<msl_keypoint>> all(p.nlevels);
        for (int level = 0; level < p.nlevels; ++level) {
            const Image &img = pyramid[level];
            const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
            const int maxBorderX = img.w - EDGE_THRESHOLD + 3, maxBorderY = img.h - EDGE_THRESHOLD + 3;
            level_candidates(level, candidates[level]);
            std::vector<Key> sel = distribute_octree(candidates[level], minBorderX, maxBorderX, minBorderY,
                                                     maxBorderY, p.perLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * p.scale[level]);  // :788
            for (const Key &k : sel) {
                msl_keypoint kp;
                kp.x = k.x + minBorderX; kp.y = k.y + minBorderY;
                kp.size = (float)scaledPatchSize; kp.angle = -1; kp.response = k.response;
                kp.octave = level; kp.class_id = -1;
                all[level].push_back(kp);
            }
        }
        for (int level = 0; level < p.nlevels; ++level)  // :801-802
            for (msl_keypoint &kp : all[level])
                kp.angle = ic_angle(pyramid[level], cv_round(kp.x), cv_round(kp.y), p.umax);
        for (int level = 0; level < p.nlevels; ++level) {  // :843-869
            if (all[level].empty()) continue;
            gaussian_blur7(pyramid[level], blurred[level]);
            for (msl_keypoint &kp : all[level]) {
                uint8_t d[32];
                orb_descriptor(blurred[level], cv_round(kp.x), cv_round(kp.y), kp.angle, d);
                desc.insert(desc.end(), d, d + 32);
            }
            if (level != 0) {
                float scale = p.scale[level];
                for (msl_keypoint &kp : all[level]) { kp.x *= scale; kp.y *= scale; }
            }
            kps.insert(kps.end(), all[level].begin(), all[level].end());
        }
        return (int)kps.size();
    }
};

Image wrap(const uint8_t *gray, int w, int h, size_t stride) {
    Image im; im.w = w; im.h = h; im.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(&im.px[(size_t)y * w], gray + (size_t)y * stride, w);
    return im;
}

}  // namespace

// =============================================================================================
// C entry points used by tests/ and bench.py (cpu_baseline).
// =============================================================================================
extern "C" {

struct mslo_orb { Extractor ex; };

__attribute__((visibility("default"))) mslo_orb *mslo_orb_create(int nfeatures, float scaleFactor, int nlevels,
                                                                 int iniTh, int minTh) {
    mslo_orb *h = new mslo_orb;
    h->ex.p = make_params(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    return h;
}
__attribute__((visibility("default"))) void mslo_orb_destroy(mslo_orb *h) { delete h; }

__attribute__((visibility("default"))) int mslo_orb_tables(mslo_orb *h, float *scale, float *invScale, float *sigma2,
                                                           float *invSigma2, int32_t *perLevel, int32_t *umax16) {
    const Params &p = h->ex.p;
    for (int i = 0; i < p.nlevels; i++) {
        if (scale) scale[i] = p.scale[i];
        if (invScale) invScale[i] = p.invScale[i];
        if (sigma2) sigma2[i] = p.sigma2[i];
        if (invSigma2) invSigma2[i] = p.invSigma2[i];
        if (perLevel) perLevel[i] = p.perLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = p.umax[i];
    return 0;
}

// Full extractor on one frame; returns the keypoint count (or -1 if cap is too small).
__attribute__((visibility("default"))) int mslo_orb_extract(mslo_orb *h, const uint8_t *gray, int w, int hh,
                                                            size_t stride, msl_keypoint *kps, uint8_t *desc, int cap) {
    if (!gray || w == 0 || hh == 0) return 0;
    std::vector<msl_keypoint> k; std::vector<uint8_t> d;
    int n = h->ex.run(wrap(gray, w, hh, stride), k, d);
    if (n > cap) return -1;
    if (n) { memcpy(kps, k.data(), sizeof(msl_keypoint) * n); memcpy(desc, d.data(), (size_t)32 * n); }
    return n;
}

__attribute__((visibility("default"))) int mslo_orb_level_size(mslo_orb *h, int level, int *w, int *hh) {
    *w = h->ex.pyramid[level].w; *hh = h->ex.pyramid[level].h; return 0;
}
__attribute__((visibility("default"))) int mslo_orb_level(mslo_orb *h, int level, int blurred, uint8_t *out) {
    const Image &im = blurred ? h->ex.blurred[level] : h->ex.pyramid[level];
    if (im.px.empty()) return -1;
    memcpy(out, im.px.data(), im.px.size());
    return 0;
}
// candidates of the last extract, level pixel coordinates (border offset added)
__attribute__((visibility("default"))) int mslo_orb_candidates(mslo_orb *h, int level, int32_t *xys, int cap) {
    const auto &c = h->ex.candidates[level];
    if ((int)c.size() > cap) return -1;
    for (size_t i = 0; i < c.size(); i++) {
        xys[3 * i] = (int)c[i].x + 16; xys[3 * i + 1] = (int)c[i].y + 16; xys[3 * i + 2] = (int)c[i].response;
    }
    return (int)c.size();
}

// ---- Frame post-ORB steps (src/Frame.cc:437-463, :495-513, :155-168, :418-427) ----------------
// cv::undistortPoints(src, dst, K, distCoeffs, noArray(), P = K) follows OpenCV 3.x cvUndistortPoints (plain C path):
// normalise with 1/fx (double), 5 fixed-point iterations of the radial/tangential model, re-project with P.
static void undistort_point(const msl_frame_params &p, float xin, float yin, float *xo, float *yo) {
    const double fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k[12] = {p.k1, p.k2, p.p1, p.p2, p.k3, 0, 0, 0, 0, 0, 0, 0};
    double x = xin, y = yin;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

__attribute__((visibility("default"))) void mslo_frame_image_bounds(msl_frame_params *p, int width, int height) {
    if (p->k1 != 0.0) {   // src/Frame.cc:466-488
        float c[4][2] = {{0.f, 0.f}, {(float)width, 0.f}, {0.f, (float)height}, {(float)width, (float)height}}, u[4][2];
        for (int i = 0; i < 4; i++) undistort_point(*p, c[i][0], c[i][1], &u[i][0], &u[i][1]);
        p->minX = std::min(u[0][0], u[2][0]); p->maxX = std::max(u[1][0], u[3][0]);
        p->minY = std::min(u[0][1], u[1][1]); p->maxY = std::max(u[2][1], u[3][1]);
    } else {
        p->minX = 0.0f; p->maxX = (float)width; p->minY = 0.0f; p->maxY = (float)height;
    }
}

// UndistortKeyPoints + ComputeStereoFromRGBD + AssignFeaturesToGrid for n keypoints of one frame
__attribute__((visibility("default"))) void mslo_frame_epilogue(const msl_frame_params *pp, const msl_keypoint *kps, int n,
                                                                const float *depth, size_t depth_stride_bytes, float *un_xy,
                                                                float *depth_out, float *uright_out, int32_t *grid_cell) {
    const msl_frame_params &p = *pp;
    const float gridW = (float)MSL_FRAME_GRID_COLS / (float)(p.maxX - p.minX);   // src/Frame.cc:137-138
    const float gridH = (float)MSL_FRAME_GRID_ROWS / (float)(p.maxY - p.minY);
    for (int i = 0; i < n; i++) {
        float ux = kps[i].x, uy = kps[i].y;
        if (p.k1 != 0.0) undistort_point(p, kps[i].x, kps[i].y, &ux, &uy);
        un_xy[2 * i] = ux; un_xy[2 * i + 1] = uy;
        const float v = kps[i].y, u = kps[i].x;
        const float d = *(const float *)((const uint8_t *)depth + (size_t)(int)v * depth_stride_bytes + sizeof(float) * (size_t)(int)u);
        depth_out[i] = -1; uright_out[i] = -1;
        if (d > 0) { depth_out[i] = d; uright_out[i] = ux - p.bf / d; }
        const int posX = (int)roundf((ux - p.minX) * gridW), posY = (int)roundf((uy - p.minY) * gridH);
        grid_cell[i] = (posX < 0 || posX >= MSL_FRAME_GRID_COLS || posY < 0 || posY >= MSL_FRAME_GRID_ROWS) ? -1
                                                                                                           : posX * MSL_FRAME_GRID_ROWS + posY;
    }
}

// ---- primitives, exposed for known-answer tests ------------------------------------------------
__attribute__((visibility("default"))) void mslo_resize_linear_u8(const uint8_t *src, int sw, int sh, uint8_t *dst,
                                                                  int dw, int dh) {
    Image s = wrap(src, sw, sh, sw), d;
    resize_linear_u8(s, d, dw, dh);
    memcpy(dst, d.px.data(), d.px.size());
}
__attribute__((visibility("default"))) void mslo_gaussian_blur7(const uint8_t *src, int w, int h, uint8_t *dst) {
    Image s = wrap(src, w, h, w), d;
    gaussian_blur7(s, d);
    memcpy(dst, d.px.data(), d.px.size());
}
__attribute__((visibility("default"))) void mslo_gaussian_kernel(int32_t k[7]) { int kk[7]; gaussian_kernel_q8(kk); for (int i = 0; i < 7; i++) k[i] = kk[i]; }
// cv::FAST on a whole image used as the view; xys = (x, y, score) triples; returns count or -1
__attribute__((visibility("default"))) int mslo_fast_view(const uint8_t *img, int w, int h, int threshold, int32_t *xys,
                                                          int cap) {
    Image s = wrap(img, w, h, w);
    std::vector<Cand> out;
    fast_view(s, 0, 0, w, h, threshold, out);
    if ((int)out.size() > cap) return -1;
    for (size_t i = 0; i < out.size(); i++) { xys[3 * i] = out[i].x; xys[3 * i + 1] = out[i].y; xys[3 * i + 2] = out[i].score; }
    return (int)out.size();
}
// quadtree on explicit candidates (border-frame coords); returns selected count
__attribute__((visibility("default"))) int mslo_distribute_octree(const float *xyr, int n, int minX, int maxX, int minY,
                                                                  int maxY, int N, float *out_xyr, int cap) {
    std::vector<Key> in(n);
    for (int i = 0; i < n; i++) in[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    std::vector<Key> sel = distribute_octree(in, minX, maxX, minY, maxY, N);
    if ((int)sel.size() > cap) return -1;
    for (size_t i = 0; i < sel.size(); i++) { out_xyr[3 * i] = sel[i].x; out_xyr[3 * i + 1] = sel[i].y; out_xyr[3 * i + 2] = sel[i].response; }
    return (int)sel.size();
}
__attribute__((visibility("default"))) float mslo_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
__attribute__((visibility("default"))) void mslo_sincos(float a, float *s, float *c) { orb_sincos_pinned(a, s, c); }
__attribute__((visibility("default"))) float mslo_ic_angle(const uint8_t *img, int w, int h, int x, int y) {
    Image s = wrap(img, w, h, w);
    Params p = make_params(1000, 1.2f, 8, 20, 7);
    return ic_angle(s, x, y, p.umax);
}
__attribute__((visibility("default"))) void mslo_orb_descriptor(const uint8_t *blurred, int w, int h, int x, int y,
                                                                float angle_deg, uint8_t *desc) {
    Image s = wrap(blurred, w, h, w);
    orb_descriptor(s, x, y, angle_deg, desc);
}

}  // extern "C"
