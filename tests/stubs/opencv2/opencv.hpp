// TEST-ONLY stand-in (see tests/stubs/README.md): the slice of the OpenCV API the adapters use.  cv::Mat, cv::FileStorage / FileNode are
// small inline containers so that tests/test_mapping_gpu.py can LINK AND RUN adapter/SurfelMapping.cpp + adapter/SurfelFusion.cpp; the array
// proxies used only by the ORB adapter stay declarations (syntax check only).  No OpenCV algorithm is implemented here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5
#define CV_16U 2
#define CV_8UC3 16
#define CV_32SC1 4
namespace cv {
struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
};
class Mat {
public:
    Mat() : rows(0), cols(0), data(nullptr), type_(0) {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
        step.v = (size_t)c * elem(type);
        own_ = std::make_shared<std::vector<unsigned char>>(step.v * (size_t)r);
        data = own_->data();
    }
    Mat(int r, int c, int type, void *external, size_t stepBytes = 0) : rows(r), cols(c), data((unsigned char *)external), type_(type) {
        step.v = stepBytes ? stepBytes : (size_t)c * elem(type);
    }
    int rows, cols;
    unsigned char *data;
    MatStep step;
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    void release() { own_.reset(); data = nullptr; rows = cols = 0; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::copy(ptr(r), ptr(r) + (size_t)cols * elem(type_), m.ptr(r));
        return m;
    }
    unsigned char *ptr(int row = 0) { return data + (size_t)row * step.v; }
    const unsigned char *ptr(int row = 0) const { return data + (size_t)row * step.v; }
    template <typename T> T *ptr(int row = 0) { return reinterpret_cast<T *>(data + (size_t)row * step.v); }
    template <typename T> const T *ptr(int row = 0) const { return reinterpret_cast<const T *>(data + (size_t)row * step.v); }
    template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T &at(int r, int c) const { return ptr<T>(r)[c]; }
    template <typename T> T &at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    template <typename T> const T &at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }

private:
    static size_t elem(int type) { return type == CV_8UC3 ? 3 : (type == CV_8U ? 1 : (type == CV_16U ? 2 : 4)); }
    int type_;
    std::shared_ptr<std::vector<unsigned char>> own_;   // shallow, reference-counted copies like cv::Mat
};
struct Point2f { float x, y; };
struct Vec3b { unsigned char v[3]; unsigned char &operator[](int i) { return v[i]; } const unsigned char &operator[](int i) const { return v[i]; } };
struct Vec3d {
    Vec3d() : v{0, 0, 0} {}
    Vec3d(double a, double b, double c) : v{a, b, c} {}
    double v[3];
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
};
class KeyPoint {
public:
    Point2f pt; float size, angle, response; int octave, class_id;
};
class _InputArray {
public:
    _InputArray(const Mat &);
    bool empty() const;
    Mat getMat() const;
};
class _OutputArray {
public:
    _OutputArray(Mat &);
    void release() const;
    void create(int rows, int cols, int type) const;
    Mat getMat() const;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
class FileNode {
public:
    explicit FileNode(double v = 0) : v_(v) {}
    operator float() const { return (float)v_; }
    operator int() const { return (int)v_; }

private:
    double v_;
};
// The settings "file" of a test: key -> value pairs registered under a path name before the object under test opens it.
class FileStorage {
public:
    enum { READ = 0 };
    static std::map<std::string, std::map<std::string, double>> &registry() {
        static std::map<std::string, std::map<std::string, double>> r;
        return r;
    }
    FileStorage(const std::string &path, int) : values_(registry()[path]) {}
    FileNode operator[](const char *key) const {
        auto it = values_.find(key);
        return FileNode(it == values_.end() ? 0.0 : it->second);
    }

private:
    std::map<std::string, double> values_;
};
}  // namespace cv
