// TEST-ONLY declarations (see tests/stubs/README.md): the slice of the OpenCV API the adapters use.  No implementation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_16U 2
#define CV_8UC3 16
#define CV_32SC1 4
namespace cv {
struct MatStep { size_t v; operator size_t() const; };
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    int rows, cols;
    unsigned char *data;
    MatStep step;
    int type() const;
    int depth() const;
    bool empty() const;
    void release();
    Mat clone() const;
    unsigned char *ptr(int row = 0);
    const unsigned char *ptr(int row = 0) const;
    template <typename T> T *ptr(int row = 0);
    template <typename T> const T *ptr(int row = 0) const;
    template <typename T> T &at(int r, int c);
    template <typename T> const T &at(int r, int c) const;
    template <typename T> T &at(int i);
    template <typename T> const T &at(int i) const;
};
struct Point2f { float x, y; };
struct Vec3b { unsigned char v[3]; unsigned char &operator[](int i); const unsigned char &operator[](int i) const; };
struct Vec3d { Vec3d(); Vec3d(double a, double b, double c); double v[3]; double &operator[](int i); const double &operator[](int i) const; };
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
    operator float() const;
    operator int() const;
};
class FileStorage {
public:
    enum { READ = 0 };
    FileStorage(const std::string &path, int flags);
    FileNode operator[](const char *key) const;
};
}  // namespace cv
