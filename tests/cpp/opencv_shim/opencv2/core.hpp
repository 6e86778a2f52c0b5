// TEST-ONLY stand-in for the handful of OpenCV core types integration/HFNetHIPModel.h touches (this image has no OpenCV).
// It exists so that the one file a maintainer of the reference actually builds goes through a compiler here
// (tests/test_adapter.py): cv::Mat headers over malloc'd storage, cv::KeyPoint, cv::Vec4i, cv::Size, the CV_* type codes.
// Same names, member names and call signatures as OpenCV 4.2 for the subset below; nothing else of OpenCV is modelled
// and nothing in the product includes this file.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <string>
#include <vector>

#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC(n) CV_MAKETYPE(CV_32F, (n))

namespace cv {

typedef unsigned char uchar;

template <class T> struct Point_ { T x = 0, y = 0; Point_() {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;

template <class T> struct Size_ { T width = 0, height = 0; Size_() {} Size_(T w, T h) : width(w), height(h) {} };
typedef Size_<int> Size;

template <class T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
    Vec(T a, T b, T c, T d) { static_assert(N == 4, "4-element constructor"); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
    const T& operator()(int i) const { return val[i]; }
    T& operator()(int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<int, 4> Vec4i;

struct KeyPoint {                     // defaults of cv::KeyPoint()
    Point2f pt;
    float size = 0.f, angle = -1.f, response = 0.f;
    int octave = 0, class_id = -1;
};

class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;                  // bytes per row

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, void* external, size_t step_bytes = 0) : rows(r), cols(c), data((uchar*)external), type_(type) {
        step = step_bytes ? step_bytes : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
        store_.reset((uchar*)std::malloc(std::max<size_t>((size_t)r * step, 1)), std::free);
        data = store_.get();
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return (size_t)channels() * ((type_ & 7) == CV_32F ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    Mat rowRange(int a, int b) const { Mat m = *this; m.rows = b - a; m.data = data + (size_t)a * step; return m; }
    Mat row(int r) const { return rowRange(r, r + 1); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }
    void copyTo(Mat dst) const {      // (dst is a header over existing storage of the same shape, like Mat::row(i))
        for (int r = 0; r < rows; ++r) std::memcpy(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * elemSize());
    }
private:
    int type_ = 0;
    std::shared_ptr<uchar> store_;
};

}  // namespace cv
