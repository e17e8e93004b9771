// putslam_compat_types.h -- the host types that cross the drop-in boundary.
//
// When OpenCV and Eigen are installed (a real PUTSLAM checkout) their own types are used and the
// drop-in classes are signature-identical to the reference.  This build image has neither library
// (SURVEY.md section 8c), so minimal stand-ins with the SAME memory layout are provided: they carry
// data across the boundary and nothing else (no arithmetic of the hot path lives here).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#if __has_include(<opencv2/core.hpp>) && __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core.hpp>
#define PUTSLAM_HAVE_CV_EIGEN 1
namespace putslam {
typedef Eigen::Transform<double, 3, Eigen::Affine> Mat34; // reference include/putslam/Defs/putslam_defs.h:34
}
#else
#define PUTSLAM_HAVE_CV_EIGEN 0

#ifndef CV_8U
#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F
#endif

namespace cv {

// same field order / size as OpenCV's cv::DMatch (16 bytes)
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

// what the hot path reads of a cv::KeyPoint: the pyramid octave (matcher.cpp:641,783); same field order as OpenCV's
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

// Header + shared pixel buffer, like cv::Mat: copies are shallow.
class Mat {
  public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type)
    {
        step = (size_t)c * elemSize();
        owner_ = std::make_shared<std::vector<unsigned char>>(step * (size_t)r, (unsigned char)0);
        data = owner_->data();
    }
    Mat(int r, int c, int type, void *ext, size_t stepBytes = 0) : rows(r), cols(c), type_(type)
    {
        step = stepBytes ? stepBytes : (size_t)c * elemSize();
        data = static_cast<unsigned char *>(ext);
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_8U ? 1 : (type_ == CV_16U ? 2 : 4); }
    template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const
    {
        return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * sizeof(T));
    }

  private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<unsigned char>> owner_;
};

} // namespace cv

namespace Eigen {

struct Vector3f { // 12 bytes, like Eigen::Vector3f
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float &operator[](int i) { return v[i]; }
    const float &operator[](int i) const { return v[i]; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
};

struct Matrix4f { // column-major 4x4, like Eigen::Matrix4f
    float m[16];
    Matrix4f() { std::memset(m, 0, sizeof m); }
    static Matrix4f Identity()
    {
        Matrix4f r;
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    float &operator()(int r, int c) { return m[4 * c + r]; }
    const float &operator()(int r, int c) const { return m[4 * c + r]; }
    float *data() { return m; }
    const float *data() const { return m; }
};

class MatrixXd { // column-major dynamic matrix, like Eigen::MatrixXd
  public:
    MatrixXd() {}
    MatrixXd(long r, long c) : r_(r), c_(c), d_((size_t)(r * c), 0.0) {}
    long rows() const { return r_; }
    long cols() const { return c_; }
    double &operator()(long r, long c) { return d_[(size_t)(c * r_ + r)]; }
    const double &operator()(long r, long c) const { return d_[(size_t)(c * r_ + r)]; }
    const double *data() const { return d_.data(); }
    double *data() { return d_.data(); }

  private:
    long r_ = 0, c_ = 0;
    std::vector<double> d_;
};

} // namespace Eigen

namespace putslam {
// stand-in for Eigen::Transform<double,3,Affine> (4x4 column-major double)
struct Mat34 {
    double m[16];
    Mat34() { setIdentity(); }
    void setIdentity()
    {
        for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    double &operator()(int r, int c) { return m[4 * c + r]; }
    const double &operator()(int r, int c) const { return m[4 * c + r]; }
    double *data() { return m; }
    const double *data() const { return m; }
};
} // namespace putslam
#endif
