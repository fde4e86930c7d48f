// cv_compat.h — the handful of OpenCV types the filter's public headers mention (cv::Mat, cv::Size,
// cv::Scalar), for builds where OpenCV itself is not installed.  With OpenCV present the real
// <opencv2/opencv.hpp> is used and this file adds nothing.  Only what slam/mcl.h, slam/pose.h,
// slam/util.h and slam/fake_lidar.h need: rows, cols, type, at<T>(i,j), clone(), Mat(Size,type,Scalar),
// shared-buffer copy semantics.
#pragma once
#if __has_include(<opencv2/opencv.hpp>) && !defined(SLAM_FORCE_CV_COMPAT)
#include <opencv2/opencv.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#ifndef CV_8UC1
#define CV_8UC1 0
#define CV_32S 4
#define CV_64F 6
#endif
namespace cv
{
struct Size
{
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar
{
    double v[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};
class Mat
{
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(Size s, int type, const Scalar& fill = Scalar()) : rows(s.height), cols(s.width), type_(type)
    {
        buf_ = std::make_shared<std::vector<unsigned char>>(static_cast<size_t>(rows) * cols * elem());
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j) set(i, j, fill.v[0]);
    }
    Mat(int r, int c, int type) : Mat(Size(c, r), type) {}
    int type() const { return type_; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return !buf_ || buf_->empty(); }
    template <class T>
    T& at(int i, int j) { return reinterpret_cast<T*>(buf_->data())[static_cast<size_t>(i) * cols + j]; }
    template <class T>
    const T& at(int i, int j) const { return reinterpret_cast<const T*>(buf_->data())[static_cast<size_t>(i) * cols + j]; }
    // row pointers, as cv::Mat has them (the shim reads and writes matrices row by row: a real cv::Mat may be a view
    // whose rows are not adjacent in memory)
    template <class T>
    T* ptr(int i) { return reinterpret_cast<T*>(buf_->data()) + static_cast<size_t>(i) * cols; }
    template <class T>
    const T* ptr(int i) const { return reinterpret_cast<const T*>(buf_->data()) + static_cast<size_t>(i) * cols; }
    bool isContinuous() const { return true; }
    unsigned char* data() { return buf_ ? buf_->data() : nullptr; }
    const unsigned char* data() const { return buf_ ? buf_->data() : nullptr; }
    Mat clone() const
    {
        Mat m;
        m.rows = rows;
        m.cols = cols;
        m.type_ = type_;
        if (buf_) m.buf_ = std::make_shared<std::vector<unsigned char>>(*buf_);
        return m;
    }

private:
    size_t elem() const { return type_ == CV_8UC1 ? 1 : type_ == CV_32S ? 4 : 8; }
    void set(int i, int j, double v)
    {
        if (type_ == CV_8UC1) at<unsigned char>(i, j) = static_cast<unsigned char>(v);
        else if (type_ == CV_32S) at<int>(i, j) = static_cast<int>(v);
        else at<double>(i, j) = v;
    }
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<unsigned char>> buf_;  // copies share the buffer, like cv::Mat
};
}  // namespace cv
#endif
