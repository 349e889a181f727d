// Test double (declarations only) of the part of <opencv2/core/core.hpp> that include/pbd_opencv_adapters.hpp and the
// reference's interface headers (IFeatures.hpp, IConvolutionEngine.hpp, Model.hpp, types.hpp) name.  See README.md.
#ifndef PBD_TEST_DOUBLE_OPENCV_CORE_HPP_
#define PBD_TEST_DOUBLE_OPENCV_CORE_HPP_
#include <stddef.h>

#include <string>
#include <vector>

#define CV_MAJOR_VERSION 2
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_StsError (-2)

namespace cv {
typedef unsigned char uchar;
template <typename T> struct DataType;
template <> struct DataType<float> { enum { depth = CV_32F, channels = 1, type = CV_32F }; };
template <> struct DataType<double> { enum { depth = CV_64F, channels = 1, type = CV_64F }; };
template <> struct DataType<int> { enum { depth = CV_32S, channels = 1, type = CV_32S }; };

template <typename T> struct Point_ { T x, y; Point_(); Point_(T x_, T y_); };
typedef Point_<int> Point;
template <typename T> struct Point3_ { T x, y, z; };
typedef Point3_<int> Point3i;
template <typename T> struct Rect_ { T x, y, width, height; Rect_(); Rect_(T x_, T y_, T w_, T h_); };
typedef Rect_<int> Rect;

class Mat {
public:
    int rows, cols;
    uchar *data;
    size_t step;          // cv::Mat::step is a MatStep convertible to size_t
    Mat();
    Mat(const Mat &);
    ~Mat();
    Mat &operator=(const Mat &);
    void create(int rows_, int cols_, int type);
    template <typename T> T *ptr(int row = 0);
    template <typename T> const T *ptr(int row = 0) const;
    int depth() const;
    int channels() const;
    bool isContinuous() const;
    Mat clone() const;
    void convertTo(Mat &dst, int rtype) const;
    Mat rowRange(int startrow, int endrow) const;
};

class FilterEngine;
template <typename T> class Ptr { T *p_; };

void error(int code, const std::string &text, const char *func, const char *file, int line) __attribute__((noreturn));
}  // namespace cv

#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#endif
