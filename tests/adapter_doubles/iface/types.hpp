// Test double, used only where /root/reference/include is absent: the typedefs of the reference's types.hpp that the
// adapter header names.  See ../README.md.
#ifndef PBD_TEST_DOUBLE_TYPES_HPP_
#define PBD_TEST_DOUBLE_TYPES_HPP_
#include <vector>
#include <opencv2/core/core.hpp>
class Candidate;
typedef std::vector<int> vectori;
typedef std::vector<float> vectorf;
typedef std::vector<cv::Mat> vectorMat;
typedef std::vector<cv::Point> vectorPoint;
typedef std::vector<Candidate> vectorCandidate;
typedef std::vector<vectori> vector2Di;
typedef std::vector<vectorf> vector2Df;
typedef std::vector<vectorMat> vector2DMat;
typedef std::vector<vector2Di> vector3Di;
#endif
