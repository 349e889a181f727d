// Test double (declarations only) of the members of the reference's Candidate (include/Candidate.hpp:56-80) that
// include/pbd_opencv_adapters.hpp calls.  See README.md.
#ifndef PBD_TEST_DOUBLE_CANDIDATE_HPP_
#define PBD_TEST_DOUBLE_CANDIDATE_HPP_
#include <opencv2/core/core.hpp>
class Candidate {
public:
    Candidate();
    virtual ~Candidate();
    void addPart(cv::Rect r, float confidence);
    void setComponent(int c);
};
#endif
