// Test double, used only where /root/reference/include is absent: the four pure virtuals of the reference's IFeatures
// that HipHOGFeatures overrides.  See ../README.md.
#ifndef PBD_TEST_DOUBLE_IFEATURES_HPP_
#define PBD_TEST_DOUBLE_IFEATURES_HPP_
#include "types.hpp"
class IFeatures {
public:
    virtual ~IFeatures() {}
    virtual size_t binsize(void) const = 0;
    virtual size_t nscales(void) const = 0;
    virtual vectorf scales(void) const = 0;
    virtual void pyramid(const cv::Mat &im, vectorMat &pyrafeatures) = 0;
};
#endif
