// Test double, used only where /root/reference/include is absent: the two pure virtuals of the reference's
// IConvolutionEngine that HipConvolutionEngine overrides.  See ../README.md.
#ifndef PBD_TEST_DOUBLE_ICONVOLUTIONENGINE_HPP_
#define PBD_TEST_DOUBLE_ICONVOLUTIONENGINE_HPP_
#include "types.hpp"
class IConvolutionEngine {
public:
    virtual ~IConvolutionEngine() {}
    virtual void pdf(const vectorMat &features, vector2DMat &responses) = 0;
    virtual void setFilters(const vectorMat &filters) = 0;
};
#endif
