// Test double, used only where /root/reference/include is absent: the accessors of the reference's Model that
// pbdbind::FlatModel reads.  See ../README.md.
#ifndef PBD_TEST_DOUBLE_MODEL_HPP_
#define PBD_TEST_DOUBLE_MODEL_HPP_
#include <string>
#include "types.hpp"
class Model {
public:
    virtual ~Model() {}
    vectorMat &filters(void);
    vector2Df &def(void);
    vectorf &bias(void);
    vectorPoint &anchors(void);
    vector3Di &filterid(void);
    vector3Di &biasid(void);
    vector3Di &defid(void);
    vector2Di &parentid(void);
    std::string name(void);
    float thresh(void) const;
    int binsize(void) const;
    int nscales(void) const;
    int flen(void) const;
    int norient(void) const;
};
#endif
