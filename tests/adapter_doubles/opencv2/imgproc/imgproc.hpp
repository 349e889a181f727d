// Test double: the reference's types.hpp includes this header; nothing from it is named by the adapter.  See README.md.
#include <opencv2/core/core.hpp>
