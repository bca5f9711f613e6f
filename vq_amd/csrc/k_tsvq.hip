// k_tsvq.hip -- TSVQ kernels (placeholder translation unit; filled in by a later milestone).
#include "kernels.hpp"
namespace vqhip {}
