// Entry points declared in include/gmsx.h whose kernels are not in this build yet: they fail loudly
// (GMSX_ERR_UNSUPPORTED) rather than fall back to any host path.
#include "device_graph.hpp"
extern "C" {
int gmsx_bk_count(const gmsx_graph *, const int32_t *, uint64_t *, gmsx_stats *) { return GMSX_ERR_UNSUPPORTED; }
int gmsx_bk_partial(const gmsx_graph *, const int32_t *, int, int, uint64_t *, gmsx_stats *) { return GMSX_ERR_UNSUPPORTED; }
}
