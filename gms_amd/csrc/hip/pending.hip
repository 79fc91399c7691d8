// All entry points declared in include/gmsx.h have kernels in this build; nothing is pending.
#include "device_graph.hpp"
