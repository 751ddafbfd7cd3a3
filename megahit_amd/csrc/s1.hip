#include "mhx_internal.h"
namespace mhx {
int run_s1(mhx_ctx *, uint32_t, uint32_t, int, mhx_s1_result *) { throw Error("read2sdbg_s1: not implemented"); }
}
