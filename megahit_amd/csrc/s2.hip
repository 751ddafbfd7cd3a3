#include "mhx_internal.h"
namespace mhx {
int run_s2(mhx_ctx *, uint32_t, uint32_t, mhx_sdbg_result *) { throw Error("read2sdbg_s2: not implemented"); }
}
