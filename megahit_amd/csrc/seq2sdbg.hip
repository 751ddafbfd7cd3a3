#include "mhx_internal.h"
namespace mhx {
int run_seq2sdbg(mhx_ctx *, uint32_t, mhx_sdbg_result *) { throw Error("seq2sdbg: not implemented"); }
}
