#include "mhx_internal.h"
namespace mhx {
int run_s1_mercy(mhx_ctx *, uint32_t, uint64_t *) { throw Error("add_mercy: not implemented"); }
int run_gen_mercy(mhx_ctx *, uint32_t, const uint32_t *, uint64_t, uint64_t, const uint64_t *, uint64_t *) { throw Error("gen_mercy_edges: not implemented"); }
}
