// LZ4HC block encoder (lib/src/hc/lz4_hc_block_encoder.dart:49-197) — device path.
#include "api_common.h"

namespace lz4mi {

int32_t run_compress_hc(lz4mi_ctx *c, const BatchArgs &a, int64_t depth) {
  (void)c; (void)a;
  if (depth < 1) return LZ4MI_RANGE;  // lib/src/hc/lz4_hc_options.dart:13-15
  return LZ4MI_UNSUPPORTED_HERE;
}

}  // namespace lz4mi
