// The tuning table of the kernel library (nsid_common.h NSID_TUNING_TABLE): named launch-heuristic constants with compiled-in
// defaults, changed only through this entry point — never through the environment.
#include <string.h>
#include "nsid_common.h"

namespace {
struct TuneEntry { const char* name; long def; };
const TuneEntry kTable[NSID_T_COUNT] = {
#define NSID_TUNE_ROW(name, def) {#name, def},
    NSID_TUNING_TABLE(NSID_TUNE_ROW)
#undef NSID_TUNE_ROW
};
int find(const char* key) {
  if (key == nullptr) return -1;
  for (int i = 0; i < NSID_T_COUNT; ++i)
    if (strcmp(kTable[i].name, key) == 0) return i;
  return -1;
}
}  // namespace

long g_nsid_tune[NSID_T_COUNT] = {
#define NSID_TUNE_DEF(name, def) def,
    NSID_TUNING_TABLE(NSID_TUNE_DEF)
#undef NSID_TUNE_DEF
};

extern "C" int nsid_set_tuning(const char* key, long value) {
  const int i = find(key);
  if (i < 0) return NSID_EINVAL;
  g_nsid_tune[i] = value;
  return NSID_OK;
}
extern "C" int nsid_get_tuning(const char* key, long* value) {
  const int i = find(key);
  if (i < 0 || value == nullptr) return NSID_EINVAL;
  *value = g_nsid_tune[i];
  return NSID_OK;
}
extern "C" int nsid_reset_tuning(void) {
  for (int i = 0; i < NSID_T_COUNT; ++i) g_nsid_tune[i] = kTable[i].def;
  return NSID_OK;
}
// ---- launch counters
namespace {
const char* const kCounterNames[NSID_C_COUNT] = {
#define NSID_CNT_NAME(name) #name,
    NSID_COUNTER_TABLE(NSID_CNT_NAME)
#undef NSID_CNT_NAME
};
}  // namespace
long g_nsid_counter[NSID_C_COUNT] = {};
extern "C" long nsid_debug_counter(const char* key) {
  if (key == nullptr) return -1;
  for (int i = 0; i < NSID_C_COUNT; ++i)
    if (strcmp(kCounterNames[i], key) == 0) return g_nsid_counter[i];
  return -1;
}
extern "C" int nsid_debug_counters_reset(void) {
  for (int i = 0; i < NSID_C_COUNT; ++i) g_nsid_counter[i] = 0;
  return NSID_OK;
}
extern "C" int nsid_debug_counter_count(void) { return NSID_C_COUNT; }
extern "C" const char* nsid_debug_counter_key(int i) { return (i >= 0 && i < NSID_C_COUNT) ? kCounterNames[i] : nullptr; }

extern "C" int nsid_tuning_count(void) { return NSID_T_COUNT; }
extern "C" const char* nsid_tuning_key(int i) { return (i >= 0 && i < NSID_T_COUNT) ? kTable[i].name : nullptr; }

// ---- one size query for the caller-provided scratch of every op. The kernels keep their working sets in LDS and
// registers, so most ops need none; the three that hand partial results from one launch to the next say how much here.
extern "C" int nsid_row_tiles(int M);
extern "C" int nsid_sumsq_blocks(long n);
extern "C" size_t nsid_ntxent_ws_floats(int Bg);
extern "C" long nsid_workspace_bytes(const char* op, long rows, long cols) {
  if (op == nullptr || rows < 0 || cols < 0) return -1;
  static const char* const kNone[] = {"knn_graph", "mr_aggregate", "linear", "linear_bwd_data", "linear_bwd_weight", "downsample3",
                                      "peak_patchify", "bn_apply", "node_mean", "l2norm", "adam", "ffn_fused", "mrconv_fused"};
  for (const char* n : kNone)
    if (strcmp(op, n) == 0) return 0;
  if (strcmp(op, "bn_stat") == 0)            // [2][row tiles][cols] fp32 partial sums of a rows x cols layer (forward and backward)
    return 2L * nsid_row_tiles((int)rows) * cols * (long)sizeof(float);
  if (strcmp(op, "ntxent") == 0)             // rows = pairs of the GLOBAL batch
    return (long)(nsid_ntxent_ws_floats((int)rows) * sizeof(float));
  if (strcmp(op, "sumsq") == 0)              // rows = elements of the flat gradient
    return (long)nsid_sumsq_blocks(rows) * (long)sizeof(float);
  return -1;
}
