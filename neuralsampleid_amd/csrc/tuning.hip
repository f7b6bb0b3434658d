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
