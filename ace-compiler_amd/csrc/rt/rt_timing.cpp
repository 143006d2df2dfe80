// rt_timing.cpp -- the per-function timing table of the runtime library (row (b) stdout contract):
// RTLIB_TIMING_OUTPUT=stdout|stderr|-|<file> makes Finalize_context print the table scripts/perf.py:251-260 parses
// (header "RTLib functions", one line per item "<indent><name>\t<count>\t<seconds> sec", "sub total" lines per nesting
// level).  Item names, order and nesting levels are the reference's (rtlib/include/common/rtlib_timing.h:28-78), the
// report layout follows common/src/rtlib_timing.c:28-94.
//
// What a line means here: device work is asynchronous and per-limb calls are executed lazily in batches, so the wall
// time of one API call is mostly issue time.  With RTLIB_TIMING_OUTPUT set the runtime therefore runs in a timing mode:
// a timed region that launches device work ends with a stream synchronisation (the queue is handed over first), so
// that its line carries the device time of what it issued -- slower than a normal run, but attributable.  HW_ADD /
// HW_MUL / HW_ROT count the queued per-limb calls; their time is the time of enqueueing (the arithmetic itself is
// executed inside whichever region hands the queue over).
#include <cstring>
#include <mutex>

#include "rt_internal.hpp"

namespace rt {

namespace {
struct Item {
  const char* name;
  int level;
};
// id order = RTLIB_TIMING_ALL() of the reference header
const Item kItems[RTM_LAST] = {
    {"FINALIZE_CONTEXT", 0}, {"PREPARE_CONTEXT", 0}, {"IO_SUBMIT", 0},      {"IO_COMPLETE", 0},     {"ENCODE_ARRAY", 0},
    {"ENCODE_VALUE", 0},     {"NTT", 0},             {"INTT", 0},           {"MAIN_GRAPH", 0},      {"HW_ADD", 1},
    {"HW_MUL", 1},           {"HW_ROT", 1},          {"COPY_POLY", 1},      {"DECOMP", 1},          {"MOD_DOWN", 1},
    {"MOD_UP", 1},           {"DECOMP_MODUP", 1},    {"RESCALE_POLY", 1},   {"COPY_CIPH", 1},       {"INIT_CIPH_SM_SC", 1},
    {"INIT_CIPH_UP_SC", 1},  {"INIT_CIPH_DN_SC", 1}, {"BOOTSTRAP", 1},      {"BS_COPY", 2},         {"BS_SETUP", 2},
    {"BS_KEYGEN", 2},        {"BS_EVAL", 2},         {"BS_PARTIAL_SUM", 3}, {"BS_COEFF_TO_SLOT", 3}, {"BS_APPROX_MOD", 3},
    {"BS_SLOT_TO_COEFF", 3}, {"PT_ENCODE", 1},       {"PT_GET", 1},
};
std::mutex g_mu;  // image threads add into one table (the reference's counters are global and unsynchronised)
uint64_t g_ns[RTM_LAST], g_cnt[RTM_LAST];
int g_on = -1;
}  // namespace

bool rtm_enabled() {
  if (g_on < 0) {
    const char* e = getenv("RTLIB_TIMING_OUTPUT");
    g_on = e != nullptr && *e != 0;
  }
  return g_on != 0;
}
void rtm_add(int id, uint64_t ns) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_ns[id] += ns;
  g_cnt[id]++;
}
uint64_t rtm_now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
RtmScope::RtmScope(int id_, bool device_work) : id(id_), t0(0), sync_at_end(device_work) {
  if (rtm_enabled()) t0 = rtm_now();
}
RtmScope::~RtmScope() {
  if (!t0) return;
  if (sync_at_end && g_ctx != nullptr) sync();
  rtm_add(id, rtm_now() - t0);
}

// Report_rtlib_timing rtlib_timing.c:28-94
void rtm_report() {
  if (!rtm_enabled()) return;
  const char* fname = getenv("RTLIB_TIMING_OUTPUT");
  FILE* fp = nullptr;
  bool need_close = false;
  if (strcmp(fname, "stdout") == 0 || strcmp(fname, "-") == 0) fp = stdout;
  else if (strcmp(fname, "stderr") == 0) fp = stderr;
  else {
    fp = fopen(fname, "w");
    if (!fp) return;
    need_close = true;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  fprintf(fp, "%-24s\t%12s\t%12s\n", "RTLib functions", "Count", "Elapse");
  fprintf(fp, "%-24s\t%12s\t%12s\n", "--------------------", "--------", "--------");
  uint64_t sum[16] = {0};
  uint32_t par[16] = {0};
  int index = 0;
  for (uint32_t i = 0; i < RTM_LAST; ++i) {
    if (g_cnt[i] == 0) continue;
    const int curr = kItems[i].level;
    while (curr < index) {  // close the deeper levels: their sub total goes under the parent's name
      fprintf(fp, "%*s%-24s\t%12s\t%12.6f sec\n", index - 1, "", kItems[par[index - 1]].name, "sub total", (double)sum[index] / 1e9);
      sum[index] = 0;
      par[index] = 0;
      --index;
    }
    sum[curr] += g_ns[i];
    par[curr] = i;
    fprintf(fp, "%*s%-24s\t%12ld\t%12.6f sec\n", curr, "", kItems[i].name, (long)g_cnt[i], (double)g_ns[i] / 1e9);
    index = curr;
  }
  while (index > 0) {
    fprintf(fp, "%*s%-24s\t%12s\t%12.6f sec\n", index - 1, "", kItems[par[index - 1]].name, "sub total", (double)sum[index] / 1e9);
    --index;
  }
  memset(g_ns, 0, sizeof(g_ns));
  memset(g_cnt, 0, sizeof(g_cnt));
  if (need_close) fclose(fp);
}

}  // namespace rt
