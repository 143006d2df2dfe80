/* common/rtlib_timing.h -- the per-function timing table (reference rtlib/include/common/rtlib_timing.h:20-116, rtlib_timing.c:28-94):
 * identifiers in the reference's order (the report's rows and nesting), RTLIB_TM_START / RTLIB_TM_END around a region of the caller's own
 * code, RTLIB_TM_REPORT.  The runtime's own regions are timed inside the library (csrc/rt/rt_timing.cpp; with RTLIB_TIMING_OUTPUT set
 * every timed library region ends with a stream synchronisation, so its row carries device time); marks taken through these macros are
 * host wall-clock intervals of the calling thread, like the reference's. */
#ifndef ACEHIP_COMMON_RTLIB_TIMING_H
#define ACEHIP_COMMON_RTLIB_TIMING_H
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#ifdef __cplusplus
extern "C" {
#endif
#define RTLIB_TIMING_MAX_LEVEL 16
typedef enum {
  RTM_FINALIZE_CONTEXT, RTM_PREPARE_CONTEXT, RTM_IO_SUBMIT, RTM_IO_COMPLETE, RTM_ENCODE_ARRAY, RTM_ENCODE_VALUE, RTM_NTT, RTM_INTT, RTM_MAIN_GRAPH,
  RTM_HW_ADD, RTM_HW_MUL, RTM_HW_ROT, RTM_COPY_POLY, RTM_DECOMP, RTM_MOD_DOWN, RTM_MOD_UP, RTM_DECOMP_MODUP, RTM_RESCALE_POLY, RTM_COPY_CIPH,
  RTM_INIT_CIPH_SM_SC, RTM_INIT_CIPH_UP_SC, RTM_INIT_CIPH_DN_SC, RTM_BOOTSTRAP, RTM_BS_COPY, RTM_BS_SETUP, RTM_BS_KEYGEN, RTM_BS_EVAL,
  RTM_BS_PARTIAL_SUM, RTM_BS_COEFF_TO_SLOT, RTM_BS_APPROX_MOD, RTM_BS_SLOT_TO_COEFF, RTM_PT_ENCODE, RTM_PT_GET, RTM_LAST
} RTLIB_TIMING_ID;
void Append_rtlib_timing(RTLIB_TIMING_ID id, uint64_t nsec); /* one more call of `id`, nsec long */
void Report_rtlib_timing(void);                              /* the table, to where RTLIB_TIMING_OUTPUT says */
static inline uint64_t Mark_rtm_start(void) { /* seconds in the high word, nanoseconds in the low one */
  struct timespec now;
  clock_gettime(CLOCK_REALTIME, &now);
  return ((uint64_t)now.tv_sec << 32) | (uint32_t)now.tv_nsec;
}
static inline void Mark_rtm_end(RTLIB_TIMING_ID id, uint64_t start) {
  struct timespec now;
  clock_gettime(CLOCK_REALTIME, &now);
  Append_rtlib_timing(id, ((uint64_t)now.tv_sec - (start >> 32)) * 1000000000ull + (uint64_t)now.tv_nsec - (uint32_t)start);
}
#define RTLIB_ENABLE_TIMING
#define RTLIB_TM_START(id, mark) uint64_t mark = Mark_rtm_start()
#define RTLIB_TM_END(id, mark)   Mark_rtm_end(id, mark)
#define RTLIB_TM_REPORT()        Report_rtlib_timing()
#ifdef __cplusplus
}
#endif
#endif
