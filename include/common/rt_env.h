/* common/rt_env.h -- names of the environment variables the runtime reads (reference rtlib/include/common/rt_env.h:12-34); the
 * ACEHIP_* additions are listed in INTEGRATION.md */
#ifndef ACEHIP_COMMON_RT_ENV_H
#define ACEHIP_COMMON_RT_ENV_H
#define ENV_RTLIB_TIMING_OUTPUT    "RTLIB_TIMING_OUTPUT"    /* stdout | stderr | <file>: the per-function table at Finalize_context */
#define ENV_RTLIB_TRACE_FILE       "RTLIB_TRACE_FILE"
#define ENV_PT_ENTRY_COUNT         "PT_ENTRY_COUNT"
#define ENV_PT_PREFETCH_COUNT      "PT_PREFETCH_COUNT"      /* honoured when ACEHIP_PT_PREFETCH is not set */
#define ENV_RT_DATA_ASYNC_READ     "RT_DATA_ASYNC_READ"     /* accepted, no effect: weight files are read once and stay in HBM */
#define ENV_BOOTSTRAP_EVEN_POLY    "RTLIB_BTS_EVEN_POLY"
#define ENV_OP_FUSION_DECOMP_MODUP "OP_FUSION_DECOMP_MODUP" /* accepted, no effect: Decomp + Mod_up and Decomp_modup give the same bits */
#define ENV_BOOTSTRAP_CLEAR_IMAG   "RT_BTS_CLEAR_IMAG"
#endif
