/* common/rt_config.h -- run-time switches of the library (reference rtlib/include/common/rt_config.h:17-46, src/rt_config.c):
 * CONF_OP_FUSION_DECOMP_MODUP (1; OP_FUSION_DECOMP_MODUP) is kept for the interface -- this runtime computes Decomp + Mod_up and
 * Decomp_modup with the same kernels and the same bits either way; CONF_BTS_CLEAR_IMAG (0; RT_BTS_CLEAR_IMAG) makes Bootstrap add the
 * conjugate to clear the imaginary part, as in the reference (ckks_bootstrap_context.c:1819-1840). */
#ifndef ACEHIP_COMMON_RT_CONFIG_H
#define ACEHIP_COMMON_RT_CONFIG_H
#include <stdint.h>
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { CONF_OP_FUSION_DECOMP_MODUP, CONF_BTS_CLEAR_IMAG, CONF_LAST } RTLIB_CONFIG_ID;
void    Init_rtlib_config(void);   /* defaults, then the environment */
int64_t Get_rtlib_config(RTLIB_CONFIG_ID id);
void    Set_rtlib_config(RTLIB_CONFIG_ID id, int64_t value);
#ifdef __cplusplus
}
#endif
#endif
