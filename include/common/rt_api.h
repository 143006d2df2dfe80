/* common/rt_api.h -- entry points called by main() and callbacks the generated program defines
 * (reference fhe-cmplr/rtlib/include/common/rt_api.h:22-68). */
#ifndef ACEHIP_COMMON_RT_API_H
#define ACEHIP_COMMON_RT_API_H
#include "common.h"
#include "tensor.h"
#ifdef __cplusplus
extern "C" {
#endif
void    Prepare_context();
void    Finalize_context();
void    Prepare_input(TENSOR* input, const char* name);
double* Handle_output(const char* name);
void    Run_main_graph();
/* defined by the generated program */
CKKS_PARAMS*  Get_context_params();
RT_DATA_INFO* Get_rt_data_info();
int           Get_input_count();
int           Get_output_count();
DATA_SCHEME*  Get_encode_scheme(int idx);
DATA_SCHEME*  Get_decode_scheme(int idx);
bool          Main_graph();
#ifdef __cplusplus
}
#endif
#endif
