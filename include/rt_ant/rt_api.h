/* rt_ant/rt_api.h -- ciphertext IO of the generated Main_graph (reference rt_ant/rt_api.h:17-20,
 * src/rtlib/rtlib.c:74-87). */
#ifndef ACEHIP_RT_ANT_RT_API_H
#define ACEHIP_RT_ANT_RT_API_H
#ifdef __cplusplus
extern "C" {
#endif
CIPHERTEXT Get_input_data(const char* name, size_t idx);
void       Set_output_data(const char* name, size_t idx, CIPHER data);
#ifdef __cplusplus
}
#endif
#endif
