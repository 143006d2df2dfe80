/* rt_ant/rt_api.h -- ciphertext IO of the generated Main_graph (reference rt_ant/rt_api.h:17-20,
 * src/rtlib/rtlib.c:74-87). */
#ifndef ACEHIP_RT_ANT_RT_API_H
#define ACEHIP_RT_ANT_RT_API_H
#ifdef __cplusplus
extern "C" {
#endif
CIPHERTEXT Get_input_data(const char* name, size_t idx);
void       Set_output_data(const char* name, size_t idx, CIPHER data);
/* Extension (not in the reference): Coeffs() pointers are HBM addresses and per-limb Hw_* calls are executed
 * lazily in batches.  Code that touches that memory itself (HIP / acehip_* calls on the raw pointers) calls
 * this first: it submits everything still queued and waits for the device. */
void       Acehip_rt_sync(void);
/* Extension: a thread other than the one that called Prepare_context attaches to that context on its first API
 * call (shared keys; own scratch, pool, queue, HIP stream); before it ends it may give those back. */
void       Acehip_rt_thread_release(void);
#ifdef __cplusplus
}
#endif
#endif
