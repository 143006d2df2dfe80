/* common/pt_mgr.h -- weight/plaintext manager API (reference pt_mgr.h:18-47, pt_mgr.c:35-191). */
#ifndef ACEHIP_COMMON_PT_MGR_H
#define ACEHIP_COMMON_PT_MGR_H
#include "common.h"
#ifdef __cplusplus
extern "C" {
#endif
bool  Pt_mgr_init(const char* fname);
void  Pt_mgr_fini();
void  Pt_prefetch(uint32_t index);
void* Pt_get(uint32_t index, size_t len, uint32_t scale, uint32_t level);
void* Pt_get_validate(float* buf, uint32_t index, size_t len, uint32_t scale, uint32_t level);
void  Pt_free(uint32_t index);
void  Pt_from_msg(void* pt, uint32_t index, size_t len, uint32_t scale, uint32_t level);
void  Pt_from_msg_validate(void* pt, float* buf, uint32_t index, size_t len, uint32_t scale, uint32_t level);
#ifdef __cplusplus
}
#endif
#endif
