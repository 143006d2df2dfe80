/* common/io_api.h -- the table of input / output ciphertexts between main() and Main_graph, by name and index (reference
 * rtlib/include/common/io_api.h:17-41, src/io_lib.c: slots come from Get_encode_scheme / Get_decode_scheme; per thread).  Prepare_input /
 * Get_input_data / Set_output_data / Handle_output (rt_ant/rt_api.h, common/rt_api.h) are built on it; a main() that brings its own
 * ciphertexts uses it directly.  The table stores pointers, it does not own them. */
#ifndef ACEHIP_COMMON_IO_API_H
#define ACEHIP_COMMON_IO_API_H
#include "common/common.h"
#ifdef __cplusplus
extern "C" {
#endif
void  Io_init(void);
void  Io_fini(void);
void  Io_set_input(const char* name, size_t idx, void* ct);
void* Io_get_input(const char* name, size_t idx);
void  Io_set_output(const char* name, size_t idx, void* ct);
void* Io_get_output(const char* name, size_t idx);
#ifdef __cplusplus
}
#endif
#endif
