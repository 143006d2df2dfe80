/* common/trace.h -- trace switch and trace file (reference rtlib/include/common/trace.h:12-52, src/trace.c).  This runtime writes no
 * per-operation trace of its own; the entry points exist so that a caller's IS_TRACE / Dump_cipher_msg lines have the file the
 * reference would give them: RTLIB_TRACE_FILE or Set_trace_file name it, stdout otherwise. */
#ifndef ACEHIP_COMMON_TRACE_H
#define ACEHIP_COMMON_TRACE_H
#include <stdbool.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
#define S_BAR "--------------------------------------------------------------------------------\n"
#ifdef Is_Trace_On
#define IS_TRACE(...)                                          \
  {                                                            \
    if (Is_trace_on()) fprintf(Get_trace_file(), __VA_ARGS__); \
  }
#define IS_TRACE_CMD(cmd)   \
  {                         \
    if (Is_trace_on()) cmd; \
  }
#else
#define IS_TRACE(...)     ((void)1)
#define IS_TRACE_CMD(cmd) ((void)1)
#endif
#define T_FILE Get_trace_file()
void  Set_trace_file(char* filename);
FILE* Get_trace_file(void);
void  Close_trace_file(void);
void  Set_trace_on(bool v);
bool  Is_trace_on(void);
#ifdef __cplusplus
}
#endif
#endif
