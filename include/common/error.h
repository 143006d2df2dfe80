/* common/error.h -- assertion macros a program sees through rt_ant/rt_ant.h (reference rtlib/include/common/error.h:14-31; the same
 * observable behaviour: IS_TRUE is an assert that NDEBUG removes, FMT_ASSERT prints "file:line: message" and aborts). */
#ifndef ACEHIP_COMMON_ERROR_H
#define ACEHIP_COMMON_ERROR_H
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef NDEBUG
#define IS_TRUE(cond, msg) ((void)1)
#else
#define IS_TRUE(cond, msg) assert((cond) && (msg))
#endif
#define FMT_ASSERT(cond, ...)                                                  \
  if (!(cond)) {                                                               \
    printf("%s:%d: ", __FILE__, __LINE__), printf(__VA_ARGS__), printf("\n"); \
    abort();                                                                   \
  }
#define DEV_WARN(fmt, ...) printf(fmt, ##__VA_ARGS__)
#endif
