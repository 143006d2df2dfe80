/* common/rtlib.h -- header a hand-written main() includes (reference common/rtlib.h:12-19). */
#ifndef ACEHIP_COMMON_RTLIB_H
#define ACEHIP_COMMON_RTLIB_H
#include <stdlib.h>
#include "rt_api.h"
#include "rt_stat.h"
#include "tensor.h"
#endif
