/* rt_ant/rt_ant.h -- the header ACE-generated C includes (lib_provider.h:66-72 builds
 * "rt_<prov>/rt_<prov>.h"); same include set as the reference rt_ant/rt_ant.h:12-21. */
#ifndef ACEHIP_RT_ANT_RT_ANT_H
#define ACEHIP_RT_ANT_RT_ANT_H
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "common/pt_mgr.h"
#include "common/rt_stat.h"
#include "common/tensor.h"
#include "rt_ant/ant_api.h"
#include "rt_ant/rt_api.h"
#endif
