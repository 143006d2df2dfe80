/* model_lib.c -- wraps an ACE-generated model source (included UNCHANGED via -DMODEL_INC) into a shared
 * library so that bench.py can drive Prepare_context / Prepare_input / Run_main_graph / Handle_output
 * in-process.  The generated source hard-codes the weight-file path of the machine it was generated on;
 * MODEL_DATA_FILE overrides it (ACEHIP_RT_DATA_SYNTH=1 needs no file at all). */
#include <stdlib.h>

#include "common/rtlib.h"

#define Get_rt_data_info Generated_get_rt_data_info
#include MODEL_INC
#undef Get_rt_data_info

RT_DATA_INFO* Get_rt_data_info() {
  static RT_DATA_INFO info;
  RT_DATA_INFO*       gen = Generated_get_rt_data_info();
  if (gen == NULL) return NULL;
  info          = *gen;
  const char* f = getenv("MODEL_DATA_FILE");
  if (f != NULL) info._file_name = f;
  return &info;
}
