/* ctx_stub.c -- test helper: supplies the callbacks a generated program would define, with parameters
 * set from the test, so that Python can drive the rt_ant API (Prepare_context, Encode_plain_from_float, ...)
 * through ctypes.  Linked against libFHErt_ant.so by tests/test_gpu_encode.py. */
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

static struct {
  CKKS_PARAMS p;
  int32_t     rot[64];
} Parm;

void Stub_set_params(uint32_t n, size_t depth, size_t q0, size_t sf, size_t dnum, size_t hw) {
  memset(&Parm, 0, sizeof(Parm));
  Parm.p._provider = LIB_ANT;
  Parm.p._poly_degree = n;
  Parm.p._mul_depth = depth;
  Parm.p._first_mod_size = q0;
  Parm.p._scaling_mod_size = sf;
  Parm.p._num_q_parts = dnum;
  Parm.p._hamming_weight = hw;
  Parm.p._num_rot_idx = 0;
}
CKKS_PARAMS* Get_context_params() { return &Parm.p; }
DATA_SCHEME* Get_encode_scheme(int idx) {
  static DATA_SCHEME scheme = {"input", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
DATA_SCHEME* Get_decode_scheme(int idx) {
  static DATA_SCHEME scheme = {"output", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
/* weight data file (DE_MSG_F32 or DE_PLAINTEXT): set before Prepare_context, which then opens it (context.c:81-84) */
static RT_DATA_INFO Data_info;
static char         Data_file[1024];
void Stub_set_data_file(const char* path, int entry_type) {
  if (path == NULL) { Data_info._file_name = NULL; return; }
  strncpy(Data_file, path, sizeof(Data_file) - 1);
  Data_info._file_name = Data_file;
  Data_info._file_uuid = "test";
  Data_info._entry_type = (DATA_ENTRY_TYPE)entry_type;
}
RT_DATA_INFO* Get_rt_data_info() { return Data_info._file_name ? &Data_info : NULL; }
int  Get_output_count() { return 1; }
int  Get_input_count() { return 1; }
bool Main_graph() { return true; }
/* sizeof / offset helpers so the Python mirror cannot drift */
size_t Stub_sizeof_plaintext() { return sizeof(PLAINTEXT); }
int64_t* Stub_plain_data(PLAINTEXT* p) { return p->_poly._data; }
size_t Stub_plain_level(PLAINTEXT* p) { return p->_poly._num_primes; }
