/* common/common.h -- provider-independent runtime types shared by the generated program and the
 * runtime (same names, field order and meaning as the reference's
 * fhe-cmplr/rtlib/include/common/common.h:19-92, which generated code initialises positionally). */
#ifndef ACEHIP_COMMON_COMMON_H
#define ACEHIP_COMMON_COMMON_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { NORMAL, CONV, CHANNEL, DIAGONAL } MAP_KIND;
typedef enum { LIB_ANT, LIB_SEAL, LIB_OPENFHE } LIB_PROV;
typedef enum { DE_MSG_F32, DE_MSG_F64, DE_PLAINTEXT } DATA_ENTRY_TYPE;

typedef struct {
  MAP_KIND _kind;
  int      _count;
  int      _start;
  int      _end;
  int      _stride;
} MAP_DESC;

typedef struct {
  size_t _n, _c, _h, _w;
} SHAPE;

typedef struct {
  const char* _name;
  SHAPE       _shape;
  int         _count;
  MAP_DESC    _desc[];
} DATA_SCHEME;

typedef struct {
  LIB_PROV _provider;
  uint32_t _poly_degree;
  size_t   _sec_level;
  size_t   _mul_depth;
  size_t   _first_mod_size;
  size_t   _scaling_mod_size;
  size_t   _num_q_parts;
  size_t   _hamming_weight;
  size_t   _num_rot_idx;
  int32_t  _rot_idxs[];
} CKKS_PARAMS;

typedef struct {
  const char*     _file_name;
  const char*     _file_uuid;
  DATA_ENTRY_TYPE _entry_type;
} RT_DATA_INFO;

#ifdef __cplusplus
}
#endif
#endif
