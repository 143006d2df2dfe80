/* rt_ant/ant_api.h -- the rt_ant provider API used by ACE-generated C, backed by HIP kernels.
 *
 * Same names, argument meaning and error behaviour as the reference provider
 * (fhe-cmplr/rtlib/include/rt_ant/ant_api.h -> rtlib/ant/include/{poly,ckks,rtlib,util}/ *.h; the
 * emitter's list of names is fhe-cmplr/include/fhe/poly/ir2c_handler.h:30-333, ir2c_core.h:50-383).
 * Difference that generated code cannot observe: POLYNOMIAL._data is a DEVICE pointer (MI355X HBM)
 * and MODULUS carries the index of its prime instead of Barrett constants.  Generated code only
 * does pointer arithmetic on both (Coeffs(), modulus + 1), never dereferences them.
 * Errors: like the reference (include/common/error.h:23-29) violations print "file:line: msg" and abort().
 */
#ifndef ACEHIP_RT_ANT_ANT_API_H
#define ACEHIP_RT_ANT_ANT_API_H
#include <assert.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/common.h"

/* reference include/util/fhe_types.h: generated code spells booleans TRUE / FALSE */
#ifndef TRUE
#define TRUE 1
#endif
#ifndef FALSE
#define FALSE 0
#endif

/* what the reference's ant_api.h brings in besides the API below (error.h, rtlib_timing.h, trace.h through its own includes): the
 * assertion and timing macros and the trace switch are visible to every program that includes rt_ant/rt_ant.h */
#include "common/error.h"
#include "common/rtlib_timing.h"
#include "common/trace.h"
/* DCMPLX as in the reference's include/util/fhe_types.h:33-40 (the same memory layout in both languages) */
#ifdef __cplusplus
#include <complex>
typedef std::complex<double> DCMPLX;
#else
#include <complex.h>
typedef double complex DCMPLX;
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* ---- MODULUS (reference include/util/fhe_utils.h:27-32); contiguous array q_0..q_{L-1} behind
 * Q_modulus() and p_0..p_{K-1} behind P_modulus() (crt.h:471-489), stepped with `m + 1` ---- */
typedef struct {
  int64_t  _val; /* the prime */
  uint32_t _gi;  /* global prime index in the HIP context (q: 0..L-1, p: L..L+K-1) */
  uint32_t _rsv;
} MODULUS;
static inline int64_t Get_mod_val(MODULUS* m) { return m->_val; }

/* ---- POLYNOMIAL (reference include/util/polynomial.h:35-44): limb-major RNS polynomial.
 * _data -> HBM; q-limbs first, p-limbs at index (_num_alloc_primes - _num_primes_p) ---- */
/* C programs zero the shells with memset like the reference's generated code does.  CKKS-level provider programs are C++
 * and declare `CIPHERTEXT out;` expecting an empty object (rt_acehip/rt_acehip.h defines ACEHIP_SHELL_ZERO_INIT): the
 * members then value-initialise; layout and C linkage are unchanged. */
#if defined(__cplusplus) && defined(ACEHIP_SHELL_ZERO_INIT)
#define ACEHIP_ZI = {}
#else
#define ACEHIP_ZI
#endif
typedef struct {
  uint32_t _ring_degree ACEHIP_ZI;
  size_t   _num_alloc_primes ACEHIP_ZI;
  size_t   _num_primes ACEHIP_ZI;
  size_t   _num_primes_p ACEHIP_ZI;
  bool     _is_ntt ACEHIP_ZI;
  int64_t* _data ACEHIP_ZI;
} POLYNOMIAL;
typedef POLYNOMIAL* POLY;

/* reference include/util/ciphertext.h:32-38, :346-353; include/util/plaintext.h:29-34 */
typedef struct {
  POLYNOMIAL _c0_poly;
  POLYNOMIAL _c1_poly;
  uint32_t   _slots ACEHIP_ZI;
  double     _scaling_factor ACEHIP_ZI;
  uint32_t   _sf_degree ACEHIP_ZI;
} CIPHERTEXT;
typedef struct {
  POLYNOMIAL _c0_poly;
  POLYNOMIAL _c1_poly;
  POLYNOMIAL _c2_poly;
  uint32_t   _slots ACEHIP_ZI;
  double     _scaling_factor ACEHIP_ZI;
  uint32_t   _sf_degree ACEHIP_ZI;
} CIPHERTEXT3;
typedef struct {
  POLYNOMIAL _poly;
  uint32_t   _slots ACEHIP_ZI;
  double     _scaling_factor ACEHIP_ZI;
  uint32_t   _sf_degree ACEHIP_ZI;
} PLAINTEXT;
typedef CIPHERTEXT*  CIPHER;
typedef CIPHERTEXT3* CIPHER3;
typedef PLAINTEXT*   PLAIN;

/* switch key = num_q_parts pairs (b_j, a_j) of polynomials over all L+K primes
 * (reference include/util/switch_key.h, public_key.h) */
typedef struct {
  POLYNOMIAL _pk0;
  POLYNOMIAL _pk1;
} PUBLIC_KEY;
typedef struct {
  size_t      _num_parts;
  PUBLIC_KEY* _parts;
} SWITCH_KEY;
typedef SWITCH_KEY* SW_KEY;
typedef PUBLIC_KEY* PUB_KEY;

/* ---- context (reference src/rtlib/context.c:140-160) ---- */
uint32_t Degree();
double   Get_default_sc();
size_t   Get_q_parts();
size_t   Get_p_cnt();
size_t   Get_part_size();                 /* limbs per key-switch digit (context.h:94) */
void     Bootstrap_precom(uint32_t num_slots); /* bootstrap tables and keys for one slot count (context.h:118, context.c:162-185) */
MODULUS* Q_modulus();
MODULUS* P_modulus();

/* ---- polynomial API (reference include/poly/poly_eval.h:29-148, poly_arith.h:27-52) ---- */
POLY Alloc_poly(uint32_t degree, size_t q_primes, bool extend_p);
void Free_poly(POLY poly);
void Free_poly_data(POLY poly);
void Copy_poly(POLY res, POLY poly);
static inline int64_t* Coeffs(POLY poly, size_t level, uint32_t degree) { return poly->_data + level * (size_t)degree; }
void                   Set_coeffs(POLY dst, uint32_t level, uint32_t degree, int64_t* src);
static inline size_t   Poly_level(POLY poly) { return poly->_num_primes; }
static inline size_t   Num_alloc(POLY poly) { return poly->_num_alloc_primes; }
static inline size_t   Num_p(POLY poly) { return poly->_num_primes_p; }
size_t                 Num_decomp(POLY poly);
int64_t* Hw_modadd(int64_t* res, int64_t* val1, int64_t* val2, MODULUS* modulus, uint32_t degree);
int64_t* Hw_modmul(int64_t* res, int64_t* val1, int64_t* val2, MODULUS* modulus, uint32_t degree);
int64_t* Hw_rotate(int64_t* res, int64_t* val, int64_t* rot_precomp, MODULUS* modulus, uint32_t degree);
POLY     Decomp(POLY res, POLY poly, uint32_t q_part_idx);
POLY     Mod_up(POLY res, POLY poly, uint32_t q_part_idx);
POLY     Decomp_modup(POLY res, POLY poly, uint32_t q_part_idx);
POLY     Mod_down(POLY res, POLY poly);
POLY     Rescale(POLY res, POLY poly);

/* ---- keys (reference include/rtlib/key_gen.h:28-75) ---- */
uint32_t Auto_idx(int32_t rot_idx);
int64_t* Auto_order(int32_t rot_idx); /* device table; only ever passed back to Hw_rotate */
SW_KEY   Swk(bool is_rot, int32_t rot_idx);
POLY     Pk0_at(SW_KEY swk, uint32_t idx);
POLY     Pk1_at(SW_KEY swk, uint32_t idx);
static inline size_t Get_level_from_pk(PUB_KEY pk) { return pk->_pk0._num_primes; }          /* key_gen.h:83-85 */
static inline void   Set_level_for_pk(PUB_KEY pk, size_t level) {                            /* key_gen.h:93-96 */
  pk->_pk0._num_primes = level;
  pk->_pk1._num_primes = level;
}

/* ---- ciphertext API (reference include/ckks/cipher_eval.h:25-171, src/ckks/cipher_eval.c) ---- */
void     Free_cipher(CIPHER ciph);
void     Init_ciph_same_scale(CIPHER res, CIPHER ciph1, CIPHER ciph2);
void     Init_ciph_same_scale_plain(CIPHER res, CIPHER ciph1, PLAIN plain);
void     Init_ciph_same_scale_ciph3(CIPHER res, CIPHER3 ciph);
void     Init_ciph3_same_scale_ciph3(CIPHER3 res, CIPHER3 ciph1, CIPHER3 ciph2);
void     Init_ciph_up_scale(CIPHER res, CIPHER ciph1, CIPHER ciph2);
void     Init_ciph_up_scale_plain(CIPHER res, CIPHER ciph1, PLAIN plain);
void     Init_ciph3_up_scale(CIPHER3 res, CIPHER ciph1, CIPHER ciph2);
void     Init_ciph_down_scale(CIPHER res, CIPHER ciph);
void     Copy_ciph(CIPHER res, CIPHER ciph);
size_t   Level(CIPHER ciph);
uint32_t Sc_degree(CIPHER ciph);
uint32_t Get_slots(CIPHER ciph);
void     Set_slots(CIPHER ciph, uint32_t slots);
double*  Get_msg(CIPHER ciph);              /* (a ciphertext over the extended basis is brought down first, cipher_eval.c:129-148) */
DCMPLX*  Get_msg_with_imag(CIPHER ciph);
void     Print_cipher_msg(FILE* fp, const char* name, CIPHER ciph, uint32_t len);
void     Print_cipher_msg_with_imag(FILE* fp, const char* name, CIPHER ciph, uint32_t len);
void     Print_cipher_range(FILE* fp, const char* name, CIPHER ciph);
void     Print_cipher_info(FILE* fp, const char* name, CIPHER ciph);
void     Print_cipher_poly(FILE* fp, const char* name, CIPHER ciph);
void     Print_poly_lite(FILE* fp, POLY input);   /* poly_eval.h:150 */
void     Dump_cipher_msg(const char* name, CIPHER ciph, uint32_t len);
CIPHER   Real_relu(CIPHER ciph);            /* decrypt, clear ReLU, encrypt again (cipher_eval.c:264-290): a debugging aid */
void     Free_ciph_poly(CIPHER ciph, uint32_t cnt);
void     Zero_ciph(CIPHER ciph);
CIPHER   Add_ciph(CIPHER res, CIPHER ciph1, CIPHER ciph2);
CIPHER   Add_plain(CIPHER res, CIPHER ciph, PLAIN plain);
CIPHER   Sub_ciph(CIPHER res, CIPHER ciph1, CIPHER ciph2);
CIPHER   Mul_ciph(CIPHER res, CIPHER ciph1, CIPHER ciph2);
CIPHER3  Mul_ciph3(CIPHER3 res, CIPHER ciph1, CIPHER ciph2);
CIPHER   Mul_plain(CIPHER res, CIPHER ciph, PLAIN plain);
CIPHER   Relin(CIPHER res, CIPHER3 ciph);
CIPHER   Rescale_ciph(CIPHER res, CIPHER ciph);
CIPHER   Upscale_ciph(CIPHER res, CIPHER ciph, uint32_t mod_size);    /* times the constant 1 encoded at scale 2^mod_size (ckks_evaluator.c:347-360) */
CIPHER   Downscale_ciph(CIPHER res, CIPHER ciph, uint32_t waterline); /* upscale to waterline + scaling bits, then rescale (:361-379) */
void     Modswitch_ciph(CIPHER ciph);
CIPHER   Rotate_ciph(CIPHER res, CIPHER ciph, int32_t rotation);
CIPHER   Bootstrap(CIPHER res, CIPHER ciph, uint32_t level_after_bts);
CIPHER   Encrypt(CIPHER res, PLAIN plain);
/* validation helpers the code generator emits when validation is on (reference include/ckks/cipher_valid.h:25-100, src/ckks/cipher_valid.c):
 * Validate reports on stderr / stdout and goes on; <op>_msg: the operation on decrypted messages; <op>_rtv: the clear tensor operation on the
 * decrypted input; <op>_ref: the clear tensor operation on clear data.  All results are malloc'ed arrays the caller frees. */
void     Validate(CIPHER ciph, double* msg, uint32_t len, int32_t epsilon);
double*  Add_plain_msg(CIPHER op0, PLAIN op1);
double*  Add_msg(CIPHER op0, CIPHER op1, uint64_t len);
double*  Add_ref(double* op0, double* op1, uint64_t len);
double*  Mul_plain_msg(CIPHER op0, PLAIN op1);
double*  Mul_msg(CIPHER op0, CIPHER op1);
double*  Rotate_msg(CIPHER op0, int32_t rotation);
double*  Relu_msg(CIPHER op0, uint64_t len);
double*  Relu_rtv(CIPHER op0, uint64_t len);
double*  Relu_ref(double* op0, uint64_t len);
double*  Bootstrap_msg(CIPHER op0);
double*  Conv_rtv(CIPHER op0, int n, int c, int h, int w, float* weight, int kn, int kc, int kh, int kw, float* bias, int bw, int sh, int sw, int pn,
                  int pc, int ph, int pw);
double*  Conv_ref(double* op0, int n, int c, int h, int w, float* weight, int kn, int kc, int kh, int kw, float* bias, int bw, int sh, int sw, int pn,
                  int pc, int ph, int pw);
double*  Gemm_rtv(CIPHER op0, int h, int w, float* weight, int wh, int ww, float* bias, int bw);
double*  Gemm_ref(double* op0, int h, int w, float* weight, int wh, int ww, float* bias, int bw);
double*  Average_pool_rtv(CIPHER op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw);
double*  Average_pool_ref(double* op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw);
double*  Max_pool_rtv(CIPHER op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw);
double*  Max_pool_ref(double* op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw);
double*  Global_average_pool_rtv(CIPHER op0, int n, int c, int h, int w);
double*  Global_average_pool_ref(double* op0, int n, int c, int h, int w);

/* ---- plaintext API (reference include/ckks/plain_eval.h:25-58, src/ckks/plain_eval.c) ---- */
void    Encode_plain_from_float(PLAIN plain, float* input, size_t len, uint32_t sc_degree, uint32_t level);
void    Encode_plain_from_double(PLAIN plain, double* input, size_t len, uint32_t sc_degree, uint32_t level);
void    Encode_plain_from_float_with_scale(PLAIN plain, float* input, size_t len, double scale, uint32_t level); /* plain_eval.h:28-33 */
DCMPLX* Get_dcmplx_msg_from_plain(PLAIN plain);
void    Free_plain(PLAIN plain);
double* Get_msg_from_plain(PLAIN plain);

#ifdef __cplusplus
}
#endif
#endif
