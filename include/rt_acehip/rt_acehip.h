/* rt_acehip/rt_acehip.h -- provider-level (ciphertext-granular) API of the MI355X runtime.
 *
 * ACE's code generator can stop lowering at the CKKS level (-P2C:lib=<provider>, fhe-cmplr/include/fhe/core/
 * lib_provider.h:18-21): the generated Main_graph then calls whole-ciphertext operators instead of spelling every
 * operation as host loops over RNS limbs -- the interface of rtlib/include/rt_seal/rt_seal.h:19-95 (Add_ciph, Add_plain,
 * Mul_ciph, Mul_plain, Rotate_ciph, Copy_ciph, Zero_ciph, Sc_degree, Level, Encode_plain_from_float, Get_input_data,
 * Set_output_data, Degree, Slice) plus the operators the ANT provider exposes at that level (Sub_ciph, Mul_ciph3, Relin,
 * Rescale_ciph, Modswitch_ciph, Bootstrap; rt_ant/ant_api.h).  With this interface one operator is a handful of batched
 * launches (fused key-switch, paired rescale) instead of ~100 per-limb calls: it removes the per-limb call granularity
 * at the source rather than coalescing it in the shim.
 *
 * The operators are the ones of libFHErt_ant (csrc/rt/rt_eval.cpp), bit-identical to the reference evaluator on identical
 * keys and inputs (tests/test_gpu_ct_parity.py).  Everything is declared by rt_ant/rt_ant.h; this header only adds what
 * a CKKS-level generated program expects beyond it:
 *   - in C++ (the provider programs are .cxx) CIPHERTEXT / CIPHERTEXT3 / PLAINTEXT value-initialise to empty shells, as the
 *     class types of the SEAL provider do: generated code declares `CIPHERTEXT output;` and passes &output as a result;
 *   - Dump_ciph / Dump_plain (rt_seal.h:90-92).
 * Link: -lFHErt_ant -lFHErt_common (or the .a names), exactly like an ANT-provider program.
 */
#ifndef ACEHIP_RT_ACEHIP_H
#define ACEHIP_RT_ACEHIP_H
#define ACEHIP_SHELL_ZERO_INIT 1 /* C++ only: see rt_ant/ant_api.h */
#include "rt_ant/rt_ant.h"
#ifdef __cplusplus
extern "C" {
#endif
/* decrypt + decode `len` slots starting at `start` and print them (debug aid of provider programs) */
void Dump_ciph(CIPHER ct, size_t start, size_t len);
void Dump_plain(PLAIN pt, size_t start, size_t len);
#ifdef __cplusplus
}
#endif
#endif
