/* acehip.h -- C ABI of the MI355X-native RNS-CKKS polynomial layer (libacehip.so).
 *
 * This is the drop-in boundary for the hot path of ACE's rt_ant runtime library: each entry point
 * replaces one reference function (cited per declaration; paths relative to
 * /root/reference/fhe-cmplr/rtlib/ant) and lands in hand-written HIP kernels for gfx950.
 * Plain pointers and sizes only; every `d_*` pointer is a DEVICE pointer (HBM).  Residues are
 * int64 in the reference API and are treated as uint64 canonical values in [0, q) here.
 *
 * Data layout (same as the reference POLYNOMIAL, include/util/polynomial.h:35-44): limb-major,
 * limb l of a polynomial at d_poly + l*N.  An "extended" polynomial at level l has l q-limbs
 * followed directly by the K p-limbs (reference Alloc_poly(N, l, extend_p=1)).  The prime of the
 * limb at position `pos` of a polynomial extended at `level` is q_pos if pos < level, else
 * p_{pos-level}.  Switch keys are stored at full level: L q-limbs then K p-limbs per key polynomial.
 *
 * Error convention: functions return 0 on success, a negative ACEHIP_E* code otherwise;
 * acehip_last_error() returns a message for the calling thread.  (The reference aborts via
 * FMT_ASSERT, include/common/error.h:23-29; the rt_ant shim on top of this ABI does the same.)
 * All launches are asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * Threads: an acehip_ctx is used by ONE host thread at a time (replica selection, workspace and statistics live in it); host
 * threads that work concurrently each create their own context on the same device -- tables and keys are read-only device memory
 * that several contexts may share (this is what the rt_ant shim does for the reference's one-thread-per-image main()).
 */
#ifndef ACEHIP_H
#define ACEHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acehip_ctx acehip_ctx;
typedef void*             acehip_stream; /* hipStream_t */

#define ACEHIP_OK 0
#define ACEHIP_EINVAL (-1)   /* bad argument (level, limb range, N ...) */
#define ACEHIP_EHIP (-2)     /* a HIP runtime call failed */
#define ACEHIP_ENODEV (-3)   /* no GPU / extension not usable */

const char* acehip_last_error(void);
int         acehip_device_count(void);
/* 16 hex digits: sha256 over the sources (ace-compiler_amd/csrc, include) this library was built from, embedded by the build
 * (ace-compiler_amd/build.py).  The Python binding compares it with the sources beside the library, the rt_ant shim with its own
 * (acehip_rt_source_fingerprint, include/rt_ant/rt_api.h): a stale binary is an error, never a silent run.  No reference counterpart. */
const char* acehip_source_fingerprint(void);

/* ---- context: Prepare_context -> Init_ckks_parameters_with_prime_size (src/rtlib/context.c:29-86,
 * src/util/ckks_parameters.c:60-101, src/util/crt.c:574-585).  Generates the q/p prime chains, psi,
 * twiddles and CRT tables exactly as the reference and uploads them to HBM of `device`.
 * dnum = 0 selects the reference default number of q parts. ---- */
acehip_ctx* acehip_ctx_create(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum, int device);
void        acehip_ctx_destroy(acehip_ctx* ctx);
/* host-only variant (tables are generated but nothing is uploaded; no GPU needed).  Used to test the
 * host logic; every launch on such a context fails with ACEHIP_ENODEV. */
acehip_ctx* acehip_ctx_create_host(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum);

uint32_t acehip_degree(const acehip_ctx* ctx);      /* Degree()            context.c:140 */
uint32_t acehip_num_q(const acehip_ctx* ctx);       /* L                                   */
uint32_t acehip_num_p(const acehip_ctx* ctx);       /* Get_p_cnt()         context.c:156 */
uint32_t acehip_num_q_parts(const acehip_ctx* ctx); /* Get_q_parts()       context.c:148 */
uint32_t acehip_part_size(const acehip_ctx* ctx);   /* alpha, crt.c:386 */
uint32_t acehip_num_decomp(const acehip_ctx* ctx, uint32_t level); /* Num_decomp, polynomial.h:158-168 */
/* prime of global index gi (0..L-1: q, L..L+K-1: p)   Q_modulus()/P_modulus() context.c:158-160 */
uint64_t acehip_prime(const acehip_ctx* ctx, uint32_t gi);
/* host copies of generated tables, for table-parity tests (what: 0 psi, 1 n_inv, 2 n_inv_prec,
 * 3 prec128_lo, 4 prec128_hi -> out[L+K]; 10 rou, 11 rou_prec, 12 rou_inv, 13 rou_inv_prec -> out[N] of
 * prime gi; 20 phat_inv_modp[K], 21 phat_inv_modp_prec[K], 22 phat_modq[L*K], 23 pinv_modq[L];
 * 30 ql_inv[L*L], 31 ql_inv_prec, 32 qlql, 33 qlql_prec).  Returns number of words written. */
int64_t acehip_get_table(const acehip_ctx* ctx, int what, uint32_t gi, uint64_t* out, size_t cap);
/* ModUp tables of (level, digit): hat_inv[n2], compl_idx[nc], hat_mod[n2*nc]; returns n2 (crt.c:426-533) */
int acehip_get_modup_tables(const acehip_ctx* ctx, uint32_t level, uint32_t digit, uint64_t* hat_inv,
                            uint32_t* compl_idx, uint64_t* hat_mod, uint32_t* nc_out);
/* rotation index -> automorphism index k (Find_automorphism_index number_theory.c:187-199; Auto_idx
 * key_gen.h:28) and the NTT-domain gather table uploaded to HBM (Auto_order key_gen.h:41,
 * Precompute_automorphism_order number_theory.c:201-214).  The returned device pointer (N uint32
 * entries) is owned by the context and cached per k. */
uint32_t        acehip_auto_index(const acehip_ctx* ctx, int32_t rot_idx);
const uint32_t* acehip_auto_order(acehip_ctx* ctx, uint32_t k);
int             acehip_auto_order_host(const acehip_ctx* ctx, uint32_t k, uint32_t* out_perm);

/* ---- device memory helpers (thin wrappers so a C caller needs no HIP headers) ---- */
void* acehip_malloc(size_t bytes);
int   acehip_free(void* d_ptr);
int   acehip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, acehip_stream stream);
int   acehip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, acehip_stream stream);
int   acehip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, acehip_stream stream);
int   acehip_memset(void* d_ptr, int value, size_t bytes, acehip_stream stream);
/* pinned host memory + a copy that does not wait (the source must stay untouched until an event recorded
 * after it has completed: acehip_event_sync) */
/* Device memory for n_limbs limbs in the reference layout (limb k at base + k * N words; h_gi[k] = prime index of limb k) -- the memory
 * of switch keys.  On a context in limb-sharded mode over RCCL (acehip_ctx_shard_rccl) only the limbs this rank owns
 * (h_gi[k] % world == rank) get physical memory of their own; every other limb position maps ONE shared scratch limb, so the whole
 * layout stays addressable (generated code indexes key limbs itself: key_gen.h:28-75) and a rank of G holds 1/G of the key bytes.
 * HIP virtual memory management, opt-in with ACEHIP_SHARD_OWNER_LIMBS=1 (on ROCm 7.2 every owned limb needs a physical handle of its own,
 * which occupies 2 MiB: it pays from 8 ranks on); a plain allocation otherwise.  Freed with
 * acehip_free.  acehip_limb_memory: bytes physically backed / bytes addressed over all live acehip_malloc_limbs blocks of the process. */
void* acehip_malloc_limbs(acehip_ctx* ctx, const uint32_t* h_gi, size_t n_limbs);
void  acehip_limb_memory(uint64_t* backed_bytes, uint64_t* addressed_bytes);
void* acehip_malloc_host(size_t bytes);
int   acehip_free_host(void* h_ptr);
int   acehip_memcpy_h2d_async(void* d_dst, const void* h_pinned_src, size_t bytes, acehip_stream stream);
int   acehip_stream_sync(acehip_stream stream);
/* HIP events on the launch stream (bench.py times kernels with these, not with host clocks) */
void* acehip_event_create(void);
int   acehip_event_record(void* event, acehip_stream stream);
int   acehip_event_elapsed_ms(void* start, void* stop, float* ms_out); /* synchronises on `stop` */
int   acehip_event_sync(void* event);   /* returns at once for an event never recorded */
int   acehip_event_destroy(void* event);

/* ---- NTT.  In-place over limbs [pos0, pos0+n_limbs) of the polynomial at d_poly extended at `level`.
 * forward: natural -> bit-reversed evaluation order  (Forward_transform ntt.c:190-264, Ftt_fwd :163,
 *          Conv_poly2ntt_inplace* polynomial.c:532-631)
 * inverse: bit-reversed -> natural, scaled by N^-1   (Inverse_transform ntt.c:268-353, Ftt_inv :177,
 *          Conv_ntt2poly_inplace* polynomial.c:633-731) ---- */
int acehip_ntt_forward(acehip_ctx* ctx, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
int acehip_ntt_inverse(acehip_ctx* ctx, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* the same transform on n_polys polynomials that lie poly_stride words apart (c0/c1 of a ciphertext,
 * the digits of a key-switch, a batch of ciphertexts): one launch, limbs x polys workgroup grid */
int acehip_ntt_batch(acehip_ctx* ctx, uint64_t* d_polys, size_t poly_stride, uint32_t n_polys, uint32_t level,
                     uint32_t pos0, uint32_t n_limbs, int inverse, acehip_stream stream);

/* ---- limb-wise ops over limbs [pos0, pos0+n_limbs) (each operand has its own base pointer; the
 * limb at position pos of every operand is at base + pos*N).
 * Hw_modadd / Hw_modmul / Hw_rotate: src/poly/poly_arith.c:14-56 (one limb per call there;
 * n_limbs = 1 reproduces that call exactly).  d_perm: table from acehip_auto_order(). ---- */
int acehip_modadd(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
int acehip_modsub(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
int acehip_modmul(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* res += a*b   (Multiply_add polynomial.c:148-183) */
int acehip_modmuladd(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
int acehip_rotate(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint32_t* d_perm, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* res_z = acc_z + automorphism_k(a_z) on one (d_res1 == NULL) or two polynomials: Automorphism_transform followed by Add_poly
 * (the outer sums of Rotate_iteration, ckks_bootstrap_context.c:1343-1377) in one pass.  auto_k: acehip_auto_index();
 * d_res may be d_acc, d_a must not alias a result. */
int acehip_rotate_add2(acehip_ctx* ctx, uint64_t* d_res0, uint64_t* d_res1, const uint64_t* d_acc0, const uint64_t* d_acc1,
                       const uint64_t* d_a0, const uint64_t* d_a1, uint32_t auto_k, uint32_t level, uint32_t pos0, uint32_t n_limbs,
                       acehip_stream stream);
/* single-limb forms with an explicit prime index, the exact shape of the generated code's calls
 * Hw_modadd(res, a, b, modulus, degree) where modulus = Q_modulus()+i or P_modulus()+i */
int acehip_hw_modadd(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t prime_gi, acehip_stream stream);
int acehip_hw_modmul(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* d_b, uint32_t prime_gi, acehip_stream stream);
int acehip_hw_rotate(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint32_t* d_perm, uint32_t prime_gi, acehip_stream stream);

/* A list of such per-limb calls handed over at once.  Result: exactly what issuing ops[0..n) one by one would
 * give (limbs are N words; two limb pointers are either equal or disjoint in every case the reference code
 * produces -- partially overlapping limbs are still handled, op by op, with every op reading its operands as
 * they were before the op started).  The library groups the ops into
 * dependency chains and runs them in a few launches instead of n: generated code calls Hw_* once per RNS limb
 * and component, ~600k times per ResNet-20 image.
 *   ADD/SUB/MUL: res = a (+|-|*) b mod prime(prime_gi);  MULADD: res += a*b (Multiply_add polynomial.c:148);
 *   MULC/ADDC: res = a (*|+) k with the residue k < prime passed as an integer in `b`
 *   (Scalar_integer_multiply_poly polynomial.c:190);  ROTATE: res[j] = a[perm[j]], b = table of
 *   acehip_auto_order(), res must not alias a;  COPY: res = a;  ZERO: res = 0. */
enum {
  ACEHIP_HW_ADD = 0, ACEHIP_HW_MUL = 1, ACEHIP_HW_ROTATE = 2, ACEHIP_HW_COPY = 3, ACEHIP_HW_ZERO = 4,
  ACEHIP_HW_SUB = 5, ACEHIP_HW_MULADD = 6, ACEHIP_HW_MULC = 7, ACEHIP_HW_ADDC = 8
};
/* prime_gi of an op whose result is not a limb of the chain (limb-sharded execution only: every rank runs it) */
#define ACEHIP_HW_ANY_RANK 0xffffffffu
typedef struct acehip_hw_op {
  uint32_t        op;        /* ACEHIP_HW_* */
  uint32_t        prime_gi;  /* global prime index of the limb (q: 0..L-1, p: L..L+K-1): the modulus of the arithmetic kinds; for
                                every kind the limb's owner under limb-sharded execution (ROTATE / COPY / ZERO ignore it otherwise) */
  uint64_t*       res;
  const uint64_t* a;
  const void*     b;         /* second operand (uint64 limb); uint32 automorphism table (ROTATE); residue (MULC/ADDC) */
} acehip_hw_op;
int acehip_hw_batch(acehip_ctx* ctx, const acehip_hw_op* ops, size_t n_ops, acehip_stream stream);
/* The same analysis without a GPU (works on a host-only context): what acehip_hw_batch WOULD launch, as a flat list of
 * ops (dead zero fills / copies removed, intermediate versions renamed to scratch limbs starting at `scratch_base`, an
 * arbitrary non-null address) with the launch and the chain segment each belongs to.  Ops of one segment run in list
 * order, segments of a launch in any order, launches in order.  Returns the number of planned ops (may exceed `cap`,
 * in which case only the first `cap` were written) or a negative error.  tests/test_hw_batch_plan.py replays plans. */
long acehip_hw_batch_plan(acehip_ctx* ctx, const acehip_hw_op* ops, size_t n_ops, acehip_hw_op* out_ops, uint32_t* out_launch,
                          uint32_t* out_segment, size_t cap, uint64_t scratch_base);
/* acehip_hw_batch for a caller that has given up some of the memory the ops name: the reference's generated code frees its
 * temporaries (Free_poly_data, e.g. resnet20_cifar10_pre.onnx.inc:1464-1471) right after the loops that use them, so by
 * the time a queued list is handed over many result limbs belong to freed blocks.  dead[0..n_dead) are disjoint ranges of
 * device memory (words of 8 bytes) whose contents after the call are UNSPECIFIED: every limb outside them ends up exactly
 * as after acehip_hw_batch; ops that only feed limbs inside are skipped, and results consumed only by the next op of their
 * chain are not written to memory.  In a plan such an op carries ACEHIP_HW_NOSTORE in `op`: its result is visible to the
 * next op of the same segment only. */
typedef struct acehip_hw_range {
  const uint64_t* ptr;
  size_t          words;
} acehip_hw_range;
#define ACEHIP_HW_NOSTORE 0x80000000u
int acehip_hw_batch_discard(acehip_ctx* ctx, const acehip_hw_op* ops, size_t n_ops, const acehip_hw_range* dead, size_t n_dead,
                            acehip_stream stream);
long acehip_hw_batch_plan_discard(acehip_ctx* ctx, const acehip_hw_op* ops, size_t n_ops, const acehip_hw_range* dead, size_t n_dead,
                                  acehip_hw_op* out_ops, uint32_t* out_launch, uint32_t* out_segment, size_t cap,
                                  uint64_t scratch_base);

/* ---- RNS basis operations (NTT-domain in, NTT-domain out) ----
 * Decomp_modup (src/poly/poly_eval.c:28 -> Decompose_modup polynomial.c:1241-1335): digit `digit` of
 *   d_in (level q-limbs) raised to the level+K limbs of d_out.
 * Mod_down (poly_eval.c:36 -> Reduce_rns_base polynomial.c:928-967): d_in has level+K limbs, d_out level.
 * Rescale (poly_eval.c:43 -> Rescale_poly polynomial.c:1097-1163): d_in level limbs -> d_out level-1. */
int acehip_decomp_modup(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, uint32_t level, uint32_t digit, acehip_stream stream);
int acehip_mod_down(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, uint32_t level, acehip_stream stream);
int acehip_rescale(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, uint32_t level, acehip_stream stream);

/* Base conversion onto a chosen subset of target limbs: the building block of limb-sharded execution, where
 * every GPU converts only the limbs it owns after the source limbs have been all-gathered.  `which` = digit index
 * (Decompose_modup polynomial.c:1297-1320, sources = the digit's limbs in the coefficient domain, unscaled) or
 * ACEHIP_CONV_MODDOWN (Fast_base_conv :755-807, sources = the K p-limbs in the coefficient domain).  d_in: the source
 * limbs, contiguous; h_out_pos (host): target limb positions in the polynomial extended at `level`; d_out: n_out limbs
 * (output k = position h_out_pos[k]), coefficient domain.  Synchronises the stream (setup-style call). */
#define ACEHIP_CONV_MODDOWN (-1)
int acehip_base_conv(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, uint32_t level, int which, const uint32_t* h_out_pos,
                     uint32_t n_out, acehip_stream stream);
/* ModRaise of bootstrapping (Transform_values_from_level0 ckks_bootstrap_context.c:1527-1551): limb 0 of d_in0
 * (and d_in1, may both be NULL with d_out1) in the NTT domain is taken to the coefficient domain, lifted to its
 * centred representative and spread to `level_out` limbs (NTT domain) at d_out0 / d_out1. */
int acehip_mod_raise(acehip_ctx* ctx, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_in0, const uint64_t* d_in1, uint32_t level_out, acehip_stream stream);
/* Mod_down / Rescale of the two polynomials of a ciphertext in the same launches (generated code calls
 * Mod_down(c0); Mod_down(c1) and Rescale(c0); Rescale(c1) back to back: resnet20_cifar10_pre.onnx.inc:7035-7036,
 * :1552-1553).  Results are those of the single-polynomial calls. */
int acehip_mod_down2(acehip_ctx* ctx, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_in0, const uint64_t* d_in1, uint32_t level, acehip_stream stream);
int acehip_rescale2(acehip_ctx* ctx, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_in0, const uint64_t* d_in1, uint32_t level, acehip_stream stream);
/* ---- key-switch core of the generated Rotate()/Relinearize()
 * (dataset/resnet20_cifar10_pre.onnx.inc:6972-7146; Fast_switch_key ckks_evaluator.c:391-416):
 *   out0 = Mod_down(sum_d key0[d] * Decomp_modup(in, d)),  out1 likewise with key1.
 * d_key: [num_q_parts][2][L+K][N] (Pk0_at/Pk1_at key_gen.h:66-73); d_in: level limbs; outs: level limbs.
 * d_out0 / d_out1 may be d_in (an output that overlaps the input takes the pipeline with stored accumulators, which has read the
 * input completely before it writes); they must not overlap each other or the key. */
int acehip_key_switch(acehip_ctx* ctx, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_in,
                      const uint64_t* d_key, uint32_t level, acehip_stream stream);

/* ---- call statistics (process-wide): per entry-point family, the number of calls, the units processed (limbs,
 * digits, key-switches ...) and the ALGORITHMIC bytes of SURVEY 8(d) (tables, scratch and re-reads excluded).
 * bench.py divides the per-image sum by the wall time for the whole-workload roofline.  Returns the number of
 * families; acehip_stat_name(i) names family i ("ntt", "elementwise", "rotate", "decomp_modup",
 * "key_inner_product", "mod_down", "rescale", "key_switch", "encode"; two subsets of "elementwise": "zero_fill_executed", the zero
 * fills that were actually executed, i.e. not proven dead inside their batch, and "elementwise_mul", the limb-ops that contain a
 * modular multiplication (units = limb multiplications): what a CPU spends its elementwise time on); and "ntt_launched": every
 * limb-transform the library launched, the ones inside the pipelines (ModUp, ModDown, Rescale, key-switch, encode) included -- "ntt"
 * only counts the direct entry points (units = limb-transforms, bytes = 16 per coefficient index: multiply by N). */
typedef struct acehip_stat {
  uint64_t calls, units, bytes;
} acehip_stat;
int         acehip_stats(acehip_stat* out, int n_out, int reset);
const char* acehip_stat_name(int family);

/* ---- setup-side entry points (key generation, encryption, encoding: SURVEY 8 rows a15-a18) ----
 * d_poly[pos][n] = d_vals[n] mod prime(pos), d_vals signed 64-bit on the device
 *   (Transform_values_to_rns polynomial.c:362-392 / Transform_values_at_level :432). */
int acehip_values_to_rns(acehip_ctx* ctx, uint64_t* d_poly, const int64_t* d_vals, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* CKKS encode of a message vector, everything on the device (Encode_at_level_with_sf ckks_encoder.c:395 ->
 * Encode_impl :199-297 with Embedding_inv ntt.c:713-753): d_vals holds `len` values (kind 0 = float real,
 * 1 = double real, 2 = complex double re/im interleaved), zero padded to `slots` (0 = N/2, a power of two);
 * the FP64 inverse embedding follows the reference's butterfly order with no FMA contraction, so the plaintext
 * is bit-identical.  Output: `level` q-limbs at d_q (times Delta^(sf_degree-1)) and n_p p-limbs at d_p
 * (may be NULL when n_p == 0), both in the NTT domain.  Coefficient overflow (|x*Delta| > 9.2e18, the
 * reference's assert) is recorded in a sticky flag: acehip_encode_status() synchronises and reports it. */
int acehip_encode(acehip_ctx* ctx, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                  double scaling_factor, uint32_t sf_degree, uint32_t level, uint32_t n_p, acehip_stream stream);
/* the same with an explicit scale instead of a power of the scaling factor (Encode_at_level_with_scale ckks_encoder.c:401 ->
 * Encode_impl_with_scale :301-378): every coefficient is llround(x * scale) -- no half is added before the rounding and nothing is
 * multiplied in afterwards */
int acehip_encode_with_scale(acehip_ctx* ctx, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                             double scale, uint32_t level, uint32_t n_p, acehip_stream stream);
/* The same for n_batch <= 8 messages of equal kind / length / slots / scale / level in one set of launches (embedding
 * kernels over the batch, one NTT with n_batch polynomials sharing every limb's twiddles): h_q[b] / h_vals[b] are HOST arrays
 * of device pointers (output q-limbs, message values).  Results are bit-identical to n_batch acehip_encode calls.  This is what
 * the weight-plaintext prefetch of the rt_ant shim issues (the reference prefetches weight plaintexts too: pt_mgr.c:128-159). */
int acehip_encode_batch(acehip_ctx* ctx, uint64_t* const* h_q, const void* const* h_vals, uint32_t n_batch, int kind, size_t len,
                        uint32_t slots, double scaling_factor, uint32_t sf_degree, uint32_t level, acehip_stream stream);
int acehip_encode_status(acehip_ctx* ctx);
/* uniformly random residues (Sample_uniform_poly polynomial.c:1349-1371; the generator differs from the
 * reference's BLAKE2 PRNG: key material is random by construction, parity is per operator) */
int acehip_sample_uniform(acehip_ctx* ctx, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, uint64_t seed, acehip_stream stream);
/* the same under a 256-bit key (the form key generation uses outside the ACEHIP_SEED test mode; the reference: Sample_uniform_poly over
 * its BLAKE2Xb PRNG, prng.c:33-69): coefficients 4t .. 4t+3 of limb position `pos` are the four 128-bit quarters of the ChaCha20 block
 * (RFC 8439 2.3) with this key, block counter t and nonce (pos, "UNIF", 0), each reduced mod the limb's prime (bias < 2^-60).
 * h_key: eight 32-bit words on the HOST. */
int acehip_sample_uniform_keyed(acehip_ctx* ctx, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, const uint32_t* h_key,
                                acehip_stream stream);
/* d_res[pos] = d_a[pos] * h_scalars[pos - pos0] mod prime(pos); h_scalars is a HOST array of n_limbs words
 * (Scalars_integer_multiply_poly polynomial.c:234-268, Scalar_integer_multiply_poly :198) */
int acehip_mul_scalars(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* the unfused pair of the generated code (eg_fhertlib_relin.inc:79-80):
 * Decomp (poly_eval.c:11 -> Decompose_poly polynomial.c:848): d_out[0..n2) = digit limbs of d_in;
 * Mod_up (poly_eval.c:19 -> Raise_rns_base_with_parts polynomial.c:877-925): d_digit (n2 limbs, NTT
 * domain) raised to level+K limbs of d_out.  Both return the digit size n2 (>0) or a negative error. */
int acehip_decomp(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_in, uint32_t level, uint32_t digit, acehip_stream stream);
int acehip_mod_up(acehip_ctx* ctx, uint64_t* d_out, const uint64_t* d_digit, uint32_t level, uint32_t digit, acehip_stream stream);

/* d_res[pos] = d_a[pos] + h_scalars[pos - pos0] mod prime(pos): adding a constant plaintext
 * (Add_const ckks_evaluator.c:116-128 with Encode_val_at_level ckks_encoder.c:464-530) */
int acehip_add_scalars(acehip_ctx* ctx, uint64_t* d_res, const uint64_t* d_a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
/* hoisting building blocks of Bootstrap (ckks_bootstrap_context.c:1237-1381):
 * Switch_key_precompute (polynomial.c:1224-1239): every digit of d_in (level limbs) raised to level+K limbs,
 *   d_ext laid out [num_decomp(level)][level+K][N];
 * Fast_switch_key_ext (ckks_evaluator.c:418-460): d_acc{0,1} = sum_d key{0,1}[d] * ext[d] over level+K
 *   limbs, no ModDown. */
int acehip_modup_digits(acehip_ctx* ctx, uint64_t* d_ext, const uint64_t* d_in, uint32_t level, acehip_stream stream);
/* The same with one output block per digit: h_ext is a HOST array of acehip_num_decomp(ctx, level) device pointers, each to
 * (level + K) limbs.  (The rt_ant shim swaps these blocks into the caller's polynomials instead of copying.) */
int acehip_modup_digits_to(acehip_ctx* ctx, uint64_t* const* h_ext, const uint64_t* d_in, uint32_t level, acehip_stream stream);
int acehip_key_inner_product(acehip_ctx* ctx, uint64_t* d_acc0, uint64_t* d_acc1, const uint64_t* d_key, const uint64_t* d_ext, uint32_t level, acehip_stream stream);
/* Fast_switch_key_ext followed by Mod_down of both accumulators (Fast_switch_key ckks_evaluator.c:391-460 from its inner product
 * on; generated code resnet20_cifar10_pre.onnx.inc:7011-7036 + the Mod_down pair behind it): d_out{0,1} = Mod_down( sum_d key{0,1}[d]
 * (*) ext[d] ), `level` limbs each.  h_ext, h_key: HOST arrays of n_digits (<= 8) device pointers -- raised digit d (level + K limbs,
 * NTT domain) and key part d ([2][L+K][N], as acehip_key_inner_product's d_key + d*2*(L+K)*N).  Results are those of
 * acehip_key_inner_product + acehip_mod_down2; at N = 2^16 the accumulators are never stored (the Mod_down passes form the sums where
 * they would load them).  acehip_keymac_fusable: 1 when that form will be used for (level, n_digits) under the current replica selection.
 * No aliasing: the outputs must not overlap each other, a raised digit or a key part (the last pass reads the digits while it writes
 * the outputs): ACEHIP_EINVAL. */
int acehip_keymac_mod_down2(acehip_ctx* ctx, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* const* h_ext, const uint64_t* const* h_key,
                            uint32_t n_digits, uint32_t level, acehip_stream stream);
int acehip_keymac_fusable(const acehip_ctx* ctx, uint32_t level, uint32_t n_digits);
/* test hook: the ACEHIP_KMAC_FUSE setting of this process (0 never, 1 where the transforms run as wide passes anyway, 2 always);
 * returns the previous one */
int acehip_debug_set_kmac_fuse(int mode);
/* Fast_rotate_ext (ckks_evaluator.c:539-575) adds P * c0 to the first accumulator before the automorphism: the same inner
 * product with d_acc0[i] += d_add0[i] * h_scalars[i] mod q_i on the q-limbs i < level (h_scalars: host array of `level`
 * residues, P mod q_i there), in the same pass. */
int acehip_key_inner_product_add(acehip_ctx* ctx, uint64_t* d_acc0, uint64_t* d_acc1, const uint64_t* d_key, const uint64_t* d_ext,
                                 uint32_t level, const uint64_t* d_add0, const uint64_t* h_scalars, acehip_stream stream);
/* The hoisted rotations of Rotate_iteration (ckks_bootstrap_context.c:1276-1290: one Switch_key_precompute, then Fast_rotate_ext with one
 * rotation key each): n_keys key inner products over the SAME raised digits d_ext in one pass over them.  h_acc0 / h_acc1 / h_keys: HOST
 * arrays of n_keys device pointers (outputs of level + K limbs; key sets [num_decomp][2][L+K][N]); d_add0 / h_scalars as in
 * acehip_key_inner_product_add (both NULL: no addend), applied to every rotation.  Results are those of n_keys single calls. */
int acehip_key_inner_products(acehip_ctx* ctx, uint64_t* const* h_acc0, uint64_t* const* h_acc1, const uint64_t* const* h_keys, uint32_t n_keys,
                              const uint64_t* d_ext, uint32_t level, const uint64_t* d_add0, const uint64_t* h_scalars, acehip_stream stream);
/* Baby-step giant-step inner products of Rotate_iteration (ckks_bootstrap_context.c:1326-1341: Mul_plaintext +
 * Add_ciphertext over one giant step, for every baby step): d_out{0,1}[i] = sum_{j<g} d_in{0,1}[j] (*) pt[i*g + j],
 * i < b, over the level+K limbs of PQ-extended ciphertexts, in ONE pass over the plaintext diagonals.  pt entries are
 * device polynomials with pt_q_limbs >= level q-limbs followed by K p-limbs (Derive_plain); a NULL entry is skipped.
 * The arrays themselves are host arrays of device pointers.  g, b <= 16, g*b <= 128. */
int acehip_bsgs_inner(acehip_ctx* ctx, uint64_t* const* d_out0, uint64_t* const* d_out1, const uint64_t* const* d_in0,
                      const uint64_t* const* d_in1, const uint64_t* const* d_pt, uint32_t g, uint32_t b, uint32_t pt_q_limbs,
                      uint32_t level, acehip_stream stream);
/* The same with the automorphisms of the hoisted rotations (Fast_rotate_ext ckks_evaluator.c:539-575 ends with
 * Automorphism_transform on both polynomials) applied while the inputs are read: h_in_auto[j] = automorphism index k
 * (acehip_auto_index) of input j, 0 = input j is taken as it is; d_in{0,1}[j] then hold the ciphertext BEFORE the
 * automorphism.  A rotated input must not alias an output. */
int acehip_bsgs_inner_rot(acehip_ctx* ctx, uint64_t* const* d_out0, uint64_t* const* d_out1, const uint64_t* const* d_in0,
                          const uint64_t* const* d_in1, const uint32_t* h_in_auto, const uint64_t* const* d_pt, uint32_t g, uint32_t b,
                          uint32_t pt_q_limbs, uint32_t level, acehip_stream stream);

/* Debug aid: with enable != 0 the pipeline entry points (acehip_key_switch, acehip_modup_digits[_to], acehip_mod_down[2],
 * acehip_rescale[2], acehip_encode[_batch]) record every range of CALLER memory they read or write (device pointer, words);
 * each call returns (up to cap of) what the calling thread's entry points recorded since the previous call and clears the log.
 * The rt_ant shim checks its declared-operand lists against it under ACEHIP_POISON=1. */
size_t acehip_debug_touches(int enable, const void** ptrs, size_t* words, size_t cap);

/* ---- replicas of the caller's polynomial memory: image batches and simulated ranks ----
 * The reference gets throughput from one OpenMP thread per image (rtlib/ant/dataset/resnet_cifar.main.inc:77-116); all
 * images run the same data-oblivious program on the same keys and weights.  The GPU form: the caller keeps its ciphertext
 * polynomials in ONE arena of which n_replicas copies exist, stride_bytes apart (replica r of the block at address p is at
 * p + r * stride_bytes), and every launch covers the selected replicas [rep0, rep0 + nrep): a device pointer argument that
 * lies inside replica 0 of the arena is moved along with the replica by the kernels, anything else (switch keys, twiddles,
 * automorphism tables, bootstrap diagonals, weight plaintexts) is shared by all replicas and read once per launch.  With the
 * default selection (0, 1) every entry point behaves exactly as without an arena.
 * workspace / hw_scratch: scratch of the pipelines (acehip_workspace_words() words) and of acehip_hw_batch (hw_scratch_limbs
 * limbs of N words), both INSIDE replica 0 of the arena so that every replica has its own; required when n_replicas > 1. */
typedef struct acehip_arena_cfg {
  void*    base;          /* replica 0 (device memory) */
  size_t   bytes;         /* size of one replica */
  size_t   stride_bytes;  /* distance between replicas, >= bytes */
  uint32_t n_replicas;
  void*    workspace;
  void*    hw_scratch;
  size_t   hw_scratch_limbs;
} acehip_arena_cfg;
int    acehip_ctx_set_arena(acehip_ctx* ctx, const acehip_arena_cfg* cfg);   /* cfg == NULL: back to no arena */
size_t acehip_workspace_words(const acehip_ctx* ctx);
int    acehip_ctx_select(acehip_ctx* ctx, uint32_t rep0, uint32_t nrep);     /* launches that follow cover these replicas */
/* host <-> device copies and fills that follow the selection: a destination inside the arena is written in every selected
 * replica, a source inside the arena is read from the first selected one.  upload / download synchronise the stream. */
int acehip_upload(acehip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, acehip_stream stream);
int acehip_download(acehip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, acehip_stream stream);
int acehip_fill(acehip_ctx* ctx, void* d_ptr, int value, size_t bytes, acehip_stream stream);
int acehip_copy(acehip_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, acehip_stream stream);

/* ---- limb-sharded execution (BASELINE configs[4]: RNS limbs spread over the GPUs of a node) ----
 * Rank r of `world` owns the limbs gi with gi % world == r (q_i: gi = i, p_j: gi = L + j) of every polynomial and
 * switch key, PACKED in ascending gi; ownership does not depend on the level.  One hybrid key-switch
 * (Decompose_modup polynomial.c:1241-1335 + Multiply_add :148-183 + Reduce_rns_base :928-967) is three local phases
 * around two all-gathers that the CALLER performs (RCCL over xGMI on a node: torch.distributed all_gather_into_tensor of
 * the send buffer, rank-major):
 *   phase1: owned q-limbs -> coefficient domain                          send [pad_q][N]      gather [world][pad_q][N]
 *   phase2: ModUp of every digit onto the owned limbs, NTT, key inner product, owned p-limbs of both accumulators ->
 *           coefficient domain                                           send [2][pad_p][N]   gather [world][2][pad_p][N]
 *   phase3: P -> owned q-limbs, NTT, (acc - conv) * P^-1                 out: owned q-limbs of both results, packed
 * Rescale (Rescale_poly :1097-1163): the owner of limb level-1 sends its coefficient-domain limb of c0 and c1 ([2][N],
 * broadcast by the caller), every rank applies it to its owned limbs.  Encode: the integer message (N words) is computed
 * once and broadcast, or computed redundantly; each rank reduces and transforms its own limbs.
 * Every call is a few batched launches on `stream` and never synchronises; results are bit-identical to the unsharded
 * entry points.  d_key_own: [num_q_parts][2][n_own][N], the rank's limbs of a switch key in packed order (owned
 * q-limbs of the full chain, then owned p-limbs). */
typedef struct acehip_shard acehip_shard;
acehip_shard* acehip_shard_create(acehip_ctx* ctx, uint32_t rank, uint32_t world);
void     acehip_shard_destroy(acehip_shard* shard);
uint32_t acehip_shard_num_q(const acehip_shard* shard, uint32_t level);  /* owned q-limbs below `level` */
uint32_t acehip_shard_num_p(const acehip_shard* shard);
uint32_t acehip_shard_pad_q(const acehip_shard* shard, uint32_t level);  /* most q-limbs any rank owns: slots per rank of exchange 1 */
uint32_t acehip_shard_pad_p(const acehip_shard* shard);
uint32_t acehip_shard_owned(const acehip_shard* shard, uint32_t level, uint32_t* q_idx_out, uint32_t* p_idx_out); /* returns num_q */
int acehip_shard_ks_phase1(acehip_shard* shard, uint64_t* d_send, const uint64_t* d_x_own, uint32_t level, acehip_stream stream);
int acehip_shard_ks_phase2(acehip_shard* shard, uint64_t* d_send2, const uint64_t* d_gathered, const uint64_t* d_x_own,
                           const uint64_t* d_key_own, uint32_t level, acehip_stream stream);
int acehip_shard_ks_phase3(acehip_shard* shard, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_gathered2, uint32_t level, acehip_stream stream);
int acehip_shard_rescale_send(acehip_shard* shard, uint64_t* d_send, const uint64_t* d_c0_own, const uint64_t* d_c1_own, uint32_t level, acehip_stream stream); /* 1: this rank is the sender */
int acehip_shard_rescale_apply(acehip_shard* shard, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_c0_own, const uint64_t* d_c1_own,
                               const uint64_t* d_last, uint32_t level, acehip_stream stream);
int acehip_encode_message(acehip_ctx* ctx, int64_t* d_msg, const void* d_vals, int kind, size_t len, uint32_t slots, double scaling_factor, acehip_stream stream);
int acehip_shard_encode_limbs(acehip_shard* shard, uint64_t* d_q_own, const int64_t* d_msg, double scaling_factor, uint32_t sf_degree, uint32_t level, acehip_stream stream);

/* ---- limb-sharded execution as a MODE of the context (BASELINE configs[4]: a whole generated program runs sharded) ----
 * Once enabled, EVERY entry point above works on the limbs its rank owns (gi % world == rank; polynomials keep the full
 * layout of the reference, the other limbs are simply not touched) and the three places where limbs have to meet -- the
 * sources of a ModUp (Decompose_modup polynomial.c:1241-1335), the P-limbs of a ModDown (Reduce_rns_base :928-967), the last
 * limb of a Rescale (Rescale_poly :1097-1163) / limb 0 of a ModRaise (ckks_bootstrap_context.c:1527-1551) -- exchange them
 * inside the call: every rank runs the same call sequence (SPMD), results are bit-identical to the unsharded library.
 *   acehip_ctx_shard_sim:  `world` simulated ranks in this process; rank r's limbs live in replica r of the arena
 *                          (acehip_ctx_set_arena with n_replicas >= world), exchanges are device copies.  For tests on one GPU.
 *   acehip_ctx_shard_rccl: this process is rank `rank` of `world`, one GPU each; an exchange is ONE RCCL collective on a stream of
 *                          its own (the owned limbs packed into a staging block, ncclAllGather over xGMI, unpacked on arrival;
 *                          ACEHIP_SHARD_PACKED=0: one grouped ncclBroadcast per limb, in place).  unique_id: the 128 bytes of acehip_rccl_unique_id()
 *                          of rank 0, handed to every rank by the caller (a file, torch.distributed, MPI ...).
 * acehip_shard_gather: limbs [pos0, pos0 + n_limbs) of a polynomial become valid on every rank (decode, serialisation). */
int      acehip_ctx_shard_sim(acehip_ctx* ctx, uint32_t world);
int      acehip_rccl_unique_id(void* out, size_t cap);   /* returns the number of bytes written (128) or a negative error */
int      acehip_ctx_shard_rccl(acehip_ctx* ctx, uint32_t rank, uint32_t world, const void* unique_id, size_t id_bytes);
int      acehip_shard_gather(acehip_ctx* ctx, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream stream);
uint32_t acehip_shard_world(const acehip_ctx* ctx);                 /* 1: not sharded */
uint32_t acehip_shard_rank(const acehip_ctx* ctx);                  /* first hosted rank */
uint32_t acehip_shard_owned_limbs(const acehip_ctx* ctx, uint32_t rank);   /* limbs of the full chain (L + K) rank owns */
/* exchange statistics of this context: steps[0] = exchange steps, steps[1] = limbs moved to this process, returns bytes moved */
uint64_t acehip_shard_traffic(const acehip_ctx* ctx, uint64_t* steps, int reset);
/* RCCL collectives issued for the exchange steps since the last reset of acehip_shard_traffic: one per step (the limbs of a step
 * travel packed: one ncclAllGather, or one ncclBroadcast when a single rank owns them all), one per limb with ACEHIP_SHARD_PACKED=0 */
uint64_t acehip_shard_collectives(const acehip_ctx* ctx);
/* The exchange schedule of one operation, for checking it without a GPU (works on a host-only context): which limb positions
 * are exchanged at each step and which rank sends them.  op: 0 ModUp of all digits, 1 ModDown, 2 Rescale, 3 ModRaise.
 * Writes up to cap entries {step, position, root rank}; returns the number of entries. */
int acehip_shard_schedule(const acehip_ctx* ctx, uint32_t world, int op, uint32_t level, uint32_t* out_step, uint32_t* out_pos,
                          uint32_t* out_root, size_t cap);

/* The constant tables of the matrix-core base conversion (the sums of Reduce_rns_base polynomial.c:928-967 /
 * Decompose_modup :1302-1320 as int8 matrix products), for checking them without a GPU (works on a host-only context).
 * digit >= 0: the ModUp of that digit at `level`; digit < 0: the ModDown at `level`.  dims[4] = {sources n_in, outputs n_out,
 * k-steps, tiles of 16 outputs}.  frag: [tile][k-step][digit b < 9][lane < 64][16 bytes] -- byte e of lane (r = lane & 15,
 * g = lane >> 4) is digit b (7 bits) of  hat(i, j) * 2^(8a) mod t_j  for source i = 8*step + 2g + (e >> 3), byte a = e & 7 of its
 * residue, output j = 16*tile + r (zero past n_in / n_out); off: [16*tiles][9] = 128 * (sum over the sources' bytes of that digit).
 * Returns the bytes of frag (copies are made when the capacities, in elements, suffice), negative on error. */
long acehip_conv_mfma_tables(const acehip_ctx* ctx, uint32_t level, int32_t digit, uint8_t* frag, size_t frag_cap, uint32_t* off,
                             size_t off_cap, uint32_t* dims);

/* algorithmic HBM bytes of one acehip_key_switch at `level` (8N(l + 2b(l+K) + 2l)) */
uint64_t acehip_key_switch_bytes(const acehip_ctx* ctx, uint32_t level);

#ifdef __cplusplus
}
#endif
#endif /* ACEHIP_H */
