// rt_internal.hpp -- internals of the rt_ant drop-in shim (libFHErt_ant.so).
//
// The shim is the host side of the boundary: it mirrors the reference provider API
// (include/rt_ant/ant_api.h) and forwards every polynomial operation to the HIP library through the
// C ABI of include/acehip.h.  No arithmetic on coefficients happens on the host except the FP64
// canonical embedding of encode/decode and the final CRT reconstruction of decode, which the
// reference also does on the CPU (ckks_encoder.c:199-297, :649-703).
#pragma once
#include <complex>
#include <deque>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "acehip.h"
#include "common/pt_mgr.h"
#include "common/rt_api.h"
#include "common/rt_stat.h"
#include "rt_ant/ant_api.h"
#include "rt_ant/rt_api.h"
#include "rt_rng.hpp"

#define RT_ASSERT(cond, ...)                              \
  do {                                                    \
    if (!(cond)) {                                        \
      fprintf(stderr, "%s:%d: ", __FILE__, __LINE__);     \
      fprintf(stderr, __VA_ARGS__);                       \
      fprintf(stderr, "\n");                              \
      abort();                                            \
    }                                                     \
  } while (0)
// Every device call of the shim goes through HIPCHK, which first hands over the per-limb Hw_* calls
// that are still queued (rt_poly.cpp hw_queue): the device sees all work in program order.
namespace rt {
void hw_flush();
void hw_flush_site(const char* file, int line);
// memory a direct launch reads or writes (words of 8 bytes)
struct Touch {
  const void* p;
  size_t words;
  bool ro = false;  // the launch only READS the range: queued ops that read it too need not run first, raised digits in it stay valid
};
void hw_flush_touching(const char* file, int line, const Touch* touch, size_t n);
// ACEHIP_POISON=1 (debug): aborts when the launch issued since hw_flush_touching touched pool memory outside `touch`
void check_declared(const char* file, int line, const Touch* touch, size_t n);
}  // namespace rt
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    rt::hw_flush_site(__FILE__, __LINE__);                                                                        \
    int rc_ = (expr);                                                                      \
    if (rc_ < 0) {                                                                         \
      fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr, acehip_last_error()); \
      abort();                                                                             \
    }                                                                                      \
  } while (0)

// HIPCHK for a launch whose device operands are ALL listed ({pointer, words}, ...; library-internal workspaces and
// tables need not be): zero fills of other memory that nothing queued has consumed yet may then stay deferred across
// the launch (rt_poly.cpp "lazy zero fills") instead of being written out now and read back by their first consumer.
// Leaving an operand out makes the launch see stale memory -- when in doubt use HIPCHK, which defers nothing.
#define HIPCHK_T(expr, ...)                                                                  \
  do {                                                                                     \
    const rt::Touch touch_[] = {{nullptr, 0}, __VA_ARGS__};                                \
    rt::hw_flush_touching(__FILE__, __LINE__, touch_, sizeof touch_ / sizeof touch_[0]);   \
    int rc_ = (expr);                                                                      \
    if (rc_ < 0) {                                                                         \
      fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr, acehip_last_error()); \
      abort();                                                                             \
    }                                                                                      \
    rt::check_declared(__FILE__, __LINE__, touch_, sizeof touch_ / sizeof touch_[0]);      \
  } while (0)

// Device work that provably touches nothing the queue names (it only writes a block that was allocated, from memory
// no queued op can reference, after the caller checked) may be launched AHEAD of the queue: no hw_flush.
#define HIPCHK_NOFLUSH(expr)                                                               \
  do {                                                                                     \
    int rc_ = (expr);                                                                      \
    if (rc_ < 0) {                                                                         \
      fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr, acehip_last_error()); \
      abort();                                                                             \
    }                                                                                      \
  } while (0)

namespace rt {

using u64 = uint64_t;
using u32 = uint32_t;
using cplx = std::complex<double>;

struct SwitchKeyStore {
  SWITCH_KEY key;                 // what Swk() returns; _parts[d]._pk0/_pk1 point into `data`
  std::vector<PUBLIC_KEY> parts;
  u64* data = nullptr;            // [dnum][2][L+K][N] on the device (layout of acehip_key_switch)
};

struct Context {
  acehip_ctx* hip = nullptr;
  CKKS_PARAMS* prm = nullptr;
  u32 N = 0, L = 0, K = 0, dnum = 0, alpha = 0, sf_bits = 0, q0_bits = 0;
  size_t hamming = 0;
  double sf = 0;                  // default scaling factor 2^sf_bits
  std::vector<MODULUS> qmod, pmod;
  std::vector<u64> primes;        // [L+K]
  // keys on the device
  u64* sk_ntt = nullptr;          // [L+K][N] NTT domain
  u64* pk0 = nullptr;             // [L][N]
  u64* pk1 = nullptr;
  SwitchKeyStore relin;
  std::map<u32, SwitchKeyStore*> auto_keys;   // automorphism index -> key
  std::map<int32_t, u32> rot2auto;            // rotation -> automorphism index
  std::vector<int64_t> sk_coef;               // ternary secret, host copy (signed)
  // Randomness (rt_rng.hpp).  drbg: every stream is ChaCha20 under master_key (256 bits from getrandom) -- one per key identity
  // (key_rng) and one per encrypting thread (rng; enc_streams counts them, primary context, under shared_mu).  Otherwise the TEST
  // mode of ACEHIP_SEED: key_seed / seed_rng as in rounds 1-4, bit for bit (the committed fixtures depend on it).
  Rng rng;                                    // this thread's encryption randomness (Acehip_rt_seed_encryptor re-seeds it: test mode)
  bool drbg = false;
  u32 master_key[8] = {};
  u64 enc_streams = 0;
  u64 key_seed = 0;                           // test mode, primary context: every key's generator derives from it (rt_context.cpp key_rng)
  Rng seed_rng;                               // test mode, primary context: seeds of attaching threads, used under shared_mu
  // FFT tables for the canonical embedding of decode (ntt.c:587-610): m = 2N
  std::vector<cplx> fft_rou;      // e^{2 pi i k / 2N}
  std::vector<u32> rot_group;     // 5^i mod 2N
  // statistics
  // (the weight-plaintext statistics of Finalize_context's report are process-wide: count_weight_plain below)
  bool keys_loaded = false, keys_strict = false;  // key set came from a container / a key missing from it is an error
  std::string keys_save_path;                    // Finalize_context writes the key set here (ACEHIP_KEYS_FILE, rt_serial.cpp)
  bool secondary = false;         // a thread's view of the primary context (shares its keys, owns its acehip_ctx)
  bool profile = false;                          // ACEHIP_PROFILE=1: host-side timers below are printed
  double t_encode = 0, t_main = 0, t_issue = 0, t_bootstrap = 0;
  size_t n_bootstrap = 0;
  size_t n_encode = 0, n_encode_ahead = 0;       // encodes / encodes launched ahead of the per-limb queue
  // Weight-plaintext prefetch (rt_io.cpp pt_encode).  An FHE program is data-oblivious: every image makes the same
  // Pt_from_msg calls in the same order.  The calls of this thread's first image are recorded; from the second image on
  // the plaintexts of the next calls are encoded together, ahead of their use (acehip_encode_batch: one set of launches
  // for up to 8 weights), and handed out as the calls arrive.  A call that differs from the record ends the prediction for
  // the rest of that image.  (The reference prefetches weight plaintexts as well: pt_mgr.c:128-159.)
  struct PtCall {
    u32 index, scale, level;
    size_t len;
    bool operator==(const PtCall& o) const { return index == o.index && scale == o.scale && level == o.level && len == o.len; }
  };
  std::vector<PtCall> pt_trace;
  bool pt_trace_done = false, pt_predict = false;
  size_t pt_pos = 0;                             // index into pt_trace of the next expected call
  std::deque<u64*> pt_ring;                      // encoded blocks for calls pt_pos, pt_pos + 1, ...
  size_t n_encode_prefetched = 0, n_encode_batches = 0;
  // Image batch (Acehip_rt_set_batch / ACEHIP_BATCH): `batch` images run through every launch of this thread; each has its
  // own replica of the pool arena, keys / bootstrap diagonals / weight plaintexts are shared (rt_poly.cpp "pool").
  u32 batch = 1;
  // Limb-sharded execution (BASELINE configs[4]): every rank runs the same program on the limbs it owns.  shard_sim: the
  // ranks are simulated in this process (ACEHIP_SHARD_SIM=G: replica r of the arena holds rank r's limbs); otherwise this
  // process is rank shard_rank of shard_world (one GPU each, RCCL; RANK / WORLD_SIZE with ACEHIP_SHARD=1).
  u32 shard_world = 1, shard_rank = 0;
  bool shard_sim = false;
};
void pt_image_boundary();                        // a new input arrives (Prepare_input): the recorded call sequence restarts

extern thread_local Context* g_ctx;  // this thread's context (its own acehip_ctx, counters, copies of the parameters)
extern Context* g_primary;           // the context Prepare_context built: owner of the keys every thread uses
// guards what threads grow lazily and share: rotation keys, bootstrap precomputation, weight/plaintext caches
std::recursive_mutex& shared_mu();
// device memory that outlives the allocating thread (keys, bootstrap plaintexts): not from the thread's pool
u64* shared_alloc(size_t words, bool zero);
u64* shared_alloc_key(size_t n_polys);  // a switch key's memory: owner-only limbs on a rank of limb-sharded execution (rt_poly.cpp)
struct SharedAllocScope {  // every dalloc of this thread inside the scope is a shared_alloc
  SharedAllocScope();
  ~SharedAllocScope();
};
void thread_release();               // give back this thread's context / pool / queue (secondary threads)
Context& ctx();

// ---- device memory pool (stream-ordered reuse; generated code does thousands of Alloc/Free) ----
// nq (limb-sharded execution: who owns which limb of a zero fill / copy): the block is a polynomial of nq q-limbs (primes
// 0..nq-1) followed by p-limbs; NQ_ANY: no such structure
constexpr u32 NQ_ANY = 0xffffffffu;
u64* dalloc(size_t words, bool zero, u32 nq = NQ_ANY);
void dfree(u64* p);
bool block_is_replicated(const u64* p);  // a pool block of which every image of the batch has its own copy
bool block_is_uniform(const u64* p);     // a pool block shared by the images of the batch
size_t arena_peak_bytes();        // highest address of the slab in use so far (0: no slab yet)
size_t arena_live_peak_bytes();   // most bytes in live blocks so far
size_t arena_bytes();
// ---- which replicas of the arena the launches of this thread cover (image batches) ----
u32 batch_size();
// One weight plaintext of `bytes` bytes was encoded (or served from the prefetch / the cache) for the images of the calling thread's
// launch.  The reference keeps ONE counter in its encoder, shared by all image threads (Append_weight_plain ckks_encoder.h:48-52, called
// per image: plain_eval.c:21), and prints it at Finalize_context (context.c:111-117; parsed by scripts/perf.py:247-250): the figure is
// per PROCESS and per IMAGE.  Here a call serves every image of the stream's batch and image streams are threads with contexts of their
// own, so the counters are process-wide atomics and a call counts once per image it serves.
void count_weight_plain(size_t bytes);
void weight_plain_totals(size_t* cnt, size_t* bytes, bool reset);
bool uniform_alloc_on();      // allocations of this thread currently come from the shared pool
u32 current_rep0();
u32 current_nrep();
bool in_image_scope();
u32 selected_image();
bool batch_aware();      // false: ACEHIP_BATCH in the environment of a program that never selects an image (Prepare_input fills all)
void set_batch_aware();
void select_image(u32 k);
void set_launch_mode(u32 rep0, u32 nrep);
struct UniformScope {  // work whose results every image shares (keys, bootstrap tables ...): one replica, blocks outside the arena
  u32 rep0, nrep;
  UniformScope();
  ~UniformScope();
};
struct ImageScope {    // work on ONE image of the batch (its input, its output)
  u32 rep0, nrep;
  bool was;
  explicit ImageScope(u32 k);
  ~ImageScope();
};
struct UniformAlloc {  // allocations inside come from the shared (uniform) pool; launches are not affected
  bool on;
  explicit UniformAlloc(bool on);
  ~UniformAlloc();
};
struct SelectGuard {   // a direct launch that covers other replicas than the thread's current mode; does NOT hand the queue over
  u32 rep0, nrep;
  bool active;
  SelectGuard(u32 r0, u32 n);
  ~SelectGuard();
};
void pool_release_all();
size_t pool_bytes_in_use();

// ---- polynomial helpers on POLYNOMIAL structs (device data) ----
// Alloc_poly_data polynomial.h:54; zero = false only where the caller overwrites every limb right away
void poly_alloc(POLYNOMIAL* p, u32 N, size_t nq, size_t np, bool zero = true);
void poly_free(POLYNOMIAL* p);                                         // Free_poly_data   :71
void poly_init_like(POLYNOMIAL* res, POLYNOMIAL* like);                // Init_poly        :331
void poly_copy(POLYNOMIAL* res, POLYNOMIAL* src);                      // Copy_polynomial
u64* q_limbs(POLYNOMIAL* p);
u64* p_limbs(POLYNOMIAL* p);
enum class Op { Add, Sub, Mul, MulAdd };
// res = a (op) b over the q-limbs of res (+ p-limbs when with_p)
void poly_ew(Op op, POLYNOMIAL* res, POLYNOMIAL* a, POLYNOMIAL* b, bool with_p);
void poly_ntt(POLYNOMIAL* p, bool inverse);                            // Conv_poly2ntt_inplace / ntt2poly
void poly_rotate(POLYNOMIAL* res, POLYNOMIAL* a, u32 auto_idx);        // Rotate_poly (NTT domain)
void poly_from_small(POLYNOMIAL* p, const std::vector<int64_t>& vals); // Transform_values_at_level(without_mod)
void sync();
// queue `n_limbs` consecutive limbs of a per-limb op (ACEHIP_HW_*) instead of launching it now
void hw_stats_print();
void hw_flush_sites_print();
bool hw_queue_empty();   // nothing queued (a held-back Mod_down / Rescale does not count)
void hw_pending_flush(); // issue a held-back Mod_down / Rescale now
// the caller is about to rewrite limbs [out, out + n_limbs) completely with a direct launch: queued fills of them are dead
void hw_cancel_fills(const u64* out, size_t n_limbs);
void hw_queue(u32 op, u32 prime_gi, u64* res, const u64* a, const void* b, size_t n_limbs = 1);
// queued multi-limb forms, argument meaning as acehip_modadd & co: limbs [pos0, pos0+n) of polynomials extended
// at `level` (limb p < level is prime p, the others are p primes); scalars[i] belongs to limb pos0+i
void q_ew(u32 op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n);
void q_scalars(u32 op, u64* r, const u64* a, const u64* scalars, u32 level, u32 pos0, u32 n);
void q_rotate(u64* r, const u64* a, const uint32_t* perm, u32 level, u32 pos0, u32 n);
void fill_zero(u64* p, size_t words, u32 nq = NQ_ANY);   // queued when whole limbs, else memset
// queued when whole disjoint limbs, else d2d copy; limb i of the copy is limb first + i of a polynomial with nq q-limbs
void copy_limbs(u64* dst, const u64* src, size_t words, u32 nq = NQ_ANY, u32 first = 0);
double wall_s();

// ---- sampling (random_sample.c) ----
void sample_triangle(std::vector<int64_t>& v, Rng& rng);                         // :78-97
void sample_ternary(std::vector<int64_t>& v, size_t hamming_weight, Rng& rng);   // :99-150
constexpr u64 KEY_TAG_SECRET = 1, KEY_TAG_PUBLIC = 2, KEY_TAG_RELIN = 3, KEY_TAG_AUTO = 1ull << 34;
Rng key_rng(u64 tag);  // the generator of one key: a function of the context's master key (test mode: key seed) and the key's identity
// limbs [pos0, pos0 + n) of a polynomial extended at `level`, uniform in [0, q): the public `a` of a key, drawn on the device
void sample_uniform_dev(u64* d, u32 level, u32 pos0, u32 n, Rng& rng);

// ---- keys (ckks_key_generator.c) ----
void generate_keys();
void shard_connect_if_asked();   // rt_context.cpp: ACEHIP_SHARD=1 joins the RCCL communicator of the launcher's ranks
SwitchKeyStore* make_switch_key(const u64* new_key_ntt /*[L+K][N]*/, const u64* old_key_ntt, u64 tag /* KEY_TAG_* */);
u32 ensure_rot_key(int32_t rotation);   // Insert_rot_map :290; returns automorphism index
SwitchKeyStore* ensure_auto_key(u32 auto_idx);
void free_switch_key(SwitchKeyStore* k);
int save_keys(const char* path);   // rt_serial.cpp: "ACEHKEY1" container
int load_keys(const char* path);
int save_eval_keys(const char* path);

// ---- encode / decode (ckks_encoder.c) ----
void embedding(std::vector<cplx>& vals);                               // ntt.c:678-711
const void* stage_to_device(const void* src, size_t bytes);           // async H2D through a pinned ring
void stage_release();
void encode_device(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt);
void encode_vector(PLAINTEXT* res, const cplx* values, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt);
void encode_value(PLAINTEXT* res, double value, u32 level, u32 sf_degree);
void encode_device_with_scale(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, double scale, u32 p_cnt);
void encode_value_with_scale(PLAINTEXT* res, double value, u32 level, double scale);   // rt_valid.cpp
void decode(std::vector<cplx>& out, PLAINTEXT* plain);
void encrypt(CIPHERTEXT* res, PLAINTEXT* plain);                       // ckks_encryptor.c:20-95
void decrypt(PLAINTEXT* res, CIPHERTEXT* ciph);                        // ckks_decryptor.c:19-65
void init_plaintext(PLAINTEXT* p, u32 slots, size_t nq, size_t np, double sf, u32 sf_degree, bool zero = true);

// ---- RTLIB_TIMING_OUTPUT table (rt_timing.cpp; ids in the order of rtlib/include/common/rtlib_timing.h:28-78) ----
typedef RTLIB_TIMING_ID RtmId;  // include/common/rtlib_timing.h (seen through rt_ant/rt_ant.h), ids in the reference's order
bool rtm_enabled();
void rtm_add(int id, uint64_t ns);
uint64_t rtm_now();
void rtm_report();
struct RtmScope {  // times the enclosing block when RTLIB_TIMING_OUTPUT is set; device_work: synchronise before stopping
  int id;
  uint64_t t0;
  bool sync_at_end;
  explicit RtmScope(int id, bool device_work = true);
  ~RtmScope();
};

void bootstrap_setup_if_needed();
void bootstrap_precom_slots(u32 num_slots);  // Bootstrap_precom(num_slots) context.c:162-185: tables and keys of one slot count
void bootstrap_release();
namespace ev { void clear_monomial_cache(); }

}  // namespace rt
