// rt_encode.cpp -- CKKS encode / decode / encrypt / decrypt.
// Reference: src/util/ckks_encoder.c:199-297 (Encode_impl, 64-bit path), :464-530 (Encode_val_at_level),
// :649-703 (Decode), src/util/ntt.c:678-753 (Embedding / Embedding_inv), src/util/ckks_encryptor.c:20-95,
// src/util/ckks_decryptor.c:19-65, src/ckks/plain_eval.c:17-58.
// Encode runs entirely on the device (acehip_encode: FP64 inverse embedding in the reference's butterfly
// order without FMA contraction, then RNS reduction, scaling, NTT); decode's forward embedding and the exact
// CRT reconstruction stay on the host as in the reference (this file is compiled with -ffp-contract=off).
#include <cmath>
#include <cstring>

#include "rt_internal.hpp"

namespace rt {

static inline u32 bitrev(u32 v, u32 width) {
  u32 r = 0;
  for (u32 i = 0; i < width; ++i) r |= ((v >> i) & 1u) << (width - 1 - i);
  return r;
}
static void bit_reverse_vec(std::vector<cplx>& v) {
  const size_t n = v.size();
  u32 w = 0;
  while ((1u << w) < n) ++w;
  std::vector<cplx> t(n);
  for (size_t i = 0; i < n; ++i) t[i] = v[bitrev((u32)i, w)];
  v.swap(t);
}
static inline cplx cmul(const cplx& a, const cplx& b) {  // (ac - bd) + (ad + bc)i, C99 operand order
  return cplx(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real());
}

// Embedding ntt.c:678-711
void embedding(std::vector<cplx>& vals) {
  Context& c = ctx();
  const size_t n = vals.size(), m = 2ull * c.N;
  u32 logn = 0;
  while ((1u << logn) < n) ++logn;
  bit_reverse_vec(vals);
  for (u32 logm = 1; logm <= logn; ++logm) {
    const size_t idx_mod = 1ull << (logm + 2), gap = m / idx_mod, num = 1ull << (logm - 1);
    for (size_t j = 0; j < n; j += (1ull << logm)) {
      for (size_t i = 0; i < num; ++i) {
        const size_t e = j + i, o = j + i + num;
        const size_t rou_idx = (c.rot_group[i] % idx_mod) * gap;
        const cplx f = cmul(c.fft_rou[rou_idx], vals[o]);
        const cplx plus = vals[e] + f, minus = vals[e] - f;
        vals[e] = plus;
        vals[o] = minus;
      }
    }
  }
}

// Init_plaintext plaintext.h:114-129
void init_plaintext(PLAINTEXT* p, u32 slots, size_t nq, size_t np, double sf, u32 sf_degree, bool zero) {
  Context& c = ctx();
  p->_scaling_factor = sf;
  p->_sf_degree = sf_degree;
  p->_slots = slots;
  POLYNOMIAL* poly = &p->_poly;
  // image batches: a block every image shares (a weight plaintext) and a block every image has a copy of are not
  // interchangeable -- a shell that holds the other kind gets a fresh block (the old one stays intact for queued readers)
  if (poly->_data != nullptr && batch_size() > 1 && (uniform_alloc_on() ? block_is_replicated((u64*)poly->_data) : block_is_uniform((u64*)poly->_data)))
    poly_free(poly);
  if (poly->_data == nullptr || (poly->_num_primes + poly->_num_primes_p) == 0) {
    if (poly->_data) poly_free(poly);
    poly_alloc(poly, c.N, nq, np, zero);
  } else {
    RT_ASSERT(poly->_num_primes == nq && poly->_num_primes_p == np, "unmatched size");
  }
}

static u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((unsigned __int128)a * b) % m); }

// Host values -> device through a small pinned ring so that the copy is asynchronous on the compute
// stream (a pageable hipMemcpy would drain the whole launch queue on every encode).
namespace {
struct StageRing {
  static constexpr int SLOTS = 8;
  size_t slot_bytes = 0;
  char* host = nullptr;     // pinned, SLOTS * slot_bytes
  char* dev = nullptr;
  void* ev[SLOTS] = {};
  int next = 0;
};
thread_local StageRing g_ring;
}  // namespace

const void* stage_to_device(const void* src, size_t bytes) {
  Context& c = ctx();
  StageRing& r = g_ring;
  const size_t need = (size_t)c.N * 8;  // N/2 complex doubles
  RT_ASSERT(bytes <= need, "staging buffer too small");
  if (r.host == nullptr || r.slot_bytes < need) {
    stage_release();
    r.slot_bytes = need;
    r.host = (char*)acehip_malloc_host(need * StageRing::SLOTS);
    r.dev = (char*)acehip_malloc(need * StageRing::SLOTS);
    RT_ASSERT(r.host && r.dev, "staging allocation failed: %s", acehip_last_error());
    for (auto& e : r.ev) e = acehip_event_create();
  }
  const int k = r.next;
  r.next = (k + 1) % StageRing::SLOTS;
  // (the staging ring is no polynomial memory: these three touch no limb)
  HIPCHK_T(acehip_event_sync(r.ev[k]));  // the copy that last read this pinned slot has finished
  memcpy(r.host + (size_t)k * r.slot_bytes, src, bytes);
  HIPCHK_T(acehip_memcpy_h2d_async(r.dev + (size_t)k * r.slot_bytes, r.host + (size_t)k * r.slot_bytes, bytes, nullptr));
  HIPCHK_T(acehip_event_record(r.ev[k], nullptr));
  return r.dev + (size_t)k * r.slot_bytes;
}
void stage_release() {
  StageRing& r = g_ring;
  if (r.host) {
    sync();
    for (auto& e : r.ev) acehip_event_destroy(e);
    acehip_free_host(r.host);
    acehip_free(r.dev);
  }
  r = StageRing{};
}

// Encode_impl ckks_encoder.c:199-297 on device-resident values (kind: 0 float, 1 double, 2 complex double)
static void encode_device_impl(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt,
                               double scale);
void encode_device(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt) {
  encode_device_impl(res, d_vals, kind, len, level, slots, sf_degree, p_cnt, 0.0);
}
// Encode_impl_with_scale ckks_encoder.c:301-378: an explicit scale; the plaintext's scale degree is the smallest whose power of the
// scaling factor reaches it (:330-334)
void encode_device_with_scale(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, double scale, u32 p_cnt) {
  RT_ASSERT(scale > 0, "invalid scale for encode");
  u32 sf_degree = (u32)floor(scale / ctx().sf);
  if (scale > ctx().sf * sf_degree) sf_degree++;
  encode_device_impl(res, d_vals, kind, len, level, slots, sf_degree, p_cnt, scale);
}
static void encode_device_impl(PLAINTEXT* res, const void* d_vals, int kind, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt,
                               double scale) {
  RtmScope rtm(RTM_ENCODE_ARRAY);
  Context& c = ctx();
  RT_ASSERT(res, "null plaintext");
  const double t0 = c.profile ? wall_s() : 0;
  const u32 N = c.N;
  if (slots == 0) slots = N / 2;
  if (level == 0) level = c.L;
  RT_ASSERT(level <= c.L, "level should not be larger than mul_depth + 1");
  RT_ASSERT(len <= slots, "slot size is too small");
  RT_ASSERT(slots <= N / 2, " slot size > N/2 ");
  RT_ASSERT(sf_degree >= 1 || scale > 0, "invalid scaling factor for encode");
  POLYNOMIAL* poly = &res->_poly;
  // Generated conv loops encode one weight plaintext per tap into the SAME PLAINTEXT shell, between the per-limb
  // multiply-accumulates of consecutive taps (resnet20_cifar10_pre.onnx.inc:1486-1503).  Flushing the per-limb queue
  // for every encode would send each tap's accumulator through memory; instead the plaintext gets a fresh block (the
  // old one stays intact for the queued ops that read it: renaming) and the encode, which writes nothing else, is
  // launched ahead of the queue.  The taps of a whole output channel then form one accumulation chain per limb.
  const bool ahead = !hw_queue_empty();
  // An image batch shares its weight plaintexts (the reference's threads share them too: pt_mgr.c:182): the block comes from
  // the shared pool and the encode covers one replica.  The input image (Prepare_input: ImageScope) and everything made
  // inside a UniformScope take the thread's current mode as it is.
  const bool shared_pt = batch_size() > 1 && !in_image_scope() && !uniform_alloc_on();
  UniformAlloc ua(shared_pt);
  c.n_encode++;
  c.n_encode_ahead += ahead;
  if (ahead && poly->_data != nullptr) {
    RT_ASSERT((poly->_num_primes + poly->_num_primes_p) == 0 || (poly->_num_primes == level && poly->_num_primes_p == p_cnt),
              "unmatched size");  // init_plaintext's check
    poly_free(poly);  // into the pool's limbo until the queue has been issued
  }
  init_plaintext(res, slots, level, p_cnt, scale > 0 ? scale : pow(c.sf, (double)sf_degree), sf_degree, false);  // encode writes every limb
  const Touch touch[3] = {{nullptr, 0}, {q_limbs(poly), (size_t)level * N}, {p_cnt ? p_limbs(poly) : nullptr, (size_t)p_cnt * N}};
  if (ahead) hw_pending_flush();
  else hw_flush_touching(__FILE__, __LINE__, touch, 3);  // (d_vals: staging ring / weights)
  {
    SelectGuard one_replica(shared_pt ? 0 : current_rep0(), shared_pt ? 1 : current_nrep());
    if (scale > 0)
      HIPCHK_NOFLUSH(acehip_encode_with_scale(c.hip, q_limbs(poly), p_cnt ? p_limbs(poly) : nullptr, d_vals, kind, len, slots, scale, level,
                                              p_cnt, nullptr));
    else
      HIPCHK_NOFLUSH(acehip_encode(c.hip, q_limbs(poly), p_cnt ? p_limbs(poly) : nullptr, d_vals, kind, len, slots, c.sf, sf_degree,
                                   level, p_cnt, nullptr));
  }
  if (!ahead) check_declared(__FILE__, __LINE__, touch, 3);
  poly->_is_ntt = true;
  if (c.profile) c.t_encode += wall_s() - t0;
}

void encode_vector(PLAINTEXT* res, const cplx* values, size_t len, u32 level, u32 slots, u32 sf_degree, u32 p_cnt) {
  static_assert(sizeof(cplx) == 16, "std::complex<double> is two doubles");
  encode_device(res, len ? stage_to_device(values, len * sizeof(cplx)) : nullptr, 2, len, level, slots, sf_degree, p_cnt);
}

// Encode_val_at_level ckks_encoder.c:464-530 (+ Scale_back_up_by_approxfactor :406-460)
void encode_value(PLAINTEXT* res, double value, u32 level, u32 sf_degree) {
  RtmScope rtm(RTM_ENCODE_VALUE);
  Context& c = ctx();
  RT_ASSERT(res && sf_degree, "invalid plaintext / scaling factor degree");
  if (level == 0) level = c.L;
  RT_ASSERT(level <= c.L, "level should not be larger than mul_depth + 1");
  const u32 N = c.N;
  init_plaintext(res, N / 2, level, 0, pow(c.sf, (double)sf_degree), sf_degree);
  const int MAX_BITS_IN_WORD = 61, MAX_LOG_STEP = 60;
  const int32_t log_sf = (int32_t)ceil(log2(fabs(value * c.sf)));
  const int32_t log_valid = log_sf <= MAX_BITS_IN_WORD ? log_sf : MAX_BITS_IN_WORD;
  int32_t log_approx = log_sf - log_valid;
  const double approx_factor = pow(2, log_approx);
  const double scaled = value / approx_factor * c.sf + 0.5;
  RT_ASSERT(scaled <= 9.2e18 && scaled >= -9.2e18, "encode overflow, please choose a smaller scaling factor");
  const int64_t val = (int64_t)scaled;
  const int64_t sfs = (int64_t)(c.sf + 0.5);
  std::vector<u64> consts(level);
  for (u32 i = 0; i < level; ++i) {
    const u64 q = c.primes[i];
    int64_t r = val % (int64_t)q;
    if (r < 0) r += (int64_t)q;
    u64 rv = (u64)r;
    for (u32 j = 1; j < sf_degree; ++j) rv = mulmod(rv, (u64)sfs % q, q);
    consts[i] = rv;
  }
  if (log_approx > 0) {
    int32_t log_step = log_approx <= MAX_BITS_IN_WORD ? log_approx : MAX_BITS_IN_WORD;
    std::vector<u64> approx(level);
    for (u32 i = 0; i < level; ++i) approx[i] = (1ull << log_step) % c.primes[i];
    int32_t rest = log_approx - log_step;
    while (rest > 0) {
      log_step = rest <= MAX_LOG_STEP ? rest : MAX_LOG_STEP;
      for (u32 i = 0; i < level; ++i) approx[i] = mulmod(approx[i], (1ull << log_step) % c.primes[i], c.primes[i]);
      rest -= log_step;
    }
    for (u32 i = 0; i < level; ++i) consts[i] = mulmod(consts[i], approx[i], c.primes[i]);
  }
  // every coefficient of limb i equals consts[i] (a constant polynomial in the NTT domain)
  fill_zero((u64*)q_limbs(&res->_poly), (size_t)level * N, level);
  q_scalars(ACEHIP_HW_ADDC, q_limbs(&res->_poly), q_limbs(&res->_poly), consts.data(), level, 0, level);
  res->_poly._is_ntt = true;
}

// exact CRT reconstruction of coefficient i over `level` primes, centred, as a double
// (Reconstruct_rns_poly_to_values polynomial.c:467-499 + mpz_get_d in Decode :669-683)
struct Crt {
  u32 level;
  std::vector<u64> inv;                  // (Q/q_i)^-1 mod q_i
  std::vector<std::vector<u64>> hat;     // Q/q_i as little-endian words
  std::vector<u64> Q, halfQ;
  size_t words;
};
static void mp_mul_small(std::vector<u64>& a, u64 b) {
  u64 carry = 0;
  for (auto& w : a) {
    unsigned __int128 t = (unsigned __int128)w * b + carry;
    w = (u64)t;
    carry = (u64)(t >> 64);
  }
  if (carry) a.push_back(carry);
}
static int mp_cmp(const std::vector<u64>& a, const std::vector<u64>& b) {
  for (size_t i = a.size(); i-- > 0;)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static Crt make_crt(u32 level) {
  Context& c = ctx();
  Crt k;
  k.level = level;
  k.Q.assign(1, 1);
  for (u32 i = 0; i < level; ++i) mp_mul_small(k.Q, c.primes[i]);
  k.words = k.Q.size() + 1;
  k.Q.resize(k.words, 0);
  k.halfQ = k.Q;
  for (size_t i = 0; i < k.words; ++i) k.halfQ[i] = (k.Q[i] >> 1) | (i + 1 < k.words ? k.Q[i + 1] << 63 : 0);
  k.hat.resize(level);
  k.inv.resize(level);
  for (u32 i = 0; i < level; ++i) {
    std::vector<u64> h(1, 1);
    u64 hm = 1;
    const u64 qi = c.primes[i];
    for (u32 j = 0; j < level; ++j)
      if (j != i) {
        mp_mul_small(h, c.primes[j]);
        hm = mulmod(hm, c.primes[j] % qi, qi);
      }
    h.resize(k.words, 0);
    k.hat[i] = h;
    u64 e = qi - 2, b = hm, r = 1;  // hm^-1 mod qi
    for (; e; e >>= 1) {
      if (e & 1) r = mulmod(r, b, qi);
      b = mulmod(b, b, qi);
    }
    k.inv[i] = r;
  }
  return k;
}
static double crt_to_double(const Crt& k, const u64* residues /*[level]*/) {
  Context& c = ctx();
  std::vector<u64> acc(k.words + 1, 0);
  for (u32 i = 0; i < k.level; ++i) {
    const u64 y = mulmod(residues[i], k.inv[i], c.primes[i]);
    u64 carry = 0;
    for (size_t w = 0; w < k.words; ++w) {
      unsigned __int128 t = (unsigned __int128)k.hat[i][w] * y + acc[w] + carry;
      acc[w] = (u64)t;
      carry = (u64)(t >> 64);
    }
    acc[k.words] += carry;
  }
  // acc < level * Q: subtract Q while >= Q
  std::vector<u64> Qx = k.Q;
  Qx.resize(k.words + 1, 0);
  while (mp_cmp(acc, Qx) >= 0) {
    u64 borrow = 0;
    for (size_t w = 0; w < acc.size(); ++w) {
      unsigned __int128 t = (unsigned __int128)acc[w] - Qx[w] - borrow;
      acc[w] = (u64)t;
      borrow = (u64)(t >> 64) ? 1 : 0;
    }
  }
  std::vector<u64> hq = k.halfQ;
  hq.resize(k.words + 1, 0);
  bool neg = mp_cmp(acc, hq) > 0;
  if (neg) {  // acc = Q - acc
    u64 borrow = 0;
    for (size_t w = 0; w < acc.size(); ++w) {
      unsigned __int128 t = (unsigned __int128)Qx[w] - acc[w] - borrow;
      acc[w] = (u64)t;
      borrow = (u64)(t >> 64) ? 1 : 0;
    }
  }
  long double r = 0;
  for (size_t w = acc.size(); w-- > 0;) r = r * 18446744073709551616.0L + (long double)acc[w];
  double d = (double)r;
  return neg ? -d : d;
}

// Decode ckks_encoder.c:649-703
void decode(std::vector<cplx>& out, PLAINTEXT* plain) {
  Context& c = ctx();
  POLYNOMIAL* poly = &plain->_poly;
  const u32 level = (u32)poly->_num_primes, N = c.N, half_n = N / 2, slots = plain->_slots, gap = half_n / slots;
  if (poly->_is_ntt) poly_ntt(poly, true);
  std::vector<u64> host((size_t)level * N);
  // (limb-sharded execution: the CRT reconstruction needs every limb -- each comes from its owner, ckks_decryptor.c:19 /
  // ckks_encoder.c:649 are the last place where the limbs of a result meet)
  if (c.shard_world > 1) HIPCHK(acehip_shard_gather(c.hip, q_limbs(poly), level, 0, level, nullptr));
  HIPCHK(acehip_download(c.hip, host.data(), q_limbs(poly), host.size() * 8, nullptr));
  Crt k = make_crt(level);
  std::vector<cplx> msg(slots);
  std::vector<u64> r(level);
  for (u32 i = 0; i < slots; ++i) {
    for (u32 l = 0; l < level; ++l) r[l] = host[(size_t)l * N + (size_t)i * gap];
    const double re = crt_to_double(k, r.data()) / plain->_scaling_factor;
    for (u32 l = 0; l < level; ++l) r[l] = host[(size_t)l * N + (size_t)i * gap + half_n];
    const double im = crt_to_double(k, r.data()) / plain->_scaling_factor;
    msg[i] = cplx(re, im);
  }
  embedding(msg);
  out = msg;
}

// Encrypt_msg ckks_encryptor.c:20-95:  c0 = pk0*v + e1 + m,  c1 = pk1*v + e2
void encrypt(CIPHERTEXT* res, PLAINTEXT* plain) {
  Context& c = ctx();
  POLYNOMIAL* m = &plain->_poly;
  const u32 l = (u32)m->_num_primes;
  res->_scaling_factor = plain->_scaling_factor;
  res->_sf_degree = plain->_sf_degree;
  res->_slots = plain->_slots;
  poly_init_like(&res->_c0_poly, m);
  poly_init_like(&res->_c1_poly, m);
  std::vector<int64_t> tri(c.N);
  POLYNOMIAL v{}, e1{}, e2{};
  poly_alloc(&v, c.N, l, 0);
  poly_alloc(&e1, c.N, l, 0);
  poly_alloc(&e2, c.N, l, 0);
  sample_triangle(tri, c.rng); poly_from_small(&v, tri); poly_ntt(&v, false);
  sample_triangle(tri, c.rng); poly_from_small(&e1, tri); poly_ntt(&e1, false);
  sample_triangle(tri, c.rng); poly_from_small(&e2, tri); poly_ntt(&e2, false);
  u64* c0 = q_limbs(&res->_c0_poly);
  u64* c1 = q_limbs(&res->_c1_poly);
  q_ew(ACEHIP_HW_MUL, c0, c.pk0, q_limbs(&v), l, 0, l);
  q_ew(ACEHIP_HW_ADD, c0, q_limbs(&e1), c0, l, 0, l);
  q_ew(ACEHIP_HW_ADD, c0, c0, q_limbs(m), l, 0, l);
  q_ew(ACEHIP_HW_MUL, c1, c.pk1, q_limbs(&v), l, 0, l);
  q_ew(ACEHIP_HW_ADD, c1, q_limbs(&e2), c1, l, 0, l);
  res->_c0_poly._is_ntt = res->_c1_poly._is_ntt = true;
  poly_free(&v);
  poly_free(&e1);
  poly_free(&e2);
}

// Decrypt ckks_decryptor.c:19-65:  m = c0 + c1*s
void decrypt(PLAINTEXT* res, CIPHERTEXT* ciph) {
  Context& c = ctx();
  HIPCHK(acehip_encode_status(c.hip));  // encode overflow (the reference's assert) surfaces here at the latest
  RT_ASSERT(c.sk_ntt != nullptr, "decrypt: this context holds an evaluation-only key set (no secret key)");
  const u32 l = (u32)ciph->_c0_poly._num_primes;
  init_plaintext(res, ciph->_slots, l, ciph->_c0_poly._num_primes_p, ciph->_scaling_factor, ciph->_sf_degree);
  u64* r = q_limbs(&res->_poly);
  q_ew(ACEHIP_HW_MUL, r, q_limbs(&ciph->_c1_poly), c.sk_ntt, l, 0, l);
  q_ew(ACEHIP_HW_ADD, r, q_limbs(&ciph->_c0_poly), r, l, 0, l);
  res->_poly._is_ntt = true;
}

}  // namespace rt

using namespace rt;

extern "C" {

// plain_eval.c:25-58
void Encode_plain_from_float(PLAIN plain, float* input, size_t len, uint32_t sc_degree, uint32_t level) {
  RtmScope rtm(RTM_PT_ENCODE);
  if (len == 1) {
    encode_value(plain, (double)*input, level, sc_degree);
    return;
  }
  encode_device(plain, stage_to_device(input, len * sizeof(float)), 0, len, level, 0, sc_degree, 0);
  count_weight_plain(plain->_poly._num_alloc_primes * (size_t)plain->_poly._ring_degree * 8);
}
void Encode_plain_from_double(PLAIN plain, double* input, size_t len, uint32_t sc_degree, uint32_t level) {
  RtmScope rtm(RTM_PT_ENCODE);
  if (len == 1) {
    encode_value(plain, *input, level, sc_degree);
    return;
  }
  encode_device(plain, stage_to_device(input, len * sizeof(double)), 1, len, level, 0, sc_degree, 0);
  count_weight_plain(plain->_poly._num_alloc_primes * (size_t)plain->_poly._ring_degree * 8);
}

CIPHER Encrypt(CIPHER res, PLAIN plain) {
  encrypt(res, plain);
  return res;
}

// cipher_eval.c:129-150: decrypt + decode, real parts
double* Get_msg_from_plain(PLAIN plain) {
  PLAINTEXT pt = *plain;  // decode converts to the coefficient domain in place: work on a copy
  pt._poly._data = nullptr;
  poly_alloc(&pt._poly, plain->_poly._ring_degree, plain->_poly._num_primes, plain->_poly._num_primes_p);
  poly_copy(&pt._poly, &plain->_poly);
  std::vector<cplx> out;
  decode(out, &pt);
  double* data = (double*)malloc(sizeof(double) * out.size());
  for (size_t i = 0; i < out.size(); ++i) data[i] = out[i].real();
  poly_free(&pt._poly);
  return data;
}
// (Get_msg, Print_cipher_msg, Dump_cipher_msg, Validate and the rest of cipher_valid.h / cipher_eval.h: rt_valid.cpp)

}  // extern "C"
