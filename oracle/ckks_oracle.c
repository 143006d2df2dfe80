/* ckks_oracle.c -- CPU restatement of the ACE rt_ant RNS-CKKS polynomial layer.
 *
 * TEST INFRASTRUCTURE ONLY (see ckks_oracle.h).  Plain C11 + unsigned __int128, single thread.
 * Reference paths are relative to /root/reference/fhe-cmplr/rtlib/ant.
 * Parity pinned by tests/test_oracle_golden.py against tests/golden/ref_*.json (outputs of the
 * reference rtlib itself, produced by oracle/ref_dump.c) and the reference unit-test KATs.
 */
#include "ckks_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef orc_u128 u128;

/* ------------------------------------------------------------------------------------------
 * modular primitives
 * ---------------------------------------------------------------------------------------- */

/* Mul_int64_with_mod, include/util/fhe_utils.h:176-182 (operands are canonical, so the signed
 * 128-bit product and Mod_int128 reduce to an unsigned product and remainder). */
uint64_t orc_mul_mod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((u128)a * b) % m); }

/* Mod_exp, src/util/number_theory.c:34-50 */
uint64_t orc_pow_mod(uint64_t a, uint64_t e, uint64_t m) {
  uint64_t r = 1, base = a % m;
  while (e > 0) {
    if (e & 1) r = orc_mul_mod(r, base, m);
    base = orc_mul_mod(base, base, m);
    e >>= 1;
  }
  return r;
}

/* Mod_inv_prime, number_theory.c:53-56: a^(m-2) mod m */
uint64_t orc_inv_mod_prime(uint64_t a, uint64_t m) { return orc_pow_mod(a, m - 2, m); }

/* Is_prime, number_theory.c:160-185.  The reference runs 200 Miller-Rabin rounds with rand()
 * bases; for 64-bit inputs the fixed base set below is a proven deterministic test, so the two
 * agree on every input (the reference's answer is only probabilistic for composites). */
int orc_is_prime(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (int i = 0; i < 12; i++) {
    if (n == small[i]) return 1;
    if (n % small[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int      s = 0;
  while ((d & 1) == 0) { d >>= 1; s++; }
  for (int i = 0; i < 12; i++) {
    uint64_t x = orc_pow_mod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int r = 1; r < s; r++) {
      x = orc_mul_mod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}

/* Find_generator, number_theory.c:92-134: smallest r>=2 with r^(phi/f) != 1 for each prime factor */
uint64_t orc_find_generator(uint64_t q) {
  uint64_t phi = q - 1, number = phi;
  uint64_t fac[64];
  int      nf = 0;
  for (uint64_t i = 2; i <= (uint64_t)sqrt((double)number); i++) {
    if (number % i == 0) {
      fac[nf++] = i;
      while (number % i == 0) number /= i;
    }
  }
  if (number > 1) fac[nf++] = number;
  for (uint64_t r = 2; r <= phi; r++) {
    int bad = 0;
    for (int i = 0; i < nf; i++) {
      if (orc_pow_mod(r, phi / fac[i], q) == 1) { bad = 1; break; }
    }
    if (!bad) return r;
  }
  return 0;
}

/* constant {order, prime, root} table: data extracted by oracle/gen_rou_table.py
 * (reference src/util/fhe_std_parms.c:200-272, Get_rou :336-344) */
static const uint64_t Rou_table[][3] = {
#include "rou_table.inc"
};

/* Root_of_unity, number_theory.c:136-157 */
uint64_t orc_root_of_unity(uint64_t order, uint64_t q) {
  if ((q - 1) % order != 0) return 0;
  for (size_t i = 0; i < sizeof(Rou_table) / sizeof(Rou_table[0]); i++) {
    if (Rou_table[i][0] == order && Rou_table[i][1] == q) return Rou_table[i][2];
  }
  uint64_t g = orc_find_generator(q);
  return orc_pow_mod(g, (q - 1) / order, q);
}

/* Reverse_bits, src/util/bit_operations.c:13 */
uint32_t orc_reverse_bits(uint32_t v, uint32_t width) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < width; i++) r |= ((v >> i) & 1u) << (width - 1 - i);
  return r;
}

/* Mod_inv (extended Euclid), number_theory.c:58-80; used for 5^-1 mod 2N */
static int64_t mod_inv_euclid(int64_t val, int64_t m) {
  int64_t mod_value = m, a = val % m, y = 0, x = 1;
  if (m == 1) return 0;
  while (a > 1) {
    int64_t t = mod_value;
    int64_t qd = a / t;
    mod_value = a % t;
    a = t;
    t = y;
    y = x - qd * y;
    x = t;
  }
  if (x < 0) x += m;
  return x;
}

/* Find_automorphism_index, number_theory.c:187-199 (modulus = 2N) */
uint32_t orc_find_automorphism_index(int32_t rot_idx, uint32_t N) {
  uint64_t m = 2ull * N;
  if (rot_idx == 0) return 1;
  if (rot_idx == (int32_t)(m - 1)) return (uint32_t)rot_idx;
  uint64_t gen = 5;
  if (rot_idx < 0) gen = (uint64_t)mod_inv_euclid(5, (int64_t)m);
  return (uint32_t)orc_pow_mod(gen, (uint64_t)(rot_idx < 0 ? -rot_idx : rot_idx), m);
}

/* Precompute_automorphism_order, number_theory.c:201-226 */
void orc_automorphism_order(int64_t* out, uint32_t k, uint32_t N, int is_ntt) {
  uint32_t logn = 0;
  while ((1u << logn) < N) logn++;
  uint64_t m = 2ull * N;
  if (is_ntt) {
    for (uint64_t j = 0; j < N; j++) {
      uint64_t jt = (j << 1) + 1;
      uint64_t idx = ((jt * k) % m) >> 1;
      out[orc_reverse_bits((uint32_t)j, logn)] = orc_reverse_bits((uint32_t)idx, logn);
    }
  } else {
    for (uint64_t j = 0; j < N; j++) {
      uint64_t shift = (j * k) % m;
      if (shift < N) out[shift] = (int64_t)j;
      else out[shift - N] = -(int64_t)j;
    }
  }
}

/* Precompute_const_128, fhe_utils.h:385-401: floor(2^128 / q) (GMP in the reference; exact) */
void orc_precompute_const_128(uint64_t q, uint64_t* lo, uint64_t* hi) {
  u128 two64 = (u128)1 << 64;
  u128 qh = two64 / q, rem = two64 % q;
  u128 ql = (rem << 64) / q;
  u128 r = (qh << 64) + ql;
  *lo = (uint64_t)r;
  *hi = (uint64_t)(r >> 64);
}

/* Mod_barrett_128, fhe_utils.h:241-280 */
uint64_t orc_mod_barrett_128(u128 val, const ORC_PRIME* p) {
  uint64_t val_l = (uint64_t)val, val_h = (uint64_t)(val >> 64);
  uint64_t mu_l = p->prec128_lo, mu_h = p->prec128_hi;
  uint64_t left_h = (uint64_t)(((u128)val_l * mu_l) >> 64);
  u128     mid = (u128)val_l * mu_h;
  uint64_t mid_l = (uint64_t)mid, mid_h = (uint64_t)(mid >> 64);
  uint64_t tmp1 = mid_l + left_h;
  uint64_t carry = tmp1 < left_h;
  uint64_t tmp2 = mid_h + carry;
  mid = (u128)val_h * mu_l;
  mid_l = (uint64_t)mid;
  mid_h = (uint64_t)(mid >> 64);
  carry = (mid_l + tmp1) < tmp1;
  left_h = mid_h + carry;
  tmp1 = val_h * mu_h + tmp2 + left_h;
  uint64_t res = val_l - tmp1 * p->q;
  while (res >= p->q) res -= p->q;
  return res;
}

/* Fast_mul_const_with_mod (Shoup), fhe_utils.h:311-318 */
uint64_t orc_shoup(uint64_t a, uint64_t w, uint64_t w_prec, uint64_t q) {
  uint64_t qq = (uint64_t)(((u128)a * w_prec) >> 64);
  uint64_t y = a * w - qq * q;
  return y >= q ? y - q : y;
}

/* Precompute_const, fhe_utils.h:378-382 */
uint64_t orc_precompute_const(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

/* Switch_modulus, fhe_utils.h:349-375 */
uint64_t orc_switch_modulus(uint64_t v, uint64_t old_q, uint64_t new_q) {
  uint64_t res = v, half = old_q >> 1;
  if (new_q > old_q) {
    if (res > half) res += (new_q - old_q);
  } else {
    uint64_t diff = new_q - (old_q % new_q);
    if (res > half) res += diff;
    if (res >= new_q) res = res % new_q;
  }
  return res;
}

static inline uint64_t add_mod(uint64_t a, uint64_t b, uint64_t q) { /* fhe_utils.h:192 */
  a += b;
  return a >= q ? a - q : a;
}
static inline uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t q) { /* fhe_utils.h:207 */
  return a >= b ? a - b : a + q - b;
}
/* Mul_int64_mod_barret, fhe_utils.h:290-300 (the value it returns; see orc_hw_modmul_faithful) */
static inline uint64_t mul_barrett(uint64_t a, uint64_t b, const ORC_PRIME* p) {
  return orc_mod_barrett_128((u128)a * b, p);
}

/* ------------------------------------------------------------------------------------------
 * prime generation, crt.c:16-125
 * ---------------------------------------------------------------------------------------- */
static uint64_t gen_first_prime(uint32_t N, uint32_t bits) { /* crt.c:16-24 */
  uint64_t order = 2ull * N, c = (1ull << bits) + order + 1;
  while (!orc_is_prime(c)) c += order;
  return c;
}
static uint64_t gen_previous_prime(uint64_t mod, uint64_t order) { /* crt.c:26-32 */
  uint64_t c = mod - order;
  while (!orc_is_prime(c)) c -= order;
  return c;
}
static uint64_t gen_next_prime(uint64_t mod, uint64_t order) { /* crt.c:34-41: starts at mod+2*order */
  uint64_t c = mod + order;
  do { c += order; } while (!orc_is_prime(c));
  return c;
}

/* Generate_q_primes, crt.c:89-125 */
void orc_generate_q_primes(uint64_t* out, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t N) {
  uint64_t order = 2ull * N;
  uint64_t first = gen_first_prime(N, sf_bits);
  out[L - 1] = first;
  uint64_t q_next = first, q_prev = first;
  if (L > 1) {
    uint32_t cnt = 0;
    for (uint32_t i = L - 2; i >= 1; i--) {
      if ((cnt % 2) == 0) { q_prev = gen_previous_prime(q_prev, order); out[i] = q_prev; }
      else                { q_next = gen_next_prime(q_next, order);     out[i] = q_next; }
      cnt++;
    }
  }
  if (q0_bits == sf_bits) out[0] = gen_previous_prime(q_prev, order);
  else                    out[0] = gen_previous_prime(gen_first_prime(N, q0_bits), order);
}

/* Generate_p_primes, crt.c:45-77 (mod_size = AUXBITS = 60, fhe_types.h:27-29) */
void orc_generate_p_primes(uint64_t* out, uint32_t K, uint32_t N, const uint64_t* q, uint32_t L) {
  uint64_t order = 2ull * N;
  uint64_t p_prev = gen_first_prime(N, 60);
  for (uint32_t i = 0; i < K; i++) {
    uint64_t c;
    int      found;
    do {
      c = gen_previous_prime(p_prev, order);
      found = 0;
      for (uint32_t j = 0; j < L; j++) if (q[j] == c) { found = 1; break; }
      p_prev = c;
    } while (found);
    out[i] = c;
  }
}

/* bit length of a product of primes (the reference uses GMP mpz_sizeinbase, crt.c:413) */
static uint32_t product_bits(const uint64_t* v, uint32_t n) {
  uint64_t w[80];
  uint32_t nw = 1;
  memset(w, 0, sizeof(w));
  w[0] = 1;
  for (uint32_t i = 0; i < n; i++) {
    uint64_t carry = 0;
    for (uint32_t k = 0; k < nw; k++) {
      u128 t = (u128)w[k] * v[i] + carry;
      w[k] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    if (carry) w[nw++] = carry;
  }
  uint32_t bits = (nw - 1) * 64;
  uint64_t top = w[nw - 1];
  while (top) { bits++; top >>= 1; }
  return bits;
}

/* Precompute_qpart, crt.c:383-424: K = ceil(max digit bits / AUXBITS) */
uint32_t orc_num_p(const uint64_t* q, uint32_t L, uint32_t dnum) {
  uint32_t alpha = (uint32_t)ceil((double)L / dnum);
  uint32_t max_bits = 0;
  for (uint32_t j = 0; j < dnum; j++) {
    uint32_t lo = j * alpha, hi = (j + 1) * alpha;
    if (hi > L) hi = L;
    uint32_t bits = lo < hi ? product_bits(q + lo, hi - lo) : 1;
    if (bits > max_bits) max_bits = bits;
  }
  return (uint32_t)ceil((double)max_bits / 60);
}

/* ------------------------------------------------------------------------------------------
 * per-prime NTT tables: Init_modulus fhe_utils.h:66-71, Init_nttcontext ntt.c:27-45,
 * Precompute_ntt ntt.c:80-127
 * ---------------------------------------------------------------------------------------- */
int orc_prime_init(ORC_PRIME* p, uint64_t q, uint32_t N) {
  memset(p, 0, sizeof(*p));
  p->q = q;
  p->br_k = 2 * ((uint64_t)log2((double)q) + 1);
  p->br_m = (uint64_t)(((u128)1 << p->br_k) / q);
  orc_precompute_const_128(q, &p->prec128_lo, &p->prec128_hi);
  uint32_t logn = 0;
  while ((1u << logn) < N) logn++;
  p->n_inv = orc_inv_mod_prime(N, q);
  p->n_inv_prec = orc_precompute_const(p->n_inv, q);
  p->psi = orc_root_of_unity(2ull * N, q);
  if (p->psi == 0) return -1;
  p->rou = (uint64_t*)malloc(sizeof(uint64_t) * N * 4);
  p->rou_prec = p->rou + N;
  p->rou_inv = p->rou + 2 * (size_t)N;
  p->rou_inv_prec = p->rou + 3 * (size_t)N;
  uint64_t psi_inv = orc_inv_mod_prime(p->psi, q);
  uint64_t pw = p->psi, pwi = psi_inv;
  p->rou[0] = 1;
  p->rou_inv[0] = 1;
  for (uint32_t i = 1; i < N; i++) {
    uint32_t r = orc_reverse_bits(i, logn);
    p->rou[r] = pw;
    p->rou_inv[r] = pwi;
    pw = orc_mul_mod(pw, p->psi, q);
    pwi = orc_mul_mod(pwi, psi_inv, q);
  }
  for (uint32_t i = 0; i < N; i++) {
    p->rou_prec[i] = orc_precompute_const(p->rou[i], q);
    p->rou_inv_prec[i] = orc_precompute_const(p->rou_inv[i], q);
  }
  return 0;
}

void orc_prime_free(ORC_PRIME* p) {
  free(p->rou);
  p->rou = NULL;
}

/* ------------------------------------------------------------------------------------------
 * NTT: Forward_transform ntt.c:190-264 (CT, natural -> bit-reversed),
 *      Inverse_transform ntt.c:268-353 (GS, bit-reversed -> natural, N^-1 folded into stage 1)
 * ---------------------------------------------------------------------------------------- */
void orc_ntt_fwd(uint64_t* a, const ORC_PRIME* p, uint32_t N) {
  const uint64_t q = p->q;
  uint32_t       n = N >> 1, t = n, logt = 0;
  while ((1u << logt) < t) logt++;
  logt += 1;
  for (uint32_t m = 1; m < n; m <<= 1, t >>= 1, --logt) {
    for (uint32_t i = 0; i < m; ++i) {
      uint64_t w = p->rou[i + m], wp = p->rou_prec[i + m];
      for (uint32_t j1 = (i << logt), j2 = j1 + t; j1 < j2; ++j1) {
        uint64_t of = orc_shoup(a[j1 + t], w, wp, q);
        uint64_t lo = a[j1];
        uint64_t hi = lo + of;
        if (hi >= q) hi -= q;
        if (lo < of) lo += q;
        lo -= of;
        a[j1] = hi;
        a[j1 + t] = lo;
      }
    }
  }
  for (uint32_t i = 0; i < (n << 1); i += 2) {
    uint64_t w = p->rou[(i >> 1) + n], wp = p->rou_prec[(i >> 1) + n];
    uint64_t of = orc_shoup(a[i + 1], w, wp, q);
    uint64_t lo = a[i];
    uint64_t hi = lo + of;
    if (hi >= q) hi -= q;
    if (lo < of) lo += q;
    lo -= of;
    a[i] = hi;
    a[i + 1] = lo;
  }
}

void orc_ntt_inv(uint64_t* a, const ORC_PRIME* p, uint32_t N) {
  const uint64_t q = p->q, ni = p->n_inv, nip = p->n_inv_prec;
  uint32_t       n = N;
  for (uint32_t i = 0; i < n; i += 2) {
    uint64_t w = p->rou_inv[(i + n) >> 1], wp = p->rou_inv_prec[(i + n) >> 1];
    uint64_t hi = a[i + 1], lo = a[i];
    uint64_t of = lo;
    if (of < hi) of += q;
    of -= hi;
    lo += hi;
    if (lo >= q) lo -= q;
    lo = orc_shoup(lo, ni, nip, q);
    of = orc_shoup(of, w, wp, q);
    of = orc_shoup(of, ni, nip, q);
    a[i] = lo;
    a[i + 1] = of;
  }
  uint32_t t = 2, logt = 2;
  for (uint32_t m = n >> 2; m >= 1; m >>= 1, t <<= 1, ++logt) {
    for (uint32_t i = 0; i < m; ++i) {
      uint64_t w = p->rou_inv[i + m], wp = p->rou_inv_prec[i + m];
      for (uint32_t j1 = i << logt, j2 = j1 + t; j1 < j2; ++j1) {
        uint64_t hi = a[j1 + t], lo = a[j1];
        uint64_t of = lo;
        if (of < hi) of += q;
        of -= hi;
        lo += hi;
        if (lo >= q) lo -= q;
        of = orc_shoup(of, w, wp, q);
        a[j1] = lo;
        a[j1 + t] = of;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * limb ops, src/poly/poly_arith.c:14-56
 * ---------------------------------------------------------------------------------------- */
void orc_hw_modadd(uint64_t* r, const uint64_t* a, const uint64_t* b, uint64_t q, uint32_t N) {
  for (uint32_t i = 0; i < N; i++) r[i] = add_mod(a[i], b[i], q);
}
void orc_hw_modmul(uint64_t* r, const uint64_t* a, const uint64_t* b, const ORC_PRIME* p, uint32_t N) {
  for (uint32_t i = 0; i < N; i++) r[i] = mul_barrett(a[i], b[i], p);
}
/* cost-faithful variant for CPU-baseline timing: the reference's Mul_int64_mod_barret
 * (fhe_utils.h:290-300) evaluates BOTH the %-based product and the Barrett product in Release
 * builds ("remove exp will got worse perf results", :292). */
void orc_hw_modmul_faithful(uint64_t* r, const uint64_t* a, const uint64_t* b, const ORC_PRIME* p, uint32_t N) {
  volatile uint64_t sink = 0;
  for (uint32_t i = 0; i < N; i++) {
    uint64_t e = orc_mul_mod(a[i], b[i], p->q);
    uint64_t v = mul_barrett(a[i], b[i], p);
    if (e != v) sink++;
    r[i] = v;
  }
  (void)sink;
}
void orc_hw_rotate(uint64_t* r, const uint64_t* a, const int64_t* perm, uint64_t q, uint32_t N) {
  for (uint32_t i = 0; i < N; i++) {
    int64_t m = perm[i];
    r[i] = m >= 0 ? a[m] : q - a[-m];
  }
}

/* ------------------------------------------------------------------------------------------
 * context: Init_crtcontext_with_prime_size crt.c:574-585 -> Precompute_crt crt.c:535-549
 * ---------------------------------------------------------------------------------------- */
ORC_CTX* orc_ctx_create_from_primes(uint32_t N, const uint64_t* q, uint32_t L, uint32_t dnum) {
  ORC_CTX* c = (ORC_CTX*)calloc(1, sizeof(ORC_CTX));
  c->N = N;
  while ((1u << c->logN) < N) c->logN++;
  c->L = L;
  if (dnum == 0) { /* Get_default_num_q_parts, fhe_std_parms.c:327-334 (mult_depth = L-1) */
    uint32_t depth = L - 1;
    dnum = depth > 3 ? 3 : (depth == 0 ? 1 : 2);
  }
  c->dnum = dnum;
  c->alpha = (uint32_t)ceil((double)L / dnum); /* crt.c:386 */
  c->K = orc_num_p(q, L, dnum);
  uint32_t K = c->K;
  uint64_t* pp = (uint64_t*)malloc(sizeof(uint64_t) * (K ? K : 1));
  orc_generate_p_primes(pp, K, N, q, L);
  c->prime = (ORC_PRIME*)calloc(L + K, sizeof(ORC_PRIME));
  for (uint32_t i = 0; i < L; i++) orc_prime_init(&c->prime[i], q[i], N);
  for (uint32_t i = 0; i < K; i++) orc_prime_init(&c->prime[L + i], pp[i], N);
  free(pp);

  /* P tables: Precompute_primes(p) crt.c:233-262 + Precompute_new_base(p, q) crt.c:330-381 */
  c->phat_inv_modp = (uint64_t*)calloc(K ? K : 1, 8);
  c->phat_inv_modp_prec = (uint64_t*)calloc(K ? K : 1, 8);
  c->phat_modq = (uint64_t*)calloc((size_t)L * (K ? K : 1), 8);
  c->pinv_modq = (uint64_t*)calloc(L, 8);
  for (uint32_t l = 0; l < K; l++) {
    uint64_t pl = c->prime[L + l].q, hat = 1;
    for (uint32_t h = 0; h < K; h++) if (h != l) hat = orc_mul_mod(hat, c->prime[L + h].q % pl, pl);
    c->phat_inv_modp[l] = orc_inv_mod_prime(hat, pl);
    c->phat_inv_modp_prec[l] = orc_precompute_const(c->phat_inv_modp[l], pl);
  }
  for (uint32_t i = 0; i < L; i++) {
    uint64_t qi = c->prime[i].q, val = 1;
    for (uint32_t l = 0; l < K; l++) {
      val = orc_mul_mod(val, c->prime[L + l].q % qi, qi);
      uint64_t hat = 1;
      for (uint32_t h = 0; h < K; h++) if (h != l) hat = orc_mul_mod(hat, c->prime[L + h].q % qi, qi);
      c->phat_modq[(size_t)i * K + l] = hat;
    }
    c->pinv_modq[i] = orc_inv_mod_prime(val, qi);
  }

  /* rescale tables, crt.c:270-326.  For dropped limb "level" = k+1 and i < level:
   *   ql_inv_mod_qi[k][i] = q_level^-1 mod q_i
   *   ql_ql_inv_mod_ql_div_ql_mod_qi[k][i] = floor((Q/q_level) * [(Q/q_level)^-1]_{q_level} / q_level) mod q_i
   * With X = (Q/q_level)*inv:  X = 1 (mod q_level) and X = 0 (mod q_i), hence floor(X/q_level)
   * = (X-1)/q_level = -q_level^-1 (mod q_i) -- the same residue the reference obtains with GMP. */
  c->ql_inv_modqi = (uint64_t*)calloc((size_t)L * L, 8);
  c->ql_inv_modqi_prec = (uint64_t*)calloc((size_t)L * L, 8);
  c->qlql = (uint64_t*)calloc((size_t)L * L, 8);
  c->qlql_prec = (uint64_t*)calloc((size_t)L * L, 8);
  for (uint32_t k = 0; k + 1 < L; k++) {
    uint32_t level = k + 1;
    uint64_t ql = c->prime[level].q;
    for (uint32_t i = 0; i < level; i++) {
      uint64_t qi = c->prime[i].q;
      uint64_t inv = orc_inv_mod_prime(ql % qi, qi);
      uint64_t neg = inv == 0 ? 0 : qi - inv;
      c->ql_inv_modqi[(size_t)k * L + i] = inv;
      c->ql_inv_modqi_prec[(size_t)k * L + i] = orc_precompute_const(inv, qi);
      c->qlql[(size_t)k * L + i] = neg;
      c->qlql_prec[(size_t)k * L + i] = orc_precompute_const(neg, qi);
    }
  }
  return c;
}

ORC_CTX* orc_ctx_create(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum) {
  uint64_t* q = (uint64_t*)malloc(sizeof(uint64_t) * L);
  orc_generate_q_primes(q, L, q0_bits, sf_bits, N);
  ORC_CTX* c = orc_ctx_create_from_primes(N, q, L, dnum);
  c->q0_bits = q0_bits;
  c->sf_bits = sf_bits;
  free(q);
  return c;
}

void orc_ctx_free(ORC_CTX* c) {
  if (!c) return;
  for (uint32_t i = 0; i < c->L + c->K; i++) orc_prime_free(&c->prime[i]);
  free(c->prime);
  free(c->phat_inv_modp);
  free(c->phat_inv_modp_prec);
  free(c->phat_modq);
  free(c->pinv_modq);
  free(c->ql_inv_modqi);
  free(c->ql_inv_modqi_prec);
  free(c->qlql);
  free(c->qlql_prec);
  free(c);
}

/* Get_num_decomp_poly, include/util/polynomial.h:158-168 */
uint32_t orc_num_decomp(const ORC_CTX* c, uint32_t level) {
  uint32_t n = (uint32_t)ceil((double)level / c->alpha);
  return n > c->dnum ? c->dnum : n;
}

/* Precompute_qpart_new_base, crt.c:426-533, evaluated on demand for one (level, digit):
 *  - digit d holds n2 = min(alpha, level - alpha*d) limbs            (polynomial.c:1253-1255)
 *  - hat_inv[i]   = (prod_{k<n2,k!=i} q_{alpha d + k})^-1 mod q_{alpha d + i}      (crt.c:439-461)
 *  - complement basis: q-limbs < level not in digit d (in order), then all p-limbs (crt.c:464-494)
 *  - hat_mod[i][j] = (prod_{k<n2,k!=i} q_{alpha d + k}) mod t_j                    (crt.c:497-531) */
uint32_t orc_modup_tables(const ORC_CTX* c, uint32_t level, uint32_t digit, uint64_t* hat_inv,
                          uint32_t* compl_idx, uint64_t* hat_mod) {
  uint32_t a = c->alpha, start = a * digit;
  uint32_t n2 = level - start < a ? level - start : a;
  uint32_t nc = 0;
  for (uint32_t i = 0; i < level; i++) if (i < start || i >= start + n2) compl_idx[nc++] = i;
  for (uint32_t j = 0; j < c->K; j++) compl_idx[nc++] = c->L + j;
  for (uint32_t i = 0; i < n2; i++) {
    uint64_t qi = c->prime[start + i].q, h = 1;
    for (uint32_t k = 0; k < n2; k++) if (k != i) h = orc_mul_mod(h, c->prime[start + k].q % qi, qi);
    hat_inv[i] = orc_inv_mod_prime(h, qi);
    for (uint32_t j = 0; j < nc; j++) {
      uint64_t t = c->prime[compl_idx[j]].q, hm = 1;
      for (uint32_t k = 0; k < n2; k++) if (k != i) hm = orc_mul_mod(hm, c->prime[start + k].q % t, t);
      hat_mod[(size_t)i * nc + j] = hm;
    }
  }
  return n2;
}

/* ------------------------------------------------------------------------------------------
 * Decompose_modup, src/util/polynomial.c:1241-1335 (NTT-domain input)
 * ---------------------------------------------------------------------------------------- */
void orc_decomp_modup(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit) {
  const uint32_t N = c->N, K = c->K;
  uint64_t       hat_inv[64];
  uint32_t       compl_idx[128];
  uint64_t*      hat_mod = (uint64_t*)malloc(sizeof(uint64_t) * 64 * 128);
  uint32_t       n2 = orc_modup_tables(c, level, digit, hat_inv, compl_idx, hat_mod);
  uint32_t       start = c->alpha * digit, nc = level - n2 + K;
  /* part2: copy digit limbs (polynomial.c:1265-1273) */
  memcpy(out + (size_t)start * N, in + (size_t)start * N, sizeof(uint64_t) * N * n2);
  /* iNTT of the digit (polynomial.c:1276-1283) */
  uint64_t* coef = (uint64_t*)malloc(sizeof(uint64_t) * N * n2);
  memcpy(coef, in + (size_t)start * N, sizeof(uint64_t) * N * n2);
  for (uint32_t i = 0; i < n2; i++) orc_ntt_inv(coef + (size_t)i * N, &c->prime[start + i], N);
  /* y_i = [x_i * hat_inv_i]_{q_i} (Barrett), polynomial.c:1299-1301 */
  for (uint32_t i = 0; i < n2; i++) {
    const ORC_PRIME* p = &c->prime[start + i];
    uint64_t*        v = coef + (size_t)i * N;
    for (uint32_t n = 0; n < N; n++) v[n] = mul_barrett(v[n], hat_inv[i], p);
  }
  /* int128 accumulate + Barrett-128 per complement limb (polynomial.c:1302-1320); output limb
   * position: complement q-limb keeps its own index, p-limb j goes to level + j */
  for (uint32_t j = 0; j < nc; j++) {
    uint32_t         gi = compl_idx[j];
    const ORC_PRIME* p = &c->prime[gi];
    uint32_t         pos = gi < c->L ? gi : level + (gi - c->L);
    uint64_t*        dst = out + (size_t)pos * N;
    for (uint32_t n = 0; n < N; n++) {
      u128 sum = 0;
      for (uint32_t i = 0; i < n2; i++) sum += (u128)coef[(size_t)i * N + n] * hat_mod[(size_t)i * nc + j];
      dst[n] = orc_mod_barrett_128(sum, p);
    }
    orc_ntt_fwd(dst, p, N); /* polynomial.c:1322-1329 */
  }
  free(coef);
  free(hat_mod);
}

/* ------------------------------------------------------------------------------------------
 * Reduce_rns_base (ModDown), polynomial.c:928-967 with Fast_base_conv :755-807
 * ---------------------------------------------------------------------------------------- */
void orc_mod_down(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level) {
  const uint32_t N = c->N, K = c->K, L = c->L;
  uint64_t*      pc = (uint64_t*)malloc(sizeof(uint64_t) * N * K);
  memcpy(pc, in + (size_t)level * N, sizeof(uint64_t) * N * K);
  for (uint32_t j = 0; j < K; j++) {
    const ORC_PRIME* p = &c->prime[L + j];
    uint64_t*        v = pc + (size_t)j * N;
    orc_ntt_inv(v, p, N); /* :943-945 */
    for (uint32_t n = 0; n < N; n++) /* :779-790 Shoup by (P/p_j)^-1 */
      v[n] = orc_shoup(v[n], c->phat_inv_modp[j], c->phat_inv_modp_prec[j], p->q);
  }
  for (uint32_t i = 0; i < level; i++) {
    const ORC_PRIME* p = &c->prime[i];
    uint64_t*        dst = out + (size_t)i * N;
    for (uint32_t n = 0; n < N; n++) { /* :791-803 */
      u128 sum = 0;
      for (uint32_t j = 0; j < K; j++) sum += (u128)pc[(size_t)j * N + n] * c->phat_modq[(size_t)i * K + j];
      dst[n] = orc_mod_barrett_128(sum, p);
    }
    orc_ntt_fwd(dst, p, N); /* :947-949 */
    const uint64_t* x = in + (size_t)i * N;
    for (uint32_t n = 0; n < N; n++) /* :956-965 */
      dst[n] = mul_barrett(sub_mod(x[n], dst[n], p->q), c->pinv_modq[i], p);
  }
  free(pc);
}

/* ------------------------------------------------------------------------------------------
 * Rescale_poly, polynomial.c:1097-1163 (NTT branch)
 * ---------------------------------------------------------------------------------------- */
void orc_rescale(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level) {
  const uint32_t   N = c->N, L = c->L;
  const ORC_PRIME* pl = &c->prime[level - 1];
  uint64_t*        last = (uint64_t*)malloc(sizeof(uint64_t) * N);
  uint64_t*        t = (uint64_t*)malloc(sizeof(uint64_t) * N);
  memcpy(last, in + (size_t)(level - 1) * N, sizeof(uint64_t) * N);
  orc_ntt_inv(last, pl, N);
  const uint64_t* inv = c->ql_inv_modqi + (size_t)(level - 2) * L;
  const uint64_t* invp = c->ql_inv_modqi_prec + (size_t)(level - 2) * L;
  const uint64_t* c1 = c->qlql + (size_t)(level - 2) * L;
  const uint64_t* c1p = c->qlql_prec + (size_t)(level - 2) * L;
  for (uint32_t i = 0; i + 1 < level; i++) {
    const ORC_PRIME* p = &c->prime[i];
    for (uint32_t n = 0; n < N; n++)
      t[n] = orc_shoup(orc_switch_modulus(last[n], pl->q, p->q), c1[i], c1p[i], p->q);
    orc_ntt_fwd(t, p, N);
    const uint64_t* x = in + (size_t)i * N;
    uint64_t*       r = out + (size_t)i * N;
    for (uint32_t n = 0; n < N; n++) r[n] = add_mod(orc_shoup(x[n], inv[i], invp[i], p->q), t[n], p->q);
  }
  free(last);
  free(t);
}

/* ------------------------------------------------------------------------------------------
 * key-switch core of the generated Rotate()/Relinearize()
 * (dataset/resnet20_cifar10_pre.onnx.inc:6972-7146; == Fast_switch_key ckks_evaluator.c:391-416
 *  + Multiply_add polynomial.c:148-183 + Reduce_rns_base)
 * ---------------------------------------------------------------------------------------- */
void orc_key_switch(const ORC_CTX* c, uint64_t* out0, uint64_t* out1, const uint64_t* in,
                    const uint64_t* key, uint32_t level) {
  const uint32_t N = c->N, K = c->K, L = c->L;
  const size_t   ext_limbs = level + K, key_limbs = L + K;
  uint64_t*      ext = (uint64_t*)malloc(sizeof(uint64_t) * N * ext_limbs);
  uint64_t*      acc0 = (uint64_t*)calloc(N * ext_limbs, sizeof(uint64_t));
  uint64_t*      acc1 = (uint64_t*)calloc(N * ext_limbs, sizeof(uint64_t));
  uint32_t       nd = orc_num_decomp(c, level);
  for (uint32_t d = 0; d < nd; d++) {
    orc_decomp_modup(c, ext, in, level, d);
    const uint64_t* k0 = key + ((size_t)d * 2 + 0) * key_limbs * N;
    const uint64_t* k1 = key + ((size_t)d * 2 + 1) * key_limbs * N;
    for (uint32_t l = 0; l < ext_limbs; l++) {
      /* q-limb l uses key limb l; p-limb (l - level) uses key limb L + (l - level)
       * (generated code: key P-limbs start at Poly_level(key) = L, inc:7020-7026) */
      uint32_t         gi = l < level ? l : L + (l - level);
      const ORC_PRIME* p = &c->prime[gi];
      const uint64_t*  e = ext + (size_t)l * N;
      const uint64_t*  a0 = k0 + (size_t)gi * N;
      const uint64_t*  a1 = k1 + (size_t)gi * N;
      uint64_t*        r0 = acc0 + (size_t)l * N;
      uint64_t*        r1 = acc1 + (size_t)l * N;
      for (uint32_t n = 0; n < N; n++) {
        r0[n] = add_mod(r0[n], mul_barrett(a0[n], e[n], p), p->q);
        r1[n] = add_mod(r1[n], mul_barrett(a1[n], e[n], p), p->q);
      }
    }
  }
  orc_mod_down(c, out0, acc0, level);
  orc_mod_down(c, out1, acc1, level);
  free(ext);
  free(acc0);
  free(acc1);
}

/* ------------------------------------------------------------------------------------------
 * checksums / deterministic inputs shared with the golden fixtures
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_sum64(const uint64_t* v, size_t n) {
  uint64_t s = 0;
  for (size_t i = 0; i < n; i++) s += v[i];
  return s;
}
uint64_t orc_xorw(const uint64_t* v, size_t n) {
  uint64_t s = 0;
  for (size_t i = 0; i < n; i++) s ^= v[i] * (2 * (uint64_t)i + 1);
  return s;
}
uint64_t orc_splitmix64(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void orc_fill_uniform(const ORC_CTX* c, uint64_t* out, const uint32_t* prime_idx, uint32_t n_limbs, uint64_t seed) {
  for (uint32_t l = 0; l < n_limbs; l++) {
    uint64_t q = c->prime[prime_idx ? prime_idx[l] : l].q;
    for (uint32_t i = 0; i < c->N; i++) out[(size_t)l * c->N + i] = orc_splitmix64(seed, (uint64_t)l * c->N + i) % q;
  }
}
