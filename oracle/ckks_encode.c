/* ckks_encode.c -- CPU restatement of the reference CKKS encoder.  TEST INFRASTRUCTURE ONLY: never linked
 * into the product; used by tests/, __graft_entry__.smoke() and nothing else.
 *
 * Follows  rtlib/ant/src/util/ntt.c:587-610   Precompute_fft   (cos/sin table over m = 2N, 5^i orbit)
 *          rtlib/ant/src/util/ntt.c:713-753   Embedding_inv    (special inverse FFT)
 *          rtlib/ant/src/util/ckks_encoder.c:199-297  Encode_impl (64-bit path)
 *          rtlib/ant/src/util/ckks_encoder.c:464-530  Encode_val_at_level
 * Pinned by tests/test_oracle_golden.py against tests/golden/ref_encode_*.json, which oracle/ref_dump.c
 * produced by calling the reference itself.  Compiled with -ffp-contract=off: the reference build
 * (plain x86-64 doubles) rounds every product before it is added. */
#define _GNU_SOURCE /* sincos */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ckks_oracle.h"

typedef struct {
  double re, im;
} orc_cplx;

static orc_cplx cx_mul(orc_cplx a, orc_cplx b) {
  orc_cplx r;
  r.re = a.re * b.re - a.im * b.im;
  r.im = a.re * b.im + a.im * b.re;
  return r;
}

/* vals[n] (n = slots, a power of two) is transformed in place */
static void embedding_inv(orc_cplx* vals, size_t n, uint32_t N) {
  const size_t m = 2 * (size_t)N;
  orc_cplx* rou = (orc_cplx*)malloc(sizeof(orc_cplx) * (m + 1));
  uint32_t* orbit = (uint32_t*)malloc(sizeof(uint32_t) * (N / 2 ? N / 2 : 1));
  for (size_t j = 0; j < m; ++j) {
    const double angle = 2 * M_PI * j / m;
    sincos(angle, &rou[j].im, &rou[j].re); /* = gcc's fusion of the reference's cos()/sin() pair (glibc) */
  }
  orbit[0] = 1;
  for (size_t i = 1; i < N / 2; ++i) orbit[i] = (uint32_t)((5ull * orbit[i - 1]) % m);
  uint32_t logn = 0;
  while (((size_t)1 << logn) < n) ++logn;
  for (uint32_t lg = logn; lg >= 1; --lg) {
    const size_t period = (size_t)1 << (lg + 2), step = m / period, span = (size_t)1 << lg, half = span >> 1;
    for (size_t blk = 0; blk < n; blk += span) {
      for (size_t i = 0; i < half; ++i) {
        orc_cplx* lo = &vals[blk + i];
        orc_cplx* hi = &vals[blk + i + half];
        const size_t k = (period - (orbit[i] % period)) * step;
        orc_cplx sum, dif;
        sum.re = lo->re + hi->re;
        sum.im = lo->im + hi->im;
        dif.re = lo->re - hi->re;
        dif.im = lo->im - hi->im;
        *lo = sum;
        *hi = cx_mul(dif, rou[k]);
      }
    }
  }
  orc_cplx* tmp = (orc_cplx*)malloc(sizeof(orc_cplx) * n);
  for (size_t i = 0; i < n; ++i) tmp[i] = vals[orc_reverse_bits((uint32_t)i, logn)];
  for (size_t i = 0; i < n; ++i) {
    vals[i].re = tmp[i].re / (double)n;
    vals[i].im = tmp[i].im / (double)n;
  }
  free(tmp);
  free(orbit);
  free(rou);
}

/* values: len complex numbers (re, im interleaved), zero padded to `slots` (0 = N/2).
 * out_q: level limbs, out_p: n_p limbs (prime index L + j), NTT domain.  Returns 0, or -1 on the
 * reference's "encode overflow" assert. */
int orc_encode(const ORC_CTX* c, uint64_t* out_q, uint64_t* out_p, const double* values, size_t len, uint32_t slots,
               uint32_t sf_degree, uint32_t level, uint32_t n_p) {
  const uint32_t N = c->N;
  if (slots == 0) slots = N / 2;
  const double sf = ldexp(1.0, (int)c->sf_bits);
  orc_cplx* v = (orc_cplx*)calloc(slots, sizeof(orc_cplx));
  for (size_t i = 0; i < len; ++i) {
    v[i].re = values[2 * i];
    v[i].im = values[2 * i + 1];
  }
  embedding_inv(v, slots, N);
  int64_t* coef = (int64_t*)calloc(N, sizeof(int64_t));
  const uint32_t gap = N / (2 * slots);
  int rc = 0;
  for (uint32_t i = 0; i < slots; ++i) {
    const double re = v[i].re * sf + 0.5, im = v[i].im * sf + 0.5;
    if (!(re <= 9.2e18 && re >= -9.2e18 && im <= 9.2e18 && im >= -9.2e18)) {
      rc = -1;
      break;
    }
    coef[(size_t)i * gap] = llround(re);
    coef[(size_t)(i + slots) * gap] = llround(im);
  }
  const uint64_t sfi = (uint64_t)sf;
  for (uint32_t l = 0; rc == 0 && l < level + n_p; ++l) {
    const ORC_PRIME* P = l < level ? &c->prime[l] : &c->prime[c->L + (l - level)];
    uint64_t* dst = l < level ? out_q + (size_t)l * N : out_p + (size_t)(l - level) * N;
    const uint64_t q = P->q;
    uint64_t pw = 1;
    if (l < level && sf_degree > 1) {
      pw = sfi % q;
      for (uint32_t d = 2; d < sf_degree; ++d) pw = orc_mul_mod(pw, sfi % q, q);
    }
    for (uint32_t i = 0; i < N; ++i) {
      const int64_t x = coef[i];
      uint64_t r = x < 0 ? (uint64_t)(-(x % (int64_t)q)) : (uint64_t)(x % (int64_t)q);
      if (x < 0 && r != 0) r = q - r;
      if (pw != 1) r = orc_mul_mod(r, pw, q);
      dst[i] = r;
    }
    orc_ntt_fwd(dst, P, N);
  }
  free(coef);
  free(v);
  return rc;
}

/* Encode_val_at_level: the constant polynomial, one residue per limb (every NTT slot holds it) */
int orc_encode_value(const ORC_CTX* c, uint64_t* out_consts, double value, uint32_t sf_degree, uint32_t level) {
  const double sf = ldexp(1.0, (int)c->sf_bits);
  const int max_word = 61, max_step = 60;
  const int32_t log_sf = (int32_t)ceil(log2(fabs(value * sf)));
  const int32_t log_valid = log_sf <= max_word ? log_sf : max_word;
  const int32_t log_approx = log_sf - log_valid;
  const double scaled = value / pow(2, log_approx) * sf + 0.5;
  if (!(scaled <= 9.2e18 && scaled >= -9.2e18)) return -1;
  const int64_t iv = (int64_t)scaled;
  const int64_t isf = (int64_t)(sf + 0.5);
  for (uint32_t l = 0; l < level; ++l) {
    const uint64_t q = c->prime[l].q;
    int64_t r = iv % (int64_t)q;
    if (r < 0) r += (int64_t)q;
    uint64_t acc = (uint64_t)r;
    for (uint32_t d = 1; d < sf_degree; ++d) acc = orc_mul_mod(acc, (uint64_t)isf % q, q);
    int32_t rest = log_approx;
    while (rest > 0) {
      const int32_t step = rest == log_approx ? (rest <= max_word ? rest : max_word) : (rest <= max_step ? rest : max_step);
      acc = orc_mul_mod(acc, ((uint64_t)1 << step) % q, q);
      rest -= step;
    }
    out_consts[l] = acc;
  }
  return 0;
}
