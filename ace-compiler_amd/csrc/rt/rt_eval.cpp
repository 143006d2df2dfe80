// rt_eval.cpp -- ciphertext-level operations: the evaluator used by Bootstrap and the CKKS-level
// provider API (Add_ciph, Mul_ciph, Rotate_ciph ...).
// Reference: src/util/ckks_evaluator.c:45-600, src/ckks/cipher_eval.c:292-364.
#include <cmath>
#include <cstring>

#include "rt_ev.hpp"

namespace rt {
namespace ev {

static void set_meta(Ct& r, double sf, u32 deg, u32 slots) {
  r.c._scaling_factor = sf;
  r.c._sf_degree = deg;
  r.c._slots = slots;
}

void init(Ct& r, u32 nq, u32 np, double sf, u32 sf_degree, u32 slots, bool zero) {
  r.reset();
  poly_alloc(&r.c._c0_poly, ctx().N, nq, np, zero);
  poly_alloc(&r.c._c1_poly, ctx().N, nq, np, zero);
  r.c._c0_poly._is_ntt = r.c._c1_poly._is_ntt = true;
  set_meta(r, sf, sf_degree, slots);
}

void copy(Ct& r, const Ct& a) {
  if (&r == &a) return;
  Ct& aa = const_cast<Ct&>(a);
  init(r, a.level(), a.np(), a.c._scaling_factor, a.c._sf_degree, a.c._slots, false);
  poly_copy(&r.c._c0_poly, &aa.c._c0_poly);
  poly_copy(&r.c._c1_poly, &aa.c._c1_poly);
}

void from_ciph(Ct& r, CIPHER a) {
  init(r, (u32)a->_c0_poly._num_primes, (u32)a->_c0_poly._num_primes_p, a->_scaling_factor, a->_sf_degree, a->_slots, false);
  poly_copy(&r.c._c0_poly, &a->_c0_poly);
  poly_copy(&r.c._c1_poly, &a->_c1_poly);
}

void to_ciph(CIPHER r, Ct& a) {
  poly_free(&r->_c0_poly);
  poly_free(&r->_c1_poly);
  *r = a.c;
  memset(&a.c, 0, sizeof(a.c));
}

void set_level(Ct& a, u32 level) {
  RT_ASSERT(level <= a.level(), "set_level: cannot raise a ciphertext");
  if (a.np() && level != a.level()) {
    // keep the "p-limbs follow the q-limbs" layout the kernels assume: compact the p part
    // (limb by limb, ascending: the destination range may overlap the source range when fewer than np limbs are dropped,
    // but a single limb never overlaps its own source, and limb i is read before limb i+1's destination is written)
    Context& c = ctx();
    for (u32 i = 0; i < a.np(); ++i) {
      copy_limbs((u64*)q_limbs(&a.c._c0_poly) + (size_t)(level + i) * c.N, (const u64*)p_limbs(&a.c._c0_poly) + (size_t)i * c.N, c.N, 0, i);
      copy_limbs((u64*)q_limbs(&a.c._c1_poly) + (size_t)(level + i) * c.N, (const u64*)p_limbs(&a.c._c1_poly) + (size_t)i * c.N, c.N, 0, i);
    }
    a.c._c0_poly._num_alloc_primes = a.c._c1_poly._num_alloc_primes = level + a.np();
  }
  a.c._c0_poly._num_primes = a.c._c1_poly._num_primes = level;
}

// res = a (op) b at level min(la, lb); operands keep their own level
static void addsub(Ct& r, Ct& a, Ct& b, Op op) {
  const u32 l = std::min(a.level(), b.level());
  RT_ASSERT(a.np() == b.np(), "unmatched mod");
  Ct* out = &r;
  Ct tmp;
  const bool alias = (&r == &a) || (&r == &b);
  if (alias && r.level() != l) out = &tmp;  // result must shrink: build it aside
  if (!alias || out == &tmp) {
    Ct& small = a.level() <= b.level() ? a : b;
    init(*out, l, a.np(), small.c._scaling_factor, small.c._sf_degree, small.c._slots);
  }
  POLYNOMIAL pa0 = a.c._c0_poly, pa1 = a.c._c1_poly, pb0 = b.c._c0_poly, pb1 = b.c._c1_poly;
  pa0._num_primes = pa1._num_primes = pb0._num_primes = pb1._num_primes = l;  // views at the common level
  poly_ew(op, &out->c._c0_poly, &pa0, &pb0, true);
  poly_ew(op, &out->c._c1_poly, &pa1, &pb1, true);
  if (out == &tmp) r.take(tmp);
}
void add(Ct& r, Ct& a, Ct& b) { addsub(r, a, b, Op::Add); }
void sub(Ct& r, Ct& a, Ct& b) { addsub(r, a, b, Op::Sub); }

static u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((unsigned __int128)a * b) % m); }

std::vector<u64> const_residues(double value, u32 level, u32 sf_degree) {
  Context& c = ctx();
  const int MAX_BITS_IN_WORD = 61, MAX_LOG_STEP = 60;
  std::vector<u64> consts(level, 0);
  if (value == 0.0) return consts;
  const int32_t log_sf = (int32_t)ceil(log2(fabs(value * c.sf)));
  const int32_t log_valid = log_sf <= MAX_BITS_IN_WORD ? log_sf : MAX_BITS_IN_WORD;
  int32_t log_approx = log_sf - log_valid;
  const double approx_factor = pow(2, log_approx);
  const double scaled = value / approx_factor * c.sf + 0.5;
  RT_ASSERT(scaled <= 9.2e18 && scaled >= -9.2e18, "encode overflow, please choose a smaller scaling factor");
  const int64_t val = (int64_t)scaled;
  const u64 sfs = (u64)(c.sf + 0.5);
  for (u32 i = 0; i < level; ++i) {
    const u64 q = c.primes[i];
    int64_t r = val % (int64_t)q;
    if (r < 0) r += (int64_t)q;
    u64 rv = (u64)r;
    for (u32 j = 1; j < sf_degree; ++j) rv = mulmod(rv, sfs % q, q);
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
  return consts;
}

void add_const(Ct& r, Ct& a, double v) {
  if (&r != &a) copy(r, a);
  const u32 l = r.level();
  std::vector<u64> k = const_residues(v, l, r.c._sf_degree);
  q_scalars(ACEHIP_HW_ADDC, q_limbs(&r.c._c0_poly), q_limbs(&r.c._c0_poly), k.data(), l, 0, l);
}

void mul_const(Ct& r, Ct& a, double v) {
  Context& c = ctx();
  if (&r != &a) copy(r, a);
  const u32 l = r.level();
  std::vector<u64> k = const_residues(v, l, 1);
  q_scalars(ACEHIP_HW_MULC, q_limbs(&r.c._c0_poly), q_limbs(&r.c._c0_poly), k.data(), l, 0, l);
  q_scalars(ACEHIP_HW_MULC, q_limbs(&r.c._c1_poly), q_limbs(&r.c._c1_poly), k.data(), l, 0, l);
  r.c._scaling_factor = r.c._scaling_factor * c.sf;
  r.c._sf_degree += 1;
}

void mul(Ct& r, Ct& a, Ct& b) {
  Context& c = ctx();
  RT_ASSERT(a.np() == 0 && b.np() == 0, "Mul_ciphertext: extended operands are not supported");
  const u32 l = std::min(a.level(), b.level());
  Ct& small = a.level() <= b.level() ? a : b;
  Ct out;
  init(out, l, 0, a.c._scaling_factor * b.c._scaling_factor, a.c._sf_degree + b.c._sf_degree, small.c._slots);
  u64* c2 = dalloc((size_t)l * c.N, false);
  u64* k0 = dalloc((size_t)l * c.N, false);
  u64* k1 = dalloc((size_t)l * c.N, false);
  u64 *a0 = q_limbs(&a.c._c0_poly), *a1 = q_limbs(&a.c._c1_poly), *b0 = q_limbs(&b.c._c0_poly), *b1 = q_limbs(&b.c._c1_poly);
  u64 *o0 = q_limbs(&out.c._c0_poly), *o1 = q_limbs(&out.c._c1_poly);
  q_ew(ACEHIP_HW_MUL, o0, a0, b0, l, 0, l);
  q_ew(ACEHIP_HW_MUL, o1, a0, b1, l, 0, l);
  q_ew(ACEHIP_HW_MULADD, o1, a1, b0, l, 0, l);
  q_ew(ACEHIP_HW_MUL, c2, a1, b1, l, 0, l);
  HIPCHK_T(acehip_key_switch(c.hip, k0, k1, c2, c.relin.data, l, nullptr), {k0, (size_t)l * c.N}, {k1, (size_t)l * c.N}, {c2, (size_t)l * c.N});
  q_ew(ACEHIP_HW_ADD, o0, o0, k0, l, 0, l);
  q_ew(ACEHIP_HW_ADD, o1, o1, k1, l, 0, l);
  dfree(c2);
  dfree(k0);
  dfree(k1);
  r.take(out);
}

void rescale(Ct& r, Ct& a) {
  Context& c = ctx();
  const u32 l = a.level();
  RT_ASSERT(l > 1, "rescale: multiply level is not big enought for more operation, try to use larger depth");
  RT_ASSERT(a.np() == 0, "rescale: extended operand");
  Ct out;
  init(out, l - 1, 0, a.c._scaling_factor / c.sf, a.c._sf_degree - 1, a.c._slots, false);
  HIPCHK_T(acehip_rescale2(c.hip, q_limbs(&out.c._c0_poly), q_limbs(&out.c._c1_poly), q_limbs(&a.c._c0_poly), q_limbs(&a.c._c1_poly), l,
                           nullptr),
           {q_limbs(&out.c._c0_poly), (size_t)(l - 1) * c.N}, {q_limbs(&out.c._c1_poly), (size_t)(l - 1) * c.N},
           {q_limbs(&a.c._c0_poly), (size_t)l * c.N}, {q_limbs(&a.c._c1_poly), (size_t)l * c.N});
  r.take(out);
}

void mul_integer(Ct& r, Ct& a, u64 k) {
  Context& c = ctx();
  if (&r != &a) copy(r, a);
  const u32 l = r.level();
  std::vector<u64> s(l + c.K);
  for (u32 i = 0; i < l; ++i) s[i] = k % c.primes[i];
  q_scalars(ACEHIP_HW_MULC, q_limbs(&r.c._c0_poly), q_limbs(&r.c._c0_poly), s.data(), l, 0, l);
  q_scalars(ACEHIP_HW_MULC, q_limbs(&r.c._c1_poly), q_limbs(&r.c._c1_poly), s.data(), l, 0, l);
  if (r.np()) {
    for (u32 j = 0; j < c.K; ++j) s[j] = k % c.primes[c.L + j];
    q_scalars(ACEHIP_HW_MULC, p_limbs(&r.c._c0_poly), p_limbs(&r.c._c0_poly), s.data(), 0, 0, c.K);
    q_scalars(ACEHIP_HW_MULC, p_limbs(&r.c._c1_poly), p_limbs(&r.c._c1_poly), s.data(), 0, 0, c.K);
  }
}

// NTT of +-x^index over all q-limbs, cached per power (Mul_by_monomial :237-264)
static thread_local std::map<u32, u64*> g_monomials;
void mul_monomial(Ct& r, Ct& a, u32 power) {
  Context& c = ctx();
  RT_ASSERT(a.np() == 0, "Mul_by_monomial: extended operand");
  const u32 pr = power % (2 * c.N), index = power % c.N;
  u64*& mono = g_monomials[pr];
  if (mono == nullptr) {
    UniformScope shared_by_all_images;
    POLYNOMIAL m{};
    poly_alloc(&m, c.N, c.L, 0);
    std::vector<int64_t> v(c.N, 0);
    v[index] = pr < c.N ? 1 : -1;
    poly_from_small(&m, v);
    poly_ntt(&m, false);
    mono = (u64*)m._data;  // kept until Finalize_context (pool)
  }
  if (&r != &a) copy(r, a);
  const u32 l = r.level();
  q_ew(ACEHIP_HW_MUL, q_limbs(&r.c._c0_poly), q_limbs(&r.c._c0_poly), mono, l, 0, l);
  q_ew(ACEHIP_HW_MUL, q_limbs(&r.c._c1_poly), q_limbs(&r.c._c1_poly), mono, l, 0, l);
}
void clear_monomial_cache() { g_monomials.clear(); }

static void switch_and_permute(Ct& r, Ct& a, u32 auto_idx) {
  Context& c = ctx();
  RT_ASSERT(a.np() == 0, "rotate: extended operand");
  SwitchKeyStore* key = ensure_auto_key(auto_idx);
  const u32 l = a.level();
  Ct out;
  init(out, l, 0, a.c._scaling_factor, a.c._sf_degree, a.c._slots);
  u64* k0 = dalloc((size_t)l * c.N, false);
  u64* k1 = dalloc((size_t)l * c.N, false);
  HIPCHK_T(acehip_key_switch(c.hip, k0, k1, q_limbs(&a.c._c1_poly), key->data, l, nullptr), {k0, (size_t)l * c.N}, {k1, (size_t)l * c.N},
           {q_limbs(&a.c._c1_poly), (size_t)l * c.N});
  q_ew(ACEHIP_HW_ADD, k0, k0, q_limbs(&a.c._c0_poly), l, 0, l);
  const uint32_t* perm = acehip_auto_order(c.hip, auto_idx);
  RT_ASSERT(perm, "automorphism table: %s", acehip_last_error());
  q_rotate(q_limbs(&out.c._c0_poly), k0, perm, l, 0, l);
  q_rotate(q_limbs(&out.c._c1_poly), k1, perm, l, 0, l);
  dfree(k0);
  dfree(k1);
  r.take(out);
}
void rotate(Ct& r, Ct& a, int32_t rotation) { switch_and_permute(r, a, ensure_rot_key(rotation)); }
void conjugate(Ct& r, Ct& a) { switch_and_permute(r, a, 2 * ctx().N - 1); }

}  // namespace ev
}  // namespace rt

using namespace rt;

extern "C" {

CIPHER Add_ciph(CIPHER res, CIPHER a, CIPHER b) {
  // provider-level programs accumulate into a ciphertext they cleared with Zero_ciph (rt_seal.h Seal_zero: "res = 0"):
  // an empty operand is the additive identity
  if (a->_c0_poly._data == nullptr || b->_c0_poly._data == nullptr) {
    CIPHER src = a->_c0_poly._data == nullptr ? b : a;
    RT_ASSERT(src->_c0_poly._data != nullptr, "Add_ciph: both operands are empty");
    Copy_ciph(res, src);
    return res;
  }
  Ct x, y, r;
  ev::from_ciph(x, a);
  if (a == b) {
    ev::add(r, x, x);
  } else {
    ev::from_ciph(y, b);
    ev::add(r, x, y);
  }
  ev::to_ciph(res, r);
  return res;
}
CIPHER Sub_ciph(CIPHER res, CIPHER a, CIPHER b) {
  Ct x, y, r;
  ev::from_ciph(x, a);
  ev::from_ciph(y, b);
  ev::sub(r, x, y);
  ev::to_ciph(res, r);
  return res;
}
// Add_plaintext ckks_evaluator.c:100-114
CIPHER Add_plain(CIPHER res, CIPHER a, PLAIN p) {
  Ct x;
  ev::from_ciph(x, a);
  poly_ew(Op::Add, &x.c._c0_poly, &x.c._c0_poly, &p->_poly, false);
  ev::to_ciph(res, x);
  return res;
}
// Mul_plaintext :183-209
CIPHER Mul_plain(CIPHER res, CIPHER a, PLAIN p) {
  Ct x;
  ev::from_ciph(x, a);
  poly_ew(Op::Mul, &x.c._c0_poly, &x.c._c0_poly, &p->_poly, false);
  poly_ew(Op::Mul, &x.c._c1_poly, &x.c._c1_poly, &p->_poly, false);
  x.c._scaling_factor = a->_scaling_factor * p->_scaling_factor;
  x.c._sf_degree = a->_sf_degree + p->_sf_degree;
  ev::to_ciph(res, x);
  return res;
}
// Mul_ciphertext3 :130-165
CIPHER3 Mul_ciph3(CIPHER3 res, CIPHER a, CIPHER b) {
  Init_ciph3_up_scale(res, a, b);
  const u32 l = (u32)res->_c0_poly._num_primes;
  q_ew(ACEHIP_HW_MUL, q_limbs(&res->_c0_poly), q_limbs(&a->_c0_poly), q_limbs(&b->_c0_poly), l, 0, l);
  q_ew(ACEHIP_HW_MUL, q_limbs(&res->_c1_poly), q_limbs(&a->_c0_poly), q_limbs(&b->_c1_poly), l, 0, l);
  q_ew(ACEHIP_HW_MULADD, q_limbs(&res->_c1_poly), q_limbs(&a->_c1_poly), q_limbs(&b->_c0_poly), l, 0, l);
  q_ew(ACEHIP_HW_MUL, q_limbs(&res->_c2_poly), q_limbs(&a->_c1_poly), q_limbs(&b->_c1_poly), l, 0, l);
  return res;
}
// Relinearize_ciph3 :266-322 == generated Relinearize()
CIPHER Relin(CIPHER res, CIPHER3 ct3) {
  Context& c = ctx();
  CIPHERTEXT out;
  memset(&out, 0, sizeof(out));
  Init_ciph_same_scale_ciph3(&out, ct3);
  const u32 l = (u32)ct3->_c0_poly._num_primes;
  u64* k0 = dalloc((size_t)l * c.N, false);
  u64* k1 = dalloc((size_t)l * c.N, false);
  HIPCHK_T(acehip_key_switch(c.hip, k0, k1, q_limbs(&ct3->_c2_poly), c.relin.data, l, nullptr), {k0, (size_t)l * c.N}, {k1, (size_t)l * c.N},
           {q_limbs(&ct3->_c2_poly), (size_t)l * c.N});
  q_ew(ACEHIP_HW_ADD, q_limbs(&out._c0_poly), k0, q_limbs(&ct3->_c0_poly), l, 0, l);
  q_ew(ACEHIP_HW_ADD, q_limbs(&out._c1_poly), k1, q_limbs(&ct3->_c1_poly), l, 0, l);
  dfree(k0);
  dfree(k1);
  Free_ciph_poly(res, 1);
  *res = out;
  return res;
}
CIPHER Mul_ciph(CIPHER res, CIPHER a, CIPHER b) {
  Ct x, y, r;
  ev::from_ciph(x, a);
  if (a == b) {
    ev::mul(r, x, x);
  } else {
    ev::from_ciph(y, b);
    ev::mul(r, x, y);
  }
  ev::to_ciph(res, r);
  return res;
}
CIPHER Rescale_ciph(CIPHER res, CIPHER a) {
  Ct x, r;
  ev::from_ciph(x, a);
  ev::rescale(r, x);
  ev::to_ciph(res, r);
  return res;
}
// Modswitch_ciphertext :381-389
void Modswitch_ciph(CIPHER a) {
  RT_ASSERT(a->_c0_poly._num_primes > 1, "Modswitch: level not enough");
  a->_c0_poly._num_primes -= 1;
  a->_c1_poly._num_primes -= 1;
}
CIPHER Rotate_ciph(CIPHER res, CIPHER a, int32_t rotation) {
  Ct x, r;
  ev::from_ciph(x, a);
  ev::rotate(r, x, rotation);
  ev::to_ciph(res, r);
  return res;
}

}  // extern "C"
