// rt_eval.cpp -- ciphertext-level operations (CKKS-level API of the provider and the building blocks
// of Bootstrap).  Reference: src/util/ckks_evaluator.c:45-600, src/ckks/cipher_eval.c:292-364.
#include <cmath>
#include <cstring>

#include "rt_internal.hpp"

using namespace rt;

namespace rt {
void bootstrap_setup_if_needed() {}
}  // namespace rt

extern "C" {

// Add_ciphertext ckks_evaluator.c:45-73
CIPHER Add_ciph(CIPHER res, CIPHER a, CIPHER b) {
  RT_ASSERT(a->_sf_degree == b->_sf_degree, "Add_ciph: scaling factor degree not match");
  Init_ciph_same_scale(res, a, b);
  poly_ew(Op::Add, &res->_c0_poly, &a->_c0_poly, &b->_c0_poly, true);
  poly_ew(Op::Add, &res->_c1_poly, &a->_c1_poly, &b->_c1_poly, true);
  return res;
}
CIPHER Sub_ciph(CIPHER res, CIPHER a, CIPHER b) {
  RT_ASSERT(a->_sf_degree == b->_sf_degree, "Sub_ciph: scaling factor degree not match");
  Init_ciph_same_scale(res, a, b);
  poly_ew(Op::Sub, &res->_c0_poly, &a->_c0_poly, &b->_c0_poly, true);
  poly_ew(Op::Sub, &res->_c1_poly, &a->_c1_poly, &b->_c1_poly, true);
  return res;
}
// Add_plaintext :100-114
CIPHER Add_plain(CIPHER res, CIPHER a, PLAIN p) {
  RT_ASSERT(a->_sf_degree == p->_sf_degree, "Add_plain: scaling factor degree not match");
  if (res != a) {
    Init_ciph_same_scale_plain(res, a, p);
    poly_copy(&res->_c1_poly, &a->_c1_poly);
  }
  poly_ew(Op::Add, &res->_c0_poly, &a->_c0_poly, &p->_poly, false);
  return res;
}
// Mul_plaintext :183-209
CIPHER Mul_plain(CIPHER res, CIPHER a, PLAIN p) {
  CIPHERTEXT tmp;
  memset(&tmp, 0, sizeof(tmp));
  CIPHER out = (res == a) ? &tmp : res;
  Init_ciph_up_scale_plain(out, a, p);
  poly_ew(Op::Mul, &out->_c0_poly, &a->_c0_poly, &p->_poly, false);
  poly_ew(Op::Mul, &out->_c1_poly, &a->_c1_poly, &p->_poly, false);
  if (out == &tmp) {
    Free_ciph_poly(res, 1);
    *res = tmp;
  }
  return res;
}
// Mul_ciphertext3 :130-165
CIPHER3 Mul_ciph3(CIPHER3 res, CIPHER a, CIPHER b) {
  Init_ciph3_up_scale(res, a, b);
  Context& c = ctx();
  const u32 l = (u32)res->_c0_poly._num_primes;
  u64* t = dalloc((size_t)l * c.N, false);
  HIPCHK(acehip_modmul(c.hip, q_limbs(&res->_c0_poly), q_limbs(&a->_c0_poly), q_limbs(&b->_c0_poly), l, 0, l, nullptr));
  HIPCHK(acehip_modmul(c.hip, q_limbs(&res->_c1_poly), q_limbs(&a->_c0_poly), q_limbs(&b->_c1_poly), l, 0, l, nullptr));
  HIPCHK(acehip_modmuladd(c.hip, q_limbs(&res->_c1_poly), q_limbs(&a->_c1_poly), q_limbs(&b->_c0_poly), l, 0, l, nullptr));
  HIPCHK(acehip_modmul(c.hip, q_limbs(&res->_c2_poly), q_limbs(&a->_c1_poly), q_limbs(&b->_c1_poly), l, 0, l, nullptr));
  dfree(t);
  return res;
}
// Relinearize_ciph3 :266-322 == generated Relinearize(): key-switch c2 with the relin key, add to (c0,c1)
CIPHER Relin(CIPHER res, CIPHER3 ct3) {
  Context& c = ctx();
  Init_ciph_same_scale_ciph3(res, ct3);
  const u32 l = (u32)ct3->_c0_poly._num_primes;
  u64* k0 = dalloc((size_t)l * c.N, false);
  u64* k1 = dalloc((size_t)l * c.N, false);
  HIPCHK(acehip_key_switch(c.hip, k0, k1, q_limbs(&ct3->_c2_poly), c.relin.data, l, nullptr));
  HIPCHK(acehip_modadd(c.hip, q_limbs(&res->_c0_poly), k0, q_limbs(&ct3->_c0_poly), l, 0, l, nullptr));
  HIPCHK(acehip_modadd(c.hip, q_limbs(&res->_c1_poly), k1, q_limbs(&ct3->_c1_poly), l, 0, l, nullptr));
  dfree(k0);
  dfree(k1);
  return res;
}
// Mul_ciphertext :167-181
CIPHER Mul_ciph(CIPHER res, CIPHER a, CIPHER b) {
  CIPHERTEXT3 t3;
  memset(&t3, 0, sizeof(t3));
  Mul_ciph3(&t3, a, b);
  CIPHERTEXT tmp;
  memset(&tmp, 0, sizeof(tmp));
  Relin(&tmp, &t3);
  poly_free(&t3._c0_poly);
  poly_free(&t3._c1_poly);
  poly_free(&t3._c2_poly);
  Free_ciph_poly(res, 1);
  *res = tmp;
  return res;
}
// Rescale_ciphertext :324-345
CIPHER Rescale_ciph(CIPHER res, CIPHER a) {
  CIPHERTEXT tmp;
  memset(&tmp, 0, sizeof(tmp));
  Init_ciph_down_scale(&tmp, a);
  Rescale(&tmp._c0_poly, &a->_c0_poly);
  Rescale(&tmp._c1_poly, &a->_c1_poly);
  if (res != a) Free_ciph_poly(res, 1);
  else Free_ciph_poly(a, 1);
  *res = tmp;
  return res;
}
// Modswitch_ciphertext :381-389: drop the last limb
void Modswitch_ciph(CIPHER a) {
  RT_ASSERT(a->_c0_poly._num_primes > 1, "Modswitch: level not enough");
  a->_c0_poly._num_primes -= 1;
  a->_c1_poly._num_primes -= 1;
}
// Fast_rotate :507-527 == generated Rotate(): key-switch c1 (key of k^-1 applied before), then automorphism
CIPHER Rotate_ciph(CIPHER res, CIPHER a, int32_t rotation) {
  Context& c = ctx();
  const u32 k = Auto_idx(rotation);
  SwitchKeyStore* key = ensure_auto_key(k);
  const u32 l = (u32)a->_c0_poly._num_primes;
  CIPHERTEXT tmp;
  memset(&tmp, 0, sizeof(tmp));
  Init_ciph_same_scale(&tmp, a, nullptr);
  u64* k0 = dalloc((size_t)l * c.N, false);
  u64* k1 = dalloc((size_t)l * c.N, false);
  HIPCHK(acehip_key_switch(c.hip, k0, k1, q_limbs(&a->_c1_poly), key->data, l, nullptr));
  HIPCHK(acehip_modadd(c.hip, k0, k0, q_limbs(&a->_c0_poly), l, 0, l, nullptr));
  const uint32_t* perm = acehip_auto_order(c.hip, k);
  HIPCHK(acehip_rotate(c.hip, q_limbs(&tmp._c0_poly), k0, perm, l, 0, l, nullptr));
  HIPCHK(acehip_rotate(c.hip, q_limbs(&tmp._c1_poly), k1, perm, l, 0, l, nullptr));
  dfree(k0);
  dfree(k1);
  if (res != a) Free_ciph_poly(res, 1);
  else Free_ciph_poly(a, 1);
  *res = tmp;
  return res;
}

CIPHER Bootstrap(CIPHER res, CIPHER ciph, uint32_t level_after_bts) {
  (void)res;
  (void)ciph;
  (void)level_after_bts;
  RT_ASSERT(false, "Bootstrap is not implemented yet in the HIP provider (round 1)");
  return res;
}

}  // extern "C"
