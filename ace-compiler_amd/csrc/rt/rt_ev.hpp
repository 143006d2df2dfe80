// rt_ev.hpp -- internal ciphertext evaluator used by the CKKS-level API and by Bootstrap
// (reference: src/util/ckks_evaluator.c).  All polynomial work runs on the device.
#pragma once
#include "rt_internal.hpp"

namespace rt {

// owning wrapper of a CIPHERTEXT whose polynomials live in the device pool
struct Ct {
  CIPHERTEXT c;
  Ct() { memset(&c, 0, sizeof(c)); }
  ~Ct() { reset(); }
  Ct(const Ct&) = delete;
  Ct& operator=(const Ct&) = delete;
  void reset() {
    poly_free(&c._c0_poly);
    poly_free(&c._c1_poly);
    memset(&c, 0, sizeof(c));
  }
  u32 level() const { return (u32)c._c0_poly._num_primes; }
  u32 np() const { return (u32)c._c0_poly._num_primes_p; }
  void take(Ct& o) {  // move o into *this
    reset();
    c = o.c;
    memset(&o.c, 0, sizeof(o.c));
  }
};

namespace ev {
// fresh polys, zeroed unless the caller overwrites every limb right away (zero = false)
void init(Ct& r, u32 nq, u32 np, double sf, u32 sf_degree, u32 slots, bool zero = true);
void copy(Ct& r, const Ct& a);
void from_ciph(Ct& r, CIPHER a);        // deep copy of a caller-owned ciphertext
void to_ciph(CIPHER r, Ct& a);          // move a into caller-owned r (frees r's old polys)
void set_level(Ct& a, u32 level);       // Set_ciph_level (drop limbs)
void add(Ct& r, Ct& a, Ct& b);          // Add_ciphertext :45  (r may alias a or b)
void sub(Ct& r, Ct& a, Ct& b);          // Sub_ciphertext :75
void add_const(Ct& r, Ct& a, double v); // Add_const :116
void mul_const(Ct& r, Ct& a, double v); // Mul_const :211
void mul(Ct& r, Ct& a, Ct& b);          // Mul_ciphertext :167 (with relinearisation)
void rescale(Ct& r, Ct& a);             // Rescale_ciphertext :324
void mul_integer(Ct& r, Ct& a, u64 k);  // Mul_integer :222
void mul_monomial(Ct& r, Ct& a, u32 power);  // Mul_by_monomial :237
void rotate(Ct& r, Ct& a, int32_t rotation); // Eval_fast_rotate :529
void conjugate(Ct& r, Ct& a);           // Conjugate :577
// per-limb residues of the constant plaintext value*Delta^sf_degree (Encode_val_at_level ckks_encoder.c:464-530)
std::vector<u64> const_residues(double value, u32 level, u32 sf_degree);
}  // namespace ev

void bootstrap(Ct& res, Ct& in, u32 raise_level);

}  // namespace rt
