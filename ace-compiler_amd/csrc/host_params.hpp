// host_params.hpp -- host-side CKKS parameter / table generation for the HIP runtime.
//
// Produces, bit-for-bit, the constants the reference derives at context creation
// (reference: fhe-cmplr/rtlib/ant/src/util/{crt.c,ntt.c,number_theory.c}, include/util/fhe_utils.h):
// q/p prime chains, psi (2N-th root of unity), bit-reversed twiddle tables with Shoup companions,
// and the CRT tables for ModUp / ModDown / Rescale.  These are uploaded to HBM once per context.
#pragma once
#include <cstdint>
#include <vector>

namespace acehip {

using u64  = uint64_t;
using u32  = uint32_t;
using u128 = unsigned __int128;

// modular helpers (host)
u64 mul_mod(u64 a, u64 b, u64 m);
u64 pow_mod(u64 a, u64 e, u64 m);
u64 inv_mod_prime(u64 a, u64 m);
bool is_prime(u64 n);
u64 find_generator(u64 q);
u64 root_of_unity(u64 order, u64 q);
u32 reverse_bits(u32 v, u32 width);
u64 shoup_prec(u64 w, u64 q);  // floor(w * 2^64 / q)
u32 find_automorphism_index(int32_t rot_idx, u32 N);
void automorphism_order_ntt(u32* perm, u32 k, u32 N);

// Per-prime constants as the kernels consume them (mirrored in HBM, see device_types.hpp)
struct PrimeConsts {
  u64 q;
  u64 barrett_mu;   // floor(2^(2n)/q) << (63-n), n = bitlen(q): qhat = mulhi(x >> (n-1), barrett_mu)
  u32 nbits;        // n
  u32 pad;
  u64 prec128_lo;   // floor(2^128 / q)  (128-bit Barrett for the base-conversion sums)
  u64 prec128_hi;
  u64 psi;
  u64 n_inv, n_inv_prec;                // N^-1 mod q and its Shoup companion
  u64 inv_w1_ninv, inv_w1_ninv_prec;    // rou_inv[1] * N^-1 mod q (last inverse stage, N^-1 folded)
};

struct HostParams {
  u32 N = 0, logN = 0, L = 0, K = 0, dnum = 0, alpha = 0, q0_bits = 0, sf_bits = 0;
  std::vector<PrimeConsts> primes;  // [L+K]
  // twiddles, [L+K][N] each: rou[bitrev(i)] = psi^i
  std::vector<u64> rou, rou_prec, rou_inv, rou_inv_prec;
  // ModDown (P -> Q)
  std::vector<u64> phat_inv_modp, phat_inv_modp_prec;  // [K]
  std::vector<u64> phat_modq;                          // [L][K]
  std::vector<u64> pinv_modq, pinv_modq_prec;          // [L]
  // Rescale: row k (= dropped limb index - 1), column i <= k
  std::vector<u64> ql_inv, ql_inv_prec, qlql, qlql_prec;  // [L][L]

  u64 q(u32 gi) const { return primes[gi].q; }
  u32 num_decomp(u32 level) const;
  // ModUp tables for (level, digit); hat_mod is [n2][nc] row-major; compl_idx are global prime indices
  struct ModUp {
    u32 n2 = 0, nc = 0, start = 0;
    std::vector<u64> hat_inv, hat_inv_prec;  // [n2]
    std::vector<u32> compl_idx;              // [nc]
    std::vector<u64> hat_mod;                // [n2*nc]
  };
  ModUp modup(u32 level, u32 digit) const;
};

void generate_q_primes(std::vector<u64>& out, u32 L, u32 q0_bits, u32 sf_bits, u32 N);
void generate_p_primes(std::vector<u64>& out, u32 K, u32 N, const std::vector<u64>& q);
u32 num_p_primes(const std::vector<u64>& q, u32 dnum);

// dnum == 0 selects the reference default (2 for depth 1..3, 3 above, 1 for depth 0)
HostParams make_params(u32 N, u32 L, u32 q0_bits, u32 sf_bits, u32 dnum);
HostParams make_params_from_primes(u32 N, const std::vector<u64>& q, u32 dnum);

}  // namespace acehip
