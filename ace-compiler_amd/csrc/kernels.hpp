// kernels.hpp -- launcher interface between the C ABI (api.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_arith.hpp"

namespace acehip {

// Tables resident in HBM for one context (passed by value to kernels).
struct DevCtx {
  const DevPrime* primes;  // [L+K]
  const u64* rou;          // [L+K][N]
  const u64* rou_prec;
  const u64* rou_inv;
  const u64* rou_inv_prec;
  u32 N, logN, L, K;
};

// prime (global index) of the limb at position pos of a polynomial extended at `level`
__host__ __device__ inline u32 limb_prime(u32 pos, u32 level, u32 L) { return pos < level ? pos : L + (pos - level); }

enum class EwOp : int { Add = 0, Sub = 1, Mul = 2, MulAdd = 3 };

// NTT over limb positions [pos0, pos0+n) of poly at `level`
// the limb at position pos lives at poly + (pos - pos_off)*N
// n_polys polynomials poly_stride words apart are transformed in the same launch
void launch_ntt(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off = 0,
                u32 n_polys = 1, size_t poly_stride = 0);
// elementwise over limb positions
void launch_ew(const DevCtx& c, EwOp op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n_limbs, hipStream_t s,
               u32 pos_off = 0);
void launch_rotate(const DevCtx& c, u64* r, const u64* a, const u32* perm, u32 pos0, u32 n_limbs, hipStream_t s);
// r[l][n] = shoup(a[l][n], w[l], wp[l]) for limbs l in [0,n) with primes gi[l]  (tables in HBM)
void launch_mul_const(const DevCtx& c, u64* r, const u64* a, const u64* w, const u64* wp, const u32* gi, u32 n_limbs, hipStream_t s);
// base conversion: out[pos[j]][n] = (sum_i in[i][n] * hat[i*hat_ld + j]) mod prime(out_gi[j])
void launch_base_conv(const DevCtx& c, u64* out, const u64* in, const u64* hat, const u32* out_gi, const u32* out_pos,
                      u32 n_in, u32 n_out, u32 hat_ld, hipStream_t s);
// ModDown tail: out[i] = shoup(x[i] - out[i], pinv[i]) for i < level
void launch_moddown_tail(const DevCtx& c, u64* out, const u64* x, const u64* pinv, const u64* pinv_prec, u32 level, hipStream_t s);
// Rescale: t[i][n] = shoup(switch_modulus(last[n], q_last, q_i), c1[i]) for i < level-1
void launch_rescale_spread(const DevCtx& c, u64* t, const u64* last, const u64* c1, const u64* c1p, u32 level, hipStream_t s);
// Rescale tail: out[i] = shoup(x[i], inv[i]) + t[i]
void launch_rescale_tail(const DevCtx& c, u64* out, const u64* x, const u64* t, const u64* inv, const u64* invp, u32 level, hipStream_t s);
// key inner product for one digit: acc{0,1}[pos] (+)= key{0,1}[gi(pos)] * ext[pos], pos < level+K
void launch_key_mac(const DevCtx& c, u64* acc0, u64* acc1, const u64* key0, const u64* key1, const u64* ext, u32 level,
                    bool accumulate, hipStream_t s);

}  // namespace acehip
