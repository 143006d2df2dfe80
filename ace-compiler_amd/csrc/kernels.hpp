// kernels.hpp -- launcher interface between the C ABI (api.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_arith.hpp"

namespace acehip {

// Throughput-sensitivity experiments (tools/ablate.sh): a library built with -DACEHIP_ABLATION skips the launches of the kernel
// families named by the bit mask in $ACEHIP_ABLATE (results are wrong; only the time of what is left is of interest).  Product
// builds compile this to nothing.
enum AblateFamily : unsigned { ABL_NTT = 1, ABL_EW = 2, ABL_ROTATE = 4, ABL_KEYMAC = 8, ABL_CONV = 16, ABL_BSGS = 32, ABL_EMBED = 64,
                               ABL_OTHER = 128 };
#ifdef ACEHIP_ABLATION
unsigned ablate_mask();
#define ACEHIP_ABLATE(family) do { if (ablate_mask() & (family)) return; } while (0)
#else
#define ACEHIP_ABLATE(family) do { } while (0)
#endif

// Tables resident in HBM for one context (passed by value to kernels).
struct DevCtx {
  const DevPrime* primes;  // [L+K]
  const ulong2* tw_fwd;    // [L+K][N] {rou[bitrev], Shoup companion}
  const ulong2* tw_inv;    // [L+K][N] {rou_inv[bitrev], Shoup companion}
  u32 N, logN, L, K;
  u32 split_bits;  // ceil(max prime bits / 2): base conversion multiplies in halves of this width (keyswitch.hip)
  // N = 2^16 transforms of at most this many limb rows (limbs x polynomials) run as narrow passes (ntt_fast.hip ntt4_*)
  u32 ntt_narrow_max_rows = 0;
  // companion-only twiddle tables [L+K][N] (8 bytes per entry; ntt_fast.hip Tp15) and the largest number of polynomials
  // per launch for which the contiguous passes read them instead of the 16-byte tables
  const u64* twp_fwd = nullptr;
  const u64* twp_inv = nullptr;
  u32 tw8_max_polys = 0;
  // the twiddles as doubles [L+K][N] (ntt_fp.hpp: FP64 butterflies for the primes below 2^50.17 in the wide N = 2^16 passes);
  // null: every limb takes the integer classes (ACEHIP_NTT_FP=0, other ring sizes)
  const double* twd_fwd = nullptr;
  const double* twd_inv = nullptr;
  // ---- replicas.  The caller's polynomial memory (the rt_ant shim's pool arena) may exist several times, rep_stride bytes
  // apart: one copy per image of a batch (the GPU form of the reference's image-parallel loop, resnet_cifar.main.inc:77-116:
  // B images run through the same launches and share every key, twiddle, bootstrap diagonal and weight plaintext), or one
  // copy per simulated rank of limb-sharded execution.  A launch covers replicas [rep0, rep0 + nrep): every pointer that lies
  // in [rep_lo, rep_lo + rep_span) is taken rep * rep_stride further by the workgroups of replica rep (reb() below); pointers
  // outside (keys, tables, plaintexts shared by all images) are used as they are.  nrep = 1, rep0 = 0: the plain case.
  u64 rep_lo = 0, rep_span = 0, rep_stride = 0;
  u32 rep0 = 0, nrep = 1;
  // ---- limb ownership (limb-sharded execution): the launch touches only the limbs whose prime index gi has
  // gi % sh_world == sh_rank; polynomials keep their full layout, the other limbs are simply not this rank's business
  u32 sh_rank = 0, sh_world = 1;
};

// prime (global index) of the limb at position pos of a polynomial extended at `level`
__host__ __device__ inline u32 limb_prime(u32 pos, u32 level, u32 L) { return pos < level ? pos : L + (pos - level); }

// Workgroup -> (tile, limb row y, polynomial z) of an NTT pass launch (1-D grid of tiles*n_limbs*n_polys blocks).
// XCD-aware: blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so the n_polys workgroups that
// work on the same (tile, limb) -- and therefore read the same twiddles -- get block ids that are congruent mod 8 and
// consecutive in time: the twiddle tile is fetched into that XCD's L2 once instead of once per polynomial (measured on the
// 32-polynomial roofline batch: contiguous-pass reads 1.57 GB -> 0.57 GB per launch).  An L2 does not keep anything
// across a kernel boundary, so there is nothing to gain from pinning a limb's two passes to one XCD (tried: same
// FETCH_SIZE, worse balance).  Placement only affects speed; any mapping is correct.
#ifdef __HIPCC__
// pointer p as replica `rep` sees it (rep = absolute replica index)
template <class T>
__device__ __forceinline__ T* reb(const DevCtx& c, T* p, u32 rep) {
  const u64 a = (u64)p;
  return (a - c.rep_lo < c.rep_span) ? (T*)(a + (u64)rep * c.rep_stride) : p;
}
__device__ __forceinline__ bool owns(const DevCtx& c, u32 gi) { return c.sh_world <= 1 || gi % c.sh_world == c.sh_rank; }
// launches that put the replica into blockIdx.z (grid.z = nz * nrep): the kernel's own z and the replica
struct RepZ {
  u32 z, rep;
};
__device__ __forceinline__ RepZ rep_of_z(const DevCtx& c) {
  if (c.nrep == 1) return RepZ{blockIdx.z, c.rep0};
  const u32 nz = gridDim.z / c.nrep, r = blockIdx.z / nz;
  return RepZ{blockIdx.z - r * nz, c.rep0 + r};
}
// 1-D launches of X * Y * nrep workgroups (x: coefficient tile, y: limb / chain segment) whose replicas share a large
// read-only operand (a switch key, bootstrap diagonals, a weight plaintext): workgroups are dealt round-robin over the 8 XCDs
// by their linear id, so the nrep workgroups of one (x, y) get ids that are congruent mod 8 -- same XCD, same L2 -- and
// consecutive in time: the shared operand's tile comes from HBM once and from that L2 for the other replicas.  With one
// replica the order is the plain x-fastest one.  Placement only affects speed; any mapping is correct.
struct RepBlk {
  u32 x, y, rep;
};
__device__ __forceinline__ RepBlk rep_block(const DevCtx& c, u32 X, u32 Y) {
  const u32 b = blockIdx.x;
  u32 x, y, r;
  if ((X & 7u) == 0) {
    u32 t = b >> 3;
    r = t % c.nrep;
    t /= c.nrep;
    const u32 Xg = X >> 3, xg = t % Xg;
    y = t / Xg;
    x = xg * 8 + (b & 7u);
  } else {
    x = b % X;
    const u32 t = b / X;
    y = t % Y;
    r = t / Y;
  }
  return RepBlk{(u32)__builtin_amdgcn_readfirstlane(x), (u32)__builtin_amdgcn_readfirstlane(y), c.rep0 + (u32)__builtin_amdgcn_readfirstlane(r)};
}
struct NttBlk {
  u32 tile, y, z;
};
__device__ __forceinline__ NttBlk ntt_block(u32 log_tiles, u32 n_limbs, u32 n_polys) {
  const u32 b = blockIdx.x, G = n_limbs << log_tiles;  // tiles per limb: a power of two
  u32 g, z;
  if ((G & 7u) == 0) {
    const u32 r = b >> 3;
    g = (r / n_polys) * 8 + (b & 7u);
    z = r % n_polys;
  } else {
    g = b % G;
    z = b / G;
  }
  // wave-uniform values: keep them (and the prime / base pointers derived from them) on the scalar unit
  g = __builtin_amdgcn_readfirstlane(g);
  z = __builtin_amdgcn_readfirstlane(z);
  return NttBlk{g & ((1u << log_tiles) - 1), g >> log_tiles, z};
}
// limb position handled by limb row y of polynomial z.  With skip_alpha != 0
// the launch covers the key-switch digits: polynomial z skips its own digit limbs
// [alpha*z, alpha*z + n2) (they are not produced by ModUp), pos0 must be 0.
__device__ __forceinline__ bool ntt_limb_pos(u32& pos, u32 pos0, u32 level, u32 K, u32 skip_alpha, u32 y, u32 z) {
  pos = pos0 + y;
  if (skip_alpha) {
    const u32 start = skip_alpha * z;
    const u32 n2 = min(skip_alpha, level - start);
    if (pos >= start) pos += n2;
    return pos < level + K;
  }
  return true;
}
#endif

// One base-conversion problem (a key-switch digit, or the P->Q conversion of ModDown); lives in HBM.
struct ConvDesc {
  const u64* hat;         // [n_in][hat_ld]  (Q_d/q_i) mod t_j
  const u64* scale;       // [n_in] per-source constant applied on load (Shoup), or nullptr
  const u64* scale_prec;  // [n_in]
  const u32* src_gi;      // [n_in] prime of each source limb
  const u32* out_gi;      // [n_out] prime of each output limb
  const u32* out_pos;     // [n_out] limb position of each output
  const u32* col;         // [n_out] column of `hat` for each output (nullptr: column j for output j)
  u32 src_pos0;           // first source limb position in the input polynomial
  u32 n_in, n_out, hat_ld;
  // matrix-core form (keyswitch.hip base_conv_mfma_kernel; nullptr: not prepared): the constants as int8 B fragments
  // [tile of 16 outputs][k-step][digit b < 9][lane] (16 bytes each) and, per output column, 9 accumulator offsets
  const void* bfrag = nullptr;
  const u32* boff = nullptr;  // [16 * tiles][9]
};
constexpr u32 kConvMfmaDigits = 9;  // 7-bit digits of a constant below 2^63

enum class EwOp : int { Add = 0, Sub = 1, Mul = 2, MulAdd = 3 };

// Key inner product computed where an NTT pass would load a stored accumulator (Fast_switch_key_ext ckks_evaluator.c:418-460; generated
// code resnet20_cifar10_pre.onnx.inc:7011-7036): the value of polynomial z at extended limb position pos (prime gi) is
//   sum_{d < nd} E_d[pos] (*) key[d][z][gi],   E_d[pos] = own + pos*N when own != nullptr and pos < level and pos / alpha == d
//                                                        (the digit's own limbs read from the key-switch input), else ext[d] + pos*N.
// The accumulators of a key-switch then never exist in memory: the inverse first pass of Mod_down's P-limbs and the Mod_down tail on the
// q-limbs form the sums themselves (exact 128-bit sums, one reduction: the canonical residue of the same integer).
struct Kmac {
  u32 nd = 0;         // digits; 0: off
  u32 alpha = 0, level = 0;
  u32 key_T = 0;      // L + K: polynomial z of key part d starts at key[d] + z*key_T*N, prime gi at + gi*N
  const u64* own = nullptr;
  const u64* ext[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const u64* key[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
constexpr u32 kKmacMaxDigits = 8;

// Work fused into the first / last pass of an N = 2^16 NTT (ntt_fast.hip), so that the neighbours of a transform
// in Rescale / ModDown / ModUp / encode need no launch and no pass over memory of their own.
struct NttFuse {
  // polynomial z lives at polyz[z] instead of poly + z*poly_stride when polyz[0] is set (batched encodes into separate
  // blocks of the caller's pool; at most 8 polynomials)
  u64* polyz[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // inverse, first pass: read polynomial z from src_z instead of transforming in place (limb layout as `poly`)
  const u64* src0 = nullptr;
  const u64* src1 = nullptr;
  // forward, first pass: every limb starts from the same signed message, reduced mod its prime and multiplied by
  // scale[pos] when given (Encode_impl ckks_encoder.c:262-285)
  const int64_t* msg = nullptr;
  size_t msg_stride = 0;          // polynomial z reads msg + z*msg_stride
  const u64* msg_scale = nullptr;
  // forward, first pass: the input of polynomial z is the fast base conversion (Decompose_modup polynomial.c:1302-1320,
  // Reduce_rns_base :928-967) of coefficient-domain source limbs, computed on the fly: limb row y of the launch is output j = y
  // of descriptor conv[z*conv_step] (n_in <= conv_max_in <= 12 sources, no `scale`: pre-factors folded into the inverse NTT
  // before); sources are read from conv_src + z*conv_src_stride at limb positions src_pos0.. .  Needs split_bits <= 30.
  const ConvDesc* conv = nullptr;
  u32 conv_step = 0, conv_max_in = 0;
  const u64* conv_src = nullptr;
  size_t conv_src_stride = 0;
  // inverse, last pass: store the centred representative (x > q/2 ? x - q : x, as int64) instead of x, i.e. the
  // `msg` of a following forward transform over other primes (Rescale, ModRaise)
  bool center_out = false;
  // inverse, last pass: per limb position {N^-1*c, its Shoup companion, w1^-1*N^-1*c, companion} replacing the prime's
  // own last-stage constants: the base-conversion pre-factor c = (Q_d/q_i)^-1 (or (P/p_j)^-1) costs nothing this way
  const u64* inv_scale = nullptr;
  // forward, last pass: v = NTT value; 1: out = x*w + v (Rescale tail polynomial.c:1145-1158),
  // 2: out = (x - v)*w (ModDown tail :956-965); x_z, out_z are polynomials of q-limbs, w/wp per limb
  // limb-sharded execution: the launch covers n PACKED limbs (limb y at poly + y*N, z-th polynomial poly_stride further);
  // limb y belongs to prime gi_tab[y] (device table).  level / pos0 / skip_alpha are ignored, per-limb tables of the
  // fused neighbours (msg_scale, inv_scale, w, wp) are indexed by y.
  const u32* gi_tab = nullptr;
  int epi = 0;
  u64* out0 = nullptr;
  u64* out1 = nullptr;
  const u64* x0 = nullptr;
  const u64* x1 = nullptr;
  const u64* w = nullptr;
  const u64* wp = nullptr;
  // inverse, first pass (with km.nd != 0): the input of polynomial z at limb position pos is the key inner product at extended
  // position km_pos0 + pos;  forward, last pass, epi == 3: the ModDown tail of epi == 2 with x_z replaced by the key inner product
  Kmac km;
  u32 km_pos0 = 0;
};
// statistics hook (api_core.cpp): `limbs` limb-transforms were just launched (direct calls and the ones inside the pipelines alike)
void ntt_count(u64 limbs);
// NTT over limb positions [pos0, pos0+n) of poly at `level`
// the limb at position pos lives at poly + (pos - pos_off)*N
// n_polys polynomials poly_stride words apart are transformed in the same launch
// N = 2^16 only: NTT with fused neighbours (f.src*/f.msg for the first pass, f.epi for the last)
void launch_ntt_fused(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                      u32 n_polys, size_t poly_stride, u32 skip_alpha, const NttFuse& f);
void launch_ntt(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off = 0,
                u32 n_polys = 1, size_t poly_stride = 0, u32 skip_alpha = 0);
// register-tiled passes (ntt_fast.hip)
void launch_ntt_fast(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                     u32 n_polys, size_t poly_stride, u32 skip_alpha);
void launch_ntt_contig8(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                        u32 n_polys, size_t poly_stride, u32 skip_alpha);
// elementwise over limb positions
void launch_ew(const DevCtx& c, EwOp op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n_limbs, hipStream_t s,
               u32 pos_off = 0);
// level: the limbs are positions [pos0, pos0 + n_limbs) of a polynomial extended at `level` (ownership filter only)
void launch_rotate(const DevCtx& c, u64* r, const u64* a, const u32* perm, u32 level, u32 pos0, u32 n_limbs, hipStream_t s);
// r_z = acc_z + automorphism_k(a_z) for one (r1 == nullptr) or two polynomials, limbs [pos0, pos0 + n_limbs) at `level`
void launch_rotate_add2(const DevCtx& c, u64* r0, u64* r1, const u64* acc0, const u64* acc1, const u64* a0, const u64* a1, u32 auto_k,
                        u32 level, u32 pos0, u32 n_limbs, hipStream_t s);
// r[l][n] = shoup(a[l][n], w[l], wp[l]) for limbs l in [0,n) with primes gi[l]  (tables in HBM)
void launch_mul_const(const DevCtx& c, u64* r, const u64* a, const u64* w, const u64* wp, const u32* gi, u32 n_limbs, hipStream_t s);
// base conversion: out[pos[j]][n] = (sum_i in[i][n] * hat[i*hat_ld + j]) mod prime(out_gi[j])
void launch_base_conv(const DevCtx& c, u64* out, const u64* in, const u64* hat, const u32* out_gi, const u32* out_pos,
                      u32 n_in, u32 n_out, u32 hat_ld, hipStream_t s);
// batched form: problem z = blockIdx.z uses descs[z*desc_step], reads in + z*in_stride (coefficient domain
// limbs at positions src_pos0..), writes out + z*out_stride
struct PtrTab8 {  // up to 8 per-problem output polynomials as a kernel argument; p[0] == nullptr: out + z*out_stride instead
  u64* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
void launch_base_conv_batch(const DevCtx& c, u64* out, size_t out_stride, const u64* in, size_t in_stride,
                            const ConvDesc* descs, u32 desc_step, u32 n_problems, u32 max_n_out, hipStream_t s,
                            u32 max_n_in = 0,  // max_n_in: largest n_in of the problems when known (selects the <= 16 kernel)
                            const PtrTab8& outz = PtrTab8{},
                            u32 mfma_steps = 0);  // 1 / 2: every problem carries B fragments of that many k-steps (ConvDesc::bfrag)
// fused key inner product over all digits (generated code inc:7011-7036 for every part):
//   acc{0,1}[pos] = sum_d key{0,1}[d][gi(pos)] * (pos in digit d ? in[pos] : ext[d][pos])
struct LimbConsts;
// add0 / w (may be null): acc0[pos] += add0[pos] * w->w[pos] on the q-limbs
void launch_key_mac_fused(const DevCtx& c, u64* acc0, u64* acc1, const u64* key, const u64* ext, size_t ext_stride,
                          const u64* in, u32 level, u32 nd, u32 alpha, hipStream_t s, const u64* add0 = nullptr,
                          const LimbConsts* w = nullptr);
// several key inner products over the same raised digits in one pass over them (keyswitch.hip key_mac_multi_kernel): rotation j < n_keys
// writes acc{0,1}[j] from key set keys[j]; nd <= 4 digits, n_keys <= KEY_MULTI_MAX
constexpr u32 KEY_MULTI_MAX = 16;
void launch_key_mac_multi(const DevCtx& c, u64* const* acc0, u64* const* acc1, const u64* const* keys, u32 n_keys, const u64* ext,
                          size_t ext_stride, u32 level, u32 nd, hipStream_t s, const u64* add0 = nullptr, const LimbConsts* w = nullptr);
// BSGS inner products (keyswitch.hip bsgs_inner_kernel): kernel-argument block, g <= 16, b <= 16, g*b <= 128
constexpr u32 BSGS_MAX_G = 16, BSGS_MAX_B = 16, BSGS_MAX_PT = 128;
struct BsgsArgs {
  const u64* in0[BSGS_MAX_G];   // c0 of the g pre-rotated PQ-extended ciphertexts (level + K limbs each)
  const u64* in1[BSGS_MAX_G];
  u64* out0[BSGS_MAX_B];        // c0 / c1 of the b results
  u64* out1[BSGS_MAX_B];
  const u64* pt[BSGS_MAX_PT];   // [b][g] plaintext polynomials (q-limbs first, p-limbs at limb pt_q_alloc); nullptr = absent
  u32 g, b, pt_q_alloc;
  // in_auto[j] != 0: input j is read through the automorphism X -> X^k of this context, k = in_auto[j] (the kernel computes
  // the index map like hw_batch_rotate_kernel): the hoisted rotations of Rotate_iteration need no pass of their own
  u32 in_auto[BSGS_MAX_G];
};
void launch_bsgs_inner(const DevCtx& c, const BsgsArgs& a, u32 level, hipStream_t s);
// ---- limb-sharded execution (shard.hip): kernels over PACKED limb lists, limb y of prime gi[y] ----
// Rescale spread on the owned limbs: t_z[y] = shoup(switch_modulus(last_z, q_last, q_gi[y]), c1[y])
void launch_packed_rescale_spread(const DevCtx& c, u64* t, size_t t_stride, const u64* last, size_t last_stride, const u32* gi,
                                  u32 gi_last, const u64* c1, const u64* c1p, u32 n_limbs, u32 n_polys, hipStream_t s);
// out_z[y] = shoup(x_z[y], inv[y]) + t_z[y]
void launch_packed_rescale_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                                const u32* gi, const u64* inv, const u64* invp, u32 n_limbs, u32 n_polys, hipStream_t s);
// out_z[y] = shoup(x_z[y] - t_z[y], w[y])      (ModDown tail on the owned q-limbs)
void launch_packed_moddown_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                                const u32* gi, const u64* w, const u64* wp, u32 n_limbs, u32 n_polys, hipStream_t s);
// acc_z[y] = sum_d key[(d*2+z)*key_stride + y*N] * e_d[y],  e_d[y] = src[d*n_limbs + y] (device pointer table: the
// digit's own limbs point into the input, the others into the raised digits)
struct PackedPtrs {  // limb pointers as a kernel argument (2 KiB): no table upload, nothing to keep alive
  const u64* p[256];
};
void launch_packed_key_mac(const DevCtx& c, u64* acc0, u64* acc1, const u64* key, size_t key_stride, const PackedPtrs& src,
                           const u32* gi, u32 nd, u32 n_limbs, hipStream_t s);
// dst[y] = src.p[y] (limb copies: assembling gathered limbs in position order)
void launch_packed_gather(const DevCtx& c, u64* dst, const PackedPtrs& src, u32 n_limbs, hipStream_t s);
// ModDown tail for two polynomials: out_z = shoup(x_z - t_z, pinv), z in {0,1}
void launch_moddown_tail2(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t0,
                          const u64* t1, const u64* pinv, const u64* pinv_prec, u32 level, hipStream_t s, u32 n_polys = 2);
// ModDown tail: out[i] = shoup(x[i] - out[i], pinv[i]) for i < level
void launch_moddown_tail(const DevCtx& c, u64* out, const u64* x, const u64* pinv, const u64* pinv_prec, u32 level, hipStream_t s);
// Rescale: t[i][n] = shoup(switch_modulus(last[n], q_last, q_i), c1[i]) for i < level-1
void launch_rescale_spread(const DevCtx& c, u64* t, size_t t_stride, const u64* last, size_t last_stride, const u64* c1,
                           const u64* c1p, u32 level, u32 n_polys, hipStream_t s);
// Rescale tail: out[i] = shoup(x[i], inv[i]) + t[i]
void launch_rescale_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                         const u64* inv, const u64* invp, u32 level, u32 n_polys, hipStream_t s);
// ---- setup-side kernels (keygen / encode), rt_kernels.hip ----
// out[pos][n] = vals[n] mod prime(pos) for signed 64-bit vals (Transform_values_to_rns polynomial.c:362-392)
void launch_center(const DevCtx& c, int64_t* out, const u64* in, u32 gi, hipStream_t s);
void launch_values_to_rns(const DevCtx& c, u64* out, const int64_t* vals, u32 level, u32 pos0, u32 n_limbs, hipStream_t s);
// uniform residues from a counter-based generator (Sample_uniform_poly polynomial.c:1349-1371)
void launch_sample_uniform(const DevCtx& c, u64* out, u32 level, u32 pos0, u32 n_limbs, u64 seed, hipStream_t s);
struct ChaChaKey {
  u32 w[8];
};
void launch_sample_uniform_keyed(const DevCtx& c, u64* out, u32 level, u32 pos0, u32 n_limbs, const ChaChaKey& key, hipStream_t s);
struct LimbConsts {
  u64 w[64];
};
// r[pos] = a[pos] * w[pos - pos0] mod prime(pos)   (Scalars_integer_multiply_poly polynomial.c:234-268)
// hw_batch.hip: a list of per-limb ops in one launch (kernel argument block, < 4 KB)
constexpr u32 HW_BATCH_MAX = 112;
enum : u32 {
  HW_OP_ADD = 0, HW_OP_MUL = 1, HW_OP_ROTATE = 2, HW_OP_COPY = 3, HW_OP_ZERO = 4,
  HW_OP_SUB = 5, HW_OP_MULADD = 6, HW_OP_MULC = 7, HW_OP_ADDC = 8,
  // ORed into HwBatchOp::kind: the result is read by the next op of the segment only (or by nobody): it stays in registers
  HW_OP_NOSTORE = 0x80000000u, HW_OP_KIND_MASK = 0xffu
};
struct HwBatchOp {
  u64* res;
  const u64* a;
  const u64* b;   // second operand; the u32 automorphism table for HW_OP_ROTATE
  u32 kind, gi;
};
template <int CAP>
struct HwBatchArgsT {
  HwBatchOp op[CAP];
  uint16_t seg_start[CAP + 1];  // chain segments of the elementwise kernel: ops [seg_start[y], seg_start[y+1])
};
using HwBatchArgs = HwBatchArgsT<HW_BATCH_MAX>;
void launch_hw_batch_ew(const DevCtx& c, const HwBatchArgs& args, u32 n_seg, hipStream_t s);
void launch_hw_batch_rotate(const DevCtx& c, const HwBatchArgs& args, u32 n_ops, hipStream_t s);
// embed.hip: rounded, scaled inverse canonical embedding (device FP64, bit-identical to the reference host code)
struct cd;
constexpr u32 EMB_BATCH_MAX = 8;
struct EmbBatch {  // messages of one batched embedding (device pointers), a kernel argument
  const void* vals[EMB_BATCH_MAX];
};
void launch_embed_inv_batch(int64_t* msg, cd* work, const EmbBatch& batch, u32 n_batch, int kind, size_t len, u32 slots, u32 N,
                            const cd* rou, const u32* rot_group, double sf, int* err_flag, hipStream_t s, double round_add = 0.5);
void launch_embed_inv(int64_t* msg, cd* work, const void* vals, int kind, size_t len, u32 slots, u32 N, const cd* rou,
                      const u32* rot_group, double sf, int* err_flag, hipStream_t s, double round_add = 0.5);
void launch_mul_scalars(const DevCtx& c, u64* r, const u64* a, const LimbConsts& w, u32 level, u32 pos0, u32 n_limbs, hipStream_t s);
// r[pos] = a[pos] + w[pos - pos0] mod prime(pos)   (adding a constant plaintext, Add_const ckks_evaluator.c:116)
void launch_add_scalars(const DevCtx& c, u64* r, const u64* a, const LimbConsts& w, u32 level, u32 pos0, u32 n_limbs, hipStream_t s);
// key inner product for one digit: acc{0,1}[pos] (+)= key{0,1}[gi(pos)] * ext[pos], pos < level+K
void launch_key_mac(const DevCtx& c, u64* acc0, u64* acc1, const u64* key0, const u64* key1, const u64* ext, u32 level,
                    bool accumulate, hipStream_t s);

}  // namespace acehip
