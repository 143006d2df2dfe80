// rt_bootstrap.cpp -- CKKS bootstrapping on the HIP polynomial layer.
// Host orchestration restated from the reference (src/util/ckks_bootstrap_context.c:220-1860,
// src/util/ckks_chebyshev.c, src/ckks/cipher_eval.c:366-404, src/rtlib/context.c:162-185):
// ModRaise from limb 0 -> CoeffsToSlots (collapsed FFT levels, BSGS with hoisted ModUp in the PQ
// basis) -> conjugate split -> Chebyshev approximate mod (Paterson-Stockmeyer) + double-angle
// iterations -> SlotsToCoeffs.  Every polynomial operation is a HIP kernel launch through acehip.h;
// the FP64 matrix precomputation and the encoding of its diagonals happen once at setup.
#include <cmath>
#include <cstring>
#include <set>

#include "rt_ev.hpp"

#include "common/rt_config.h"  // Get_rtlib_config (rt_valid.cpp)

namespace rt {

namespace {

#include "bts_coeffs.inc"

enum { LEVEL_BUDGET, LAYERS_COLL, LAYERS_REM, NUM_ROTATIONS, BABY_STEP, GIANT_STEP, NUM_ROTATIONS_REM, BABY_STEP_REM, GIANT_STEP_REM, TOTAL_PARAMS };

struct SinPoly {
  u32 upper_bound, double_angle, coeff_size;
  const double* coeff;
  bool even;
};

// Get_eval_sin_poly_info ckks_bootstrap_context.c:44-64 (UNIFORM_TERNARY build)
SinPoly sin_poly(size_t hw) {
  const char* e = getenv("RTLIB_BTS_EVEN_POLY");
  const bool even = e != nullptr && atoi(e) != 0;
  const bool under = hw > 0 && hw <= 192;
  if (even) return under ? SinPoly{32, 3, 55, kCoeffUniformEvenHw192, true} : SinPoly{512, 7, 55, kCoeffUniformEven, true};
  return under ? SinPoly{32, 3, 55, kCoeffUniformHw192, false} : SinPoly{512, 6, 89, kCoeffUniform, false};
}

// Gen_depth_by_degree_table :66-82
u32 depth_by_degree(size_t degree) {
  RT_ASSERT(degree >= 5 && degree <= 2031, "Polynomial degree is supported from 5 to 2031 inclusive.");
  if (degree == 5) return 4;
  if (degree <= 13) return 5;
  if (degree <= 27) return 6;
  if (degree <= 59) return 7;
  if (degree <= 119) return 8;
  if (degree <= 247) return 9;
  if (degree <= 495) return 10;
  if (degree <= 1007) return 11;
  return 12;
}
u32 approx_mod_depth(size_t hw) {  // Get_mul_depth_internal
  SinPoly p = sin_poly(hw);
  return depth_by_degree(p.coeff_size - 1) - 1 + p.double_angle;
}

struct Precom {
  u32 slots = 0;
  int enc[TOTAL_PARAMS], dec[TOTAL_PARAMS];
  std::vector<std::vector<PLAINTEXT*>> u0hatt_fft, u0_fft;
  bool keys = false;
};
std::map<u32, Precom*> g_precom;  // shared by all threads (guarded by shared_mu), built by whichever thread needs it first

// Reduce_rotation :220-233
u32 reduce_rotation(int32_t index, u32 slots) {
  const int32_t is = (int32_t)slots;
  if ((slots & (slots - 1)) == 0) {
    const int32_t n = (int32_t)log2((double)slots);
    if (index >= 0) return (u32)(index - ((index >> n) << n));
    return (u32)(index + is + (((-index) >> n) << n));
  }
  return (u32)((is + index % is) % is);
}

// Select_layers :514-548
void select_layers(u32 log_slots, u32 budget, u32& layers, u32& rows, u32& rem) {
  layers = (u32)ceil((double)log_slots / budget);
  rows = log_slots / layers;
  rem = log_slots % layers;
  u32 dim = rows;
  if (rem != 0) dim = rows + 1;
  if (dim < budget) {
    layers -= 1;
    rows = log_slots / layers;
    rem = log_slots - rows * layers;
    dim = rows;
    if (rem != 0) dim = rows + 1;
    while (dim != budget) {
      rows -= 1;
      rem = log_slots - rows * layers;
      dim = rows;
      if (rem != 0) dim = rows + 1;
    }
  }
}

// Get_colls_fft_params :550-603
void colls_fft_params(int* out, u32 slots, u32 level_budget, u32 dim1) {
  const u32 log_slots = (u32)log2((double)slots);
  u32 layers, rows, rem;
  select_layers(log_slots, level_budget, layers, rows, rem);
  const int layers_coll = (int)layers, rem_coll = (int)rem;
  const bool flag_rem = rem_coll != 0;
  const u32 num_rot = (1u << (layers_coll + 1)) - 1, num_rot_rem = (1u << (rem_coll + 1)) - 1;
  int b, g;
  if (dim1 == 0 || dim1 > num_rot) g = num_rot > 7 ? (1 << (layers_coll / 2 + 2)) : (1 << (layers_coll / 2 + 1));
  else g = (int)dim1;
  b = (int)(num_rot + 1) / g;
  int b_rem = 0, g_rem = 0;
  if (flag_rem) {
    g_rem = num_rot_rem > 7 ? (1 << (rem_coll / 2 + 2)) : (1 << (rem_coll / 2 + 1));
    b_rem = (int)(num_rot_rem + 1) / g_rem;
  }
  const int vals[TOTAL_PARAMS] = {(int)level_budget, layers_coll, rem_coll, (int)num_rot, b, g, (int)num_rot_rem, b_rem, g_rem};
  memcpy(out, vals, sizeof(vals));
}

using Mat = std::vector<std::vector<cplx>>;

// Coeff_enc_one_level :409-457 / Coeff_dec_one_level :459-512
Mat coeff_one_level(const std::vector<cplx>& ksipows, const std::vector<u32>& rot_group, bool flag, bool encoding) {
  const u32 dim = (u32)ksipows.size() - 1, slots = (u32)rot_group.size(), log_slots = (u32)log2((double)slots);
  Mat coeff(3 * log_slots, std::vector<cplx>(slots, cplx(0, 0)));
  for (u32 m = slots; m > 1; m >>= 1) {
    const u32 s = (u32)log2((double)m) - 1;
    auto& c_s = coeff[s];
    auto& c_l = coeff[s + log_slots];
    auto& c_2l = coeff[s + 2 * log_slots];
    for (u32 k = 0; k < slots; k += m) {
      const u32 lenh = m >> 1, lenq = m << 2;
      for (u32 j = 0; j < lenh; j++) {
        if (encoding) {
          const u32 jt = (lenq - rot_group[j] % lenq) * (dim / lenq);
          if (flag && m == 2) {
            const cplx val = std::exp(cplx(0, -M_PI / 2));
            const cplx w = val * ksipows[jt];
            c_l[j + k] = val;
            c_2l[j + k] = val;
            c_l[j + k + lenh] = -w;
            c_s[j + k + lenh] = w;
          } else {
            const cplx w = ksipows[jt];
            c_l[j + k] = 1;
            c_2l[j + k] = 1;
            c_l[j + k + lenh] = -w;
            c_s[j + k + lenh] = w;
          }
        } else {
          const u32 jt = (rot_group[j] % lenq) * (dim / lenq);
          if (flag && m == 2) {
            const cplx val = std::exp(cplx(0, M_PI / 2));
            const cplx w = val * ksipows[jt];
            c_l[j + k] = val;
            c_2l[j + k] = w;
            c_l[j + k + lenh] = -w;
            c_s[j + k + lenh] = val;
          } else {
            const cplx w = ksipows[jt];
            c_l[j + k] = 1;
            c_2l[j + k] = w;
            c_l[j + k + lenh] = -w;
            c_s[j + k + lenh] = 1;
          }
        }
      }
    }
  }
  return coeff;
}

// Coeff_collapse :605-774
std::vector<Mat> coeff_collapse(const std::vector<cplx>& ksipows, const std::vector<u32>& rot_group, u32 level_budget,
                                bool flag, bool encoding) {
  const u32 slots = (u32)rot_group.size(), log_slots = (u32)log2((double)slots);
  u32 layers, rows, rem;
  select_layers(log_slots, level_budget, layers, rows, rem);
  const int layers_coll = (int)layers, rem_coll = (int)rem, dim_coll = (int)level_budget;
  const bool flag_rem = rem_coll != 0;
  const u32 num_rot = (1u << (layers_coll + 1)) - 1, num_rot_rem = (1u << (rem_coll + 1)) - 1;
  Mat coeff1 = coeff_one_level(ksipows, rot_group, flag, encoding);
  std::vector<Mat> coeff(dim_coll);
  for (int idx = 0; idx < dim_coll; ++idx) {
    u32 n = num_rot;
    if (flag_rem) {
      const bool after_rem = (encoding && idx >= 1) || (!encoding && idx < (int)level_budget - 1);
      n = after_rem ? num_rot : num_rot_rem;
    }
    coeff[idx].assign(n, std::vector<cplx>(slots, cplx(0, 0)));
  }
  for (int s = 0; s < dim_coll; s++) {
    const int top = encoding ? (int)log_slots - (dim_coll - 1 - s) * layers_coll - 1 : s * layers_coll;
    const bool is_rem = flag_rem && ((encoding && s == 0) || (!encoding && s == dim_coll - 1));
    const int end_l = is_rem ? rem_coll : layers_coll;
    for (int l = 0; l < end_l; l++) {
      if (l == 0) {
        coeff[s][0] = coeff1[top];
        coeff[s][1] = coeff1[top + log_slots];
        coeff[s][2] = coeff1[top + 2 * log_slots];
      } else {
        Mat temp(coeff[s].size(), std::vector<cplx>(slots, cplx(0, 0)));
        u32 t = 0;
        if (encoding) {
          for (int u = 0; u < (1 << (l + 1)) - 1; u++) {
            const auto& temp_u = coeff[s][u];
            for (u32 k = 0; k < slots; k++) {
              const u32 r1 = reduce_rotation((int32_t)k - (1 << (top - l)), slots);
              const u32 r2 = reduce_rotation((int32_t)k + (1 << (top - l)), slots);
              temp[u + t][k] += coeff1[top - l][k] * temp_u[r1];
              temp[u + t + 1][k] += coeff1[top - l + log_slots][k] * temp_u[k];
              temp[u + t + 2][k] += coeff1[top - l + 2 * log_slots][k] * temp_u[r2];
            }
            t += 1;
          }
        } else {
          for (; t < 3; t++) {
            for (int u = 0; u < (1 << (l + 1)) - 1; u++) {
              const auto& temp_u = coeff[s][u];
              for (u32 k = 0; k < slots; k++) {
                if (t == 0) temp[u][k] += coeff1[top + l][k] * temp_u[k];
                if (t == 1) temp[u + (1 << l)][k] += coeff1[top + l + log_slots][k] * temp_u[k];
                if (t == 2) temp[u + (1 << (l + 1))][k] += coeff1[top + l + 2 * log_slots][k] * temp_u[k];
              }
            }
          }
        }
        coeff[s] = temp;
      }
    }
  }
  return coeff;
}

// Rotate_precomp :300-407: rotate + (scale) + encode the collapsed coefficients as extended plaintexts
std::vector<std::vector<PLAINTEXT*>> rotate_precomp(Precom* pre, std::vector<Mat>& coeffs, double scale, u32 level,
                                                    bool encoding) {
  Context& c = ctx();
  const size_t m = 2ull * c.N;
  const int* prm = pre->enc;  // the reference reads the encode params in both directions
  const int level_budget = prm[LEVEL_BUDGET], layers_collapse = prm[LAYERS_COLL], rem_collapse = prm[LAYERS_REM];
  const int num_rot = prm[NUM_ROTATIONS], b = prm[BABY_STEP], g = prm[GIANT_STEP];
  const int num_rot_rem = prm[NUM_ROTATIONS_REM], b_rem = prm[BABY_STEP_REM], g_rem = prm[GIANT_STEP_REM];
  int stop = -1, flag_rem = 0;
  if (rem_collapse != 0) {
    stop = 0;
    flag_rem = 1;
  }
  const u32 rem_index = encoding ? 0 : (u32)level_budget - 1;
  std::vector<std::vector<PLAINTEXT*>> out(level_budget);
  for (u32 i = 0; i < (u32)level_budget; i++) out[i].assign((flag_rem == 1 && i == rem_index) ? num_rot_rem : num_rot, nullptr);
  const int start = encoding ? stop + 1 : 0;
  const int end = encoding ? level_budget : level_budget - flag_rem;
  const int cond = encoding ? start : end - 1;
  const u32 enc_level = level ? level + 1 : c.L - level_budget + 1;
  const u32 dec_level = level ? level + level_budget : c.L;
  auto encode_rot = [&](std::vector<cplx>& vl, u32 rot, u32 plain_level) {
    const size_t n = vl.size();
    std::vector<cplx> rv(n);
    for (size_t i = 0; i < n; ++i) rv[i] = vl[(i + rot) % n];  // Rotate_vector matrix_operations.c:106
    PLAINTEXT* pt = (PLAINTEXT*)calloc(1, sizeof(PLAINTEXT));
    encode_vector(pt, rv.data(), n, plain_level, (u32)n, 1, c.K);  // Encode_ext_at_level
    return pt;
  };
  for (int s = start; s < end; s++) {
    const u32 plain_level = encoding ? enc_level + s : dec_level - s;
    for (int i = 0; i < b; i++) {
      for (int j = 0; j < g; j++) {
        const int dim2 = g * i + j;
        if (dim2 == num_rot) continue;
        const int shift = encoding ? ((s - flag_rem) * layers_collapse + rem_collapse) : (s * layers_collapse);
        const u32 rot = reduce_rotation(-g * i * (1 << shift), (u32)(m / 4));
        auto& vl = coeffs[s][dim2];
        if (flag_rem == 0 && s == cond)
          for (auto& x : vl) x *= scale;
        out[s][dim2] = encode_rot(vl, rot, plain_level);
      }
    }
  }
  if (flag_rem) {
    const int dim1 = encoding ? stop : level_budget - flag_rem;
    const int shift_value = encoding ? 1 : (1 << (dim1 * layers_collapse));
    const u32 plain_level = encoding ? enc_level : dec_level - level_budget + flag_rem;
    for (int i = 0; i < b_rem; i++) {
      for (int j = 0; j < g_rem; j++) {
        const int dim2 = g_rem * i + j;
        if (dim2 == num_rot_rem) continue;
        const u32 rot = reduce_rotation(-g_rem * i * shift_value, (u32)(m / 4));
        auto& vl = coeffs[dim1][dim2];
        for (auto& x : vl) x *= scale;
        out[dim1][dim2] = encode_rot(vl, rot, plain_level);
      }
    }
  }
  return out;
}

// Coeffs2slots_precomp :776-858 / Slots2coeffs_precomp :860-916
std::vector<std::vector<PLAINTEXT*>> fft_precomp(Precom* pre, const std::vector<cplx>& ksipows,
                                                 const std::vector<u32>& rot_group, double scale, u32 level, bool encoding) {
  Context& c = ctx();
  const size_t m = 2ull * c.N, slots = rot_group.size();
  const u32 level_budget = (u32)pre->enc[LEVEL_BUDGET];
  std::vector<Mat> coeffs;
  if (slots == m / 4) {
    coeffs = coeff_collapse(ksipows, rot_group, level_budget, false, encoding);
  } else {  // sparsely packed: concatenate the flag=false and flag=true coefficient sets
    std::vector<Mat> c1 = coeff_collapse(ksipows, rot_group, level_budget, false, encoding);
    std::vector<Mat> c2 = coeff_collapse(ksipows, rot_group, level_budget, true, encoding);
    coeffs.resize(c1.size());
    for (size_t i = 0; i < c1.size(); ++i) {
      coeffs[i].resize(c1[i].size());
      for (size_t j = 0; j < c1[i].size(); ++j) {
        coeffs[i][j] = c1[i][j];
        coeffs[i][j].insert(coeffs[i][j].end(), c2[i][j].begin(), c2[i][j].end());
      }
    }
  }
  if (encoding) {  // fold 1/N * 1/K * 1/(q0/sf) into the matrices (:826-850)
    double factor = 1.0 / c.N;
    factor /= sin_poly(c.hamming).upper_bound;
    const double q0_sf_ratio = round(log2((double)c.primes[0] / c.sf));
    factor /= pow(2, q0_sf_ratio);
    factor = pow(factor, 1. / level_budget);
    for (auto& lvl : coeffs)
      for (auto& row : lvl)
        for (auto& x : row) x *= factor;
  }
  return rotate_precomp(pre, coeffs, scale, level, encoding);
}

// Bootstrap_setup :1050-1192 (FFT variant only: level budget {3,3})
Precom* bootstrap_setup(u32 num_slots) {
  Context& c = ctx();
  const size_t m = 2ull * c.N;
  const u32 slots = num_slots == 0 ? (u32)(m / 4) : num_slots;
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  auto it = g_precom.find(slots);
  if (it != g_precom.end()) return it->second;
  SharedAllocScope shared_plaintexts;  // the encoded diagonals outlive this thread's pool
  UniformScope shared_by_all_images;   // ... and serve every image of a batch: encoded once
  Precom* pre = new Precom();
  pre->slots = slots;
  u32 budget[2] = {3, 3};
  const double log_slots = log2((double)slots);
  for (auto& bgt : budget) {
    if (bgt > log_slots) bgt = (u32)log_slots;
    if (bgt < 1) bgt = 1;
  }
  colls_fft_params(pre->enc, slots, budget[0], 0);
  colls_fft_params(pre->dec, slots, budget[1], 0);
  const u32 slots4 = 4 * slots;
  std::vector<u32> rot_group(slots);
  u32 five = 1;
  for (auto& r : rot_group) {
    r = five;
    five = (u32)(((u64)five * 5) % slots4);
  }
  std::vector<cplx> ksi(slots4 + 1);
  for (size_t i = 0; i < slots4; ++i) {
    const double angle = 2.0 * M_PI * i / slots4;
    ksi[i] = cplx(cos(angle), sin(angle));
  }
  ksi[slots4] = ksi[0];
  const double dq0 = (double)c.primes[0];
  const double pre_f = dq0 / pow(2.0, round(log2(dq0)));
  const double scale_enc = pre_f, scale_dec = 1 / pre_f;
  const u32 amd = approx_mod_depth(c.hamming);
  const u32 enc_budget = (u32)pre->enc[LEVEL_BUDGET], dec_budget = (u32)pre->dec[LEVEL_BUDGET];
  const u32 bts_depth = amd + enc_budget + dec_budget;
  const u32 level_0 = c.L;  // mult depth + 1
  RT_ASSERT(level_0 > enc_budget, "not enough levels");
  RT_ASSERT(level_0 > bts_depth, "need set a larger multiply depth");
  RT_ASSERT(!(enc_budget == 1 && dec_budget == 1), "linear-transform bootstrapping (level budget 1/1) is not implemented");
  pre->u0hatt_fft = fft_precomp(pre, ksi, rot_group, scale_enc, level_0 - enc_budget, true);
  pre->u0_fft = fft_precomp(pre, ksi, rot_group, scale_dec, level_0 - bts_depth, false);
  sync();  // encoded on this thread's stream: complete before other threads can find it
  g_precom[slots] = pre;
  return pre;
}

// Find_rot_indices :300 / Find_coeffslots_rot_index :239-298
void find_rot_index(std::set<int32_t>& out, Precom* pre, u32 slots, u32 m, bool encoding) {
  const int* prm = encoding ? pre->enc : pre->dec;
  const int level_budget = prm[LEVEL_BUDGET], layers_collapse = prm[LAYERS_COLL], rem_collapse = prm[LAYERS_REM];
  const int num_rot = prm[NUM_ROTATIONS], b = prm[BABY_STEP], g = prm[GIANT_STEP];
  const int num_rot_rem = prm[NUM_ROTATIONS_REM], b_rem = prm[BABY_STEP_REM], g_rem = prm[GIANT_STEP_REM];
  int stop = -1, flag_rem = 0;
  const u32 mdiv4 = m / 4;
  if (rem_collapse != 0) {
    stop = 0;
    flag_rem = 1;
  }
  const int start = encoding ? stop + 1 : 0, end = level_budget;
  const int slots_value = encoding ? (int)slots : (int)mdiv4;
  for (int s = start; s < end; s++) {
    const int shift = encoding ? 1 << ((s - flag_rem) * layers_collapse + rem_collapse) : 1 << (s * layers_collapse);
    for (int j = 0; j < g; j++) out.insert((int32_t)reduce_rotation((j - ((num_rot + 1) / 2) + 1) * shift, (u32)slots_value));
    for (int i = 0; i < b; i++) out.insert((int32_t)reduce_rotation((g * i) * shift, mdiv4));
  }
  if (flag_rem) {
    const int s = level_budget - flag_rem;
    const int shift = encoding ? 1 : 1 << (s * layers_collapse);
    for (int j = 0; j < g_rem; j++) out.insert((int32_t)reduce_rotation((j - ((num_rot_rem + 1) / 2) + 1) * shift, (u32)slots_value));
    for (int i = 0; i < b_rem; i++) out.insert((int32_t)reduce_rotation(g_rem * i * shift, mdiv4));
  }
  const u32 slots4 = slots * 4;
  if (slots4 != m)
    for (u32 j = 1; j < m / slots4; j <<= 1) out.insert((int32_t)(j * slots));
}

// Bootstrap_keygen :1194-1226
void bootstrap_keygen(Precom* pre) {
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  if (pre->keys) return;
  Context& c = ctx();
  UniformScope shared_by_all_images;
  const u32 m = 2 * c.N;
  std::set<int32_t> idx;
  find_rot_index(idx, pre, pre->slots, m, true);
  find_rot_index(idx, pre, pre->slots, m, false);
  idx.erase(0);
  idx.erase((int32_t)(m / 4));
  for (int32_t r : idx) ensure_rot_key(r);
  ensure_auto_key(m - 1);  // conjugation key
  pre->keys = true;
}

// ---------------------------------------------------------------------------------------------
// extended-basis (PQ) helpers for the hoisted BSGS of Rotate_iteration
// ---------------------------------------------------------------------------------------------
std::vector<u64> p_mod_q(u32 level) {
  Context& c = ctx();
  std::vector<u64> s(level);
  for (u32 i = 0; i < level; ++i) {
    unsigned __int128 r = 1;
    for (u32 j = 0; j < c.K; ++j) r = (r * (c.primes[c.L + j] % c.primes[i])) % c.primes[i];
    s[i] = (u64)r;
  }
  return s;
}

// Fast_rotate_ext ckks_evaluator.c:539-575: automorphism( <key, digits> [+ P*c0] ) in the PQ basis
// defer_k != nullptr: the automorphism is left to the consumer (acehip_bsgs_inner_rot reads its inputs through it): rot gets
// the inner product as it is and *defer_k the automorphism index
void fast_rotate_ext(Ct& rot, Ct& in, int32_t rotation, const u64* digits, bool add_first, u32* defer_k = nullptr) {
  Context& c = ctx();
  const u32 l = in.level();
  const u32 k = ensure_rot_key(rotation);
  SwitchKeyStore* key = ensure_auto_key(k);
  Ct tmp;
  ev::init(tmp, l, c.K, in.c._scaling_factor, in.c._sf_degree, in.c._slots, false);  // the inner product writes every limb
  u64 *t0 = q_limbs(&tmp.c._c0_poly), *t1 = q_limbs(&tmp.c._c1_poly);
  if (add_first) {  // + P*c0 on the q-limbs, in the pass that forms the inner product
    std::vector<u64> pm = p_mod_q(l);
    HIPCHK(acehip_key_inner_product_add(c.hip, t0, t1, key->data, digits, l, q_limbs(&in.c._c0_poly), pm.data(), nullptr));
  } else {
    HIPCHK(acehip_key_inner_product(c.hip, t0, t1, key->data, digits, l, nullptr));
  }
  if (defer_k) {
    *defer_k = k;
    rot.take(tmp);
    return;
  }
  ev::init(rot, l, c.K, in.c._scaling_factor, in.c._sf_degree, in.c._slots, false);  // the automorphism writes every limb
  const uint32_t* perm = acehip_auto_order(c.hip, k);
  q_rotate(q_limbs(&rot.c._c0_poly), t0, perm, l, 0, l + c.K);
  q_rotate(q_limbs(&rot.c._c1_poly), t1, perm, l, 0, l + c.K);
}

// Switch_key_ext :462-490 with add_first: (P*c0, P*c1) on the q-limbs, zero p-limbs
void switch_key_ext(Ct& res, Ct& in) {
  Context& c = ctx();
  const u32 l = in.level();
  ev::init(res, l, c.K, in.c._scaling_factor, in.c._sf_degree, in.c._slots);  // zero-filled
  std::vector<u64> pm = p_mod_q(l);
  q_scalars(ACEHIP_HW_MULC, q_limbs(&res.c._c0_poly), q_limbs(&in.c._c0_poly), pm.data(), l, 0, l);
  q_scalars(ACEHIP_HW_MULC, q_limbs(&res.c._c1_poly), q_limbs(&in.c._c1_poly), pm.data(), l, 0, l);
}

// Mul_plaintext in the PQ basis with a plaintext encoded at a level >= the ciphertext's (Derive_plain)
void mul_plain_ext(Ct& res, Ct& a, PLAINTEXT* p, bool accumulate) {
  Context& c = ctx();
  const u32 l = a.level();
  POLYNOMIAL pv = p->_poly;  // view: first l q-limbs + the plaintext's p-limbs
  RT_ASSERT(pv._num_primes >= l && pv._num_primes_p == c.K, "plaintext level too low for the ciphertext");
  pv._num_primes = l;
  if (!accumulate) ev::init(res, l, c.K, a.c._scaling_factor * p->_scaling_factor, a.c._sf_degree + p->_sf_degree, a.c._slots);
  poly_ew(accumulate ? Op::MulAdd : Op::Mul, &res.c._c0_poly, &a.c._c0_poly, &pv, true);
  poly_ew(accumulate ? Op::MulAdd : Op::Mul, &res.c._c1_poly, &a.c._c1_poly, &pv, true);
}

void poly_add_ext(POLYNOMIAL* r, POLYNOMIAL* a, POLYNOMIAL* b) { poly_ew(Op::Add, r, a, b, true); }

// Rotate_iteration :1237-1381
void rotate_iteration(Ct& result, Precom* pre, std::vector<std::vector<PLAINTEXT*>>& conj_pre,
                      std::vector<std::vector<int32_t>>& rot_in, std::vector<std::vector<int32_t>>& rot_out, int step,
                      bool encoding, bool is_rem) {
  Context& c = ctx();
  const int* prm = encoding ? pre->enc : pre->dec;
  const int level_budget = prm[LEVEL_BUDGET];
  const int giant_step = is_rem ? prm[GIANT_STEP_REM] : prm[GIANT_STEP];
  const int baby_step = is_rem ? prm[BABY_STEP_REM] : prm[BABY_STEP];
  const int num_rot = is_rem ? prm[NUM_ROTATIONS_REM] : prm[NUM_ROTATIONS];
  const int level_idx = encoding ? level_budget - 1 : 0;
  if (is_rem || step != level_idx) ev::rescale(result, result);
  const u32 l = result.level();
  const size_t E = (size_t)(l + c.K) * c.N;
  // hoisted ModUp of c1 (Switch_key_precompute), shared by all inner rotations
  const u32 nd = acehip_num_decomp(c.hip, l);
  u64* digits = dalloc(nd * E, false);
  HIPCHK(acehip_modup_digits(c.hip, digits, q_limbs(&result.c._c1_poly), l, nullptr));
  // inner_i = sum_j fast_rot[j] (*) diag[giant_step*i + j] for all baby steps in one pass over the diagonals
  // (acehip_bsgs_inner_rot); the per-output multiply-accumulate chains are the fallback for shapes the kernel does not take
  bool fused = giant_step <= 16 && baby_step <= 16 && giant_step * baby_step <= 128;
  u32 pt_q = 0;
  if (fused) {
    for (int i = 0; i < baby_step && fused; i++)
      for (int j = 0; j < giant_step; j++) {
        if (j > 0 && giant_step * i + j == num_rot) continue;  // the one diagonal a full grid would have too many
        PLAINTEXT* p = conj_pre[step][giant_step * i + j];
        const u32 nq = (u32)(p->_poly._num_alloc_primes - p->_poly._num_primes_p);
        if (p->_poly._num_primes_p != c.K || p->_poly._num_primes < l || (pt_q && nq != pt_q)) fused = false;
        pt_q = nq;
      }
  }
  // with the fused kernel the automorphisms of the hoisted rotations are applied where that kernel reads its inputs
  std::vector<Ct> fast_rot(giant_step);
  std::vector<u32> rot_k(giant_step, 0);
  if (fused) {
    // every hoisted rotation's <key, digits> + P*c0 (Fast_rotate_ext without its automorphism) in ONE pass over the digits
    // (acehip_key_inner_products): they are beta (l+K) limbs per image and were read once per rotation
    std::vector<u64*> a0, a1;
    std::vector<const u64*> keys;
    for (int j = 0; j < giant_step; j++) {
      const int32_t val = rot_in[step][j];
      if (val == 0) {
        switch_key_ext(fast_rot[j], result);
        continue;
      }
      rot_k[j] = ensure_rot_key(val);
      SwitchKeyStore* key = ensure_auto_key(rot_k[j]);
      ev::init(fast_rot[j], l, c.K, result.c._scaling_factor, result.c._sf_degree, result.c._slots, false);  // the inner product writes every limb
      a0.push_back(q_limbs(&fast_rot[j].c._c0_poly));
      a1.push_back(q_limbs(&fast_rot[j].c._c1_poly));
      keys.push_back(key->data);
    }
    if (!keys.empty()) {
      std::vector<u64> pm = p_mod_q(l);
      HIPCHK(acehip_key_inner_products(c.hip, a0.data(), a1.data(), keys.data(), (u32)keys.size(), digits, l, q_limbs(&result.c._c0_poly),
                                       pm.data(), nullptr));
    }
  } else {
    for (int j = 0; j < giant_step; j++) {
      const int32_t val = rot_in[step][j];
      if (val != 0) fast_rotate_ext(fast_rot[j], result, val, digits, true, nullptr);
      else switch_key_ext(fast_rot[j], result);
    }
  }
  dfree(digits);
  POLYNOMIAL first{};
  poly_alloc(&first, c.N, l, c.K);
  first._is_ntt = true;
  std::vector<Ct> inners(baby_step);
  if (fused) {
    std::vector<u64*> o0(baby_step), o1(baby_step);
    std::vector<const u64*> i0(giant_step), i1(giant_step), pts((size_t)baby_step * giant_step, nullptr);
    for (int j = 0; j < giant_step; j++) {
      i0[j] = q_limbs(&fast_rot[j].c._c0_poly);
      i1[j] = q_limbs(&fast_rot[j].c._c1_poly);
    }
    for (int i = 0; i < baby_step; i++) {
      PLAINTEXT* p0 = conj_pre[step][giant_step * i];
      ev::init(inners[i], l, c.K, fast_rot[0].c._scaling_factor * p0->_scaling_factor, fast_rot[0].c._sf_degree + p0->_sf_degree,
               fast_rot[0].c._slots, false);  // the kernel writes every limb
      o0[i] = q_limbs(&inners[i].c._c0_poly);
      o1[i] = q_limbs(&inners[i].c._c1_poly);
      for (int j = 0; j < giant_step; j++)
        if (j == 0 || giant_step * i + j != num_rot)
          pts[(size_t)i * giant_step + j] = q_limbs(&conj_pre[step][giant_step * i + j]->_poly);
    }
    HIPCHK(acehip_bsgs_inner_rot(c.hip, o0.data(), o1.data(), i0.data(), i1.data(), rot_k.data(), pts.data(), (u32)giant_step, (u32)baby_step,
                                 pt_q, l, nullptr));
  }
  Ct outer;
  for (int i = 0; i < baby_step; i++) {
    const int giant = giant_step * i;
    Ct& inner = inners[i];
    if (!fused) {
      mul_plain_ext(inner, fast_rot[0], conj_pre[step][giant], false);
      for (int j = 1; j < giant_step; j++)
        if (giant + j != num_rot) mul_plain_ext(inner, fast_rot[j], conj_pre[step][giant + j], true);
    }
    if (i == 0) {
      // first = inner.c0, outer = (0, inner.c1): by exchanging blocks -- `first` was allocated zero-filled with the very shape of
      // inner.c0 (l q-limbs + K p-limbs), so after the exchange inner.c0 IS the zero polynomial the reference fills in
      std::swap(first._data, inner.c._c0_poly._data);
      outer.take(inner);
    } else {
      const int32_t val = rot_out[step][i];
      if (val != 0) {
        // c1 back to Q, re-raise, key-switch + automorphism in PQ; c0 only needs the automorphism
        u64* c1q = dalloc((size_t)l * c.N, false);
        HIPCHK(acehip_mod_down(c.hip, c1q, q_limbs(&inner.c._c1_poly), l, nullptr));
        const u32 k = ensure_rot_key(val);
        // first += automorphism(inner.c0), outer += automorphism(key-switched c1): each sum in the pass that applies the map
        HIPCHK(acehip_rotate_add2(c.hip, q_limbs(&first), nullptr, q_limbs(&first), nullptr, q_limbs(&inner.c._c0_poly), nullptr, k, l, 0,
                                  l + c.K, nullptr));
        u64* idig = dalloc(nd * E, false);
        HIPCHK(acehip_modup_digits(c.hip, idig, c1q, l, nullptr));
        // (without add_first Fast_rotate_ext takes only level and scale from its ciphertext argument: the reduced c1 itself
        // enters through its digits)
        Ct tmp;
        u32 k2 = 0;
        fast_rotate_ext(tmp, inner, val, idig, false, &k2);
        dfree(idig);
        dfree(c1q);
        HIPCHK(acehip_rotate_add2(c.hip, q_limbs(&outer.c._c0_poly), q_limbs(&outer.c._c1_poly), q_limbs(&outer.c._c0_poly),
                                  q_limbs(&outer.c._c1_poly), q_limbs(&tmp.c._c0_poly), q_limbs(&tmp.c._c1_poly), k2, l, 0, l + c.K, nullptr));
      } else {
        poly_add_ext(&first, &first, &inner.c._c0_poly);
        poly_add_ext(&outer.c._c1_poly, &outer.c._c1_poly, &inner.c._c1_poly);
      }
    }
  }
  poly_add_ext(&outer.c._c0_poly, &outer.c._c0_poly, &first);
  Ct out;
  ev::init(out, l, 0, outer.c._scaling_factor, outer.c._sf_degree, outer.c._slots, false);
  HIPCHK(acehip_mod_down2(c.hip, q_limbs(&out.c._c0_poly), q_limbs(&out.c._c1_poly), q_limbs(&outer.c._c0_poly),
                          q_limbs(&outer.c._c1_poly), l, nullptr));
  poly_free(&first);
  result.take(out);
}

// Coeff_slots_transform :1383-1492
void coeff_slots_transform(Ct& result, Ct& ciph, std::vector<std::vector<PLAINTEXT*>>& conj_pre, Precom* pre, bool encoding) {
  Context& c = ctx();
  const u32 order = 2 * c.N, slots = ciph.c._slots;
  const int* prm = encoding ? pre->enc : pre->dec;
  const int level_budget = prm[LEVEL_BUDGET], layers_collapse = prm[LAYERS_COLL], rem_collapse = prm[LAYERS_REM];
  const int num_rots = prm[NUM_ROTATIONS], g = prm[GIANT_STEP], b = prm[BABY_STEP];
  const int num_rots_rem = prm[NUM_ROTATIONS_REM], g_rem = prm[GIANT_STEP_REM], b_rem = prm[BABY_STEP_REM];
  int stop = -1, flag_rem = 0;
  if (rem_collapse) {
    stop = 0;
    flag_rem = 1;
  }
  const int start = encoding ? stop + 1 : 0, end = encoding ? level_budget : level_budget - flag_rem;
  const u32 slots_value = encoding ? slots : order / 4;
  std::vector<std::vector<int32_t>> rot_in(level_budget), rot_out(level_budget);
  const u32 rem_index = encoding ? 0 : (u32)level_budget - 1;
  for (u32 i = 0; i < (u32)level_budget; i++) {
    rot_in[i].assign((flag_rem == 1 && i == rem_index) ? num_rots_rem + 1 : num_rots + 1, 0);
    rot_out[i].assign(b + b_rem, 0);
  }
  for (int s = start; s < end; s++) {
    const int shift = encoding ? ((s - flag_rem) * layers_collapse + rem_collapse) : (s * layers_collapse);
    for (int j = 0; j < g; j++) rot_in[s][j] = (int32_t)reduce_rotation((j - ((num_rots + 1) / 2) + 1) * (1 << shift), slots_value);
    for (int i = 0; i < b; i++) rot_out[s][i] = (int32_t)reduce_rotation((g * i) * (1 << shift), order / 4);
  }
  if (flag_rem) {
    const int s = encoding ? stop : level_budget - flag_rem;
    const int shift_value = encoding ? 1 : (1 << (s * layers_collapse));
    for (int j = 0; j < g_rem; j++) rot_in[s][j] = (int32_t)reduce_rotation((j - ((num_rots_rem + 1) / 2) + 1) * shift_value, slots_value);
    for (int i = 0; i < b_rem; i++) rot_out[s][i] = (int32_t)reduce_rotation((g_rem * i) * shift_value, order / 4);
  }
  ev::copy(result, ciph);
  if (encoding) {
    for (int s = end - 1; s > start - 1; s--) rotate_iteration(result, pre, conj_pre, rot_in, rot_out, s, encoding, false);
  } else {
    for (int s = start; s < end; s++) rotate_iteration(result, pre, conj_pre, rot_in, rot_out, s, encoding, false);
  }
  if (flag_rem) {
    const int s = encoding ? stop : level_budget - flag_rem;
    rotate_iteration(result, pre, conj_pre, rot_in, rot_out, s, encoding, true);
  }
}

// ---------------------------------------------------------------------------------------------
// Chebyshev series evaluation, Paterson-Stockmeyer (ckks_chebyshev.c)
// ---------------------------------------------------------------------------------------------
using Vd = std::vector<double>;

u32 degree_of(const Vd& v) {  // Get_degree_from_coeffs :38-50
  if (v.empty()) return 0;
  u32 deg = 1;
  for (int i = (int)v.size() - 1; i > 0; i--) {
    if (v[i] == 0) deg += 1;
    else break;
  }
  return (u32)v.size() - deg;
}
bool is_even_poly(const Vd& v) {
  const u32 d = degree_of(v);
  for (u32 i = 1; i <= d; i += 2)
    if (v[i] != 0.) return false;
  return true;
}
void compute_degree_ps(u32 n, u32& k, u32& m) {  // Compute_degree_ps :95-131 (table part, n <= 2204)
  static const u32 ranges[16] = {2, 11, 13, 17, 55, 59, 76, 239, 247, 284, 991, 1007, 1083, 2015, 2031, 2204};
  static const u32 values[16] = {1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6, 7, 6};
  RT_ASSERT(n > 0 && n <= 2204, "unsupported polynomial degree");
  m = 0;
  for (int i = 0; i < 16; ++i)
    if (n - 1 < ranges[i]) {
      m = values[i];
      break;
    }
  k = (u32)floor((double)n / ((1 << m) - 1)) + 1;
}
const double kPrec = 9.5367431640625e-07;
bool not_one(double v) { return (1 - kPrec >= v) || (1 + kPrec <= v); }

// Long_div_chebyshev :150-263
void long_div_chebyshev(Vd& q, Vd& r, const Vd& f, const Vd& g) {
  u32 n = degree_of(f);
  const u32 k = degree_of(g);
  RT_ASSERT(n == f.size() - 1, "The dominant coefficient of the divident is zero");
  RT_ASSERT(k == g.size() - 1, "The dominant coefficient of the divisor is zero");
  r = f;
  if (n >= k) {
    q.assign(n - k + 1, 0.0);
    while (n > k) {
      double qnk = 2 * r.back();
      q[n - k] = qnk;
      if (not_one(g[k])) q[n - k] = qnk / g.back();
      Vd d(n + 1, 0.0);
      if (k == n - k) {
        d[0] = 2 * g[n - k];
        for (u32 i = 1; i < 2 * k + 1; i++) d[i] = g[(u32)abs((int)(n - k) - (int)i)];
      } else if ((int)k > (int)(n - k)) {
        d[0] = 2 * g[n - k];
        for (u32 i = 1; i < k - (n - k) + 1; i++) d[i] = g[(u32)abs((int)(n - k) - (int)i)] + g[n - k + i];
        for (u32 i = k - (n - k) + 1; i < n + 1; i++) d[i] = g[(u32)abs((int)i - (int)n + (int)k)];
      } else {
        d[n - k] = g[0];
        for (u32 i = n - 2 * k; i < n + 1; i++) d[i] = g[(u32)abs((int)i - (int)n + (int)k)];
      }
      const double r_back = r.back();
      if (not_one(r_back))
        for (auto& x : d) x *= r_back;
      const double g_back = g.back();
      if (not_one(g_back))
        for (auto& x : d) x /= g_back;
      for (size_t i = 0; i < r.size(); ++i) r[i] -= d[i];
      if (r.size() > 1) {
        n = degree_of(r);
        r.resize(n + 1, 0.0);
      }
    }
    if (n == k) {
      const double r_back = r.back(), g_back = g.back();
      q[0] = r_back;
      if (not_one(g_back)) q[0] = r_back / g_back;
      Vd d = g;
      if (not_one(r_back))
        for (auto& x : d) x *= r_back;
      if (not_one(g_back))
        for (auto& x : d) x /= g_back;
      for (size_t i = 0; i < r.size(); ++i) r[i] -= d[i];
      if (r.size() > 1) {
        n = degree_of(r);
        r.resize(n + 1, 0.0);
      }
    }
    q[0] *= 2;
  } else {
    q.assign(1, 0.0);
  }
}

// Eval_linear_wsum(_mutable) :275-312: out = rescale( sum_i w_i * T_i )
void eval_linear_wsum(Ct& out, std::vector<Ct>& t_list, size_t size, const double* weights) {
  bool first = true;
  Ct tmp;
  for (size_t i = 0; i < size; i++) {
    if (weights[i] == 0.) continue;
    ev::mul_const(tmp, t_list[i], weights[i]);
    if (first) {
      ev::copy(out, tmp);
      first = false;
    } else {
      ev::add(out, out, tmp);
    }
  }
  RT_ASSERT(!first, "polynomial has no non_zero coefficient");
  ev::rescale(out, out);
}

// Eval_quot_or_rem :314-375
void eval_quot_or_rem(Ct& out, std::vector<Ct>& t_list, const Vd& quot_rem, u32 k, bool is_quotient, bool in_recursion) {
  Vd qr = quot_rem;
  qr.resize(k, 0.0);
  Ct& t_k_1 = t_list[k - 1];
  const size_t dg = degree_of(qr);
  if (dg > 0) {
    eval_linear_wsum(out, t_list, dg, qr.data() + 1);
    if (is_quotient) {
      if (in_recursion) {
        const double quot_last = quot_rem.back();
        Ct sum;
        ev::copy(sum, t_k_1);
        for (u32 i = 0; i < log2(quot_last); i++) ev::add(sum, sum, sum);
        ev::add(out, out, sum);
      } else {
        ev::add(out, out, t_k_1);
        ev::add(out, out, t_k_1);
      }
    } else {
      ev::add(out, out, t_k_1);
    }
  } else {
    ev::copy(out, t_k_1);
    if (is_quotient) {
      const double quot_last = quot_rem.back();
      const u32 end = in_recursion ? (u32)log2(quot_last) : (u32)quot_last;
      for (u32 i = 0; i < end; i++) ev::add(out, out, t_k_1);
    }
  }
  ev::add_const(out, out, quot_rem[0] / 2);
}

// Inner_eval_chebyshev_ps :377-488
void inner_eval_chebyshev_ps(Ct& out, const Vd& coeffs, u32 k, u32 m, std::vector<Ct>& t_list, std::vector<Ct>& t2_list,
                             bool in_recursion) {
  const u32 k2m2k = k * (1u << (m - 1)) - k;
  Vd tkm(k2m2k + k + 1, 0.0);
  tkm.back() = 1;
  Vd div_q, div_r;
  long_div_chebyshev(div_q, div_r, coeffs, tkm);
  Vd r2 = div_r;
  if (k2m2k <= degree_of(div_r)) {
    r2[k2m2k] -= 1;
    r2.resize(degree_of(r2) + 1, 0.0);
  } else {
    r2.resize(k2m2k + 1, 0.0);
    r2.back() = -1;
  }
  Vd divr2_q, divr2_r;
  long_div_chebyshev(divr2_q, divr2_r, r2, div_q);
  size_t s2_len = std::max(divr2_r.size(), (size_t)k2m2k + 1);
  Vd s2 = divr2_r;
  s2.resize(s2_len, 0.0);
  s2[s2_len - 1] = 1;
  Ct cu;
  const u32 dc = degree_of(divr2_q);
  bool flag_c = false;
  if (dc >= 1) {
    if (dc == 1) {
      const double q1 = divr2_q[1];
      if (q1 != 1) {
        ev::mul_const(cu, t_list[0], q1);
        ev::rescale(cu, cu);
      } else {
        ev::copy(cu, t_list[0]);
      }
    } else {
      eval_linear_wsum(cu, t_list, dc, divr2_q.data() + 1);
    }
    ev::add_const(cu, cu, divr2_q[0] / 2);
    flag_c = true;
  }
  Ct qu, su;
  if (degree_of(div_q) > k) inner_eval_chebyshev_ps(qu, div_q, k, m - 1, t_list, t2_list, true);
  else eval_quot_or_rem(qu, t_list, div_q, k, true, in_recursion);
  if (degree_of(s2) > k) inner_eval_chebyshev_ps(su, s2, k, m - 1, t_list, t2_list, true);
  else eval_quot_or_rem(su, t_list, s2, k, false, in_recursion);
  Ct& t2_m_1 = t2_list[m - 1];
  Ct res;
  if (flag_c) {
    ev::set_level(cu, std::min(cu.level(), t2_m_1.level()));
    ev::add(res, t2_m_1, cu);
  } else {
    ev::add_const(res, t2_m_1, divr2_q[0] / 2);
  }
  ev::mul(res, res, qu);
  ev::rescale(res, res);
  ev::add(res, res, su);
  out.take(res);
}

// T_j = 2 * a * b - (1 | y | T_2) helper: prod = a*b; t = rescale(prod + prod)
void cheb_double_product(Ct& t, Ct& a, Ct& b) {
  Ct prod;
  ev::mul(prod, a, b);
  ev::add(t, prod, prod);
  ev::rescale(t, t);
}

// Eval_chebyshev_ps :490-672 with [a,b] = [-1,1]
void eval_chebyshev_ps(Ct& out, Ct& in, const Vd& coeffs) {
  const u32 n = degree_of(coeffs);
  const bool even = is_even_poly(coeffs);
  Vd f2(coeffs.begin(), coeffs.begin() + (coeffs.back() == 0 ? n + 1 : coeffs.size()));
  u32 k, m;
  compute_degree_ps(n, k, m);
  if (even && (k % 2 == 1)) k += 1;
  std::vector<Ct> t_list(k);
  ev::copy(t_list[0], in);
  Ct y;
  ev::copy(y, t_list[0]);
  for (u32 i = 2; i <= k; i++) {
    const u32 j = i - 1;
    if (!(i & (i - 1))) {  // power of two: T_i = 2 T_{i/2}^2 - 1
      cheb_double_product(t_list[j], t_list[i / 2 - 1], t_list[i / 2 - 1]);
      ev::add_const(t_list[j], t_list[j], -1.0);
    } else if (i % 2 == 1) {
      if (even) continue;
      cheb_double_product(t_list[j], t_list[i / 2 - 1], t_list[i / 2]);  // 2 T_{(i-1)/2} T_{(i+1)/2} - y
      ev::sub(t_list[j], t_list[j], y);
    } else {
      u32 ih1 = i / 2;
      if (even && (ih1 % 2 == 1)) ih1 += 1;
      const u32 ih2 = i - ih1;
      cheb_double_product(t_list[j], t_list[ih1 - 1], t_list[ih2 - 1]);
      if (ih1 == ih2) ev::add_const(t_list[j], t_list[j], -1.0);
      else ev::sub(t_list[j], t_list[j], t_list[1]);
    }
  }
  // FIXED_MANUAL: bring every T_i to the level of T_k (:608-624)
  for (size_t i = 1; i < k; i++) {
    if (even && i % 2 == 1) continue;
    if (t_list[i - 1].level() > t_list[k - 1].level()) ev::set_level(t_list[i - 1], t_list[k - 1].level());
  }
  std::vector<Ct> t2_list(m);
  ev::copy(t2_list[0], t_list[k - 1]);
  for (u32 i = 1; i < m; i++) {
    cheb_double_product(t2_list[i], t2_list[i - 1], t2_list[i - 1]);
    ev::add_const(t2_list[i], t2_list[i], -1.0);
  }
  Ct t2km1;
  ev::copy(t2km1, t2_list[0]);
  for (u32 i = 1; i < m; i++) {
    cheb_double_product(t2km1, t2km1, t2_list[i]);
    ev::sub(t2km1, t2km1, t2_list[0]);
  }
  const u32 k2m2k = k * (1u << (m - 1)) - k;
  f2.resize(2 * k2m2k + k + 1, 0.0);
  f2.back() = 1;
  Ct res;
  inner_eval_chebyshev_ps(res, f2, k, m, t_list, t2_list, false);
  ev::sub(res, res, t2km1);
  out.take(res);
}

// Apply_double_angle_iterations :1512-1525
void double_angle(Ct& ct, u32 r) {
  for (int j = 1; j < (int)r + 1; j++) {
    ev::mul(ct, ct, ct);
    ev::add(ct, ct, ct);
    ev::add_const(ct, ct, -1.0 / pow(2.0 * M_PI, pow(2.0, j - (int)r)));
    ev::rescale(ct, ct);
  }
}

// Eval_approx_mod :1553-1582
void eval_approx_mod(Ct& out, Ct& in) {
  SinPoly sp = sin_poly(ctx().hamming);
  if (sp.even) ev::add_const(in, in, -1. / (4. * sp.upper_bound));
  Vd coeffs(sp.coeff, sp.coeff + sp.coeff_size);
  Ct r;
  eval_chebyshev_ps(r, in, coeffs);
  double_angle(r, sp.double_angle);
  out.take(r);
}

}  // namespace

void bootstrap_setup_if_needed() {  // Bootstrap_precom context.c:162-185
  Context& c = ctx();
  const u32 bts_depth = approx_mod_depth(c.hamming) + 3 + 3;
  if (c.L - 1 > bts_depth) {
    Precom* pre;
    {
      RtmScope rtm(RTM_BS_SETUP);
      pre = bootstrap_setup(c.N / 2);
    }
    RtmScope rtm(RTM_BS_KEYGEN);
    bootstrap_keygen(pre);
  }
}

void bootstrap_precom_slots(u32 num_slots) {
  Context& c = ctx();
  const u32 bts_depth = approx_mod_depth(c.hamming) + 3 + 3;
  if (c.L - 1 <= bts_depth) return;  // (the reference: nothing to set up when the depth does not allow a bootstrap)
  UniformScope shared_by_all_images;
  Precom* pre;
  {
    RtmScope rtm(RTM_BS_SETUP);
    pre = bootstrap_setup(num_slots);
  }
  RtmScope rtm(RTM_BS_KEYGEN);
  bootstrap_keygen(pre);
}

void bootstrap_release() {
  for (auto& kv : g_precom) {
    for (auto* tabs : {&kv.second->u0hatt_fft, &kv.second->u0_fft})
      for (auto& row : *tabs)
        for (PLAINTEXT* p : row)
          if (p) {
            poly_free(&p->_poly);
            free(p);
          }
    delete kv.second;
  }
  g_precom.clear();
}

// Eval_bootstrap :1584-1860
void bootstrap(Ct& res, Ct& ciph, u32 raise_level) {
  Context& c = ctx();
  const u32 slots = ciph.c._slots, N = c.N, m = 2 * N;
  Precom* pre = bootstrap_setup(slots);
  bootstrap_keygen(pre);
  RtmScope rtm_eval(RTM_BS_EVAL);
  const int32_t deg = (int32_t)round(log2((double)c.primes[0] / c.sf));
  Ct raised;
  ev::copy(raised, ciph);
  while (raised.c._sf_degree > 1) ev::rescale(raised, raised);
  if (!raise_level) raise_level = c.L;
  RT_ASSERT(raise_level <= c.L, "The raise level must be less than or equal to q_cnt");
  // ModRaise: limb 0 (coefficient domain) spread to raise_level limbs, centred (Transform_values_from_level0 :1527-1551)
  Ct nc;
  ev::init(nc, raise_level, 0, raised.c._scaling_factor, raised.c._sf_degree, slots, false);
  HIPCHK(acehip_mod_raise(c.hip, q_limbs(&nc.c._c0_poly), q_limbs(&nc.c._c1_poly), q_limbs(&raised.c._c0_poly),
                          q_limbs(&raised.c._c1_poly), raise_level, nullptr));
  nc.c._c0_poly._is_ntt = nc.c._c1_poly._is_ntt = true;
  raised.reset();
  auto& u0hatt = pre->u0hatt_fft;
  auto& u0 = pre->u0_fft;
  Ct enc;
  if (slots == m / 4) {
    {
      RtmScope rtm(RTM_BS_COEFF_TO_SLOT);
      coeff_slots_transform(enc, nc, u0hatt, pre, true);
    }
    Ct conj, enc_sub;
    ev::conjugate(conj, enc);
    ev::sub(enc_sub, enc, conj);
    ev::add(enc, enc, conj);
    ev::mul_monomial(enc_sub, enc_sub, 3 * m / 4);
    while (enc.c._sf_degree > 1) {
      ev::rescale(enc, enc);
      ev::rescale(enc_sub, enc_sub);
    }
    {
      RtmScope rtm(RTM_BS_APPROX_MOD);
      eval_approx_mod(enc, enc);
      eval_approx_mod(enc_sub, enc_sub);
    }
    ev::mul_monomial(enc_sub, enc_sub, m / 4);
    ev::add(enc, enc, enc_sub);
    {
      RtmScope rtm(RTM_BS_SLOT_TO_COEFF);
      coeff_slots_transform(res, enc, u0, pre, false);
    }
  } else {
    // sparsely packed: partial sums first (:1770-1777)
    Ct temp;
    {
      RtmScope rtm(RTM_BS_PARTIAL_SUM);
      for (u32 j = 1; j < N / (2 * slots); j <<= 1) {
        ev::rotate(temp, nc, (int32_t)(j * slots));
        ev::add(nc, nc, temp);
      }
    }
    {
      RtmScope rtm(RTM_BS_COEFF_TO_SLOT);
      coeff_slots_transform(enc, nc, u0hatt, pre, true);
    }
    Ct conj;
    ev::conjugate(conj, enc);
    ev::add(enc, enc, conj);
    while (enc.c._sf_degree > 1) ev::rescale(enc, enc);
    {
      RtmScope rtm(RTM_BS_APPROX_MOD);
      eval_approx_mod(enc, enc);
    }
    {
      RtmScope rtm(RTM_BS_SLOT_TO_COEFF);
      coeff_slots_transform(res, enc, u0, pre, false);
    }
    Ct rot;
    ev::rotate(rot, res, (int32_t)slots);
    ev::add(res, res, rot);
  }
  if (Get_rtlib_config(CONF_BTS_CLEAR_IMAG /* RT_BTS_CLEAR_IMAG or Set_rtlib_config */) != 0 && deg >= 1) {
    Ct conj;
    ev::conjugate(conj, res);
    ev::add(res, res, conj);
    const u64 ratio = (u64)pow(2., deg - 1);
    if (ratio > 1) ev::mul_integer(res, res, ratio);
  } else {
    ev::mul_integer(res, res, (u64)pow(2., deg));
  }
  while (res.c._sf_degree > 1) ev::rescale(res, res);
  if (res.level() <= ciph.level()) {
    fprintf(stderr, "WARNING: q_cnt(after):%u <= q_cnt(before):%u, bootstrapping earns too small, just return.\n", res.level(), ciph.level());
    ev::copy(res, ciph);
  }
}

}  // namespace rt

using namespace rt;

extern "C" {

// cipher_eval.c:366-404
CIPHER Bootstrap(CIPHER res, CIPHER ciph, uint32_t level_after_bts) {
  Context& c = ctx();
  const u32 bts_depth = approx_mod_depth(c.hamming) + 3 + 3;
  if (ciph->_sf_degree == 1 && ciph->_c0_poly._num_primes >= level_after_bts) {
    RtmScope rtm(RTM_BS_COPY);  // cipher_eval.c:389-396 files the early return under BS_COPY
    if (res != ciph) Copy_ciph(res, ciph);
    return res;
  }
  RtmScope rtm(RTM_BOOTSTRAP);
  RT_ASSERT(!level_after_bts || level_after_bts <= c.L - bts_depth, "The level set after bootstrapping is excessively high");
  const u32 raise_level = level_after_bts ? level_after_bts + bts_depth : c.L;
  Ct in, out;
  double t0 = 0;
  if (c.profile) {  // attribute device time to the bootstrap: drain the queue on both sides
    sync();
    t0 = wall_s();
  }
  ev::from_ciph(in, ciph);
  bootstrap(out, in, raise_level);
  ev::to_ciph(res, out);
  if (c.profile) {
    sync();
    c.t_bootstrap += wall_s() - t0;
    c.n_bootstrap++;
  }
  return res;
}

}  // extern "C"
