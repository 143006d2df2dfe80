// rt_valid.cpp -- the rest of the API surface the reference exposes through rt_ant/rt_ant.h beyond what the checked-in generated
// programs call: the validation helpers the code generator emits when validation is switched on (include/ckks/cipher_valid.h:
// <op>_msg / _rtv / _ref; emitted by fhe/sihe/ir2c_handler.h:69 and the *_rtv handlers of vector2sihe_impl.h), the diagnostics and scale
// operators of include/ckks/cipher_eval.h (Get_msg_with_imag, Print_cipher_*, Real_relu, Upscale_ciph, Downscale_ciph), the with-scale
// encoders of include/ckks/plain_eval.h, Print_poly_lite (poly_eval.h) and two context entries (Get_part_size, Bootstrap_precom).
// Everything that touches ciphertexts goes through the same device paths as the rest of the shim (decrypt / decode / encode / Mul_plain /
// Rescale): nothing here computes a ciphertext operation on the host.  Message-level reference arithmetic (the *_ref family: what a plain
// tensor program would have produced) is host code by definition, as it is in the reference (src/ckks/cipher_valid.c).
#include <cmath>
#include <cstring>

#include "rt_internal.hpp"
#include "common/rt_config.h"

namespace rt {

// Encode_val_at_level_with_scale ckks_encoder.c:509-592: the constant polynomial of (int64)(value / 2^log_approx * scale), scaled back
void encode_value_with_scale(PLAINTEXT* res, double value, u32 level, double scale) {
  RtmScope rtm(RTM_ENCODE_VALUE);
  Context& c = ctx();
  RT_ASSERT(res != nullptr, "null plaintext");
  if (level == 0) level = c.L;
  RT_ASSERT(level <= c.L, "level should not be larger than mul_depth + 1");
  RT_ASSERT(scale > 0, "invalid scale for encode");
  u32 sf_degree = (u32)floor(scale / c.sf);
  if (scale > c.sf * sf_degree) sf_degree++;
  const u32 N = c.N;
  init_plaintext(res, N / 2, level, 0, scale, sf_degree);
  const int MAX_BITS_IN_WORD = 61, MAX_LOG_STEP = 60;
  const double mag = fabs(value * scale);
  const int32_t log_scale = mag >= 1.0 ? (int32_t)ceil(log2(mag)) : 0;  // (below 1 the value fits a word anyway; log2(0) is not asked for)
  const int32_t log_valid = log_scale <= MAX_BITS_IN_WORD ? log_scale : MAX_BITS_IN_WORD;
  const int32_t log_approx = log_scale - log_valid;
  const double scaled = value / pow(2, log_approx) * scale;
  RT_ASSERT(scaled <= 9.2e18 && scaled >= -9.2e18, "encode %f with scale %f overflow, please choose a smaller scale", value, scale);
  const int64_t val = (int64_t)scaled;
  std::vector<u64> consts(level);
  for (u32 i = 0; i < level; ++i) {
    const int64_t q = (int64_t)c.primes[i];
    int64_t r = val % q;
    if (r < 0) r += q;
    consts[i] = (u64)r;
  }
  if (log_approx > 0) {  // Scale_back_up_by_approxfactor :414-455
    auto mulmod = [](u64 a, u64 b, u64 m) { return (u64)(((unsigned __int128)a * b) % m); };
    int32_t rest = log_approx, log_step = rest <= MAX_BITS_IN_WORD ? rest : MAX_BITS_IN_WORD;
    std::vector<u64> approx(level);
    for (u32 i = 0; i < level; ++i) approx[i] = (1ull << log_step) % c.primes[i];
    rest -= log_step;
    while (rest > 0) {
      log_step = rest <= MAX_LOG_STEP ? rest : MAX_LOG_STEP;
      for (u32 i = 0; i < level; ++i) approx[i] = mulmod(approx[i], (1ull << log_step) % c.primes[i], c.primes[i]);
      rest -= log_step;
    }
    for (u32 i = 0; i < level; ++i) consts[i] = mulmod(consts[i], approx[i], c.primes[i]);
  }
  fill_zero((u64*)q_limbs(&res->_poly), (size_t)level * N, level);
  q_scalars(ACEHIP_HW_ADDC, q_limbs(&res->_poly), q_limbs(&res->_poly), consts.data(), level, 0, level);
  res->_poly._is_ntt = true;
}

// the decoded slots of a ciphertext, real and imaginary parts; a ciphertext over the extended basis is brought down first
// (Get_msg / Get_msg_with_imag cipher_eval.c:129-169)
static void decrypt_decode(std::vector<cplx>& out, CIPHER ciph) {
  CIPHERTEXT down;
  memset(&down, 0, sizeof(down));
  CIPHER src = ciph;
  if (ciph->_c0_poly._num_primes_p != 0) {
    down._slots = ciph->_slots;
    down._scaling_factor = ciph->_scaling_factor;
    down._sf_degree = ciph->_sf_degree;
    poly_alloc(&down._c0_poly, ciph->_c0_poly._ring_degree, ciph->_c0_poly._num_primes, 0);
    poly_alloc(&down._c1_poly, ciph->_c1_poly._ring_degree, ciph->_c1_poly._num_primes, 0);
    Mod_down(&down._c0_poly, &ciph->_c0_poly);
    Mod_down(&down._c1_poly, &ciph->_c1_poly);
    src = &down;
  }
  PLAINTEXT pt;
  memset(&pt, 0, sizeof(pt));
  decrypt(&pt, src);
  decode(out, &pt);
  poly_free(&pt._poly);
  if (src == &down) {
    poly_free(&down._c0_poly);
    poly_free(&down._c1_poly);
  }
}

}  // namespace rt

using namespace rt;

namespace {
typedef std::complex<double> dcmplx;

// Print_msg_range cipher_eval.c:171-203: extremes of the real and of the imaginary parts with their positions
void print_msg_range(FILE* fp, const std::vector<cplx>& m) {
  if (m.empty()) return;
  double max_r = m[0].real(), min_r = max_r, max_i = m[0].imag(), min_i = max_i;
  uint32_t max_rp = 0, min_rp = 0, max_ip = 0, min_ip = 0;
  for (uint32_t i = 1; i < m.size(); ++i) {
    const double re = m[i].real(), im = m[i].imag();
    if (re > max_r) {
      max_r = re;
      max_rp = i;
    } else if (re < min_r) {
      min_r = re;
      min_rp = i;
    }
    if (im > max_i) {
      max_i = im;
      max_ip = i;
    } else if (im < min_i) {
      min_i = im;
      min_ip = i;
    }
  }
  fprintf(fp, "msg_range[%d]: real(%.17f[%d] ~ %.17f[%d]), imag(%.17f[%d] ~ %.17f[%d])\n", (int)m.size(), min_r, min_rp, max_r, max_rp, min_i,
          min_ip, max_i, max_ip);
}

double* reals_of(const std::vector<cplx>& v) {
  double* d = (double*)malloc(sizeof(double) * v.size());
  for (size_t i = 0; i < v.size(); ++i) d[i] = v[i].real();
  return d;
}

// ---- message-level reference arithmetic of cipher_valid.c (plain tensor programs) ----
double* add_impl(const double* a, const double* b, uint64_t len) {
  double* r = (double*)malloc(sizeof(double) * len);
  for (uint64_t i = 0; i < len; ++i) r[i] = a[i] + b[i];
  return r;
}
double* relu_impl(const double* m, uint64_t len) {  // :126-141 (also reports the value range it saw)
  if (len == 0) return (double*)malloc(sizeof(double));
  double lo = m[0], hi = m[0];
  double* r = (double*)malloc(sizeof(double) * len);
  for (uint64_t i = 0; i < len; ++i) {
    hi = m[i] > hi ? m[i] : hi;
    lo = m[i] < lo ? m[i] : lo;
    r[i] = m[i] < 0 ? 0 : m[i];
  }
  fprintf(stderr, "INFO: relu value range [%.4f, %.4f].\n", lo, hi);
  return r;
}
// zero border of ph rows / pw columns around every (n, c) plane (:166-187)
double* pad_planes(const double* m, int n, int c, int h, int w, int ph, int pw) {
  const int nh = h + 2 * ph, nw = w + 2 * pw;
  double* r = (double*)calloc((size_t)n * c * nh * nw, sizeof(double));
  for (int p = 0; p < n * c; ++p)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) r[((size_t)p * nh + y + ph) * nw + x + pw] = m[((size_t)p * h + y) * w + x];
  return r;
}
// direct convolution, unit strides, NCHW input, weights [kn][kc][kh][kw], one bias per output channel (:189-250)
double* conv_impl(const double* in, int n, int c, int h, int w, const float* weight, int kn, int kc, int kh, int kw, const float* bias, int bw,
                  int sh, int sw, int ph, int pw) {
  RT_ASSERT(sh == 1 && sw == 1, "TODO: strides not 1");
  RT_ASSERT(kc == c, "channel mismatch");
  RT_ASSERT(kn == bw, "bias length mismatch");
  double* padded = nullptr;
  if (ph != 0 || pw != 0) {
    padded = pad_planes(in, n, c, h, w, ph, pw);
    in = padded;
    h += 2 * ph;
    w += 2 * pw;
  }
  const int oh = (h - (kh - 1) - 1) / sh + 1, ow = (w - (kw - 1) - 1) / sw + 1;
  double* out = (double*)malloc(sizeof(double) * (size_t)n * kn * oh * ow);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < kn; ++j)
      for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
          double total = 0;
          for (int m = 0; m < kc; ++m)
            for (int dy = 0; dy < kh; ++dy) {
              if (y * sh + dy >= h) continue;
              for (int dx = 0; dx < kw; ++dx) {
                if (x * sw + dx >= w) continue;
                total += in[(((size_t)i * c + m) * h + y * sh + dy) * w + x * sw + dx] * (double)weight[(((size_t)j * kc + m) * kh + dy) * kw + dx];
              }
            }
          out[(((size_t)i * kn + j) * oh + y) * ow + x] = total + bias[j];
        }
  free(padded);
  return out;
}
double* gemm_impl(const double* in, int h, int w, const float* weight, int wh, int ww, const float* bias, int bw) {  // :272-293
  RT_ASSERT(h == 1, "height not 1?");
  RT_ASSERT(w == ww, "weight width mismatch");
  RT_ASSERT(wh == bw, "bias width mismatch");
  double* out = (double*)malloc(sizeof(double) * (size_t)h * wh);
  for (int j = 0; j < wh; ++j) {
    double t = 0;
    for (int k = 0; k < ww; ++k) t += in[k] * (double)weight[(size_t)j * ww + k];
    out[j] = t;
  }
  for (int i = 0; i < bw; ++i) out[i] += bias[i];
  return out;
}
// non-overlapping kh x kw windows (the stride arguments are not consulted: the window of output (k, l) starts at row k*kh, column l*kw,
// k runs over the output WIDTH and l over the output HEIGHT -- cipher_valid.c:313-345 as it is, square planes in every use)
double* avg_pool_impl(const double* in, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw) {
  double* padded = nullptr;
  if (ph != 0 || pw != 0) {
    padded = pad_planes(in, n, c, h, w, ph, pw);
    in = padded;
    h += 2 * ph;
    w += 2 * pw;
  }
  const int oh = (h - kh) / sh + 1, ow = (w - kw) / sw + 1;
  const double scale = 1.0 / (double)(kh * kw);
  double* out = (double*)malloc(sizeof(double) * (size_t)n * c * oh * ow);
  for (int p = 0; p < n * c; ++p)
    for (int k = 0; k < ow; ++k)
      for (int l = 0; l < oh; ++l) {
        double sum = 0.0;
        for (int m = 0; m < kh; ++m)
          for (int q = 0; q < kw; ++q) sum += in[((size_t)p * h + k * kh + m) * w + l * kw + q];
        out[((size_t)p * oh + k) * ow + l] = sum * scale;
      }
  free(padded);
  return out;
}
double* gap_impl(const double* in, int n, int c, int h, int w) {  // :375-387
  double* out = (double*)malloc(sizeof(double) * (size_t)n * c);
  const double scale = 1.0 / (double)(h * w);
  for (int i = 0; i < n * c; ++i) {
    double t = 0;
    for (int j = 0; j < h * w; ++j) t += in[(size_t)i * h * w + j];
    out[i] = t * scale;
  }
  return out;
}
int g_relu_n, g_relu_e2e_n, g_bts_n, g_conv_n, g_conv_e2e_n, g_gemm_n, g_gemm_e2e_n, g_pool_n, g_pool_e2e_n, g_gap_n, g_gap_e2e_n;
}  // namespace

extern "C" {

// ---- common/rtlib_timing.h, trace.h, rt_config.h (declared there; the enum values are those of rt_internal.hpp RtmId) ----
void Append_rtlib_timing(RTLIB_TIMING_ID id, uint64_t nsec) {
  if ((int)id >= 0 && id < RTM_LAST) rtm_add((int)id, nsec);
}
void Report_rtlib_timing(void) { rtm_report(); }

static FILE* g_trace_fp = nullptr;
static bool g_trace_on = false, g_trace_env_read = false;
void Set_trace_file(char* filename) {
  if (g_trace_fp != nullptr && g_trace_fp != stdout) fclose(g_trace_fp);
  g_trace_fp = (filename != nullptr && *filename) ? fopen(filename, "w") : nullptr;
}
FILE* Get_trace_file(void) {
  if (g_trace_fp == nullptr && !g_trace_env_read) {  // RTLIB_TRACE_FILE names it the first time it is asked for
    g_trace_env_read = true;
    const char* e = getenv("RTLIB_TRACE_FILE");
    if (e != nullptr && *e) g_trace_fp = fopen(e, "w");
  }
  return g_trace_fp != nullptr ? g_trace_fp : stdout;
}
void Close_trace_file(void) {
  if (g_trace_fp != nullptr && g_trace_fp != stdout) fclose(g_trace_fp);
  g_trace_fp = nullptr;
}
void Set_trace_on(bool v) { g_trace_on = v; }
bool Is_trace_on(void) { return g_trace_on; }

static int64_t g_config[2] = {1, 0};  // CONF_OP_FUSION_DECOMP_MODUP, CONF_BTS_CLEAR_IMAG
static bool g_config_ready = false;
void Init_rtlib_config(void) {
  g_config[0] = 1;
  g_config[1] = 0;
  if (const char* e = getenv("OP_FUSION_DECOMP_MODUP")) g_config[0] = atoi(e) != 0;
  if (const char* e = getenv("RT_BTS_CLEAR_IMAG")) g_config[1] = atoi(e) != 0;
  g_config_ready = true;
}
int64_t Get_rtlib_config(RTLIB_CONFIG_ID id) {
  if (!g_config_ready) Init_rtlib_config();
  return (int)id >= 0 && (int)id < 2 ? g_config[id] : 0;
}
void Set_rtlib_config(RTLIB_CONFIG_ID id, int64_t value) {
  if (!g_config_ready) Init_rtlib_config();
  if ((int)id >= 0 && (int)id < 2) g_config[id] = value;
}

// ---- context.h ----
size_t Get_part_size() { return ctx().alpha; }
void Bootstrap_precom(uint32_t num_slots) { bootstrap_precom_slots(num_slots); }

// ---- plain_eval.h:28-33,44-46 ----
void Encode_plain_from_float_with_scale(PLAIN plain, float* input, size_t len, double scale, uint32_t level) {
  RtmScope rtm(RTM_PT_ENCODE);
  if (len == 1) {
    encode_value_with_scale(plain, (double)*input, level, scale);
    return;
  }
  encode_device_with_scale(plain, stage_to_device(input, len * sizeof(float)), 0, len, level, 0, scale, 0);
}
DCMPLX* Get_dcmplx_msg_from_plain(PLAIN plain) {
  PLAINTEXT pt = *plain;  // decode converts to the coefficient domain in place: work on a copy
  pt._poly._data = nullptr;
  poly_alloc(&pt._poly, plain->_poly._ring_degree, plain->_poly._num_primes, plain->_poly._num_primes_p);
  poly_copy(&pt._poly, &plain->_poly);
  std::vector<cplx> out;
  decode(out, &pt);
  poly_free(&pt._poly);
  dcmplx* d = (dcmplx*)malloc(sizeof(dcmplx) * out.size());
  for (size_t i = 0; i < out.size(); ++i) d[i] = dcmplx(out[i].real(), out[i].imag());
  return d;
}

// ---- cipher_eval.h:80-105 ----
double* Get_msg(CIPHER ciph) {
  std::vector<cplx> out;
  decrypt_decode(out, ciph);
  return reals_of(out);
}
DCMPLX* Get_msg_with_imag(CIPHER ciph) {
  std::vector<cplx> out;
  decrypt_decode(out, ciph);
  dcmplx* d = (dcmplx*)malloc(sizeof(dcmplx) * out.size());
  for (size_t i = 0; i < out.size(); ++i) d[i] = dcmplx(out[i].real(), out[i].imag());
  return d;
}
void Print_cipher_info(FILE* fp, const char* name, CIPHER ciph) {
  fprintf(fp, "\n[%s] ciph_info: %d %d %ld %ld\n", name, (int)ciph->_sf_degree, (int)ciph->_slots, (long)ciph->_c0_poly._num_primes,
          (long)ciph->_c0_poly._num_primes_p);
}
void Print_cipher_msg(FILE* fp, const char* name, CIPHER ciph, uint32_t len) {
  std::vector<cplx> m;
  decrypt_decode(m, ciph);
  Print_cipher_info(fp, name, ciph);
  fprintf(fp, "[%s] msg: [ ", name);
  for (uint32_t i = 0; i < len && i < ciph->_slots; ++i) fprintf(fp, "%.17f ", m[i].real());
  fprintf(fp, "] ");
  print_msg_range(fp, m);
}
void Print_cipher_msg_with_imag(FILE* fp, const char* name, CIPHER ciph, uint32_t len) {
  std::vector<cplx> m;
  decrypt_decode(m, ciph);
  Print_cipher_info(fp, name, ciph);
  fprintf(fp, "[%s] msg: [ ", name);
  for (uint32_t i = 0; i < len && i < ciph->_slots; ++i) fprintf(fp, "(%.17f, %.17fI) ", m[i].real(), m[i].imag());
  fprintf(fp, "] ");
  print_msg_range(fp, m);
}
void Print_cipher_range(FILE* fp, const char* name, CIPHER ciph) {
  std::vector<cplx> m;
  decrypt_decode(m, ciph);
  fprintf(fp, "[%s] ", name);
  print_msg_range(fp, m);
}
// poly_eval.c:51-90: the first 8 coefficients (coefficient domain) of the first 3 q-limbs and p-limbs
void Print_poly_lite(FILE* fp, POLY input) {
  POLYNOMIAL p;
  memset(&p, 0, sizeof(p));
  poly_alloc(&p, input->_ring_degree, input->_num_primes, input->_num_primes_p);
  poly_copy(&p, input);
  if (input->_is_ntt) poly_ntt(&p, true);
  const size_t N = p._ring_degree, nq = p._num_primes < 3 ? p._num_primes : 3, np = p._num_primes_p < 3 ? p._num_primes_p : 3;
  const size_t n8 = N < 8 ? N : 8;
  int64_t v[8];
  sync();
  for (size_t i = 0; i < nq + np; ++i) {
    const bool is_p = i >= nq;
    const u64* limb = is_p ? p_limbs(&p) + (i - nq) * N : q_limbs(&p) + i * N;
    HIPCHK(acehip_download(ctx().hip, v, limb, n8 * 8, nullptr));
    fprintf(fp, "%c%ld: [", is_p ? 'P' : 'Q', (long)(is_p ? i - nq : i));
    for (size_t j = 0; j < n8; ++j) fprintf(fp, "%ld ", (long)v[j]);
    fprintf(fp, " ]\n");
  }
  poly_free(&p);
}
void Print_cipher_poly(FILE* fp, const char* name, CIPHER ciph) {
  Print_cipher_info(fp, name, ciph);
  fprintf(fp, "@c0:\n");
  Print_poly_lite(fp, &ciph->_c0_poly);
  fprintf(fp, "@c1:\n");
  Print_poly_lite(fp, &ciph->_c1_poly);
}
void Dump_cipher_msg(const char* name, CIPHER ciph, uint32_t len) { Print_cipher_msg(Get_trace_file(), name, ciph, len); }  // cipher_eval.c:260

// cipher_eval.c:264-290: the clear ReLU of the decrypted message, encrypted again at the ciphertext's level and scale degree
CIPHER Real_relu(CIPHER ciph) {
  std::vector<cplx> m;
  decrypt_decode(m, ciph);
  for (auto& v : m)
    if (v.real() <= 0) v *= 0.0;
  PLAINTEXT pt;
  memset(&pt, 0, sizeof(pt));
  encode_vector(&pt, m.data(), m.size(), (u32)ciph->_c0_poly._num_primes, ciph->_slots, ciph->_sf_degree, 0);
  CIPHER res = (CIPHER)calloc(1, sizeof(CIPHERTEXT));
  encrypt(res, &pt);
  poly_free(&pt._poly);
  return res;
}

// ckks_evaluator.c:347-379
CIPHER Upscale_ciph(CIPHER res, CIPHER ciph, uint32_t mod_size) {
  PLAINTEXT pt;
  memset(&pt, 0, sizeof(pt));
  encode_value_with_scale(&pt, 1.0, (u32)ciph->_c0_poly._num_primes, pow(2.0, (double)mod_size));
  Mul_plain(res, ciph, &pt);
  poly_free(&pt._poly);
  return res;
}
CIPHER Downscale_ciph(CIPHER res, CIPHER ciph, uint32_t waterline) {
  Context& c = ctx();
  RT_ASSERT(ciph->_c0_poly._num_primes > 1, "Downscale: multiply level is not big enought for more operation, try to use larger depth");
  const uint32_t sf_mod_size = (uint32_t)log2(c.sf);
  RT_ASSERT(waterline <= sf_mod_size, "Downscale: waterline should not larger than scaling factor");
  const uint32_t ciph_sf_mod_size = (uint32_t)log2(ciph->_scaling_factor);
  RT_ASSERT(ciph_sf_mod_size > sf_mod_size && ciph_sf_mod_size < waterline + sf_mod_size,
            "Downscale: waterline is set too low or the scale of input ciph is too high");
  Upscale_ciph(ciph, ciph, waterline + sf_mod_size - ciph_sf_mod_size);
  ciph->_sf_degree += 1;
  Rescale_ciph(res, ciph);
  return res;
}

// ---- cipher_valid.h ----
// cipher_valid.c:20-53: reports and goes on (the generated program decides what a failed validation means)
void Validate(CIPHER ciph, double* msg, uint32_t len, int32_t epsilon) {
  double* res = Get_msg(ciph);
  const double error = pow(10, epsilon);
  bool bad = false;
  uint32_t i;
  for (i = 0; i < len; ++i)
    if (fabs(res[i] - msg[i]) > error) {
      fprintf(stderr, "ERROR: validation failed at %d. %f != %f\n", (int)i, res[i], msg[i]);
      bad = true;
      break;
    }
  if (bad) {
    const int32_t start = i > 8 ? (int32_t)i - 8 : 0, end = start + 16 < (int32_t)len ? start + 16 : (int32_t)len;
    fprintf(stderr, "idx: ");
    for (int32_t j = start; j < end; ++j) fprintf(stderr, "%7d%c ", j, j == (int32_t)i ? '*' : ' ');
    fprintf(stderr, "\nres: ");
    for (int32_t j = start; j < end; ++j) fprintf(stderr, "%8.4f ", res[j]);
    fprintf(stderr, "\nstd: ");
    for (int32_t j = start; j < end; ++j) fprintf(stderr, "%8.4f ", msg[j]);
    fprintf(stderr, "\n");
  }
  free(res);
  fprintf(stdout, "%s: internal validation %s.\n", bad ? "ERROR" : "INFO", bad ? "fail" : "pass");
}
double* Add_plain_msg(CIPHER op0, PLAIN op1) {
  double *a = Get_msg(op0), *b = Get_msg_from_plain(op1);
  for (uint32_t i = 0; i < op0->_slots; ++i) a[i] += b[i];
  free(b);
  return a;
}
double* Add_msg(CIPHER op0, CIPHER op1, uint64_t len) {
  double *a = Get_msg(op0), *b = Get_msg(op1);
  double* r = add_impl(a, b, len);
  free(a);
  free(b);
  return r;
}
double* Add_ref(double* op0, double* op1, uint64_t len) { return add_impl(op0, op1, len); }
double* Mul_plain_msg(CIPHER op0, PLAIN op1) {
  double *a = Get_msg(op0), *b = Get_msg_from_plain(op1);
  for (uint32_t i = 0; i < op0->_slots; ++i) a[i] *= b[i];
  free(b);
  return a;
}
double* Mul_msg(CIPHER op0, CIPHER op1) {
  double *a = Get_msg(op0), *b = Get_msg(op1);
  for (uint32_t i = 0; i < op0->_slots; ++i) a[i] *= b[i];
  free(b);
  return a;
}
double* Rotate_msg(CIPHER op0, int32_t rotation) {  // slot i of the result is slot i + rotation (cyclically) of the message
  double* m = Get_msg(op0);
  const int32_t len = (int32_t)op0->_slots;
  double* r = (double*)malloc(sizeof(double) * len);
  for (int32_t i = 0; i < len; ++i) r[i] = m[(((i + rotation) % len) + len) % len];
  free(m);
  return r;
}
double* Relu_msg(CIPHER op0, uint64_t len) {
  fprintf(stderr, "INFO: validate %d relu.\n", ++g_relu_n);
  double* m = Get_msg(op0);
  double* r = relu_impl(m, len);
  free(m);
  return r;
}
double* Relu_rtv(CIPHER op0, uint64_t len) { return Relu_msg(op0, len); }
double* Relu_ref(double* op0, uint64_t len) {
  fprintf(stderr, "INFO: validate %d relu_e2e.\n", ++g_relu_e2e_n);
  return relu_impl(op0, len);
}
double* Bootstrap_msg(CIPHER op0) {
  fprintf(stderr, "INFO: validate %d bootstrap.\n", ++g_bts_n);
  return Get_msg(op0);
}
double* Conv_rtv(CIPHER op0, int n, int c, int h, int w, float* weight, int kn, int kc, int kh, int kw, float* bias, int bw, int sh, int sw, int pn,
                 int pc, int ph, int pw) {
  fprintf(stderr, "INFO: validate %d conv.\n", ++g_conv_n);
  RT_ASSERT((uint32_t)(n * c * h * w) <= op0->_slots, "input data too small");
  double* m = Get_msg(op0);
  double* r = conv_impl(m, n, c, h, w, weight, kn, kc, kh, kw, bias, bw, sh, sw, ph, pw);
  free(m);
  return r;
}
double* Conv_ref(double* op0, int n, int c, int h, int w, float* weight, int kn, int kc, int kh, int kw, float* bias, int bw, int sh, int sw, int pn,
                 int pc, int ph, int pw) {
  fprintf(stderr, "INFO: validate %d conv_e2e.\n", ++g_conv_e2e_n);
  return conv_impl(op0, n, c, h, w, weight, kn, kc, kh, kw, bias, bw, sh, sw, ph, pw);
}
double* Gemm_rtv(CIPHER op0, int h, int w, float* weight, int wh, int ww, float* bias, int bw) {
  fprintf(stderr, "INFO: validate %d gemm.\n", ++g_gemm_n);
  RT_ASSERT((uint32_t)(h * w) <= op0->_slots, "input data too small");
  double* m = Get_msg(op0);
  double* r = gemm_impl(m, h, w, weight, wh, ww, bias, bw);
  free(m);
  return r;
}
double* Gemm_ref(double* op0, int h, int w, float* weight, int wh, int ww, float* bias, int bw) {
  fprintf(stderr, "INFO: validate %d gemm_e2e.\n", ++g_gemm_e2e_n);
  return gemm_impl(op0, h, w, weight, wh, ww, bias, bw);
}
double* Average_pool_rtv(CIPHER op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw) {
  fprintf(stderr, "INFO: validate %d avg_pool.\n", ++g_pool_n);
  RT_ASSERT((uint32_t)(n * c * h * w) <= op0->_slots, "input data too small");
  double* m = Get_msg(op0);
  double* r = avg_pool_impl(m, n, c, h, w, kh, kw, sh, sw, ph, pw);
  free(m);
  return r;
}
double* Average_pool_ref(double* op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw) {
  fprintf(stderr, "INFO: validate %d avg_pool_e2e.\n", ++g_pool_e2e_n);
  return avg_pool_impl(op0, n, c, h, w, kh, kw, sh, sw, ph, pw);
}
double* Global_average_pool_rtv(CIPHER op0, int n, int c, int h, int w) {
  fprintf(stderr, "INFO: validate %d global_avg_pool.\n", ++g_gap_n);
  RT_ASSERT((uint32_t)(n * c * h * w) <= op0->_slots, "input data too small");
  double* m = Get_msg(op0);
  double* r = gap_impl(m, n, c, h, w);
  free(m);
  return r;
}
double* Global_average_pool_ref(double* op0, int n, int c, int h, int w) {
  fprintf(stderr, "INFO: validate %d global_avg_pool_e2e.\n", ++g_gap_e2e_n);
  return gap_impl(op0, n, c, h, w);
}
// (the reference validates max-pooling as the average pooling its compiler replaces it with, cipher_valid.c:403-413)
double* Max_pool_rtv(CIPHER op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw) {
  return Average_pool_rtv(op0, n, c, h, w, kh, kw, sh, sw, pn, pc, ph, pw);
}
double* Max_pool_ref(double* op0, int n, int c, int h, int w, int kh, int kw, int sh, int sw, int pn, int pc, int ph, int pw) {
  return Average_pool_ref(op0, n, c, h, w, kh, kw, sh, sw, pn, pc, ph, pw);
}

}  // extern "C"
