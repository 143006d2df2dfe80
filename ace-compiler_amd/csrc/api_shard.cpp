// api_shard.cpp -- packed-list phases of limb-sharded execution (include/acehip.h "Limb-sharded execution").
#include "api_internal.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------------
// Limb-sharded execution (BASELINE configs[4]): rank r of `world` holds the limbs gi with gi % world == r
// of every polynomial and key (q_i: gi = i, p_j: gi = L + j), packed in ascending gi.  NTT, limb-wise arithmetic and
// the key inner product are local; the two base conversions of a key-switch and the last limb of a rescale need the
// other ranks' limbs: the caller moves those with one all-gather / broadcast each (RCCL over xGMI on a node) between
// the phases below.  Every phase is a handful of batched launches on the caller's stream and never synchronises.
// The arithmetic is that of acehip_key_switch / acehip_rescale: gathering every rank's result reproduces them bit
// for bit (tests/test_gpu_shard.py runs world simulated ranks on one GPU).
// ------------------------------------------------------------------------------------------------
struct ShardLevelPlan {
  u32 nq = 0, nd = 0;
  ConvDesc* d_up = nullptr;             // [nd] ModUp onto the owned complement limbs of each digit
  ConvDesc* d_down = nullptr;           // [1]  ModDown onto the owned q-limbs
  std::vector<u32> n_tgt;               // per digit: converted limbs
  std::vector<std::vector<u32>> tgt_y;  // per digit: packed index (among the owned limbs at this level) of each target
  std::vector<u32*> d_tgt_gi;           // per digit: primes of the targets (NTT of the raised limbs)
  u32 max_tgt = 0;
  u64 *d_pinv = nullptr, *d_pinv_prec = nullptr;                      // [nq]  P^-1 mod q (ModDown tail)
  u64 *d_rs_c1 = nullptr, *d_rs_c1p = nullptr, *d_rs_inv = nullptr, *d_rs_invp = nullptr;  // rescale constants of the owned limbs < level-1
  u32 nq_rs = 0;
};
struct acehip_shard {
  acehip_ctx* c = nullptr;
  u32 rank = 0, world = 1;
  std::vector<u32> q_own, p_own;        // owned q indices i (ascending), owned p indices j
  u32 *d_q_gi = nullptr, *d_p_gi = nullptr, *d_own_gi_full = nullptr;  // device: gi of owned q-limbs / p-limbs / both
  u64 *full = nullptr, *ext = nullptr, *acc = nullptr, *pfull = nullptr, *conv = nullptr, *tmp = nullptr;
  std::map<u32, ShardLevelPlan> plans;
  std::mutex mu;
};

namespace {
using PtrTab = PackedPtrs;
u32 shard_nq(const acehip_shard* sh, u32 level) {
  u32 n = 0;
  while (n < sh->q_own.size() && sh->q_own[n] < level) ++n;
  return n;
}
u32 shard_pad(const acehip_shard* sh, u32 n_total) {  // most limbs any rank owns out of gi in [0, n_total)
  return (n_total + sh->world - 1) / sh->world;
}
// NTT of n packed limbs of primes h_gi[0..n) (device copy d_gi), n_polys polynomials `stride` words apart, optionally
// reading the input out of place (inverse, N = 2^16) from src0 / src1
void ntt_packed(acehip_ctx* c, u64* data, const u64* src0, const u64* src1, const u32* d_gi, const u32* h_gi, u32 n, u32 n_polys,
                size_t stride, bool inverse, hipStream_t s) {
  if (n == 0) return;
  const size_t N = c->hp.N;
  if (c->dc.logN == 16) {
    NttFuse f;
    f.gi_tab = d_gi;
    f.src0 = src0;
    f.src1 = src1;
    launch_ntt_fused(c->dc, data, 0, 0, n, inverse, s, 0, n_polys, stride, 0, f);
    return;
  }
  for (u32 z = 0; z < n_polys; ++z) {
    const u64* src = z ? src1 : src0;
    if (src) (void)hipMemcpyAsync(data + z * stride, src, (size_t)n * N * sizeof(u64), hipMemcpyDeviceToDevice, s);
    for (u32 k = 0; k < n; ++k) {
      const u32 gi = h_gi[k];
      u64* ptr = data + z * stride + (size_t)k * N;
      if (gi < c->hp.L) launch_ntt(c->dc, ptr, c->hp.L, gi, 1, inverse, s, gi);
      else              launch_ntt(c->dc, ptr, 0, gi - c->hp.L, 1, inverse, s, gi - c->hp.L);
    }
  }
}
const ShardLevelPlan* shard_plan(acehip_shard* sh, u32 level) {
  std::lock_guard<std::mutex> g(sh->mu);
  auto it = sh->plans.find(level);
  if (it != sh->plans.end()) return &it->second;
  acehip_ctx* c = sh->c;
  const HostParams& hp = c->hp;
  ShardLevelPlan pl;
  pl.nq = shard_nq(sh, level);
  pl.nd = hp.num_decomp(level);
  // owned limbs at this level, in packed order: (position, prime)
  std::vector<std::pair<u32, u32>> own;
  for (u32 k = 0; k < pl.nq; ++k) own.push_back({sh->q_own[k], sh->q_own[k]});
  for (u32 j : sh->p_own) own.push_back({level + j, hp.L + j});
  std::vector<ConvDesc> up(pl.nd);
  pl.n_tgt.resize(pl.nd);
  pl.tgt_y.resize(pl.nd);
  pl.d_tgt_gi.resize(pl.nd, nullptr);
  for (u32 d = 0; d < pl.nd; ++d) {
    const DevModUp* t = get_modup(c, level, d);
    if (!t) return nullptr;
    HostParams::ModUp hm = hp.modup(level, d);
    std::vector<u32> gi, col, pos;
    for (u32 y = 0; y < own.size(); ++y) {
      const u32 p = own[y].first, want = own[y].second;
      if (p >= hm.start && p < hm.start + hm.n2) continue;  // the digit's own limbs pass through
      u32 j = 0;
      while (j < hm.nc && hm.compl_idx[j] != want) ++j;
      if (j == hm.nc) return nullptr;
      gi.push_back(want);
      col.push_back(j);
      pos.push_back((u32)pl.tgt_y[d].size());
      pl.tgt_y[d].push_back(y);
    }
    pl.n_tgt[d] = (u32)gi.size();
    pl.max_tgt = std::max(pl.max_tgt, pl.n_tgt[d]);
    ConvDesc cd{};
    cd.hat = t->hat_mod;
    cd.scale = t->hat_inv;
    cd.scale_prec = t->hat_inv_prec;
    cd.src_gi = t->src_gi;
    cd.n_in = t->n2;
    cd.hat_ld = t->nc;
    cd.src_pos0 = hm.start;
    cd.n_out = pl.n_tgt[d];
    if (cd.n_out) {
      cd.out_gi = pl.d_tgt_gi[d] = c->up(gi);
      cd.col = c->up(col);
      cd.out_pos = c->up(pos);
      if (!cd.out_gi || !cd.col || !cd.out_pos) return nullptr;
    }
    up[d] = cd;
  }
  pl.d_up = c->up(up);
  if (pl.nq) {
    std::vector<u32> gi(pl.nq), pos(pl.nq);
    std::vector<u64> pinv(pl.nq), pinvp(pl.nq);
    for (u32 k = 0; k < pl.nq; ++k) {
      gi[k] = sh->q_own[k];
      pos[k] = k;
      pinv[k] = hp.pinv_modq[gi[k]];
      pinvp[k] = hp.pinv_modq_prec[gi[k]];
    }
    ConvDesc md{};
    md.hat = c->phat_modq_t;
    md.scale = c->phat_inv;
    md.scale_prec = c->phat_inv_prec;
    md.src_gi = c->p_gi;
    md.n_in = hp.K;
    md.hat_ld = hp.L;
    md.src_pos0 = 0;
    md.n_out = pl.nq;
    md.out_gi = c->up(gi);
    md.col = md.out_gi;  // column of phat_modq_t = the q index
    md.out_pos = c->up(pos);
    std::vector<ConvDesc> one(1, md);
    pl.d_down = c->up(one);
    pl.d_pinv = c->up(pinv);
    pl.d_pinv_prec = c->up(pinvp);
    if (!pl.d_down || !pl.d_pinv || !pl.d_pinv_prec) return nullptr;
  }
  if (level > 1) {  // rescale constants (crt.c:270-326) of the owned limbs below the last one
    const size_t row = (size_t)(level - 2) * hp.L;
    std::vector<u64> c1, c1p, inv, invp;
    for (u32 k = 0; k < pl.nq && sh->q_own[k] < level - 1; ++k) {
      const u32 i = sh->q_own[k];
      c1.push_back(hp.qlql[row + i]);
      c1p.push_back(hp.qlql_prec[row + i]);
      inv.push_back(hp.ql_inv[row + i]);
      invp.push_back(hp.ql_inv_prec[row + i]);
    }
    pl.nq_rs = (u32)c1.size();
    if (pl.nq_rs) {
      pl.d_rs_c1 = c->up(c1);
      pl.d_rs_c1p = c->up(c1p);
      pl.d_rs_inv = c->up(inv);
      pl.d_rs_invp = c->up(invp);
    }
  }
  return &(sh->plans[level] = pl);
}
}  // namespace

acehip_shard* acehip_shard_create(acehip_ctx* c, uint32_t rank, uint32_t world) {
  if (!c) {
    fail(ACEHIP_EINVAL, "null context");
    return nullptr;
  }
  if (world == 0 || rank >= world) {
    fail(ACEHIP_EINVAL, "acehip_shard_create: rank outside [0, world)");
    return nullptr;
  }
  const HostParams& hp = c->hp;
  auto* sh = new acehip_shard();
  sh->c = c;
  sh->rank = rank;
  sh->world = world;
  std::vector<u32> qg, pg, all;
  for (u32 i = 0; i < hp.L; ++i)
    if (i % world == rank) sh->q_own.push_back(i), qg.push_back(i), all.push_back(i);
  for (u32 j = 0; j < hp.K; ++j)
    if ((hp.L + j) % world == rank) sh->p_own.push_back(j), pg.push_back(hp.L + j), all.push_back(hp.L + j);
  if (!c->on_device) return sh;  // host-only context: ownership / exchange-layout queries only, every phase returns ENODEV
  sh->d_q_gi = c->up(qg);
  sh->d_p_gi = c->up(pg);
  sh->d_own_gi_full = c->up(all);
  const size_t N = hp.N, nown = std::max<size_t>(all.size(), 1);
  auto dev = [&](size_t limbs) {
    u64* p = nullptr;
    if (hipMalloc((void**)&p, std::max<size_t>(limbs, 1) * N * sizeof(u64)) != hipSuccess) return (u64*)nullptr;
    c->owned.push_back(p);
    return p;
  };
  sh->full = dev(hp.L);
  sh->ext = dev((size_t)hp.dnum * nown);
  sh->acc = dev(2 * nown);
  sh->pfull = dev(2ull * hp.K);
  sh->conv = dev(2 * nown);
  sh->tmp = dev(2 * nown);
  if (!sh->full || !sh->ext || !sh->acc || !sh->pfull || !sh->conv || !sh->tmp) {
    fail(ACEHIP_EHIP, "acehip_shard_create: device allocation failed");
    delete sh;
    return nullptr;
  }
  return sh;
}
void acehip_shard_destroy(acehip_shard* sh) { delete sh; }  // device memory belongs to the context
uint32_t acehip_shard_num_q(const acehip_shard* sh, uint32_t level) { return sh ? shard_nq(sh, level) : 0; }
uint32_t acehip_shard_num_p(const acehip_shard* sh) { return sh ? (u32)sh->p_own.size() : 0; }
uint32_t acehip_shard_pad_q(const acehip_shard* sh, uint32_t level) { return sh ? shard_pad(sh, level) : 0; }
uint32_t acehip_shard_pad_p(const acehip_shard* sh) {
  if (!sh) return 0;
  u32 m = 0;  // p-limbs start at gi = L: count per rank
  for (u32 r = 0; r < sh->world; ++r) {
    u32 n = 0;
    for (u32 j = 0; j < sh->c->hp.K; ++j) n += (sh->c->hp.L + j) % sh->world == r;
    m = std::max(m, n);
  }
  return m;
}
uint32_t acehip_shard_owned(const acehip_shard* sh, uint32_t level, uint32_t* q_out, uint32_t* p_out) {
  if (!sh) return 0;
  const u32 nq = shard_nq(sh, level);
  if (q_out) std::copy(sh->q_own.begin(), sh->q_own.begin() + nq, q_out);
  if (p_out) std::copy(sh->p_own.begin(), sh->p_own.end(), p_out);
  return nq;
}

// phase 1: the owned q-limbs of the key-switch input (NTT domain, packed) to the coefficient domain, written into the
// send buffer of the first exchange ([pad_q][N]; the caller all-gathers it rank-major)
int acehip_shard_ks_phase1(acehip_shard* sh, uint64_t* d_send, const uint64_t* d_x_own, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || !d_send || !d_x_own) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase1: bad arguments");
  const u32 nq = shard_nq(sh, level);
  ntt_packed(c, d_send, d_x_own, nullptr, sh->d_q_gi, sh->q_own.data(), nq, 1, 0, true, (hipStream_t)s_);
  stat(ST_NTT, 1, 16ull * c->hp.N * nq);
  return post_launch();
}

// phase 2: from the gathered coefficient-domain limbs ([world][pad_q][N], rank-major): ModUp of every digit onto the owned
// limbs, NTT, key inner product over the owned limbs (d_key_own: [dnum][2][n_own][N], this rank's limbs of the switch
// key in packed order: owned q-limbs of the full chain, then owned p-limbs), inverse NTT of the owned p-limbs of both
// accumulators into the send buffer of the second exchange ([2][pad_p][N])
int acehip_shard_ks_phase2(acehip_shard* sh, uint64_t* d_send2, const uint64_t* d_gath, const uint64_t* d_x_own, const uint64_t* d_key_own,
                           uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_send2 || !d_gath || !d_x_own || !d_key_own) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase2: bad arguments");
  if (hp.L > 256 || hp.dnum * (sh->q_own.size() + sh->p_own.size()) > 256) return fail(ACEHIP_EINVAL, "acehip_shard: pointer table too large");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 G = sh->world, pad_q = shard_pad(sh, level), nq = pl->nq, np = (u32)sh->p_own.size(), nown = nq + np;
  const u32 nq_full = (u32)sh->q_own.size(), nown_full = nq_full + np;
  // (a) every q-limb in position order: limb i is the (i / G)-th owned limb of rank i % G
  PtrTab gt{};
  for (u32 i = 0; i < level; ++i) gt.p[i] = d_gath + ((size_t)(i % G) * pad_q + i / G) * N;
  launch_packed_gather(c->dc, sh->full, gt, level, s);
  // (b) ModUp: all digits in one launch onto the owned complement limbs; ext digit d at ext + d*nown_full*N
  const size_t ext_stride = (size_t)nown_full * N;
  if (pl->max_tgt) launch_base_conv_batch(c->dc, sh->ext, ext_stride, sh->full, 0, pl->d_up, 1, pl->nd, pl->max_tgt, s, hp.alpha);
  // (c) NTT of the raised limbs
  for (u32 d = 0; d < pl->nd; ++d) {
    std::vector<u32> h_gi(pl->n_tgt[d]);
    for (u32 k = 0; k < pl->n_tgt[d]; ++k) {
      const u32 y = pl->tgt_y[d][k];
      h_gi[k] = y < nq ? sh->q_own[y] : hp.L + sh->p_own[y - nq];
    }
    ntt_packed(c, sh->ext + d * ext_stride, nullptr, nullptr, pl->d_tgt_gi[d], h_gi.data(), pl->n_tgt[d], 1, 0, false, s);
  }
  // (d) key inner product over the owned limbs; e_d[y] = the input's own limb (y in digit d) or the raised limb
  PtrTab qt{}, pt{};
  for (u32 d = 0; d < pl->nd; ++d) {
    const u32 start = hp.alpha * d, n2 = std::min(hp.alpha, level - start);
    std::vector<const u64*> of_y(nown, nullptr);
    for (u32 k = 0; k < pl->n_tgt[d]; ++k) of_y[pl->tgt_y[d][k]] = sh->ext + d * ext_stride + (size_t)k * N;
    for (u32 y = 0; y < nq; ++y)
      if (sh->q_own[y] >= start && sh->q_own[y] < start + n2) of_y[y] = d_x_own + (size_t)y * N;
    for (u32 y = 0; y < nq; ++y) qt.p[d * nq + y] = of_y[y];
    for (u32 y = 0; y < np; ++y) pt.p[d * np + y] = of_y[nq + y];
  }
  u64 *acc0 = sh->acc, *acc1 = sh->acc + (size_t)nown_full * N;
  const size_t key_stride = (size_t)nown_full * N;
  launch_packed_key_mac(c->dc, acc0, acc1, d_key_own, key_stride, qt, sh->d_q_gi, pl->nd, nq, s);
  launch_packed_key_mac(c->dc, acc0 + (size_t)nq * N, acc1 + (size_t)nq * N, d_key_own + (size_t)nq_full * N, key_stride, pt,
                        sh->d_p_gi, pl->nd, np, s);
  // (e) owned p-limbs of both accumulators to the coefficient domain, into the send buffer [2][pad_p][N]
  const u32 pad_p = acehip_shard_pad_p(sh);
  std::vector<u32> h_pgi(np);
  for (u32 k = 0; k < np; ++k) h_pgi[k] = hp.L + sh->p_own[k];
  ntt_packed(c, d_send2, acc0 + (size_t)nq * N, acc1 + (size_t)nq * N, sh->d_p_gi, h_pgi.data(), np, 2, (size_t)pad_p * N, true, s);
  stat(ST_KEYMAC, 1, 8ull * N * nown * (3ull * pl->nd + 2));
  return post_launch();
}

// phase 3: from the gathered coefficient-domain p-limbs of both accumulators ([world][2][pad_p][N]): conversion P -> owned
// q-limbs, NTT, out = (acc - conv) * P^-1 on the owned q-limbs (packed)
int acehip_shard_ks_phase3(acehip_shard* sh, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_gath2, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_out0 || !d_out1 || !d_gath2) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase3: bad arguments");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 G = sh->world, pad_p = acehip_shard_pad_p(sh), nq = pl->nq, nq_full = (u32)sh->q_own.size(), np = (u32)sh->p_own.size();
  const u32 nown_full = nq_full + np;
  if (nq == 0) return 0;
  // p-limb j of accumulator z: rank (L + j) % G, its k-th owned p-limb
  PtrTab gt{};
  for (u32 z = 0; z < 2; ++z)
    for (u32 j = 0; j < hp.K; ++j) {
      const u32 r = (hp.L + j) % G;
      u32 k = 0;
      for (u32 jj = 0; jj < j; ++jj) k += (hp.L + jj) % G == r;
      gt.p[z * hp.K + j] = d_gath2 + (((size_t)r * 2 + z) * pad_p + k) * N;
    }
  launch_packed_gather(c->dc, sh->pfull, gt, 2 * hp.K, s);
  // both accumulators: P -> owned q-limbs (problem z reads pfull + z*K*N, writes conv + z*nq_full*N)
  launch_base_conv_batch(c->dc, sh->conv, (size_t)nq_full * N, sh->pfull, (size_t)hp.K * N, pl->d_down, 0, 2, nq, s, hp.K);
  ntt_packed(c, sh->conv, nullptr, nullptr, sh->d_q_gi, sh->q_own.data(), nq, 2, (size_t)nq_full * N, false, s);
  u64 *acc0 = sh->acc, *acc1 = sh->acc + (size_t)nown_full * N;
  launch_packed_moddown_tail(c->dc, d_out0, d_out1, acc0, acc1, sh->conv, (size_t)nq_full * N, sh->d_q_gi, pl->d_pinv, pl->d_pinv_prec, nq, 2, s);
  stat(ST_MODDOWN, 2, 8ull * N * (2ull * hp.K + 2ull * nq));
  return post_launch();
}

// Rescale (Rescale_poly polynomial.c:1097-1163), exchange = broadcast of the last limb of c0 and c1 in the coefficient
// domain.  _send: on the rank that owns limb level-1, writes it for both polynomials into d_send ([2][N]) and returns 1;
// the other ranks return 0 and receive the broadcast.  _apply: every rank, owned limbs below level-1 (packed, in and out).
int acehip_shard_rescale_send(acehip_shard* sh, uint64_t* d_send, const uint64_t* d_c0_own, const uint64_t* d_c1_own, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L || !d_send || !d_c0_own || !d_c1_own) return fail(ACEHIP_EINVAL, "acehip_shard_rescale_send: bad arguments");
  if ((level - 1) % sh->world != sh->rank) return 0;
  const size_t N = c->hp.N;
  const u32 y = shard_nq(sh, level) - 1;  // the last owned limb is limb level-1
  const u32 gi = level - 1;
  ntt_packed(c, d_send, d_c0_own + (size_t)y * N, d_c1_own + (size_t)y * N, sh->d_q_gi + y, &gi, 1, 2, N, true, (hipStream_t)s_);
  if (int e = post_launch()) return e;
  return 1;
}
int acehip_shard_rescale_apply(acehip_shard* sh, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_c0_own, const uint64_t* d_c1_own,
                               const uint64_t* d_last, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L || !d_out0 || !d_out1 || !d_c0_own || !d_c1_own || !d_last) return fail(ACEHIP_EINVAL, "acehip_shard_rescale_apply: bad arguments");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = c->hp.N;
  const u32 n = pl->nq_rs, nq_full = (u32)sh->q_own.size();
  if (n == 0) return 0;
  u64* t = sh->tmp;  // [2][nq_full][N]
  launch_packed_rescale_spread(c->dc, t, (size_t)nq_full * N, d_last, N, sh->d_q_gi, level - 1, pl->d_rs_c1, pl->d_rs_c1p, n, 2, s);
  ntt_packed(c, t, nullptr, nullptr, sh->d_q_gi, sh->q_own.data(), n, 2, (size_t)nq_full * N, false, s);
  launch_packed_rescale_tail(c->dc, d_out0, d_out1, d_c0_own, d_c1_own, t, (size_t)nq_full * N, sh->d_q_gi, pl->d_rs_inv, pl->d_rs_invp, n, 2, s);
  stat(ST_RESCALE, 2, 8ull * N * (2ull * n + 1) * 2);
  return post_launch();
}

// Encode (SURVEY 8e collective 4): the integer message of a plaintext (rounded, scaled inverse embedding: N signed words) is
// computed once (acehip_encode_message, any rank) and broadcast; every rank then reduces it into its own limbs and
// transforms them (acehip_shard_encode_limbs).  Ranks may also each compute the message themselves: no exchange at all.
int acehip_encode_message(acehip_ctx* c, int64_t* d_msg, const void* d_vals, int kind, size_t len, uint32_t slots, double sf, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) || len > slots || !d_msg || (!d_vals && len))
    return fail(ACEHIP_EINVAL, "acehip_encode_message: bad arguments");
  if (int e = ensure_embed_tables(c)) return e;
  launch_embed_inv(d_msg, c->emb_work, d_vals, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, (hipStream_t)s);
  return post_launch();
}
int acehip_shard_encode_limbs(acehip_shard* sh, uint64_t* d_q_own, const int64_t* d_msg, double sf, uint32_t sf_degree, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || sf_degree < 1 || !d_q_own || !d_msg) return fail(ACEHIP_EINVAL, "acehip_shard_encode_limbs: bad arguments");
  hipStream_t s = (hipStream_t)s_;
  const u32 nq = shard_nq(sh, level);
  if (nq == 0) return 0;
  const size_t N = hp.N;
  const u64 sfi = (u64)sf;
  std::vector<u64> w(nq, 1);
  for (u32 k = 0; k < nq && sf_degree > 1; ++k) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1)
    const u64 q = hp.primes[sh->q_own[k]].q;
    u64 pw = sfi % q;
    for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
    w[k] = pw;
  }
  if (c->dc.logN == 16) {
    NttFuse f;
    f.gi_tab = sh->d_q_gi;
    f.msg = d_msg;
    if (sf_degree > 1) {
      u64* tab = sh->tmp;  // [nq] words at the head of tmp (stream ordered)
      HIP_TRY(hipMemcpyAsync(tab, w.data(), nq * sizeof(u64), hipMemcpyHostToDevice, s));
      f.msg_scale = tab;
    }
    launch_ntt_fused(c->dc, d_q_own, 0, 0, nq, false, s, 0, 1, 0, 0, f);
  } else {
    for (u32 k = 0; k < nq; ++k) {
      const u32 gi = sh->q_own[k];
      u64* ptr = d_q_own + (size_t)k * N;
      launch_values_to_rns(c->dc, ptr - (size_t)gi * N, d_msg, hp.L, gi, 1, s);
      if (sf_degree > 1) {
        LimbConsts lc{};
        lc.w[0] = w[k];
        launch_mul_scalars(c->dc, ptr - (size_t)gi * N, ptr - (size_t)gi * N, lc, hp.L, gi, 1, s);
      }
      launch_ntt(c->dc, ptr, hp.L, gi, 1, false, s, gi);
    }
  }
  stat(ST_ENCODE, 1, 8ull * N * (nq + 1));
  return post_launch();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Limb-sharded execution as a mode of the context (include/acehip.h): the exchange steps of the pipelines in api_ops.cpp.
// Simulated ranks (one process, replica h of the arena = hosted rank h): an exchange copies every listed limb from its
// owner's replica into the other replicas.  Real ranks (one process per GPU): grouped RCCL broadcasts from the owner, in place,
// on the launch stream.  RCCL is loaded with dlopen when first needed: unsharded programs never map it.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <thread>

struct Id128 {  // ncclUniqueId (rccl.h): passed by value to ncclCommInitRank
  char b[128];
};
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;  // optional: without it every exchange is grouped broadcasts
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // ACEHIP_RCCL_LIB: another library with RCCL's entry points (tests/c/mock_rccl.c lets the processes of a one-GPU test box act as
    // ranks); when set it is the only candidate, so a typo cannot silently fall back to the real library
    const char* forced = getenv("ACEHIP_RCCL_LIB");
    if (forced != nullptr && *forced) {
      api.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
    }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, Id128, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
    api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.GroupStart || !api.GroupEnd || !api.Broadcast) api.lib = nullptr;
  });
  return api.lib ? &api : nullptr;
}
struct RcclComm {
  void* comm = nullptr;
  u32 rank = 0, world = 1;
  // RCCL gets a stream of its own: this library is built with -fgpu-default-stream=per-thread, so the NULL stream its callers pass
  // means "the calling thread's stream" here and "the legacy stream" inside RCCL.  Every exchange is ordered after the launches
  // issued so far, and the launches that follow after the exchange, by events (no host synchronisation).
  hipStream_t xs = nullptr;
  hipEvent_t before = nullptr, after = nullptr;
  bool pending = false;  // an exchange was begun and not yet joined (shard_exchange_begin / _end)
  // ---- packed exchanges (round 6): ONE collective per exchange.  The limbs an exchange names are scattered over the polynomials'
  // reference layout (owner = gi % world), so every rank copies the limbs it owns into its slice of a staging block, one ncclAllGather
  // (one ncclBroadcast when a single rank owns them all) moves the block, and the limbs of the other ranks are copied from it to
  // where the pipelines read them.  The local copies run at HBM speed (0.1 us per limb); what crosses xGMI is one message per peer
  // instead of one per limb.  stage: [stage_words]; begin() takes regions from stage_used on, end() enqueues the pending copies out
  // of them on the launch stream and starts over.
  u64* stage = nullptr;
  size_t stage_words = 0, stage_used = 0;
  std::vector<std::pair<u64*, const u64*>> unpack;  // (destination limb, limb in the staging block) of the exchanges begun since the last end()
};
constexpr int kNcclUint64 = 5;  // ncclDataType_t (rccl.h)
}  // namespace

// limb copies between absolute addresses (no replica rebasing, no ownership filter), 112 per launch of the per-limb batch kernel
static void copy_limbs_abs(acehip_ctx* c, const std::vector<std::pair<u64*, const u64*>>& cps, hipStream_t s) {
  DevCtx dc = c->dc;
  dc.rep_span = 0;
  dc.rep0 = 0;
  dc.nrep = 1;
  dc.sh_world = 1;
  HwBatchArgs cp;
  u32 m = 0;
  for (size_t k = 0; k < cps.size(); ++k) {
    cp.seg_start[m] = (uint16_t)m;
    cp.op[m++] = HwBatchOp{cps[k].first, cps[k].second, nullptr, HW_OP_COPY, 0};
    if (m == HW_BATCH_MAX || k + 1 == cps.size()) {
      cp.seg_start[m] = (uint16_t)m;
      launch_hw_batch_ew(dc, cp, m, s);
      m = 0;
    }
  }
}
static bool packed_exchanges_on() {
  static const bool on = [] { const char* e = getenv("ACEHIP_SHARD_PACKED"); return !e || atoi(e) != 0; }();
  return on;
}
// One collective for the whole exchange (RcclComm: packed exchanges).  *done = false: not applicable here, use the grouped broadcasts.
static int shard_exchange_begin_packed(acehip_ctx* c, const XItem* items, size_t n, hipStream_t s, bool* done) {
  *done = false;
  RcclComm* rc = (RcclComm*)c->rccl;
  RcclApi* api = rccl_api();
  if (!packed_exchanges_on()) return ACEHIP_OK;
  const size_t N = c->hp.N;
  const u32 world = rc->world;
  const u64 lo = c->dc.rep_lo, span = c->dc.rep_span, stride = c->dc.rep_stride;
  // every (limb, replica) of the exchange with its owner, in the order all ranks enumerate alike
  std::vector<std::pair<u64*, u32>> limbs;
  std::vector<u32> cnt(world, 0);
  for (size_t k = 0; k < n; ++k) {
    const u64 a = (u64)items[k].ptr;
    const bool in_arena = a - lo < span && stride != 0;
    if (items[k].root >= world) return fail(ACEHIP_EINVAL, "shard_exchange: owner out of range");
    for (u32 r = c->sel0; r < c->sel0 + (in_arena ? c->seln : 1); ++r) {
      limbs.push_back({(u64*)(in_arena ? a + (u64)r * stride : a), items[k].root});
      cnt[items[k].root]++;
    }
  }
  if (limbs.size() < 2) return ACEHIP_OK;  // (a single limb is one broadcast either way, in place)
  u32 roots = 0, m = 0, single_root = 0;
  for (u32 r = 0; r < world; ++r) {
    if (cnt[r]) {
      roots++;
      single_root = r;
    }
    m = std::max(m, cnt[r]);
  }
  const bool gather = roots > 1;
  if (gather && api->AllGather == nullptr) return ACEHIP_OK;
  // slots per rank are padded to the largest owner's count: ownership is round robin, so the padding is at most one limb per rank
  const size_t need = (gather ? (size_t)world * m : (size_t)m) * N;
  if (rc->stage_used + need > rc->stage_words) {
    // (exchanges in flight use the block: this one goes limb by limb.  Decided by stage_used alone, which every rank advances alike:
    //  all ranks must take the same form of every exchange)
    if (rc->stage_used != 0) return ACEHIP_OK;
    HIP_TRY(hipStreamSynchronize(s));  // the block grows a few times at most while a program warms up
    HIP_TRY(hipStreamSynchronize(rc->xs));
    if (rc->stage) (void)hipFree(rc->stage);
    rc->stage = nullptr;
    rc->stage_words = 0;
    const size_t want = std::max(need * 2, (size_t)4 * (c->hp.L + c->hp.K) * N);
    HIP_TRY(hipMalloc((void**)&rc->stage, want * sizeof(u64)));
    rc->stage_words = want;
  }
  u64* blk = rc->stage + rc->stage_used;
  rc->stage_used += need;
  std::vector<std::pair<u64*, const u64*>> pack;
  std::vector<u32> slot(world, 0);
  for (const auto& lb : limbs) {
    const u32 root = lb.second;
    u64* st = blk + ((gather ? (size_t)root * m : 0) + slot[root]++) * N;
    if (root == rc->rank) pack.push_back({st, lb.first});
    else {
      rc->unpack.push_back({lb.first, st});
      c->xchg_bytes += N * 8;
    }
  }
  c->xchg_calls++;
  c->xchg_collectives++;
  copy_limbs_abs(c, pack, s);
  HIP_TRY(hipEventRecord(rc->before, s));
  HIP_TRY(hipStreamWaitEvent(rc->xs, rc->before, 0));
  int e;
  if (gather) e = api->AllGather(blk + (size_t)rc->rank * m * N, blk, (size_t)m * N, kNcclUint64, rc->comm, rc->xs);  // in place
  else        e = api->Broadcast(blk, blk, (size_t)m * N, kNcclUint64, (int)single_root, rc->comm, rc->xs);
  HIP_TRY(hipEventRecord(rc->after, rc->xs));
  rc->pending = true;
  *done = true;
  if (e) return fail(ACEHIP_EHIP, std::string(gather ? "RCCL all-gather: " : "RCCL broadcast: ") + (api->GetErrorString ? api->GetErrorString(e) : "error"));
  return ACEHIP_OK;
}

// An exchange in two halves, so that a pipeline can put independent launches between them: begin() orders the broadcasts after
// everything issued on s so far and enqueues them on the exchange stream, end() makes s wait for every exchange begun since the
// last end().  Between the two, launches on s run beside the broadcasts (the ModDown of a ciphertext: the inverse transform of c1's
// P-limbs while c0's travel).  Simulated ranks copy on s itself: begin() is the whole exchange, end() nothing.
int shard_exchange_begin(acehip_ctx* c, const XItem* items, size_t n, hipStream_t s) {
  if (c->sh_world <= 1 || n == 0) return ACEHIP_OK;
  if (c->rccl == nullptr) return shard_exchange(c, items, n, s);
  const size_t N = c->hp.N;
  const u64 lo = c->dc.rep_lo, span = c->dc.rep_span, stride = c->dc.rep_stride;
  {
    bool done = false;
    const int e = shard_exchange_begin_packed(c, items, n, s, &done);
    if (e || done) return e;
  }
  c->xchg_calls++;
  RcclComm* rc = (RcclComm*)c->rccl;
  RcclApi* api = rccl_api();
  HIP_TRY(hipEventRecord(rc->before, s));
  HIP_TRY(hipStreamWaitEvent(rc->xs, rc->before, 0));
  int e = api->GroupStart();
  for (size_t k = 0; k < n && e == 0; ++k) {
    const u64 a = (u64)items[k].ptr;
    const bool in_arena = a - lo < span && stride != 0;
    for (u32 r = c->sel0; r < c->sel0 + (in_arena ? c->seln : 1) && e == 0; ++r) {
      void* p = (void*)(in_arena ? a + (u64)r * stride : a);
      e = api->Broadcast(p, p, N, kNcclUint64, (int)items[k].root, rc->comm, rc->xs);
      c->xchg_collectives++;
      if (items[k].root != rc->rank) c->xchg_bytes += N * 8;
    }
  }
  const int e2 = api->GroupEnd();
  // (also after a failed group: whatever was enqueued on the exchange stream is ordered before the launches that follow end())
  HIP_TRY(hipEventRecord(rc->after, rc->xs));
  rc->pending = true;
  if (e || e2) return fail(ACEHIP_EHIP, std::string("RCCL broadcast: ") + (api->GetErrorString ? api->GetErrorString(e ? e : e2) : "error"));
  return ACEHIP_OK;
}
int shard_exchange_end(acehip_ctx* c, hipStream_t s) {
  if (c->sh_world <= 1 || c->rccl == nullptr) return ACEHIP_OK;
  RcclComm* rc = (RcclComm*)c->rccl;
  if (!rc->pending) return ACEHIP_OK;
  rc->pending = false;
  HIP_TRY(hipStreamWaitEvent(s, rc->after, 0));  // (the exchange stream is in order: the last recorded event covers every begin())
  if (!rc->unpack.empty()) {  // packed exchanges: the other ranks' limbs go from the staging block to where the pipelines read them
    copy_limbs_abs(c, rc->unpack, s);
    rc->unpack.clear();
  }
  rc->stage_used = 0;  // (the next begin() packs on s, behind these copies; its collective waits for that)
  return ACEHIP_OK;
}

int shard_exchange(acehip_ctx* c, const XItem* items, size_t n, hipStream_t s) {
  if (c->sh_world <= 1 || n == 0) return ACEHIP_OK;
  if (c->rccl != nullptr) {  // one process per rank: broadcast from the owner, in place, for every selected replica
    const int e = shard_exchange_begin(c, items, n, s);
    const int e2 = shard_exchange_end(c, s);
    return e ? e : e2;
  }
  const size_t N = c->hp.N;
  const u64 lo = c->dc.rep_lo, span = c->dc.rep_span, stride = c->dc.rep_stride;
  c->xchg_calls++;
  // simulated ranks: owner's replica -> every other hosted replica (absolute addresses: a DevCtx without rebasing)
  DevCtx dc = c->dc;
  dc.rep_span = 0;
  dc.rep0 = 0;
  dc.nrep = 1;
  dc.sh_world = 1;
  HwBatchArgs cp;
  u32 m = 0;
  auto flush = [&] {
    if (m == 0) return;
    cp.seg_start[m] = (uint16_t)m;
    launch_hw_batch_ew(dc, cp, m, s);
    m = 0;
  };
  for (size_t k = 0; k < n; ++k) {
    const u64 a = (u64)items[k].ptr;
    if (!(a - lo < span) || stride == 0) continue;  // memory all simulated ranks share
    u32 h_root = UINT32_MAX;
    for (u32 h = 0; h < c->sh_hosted.size(); ++h)
      if (c->sh_hosted[h] == items[k].root) h_root = h;
    if (h_root == UINT32_MAX) return fail(ACEHIP_EINVAL, "shard_exchange: the owner of a limb is not hosted here");
    for (u32 h = 0; h < c->sh_hosted.size(); ++h) {
      if (h == h_root) continue;
      if (m == HW_BATCH_MAX) flush();
      cp.seg_start[m] = (uint16_t)m;
      cp.op[m++] = HwBatchOp{(u64*)(a + (u64)h * stride), (const u64*)(a + (u64)h_root * stride), nullptr, HW_OP_COPY, 0};
      c->xchg_bytes += N * 8;
    }
  }
  flush();
  return post_launch();
}

// called by acehip_ctx_destroy: the communicator, its stream and events go with the context
void shard_release(acehip_ctx* c) {
  if (!c || !c->rccl) return;
  RcclComm* rc = (RcclComm*)c->rccl;
  RcclApi* api = rccl_api();
  if (rc->xs) (void)hipStreamSynchronize(rc->xs);
  if (rc->comm && api && api->CommDestroy) (void)api->CommDestroy(rc->comm);
  if (rc->before) (void)hipEventDestroy(rc->before);
  if (rc->after) (void)hipEventDestroy(rc->after);
  if (rc->xs) (void)hipStreamDestroy(rc->xs);
  if (rc->stage) (void)hipFree(rc->stage);
  delete rc;
  c->rccl = nullptr;
}

extern "C" {

int acehip_ctx_shard_sim(acehip_ctx* c, uint32_t world) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if (world == 0 || world > 16) return fail(ACEHIP_EINVAL, "acehip_ctx_shard_sim: 1..16 simulated ranks");
  if (world == 1) {
    c->sh_world = 1;
    c->sh_hosted.clear();
    return ACEHIP_OK;
  }
  if (c->on_device && (c->n_replicas < world || !c->ws_external || !c->scratch_external))
    return fail(ACEHIP_EINVAL, "acehip_ctx_shard_sim: needs an arena with one replica per rank, workspace and hw scratch inside (acehip_ctx_set_arena)");
  c->sh_world = world;
  c->sh_hosted.resize(world);
  for (u32 r = 0; r < world; ++r) c->sh_hosted[r] = r;
  c->sel0 = 0;
  c->seln = 1;
  return ACEHIP_OK;
}

int acehip_rccl_unique_id(void* out, size_t cap) {
  if (!out || cap < 128) return fail(ACEHIP_EINVAL, "acehip_rccl_unique_id: 128 bytes needed");
  RcclApi* api = rccl_api();
  if (!api) return fail(ACEHIP_ENODEV, "librccl.so could not be loaded");
  Id128 id;
  if (int e = api->GetUniqueId(&id)) return fail(ACEHIP_EHIP, std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(e) : "error"));
  std::memcpy(out, id.b, 128);
  return 128;
}

int acehip_ctx_shard_rccl(acehip_ctx* c, uint32_t rank, uint32_t world, const void* unique_id, size_t id_bytes) {
  if (int e = check_dev(c)) return e;
  if (world == 0 || rank >= world || !unique_id || id_bytes != 128) return fail(ACEHIP_EINVAL, "acehip_ctx_shard_rccl: bad arguments");
  if (c->sh_world > 1) return fail(ACEHIP_EINVAL, "acehip_ctx_shard_rccl: the context is sharded already");
  RcclApi* api = rccl_api();
  if (!api) return fail(ACEHIP_ENODEV, "librccl.so could not be loaded");
  auto* rc = new RcclComm();
  rc->rank = rank;
  rc->world = world;
  Id128 id;
  std::memcpy(id.b, unique_id, 128);
  // ncclCommInitRank returns when EVERY rank has joined and cannot be cancelled: a rank that read a stale id, or whose peers died,
  // would wait for ever.  It runs on a helper thread and the caller gives up after ACEHIP_RCCL_INIT_TIMEOUT_S (default 180 s) with
  // an error (the shim aborts the process on it); the helper is left behind in that case.
  struct Join {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int err = 0;
    void* comm = nullptr;
  };
  auto join = std::make_shared<Join>();
  const int dev = c->device;
  std::thread([join, api, id, world, rank, dev] {
    (void)hipSetDevice(dev);
    void* comm = nullptr;
    const int e = api->CommInitRank(&comm, (int)world, id, (int)rank);
    std::lock_guard<std::mutex> lk(join->mu);
    join->err = e;
    join->comm = comm;
    join->done = true;
    join->cv.notify_all();
  }).detach();
  {
    const char* t = getenv("ACEHIP_RCCL_INIT_TIMEOUT_S");
    const int limit_s = t && atoi(t) > 0 ? atoi(t) : 180;
    std::unique_lock<std::mutex> lk(join->mu);
    if (!join->cv.wait_for(lk, std::chrono::seconds(limit_s), [&] { return join->done; })) {
      delete rc;
      return fail(ACEHIP_EHIP, "ncclCommInitRank: rank " + std::to_string(rank) + " of " + std::to_string(world) + " did not join within " +
                                   std::to_string(limit_s) + " s (stale communicator id, or a peer that never started?)");
    }
    if (join->err) {
      delete rc;
      return fail(ACEHIP_EHIP, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(join->err) : "error"));
    }
    rc->comm = join->comm;
  }
  if (hipStreamCreateWithFlags(&rc->xs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&rc->before, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&rc->after, hipEventDisableTiming) != hipSuccess) {
    c->rccl = rc;
    shard_release(c);  // (communicator, and whichever of the stream / events exist)
    return fail(ACEHIP_EHIP, "acehip_ctx_shard_rccl: stream / event creation failed");
  }
  c->rccl = rc;
  if (world > 1) {
    c->sh_world = world;
    c->sh_hosted.assign(1, rank);
  }
  return ACEHIP_OK;
}

int acehip_shard_gather(acehip_ctx* c, uint64_t* d_poly, uint32_t level, uint32_t pos0, uint32_t n_limbs, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n_limbs)) return e;
  if (c->sh_world <= 1) return ACEHIP_OK;
  std::vector<XItem> x;
  for (u32 p = pos0; p < pos0 + n_limbs; ++p) x.push_back(XItem{d_poly + (size_t)p * c->hp.N, limb_prime(p, level, c->hp.L) % c->sh_world});
  return shard_exchange(c, x.data(), x.size(), (hipStream_t)s);
}

uint32_t acehip_shard_world(const acehip_ctx* c) { return c ? c->sh_world : 1; }
uint32_t acehip_shard_rank(const acehip_ctx* c) { return c && c->sh_world > 1 ? c->sh_hosted[0] : 0; }
uint32_t acehip_shard_owned_limbs(const acehip_ctx* c, uint32_t rank) {
  if (!c) return 0;
  u32 n = 0;
  for (u32 gi = 0; gi < c->hp.L + c->hp.K; ++gi) n += gi % c->sh_world == rank;
  return n;
}
uint64_t acehip_shard_traffic(const acehip_ctx* c_, uint64_t* steps, int reset) {
  acehip_ctx* c = const_cast<acehip_ctx*>(c_);
  if (!c) return 0;
  const u64 b = c->xchg_bytes;
  if (steps) {
    steps[0] = c->xchg_calls;
    steps[1] = b / (8ull * c->hp.N);
  }
  if (reset) c->xchg_bytes = c->xchg_calls = c->xchg_collectives = 0;
  return b;
}
uint64_t acehip_shard_collectives(const acehip_ctx* c) { return c ? c->xchg_collectives : 0; }

// which limbs meet where: the exchange steps of the pipelines in api_ops.cpp, stated once more as data so that the schedule
// can be checked against the CPU oracle without a GPU (tests/test_dist_gloo.py runs it over gloo with two processes)
int acehip_shard_schedule(const acehip_ctx* c, uint32_t world, int op, uint32_t level, uint32_t* out_step, uint32_t* out_pos,
                          uint32_t* out_root, size_t cap) {
  if (!c || world == 0 || level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_shard_schedule: bad arguments");
  const HostParams& hp = c->hp;
  size_t n = 0;
  auto put = [&](u32 step, u32 pos, u32 gi) {
    if (n < cap) {
      out_step[n] = step;
      out_pos[n] = pos;
      out_root[n] = gi % world;
    }
    ++n;
  };
  switch (op) {
    case 0:  // ModUp of all digits: the coefficient-domain q-limbs (api_ops.cpp modup_digits_to / acehip_key_switch step 1)
      for (u32 i = 0; i < level; ++i) put(0, i, i);
      break;
    case 1:  // ModDown: the coefficient-domain P-limbs at positions level.. (do_mod_down_n / acehip_key_switch step 5)
      for (u32 j = 0; j < hp.K; ++j) put(0, level + j, hp.L + j);
      break;
    case 2:  // Rescale: the last limb (do_rescale)
      if (level < 2) return fail(ACEHIP_EINVAL, "acehip_shard_schedule: rescale needs two limbs");
      put(0, level - 1, level - 1);
      break;
    case 3:  // ModRaise: limb 0 (acehip_mod_raise)
      put(0, 0, 0);
      break;
    default:
      return fail(ACEHIP_EINVAL, "acehip_shard_schedule: unknown operation");
  }
  return (int)n;
}

}  // extern "C"
