// api_ops.cpp -- C ABI (include/acehip.h): the launch entry points and the host-side sequencing of Decomp_modup / Mod_down /
// Rescale / key-switch / encode over the HIP kernels.
#include "api_internal.hpp"

// The base conversion can ride in the first pass of the following forward NTT (N = 2^16: ntt_fast.hip SRC_CONV*): the converted
// limbs are then never written / re-read in coefficient form (-104 MB of HBM traffic per C3 key-switch), but every output limb
// re-reads its alpha sources through L2 and the pass becomes VALU-bound.  Measured (MI355X, round 2): key-switch 0.256 ms
// either way, ResNet-20 1.51 images/s fused vs 1.56 unfused -- so it is OFF unless ACEHIP_CONV_FUSION=1 (bit-exact both ways,
// tests/test_gpu_parity.py::test_conv_fusion_matches).
// The base conversion computed by the first pass of the forward NTT while it loads its input (ntt_fast.hip SRC_CONV*): every output limb's
// workgroups read all n_in source tiles again (through L2), the converted limbs are never written in coefficient form.
//   ACEHIP_CONV_FUSION=1: wherever it is possible (measured slower on the whole image: the re-reads and the multiply-adds of 12 sources per output)
//   ACEHIP_CONV_FUSION_UP=n: ModUp conversions with at most n source limbs (the low levels of a network: few sources, K outputs)
//   ACEHIP_CONV_FUSION_DOWN=n: ModDown conversions with at most n output limbs (K sources each)
static u32 env_u32(const char* name, u32 dflt) {
  const char* e = getenv(name);
  return e ? (u32)strtoul(e, nullptr, 0) : dflt;
}
bool conv_fusable(const acehip_ctx* c, u32 n_in) {
  static const bool on = [] { const char* e = getenv("ACEHIP_CONV_FUSION"); return e && *e == '1'; }();
  return on && c->dc.logN == 16 && c->dc.split_bits <= 30 && n_in <= 12 && c->sh_world <= 1;
}
static bool conv_fusable_up(const acehip_ctx* c, u32 level) {  // every digit's conversion of a ModUp at `level`
  static const u32 cap = env_u32("ACEHIP_CONV_FUSION_UP", 0);
  const u32 n_in = std::min(c->hp.alpha, level);
  if (conv_fusable(c, c->hp.alpha)) return true;
  return n_in <= cap && c->dc.logN == 16 && c->dc.split_bits <= 30 && n_in <= 12 && c->sh_world <= 1;
}
static bool conv_fusable_down(const acehip_ctx* c, u32 level) {
  static const u32 cap = env_u32("ACEHIP_CONV_FUSION_DOWN", 0);
  if (conv_fusable(c, c->hp.K)) return true;
  return level <= cap && c->dc.logN == 16 && c->dc.split_bits <= 30 && c->hp.K <= 12 && c->sh_world <= 1;
}

static int do_mod_down_n(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s);

static int do_decomp_modup(acehip_ctx* c, u64* out, const u64* in, u32 level, u32 digit, u64* scratch, hipStream_t s) {
  const HostParams& hp = c->hp;
  const DevModUp* t = get_modup(c, level, digit);
  if (!t) return fail(ACEHIP_EHIP, "ModUp table upload failed");
  const size_t N = hp.N;
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs) {
    // digit limbs pass through unchanged (polynomial.c:1265-1273)
    copy_limbs_dc(dc, out + t->start * N, in + t->start * N, t->n2, t->start, s);
    // iNTT of the digit limbs in scratch, scaled by (Q_d/q_i)^-1 mod q_i (polynomial.c:1276-1301)
    copy_limbs_dc(dc, scratch, in + t->start * N, t->n2, t->start, s);
    // scratch limb i has prime start+i: run the iNTT as "positions [start, start+n2) of a level-L poly"
    launch_ntt(dc, scratch, hp.L, t->start, t->n2, true, s, t->start);
    launch_mul_const(dc, scratch, scratch, t->hat_inv, t->hat_inv_prec, t->src_gi, t->n2, s);
  }
  if (sharded(c)) {  // the conversion needs every source limb of the digit: each comes from its owner
    std::vector<XItem> x;
    for (u32 i = 0; i < t->n2; ++i) x.push_back(XItem{scratch + (size_t)i * N, (t->start + i) % c->sh_world});
    if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
  }
  for (const DevCtx& dc : dcs) {
    // exact 128-bit sums + reduction into the complement limbs (polynomial.c:1302-1320)
    launch_base_conv(dc, out, scratch, t->hat_mod, t->out_gi, t->out_pos, t->n2, t->nc, t->nc, s);
    // NTT of the complement limbs (polynomial.c:1322-1329)
    launch_ntt(dc, out, level, 0, t->start, false, s);
    launch_ntt(dc, out, level, t->start + t->n2, level + hp.K - (t->start + t->n2), false, s);
  }
  return post_launch();
}

static int do_mod_down(acehip_ctx* c, u64* out, const u64* in, u32 level, u64* scratch, hipStream_t s) {
  const HostParams& hp = c->hp;
  const size_t N = hp.N;
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs) {
    // P part -> coefficient domain, times (P/p_j)^-1 mod p_j  (polynomial.c:941-945, 779-790)
    copy_limbs_dc(dc, scratch, in + level * N, hp.K, hp.L, s);
    launch_ntt(dc, scratch, 0, 0, hp.K, true, s);  // level 0: position j -> prime p_j
    launch_mul_const(dc, scratch, scratch, c->phat_inv, c->phat_inv_prec, c->p_gi, hp.K, s);
  }
  if (sharded(c)) {
    std::vector<XItem> x;
    for (u32 j = 0; j < hp.K; ++j) x.push_back(XItem{scratch + (size_t)j * N, (hp.L + j) % c->sh_world});
    if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
  }
  for (const DevCtx& dc : dcs) {
    // conv P -> Q (polynomial.c:791-803); phat_modq_t is [K][L]: use its first `level` columns via n_out = L stride
    launch_base_conv(dc, out, scratch, c->phat_modq_t, c->q_gi, c->q_pos, hp.K, level, hp.L, s);
    launch_ntt(dc, out, level, 0, level, false, s);
    launch_moddown_tail(dc, out, in, c->pinv, c->pinv_prec, level, s);
  }
  return post_launch();
}

extern "C" {

int acehip_ntt_forward(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (int e = for_replica_chunks(c, [&] {
        for (const DevCtx& dc : launch_dcs(c)) launch_ntt(dc, d, level, pos0, n, false, (hipStream_t)s);
        return 0;
      }))
    return e;
  stat(ST_NTT, n, 16ull * c->hp.N * n);
  return post_launch();
}
int acehip_ntt_inverse(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (int e = for_replica_chunks(c, [&] {
        for (const DevCtx& dc : launch_dcs(c)) launch_ntt(dc, d, level, pos0, n, true, (hipStream_t)s);
        return 0;
      }))
    return e;
  stat(ST_NTT, n, 16ull * c->hp.N * n);
  return post_launch();
}

int acehip_ntt_batch(acehip_ctx* c, uint64_t* d, size_t poly_stride, uint32_t n_polys, uint32_t level, uint32_t pos0,
                     uint32_t n, int inverse, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n_polys == 0) return ACEHIP_OK;
  if (n_polys > 65535) return fail(ACEHIP_EINVAL, "acehip_ntt_batch: at most 65535 polynomials per launch");
  if (int e = for_replica_chunks(c, [&] {
        for (const DevCtx& dc : launch_dcs(c)) launch_ntt(dc, d, level, pos0, n, inverse != 0, (hipStream_t)s, 0, n_polys, poly_stride);
        return 0;
      }))
    return e;
  stat(ST_NTT, (u64)n * n_polys, 16ull * c->hp.N * n * n_polys);
  return post_launch();
}

static int ew(acehip_ctx* c, EwOp op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n, acehip_stream s) {
  if (c) stat(ST_EW, n, (op == EwOp::MulAdd ? 32ull : 24ull) * c->hp.N * n);
  if (c && (op == EwOp::Mul || op == EwOp::MulAdd)) {
    stat(ST_EW_MUL, n, 24ull * c->hp.N * n);
    acehip_stat_slots()[ST_EW_MUL].calls--;
  }
  if (int e = check_range(c, level, pos0, n)) return e;
  for (const DevCtx& dc : launch_dcs(c)) launch_ew(dc, op, r, a, b, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
int acehip_modadd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Add, r, a, b, level, pos0, n, s); }
int acehip_modsub(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Sub, r, a, b, level, pos0, n, s); }
int acehip_modmul(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Mul, r, a, b, level, pos0, n, s); }
int acehip_modmuladd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::MulAdd, r, a, b, level, pos0, n, s); }

int acehip_rotate(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint32_t* perm, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (r == a) return fail(ACEHIP_EINVAL, "acehip_rotate: in-place rotation is not supported");
  for (const DevCtx& dc : launch_dcs(c)) launch_rotate(dc, r, a, perm, level, pos0, n, (hipStream_t)s);
  return post_launch();
}

int acehip_rotate_add2(acehip_ctx* c, uint64_t* r0, uint64_t* r1, const uint64_t* acc0, const uint64_t* acc1, const uint64_t* a0,
                       const uint64_t* a1, uint32_t auto_k, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (!r0 || !acc0 || !a0 || (r1 && (!acc1 || !a1))) return fail(ACEHIP_EINVAL, "acehip_rotate_add2: null operand");
  if (auto_k % 2 == 0 || auto_k >= 2 * c->hp.N) return fail(ACEHIP_EINVAL, "acehip_rotate_add2: automorphism index must be odd and below 2N");
  if (r0 == a0 || r0 == a1 || (r1 && (r1 == a0 || r1 == a1)))
    return fail(ACEHIP_EINVAL, "acehip_rotate_add2: the rotated operand must not alias a result");
  for (const DevCtx& dc : launch_dcs(c)) launch_rotate_add2(dc, r0, r1, acc0, acc1, a0, a1, auto_k, level, pos0, n, (hipStream_t)s);
  stat(ST_ROTATE, (r1 ? 2u : 1u) * n, (r1 ? 2ull : 1ull) * n * 24ull * c->hp.N);
  return post_launch();
}

// single-limb forms: the limb pointers are used directly; prime_gi selects the modulus
static int hw(acehip_ctx* c, EwOp op, u64* r, const u64* a, const u64* b, u32 gi, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 L = c->hp.L;
  if (gi >= L + c->hp.K) return fail(ACEHIP_EINVAL, "prime index out of range");
  for (const DevCtx& dc : launch_dcs(c)) {
    if (gi < L) launch_ew(dc, op, r, a, b, L, gi, 1, (hipStream_t)s, gi);
    else launch_ew(dc, op, r, a, b, 0, gi - L, 1, (hipStream_t)s, gi - L);
  }
  return post_launch();
}
int acehip_hw_modadd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t gi, acehip_stream s) { return hw(c, EwOp::Add, r, a, b, gi, s); }
int acehip_hw_modmul(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t gi, acehip_stream s) { return hw(c, EwOp::Mul, r, a, b, gi, s); }
int acehip_hw_rotate(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint32_t* perm, uint32_t gi, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (gi >= c->hp.L + c->hp.K) return fail(ACEHIP_EINVAL, "prime index out of range");
  if (r == a) return fail(ACEHIP_EINVAL, "acehip_hw_rotate: in-place rotation is not supported");
  for (DevCtx dc : launch_dcs(c)) {  // the limb pointers are used directly: ownership is decided here, by the prime index
    if (!dc_owns(dc, gi)) continue;
    dc.sh_world = 1;
    launch_rotate(dc, r, a, perm, 0, 0, 1, (hipStream_t)s);
  }
  return post_launch();
}


int acehip_decomp_modup(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_decomp_modup: bad level/digit");
  (void)hipSetDevice(c->device);
  stat(ST_MODUP, 1, 8ull * c->hp.N * (std::min(c->hp.alpha, level - c->hp.alpha * digit) + level + c->hp.K));
  return do_decomp_modup(c, out, in, level, digit, ws_at(c, 0), (hipStream_t)s);
}

int acehip_mod_down(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_mod_down: bad level");
  if (out == in) return fail(ACEHIP_EINVAL, "acehip_mod_down: out must not alias in");
  if (c->dc.logN == 16) return for_replica_chunks(c, [&] { return do_mod_down_n(c, out, nullptr, in, nullptr, level, (hipStream_t)s); });
  stat(ST_MODDOWN, 1, 8ull * c->hp.N * (2 * level + c->hp.K));
  dbg_touch(out, (size_t)level * c->hp.N);
  dbg_touch(in, (size_t)(level + c->hp.K) * c->hp.N);
  return do_mod_down(c, out, in, level, ws_at(c, 0), (hipStream_t)s);
}

// Mod_down of one or two extended polynomials (the two accumulators of a key-switch) in the same launches
static int do_mod_down_n(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s) {
  const HostParams& hp = c->hp;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const u32 np = in1 ? 2 : 1;
  const size_t N = hp.N, PK = (size_t)hp.K * N, QL = (size_t)level * N;
  dbg_touch(out0, QL);
  dbg_touch(out1, QL);
  dbg_touch(in0, QL + PK);
  dbg_touch(in1, QL + PK);
  u64* pc = c->ws;            // [2][K][N] p-limbs in the coefficient domain
  u64* tmp = pc + 2 * PK;     // [2][level][N]
  const DcList dcs = launch_dcs(c);
  auto p_limbs_of = [&](u32 z) {
    std::vector<XItem> x;
    for (u32 j = 0; j < hp.K; ++j) x.push_back(XItem{pc + z * PK + (size_t)j * N, (hp.L + j) % c->sh_world});
    return x;
  };
  if (sharded(c) && c->rccl != nullptr && np == 2 && hp.logN == 16) {
    // ranks on separate GPUs: the P-limbs of c0 travel while the inverse transform of c1's runs (SURVEY 8e collective 2, overlapped)
    for (u32 z = 0; z < 2; ++z) {
      for (const DevCtx& dc : dcs) {
        NttFuse fi;
        fi.src0 = (z ? in1 : in0) + QL;
        fi.inv_scale = plan->inv_down;
        launch_ntt_fused(dc, pc + z * PK, 0, 0, hp.K, true, s, 0, 1, PK, 0, fi);
      }
      const std::vector<XItem> x = p_limbs_of(z);
      if (int e = shard_exchange_begin(c, x.data(), x.size(), s)) {
        // broadcasts of z = 0 may already be queued on the exchange stream and write into the workspace: order s behind them
        // before anything else reuses it (the error is what the caller sees; the hand-back must still happen)
        (void)shard_exchange_end(c, s);
        return e;
      }
    }
    if (int e = shard_exchange_end(c, s)) return e;
  } else {
    for (const DevCtx& dc : dcs) {
      if (dc.logN == 16) {
        NttFuse fi;
        fi.src0 = in0 + QL;
        fi.src1 = in1 ? in1 + QL : nullptr;
        fi.inv_scale = plan->inv_down;  // (P/p_j)^-1 folded into the last inverse stage
        launch_ntt_fused(dc, pc, 0, 0, hp.K, true, s, 0, np, PK, 0, fi);  // level 0: position j -> prime p_j
      } else {
        copy_limbs_dc(dc, pc, in0 + QL, hp.K, hp.L, s);
        if (in1) copy_limbs_dc(dc, pc + PK, in1 + QL, hp.K, hp.L, s);
        launch_ntt(dc, pc, 0, 0, hp.K, true, s, 0, np, PK);
      }
    }
    if (sharded(c)) {  // the conversion P -> Q needs every P-limb: each comes from its owner (SURVEY 8e collective 2)
      std::vector<XItem> x;
      for (u32 z = 0; z < np; ++z) {
        const std::vector<XItem> xz = p_limbs_of(z);
        x.insert(x.end(), xz.begin(), xz.end());
      }
      if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
    }
  }
  // (descriptor nd + 1: the ModDown problem with its K sources at limb positions 0.. of `pc`)
  const bool conv_in_ntt = conv_fusable_down(c, level);
  for (const DevCtx& dc : dcs) {
    if (!conv_in_ntt) launch_base_conv_batch(dc, tmp, QL, pc, PK, plan->d_descs + plan->nd + 1, 0, np, level, s, hp.K, PtrTab8{}, plan->mfma_down);
    if (dc.logN == 16) {
      NttFuse fo;
      if (conv_in_ntt) {  // the conversion P -> Q rides in the first pass of the NTT
        fo.conv = plan->d_descs + plan->nd + 1;
        fo.conv_step = 0;
        fo.conv_max_in = hp.K;
        fo.conv_src = pc;
        fo.conv_src_stride = PK;
      }
      fo.epi = 2;
      fo.out0 = out0;
      fo.out1 = out1;
      fo.x0 = in0;
      fo.x1 = in1;
      fo.w = c->pinv;
      fo.wp = c->pinv_prec;
      launch_ntt_fused(dc, tmp, level, 0, level, false, s, 0, np, QL, 0, fo);
    } else {
      launch_ntt(dc, tmp, level, 0, level, false, s, 0, np, QL);
      launch_moddown_tail2(dc, out0, out1 ? out1 : out0, in0, in1 ? in1 : in0, tmp, tmp + (in1 ? QL : 0), c->pinv, c->pinv_prec,
                           level, s, np);
    }
  }
  stat(ST_MODDOWN, np, 8ull * np * N * (2 * level + hp.K));
  return post_launch();
}
int acehip_mod_down2(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1, uint32_t level,
                     acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_mod_down2: bad level");
  if (!out0 || !out1 || !in0 || !in1 || out0 == in0 || out1 == in1 || out0 == in1 || out1 == in0 || out0 == out1)
    return fail(ACEHIP_EINVAL, "acehip_mod_down2: outputs must not alias inputs or each other");
  return for_replica_chunks(c, [&] { return do_mod_down_n(c, out0, out1, in0, in1, level, (hipStream_t)s); });
}

}  // extern "C"
// Key inner product + Mod_down of both accumulators with the accumulators never stored (kernels.hpp Kmac): the inverse first pass of
// the P-limbs and the Mod_down tail on the q-limbs form the sums where they would load them.  Per key-switch 4 (level + K) limb
// transfers less than key inner product kernel + acehip_mod_down2 (2 (level + K) written, 2 (level + K) read back).
//   ACEHIP_KMAC_FUSE=0: never (callers use the unfused pipelines), 2: also where the transforms would run as narrow passes
static std::atomic<int>& kmac_fuse_slot() {
  static std::atomic<int> m{[] { const char* e = getenv("ACEHIP_KMAC_FUSE"); return e ? atoi(e) : 1; }()};
  return m;
}
static int kmac_fuse_mode() { return kmac_fuse_slot().load(std::memory_order_relaxed); }
bool kmac_fusable(const acehip_ctx* c, u32 level, u32 nd) {
  const int mode = kmac_fuse_mode();
  if (mode == 0 || !c->on_device || c->dc.logN != 16 || sharded(c) || nd == 0 || nd > kKmacMaxDigits) return false;
  if (mode >= 2) return true;
  // (measured on the single C3 key-switch, 14 P-limb rows and 50 q-limb rows: 0.230 -> 0.221 ms although the P-limbs' inverse passes turn
  //  wide -- one launch less; where the q-limb passes are narrow too the narrow pipeline stays)
  const u32 rows_q = 2 * level * c->seln;
  return rows_q > c->dc.ntt_narrow_max_rows;
}
// word ranges [a, a + na) and [b, b + nb) share a word
static bool overlaps(const u64* a, size_t na, const u64* b, size_t nb) { return a < b + nb && b < a + na; }
// own_stats: record ST_KEYMAC / ST_MODDOWN here (false: the caller accounts for the whole operation, e.g. ST_KEYSWITCH)
int do_keymac_mod_down2(acehip_ctx* c, u64* out0, u64* out1, const Kmac& km, u32 level, hipStream_t s, u64* scratch = nullptr,
                        bool own_stats = true) {
  const HostParams& hp = c->hp;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const size_t N = hp.N, PK = (size_t)hp.K * N, QL = (size_t)level * N, E = QL + PK;
  dbg_touch(out0, QL);
  dbg_touch(out1, QL);
  for (u32 d = 0; d < km.nd; ++d) {
    dbg_touch(km.ext[d], E);
    dbg_touch(km.key[d], 2 * (size_t)(hp.L + hp.K) * N);
  }
  if (km.own) dbg_touch(km.own, QL);
  u64* pc = scratch ? scratch : c->ws;  // [2][K][N] p-limbs of both accumulators in the coefficient domain
  u64* tmp = pc + 2 * PK;               // [2][level][N]
  const bool conv_in_ntt = conv_fusable_down(c, level);
  for (const DevCtx& dc : launch_dcs(c)) {
    NttFuse fi;
    fi.km = km;
    fi.km_pos0 = level;             // launch position j = extended position level + j
    fi.inv_scale = plan->inv_down;  // (P/p_j)^-1 folded into the last inverse stage
    launch_ntt_fused(dc, pc, 0, 0, hp.K, true, s, 0, 2, PK, 0, fi);
    if (!conv_in_ntt) launch_base_conv_batch(dc, tmp, QL, pc, PK, plan->d_descs + plan->nd + 1, 0, 2, level, s, hp.K, PtrTab8{}, plan->mfma_down);
    NttFuse fo;
    if (conv_in_ntt) {
      fo.conv = plan->d_descs + plan->nd + 1;
      fo.conv_step = 0;
      fo.conv_max_in = hp.K;
      fo.conv_src = pc;
      fo.conv_src_stride = PK;
    }
    fo.epi = 3;
    fo.km = km;
    fo.km_pos0 = 0;
    fo.out0 = out0;
    fo.out1 = out1;
    fo.w = c->pinv;
    fo.wp = c->pinv_prec;
    launch_ntt_fused(dc, tmp, level, 0, level, false, s, 0, 2, QL, 0, fo);
  }
  if (own_stats) {
    stat(ST_KEYMAC, 1, 8ull * E * (3ull * km.nd + 2));
    stat(ST_MODDOWN, 2, 8ull * 2 * N * (2 * level + hp.K));
  }
  return post_launch();
}
extern "C" {
int acehip_debug_set_kmac_fuse(int mode) { return kmac_fuse_slot().exchange(mode); }  // tests: the ACEHIP_KMAC_FUSE setting; returns the old one
int acehip_keymac_fusable(const acehip_ctx* c, uint32_t level, uint32_t n_digits) {
  return c && level >= 1 && level <= c->hp.L && kmac_fusable(c, level, n_digits) ? 1 : 0;
}
int acehip_keymac_mod_down2(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* const* h_ext, const uint64_t* const* h_key,
                            uint32_t n_digits, uint32_t level, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: bad level");
  if (!out0 || !out1 || out0 == out1 || !h_ext || !h_key || n_digits == 0 || n_digits > kKmacMaxDigits)
    return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: bad arguments");
  const HostParams& hp = c->hp;
  const size_t N = hp.N, E = (size_t)(level + hp.K) * N, T = hp.L + hp.K;
  const size_t QLw = (size_t)level * N;
  if (overlaps(out0, QLw, out1, QLw)) return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: the outputs overlap");
  for (u32 d = 0; d < n_digits; ++d) {
    if (!h_ext[d] || !h_key[d]) return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: null digit or key part");
    // the last pass reads the digits while it writes the outputs (the two polynomials' workgroups of a tile read the same digit words)
    if (overlaps(out0, QLw, h_ext[d], E) || overlaps(out1, QLw, h_ext[d], E))
      return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: an output overlaps a raised digit");
  }
  hipStream_t s = (hipStream_t)s_;
  if (kmac_fusable(c, level, n_digits)) {
    Kmac km;
    km.nd = n_digits;
    km.level = level;
    km.key_T = (u32)T;
    for (u32 d = 0; d < n_digits; ++d) {
      km.ext[d] = h_ext[d];
      km.key[d] = h_key[d];
    }
    return for_replica_chunks(c, [&] { return do_keymac_mod_down2(c, out0, out1, km, level, s); });
  }
  // unfused: the accumulators in the workspace (behind Mod_down's own scratch), one key inner product launch per digit
  u64* acc0 = c->ws + 2 * (size_t)hp.K * N + 2 * (size_t)level * N;
  u64* acc1 = acc0 + E;
  if ((size_t)(acc1 + E - c->ws) > c->ws_words) return fail(ACEHIP_EINVAL, "acehip_keymac_mod_down2: workspace too small");
  return for_replica_chunks(c, [&] {
    for (const DevCtx& dc : launch_dcs(c))
      for (u32 d = 0; d < n_digits; ++d) launch_key_mac(dc, acc0, acc1, h_key[d], h_key[d] + T * N, h_ext[d], level, d != 0, s);
    stat(ST_KEYMAC, 1, 8ull * E * (3ull * n_digits + 2));
    return do_mod_down_n(c, out0, out1, acc0, acc1, level, s);
  });
}

// ModRaise of bootstrapping (Transform_values_from_level0 ckks_bootstrap_context.c:1527-1551): limb 0 of each
// polynomial (NTT domain) -> coefficient domain -> centred lift -> residues on `level_out` limbs -> NTT domain
int acehip_mod_raise(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1,
                     uint32_t level_out, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level_out == 0 || level_out > c->hp.L || !out0 || !in0 || (in1 != nullptr) != (out1 != nullptr))
    return fail(ACEHIP_EINVAL, "acehip_mod_raise: bad arguments");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 np = in1 ? 2 : 1;
  u64* last = ws_at(c, 0);  // [np][N]
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs) {  // limb 0 (its owner only) -> coefficient domain, centred lift
    if (dc.logN == 16) {
      NttFuse fi;
      fi.src0 = in0;
      fi.src1 = in1;
      fi.center_out = true;
      launch_ntt_fused(dc, last, hp.L, 0, 1, true, s, 0, np, N, 0, fi);
    } else {
      for (u32 z = 0; z < np; ++z) {
        copy_limbs_dc(dc, last + z * N, z ? in1 : in0, 1, 0, s);
        launch_ntt(dc, last + z * N, hp.L, 0, 1, true, s);
        launch_center(dc, (int64_t*)(last + z * N), last + z * N, 0, s);
      }
    }
  }
  if (sharded(c)) {
    XItem x[2] = {{last, 0}, {last + N, 0}};
    if (int e = shard_exchange(c, x, np, s)) return e;
  }
  for (const DevCtx& dc : dcs) {
    if (dc.logN == 16) {
      NttFuse fo;
      fo.msg = (const int64_t*)last;
      fo.msg_stride = N;
      launch_ntt_fused(dc, out0, level_out, 0, level_out, false, s, 0, np, (size_t)(out1 - out0), 0, fo);
    } else {
      for (u32 z = 0; z < np; ++z) {
        u64* out = z ? out1 : out0;
        launch_values_to_rns(dc, out, (const int64_t*)(last + z * N), level_out, 0, level_out, s);
        launch_ntt(dc, out, level_out, 0, level_out, false, s);
      }
    }
  }
  stat(ST_RESCALE, np, 8ull * N * (1 + level_out) * np);
  return post_launch();
}

// Base conversion onto a SUBSET of the target limbs (limb-sharded execution, SURVEY 8e: every GPU converts only the
// limbs it owns once the source limbs have been gathered).  which = digit index (ModUp of that digit at `level`,
// sources = the digit's limbs in the coefficient domain, NOT yet scaled) or ACEHIP_CONV_MODDOWN (sources = the K
// p-limbs in the coefficient domain).  d_in: the n_in source limbs, contiguous; h_out_pos: target limb positions (in the
// polynomial extended at `level`); d_out: n_out limbs, output k is the limb at h_out_pos[k], coefficient domain.
int acehip_base_conv(acehip_ctx* c, uint64_t* d_out, const uint64_t* d_in, uint32_t level, int which, const uint32_t* h_out_pos,
                     uint32_t n_out, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_out || !d_in || !h_out_pos || n_out == 0 || n_out > hp.L + hp.K)
    return fail(ACEHIP_EINVAL, "acehip_base_conv: bad arguments");
  std::vector<u32> gi(n_out), col(n_out), pos(n_out);
  ConvDesc cd{};
  if (which == ACEHIP_CONV_MODDOWN) {
    for (u32 k = 0; k < n_out; ++k) {
      if (h_out_pos[k] >= level) return fail(ACEHIP_EINVAL, "acehip_base_conv: ModDown targets are q-limbs below the level");
      gi[k] = col[k] = h_out_pos[k];
      pos[k] = k;
    }
    cd.hat = c->phat_modq_t;
    cd.scale = c->phat_inv;
    cd.scale_prec = c->phat_inv_prec;
    cd.src_gi = c->p_gi;
    cd.n_in = hp.K;
    cd.hat_ld = hp.L;
  } else {
    if (which < 0 || (u32)which >= hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_base_conv: bad digit");
    const DevModUp* t = get_modup(c, level, (u32)which);
    if (!t) return fail(ACEHIP_EHIP, "ModUp table upload failed");
    HostParams::ModUp hm = hp.modup(level, (u32)which);
    for (u32 k = 0; k < n_out; ++k) {
      const u32 p = h_out_pos[k];
      const u32 want = p < level ? p : hp.L + (p - level);  // global prime index of the target
      u32 j = 0;
      while (j < hm.nc && hm.compl_idx[j] != want) ++j;
      if (p >= level + hp.K || j == hm.nc) return fail(ACEHIP_EINVAL, "acehip_base_conv: target is not a complement limb of the digit");
      gi[k] = want;
      col[k] = j;
      pos[k] = k;
    }
    cd.hat = t->hat_mod;
    cd.scale = t->hat_inv;
    cd.scale_prec = t->hat_inv_prec;
    cd.src_gi = t->src_gi;
    cd.n_in = t->n2;
    cd.hat_ld = t->nc;
  }
  // the three index lists and the descriptor travel in one small upload; freed after the launch has been ordered
  std::vector<u32> blob;
  blob.insert(blob.end(), gi.begin(), gi.end());
  blob.insert(blob.end(), col.begin(), col.end());
  blob.insert(blob.end(), pos.begin(), pos.end());
  u32* d_blob = nullptr;
  ConvDesc* d_desc = nullptr;
  HIP_TRY(hipMalloc((void**)&d_blob, blob.size() * sizeof(u32)));
  HIP_TRY(hipMalloc((void**)&d_desc, sizeof(ConvDesc)));
  HIP_TRY(hipMemcpy(d_blob, blob.data(), blob.size() * sizeof(u32), hipMemcpyHostToDevice));
  cd.out_gi = d_blob;
  cd.col = d_blob + n_out;
  cd.out_pos = d_blob + 2 * n_out;
  cd.src_pos0 = 0;
  cd.n_out = n_out;
  HIP_TRY(hipMemcpy(d_desc, &cd, sizeof(ConvDesc), hipMemcpyHostToDevice));
  hipStream_t s = (hipStream_t)s_;
  for (DevCtx dc : launch_dcs(c)) {
    dc.sh_world = 1;  // (the caller names the outputs it wants)
    launch_base_conv_batch(dc, d_out, 0, d_in, 0, d_desc, 0, 1, n_out, s, cd.n_in);
  }
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(d_blob);
  (void)hipFree(d_desc);
  return post_launch();
}

// one or two polynomials (c0, c1 of a ciphertext) through Rescale_poly in the same launches
static int do_rescale(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s) {
  const HostParams& hp = c->hp;
  const size_t N = hp.N;
  const u32 np = in1 ? 2 : 1;
  u64* last = ws_at(c, 0);   // [np][N]
  u64* t = ws_at(c, 2);      // [np][level-1][N]
  dbg_touch(out0, (size_t)(level - 1) * N);
  dbg_touch(out1, (size_t)(level - 1) * N);
  dbg_touch(in0, (size_t)level * N);
  dbg_touch(in1, (size_t)level * N);
  const size_t t_stride = (size_t)(level - 1) * N;
  const size_t row = (size_t)(level - 2) * hp.L;
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs) {  // the last limb (its owner only) -> coefficient domain
    if (dc.logN == 16) {
      // fused: the iNTT reads the last limbs where they lie and leaves their centred lift; the forward NTT of the
      // remaining limbs starts from that lift (modulus switch and constant folded into its first pass) and applies
      // the Rescale tail in its last pass: 4 launches, no intermediate polynomial in memory
      NttFuse fi;
      fi.src0 = in0 + (size_t)(level - 1) * N;
      fi.src1 = in1 ? in1 + (size_t)(level - 1) * N : nullptr;
      fi.center_out = true;
      launch_ntt_fused(dc, last, hp.L, level - 1, 1, true, s, level - 1, np, N, 0, fi);
    } else {
      for (u32 z = 0; z < np; ++z) copy_limbs_dc(dc, last + z * N, (z ? in1 : in0) + (size_t)(level - 1) * N, 1, level - 1, s);
      launch_ntt(dc, last, hp.L, level - 1, 1, true, s, level - 1, np, N);
    }
  }
  if (sharded(c)) {  // SURVEY 8e collective 3: the owner of limb level-1 sends it to everyone
    XItem x[2] = {{last, (level - 1) % c->sh_world}, {last + N, (level - 1) % c->sh_world}};
    if (int e = shard_exchange(c, x, np, s)) return e;
  }
  for (const DevCtx& dc : dcs) {
    if (dc.logN == 16) {
      NttFuse fo;
      fo.msg = (const int64_t*)last;
      fo.msg_stride = N;
      fo.msg_scale = c->qlql + row;
      fo.epi = 1;
      fo.out0 = out0;
      fo.out1 = out1;
      fo.x0 = in0;
      fo.x1 = in1;
      fo.w = c->ql_inv + row;
      fo.wp = c->ql_inv_prec + row;
      launch_ntt_fused(dc, t, hp.L, 0, level - 1, false, s, 0, np, t_stride, 0, fo);
    } else {
      launch_rescale_spread(dc, t, t_stride, last, N, c->qlql + row, c->qlql_prec + row, level, np, s);
      launch_ntt(dc, t, hp.L, 0, level - 1, false, s, 0, np, t_stride);
      launch_rescale_tail(dc, out0, out1, in0, in1, t, t_stride, c->ql_inv + row, c->ql_inv_prec + row, level, np, s);
    }
  }
  stat(ST_RESCALE, np, 8ull * N * (2 * level - 1) * np);
  return post_launch();
}
int acehip_rescale(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_rescale: level must be in [2, L]");
  return for_replica_chunks(c, [&] { return do_rescale(c, out, nullptr, in, nullptr, level, (hipStream_t)s); });
}
int acehip_rescale2(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1, uint32_t level,
                    acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_rescale2: level must be in [2, L]");
  if (!out0 || !out1 || !in0 || !in1) return fail(ACEHIP_EINVAL, "acehip_rescale2: null polynomial");
  return for_replica_chunks(c, [&] { return do_rescale(c, out0, out1, in0, in1, level, (hipStream_t)s); });
}

}  // extern "C"
// B fragments and accumulator offsets of the matrix-core base conversion (keyswitch.hip base_conv_mfma_kernel) for one problem:
// hat(i, j) = constant of source i and output j, t(j) = output prime.  Layout [tile][step][digit][lane][16 bytes]; byte e of
// lane (r = lane & 15, g = lane >> 4) belongs to source limb i = 8*step + 2g + (e >> 3), byte a = e & 7 of its residue, and to
// output j = 16*tile + r: digit b of  hat(i, j) * 2^(8a) mod t(j).
template <class Hat, class Prime>
static void conv_mfma_tables(u32 n_in, u32 n_out, u32 steps, Hat hat, Prime t, std::vector<uint8_t>& frag, std::vector<u32>& off) {
  const u32 tiles = (n_out + 15) / 16, NB = kConvMfmaDigits;
  frag.assign((size_t)tiles * steps * NB * 64 * 16, 0);
  off.assign((size_t)tiles * 16 * NB, 0);
  for (u32 j = 0; j < n_out; ++j) {
    const u64 tj = t(j);
    const u32 tile = j / 16, r = j % 16;
    for (u32 i = 0; i < n_in; ++i) {
      u64 gv = hat(i, j) % tj;
      for (u32 a = 0; a < 8; ++a) {
        const u32 step = i / 8, g = (i % 8) / 2, e = (i % 2) * 8 + a, lane = g * 16 + r;
        for (u32 b = 0; b < NB; ++b) {
          const u32 dig = (u32)(gv >> (7 * b)) & 127u;
          frag[((((size_t)tile * steps + step) * NB + b) * 64 + lane) * 16 + e] = (uint8_t)dig;
          off[(size_t)j * NB + b] += 128u * dig;
        }
        gv = (u64)(((unsigned __int128)gv << 8) % tj);
      }
    }
  }
}

const KsPlan* get_ks_plan(acehip_ctx* c, u32 level) {
  {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->ks_plans.find(level);
    if (it != c->ks_plans.end()) return &it->second;
  }
  const HostParams& hp = c->hp;
  KsPlan plan;
  plan.nd = hp.num_decomp(level);
  const bool fold = c->dc.logN == 16;
  // matrix-core base conversion (keyswitch.hip base_conv_mfma_kernel): whole workgroups of 1024 coefficients, digits / K of at
  // most 16 limbs, primes between 2^32 and 2^61 (nine 7-bit digits; keyswitch.hip reduce80).  ACEHIP_CONV_MFMA=0 keeps the multiply-add kernels (measurement).
  static const bool mfma_on = [] { const char* e = getenv("ACEHIP_CONV_MFMA"); return !e || atoi(e) != 0; }();
  const u32 steps_up = (hp.alpha + 7) / 8, steps_down = (hp.K + 7) / 8;
  bool mfma = mfma_on && c->on_device && hp.N % 1024 == 0 && hp.alpha <= 16 && hp.K <= 16;
  for (u32 i = 0; i < hp.L + hp.K && mfma; ++i) mfma = (hp.primes[i].q >> 61) == 0 && (hp.primes[i].q >> 32) != 0;  // (reduce80: 5q < 2^64, mh < 2^32)
  if (mfma) {
    plan.mfma_up = steps_up;
    plan.mfma_down = steps_down;
  }
  std::vector<u64> inv_up(4 * (size_t)level, 0), inv_down(4 * (size_t)hp.K, 0);
  std::vector<ConvDesc> descs;
  for (u32 d = 0; d < plan.nd; ++d) {
    const DevModUp* t = get_modup(c, level, d);
    if (!t) return nullptr;
    ConvDesc cd{};
    cd.hat = t->hat_mod;
    cd.scale = fold ? nullptr : t->hat_inv;
    cd.scale_prec = fold ? nullptr : t->hat_inv_prec;
    if (fold) {
      HostParams::ModUp hm = hp.modup(level, d);
      for (u32 i = 0; i < hm.n2; ++i) {
        const PrimeConsts& P = hp.primes[hm.start + i];
        const u64 tn = mul_mod(P.n_inv, hm.hat_inv[i], P.q), tw = mul_mod(P.inv_w1_ninv, hm.hat_inv[i], P.q);
        u64* o = &inv_up[4 * (size_t)(hm.start + i)];
        o[0] = tn;
        o[1] = shoup_prec(tn, P.q);
        o[2] = tw;
        o[3] = shoup_prec(tw, P.q);
      }
    }
    cd.src_gi = t->src_gi;
    cd.out_gi = t->out_gi;
    cd.out_pos = t->out_pos;
    cd.src_pos0 = t->start;
    cd.n_in = t->n2;
    cd.n_out = t->nc;
    cd.hat_ld = t->nc;
    plan.max_nc = std::max(plan.max_nc, t->nc);
    if (mfma) {
      HostParams::ModUp hm = hp.modup(level, d);
      std::vector<uint8_t> frag;
      std::vector<u32> off;
      conv_mfma_tables(hm.n2, hm.nc, steps_up, [&](u32 i, u32 j) { return hm.hat_mod[(size_t)i * hm.nc + j]; },
                       [&](u32 j) { return hp.primes[hm.compl_idx[j]].q; }, frag, off);
      std::lock_guard<std::mutex> lk(c->mu);
      cd.bfrag = c->up(frag);
      cd.boff = c->up(off);
      if (!cd.bfrag || !cd.boff) return nullptr;
    }
    descs.push_back(cd);
  }
  ConvDesc md{};  // ModDown: K p-limbs at positions level.. -> level q-limbs (polynomial.c:755-807)
  md.hat = c->phat_modq_t;
  md.scale = fold ? nullptr : c->phat_inv;
  md.scale_prec = fold ? nullptr : c->phat_inv_prec;
  if (fold)
    for (u32 j = 0; j < hp.K; ++j) {
      const PrimeConsts& P = hp.primes[hp.L + j];
      const u64 tn = mul_mod(P.n_inv, hp.phat_inv_modp[j], P.q), tw = mul_mod(P.inv_w1_ninv, hp.phat_inv_modp[j], P.q);
      inv_down[4 * j + 0] = tn;
      inv_down[4 * j + 1] = shoup_prec(tn, P.q);
      inv_down[4 * j + 2] = tw;
      inv_down[4 * j + 3] = shoup_prec(tw, P.q);
    }
  md.src_gi = c->p_gi;
  md.out_gi = c->q_gi;
  md.out_pos = c->q_pos;
  md.src_pos0 = level;
  md.n_in = hp.K;
  md.n_out = level;
  md.hat_ld = hp.L;
  if (mfma) {
    std::vector<uint8_t> frag;
    std::vector<u32> off;
    conv_mfma_tables(hp.K, level, steps_down, [&](u32 i, u32 j) { return hp.phat_modq[(size_t)j * hp.K + i]; },
                     [&](u32 j) { return hp.primes[j].q; }, frag, off);
    std::lock_guard<std::mutex> lk(c->mu);
    md.bfrag = c->up(frag);
    md.boff = c->up(off);
    if (!md.bfrag || !md.boff) return nullptr;
  }
  descs.push_back(md);
  md.src_pos0 = 0;  // [nd + 1]: the same problem with the K sources at limb positions 0.. (Mod_down: the P-limbs sit in a scratch of their own)
  descs.push_back(md);
  std::lock_guard<std::mutex> lk(c->mu);
  plan.d_descs = c->up(descs);
  if (!plan.d_descs) return nullptr;
  if (fold) {
    plan.inv_up = c->up(inv_up);
    plan.inv_down = c->up(inv_down);
    if (!plan.inv_up || !plan.inv_down) return nullptr;
  }
  return &(c->ks_plans[level] = plan);
}
extern "C" {

// the tables of the matrix-core base conversion for one problem, as get_ks_plan uploads them (host arithmetic only)
long acehip_conv_mfma_tables(const acehip_ctx* c, uint32_t level, int32_t digit, uint8_t* frag, size_t frag_cap, uint32_t* off, size_t off_cap,
                             uint32_t* dims) {
  if (!c || !dims) return fail(ACEHIP_EINVAL, "acehip_conv_mfma_tables: null argument");
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L) return fail(ACEHIP_EINVAL, "acehip_conv_mfma_tables: bad level");
  if (hp.alpha > 16 || hp.K > 16) return fail(ACEHIP_EINVAL, "acehip_conv_mfma_tables: digits of more than 16 limbs have no matrix-core form");
  std::vector<uint8_t> f;
  std::vector<u32> o;
  u32 n_in, n_out, steps;
  if (digit >= 0) {
    if ((u32)digit >= hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_conv_mfma_tables: bad digit");
    HostParams::ModUp hm = hp.modup(level, (u32)digit);
    n_in = hm.n2;
    n_out = hm.nc;
    steps = (hp.alpha + 7) / 8;
    conv_mfma_tables(n_in, n_out, steps, [&](u32 i, u32 j) { return hm.hat_mod[(size_t)i * hm.nc + j]; },
                     [&](u32 j) { return hp.primes[hm.compl_idx[j]].q; }, f, o);
  } else {
    n_in = hp.K;
    n_out = level;
    steps = (hp.K + 7) / 8;
    conv_mfma_tables(n_in, n_out, steps, [&](u32 i, u32 j) { return hp.phat_modq[(size_t)j * hp.K + i]; },
                     [&](u32 j) { return hp.primes[j].q; }, f, o);
  }
  dims[0] = n_in;
  dims[1] = n_out;
  dims[2] = steps;
  dims[3] = (n_out + 15) / 16;
  if (frag && frag_cap >= f.size()) std::memcpy(frag, f.data(), f.size());
  if (off && off_cap >= o.size()) std::memcpy(off, o.data(), o.size() * sizeof(u32));
  return (long)f.size();
}

static int key_switch_impl(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in, const uint64_t* key, uint32_t level, acehip_stream s_);
int acehip_key_switch(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in, const uint64_t* key,
                      uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  return for_replica_chunks(c, [&] { return key_switch_impl(c, out0, out1, in, key, level, s); });
}
static int key_switch_impl(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in, const uint64_t* key,
                           uint32_t level, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_switch: bad level");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const size_t N = hp.N, E = (size_t)(level + hp.K) * N;  // words per extended polynomial
  const u32 nd = plan->nd;
  // workspace: coef (level limbs) | ext[nd] | acc0 | acc1 | tmp[2] (level limbs each)
  dbg_touch(out0, (size_t)level * N);
  dbg_touch(out1, (size_t)level * N);
  dbg_touch(in, (size_t)level * N);
  dbg_touch(key, (size_t)nd * 2 * (hp.L + hp.K) * N);
  u64* coef = c->ws;
  u64* ext = coef + (size_t)level * N;
  u64* acc0 = ext + nd * E;
  u64* acc1 = acc0 + E;
  u64* tmp = acc1 + E;
  const bool fused = c->dc.logN == 16;
  const DcList dcs = launch_dcs(c);
  const u32 n_ext_rows = level + hp.K - std::min(hp.alpha, level - hp.alpha * (nd - 1));
  // 1. all digit limbs to the coefficient domain in one launch (polynomial.c:1276-1283 for every part)
  for (const DevCtx& dc : dcs) {
    if (fused) {
      NttFuse fi;
      fi.src0 = in;
      fi.inv_scale = plan->inv_up;  // (Q_d/q_i)^-1 folded into the last inverse stage
      launch_ntt_fused(dc, coef, hp.L, 0, level, true, s, 0, 1, 0, 0, fi);
    } else {
      copy_limbs_dc(dc, coef, in, level, 0, s);
      launch_ntt(dc, coef, hp.L, 0, level, true, s);
    }
  }
  if (sharded(c)) {  // SURVEY 8e collective 1: every rank needs every source limb of the conversions
    std::vector<XItem> x;
    for (u32 i = 0; i < level; ++i) x.push_back(XItem{coef + (size_t)i * N, i % c->sh_world});
    if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
  }
  for (const DevCtx& dc : dcs) {
    // 2. every digit's base conversion (scaling by (Q_d/q_i)^-1 folded into the inverse NTT) and
    // 3. the NTT of every digit's complement limbs (own digit limbs are skipped): at N = 2^16 one pipeline, the conversion
    //    is computed by the first NTT pass while it loads its input
    if (conv_fusable_up(c, level)) {
      NttFuse fc;
      fc.conv = plan->d_descs;
      fc.conv_step = 1;
      fc.conv_max_in = std::min(hp.alpha, level);
      fc.conv_src = coef;
      fc.conv_src_stride = 0;
      launch_ntt_fused(dc, ext, level, 0, n_ext_rows, false, s, 0, nd, E, hp.alpha, fc);
    } else {
      launch_base_conv_batch(dc, ext, E, coef, 0, plan->d_descs, 1, nd, plan->max_nc, s, hp.alpha, PtrTab8{}, plan->mfma_up);
      launch_ntt(dc, ext, level, 0, n_ext_rows, false, s, 0, nd, E, hp.alpha);
    }
  }
  // (the fused last pass reads `in` -- the digits' own limbs -- while it writes out0 / out1, and both polynomials' workgroups of a tile read
  //  the same words of `in`: an output that overlaps the input takes the pipeline with stored accumulators, which has read `in` completely
  //  before it writes)
  const bool out_aliases_in = overlaps(out0, (size_t)level * N, in, (size_t)level * N) || overlaps(out1, (size_t)level * N, in, (size_t)level * N);
  if (kmac_fusable(c, level, nd) && !out_aliases_in) {
    // 4 + 5. the key inner product is formed by Mod_down's own passes (the accumulators are never stored); a digit's own limbs are
    // read from `in`.  Mod_down's scratch (2 K + 2 level limbs) takes the place of the accumulators behind the digits.
    Kmac km;
    km.nd = nd;
    km.alpha = hp.alpha;
    km.level = level;
    km.key_T = hp.L + hp.K;
    km.own = in;
    for (u32 d = 0; d < nd; ++d) {
      km.ext[d] = ext + d * E;
      km.key[d] = key + (size_t)d * 2 * (hp.L + hp.K) * N;
    }
    stat(ST_KEYSWITCH, 1, acehip_key_switch_bytes(c, level));
    return do_keymac_mod_down2(c, out0, out1, km, level, s, acc0, false);
  }
  for (const DevCtx& dc : dcs) {
    // 4. key inner product fused over digits; a digit's own limbs are read from `in` directly
    launch_key_mac_fused(dc, acc0, acc1, key, ext, E, in, level, nd, hp.alpha, s);
    // 5. ModDown of both accumulators together (polynomial.c:928-967)
    if (fused) {
      NttFuse fa;
      fa.inv_scale = plan->inv_down - 4 * (size_t)level;  // the p-limbs sit at positions level .. level+K-1
      launch_ntt_fused(dc, acc0, level, level, hp.K, true, s, 0, 2, E, 0, fa);
    } else {
      launch_ntt(dc, acc0, level, level, hp.K, true, s, 0, 2, E);
    }
  }
  if (sharded(c)) {  // collective 2: the P-limbs of both accumulators
    std::vector<XItem> x;
    for (u32 z = 0; z < 2; ++z)
      for (u32 j = 0; j < hp.K; ++j) x.push_back(XItem{acc0 + z * E + (size_t)(level + j) * N, (hp.L + j) % c->sh_world});
    if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
  }
  const bool conv_in_ntt = fused && conv_fusable_down(c, level);
  for (const DevCtx& dc : dcs) {
    if (!conv_in_ntt) launch_base_conv_batch(dc, tmp, (size_t)level * N, acc0, E, plan->d_descs + nd, 0, 2, level, s, hp.K, PtrTab8{}, plan->mfma_down);
    if (fused) {  // the ModDown tail rides in the last NTT pass, the conversion P -> Q in the first
      NttFuse fo;
      if (conv_in_ntt) {
        fo.conv = plan->d_descs + nd;
        fo.conv_step = 0;
        fo.conv_max_in = hp.K;
        fo.conv_src = acc0;
        fo.conv_src_stride = E;
      }
      fo.epi = 2;
      fo.out0 = out0;
      fo.out1 = out1;
      fo.x0 = acc0;
      fo.x1 = acc1;
      fo.w = c->pinv;
      fo.wp = c->pinv_prec;
      launch_ntt_fused(dc, tmp, level, 0, level, false, s, 0, 2, (size_t)level * N, 0, fo);
    } else {
      launch_ntt(dc, tmp, level, 0, level, false, s, 0, 2, (size_t)level * N);
      launch_moddown_tail2(dc, out0, out1, acc0, acc1, tmp, tmp + (size_t)level * N, c->pinv, c->pinv_prec, level, s);
    }
  }
  stat(ST_KEYSWITCH, 1, acehip_key_switch_bytes(c, level));
  return post_launch();
}

int acehip_values_to_rns(acehip_ctx* c, uint64_t* d, const int64_t* vals, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  for (const DevCtx& dc : launch_dcs(c)) launch_values_to_rns(dc, d, vals, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
int acehip_sample_uniform_keyed(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, const uint32_t* h_key, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (!h_key || !d) return fail(ACEHIP_EINVAL, "acehip_sample_uniform_keyed: null argument");
  if (c->hp.N % 4 != 0) return fail(ACEHIP_EINVAL, "acehip_sample_uniform_keyed: N must be a multiple of 4");
  ChaChaKey k;
  std::memcpy(k.w, h_key, sizeof k.w);
  for (const DevCtx& dc : launch_dcs(c)) launch_sample_uniform_keyed(dc, d, level, pos0, n, k, (hipStream_t)s);
  return post_launch();
}
int acehip_sample_uniform(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, uint64_t seed, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  for (const DevCtx& dc : launch_dcs(c)) launch_sample_uniform(dc, d, level, pos0, n, seed, (hipStream_t)s);
  return post_launch();
}
int acehip_mul_scalars(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n > 64) return fail(ACEHIP_EINVAL, "acehip_mul_scalars: at most 64 limbs per call");
  LimbConsts w{};
  for (u32 i = 0; i < n; ++i) w.w[i] = h_scalars[i];
  for (const DevCtx& dc : launch_dcs(c)) launch_mul_scalars(dc, r, a, w, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
}  // extern "C"
// Encode_at_level_with_sf ckks_encoder.c:395 -> Encode_impl :199-297 (64-bit path)
int ensure_embed_tables(acehip_ctx* c) {
  std::lock_guard<std::mutex> g(c->mu);
  if (c->emb_rou) return 0;
  const size_t N = c->hp.N, m = 2 * N;
  std::vector<double> rou(2 * m);
  for (size_t i = 0; i < m; ++i) {  // Precompute_fft ntt.c:587-610
    // glibc's sincos(), which is what gcc makes of the reference's cos(angle) + sin(angle) pair: it differs from
    // separate cos()/sin() calls in the last bit for ~0.1% of the entries, and clang would emit the latter
    const double angle = 2 * M_PI * i / m;
    sincos(angle, &rou[2 * i + 1], &rou[2 * i]);
  }
  std::vector<u32> rot(N / 2 ? N / 2 : 1, 1);
  for (size_t i = 1; i < N / 2; ++i) rot[i] = (u32)((5ull * rot[i - 1]) % m);
  // per-stage layout of the twiddles Embedding_inv uses: stage logm, butterfly i < 2^(logm-1)
  std::vector<double> tws(2 * (N / 2 ? N / 2 : 1), 0.0);
  for (u32 logm = 1; (1ull << logm) <= N / 2; ++logm) {
    const size_t idx_mod = 1ull << (logm + 2), gap = m / idx_mod, half = 1ull << (logm - 1);
    for (size_t i = 0; i < half; ++i) {
      const size_t k = (idx_mod - (rot[i] % idx_mod)) * gap;
      tws[2 * (half - 1 + i)] = rou[2 * k];
      tws[2 * (half - 1 + i) + 1] = rou[2 * k + 1];
    }
  }
  rou.swap(tws);  // the device gets the per-stage table (the flat one is only needed to build it)
  u32* d_rot = c->up(rot);
  double* d_rou = c->up(rou);
  void *work = nullptr, *msg = nullptr, *err = nullptr;
  // scratch for a batch of EMB_BATCH_MAX messages (acehip_encode_batch)
  if (!d_rot || !d_rou || hipMalloc(&work, EMB_BATCH_MAX * (N / 2 * 16) + 16) != hipSuccess ||
      hipMalloc(&msg, (size_t)EMB_BATCH_MAX * N * 8) != hipSuccess ||
      hipMalloc(&err, 64) != hipSuccess)
    return fail(ACEHIP_EHIP, "acehip_encode: table allocation failed");
  c->owned.push_back(work);
  c->owned.push_back(msg);
  c->owned.push_back(err);
  if (hipMemset(err, 0, 64) != hipSuccess) return fail(ACEHIP_EHIP, "acehip_encode: memset failed");
  c->emb_rot = d_rot;
  c->emb_work = (cd*)work;
  c->emb_msg = (int64_t*)msg;
  c->emb_err = (int*)err;
  c->emb_rou = (cd*)d_rou;
  return 0;
}
extern "C" {

// [L] Delta^(sf_degree-1) mod q_i on the device (ckks_encoder.c:270-285), cached per (Delta, sf_degree)
static const u64* encode_scale_table(acehip_ctx* c, u64 sfi, u32 sf_degree) {
  std::lock_guard<std::mutex> g(c->mu);
  u64*& tab = c->enc_scales[{sfi, sf_degree}];
  if (!tab) {
    std::vector<u64> w(c->hp.L);
    for (u32 i = 0; i < c->hp.L; ++i) {
      const u64 q = c->hp.primes[i].q;
      u64 pw = sfi % q;
      for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
      w[i] = pw;
    }
    tab = c->up(w);
  }
  return tab;
}

int acehip_encode_batch(acehip_ctx* c, uint64_t* const* h_q, const void* const* h_vals, uint32_t n_batch, int kind, size_t len,
                        uint32_t slots, double sf, uint32_t sf_degree, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (n_batch == 0 || n_batch > EMB_BATCH_MAX || !h_q || !h_vals || kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) ||
      len > slots || len == 0 || sf_degree < 1 || level == 0 || level > c->hp.L)
    return fail(ACEHIP_EINVAL, "acehip_encode_batch: bad arguments");
  for (u32 b = 0; b < n_batch; ++b)
    if (!h_q[b] || !h_vals[b]) return fail(ACEHIP_EINVAL, "acehip_encode_batch: null pointer in the batch");
  if (c->dc.logN != 16 || n_batch == 1) {  // no batched form below N = 2^16: one encode after the other
    for (u32 b = 0; b < n_batch; ++b)
      if (int e = acehip_encode(c, h_q[b], nullptr, h_vals[b], kind, len, slots, sf, sf_degree, level, 0, s)) return e;
    return ACEHIP_OK;
  }
  if (int e = ensure_embed_tables(c)) return e;
  hipStream_t st = (hipStream_t)s;
  EmbBatch eb{};
  for (u32 b = 0; b < n_batch; ++b) eb.vals[b] = h_vals[b];
  launch_embed_inv_batch(c->emb_msg, c->emb_work, eb, n_batch, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, st);
  NttFuse f;
  f.msg = c->emb_msg;
  f.msg_stride = N;
  for (u32 b = 0; b < n_batch; ++b) {
    f.polyz[b] = h_q[b];
    dbg_touch(h_q[b], (size_t)level * N);
  }
  if (sf_degree > 1) {
    f.msg_scale = encode_scale_table(c, (u64)sf, sf_degree);
    if (!f.msg_scale) return fail(ACEHIP_EHIP, "acehip_encode: scale table upload failed");
  }
  for (const DevCtx& dc : launch_dcs(c)) launch_ntt_fused(dc, h_q[0], level, 0, level, false, st, 0, n_batch, 0, 0, f);
  stat(ST_ENCODE, n_batch, n_batch * (8ull * N * level + len * (kind == 0 ? 4 : kind == 1 ? 8 : 16)));
  return post_launch();
}

static int encode_impl(acehip_ctx* c, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                       double sf, uint32_t sf_degree, uint32_t level, uint32_t n_p, acehip_stream s, double round_add);
int acehip_encode(acehip_ctx* c, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                  double sf, uint32_t sf_degree, uint32_t level, uint32_t n_p, acehip_stream s) {
  return encode_impl(c, d_q, d_p, d_vals, kind, len, slots, sf, sf_degree, level, n_p, s, 0.5);
}
// Encode_impl_with_scale (ckks_encoder.c:301-378): the message is llround(x * scale) -- no half added before the rounding, no power of
// the scaling factor multiplied in afterwards
int acehip_encode_with_scale(acehip_ctx* c, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                             double scale, uint32_t level, uint32_t n_p, acehip_stream s) {
  if (!(scale > 0)) return fail(ACEHIP_EINVAL, "acehip_encode_with_scale: invalid scale for encode");
  return encode_impl(c, d_q, d_p, d_vals, kind, len, slots, scale, 1, level, n_p, s, 0.0);
}
static int encode_impl(acehip_ctx* c, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                       double sf, uint32_t sf_degree, uint32_t level, uint32_t n_p, acehip_stream s, double round_add) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) || len > slots || sf_degree < 1 || level == 0 ||
      level > c->hp.L || n_p > c->hp.K || (n_p && !d_p) || !d_q || (!d_vals && len))
    return fail(ACEHIP_EINVAL, "acehip_encode: bad arguments");
  if (int e = ensure_embed_tables(c)) return e;
  hipStream_t st = (hipStream_t)s;
  dbg_touch(d_q, (size_t)level * N);
  dbg_touch(d_p, (size_t)n_p * N);
  launch_embed_inv(c->emb_msg, c->emb_work, d_vals, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, st, round_add);
  const u64 sfi = (u64)sf;
  for (const DevCtx& dc : launch_dcs(c)) {
    if (dc.logN == 16) {  // the first NTT pass reduces (and scales) the message itself: no residue pass over memory
      NttFuse f;
      f.msg = c->emb_msg;
      if (sf_degree > 1) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1) on the q limbs
        f.msg_scale = encode_scale_table(c, sfi, sf_degree);
        if (!f.msg_scale) return fail(ACEHIP_EHIP, "acehip_encode: scale table upload failed");
      }
      launch_ntt_fused(dc, d_q, level, 0, level, false, st, 0, 1, 0, 0, f);
      if (n_p) {
        f.msg_scale = nullptr;
        launch_ntt_fused(dc, d_p, 0, 0, n_p, false, st, 0, 1, 0, 0, f);
      }
    } else {
      launch_values_to_rns(dc, d_q, c->emb_msg, level, 0, level, st);
      if (n_p) launch_values_to_rns(dc, d_p, c->emb_msg, 0, 0, n_p, st);
      if (sf_degree > 1) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1) on the q limbs
        for (u32 l0 = 0; l0 < level; l0 += 64) {
          LimbConsts w{};
          const u32 n = std::min(64u, level - l0);
          for (u32 i = 0; i < n; ++i) {
            const u64 q = c->hp.primes[l0 + i].q;
            u64 pw = sfi % q;
            for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
            w.w[i] = pw;
          }
          launch_mul_scalars(dc, d_q, d_q, w, level, l0, n, st);
        }
      }
      launch_ntt(dc, d_q, level, 0, level, false, st);
      if (n_p) launch_ntt(dc, d_p, 0, 0, n_p, false, st);
    }
  }
  stat(ST_ENCODE, 1, 8ull * N * (level + n_p) + len * (kind == 0 ? 4 : kind == 1 ? 8 : 16));
  return post_launch();
}

int acehip_encode_status(acehip_ctx* c) {
  if (int e = check_dev(c)) return e;
  if (!c->emb_err) return 0;
  int flag = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&flag, c->emb_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIP_TRY(hipMemset(c->emb_err, 0, sizeof(int)));
    return fail(ACEHIP_EINVAL, "encode overflow, please choose a smaller scaling factor");
  }
  return 0;
}

int acehip_decomp(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_decomp: bad level/digit");
  const u32 start = c->hp.alpha * digit, n2 = std::min(c->hp.alpha, level - start);
  for (const DevCtx& dc : launch_dcs(c)) copy_limbs_dc(dc, out, in + (size_t)start * c->hp.N, n2, start, (hipStream_t)s);
  if (int e = post_launch()) return e;
  return (int)n2;
}
int acehip_mod_up(acehip_ctx* c, uint64_t* out, const uint64_t* digit_limbs, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_mod_up: bad level/digit");
  const u32 start = c->hp.alpha * digit, n2 = std::min(c->hp.alpha, level - start);
  // same pipeline as Decomp_modup with the digit limbs supplied separately
  if (int e = do_decomp_modup(c, out, digit_limbs - (size_t)start * c->hp.N, level, digit, ws_at(c, 0), (hipStream_t)s)) return e;
  return (int)n2;
}

int acehip_add_scalars(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n > 64) return fail(ACEHIP_EINVAL, "acehip_add_scalars: at most 64 limbs per call");
  LimbConsts w{};
  for (u32 i = 0; i < n; ++i) w.w[i] = h_scalars[i];
  for (const DevCtx& dc : launch_dcs(c)) launch_add_scalars(dc, r, a, w, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
// Switch_key_precompute (polynomial.c:1224-1239, 1337-1343): every digit of d_in raised to level+K limbs.
// h_ext[d] = output polynomial of digit d (separate blocks: the rt_ant shim hands them to the caller's polynomials without a copy)
static int modup_digits_to(acehip_ctx* c, uint64_t* const* h_ext, const uint64_t* in, uint32_t level, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_modup_digits: bad level");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const size_t N = hp.N;
  const u32 nd = plan->nd;
  if (nd > 8) return fail(ACEHIP_EINVAL, "acehip_modup_digits: more than 8 digits");
  PtrTab8 outz;
  for (u32 d = 0; d < nd; ++d) {
    if (!h_ext[d]) return fail(ACEHIP_EINVAL, "acehip_modup_digits: null output");
    outz.p[d] = h_ext[d];
    dbg_touch(h_ext[d], (size_t)(level + hp.K) * N);
  }
  dbg_touch(in, (size_t)level * N);
  u64* coef = c->ws;
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs) {
    if (dc.logN == 16) {
      NttFuse fi;
      fi.src0 = in;
      fi.inv_scale = plan->inv_up;  // (Q_d/q_i)^-1 folded into the last inverse stage
      launch_ntt_fused(dc, coef, hp.L, 0, level, true, s, 0, 1, 0, 0, fi);
    } else {
      copy_limbs_dc(dc, coef, in, level, 0, s);
      launch_ntt(dc, coef, hp.L, 0, level, true, s);
    }
  }
  if (sharded(c)) {  // SURVEY 8e collective 1: the source limbs of every digit's conversion come from their owners
    std::vector<XItem> x;
    for (u32 i = 0; i < level; ++i) x.push_back(XItem{coef + (size_t)i * N, i % c->sh_world});
    if (int e = shard_exchange(c, x.data(), x.size(), s)) return e;
  }
  const u32 n_ext_rows = level + hp.K - std::min(hp.alpha, level - hp.alpha * (nd - 1));
  for (const DevCtx& dc : dcs) {
    if (conv_fusable_up(c, level)) {  // the conversions ride in the first pass of the NTT
      NttFuse fc;
      fc.conv = plan->d_descs;
      fc.conv_step = 1;
      fc.conv_max_in = std::min(hp.alpha, level);
      fc.conv_src = coef;
      fc.conv_src_stride = 0;
      for (u32 d = 0; d < nd; ++d) fc.polyz[d] = outz.p[d];
      launch_ntt_fused(dc, outz.p[0], level, 0, n_ext_rows, false, s, 0, nd, 0, hp.alpha, fc);
    } else {
      launch_base_conv_batch(dc, outz.p[0], 0, coef, 0, plan->d_descs, 1, nd, plan->max_nc, s, hp.alpha, outz, plan->mfma_up);
      if (dc.logN == 16) {
        NttFuse fz;
        for (u32 d = 0; d < nd; ++d) fz.polyz[d] = outz.p[d];
        launch_ntt_fused(dc, outz.p[0], level, 0, n_ext_rows, false, s, 0, nd, 0, hp.alpha, fz);
      } else {
        for (u32 d = 0; d < nd; ++d) {  // the generic passes address polynomials by stride: one digit at a time
          const u32 start = hp.alpha * d, n2 = std::min(hp.alpha, level - start);
          if (start) launch_ntt(dc, outz.p[d], level, 0, start, false, s);
          launch_ntt(dc, outz.p[d], level, start + n2, level + hp.K - (start + n2), false, s);
        }
      }
    }
    {  // digit limbs pass through (polynomial.c:1265-1273): `level` limb copies in one launch
      HwBatchArgs cp;
      u32 n_ops = 0;
      for (u32 pos = 0; pos < level; ++pos) {
        if (!dc_owns(dc, pos)) continue;
        if (n_ops == HW_BATCH_MAX) return fail(ACEHIP_EINVAL, "acehip_modup_digits: too many limbs");
        cp.seg_start[n_ops] = (uint16_t)n_ops;
        cp.op[n_ops++] = HwBatchOp{outz.p[pos / hp.alpha] + (size_t)pos * N, in + (size_t)pos * N, nullptr, HW_OP_COPY, 0};
      }
      cp.seg_start[n_ops] = (uint16_t)n_ops;
      launch_hw_batch_ew(dc, cp, n_ops, s);
    }
  }
  stat(ST_MODUP, nd, 8ull * N * (level + (u64)nd * (level + hp.K)));
  return post_launch();
}
int acehip_modup_digits(acehip_ctx* c, uint64_t* ext, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (!c || level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_modup_digits: bad level");
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  uint64_t* tab[8];
  const u32 nd = c->hp.num_decomp(level);
  if (nd > 8) return fail(ACEHIP_EINVAL, "acehip_modup_digits: more than 8 digits");
  for (u32 d = 0; d < nd; ++d) tab[d] = ext + d * E;
  return for_replica_chunks(c, [&] { return modup_digits_to(c, tab, in, level, s); });
}
int acehip_modup_digits_to(acehip_ctx* c, uint64_t* const* h_ext, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (!c || !h_ext) return fail(ACEHIP_EINVAL, "acehip_modup_digits_to: null argument");
  return for_replica_chunks(c, [&] { return modup_digits_to(c, h_ext, in, level, s); });
}
// Fast_switch_key_ext (ckks_evaluator.c:418-460): acc{0,1} = sum_d key{0,1}[d] * ext[d] over level+K limbs, no ModDown
int acehip_key_inner_product(acehip_ctx* c, uint64_t* acc0, uint64_t* acc1, const uint64_t* key, const uint64_t* ext, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_inner_product: bad level");
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  for (const DevCtx& dc : launch_dcs(c)) launch_key_mac_fused(dc, acc0, acc1, key, ext, E, nullptr, level, c->hp.num_decomp(level), c->hp.alpha, (hipStream_t)s);
  stat(ST_KEYMAC, 1, 8ull * E * (3ull * c->hp.num_decomp(level) + 2));
  return post_launch();
}

int acehip_key_inner_product_add(acehip_ctx* c, uint64_t* acc0, uint64_t* acc1, const uint64_t* key, const uint64_t* ext, uint32_t level,
                                 const uint64_t* add0, const uint64_t* h_scalars, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: bad level");
  if (!add0 || !h_scalars) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: null addend");
  if (level > 64) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: at most 64 q-limbs");
  LimbConsts w{};
  for (u32 i = 0; i < level; ++i) {
    if (h_scalars[i] >= c->hp.primes[i].q) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: scalar is not a residue of its prime");
    w.w[i] = h_scalars[i];
  }
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  for (const DevCtx& dc : launch_dcs(c)) launch_key_mac_fused(dc, acc0, acc1, key, ext, E, nullptr, level, c->hp.num_decomp(level), c->hp.alpha, (hipStream_t)s, add0, &w);
  stat(ST_KEYMAC, 1, 8ull * E * (3ull * c->hp.num_decomp(level) + 2) + 8ull * level * c->hp.N);
  return post_launch();
}

// The hoisted rotations of Rotate_iteration (ckks_bootstrap_context.c:1276-1290): n key inner products over the same raised digits, each as
// acehip_key_inner_product[_add] would form it, in one pass over the digits
int acehip_key_inner_products(acehip_ctx* c, uint64_t* const* h_acc0, uint64_t* const* h_acc1, const uint64_t* const* h_keys, uint32_t n_keys,
                              const uint64_t* ext, uint32_t level, const uint64_t* add0, const uint64_t* h_scalars, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: bad level");
  if (!h_acc0 || !h_acc1 || !h_keys || !ext || n_keys == 0) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: null argument");
  if ((add0 != nullptr) != (h_scalars != nullptr)) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: addend and scalars go together");
  if (add0 && level > 64) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: at most 64 q-limbs with an addend");
  for (u32 j = 0; j < n_keys; ++j)
    if (!h_acc0[j] || !h_acc1[j] || !h_keys[j] || h_acc0[j] == h_acc1[j]) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: bad output / key of a rotation");
  LimbConsts w{};
  if (add0)
    for (u32 i = 0; i < level; ++i) {
      if (h_scalars[i] >= c->hp.primes[i].q) return fail(ACEHIP_EINVAL, "acehip_key_inner_products: scalar is not a residue of its prime");
      w.w[i] = h_scalars[i];
    }
  const u32 nd = c->hp.num_decomp(level);
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  static const bool multi_on = [] { const char* e = getenv("ACEHIP_KEYMAC_MULTI"); return !e || atoi(e) != 0; }();
  for (u32 j0 = 0; j0 < n_keys; j0 += KEY_MULTI_MAX) {
    const u32 n = std::min(KEY_MULTI_MAX, n_keys - j0);
    for (const DevCtx& dc : launch_dcs(c)) {
      // the multi-key kernel reads a key part ONCE for all images of the launch (keys are shared: every caller allocates them outside the
      // replicated arena).  A key set INSIDE the arena differs per replica: those launches take the single-key kernels, which decide
      // per key (launch_key_mac_fused: key_shared)
      bool keys_shared = true;
      if (dc.nrep > 1)
        for (u32 j = j0; j < j0 + n; ++j) keys_shared = keys_shared && !((u64)h_keys[j] - dc.rep_lo < dc.rep_span);
      if (multi_on && nd <= 4 && n > 1 && keys_shared) {
        launch_key_mac_multi(dc, h_acc0 + j0, h_acc1 + j0, h_keys + j0, n, ext, E, level, nd, (hipStream_t)s, add0, add0 ? &w : nullptr);
      } else {  // one rotation (or more digits than the registers hold): the single-key kernels
        for (u32 j = j0; j < j0 + n; ++j)
          launch_key_mac_fused(dc, h_acc0[j], h_acc1[j], h_keys[j], ext, E, nullptr, level, nd, c->hp.alpha, (hipStream_t)s, add0, add0 ? &w : nullptr);
      }
    }
  }
  for (u32 j = 0; j < n_keys; ++j) stat(ST_KEYMAC, 1, 8ull * E * (3ull * nd + 2) + (add0 ? 8ull * level * c->hp.N : 0));
  return post_launch();
}

// Rotate_iteration's inner loop (ckks_bootstrap_context.c:1326-1341): out_i = sum_j rot_j (*) pt_{i,j} in the PQ basis
int acehip_bsgs_inner(acehip_ctx* c, uint64_t* const* out0, uint64_t* const* out1, const uint64_t* const* in0, const uint64_t* const* in1,
                      const uint64_t* const* pt, uint32_t g, uint32_t b, uint32_t pt_q_limbs, uint32_t level, acehip_stream s) {
  return acehip_bsgs_inner_rot(c, out0, out1, in0, in1, nullptr, pt, g, b, pt_q_limbs, level, s);
}

int acehip_bsgs_inner_rot(acehip_ctx* c, uint64_t* const* out0, uint64_t* const* out1, const uint64_t* const* in0, const uint64_t* const* in1,
                          const uint32_t* in_auto, const uint64_t* const* pt, uint32_t g, uint32_t b, uint32_t pt_q_limbs, uint32_t level,
                          acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || pt_q_limbs < level || pt_q_limbs > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: bad level");
  if (g == 0 || b == 0 || g > BSGS_MAX_G || b > BSGS_MAX_B || g * b > BSGS_MAX_PT) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: g, b out of range");
  BsgsArgs a{};
  u32 n_pt = 0;
  for (u32 j = 0; j < g; ++j) {
    if (!in0[j] || !in1[j]) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: null input");
    a.in0[j] = in0[j];
    a.in1[j] = in1[j];
    a.in_auto[j] = in_auto ? in_auto[j] : 0;
    if (a.in_auto[j] != 0 && (a.in_auto[j] % 2 == 0 || a.in_auto[j] >= 2 * c->hp.N))
      return fail(ACEHIP_EINVAL, "acehip_bsgs_inner_rot: automorphism index must be odd and below 2N");
    for (u32 i = 0; i < b && a.in_auto[j] != 0; ++i)  // a gathered input is read at other lanes' positions: it cannot be an output
      if (out0[i] == in0[j] || out0[i] == in1[j] || out1[i] == in0[j] || out1[i] == in1[j])
        return fail(ACEHIP_EINVAL, "acehip_bsgs_inner_rot: a rotated input aliases an output");
  }
  for (u32 i = 0; i < b; ++i) {
    if (!out0[i] || !out1[i]) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: null output");
    a.out0[i] = out0[i];
    a.out1[i] = out1[i];
    for (u32 j = 0; j < g; ++j) {
      a.pt[i * g + j] = pt[i * g + j];
      n_pt += pt[i * g + j] != nullptr;
    }
  }
  a.g = g;
  a.b = b;
  a.pt_q_alloc = pt_q_limbs;
  for (const DevCtx& dc : launch_dcs(c)) launch_bsgs_inner(dc, a, level, (hipStream_t)s);
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  stat(ST_EW, n_pt, 8ull * E * (2ull * g + n_pt + 2ull * b));
  stat(ST_EW_MUL, 2ull * n_pt * (level + c->hp.K), 2ull * n_pt * 24ull * E);  // (c0 and c1 of every product: limb-multiplications)
  acehip_stat_slots()[ST_EW_MUL].calls--;
  return post_launch();
}

}  // extern "C"
