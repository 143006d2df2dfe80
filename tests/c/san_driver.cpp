// san_driver.cpp -- host-side logic of libacehip under sanitizers (test infrastructure; CPU only, no GPU needed).
// Built by `make -C oracle asan` (-fsanitize=address,undefined) and `make -C oracle tsan` (-fsanitize=thread) from the
// product sources; exercises what runs on the host: parameter/table generation (host_params.cpp), context life cycle,
// automorphism tables, the ModUp/ModDown constant caches and the batch planner of acehip_hw_batch (api_hw_batch.cpp) -- from one
// thread and from four threads, on separate contexts and on a shared one (the shim gives every image thread its own
// context, but the C ABI documents contexts as usable from any thread).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "acehip.h"

static int g_fail = 0;
#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
      ++g_fail;                                                   \
    }                                                             \
  } while (0)

static void plan_random(acehip_ctx* c, uint32_t N, uint32_t limbs, unsigned seed, int rounds) {
  std::mt19937 rng(seed);
  const uint64_t base = 0x10000000ull, span = (uint64_t)N * 8;
  for (int r = 0; r < rounds; ++r) {
    const size_t n_ops = 8 + rng() % 120;
    std::vector<acehip_hw_op> prog(n_ops);
    for (auto& o : prog) {
      o.op = rng() % 9;
      o.prime_gi = rng() % limbs;
      o.res = (uint64_t*)(base + (rng() % 24) * span);
      o.a = (const uint64_t*)(base + (rng() % 24) * span);
      o.b = (o.op == ACEHIP_HW_MULC || o.op == ACEHIP_HW_ADDC) ? (const void*)(uintptr_t)(rng() % 1000)
            : (o.op == ACEHIP_HW_ROTATE)                       ? (const void*)acehip_auto_order(c, 5)  /* host ctx: may be null */
                                                               : (const void*)(base + (rng() % 24) * span);
      if (o.op == ACEHIP_HW_ROTATE && (o.b == nullptr || o.res == o.a)) o.op = ACEHIP_HW_COPY;
      if (o.op == ACEHIP_HW_COPY && o.res == o.a) o.op = ACEHIP_HW_ZERO;
      if (o.op == ACEHIP_HW_ZERO) o.a = nullptr;
    }
    const size_t cap = 4 * n_ops + 16;
    std::vector<acehip_hw_op> out(cap);
    std::vector<uint32_t> launch(cap), seg(cap);
    const long n = acehip_hw_batch_plan(c, prog.data(), n_ops, out.data(), launch.data(), seg.data(), cap, 0x7F0000000000ull);
    CHECK(n >= 0 && (size_t)n <= cap);
    // the same list with some of its limbs given up (acehip_hw_batch_discard): never more ops than without the hint
    std::vector<acehip_hw_range> dead;
    for (uint64_t l = 0; l < 24; l += 1 + rng() % 4) dead.push_back(acehip_hw_range{(const uint64_t*)(base + l * span), (size_t)N * (1 + rng() % 2)});
    for (size_t i = 1; i < dead.size(); ++i)
      if (dead[i - 1].ptr + dead[i - 1].words > dead[i].ptr) dead[i - 1].words = N;  // keep them disjoint
    const long nd = acehip_hw_batch_plan_discard(c, prog.data(), n_ops, dead.data(), dead.size(), out.data(), launch.data(), seg.data(), cap,
                                                 0x7F0000000000ull);
    CHECK(nd >= 0 && nd <= n);
  }
}

static void tables(acehip_ctx* c) {
  const uint32_t L = acehip_num_q(c), K = acehip_num_p(c), N = acehip_degree(c);
  std::vector<uint64_t> t(L + K);
  for (int what = 0; what < 4; ++what) CHECK(acehip_get_table(c, what, 0, t.data(), t.size()) >= 0);
  for (uint32_t level = 1; level <= L; ++level)
    for (uint32_t d = 0; d < acehip_num_decomp(c, level); ++d) {
      std::vector<uint64_t> hat_inv(64), hat_mod(64 * 128);
      std::vector<uint32_t> compl_(128);
      uint32_t nc = 0;
      CHECK(acehip_get_modup_tables(c, level, d, hat_inv.data(), compl_.data(), hat_mod.data(), &nc) > 0 && nc > 0);
    }
  std::vector<uint32_t> perm(N);
  for (int32_t rot : {1, -1, 3, (int32_t)(N / 4)}) CHECK(acehip_auto_order_host(c, acehip_auto_index(c, rot), perm.data()) == 0);
  CHECK(acehip_key_switch_bytes(c, L) > 0);
}

int main() {
  struct Set { uint32_t N, L, q0, sf, dnum; } sets[] = {{8, 4, 60, 56, 2}, {64, 7, 60, 51, 3}, {1024, 7, 60, 51, 3}, {64, 40, 60, 50, 2}};
  for (auto& s : sets) {
    acehip_ctx* c = acehip_ctx_create_host(s.N, s.L, s.q0, s.sf, s.dnum);
    CHECK(c != nullptr);
    if (!c) continue;
    tables(c);
    plan_random(c, s.N, s.L, 1, 50);
    // every launch on a host-only context must fail cleanly (no CPU fallback)
    CHECK(acehip_ntt_forward(c, (uint64_t*)0x1000, s.L, 0, 1, nullptr) < 0);
    acehip_ctx_destroy(c);
  }
  CHECK(acehip_ctx_create_host(16, 70, 40, 30, 1) == nullptr);  // alpha > 64: refused
  CHECK(acehip_last_error() != nullptr && strlen(acehip_last_error()) > 0);
  // four threads, own contexts (creation shares nothing but the prime search and the error slot)
  {
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
      th.emplace_back([t] {
        acehip_ctx* c = acehip_ctx_create_host(64, 7, 60, 51, 3);
        if (c) {
          tables(c);
          plan_random(c, 64, 7, 100 + t, 40);
          acehip_ctx_destroy(c);
        }
      });
    for (auto& x : th) x.join();
  }
  // four threads, ONE context: table caches and plans are built under the context's lock
  {
    acehip_ctx* c = acehip_ctx_create_host(64, 7, 60, 51, 3);
    CHECK(c != nullptr);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t) th.emplace_back([c, t] { tables(c); plan_random(c, 64, 7, 200 + t, 40); });
    for (auto& x : th) x.join();
    acehip_ctx_destroy(c);
  }
  printf(g_fail ? "san_driver: %d checks failed\n" : "san_driver: OK\n", g_fail);
  return g_fail ? 1 : 0;
}
