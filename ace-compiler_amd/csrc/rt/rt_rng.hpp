// rt_rng.hpp -- the randomness behind key generation and encryption of the rt_ant shim.
//
// Reference: a BLAKE2Xb counter PRNG seeded from /dev/urandom (src/util/prng.c:33-69, random_sample.c:20-24).  Here: the ChaCha20
// block function (RFC 8439 section 2.3) as a deterministic random bit generator -- 256-bit key from getrandom(2), 96-bit nonce =
// the stream's identity, 32-bit block counter extended into the nonce's last word -- one independent stream per KEY identity and per
// ENCRYPTING thread, all derived from one 256-bit master key that never leaves the process (limb-sharded ranks share it through the
// 0600 rendezvous record).  The device-side uniform sampler of the public `a` polynomials is keyed from the same streams
// (acehip_sample_uniform_keyed: ChaCha20 blocks on the GPU).
//
// ACEHIP_SEED / Acehip_rt_seed_encryptor select the TEST mode instead: std::mt19937_64 seeded with 64 bits, exactly the derivation the
// committed fixtures were made with (tests/c/gen_parity_ref.c restates it for the reference side).  Reproducible, therefore not secret:
// for tests and benchmarks only.
#pragma once
#include <cstdint>
#include <cstring>
#include <random>

namespace rt {

struct ChaCha20 {
  uint32_t key[8] = {}, nonce[3] = {};
  uint32_t counter = 0;
  uint32_t buf[16];
  int pos = 16;  // words of buf already handed out
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static inline void qr(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    a += b; d ^= a; d = rotl(d, 16);
    c += d; b ^= c; b = rotl(b, 12);
    a += b; d ^= a; d = rotl(d, 8);
    c += d; b ^= c; b = rotl(b, 7);
  }
  // RFC 8439 2.3: state = constants | key | counter | nonce; 20 rounds; add the input state
  static void block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      counter, nonce[0], nonce[1], nonce[2]};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int i = 0; i < 10; ++i) {
      qr(x[0], x[4], x[8], x[12]); qr(x[1], x[5], x[9], x[13]); qr(x[2], x[6], x[10], x[14]); qr(x[3], x[7], x[11], x[15]);
      qr(x[0], x[5], x[10], x[15]); qr(x[1], x[6], x[11], x[12]); qr(x[2], x[7], x[8], x[13]); qr(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
  }
  uint64_t next() {
    if (pos >= 16) {
      block(key, counter, nonce, buf);
      if (++counter == 0) ++nonce[2];  // 2^32 blocks = 256 GiB of one stream: carry into the word the stream identities leave at zero
      pos = 0;
    }
    const uint64_t v = (uint64_t)buf[pos] | ((uint64_t)buf[pos + 1] << 32);
    pos += 2;
    return v;
  }
};

// What the samplers draw from.  mode 0: not initialised (drawing aborts), 1: ChaCha20 stream, 2: mt19937_64 (test mode)
class Rng {
 public:
  uint64_t operator()() {
    if (mode_ == 1) return cc_.next();
    if (mode_ == 2) return mt_();
    abort_unseeded();
    return 0;
  }
  // test mode: the generator the fixtures were made with
  void seed(uint64_t s) {
    mt_.seed(s);
    mode_ = 2;
  }
  // a ChaCha20 stream of the master key: domain separates the uses ('K' keys, 'E' encrypting threads), id names the stream
  void key(const uint32_t master[8], uint32_t domain, uint64_t id) {
    memcpy(cc_.key, master, sizeof cc_.key);
    cc_.nonce[0] = (uint32_t)id;
    cc_.nonce[1] = (uint32_t)(id >> 32);
    cc_.nonce[2] = domain << 24;  // (the low 24 bits take the carry of the block counter)
    cc_.counter = 0;
    cc_.pos = 16;
    mode_ = 1;
  }
  bool test_mode() const { return mode_ == 2; }
  // 256 fresh bits of this stream: the key of a device-side sampler launch
  void draw_key(uint32_t out[8]) {
    for (int i = 0; i < 4; ++i) {
      const uint64_t v = (*this)();
      out[2 * i] = (uint32_t)v;
      out[2 * i + 1] = (uint32_t)(v >> 32);
    }
  }

 private:
  static void abort_unseeded();
  int mode_ = 0;
  ChaCha20 cc_;
  std::mt19937_64 mt_;
};

}  // namespace rt
