/* rt_ant/rt_api.h -- ciphertext IO of the generated Main_graph (reference rt_ant/rt_api.h:17-20,
 * src/rtlib/rtlib.c:74-87). */
#ifndef ACEHIP_RT_ANT_RT_API_H
#define ACEHIP_RT_ANT_RT_API_H
#ifdef __cplusplus
extern "C" {
#endif
CIPHERTEXT Get_input_data(const char* name, size_t idx);
void       Set_output_data(const char* name, size_t idx, CIPHER data);
/* Extension (not in the reference): Coeffs() pointers are HBM addresses and per-limb Hw_* calls are executed
 * lazily in batches.  Code that touches that memory itself (HIP / acehip_* calls on the raw pointers) calls
 * this first: it submits everything still queued and waits for the device. */
void       Acehip_rt_sync(void);
/* Extension: fingerprint of the sources this library was built from (16 hex digits, embedded by the build); Prepare_context aborts
 * when it differs from libacehip's acehip_source_fingerprint() -- the two libraries of one build always agree. */
const char* acehip_rt_source_fingerprint(void);
/* Extension (test hook): the ChaCha20 block function (RFC 8439 2.3) behind the runtime's random streams -- 8 key words, block
 * counter, 3 nonce words -> 16 output words.  Keys and encryption randomness are ChaCha20 streams of a 256-bit master key from
 * getrandom(2) (the reference: BLAKE2Xb over /dev/urandom, src/util/prng.c:33-69); ACEHIP_SEED selects a reproducible TEST mode. */
void        acehip_rt_debug_chacha20_block(const uint32_t* key, uint32_t counter, const uint32_t* nonce, uint32_t* out);
/* Extension: a thread other than the one that called Prepare_context attaches to that context on its first API
 * call (shared keys; own scratch, pool, queue, HIP stream); before it ends it may give those back. */
void       Acehip_rt_thread_release(void);
/* Extension: weight-plaintext prefetch.  The runtime records the Pt_from_msg calls of a thread's first input and, for later
 * inputs, encodes the plaintexts of upcoming calls in batches ahead of their use (ACEHIP_PT_PREFETCH=<batch>, default 8,
 * 0 turns it off); a call that differs from the record falls back to a direct encode.  Prepare_input marks the start of an
 * input; a main() that feeds ciphertexts by other means calls Acehip_rt_next_input() there instead.
 * Acehip_rt_prefetched_count() = plaintexts this thread received from the prefetcher so far (statistics, tests). */
void       Acehip_rt_next_input(void);
size_t     Acehip_rt_prefetched_count(void);
/* Extension: the calling thread's ENCRYPTION randomness (the v, e1, e2 of Prepare_input / Encrypt) restarts from `seed`.  Keys never
 * draw from that stream: each key has a generator of its own derived from ACEHIP_SEED and the key's identity (csrc/rt/rt_context.cpp),
 * so ACEHIP_SEED + this call make the ciphertext of an input -- and with it every output of Main_graph -- reproducible whichever
 * thread, stream or image batch carries it (tests/test_gpu_gen_parity.py, bench.py "verified"). */
void       Acehip_rt_seed_encryptor(uint64_t seed);
/* Extension: the calling thread's next Set_output_data also writes its ciphertext to <prefix>.<image> (ACEHCT01; one file per image
 * of the batch).  One shot, per thread.  (ACEHIP_DUMP_OUTPUT=<prefix> in the environment does the same for every call of every thread:
 * <prefix>.<call>.<image>.) */
void       Acehip_rt_dump_next_output(const char* prefix);
/* Extension: on-disk containers (the reference has none).  All return 0 or a negative code
 * (-1 cannot open, -2 truncated / wrong magic, -3 written for other CKKS parameters).
 *   "ACEHCT01" ciphertext / plaintext: u32 n_polys, N, level, num_p, is_ntt, slots, sf_degree, 0; f64 scaling_factor;
 *              per polynomial `level` q-limbs then `num_p` p-limbs of N u64 residues
 *   "ACEHKEY1" key set (secret, public, relinearisation and every automorphism key + the rotation map); layout in
 *              csrc/rt/rt_serial.cpp.  ACEHIP_KEYS_FILE=<path> makes Prepare_context load it when it exists and
 *              Finalize_context write it when it does not; ACEHIP_KEYS_STRICT=1 turns a key missing from a loaded set
 *              into an error instead of generating a fresh one. */
int        Acehip_rt_save_ciph(const char* path, CIPHER ciph);
int        Acehip_rt_save_ciph3(const char* path, CIPHER3 ciph);
int        Acehip_rt_save_plain(const char* path, PLAIN plain);
int        Acehip_rt_load_ciph(CIPHER ciph, const char* path);   /* frees what ciph held, allocates from the pool */
int        Acehip_rt_load_ciph3(CIPHER3 ciph, const char* path);
int        Acehip_rt_load_plain(PLAIN plain, const char* path);
int        Acehip_rt_save_keys(const char* path);
int        Acehip_rt_load_keys(const char* path);                 /* replaces the keys of the prepared context */
/* The key set without the secret key (header flag): what a client hands to the party that only runs Main_graph.  A context
 * that loaded such a set evaluates but cannot decrypt, and cannot make further rotation keys.  Key files get mode 0600. */
int        Acehip_rt_save_eval_keys(const char* path);
/* Extension: image batches -- the GPU form of the reference's image-parallel loop (rtlib/ant/dataset/resnet_cifar.main.inc:77-116:
 * one OpenMP thread per image on shared keys and weights).  After Acehip_rt_set_batch(B) (or with ACEHIP_BATCH=B in the
 * environment) every launch of this thread carries B images: each has its own copy of every ciphertext buffer, while switch keys,
 * twiddles, bootstrap tables and the encoded weight plaintexts are read / produced once per batch.  Per batch:
 *     for (k = 0; k < B; k++) { Acehip_rt_select_image(k); Prepare_input(tensor_k, "input"); }
 *     Run_main_graph();                                     // once
 *     for (k = 0; k < B; k++) { Acehip_rt_select_image(k); out_k = Handle_output("output"); }
 * Results are bit-identical, image by image, to B separate runs with the same keys and encryption randomness.  Call
 * Acehip_rt_set_batch after Prepare_context and before the thread's first Prepare_input.  Acehip_rt_save_ciph / load_ciph,
 * Get_msg & co. address the selected image. */
void       Acehip_rt_set_batch(uint32_t n_images);
uint32_t   Acehip_rt_batch(void);
void       Acehip_rt_select_image(uint32_t k);
/* Extension: limb-sharded execution (BASELINE configs[4]): the RNS limbs of every polynomial are spread over the ranks
 * (limb gi on rank gi % world), every rank runs the SAME program (SPMD) and the limbs meet -- RCCL broadcasts over xGMI --
 * only where the algorithm needs them together: Decomp_modup (polynomial.c:1241-1335), Mod_down (:928-967), Rescale
 * (:1097-1163), the ModRaise of Bootstrap (ckks_bootstrap_context.c:1527-1551) and decode.  Nothing changes in the program.
 *   ACEHIP_SHARD=1      the processes of a launcher (RANK / WORLD_SIZE / LOCAL_RANK: torchrun, mpirun) are the ranks, one GPU
 *                       each; all need the same ACEHIP_SEED or ACEHIP_KEYS_FILE (else the seed is taken from the RCCL id)
 *   ACEHIP_SHARD_SIM=G  G ranks simulated on one GPU (tests): exchanges are device copies between per-rank buffers
 * Every rank's Handle_output returns the full result.  Acehip_rt_shard_traffic: bytes this process received through
 * exchanges so far; steps_limbs[0] = exchange steps, [1] = limbs. */
uint32_t   Acehip_rt_shard_world(void);
uint32_t   Acehip_rt_shard_rank(void);
uint64_t   Acehip_rt_shard_traffic(uint64_t* steps_limbs, int reset);
#ifdef __cplusplus
}
#endif
#endif
