/* model_main.c -- driver for an ACE-generated model source on a synthetic input.
 *
 * Our own main (the reference's dataset harness needs the CIFAR binaries, which are not in the tree:
 * rtlib/ant/dataset/resnet_cifar.main.inc).  The generated source is included UNCHANGED from where it
 * lies (-DMODEL_INC='"<path>.inc"'); see `make -C oracle models`.
 *   usage: model <n_images> [c h w]      env: ACEHIP_RT_DATA_SYNTH=1 for synthetic weights
 *   MODEL_ENC_SEED=<s>: image i is encrypted with the randomness of seed s + i (Acehip_rt_seed_encryptor), whatever batch or
 *   stream carries it; with ACEHIP_SEED that makes every output ciphertext a function of (seeds, image index) alone.  The same
 *   source is built against the REFERENCE rtlib with tests/c/gen_parity_ref.c, which serves the same call on the CPU.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>

#include "common/rtlib.h"

void Acehip_rt_seed_encryptor(uint64_t seed); /* include/rt_ant/rt_api.h (tests/c/gen_parity_ref.c in the reference build) */
void Acehip_rt_set_batch(uint32_t n_images);
void Acehip_rt_select_image(uint32_t k);

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char* argv[]) {
  int    n_images = argc > 1 ? atoi(argv[1]) : 1;
  size_t c = argc > 4 ? atoi(argv[2]) : 3, h = argc > 4 ? atoi(argv[3]) : 32, w = argc > 4 ? atoi(argv[4]) : 32;
  /* MODEL_BATCH=B: B images per Run_main_graph (Acehip_rt_set_batch, include/rt_ant/rt_api.h) -- the loop below is the
   * reference's image loop (rtlib/ant/dataset/resnet_cifar.main.inc:77-116) with its body split at the batch boundary */
  int    batch = getenv("MODEL_BATCH") ? atoi(getenv("MODEL_BATCH")) : 1;
  double t0 = now_s();
  Prepare_context();
  double t1 = now_s();
  printf("[MODEL] Prepare_context: %.3f s\n", t1 - t0);
  if (batch < 1) batch = 1;
  if (batch > 1) Acehip_rt_set_batch((uint32_t)batch);
  const char*        enc_seed = getenv("MODEL_ENC_SEED");
  unsigned long long z = 1;
  for (int img0 = 0; img0 < n_images; img0 += batch) {
    const int nb = n_images - img0 < batch ? n_images - img0 : batch;
    double    a  = now_s();
    for (int k = 0; k < batch; ++k) { /* (a short last batch repeats its last image: every slot of the batch holds an input) */
      TENSOR* in = Alloc_tensor(1, c, h, w, NULL);
      unsigned long long zk = z;
      for (size_t i = 0; i < c * h * w; ++i) { /* U(-1,1), seed 1 */
        zk ^= zk << 13; zk ^= zk >> 7; zk ^= zk << 17;
        in->_vals[i] = (double)(zk >> 11) / 9007199254740992.0 * 2.0 - 1.0;
      }
      if (k < nb) z = zk;
      if (batch > 1) Acehip_rt_select_image((uint32_t)k);
      if (enc_seed) Acehip_rt_seed_encryptor(strtoull(enc_seed, NULL, 10) + (unsigned long long)(img0 + (k < nb ? k : nb - 1)));
      Prepare_input(in, "input");
      Free_tensor(in);
    }
    double b = now_s();
    Run_main_graph();
    double e0 = now_s();
    for (int k = 0; k < batch; ++k) {
      if (batch > 1) Acehip_rt_select_image((uint32_t)k);
      double* out = Handle_output("output");
      if (k < nb) {
        printf("[MODEL] image %d: encrypt %.3f s, Main_graph+decrypt %.3f s, logits:", img0 + k, (b - a) / batch, (now_s() - b) / batch);
        for (int i = 0; i < 10; ++i) printf(" %.4f", out[i]);
        printf("\n[MODEL] image %d logits9:", img0 + k);
        for (int i = 0; i < 10; ++i) printf(" %.9f", out[i]);
        printf("\n");
      }
      free(out);
    }
    (void)e0;
  }
  Finalize_context();
  printf("[MODEL] total %.3f s\n", now_s() - t0);
  return 0;
}

/* The generated source hard-codes the weight file path of the machine it was compiled on
 * (/app/release/....msg); rename its Get_rt_data_info and supply one that reads MODEL_DATA_FILE. */
#define Get_rt_data_info Generated_get_rt_data_info
#include MODEL_INC
#undef Get_rt_data_info
RT_DATA_INFO* Get_rt_data_info() {
  static RT_DATA_INFO info;
  RT_DATA_INFO*       gen = Generated_get_rt_data_info();
  if (gen == NULL) return NULL;
  info               = *gen;
  const char* f      = getenv("MODEL_DATA_FILE");
  if (f != NULL) info._file_name = f;
  return &info;
}
