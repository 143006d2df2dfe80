/* model_main_omp.c -- the structure of the reference's model main (rtlib/ant/dataset/resnet_cifar.main.inc:77-116)
 * without its CIFAR reader: Prepare_context once, then `#pragma omp parallel for` over images with Prepare_input /
 * Run_main_graph / Handle_output per thread, Finalize_context once.  Worker threads never call Prepare_context: they
 * attach to the prepared context on first use.  Build: gcc -fopenmp -DMODEL_INC='"....onnx.inc"' ...
 * MODEL_ENC_SEED=<s>: every iteration encrypts with the randomness of seed s (the same image: the same ciphertext);
 * MODEL_DUMP_PREFIX=<p>: iteration i writes its output ciphertext to <p>.img<i>.0 (Acehip_rt_dump_next_output). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "common/rtlib.h"

void Acehip_rt_seed_encryptor(uint64_t seed); /* include/rt_ant/rt_api.h */
void Acehip_rt_dump_next_output(const char* prefix);

static double now_s() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char* argv[]) {
  int n_images = argc > 1 ? atoi(argv[1]) : 4;
  Prepare_context();
  double t0 = now_s();
  double first[10] = {0};
  int bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1)
  for (int img = 0; img < n_images; ++img) {
    TENSOR* in = Alloc_tensor(1, 3, 32, 32, NULL);
    unsigned long long z = 1;  /* the same image on every iteration: all logits must agree */
    for (size_t i = 0; i < 3 * 32 * 32; ++i) {
      z ^= z << 13; z ^= z >> 7; z ^= z << 17;
      in->_vals[i] = (double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    }
    if (getenv("MODEL_ENC_SEED")) Acehip_rt_seed_encryptor(strtoull(getenv("MODEL_ENC_SEED"), NULL, 10));
    Prepare_input(in, "input");
    Free_tensor(in);
    if (getenv("MODEL_DUMP_PREFIX")) {
      char prefix[1024];
      snprintf(prefix, sizeof prefix, "%s.img%d", getenv("MODEL_DUMP_PREFIX"), img);
      Acehip_rt_dump_next_output(prefix);
    }
    Run_main_graph();
    double* out = Handle_output("output");
#pragma omp critical
    {
      printf("[MODEL] image %d: logits:", img);
      for (int i = 0; i < 10; ++i) printf(" %.4f", out[i]);
      printf("\n");
      if (img == 0) for (int i = 0; i < 10; ++i) first[i] = out[i];
    }
    free(out);
  }
  double dt = now_s() - t0;
  printf("[MODEL] %d images in %.3f s = %.3f images/s\n", n_images, dt, n_images / dt);
  Finalize_context();
  return bad;
}

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
