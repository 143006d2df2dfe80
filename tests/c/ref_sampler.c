/* ref_sampler.c -- where does the REFERENCE runtime spend an image?  A sampling profiler for the reference-side executables
 * (`make -C oracle refgen`), test infrastructure and our own code: nothing of the product links it.
 *
 * The dev container has no perf / gdb.  setitimer(ITIMER_PROF) delivers SIGPROF at the kernel's timer rate (asked for 1 ms, ticks every 4 ms
 * here) of CPU time (user + system: a page fault taken inside calloc shows up at the faulting instruction); the handler stores the interrupted
 * program counter; the CPU time of the span is written along, so a sample's weight is known.  At the end the
 * counters are written as "<module path> <offset in module> <samples>" lines, which tools/ref_profile_report.py buckets into functions
 * with the module's own symbol table (`nm -n`, static functions included) and into the families bench.py's price_image() knows.
 *
 *   REF_SAMPLER_OUT=<file>   switches it on; gen_parity_ref.c starts it after the input is encrypted (end of Prepare_input) and stops
 *                            it when the first output ciphertext arrives (Set_output_data): exactly the span RTM_MAIN_GRAPH reports.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <ucontext.h>

#define SAMPLER_CAP (8u << 20) /* 8 M samples = 2.3 h at 1 kHz */
static uintptr_t*       Pc;
static volatile size_t  N_pc;
static volatile size_t  N_lost;
static int              Running;
static double           Cpu0; /* process CPU time at the start: ITIMER_PROF ticks with the kernel's HZ, not with the period asked for */
static double cpu_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void on_prof(int sig, siginfo_t* si, void* uc_) {
  (void)sig;
  (void)si;
  const ucontext_t* uc = (const ucontext_t*)uc_;
  const size_t      k  = N_pc;
  if (k < SAMPLER_CAP) {
    Pc[k] = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
    N_pc  = k + 1;
  } else {
    ++N_lost;
  }
}

void Ref_sampler_start(void) {
  if (Running || !getenv("REF_SAMPLER_OUT")) return;
  if (!Pc) Pc = (uintptr_t*)malloc(sizeof(uintptr_t) * SAMPLER_CAP);
  if (!Pc) return;
  memset(Pc, 0, sizeof(uintptr_t) * SAMPLER_CAP); /* touch the pages now, not inside the handler */
  N_pc = 0;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags     = SA_SIGINFO | SA_RESTART;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGPROF, &sa, NULL);
  Cpu0 = cpu_now();
  struct itimerval it = {{0, 1000}, {0, 1000}};
  setitimer(ITIMER_PROF, &it, NULL);
  Running = 1;
}

static int cmp_pc(const void* a, const void* b) {
  const uintptr_t x = *(const uintptr_t*)a, y = *(const uintptr_t*)b;
  return x < y ? -1 : x > y;
}

void Ref_sampler_stop(void) {
  if (!Running) return;
  struct itimerval off = {{0, 0}, {0, 0}};
  const double cpu_s = cpu_now() - Cpu0;
  setitimer(ITIMER_PROF, &off, NULL);
  signal(SIGPROF, SIG_IGN);
  Running = 0;
  const char* path = getenv("REF_SAMPLER_OUT");
  FILE*       f    = fopen(path, "w");
  if (!f) { perror(path); return; }
  const size_t n = N_pc;
  qsort(Pc, n, sizeof(uintptr_t), cmp_pc);
  fprintf(f, "# samples %zu lost %zu period_us 1000 cpu_s %.3f\n", n, (size_t)N_lost, cpu_s);
  for (size_t i = 0; i < n;) {
    size_t j = i;
    while (j < n && Pc[j] == Pc[i]) ++j;
    Dl_info info;
    if (dladdr((void*)Pc[i], &info) && info.dli_fname)
      fprintf(f, "%s %lx %zu %s\n", info.dli_fname, (unsigned long)(Pc[i] - (uintptr_t)info.dli_fbase), j - i, info.dli_sname ? info.dli_sname : "?");
    else
      fprintf(f, "? %lx %zu ?\n", (unsigned long)Pc[i], j - i);
    i = j;
  }
  fclose(f);
}
