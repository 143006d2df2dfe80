/* rt_seal/rt_seal.h -- lets a program that ACE generated for the CKKS-level provider interface (-P2C:lib=seal:
 * `#include "rt_seal/rt_seal.h"`, reference rtlib/include/rt_seal/rt_seal.h:19-95) compile UNCHANGED against the MI355X
 * runtime: the same operator names and argument meaning, implemented by libFHErt_ant on the GPU.  See
 * rt_acehip/rt_acehip.h.  (The reference's rtlib/seal/example/eg_rtseal_*.cxx are built this way by `make -C workloads
 * provider` and run by tests/test_gpu_dropin.py.) */
#ifndef ACEHIP_RT_SEAL_COMPAT_H
#define ACEHIP_RT_SEAL_COMPAT_H
#include "rt_acehip/rt_acehip.h"
#endif
