#!/bin/bash
# Throughput sensitivity of the headline workload to each kernel family: run on the GPU box after building
# an ablation library on the dev container:
#   . tools/exp_build.sh; exp_build -DACEHIP_ABLATION   (the flags are part of the library fingerprint: keep them exported for the runs)
#   python ace-compiler_amd/build.py --force        # restore the product library
#   gpurun -- 'bash tools/ablate.sh'
# $ACEHIP_ABLATE is the bit mask of families whose launches are skipped (kernels.hpp AblateFamily): results are wrong by
# construction, only the time of what is left is of interest; the product library ignores the variable.
export ACEHIP_BENCH_NO_VERIFY=1
cp gpurun_exp/libacehip_ablate.so ace-compiler_amd/lib/libacehip.so
for m in 0 1 2 4 8 16 32 64 128 255; do
  echo "ABLATE $m: $(ACEHIP_ABLATE=$m timeout 300 python bench.py --no-verify --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
