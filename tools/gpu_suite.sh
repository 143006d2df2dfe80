#!/bin/bash
# the whole GPU suite: the long-standing tests first (-x), then image batches / simulated limb-sharded runs (all of them, no -x)
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_batch_shard.py > gpurun_out/$1_gpu_tests.log 2>&1
echo exit=$? >> gpurun_out/$1_gpu_tests.log
tail -3 gpurun_out/$1_gpu_tests.log
timeout -k 10 1000 python -m pytest tests/test_gpu_batch_shard.py -m gpu -q > gpurun_out/$1_gpu_batch_shard.log 2>&1
echo exit=$? >> gpurun_out/$1_gpu_batch_shard.log
tail -25 gpurun_out/$1_gpu_batch_shard.log
