#!/usr/bin/env python3
"""Throughput of W concurrent image streams in ONE process: W host threads attached to one rt_ant context
(shared keys; own scratch context, pool, queue and HIP stream each).  usage: ubench_streams.py W [images]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per image stream (see bench.py)
json_fd = os.dup(1)
os.dup2(2, 1)
import bench  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
IMAGES = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fhe, _ = bench.load_model_runtime(0)
fhe.Prepare_context()  # keys once; the image threads attach to this context
barrier = threading.Barrier(W + 1)
logits = [None] * W


def run(i):
    _, step = bench.load_model_runtime(0)   # same library handle; a step closure (image generator) per thread
    fhe.Prepare_context()                   # thread-local context
    step()                                  # warm-up image (lazy key generation, bootstrap precompute)
    barrier.wait()
    for _ in range(IMAGES):
        logits[i] = step()
    barrier.wait()
    fhe.Finalize_context()


threads = [threading.Thread(target=run, args=(i,)) for i in range(W)]
for t in threads:
    t.start()
barrier.wait()
t0 = time.perf_counter()
barrier.wait()
dt = time.perf_counter() - t0
for t in threads:
    t.join()
fhe.Finalize_context()
os.write(json_fd, ("%d streams: %d images in %.3f s = %.3f images/s (%.3f s per image per stream); logits[0..2] %s\n" % (
    W, W * IMAGES, dt, W * IMAGES / dt, dt / IMAGES, [[round(v, 4) for v in l[:3]] for l in logits])).encode())
