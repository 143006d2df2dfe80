/* mock_rccl.c -- TEST DOUBLE for librccl (test infrastructure, never shipped or measured).
 *
 * A GPU test box has ONE GPU and RCCL refuses two ranks on one device, so the multi-process form of limb-sharded execution
 * (ACEHIP_SHARD=1: csrc/api_shard.cpp shard_exchange -> grouped ncclBroadcast from the owning rank) could not run with more
 * than one rank there.  This library implements the handful of entry points the product dlopen()s -- same names, same
 * signatures (rccl.h) -- on top of POSIX shared memory, so that N processes that share one GPU can act as N ranks:
 *   ncclBroadcast(root)  root: stream sync, device -> shared slot, publish;   others: wait, shared slot -> device
 *   ncclAllGather        one such broadcast per rank and slot-sized chunk of its contribution, in rank order
 * Operations are matched by their sequence number on the communicator, exactly what RCCL requires of its callers (every rank
 * issues the same collectives in the same order with the same root and count): a rank that issues a different sequence makes
 * the test fail (root / count mismatch is checked) or time out.  Selected with ACEHIP_RCCL_LIB=<this .so>
 * (tests/test_gpu_batch_shard.py).  Everything is synchronous: a superset of the ordering RCCL gives on its stream.
 */
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define SLOTS 32
#define SLOT_BYTES (1u << 20) /* one limb of N = 2^17 */
#define WAIT_S 120.0

typedef struct {
  char internal[128];
} ncclUniqueId;

typedef struct {
  _Atomic uint32_t joined, left;
  _Atomic uint64_t published[SLOTS]; /* op + 1 once the slot holds the payload of operation `op` */
  _Atomic uint64_t acks[SLOTS];      /* readers that have finished with the slot, over all its uses */
  uint64_t count[SLOTS];
  int32_t root[SLOTS];
  char data[SLOTS][SLOT_BYTES];
} Shm;

typedef struct {
  Shm* shm;
  char name[128];
  int rank, world;
  uint64_t op; /* next operation number of this rank */
} Comm;

static const char* g_err = "mock rccl: ok";
static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static int fail(const char* m) {
  g_err = m;
  fprintf(stderr, "[mock rccl] %s\n", m);
  return 1; /* ncclUnhandledCudaError-ish: any non-zero */
}

const char* ncclGetErrorString(int e) { return e ? g_err : "no error"; }

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/acehip_mock_rccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
  int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return fail("shm_open(create) failed");
  if (ftruncate(fd, sizeof(Shm)) != 0) {
    close(fd);
    return fail("ftruncate failed");
  }
  close(fd); /* zero-filled: every counter starts at 0 */
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  Comm* c = calloc(1, sizeof *c);
  id.internal[sizeof id.internal - 1] = 0;
  snprintf(c->name, sizeof c->name, "%s", id.internal);
  const double t0 = now_s();
  int fd = -1;
  while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0) {
    if (now_s() - t0 > WAIT_S) return fail("shm_open(join) timed out");
    usleep(1000);
  }
  struct stat st;
  while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(Shm)) {
    if (now_s() - t0 > WAIT_S) return fail("shared segment never reached its size");
    usleep(1000);
  }
  c->shm = mmap(NULL, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->shm == MAP_FAILED) return fail("mmap failed");
  c->rank = rank;
  c->world = nranks;
  atomic_fetch_add(&c->shm->joined, 1);
  while (atomic_load(&c->shm->joined) < (uint32_t)nranks) {
    if (now_s() - t0 > WAIT_S) return fail("not every rank joined");
    usleep(200);
  }
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = comm;
  if (!c) return 0;
  if (atomic_fetch_add(&c->shm->left, 1) + 1 == (uint32_t)c->world) shm_unlink(c->name);
  munmap(c->shm, sizeof(Shm));
  free(c);
  return 0;
}

int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }

static size_t dtype_bytes(int dt) {
  switch (dt) {
    case 0: case 1: return 1;          /* ncclInt8, ncclUint8 */
    case 2: case 3: case 7: return 4;  /* ncclInt32, ncclUint32, ncclFloat32 */
    case 4: case 5: case 8: return 8;  /* ncclInt64, ncclUint64, ncclFloat64 */
    case 6: case 9: return 2;          /* ncclFloat16, ncclBfloat16 */
    default: return 0;
  }
}

/* one payload of at most a slot */
static int bcast_chunk(const void* sendbuff, void* recvbuff, size_t bytes, int root, void* comm, hipStream_t stream) {
  Comm* c = comm;
  Shm* s = c->shm;
  if (bytes == 0 || bytes > SLOT_BYTES) return fail("payload size not supported by the mock");
  if (root < 0 || root >= c->world) return fail("root out of range");
  const uint64_t op = c->op++;
  const unsigned slot = op % SLOTS;
  const uint64_t use = op / SLOTS;
  const double t0 = now_s();
  if (hipStreamSynchronize(stream) != hipSuccess) return fail("hipStreamSynchronize failed");
  if (c->rank == root) {
    /* every reader of the slot's previous uses must be done before it is overwritten */
    while (atomic_load(&s->acks[slot]) < use * (uint64_t)(c->world - 1)) {
      if (now_s() - t0 > WAIT_S) return fail("timed out waiting for a slot to drain (ranks issue different sequences?)");
      usleep(50);
    }
    if (hipMemcpy(s->data[slot], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("device -> shared copy failed");
    s->count[slot] = bytes;
    s->root[slot] = root;
    atomic_store(&s->published[slot], op + 1);
    if (recvbuff != sendbuff && hipMemcpy(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return fail("root copy failed");
  } else {
    while (atomic_load(&s->published[slot]) != op + 1) {
      if (now_s() - t0 > WAIT_S) return fail("timed out waiting for the root (ranks issue different sequences?)");
      usleep(50);
    }
    if (s->count[slot] != bytes || s->root[slot] != root) return fail("ranks disagree on the root or the size of an operation");
    if (hipMemcpy(recvbuff, s->data[slot], bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("shared -> device copy failed");
    atomic_fetch_add(&s->acks[slot], 1);
  }
  return 0;
}

/* any size: slot-sized pieces one after the other (every rank cuts the same count the same way) */
int ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, void* comm, hipStream_t stream) {
  const size_t bytes = count * dtype_bytes(datatype);
  if (bytes == 0) return fail("payload size not supported by the mock");
  for (size_t off = 0; off < bytes; off += SLOT_BYTES) {
    const size_t n = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
    const int e = bcast_chunk((const char*)sendbuff + off, (char*)recvbuff + off, n, root, comm, stream);
    if (e) return e;
  }
  return 0;
}

/* every rank's sendcount elements land at recvbuff + rank * sendcount on every rank (in place when sendbuff already points there):
 * the ranks' contributions one after the other, each cut into chunks a shared slot holds */
int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
  Comm* c = comm;
  const size_t bytes = sendcount * dtype_bytes(datatype);
  if (bytes == 0) return fail("payload size not supported by the mock");
  for (int r = 0; r < c->world; ++r)
    for (size_t off = 0; off < bytes; off += SLOT_BYTES) {
      const size_t n = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
      char* dst = (char*)recvbuff + (size_t)r * bytes + off;
      const void* src = r == c->rank ? (const char*)sendbuff + off : dst;
      const int e = bcast_chunk(src, dst, n, r, comm, stream);
      if (e) return e;
    }
  return 0;
}
