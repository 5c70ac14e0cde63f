// TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that lets TWO (or more) processes sharing ONE GPU run
// the engine's own gradient-exchange protocol (csrc/engine.cpp: plb_comm_* / reduce_piece / pieces_done) at world > 1.
// RCCL itself refuses two ranks on one device, and the test box has one GPU, so without this the piecewise exchange had
// only ever executed where ncclAllReduce is the identity (world 1).
//
// It exports exactly the seven symbols engine.cpp resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclAllReduce, ncclBroadcast, ncclGetVersion, ncclGetErrorString) and is selected with PLBERT_RCCL_LIB=<this .so>.
// Collectives are stream-ordered like the real ones: device -> pinned host copy, a host function on the stream that meets
// the other ranks in a POSIX shared-memory segment (sum in rank order: deterministic), pinned host -> device copy. Every
// collective also posts (kind, count) and checks that all ranks posted the same: a rank that issues a different
// sequence of collectives — the failure mode of a piecewise exchange — is reported through fake_rccl_errors() instead of
// silently summing mismatched ranges. Waits are bounded (FAKE_RCCL_TIMEOUT_S, default 120 s).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t kSlotBytes = 64u << 20;  // per rank: the largest single collective the tests issue (token head: 512 x 256 floats)
constexpr int kMaxRanks = 8;

struct Header {
  std::atomic<uint32_t> count;
  std::atomic<uint32_t> gen;
  std::atomic<uint32_t> errors;
  std::atomic<uint32_t> collectives;
  uint64_t posted_kind[kMaxRanks];
  uint64_t posted_count[kMaxRanks];
};

struct Comm {
  int rank = 0, world = 1;
  char name[64] = {0};
  Header* hdr = nullptr;
  char* slots = nullptr;  // world x kSlotBytes, after the header
  size_t map_bytes = 0;
  char* staging = nullptr;  // pinned host buffer for the device copies
};

std::atomic<uint32_t> g_local_errors{0};
double timeout_s() {
  const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

bool barrier(Comm* c) {
  Header* h = c->hdr;
  const uint32_t g = h->gen.load(std::memory_order_acquire);
  if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
    h->count.store(0, std::memory_order_relaxed);
    h->gen.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->gen.load(std::memory_order_acquire) == g) {
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      fprintf(stderr, "fake_rccl: rank %d timed out waiting for the other ranks (a rank issued fewer collectives?)\n", c->rank);
      h->errors.fetch_add(1);
      g_local_errors.fetch_add(1);
      return false;
    }
  }
  return true;
}

struct Op {
  Comm* c;
  int kind;  // 0 all-reduce sum f32, 1 broadcast
  int root;
  size_t count;
};

void run_op(void* arg) {  // on the stream, between the D2H and the H2D copy: no HIP calls in here
  Op* op = (Op*)arg;
  Comm* c = op->c;
  const size_t bytes = op->count * sizeof(float);
  char* mine = c->slots + (size_t)c->rank * kSlotBytes;
  c->hdr->posted_kind[c->rank] = (uint64_t)op->kind * 16 + (uint64_t)op->root;
  c->hdr->posted_count[c->rank] = op->count;
  if (op->kind == 0 || c->rank == op->root) memcpy(mine, c->staging, bytes);
  if (!barrier(c)) { delete op; return; }
  for (int r = 0; r < c->world; ++r)
    if (c->hdr->posted_kind[r] != c->hdr->posted_kind[c->rank] || c->hdr->posted_count[r] != op->count) {
      fprintf(stderr, "fake_rccl: rank %d posted (kind %llu, count %llu) but rank %d posted (kind %llu, count %llu)\n",
              c->rank, (unsigned long long)c->hdr->posted_kind[c->rank], (unsigned long long)op->count, r,
              (unsigned long long)c->hdr->posted_kind[r], (unsigned long long)c->hdr->posted_count[r]);
      c->hdr->errors.fetch_add(1);
      g_local_errors.fetch_add(1);
    }
  float* out = (float*)c->staging;
  if (op->kind == 0) {
    const float* s0 = (const float*)c->slots;
    for (size_t i = 0; i < op->count; ++i) out[i] = s0[i];
    for (int r = 1; r < c->world; ++r) {
      const float* sr = (const float*)(c->slots + (size_t)r * kSlotBytes);
      for (size_t i = 0; i < op->count; ++i) out[i] += sr[i];  // rank order: the same sum on every rank
    }
  } else {
    memcpy(out, c->slots + (size_t)op->root * kSlotBytes, bytes);
  }
  if (c->rank == 0) c->hdr->collectives.fetch_add(1);
  barrier(c);  // nobody re-fills a slot before everyone has read it
  delete op;
}

int collective(const void* send, void* recv, size_t count, int kind, int root, Comm* c, hipStream_t s) {
  if (!c) return 4;
  if (count * sizeof(float) > kSlotBytes) {
    fprintf(stderr, "fake_rccl: %zu floats exceed the %zu-byte slot\n", count, kSlotBytes);
    return 5;
  }
  if (count == 0) return 0;
  if (hipMemcpyAsync(c->staging, send, count * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
  if (hipLaunchHostFunc(s, run_op, new Op{c, kind, root, count}) != hipSuccess) return 1;
  if (hipMemcpyAsync(recv, c->staging, count * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) return 1;
  return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetVersion(int* v) { if (v) *v = 29999; return 0; }  // recognisable: no real RCCL reports this
const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "success";
    case 1: return "fake_rccl: HIP call failed";
    case 3: return "fake_rccl: shared-memory segment";
    case 4: return "fake_rccl: invalid argument";
    case 5: return "fake_rccl: collective larger than the slot";
    default: return "fake_rccl: error";
  }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return 4;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/plbfake_%d_%llx", (int)getpid(),
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
  Comm* c = new Comm;
  c->rank = rank; c->world = nranks;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->map_bytes = 4096 + (size_t)nranks * kSlotBytes;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return 3; }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (true) {  // rank 0 creates and sizes the segment
      fd = shm_open(c->name, O_RDWR, 0600);
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
      if (fd >= 0) { close(fd); fd = -1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { delete c; return 3; }
    }
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return 3; }
  c->hdr = (Header*)m;  // a fresh segment is zero-filled: count = gen = errors = 0
  c->slots = (char*)m + 4096;
  if (hipHostMalloc((void**)&c->staging, kSlotBytes, hipHostMallocDefault) != hipSuccess) { munmap(m, c->map_bytes); delete c; return 1; }
  if (!barrier(c)) { return 3; }
  if (rank == 0) shm_unlink(c->name);  // everyone has it mapped: the name can go, the memory lives until the last unmap
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  if (c->staging) (void)hipHostFree(c->staging);
  if (c->hdr) munmap((void*)c->hdr, c->map_bytes);
  delete c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) {
  if (dtype != 7 || op != 0) return 4;  // ncclFloat32, ncclSum: all the engine uses
  return collective(send, recv, count, 0, 0, (Comm*)comm, s);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t s) {
  if (dtype != 7) return 4;
  return collective(send, recv, count, 1, root, (Comm*)comm, s);
}

// test hooks (not part of the RCCL API): sequence mismatches / time-outs seen by this process, collectives completed
unsigned fake_rccl_errors(void) { return g_local_errors.load(); }
unsigned fake_rccl_collectives(void* comm) { Comm* c = (Comm*)comm; return c && c->hdr ? c->hdr->collectives.load() : 0u; }

}  // extern "C"
