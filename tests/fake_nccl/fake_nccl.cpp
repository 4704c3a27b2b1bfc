// fake_nccl.cpp -- TEST DOUBLE for librccl: the handful of nccl* entry points libmcl_hip.so calls, between THREADS of
// one process whose ranks all live on the same GPU.  Test infrastructure only (tests/test_gpu_multirank_shim.py loads it
// with LD_PRELOAD in a child process): the GPU boxes of this pool have one device and RCCL refuses two ranks on one
// ("Duplicate GPU detected"), so the multi-rank branches of the product's host code -- mcl_comm_init_ex, the max / totals /
// hand-over-record collectives, exchange_dupes' grouped ncclSend / ncclRecv, the pinned-word spin, mcl_comm_selftest,
// shutdown and re-init -- had never run with world > 1.  This library makes them run: NOT RCCL's transport, NOT a
// performance path, but the same call sequence with every argument checked the way RCCL would rely on it:
//   * a collective completes only when ALL ranks of the communicator have called it with the same count / type / op
//     (a mismatch is reported as ncclInvalidArgument on every rank instead of RCCL's hang);
//   * a ncclRecv matches the peer's ncclSend of the same group in posting order and must agree in count and type;
//   * ncclGroupStart / End defer the operations exactly like nccl.h says (nothing moves before the outermost End).
// Semantics are SYNCHRONOUS (the caller's stream is drained before and after every operation): stricter than RCCL's
// stream-ordered asynchrony, so a data-flow bug cannot hide behind timing; overlap is not modelled.
// Reductions are taken in rank order.  Waits time out after FAKE_NCCL_TIMEOUT_S (default 120 s) with ncclSystemError.
//
// Two transports behind the same checks.  THREADS (default): the ranks are threads of one process, bytes move by device
// copies.  PROCESSES (FAKE_NCCL_SHM=1 when the unique id is made; the id carries the choice to the other ranks): the ranks
// are separate processes -- `python bench.py --gpus 2` under LD_PRELOAD, every rank on device 0 -- that meet in one POSIX
// shared-memory segment: a header (process-shared robust mutex + condition variable, the posted operations, one mailbox
// per (source, destination) pair) and one staging slot per rank; a contribution travels device -> the sender's slot ->
// the receiver's device.  A rank that dies is noticed (its pid is polled, a mutex it held comes back EOWNERDEAD) and every
// waiting rank returns ncclSystemError instead of hanging.  The segment's name is unlinked as soon as every rank has
// mapped it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// -DFAKE_NCCL_HOST_ONLY (tests/test_fake_nccl_processes.py, CPU): "device" pointers are host pointers and a copy is a
// memcpy -- the rendezvous, matching, mailbox and dead-rank logic of the double itself, without a GPU
#ifdef FAKE_NCCL_HOST_ONLY
#define hipMemcpy(dst, src, bytes, kind) (memcpy((dst), (src), (bytes)), hipSuccess)
#define hipMemcpyAsync(dst, src, bytes, kind, stream) (memcpy((dst), (src), (bytes)), hipSuccess)
#define hipStreamSynchronize(stream) (hipSuccess)
#endif

namespace {

struct Op {
  int kind = 0;  // 1 all-reduce, 2 all-gather, 3 send, 4 recv
  const void* send = nullptr;
  void* recv = nullptr;
  size_t count = 0;
  ncclDataType_t type = ncclInt8;
  ncclRedOp_t op = ncclSum;
  int peer = -1;
  hipStream_t stream = nullptr;
};

struct Post {  // a posted send waiting for its receive
  const void* ptr;
  size_t count;
  ncclDataType_t type;
  bool done;
};

struct World {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long gen = 0;
  int joined = 0, left = 0;
  bool aborted = false;
  bool mismatch = false;
  std::vector<Op> posted;                            // the collective every rank is in
  std::vector<std::vector<unsigned char>> host;      // all-reduce staging, one per rank
  std::map<std::pair<int, int>, std::deque<Post*>> mail;  // (src, dst) -> sends in posting order
  World* split_child = nullptr;                      // ncclCommSplit: the communicator being made
};

// what rank 0 of this process has been asked for so far (fake_nccl_counters): collectives, groups of point-to-point
// operations, point-to-point operations -- the per-step budget of the product's exchange is asserted on these
std::atomic<unsigned long long> g_n_coll{0}, g_n_groups{0}, g_n_p2p{0};
std::atomic<int> g_timeout_s{0};   // fake_nccl_set_timeout(); 0 = FAKE_NCCL_TIMEOUT_S or 120
int timeout_s() {
  static const int env = getenv("FAKE_NCCL_TIMEOUT_S") ? atoi(getenv("FAKE_NCCL_TIMEOUT_S")) : 120;
  const int t = g_timeout_s.load() > 0 ? g_timeout_s.load() : env;
  return t > 0 ? t : 120;
}

// all ranks of w arrive; false on abort / timeout
bool barrier(World* w, std::unique_lock<std::mutex>& lk) {
  const long g = w->gen;
  if (++w->arrived == w->n) {
    w->arrived = 0;
    ++w->gen;
    w->cv.notify_all();
    return !w->aborted;
  }
  const bool ok = w->cv.wait_for(lk, std::chrono::seconds(timeout_s()), [&] { return w->gen != g || w->aborted; });
  return ok && !w->aborted;
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <typename T>
void reduce_into(T* acc, const T* v, size_t n, ncclRedOp_t op) {
  for (size_t k = 0; k < n; ++k) {
    if (op == ncclSum) acc[k] = acc[k] + v[k];
    else if (op == ncclMax) acc[k] = v[k] > acc[k] ? v[k] : acc[k];
    else if (op == ncclMin) acc[k] = v[k] < acc[k] ? v[k] : acc[k];
  }
}

std::mutex g_reg_m;
std::map<std::string, World*> g_registry;   // unique id -> world being assembled
unsigned long long g_next_id = 1;

thread_local int t_depth = 0;
thread_local std::vector<std::pair<ncclComm_t, Op>> t_ops;

}  // namespace

struct ShmWorld;
struct ncclComm {
  World* w;        // threads of one process ...
  int rank;
  ShmWorld* sw;    // ... or processes that meet in shared memory (then w == nullptr)
};

// ------------------------------------------------------------------ ranks as PROCESSES: the shared-memory transport
namespace {

constexpr int SHM_MAX_RANKS = 16;
constexpr int SHM_MAIL_DEPTH = 8;

struct ShmOp {   // what a rank posted for the collective everybody is in
  int kind;
  size_t count;
  int type, op;
};
struct ShmPost {   // a posted send: where its bytes lie in the sender's slot
  size_t offset, count;
  int type;
};
struct ShmMail {   // sends of one (source, destination) pair in posting order
  unsigned long long tail, head;   // posted / consumed so far
  ShmPost d[SHM_MAIL_DEPTH];
};
struct ShmHdr {
  pthread_mutex_t m;
  pthread_cond_t cv;
  int n, arrived, joined, left, aborted, mismatch, splits;
  long gen;
  size_t slot_bytes, data_off;
  int pid[SHM_MAX_RANKS];
  ShmOp posted[SHM_MAX_RANKS];
  ShmMail mail[SHM_MAX_RANKS][SHM_MAX_RANKS];
};
}  // namespace

struct ShmWorld {
  ShmHdr* h = nullptr;
  size_t map_bytes = 0;
  std::string name;
  unsigned char* slot(int r) const { return (unsigned char*)h + h->data_off + (size_t)r * h->slot_bytes; }
};

namespace {

size_t shm_slot_bytes() {
  const char* e = getenv("FAKE_NCCL_SLOT_MB");
  const long mb = e ? atol(e) : 256;   // (sparse: tmpfs only backs the pages a run touches)
  return (size_t)(mb > 0 ? mb : 256) << 20;
}
size_t shm_total(int n, size_t slot) { return ((sizeof(ShmHdr) + 4095) & ~(size_t)4095) + (size_t)n * slot; }

// lock; a mutex whose owner died comes back EOWNERDEAD: the world is broken, say so to everybody
void shm_lock(ShmHdr* h) {
  const int rc = pthread_mutex_lock(&h->m);
  if (rc == EOWNERDEAD) {
    pthread_mutex_consistent(&h->m);
    h->aborted = 1;
    pthread_cond_broadcast(&h->cv);
  }
}
void shm_unlock(ShmHdr* h) { pthread_mutex_unlock(&h->m); }

// wait (mutex held) until pred() or abort or timeout; polls the other ranks' pids so that a killed rank ends the wait
template <typename F>
bool shm_wait(ShmHdr* h, F pred) {
  timespec t0;
  clock_gettime(CLOCK_REALTIME, &t0);
  const long limit = timeout_s();
  while (!pred() && !h->aborted) {
    timespec now;
    clock_gettime(CLOCK_REALTIME, &now);
    if (now.tv_sec - t0.tv_sec > limit) return false;
    for (int r = 0; r < h->n; ++r)
      if (h->pid[r] > 0 && kill(h->pid[r], 0) != 0 && errno == ESRCH) {
        fprintf(stderr, "[fake_nccl] rank process %d (rank %d) is gone\n", h->pid[r], r);
        h->aborted = 1;
        pthread_cond_broadcast(&h->cv);
        return false;
      }
    timespec until = now;
    until.tv_nsec += 100000000L;   // 0.1 s
    if (until.tv_nsec >= 1000000000L) { until.tv_nsec -= 1000000000L; until.tv_sec += 1; }
    const int rc = pthread_cond_timedwait(&h->cv, &h->m, &until);
    if (rc == EOWNERDEAD) {
      pthread_mutex_consistent(&h->m);
      h->aborted = 1;
      pthread_cond_broadcast(&h->cv);
    }
  }
  return !h->aborted;
}

bool shm_barrier(ShmHdr* h) {   // mutex held
  const long g = h->gen;
  if (++h->arrived == h->n) {
    h->arrived = 0;
    ++h->gen;
    pthread_cond_broadcast(&h->cv);
    return !h->aborted;
  }
  return shm_wait(h, [&] { return h->gen != g; });
}

// segments this process made whose names may still exist (a rank that never joined): removed at exit
std::mutex g_names_m;
std::vector<std::string> g_names;
void shm_unlink_leftovers() {
  std::lock_guard<std::mutex> lk(g_names_m);
  for (const std::string& n : g_names) shm_unlink(n.c_str());
}

ShmWorld* shm_create(const std::string& name, int n) {
  const size_t slot = shm_slot_bytes(), total = shm_total(n, slot);
  const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return nullptr;
  if (ftruncate(fd, (off_t)total) != 0) {
    close(fd);
    shm_unlink(name.c_str());
    return nullptr;
  }
  void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    shm_unlink(name.c_str());
    return nullptr;
  }
  ShmHdr* h = (ShmHdr*)p;   // (a fresh segment is zero-filled)
  pthread_mutexattr_t ma;
  pthread_mutexattr_init(&ma);
  pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
  pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
  pthread_mutex_init(&h->m, &ma);
  pthread_condattr_t ca;
  pthread_condattr_init(&ca);
  pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
  pthread_cond_init(&h->cv, &ca);
  h->slot_bytes = slot;
  h->data_off = (sizeof(ShmHdr) + 4095) & ~(size_t)4095;
  __sync_synchronize();
  h->n = n;   // (last: a rank that attaches waits for it)
  {
    std::lock_guard<std::mutex> lk(g_names_m);
    if (g_names.empty()) atexit(shm_unlink_leftovers);
    g_names.push_back(name);
  }
  ShmWorld* w = new ShmWorld;
  w->h = h;
  w->map_bytes = total;
  w->name = name;
  return w;
}

// n == 0: take the rank count from the header once it is there
ShmWorld* shm_attach(const std::string& name, int n) {
  int fd = -1;
  for (int tries = 0; tries < 50 * timeout_s() && fd < 0; ++tries) {   // (the creator may still be on its way)
    fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd < 0) usleep(20000);
  }
  if (fd < 0) return nullptr;
  struct stat st;
  for (int tries = 0; tries < 500; ++tries) {   // (created but not sized yet)
    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(ShmHdr)) break;
    usleep(10000);
  }
  const size_t total = (size_t)st.st_size;
  if (total < sizeof(ShmHdr)) {
    close(fd);
    return nullptr;
  }
  void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  ShmHdr* h = (ShmHdr*)p;
  for (int tries = 0; tries < 500 && *(volatile int*)&h->n == 0; ++tries) usleep(10000);   // (header not initialised yet)
  if (h->n == 0 || (n && h->n != n)) {
    munmap(p, total);
    return nullptr;
  }
  ShmWorld* w = new ShmWorld;
  w->h = h;
  w->map_bytes = total;
  w->name = name;
  return w;
}

// every rank joins: pid table, rendezvous, then the name goes away (the mappings stay)
ncclResult_t shm_join(ShmWorld* w, int rank) {
  ShmHdr* h = w->h;
  shm_lock(h);
  h->pid[rank] = (int)getpid();
  const bool last = ++h->joined == h->n;
  const bool ok = shm_barrier(h);
  shm_unlock(h);
  if (last) shm_unlink(w->name.c_str());
  return ok ? ncclSuccess : ncclSystemError;
}

bool same_shape(const ShmOp& a, const Op& b) { return a.kind == b.kind && a.count == b.count && a.type == (int)b.type && a.op == (int)b.op; }

ncclResult_t shm_collective(ncclComm_t c, const Op& o) {
  if (c->rank == 0) ++g_n_coll;
  ShmWorld* w = c->sw;
  ShmHdr* h = w->h;
  const size_t bytes = o.count * type_bytes(o.type);
  if (type_bytes(o.type) == 0) return ncclInvalidArgument;
  if (bytes > h->slot_bytes) {
    fprintf(stderr, "[fake_nccl] rank %d: a contribution of %zu bytes exceeds the staging slot (FAKE_NCCL_SLOT_MB)\n", c->rank, bytes);
    return ncclInvalidArgument;
  }
  if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // my contribution is final
  if (bytes && hipMemcpy(w->slot(c->rank), o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  shm_lock(h);
  h->posted[c->rank] = ShmOp{o.kind, o.count, (int)o.type, (int)o.op};
  if (!shm_barrier(h)) { shm_unlock(h); return ncclSystemError; }
  for (int r = 0; r < h->n; ++r)
    if (!same_shape(h->posted[r], o)) h->mismatch = 1;
  const bool bad = h->mismatch != 0;
  shm_unlock(h);
  ncclResult_t rc = ncclSuccess;
  if (!bad && bytes) {
    if (o.kind == 1) {
      std::vector<unsigned char> acc(w->slot(0), w->slot(0) + bytes);
      for (int r = 1; r < h->n; ++r) {
        if (o.type == ncclFloat64) reduce_into((double*)acc.data(), (const double*)w->slot(r), o.count, o.op);
        else if (o.type == ncclUint64) reduce_into((unsigned long long*)acc.data(), (const unsigned long long*)w->slot(r), o.count, o.op);
        else if (o.type == ncclInt32) reduce_into((int*)acc.data(), (const int*)w->slot(r), o.count, o.op);
        else if (o.type == ncclUint32) reduce_into((unsigned*)acc.data(), (const unsigned*)w->slot(r), o.count, o.op);
        else rc = ncclInvalidArgument;
      }
      if (rc == ncclSuccess && hipMemcpy(o.recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    } else {
      for (int r = 0; r < h->n && rc == ncclSuccess; ++r) {
        void* dst = (unsigned char*)o.recv + (size_t)r * bytes;
        if (r == c->rank && dst == o.send) continue;   // (in place: my own block is where it belongs)
        if (hipMemcpy(dst, w->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
      }
    }
  }
  shm_lock(h);
  bool ok = shm_barrier(h);   // nobody overwrites its slot before everybody has read it
  if (ok && c->rank == 0) h->mismatch = 0;
  ok = ok && shm_barrier(h);
  shm_unlock(h);
  if (!ok) return ncclSystemError;
  if (bad) {
    fprintf(stderr, "[fake_nccl] rank %d: collective called with different count / type / op on different ranks\n", c->rank);
    return ncclInvalidArgument;
  }
  return rc;
}

// the point-to-point operations of one group (all on one communicator: what the product posts): every send is staged and
// posted before any receive waits
ncclResult_t shm_p2p(std::vector<std::pair<ncclComm_t, Op>>& ops, size_t first, size_t last) {
  if (ops[first].first->rank == 0) {
    ++g_n_groups;
    g_n_p2p += last - first;
  }
  ncclResult_t rc = ncclSuccess;
  size_t cursor = 0;
  struct Mine { ShmWorld* w; int rank, peer; unsigned long long seq; size_t count; };
  std::vector<Mine> mine;
  for (size_t k = first; k < last; ++k) {
    const Op& o = ops[k].second;
    if (o.kind != 3) continue;
    ncclComm_t c = ops[k].first;
    ShmWorld* w = c->sw;
    ShmHdr* h = w->h;
    const size_t bytes = o.count * type_bytes(o.type);
    if (o.peer < 0 || o.peer >= h->n || type_bytes(o.type) == 0) return ncclInvalidArgument;
    if (cursor + bytes > h->slot_bytes) {
      fprintf(stderr, "[fake_nccl] rank %d: the sends of one group exceed the staging slot (FAKE_NCCL_SLOT_MB)\n", c->rank);
      return ncclInvalidArgument;
    }
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(w->slot(c->rank) + cursor, o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    shm_lock(h);
    ShmMail& q = h->mail[c->rank][o.peer];
    if (q.tail - q.head >= (unsigned long long)SHM_MAIL_DEPTH) {
      shm_unlock(h);
      fprintf(stderr, "[fake_nccl] rank %d: more than %d sends to rank %d in flight\n", c->rank, SHM_MAIL_DEPTH, o.peer);
      return ncclInvalidArgument;
    }
    q.d[q.tail % SHM_MAIL_DEPTH] = ShmPost{cursor, o.count, (int)o.type};
    mine.push_back(Mine{w, c->rank, o.peer, q.tail, o.count});
    ++q.tail;
    pthread_cond_broadcast(&h->cv);
    shm_unlock(h);
    cursor += (bytes + 255) & ~(size_t)255;
  }
  for (size_t k = first; k < last && rc == ncclSuccess; ++k) {
    const Op& o = ops[k].second;
    if (o.kind != 4) continue;
    ncclComm_t c = ops[k].first;
    ShmWorld* w = c->sw;
    ShmHdr* h = w->h;
    if (o.peer < 0 || o.peer >= h->n || type_bytes(o.type) == 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    shm_lock(h);
    ShmMail& q = h->mail[o.peer][c->rank];
    if (!shm_wait(h, [&] { return q.tail > q.head; })) {
      shm_unlock(h);
      fprintf(stderr, "[fake_nccl] rank %d: ncclRecv from %d of %zu elements never met a send\n", c->rank, o.peer, o.count);
      return ncclSystemError;
    }
    const ShmPost p = q.d[q.head % SHM_MAIL_DEPTH];
    shm_unlock(h);
    if (p.count != o.count || p.type != (int)o.type) {
      fprintf(stderr, "[fake_nccl] rank %d: ncclRecv from %d expects %zu elements, the send has %zu\n", c->rank, o.peer, o.count, p.count);
      rc = ncclInvalidArgument;
    } else if (o.count) {
      if (hipMemcpy(o.recv, w->slot(o.peer) + p.offset, o.count * type_bytes(o.type), hipMemcpyHostToDevice) != hipSuccess)
        rc = ncclUnhandledCudaError;
    }
    shm_lock(h);
    ++q.head;   // (consumed -- also a mismatched one: its sender must not wait for ever)
    pthread_cond_broadcast(&h->cv);
    shm_unlock(h);
  }
  // my sends are complete when their receivers have copied
  for (const Mine& s : mine) {
    ShmHdr* h = s.w->h;
    shm_lock(h);
    ShmMail& q = h->mail[s.rank][s.peer];
    const bool ok = shm_wait(h, [&] { return q.head > s.seq; });
    shm_unlock(h);
    if (!ok) {
      fprintf(stderr, "[fake_nccl] ncclSend to %d of %zu elements never met a receive\n", s.peer, s.count);
      if (rc == ncclSuccess) rc = ncclSystemError;
    }
  }
  return rc;
}

}  // namespace

namespace {

bool same_shape(const Op& a, const Op& b) { return a.kind == b.kind && a.count == b.count && a.type == b.type && a.op == b.op; }

ncclResult_t run_collective(ncclComm_t c, const Op& o) {
  if (c->rank == 0) ++g_n_coll;
  World* w = c->w;
  const size_t bytes = o.count * type_bytes(o.type);
  if (type_bytes(o.type) == 0) return ncclInvalidArgument;
  if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // my contribution is final
  std::unique_lock<std::mutex> lk(w->m);
  w->posted[c->rank] = o;
  if (o.kind == 1) {
    w->host[c->rank].resize(bytes);
    lk.unlock();
    if (hipMemcpy(w->host[c->rank].data(), o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    lk.lock();
  }
  if (!barrier(w, lk)) return ncclSystemError;
  for (int r = 0; r < w->n; ++r)
    if (!same_shape(w->posted[r], o)) w->mismatch = true;
  const bool bad = w->mismatch;
  ncclResult_t rc = ncclSuccess;
  if (!bad) {
    if (o.kind == 1) {
      std::vector<unsigned char> acc(w->host[0]);
      for (int r = 1; r < w->n; ++r) {
        if (o.type == ncclFloat64) reduce_into((double*)acc.data(), (const double*)w->host[r].data(), o.count, o.op);
        else if (o.type == ncclUint64) reduce_into((unsigned long long*)acc.data(), (const unsigned long long*)w->host[r].data(), o.count, o.op);
        else if (o.type == ncclInt32) reduce_into((int*)acc.data(), (const int*)w->host[r].data(), o.count, o.op);
        else if (o.type == ncclUint32) reduce_into((unsigned*)acc.data(), (const unsigned*)w->host[r].data(), o.count, o.op);
        else rc = ncclInvalidArgument;
      }
      lk.unlock();
      if (rc == ncclSuccess && hipMemcpy(o.recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
      lk.lock();
    } else {
      std::vector<const void*> src(w->n);
      for (int r = 0; r < w->n; ++r) src[r] = w->posted[r].send;
      lk.unlock();
      for (int r = 0; r < w->n && rc == ncclSuccess; ++r) {
        void* dst = (unsigned char*)o.recv + (size_t)r * bytes;
        if (dst == src[r] || bytes == 0) continue;   // (in place: my own block is where it belongs)
        if (hipMemcpyAsync(dst, src[r], bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      }
      if (rc == ncclSuccess && hipStreamSynchronize(o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      lk.lock();
    }
  }
  if (!barrier(w, lk)) return ncclSystemError;   // nobody reuses its send buffer before everybody has read it
  if (c->rank == 0) w->mismatch = false;
  if (!barrier(w, lk)) return ncclSystemError;
  if (bad) {
    fprintf(stderr, "[fake_nccl] rank %d: collective called with different count / type / op on different ranks\n", c->rank);
    return ncclInvalidArgument;
  }
  return rc;
}

// the point-to-point operations of one group: every send is posted before any receive waits, so two ranks that send
// to each other and then receive from each other do not deadlock (nccl.h: grouped send / recv progress together)
ncclResult_t run_p2p(std::vector<std::pair<ncclComm_t, Op>>& ops, size_t first, size_t last) {
  if (ops[first].first->rank == 0) {
    ++g_n_groups;
    g_n_p2p += last - first;
  }
  std::vector<Post*> mine;
  ncclResult_t rc = ncclSuccess;
  for (size_t k = first; k < last; ++k) {
    const Op& o = ops[k].second;
    if (o.kind != 3) continue;
    ncclComm_t c = ops[k].first;
    if (o.peer < 0 || o.peer >= c->w->n || type_bytes(o.type) == 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    Post* p = new Post{o.send, o.count, o.type, false};
    mine.push_back(p);
    std::lock_guard<std::mutex> lk(c->w->m);
    c->w->mail[{c->rank, o.peer}].push_back(p);
    c->w->cv.notify_all();
  }
  for (size_t k = first; k < last && rc == ncclSuccess; ++k) {
    const Op& o = ops[k].second;
    if (o.kind != 4) continue;
    ncclComm_t c = ops[k].first;
    World* w = c->w;
    if (o.peer < 0 || o.peer >= w->n || type_bytes(o.type) == 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(w->m);
    auto& q = w->mail[{o.peer, c->rank}];
    if (!w->cv.wait_for(lk, std::chrono::seconds(timeout_s()), [&] { return !q.empty() || w->aborted; }) || w->aborted) {
      fprintf(stderr, "[fake_nccl] rank %d: ncclRecv from %d of %zu elements never met a send\n", c->rank, o.peer, o.count);
      return ncclSystemError;
    }
    Post* p = q.front();
    q.pop_front();
    lk.unlock();
    if (p->count != o.count || p->type != o.type) {
      fprintf(stderr, "[fake_nccl] rank %d: ncclRecv from %d expects %zu elements, the send has %zu\n", c->rank, o.peer, o.count, p->count);
      rc = ncclInvalidArgument;
    } else if (o.count) {
      const size_t bytes = o.count * type_bytes(o.type);
      if (hipMemcpyAsync(o.recv, p->ptr, bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
          hipStreamSynchronize(o.stream) != hipSuccess)
        rc = ncclUnhandledCudaError;
    }
    lk.lock();
    p->done = true;
    w->cv.notify_all();
  }
  // my sends are complete when their receivers have copied
  for (size_t k = first, j = 0; k < last; ++k) {
    if (ops[k].second.kind != 3) continue;
    World* w = ops[k].first->w;
    Post* p = mine[j++];
    std::unique_lock<std::mutex> lk(w->m);
    if (!w->cv.wait_for(lk, std::chrono::seconds(timeout_s()), [&] { return p->done || w->aborted; }) || w->aborted) {
      fprintf(stderr, "[fake_nccl] rank %d: ncclSend to %d of %zu elements never met a receive\n", ops[k].first->rank,
              ops[k].second.peer, ops[k].second.count);
      if (rc == ncclSuccess) rc = ncclSystemError;
      continue;   // (the post stays in the mailbox: leaked on purpose, the world is broken)
    }
    lk.unlock();
    delete p;
  }
  return rc;
}

ncclResult_t run_ops(std::vector<std::pair<ncclComm_t, Op>>& ops) {
  ncclResult_t rc = ncclSuccess;
  size_t k = 0;
  while (k < ops.size() && rc == ncclSuccess) {
    const bool shm = ops[k].first->sw != nullptr;
    if (ops[k].second.kind <= 2) {
      rc = shm ? shm_collective(ops[k].first, ops[k].second) : run_collective(ops[k].first, ops[k].second);
      ++k;
    } else {
      size_t e = k;
      while (e < ops.size() && ops[e].second.kind >= 3 && (ops[e].first->sw != nullptr) == shm) ++e;
      rc = shm ? shm_p2p(ops, k, e) : run_p2p(ops, k, e);
      k = e;
    }
  }
  ops.clear();
  return rc;
}

ncclResult_t submit(ncclComm_t c, const Op& o) {
  if (!c || (!c->w && !c->sw)) return ncclInvalidArgument;
  if (c->sw ? c->sw->h->aborted != 0 : c->w->aborted) return ncclSystemError;
  t_ops.push_back({c, o});
  if (t_depth > 0) return ncclSuccess;
  return run_ops(t_ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  std::lock_guard<std::mutex> lk(g_reg_m);
  const char* shm = getenv("FAKE_NCCL_SHM");
  if (shm && shm[0] == '1')   // ranks as processes: the id names the shared-memory segment they will meet in
    snprintf(id->internal, sizeof id->internal, "/fake-nccl-shm-%d-%llu", (int)getpid(), g_next_id++);
  else
    snprintf(id->internal, sizeof id->internal, "fake-nccl-%llu", g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
  if (key.compare(0, 15, "/fake-nccl-shm-") == 0) {
    if (nranks > SHM_MAX_RANKS) return ncclInvalidArgument;
    // rank 0 makes the segment, the others wait for it
    ShmWorld* sw = rank == 0 ? shm_create(key, nranks) : shm_attach(key, nranks);
    if (!sw) {
      fprintf(stderr, "[fake_nccl] rank %d: no shared-memory segment %s (%s)\n", rank, key.c_str(), strerror(errno));
      return ncclSystemError;
    }
    const ncclResult_t rc = shm_join(sw, rank);
    if (rc != ncclSuccess) return rc;
    *comm = new ncclComm{nullptr, rank, sw};
    return ncclSuccess;
  }
  World* w;
  {
    std::lock_guard<std::mutex> lk(g_reg_m);
    World*& slot = g_registry[key];
    if (!slot) {
      slot = new World;
      slot->n = nranks;
      slot->posted.resize(nranks);
      slot->host.resize(nranks);
    }
    w = slot;
    if (w->n != nranks) return ncclInvalidArgument;
    if (++w->joined == nranks) g_registry.erase(key);   // complete: a later init with the same id makes a new world
  }
  std::unique_lock<std::mutex> lk(w->m);
  if (!barrier(w, lk)) return ncclSystemError;   // (RCCL's init is a rendezvous of all ranks too)
  lk.unlock();
  *comm = new ncclComm{w, rank, nullptr};
  return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  if (!comm || !newcomm) return ncclInvalidArgument;
  (void)key;
  if (color != 0) return ncclInvalidArgument;   // (the product splits into ONE colour, ranks kept)
  if (comm->sw) {
    // the child: a segment of its own, named after the parent and the number of splits so far; rank 0 makes it
    ShmHdr* h = comm->sw->h;
    shm_lock(h);
    const int k = h->splits;
    bool ok = shm_barrier(h);
    if (ok && comm->rank == 0) ++h->splits;
    shm_unlock(h);
    if (!ok) return ncclSystemError;
    const std::string name = comm->sw->name + "-s" + std::to_string(k);
    ShmWorld* sw = comm->rank == 0 ? shm_create(name, h->n) : shm_attach(name, h->n);
    if (!sw) return ncclSystemError;
    const ncclResult_t rc = shm_join(sw, comm->rank);
    if (rc != ncclSuccess) return rc;
    *newcomm = new ncclComm{nullptr, comm->rank, sw};
    return ncclSuccess;
  }
  World* w = comm->w;
  std::unique_lock<std::mutex> lk(w->m);
  if (comm->rank == 0) {
    World* c = new World;
    c->n = w->n;
    c->posted.resize(w->n);
    c->host.resize(w->n);
    w->split_child = c;
  }
  if (!barrier(w, lk)) return ncclSystemError;
  World* c = w->split_child;
  if (!barrier(w, lk)) return ncclSystemError;
  *newcomm = new ncclComm{c, comm->rank, nullptr};
  return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort) {
  if (!comm) return ncclSuccess;
  if (comm->sw) {
    ShmHdr* h = comm->sw->h;
    shm_lock(h);
    if (abort) {
      h->aborted = 1;
      pthread_cond_broadcast(&h->cv);
    }
    ++h->left;
    h->pid[comm->rank] = 0;   // (gone on purpose: not a dead rank)
    shm_unlock(h);
    munmap((void*)h, comm->sw->map_bytes);
    delete comm->sw;
    delete comm;
    return ncclSuccess;
  }
  World* w = comm->w;
  bool last;
  {
    std::lock_guard<std::mutex> lk(w->m);
    if (abort) {
      w->aborted = true;
      w->cv.notify_all();
    }
    last = ++w->left == w->n;
  }
  if (last) delete w;
  delete comm;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return leave(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return leave(comm, true); }

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_nccl)";
    case ncclSystemError: return "system error: abort or timeout (fake_nccl)";
    case ncclInvalidArgument: return "invalid argument (fake_nccl: counts / types / ops of the ranks do not match)";
    default: return "error (fake_nccl)";
  }
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  return run_ops(t_ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Op o;
  o.kind = 1; o.send = send; o.recv = recv; o.count = count; o.type = type; o.op = op; o.stream = stream;
  return submit(comm, o);
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
  Op o;
  o.kind = 2; o.send = send; o.recv = recv; o.count = count; o.type = type; o.stream = stream;
  return submit(comm, o);
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  Op o;
  o.kind = 3; o.send = send; o.count = count; o.type = type; o.peer = peer; o.stream = stream;
  return submit(comm, o);
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  Op o;
  o.kind = 4; o.recv = recv; o.count = count; o.type = type; o.peer = peer; o.stream = stream;
  return submit(comm, o);
}

// marker the test driver checks: the preload really is in front of librccl
int fake_nccl_present(void) { return 1; }
// {collectives, point-to-point groups, point-to-point operations} rank 0 of this process has issued so far
void fake_nccl_counters(unsigned long long out[3]) {
  out[0] = g_n_coll.load();
  out[1] = g_n_groups.load();
  out[2] = g_n_p2p.load();
}
// waits that begin after this call give up after `seconds` (0 = back to FAKE_NCCL_TIMEOUT_S)
void fake_nccl_set_timeout(int seconds) { g_timeout_s.store(seconds); }

}  // extern "C"
