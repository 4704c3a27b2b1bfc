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
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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

struct ncclComm {
  World* w;
  int rank;
};

namespace {

bool same_shape(const Op& a, const Op& b) { return a.kind == b.kind && a.count == b.count && a.type == b.type && a.op == b.op; }

ncclResult_t run_collective(ncclComm_t c, const Op& o) {
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
    if (ops[k].second.kind <= 2) {
      rc = run_collective(ops[k].first, ops[k].second);
      ++k;
    } else {
      size_t e = k;
      while (e < ops.size() && ops[e].second.kind >= 3) ++e;
      rc = run_p2p(ops, k, e);
      k = e;
    }
  }
  ops.clear();
  return rc;
}

ncclResult_t submit(ncclComm_t c, const Op& o) {
  if (!c || !c->w) return ncclInvalidArgument;
  if (c->w->aborted) return ncclSystemError;
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
  snprintf(id->internal, sizeof id->internal, "fake-nccl-%llu", g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
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
  *comm = new ncclComm{w, rank};
  return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  if (!comm || !newcomm) return ncclInvalidArgument;
  (void)key;
  if (color != 0) return ncclInvalidArgument;   // (the product splits into ONE colour, ranks kept)
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
  *newcomm = new ncclComm{c, comm->rank};
  return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort) {
  if (!comm) return ncclSuccess;
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
// waits that begin after this call give up after `seconds` (0 = back to FAKE_NCCL_TIMEOUT_S)
void fake_nccl_set_timeout(int seconds) { g_timeout_s.store(seconds); }

}  // extern "C"
