// Multi-GPU surface of the C ABI (include/tc_amd.h "tc_group_*"): ONE host process, one
// context + one worker thread per GPU, the batch sharded contiguously, RCCL for the only exchange steps of the
// path -- the broadcast of the key-set parameters from rank 0 over xGMI and the all-reduce of the valid counts.
// (The Python/bench path reaches the same thing with one PROCESS per GPU through torch.distributed, whose
// "nccl" backend is RCCL: threshold_crypto_amd/parallel.py.)  Host code only; all compute is the gfx950 kernels
// behind the per-GPU tc_ctx entry points.  RCCL is bound at run time (dlopen) so that libtc_amd.so carries no
// link-time dependency and shares the process with whatever RCCL the host application already loaded.
#include "../../include/tc_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// the few RCCL entry points the path needs (rccl.h: ncclResult_t = int, ncclSuccess = 0)
typedef void* ncclComm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int /*ncclDataType_t*/, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int /*ncclRedOp_t*/, ncclComm_t, hipStream_t) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && AllReduce;
  }
};
constexpr int kNcclUint8 = 1, kNcclUint64 = 5, kNcclSum = 0;  // rccl.h: ncclUint8 = 1, ncclUint64 = 5, ncclSum = 0

// One PERSISTENT worker thread per GPU (created with the group, bound to its device once): a call hands every worker
// its slice and waits; no thread is created or joined per call.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> task;
  bool has_task = false, done = false, quit = false;
  int rc = 0;
};

// grow-only device buffers of one rank (the slice's hash points, shares, results live here between the kernels of
// a call: they never visit host memory)
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};
enum { kBufMsgs, kBufOff, kBufIdx, kBufHash, kBufShares, kBufSt, kBufSig, kBufStc, kBufOk, kBufSk, kBufCount };

}  // namespace

struct tc_group {
  std::vector<int> devices;
  std::vector<tc_ctx*> ctx;
  std::vector<hipStream_t> streams;   // one stream per rank: the rank's context runs on it, and so do the group's copies
  std::vector<ncclComm_t> comms;      // empty when the group runs without RCCL (duplicate devices: tests)
  std::vector<uint8_t*> d_keyset;     // per rank: (t+1) x 96 B commitment in HBM
  std::vector<uint64_t*> d_count;     // per rank: one u64 for the all-reduced valid count
  std::vector<Worker*> workers;
  std::vector<std::vector<DevBuf>> bufs;
  std::atomic<uint64_t> h2d{0}, d2h{0};  // bytes the group's own copies moved over PCIe (the contexts count theirs)
  size_t t = 0;
  bool have_keyset = false;
  Rccl rccl;
  std::string err;
};

namespace {

void shard(size_t total, int world, int rank, size_t* start, size_t* count) {
  const size_t base = total / (size_t)world, extra = total % (size_t)world;
  *start = (size_t)rank * base + ((size_t)rank < extra ? (size_t)rank : extra);
  *count = base + ((size_t)rank < extra ? 1 : 0);
}

void worker_main(tc_group* g, int r) {
  (void)hipSetDevice(g->devices[r]);
  Worker* w = g->workers[r];
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_task || w->quit; });
    if (w->quit) return;
    std::function<int()> task = std::move(w->task);
    w->has_task = false;
    lk.unlock();
    int rc;
    try {
      rc = task();
    } catch (...) {
      rc = TC_ERR_HOST;  // (an exception must not leave a worker thread: std::terminate)
    }
    lk.lock();
    w->rc = rc;
    w->done = true;
    w->cv.notify_all();
  }
}

// every rank runs body(rank) on its worker thread; returns the first non-zero result
int run_ranks(tc_group* g, const std::function<int(int)>& body) {
  const int n = (int)g->ctx.size();
  for (int r = 0; r < n; r++) {
    Worker* w = g->workers[r];
    std::lock_guard<std::mutex> lk(w->m);
    w->task = [&body, r]() { return body(r); };
    w->has_task = true;
    w->done = false;
    w->cv.notify_all();
  }
  int first = TC_OK;
  for (int r = 0; r < n; r++) {
    Worker* w = g->workers[r];
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc != TC_OK && first == TC_OK) {
      first = w->rc;
      g->err = std::string("rank ") + std::to_string(r) + ": " + tc_last_error(g->ctx[r]);
    }
  }
  return first;
}

// a device buffer of the rank, at least `bytes` long (called on the rank's worker thread)
void* dev_buf(tc_group* g, int r, int which, size_t bytes) {
  DevBuf& b = g->bufs[r][which];
  if (bytes == 0) bytes = 8;
  if (b.cap < bytes) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t cap = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, cap) != hipSuccess) return nullptr;
    b.cap = cap;
  }
  return b.p;
}
bool h2d(tc_group* g, int r, void* dst, const void* src, size_t n) {
  if (!n) return true;
  g->h2d += n;
  return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, g->streams[r]) == hipSuccess;
}
bool d2h(tc_group* g, int r, void* dst, const void* src, size_t n) {
  if (!n) return true;
  g->d2h += n;
  return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, g->streams[r]) == hipSuccess;
}

// sum of the per-rank counts: ncclAllReduce of one u64 per rank (every rank ends with the total); without RCCL
// (several ranks on one GPU: tests) the host adds them
int all_reduce_counts(tc_group* g, const std::vector<uint64_t>& local, uint64_t* total) {
  const int n = (int)g->ctx.size();
  if (g->comms.empty()) {
    uint64_t tot = 0;
    for (int r = 0; r < n; r++) tot += local[r];
    *total = tot;
    return TC_OK;
  }
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(g->devices[r]);
    if (!h2d(g, r, g->d_count[r], &local[r], 8)) return TC_ERR_HIP;
  }
  if (g->rccl.GroupStart() != 0) return TC_ERR_HIP;
  for (int r = 0; r < n; r++)
    if (g->rccl.AllReduce(g->d_count[r], g->d_count[r], 1, kNcclUint64, kNcclSum, g->comms[r], g->streams[r]) != 0) return TC_ERR_HIP;
  if (g->rccl.GroupEnd() != 0) return TC_ERR_HIP;
  (void)hipSetDevice(g->devices[0]);
  if (!d2h(g, 0, total, g->d_count[0], 8)) return TC_ERR_HIP;
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(g->devices[r]);
    if (hipStreamSynchronize(g->streams[r]) != hipSuccess) return TC_ERR_HIP;
  }
  return TC_OK;
}

}  // namespace

static int on_exception_group(tc_group* g) noexcept {
  if (g) {
    try {
      g->err = "C++ exception in the host code (std::bad_alloc?)";
    } catch (...) {
    }
  }
  return TC_ERR_HOST;
}
static int on_exception(void*) noexcept { return TC_ERR_HOST; }

extern "C" {

int tc_group_create(tc_group** out, const int* devices, int ndev) try {
  if (!out || !devices || ndev <= 0) return TC_ERR_INVALID_ARG;
  *out = nullptr;
  tc_group* g = new tc_group();
  g->devices.assign(devices, devices + ndev);
  bool distinct = true;
  for (int a = 0; a < ndev; a++)
    for (int b = a + 1; b < ndev; b++) distinct = distinct && devices[a] != devices[b];
  for (int r = 0; r < ndev; r++) {
    tc_ctx* c = nullptr;
    const int rc = tc_ctx_create(&c, devices[r]);
    if (rc != TC_OK) {
      tc_group_destroy(g);
      return rc;
    }
    g->ctx.push_back(c);
    hipStream_t s = nullptr;
    uint64_t* cnt = nullptr;
    if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&cnt, 8) != hipSuccess) {
      tc_group_destroy(g);
      return TC_ERR_HIP;
    }
    (void)tc_ctx_set_stream(c, s);  // the rank's kernels and the group's copies share one stream: stream order is data order
    g->streams.push_back(s);
    g->d_count.push_back(cnt);
    g->d_keyset.push_back(nullptr);
    g->bufs.emplace_back(kBufCount);
  }
  if (distinct) {
    // one RCCL communicator per GPU of this process (ncclCommInitAll): the rings run over xGMI
    if (!g->rccl.load()) {
      tc_group_destroy(g);
      return TC_ERR_NO_DEVICE;  // RCCL not found: a multi-GPU group cannot exchange its key set
    }
    g->comms.resize(ndev);
    if (g->rccl.CommInitAll(g->comms.data(), ndev, g->devices.data()) != 0) {
      g->comms.clear();
      tc_group_destroy(g);
      return TC_ERR_HIP;
    }
  }
  // duplicate device ids (several ranks on ONE GPU) are accepted so that the sharding / threading logic can be
  // exercised on a single-GPU box; the exchange steps then use device-to-device copies instead of RCCL
  // (all Worker objects exist before the first thread starts: a worker reads g->workers[r] on entry)
  for (int r = 0; r < ndev; r++) g->workers.push_back(new Worker());
  for (int r = 0; r < ndev; r++) g->workers[r]->th = std::thread(worker_main, g, r);
  *out = g;
  return TC_OK;
} catch (...) {
  return on_exception(nullptr);
}

void tc_group_destroy(tc_group* g) try {
  if (!g) return;
  for (auto w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
      w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  for (auto c : g->comms)
    if (c) g->rccl.CommDestroy(c);
  for (size_t r = 0; r < g->devices.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
    if (r < g->d_keyset.size() && g->d_keyset[r]) (void)hipFree(g->d_keyset[r]);
    if (r < g->d_count.size() && g->d_count[r]) (void)hipFree(g->d_count[r]);
    if (r < g->bufs.size())
      for (auto& b : g->bufs[r])
        if (b.p) (void)hipFree(b.p);
  }
  for (auto c : g->ctx) tc_ctx_destroy(c);
  for (size_t r = 0; r < g->streams.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
    if (g->streams[r]) (void)hipStreamDestroy(g->streams[r]);
  }
  delete g;
} catch (...) {
}

int tc_group_size(const tc_group* g) { return g ? (int)g->ctx.size() : 0; }
tc_ctx* tc_group_ctx(tc_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr; }
const char* tc_group_last_error(const tc_group* g) { return g ? g->err.c_str() : "null group"; }
int tc_group_uses_rccl(const tc_group* g) { return g && !g->comms.empty(); }

int tc_group_shard(const tc_group* g, size_t B, int rank, size_t* start, size_t* count) try {
  if (!g || !start || !count || rank < 0 || rank >= (int)g->ctx.size()) return TC_ERR_INVALID_ARG;
  shard(B, (int)g->ctx.size(), rank, start, count);
  return TC_OK;
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// bytes that crossed PCIe since the group was created: the group's own copies plus its contexts' staging copies
int tc_group_transfer_bytes(const tc_group* g, uint64_t* h2d_bytes, uint64_t* d2h_bytes) try {
  if (!g) return TC_ERR_INVALID_ARG;
  uint64_t up = g->h2d.load(), down = g->d2h.load();
  for (auto c : g->ctx) {
    uint64_t a = 0, b = 0;
    (void)tc_ctx_transfer_bytes(c, &a, &b);
    up += a;
    down += b;
  }
  if (h2d_bytes) *h2d_bytes = up;
  if (d2h_bytes) *d2h_bytes = down;
  return TC_OK;
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// PublicKeySet { commit } (src/lib.rs:539-543) -> every GPU: rank 0's HBM copy is broadcast over RCCL
int tc_group_set_keyset(tc_group* g, size_t t, const uint8_t* commit) try {
  if (!g || !commit || t >= (1u << 20)) return TC_ERR_INVALID_ARG;
  const size_t bytes = (t + 1) * 96;
  const int n = (int)g->ctx.size();
  for (int r = 0; r < n; r++) {
    if (hipSetDevice(g->devices[r]) != hipSuccess) return TC_ERR_HIP;
    if (g->d_keyset[r]) (void)hipFree(g->d_keyset[r]);
    g->d_keyset[r] = nullptr;
    if (hipMalloc((void**)&g->d_keyset[r], bytes) != hipSuccess) return TC_ERR_HIP;
  }
  (void)hipSetDevice(g->devices[0]);
  if (!h2d(g, 0, g->d_keyset[0], commit, bytes)) return TC_ERR_HIP;
  if (!g->comms.empty()) {
    if (g->rccl.GroupStart() != 0) return TC_ERR_HIP;
    for (int r = 0; r < n; r++)
      if (g->rccl.Broadcast(g->d_keyset[0], g->d_keyset[r], bytes, kNcclUint8, 0, g->comms[r], g->streams[r]) != 0) return TC_ERR_HIP;
    if (g->rccl.GroupEnd() != 0) return TC_ERR_HIP;
  } else {
    if (hipStreamSynchronize(g->streams[0]) != hipSuccess) return TC_ERR_HIP;
    for (int r = 1; r < n; r++)
      if (hipMemcpyAsync(g->d_keyset[r], g->d_keyset[0], bytes, hipMemcpyDeviceToDevice, g->streams[r]) != hipSuccess) return TC_ERR_HIP;
  }
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(g->devices[r]);
    if (hipStreamSynchronize(g->streams[r]) != hipSuccess) return TC_ERR_HIP;
  }
  g->t = t;
  g->have_keyset = true;
  return TC_OK;
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// a rank's copy of the key set (host buffer of (t+1) x 96 B): what tests read back to see the broadcast
int tc_group_get_keyset(tc_group* g, int rank, uint8_t* out_commit) try {
  if (!g || !out_commit || !g->have_keyset || rank < 0 || rank >= (int)g->ctx.size()) return TC_ERR_INVALID_ARG;
  if (hipSetDevice(g->devices[rank]) != hipSuccess) return TC_ERR_HIP;
  return hipMemcpy(out_commit, g->d_keyset[rank], (g->t + 1) * 96, hipMemcpyDeviceToHost) == hipSuccess ? TC_OK : TC_ERR_HIP;
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// PublicKeySet::combine_signatures (src/lib.rs:608-615) for B jobs in host memory, sharded over the GPUs.  The share
// signatures are the INPUT here (they arrive from the signers): each rank stages its own slice, nothing else moves.
int tc_group_combine_signatures(tc_group* g, size_t n_per_job, const uint64_t* idx, const uint8_t* shares, size_t B, uint8_t* out,
                                uint8_t* status) try {
  if (!g || !g->have_keyset || !idx || !shares || !out || !status) return TC_ERR_INVALID_ARG;
  return run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, (int)g->ctx.size(), r, &s, &c);
    if (!c) return (int)TC_OK;
    const int dio = tc_ctx_get_device_io(g->ctx[r]);  // host buffers here, whatever mode the user left the rank's context in
    (void)tc_ctx_set_device_io(g->ctx[r], 0);
    const int e = tc_combine_g2_batch(g->ctx[r], g->t, n_per_job, idx + s * n_per_job, shares + s * n_per_job * 192, c, out + s * 192, status + s);
    (void)tc_ctx_set_device_io(g->ctx[r], dio);
    return e;
  });
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// PublicKey::verify_g2 (src/lib.rs:108-110) under the key set's master key (commit[0], already resident on every
// GPU); n_valid (optional) = sum over ranks of the valid counts, all-reduced over RCCL
int tc_group_verify_g2(tc_group* g, const uint8_t* sig, const uint8_t* hash, size_t B, uint8_t* ok, uint64_t* n_valid) try {
  if (!g || !g->have_keyset || !sig || !hash || !ok) return TC_ERR_INVALID_ARG;
  const int n = (int)g->ctx.size();
  std::vector<uint64_t> local(n, 0);
  const int rc = run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, n, r, &s, &c);
    if (!c) return (int)TC_OK;
    // signatures and hash points are this call's input: staged into the rank's HBM; the master key is already there
    uint8_t* d_sig = (uint8_t*)dev_buf(g, r, kBufSig, c * 192);
    uint8_t* d_hash = (uint8_t*)dev_buf(g, r, kBufHash, c * 192);
    uint8_t* d_ok = (uint8_t*)dev_buf(g, r, kBufOk, c);
    if (!d_sig || !d_hash || !d_ok) return (int)TC_ERR_HIP;
    if (!h2d(g, r, d_sig, sig + s * 192, c * 192) || !h2d(g, r, d_hash, hash + s * 192, c * 192)) return (int)TC_ERR_HIP;
    const int dio = tc_ctx_get_device_io(g->ctx[r]);  // the rank's context is also handed to the user (tc_group_ctx): leave its mode as found
    (void)tc_ctx_set_device_io(g->ctx[r], 1);
    int e = tc_verify_g2_batch(g->ctx[r], g->d_keyset[r], 0, d_sig, d_hash, c, d_ok);
    (void)tc_ctx_set_device_io(g->ctx[r], dio);
    if (e == TC_OK && !d2h(g, r, ok + s, d_ok, c)) e = TC_ERR_HIP;
    if (e == TC_OK && hipStreamSynchronize(g->streams[r]) != hipSuccess) e = TC_ERR_HIP;
    for (size_t j = 0; j < c && e == TC_OK; j++) local[r] += ok[s + j] ? 1 : 0;
    return e;
  });
  if (rc != TC_OK || !n_valid) return rc;
  return all_reduce_counts(g, local, n_valid);
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

// BASELINE config 5 through the C ABI: for each of B messages sign the n shares of its signer subset on the device
// (sk_table: N x 32 B, sent with the call), combine them, verify the result under the master key.
// Host buffers: idx B x n (signer indices, ascending), msgs/off, out sig B x 192, ok B.  Per rank only its slice of
// msgs / off / idx goes up and its slice of sig / ok comes back; hash points and share signatures -- B x n x 192 B, the
// bulk of the data -- are made and consumed in the rank's HBM.
int tc_group_sign_combine_verify(tc_group* g, const uint8_t* sk_table, size_t N, const uint64_t* idx, size_t n, const uint8_t* msgs,
                                 const uint64_t* off, size_t B, uint8_t* sig, uint8_t* ok, uint64_t* n_valid) try {
  if (!g || !g->have_keyset || !sk_table || !idx || !off || !sig || !ok || n == 0 || n <= g->t || N == 0) return TC_ERR_INVALID_ARG;
  // offsets are read before anything is sharded: they must start at 0 and never decrease
  if (off[0] != 0) return TC_ERR_INVALID_ARG;
  for (size_t j = 0; j < B; j++)
    if (off[j + 1] < off[j]) return TC_ERR_INVALID_ARG;
  if (off[B] != 0 && !msgs) return TC_ERR_INVALID_ARG;
  const int nr = (int)g->ctx.size();
  std::vector<uint64_t> local(nr, 0);
  const int rc = run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, nr, r, &s, &c);
    if (!c) return (int)TC_OK;
    tc_ctx* ctx = g->ctx[r];
    const size_t mbytes = (size_t)(off[s + c] - off[s]);
    std::vector<uint64_t> o(c + 1);  // the slice's offsets rebased to 0
    for (size_t j = 0; j <= c; j++) o[j] = off[s + j] - off[s];
    uint8_t* d_msgs = (uint8_t*)dev_buf(g, r, kBufMsgs, mbytes);
    uint64_t* d_off = (uint64_t*)dev_buf(g, r, kBufOff, (c + 1) * 8);
    uint64_t* d_idx = (uint64_t*)dev_buf(g, r, kBufIdx, c * n * 8);
    uint8_t* d_sk = (uint8_t*)dev_buf(g, r, kBufSk, N * 32);
    uint8_t* d_hash = (uint8_t*)dev_buf(g, r, kBufHash, c * 192);
    uint8_t* d_shares = (uint8_t*)dev_buf(g, r, kBufShares, c * n * 192);
    uint8_t* d_st = (uint8_t*)dev_buf(g, r, kBufSt, c * n);
    uint8_t* d_sig = (uint8_t*)dev_buf(g, r, kBufSig, c * 192);
    uint8_t* d_stc = (uint8_t*)dev_buf(g, r, kBufStc, c);
    uint8_t* d_ok = (uint8_t*)dev_buf(g, r, kBufOk, c);
    if (!d_msgs || !d_off || !d_idx || !d_sk || !d_hash || !d_shares || !d_st || !d_sig || !d_stc || !d_ok) return (int)TC_ERR_HIP;
    bool up = h2d(g, r, d_msgs, msgs ? msgs + off[s] : nullptr, mbytes) && h2d(g, r, d_off, o.data(), (c + 1) * 8) &&
              h2d(g, r, d_idx, idx + s * n, c * n * 8) && h2d(g, r, d_sk, sk_table, N * 32);
    if (!up) {
      (void)hipMemsetAsync(d_sk, 0, N * 32, g->streams[r]);  // (part of) the secret share table may have gone up: wiped on this path too
      (void)hipStreamSynchronize(g->streams[r]);
      return (int)TC_ERR_HIP;
    }
    const int dio = tc_ctx_get_device_io(ctx);
    (void)tc_ctx_set_device_io(ctx, 1);
    // hash points and shares are made right here by the library's own kernels: known group members (tc_amd.h "Decoding")
    const int checks = tc_ctx_get_input_checks(ctx);
    (void)tc_ctx_set_input_checks(ctx, 0);
    int e = tc_hash_g2_batch(ctx, d_msgs, d_off, c, d_hash);
    if (e == TC_OK) e = tc_sign_shares_g2_batch(ctx, d_sk, N, d_idx, d_hash, n, c, d_shares, d_st);
    if (e == TC_OK) e = tc_combine_g2_batch(ctx, g->t, n, d_idx, d_shares, c, d_sig, d_stc);
    if (e == TC_OK) e = tc_verify_g2_batch(ctx, g->d_keyset[r], 0, d_sig, d_hash, c, d_ok);
    (void)hipMemsetAsync(d_sk, 0, N * 32, g->streams[r]);  // the secret key shares do not outlive the call
    (void)tc_ctx_set_input_checks(ctx, checks);
    (void)tc_ctx_set_device_io(ctx, dio);
    std::vector<uint8_t> stc(c);
    if (e == TC_OK && !(d2h(g, r, sig + s * 192, d_sig, c * 192) && d2h(g, r, ok + s, d_ok, c) && d2h(g, r, stc.data(), d_stc, c))) e = TC_ERR_HIP;
    if (hipStreamSynchronize(g->streams[r]) != hipSuccess && e == TC_OK) e = TC_ERR_HIP;
    for (size_t j = 0; j < c && e == TC_OK; j++) {
      if (stc[j] != TC_JOB_OK) ok[s + j] = 0;
      local[r] += ok[s + j] ? 1 : 0;
    }
    return e;
  });
  if (rc != TC_OK || !n_valid) return rc;
  return all_reduce_counts(g, local, n_valid);
} catch (...) {
  return on_exception_group(const_cast<tc_group*>(g));
}

}  // extern "C"
