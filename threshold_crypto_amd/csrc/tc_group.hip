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

#include <functional>
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

}  // namespace

struct tc_group {
  std::vector<int> devices;
  std::vector<tc_ctx*> ctx;
  std::vector<hipStream_t> streams;   // one communication stream per rank
  std::vector<ncclComm_t> comms;      // empty when the group runs without RCCL (duplicate devices: tests)
  std::vector<uint8_t*> d_keyset;     // per rank: (t+1) x 96 B commitment in HBM
  std::vector<uint64_t*> d_count;     // per rank: one u64 for the all-reduced valid count
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

// one worker thread per GPU; returns the first non-zero result
int run_ranks(tc_group* g, const std::function<int(int)>& body) {
  const int n = (int)g->ctx.size();
  std::vector<int> rc(n, TC_OK);
  std::vector<std::thread> th;
  for (int r = 0; r < n; r++) th.emplace_back([&, r]() {
    (void)hipSetDevice(g->devices[r]);
    rc[r] = body(r);
  });
  for (auto& t : th) t.join();
  for (int r = 0; r < n; r++)
    if (rc[r] != TC_OK) {
      g->err = std::string("rank ") + std::to_string(r) + ": " + tc_last_error(g->ctx[r]);
      return rc[r];
    }
  return TC_OK;
}

}  // namespace

extern "C" {

int tc_group_create(tc_group** out, const int* devices, int ndev) {
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
    g->streams.push_back(s);
    g->d_count.push_back(cnt);
    g->d_keyset.push_back(nullptr);
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
  *out = g;
  return TC_OK;
}

void tc_group_destroy(tc_group* g) {
  if (!g) return;
  for (auto c : g->comms)
    if (c) g->rccl.CommDestroy(c);
  for (size_t r = 0; r < g->devices.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
    if (r < g->d_keyset.size() && g->d_keyset[r]) (void)hipFree(g->d_keyset[r]);
    if (r < g->d_count.size() && g->d_count[r]) (void)hipFree(g->d_count[r]);
    if (r < g->streams.size() && g->streams[r]) (void)hipStreamDestroy(g->streams[r]);
  }
  for (auto c : g->ctx) tc_ctx_destroy(c);
  delete g;
}

int tc_group_size(const tc_group* g) { return g ? (int)g->ctx.size() : 0; }
tc_ctx* tc_group_ctx(tc_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr; }
const char* tc_group_last_error(const tc_group* g) { return g ? g->err.c_str() : "null group"; }
int tc_group_uses_rccl(const tc_group* g) { return g && !g->comms.empty(); }

int tc_group_shard(const tc_group* g, size_t B, int rank, size_t* start, size_t* count) {
  if (!g || !start || !count || rank < 0 || rank >= (int)g->ctx.size()) return TC_ERR_INVALID_ARG;
  shard(B, (int)g->ctx.size(), rank, start, count);
  return TC_OK;
}

// PublicKeySet { commit } (src/lib.rs:539-543) -> every GPU: rank 0's HBM copy is broadcast over RCCL
int tc_group_set_keyset(tc_group* g, size_t t, const uint8_t* commit) {
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
  if (hipMemcpyAsync(g->d_keyset[0], commit, bytes, hipMemcpyHostToDevice, g->streams[0]) != hipSuccess) return TC_ERR_HIP;
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
}

// a rank's copy of the key set (host buffer of (t+1) x 96 B): what tests read back to see the broadcast
int tc_group_get_keyset(tc_group* g, int rank, uint8_t* out_commit) {
  if (!g || !out_commit || !g->have_keyset || rank < 0 || rank >= (int)g->ctx.size()) return TC_ERR_INVALID_ARG;
  if (hipSetDevice(g->devices[rank]) != hipSuccess) return TC_ERR_HIP;
  return hipMemcpy(out_commit, g->d_keyset[rank], (g->t + 1) * 96, hipMemcpyDeviceToHost) == hipSuccess ? TC_OK : TC_ERR_HIP;
}

// PublicKeySet::combine_signatures (src/lib.rs:608-615) for B jobs in host memory, sharded over the GPUs
int tc_group_combine_signatures(tc_group* g, size_t n_per_job, const uint64_t* idx, const uint8_t* shares, size_t B, uint8_t* out,
                                uint8_t* status) {
  if (!g || !g->have_keyset || !idx || !shares || !out || !status) return TC_ERR_INVALID_ARG;
  return run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, (int)g->ctx.size(), r, &s, &c);
    return c ? tc_combine_g2_batch(g->ctx[r], g->t, n_per_job, idx + s * n_per_job, shares + s * n_per_job * 192, c, out + s * 192,
                                   status + s)
             : TC_OK;
  });
}

// PublicKey::verify_g2 (src/lib.rs:108-110) under the key set's master key (commit[0], already resident on every
// GPU); n_valid (optional) = sum over ranks of the valid counts, all-reduced over RCCL
int tc_group_verify_g2(tc_group* g, const uint8_t* sig, const uint8_t* hash, size_t B, uint8_t* ok, uint64_t* n_valid) {
  if (!g || !g->have_keyset || !sig || !hash || !ok) return TC_ERR_INVALID_ARG;
  const int n = (int)g->ctx.size();
  std::vector<uint8_t> pk(96);
  (void)hipSetDevice(g->devices[0]);
  if (hipMemcpy(pk.data(), g->d_keyset[0], 96, hipMemcpyDeviceToHost) != hipSuccess) return TC_ERR_HIP;
  std::vector<uint64_t> local(n, 0);
  const int rc = run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, n, r, &s, &c);
    if (!c) return TC_OK;
    const int e = tc_verify_g2_batch(g->ctx[r], pk.data(), 0, sig + s * 192, hash + s * 192, c, ok + s);
    for (size_t j = 0; j < c; j++) local[r] += ok[s + j] ? 1 : 0;
    return e;
  });
  if (rc != TC_OK || !n_valid) return rc;
  // sum of the per-rank counts: ncclAllReduce of one u64 per rank (every rank ends with the total)
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(g->devices[r]);
    if (hipMemcpyAsync(g->d_count[r], &local[r], 8, hipMemcpyHostToDevice, g->streams[r]) != hipSuccess) return TC_ERR_HIP;
  }
  if (!g->comms.empty()) {
    if (g->rccl.GroupStart() != 0) return TC_ERR_HIP;
    for (int r = 0; r < n; r++)
      if (g->rccl.AllReduce(g->d_count[r], g->d_count[r], 1, kNcclUint64, kNcclSum, g->comms[r], g->streams[r]) != 0) return TC_ERR_HIP;
    if (g->rccl.GroupEnd() != 0) return TC_ERR_HIP;
    (void)hipSetDevice(g->devices[0]);
    if (hipMemcpyAsync(n_valid, g->d_count[0], 8, hipMemcpyDeviceToHost, g->streams[0]) != hipSuccess) return TC_ERR_HIP;
    for (int r = 0; r < n; r++) {
      (void)hipSetDevice(g->devices[r]);
      if (hipStreamSynchronize(g->streams[r]) != hipSuccess) return TC_ERR_HIP;
    }
  } else {
    uint64_t tot = 0;
    for (int r = 0; r < n; r++) {
      (void)hipSetDevice(g->devices[r]);
      (void)hipStreamSynchronize(g->streams[r]);
      tot += local[r];
    }
    *n_valid = tot;
  }
  return TC_OK;
}

// BASELINE config 5 through the C ABI: for each of B messages sign the n shares of its signer subset on the device
// (sk_table: N x 32 B, broadcast with the call), combine them, verify the result under the master key.
// Host buffers: idx B x n (signer indices, ascending), msgs/off, out sig B x 192, ok B.
int tc_group_sign_combine_verify(tc_group* g, const uint8_t* sk_table, size_t N, const uint64_t* idx, size_t n, const uint8_t* msgs,
                                 const uint64_t* off, size_t B, uint8_t* sig, uint8_t* ok, uint64_t* n_valid) {
  if (!g || !g->have_keyset || !sk_table || !idx || !off || !sig || !ok || n == 0 || n <= g->t) return TC_ERR_INVALID_ARG;
  const int nr = (int)g->ctx.size();
  std::vector<uint8_t> pk(96);
  (void)hipSetDevice(g->devices[0]);
  if (hipMemcpy(pk.data(), g->d_keyset[0], 96, hipMemcpyDeviceToHost) != hipSuccess) return TC_ERR_HIP;
  const int rc = run_ranks(g, [&](int r) {
    size_t s, c;
    shard(B, nr, r, &s, &c);
    if (!c) return TC_OK;
    // the slice's messages with offsets rebased to 0
    std::vector<uint64_t> o(c + 1);
    for (size_t j = 0; j <= c; j++) o[j] = off[s + j] - off[s];
    std::vector<uint8_t> hashes(c * 192), shares(c * n * 192), st(c * n), stc(c);
    int e = tc_hash_g2_batch(g->ctx[r], msgs ? msgs + off[s] : nullptr, o.data(), c, hashes.data());
    if (e == TC_OK) e = tc_sign_shares_g2_batch(g->ctx[r], sk_table, N, idx + s * n, hashes.data(), n, c, shares.data(), st.data());
    if (e == TC_OK) e = tc_combine_g2_batch(g->ctx[r], g->t, n, idx + s * n, shares.data(), c, sig + s * 192, stc.data());
    if (e == TC_OK) e = tc_verify_g2_batch(g->ctx[r], pk.data(), 0, sig + s * 192, hashes.data(), c, ok + s);
    for (size_t j = 0; j < c && e == TC_OK; j++)
      if (stc[j] != TC_JOB_OK) ok[s + j] = 0;
    return e;
  });
  if (rc != TC_OK) return rc;
  if (n_valid) {
    uint64_t tot = 0;
    for (size_t j = 0; j < B; j++) tot += ok[j] ? 1 : 0;
    *n_valid = tot;
  }
  return TC_OK;
}

}  // extern "C"
