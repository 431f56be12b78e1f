// C ABI of libtc_amd.so (declared in include/tc_amd.h): context, device staging and the
// batch entry points.  Host code only; every computation is a gfx950 kernel launch (see
// tc_launch.h).  There is no CPU compute path: without a HIP device tc_ctx_create fails.
#include "tc_launch.h"
#include "../../include/tc_amd.h"

#include <stdio.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>

namespace {

struct Slot {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct tc_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool device_io = false;
  bool timing = false;
  // validate uncompressed point operands (order-r subgroup) before using them: ON unless the caller opts out
  // (tc_ctx_set_input_checks(ctx, 0)) for operands it knows to be group members -- outputs of this library, values that
  // passed the checked decode
  bool input_checks = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the membership tests of checked-input mode run on a SECOND stream beside the call's main kernels (Call::run_checks): a
  // LOW-priority stream, created by the first call that needs it -- a context whose caller opted out of the tests never has
  // one (HIP multiplexes its streams onto a few hardware queues: a stream nobody uses must not take a queue from the main
  // streams of several contexts that run side by side, bench.py `streaming`) -- and the two ordering events
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  double last_ms = 0.0;
  std::string err;
  std::vector<Slot> slots;  // grow-only device staging buffers, reused across calls
  size_t next_slot = 0;
  uint8_t* g1_gen = nullptr;  // 96 B uncompressed G1 generator on the device
  // [3 (x^2 - 1)] g1: the generator side of a pairing check whose other side holds a hash point without
  // its last constant multiplication (tc_gls.h g2_clear_cofactor): e(pk, [c] Q') == e(g1, sig)  <=>
  // e(pk, Q') == e([1/c] g1, sig)
  uint8_t* g1_gen_unfix = nullptr;
  int32_t* fb_table = nullptr;  // signed 4-bit window table of the G1 generator (tc_dkg.h), built on first use
  // per-job ladder tables of the G2 kernels (tc_table.h): 256 MB + one flag word per wave slot, allocated on first use
  int32_t* tbl_mem = nullptr;
  uint32_t* tbl_flags = nullptr;
  bool tbl_reset = false;  // a call failed: a kernel may have died holding slots, clear the flags before the next use
  int cus = 0;
  tc::Tuning tuning;  // form choices, fixed when the context is created (tc_launch.h)
  uint64_t h2d_bytes = 0, d2h_bytes = 0;  // bytes this context's staging copies moved over PCIe (tc_ctx_transfer_bytes)
};

namespace {

struct Call {
  tc_ctx* c;
  bool failed = false;
  bool arg_error = false;
  struct CopyBack {
    void* host;
    const void* dev;
    size_t n;
  };
  std::vector<CopyBack> outs;
  std::vector<std::pair<void*, size_t>> wipe;
  std::vector<std::pair<void*, size_t>> wipe_after_copy;  // secret RESULTS staged for the host: zeroed after the D2H copy
  struct Pending {
    const uint8_t* valid;
    size_t per_job, group;
    // the membership-test launch that will fill `valid` (pts == nullptr: already filled, e.g. by a checked decode)
    bool g2 = false;
    const uint8_t* pts = nullptr;
    size_t stride = 0, n_per_job = 0, take = 0, n = 0;
    size_t bytes = 0;  // extent of the operand the deferred test reads (records * n_per_job * stride)
  };
  std::vector<Pending> checks;  // checked-input mode: validity bytes waiting to be applied to the jobs
  // device-io mode: the caller's output buffers.  A deferred membership test reads its operand WHILE the main kernels write
  // their results, so an operand that overlaps an output (an in-place call) keeps the order of a one-stream context: test first.
  std::vector<std::pair<const uint8_t*, size_t>> out_ranges;
  static bool ranges_overlap(const uint8_t* a, size_t na, const uint8_t* b, size_t nb) {
    return a && b && na && nb && (uintptr_t)a < (uintptr_t)b + nb && (uintptr_t)b < (uintptr_t)a + na;
  }
  bool overlaps_output(const uint8_t* p, size_t n) const {
    for (auto& r : out_ranges)
      if (ranges_overlap(p, n, r.first, r.second)) return true;
    return false;
  }

  explicit Call(tc_ctx* ctx) : c(ctx) {
    c->next_slot = 0;
    c->err.clear();
    if (hipSetDevice(c->device) != hipSuccess) fail("hipSetDevice failed");
    // hipGetLastError() is how finish() learns of a failed kernel LAUNCH, and the runtime keeps the last error of this host thread
    // until somebody reads it: a hipMalloc that failed in an EARLIER call (HBM full) would otherwise be reported, as "kernel
    // launch: out of memory", by the first call that runs after the memory came back
    // (tests/test_gpu_tables.py::test_failed_device_allocation_is_an_error_and_the_context_survives).  A call starts clean.
    (void)hipGetLastError();
  }
  void fail(const std::string& m) {
    if (!failed) c->err = m;
    failed = true;
  }
  bool check(hipError_t e, const char* what) {
    if (e != hipSuccess) {
      fail(std::string(what) + ": " + hipGetErrorString(e));
      return false;
    }
    return true;
  }
  // the table arena of the G2 ladder kernels (tc_table.h)
  tc::TableArena tables() {
    // (either half may be missing: an earlier call can have failed between the two allocations -- HBM full -- and a launcher
    // handed an arena without flags returns without doing anything)
    if ((!c->tbl_mem || !c->tbl_flags) && !failed) {
      if (!c->tbl_mem && !check(hipMalloc((void**)&c->tbl_mem, tc::kTableArenaWords * sizeof(int32_t)), "hipMalloc")) c->tbl_mem = nullptr;
      if (!failed && !c->tbl_flags) {
        if (check(hipMalloc((void**)&c->tbl_flags, tc::kTableArenaFlags * sizeof(uint32_t)), "hipMalloc"))
          check(hipMemsetAsync(c->tbl_flags, 0, tc::kTableArenaFlags * sizeof(uint32_t), c->stream), "memset");
        else
          c->tbl_flags = nullptr;
      }
    }
    if (c->tbl_flags && c->tbl_reset && !failed) {
      // stream-ordered after whatever ran before; nothing of this context is in flight beyond the stream
      if (check(hipMemsetAsync(c->tbl_flags, 0, tc::kTableArenaFlags * sizeof(uint32_t), c->stream), "memset")) c->tbl_reset = false;
    }
    return tc::TableArena{c->tbl_mem, c->tbl_flags};
  }
  void* scratch(size_t n) {
    if (n == 0) n = 8;
    if (c->next_slot >= c->slots.size()) c->slots.emplace_back();
    Slot& s = c->slots[c->next_slot++];
    if (s.cap < n) {
      if (s.p) (void)hipFree(s.p);
      s.p = nullptr;
      s.cap = 0;
      size_t cap = n + n / 4 + 256;
      if (!check(hipMalloc(&s.p, cap), "hipMalloc")) {
        s.p = nullptr;  // (whatever the failed call left in the slot is not a pointer to free later)
        return nullptr;
      }
      s.cap = cap;
    }
    return s.p;
  }
  // input operand: device pointer usable by kernels
  template <class T>
  const T* in(const T* p, size_t count, bool secret = false) {
    if (p) operand_bytes += count * sizeof(T);
    if (c->device_io || p == nullptr) return p;
    size_t n = count * sizeof(T);
    void* d = scratch(n);
    if (!d) return nullptr;
    if (n && !check(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, c->stream), "H2D copy")) return nullptr;
    c->h2d_bytes += n;
    if (secret) wipe.emplace_back(d, n);
    return (const T*)d;
  }
  // output operand: device pointer the kernels write; copied back in finish()
  template <class T>
  T* out(T* p, size_t count, bool zero = false) {
    if (p == nullptr) return nullptr;
    size_t n = count * sizeof(T);
    operand_bytes += n;
    T* d = p;
    if (!c->device_io) {
      d = (T*)scratch(n);
      if (!d) return nullptr;
      outs.push_back({p, d, n});
    } else if (n) {
      out_ranges.emplace_back((const uint8_t*)d, n);
      // (an operand registered before this output and deferred already: test it now, on the main stream, before anything writes)
      for (auto& pc : checks)
        if (pc.pts && ranges_overlap(pc.pts, pc.bytes, (const uint8_t*)d, n)) launch_pending(pc, c->stream);
    }
    if (zero && n) check(hipMemsetAsync(d, 0, n, c->stream), "memset");
    return d;
  }
  template <class T>
  T* temp(size_t count, bool zero = false) {
    size_t n = count * sizeof(T);
    T* d = (T*)scratch(n);
    if (d && zero && n) check(hipMemsetAsync(d, 0, n, c->stream), "memset");
    return d;
  }
  // the Miller values between the two kernels of a pairing check (k_pairing.hip)
  // (+ the line buffer of the prepared form, one tile of checks sized to the memory this call may spend: msm_table_budget)
  tc::PairingWs pairing_ws(size_t B);
  // Checked-input mode (tc_ctx_set_input_checks): validate the first `take` points of `records` records of
  // n_per_job points each (stride 0 = ONE point shared by every job); job j owns record j / group.
  // always: the test runs whatever the context's switch says (entries that multiply a SECRET by the operand)
  void check_points(bool g2, const uint8_t* d_pts, size_t stride, size_t n_per_job, size_t take, size_t records, size_t group,
                    bool always = false) {
    if (!(c->input_checks || always) || failed || !d_pts || !records || !take) return;
    const size_t PB = g2 ? 192 : 96;
    if (stride == 0) {
      stride = PB;
      records = 1;
      n_per_job = take = 1;
      group = (size_t)-1;
    }
    const size_t n = records * take;
    const size_t bytes = records * n_per_job * stride;
    uint8_t* v = temp<uint8_t>(n);
    if (!v) return;
    // The tests are NOT launched here.  They read operands only and nothing reads their verdict before apply_checks, so they
    // run on the context's second stream BESIDE the main kernels of the call, submitted after them (apply_checks): a launch
    // that ends with idle SIMDs -- the 65 536-job combination's last third, DESIGN.md 5.2 -- gets its tail filled (default-mode
    // combine_signatures 10.2 -> 9.3 ms per call, profiles/r06_checks_beside_ab.txt), and a small batch runs both at once.
    // ev_fork marks where the operands are ready in the main stream (every check_points call precedes the main kernels).
    const bool beside = c->tuning.checks_beside && !overlaps_output(d_pts, bytes) && side_stream_ready();
    if (beside && !check(hipEventRecord(c->ev_fork, c->stream), "event record")) return;
    checks.push_back({v, take, group, g2, d_pts, stride, n_per_job, take, n, bytes});
    if (!beside) launch_pending(checks.back(), c->stream);
  }
  // the context's second stream, made on first use: non-blocking and of the LOWEST priority the device offers, so that its
  // kernels take the wave slots the main kernels leave idle instead of competing with them for the dispatcher, and so that it
  // draws its hardware queue from another pool than the normal-priority streams
  bool side_stream_ready() {
    if (c->side_stream) return true;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, c->tuning.checks_beside == 2 ? 0 : least) != hipSuccess) {
      c->side_stream = nullptr;
      (void)hipGetLastError();
      return false;  // no second stream: the tests run before the main kernels, on the one stream (correct, just not overlapped)
    }
    return true;
  }
  // the event a launcher may re-record just before the call's one long kernel (k_combine.hip launch_combine_g2): the deferred
  // tests then start behind that kernel instead of beside the small kernels in front of it; nullptr when nothing is deferred
  hipEvent_t fork_event() const {
    for (auto& p : checks)
      if (p.pts) return c->ev_fork;
    return nullptr;
  }
  void launch_pending(Pending& p, hipStream_t st) {
    if (!p.pts) return;
    if (p.g2) tc::launch_subgroup_check_g2(st, p.pts, p.stride, p.n_per_job, p.take, p.n, const_cast<uint8_t*>(p.valid));
    else tc::launch_subgroup_check_g1(st, p.pts, p.stride, p.n_per_job, p.take, p.n, const_cast<uint8_t*>(p.valid));
    p.pts = nullptr;  // launched
  }
  // the tests check_points deferred: submitted now (after the call's main kernels) on the second stream, their verdicts
  // ordered before whatever the main stream does next.  Every reader of Pending::valid goes through here first.
  void run_checks() {
    if (failed) return;
    bool beside = false;
    for (auto& p : checks) beside = beside || p.pts != nullptr;
    if (!beside) return;
    // second stream: wait for the operands, run the tests, hand the verdicts back to the main stream
    if (!check(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0), "stream wait")) return;
    for (auto& p : checks) launch_pending(p, c->side_stream);
    if (check(hipEventRecord(c->ev_join, c->side_stream), "event record")) check(hipStreamWaitEvent(c->stream, c->ev_join, 0), "stream wait");
  }
  // after the main kernels: jobs that own an invalid point fail like undecodable ones
  void apply_checks(size_t B, uint8_t* status, uint8_t* out, size_t out_bytes, uint8_t* ok) {
    run_checks();
    if (!failed)
      for (auto& p : checks) tc::launch_invalidate_jobs(c->stream, p.valid, p.per_job, p.group, B, status, out, out_bytes, ok);
    checks.clear();
  }
  // Every entry calls begin_timing() after its allocations and before its first launch: the place to refuse a call the RUNTIME
  // could not survive.  The kernels' private segments (spilled registers, ladder tables: up to 11 KB per lane) are allocated by
  // the ROCm runtime when a dispatch needs them, per queue, for as many waves as the dispatch has (at most 32 per CU); when that
  // allocation fails the runtime's queue-error callback calls abort() -- the process dies on a runtime thread, no error code
  // reaches anybody (reproduced with ~0.4 GB of HBM free: DESIGN.md 7).  So a call whose size could ask for more than is free is
  // turned away HERE, as TC_ERR_HIP.  The bound is generous (operand bytes / 48 lanes, every lane at the largest private
  // segment of the library): it only ever bites with the HBM nearly full.
  size_t operand_bytes = 0;
  void guard_private() {
    if (failed || c->tuning.private_reserve == 0) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    size_t need = c->tuning.private_reserve;
    if (need == (size_t)-1) {
      const size_t device_waves = (size_t)(c->cus > 0 ? c->cus : 256) * 32;
      size_t waves = operand_bytes / 48 / 64 + 1;  // (no entry moves fewer than 97 operand bytes per job, no kernel runs more than two lanes per 97 bytes)
      if (waves > device_waves) waves = device_waves;
      need = waves * 64 * tc::kMaxPrivateBytesPerLane + ((size_t)64 << 20);
    }
    if (free_b < need)
      fail("out of memory: " + std::to_string(free_b >> 20) + " MB of HBM free, the kernels' private segments may need " + std::to_string(need >> 20) +
           " MB (the ROCm runtime aborts the process when it cannot allocate them; TC_PRIVATE_RESERVE overrides)");
  }
  void begin_timing() {
    guard_private();
    if (c->timing && !failed) check(hipEventRecord(c->ev0, c->stream), "event record");
  }
  void end_timing() {
    if (c->timing && !failed) check(hipEventRecord(c->ev1, c->stream), "event record");
  }
  int finish() {
    if (!failed) check(hipGetLastError(), "kernel launch");
    for (auto& w : wipe) (void)hipMemsetAsync(w.first, 0, w.second, c->stream);  // zero secret scalars
    if (!failed)
      for (auto& o : outs)
        if (o.n) {
          check(hipMemcpyAsync(o.host, o.dev, o.n, hipMemcpyDeviceToHost, c->stream), "D2H copy");
          c->d2h_bytes += o.n;
        }
    for (auto& w : wipe_after_copy) (void)hipMemsetAsync(w.first, 0, w.second, c->stream);  // stream-ordered after the copies
    if (!c->device_io || c->timing || failed) {
      hipError_t e = hipStreamSynchronize(c->stream);
      if (!failed) check(e, "stream sync");
    }
    if (c->timing && !failed) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    }
    if (failed && !arg_error) c->tbl_reset = true;
    if (failed) (void)hipGetLastError();  // this call's own failure has been reported through `err`: do not leave it for the caller's next HIP check
    return failed ? (arg_error ? TC_ERR_INVALID_ARG : TC_ERR_HIP) : TC_OK;
  }
};

// total message bytes = off[B]; in device-io mode read it back (8 bytes)
bool total_bytes(Call& k, const uint64_t* off, size_t B, uint64_t* total) {
  if (!k.c->device_io) {
    // host offsets are validated here (O(B)); in device-io mode the caller guarantees off[0] = 0 and a
    // non-decreasing array (the kernels read msgs + off[j] .. off[j+1] as they are)
    bool good = off[0] == 0;
    for (size_t j = 0; j < B && good; j++) good = off[j + 1] >= off[j];
    if (!good) {
      k.fail("invalid argument: offsets must start at 0 and be non-decreasing");
      k.arg_error = true;
      return false;
    }
    *total = off[B];
    return true;
  }
  if (!k.check(hipMemcpyAsync(total, off + B, 8, hipMemcpyDeviceToHost, k.c->stream), "offset readback")) return false;
  k.c->d2h_bytes += 8;
  return k.check(hipStreamSynchronize(k.c->stream), "stream sync");
}

// sum_i scalars[j][i] * points[j][i] in G2 for n >= kMsmMinPoints through the two-stage kernels (k_msm.hip);
// d_st must hold B zeroed-or-flagged status bytes.  The per-share tables live in HBM (2 KB per share): batches
// whose tables would exceed kMsmTableBudget run as consecutive tiles of jobs through ONE table buffer (the stream
// orders a tile's ladder before the next tile's table stage).
constexpr size_t kMsmTableBudget = (size_t)24 << 30;
// What a call may spend on table buffers: a third of the HBM that is free right now (plus whatever the context's staging
// slots already hold), at most 24 GiB, at least 1 GiB -- several contexts on one GPU, or a smaller card, tile their batches
// finer instead of failing an allocation.  tc_ctx_trim() gives the slots back.
size_t msm_table_budget(Call& k) {
  if (k.c->tuning.msm_budget) return k.c->tuning.msm_budget;  // (TC_MSM_BUDGET at context creation: the tile loops under test)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)1 << 30;
  size_t held = 0;
  for (auto& s : k.c->slots) held += s.cap;
  size_t b = (free_b + held) / 3;
  if (b > kMsmTableBudget) b = kMsmTableBudget;
  if (b < ((size_t)1 << 30)) b = (size_t)1 << 30;
  return b;
}
// What the line buffer of the prepared pairing form may take: a third of the HBM that is free right now (plus the context's own
// staging slots), at most 24 GiB -- and NO floor (the smallest tile's buffer is 1.013 GB, just under msm_table_budget's
// 1 GiB floor, so with the floor a card with less than that free still tried the allocation and failed; without it pairing_tile
// answers 0 and the one-loop form, which needs no line buffer, runs).
size_t pairing_line_budget(Call& k) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  size_t held = 0;
  for (auto& s : k.c->slots) held += s.cap;
  const size_t b = (free_b + held) / 3;
  return b > kMsmTableBudget ? kMsmTableBudget : b;
}
tc::PairingWs Call::pairing_ws(size_t B) {
  const tc::Tuning& tn = c->tuning;
  const int form = tc::pairing_form(B, tn);
  // (the budget -- hipMemGetInfo and a walk over the slots -- is only looked up by the form that has a line buffer: small
  // batches, which run the four-lane form, pay nothing for it)
  const size_t tile = tc::pairing_form_needs_lines(form) ? tc::pairing_tile(B, tn.pairing_budget ? tn.pairing_budget : pairing_line_budget(*this)) : 0;
  return tc::PairingWs{temp<int32_t>(tc::pairing_ws_words(B, tile)), tile, form};
}
void msm_g2(Call& k, size_t n, size_t pts_stride, const uint8_t* d_pts, const uint32_t* d_scalars, size_t B, uint8_t* d_out,
            uint8_t* d_st, int nbits = 64, tc::MsmFilter filter = tc::MsmFilter(), bool secret_scalars = false) {
  const size_t per_job = tc::msm_table_bytes(n, 1);
  size_t tile = msm_table_budget(k) / (per_job ? per_job : 1);
  if (tile < 1) tile = 1;
  if (tile > B) tile = B;
  int32_t* d_tbl = k.temp<int32_t>(tc::msm_table_bytes(n, tile) / sizeof(int32_t));
  uint8_t* d_codes = k.temp<uint8_t>(tc::msm_code_bytes(n, tile));
  // the digit codes are a lossless recoding of the scalars: wiped like them when they are secret
  if (secret_scalars && d_codes) k.wipe.emplace_back(d_codes, tc::msm_code_bytes(n, tile));
  for (size_t lo = 0; lo < B && !k.failed; lo += tile) {
    const size_t cnt = (B - lo < tile) ? B - lo : tile;
    tc::MsmFilter f = filter;
    if (f.idx) f.idx += lo * f.n_per_job;
    tc::launch_msm_g2(k.c->stream, n, pts_stride, d_pts + lo * pts_stride, d_scalars + lo * n * 8, cnt, d_tbl, d_codes, d_out + lo * 192,
                      d_st + lo, nbits, f);
  }
}

// the same in G1 (k_msm.hip launch_msm_g1).  pts_stride 0 with short scalars: ONE table set for the whole call.
void msm_g1(Call& k, size_t n, size_t pts_stride, const uint8_t* d_pts, const uint32_t* d_scalars, size_t B, uint8_t* d_out, uint8_t* d_st,
            int nbits = 128, bool secret_scalars = false) {
  const bool shared = tc::msm_table_jobs_g1(pts_stride, nbits, B) == 1;
  const size_t per_job = shared ? tc::msm_code_bytes(n, 1) : tc::msm_table_bytes_g1(n, 1);
  size_t tile = msm_table_budget(k) / (per_job ? per_job : 1);
  if (tile < 1) tile = 1;
  if (tile > B) tile = B;
  int32_t* d_tbl = k.temp<int32_t>(tc::msm_table_bytes_g1(n, shared ? 1 : tile) / sizeof(int32_t));
  uint8_t* d_codes = k.temp<uint8_t>(tc::msm_code_bytes(n, tile));
  // the digit codes are a lossless recoding of the scalars: wiped like them when they are secret
  if (secret_scalars && d_codes) k.wipe.emplace_back(d_codes, tc::msm_code_bytes(n, tile));
  for (size_t lo = 0; lo < B && !k.failed; lo += tile) {
    const size_t cnt = (B - lo < tile) ? B - lo : tile;
    tc::launch_msm_g1(k.c->stream, n, pts_stride, d_pts + lo * pts_stride, d_scalars + lo * n * 8, cnt, d_tbl, d_codes, d_out + lo * 96, d_st + lo,
                      nbits);
  }
}

// TC_DUO_MIN / TC_PAIRING_FORM / TC_PAIRING_BUDGET (tests, experiments): read here, once per context, never on the launch path
tc::Tuning tuning_from_env() {
  tc::Tuning tn;
  if (const char* e = getenv("TC_DUO_MIN")) tn.duo_min_decode = tn.duo_min_hash = (size_t)strtoull(e, nullptr, 10);
  if (const char* f = getenv("TC_PAIRING_FORM")) tn.pairing_form = f[0] == 'q' ? 1 : f[0] == 'l' ? 2 : f[0] == 'p' ? 3 : f[0] == 'f' ? 4 : 0;
  if (const char* b = getenv("TC_PAIRING_BUDGET")) tn.pairing_budget = (size_t)strtoull(b, nullptr, 10);
  if (const char* m = getenv("TC_MSM_BUDGET")) tn.msm_budget = (size_t)strtoull(m, nullptr, 10);
  if (const char* r = getenv("TC_PRIVATE_RESERVE")) tn.private_reserve = (size_t)strtoull(r, nullptr, 10);
  if (const char* o = getenv("TC_CHECKS_BESIDE")) tn.checks_beside = (o[0] >= '0' && o[0] <= '2') ? o[0] - '0' : 1;  // 0 one stream, 1 low priority, 2 normal
  return tn;
}

#define TC_REQUIRE(cond)            \
  do {                              \
    if (!(cond)) {                  \
      if (ctx) ctx->err = "invalid argument: " #cond; \
      return TC_ERR_INVALID_ARG;    \
    }                               \
  } while (0)

}  // namespace

// "never throws across the boundary" (include/tc_amd.h): every entry point below is a function-try-block that ends here.
// The host code allocates (std::vector, std::string); under memory pressure that throws, and an exception that crossed the C ABI
// into Rust or ctypes would be undefined behaviour.  The context stays usable: the arena flags are reset before the next use.
static int on_exception(tc_ctx* ctx) noexcept {
  const char* what = "C++ exception";
  try {
    throw;
  } catch (const std::bad_alloc&) {
    what = "host allocation failed (std::bad_alloc)";
  } catch (const std::exception&) {
    what = "C++ exception in the host code";
  } catch (...) {
  }
  if (ctx) {
    ctx->tbl_reset = true;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
    try {
      ctx->err = what;
    } catch (...) {
    }
  }
  return TC_ERR_HOST;
}

extern "C" {

const char* tc_version(void) { return "tc_amd 0.1.0 (gfx950)"; }

int tc_ctx_create(tc_ctx** out, int device) try {
  if (!out) return TC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TC_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return TC_ERR_INVALID_ARG;
  if (hipSetDevice(device) != hipSuccess) return TC_ERR_HIP;
  tc_ctx* c = new tc_ctx();
  c->device = device;
  c->tuning = tuning_from_env();
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipMalloc((void**)&c->g1_gen, 96) != hipSuccess || hipMalloc((void**)&c->g1_gen_unfix, 96) != hipSuccess) {
    tc_ctx_destroy(c);
    return TC_ERR_HIP;
  }
  c->stream = c->own_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cus = prop.multiProcessorCount;
  }
  (void)hipGetLastError();  // (whatever this thread's earlier HIP calls left behind is not this launch's)
  tc::launch_fill_g1_generator(c->stream, c->g1_gen, c->g1_gen_unfix);
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    // typically: no gfx950 code object for this device
    tc_ctx_destroy(c);
    return TC_ERR_NO_DEVICE;
  }
  *out = c;
  return TC_OK;
} catch (...) {
  return on_exception(nullptr);
}

void tc_ctx_destroy(tc_ctx* c) try {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // device-I/O calls may still be running on the context's (or the caller's) stream: nothing is freed under them
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);  // (joined into the main stream by every call that completed; a call that FAILED between launch and join is the exception)
  for (auto& s : c->slots)
    if (s.p) (void)hipFree(s.p);
  if (c->g1_gen) (void)hipFree(c->g1_gen);
  if (c->g1_gen_unfix) (void)hipFree(c->g1_gen_unfix);
  if (c->fb_table) (void)hipFree(c->fb_table);
  if (c->tbl_mem) (void)hipFree(c->tbl_mem);
  if (c->tbl_flags) (void)hipFree(c->tbl_flags);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
} catch (...) {
}

int tc_ctx_set_device_io(tc_ctx* ctx, int enabled) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  ctx->device_io = enabled != 0;
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ctx_set_stream(tc_ctx* ctx, void* hip_stream) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ctx_set_input_checks(tc_ctx* ctx, int enabled) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  ctx->input_checks = enabled != 0;
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// gives the context's grow-only staging / table buffers back to the device (they are allocated again on demand)
int tc_ctx_trim(tc_ctx* ctx) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return TC_ERR_HIP;
  for (auto& s : ctx->slots) {
    // staging slots may have held secret operands or their recodings: zeroed before they go back to the allocator
    if (s.p) (void)hipMemsetAsync(s.p, 0, s.cap, ctx->stream);
  }
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->slots) {
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
  }
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ctx_get_input_checks(const tc_ctx* ctx) { return (ctx && ctx->input_checks) ? 1 : 0; }
int tc_ctx_get_device_io(const tc_ctx* ctx) { return (ctx && ctx->device_io) ? 1 : 0; }
int tc_ctx_get_tuning(const tc_ctx* ctx, uint64_t* out8) try {
  if (!ctx || !out8) return TC_ERR_INVALID_ARG;
  out8[0] = ctx->tuning.duo_min_decode;
  out8[1] = ctx->tuning.duo_min_hash;
  out8[2] = (uint64_t)ctx->tuning.pairing_form;
  out8[3] = ctx->tuning.pairing_budget;
  out8[4] = (uint64_t)ctx->tuning.checks_beside;
  out8[5] = ctx->tuning.msm_budget;
  out8[6] = ctx->tuning.private_reserve;  // ~0 = by the size of the call
  out8[7] = 0;  // reserved
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ctx_transfer_bytes(const tc_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  if (h2d_bytes) *h2d_bytes = ctx->h2d_bytes;
  if (d2h_bytes) *d2h_bytes = ctx->d2h_bytes;
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ctx_set_timing(tc_ctx* ctx, int enabled) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  ctx->timing = enabled != 0;
  return TC_OK;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

double tc_last_kernel_ms(const tc_ctx* ctx) { return ctx ? ctx->last_ms : 0.0; }

int tc_sync(tc_ctx* ctx) try {
  if (!ctx) return TC_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return TC_ERR_HIP;
  return hipStreamSynchronize(ctx->stream) == hipSuccess ? TC_OK : TC_ERR_HIP;
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

const char* tc_last_error(const tc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ---- hashing ------------------------------------------------------------------------------
int tc_hash_g2_batch(tc_ctx* ctx, const uint8_t* msgs, const uint64_t* off, size_t B, uint8_t* out_g2) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && off && out_g2);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || msgs);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_out = k.out(out_g2, B * 192);
  k.begin_timing();
  if (!k.failed) tc::launch_hash_g2(ctx->tuning, ctx->stream, d_msgs, d_off, B, d_out);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_hash_g1_g2_batch(tc_ctx* ctx, const uint8_t* g1, const uint8_t* msgs, const uint64_t* off, size_t B,
                        uint8_t* out_g2, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && g1 && off && out_g2);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || msgs);
  const uint8_t* d_g1 = k.in(g1, B * 96);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_out = k.out(out_g2, B * 192);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  k.check_points(false, d_g1, 96, 1, 1, B, 1);
  if (!k.failed) tc::launch_hash_g1_g2(ctx->tuning, ctx->stream, d_g1, d_msgs, d_off, B, d_out, d_st);
  k.apply_checks(B, d_st, d_out, 192, nullptr);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- scalar multiplication -----------------------------------------------------------------
static int point_mul(tc_ctx* ctx, bool g2, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                     uint8_t* status) {
  TC_REQUIRE(ctx);
  if (S == 0 || B == 0) return TC_OK;
  TC_REQUIRE(ctx && fr && pts && out);
  const size_t PB = g2 ? 192 : 96;
  Call k(ctx);
  const uint8_t* d_fr = k.in(fr, S * 32, /*secret=*/true);
  const uint8_t* d_pts = k.in(pts, B * PB);
  uint8_t* d_out = k.out(out, S * B * PB);
  uint8_t* d_st = k.out(status, S * B);
  k.begin_timing();
  k.check_points(g2, d_pts, PB, 1, 1, B, S);
  if (!k.failed) {
    if (g2) tc::launch_g2_mul(ctx->stream, k.tables(), d_fr, d_pts, S, B, d_out, d_st);
    else tc::launch_g1_mul(ctx->stream, k.tables(), d_fr, d_pts, S, B, d_out, d_st);
  }
  k.apply_checks(S * B, d_st, d_out, PB, nullptr);
  k.end_timing();
  return k.finish();
}

int tc_g2_mul_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                    uint8_t* status) try {
  return point_mul(ctx, true, fr, pts, S, B, out, status);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_g1_mul_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* pts, size_t S, size_t B, uint8_t* out,
                    uint8_t* status) try {
  return point_mul(ctx, false, fr, pts, S, B, out, status);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_sign_shares_g2_batch(tc_ctx* ctx, const uint8_t* sk_table, size_t N, const uint64_t* idx, const uint8_t* hashes, size_t n,
                            size_t B, uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (n == 0 || B == 0) return TC_OK;
  TC_REQUIRE(ctx && sk_table && idx && hashes && out && N > 0);
  Call k(ctx);
  const uint8_t* d_sk = k.in(sk_table, N * 32, /*secret=*/true);
  const uint64_t* d_idx = k.in(idx, B * n);
  const uint8_t* d_h = k.in(hashes, B * 192);
  uint8_t* d_out = k.out(out, B * n * 192);
  uint8_t* d_st = k.out(status, B * n);
  k.begin_timing();
  k.check_points(true, d_h, 192, 1, 1, B, n);
  if (!k.failed && n >= tc::kCombMinSigners && B >= tc::kCombMinBatch) {
    // many signers per message: the doublings are done once per message, on a comb of its table (k_comb.hip); the combs
    // (133 KB per message) live in one HBM buffer, messages run through it in tiles
    const size_t per_msg = tc::comb_table_bytes(1);
    size_t tile = msm_table_budget(k) / per_msg;
    if (tile < 1) tile = 1;
    if (tile > B) tile = B;
    int32_t* d_tbl = k.temp<int32_t>(tc::comb_table_bytes(tile) / sizeof(int32_t));
    uint8_t* d_ok = k.temp<uint8_t>(tile);
    const tc::TableArena ta = k.tables();
    for (size_t lo = 0; lo < B && !k.failed; lo += tile) {
      const size_t cnt = (B - lo < tile) ? B - lo : tile;
      tc::launch_comb_sign(ctx->stream, ta, d_sk, N, d_idx + lo * n, d_h + lo * 192, n, cnt, d_tbl, d_ok, d_out + lo * n * 192,
                           d_st ? d_st + lo * n : nullptr);
    }
  } else if (!k.failed) {
    tc::launch_g2_mul_gather(ctx->stream, k.tables(), d_sk, N, d_idx, d_h, n, B, d_out, d_st);
  }
  k.apply_checks(B * n, d_st, d_out, 192, nullptr);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_sign_batch(tc_ctx* ctx, const uint8_t* fr, const uint8_t* msgs, const uint64_t* off, size_t S, size_t B,
                  uint8_t* out_g2, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (S == 0 || B == 0) return TC_OK;
  TC_REQUIRE(ctx && fr && off && out_g2);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || msgs);
  const uint8_t* d_fr = k.in(fr, S * 32, true);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_frc = k.temp<uint8_t>(S * 32);
  if (d_frc) k.wipe.emplace_back(d_frc, S * 32);
  uint8_t* d_out = k.out(out_g2, S * B * 192);
  uint8_t* d_st = k.out(status, S * B);
  k.begin_timing();
  if (!k.failed) {
    // sk * hash_g2(m) = (sk c) * Q': the hash skips its last constant multiplication, the scalars carry it
    tc::launch_hash_g2(ctx->tuning, ctx->stream, d_msgs, d_off, B, d_hash, /*fix=*/false);
    tc::launch_fr_scale_cofactor_fix(ctx->stream, d_fr, S, d_frc);
    tc::launch_g2_mul(ctx->stream, k.tables(), d_frc, d_hash, S, B, d_out, d_st);
  }
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- combination -----------------------------------------------------------------------------
// The launches of one combination over device-resident operands: Lagrange stage + the group's kernels.  d_idx: B x n u64
// abscissae (IntoFr for u64); d_idx_fr != nullptr: B x n Fr abscissae (8 canonical words each) instead -- every job takes
// the general path then, coefficients from k_lagrange_fr.  d_st: B zeroed status bytes (or already flagged jobs).
static void combine_launch(Call& k, bool g2, size_t t, size_t n, const uint64_t* d_idx, const uint32_t* d_idx_fr, const uint8_t* d_sh, size_t B,
                           uint8_t* d_pt, uint8_t* d_st) {
  tc_ctx* ctx = k.c;
  const size_t PB = g2 ? 192 : 96;
  uint32_t* d_lam = k.temp<uint32_t>(B * (t + 1) * 8);
  // G2, t <= 3: jobs are grouped by the class of their Lagrange denominator so that whole waves take
  // the cheap forms of the final division (tc_jobs.h combine_divide); not worth three launches for a
  // batch that fills a fraction of the machine anyway
  const bool group = !d_idx_fr && t >= 1 && t <= 3 && (g2 ? B >= 4096 : B >= tc::kG1GroupMinJobs);
  uint8_t* d_cls = group ? k.temp<uint8_t>(B) : nullptr;
  uint32_t* d_counters = group ? k.temp<uint32_t>(8) : nullptr;
  uint32_t* d_perm = group ? k.temp<uint32_t>(tc::combine_group_slots(B)) : nullptr;
  uint32_t* d_need = (t > 0 && !d_idx_fr) ? k.temp<uint32_t>(1, /*zero=*/true) : nullptr;
  if (k.failed) return;
  if (d_idx_fr) {
    // `T: IntoFr` abscissae beyond u64 (src/into_fr.rs:10-14, 28-56): no small-index fast path, no integer differences
    if (t > 0) tc::launch_lagrange_fr(ctx->stream, d_idx_fr, n, t, B, d_lam, d_st);
    if (g2 && t >= 1) msm_g2(k, t + 1, n * PB, d_sh, d_lam, B, d_pt, d_st);
    else if (g2) tc::launch_combine_g2(ctx->stream, k.tables(), t, n, nullptr, d_sh, d_lam, B, d_pt, d_st, nullptr, nullptr, nullptr, nullptr);
    else if (t + 1 >= tc::kMsmMinPoints) msm_g1(k, t + 1, n * PB, d_sh, d_lam, B, d_pt, d_st);
    else tc::launch_combine_g1(ctx->stream, t, n, nullptr, d_sh, d_lam, B, d_pt, d_st, nullptr);
    return;
  }
  if (t + 1 >= tc::kMsmMinPoints) {
    uint32_t* d_ws = k.temp<uint32_t>(tc::lagrange_all_ws_words(t, B));
    if (!k.failed) tc::launch_lagrange_all(ctx->stream, d_idx, n, t, B, d_lam, d_ws, d_st);
  } else if (t > 0) {
    tc::launch_lagrange(ctx->stream, d_idx, n, t, B, d_lam, d_st, d_need);
  }
  if (g2 && t + 1 >= tc::kMsmMinPoints) {
    msm_g2(k, t + 1, n * PB, d_sh, d_lam, B, d_pt, d_st);  // large thresholds: every job, coefficients from the one-inversion kernels
  } else if (g2) {
    // t <= 3: small-index fast path first; then (t >= 1) the jobs it left, through the two-stage kernels
    tc::launch_combine_g2(ctx->stream, k.tables(), t, n, d_idx, d_sh, d_lam, B, d_pt, d_st, d_cls, d_counters, d_perm, d_need,
                          k.fork_event());
    if (t >= 1) {
      tc::MsmFilter f;
      f.need = d_need;
      f.idx = d_idx;
      f.n_per_job = n;
      f.t = t;
      msm_g2(k, t + 1, n * PB, d_sh, d_lam, B, d_pt, d_st, 64, f);
    }
  }
  else if (t + 1 >= tc::kMsmMinPoints) msm_g1(k, t + 1, n * PB, d_sh, d_lam, B, d_pt, d_st);  // large thresholds in G1: the same two stages
  else tc::launch_combine_g1(ctx->stream, t, n, d_idx, d_sh, d_lam, B, d_pt, d_st, d_need,
                             k.tables(), d_cls, d_counters, d_perm, k.fork_event());
}

// samples.len() <= t  =>  Err(NotEnoughShares) for every job        (src/lib.rs:731-733)
static int not_enough_shares(Call& k, size_t B, uint8_t* out, size_t out_bytes, uint8_t* status) {
  if (k.c->device_io) {
    k.check(hipMemsetAsync(status, TC_JOB_NOT_ENOUGH_SHARES, B, k.c->stream), "memset");
    if (out && out_bytes) k.check(hipMemsetAsync(out, 0, B * out_bytes, k.c->stream), "memset");
  } else {
    memset(status, TC_JOB_NOT_ENOUGH_SHARES, B);
    if (out && out_bytes) memset(out, 0, B * out_bytes);
  }
  return k.finish();
}

// idx_fr != nullptr: the abscissae as 32-byte Fr values (idx unused); wire: the shares arrive compressed (48 / 96 B, checked
// decode) and a G2 result leaves compressed
static int combine(tc_ctx* ctx, bool g2, size_t t, size_t n, const uint64_t* idx, const uint8_t* idx_fr, const uint8_t* shares, size_t B,
                   uint8_t* out, uint8_t* status, const uint8_t* v, const uint64_t* off, uint8_t* plain, bool wire = false,
                   bool plain_unbacked = false) {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && out && status);
  TC_REQUIRE(t < (1u << 20));
  const size_t PB = g2 ? 192 : 96, CB = PB / 2;
  const size_t OB = plain ? 0 : (wire ? CB : PB);  // bytes of a point result
  Call k(ctx);
  if (n <= t) return not_enough_shares(k, B, plain ? nullptr : out, OB, status);
  TC_REQUIRE((idx || idx_fr) && shares);
  uint64_t total = 0;
  if (plain) {
    TC_REQUIRE(off);
    if (!total_bytes(k, off, B, &total)) return k.finish();
    TC_REQUIRE(total == 0 || v);
    TC_REQUIRE(total == 0 || !plain_unbacked);  // (a null plaintext buffer is only acceptable when there is nothing to write)
  }
  const uint64_t* d_idx = idx_fr ? nullptr : k.in(idx, B * n);
  const uint32_t* d_idx_fr = idx_fr ? reinterpret_cast<const uint32_t*>(k.in(idx_fr, B * n * 32)) : nullptr;
  const uint8_t* d_sh = k.in(shares, B * n * (wire ? CB : PB));
  uint8_t* d_st = k.out(status, B, /*zero=*/true);
  uint8_t* d_pt = (plain || wire) ? k.temp<uint8_t>(B * PB) : k.out(out, B * PB);
  uint8_t* d_wire_out = (wire && !plain) ? k.out(out, B * CB) : nullptr;
  const uint8_t* d_v = nullptr;
  const uint64_t* d_off = nullptr;
  uint8_t* d_plain = nullptr;
  if (plain) {
    d_v = k.in(v, (size_t)total);
    d_off = k.in(off, B + 1);
    d_plain = k.out(plain, (size_t)total, /*zero=*/true);  // failed jobs leave zeros, never stale staging bytes
  }
  k.begin_timing();
  size_t n_eff = n;
  if (idx_fr && !k.failed) {
    // abscissae that merely ARRIVED as Fr but fit 64 bits take the u64 kernels (fast path included); one word comes back
    uint64_t* d_idx64 = k.temp<uint64_t>(B * n);
    uint32_t* d_wide = k.temp<uint32_t>(1, /*zero=*/true);
    uint8_t* d_canon = k.temp<uint8_t>(B);
    if (!k.failed) {
      k.check(hipMemsetAsync(d_canon, 1, B, ctx->stream), "memset");
      tc::launch_fr_idx_narrow(ctx->stream, d_idx_fr, n, t + 1, B, d_idx64, d_wide, d_canon);
      uint32_t wide = 0;
      k.check(hipMemcpyAsync(&wide, d_wide, 4, hipMemcpyDeviceToHost, ctx->stream), "readback");
      ctx->d2h_bytes += 4;
      k.check(hipStreamSynchronize(ctx->stream), "stream sync");
      k.checks.push_back({d_canon, 1, 1});  // a job that owns a non-canonical abscissa (>= r) fails like an undecodable operand
      if (wide == 0) {
        d_idx = d_idx64;
        d_idx_fr = nullptr;
      }
    }
  }
  if (wire && !k.failed) {
    // the first t+1 samples of every job through the checked decode of from_bytes (src/lib.rs:140-146, 246-252), compactly
    uint8_t* d_dec = k.temp<uint8_t>(B * (t + 1) * PB);
    uint8_t* d_valid = k.temp<uint8_t>(B * (t + 1));
    if (!k.failed) {
      tc::launch_decompress_take(ctx->tuning, ctx->stream, g2, d_sh, n, t + 1, B, d_dec, d_valid);
      k.checks.push_back({d_valid, t + 1, 1});
      if (d_idx_fr) {
        uint32_t* d_fr2 = k.temp<uint32_t>(B * (t + 1) * 8);
        if (d_fr2) tc::launch_take_u64(ctx->stream, reinterpret_cast<const uint64_t*>(d_idx_fr), n * 4, (t + 1) * 4, B, reinterpret_cast<uint64_t*>(d_fr2));
        d_idx_fr = d_fr2;
      } else {
        uint64_t* d_idx2 = k.temp<uint64_t>(B * (t + 1));
        if (d_idx2) tc::launch_take_u64(ctx->stream, d_idx, n, t + 1, B, d_idx2);
        d_idx = d_idx2;
      }
      d_sh = d_dec;
      n_eff = t + 1;
    }
  } else {
    k.check_points(g2, d_sh, PB, n, t + 1, B, 1);  // exactly the first t+1 samples interpolate() takes
  }
  if (!k.failed) {
    combine_launch(k, g2, t, n_eff, d_idx, d_idx_fr, d_sh, B, d_pt, d_st);
    k.apply_checks(B, d_st, d_pt, PB, nullptr);
    if (plain) tc::launch_xor_with_hash(ctx->stream, d_pt, d_v, d_off, B, d_plain, d_st);
    else if (wire) {
      if (g2) tc::launch_g2_compress(ctx->stream, d_pt, B, d_wire_out, nullptr);   // Signature::to_bytes src/lib.rs:255-259
      else tc::launch_g1_compress(ctx->stream, d_pt, B, d_wire_out, nullptr);
    }
  }
  k.end_timing();
  return k.finish();
}

int tc_combine_g2_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                        size_t B, uint8_t* out, uint8_t* status) try {
  return combine(ctx, true, t, n_per_job, idx, nullptr, shares, B, out, status, nullptr, nullptr, nullptr);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_combine_g1_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares,
                        size_t B, uint8_t* out, uint8_t* status) try {
  return combine(ctx, false, t, n_per_job, idx, nullptr, shares, B, out, status, nullptr, nullptr, nullptr);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// The plaintext buffer of a batch whose plaintexts are ALL empty has no address when it is device-resident (an empty tensor's
// data pointer is null): the decrypt entries then stand a private byte in for it -- never written, combine() insists that the
// total length is 0 -- so that `SecretKey::decrypt(encrypt(b""))` works with either residency.
static uint8_t g_no_plaintext_bytes;
int tc_decrypt_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares_g1,
                     const uint8_t* v, const uint64_t* off, size_t B, uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;  // (an empty batch is a no-op for every entry: a device-resident empty buffer has no address)
  TC_REQUIRE(ctx && status && off);
  const bool unbacked = out == nullptr;
  if (unbacked) out = &g_no_plaintext_bytes;
  return combine(ctx, false, t, n_per_job, idx, nullptr, shares_g1, B, out /*non-null marker*/, status, v, off, out, /*wire=*/false, unbacked);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// `T: IntoFr` beyond u64 (src/into_fr.rs:10-14, 28-56; combine_signatures / decrypt are generic over it, src/lib.rs:608-622)
int tc_combine_g2_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares, size_t B, uint8_t* out,
                           uint8_t* status) try {
  return combine(ctx, true, t, n_per_job, nullptr, idx_fr, shares, B, out, status, nullptr, nullptr, nullptr);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}
int tc_combine_g1_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares, size_t B, uint8_t* out,
                           uint8_t* status) try {
  return combine(ctx, false, t, n_per_job, nullptr, idx_fr, shares, B, out, status, nullptr, nullptr, nullptr);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}
int tc_decrypt_fr_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint8_t* idx_fr, const uint8_t* shares_g1, const uint8_t* v,
                        const uint64_t* off, size_t B, uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;  // (an empty batch is a no-op for every entry: a device-resident empty buffer has no address)
  TC_REQUIRE(ctx && status && off);
  const bool unbacked = out == nullptr;
  if (unbacked) out = &g_no_plaintext_bytes;
  return combine(ctx, false, t, n_per_job, nullptr, idx_fr, shares_g1, B, out, status, v, off, out, /*wire=*/false, unbacked);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// wire-level forms: shares as they travel (Signature::to_bytes / DecryptionShare's compressed G1, checked decode of
// from_bytes src/lib.rs:140-146, 246-252), the combined signature back as Signature::to_bytes (src/lib.rs:255-259)
int tc_combine_signatures_wire_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares96, size_t B,
                                     uint8_t* out96, uint8_t* status) try {
  return combine(ctx, true, t, n_per_job, idx, nullptr, shares96, B, out96, status, nullptr, nullptr, nullptr, /*wire=*/true);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}
int tc_decrypt_wire_batch(tc_ctx* ctx, size_t t, size_t n_per_job, const uint64_t* idx, const uint8_t* shares48, const uint8_t* v,
                          const uint64_t* off, size_t B, uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;  // (an empty batch is a no-op for every entry: a device-resident empty buffer has no address)
  TC_REQUIRE(ctx && status && off);
  const bool unbacked = out == nullptr;
  if (unbacked) out = &g_no_plaintext_bytes;
  return combine(ctx, false, t, n_per_job, idx, nullptr, shares48, B, out, status, v, off, out, /*wire=*/true, unbacked);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

static int lincomb(tc_ctx* ctx, bool g2, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                   uint8_t* status) {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && out && (n == 0 || (scalars && points)));
  const size_t PB = g2 ? 192 : 96;
  Call k(ctx);
  const uint8_t* d_sc = k.in(scalars, B * n * 32, /*secret=*/true);
  const uint8_t* d_pt = k.in(points, B * n * PB);
  uint8_t* d_out = k.out(out, B * PB);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  k.check_points(g2, d_pt, PB, n, n, B, 1);
  if (!k.failed) {
    if (g2 && n == 0) {
      // the empty sum: the identity (flag byte 0x40, then zeros) for every job
      k.check(hipMemsetAsync(d_out, 0, B * PB, ctx->stream), "memset");
      k.check(hipMemset2DAsync(d_out, PB, 0x40, 1, B, ctx->stream), "memset");
      if (d_st) k.check(hipMemsetAsync(d_st, 0, B, ctx->stream), "memset");
    } else if (g2) {
      uint8_t* st_buf = d_st ? d_st : k.temp<uint8_t>(B);
      if (st_buf) k.check(hipMemsetAsync(st_buf, 0, B, ctx->stream), "memset");
      msm_g2(k, n, n * PB, d_pt, reinterpret_cast<const uint32_t*>(d_sc), B, d_out, st_buf, 64, tc::MsmFilter(), /*secret_scalars=*/true);
    } else if (n >= tc::kMsmMinPoints) {
      uint8_t* st_buf = d_st ? d_st : k.temp<uint8_t>(B);
      if (st_buf) k.check(hipMemsetAsync(st_buf, 0, B, ctx->stream), "memset");
      msm_g1(k, n, n * PB, d_pt, reinterpret_cast<const uint32_t*>(d_sc), B, d_out, st_buf, 128, /*secret_scalars=*/true);
    } else tc::launch_lincomb_g1(ctx->stream, n, d_sc, d_pt, B, d_out, d_st);
  }
  k.apply_checks(B, d_st, d_out, PB, nullptr);
  k.end_timing();
  return k.finish();
}

int tc_g1_lincomb_batch(tc_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                        uint8_t* status) try {
  return lincomb(ctx, false, n, scalars, points, B, out, status);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_g2_lincomb_batch(tc_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, size_t B, uint8_t* out,
                        uint8_t* status) try {
  return lincomb(ctx, true, n, scalars, points, B, out, status);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_xor_with_hash_batch(tc_ctx* ctx, const uint8_t* g1, const uint8_t* data, const uint64_t* off, size_t B,
                           uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && g1 && off);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || (data && out));  // (all-empty inputs: a device-resident empty buffer has no address)
  const uint8_t* d_g1 = k.in(g1, B * 96);
  const uint8_t* d_data = k.in(data, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_out = k.out(out, (size_t)total, /*zero=*/true);
  uint8_t* d_st = k.out(status, B, /*zero=*/true);
  k.begin_timing();
  k.check_points(false, d_g1, 96, 1, 1, B, 1);
  if (!k.failed) {
    k.apply_checks(B, d_st, nullptr, 0, nullptr);  // the keystream kernel skips jobs already flagged
    tc::launch_xor_with_hash(ctx->stream, d_g1, d_data, d_off, B, d_out, d_st);
  }
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- pairing checks ---------------------------------------------------------------------------
int tc_pairing_check_batch(tc_ctx* ctx, const uint8_t* a, size_t sa, const uint8_t* b, size_t sb, const uint8_t* c,
                           size_t sc, const uint8_t* d, size_t sd, size_t B, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && a && b && c && d && ok);
  TC_REQUIRE((sa == 0 || sa >= 96) && (sc == 0 || sc >= 96) && (sb == 0 || sb >= 192) && (sd == 0 || sd >= 192));
  Call k(ctx);
  auto span = [&](size_t stride, size_t bytes) { return stride ? (B - 1) * stride + bytes : bytes; };
  const uint8_t* da = k.in(a, span(sa, 96));
  const uint8_t* db = k.in(b, span(sb, 192));
  const uint8_t* dc = k.in(c, span(sc, 96));
  const uint8_t* dd = k.in(d, span(sd, 192));
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, da, sa, 1, 1, B, 1);
  k.check_points(true, db, sb, 1, 1, B, 1);
  k.check_points(false, dc, sc, 1, 1, B, 1);
  k.check_points(true, dd, sd, 1, 1, B, 1);
  if (!k.failed) tc::launch_pairing_check(ctx->stream, da, sa, db, sb, dc, sc, dd, sd, B, d_ok, k.pairing_ws(B));
  k.apply_checks(B, nullptr, nullptr, 0, d_ok);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_verify_g2_batch(tc_ctx* ctx, const uint8_t* pk, size_t pk_stride, const uint8_t* sig, const uint8_t* hash,
                       size_t B, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pk && sig && hash && ok);
  TC_REQUIRE(pk_stride == 0 || pk_stride >= 96);
  Call k(ctx);
  const uint8_t* d_pk = k.in(pk, pk_stride ? (B - 1) * pk_stride + 96 : 96);
  const uint8_t* d_sig = k.in(sig, B * 192);
  const uint8_t* d_hash = k.in(hash, B * 192);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, d_pk, pk_stride, 1, 1, B, 1);
  k.check_points(true, d_sig, 192, 1, 1, B, 1);
  k.check_points(true, d_hash, 192, 1, 1, B, 1);
  // e(pk, hash) == e(g1, sig)                                           (src/lib.rs:109)
  if (!k.failed) tc::launch_pairing_check(ctx->stream, d_pk, pk_stride, d_hash, 192, ctx->g1_gen, 0, d_sig, 192, B, d_ok, k.pairing_ws(B));
  k.apply_checks(B, nullptr, nullptr, 0, d_ok);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_verify_sig_batch(tc_ctx* ctx, const uint8_t* pk, size_t pk_stride, const uint8_t* sig, const uint8_t* msgs,
                        const uint64_t* off, size_t B, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pk && sig && off && ok);
  TC_REQUIRE(pk_stride == 0 || pk_stride >= 96);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || msgs);
  const uint8_t* d_pk = k.in(pk, pk_stride ? (B - 1) * pk_stride + 96 : 96);
  const uint8_t* d_sig = k.in(sig, B * 192);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, d_pk, pk_stride, 1, 1, B, 1);
  k.check_points(true, d_sig, 192, 1, 1, B, 1);
  if (!k.failed) {
    // e(pk, [c] Q') == e(g1, sig)  <=>  e(pk, Q') == e([1/c] g1, sig): the hash skips its last constant
    // multiplication and the generator side uses the context's pre-scaled generator
    tc::launch_hash_g2(ctx->tuning, ctx->stream, d_msgs, d_off, B, d_hash, /*fix=*/false);
    tc::launch_pairing_check(ctx->stream, d_pk, pk_stride, d_hash, 192, ctx->g1_gen_unfix, 0, d_sig, 192, B, d_ok, k.pairing_ws(B));
  }
  k.apply_checks(B, nullptr, nullptr, 0, d_ok);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// Share validation by ONE random linear combination per message (opt-in fast path of the loop at
// examples/threshold_sig.rs:115-131): instead of N pairing checks  e(pk_i, H(m)) == e(g1, sig_i)  the device
// checks  e(sum_i r_i pk_i, H(m)) == e(g1, sum_i r_i sig_i)  with secret r_i (2^63 values), which holds whenever all N
// do and fails with probability >= 1 - 2^-63 otherwise; messages whose combined check fails are re-checked
// share by share, so ok[] equals the per-share path's (up to that 2^-63).
int tc_verify_shares_rlc_batch(tc_ctx* ctx, const uint8_t* pk_shares, size_t N, const uint8_t* sig_shares, const uint8_t* msgs,
                               const uint64_t* off, size_t B, const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback) try {
  TC_REQUIRE(ctx);
  if (n_fallback) *n_fallback = 0;
  if (B == 0 || N == 0) return TC_OK;
  TC_REQUIRE(ctx && pk_shares && sig_shares && off && seed32 && ok);
  TC_REQUIRE(B * N < (1ull << 32));
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || msgs);
  const uint8_t* d_pk = k.in(pk_shares, N * 96);
  const uint8_t* d_sig = k.in(sig_shares, B * N * 192);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  // the seed is host memory in both I/O modes (32 bytes; it must stay secret until the call returns)
  uint8_t* d_seed = k.temp<uint8_t>(32);
  if (d_seed) {
    k.check(hipMemcpyAsync(d_seed, seed32, 32, hipMemcpyHostToDevice, ctx->stream), "seed copy");
    ctx->h2d_bytes += 32;
    k.wipe.emplace_back(d_seed, 32);
  }
  uint8_t* d_r = k.temp<uint8_t>(B * N * 32);
  if (d_r) k.wipe.emplace_back(d_r, B * N * 32);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_S = k.temp<uint8_t>(B * 192);
  uint8_t* d_P = k.temp<uint8_t>(B * 96);
  uint8_t* d_stS = k.temp<uint8_t>(B);
  uint8_t* d_okmsg = k.temp<uint8_t>(B);
  uint8_t* d_ok = k.out(ok, B * N);
  k.begin_timing();
  k.check_points(false, d_pk, 96, N, N, 1, (size_t)-1);
  k.check_points(true, d_sig, 192, N, N, B, 1);
  std::vector<uint8_t> h_okmsg(B);
  if (!k.failed) {
    tc::launch_rlc_scalars(ctx->stream, d_seed, B * N, d_r);
    // (the two-stage kernels' short-scalar mode: 16 doublings for the 64-bit random scalars)
    k.check(hipMemsetAsync(d_stS, 0, B, ctx->stream), "memset");
    msm_g2(k, N, N * 192, d_sig, reinterpret_cast<const uint32_t*>(d_r), B, d_S, d_stS, /*nbits=*/16, tc::MsmFilter(), /*secret_scalars=*/true);
    tc::launch_lincomb_g1(ctx->stream, N, d_r, d_pk, B, d_P, nullptr, /*shared_points=*/true);
    tc::launch_hash_g2(ctx->tuning, ctx->stream, d_msgs, d_off, B, d_hash, /*fix=*/false);
    // e(P, [c] Q') == e(g1, S)  <=>  e(P, Q') == e([1/c] g1, S)   (the folded hash constant of tc_verify_sig_batch)
    tc::launch_pairing_check(ctx->stream, d_P, 96, d_hash, 192, ctx->g1_gen_unfix, 0, d_S, 192, B, d_okmsg, k.pairing_ws(B));
    // a share that does not decode (status below) or, in checked-input mode, is no group member sends its
    // message to the per-share fallback
    k.apply_checks(B, nullptr, nullptr, 0, d_okmsg);
    k.check(hipMemsetAsync(d_ok, 1, B * N, ctx->stream), "memset");
    k.check(hipMemcpyAsync(h_okmsg.data(), d_okmsg, B, hipMemcpyDeviceToHost, ctx->stream), "ok readback");
    std::vector<uint8_t> h_stS(B);
    k.check(hipMemcpyAsync(h_stS.data(), d_stS, B, hipMemcpyDeviceToHost, ctx->stream), "status readback");
    ctx->d2h_bytes += 2 * B;
    k.check(hipStreamSynchronize(ctx->stream), "stream sync");
    std::vector<uint32_t> failed;
    if (!k.failed)
      for (size_t j = 0; j < B; j++)
        if (!h_okmsg[j] || h_stS[j] != TC_JOB_OK) failed.push_back((uint32_t)j);
    if (!k.failed && !failed.empty()) {
      // per-share pairing checks for every share of the failed messages, on compacted operands
      const size_t F = failed.size(), R = F * N;
      std::vector<uint32_t> m_sig(R), m_hash(R), m_pk(R);
      for (size_t f = 0; f < F; f++)
        for (size_t i = 0; i < N; i++) {
          m_sig[f * N + i] = (uint32_t)(failed[f] * N + i);
          m_hash[f * N + i] = failed[f];
          m_pk[f * N + i] = (uint32_t)i;
        }
      uint32_t* d_maps = k.temp<uint32_t>(3 * R);
      uint8_t* c_sig = k.temp<uint8_t>(R * 192);
      uint8_t* c_hash = k.temp<uint8_t>(R * 192);
      uint8_t* c_pk = k.temp<uint8_t>(R * 96);
      uint8_t* c_ok = k.temp<uint8_t>(R);
      if (!k.failed) {
        k.check(hipMemcpyAsync(d_maps, m_sig.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        k.check(hipMemcpyAsync(d_maps + R, m_hash.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        k.check(hipMemcpyAsync(d_maps + 2 * R, m_pk.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        ctx->h2d_bytes += 12 * R;
        tc::launch_gather_rows(ctx->stream, d_sig, 192, d_maps, R, c_sig);
        tc::launch_gather_rows(ctx->stream, d_hash, 192, d_maps + R, R, c_hash);
        tc::launch_gather_rows(ctx->stream, d_pk, 96, d_maps + 2 * R, R, c_pk);
        tc::launch_pairing_check(ctx->stream, c_pk, 96, c_hash, 192, ctx->g1_gen_unfix, 0, c_sig, 192, R, c_ok, k.pairing_ws(R));
        if (ctx->input_checks) {  // members only, as the per-share path would require
          uint8_t* v = k.temp<uint8_t>(R);
          tc::launch_subgroup_check_g2(ctx->stream, c_sig, 192, 1, 1, R, v);
          tc::launch_invalidate_jobs(ctx->stream, v, 1, 1, R, nullptr, nullptr, 0, c_ok);
          tc::launch_subgroup_check_g1(ctx->stream, c_pk, 96, 1, 1, R, v);
          tc::launch_invalidate_jobs(ctx->stream, v, 1, 1, R, nullptr, nullptr, 0, c_ok);
        }
        tc::launch_scatter_bytes(ctx->stream, c_ok, d_maps, R, d_ok);
        k.check(hipStreamSynchronize(ctx->stream), "stream sync");  // the host maps go out of scope
      }
      if (n_fallback) *n_fallback = F;
    }
  }
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// Signature batches under ONE key (BASELINE config 3's shape: 65 536 verifies under the master key) by random linear
// combination, opt-in: the batch is cut into groups of `group` jobs, and a group passes with ONE check
//     e(pk, sum_j r_j H_j) == e(g1, sum_j r_j sig_j)          (bilinearity; r_j secret, 2^63 values each)
// instead of `group` checks of src/lib.rs:109; a group that fails (or holds an undecodable / non-member operand) is
// re-checked job by job, so ok[] equals tc_verify_g2_batch's up to the 2^-63 of a wrongly passing group.  The two sums
// are 16-column ladders over the psi-images (the two-stage kernels' short-scalar mode).  hash == nullptr: the hash points
// are made on the device from msgs / off first (tc_verify_sig_batch's composition, hash constant folded into g1).
static int verify_rlc(tc_ctx* ctx, const uint8_t* pk, const uint8_t* sig, const uint8_t* hash, const uint8_t* msgs, const uint64_t* off,
                      size_t B, size_t group, const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback) {
  TC_REQUIRE(ctx);
  if (n_fallback) *n_fallback = 0;
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pk && sig && (hash || off) && seed32 && ok);
  TC_REQUIRE(B < (1ull << 32));
  if (group == 0) group = 64;
  if (group > 1024) group = 1024;
  if (group > B) group = B;
  Call k(ctx);
  uint64_t total = 0;
  if (!hash) {
    if (!total_bytes(k, off, B, &total)) return k.finish();
    TC_REQUIRE(total == 0 || msgs);
  }
  const uint8_t* d_pk = k.in(pk, 96);
  const uint8_t* d_sig = k.in(sig, B * 192);
  const uint8_t* d_hash_in = hash ? k.in(hash, B * 192) : nullptr;
  const uint8_t* d_msgs = hash ? nullptr : k.in(msgs, (size_t)total);
  const uint64_t* d_off = hash ? nullptr : k.in(off, B + 1);
  uint8_t* d_hash_own = hash ? nullptr : k.temp<uint8_t>(B * 192);
  const uint8_t* d_hash = hash ? d_hash_in : d_hash_own;
  const uint8_t* d_g1 = hash ? ctx->g1_gen : ctx->g1_gen_unfix;
  uint8_t* d_seed = k.temp<uint8_t>(32);
  if (d_seed) {
    k.check(hipMemcpyAsync(d_seed, seed32, 32, hipMemcpyHostToDevice, ctx->stream), "seed copy");
    ctx->h2d_bytes += 32;
    k.wipe.emplace_back(d_seed, 32);
  }
  uint8_t* d_r = k.temp<uint8_t>(B * 32);
  if (d_r) k.wipe.emplace_back(d_r, B * 32);
  const size_t G = B / group, tail = B % group, NG = G + (tail ? 1 : 0);
  uint8_t* d_S = k.temp<uint8_t>(NG * 192);
  uint8_t* d_H = k.temp<uint8_t>(NG * 192);
  uint8_t* d_st = k.temp<uint8_t>(2 * NG);
  uint8_t* d_okg = k.temp<uint8_t>(NG);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, d_pk, 0, 1, 1, B, 1);
  k.check_points(true, d_sig, 192, 1, 1, B, 1);
  if (hash) k.check_points(true, d_hash, 192, 1, 1, B, 1);
  std::vector<uint8_t> h_okg(NG), h_st(2 * NG), h_valid;
  if (!k.failed) {
    if (!hash) tc::launch_hash_g2(ctx->tuning, ctx->stream, d_msgs, d_off, B, d_hash_own, /*fix=*/false);
    tc::launch_rlc_scalars(ctx->stream, d_seed, B, d_r);
    k.check(hipMemsetAsync(d_st, 0, 2 * NG, ctx->stream), "memset");
    const uint32_t* rr = reinterpret_cast<const uint32_t*>(d_r);
    if (G) {
      msm_g2(k, group, group * 192, d_sig, rr, G, d_S, d_st, /*nbits=*/16, tc::MsmFilter(), /*secret_scalars=*/true);
      msm_g2(k, group, group * 192, d_hash, rr, G, d_H, d_st + NG, 16, tc::MsmFilter(), true);
    }
    if (tail) {
      msm_g2(k, tail, tail * 192, d_sig + G * group * 192, rr + G * group * 8, 1, d_S + G * 192, d_st + G, 16, tc::MsmFilter(), true);
      msm_g2(k, tail, tail * 192, d_hash + G * group * 192, rr + G * group * 8, 1, d_H + G * 192, d_st + NG + G, 16, tc::MsmFilter(), true);
    }
    tc::launch_pairing_check(ctx->stream, d_pk, 0, d_H, 192, d_g1, 0, d_S, 192, NG, d_okg, k.pairing_ws(NG));
    k.check(hipMemsetAsync(d_ok, 1, B, ctx->stream), "memset");
    k.check(hipMemcpyAsync(h_okg.data(), d_okg, NG, hipMemcpyDeviceToHost, ctx->stream), "ok readback");
    k.check(hipMemcpyAsync(h_st.data(), d_st, 2 * NG, hipMemcpyDeviceToHost, ctx->stream), "status readback");
    ctx->d2h_bytes += 3 * NG;
    // checked-input mode: a group that owns a non-member operand goes to the per-job path as well
    std::vector<const uint8_t*> valid_ptrs;
    k.run_checks();  // (the membership tests ran beside the two sums and the group checks; their verdicts are read back below)
    for (auto& p : k.checks) valid_ptrs.push_back(p.valid);
    const size_t n_checks = k.checks.size();
    k.checks.clear();
    if (n_checks) {
      h_valid.resize(n_checks * B);
      for (size_t q = 0; q < n_checks; q++) {
        const bool shared = (q == 0);  // the first pending check is the key's (one byte for the whole batch)
        k.check(hipMemcpyAsync(h_valid.data() + q * B, valid_ptrs[q], shared ? 1 : B, hipMemcpyDeviceToHost, ctx->stream), "valid readback");
        ctx->d2h_bytes += shared ? 1 : B;
      }
    }
    k.check(hipStreamSynchronize(ctx->stream), "stream sync");
    std::vector<uint32_t> failed;
    if (!k.failed) {
      const bool key_bad = n_checks && h_valid[0] == 0;
      for (size_t g = 0; g < NG; g++) {
        bool bad = key_bad || !h_okg[g] || h_st[g] != TC_JOB_OK || h_st[NG + g] != TC_JOB_OK;
        const size_t lo = g * group, hi = (lo + group < B) ? lo + group : B;
        for (size_t q = 1; q < n_checks && !bad; q++)
          for (size_t j = lo; j < hi && !bad; j++) bad = h_valid[q * B + j] == 0;
        if (bad)
          for (size_t j = lo; j < hi; j++) failed.push_back((uint32_t)j);
      }
    }
    if (!k.failed && !failed.empty()) {
      // per-job pairing checks for every job of the failed groups, on compacted operands
      const size_t R = failed.size();
      uint32_t* d_map = k.temp<uint32_t>(R);
      uint8_t* c_sig = k.temp<uint8_t>(R * 192);
      uint8_t* c_hash = k.temp<uint8_t>(R * 192);
      uint8_t* c_ok = k.temp<uint8_t>(R);
      if (!k.failed) {
        k.check(hipMemcpyAsync(d_map, failed.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        ctx->h2d_bytes += 4 * R;
        tc::launch_gather_rows(ctx->stream, d_sig, 192, d_map, R, c_sig);
        tc::launch_gather_rows(ctx->stream, d_hash, 192, d_map, R, c_hash);
        tc::launch_pairing_check(ctx->stream, d_pk, 0, c_hash, 192, d_g1, 0, c_sig, 192, R, c_ok, k.pairing_ws(R));
        if (ctx->input_checks) {  // members only, as the per-job path would require
          uint8_t* v = k.temp<uint8_t>(R);
          tc::launch_subgroup_check_g2(ctx->stream, c_sig, 192, 1, 1, R, v);
          tc::launch_invalidate_jobs(ctx->stream, v, 1, 1, R, nullptr, nullptr, 0, c_ok);
          if (hash) {
            tc::launch_subgroup_check_g2(ctx->stream, c_hash, 192, 1, 1, R, v);
            tc::launch_invalidate_jobs(ctx->stream, v, 1, 1, R, nullptr, nullptr, 0, c_ok);
          }
          tc::launch_subgroup_check_g1(ctx->stream, d_pk, 0, 1, 1, 1, v);
          tc::launch_invalidate_jobs(ctx->stream, v, 1, (size_t)-1, R, nullptr, nullptr, 0, c_ok);
        }
        tc::launch_scatter_bytes(ctx->stream, c_ok, d_map, R, d_ok);
        k.check(hipStreamSynchronize(ctx->stream), "stream sync");  // the host map goes out of scope
      }
      if (n_fallback) *n_fallback = R;
    }
  }
  k.end_timing();
  return k.finish();
}

int tc_verify_g2_rlc_batch(tc_ctx* ctx, const uint8_t* pk, const uint8_t* sig, const uint8_t* hash, size_t B, size_t group,
                           const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback) try {
  if (ctx && !hash) {
    ctx->err = "invalid argument: hash";
    return TC_ERR_INVALID_ARG;
  }
  return verify_rlc(ctx, pk, sig, hash, nullptr, nullptr, B, group, seed32, ok, n_fallback);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_verify_sig_rlc_batch(tc_ctx* ctx, const uint8_t* pk, const uint8_t* sig, const uint8_t* msgs, const uint64_t* off, size_t B,
                            size_t group, const uint8_t* seed32, uint8_t* ok, uint64_t* n_fallback) try {
  if (ctx && !off) {
    ctx->err = "invalid argument: off";
    return TC_ERR_INVALID_ARG;
  }
  return verify_rlc(ctx, pk, sig, nullptr, msgs, off, B, group, seed32, ok, n_fallback);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_ciphertext_verify_batch(tc_ctx* ctx, const uint8_t* u, const uint8_t* v, const uint64_t* off,
                               const uint8_t* w, size_t B, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && u && off && w && ok);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || v);
  const uint8_t* d_u = k.in(u, B * 96);
  const uint8_t* d_v = k.in(v, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  const uint8_t* d_w = k.in(w, B * 192);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, d_u, 96, 1, 1, B, 1);
  k.check_points(true, d_w, 192, 1, 1, B, 1);
  if (!k.failed) {
    // an undecodable u leaves an infinity hash; the pairing kernel then rejects u itself
    tc::launch_hash_g1_g2(ctx->tuning, ctx->stream, d_u, d_v, d_off, B, d_hash, nullptr, /*fix=*/false);
    // e(g1, w) == e(u, [c] Q')  <=>  e([1/c] g1, w) == e(u, Q')         (src/lib.rs:511)
    tc::launch_pairing_check(ctx->stream, ctx->g1_gen_unfix, 0, d_w, 192, d_u, 96, d_hash, 192, B, d_ok, k.pairing_ws(B));
  }
  k.apply_checks(B, nullptr, nullptr, 0, d_ok);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// Ciphertext::verify, then [sk] u for the ciphertexts that pass: SecretKeyShare::decrypt_share (src/lib.rs:452-457) when
// plain == nullptr, SecretKey::decrypt (src/lib.rs:384-391: the same, then xor_with_hash) otherwise
static int verified_decrypt(tc_ctx* ctx, const uint8_t* sk, const uint8_t* u, const uint8_t* v, const uint64_t* off, const uint8_t* w,
                            size_t B, uint8_t* out_g1, uint8_t* plain, uint8_t* ok, bool plain_unbacked = false) {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && sk && u && off && w && ok && (out_g1 || plain));
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || (v && !plain_unbacked));
  const uint8_t* d_sk = k.in(sk, 32, /*secret=*/true);
  const uint8_t* d_u = k.in(u, B * 96);
  const uint8_t* d_v = k.in(v, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  const uint8_t* d_w = k.in(w, B * 192);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_ok = k.out(ok, B);
  uint8_t* d_st = k.temp<uint8_t>(B, /*zero=*/true);
  uint8_t* d_pt = plain ? k.temp<uint8_t>(B * 96) : k.out(out_g1, B * 96);
  uint8_t* d_plain = plain ? k.out(plain, (size_t)total, /*zero=*/true) : nullptr;
  if (plain && d_pt) k.wipe_after_copy.emplace_back(d_pt, B * 96);  // g = [sk] u opens the ciphertext: not left in a staging slot
  k.begin_timing();
  // ALWAYS, whatever tc_ctx_set_input_checks says: the pairing check cannot see a small-order component of u
  // (u = u1 + T, T in E(Fq)[h1]: e(u, H) = e(u1, H)), and [sk] u would leak sk modulo the cofactor's small factors.  The
  // reference cannot reach that state: Ciphertext deserialisation always runs the subgroup test (src/lib.rs:140-146).
  k.check_points(false, d_u, 96, 1, 1, B, 1, /*always=*/true);
  k.check_points(true, d_w, 192, 1, 1, B, 1, /*always=*/true);
  if (!k.failed) {
    tc::launch_hash_g1_g2(ctx->tuning, ctx->stream, d_u, d_v, d_off, B, d_hash, nullptr, /*fix=*/false);
    tc::launch_pairing_check(ctx->stream, ctx->g1_gen_unfix, 0, d_w, 192, d_u, 96, d_hash, 192, B, d_ok, k.pairing_ws(B));  // src/lib.rs:511
    k.apply_checks(B, nullptr, nullptr, 0, d_ok);
    tc::launch_g1_mul(ctx->stream, k.tables(), d_sk, d_u, 1, B, d_pt, d_st);
    // a multiplication that reported an error (a non-canonical secret key: >= r) is not a decryption: ok = 0, like the
    // reference, which cannot hold such an Fr
    tc::launch_ok_and_status(ctx->stream, d_st, B, d_ok);
    // `if !ct.verify() { return None; }` (src/lib.rs:385, 453): a ciphertext that fails the check gets the identity, never [sk] u
    tc::launch_invalidate_jobs(ctx->stream, d_ok, 1, 1, B, d_st, d_pt, 96, nullptr);
    if (plain) tc::launch_xor_with_hash(ctx->stream, d_pt, d_v, d_off, B, d_plain, d_st);  // skips flagged jobs: zeros
  }
  k.end_timing();
  return k.finish();
}

int tc_decrypt_share_batch(tc_ctx* ctx, const uint8_t* sk_fr, const uint8_t* u, const uint8_t* v, const uint64_t* off, const uint8_t* w,
                           size_t B, uint8_t* out_g1, uint8_t* ok) try {
  TC_REQUIRE(ctx && (B == 0 || out_g1));
  return verified_decrypt(ctx, sk_fr, u, v, off, w, B, out_g1, nullptr, ok);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_secret_key_decrypt_batch(tc_ctx* ctx, const uint8_t* sk_fr, const uint8_t* u, const uint8_t* v, const uint64_t* off, const uint8_t* w,
                                size_t B, uint8_t* out, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  const bool unbacked = out == nullptr;  // (all-empty plaintexts, device-resident: see g_no_plaintext_bytes)
  if (unbacked) out = &g_no_plaintext_bytes;
  return verified_decrypt(ctx, sk_fr, u, v, off, w, B, nullptr, out, ok, unbacked);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_verify_decryption_share_batch(tc_ctx* ctx, const uint8_t* pk_share, size_t pk_stride, const uint8_t* share,
                                     const uint8_t* u, const uint8_t* v, const uint64_t* off, const uint8_t* w,
                                     size_t B, uint8_t* ok) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pk_share && share && u && off && w && ok);
  TC_REQUIRE(pk_stride == 0 || pk_stride >= 96);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || v);
  const uint8_t* d_pk = k.in(pk_share, pk_stride ? (B - 1) * pk_stride + 96 : 96);
  const uint8_t* d_share = k.in(share, B * 96);
  const uint8_t* d_u = k.in(u, B * 96);
  const uint8_t* d_v = k.in(v, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  const uint8_t* d_w = k.in(w, B * 192);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_st = k.temp<uint8_t>(B);
  uint8_t* d_sharec = k.temp<uint8_t>(B * 96);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  k.check_points(false, d_pk, pk_stride, 1, 1, B, 1);
  k.check_points(false, d_share, 96, 1, 1, B, 1);
  k.check_points(false, d_u, 96, 1, 1, B, 1);
  k.check_points(true, d_w, 192, 1, 1, B, 1);
  if (!k.failed) {
    tc::launch_hash_g1_g2(ctx->tuning, ctx->stream, d_u, d_v, d_off, B, d_hash, d_st, /*fix=*/false);
    tc::launch_g1_scale_cofactor_fix(ctx->stream, d_share, 96, B, d_sharec);
    // e(share, hash) = e([c] share, Q') == e(pk_share, w)               (src/lib.rs:185)
    tc::launch_pairing_check(ctx->stream, d_sharec, 96, d_hash, 192, d_pk, pk_stride, d_w, 192, B, d_ok, k.pairing_ws(B));
  }
  k.apply_checks(B, nullptr, nullptr, 0, d_ok);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// Decryption-share validation by ONE random linear combination per ciphertext (opt-in fast path of the loop of
// examples/threshold_enc.rs over PublicKeyShare::verify_decryption_share, src/lib.rs:182-186): instead of N checks
// e(share_i, H) == e(pk_i, w) per ciphertext (H = hash_g1_g2(u, v)) the device checks
//     e(sum_i r_i share_i, H) == e(sum_i r_i pk_i, w)
// with secret r_i = a_i + b_i x^2 (2^63 values; a 16-step ladder through phi); ciphertexts whose combined check fails are
// re-checked share by share, so ok[] equals the per-share path's up to 2^-63.
int tc_verify_decryption_shares_rlc_batch(tc_ctx* ctx, const uint8_t* pk_shares, size_t N, const uint8_t* shares, const uint8_t* u,
                                          const uint8_t* v, const uint64_t* off, const uint8_t* w, size_t B, const uint8_t* seed32,
                                          uint8_t* ok, uint64_t* n_fallback) try {
  TC_REQUIRE(ctx);
  if (n_fallback) *n_fallback = 0;
  if (B == 0 || N == 0) return TC_OK;
  TC_REQUIRE(ctx && pk_shares && shares && u && off && w && seed32 && ok);
  TC_REQUIRE(B * N < (1ull << 32));
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || v);
  const uint8_t* d_pk = k.in(pk_shares, N * 96);
  const uint8_t* d_sh = k.in(shares, B * N * 96);
  const uint8_t* d_u = k.in(u, B * 96);
  const uint8_t* d_v = k.in(v, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  const uint8_t* d_w = k.in(w, B * 192);
  uint8_t* d_seed = k.temp<uint8_t>(32);
  if (d_seed) {
    k.check(hipMemcpyAsync(d_seed, seed32, 32, hipMemcpyHostToDevice, ctx->stream), "seed copy");
    ctx->h2d_bytes += 32;
    k.wipe.emplace_back(d_seed, 32);
  }
  uint8_t* d_r = k.temp<uint8_t>(B * N * 32);
  if (d_r) k.wipe.emplace_back(d_r, B * N * 32);
  uint8_t* d_hash = k.temp<uint8_t>(B * 192);
  uint8_t* d_D = k.temp<uint8_t>(B * 96);
  uint8_t* d_Dc = k.temp<uint8_t>(B * 96);
  uint8_t* d_P = k.temp<uint8_t>(B * 96);
  uint8_t* d_st = k.temp<uint8_t>(3 * B);
  uint8_t* d_okct = k.temp<uint8_t>(B);
  uint8_t* d_ok = k.out(ok, B * N);
  k.begin_timing();
  k.check_points(false, d_pk, 96, N, N, 1, (size_t)-1);
  k.check_points(false, d_sh, 96, N, N, B, 1);
  k.check_points(false, d_u, 96, 1, 1, B, 1);
  k.check_points(true, d_w, 192, 1, 1, B, 1);
  std::vector<uint8_t> h_okct(B), h_st(3 * B);
  if (!k.failed) {
    tc::launch_rlc_scalars_g1(ctx->stream, d_seed, B * N, d_r);
    k.check(hipMemsetAsync(d_st, 0, 3 * B, ctx->stream), "memset");
    const uint32_t* rr = reinterpret_cast<const uint32_t*>(d_r);
    msm_g1(k, N, N * 96, d_sh, rr, B, d_D, d_st, /*nbits=*/32, /*secret_scalars=*/true);
    msm_g1(k, N, 0, d_pk, rr, B, d_P, d_st + B, 32, true);   // the N key shares are the same for every ciphertext: one table set
    tc::launch_hash_g1_g2(ctx->tuning, ctx->stream, d_u, d_v, d_off, B, d_hash, d_st + 2 * B, /*fix=*/false);
    tc::launch_g1_scale_cofactor_fix(ctx->stream, d_D, 96, B, d_Dc);
    // e(D, [c] Q') = e([c] D, Q') == e(P, w)      (the folded hash constant of tc_verify_decryption_share_batch)
    tc::launch_pairing_check(ctx->stream, d_Dc, 96, d_hash, 192, d_P, 96, d_w, 192, B, d_okct, k.pairing_ws(B));
    k.apply_checks(B, nullptr, nullptr, 0, d_okct);  // checked-input mode: a ciphertext owning a non-member operand falls back
    k.check(hipMemsetAsync(d_ok, 1, B * N, ctx->stream), "memset");
    k.check(hipMemcpyAsync(h_okct.data(), d_okct, B, hipMemcpyDeviceToHost, ctx->stream), "ok readback");
    k.check(hipMemcpyAsync(h_st.data(), d_st, 3 * B, hipMemcpyDeviceToHost, ctx->stream), "status readback");
    ctx->d2h_bytes += 4 * B;
    k.check(hipStreamSynchronize(ctx->stream), "stream sync");
    std::vector<uint32_t> failed;
    if (!k.failed)
      for (size_t j = 0; j < B; j++)
        if (!h_okct[j] || h_st[j] != TC_JOB_OK || h_st[B + j] != TC_JOB_OK || h_st[2 * B + j] != TC_JOB_OK) failed.push_back((uint32_t)j);
    if (!k.failed && !failed.empty()) {
      // per-share pairing checks for every share of the failed ciphertexts, on compacted operands
      const size_t F = failed.size(), R = F * N;
      std::vector<uint32_t> m_sh(R), m_ct(R), m_pk(R);
      for (size_t f = 0; f < F; f++)
        for (size_t i = 0; i < N; i++) {
          m_sh[f * N + i] = (uint32_t)(failed[f] * N + i);
          m_ct[f * N + i] = failed[f];
          m_pk[f * N + i] = (uint32_t)i;
        }
      uint32_t* d_maps = k.temp<uint32_t>(3 * R);
      uint8_t* c_sh = k.temp<uint8_t>(R * 96);
      uint8_t* c_shc = k.temp<uint8_t>(R * 96);
      uint8_t* c_hash = k.temp<uint8_t>(R * 192);
      uint8_t* c_w = k.temp<uint8_t>(R * 192);
      uint8_t* c_pk = k.temp<uint8_t>(R * 96);
      uint8_t* c_ok = k.temp<uint8_t>(R);
      if (!k.failed) {
        k.check(hipMemcpyAsync(d_maps, m_sh.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        k.check(hipMemcpyAsync(d_maps + R, m_ct.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        k.check(hipMemcpyAsync(d_maps + 2 * R, m_pk.data(), R * 4, hipMemcpyHostToDevice, ctx->stream), "map copy");
        ctx->h2d_bytes += 12 * R;
        tc::launch_gather_rows(ctx->stream, d_sh, 96, d_maps, R, c_sh);
        tc::launch_gather_rows(ctx->stream, d_hash, 192, d_maps + R, R, c_hash);
        tc::launch_gather_rows(ctx->stream, d_w, 192, d_maps + R, R, c_w);
        tc::launch_gather_rows(ctx->stream, d_pk, 96, d_maps + 2 * R, R, c_pk);
        tc::launch_g1_scale_cofactor_fix(ctx->stream, c_sh, 96, R, c_shc);
        tc::launch_pairing_check(ctx->stream, c_shc, 96, c_hash, 192, c_pk, 96, c_w, 192, R, c_ok, k.pairing_ws(R));
        if (ctx->input_checks) {  // members only, as the per-share path would require
          uint8_t* vv = k.temp<uint8_t>(R);
          tc::launch_subgroup_check_g1(ctx->stream, c_sh, 96, 1, 1, R, vv);
          tc::launch_invalidate_jobs(ctx->stream, vv, 1, 1, R, nullptr, nullptr, 0, c_ok);
          tc::launch_subgroup_check_g1(ctx->stream, c_pk, 96, 1, 1, R, vv);
          tc::launch_invalidate_jobs(ctx->stream, vv, 1, 1, R, nullptr, nullptr, 0, c_ok);
          tc::launch_subgroup_check_g2(ctx->stream, c_w, 192, 1, 1, R, vv);
          tc::launch_invalidate_jobs(ctx->stream, vv, 1, 1, R, nullptr, nullptr, 0, c_ok);
          // ... and u, which tc_verify_decryption_share_batch checks too (an on-curve u outside G1 used to
          // reach the bare pairing result here)
          uint8_t* c_u = k.temp<uint8_t>(R * 96);
          if (c_u) {
            tc::launch_gather_rows(ctx->stream, d_u, 96, d_maps + R, R, c_u);
            tc::launch_subgroup_check_g1(ctx->stream, c_u, 96, 1, 1, R, vv);
            tc::launch_invalidate_jobs(ctx->stream, vv, 1, 1, R, nullptr, nullptr, 0, c_ok);
          }
        }
        tc::launch_scatter_bytes(ctx->stream, c_ok, d_maps, R, d_ok);
        k.check(hipStreamSynchronize(ctx->stream), "stream sync");  // the host maps go out of scope
      }
      if (n_fallback) *n_fallback = F;
    }
  }
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- DKG algebra -----------------------------------------------------------------------------------
int tc_g1_commitment_batch(tc_ctx* ctx, const uint8_t* coeff_fr, size_t M, uint8_t* out, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (M == 0) return TC_OK;
  TC_REQUIRE(ctx && coeff_fr && out);
  Call k(ctx);
  if (!ctx->fb_table && !k.failed) {
    if (k.check(hipMalloc((void**)&ctx->fb_table, tc::fixed_base_table_bytes()), "hipMalloc"))
      tc::launch_fixed_base_table(ctx->stream, ctx->fb_table);
    else
      ctx->fb_table = nullptr;
  }
  const uint8_t* d_fr = k.in(coeff_fr, M * 32, /*secret=*/true);
  uint8_t* d_out = k.out(out, M * 96);
  uint8_t* d_st = k.out(status, M);
  k.begin_timing();
  if (!k.failed) tc::launch_g1_fixed_base(ctx->stream, ctx->fb_table, d_fr, M, d_out, d_st, ctx->cus);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_bivar_commitment_row_batch(tc_ctx* ctx, const uint8_t* commit, size_t degree, const uint64_t* xs, size_t M, uint8_t* out,
                                  uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (M == 0) return TC_OK;
  TC_REQUIRE(ctx && commit && xs && out);
  TC_REQUIRE(degree < (1u << 16));
  const size_t ncoeff = (degree + 1) * (degree + 2) / 2;
  Call k(ctx);
  const uint8_t* d_c = k.in(commit, ncoeff * 96);
  const uint64_t* d_x = k.in(xs, M);
  uint8_t* d_out = k.out(out, M * (degree + 1) * 96);
  uint8_t* d_st = k.out(status, M * (degree + 1));
  k.begin_timing();
  k.check_points(false, d_c, 96, ncoeff, ncoeff, 1, (size_t)-1);
  if (!k.failed) tc::launch_bivar_commitment_row(ctx->stream, d_c, degree, d_x, M, d_out, d_st);
  k.apply_checks(M * (degree + 1), d_st, d_out, 96, nullptr);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_fr_interpolate_batch(tc_ctx* ctx, size_t n, const uint8_t* xs, const uint8_t* ys, size_t B, uint8_t* out_coeff, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0 || n == 0) return TC_OK;
  TC_REQUIRE(ctx && xs && ys && out_coeff);
  TC_REQUIRE(n < (1u << 16));
  Call k(ctx);
  const uint8_t* d_x = k.in(xs, B * n * 32);
  const uint8_t* d_y = k.in(ys, B * n * 32, /*secret=*/true);
  uint8_t* d_out = k.out(out_coeff, B * n * 32);
  // host-I/O mode: the interpolated (secret) coefficients pass through a staging slot that outlives the call: wiped after
  // the copy back, like every other secret operand
  if (d_out && !ctx->device_io) k.wipe_after_copy.emplace_back(d_out, B * n * 32);
  uint32_t* d_ws = k.temp<uint32_t>(B * 2 * (n + 1) * 8);
  if (d_ws) k.wipe.emplace_back(d_ws, B * 2 * (n + 1) * 8 * sizeof(uint32_t));
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  if (!k.failed)
    tc::launch_fr_interpolate(ctx->stream, n, (const uint32_t*)d_x, (const uint32_t*)d_y, B, (uint32_t*)d_out, d_ws, d_st);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- membership tests ---------------------------------------------------------------------------
static int subgroup_check(tc_ctx* ctx, bool g2, const uint8_t* pts, size_t B, uint8_t* ok) {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pts && ok);
  const size_t PB = g2 ? 192 : 96;
  Call k(ctx);
  const uint8_t* d_pts = k.in(pts, B * PB);
  uint8_t* d_ok = k.out(ok, B);
  k.begin_timing();
  if (!k.failed) {
    if (g2) tc::launch_subgroup_check_g2(ctx->stream, d_pts, PB, 1, 1, B, d_ok);
    else tc::launch_subgroup_check_g1(ctx->stream, d_pts, PB, 1, 1, B, d_ok);
  }
  k.end_timing();
  return k.finish();
}
int tc_g1_subgroup_check_batch(tc_ctx* ctx, const uint8_t* pts96, size_t B, uint8_t* ok) try {
  return subgroup_check(ctx, false, pts96, B, ok);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}
int tc_g2_subgroup_check_batch(tc_ctx* ctx, const uint8_t* pts192, size_t B, uint8_t* ok) try {
  return subgroup_check(ctx, true, pts192, B, ok);
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

// ---- wire formats ------------------------------------------------------------------------------
int tc_g1_compress_batch(tc_ctx* ctx, const uint8_t* in96, size_t B, uint8_t* out48, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && in96 && out48);
  Call k(ctx);
  const uint8_t* d_in = k.in(in96, B * 96);
  uint8_t* d_out = k.out(out48, B * 48);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  if (!k.failed) tc::launch_g1_compress(ctx->stream, d_in, B, d_out, d_st);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_g2_compress_batch(tc_ctx* ctx, const uint8_t* in192, size_t B, uint8_t* out96, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && in192 && out96);
  Call k(ctx);
  const uint8_t* d_in = k.in(in192, B * 192);
  uint8_t* d_out = k.out(out96, B * 96);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  if (!k.failed) tc::launch_g2_compress(ctx->stream, d_in, B, d_out, d_st);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_encrypt_batch(tc_ctx* ctx, const uint8_t* pk, size_t pk_stride, const uint8_t* r, const uint8_t* msgs,
                     const uint64_t* off, size_t B, uint8_t* out_u, uint8_t* out_v, uint8_t* out_w, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && pk && r && off && out_u && out_w);
  TC_REQUIRE(pk_stride == 0 || pk_stride >= 96);
  Call k(ctx);
  uint64_t total = 0;
  if (!total_bytes(k, off, B, &total)) return k.finish();
  TC_REQUIRE(total == 0 || (msgs && out_v));  // (all-empty messages: no v bytes, and a device-resident empty buffer has no address)
  const uint8_t* d_pk = k.in(pk, pk_stride ? (B - 1) * pk_stride + 96 : 96);
  const uint8_t* d_r = k.in(r, B * 32, /*secret=*/true);
  const uint8_t* d_msgs = k.in(msgs, (size_t)total);
  const uint64_t* d_off = k.in(off, B + 1);
  uint8_t* d_u = k.out(out_u, B * 96);
  uint8_t* d_v = k.out(out_v, (size_t)total);
  uint8_t* d_w = k.out(out_w, B * 192);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  k.check_points(false, d_pk, pk_stride, 1, 1, B, 1);
  if (!k.failed) tc::launch_encrypt(ctx->stream, k.tables(), d_pk, pk_stride, d_r, d_msgs, d_off, B, d_u, d_v, d_w, d_st);
  if (!k.checks.empty()) {
    // an invalid key fails the job: status + identity u and w (v keeps the kernel's bytes and must be ignored)
    auto pending = k.checks;
    for (auto& p : pending) p.pts = nullptr;  // (the first apply_checks launches the tests; the second only applies their verdicts)
    k.apply_checks(B, d_st, d_u, 96, nullptr);
    k.checks = pending;
    k.apply_checks(B, nullptr, d_w, 192, nullptr);
  }
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_public_key_share_batch(tc_ctx* ctx, const uint8_t* commit, size_t t, const uint64_t* idx, size_t M, uint8_t* out,
                              uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (M == 0) return TC_OK;
  TC_REQUIRE(ctx && commit && idx && out);
  TC_REQUIRE(t < (1u << 20));
  Call k(ctx);
  const uint8_t* d_c = k.in(commit, (t + 1) * 96);
  const uint64_t* d_idx = k.in(idx, M);
  uint8_t* d_out = k.out(out, M * 96);
  uint8_t* d_st = k.out(status, M);
  k.begin_timing();
  k.check_points(false, d_c, 96, t + 1, t + 1, 1, (size_t)-1);  // the t+1 coefficients, shared by every index
  if (!k.failed) tc::launch_commitment_evaluate(ctx->stream, d_c, t, d_idx, M, d_out, d_st);
  k.apply_checks(M, d_st, d_out, 96, nullptr);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_g1_decompress_batch(tc_ctx* ctx, const uint8_t* in48, size_t B, uint8_t* out96, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && in48 && out96);
  Call k(ctx);
  const uint8_t* d_in = k.in(in48, B * 48);
  uint8_t* d_out = k.out(out96, B * 96);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  if (!k.failed) tc::launch_g1_decompress(ctx->stream, d_in, B, d_out, d_st);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

int tc_g2_decompress_batch(tc_ctx* ctx, const uint8_t* in96, size_t B, uint8_t* out192, uint8_t* status) try {
  TC_REQUIRE(ctx);
  if (B == 0) return TC_OK;
  TC_REQUIRE(ctx && in96 && out192);
  Call k(ctx);
  const uint8_t* d_in = k.in(in96, B * 96);
  uint8_t* d_out = k.out(out192, B * 192);
  uint8_t* d_st = k.out(status, B);
  k.begin_timing();
  if (!k.failed) tc::launch_g2_decompress(ctx->tuning, ctx->stream, d_in, B, d_out, d_st);
  k.end_timing();
  return k.finish();
} catch (...) {
  return on_exception((tc_ctx*)ctx);
}

}  // extern "C"
