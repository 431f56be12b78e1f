// Per-job point tables in HBM: the "table arena".
//
// A windowed ladder over a VARIABLE base needs a table per job (8 affine G2 points for the sign-aligned
// 4-dimensional GLS ladder of tc_gls.h: 1.8 KB per lane pair), and every column reads the entry its own digits
// select.  160 KB of LDS per CU is 320 B per lane at two waves per SIMD: the tables cannot live there.  Held as a
// private array they live in scratch memory, which the hardware interleaves dword by dword across the 64 lanes
// of a wave: a look-up with a PER-LANE index then touches one 256-byte row per lane and limb -- 28 x 64 cache lines
// for 28 x 64 x 4 useful bytes (measured on k_point_mul<Fq2>: 15 % of all wave cycles parked in s_waitcnt, 5.8 GB
// fetched per 65 536-job combine launch for 1.1 GB of entries).
//
// So the tables go to global memory in the layout the look-up wants: one entry = 256 contiguous bytes
// (x of the even lane, x of the odd lane, y even, y odd: four 64-byte rows of 14 limbs + 2 words), a lane reads its
// two rows with 8 global_load_dwordx4, a lane pair's entry is two full 128-byte lines.  The arena does not grow
// with the batch: a wave borrows a SLOT (32 lane pairs x kPairTableWords) for its lifetime --
//   * slots are partitioned by XCD (HW_REG_XCC_ID): a slot is only ever touched through ONE L2, so no
//     cross-XCD write-back/invalidate is needed when it passes to the next wave;
//   * inside the partition a wave claims a free slot with one atomicCAS on a flag word (linear probing from
//     blockIdx / 8; at most kSlotsPerXcc = 512 waves of these 256-register kernels are resident per XCD while
//     256 fit) and clears the flag after a __threadfence() when it is done;
//   * the slot's base address is kept in LDS (8 bytes per workgroup), where the out-of-line ladder routines
//     find it without a parameter threaded through every caller.
// 8 XCDs x 512 slots x 64 KB = 256 MB per context (tc_api.hip), and the working set of the waves that are
// resident (2048 x 64 KB = 128 MB) fits the 256 MB Infinity Cache.
//
// The g++ test build (tests/hostsim) keeps one table per host thread.
#pragma once
#include "tc_arena.h"
#include "tc_curve.h"

namespace tc {

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) int32_t tbl_word;  // global address space: global_load/store, not flat
__shared__ tbl_word* tc_wave_tables;

// this lane pair's table (kBlock = 64: one wave per workgroup)
__device__ __forceinline__ tbl_word* pair_table() { return tc_wave_tables + (threadIdx.x >> 1) * kPairTableWords; }

// G1 kernels (one lane per job): the slot holds one 8-entry table per LANE -- 8 x 128 bytes (x, y: 14 limbs each + 4 words of
// padding: a full-line entry, read with seven global_load_dwordx4) = 1 KB per lane, 64 KB per wave, the same slot size.
__device__ __forceinline__ tbl_word* lane_table() { return tc_wave_tables + threadIdx.x * kLaneTableWords; }

// Call once per wave, in convergent code, before the first table is built; returns the slot for table_slot_release.
__device__ __forceinline__ uint32_t table_slot_acquire(const TableArena& ta) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (kTableXccs - 1);  // HW_REG_XCC_ID [3:0]
  uint32_t slot = 0;
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {  // first active lane
    uint32_t i = (blockIdx.x >> 3) & (kSlotsPerXcc - 1);
    uint32_t probes = 0;
    while (atomicCAS(ta.flags + xcc * kSlotsPerXcc + i, 0u, 1u) != 0u) {
      i = (i + 1) & (kSlotsPerXcc - 1);
      // cannot happen (more slots than resident waves); a corrupted flag array must fail the launch, not hang the GPU
      if (++probes > (64u << 20)) __builtin_trap();
    }
    slot = xcc * kSlotsPerXcc + i;
    tc_wave_tables = (tbl_word*)(ta.mem + (size_t)slot * kWaveTableWords);
  }
  slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  __syncthreads();  // one wave: orders the LDS write before the other lanes' reads
  return slot;
}
__device__ __forceinline__ void table_slot_release(const TableArena& ta, uint32_t slot) {
  __threadfence();  // this wave's table stores have landed before the next owner writes
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicExch(ta.flags + slot, 0u);
}
#else  // g++ test build, and the host pass of hipcc (which only has to parse)
typedef int32_t tbl_word;
TC_HD tbl_word* pair_table() {
  static thread_local tbl_word buf[kPairTableWords];
  return buf;
}
TC_HD tbl_word* lane_table() {
  static thread_local tbl_word buf[kLaneTableWords];
  return buf;
}
TC_HD uint32_t table_slot_acquire(const TableArena&) { return 0; }
TC_HD void table_slot_release(const TableArena&, uint32_t) {}
#endif

// this lane's half of an affine G2 point -> its two 64-byte rows of a table entry (hipcc: one coefficient per
// lane of the pair; g++ test build: both coefficients)
TC_HD void tbl_store_g2(tbl_word* e, const G2Affine& p) {
#if TC_PAIR
  const int o = pair_odd() * kTblCoordWords;
  const Fq x = p.x.m.norm(), y = p.y.m.norm();
#if defined(TC_BOUND_CHECK)
  if (x.val() > 2.1f || y.val() > 2.1f) tc_bound_fail(x.val(), y.val());
#endif
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    e[o + i] = x.l[i];
    e[2 * kTblCoordWords + o + i] = y.l[i];
  }
  e[o + 14] = 0;
  e[o + 15] = p.inf ? 1 : 0;
  e[2 * kTblCoordWords + o + 14] = 0;
  e[2 * kTblCoordWords + o + 15] = 0;
#else
  const Fq xs[2] = {p.x.c0.norm(), p.x.c1.norm()}, ys[2] = {p.y.c0.norm(), p.y.c1.norm()};
  for (int h = 0; h < 2; h++) {
#if defined(TC_BOUND_CHECK)
    if (xs[h].val() > 2.1f || ys[h].val() > 2.1f) tc_bound_fail(xs[h].val(), ys[h].val());
#endif
    for (int i = 0; i < FQ_LIMBS; i++) {
      e[h * kTblCoordWords + i] = xs[h].l[i];
      e[2 * kTblCoordWords + h * kTblCoordWords + i] = ys[h].l[i];
    }
    e[h * kTblCoordWords + 14] = 0;
    e[h * kTblCoordWords + 15] = p.inf ? 1 : 0;
    e[2 * kTblCoordWords + h * kTblCoordWords + 14] = 0;
    e[2 * kTblCoordWords + h * kTblCoordWords + 15] = 0;
  }
#endif
}
// a stored coordinate: limbs carry-normalised, value below 2.1 p (checked at the store in the bound-check build)
TC_HD Fq tbl_load_fq(const tbl_word* w) {
  Fq r;
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) r.l[i] = w[i];
  r.set_range(-0.001f, 1.001f);
  r.set_val(2.1f);
  return r;
}
// an affine G1 point (of the common-Z curve of a ladder table) <-> its 128-byte entry
TC_HD void tbl_store_g1(tbl_word* e, const G1Affine& p) {
  const Fq x = p.x.norm(), y = p.y.norm();
#if defined(TC_BOUND_CHECK)
  if (x.val() > 2.1f || y.val() > 2.1f) tc_bound_fail(x.val(), y.val());
#endif
  TC_UNROLL for (int i = 0; i < FQ_LIMBS; i++) {
    e[i] = x.l[i];
    e[FQ_LIMBS + i] = y.l[i];
  }
}
TC_HD G1Affine tbl_load_g1(const tbl_word* e) { return G1Affine{tbl_load_fq(e), tbl_load_fq(e + FQ_LIMBS), false}; }
TC_HD G2Affine tbl_load_g2(const tbl_word* e) {
  G2Affine p;
#if TC_PAIR
  const int o = pair_odd() * kTblCoordWords;
  p.x = Fq2{tbl_load_fq(e + o)};
  p.y = Fq2{tbl_load_fq(e + 2 * kTblCoordWords + o)};
  p.inf = e[o + 15] != 0;
#else
  p.x = Fq2::make(tbl_load_fq(e), tbl_load_fq(e + kTblCoordWords));
  p.y = Fq2::make(tbl_load_fq(e + 2 * kTblCoordWords), tbl_load_fq(e + 3 * kTblCoordWords));
  p.inf = e[15] != 0;
#endif
  return p;
}

}  // namespace tc
