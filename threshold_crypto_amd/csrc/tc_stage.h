// Whole-wave operand staging through LDS (hipcc only): the C ABI hands over array-of-structs records (one
// 96/192-byte point per job), and a job is one lane (G1) or one lane pair (G2), so lane-by-lane access to "its"
// record is a strided byte gather.  Here the 64 lanes of the wave load the wave's rows TOGETHER -- consecutive
// lanes take consecutive 8-byte words of a record, every load instruction covers whole 96/192-byte runs -- into a
// row-per-job LDS buffer, each lane then parses its own row out of LDS; results travel the same way back
// (encode into the LDS row, one cooperative store).  One buffer of 64 / L rows is reused for every operand.
#pragma once
#include "tc_jobs.h"

namespace tc {

#if defined(__HIPCC__)  // (the host pass of hipcc parses the kernels too)
template <int PB, int L>
struct WaveRowIO {
  static constexpr int JOBS = 64 / L;        // jobs of one wave
  static constexpr int ROW = PB + 16;        // LDS row pitch: the pad spreads consecutive rows over the banks
  static constexpr int WORDS = PB / 8;       // 8-byte words per record (the C ABI guarantees 8-byte alignment)
  static constexpr int BYTES = JOBS * ROW;   // LDS bytes per wave
  uint8_t* lds;        // this wave's buffer
  const uint8_t* in;   // this lane's job: operand k is the record at in + k * stride; nullptr = no job
  size_t stride;
  uint8_t* out;        // this lane's job: result record; nullptr = none

  __device__ static const uint8_t* lane_ptr(const uint8_t* p, int src_lane) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src_lane, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src_lane, 64);
    return (const uint8_t*)((uint64_t)lo | ((uint64_t)hi << 32));
  }
  // stage operand k of every job of the wave; returns this lane's row
  __device__ const uint8_t* operand(int k) {
    const uint8_t* mine = in ? in + (size_t)k * stride : nullptr;
    const int lane = (int)(threadIdx.x & 63);
    __syncthreads();  // one-wave workgroup: orders the previous operand's LDS reads before these writes
    TC_NOUNROLL for (int f = lane; f < JOBS * WORDS; f += 64) {
      const int s = f / WORDS, w = f % WORDS;
      const uint8_t* p = lane_ptr(mine, s * L);
      if (p) reinterpret_cast<uint64_t*>(lds + s * ROW)[w] = reinterpret_cast<const uint64_t*>(p)[w];
    }
    __syncthreads();
    return lds + (lane / L) * ROW;
  }
  __device__ uint8_t* result() {
    __syncthreads();
    return lds + ((int)(threadIdx.x & 63) / L) * ROW;
  }
  // wrote: this lane's job left a record in its row; rows are stored back with coalesced 8-byte words
  __device__ void commit(bool wrote) {
    const uint8_t* mine = wrote ? out : nullptr;
    const int lane = (int)(threadIdx.x & 63);
    __syncthreads();
    TC_NOUNROLL for (int f = lane; f < JOBS * WORDS; f += 64) {
      const int s = f / WORDS, w = f % WORDS;
      uint8_t* p = const_cast<uint8_t*>(lane_ptr(mine, s * L));
      if (p) reinterpret_cast<uint64_t*>(p)[w] = reinterpret_cast<const uint64_t*>(lds + s * ROW)[w];
    }
  }
};
#endif

}  // namespace tc
