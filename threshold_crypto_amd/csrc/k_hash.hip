// gfx950 kernels: SHA3 -> ChaCha20 -> G2 hashing and the keystream XOR.
#include "tc_jobs.h"
#include "tc_launch.h"

namespace tc {

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_hash_g2(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off,
                                                    size_t B, uint8_t* __restrict__ out, int fix) {
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j >= B) return;
  job_hash_g2(msgs + off[j], (size_t)(off[j + 1] - off[j]), out + j * 192, fix != 0);
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_hash_g1_g2(const uint8_t* __restrict__ g1, const uint8_t* __restrict__ msgs,
                                                       const uint64_t* __restrict__ off, size_t B,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ status, int fix) {
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j >= B) return;
  uint8_t st = job_hash_g1_g2(g1 + j * 96, msgs + off[j], (size_t)(off[j + 1] - off[j]), out + j * 192, fix != 0);
  if (status && pair_leader()) status[j] = st;
}

// batches from 131 072 messages on (tc_launch.h kDuoMinHash): a lane pair takes TWO messages (tc_duo.h) -- pair p hashes messages 2p and 2p + 1
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_hash_g2_x2(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off,
                                                                 size_t B, uint8_t* __restrict__ out, int fix) {
  const size_t p = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t ja = 2 * p;
  if (ja >= B) return;
  const bool has_b = ja + 1 < B;
  const size_t jb = has_b ? ja + 1 : ja;
  job_hash_g2_x2(msgs + off[ja], (size_t)(off[ja + 1] - off[ja]), msgs + off[jb], (size_t)(off[jb + 1] - off[jb]), out + ja * 192,
                 has_b ? out + jb * 192 : nullptr, fix != 0);
}
__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_hash_g1_g2_x2(const uint8_t* __restrict__ g1, const uint8_t* __restrict__ msgs,
                                                                    const uint64_t* __restrict__ off, size_t B,
                                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ status, int fix) {
  const size_t p = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  const size_t ja = 2 * p;
  if (ja >= B) return;
  const bool has_b = ja + 1 < B;
  const size_t jb = has_b ? ja + 1 : ja;
  uint8_t sa, sb;
  job_hash_g1_g2_x2(g1 + ja * 96, msgs + off[ja], (size_t)(off[ja + 1] - off[ja]), g1 + jb * 96, msgs + off[jb],
                    (size_t)(off[jb + 1] - off[jb]), out + ja * 192, has_b ? out + jb * 192 : nullptr, fix != 0, sa, sb);
  if (status && pair_leader()) {
    status[ja] = sa;
    if (has_b) status[jb] = sb;
  }
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G1) void k_xor_with_hash(const uint8_t* __restrict__ g1,
                                                          const uint8_t* __restrict__ data,
                                                          const uint64_t* __restrict__ off, size_t B,
                                                          uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= B) return;
  if (status && status[j] != TC_JOB_OK) return;  // an earlier stage (combine) flagged the job
  uint8_t st = job_xor_with_hash(g1 + j * 96, data + off[j], (size_t)(off[j + 1] - off[j]), out + off[j]);
  if (status) status[j] = st;
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G2) void k_encrypt(const uint8_t* __restrict__ pk, size_t pk_stride,
                                                    const uint8_t* __restrict__ r, const uint8_t* __restrict__ msgs,
                                                    const uint64_t* __restrict__ off, size_t B,
                                                    uint8_t* __restrict__ out_u, uint8_t* __restrict__ out_v,
                                                    uint8_t* __restrict__ out_w, uint8_t* __restrict__ status, TableArena ta) {
  const uint32_t tslot = table_slot_acquire(ta);  // W = r H(U, V): a GLS ladder with its table in the arena
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) / kG2Lanes;
  if (j < B) {
    uint8_t st = job_encrypt(pk + j * pk_stride, r + j * 32, msgs + off[j], (size_t)(off[j + 1] - off[j]), out_u + j * 96,
                             out_v + off[j], out_w + j * 192);
    if (status && pair_leader()) status[j] = st;
  }
  table_slot_release(ta, tslot);
}

__global__ __launch_bounds__(kBlock, TC_WAVES_G1_AUX) void k_commitment_evaluate(const uint8_t* __restrict__ commit, size_t t,
                                                                const uint64_t* __restrict__ idx, size_t M,
                                                                uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
  const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= M) return;
  uint8_t st = job_commitment_evaluate(commit, (int)t, idx[j], out + j * 96);
  if (status) status[j] = st;
}

void launch_encrypt(hipStream_t st, TableArena ta, const uint8_t* pk, size_t pk_stride, const uint8_t* r, const uint8_t* msgs,
                    const uint64_t* off, size_t B, uint8_t* out_u, uint8_t* out_v, uint8_t* out_w, uint8_t* status) {
  if (B && ta.mem && ta.flags) hipLaunchKernelGGL(k_encrypt, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, pk, pk_stride, r, msgs, off, B, out_u, out_v, out_w, status, ta);
}
void launch_commitment_evaluate(hipStream_t st, const uint8_t* commit, size_t t, const uint64_t* idx, size_t M, uint8_t* out,
                                uint8_t* status) {
  if (M) hipLaunchKernelGGL(k_commitment_evaluate, dim3(grid_for(M)), dim3(kBlock), 0, st, commit, t, idx, M, out, status);
}
void launch_hash_g2(const Tuning& tn, hipStream_t st, const uint8_t* msgs, const uint64_t* off, size_t B, uint8_t* out, bool fix) {
  if (!B) return;
  if (duo_form(B, tn.duo_min_hash)) hipLaunchKernelGGL(k_hash_g2_x2, dim3(grid_for((B + 1) / 2 * kG2Lanes)), dim3(kBlock), 0, st, msgs, off, B, out, fix ? 1 : 0);
  else hipLaunchKernelGGL(k_hash_g2, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, msgs, off, B, out, fix ? 1 : 0);
}
void launch_hash_g1_g2(const Tuning& tn, hipStream_t st, const uint8_t* g1, const uint8_t* msgs, const uint64_t* off, size_t B,
                       uint8_t* out, uint8_t* status, bool fix) {
  if (!B) return;
  if (duo_form(B, tn.duo_min_hash)) hipLaunchKernelGGL(k_hash_g1_g2_x2, dim3(grid_for((B + 1) / 2 * kG2Lanes)), dim3(kBlock), 0, st, g1, msgs, off, B, out, status, fix ? 1 : 0);
  else hipLaunchKernelGGL(k_hash_g1_g2, dim3(grid_for(B * kG2Lanes)), dim3(kBlock), 0, st, g1, msgs, off, B, out, status, fix ? 1 : 0);
}
void launch_xor_with_hash(hipStream_t st, const uint8_t* g1, const uint8_t* data, const uint64_t* off, size_t B,
                          uint8_t* out, uint8_t* status) {
  if (B) hipLaunchKernelGGL(k_xor_with_hash, dim3(grid_for(B)), dim3(kBlock), 0, st, g1, data, off, B, out, status);
}

}  // namespace tc
