// spmm.hip — T2: sum-reduce CSR SpMM,  Y[i,:] = sum_e val[e] * X[colind[e],:].
//
// Replaces torch_sparse.matmul(adj, x) at large/ours.py:34 (and its autograd backward, which is
// the same product with A^T).  HBM-bound gather: per stored entry the kernel reads 8 B of CSR
// (int32 colind + fp32 val) and one d-wide row of X.
//
// gfx950 mapping
//   * k_spmm_wave: one 64-lane wavefront per row; lane l owns features [4l, 4l+4) of each
//     256-feature panel, so one neighbour row is ONE fully coalesced 1 KiB (fp32) / 512 B (bf16)
//     global_load_dwordx4 / dwordx2 per wave.  The row index is wave-uniform, so rowptr / colind /
//     val are fetched with scalar loads (s_load_dword) and the X row base lives in SGPRs.  UNROLL
//     independent row loads are kept in flight per wave; with ~24 VGPRs the CU holds 32 waves, i.e.
//     up to 32 * UNROLL KiB of gathers in flight per CU, which is what hides HBM latency here.
//   * k_spmm_sub<LPR>: for d <= 128 a row needs fewer than 64 lanes, so 64/LPR rows share a wave.
//   * accumulation is fp32 in stored (ascending-source) order per feature — the same order as the
//     sequential CPU kernel of torch_sparse, so fp32 results agree to the last few ulps.
//   * block -> row-range mapping is XCD-aware: the dispatcher places block b on XCD b % 8, so the
//     remap gives each XCD a contiguous range of rows and its private 4 MiB L2 caches the X rows of
//     ONE graph neighbourhood instead of 1/8 of everybody's.
#include "common.h"

namespace sgf {
namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nblocks) {
  // bijective for any nblocks: the first `full` blocks are spread as 8 contiguous chunks.
  const int64_t per = nblocks / kNumXCD;
  const int64_t full = per * kNumXCD;
  if (b >= full) return b;
  return (b % kNumXCD) * per + b / kNumXCD;
}

__device__ __forceinline__ void fma4(float4& acc, float v, const float4& x) {
  acc.x = fmaf(v, x.x, acc.x);
  acc.y = fmaf(v, x.y, acc.y);
  acc.z = fmaf(v, x.z, acc.z);
  acc.w = fmaf(v, x.w, acc.w);
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_wave(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const float* __restrict__ val, const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
    int64_t ldy, int64_t n_rows, int32_t d) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row = blk * kWavesPerBlock + wid;  // wave-uniform
  if (row >= n_rows) return;
  const int64_t e0 = rowptr[row];
  const int64_t e1 = rowptr[row + 1];

  for (int f0 = 0; f0 < d; f0 += 256) {
    const int fc = f0 + lane * 4;
    const bool active = fc < d;
    const T* xl = x + (active ? fc : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t e = e0;
    for (; e + UNROLL <= e1; e += UNROLL) {
      int32_t c[UNROLL];
      float v[UNROLL];
      float4 xv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        c[u] = colind[e + u];
        v[u] = val[e + u];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        xv[u] = load4<T>(xl + static_cast<int64_t>(c[u]) * ldx);  // idle lanes re-read col 0
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) fma4(acc, v[u], xv[u]);
    }
    for (; e < e1; ++e) {
      const int32_t c = colind[e];
      const float v = val[e];
      const float4 xv = load4<T>(xl + static_cast<int64_t>(c) * ldx);
      fma4(acc, v, xv);
    }
    if (active) store4<T>(y + row * ldy + fc, acc);
  }
}

// LPR lanes per row (power of two, < 64); 64/LPR rows per wave.  d <= 4*LPR.
template <typename T, int LPR, int UNROLL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_sub(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const float* __restrict__ val, const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
    int64_t ldy, int64_t n_rows, int32_t d) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row = (blk * kWavesPerBlock + wid) * RPW + lane / LPR;
  const int fc = (lane % LPR) * 4;
  if (row >= n_rows || fc >= d) return;
  const int64_t e0 = rowptr[row];
  const int64_t e1 = rowptr[row + 1];
  const T* xl = x + fc;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t e = e0;
  for (; e + UNROLL <= e1; e += UNROLL) {
    int32_t c[UNROLL];
    float v[UNROLL];
    float4 xv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      c[u] = colind[e + u];  // same address across the LPR lanes of a row: one broadcast fetch
      v[u] = val[e + u];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) xv[u] = load4<T>(xl + static_cast<int64_t>(c[u]) * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) fma4(acc, v[u], xv[u]);
  }
  for (; e < e1; ++e) {
    const float4 xv = load4<T>(xl + static_cast<int64_t>(colind[e]) * ldx);
    fma4(acc, val[e], xv);
  }
  store4<T>(y + row * ldy + fc, acc);
}

template <typename T>
int launch(const int64_t* rowptr, const int32_t* colind, const float* val, const T* x, int64_t ldx,
           T* y, int64_t ldy, int64_t n_rows, int32_t d, hipStream_t st) {
  constexpr int UNROLL = 8;
  const dim3 block(kWavesPerBlock * 64);
  if (d > 128) {
    const int64_t nb = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL((k_spmm_wave<T, UNROLL>), dim3(static_cast<unsigned>(nb)), block, 0, st,
                       rowptr, colind, val, x, ldx, y, ldy, n_rows, d);
  } else {
#define SGF_SUB(LPR_)                                                                         \
  {                                                                                           \
    constexpr int RPB = kWavesPerBlock * (64 / LPR_);                                         \
    const int64_t nb = (n_rows + RPB - 1) / RPB;                                              \
    hipLaunchKernelGGL((k_spmm_sub<T, LPR_, UNROLL>), dim3(static_cast<unsigned>(nb)), block, \
                       0, st, rowptr, colind, val, x, ldx, y, ldy, n_rows, d);                \
  }
    if (d > 64) SGF_SUB(32)
    else if (d > 32) SGF_SUB(16)
    else if (d > 16) SGF_SUB(8)
    else if (d > 8) SGF_SUB(4)
    else if (d > 4) SGF_SUB(2)
    else SGF_SUB(1)
#undef SGF_SUB
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" int sgf_spmm(const int64_t* rowptr, const int32_t* colind, const float* val,
                        const void* x, int64_t ldx, void* y, int64_t ldy, int64_t n_rows,
                        int32_t d, int32_t dtype, void* stream) {
  SGF_REQUIRE(n_rows >= 0 && d >= 0, SGF_E_INVALID, "sgf_spmm: negative size");
  if (n_rows == 0 || d == 0) return SGF_OK;
  SGF_REQUIRE(rowptr && x && y, SGF_E_INVALID, "sgf_spmm: null pointer");
  SGF_REQUIRE(n_rows < (static_cast<int64_t>(1) << 31) * kWavesPerBlock, SGF_E_UNSUPPORTED,
              "sgf_spmm: n_rows too large for one launch");
  SGF_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, SGF_E_INVALID,
              "sgf_spmm: d, ldx, ldy must be multiples of 4 with ld >= d (d=%d ldx=%lld ldy=%lld)",
              d, static_cast<long long>(ldx), static_cast<long long>(ldy));
  const size_t esz = dtype == SGF_BF16 ? 2 : 4;
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(x) % (4 * esz) == 0 &&
                  reinterpret_cast<uintptr_t>(y) % (4 * esz) == 0,
              SGF_E_INVALID, "sgf_spmm: x / y must be aligned to 4 elements");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return launch<float>(rowptr, colind, val, static_cast<const float*>(x), ldx,
                         static_cast<float*>(y), ldy, n_rows, d, st);
  if (dtype == SGF_BF16)
    return launch<uint16_t>(rowptr, colind, val, static_cast<const uint16_t*>(x), ldx,
                            static_cast<uint16_t*>(y), ldy, n_rows, d, st);
  set_error("sgf_spmm: unknown dtype %d", dtype);
  return SGF_E_INVALID;
}
