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
//   * measured alternatives (MI355X, products-size uniform graph, bf16 d = 256): UNROLL 4 / 8 / 16 run
//     9.99 / 10.07 / 10.29 ms, non-temporal gathers 11.85 ms — the kernel sits at the memory system's
//     gather limit (6.5 TB/s of 512-byte rows), not at an issue or latency limit.
//   * accumulation is fp32 in stored (ascending-source) order per feature — the same order as the
//     sequential CPU kernel of torch_sparse, so fp32 results agree to the last few ulps.
//   * block -> row-range mapping is XCD-aware (xcd_remap): each XCD walks 4096-row chunks, so its
//     private 4 MiB L2 caches the X rows of ONE graph neighbourhood instead of 1/8 of everybody's;
//     chunks are dealt round-robin so that skewed graphs stay load-balanced across XCDs.
#include "common.h"
#include "spmm_shared.h"

#include <climits>
#include <string>
#include <cstdlib>

namespace sgf {
namespace {

constexpr int kWavesPerBlock = 4;

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
    int64_t ldy, int64_t n_rows, int32_t d, LongQueue lq) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row = blk * kWavesPerBlock + wid;  // wave-uniform
  if (row >= n_rows) return;
  const int64_t e0 = rowptr[row];
  const int64_t e1 = rowptr[row + 1];
  if (e1 - e0 > lq.long_len) {   // wave-uniform
    push_long_row(lq, row, e1 - e0, lane, 64);
    return;
  }

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

// ---- wave per row, 32-bit buffer addressing ------------------------------------------------------------
// rocprofv3 on the community graph (profiles/r02_spmm_pmc.md) showed k_spmm_wave ISSUE-bound once its gathers
// hit in L2: 21 wave-instructions per stored entry — 11 of them scalar ALU for the 64-bit address c * ldx —
// at one instruction per SIMD per 4 cycles is the whole launch time.  This variant addresses X through a
// buffer descriptor: the row offset c * pitch is ONE s_mul_i32 into the instruction's scalar offset, the
// lane's column offset a loop-invariant VGPR, so an entry costs 1 SALU + 1 buffer_load + 4 unpack + 2 packed
// FMA (+ 1/4 of the two s_load_dwordx8 that fetch 8 codes and 8 values).  The next batch's codes / values
// are requested before the current batch's FMAs, so the scalar-memory latency overlaps the gathers instead
// of following them; the row tail goes through a 4 / 2 / 1 ladder (at most 3 dependent steps, not 7).
// Needs n_cols * ldx * sizeof(T) < 2^32 (offsets are 32-bit); larger operands keep k_spmm_wave.
template <typename T> struct BufRow;
template <> struct BufRow<uint16_t> {
  using Raw = u32x2;
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t r, int voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, static_cast<int>(soff), 0);
  }
  static __device__ __forceinline__ float4 widen(const Raw& v) {
    float4 f;
    f.x = __uint_as_float(v.x << 16);
    f.y = __uint_as_float(v.x & 0xffff0000u);
    f.z = __uint_as_float(v.y << 16);
    f.w = __uint_as_float(v.y & 0xffff0000u);
    return f;
  }
};
template <> struct BufRow<float> {
  using Raw = u32x4;
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t r, int voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, static_cast<int>(soff), 0);
  }
  static __device__ __forceinline__ float4 widen(const Raw& v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
};

template <typename T, int N>
__device__ __forceinline__ void gather_batch(__amdgpu_buffer_rsrc_t rsrc, int voff, uint32_t pitch,
                                             const int32_t* __restrict__ colind, const float* __restrict__ val,
                                             int e, float4& acc) {
  int32_t c[N];
  float v[N];
  typename BufRow<T>::Raw raw[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    c[u] = colind[e + u];
    v[u] = val[e + u];
  }
#pragma unroll
  for (int u = 0; u < N; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c[u]) * pitch);
#pragma unroll
  for (int u = 0; u < N; ++u) fma4(acc, v[u], BufRow<T>::widen(raw[u]));
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_row(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
    const T* __restrict__ x, uint32_t pitch, uint32_t x_bytes, T* __restrict__ y, int64_t ldy, int64_t n_rows,
    int32_t d, LongQueue lq) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row = blk * kWavesPerBlock + wid;  // wave-uniform
  if (row >= n_rows) return;
  const int64_t e0 = rowptr[row];
  const int64_t e1 = rowptr[row + 1];
  if (e1 - e0 > lq.long_len) {   // wave-uniform
    push_long_row(lq, row, e1 - e0, lane, 64);
    return;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, x_bytes, 0x00020000);
  const int fc = lane * 4;
  const bool active = fc < d;
  const int voff = active ? fc * static_cast<int>(sizeof(T)) : 0;   // idle lanes re-read column 0
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // 32-bit positions relative to the row start (scalar compares; a 64-bit induction variable costs VALU compares)
  const int32_t* __restrict__ ci = colind + e0;
  const float* __restrict__ va = val + e0;
  const int len = static_cast<int>(e1 - e0);
  int i = 0;
  if (UNROLL <= len) {
    // full batches, two per trip: A uses (c0, v0) and requests (c1, v1), B the other way round — the scalar
    // loads of the NEXT batch are issued before the FMAs of the current one wait for its gathers
    int32_t c0[UNROLL], c1[UNROLL];
    float v0[UNROLL], v1[UNROLL];
    typename BufRow<T>::Raw raw[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      c0[u] = ci[u];
      v0[u] = va[u];
    }
    for (;;) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c0[u]) * pitch);
      i += UNROLL;
      const bool more_a = i + UNROLL <= len;
      if (more_a) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          c1[u] = ci[i + u];
          v1[u] = va[i + u];
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) fma4(acc, v0[u], BufRow<T>::widen(raw[u]));
      if (!more_a) break;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c1[u]) * pitch);
      i += UNROLL;
      const bool more_b = i + UNROLL <= len;
      if (more_b) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          c0[u] = ci[i + u];
          v0[u] = va[i + u];
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) fma4(acc, v1[u], BufRow<T>::widen(raw[u]));
      if (!more_b) break;
    }
  }
  if (i + 4 <= len) { gather_batch<T, 4>(rsrc, voff, pitch, ci, va, i, acc); i += 4; }
  if (i + 2 <= len) { gather_batch<T, 2>(rsrc, voff, pitch, ci, va, i, acc); i += 2; }
  if (i < len) gather_batch<T, 1>(rsrc, voff, pitch, ci, va, i, acc);
  if (active) store4<T>(y + row * ldy + fc, acc);
}

// ---- flattened (segmented) wave kernel --------------------------------------------------------------------
// Why: with ~46 stored entries per row, a wave-per-row kernel spends its life in a chain of DEPENDENT memory
// latencies — kernel arguments, rowptr, then per 8-entry batch (codes/values -> gathers), then a tail ladder —
// about ten of them, ~12 us per row (measured: 2.45 M waves in 3.6 ms at 32 waves per CU).  Once the gathers
// hit in L2 that chain, not bandwidth and not instruction issue, is the launch time.  Here a wave owns
// kSegRows CONSECUTIVE rows and walks their stored entries as ONE stream in groups of G: the gathers of a
// group are all in flight together, the codes / values of the NEXT group are requested before the current
// group's FMAs, and a row boundary inside a group costs one scalar compare per entry plus a store when it
// is hit.  Per row there is no dependent latency left at all; per group there is one.
// Rows longer than lq.long_len go to the long-row queue as in the other kernels: a wave that owns one (or
// whose entry count does not fit the 32-bit stream positions) takes the per-row fallback below.
constexpr int kSegRows = 4;   // 2: 3.42 ms, 4: 3.23, 8: 3.28-3.31, 16: 4.4 (community graph after sgf_reorder; more rows per wave = a larger
                              // concurrent footprint per XCD than its 4 MiB L2 holds)

template <typename T, int G>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_seg(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
    const T* __restrict__ x, uint32_t pitch, uint32_t x_bytes, T* __restrict__ y, int64_t ldy, int64_t n_rows,
    int32_t d, int32_t chunk_blocks, LongQueue lq) {
  using Raw = typename BufRow<T>::Raw;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x, chunk_blocks);
  const int64_t r_first = (blk * kWavesPerBlock + wid) * kSegRows;
  if (r_first >= n_rows) return;
  const int nr = n_rows - r_first < kSegRows ? static_cast<int>(n_rows - r_first) : kSegRows;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, x_bytes, 0x00020000);
  const int fc = lane * 4;
  const bool active = fc < d;
  const int voff = active ? fc * static_cast<int>(sizeof(T)) : 0;
  T* __restrict__ yl = y + r_first * ldy + (active ? fc : 0);

  // lane i (< nr) holds the END of row r_first + i relative to the wave's first entry
  const int64_t e_first = rowptr[r_first];
  const int64_t end_abs = rowptr[r_first + 1 + (lane < nr ? lane : nr - 1)];
  const int64_t begin_abs = __shfl_up(end_abs, 1, 64);
  const int64_t len_i = end_abs - (lane == 0 ? e_first : begin_abs);
  const bool long_i = lane < nr && len_i > lq.long_len;
  // (readlane results are wave-uniform AND known to be so to the compiler: scalar loads / branches downstream)
  auto lane64 = [&](int64_t v, int i) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffff), i);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), i);
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
  };
  const int64_t total64 = lane64(end_abs, nr - 1) - e_first;
  const int32_t* __restrict__ ci = colind + e_first;
  const float* __restrict__ va = val + e_first;

  if (__ballot(long_i) != 0 || total64 >= (static_cast<int64_t>(1) << 31)) {
    // fallback: row by row (rare: a hub row among this wave's rows)
    for (int r = 0; r < nr; ++r) {
      const int64_t re = lane64(end_abs, r);
      const int64_t rb = r == 0 ? e_first : lane64(end_abs, r - 1);
      if (re - rb > lq.long_len) {
        push_long_row(lq, r_first + r, re - rb, lane, 64);
        continue;
      }
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t e = rb; e < re; ++e)
        fma4(acc, val[e], BufRow<T>::widen(BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(colind[e]) * pitch)));
      if (active) store4<T>(yl + r * ldy, acc);
    }
    return;
  }

  const int rel_v = static_cast<int>(end_abs - e_first);       // lane i: end of local row i in stream positions
  const int total = static_cast<int>(total64);
  int row = 0;
  int row_end = __builtin_amdgcn_readlane(rel_v, 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // a row ends at stream position p: write it, start the next (empty rows fall through the loop and store zeros)
  auto boundary = [&](int p) {
    while (p == row_end && row < nr) {
      if (active) store4<T>(yl + row * ldy, acc);
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
      ++row;
      row_end = row < nr ? __builtin_amdgcn_readlane(rel_v, row < kSegRows ? row : kSegRows - 1) : 0x7fffffff;
    }
  };

  int pos = 0;
  int32_t c0[G], c1[G];
  float v0[G], v1[G];
  Raw raw[G];
  if (G <= total) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      c0[u] = ci[u];
      v0[u] = va[u];
    }
    for (;;) {
      // ---- body A: (c0, v0) current, (c1, v1) requested
#pragma unroll
      for (int u = 0; u < G; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c0[u]) * pitch);
      const bool more_a = pos + 2 * G <= total;
      if (more_a) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          c1[u] = ci[pos + G + u];
          v1[u] = va[pos + G + u];
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (pos + u == row_end) boundary(pos + u);
        fma4(acc, v0[u], BufRow<T>::widen(raw[u]));
      }
      pos += G;
      if (!more_a) break;
      // ---- body B: roles swapped
#pragma unroll
      for (int u = 0; u < G; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c1[u]) * pitch);
      const bool more_b = pos + 2 * G <= total;
      if (more_b) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          c0[u] = ci[pos + G + u];
          v0[u] = va[pos + G + u];
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (pos + u == row_end) boundary(pos + u);
        fma4(acc, v1[u], BufRow<T>::widen(raw[u]));
      }
      pos += G;
      if (!more_b) break;
    }
  }
  // the stream's tail (< G entries): gathers clamped to the last entry, each FMA behind a uniform test
  if (pos < total) {
    const int g = total - pos;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int p = pos + (u < g ? u : g - 1);
      c0[u] = ci[p];
      v0[u] = va[p];
    }
#pragma unroll
    for (int u = 0; u < G; ++u) raw[u] = BufRow<T>::load(rsrc, voff, static_cast<uint32_t>(c0[u]) * pitch);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (u < g) {
        if (pos + u == row_end) boundary(pos + u);
        fma4(acc, v0[u], BufRow<T>::widen(raw[u]));
      }
    }
    pos = total;
  }
  boundary(total);   // closes the last non-empty row and any trailing empty rows (row_end == total for all of them)
}

// ---- the same stream, bf16 rows fetched TWO PER INSTRUCTION ---------------------------------------------------
// Four differently built kernels (wave per row with 64-bit and with buffer addressing, the flattened stream above,
// the LDS-staged row blocks) all ran the re-ordered community graph in 3.6-4.1 ms = 15 TB/s of L2-served
// gathers: the limit is the vector memory pipe's address rate, which is per LANE — an 8-byte-per-lane load
// (one 512-byte bf16 row per wave instruction) moves half the bytes of a 16-byte-per-lane load for the same
// lane work (MI355X_MICROARCH.md: 8-B accesses run at 0.54-0.70x the 16-B rate).  So: lanes 0-31 fetch the
// row of stream entry 2j (16 B = 8 bf16 each), lanes 32-63 the row of entry 2j+1, in ONE dwordx4 buffer load;
// each half accumulates its own partial sums of the current row (even / odd stream positions) and the two
// halves are added when the row ends.  A row boundary between the two entries of a pair takes a slow path
// (the two halves are applied one after the other under exec masks).  Deterministic: a row's summation order
// depends only on where its entries sit in the wave's stream.
template <int G>   // stored entries per group (even): G/2 pair loads in flight per wave
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_seg_bf16x2(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
    const uint16_t* __restrict__ x, uint32_t pitch, uint32_t x_bytes, uint16_t* __restrict__ y, int64_t ldy,
    int64_t n_rows, int32_t d, int32_t chunk_blocks, LongQueue lq) {
  constexpr int P = G / 2;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x, chunk_blocks);
  const int64_t r_first = (blk * kWavesPerBlock + wid) * kSegRows;
  if (r_first >= n_rows) return;
  const int nr = n_rows - r_first < kSegRows ? static_cast<int>(n_rows - r_first) : kSegRows;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, x_bytes, 0x00020000);
  const int q = lane & 31;
  const bool hi = lane >= 32;
  const uint32_t hmask = hi ? 0xffffffffu : 0u;
  const int fc = q * 8;
  const bool active = fc < d;                      // d % 8 == 0 (checked by the launcher)
  const uint32_t lanebase = active ? static_cast<uint32_t>(fc) * 2u : 0u;
  uint16_t* __restrict__ yl = y + r_first * ldy + (active ? fc : 0);

  auto lane64 = [&](int64_t v, int i) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffff), i);
    const uint32_t hi32 = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), i);
    return static_cast<int64_t>((static_cast<uint64_t>(hi32) << 32) | lo);
  };
  const int64_t e_first = rowptr[r_first];
  const int64_t end_abs = rowptr[r_first + 1 + (lane < nr ? lane : nr - 1)];
  const int64_t begin_abs = __shfl_up(end_abs, 1, 64);
  const int64_t len_i = end_abs - (lane == 0 ? e_first : begin_abs);
  const bool long_i = lane < nr && len_i > lq.long_len;
  const int64_t total64 = lane64(end_abs, nr - 1) - e_first;
  const int32_t* __restrict__ ci = colind + e_first;
  const float* __restrict__ va = val + e_first;

  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  auto fma8 = [&](float v, const u32x4& r) {
    acc[0] = fmaf(v, __uint_as_float(r.x << 16), acc[0]);
    acc[1] = fmaf(v, __uint_as_float(r.x & 0xffff0000u), acc[1]);
    acc[2] = fmaf(v, __uint_as_float(r.y << 16), acc[2]);
    acc[3] = fmaf(v, __uint_as_float(r.y & 0xffff0000u), acc[3]);
    acc[4] = fmaf(v, __uint_as_float(r.z << 16), acc[4]);
    acc[5] = fmaf(v, __uint_as_float(r.z & 0xffff0000u), acc[5]);
    acc[6] = fmaf(v, __uint_as_float(r.w << 16), acc[6]);
    acc[7] = fmaf(v, __uint_as_float(r.w & 0xffff0000u), acc[7]);
  };
  // row finished: add the two halves, lanes 0-31 write 8 bf16 each, both halves start the next row at zero
  auto flush_row = [&](int local_row) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = acc[k] + __shfl_xor(acc[k], 32, 64);
    if (active && !hi) {
      uint4 o;
      o.x = static_cast<uint32_t>(f32_to_bf16(t[0])) | (static_cast<uint32_t>(f32_to_bf16(t[1])) << 16);
      o.y = static_cast<uint32_t>(f32_to_bf16(t[2])) | (static_cast<uint32_t>(f32_to_bf16(t[3])) << 16);
      o.z = static_cast<uint32_t>(f32_to_bf16(t[4])) | (static_cast<uint32_t>(f32_to_bf16(t[5])) << 16);
      o.w = static_cast<uint32_t>(f32_to_bf16(t[6])) | (static_cast<uint32_t>(f32_to_bf16(t[7])) << 16);
      *reinterpret_cast<uint4*>(yl + local_row * ldy) = o;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  };
  auto pair_load = [&](int32_t ca, int32_t cb) -> u32x4 {
    // offsets are added as unsigned numbers (no wrap): select, do not subtract (cb may be smaller than ca
    // when the pair straddles a row boundary)
    const uint32_t sa = static_cast<uint32_t>(ca) * pitch, sb = static_cast<uint32_t>(cb) * pitch;
    const uint32_t voff = lanebase + (sa ^ ((sa ^ sb) & hmask));
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(voff), 0, 0);
  };
  auto pick = [&](float a, float b) -> float {   // lanes 0-31: a, lanes 32-63: b
    const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    return __uint_as_float(ua ^ ((ua ^ ub) & hmask));
  };

  if (__ballot(long_i) != 0 || total64 >= (static_cast<int64_t>(1) << 31)) {
    // fallback: row by row, one entry per instruction (rare: a hub row among this wave's rows)
    for (int r = 0; r < nr; ++r) {
      const int64_t re = lane64(end_abs, r);
      const int64_t rb = r == 0 ? e_first : lane64(end_abs, r - 1);
      if (re - rb > lq.long_len) {
        push_long_row(lq, r_first + r, re - rb, lane, 64);
        continue;
      }
      for (int64_t e = rb; e < re; ++e) {
        const u32x4 raw = pair_load(colind[e], colind[e]);
        if (!hi) fma8(val[e], raw);
      }
      flush_row(r);
    }
    return;
  }

  const int rel_v = static_cast<int>(end_abs - e_first);       // lane i: end of local row i in stream positions
  const int total = static_cast<int>(total64);
  int row = 0;
  int row_end = __builtin_amdgcn_readlane(rel_v, 0);
  auto boundary = [&](int p) {     // rows ending at stream position p (empty rows store zeros)
    while (p == row_end && row < nr) {
      flush_row(row);
      ++row;
      row_end = row < nr ? __builtin_amdgcn_readlane(rel_v, row < kSegRows ? row : kSegRows - 1) : 0x7fffffff;
    }
  };
  // one pair at stream positions (p, p + 1): fast unless a row ends exactly between them
  auto apply_pair = [&](int p, float v_a, float v_b, const u32x4& raw) {
    if (p == row_end) boundary(p);
    if (p + 1 == row_end) {
      if (!hi) fma8(v_a, raw);
      boundary(p + 1);
      if (hi) fma8(v_b, raw);
    } else {
      fma8(pick(v_a, v_b), raw);
    }
  };

  int pos = 0;
  int32_t c0[G], c1[G];
  float v0[G], v1[G];
  u32x4 raw[P];
  if (G <= total) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      c0[u] = ci[u];
      v0[u] = va[u];
    }
    for (;;) {
#pragma unroll
      for (int j = 0; j < P; ++j) raw[j] = pair_load(c0[2 * j], c0[2 * j + 1]);
      const bool more_a = pos + 2 * G <= total;
      if (more_a) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          c1[u] = ci[pos + G + u];
          v1[u] = va[pos + G + u];
        }
      }
#pragma unroll
      for (int j = 0; j < P; ++j) apply_pair(pos + 2 * j, v0[2 * j], v0[2 * j + 1], raw[j]);
      pos += G;
      if (!more_a) break;
#pragma unroll
      for (int j = 0; j < P; ++j) raw[j] = pair_load(c1[2 * j], c1[2 * j + 1]);
      const bool more_b = pos + 2 * G <= total;
      if (more_b) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          c0[u] = ci[pos + G + u];
          v0[u] = va[pos + G + u];
        }
      }
#pragma unroll
      for (int j = 0; j < P; ++j) apply_pair(pos + 2 * j, v1[2 * j], v1[2 * j + 1], raw[j]);
      pos += G;
      if (!more_b) break;
    }
  }
  // the stream's tail (< G entries): indices clamped to the last entry, values of the padding are zero
  if (pos < total) {
    const int g = total - pos;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int p = pos + (u < g ? u : g - 1);
      c0[u] = ci[p];
      v0[u] = u < g ? va[p] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) raw[j] = pair_load(c0[2 * j], c0[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      if (2 * j < g) {
        const int p = pos + 2 * j;
        if (p == row_end) boundary(p);
        if (2 * j + 1 < g) {
          apply_pair(p, v0[2 * j], v0[2 * j + 1], raw[j]);
        } else if (!hi) {
          fma8(v0[2 * j], raw[j]);                       // last entry of the stream: lanes 0-31 only
        }
      }
    }
    pos = total;
  }
  boundary(total);   // closes the last non-empty row and any trailing empty rows
}

// Tried and dropped (r02): a second version of this kernel that parks the stream's {code, value} pairs in a per-wave
// LDS window (one coalesced load per 64 entries, one ds_read_b64 + one v_mad_u32_u24 per pair, no row-end test inside
// 16-entry groups).  Bit-identical results, a much shorter front end — and 3.38 ms against 3.28 ms for the kernel above
// on the same box (community graph, sgf_reorder order).  The kernel is not bound by its instruction count but by the
// gathers themselves: scripts/l2_gather_probe.hip puts the ceiling for 512-byte rows that HIT in L2 at 25-27 TB/s
// (16 B per lane; 14.7 TB/s at 8 B per lane), this launch moves 58 GB of rows with a 70 % L2 hit rate, the rest at the
// 7-8 TB/s fabric rate.

// LPR lanes per row (power of two, < 64); 64/LPR rows per wave.  d <= 4*LPR.
template <typename T, int LPR, int UNROLL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_sub(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const float* __restrict__ val, const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
    int64_t ldy, int64_t n_rows, int32_t d, LongQueue lq) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row = (blk * kWavesPerBlock + wid) * RPW + lane / LPR;
  const int fc = (lane % LPR) * 4;
  if (row >= n_rows) return;
  const int64_t e0 = rowptr[row];
  const int64_t e1 = rowptr[row + 1];
  if (e1 - e0 > lq.long_len) {   // uniform over the LPR lanes of this row
    push_long_row(lq, row, e1 - e0, lane % LPR, LPR);
    return;
  }
  if (fc >= d) return;
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

// one workgroup per queued (row, segment): wave w takes the 64-entry pieces w, w + NW, ... of the segment;
// lane l owns features [4l, 4l+4) of each 256-feature panel; partial[slot][d] fp32.
// r05: a piece's source ids and values arrive as ONE coalesced vector load each (lane j <- entry j) and are handed to the
// gathers through v_readlane — the row index of a gather is then wave-uniform (scalar address arithmetic, one vector load
// per entry, UNROLL of them in flight) where the first form issued two same-address vector loads and a 64-bit multiply per
// entry and lane: on the R-MAT graph (hub rows of 10^4-10^5 entries: 45 % of the stored entries take this path) the
// kernel moved its bytes at 3.9 TB/s against 7.2 for the row kernel next to it (profiles/r05_spmm_pmc.md).
template <typename T, int UNROLL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_spmm_long_seg(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const float* __restrict__ val, const T* __restrict__ x, int64_t ldx, int32_t d, LongQueue lq,
    float* __restrict__ partial) {
  __shared__ float4 red[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int cnt = *lq.count;
  if (cnt > lq.cap) cnt = lq.cap;
  for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
    const LongEntry en = lq.entries[i];
    const int64_t r0 = rowptr[en.row];
    const int64_t e0 = r0 + static_cast<int64_t>(en.seg) * kSegLen;
    int64_t e1 = e0 + kSegLen;
    const int64_t rend = rowptr[en.row + 1];
    if (e1 > rend) e1 = rend;
    for (int f0 = 0; f0 < d; f0 += 256) {
      const int fc = f0 + lane * 4;
      const bool active = fc < d;
      const T* xl = x + (active ? fc : 0);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t e = e0 + wave * 64; e < e1; e += kWavesPerBlock * 64) {
        const int64_t mine = e + lane;
        const int32_t c_l = mine < e1 ? colind[mine] : 0;              // lane j <- entry j of this piece
        const float v_l = mine < e1 ? val[mine] : 0.f;
        const int n = e1 - e < 64 ? static_cast<int>(e1 - e) : 64;     // wave-uniform
        for (int j0 = 0; j0 < n; j0 += UNROLL) {
          float4 xv[UNROLL];
          float vv[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const int j = j0 + u < n ? j0 + u : n - 1;                  // (a clamped repeat carries weight 0)
            const int32_t cj = __builtin_amdgcn_readlane(c_l, j);
            vv[u] = j0 + u < n ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_l), j)) : 0.f;
            xv[u] = load4<T>(xl + static_cast<int64_t>(cj) * ldx);
          }
#pragma unroll
          for (int u = 0; u < UNROLL; ++u)
            if (j0 + u < n) fma4(acc, vv[u], xv[u]);
        }
      }
      red[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && active) {
        float4 t = red[0][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
          t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w;
        }
        *reinterpret_cast<float4*>(&partial[static_cast<int64_t>(i) * d + fc]) = t;
      }
      __syncthreads();
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_spmm_long_fin(LongQueue lq, const float* __restrict__ partial,
                                                        int32_t d, T* __restrict__ y, int64_t ldy) {
  int cnt = *lq.count;
  if (cnt > lq.cap) cnt = lq.cap;
  const int f4 = d / 4;
  for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
    const LongEntry en = lq.entries[i];
    if (en.seg != 0) continue;   // the head entry of a row owns slots [i, i + k)
    for (int c = threadIdx.x; c < f4; c += blockDim.x) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < en.k; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(&partial[static_cast<int64_t>(i + s) * d + 4 * c]);
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      store4<T>(y + static_cast<int64_t>(en.row) * ldy + 4 * c, t);
    }
  }
}


// ---- row-block kernel with LDS staging of shared neighbour rows (plan: spmm_plan.hip) ------------------
// One workgroup = R consecutive target rows, 8 CONSECUTIVE rows per wave.
//   Phase 1  the rows of X that at least two of the block's stored entries reference (sh_cols, at most
//            `lds_rows`, most-referenced first) are fetched ONCE — one coalesced 512 B / 1 KiB wave load each,
//            all of a wave's loads in flight together — and parked in LDS in storage format.
//   Phase 2  wave per row, but built so that no step of the per-row dependency chain is a scalar-memory
//            latency (the wave-per-row kernels wait for an s_load of codes / values before every batch of
//            gathers; with 8-entry batches that is ~12 dependent memory latencies per 46-entry row, and it —
//            not bandwidth — bounds them once the gathers hit in L2):
//              * a row's codes and values are fetched as ONE coalesced vector load (lane j <- entry j of the
//                current <= 64-entry piece), issued one piece AHEAD and parked in a 512-byte per-wave LDS
//                scratch; an entry is then read back with a broadcast ds_read_b64 (~100 cycles, LDS port);
//              * the piece's first gathers (entries the plan left on the L2 / HBM path) are issued BEFORE its
//                LDS entries are processed, so their latency is covered by LDS work;
//              * an LDS entry costs two ds_reads (code/value broadcast, then the staged row: lane l owns bytes
//                [l*8, l*8+8) of the slot, conflict-free) and no scalar or vector memory instruction.
// Per row the LDS entries are accumulated in stored order into one accumulator and the gathered ones into
// another, added at the end: a fixed order, so results are deterministic (and equal to k_spmm_wave up to fp32
// rounding of the different summation order).
template <typename T> struct Stored;
template <> struct Stored<float> { using type = float4; };
template <> struct Stored<uint16_t> { using type = uint2; };

template <typename T>
__device__ __forceinline__ typename Stored<T>::type load_raw(const T* p) {
  return *reinterpret_cast<const typename Stored<T>::type*>(p);
}
__device__ __forceinline__ float4 widen(const float4& r) { return r; }
__device__ __forceinline__ float4 widen(const uint2& r) {
  float4 f;
  f.x = __uint_as_float(r.x << 16);
  f.y = __uint_as_float(r.x & 0xffff0000u);
  f.z = __uint_as_float(r.y << 16);
  f.w = __uint_as_float(r.y & 0xffff0000u);
  return f;
}

constexpr int kBlkRowsPerWave = 8;
constexpr int kPiece = 64;                       // entries per piece = lanes per wave
constexpr int kScratchPerWave = 2 * kPiece * 8;  // bytes: two pieces of {code, value}

template <typename T, int DG, int LB, int MINW>
__global__ __launch_bounds__(1024, MINW) void k_spmm_blk(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ ecode, const float* __restrict__ eval,
    const int32_t* __restrict__ nlds, const int32_t* __restrict__ sh_ptr, const int32_t* __restrict__ sh_cols,
    const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int64_t n_rows, int32_t d,
    int32_t rows_per_block, int32_t lds_rows, int32_t chunk_blocks, LongQueue lq, int32_t dbg) {
  using V = typename Stored<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  V* tile = reinterpret_cast<V*>(smem_raw);           // [slot][64 lanes]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(blockDim.x >> 6);
  uint2* scratch = reinterpret_cast<uint2*>(smem_raw + static_cast<size_t>(lds_rows) * 64 * sizeof(V)) +
                   wid * (2 * kPiece);                 // [2][64] {code, value bits}, private to this wave
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x, chunk_blocks);
  const int fc = lane * 4;
  const bool active = fc < d;
  const T* xl = x + (active ? fc : 0);

  // ---- phase 1: stage the shared rows.  Wave w takes slots w, w + nw, ...; kStage loads are issued back to
  // back before the first LDS write, so a full block costs one or two memory latencies, not one per row.
  const int s0 = sh_ptr[blk];
  const int ns = sh_ptr[blk + 1] - s0;
  constexpr int kStage = 9;
  for (int u0 = wid; u0 < ns && !(dbg & 1); u0 += nw * kStage) {
    V r[kStage];
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      const int u = u0 + k * nw;
      const int uu = u < ns ? u : u0;                 // clamp: the tail re-reads a row it already has (no branch)
      r[k] = load_raw<T>(xl + static_cast<int64_t>(sh_cols[s0 + uu]) * ldx);
    }
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      const int u = u0 + k * nw;
      if (u < ns) tile[u * 64 + lane] = r[k];
    }
  }
  __syncthreads();

  // ---- phase 2
  const int64_t r_first = blk * rows_per_block + static_cast<int64_t>(wid) * kBlkRowsPerWave;
  if (r_first >= n_rows) return;
  const int nr = n_rows - r_first < kBlkRowsPerWave ? static_cast<int>(n_rows - r_first) : kBlkRowsPerWave;
  const int64_t nnz = rowptr[n_rows];
  // lane i holds rowptr[r_first + i] (i <= nr) and nlds[r_first + i] (i < nr)
  const int64_t rp_v = rowptr[r_first + (lane < nr ? lane : nr)];
  const int32_t nl_v = nlds[r_first + (lane < nr ? lane : nr - 1)];
  auto rp = [&](int i) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(rp_v & 0xffffffff), i);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<int>(rp_v >> 32), i);
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
  };

  // piece iterator: (row i, piece start p, row end e1); rows without entries are written (zeros) and long
  // rows handed to the queue as the iterator passes them — it visits every row of the wave exactly once
  int it_i = -1;
  int64_t it_p = 0, it_e1 = 0;
  auto next_piece = [&]() {     // returns false at the end
    it_p += kPiece;
    while (it_p >= it_e1) {
      ++it_i;
      if (it_i >= nr) return false;
      const int64_t e0 = rp(it_i), e1 = rp(it_i + 1);
      if (e1 - e0 > lq.long_len) {                      // the plan left plain source ids for these rows
        push_long_row(lq, r_first + it_i, e1 - e0, lane, 64);
        continue;
      }
      if (e1 == e0) {
        if (active) store4<T>(y + (r_first + it_i) * ldy + fc, make_float4(0.f, 0.f, 0.f, 0.f));
        continue;
      }
      it_p = e0;
      it_e1 = e1;
    }
    return true;
  };
  auto fetch = [&](int64_t p) -> uint2 {                // lane j <- entry p + j (clamped inside the arrays)
    int64_t i = p + lane;
    if (i >= nnz) i = nnz - 1;
    return make_uint2(static_cast<uint32_t>(ecode[i]), __float_as_uint(eval[i]));
  };

  it_p = -kPiece;   // so that the first next_piece() starts at row 0
  it_e1 = -kPiece;
  if (!next_piece()) return;
  uint2 pre = fetch(it_p);
  int buf = 0;
  float4 acc_l = make_float4(0.f, 0.f, 0.f, 0.f), acc_d = make_float4(0.f, 0.f, 0.f, 0.f);
  for (;;) {
    const int ci = it_i;
    const int64_t cp = it_p, ce1 = it_e1;
    uint2* sc = scratch + buf * kPiece;
    sc[lane] = pre;                                     // this piece's codes / values -> LDS scratch
    const bool more = next_piece();
    if (more) pre = fetch(it_p);                        // the NEXT piece's loads fly while this one is processed

    const int64_t e0 = rp(ci);
    const int64_t el = e0 + __builtin_amdgcn_readlane(nl_v, ci);
    const int len = ce1 - cp < kPiece ? static_cast<int>(ce1 - cp) : kPiece;
    const int l0 = 0;                                   // piece-local [l0, l1) = LDS entries, [d0, len) = gathered
    const int l1x = el > cp ? (el - cp < len ? static_cast<int>(el - cp) : len) : 0;
    const int d0 = l1x;
    const int l1 = (dbg & 2) ? 0 : l1x;
    if (cp == e0) {
      acc_l = make_float4(0.f, 0.f, 0.f, 0.f);
      acc_d = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // first group of gathers: issue now, consume after the LDS entries
    uint2 cv[DG];
    V xv[DG];
    const int dg0 = (dbg & 4) ? 0 : (len - d0 < DG ? len - d0 : DG);
    if (dg0 > 0) {
#pragma unroll
      for (int u = 0; u < DG; ++u) cv[u] = sc[d0 + (u < dg0 ? u : dg0 - 1)];
#pragma unroll
      for (int u = 0; u < DG; ++u) xv[u] = load_raw<T>(xl + static_cast<int64_t>(static_cast<int32_t>(cv[u].x)) * ldx);
    }
    // LDS entries
    int j = l0;
    for (; j + LB <= l1; j += LB) {
      uint2 c[LB];
      V t[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) c[u] = sc[j + u];
#pragma unroll
      for (int u = 0; u < LB; ++u) t[u] = tile[(c[u].x & 0x7fffffffu) * 64 + lane];
#pragma unroll
      for (int u = 0; u < LB; ++u) fma4(acc_l, __uint_as_float(c[u].y), widen(t[u]));
    }
    if (j < l1) {
      uint2 c[LB];
      V t[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) c[u] = sc[j + u < l1 ? j + u : l1 - 1];
#pragma unroll
      for (int u = 0; u < LB; ++u) t[u] = tile[(c[u].x & 0x7fffffffu) * 64 + lane];
#pragma unroll
      for (int u = 0; u < LB; ++u)
        if (j + u < l1) fma4(acc_l, __uint_as_float(c[u].y), widen(t[u]));
    }
    // gathered entries
    if (dg0 > 0) {
#pragma unroll
      for (int u = 0; u < DG; ++u)
        if (u < dg0) fma4(acc_d, __uint_as_float(cv[u].y), widen(xv[u]));
      for (int g = d0 + DG; g < len; g += DG) {
        const int dg = len - g < DG ? len - g : DG;
#pragma unroll
        for (int u = 0; u < DG; ++u) cv[u] = sc[g + (u < dg ? u : dg - 1)];
#pragma unroll
        for (int u = 0; u < DG; ++u) xv[u] = load_raw<T>(xl + static_cast<int64_t>(static_cast<int32_t>(cv[u].x)) * ldx);
#pragma unroll
        for (int u = 0; u < DG; ++u)
          if (u < dg) fma4(acc_d, __uint_as_float(cv[u].y), widen(xv[u]));
      }
    }
    if (cp + kPiece >= ce1 && active)                   // last piece of the row
      store4<T>(y + (r_first + ci) * ldy + fc,
                make_float4(acc_l.x + acc_d.x, acc_l.y + acc_d.y, acc_l.z + acc_d.z, acc_l.w + acc_d.w));
    if (!more) break;
    buf ^= 1;
  }
}

// ---- LDS-staged row blocks, bf16, two entries per instruction (k_spmm_blk2) ------------------------------------------
// k_spmm_blk above spends as many issue slots on an LDS-served entry as on a gathered one (13.8 VALU per entry, 8 bytes
// per lane) and runs one 16-wave block per CU, so its phases add up.  This form keeps the plan (sgf_spmm_plan) and
//   * treats a neighbour row as 32 lanes x 16 bytes: lanes 0-31 work on entry 2 j, lanes 32-63 on entry 2 j + 1 — for LDS
//     entries one ds_read_b128 serves both (two slots), for gathered entries one 16-byte-per-lane buffer load (the
//     k_spmm_seg_bf16x2 access), the two half-waves' sums are added when the row ends;
//   * a lane gets ITS entry's {code, value} with one ds_read_b64 from the wave's scratch piece (base + 8 * half);
//   * runs 64-row blocks with <= 144 staged rows (72 KiB): two 8-wave blocks per CU, one staging while the other multiplies.
constexpr int kBlk2Stage = 8;                       // staging loads in flight per lane (16 rows per wave)

template <int DG, int LB>
__global__ __launch_bounds__(512, 2) void k_spmm_blk2(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ ecode, const float* __restrict__ eval,
    const int32_t* __restrict__ nlds, const int32_t* __restrict__ sh_ptr, const int32_t* __restrict__ sh_cols,
    const uint16_t* __restrict__ x, uint32_t pitch, uint32_t x_bytes, uint16_t* __restrict__ y, int64_t ldy,
    int64_t n_rows, int32_t d, int32_t rows_per_block, int32_t lds_rows, int32_t chunk_blocks, LongQueue lq,
    int32_t dbg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned char* const tile = smem_raw;               // [slot][512 bytes]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int nw = static_cast<int>(blockDim.x >> 6);
  uint2* const scratch = reinterpret_cast<uint2*>(smem_raw + static_cast<size_t>(lds_rows) * 512) + wid * (2 * kPiece);
  const int64_t blk = xcd_remap(blockIdx.x, gridDim.x, chunk_blocks);
  const int q = lane & 31;
  const int hi = lane >> 5;
  const bool active = q * 8 < d;                      // d % 8 == 0 (checked by the launcher)
  const uint32_t lanebase = active ? static_cast<uint32_t>(q) * 16u : 0u;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, x_bytes, 0x00020000);
  auto row_load = [&](uint32_t col) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(col * pitch + lanebase), 0, 0);
  };

  // ---- phase 1: stage the shared rows, two per load instruction (half-wave each) ----
  const int s0 = sh_ptr[blk];
  const int ns = sh_ptr[blk + 1] - s0;
  for (int u0 = 2 * wid; u0 < ns && !(dbg & 1); u0 += 2 * nw * kBlk2Stage) {
    u32x4 r[kBlk2Stage];
#pragma unroll
    for (int k = 0; k < kBlk2Stage; ++k) {
      const int u = u0 + 2 * nw * k + hi;
      const int uu = u < ns ? u : ns - 1;             // clamp: re-read a valid row, not stored
      r[k] = row_load(static_cast<uint32_t>(sh_cols[s0 + uu]));
    }
#pragma unroll
    for (int k = 0; k < kBlk2Stage; ++k) {
      const int u = u0 + 2 * nw * k + hi;
      if (u < ns) *reinterpret_cast<u32x4*>(tile + u * 512 + q * 16) = r[k];
    }
  }
  __syncthreads();

  // ---- phase 2 ----
  const int64_t r_first = blk * rows_per_block + static_cast<int64_t>(wid) * kBlkRowsPerWave;
  if (r_first >= n_rows) return;
  const int nr = n_rows - r_first < kBlkRowsPerWave ? static_cast<int>(n_rows - r_first) : kBlkRowsPerWave;
  const int64_t nnz = rowptr[n_rows];
  const int64_t rp_v = rowptr[r_first + (lane < nr ? lane : nr)];
  const int32_t nl_v = nlds[r_first + (lane < nr ? lane : nr - 1)];
  auto rp = [&](int i) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(rp_v & 0xffffffff), i);
    const uint32_t hi32 = __builtin_amdgcn_readlane(static_cast<int>(rp_v >> 32), i);
    return static_cast<int64_t>((static_cast<uint64_t>(hi32) << 32) | lo);
  };
  uint16_t* const yl = y + (active ? q * 8 : 0);
  auto store_row = [&](int64_t row, const float (&a)[8]) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = a[k] + __shfl_xor(a[k], 32, 64);
    if (active && !hi) {
      uint4 o;
      o.x = static_cast<uint32_t>(f32_to_bf16(t[0])) | (static_cast<uint32_t>(f32_to_bf16(t[1])) << 16);
      o.y = static_cast<uint32_t>(f32_to_bf16(t[2])) | (static_cast<uint32_t>(f32_to_bf16(t[3])) << 16);
      o.z = static_cast<uint32_t>(f32_to_bf16(t[4])) | (static_cast<uint32_t>(f32_to_bf16(t[5])) << 16);
      o.w = static_cast<uint32_t>(f32_to_bf16(t[6])) | (static_cast<uint32_t>(f32_to_bf16(t[7])) << 16);
      *reinterpret_cast<uint4*>(yl + row * ldy) = o;
    }
  };

  int it_i = -1;
  int64_t it_p = 0, it_e1 = 0;
  auto next_piece = [&]() {
    it_p += kPiece;
    while (it_p >= it_e1) {
      ++it_i;
      if (it_i >= nr) return false;
      const int64_t e0 = rp(it_i), e1 = rp(it_i + 1);
      if (e1 - e0 > lq.long_len) {
        push_long_row(lq, r_first + it_i, e1 - e0, lane, 64);
        continue;
      }
      if (e1 == e0) {
        if (active && !hi) *reinterpret_cast<uint4*>(yl + (r_first + it_i) * ldy) = make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      it_p = e0;
      it_e1 = e1;
    }
    return true;
  };
  auto fetch = [&](int64_t p) -> uint2 {
    int64_t i = p + lane;
    if (i >= nnz) i = nnz - 1;
    return make_uint2(static_cast<uint32_t>(ecode[i]), __float_as_uint(eval[i]));
  };

  it_p = -kPiece;
  it_e1 = -kPiece;
  if (!next_piece()) return;
  uint2 pre = fetch(it_p);
  int buf = 0;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  auto fma8 = [&](float v, const u32x4& r) {
    acc[0] = fmaf(v, __uint_as_float(r.x << 16), acc[0]);
    acc[1] = fmaf(v, __uint_as_float(r.x & 0xffff0000u), acc[1]);
    acc[2] = fmaf(v, __uint_as_float(r.y << 16), acc[2]);
    acc[3] = fmaf(v, __uint_as_float(r.y & 0xffff0000u), acc[3]);
    acc[4] = fmaf(v, __uint_as_float(r.z << 16), acc[4]);
    acc[5] = fmaf(v, __uint_as_float(r.z & 0xffff0000u), acc[5]);
    acc[6] = fmaf(v, __uint_as_float(r.w << 16), acc[6]);
    acc[7] = fmaf(v, __uint_as_float(r.w & 0xffff0000u), acc[7]);
  };
  for (;;) {
    const int ci = it_i;
    const int64_t cp = it_p, ce1 = it_e1;
    uint2* const sc = scratch + buf * kPiece;
    sc[lane] = pre;
    // (read back below by other lanes of this wave: LDS executes a wave's instructions in order, the compiler must not
    // re-order them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool more = next_piece();
    if (more) pre = fetch(it_p);

    const int64_t e0 = rp(ci);
    const int64_t el = e0 + __builtin_amdgcn_readlane(nl_v, ci);
    const int len = ce1 - cp < kPiece ? static_cast<int>(ce1 - cp) : kPiece;
    const int l1x = el > cp ? (el - cp < len ? static_cast<int>(el - cp) : len) : 0;
    const int d0 = l1x;                                 // [0, l1) LDS entries, [d0, len) gathered
    const int l1 = (dbg & 2) ? 0 : l1x;
    if (cp == e0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    }
    // pair starting at piece-local entry i of a range ending at b: this half-wave's entry is i + hi; beyond the range the
    // address is clamped to the range's last entry and the weight is 0
    auto get = [&](int i, int b) -> uint2 {
      const int idx = i + hi;
      uint2 c = sc[idx < b ? idx : b - 1];
      if (idx >= b) c.y = 0u;
      return c;
    };

    // first group of gathers: issued now, consumed after the LDS entries
    uint2 cv[DG];
    u32x4 xv[DG];
    const int ng = (dbg & 4) ? 0 : len - d0;            // gathered entries in this piece
    if (ng > 0) {
#pragma unroll
      for (int u = 0; u < DG; ++u) cv[u] = get(d0 + 2 * u, len);
#pragma unroll
      for (int u = 0; u < DG; ++u) xv[u] = row_load(cv[u].x);
    }
    // LDS entries, LB pairs at a time
    for (int j = 0; j < l1; j += 2 * LB) {
      uint2 c[LB];
      u32x4 t[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) c[u] = get(j + 2 * u, l1);
#pragma unroll
      for (int u = 0; u < LB; ++u)
        t[u] = *reinterpret_cast<const u32x4*>(tile + (c[u].x & 0x7fffffffu) * 512 + q * 16);
#pragma unroll
      for (int u = 0; u < LB; ++u) fma8(__uint_as_float(c[u].y), t[u]);
    }
    // gathered entries
    if (ng > 0) {
#pragma unroll
      for (int u = 0; u < DG; ++u) fma8(__uint_as_float(cv[u].y), xv[u]);
      for (int g = d0 + 2 * DG; g < len; g += 2 * DG) {
#pragma unroll
        for (int u = 0; u < DG; ++u) cv[u] = get(g + 2 * u, len);
#pragma unroll
        for (int u = 0; u < DG; ++u) xv[u] = row_load(cv[u].x);
#pragma unroll
        for (int u = 0; u < DG; ++u) fma8(__uint_as_float(cv[u].y), xv[u]);
      }
    }
    if (cp + kPiece >= ce1) store_row(r_first + ci, acc);   // last piece of the row
    if (!more) break;
    buf ^= 1;
  }
}

template <typename T>
int launch_blocked(const int64_t* rowptr, const int32_t* ecode, const float* eval, const int32_t* nlds,
                   const int32_t* sh_ptr, const int32_t* sh_cols, const T* x, int64_t ldx, T* y, int64_t ldy,
                   int64_t n_rows, int32_t d, int32_t rows_per_block, int32_t lds_rows, hipStream_t st,
                   const LongQueue& lq, float* partial) {
  constexpr int UNROLL = 8;
  using V = typename Stored<T>::type;
  const int threads = rows_per_block / kBlkRowsPerWave * 64;
  const size_t lds_bytes = static_cast<size_t>(lds_rows) * 64 * sizeof(V) +
                           static_cast<size_t>(threads / 64) * kScratchPerWave;
  const int64_t nb = (n_rows + rows_per_block - 1) / rows_per_block;
  // one XCD walks ~4096 consecutive rows at a time (its L2 then holds one neighbourhood, as in xcd_remap)
  int chunk = 4096 / rows_per_block;
  if (chunk < 1) chunk = 1;
  // Two register budgets: "deep" keeps 4 KiB of gathers + 8 LDS entries in flight per wave (<= 128 VGPRs, 16 waves
  // per CU); "lean" halves both and fits 64 VGPRs, for block shapes of which the LDS admits 32 waves per CU.
  // SGF_SPMM_BLK_DEBUG (timing experiments only, results are then wrong): 1 = skip the staging loads,
  // 2 = skip the LDS entries, 4 = skip the gathered entries
#ifdef SGF_PROBES   // (make PROBES=1; the release library takes no debug mask)
  const char* dbg_env = getenv("SGF_SPMM_BLK_DEBUG");
  const int dbg = dbg_env ? atoi(dbg_env) : 0;
#else
  const int dbg = 0;
#endif
  const size_t per_cu = 160 * 1024;
  const bool lean = (per_cu / lds_bytes) * static_cast<size_t>(threads / 64) > 16;
  constexpr int DGd = sizeof(T) == 4 ? 4 : 8, DGl = sizeof(T) == 4 ? 2 : 4;
  auto deep_fn = &k_spmm_blk<T, DGd, DGd, 4>;
  auto lean_fn = &k_spmm_blk<T, DGl, DGl, 8>;
  static thread_local size_t attr_set[4] = {0, 0, 0, 0};   // per (dtype, variant): largest dynamic LDS enabled so far
  const int which = (sizeof(T) == 4 ? 0 : 2) + (lean ? 1 : 0);
  if (lds_bytes > attr_set[which]) {
    SGF_CHECK_HIP(hipFuncSetAttribute(lean ? reinterpret_cast<const void*>(lean_fn) : reinterpret_cast<const void*>(deep_fn),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    attr_set[which] = lds_bytes;
  }
  const char* v2_env = getenv("SGF_SPMM_BLK2");         // "0": keep the 8-byte-per-lane kernel (A/B)
  const uint64_t xb = static_cast<uint64_t>(n_rows) * static_cast<uint64_t>(ldx) * sizeof(T);
  if (sizeof(T) == 2 && d % 8 == 0 && d <= 256 && ldx % 8 == 0 && ldy % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(y) % 16 == 0 && xb < (static_cast<uint64_t>(1) << 32) && threads <= 512 &&
      !(v2_env && v2_env[0] == '0')) {
    const size_t lds2 = static_cast<size_t>(lds_rows) * 512 + static_cast<size_t>(threads / 64) * kScratchPerWave;
    auto fn2 = &k_spmm_blk2<8, 4>;
    static thread_local size_t attr2 = 0;
    if (lds2 > attr2) {
      SGF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(lds2)));
      attr2 = lds2;
    }
    hipLaunchKernelGGL(fn2, dim3(static_cast<unsigned>(nb)), dim3(threads), lds2, st, rowptr, ecode, eval, nlds, sh_ptr,
                       sh_cols, reinterpret_cast<const uint16_t*>(x), static_cast<uint32_t>(ldx * sizeof(T)),
                       static_cast<uint32_t>(xb), reinterpret_cast<uint16_t*>(y), ldy, n_rows, d, rows_per_block,
                       lds_rows, chunk, lq, dbg);
  } else if (lean)
    hipLaunchKernelGGL(lean_fn, dim3(static_cast<unsigned>(nb)), dim3(threads), lds_bytes, st, rowptr, ecode, eval,
                       nlds, sh_ptr, sh_cols, x, ldx, y, ldy, n_rows, d, rows_per_block, lds_rows, chunk, lq, dbg);
  else
    hipLaunchKernelGGL(deep_fn, dim3(static_cast<unsigned>(nb)), dim3(threads), lds_bytes, st, rowptr, ecode, eval,
                       nlds, sh_ptr, sh_cols, x, ldx, y, ldy, n_rows, d, rows_per_block, lds_rows, chunk, lq, dbg);
  SGF_LAUNCH_CHECK();
  if (lq.cap > 0) {
    const dim3 block(kWavesPerBlock * 64);
    const int nbl = lq.cap < 2048 ? lq.cap : 2048;
    hipLaunchKernelGGL((k_spmm_long_seg<T, UNROLL>), dim3(nbl), block, 0, st, rowptr, ecode, eval, x, ldx, d, lq,
                       partial);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_spmm_long_fin<T>), dim3(nbl), dim3(256), 0, st, lq, partial, d, y, ldy);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

template <typename T>
int launch(const int64_t* rowptr, const int32_t* colind, const float* val, const T* x, int64_t ldx, int64_t n_cols,
           T* y, int64_t ldy, int64_t n_rows, int32_t d, hipStream_t st, const LongQueue& lq,
           float* partial, bool stream) {
  constexpr int UNROLL = 8;
  const dim3 block(kWavesPerBlock * 64);
  const char* force = getenv("SGF_SPMM_KERNEL");          // timing experiments: wave | row | seg | seg2 | sub
  const std::string fk = force ? force : "";
  const uint64_t x_bytes = static_cast<uint64_t>(n_cols) * static_cast<uint64_t>(ldx) * sizeof(T);
  const bool fits32 = d <= 256 && n_cols > 0 && x_bytes < (static_cast<uint64_t>(1) << 32);   // 32-bit offsets reach all of X
  const bool pair_ok = sizeof(T) == 2 && d % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 &&
                       reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  // 64 < d <= 128, bf16 (the 100M recipe's hidden width): the stream kernel with half of each half-wave idle still
  // beats the half-wave-per-row kernel — 7.26 vs 8.39 ms = 6.35 vs 5.5 TB/s of 256-byte gathers on a uniform graph of
  // 6 M nodes, degree 29 (scripts/spmm_d128_probe.py): no per-row tail of dependent single loads, no idle half when the
  // two rows of a wave differ in length
  const bool narrow_stream = d > 64 && d <= 128 && fits32 && pair_ok && fk != "sub";
  if (d > 128 || narrow_stream) {
    const int64_t nb = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    // Kernel choice (A/B on one MI355X, ogbn-products scale, profiles/r02_spmm_structured.md):
    //   gathers out of HBM (uniform graph)      wave 10.46  row 10.49  seg 10.65  seg_bf16x2 10.66 ms  -> row
    //   gathers out of L2 (re-ordered, bf16)    wave  3.94  row  3.84  seg  3.96  seg_bf16x2  3.31 ms  -> seg_bf16x2
    //   gathers out of L2 (re-ordered, fp32)    wave  6.00  row  5.97  seg  7.37                       -> row
    // `stream` is the caller's statement that the CSR's gathers mostly hit in L2 (sgf_spmm_stream).
    const int64_t nbs = (n_rows + kWavesPerBlock * kSegRows - 1) / (kWavesPerBlock * kSegRows);
    int chunk_rows = 4096;                                   // one XCD walks ~4096 consecutive rows at a time
    if (const char* e = getenv("SGF_SPMM_CHUNK_ROWS")) chunk_rows = atoi(e) > 0 ? atoi(e) : chunk_rows;   // experiments
    const int chunk = chunk_rows / (kWavesPerBlock * kSegRows) > 0 ? chunk_rows / (kWavesPerBlock * kSegRows) : 1;
    if (fits32 && pair_ok && (stream || fk == "seg2" || (narrow_stream && fk.empty())))
      hipLaunchKernelGGL((k_spmm_seg_bf16x2<16>), dim3(static_cast<unsigned>(nbs)), block, 0, st, rowptr, colind, val,
                         reinterpret_cast<const uint16_t*>(x), static_cast<uint32_t>(ldx * sizeof(T)),
                         static_cast<uint32_t>(x_bytes), reinterpret_cast<uint16_t*>(y), ldy, n_rows, d, chunk, lq);
    else if (fits32 && fk == "seg")
      hipLaunchKernelGGL((k_spmm_seg<T, (sizeof(T) == 4 ? 8 : 16)>), dim3(static_cast<unsigned>(nbs)), block, 0, st,
                         rowptr, colind, val, x, static_cast<uint32_t>(ldx * sizeof(T)),
                         static_cast<uint32_t>(x_bytes), y, ldy, n_rows, d, chunk, lq);
    else if (fits32 && fk != "wave")
      hipLaunchKernelGGL((k_spmm_row<T, UNROLL>), dim3(static_cast<unsigned>(nb)), block, 0, st, rowptr, colind,
                         val, x, static_cast<uint32_t>(ldx * sizeof(T)), static_cast<uint32_t>(x_bytes), y, ldy,
                         n_rows, d, lq);
    else
      hipLaunchKernelGGL((k_spmm_wave<T, UNROLL>), dim3(static_cast<unsigned>(nb)), block, 0, st,
                         rowptr, colind, val, x, ldx, y, ldy, n_rows, d, lq);
  } else {
#define SGF_SUB(LPR_)                                                                         \
  {                                                                                           \
    constexpr int RPB = kWavesPerBlock * (64 / LPR_);                                         \
    const int64_t nb = (n_rows + RPB - 1) / RPB;                                              \
    hipLaunchKernelGGL((k_spmm_sub<T, LPR_, UNROLL>), dim3(static_cast<unsigned>(nb)), block, \
                       0, st, rowptr, colind, val, x, ldx, y, ldy, n_rows, d, lq);            \
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
  if (lq.cap > 0) {
    const int nb = lq.cap < 2048 ? lq.cap : 2048;
    hipLaunchKernelGGL((k_spmm_long_seg<T, UNROLL>), dim3(nb), block, 0, st, rowptr, colind, val, x, ldx, d,
                       lq, partial);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_spmm_long_fin<T>), dim3(nb), dim3(256), 0, st, lq, partial, d, y, ldy);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

}  // namespace

int spmm_long_rows(int dtype, const int64_t* rowptr, const int32_t* colind, const float* val, const void* x, int64_t ldx,
                   int32_t d, const LongQueue& lq, float* partial, void* y, int64_t ldy, hipStream_t st) {
  if (lq.cap <= 0) return SGF_OK;
  constexpr int UNROLL = 8;
  const dim3 block(kWavesPerBlock * 64);
  const int nb = lq.cap < 2048 ? lq.cap : 2048;
  if (dtype == SGF_BF16) {
    hipLaunchKernelGGL((k_spmm_long_seg<uint16_t, UNROLL>), dim3(nb), block, 0, st, rowptr, colind, val,
                       static_cast<const uint16_t*>(x), ldx, d, lq, partial);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_spmm_long_fin<uint16_t>), dim3(nb), dim3(256), 0, st, lq, partial, d,
                       static_cast<uint16_t*>(y), ldy);
  } else {
    hipLaunchKernelGGL((k_spmm_long_seg<float, UNROLL>), dim3(nb), block, 0, st, rowptr, colind, val,
                       static_cast<const float*>(x), ldx, d, lq, partial);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_spmm_long_fin<float>), dim3(nb), dim3(256), 0, st, lq, partial, d, static_cast<float*>(y), ldy);
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace sgf

using namespace sgf;

namespace {
int spmm_common(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x, int64_t ldx,
                int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d, int32_t dtype, const LongQueue& lq,
                float* partial, hipStream_t st, const char* fn, bool stream = false) {
  SGF_REQUIRE(n_rows >= 0 && d >= 0 && n_cols >= 0, SGF_E_INVALID, "%s: negative size", fn);
  if (n_rows == 0 || d == 0) return SGF_OK;
  SGF_REQUIRE(rowptr && x && y, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(n_rows < (static_cast<int64_t>(1) << 31) * kWavesPerBlock, SGF_E_UNSUPPORTED,
              "%s: n_rows too large for one launch", fn);
  SGF_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, SGF_E_INVALID,
              "%s: d, ldx, ldy must be multiples of 4 with ld >= d (d=%d ldx=%lld ldy=%lld)", fn, d,
              static_cast<long long>(ldx), static_cast<long long>(ldy));
  const size_t esz = dtype == SGF_BF16 ? 2 : 4;
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(x) % (4 * esz) == 0 &&
                  reinterpret_cast<uintptr_t>(y) % (4 * esz) == 0,
              SGF_E_INVALID, "%s: x / y must be aligned to 4 elements", fn);
  if (dtype == SGF_F32)
    return launch<float>(rowptr, colind, val, static_cast<const float*>(x), ldx, n_cols, static_cast<float*>(y), ldy,
                         n_rows, d, st, lq, partial, stream);
  if (dtype == SGF_BF16)
    return launch<uint16_t>(rowptr, colind, val, static_cast<const uint16_t*>(x), ldx, n_cols,
                            static_cast<uint16_t*>(y), ldy, n_rows, d, st, lq, partial, stream);
  set_error("%s: unknown dtype %d", fn, dtype);
  return SGF_E_INVALID;
}
}  // namespace

extern "C" int sgf_spmm(const int64_t* rowptr, const int32_t* colind, const float* val,
                        const void* x, int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows,
                        int32_t d, int32_t dtype, void* stream) {
  const LongQueue none{nullptr, nullptr, 0, INT64_MAX};
  return spmm_common(rowptr, colind, val, x, ldx, n_cols, y, ldy, n_rows, d, dtype, none, nullptr,
                     static_cast<hipStream_t>(stream), "sgf_spmm");
}

extern "C" int32_t sgf_spmm_segment_len(void) { return kSegLen; }

extern "C" size_t sgf_spmm_split_workspace_bytes(int64_t long_segments, int32_t d) {
  if (long_segments < 0 || d < 0) return 0;
  return 256 + align_up(static_cast<size_t>(long_segments) * sizeof(LongEntry), 256) +
         static_cast<size_t>(long_segments) * static_cast<size_t>(d) * sizeof(float);
}

static int spmm_split_impl(bool stream_hint, const char* fn, const int64_t* rowptr, const int32_t* colind, const float* val,
                              const void* x, int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows,
                              int32_t d, int32_t dtype, int64_t long_len, int64_t long_segments,
                              void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(long_len >= 1 && long_segments >= 0 && long_segments < (static_cast<int64_t>(1) << 31),
              SGF_E_INVALID, "%s: bad long_len / long_segments", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (long_segments == 0) {
    const LongQueue none{nullptr, nullptr, 0, INT64_MAX};
    return spmm_common(rowptr, colind, val, x, ldx, n_cols, y, ldy, n_rows, d, dtype, none, nullptr, st, fn,
                       stream_hint);
  }
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_spmm_split_workspace_bytes(long_segments, d),
              SGF_E_WORKSPACE, "%s: workspace too small", fn);
  char* ws = static_cast<char*>(workspace);
  LongQueue lq;
  lq.count = reinterpret_cast<int32_t*>(ws);
  lq.entries = reinterpret_cast<LongEntry*>(ws + 256);
  lq.cap = static_cast<int32_t>(long_segments);
  lq.long_len = long_len;
  float* partial = reinterpret_cast<float*>(
      ws + 256 + align_up(static_cast<size_t>(long_segments) * sizeof(LongEntry), 256));
  SGF_CHECK_HIP(hipMemsetAsync(lq.count, 0, sizeof(int32_t), st));
  return spmm_common(rowptr, colind, val, x, ldx, n_cols, y, ldy, n_rows, d, dtype, lq, partial, st, fn, stream_hint);
}

extern "C" int sgf_spmm_split(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x,
                              int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d,
                              int32_t dtype, int64_t long_len, int64_t long_segments, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return spmm_split_impl(false, "sgf_spmm_split", rowptr, colind, val, x, ldx, n_cols, y, ldy, n_rows, d, dtype,
                         long_len, long_segments, workspace, workspace_bytes, stream);
}

// sgf_spmm_split for a CSR whose gathers mostly hit in L2 (a locality-restoring node order, sgf_reorder): bf16 rows
// are then fetched two per 16-byte-per-lane load by the flattened stream kernel (k_spmm_seg_bf16x2)
extern "C" int sgf_spmm_stream(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x,
                               int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d,
                               int32_t dtype, int64_t long_len, int64_t long_segments, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return spmm_split_impl(true, "sgf_spmm_stream", rowptr, colind, val, x, ldx, n_cols, y, ldy, n_rows, d, dtype,
                         long_len, long_segments, workspace, workspace_bytes, stream);
}

// ---- LDS-staged row-block SpMM (plan from sgf_spmm_plan) ------------------------------------------------
extern "C" int32_t sgf_spmm_lds_rows_len(int32_t dtype) {
  // 144 KiB of the CU's 160 KiB LDS for staged rows (64 lanes x 8 B = 512 B per bf16 slot, 1 KiB per fp32 slot)
  return dtype == SGF_BF16 ? 288 : 144;   // + 1 KiB of code / value scratch per wave = exactly 160 KiB at 16 waves
}

extern "C" int sgf_spmm_blocked(const int64_t* rowptr, const int32_t* ecode, const float* eval,
                                const int32_t* nlds, const int32_t* sh_ptr, const int32_t* sh_cols,
                                const void* x, int64_t ldx, void* y, int64_t ldy, int64_t n_rows, int32_t d,
                                int32_t dtype, int32_t rows_per_block, int32_t lds_rows, int64_t long_len,
                                int64_t long_segments, void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_spmm_blocked";
  SGF_REQUIRE(n_rows >= 0 && d >= 0, SGF_E_INVALID, "%s: negative size", fn);
  if (n_rows == 0 || d == 0) return SGF_OK;
  SGF_REQUIRE(rowptr && ecode && eval && nlds && sh_ptr && sh_cols && x && y, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "%s: unknown dtype %d", fn, dtype);
  SGF_REQUIRE(d <= 256, SGF_E_UNSUPPORTED, "%s: d = %d > 256 (one wave per row)", fn, d);
  SGF_REQUIRE(rows_per_block >= kBlkRowsPerWave && rows_per_block <= 128 && rows_per_block % kBlkRowsPerWave == 0,
              SGF_E_INVALID, "%s: rows_per_block must be a multiple of %d in [%d, 128]", fn, kBlkRowsPerWave,
              kBlkRowsPerWave);
  SGF_REQUIRE(lds_rows >= 1 && lds_rows <= sgf_spmm_lds_rows_len(dtype), SGF_E_INVALID,
              "%s: lds_rows %d outside [1, %d]", fn, lds_rows, sgf_spmm_lds_rows_len(dtype));
  SGF_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, SGF_E_INVALID,
              "%s: d, ldx, ldy must be multiples of 4 with ld >= d", fn);
  const size_t esz = dtype == SGF_BF16 ? 2 : 4;
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(x) % (4 * esz) == 0 && reinterpret_cast<uintptr_t>(y) % (4 * esz) == 0,
              SGF_E_INVALID, "%s: x / y must be aligned to 4 elements", fn);
  SGF_REQUIRE(long_len >= 1 && long_segments >= 0 && long_segments < (static_cast<int64_t>(1) << 31),
              SGF_E_INVALID, "%s: bad long_len / long_segments", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  LongQueue lq{nullptr, nullptr, 0, long_len};
  float* partial = nullptr;
  if (long_segments > 0) {
    SGF_REQUIRE(workspace && workspace_bytes >= sgf_spmm_split_workspace_bytes(long_segments, d), SGF_E_WORKSPACE,
                "%s: workspace too small", fn);
    char* ws = static_cast<char*>(workspace);
    lq.count = reinterpret_cast<int32_t*>(ws);
    lq.entries = reinterpret_cast<LongEntry*>(ws + 256);
    lq.cap = static_cast<int32_t>(long_segments);
    partial = reinterpret_cast<float*>(ws + 256 + align_up(static_cast<size_t>(long_segments) * sizeof(LongEntry), 256));
    SGF_CHECK_HIP(hipMemsetAsync(lq.count, 0, sizeof(int32_t), st));
  } else {
    lq.long_len = INT64_MAX;   // no queue: every row is walked by its wave (the plan then has LDS codes everywhere)
  }
  if (dtype == SGF_F32)
    return launch_blocked<float>(rowptr, ecode, eval, nlds, sh_ptr, sh_cols, static_cast<const float*>(x), ldx,
                                 static_cast<float*>(y), ldy, n_rows, d, rows_per_block, lds_rows, st, lq, partial);
  return launch_blocked<uint16_t>(rowptr, ecode, eval, nlds, sh_ptr, sh_cols, static_cast<const uint16_t*>(x), ldx,
                                  static_cast<uint16_t*>(y), ldy, n_rows, d, rows_per_block, lds_rows, st, lq, partial);
}
