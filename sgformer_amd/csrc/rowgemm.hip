// rowgemm.hip — K8: the GCN layer's dense half as ONE streaming pass.
//
// Reference (large/ours.py:36-40, 87-93):   x = gcn_conv(...)            # SpMM, csrc/spmm.hip
//                                           x = self.W(x)                # Linear d -> d        <- here
//                                           x = self.bns[i](x)           # needs column mean / var of THAT   <- Σ, Σ² here
//                                           x = relu(x); x = x + layer_[i]        # sgf_bn_apply
//
//   sgf_gcn_epilogue_stats : y = a W^T + b (bf16 out, fp32 accumulate) and, in the same pass, the shifted column sums
//                            Σ_i (y_ij - shift_j), Σ_i (y_ij - shift_j)² that BatchNorm needs (of the ROUNDED y, the
//                            tensor sgf_bn_apply normalises afterwards) — one read of a, one write of y, nothing else.
//   sgf_gcn_epilogue_dx    : dx = dy W, the same kernel with W loaded transposed.
//
// Replaces hipBLASLt's [N, d] x [d, d] GEMM (0.61-0.63 ms at N = 2.45 M, d = 256: 4.0 TB/s of the 2.5 GB it must move)
// plus the separate sgf_colstats pass over y (0.33 ms).  Shape of the problem: 2 N d² flop = 0.13 ms of bf16 MFMA
// against 0.40 ms of HBM time — a copy with a matrix product attached, so the kernel is organised as a copy:
//
//   * W (d x d bf16 = 128 KiB) is loaded ONCE per block into LDS, laid out so that a matrix-core B fragment
//     (8 consecutive k of one output column) is one conflict-free ds_read_b128.  One 8-wave block per CU, persistent.
//   * every wave works ALONE on 32-row tiles — no block barrier after the W fill.  It loads its tile straight into
//     MFMA A-fragment registers (lane l: row l & 31, 16 bytes at k = 16 s + 8 (l >> 5); four consecutive loads cover
//     the same 128-byte lines), and requests the NEXT tile before computing the current one: 8 waves x 16 KiB x 2 in
//     flight per CU.
//   * per 32-column strip: 16 x v_mfma_f32_32x32x16_bf16; the accumulator layout (lane = output column, registers =
//     16 rows) makes the column statistics register-local adds; bias add, rounding, then the strip goes through a
//     2.5 KiB per-wave LDS patch to come back row-major for 64-byte-per-row coalesced stores.
#include "common.h"

#include <cstdlib>

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRgWaves = 8;
constexpr int kRgThreads = kRgWaves * 64;
constexpr int kStageStride = 144;                      // bytes per staged row (64 bf16 + pad: the two half-waves' rows
                                                       // land 16 banks apart)
constexpr int kStageBytes = 16 * kStageStride;         // per wave: 16 rows x two column strips

struct RowGemmArgs {
  const uint16_t* a; int64_t lda;
  const uint16_t* w; int64_t ldw; int trans;           // trans = 0: B^T[j][k] = w[j ldw + k];  1: = w[k ldw + j]
  const float* bias;                                   // [d_out] or null
  const float* shift;                                  // [d_out] or null (statistics only)
  float* spart;                                        // [gridDim.x][2][d_out] statistics partials, or null
  uint16_t* y; int64_t ldy;
  int64_t n;
  uint4* part;                                         // IO 1 (written) / IO 2 (read): [tiles][NS][2][64 lanes] x 16 B
  // PAIRED launch (pair != 0): blocks b and b + 8 (same XCD under the observed b % 8 placement) walk the SAME tiles, one
  // with (w, y), the other with (w2, y2) — two products of one operand whose second read is served by the XCD's L2.
  const uint16_t* w2; uint16_t* y2; int64_t ldy2; int pair;
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  const f32x2 v = {lo, hi};
  const bf16x2 b = __builtin_convertvector(v, bf16x2);
  return *reinterpret_cast<const uint32_t*>(&b);
}

// Writes of one lane read back by another lane of the same wave: LDS executes a wave's instructions in order, the
// compiler only has to be kept from re-ordering them.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// D = d_in = d_out (64 / 128 / 256); KS = D / 16 k-steps, NS = D / 32 column strips, processed in pairs so that a row
// leaves in 128-byte pieces (full cache lines: 64-byte pieces measured 0.71 ms against 0.55 ms at N = 2.45 M).
// IO: 0  y (row-major) = a W^T + bias
//     1  part = a W^T + bias kept in ACCUMULATOR layout (lane-contiguous 16-byte pieces, rounded to bf16): the first
//        half of a two-operand Linear [a1 | a2] W^T; no LDS patch, no statistics
//     2  y (row-major) = a W^T + part: the second half; the addend comes back into the registers it left from, so the
//        sum is rounded once, like a library GEMM with beta = 1 on the rounded first product
template <int D, bool STATS, int IO = 0, int DBG = 0>   // DBG: timing ablations only (SGF_ROWGEMM_DEBUG): 2 no stores, 4 no re-loads
__global__ __launch_bounds__(kRgThreads, 2) void k_rowgemm_bf16(RowGemmArgs p) {
  constexpr int KS = D / 16, NS = D / 32;
  constexpr int BT = D * 2 + 16;                       // bytes per row of B^T in LDS (16-byte slots rotate by one per row)
  __shared__ __attribute__((aligned(16))) unsigned char lds[D * BT + kRgWaves * kStageBytes + 2 * D * 4];
  unsigned char* const ldsB = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const stg = lds + D * BT + wave * kStageBytes;
  float* const cvec = reinterpret_cast<float*>(lds + D * BT + kRgWaves * kStageBytes);   // [bias | shift]
  // paired launch: role 1 multiplies with w2 into y2; both roles of a pair walk the same tiles
  const int role = p.pair ? (blockIdx.x >> 3) & 1 : 0;
  const int64_t vblock = p.pair ? (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3) : blockIdx.x;
  const int64_t vgrid = p.pair ? gridDim.x / 2 : gridDim.x;
  if (role) {
    p.w = p.w2;
    p.y = p.y2;
    p.ldy = p.ldy2;
  }

  // ---- W -> LDS, once ----
  if (!p.trans) {
    constexpr int CH = D / 8;                          // 16-byte chunks per row
    for (int c = tid; c < D * CH; c += kRgThreads) {
      const int j = c / CH, q = c % CH;
      *reinterpret_cast<uint4*>(ldsB + j * BT + 16 * q) = *reinterpret_cast<const uint4*>(p.w + j * p.ldw + 8 * q);
    }
  } else {
    constexpr int CH = D / 8;
    for (int c = tid; c < D * CH; c += kRgThreads) {
      const int k = c / CH, q = c % CH;                // 8 output columns j = 8 q .. 8 q + 7 of contraction index k
      const uint4 v = *reinterpret_cast<const uint4*>(p.w + k * p.ldw + 8 * q);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<uint16_t*>(ldsB + (8 * q + t) * BT + 2 * k) =
            static_cast<uint16_t>(t & 1 ? u[t >> 1] >> 16 : u[t >> 1] & 0xffffu);
    }
  }
  for (int c = tid; c < D; c += kRgThreads) {
    cvec[c] = p.bias ? p.bias[c] : 0.f;
    cvec[D + c] = (STATS && p.shift) ? p.shift[c] : 0.f;
  }
  __syncthreads();

  float s1[NS], s2[NS];
#pragma unroll
  for (int w = 0; w < NS; ++w) {
    s1[w] = 0.f;
    s2[w] = 0.f;
  }

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = vgrid * kRgWaves;
  auto load_tile = [&](int64_t t, bf16x8 (&dst)[KS]) {
    int64_t row = t * 32 + i31;
    if (row >= p.n) row = p.n - 1;                     // ragged end: any valid row; its results are masked
    const uint16_t* src = p.a + row * p.lda + 8 * hi;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };

  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  // staging patch: 16 rows x 64 columns (two strips).  write: value (row, column 32 c + i31) at
  // row * stride + 64 c + 2 i31.  read back: lane -> row lane >> 3 (+8), 16-byte chunk lane & 7.
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);

  bf16x8 cur[KS], nxt[KS];
  int64_t t = vblock * kRgWaves + wave;
  if (t < ntiles) load_tile(t, cur);
  for (; t < ntiles; t += nwaves) {
    const int64_t tn = t + nwaves;
    if (tn < ntiles && !(DBG & 4)) load_tile(tn, nxt);
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    uint16_t* const yrow = p.y + (row0 + (lane >> 3)) * p.ldy + 8 * (lane & 7);
    uint4* const ptile = IO != 0 ? p.part + t * (NS * 2 * 64) + lane : nullptr;
#pragma unroll
    for (int u = 0; u < NS / 2; ++u) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][r] = 0.f;
        acc[1][r] = 0.f;
      }
      const unsigned char* const bu = bfrag0 + 64 * u * BT;
      uint4 addend[2][2];
      if (IO == 2) {                                   // requested before the MFMAs, consumed after them
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) addend[c][q] = ptile[((2 * u + c) * 2 + q) * 64];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bu + 32 * s);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bu + 32 * BT + 32 * s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b1, acc[1], 0, 0, 0);
      }
      float bias_c[2], shift_c[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bias_c[c] = cvec[32 * (2 * u + c) + i31];
        shift_c[c] = STATS ? cvec[D + 32 * (2 * u + c) + i31] : 0.f;
      }
      // accumulator register r of lane (i31, hi): row (r & 3) + 8 (r >> 2) + 4 hi, column 32 w + i31.
      if (IO == 2) {                                   // the first operand's product, in this very layout
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint4 v = addend[c][q];
            const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[c][8 * q + 2 * e] += __uint_as_float(d4[e] << 16);
              acc[c][8 * q + 2 * e + 1] += __uint_as_float(d4[e] & 0xffff0000u);
            }
          }
      }
      if (IO == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            uint4 v;
            v.x = cvt_pk_bf16(acc[c][8 * q + 0] + bias_c[c], acc[c][8 * q + 1] + bias_c[c]);
            v.y = cvt_pk_bf16(acc[c][8 * q + 2] + bias_c[c], acc[c][8 * q + 3] + bias_c[c]);
            v.z = cvt_pk_bf16(acc[c][8 * q + 4] + bias_c[c], acc[c][8 * q + 5] + bias_c[c]);
            v.w = cvt_pk_bf16(acc[c][8 * q + 6] + bias_c[c], acc[c][8 * q + 7] + bias_c[c]);
            if (!(DBG & 2)) ptile[((2 * u + c) * 2 + q) * 64] = v;
          }
        __builtin_amdgcn_sched_barrier(0);             // keep the strip pairs apart (32 accumulator registers, not 128)
        continue;
      }
      // Rows 0-15 are registers 0-7, rows 16-31 registers 8-15: two passes through the 16-row patch.
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        uint32_t pk[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
            const uint32_t v = cvt_pk_bf16(acc[c][r] + bias_c[c], acc[c][r + 1] + bias_c[c]);
            const int rl = (r & 3) + 8 * ((r >> 2) & 1);                 // row inside the patch (+ 4 hi in st_w)
            *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
            pk[c][(r >> 1) & 3] = v;
          }
        }
        if (STATS) {
          if (!tail) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v0 = __uint_as_float(pk[c][q] << 16) - shift_c[c];
                const float v1 = __uint_as_float(pk[c][q] & 0xffff0000u) - shift_c[c];
                s1[2 * u + c] += v0 + v1;
                s2[2 * u + c] = fmaf(v0, v0, fmaf(v1, v1, s2[2 * u + c]));
              }
          } else {                                     // the one ragged tile of the launch: rows >= n do not count
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int64_t ra = row0 + 16 * hh + ((2 * q) & 3) + 8 * (q >> 1) + 4 * hi;
                const float v0 = ra < p.n ? __uint_as_float(pk[c][q] << 16) - shift_c[c] : 0.f;
                const float v1 = ra + 1 < p.n ? __uint_as_float(pk[c][q] & 0xffff0000u) - shift_c[c] : 0.f;
                s1[2 * u + c] += v0 + v1;
                s2[2 * u + c] = fmaf(v0, v0, fmaf(v1, v1, s2[2 * u + c]));
              }
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
          if ((!tail || row0 + 16 * hh + 8 * q + (lane >> 3) < p.n) && !(DBG & 2))
            *reinterpret_cast<uint4*>(yrow + (16 * hh + 8 * q) * p.ldy + 64 * u) = v;
        }
        wave_lds_sync();
      }
    }
    if (!(DBG & 4)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
  }

  if (STATS) {
    // hi = 0 / 1 hold the same columns, different rows; then the 8 waves of the block
    __syncthreads();                                   // every wave is done with its staging patch
    static_assert(kRgWaves * 2 * D * 4 <= kRgWaves * kStageBytes, "reduction scratch must fit the staging patches");
    float* const redb = reinterpret_cast<float*>(lds + D * BT);   // [wave][2][D]
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      const float a1s = s1[w] + __shfl_xor(s1[w], 32, 64);
      const float a2s = s2[w] + __shfl_xor(s2[w], 32, 64);
      if (hi == 0) {
        redb[(wave * 2 + 0) * D + 32 * w + i31] = a1s;
        redb[(wave * 2 + 1) * D + 32 * w + i31] = a2s;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * D; c += kRgThreads) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < kRgWaves; ++v) s += redb[v * 2 * D + c];
      p.spart[static_cast<int64_t>(blockIdx.x) * 2 * D + c] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Both input gradients of the two-operand Linear z = [y | x0] W^T (large/ours.py:36-38 differentiated) in one launch, with the
// gradient of x0 = layer_[0] ACCUMULATED IN PLACE across the layers (large/ours.py:83-93: x0 feeds every layer and its
// residual):
//     dy      = dz W[:, :D]
//     acc_out = dz W[:, D:] + gadd + acc_in          (gadd: the residual's gradient of this layer; acc_in: the layers before)
// Balanced roles: a block produces HC = D / ROLES output columns of BOTH results — its B^T in LDS is the virtual matrix
// [W[:, c0 : c0 + HC] | W[:, D + c0 : D + c0 + HC]] — so at D = 256 the two workgroups b and b + 8 of a pair read the same
// dz tiles (the second read an L2 hit) and the same number of addend bytes each, and neither falls behind the other.
// The addends arrive in the read-back layout of the staging patch (16 bytes per lane, requested before the strip's MFMAs).
struct Dx2AccArgs {
  const uint16_t* a; int64_t lda;                      // dz
  const uint16_t* w; int64_t ldw;                      // W [D, 2 D]
  uint16_t* y; int64_t ldy;                            // dy
  const uint16_t* g; int64_t ldg;                      // gadd or null
  const uint16_t* ai; int64_t ldai;                    // acc_in or null
  uint16_t* ao; int64_t ldao;                          // acc_out
  int64_t n;
};

template <int DK, int ROLES>
__global__ __launch_bounds__(kRgThreads) void k_dx2acc_bf16(Dx2AccArgs p) {
  constexpr int HC = DK / ROLES;                       // columns of each result per block
  constexpr int DN = 2 * HC;                           // virtual output width
  constexpr int KS = DK / 16, NS = DN / 32;
  constexpr int BT = DK * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[DN * BT + kRgWaves * kStageBytes];
  unsigned char* const ldsB = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const stg = lds + DN * BT + wave * kStageBytes;
  const int role = ROLES == 2 ? (blockIdx.x >> 3) & 1 : 0;
  const int64_t vblock = ROLES == 2 ? (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3) : blockIdx.x;
  const int64_t vgrid = ROLES == 2 ? gridDim.x / 2 : gridDim.x;
  const int c0 = role * HC;

  // ---- B^T -> LDS, once: B^T[j][k] = W[k][col(j)], col(j) = j < HC ? c0 + j : DK + c0 + (j - HC) ----
  {
    constexpr int CH = DN / 8;
    for (int c = tid; c < DK * CH; c += kRgThreads) {
      const int k = c / CH, q = c % CH;
      const int jv = 8 * q;
      const int col = jv < HC ? c0 + jv : DK + c0 + (jv - HC);
      const uint4 v = *reinterpret_cast<const uint4*>(p.w + k * p.ldw + col);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<uint16_t*>(ldsB + (jv + t) * BT + 2 * k) =
            static_cast<uint16_t>(t & 1 ? u[t >> 1] >> 16 : u[t >> 1] & 0xffffu);
    }
  }
  __syncthreads();

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = vgrid * kRgWaves;
  auto load_tile = [&](int64_t t, bf16x8 (&dst)[KS]) {
    int64_t row = t * 32 + i31;
    if (row >= p.n) row = p.n - 1;
    const uint16_t* src = p.a + row * p.lda + 8 * hi;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };
  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);

  bf16x8 cur[KS], nxt[KS];
  int64_t t = vblock * kRgWaves + wave;
  if (t < ntiles) load_tile(t, cur);
  for (; t < ntiles; t += nwaves) {
    const int64_t tn = t + nwaves;
    if (tn < ntiles) load_tile(tn, nxt);
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    const int64_t rb = row0 + (lane >> 3);               // this lane's row in the read-back, + 8 q + 16 hh
    const int cl = c0 + 8 * (lane & 7);                  // its first column inside a 64-column strip pair, + 64 u'
#pragma unroll
    for (int u = 0; u < NS / 2; ++u) {
      const bool is_acc = 64 * u >= HC;                  // compile-time after unrolling
      const int cu = cl + (is_acc ? 64 * u - HC : 64 * u);
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][r] = 0.f;
        acc[1][r] = 0.f;
      }
      // the addends of this strip pair, in the read-back layout: rows rb + 16 hh + 8 q
      uint4 ag[2][2], aa[2][2];
      if (is_acc) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            int64_t r = rb + 16 * hh + 8 * q;
            if (r >= p.n) r = p.n - 1;
            ag[hh][q] = p.g ? *reinterpret_cast<const uint4*>(p.g + r * p.ldg + cu) : make_uint4(0u, 0u, 0u, 0u);
            aa[hh][q] = p.ai ? *reinterpret_cast<const uint4*>(p.ai + r * p.ldai + cu) : make_uint4(0u, 0u, 0u, 0u);
          }
      }
      const unsigned char* const bu = bfrag0 + 64 * u * BT;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bu + 32 * s);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bu + 32 * BT + 32 * s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b1, acc[1], 0, 0, 0);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
            const uint32_t v = cvt_pk_bf16(acc[c][r], acc[c][r + 1]);
            const int rl = (r & 3) + 8 * ((r >> 2) & 1);
            *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
          const int64_t r = rb + 16 * hh + 8 * q;
          if (is_acc) {
            const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
            const uint32_t pg[4] = {ag[hh][q].x, ag[hh][q].y, ag[hh][q].z, ag[hh][q].w};
            const uint32_t pa[4] = {aa[hh][q].x, aa[hh][q].y, aa[hh][q].z, aa[hh][q].w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = __uint_as_float(pv[e] << 16) + __uint_as_float(pg[e] << 16) + __uint_as_float(pa[e] << 16);
              const float hi2 = __uint_as_float(pv[e] & 0xffff0000u) + __uint_as_float(pg[e] & 0xffff0000u) +
                                __uint_as_float(pa[e] & 0xffff0000u);
              o[e] = cvt_pk_bf16(lo, hi2);
            }
            v = make_uint4(o[0], o[1], o[2], o[3]);
            if (!tail || r < p.n) *reinterpret_cast<uint4*>(p.ao + r * p.ldao + cu) = v;
          } else {
            if (!tail || r < p.n) *reinterpret_cast<uint4*>(p.y + r * p.ldy + cu) = v;
          }
        }
        wave_lds_sync();
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The two-operand Linear  y = [a1 | a2] W^T + bias  (GraphConvLayer with use_init, large/ours.py:36-38) in ONE pass
// (+ BatchNorm's column sums).  The contraction runs over 2 D indices, so W is [D, 2 D]:
//   D <= 128: 64 KiB — one block holds all of it, a wave multiplies a1's tile, then a2's tile into the same accumulators;
//   D = 256 : 256 KiB, more than a CU's LDS — the launch is PAIRED: blocks b and b + 8 (one XCD under the observed
//             b % 8 placement) walk the same row tiles, each holding the 128 rows of W that produce ITS half of the output
//             columns (ROLES = 2).  Both read the a1 / a2 tiles; whichever runs behind finds them in the XCD's L2 (and
//             catches up: its loads return sooner), so the operands leave HBM once.  Correctness does not depend on the
//             placement or on the timing — only the traffic does.
// Per tile a wave keeps all NS accumulators of its role live (K-outer order): a1's 16 k-steps, then the load of the NEXT
// tile's a1 rows into the registers just freed, a2's k-steps, the load of the next a2 tile, epilogue as in k_rowgemm_bf16.
struct RowGemm2Args {
  const uint16_t* a1; int64_t lda1;
  const uint16_t* a2; int64_t lda2;
  const uint16_t* w; int64_t ldw;                      // [D, 2 D] row-major (row j = output column j)
  const float* bias;                                   // [D] or null
  const float* shift;                                  // [D] or null (statistics only)
  float* spart;                                        // [virtual blocks][2][D] statistics partials, or null
  uint16_t* y; int64_t ldy;
  int64_t n;
};

template <int D, int ROLES, bool STATS>
__global__ __launch_bounds__(kRgThreads, 2) void k_rowgemm2_bf16(RowGemm2Args p) {
  constexpr int KS = D / 16;                           // k-steps per operand
  constexpr int DC = D / ROLES;                        // output columns of one role
  constexpr int NS = DC / 32;                          // 32-column strips of one role
  constexpr int BT = 2 * D * 2 + 16;                   // bytes per row of B^T = W[j][0 .. 2 D) (+ one 16-byte slot of skew)
  static_assert(NS % 2 == 0, "strips are drained in pairs");
  __shared__ __attribute__((aligned(16))) unsigned char lds[DC * BT + kRgWaves * kStageBytes + 2 * DC * 4];
  unsigned char* const ldsB = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const stg = lds + DC * BT + wave * kStageBytes;
  float* const cvec = reinterpret_cast<float*>(lds + DC * BT + kRgWaves * kStageBytes);   // [bias | shift] of this role
  const int role = ROLES == 2 ? (blockIdx.x >> 3) & 1 : 0;
  const int64_t vblock = ROLES == 2 ? (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3) : blockIdx.x;
  const int64_t vgrid = ROLES == 2 ? gridDim.x / 2 : gridDim.x;
  const int c0 = role * DC;                            // first output column of this role

  // ---- this role's rows of W -> LDS, once ----
  {
    constexpr int CH = 2 * D / 8;                      // 16-byte chunks per row
    for (int c = tid; c < DC * CH; c += kRgThreads) {
      const int j = c / CH, q = c % CH;
      *reinterpret_cast<uint4*>(ldsB + j * BT + 16 * q) =
          *reinterpret_cast<const uint4*>(p.w + static_cast<int64_t>(c0 + j) * p.ldw + 8 * q);
    }
  }
  for (int c = tid; c < DC; c += kRgThreads) {
    cvec[c] = p.bias ? p.bias[c0 + c] : 0.f;
    cvec[DC + c] = (STATS && p.shift) ? p.shift[c0 + c] : 0.f;
  }
  __syncthreads();

  float s1[NS], s2[NS];
#pragma unroll
  for (int w = 0; w < NS; ++w) {
    s1[w] = 0.f;
    s2[w] = 0.f;
  }

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = vgrid * kRgWaves;
  auto load_tile = [&](const uint16_t* base, int64_t ld, int64_t t, bf16x8 (&dst)[KS]) {
    int64_t row = t * 32 + i31;
    if (row >= p.n) row = p.n - 1;                     // ragged end: any valid row; its results are masked
    const uint16_t* src = base + row * ld + 8 * hi;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };
  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);

  bf16x8 bufa[KS], bufb[KS];
  int64_t t = vblock * kRgWaves + wave;
  if (t < ntiles) {
    load_tile(p.a1, p.lda1, t, bufa);
    load_tile(p.a2, p.lda2, t, bufb);
  }
  for (; t < ntiles; t += nwaves) {
    const int64_t tn = t + nwaves;
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    f32x16 acc[NS];
#pragma unroll
    for (int w = 0; w < NS; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[w][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int w = 0; w < NS; ++w) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(bfrag0 + 32 * w * BT + 32 * s);
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bufa[s], b, acc[w], 0, 0, 0);
      }
    if (tn < ntiles) load_tile(p.a1, p.lda1, tn, bufa);            // into the registers the loop above has just read
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int w = 0; w < NS; ++w) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(bfrag0 + 32 * w * BT + 2 * D + 32 * s);
        acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bufb[s], b, acc[w], 0, 0, 0);
      }
    if (tn < ntiles) load_tile(p.a2, p.lda2, tn, bufb);

    uint16_t* const yrow = p.y + (row0 + (lane >> 3)) * p.ldy + c0 + 8 * (lane & 7);
#pragma unroll
    for (int u = 0; u < NS / 2; ++u) {
      float bias_c[2], shift_c[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bias_c[c] = cvec[32 * (2 * u + c) + i31];
        shift_c[c] = STATS ? cvec[DC + 32 * (2 * u + c) + i31] : 0.f;
      }
      // accumulator register r of lane (i31, hi): row (r & 3) + 8 (r >> 2) + 4 hi, column 32 w + i31.
      // Rows 0-15 are registers 0-7, rows 16-31 registers 8-15: two passes through the 16-row patch.
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        uint32_t pk[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
            const uint32_t v = cvt_pk_bf16(acc[2 * u + c][r] + bias_c[c], acc[2 * u + c][r + 1] + bias_c[c]);
            const int rl = (r & 3) + 8 * ((r >> 2) & 1);                 // row inside the patch (+ 4 hi in st_w)
            *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
            pk[c][(r >> 1) & 3] = v;
          }
        }
        if (STATS) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int64_t ra = row0 + 16 * hh + ((2 * q) & 3) + 8 * (q >> 1) + 4 * hi;
              const float v0 = (!tail || ra < p.n) ? __uint_as_float(pk[c][q] << 16) - shift_c[c] : 0.f;
              const float v1 = (!tail || ra + 1 < p.n) ? __uint_as_float(pk[c][q] & 0xffff0000u) - shift_c[c] : 0.f;
              s1[2 * u + c] += v0 + v1;
              s2[2 * u + c] = fmaf(v0, v0, fmaf(v1, v1, s2[2 * u + c]));
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
          if (!tail || row0 + 16 * hh + 8 * q + (lane >> 3) < p.n)
            *reinterpret_cast<uint4*>(yrow + (16 * hh + 8 * q) * p.ldy + 64 * u) = v;
        }
        wave_lds_sync();
      }
    }
  }

  if (STATS) {
    __syncthreads();                                   // every wave is done with its staging patch
    static_assert(kRgWaves * 2 * DC * 4 <= kRgWaves * kStageBytes, "reduction scratch must fit the staging patches");
    float* const redb = reinterpret_cast<float*>(lds + DC * BT);   // [wave][2][DC]
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      const float a1s = s1[w] + __shfl_xor(s1[w], 32, 64);
      const float a2s = s2[w] + __shfl_xor(s2[w], 32, 64);
      if (hi == 0) {
        redb[(wave * 2 + 0) * DC + 32 * w + i31] = a1s;
        redb[(wave * 2 + 1) * DC + 32 * w + i31] = a2s;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * DC; c += kRgThreads) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < kRgWaves; ++v) s += redb[v * 2 * DC + c];
      p.spart[vblock * 2 * D + (c / DC) * D + c0 + (c % DC)] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the GCN layer's dense half in ONE launch (large/ours.py:36-40 + :87-93 differentiated):
//     out = relu(BatchNorm(z)) + x0,   z = [Ax | x0] W^T + b          (forward)
//     dz  = BatchNorm'(relu'(gy))                                       (sgf_bn_bwd_apply's arithmetic, from gy, z and the
//                                                                        reduced statistics [sum g' | sum g' xhat])
//     d[Ax] = dz W[:, :D],     dx0 += gy + dz W[:, D:]                  (both input gradients + the residual's)
// gy and z are read in matrix-core A-fragment layout, dz is formed in registers per k-step and multiplied right away
// (K-outer order: all accumulators of a block live, the fragment registers of step s re-loaded with the NEXT tile's step s
// as soon as they have been consumed — a full tile of gy and z in flight per wave at all times).  The virtual output
// [d[Ax] | dx0] is 2 D columns wide and W is up to 256 KiB, so a block produces 128 virtual columns from its 128 columns of
// W and the launch runs ROLES = 2 D / 128 blocks per row tile: blocks b, b + 8, (b + 16, b + 24) sit on one XCD under the
// observed b % 8 placement and walk the same tiles, the XCD's L2 serves all but the first read of a tile.  The residual's
// gradient gy enters dx0 through the matrix cores as well: gy's fragments times an identity fragment (exact).  dx0 is
// ACCUMULATED across the layers of the branch (acc_in -> acc_out, bf16, kept in the matrix cores' accumulator layout — an
// opaque buffer of sgf_gcn_epilogue_partial_bytes) and leaves row-major after the last one: the 8-stream gradient sum of
// x0's consumers (k_sum_n<7>: 10 GB per step at ogbn-products scale) does not exist any more.
struct BnBwdDxArgs {
  const uint16_t* gy; int64_t ldg;
  const uint16_t* z; int64_t ldz;
  const float* mean; const float* rstd; const float* gamma; const float* beta;   // gamma / beta may be null
  const float* stats; float inv_n; int training; int relu;                       // stats [2 D] (training only)
  const uint16_t* w; int64_t ldw;                      // [D, 2 D] row-major
  uint16_t* dz; int64_t lddz;                          // out, row-major
  uint16_t* dy; int64_t lddy;                          // out, row-major: dz W[:, :D]
  const uint4* acc_in;                                 // opaque running dx0, or null
  uint4* acc_out;                                      // opaque running dx0 (when dx0 == null)
  uint16_t* dx0; int64_t lddx0;                        // out, row-major (the last layer of the chain), or null
  int add_gy;                                          // the layer has the residual  + x0
  int64_t n;
  uint32_t* sync;                                      // [virtual blocks][8 waves] arrival counters, zeroed before the launch
};

// Every block of a tile's group does the SAME work (so none of them drifts away from the others and out of the L2 window — a
// first version gave the d[Ax] columns to two blocks and the dx0 columns to the other two: the dx0 blocks, which also read
// and write the running sum, fell behind, lost their L2 hits and the launch ran 2.5 ms): block r of ROLES = D / 64 owns
// d[Ax] strips 2 r, 2 r + 1 AND dx0 strips 2 r, 2 r + 1, stores the dz columns of k-steps [4 r, 4 r + 4) and adds gy's
// columns of exactly those k-steps (they are its dx0 strips' own columns) through the identity fragments.  With RING = 4
// k-steps per phase that is: "in phase r of the tile, also store dz and run the identity multiplies" — one uniform branch
// per phase, none inside a phase.
// Cache policy of the streams only ONE block touches (gfx950 buffer instructions, aux operand): the outputs are written
// through and dropped from the XCD's L2 (sc1), the running sum is read with the streaming hint (nt) — the L2 then holds
// what the blocks of a group SHARE, the gy / z tiles, for long enough that the blocks behind the first one find them
// (with plain stores the write-allocated lines pushed them out first: PMC showed gy and z leaving HBM four times).
constexpr int kStreamOut = 16;                         // sc1
constexpr int kStreamIn = 2;                           // nt

// Rendezvous of the ROLES waves (one per block of a group) that walk the same tiles, once per tile: the blocks share
// the gy / z tiles through their XCD's L2, which keeps a line for a few microseconds only — left alone, the four blocks
// drift apart (nothing pulls them back together: the waves wait on LDS and the matrix cores, not on these loads) and each
// fetches the tiles from HBM for itself (PMC: 9 T of reads instead of 3 T).  Timing only, no data is handed over, so relaxed
// device-scope atomics suffice.  BOUNDED: a wave that waits longer than ~1 ms (its partners are not resident — the grid
// exceeds the free CUs) gives up for the rest of the launch; the result never depends on the rendezvous.
__device__ __forceinline__ bool tile_rendezvous(uint32_t* cnt, uint32_t target, int lane) {
  uint32_t ok = 1;
  if (lane == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 4000) {
        ok = 0;
        break;
      }
    }
  }
  return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// DBG (timing ablations only, SGF_GCN_BWD_DEBUG; results are then wrong): 1 no element-wise work (dz = gy), 2 no matrix-core
// work, 4 no stores, 8 no re-loads of the fragment slots
template <int D, int ROLES, bool MINE, int DBG>
__device__ __forceinline__ void bn_bwd_dx_phase(const int half, bf16x8 (&G)[4], bf16x8 (&Z)[4], f32x16 (&acc)[4],
                                                const float* __restrict__ coef, const unsigned char* __restrict__ bfrag0,
                                                const unsigned char* __restrict__ identp, const __amdgpu_buffer_rsrc_t dzrsrc,
                                                const uint32_t dzoff, const bool row_ok, const __amdgpu_buffer_rsrc_t grsrc,
                                                const __amdgpu_buffer_rsrc_t zrsrc, const uint32_t cgo, const uint32_t czo,
                                                const uint32_t ngo, const uint32_t nzo, const int hi) {
  constexpr int KS = D / 16, RING = 4, NS = 4;
  constexpr int BT = D * 2 + 16;
  bf16x8 dzr[RING];
  // (A) element-wise: dz fragments into registers, the dz store, the refill of the consumed slots — RING independent chains
#pragma unroll
  for (int j = 0; j < RING; ++j) {
    const int s = half * RING + j;
    const float* cf = coef + 16 * s + 8 * hi;
    const uint4 gq = *reinterpret_cast<const uint4*>(&G[j]);
    const uint4 zq = *reinterpret_cast<const uint4*>(&Z[j]);
    const uint32_t gu[4] = {gq.x, gq.y, gq.z, gq.w}, zu[4] = {zq.x, zq.y, zq.z, zq.w};
    uint32_t du[4] = {gq.x ^ zq.x, gq.y ^ zq.y, gq.z ^ zq.z, gq.w ^ zq.w};
#pragma unroll
    for (int h = 0; h < ((DBG & 1) ? 0 : 2); ++h) {
      const float4 t1 = *reinterpret_cast<const float4*>(cf + 4 * h);
      const float4 t0 = *reinterpret_cast<const float4*>(cf + D + 4 * h);
      const float4 a1 = *reinterpret_cast<const float4*>(cf + 2 * D + 4 * h);
      const float4 z1 = *reinterpret_cast<const float4*>(cf + 3 * D + 4 * h);
      const float4 z0 = *reinterpret_cast<const float4*>(cf + 4 * D + 4 * h);
      const float t1v[4] = {t1.x, t1.y, t1.z, t1.w}, t0v[4] = {t0.x, t0.y, t0.z, t0.w};
      const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, z1v[4] = {z1.x, z1.y, z1.z, z1.w}, z0v[4] = {z0.x, z0.y, z0.z, z0.w};
      float dv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t gw_ = gu[2 * h + (e >> 1)], zw_ = zu[2 * h + (e >> 1)];
        const float g = (e & 1) ? __uint_as_float(gw_ & 0xffff0000u) : __uint_as_float(gw_ << 16);
        const float zv = (e & 1) ? __uint_as_float(zw_ & 0xffff0000u) : __uint_as_float(zw_ << 16);
        const float act = fmaf(zv, t1v[e], t0v[e]);
        const float gm = act > 0.f ? g : 0.f;
        dv[e] = fmaf(a1v[e], gm, fmaf(zv, z1v[e], z0v[e]));
      }
      du[2 * h] = cvt_pk_bf16(dv[0], dv[1]);
      du[2 * h + 1] = cvt_pk_bf16(dv[2], dv[3]);
    }
    const u32x4 dq = {du[0], du[1], du[2], du[3]};
    dzr[j] = *reinterpret_cast<const bf16x8*>(&dq);
    // a lane past the last row gets an out-of-range offset, which the hardware drops (no branch)
    if (MINE && !(DBG & 4))
      __builtin_amdgcn_raw_buffer_store_b128(dq, dzrsrc, row_ok ? dzoff + 32u * s : 0xffffff00u, 0, kStreamOut);
    // step s + RING (of this tile, or of the next one) into the slots just consumed; gy's slot stays until phase (B) when
    // it still has to go through the identity multiply
    const uint32_t rz = s + RING < KS ? czo + 32u * (s + RING) : nzo + 32u * (s + RING - KS);
    const uint32_t rg = s + RING < KS ? cgo + 32u * (s + RING) : ngo + 32u * (s + RING - KS);
    if (DBG & 8) continue;
    const u32x4 zl = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, static_cast<int>(rz), 0, 0);
    Z[j] = *reinterpret_cast<const bf16x8*>(&zl);
    if (!MINE) {
      const u32x4 gl = __builtin_amdgcn_raw_buffer_load_b128(grsrc, static_cast<int>(rg), 0, 0);
      G[j] = *reinterpret_cast<const bf16x8*>(&gl);
    }
    if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two steps' chains overlap; four do not fit the registers
  }
  // (B) matrix cores
#pragma unroll
  for (int j = 0; j < RING; ++j) {
    const int s = half * RING + j;
    if (DBG & 2) {
      acc[j][0] += __uint_as_float(static_cast<uint32_t>(dzr[j][0]) << 16);
      continue;
    }
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(bfrag0 + 32 * w * BT + 32 * s);
      acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzr[j], b, acc[w], 0, 0, 0);
    }
    if (MINE) {
      // the residual's gradient: these 16 columns of gy are columns of this block's dx0 strip 2 + (j >> 1) — added through an
      // identity fragment (exact)
      const bf16x8 idf = *reinterpret_cast<const bf16x8*>(identp + 1024 * (j & 1));
      acc[2 + (j >> 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G[j], idf, acc[2 + (j >> 1)], 0, 0, 0);
      if (!(DBG & 8)) {
        const uint32_t rg = s + RING < KS ? cgo + 32u * (s + RING) : ngo + 32u * (s + RING - KS);
        const u32x4 gl = __builtin_amdgcn_raw_buffer_load_b128(grsrc, static_cast<int>(rg), 0, 0);
        G[j] = *reinterpret_cast<const bf16x8*>(&gl);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int D, int ROLES, int DBG = 0>
__global__ __launch_bounds__(kRgThreads, 2) void k_bn_bwd_dx_bf16(BnBwdDxArgs p) {
  constexpr int KS = D / 16;                           // k-steps
  constexpr int RING = 4;                              // fragment slots per stream: step s lives in slot s % RING and is
                                                       // re-loaded with step s + RING (of this tile or the next) once consumed
  constexpr int NS = 4;                                // strips per block: 0, 1 = d[Ax] strips 2 r, 2 r + 1;  2, 3 = dx0 strips
  constexpr int NSD = D / 32;                          // strips of d[Ax] = strips of dx0
  constexpr int BT = D * 2 + 16;                       // bytes per row of B^T
  static_assert(ROLES * 64 == D && KS / RING == ROLES, "block r owns 64 columns of each product and phase r of a tile");
  __shared__ __attribute__((aligned(16))) unsigned char lds[128 * BT + kRgWaves * kStageBytes + 5 * D * 4 + 2048];
  unsigned char* const ldsB = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const stg = lds + 128 * BT + wave * kStageBytes;
  float* const coef = reinterpret_cast<float*>(lds + 128 * BT + kRgWaves * kStageBytes);   // [T1 | T0 | A1 | Z1 | Z0][D]
  unsigned char* const identl = lds + 128 * BT + kRgWaves * kStageBytes + 5 * D * 4;         // [2][64 lanes] x 16 B
  // blocks b, b + 8, ... (b + 8 (ROLES - 1)) of a group of 8 ROLES consecutive blocks: one XCD, the same row tiles
  const int role = ROLES == 1 ? 0 : (blockIdx.x >> 3) % ROLES;
  const int64_t vblock = ROLES == 1 ? blockIdx.x : (blockIdx.x & 7) | ((blockIdx.x / (8 * ROLES)) << 3);
  const int64_t vgrid = gridDim.x / ROLES;
  const int c0 = 64 * role;                            // first column of this block's 64 columns of d[Ax] and of dx0

  // ---- B^T -> LDS: B^T[j][k] = W[k][c0 + j] (j < 64: d[Ax]),  W[k][D + c0 + j - 64] (j >= 64: dx0) ----
  {
    constexpr int CH = 128 / 8;
    for (int c = tid; c < D * CH; c += kRgThreads) {
      const int k = c / CH, q = c % CH;
      const int col = q < 8 ? c0 + 8 * q : D + c0 + 8 * (q - 8);
      const uint4 v = *reinterpret_cast<const uint4*>(p.w + static_cast<int64_t>(k) * p.ldw + col);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<uint16_t*>(ldsB + (8 * q + e) * BT + 2 * k) =
            static_cast<uint16_t>(e & 1 ? u[e >> 1] >> 16 : u[e >> 1] & 0xffffu);
    }
  }
  // per column, with xhat = z rs + m (m = -mean rs), ga rs = cs:   act = z T1 + T0   (the relu argument xhat ga + be),
  //   dz = cs (g' - k0 - xhat k1) = A1 g' + z Z1 + Z0,   A1 = cs, Z1 = -cs k1 rs, Z0 = -cs (k0 + k1 m)
  for (int c = tid; c < D; c += kRgThreads) {
    const float rs = p.rstd[c], m = -p.mean[c] * rs;
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
    const float k0 = p.training ? p.stats[c] * p.inv_n : 0.f, k1 = p.training ? p.stats[D + c] * p.inv_n : 0.f;
    const float cs = ga * rs;
    coef[c] = p.relu ? rs * ga : 0.f;                  // T1
    coef[D + c] = p.relu ? m * ga + be : 1.f;          // T0  (no activation: act = 1 > 0 for every element)
    coef[2 * D + c] = cs;                              // A1
    coef[3 * D + c] = -cs * k1 * rs;                   // Z1
    coef[4 * D + c] = -cs * (k0 + k1 * m);             // Z0
  }
  // identity fragments: B[k = 8 hi + e][j = i31] of the two k-steps that cover a strip's own 32 columns; zero when the
  // layer has no residual (the multiply still runs: no branch)
  if (tid < 128) {
    const int f = tid >> 6, l = tid & 63, li = l & 31, lh = l >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      reinterpret_cast<uint16_t*>(identl + 1024 * f + 16 * l)[e] =
          static_cast<uint16_t>((p.add_gy && li == 16 * f + 8 * lh + e) ? 0x3f80 : 0);
  }
  __syncthreads();

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = vgrid * kRgWaves;
  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);
  const uint64_t dz_bytes = static_cast<uint64_t>(p.n) * static_cast<uint64_t>(p.lddz) * 2;
  const __amdgpu_buffer_rsrc_t dzrsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.dz, 0, static_cast<uint32_t>(dz_bytes), 0x00020000);
  // gy and z through buffer descriptors too: 32-bit per-lane offsets instead of 64-bit pointers (registers)
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.gy), 0, static_cast<uint32_t>(static_cast<uint64_t>(p.n) * static_cast<uint64_t>(p.ldg) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(p.z), 0, static_cast<uint32_t>(static_cast<uint64_t>(p.n) * static_cast<uint64_t>(p.ldz) * 2), 0x00020000);
  const unsigned char* const identp = identl + 16 * lane;
  const __amdgpu_buffer_rsrc_t dyrsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.dy, 0, static_cast<uint32_t>(static_cast<uint64_t>(p.n) * static_cast<uint64_t>(p.lddy) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t dxrsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.dx0, 0, p.dx0 ? static_cast<uint32_t>(static_cast<uint64_t>(p.n) * static_cast<uint64_t>(p.lddx0) * 2) : 0u, 0x00020000);
  const uint32_t acc_bytes = static_cast<uint32_t>(((p.n + 31) / 32) * (NSD * 2 * 64 * 16));
  const __amdgpu_buffer_rsrc_t airsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.acc_in), 0, p.acc_in ? acc_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t aorsrc = __builtin_amdgcn_make_buffer_rsrc(p.acc_out, 0, p.acc_out ? acc_bytes : 0u, 0x00020000);

  bf16x8 G[RING], Z[RING];
  int64_t t = vblock * kRgWaves + wave;
  auto rowptr_of = [&](int64_t tt) {
    int64_t row = tt * 32 + i31;
    if (row >= p.n) row = p.n - 1;                     // ragged end: any valid row; its results are masked
    return row;
  };
  if (t < ntiles) {
    const int64_t row = rowptr_of(t);
    const uint32_t sg = static_cast<uint32_t>((row * p.ldg + 8 * hi) * 2), sz = static_cast<uint32_t>((row * p.ldz + 8 * hi) * 2);
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      const u32x4 gl = __builtin_amdgcn_raw_buffer_load_b128(grsrc, static_cast<int>(sg + 32u * s), 0, 0);
      const u32x4 zl = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, static_cast<int>(sz + 32u * s), 0, 0);
      G[s] = *reinterpret_cast<const bf16x8*>(&gl);
      Z[s] = *reinterpret_cast<const bf16x8*>(&zl);
    }
  }

  uint32_t* const my_sync = p.sync + vblock * kRgWaves + wave;
  bool in_step = ROLES > 1 && p.sync != nullptr;
  uint32_t arrivals = 0;
  for (; t < ntiles; t += nwaves) {
    if (in_step) {
      arrivals += ROLES;
      in_step = tile_rendezvous(my_sync, arrivals, lane);
    }
    const int64_t tn = t + nwaves;
    const bool has_next = tn < ntiles;
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    const int64_t rowc = rowptr_of(t), rown = has_next ? rowptr_of(tn) : rowc;   // no next tile: re-load this one (unused)
    const uint32_t cg = static_cast<uint32_t>((rowc * p.ldg + 8 * hi) * 2), cz = static_cast<uint32_t>((rowc * p.ldz + 8 * hi) * 2);
    const uint32_t ng = static_cast<uint32_t>((rown * p.ldg + 8 * hi) * 2), nz = static_cast<uint32_t>((rown * p.ldz + 8 * hi) * 2);
    const bool row_ok = row0 + i31 < p.n;
    const uint32_t dzoff = static_cast<uint32_t>(((row0 + i31) * p.lddz + 8 * hi) * 2);

    f32x16 acc[NS];
#pragma unroll
    for (int w = 0; w < NS; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[w][r] = 0.f;

#pragma unroll
    for (int half = 0; half < KS / RING; ++half) {
      if (half == role)
        bn_bwd_dx_phase<D, ROLES, true, DBG>(half, G, Z, acc, coef, bfrag0, identp, dzrsrc, dzoff, row_ok, grsrc, zrsrc, cg, cz,
                                             ng, nz, hi);
      else
        bn_bwd_dx_phase<D, ROLES, false, DBG>(half, G, Z, acc, coef, bfrag0, identp, dzrsrc, dzoff, row_ok, grsrc, zrsrc, cg, cz,
                                              ng, nz, hi);
    }

    // ---- epilogue: the d[Ax] pair leaves row-major through the patch; the dx0 pair row-major or in accumulator layout ----
    const uint32_t acc_off = static_cast<uint32_t>((t * (NSD * 2 * 64) + (2 * role * 2) * 64 + lane) * 16);
    if (p.acc_in != nullptr) {                         // the running dx0 of this tile's two strips
      u32x4 av[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          av[c][q] = __builtin_amdgcn_raw_buffer_load_b128(airsrc, static_cast<int>(acc_off + (c * 2 + q) * 1024), 0, kStreamIn);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t d4[4] = {av[c][q].x, av[c][q].y, av[c][q].z, av[c][q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[2 + c][8 * q + 2 * e] += __uint_as_float(d4[e] << 16);
            acc[2 + c][8 * q + 2 * e + 1] += __uint_as_float(d4[e] & 0xffff0000u);
          }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && p.dx0 == nullptr) {                // running sum stays in accumulator layout
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const u32x4 v = {cvt_pk_bf16(acc[2 + c][8 * q + 0], acc[2 + c][8 * q + 1]),
                             cvt_pk_bf16(acc[2 + c][8 * q + 2], acc[2 + c][8 * q + 3]),
                             cvt_pk_bf16(acc[2 + c][8 * q + 4], acc[2 + c][8 * q + 5]),
                             cvt_pk_bf16(acc[2 + c][8 * q + 6], acc[2 + c][8 * q + 7])};
            if (!(DBG & 4))
              __builtin_amdgcn_raw_buffer_store_b128(v, aorsrc, static_cast<int>(acc_off + (c * 2 + q) * 1024), 0, kStreamOut);
          }
        continue;
      }
      const int64_t ldo = u == 1 ? p.lddx0 : p.lddy;
      const uint32_t ooff = static_cast<uint32_t>(((row0 + (lane >> 3)) * ldo + c0 + 8 * (lane & 7)) * 2);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
            const uint32_t v = cvt_pk_bf16(acc[2 * u + c][r], acc[2 * u + c][r + 1]);
            const int rl = (r & 3) + 8 * ((r >> 2) & 1);
            *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint4 v4 = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
          const u32x4 v = {v4.x, v4.y, v4.z, v4.w};
          const bool ok = !tail || row0 + 16 * hh + 8 * q + (lane >> 3) < p.n;
          const uint32_t off = ok ? ooff + static_cast<uint32_t>((16 * hh + 8 * q) * ldo * 2) : 0xffffff00u;
          if (DBG & 4) continue;
          if (u == 1) __builtin_amdgcn_raw_buffer_store_b128(v, dxrsrc, static_cast<int>(off), 0, kStreamOut);
          else __builtin_amdgcn_raw_buffer_store_b128(v, dyrsrc, static_cast<int>(off), 0, kStreamOut);
        }
        wave_lds_sync();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same skeleton for the attention-from-input apply passes (include/sgf.h, sgf_attn_h_fwd / sgf_attn_h_bwd_apply;
// large/ours.py:130-151 with the projections folded into d x d matrices): a streamed [n, d] operand times a resident
// d x d matrix, with PER-ROW scalars in the epilogue.  In the accumulator layout a lane's 16 registers are 16 rows, so
// a row scalar is one register per accumulator register:
//   F   out = (h M + m) / den,  den = h.w + beta.   h.w comes out of the matrix cores too: a ninth "strip" whose B
//       fragment is w for EVERY lane gives D[row][*] = h_row . w — already one value per accumulator register
//       (w enters as hi + lo bf16 halves, two MFMA chains: the denominators keep fp32-grade accuracy).
//   B1  part = (g M^T) / den - ((g.o) / den) w        (dnum M^T + dden w, scaled after the product, in fp32);
//       g.o is a lane-local dot over the fragments of g and o, den is read; kept in accumulator layout
//   B2  dh = h D + ds + part
// B1 + B2 move 6 [n, d] tensors like the two k_apply_bf16 launches they replace (1.28 + 1.66 ms at N = 2.45 M),
// with whole-line stores, lane-contiguous partials and no block barrier.
constexpr int kHF = 0, kHB1 = 1, kHB2 = 2;

struct HRowArgs {
  const uint16_t* a; int64_t lda;        // F, B2: h    B1: g
  const uint16_t* a2; int64_t lda2;      // B1: o
  const float* bmat; int trans_b;        // fp32 [d, d]; B[k][j] = trans_b ? bmat[j d + k] : bmat[k d + j]
  const float* cvec;                     // F: m    B1: w    B2: ds
  const float* dvec; const float* beta;  // F: w and the device scalar beta
  float* den;                            // F: written    B1: read      [n]
  uint16_t* out; int64_t ldo;            // F, B2
  uint4* part;                           // B1: written    B2: read
  int64_t n;
  float2* rowscal;                       // B1, optional: (1 / den, dden = -(g.o) / den) per row, for the reduce pass
  const uint16_t* addend; int64_t ldadd;  // B2, optional: a row-major tensor added to the (rounded) result — the second
                                         // gradient of a tensor with two consumers, instead of a separate add pass
};

template <int D, int MODE>
__global__ __launch_bounds__(kRgThreads, 2) void k_hrow_bf16(HRowArgs p) {
  constexpr int KS = D / 16, NS = D / 32;
  constexpr int BT = D * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[D * BT + kRgWaves * kStageBytes + D * 4 + 2 * D * 2];
  unsigned char* const ldsB = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const stg = lds + D * BT + wave * kStageBytes;
  float* const cvec = reinterpret_cast<float*>(lds + D * BT + kRgWaves * kStageBytes);
  // F: w as two bf16 vectors, w = hi + lo to 16 mantissa bits (the denominators keep fp32-grade accuracy)
  uint16_t* const wdot = reinterpret_cast<uint16_t*>(lds + D * BT + kRgWaves * kStageBytes + D * 4);

  // ---- B^T -> LDS as bf16: B^T[j][k] = B[k][j] ----
  if (p.trans_b) {                                     // B[k][j] = bmat[j d + k]: rows of bmat are rows of B^T
    for (int c = tid; c < D * (D / 8); c += kRgThreads) {
      const int j = c / (D / 8), q = c % (D / 8);
      const float4 f0 = *reinterpret_cast<const float4*>(p.bmat + j * D + 8 * q);
      const float4 f1 = *reinterpret_cast<const float4*>(p.bmat + j * D + 8 * q + 4);
      uint4 v;
      v.x = cvt_pk_bf16(f0.x, f0.y); v.y = cvt_pk_bf16(f0.z, f0.w);
      v.z = cvt_pk_bf16(f1.x, f1.y); v.w = cvt_pk_bf16(f1.z, f1.w);
      *reinterpret_cast<uint4*>(ldsB + j * BT + 16 * q) = v;
    }
  } else {                                             // B[k][j] = bmat[k d + j]: scatter a row of bmat down a column
    for (int c = tid; c < D * (D / 4); c += kRgThreads) {
      const int k = c / (D / 4), q = c % (D / 4);
      const float4 f = *reinterpret_cast<const float4*>(p.bmat + k * D + 4 * q);
      const uint32_t lo = cvt_pk_bf16(f.x, f.y), hi2 = cvt_pk_bf16(f.z, f.w);
      *reinterpret_cast<uint16_t*>(ldsB + (4 * q + 0) * BT + 2 * k) = static_cast<uint16_t>(lo & 0xffffu);
      *reinterpret_cast<uint16_t*>(ldsB + (4 * q + 1) * BT + 2 * k) = static_cast<uint16_t>(lo >> 16);
      *reinterpret_cast<uint16_t*>(ldsB + (4 * q + 2) * BT + 2 * k) = static_cast<uint16_t>(hi2 & 0xffffu);
      *reinterpret_cast<uint16_t*>(ldsB + (4 * q + 3) * BT + 2 * k) = static_cast<uint16_t>(hi2 >> 16);
    }
  }
  for (int c = tid; c < D; c += kRgThreads) {
    cvec[c] = p.cvec[c];
    if (MODE == kHF) {
      const float wv = p.dvec[c];
      const uint32_t whi = cvt_pk_bf16(wv, 0.f) & 0xffffu;
      wdot[c] = static_cast<uint16_t>(whi);
      wdot[D + c] = static_cast<uint16_t>(cvt_pk_bf16(wv - __uint_as_float(whi << 16), 0.f) & 0xffffu);
    }
  }
  __syncthreads();
  const float beta = MODE == kHF ? p.beta[0] : 0.f;

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kRgWaves;
  auto load_tile = [&](int64_t t, bf16x8 (&dst)[KS]) {
    int64_t row = t * 32 + i31;
    if (row >= p.n) row = p.n - 1;
    const uint16_t* src = p.a + row * p.lda + 8 * hi;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };
  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);

  bf16x8 cur[KS], nxt[KS];
  int64_t t = static_cast<int64_t>(blockIdx.x) * kRgWaves + wave;
  if (t < ntiles) load_tile(t, cur);
  for (; t < ntiles; t += nwaves) {
    const int64_t tn = t + nwaves;
    if (tn < ntiles) load_tile(tn, nxt);
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    uint16_t* const yrow = p.out + (row0 + (lane >> 3)) * p.ldo + 8 * (lane & 7);
    uint4* const ptile = MODE != kHF ? p.part + t * (NS * 2 * 64) + lane : nullptr;

    // ---- per-row scalars, one per accumulator register: rs[r] multiplies the product, rc[r] the vector ----
    float rs[16], rc[16];
    if (MODE == kHF) {
      f32x16 dacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(wdot) + 32 * s + 16 * hi;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wb);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wb + 2 * D);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b, dacc, 0, 0, 0);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], bl, dacc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float den = dacc[r] + beta;
        rs[r] = 1.0f / den;
        rc[r] = rs[r];
        dacc[r] = den;
      }
      if (i31 == 0) {                                  // every lane of a half-wave holds the same 16 denominators
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t r0 = row0 + 8 * q + 4 * hi;
          if (!tail) {
            *reinterpret_cast<float4*>(p.den + r0) = make_float4(dacc[4 * q], dacc[4 * q + 1], dacc[4 * q + 2], dacc[4 * q + 3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (r0 + j < p.n) p.den[r0 + j] = dacc[4 * q + j];
          }
        }
      }
    } else if (MODE == kHB1) {
      // g.o of row i31: lane-local over this lane's k chunks, then the other half-wave's
      int64_t row = row0 + i31;
      if (row >= p.n) row = p.n - 1;
      const uint16_t* po = p.a2 + row * p.lda2 + 8 * hi;
      float dot = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint4 o = *reinterpret_cast<const uint4*>(po + 16 * s);
        const uint4 g = *reinterpret_cast<const uint4*>(&cur[s]);
        const uint32_t ov[4] = {o.x, o.y, o.z, o.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dot = fmaf(__uint_as_float(gv[e] << 16), __uint_as_float(ov[e] << 16), dot);
          dot = fmaf(__uint_as_float(gv[e] & 0xffff0000u), __uint_as_float(ov[e] & 0xffff0000u), dot);
        }
      }
      dot += __shfl_xor(dot, 32, 64);
      const float inv_row = 1.0f / p.den[row];
      const float coef_row = -dot * inv_row;                             // dden = -(g.o) / den
      if (p.rowscal && hi == 0 && row0 + i31 < p.n) p.rowscal[row0 + i31] = make_float2(inv_row, coef_row);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int src = (r & 3) + 8 * (r >> 2) + 4 * hi;                 // the lane that owns this register's row
        rs[r] = __shfl(inv_row, src, 64);
        rc[r] = __shfl(coef_row, src, 64);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rs[r] = 1.f;
        rc[r] = 1.f;
      }
    }

#pragma unroll
    for (int u = 0; u < NS / 2; ++u) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[0][r] = 0.f;
        acc[1][r] = 0.f;
      }
      const unsigned char* const bu = bfrag0 + 64 * u * BT;
      uint4 addend[2][2];
      if (MODE == kHB2) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) addend[c][q] = ptile[((2 * u + c) * 2 + q) * 64];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bu + 32 * s);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bu + 32 * BT + 32 * s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b1, acc[1], 0, 0, 0);
      }
      float vc[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) vc[c] = cvec[32 * (2 * u + c) + i31];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (MODE == kHB2) acc[c][r] += vc[c];
          else acc[c][r] = fmaf(rs[r], acc[c][r], rc[r] * vc[c]);
        }
      if (MODE == kHB2) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint4 v = addend[c][q];
            const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[c][8 * q + 2 * e] += __uint_as_float(d4[e] << 16);
              acc[c][8 * q + 2 * e + 1] += __uint_as_float(d4[e] & 0xffff0000u);
            }
          }
      }
      if (MODE == kHB1) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            uint4 v;
            v.x = cvt_pk_bf16(acc[c][8 * q + 0], acc[c][8 * q + 1]);
            v.y = cvt_pk_bf16(acc[c][8 * q + 2], acc[c][8 * q + 3]);
            v.z = cvt_pk_bf16(acc[c][8 * q + 4], acc[c][8 * q + 5]);
            v.w = cvt_pk_bf16(acc[c][8 * q + 6], acc[c][8 * q + 7]);
            ptile[((2 * u + c) * 2 + q) * 64] = v;
          }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
            const uint32_t v = cvt_pk_bf16(acc[c][r], acc[c][r + 1]);
            const int rl = (r & 3) + 8 * ((r >> 2) & 1);
            *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
          if (!tail || row0 + 16 * hh + 8 * q + (lane >> 3) < p.n) {
            if (MODE == kHB2 && p.addend) {
              const uint4 ad = *reinterpret_cast<const uint4*>(p.addend + (row0 + 16 * hh + 8 * q + (lane >> 3)) * p.ldadd +
                                                               64 * u + 8 * (lane & 7));
              const uint32_t a4[4] = {ad.x, ad.y, ad.z, ad.w};
              uint32_t v4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v4[e] = cvt_pk_bf16(__uint_as_float(v4[e] << 16) + __uint_as_float(a4[e] << 16),
                                    __uint_as_float(v4[e] & 0xffff0000u) + __uint_as_float(a4[e] & 0xffff0000u));
              v = make_uint4(v4[0], v4[1], v4[2], v4[3]);
            }
            *reinterpret_cast<uint4*>(yrow + (16 * hh + 8 * q) * p.ldo + 64 * u) = v;
          }
        }
        wave_lds_sync();
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K10: the two input stems read the SAME node features (large/ours.py:77 and :198,
// x = self.fcs[0](x) in GraphConv and in TransConv):  y0 = x W0^T + b0 (+ BatchNorm's column sums),  y1 = x W1^T + b1
// in ONE pass over x.  x is [n, f] with a small f (100 / 128): its rows are only 8-byte aligned and the contraction is
// 7-8 k-steps, so the A fragments are two 8-byte loads per k-step with the ragged last step zero-filled, a tile's
// fragments are 28-32 registers, and both weight matrices (2 D rows of <= 128 k) sit in LDS together.
constexpr int kStemMaxK = 128;                         // f <= 128
constexpr int kStemKS = kStemMaxK / 16;

struct StemArgs {
  const uint16_t* x; int64_t ldx; int d_in;            // d_in % 4 == 0
  const uint16_t* w0; int64_t ldw0; const float* b0;
  const uint16_t* w1; int64_t ldw1; const float* b1;   // w1 may be null: one output
  const float* shift;                                  // statistics of output 0 only
  float* spart;
  uint16_t* y0; int64_t ldy0;
  uint16_t* y1; int64_t ldy1;
  int64_t n;
};

template <int D, bool STATS>
__global__ __launch_bounds__(kRgThreads, 2) void k_stem_bf16(StemArgs p) {
  constexpr int NS1 = D / 32;                          // strips per output
  constexpr int BT = kStemMaxK * 2 + 16;               // bytes per row of B^T
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * D * BT + kRgWaves * kStageBytes + 3 * D * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  unsigned char* const ldsB = lds;
  unsigned char* const stg = lds + 2 * D * BT + wave * kStageBytes;
  float* const cvec = reinterpret_cast<float*>(lds + 2 * D * BT + kRgWaves * kStageBytes);   // [b0 | b1 | shift0]
  const int nout = p.w1 ? 2 : 1;
  const int ks = (p.d_in + 15) / 16;

  // ---- [W0 ; W1] -> LDS: B^T[j][k] = W[j][k], k >= d_in zero ----
  for (int c = tid; c < nout * D * (kStemMaxK / 4); c += kRgThreads) {
    const int j = c / (kStemMaxK / 4), q = c % (kStemMaxK / 4);
    const uint16_t* wsrc = j < D ? p.w0 + static_cast<int64_t>(j) * p.ldw0 : p.w1 + static_cast<int64_t>(j - D) * p.ldw1;
    uint2 v = make_uint2(0u, 0u);
    if (4 * q < p.d_in) v = *reinterpret_cast<const uint2*>(wsrc + 4 * q);
    *reinterpret_cast<uint2*>(ldsB + j * BT + 8 * q) = v;
  }
  for (int c = tid; c < D; c += kRgThreads) {
    cvec[c] = p.b0 ? p.b0[c] : 0.f;
    cvec[D + c] = (p.w1 && p.b1) ? p.b1[c] : 0.f;
    cvec[2 * D + c] = (STATS && p.shift) ? p.shift[c] : 0.f;
  }
  __syncthreads();

  float s1[NS1], s2[NS1];
#pragma unroll
  for (int w = 0; w < NS1; ++w) {
    s1[w] = 0.f;
    s2[w] = 0.f;
  }

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kRgWaves;
  auto load_tile = [&](int64_t t, bf16x8 (&dst)[kStemKS]) {
    int64_t row = t * 32 + i31;
    if (row >= p.n) row = p.n - 1;
    const uint16_t* src = p.x + row * p.ldx + 8 * hi;
#pragma unroll
    for (int s = 0; s < kStemKS; ++s) {
      const int k0 = 16 * s + 8 * hi;
      uint2 lo = make_uint2(0u, 0u), hi2 = make_uint2(0u, 0u);
      if (k0 < p.d_in) lo = *reinterpret_cast<const uint2*>(src + 16 * s);
      if (k0 + 4 < p.d_in) hi2 = *reinterpret_cast<const uint2*>(src + 16 * s + 4);
      const uint4 v = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
      dst[s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  const unsigned char* const bfrag0 = ldsB + i31 * BT + 16 * hi;
  unsigned char* const st_w = stg + 4 * hi * kStageStride + 2 * i31;
  const unsigned char* const st_r = stg + (lane >> 3) * kStageStride + 16 * (lane & 7);

  bf16x8 cur[kStemKS], nxt[kStemKS];
  int64_t t = static_cast<int64_t>(blockIdx.x) * kRgWaves + wave;
  if (t < ntiles) load_tile(t, cur);
  for (; t < ntiles; t += nwaves) {
    const int64_t tn = t + nwaves;
    if (tn < ntiles) load_tile(tn, nxt);
    const int64_t row0 = t * 32;
    const bool tail = row0 + 32 > p.n;
    for (int o = 0; o < nout; ++o) {
      uint16_t* const yrow = (o == 0 ? p.y0 + (row0 + (lane >> 3)) * p.ldy0 : p.y1 + (row0 + (lane >> 3)) * p.ldy1) +
                             8 * (lane & 7);
      const int64_t ldy = o == 0 ? p.ldy0 : p.ldy1;
#pragma unroll
      for (int u = 0; u < NS1 / 2; ++u) {
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[0][r] = 0.f;
          acc[1][r] = 0.f;
        }
        const unsigned char* const bu = bfrag0 + (o * D + 64 * u) * BT;
#pragma unroll
        for (int s = 0; s < kStemKS; ++s) {
          if (s < ks) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bu + 32 * s);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bu + 32 * BT + 32 * s);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s], b1, acc[1], 0, 0, 0);
          }
        }
        float bias_c[2], shift_c[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bias_c[c] = cvec[o * D + 32 * (2 * u + c) + i31];
          shift_c[c] = STATS ? cvec[2 * D + 32 * (2 * u + c) + i31] : 0.f;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          uint32_t pk[2][4];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; r += 2) {
              const uint32_t v = cvt_pk_bf16(acc[c][r] + bias_c[c], acc[c][r + 1] + bias_c[c]);
              const int rl = (r & 3) + 8 * ((r >> 2) & 1);
              *reinterpret_cast<uint16_t*>(st_w + rl * kStageStride + 64 * c) = static_cast<uint16_t>(v & 0xffffu);
              *reinterpret_cast<uint16_t*>(st_w + (rl + 1) * kStageStride + 64 * c) = static_cast<uint16_t>(v >> 16);
              pk[c][(r >> 1) & 3] = v;
            }
          }
          if (STATS && o == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int64_t ra = row0 + 16 * hh + ((2 * q) & 3) + 8 * (q >> 1) + 4 * hi;
                const float v0 = (!tail || ra < p.n) ? __uint_as_float(pk[c][q] << 16) - shift_c[c] : 0.f;
                const float v1 = (!tail || ra + 1 < p.n) ? __uint_as_float(pk[c][q] & 0xffff0000u) - shift_c[c] : 0.f;
                s1[2 * u + c] += v0 + v1;
                s2[2 * u + c] = fmaf(v0, v0, fmaf(v1, v1, s2[2 * u + c]));
              }
          }
          wave_lds_sync();
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(st_r + 8 * q * kStageStride);
            if (!tail || row0 + 16 * hh + 8 * q + (lane >> 3) < p.n)
              *reinterpret_cast<uint4*>(yrow + (16 * hh + 8 * q) * ldy + 64 * u) = v;
          }
          wave_lds_sync();
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kStemKS; ++s) cur[s] = nxt[s];
  }

  if (STATS) {
    __syncthreads();
    static_assert(kRgWaves * 2 * D * 4 <= kRgWaves * kStageBytes, "reduction scratch must fit the staging patches");
    float* const redb = reinterpret_cast<float*>(lds + 2 * D * BT);   // [wave][2][D]
#pragma unroll
    for (int w = 0; w < NS1; ++w) {
      const float a1s = s1[w] + __shfl_xor(s1[w], 32, 64);
      const float a2s = s2[w] + __shfl_xor(s2[w], 32, 64);
      if (hi == 0) {
        redb[(wave * 2 + 0) * D + 32 * w + i31] = a1s;
        redb[(wave * 2 + 1) * D + 32 * w + i31] = a2s;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * D; c += kRgThreads) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < kRgWaves; ++v) s += redb[v * 2 * D + c];
      p.spart[static_cast<int64_t>(blockIdx.x) * 2 * D + c] = s;
    }
  }
}

// stats[c] = Σ_blocks part[b][c], fixed order (deterministic).  One block per 64 entries, 4 block-groups per entry.
__global__ __launch_bounds__(256) void k_rowgemm_stats(const float* __restrict__ part, int nblk, int len,
                                                       float* __restrict__ stats) {
  __shared__ float red[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.f;
  if (c < len)
    for (int b = g; b < nblk; b += 4) s += part[static_cast<int64_t>(b) * len + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && c < len) stats[c] = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
}

int grid_blocks(int64_t n) {
  const int64_t tiles = (n + 31) / 32;
  int64_t b = (tiles + kRgWaves - 1) / kRgWaves;
  if (b > kNumCU) b = kNumCU;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

template <bool STATS, int IO>
int launch_rowgemm(const RowGemmArgs& a, int d, int blocks, hipStream_t st) {
  switch (d) {
    case 64: hipLaunchKernelGGL((k_rowgemm_bf16<64, STATS, IO>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
    case 128: hipLaunchKernelGGL((k_rowgemm_bf16<128, STATS, IO>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
    case 256: {
#ifdef SGF_PROBES   // timing ablations (make PROBES=1): SGF_ROWGEMM_DEBUG; not compiled into the release library
      static EnvInt dbg_env{"SGF_ROWGEMM_DEBUG", 0};
      const int dbg = dbg_env.get();
      switch (IO == 0 && !STATS ? dbg : 0) {
#define SGF_RG_DBG(X) case X: hipLaunchKernelGGL((k_rowgemm_bf16<256, false, 0, X>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
        SGF_RG_DBG(2) SGF_RG_DBG(4) SGF_RG_DBG(6)
#undef SGF_RG_DBG
        default: hipLaunchKernelGGL((k_rowgemm_bf16<256, STATS, IO>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
      }
#else
      hipLaunchKernelGGL((k_rowgemm_bf16<256, STATS, IO>), dim3(blocks), dim3(kRgThreads), 0, st, a);
#endif
      break;
    }
    default: set_error("sgf_gcn_epilogue: width %d not in {64, 128, 256}", d); return SGF_E_UNSUPPORTED;
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

bool supported(int32_t d_in, int32_t d_out, int32_t dtype) {
  if (dtype == SGF_F32) return linear_f32_supported(d_in, d_out);   // fp32 storage: csrc/linear_f32.hip
  return dtype == SGF_BF16 && d_in == d_out && (d_in == 64 || d_in == 128 || d_in == 256);
}

int check_common(const char* who, const void* a, int64_t lda, const void* w, int64_t ldw, int64_t n, int32_t d_in,
                 int32_t d_out, int32_t dtype, const void* y, int64_t ldy) {
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", who);
  SGF_REQUIRE(supported(d_in, d_out, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage with d_in == d_out in {64, 128, 256}, or fp32 storage with widths %% 4 == 0 up to 256 "
              "(got %d -> %d, dtype %d)", who, d_in, d_out, dtype);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(a && w && y, SGF_E_INVALID, "%s: null pointer", who);
  if (dtype == SGF_F32) {
    SGF_REQUIRE(lda >= d_in && ldy >= d_out && lda % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(a) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0,
                SGF_E_INVALID, "%s: rows must be 16-byte aligned (pointers %% 16, leading dims %% 4 elements)", who);
    return SGF_OK;
  }
  SGF_REQUIRE(lda >= d_in && ldy >= d_out && ldw >= (d_in > d_out ? d_in : d_out), SGF_E_INVALID,
              "%s: leading dimension smaller than the width", who);
  SGF_REQUIRE(lda % 8 == 0 && ldy % 8 == 0 && ldw % 8 == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(w) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0,
              SGF_E_INVALID, "%s: rows must be 16-byte aligned (pointers % 16, leading dims % 8 elements)", who);
  return SGF_OK;
}

template <int MODE>
int launch_hrow(const HRowArgs& a, int d, hipStream_t st) {
  const int blocks = grid_blocks(a.n);
  switch (d) {
    case 64: hipLaunchKernelGGL((k_hrow_bf16<64, MODE>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
    case 128: hipLaunchKernelGGL((k_hrow_bf16<128, MODE>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
    case 256: hipLaunchKernelGGL((k_hrow_bf16<256, MODE>), dim3(blocks), dim3(kRgThreads), 0, st, a); break;
    default: set_error("hrow: width %d not in {64, 128, 256}", d); return SGF_E_UNSUPPORTED;
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

bool rows16(const void* p, int64_t ld) { return reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld % 8 == 0; }

}  // namespace

// ---- internal entry points used by attn.hip (declared in common.h) ----
bool hrow_supported(int d, int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc,
                    const void* o, int64_t ldo) {
  return dtype == SGF_BF16 && (d == 64 || d == 128 || d == 256) && rows16(a, lda) && (!b || rows16(b, ldb)) &&
         (!c || rows16(c, ldc)) && rows16(o, ldo);
}

size_t hrow_partial_bytes(int64_t n, int d) { return static_cast<size_t>((n + 31) / 32) * 32 * static_cast<size_t>(d) * 2; }

int hrow_fwd(const void* h, int64_t ldh, int64_t n, int d, const float* M, const float* m, const float* w,
             const float* beta, void* out, int64_t ldo, float* den, hipStream_t st) {
  HRowArgs a{static_cast<const uint16_t*>(h), ldh, nullptr, 0, M, 0, m, w, beta, den, static_cast<uint16_t*>(out), ldo,
             nullptr, n, nullptr, nullptr, 0};
  return launch_hrow<kHF>(a, d, st);
}

// part = (g M^T) / den - ((g.o) / den) w;  rowscal (optional) = (1 / den, -(g.o) / den) per row
int hrow_bwd_pre(const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n, int d,
                 const float* M, const float* w, void* partial, float* rowscal, hipStream_t st) {
  HRowArgs a{static_cast<const uint16_t*>(g), ldg, static_cast<const uint16_t*>(o), ldo, M, 1, w, nullptr, nullptr,
             const_cast<float*>(den), nullptr, 0, static_cast<uint4*>(partial), n, reinterpret_cast<float2*>(rowscal),
             nullptr, 0};
  return launch_hrow<kHB1>(a, d, st);
}

// dh = h D + ds + part
int hrow_bwd_post(const void* h, int64_t ldh, int64_t n, int d, const float* Dm, const float* ds, const void* partial,
                  const void* addend, int64_t ldadd, void* dh, int64_t lddh, hipStream_t st) {
  HRowArgs b{static_cast<const uint16_t*>(h), ldh, nullptr, 0, Dm, 0, ds, nullptr, nullptr, nullptr,
             static_cast<uint16_t*>(dh), lddh, const_cast<uint4*>(static_cast<const uint4*>(partial)), n, nullptr,
             static_cast<const uint16_t*>(addend), ldadd};
  return launch_hrow<kHB2>(b, d, st);
}

int hrow_bwd(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den,
             int64_t n, int d, const float* M, const float* w, const float* Dm, const float* ds, void* dh, int64_t lddh,
             void* partial, hipStream_t st) {
  int rc = hrow_bwd_pre(g, ldg, o, ldo, den, n, d, M, w, partial, nullptr, st);
  if (rc != SGF_OK) return rc;
  return hrow_bwd_post(h, ldh, n, d, Dm, ds, partial, nullptr, 0, dh, lddh, st);
}

}  // namespace sgf

using namespace sgf;

extern "C" int32_t sgf_gcn_epilogue_supported(int32_t d_in, int32_t d_out, int32_t dtype) {
  return supported(d_in, d_out, dtype) ? 1 : 0;
}

extern "C" size_t sgf_gcn_epilogue_workspace_bytes(int64_t n, int32_t d_out) {
  if (n < 0 || d_out <= 0) return 0;
  int b = grid_blocks(n) > linear_f32_blocks(n) ? grid_blocks(n) : linear_f32_blocks(n);
  if (b < 8) b = 8;                                    // a paired launch runs whole groups of 8 pairs
  return static_cast<size_t>(b) * 2 * static_cast<size_t>(d_out) * sizeof(float);
}

extern "C" int sgf_gcn_epilogue_stats(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                                      int64_t n, int32_t d_in, int32_t d_out, int32_t dtype, void* y, int64_t ldy,
                                      const float* shift, float* stats, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  int rc = check_common("sgf_gcn_epilogue_stats", a, lda, w, ldw, n, d_in, d_out, dtype, y, ldy);
  if (rc != SGF_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (stats) SGF_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * static_cast<size_t>(d_out) * sizeof(float), st));
    return SGF_OK;
  }
  if (dtype == SGF_F32) {
    float* spart = nullptr;
    if (stats) {
      SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d_out), SGF_E_WORKSPACE,
                  "sgf_gcn_epilogue_stats: workspace %zu < %zu", workspace_bytes, sgf_gcn_epilogue_workspace_bytes(n, d_out));
      spart = static_cast<float*>(workspace);
    }
    rc = linear_f32(static_cast<const float*>(a), lda, n, d_in, d_out, static_cast<const float*>(w), ldw, 1, bias, nullptr,
                    0, shift, static_cast<float*>(y), ldy, spart, st);
    if (rc != SGF_OK || !stats) return rc;
    hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d_out + 63) / 64), dim3(256), 0, st, spart, linear_f32_blocks(n),
                       2 * d_out, stats);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int blocks = grid_blocks(n);
  RowGemmArgs args{static_cast<const uint16_t*>(a), lda, static_cast<const uint16_t*>(w), ldw, 0, bias, shift,
                   nullptr, static_cast<uint16_t*>(y), ldy, n, nullptr};
  if (!stats) return launch_rowgemm<false, 0>(args, d_out, blocks, st);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d_out), SGF_E_WORKSPACE,
              "sgf_gcn_epilogue_stats: workspace %zu < %zu", workspace_bytes, sgf_gcn_epilogue_workspace_bytes(n, d_out));
  args.spart = static_cast<float*>(workspace);
  rc = launch_rowgemm<true, 0>(args, d_out, blocks, st);
  if (rc != SGF_OK) return rc;
  hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d_out + 63) / 64), dim3(256), 0, st, args.spart, blocks, 2 * d_out,
                     stats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_gcn_epilogue_dx(const void* dy, int64_t lddy, const void* w, int64_t ldw, int64_t n, int32_t d_in,
                                   int32_t d_out, int32_t dtype, void* dx, int64_t lddx, void* stream) {
  // dx [n, d_in] = dy [n, d_out] W [d_out, d_in]: contraction over W's rows
  int rc = check_common("sgf_gcn_epilogue_dx", dy, lddy, w, ldw, n, d_out, d_in, dtype, dx, lddx);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  if (dtype == SGF_F32)
    return linear_f32(static_cast<const float*>(dy), lddy, n, d_out, d_in, static_cast<const float*>(w), ldw, 0, nullptr,
                      nullptr, 0, nullptr, static_cast<float*>(dx), lddx, nullptr, static_cast<hipStream_t>(stream));
  RowGemmArgs args{static_cast<const uint16_t*>(dy), lddy, static_cast<const uint16_t*>(w), ldw, 1, nullptr, nullptr,
                   nullptr, static_cast<uint16_t*>(dx), lddx, n, nullptr};
  return launch_rowgemm<false, 0>(args, d_in, grid_blocks(n), static_cast<hipStream_t>(stream));
}

// Both input gradients of the two-operand Linear from ONE read of dy out of HBM:  dx1 = dy W[:, :d], dx2 = dy W[:, d:]
// (large/ours.py:36-38 differentiated).  [W1 | W2] is 256 KiB in bf16 — more than a CU's LDS — so the launch is PAIRED:
// blocks b and b + 8 (one XCD) walk the same row tiles, one holding W1, the other W2; whichever runs behind finds the
// tile in the XCD's L2, which also pulls the two back together.  pair = 0 runs the two products as two launches.
extern "C" int sgf_gcn_epilogue_dx2(const void* dy, int64_t lddy, const void* w1, const void* w2, int64_t ldw, int64_t n,
                                    int32_t d, int32_t dtype, void* dx1, int64_t lddx1, void* dx2, int64_t lddx2,
                                    int32_t pair, void* stream) {
  int rc = check_common("sgf_gcn_epilogue_dx2", dy, lddy, w1, ldw, n, d, d, dtype, dx1, lddx1);
  if (rc != SGF_OK) return rc;
  rc = check_common("sgf_gcn_epilogue_dx2", dy, lddy, w2, ldw, n, d, d, dtype, dx2, lddx2);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32 || !pair || grid_blocks(n) < 16) {
    rc = sgf_gcn_epilogue_dx(dy, lddy, w1, ldw, n, d, d, dtype, dx1, lddx1, stream);
    if (rc != SGF_OK) return rc;
    return sgf_gcn_epilogue_dx(dy, lddy, w2, ldw, n, d, d, dtype, dx2, lddx2, stream);
  }
  RowGemmArgs args{static_cast<const uint16_t*>(dy), lddy, static_cast<const uint16_t*>(w1), ldw, 1, nullptr, nullptr,
                   nullptr, static_cast<uint16_t*>(dx1), lddx1, n, nullptr,
                   static_cast<const uint16_t*>(w2), static_cast<uint16_t*>(dx2), lddx2, 1};
  const int pairs = grid_blocks(n) / 16 * 8;          // whole groups of 8 pairs = 16 consecutive blocks
  return launch_rowgemm<false, 0>(args, d, 2 * pairs, st);
}

// dy = dz W[:, :d], acc_out = dz W[:, d:] + gadd + acc_in (gadd / acc_in nullable): see k_dx2acc_bf16
extern "C" int32_t sgf_gcn_epilogue_dx2_acc_supported(int32_t d, int32_t dtype) {
  return dtype == SGF_BF16 && (d == 64 || d == 128 || d == 256) ? 1 : 0;
}

extern "C" int sgf_gcn_epilogue_dx2_acc(const void* dz, int64_t lddz, const void* w, int64_t ldw, int64_t n, int32_t d,
                                        int32_t dtype, void* dy, int64_t lddy, const void* gadd, int64_t ldg,
                                        const void* acc_in, int64_t ldai, void* acc_out, int64_t ldao, void* stream) {
  const char* fn = "sgf_gcn_epilogue_dx2_acc";
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", fn);
  SGF_REQUIRE(sgf_gcn_epilogue_dx2_acc_supported(d, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, d in {64, 128, 256} (got d = %d, dtype %d)", fn, d, dtype);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(dz && w && dy && acc_out, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(lddz >= d && lddy >= d && ldao >= d && ldw >= 2 * d && (!gadd || ldg >= d) && (!acc_in || ldai >= d),
              SGF_E_INVALID, "%s: leading dimension smaller than the width", fn);
  SGF_REQUIRE(rows16(dz, lddz) && rows16(w, ldw) && rows16(dy, lddy) && rows16(acc_out, ldao) &&
                  (!gadd || rows16(gadd, ldg)) && (!acc_in || rows16(acc_in, ldai)),
              SGF_E_INVALID, "%s: rows must be 16-byte aligned (pointers %% 16, leading dims %% 8 elements)", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Dx2AccArgs a{static_cast<const uint16_t*>(dz), lddz, static_cast<const uint16_t*>(w), ldw, static_cast<uint16_t*>(dy), lddy,
               static_cast<const uint16_t*>(gadd), ldg, static_cast<const uint16_t*>(acc_in), ldai,
               static_cast<uint16_t*>(acc_out), ldao, n};
  if (d == 256) {
    int pairs = grid_blocks(n) / 16 * 8;                // whole groups of 8 pairs = 16 consecutive blocks
    if (pairs < 8) pairs = 8;
    hipLaunchKernelGGL((k_dx2acc_bf16<256, 2>), dim3(2 * pairs), dim3(kRgThreads), 0, st, a);
  } else if (d == 128) {
    hipLaunchKernelGGL((k_dx2acc_bf16<128, 1>), dim3(grid_blocks(n)), dim3(kRgThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((k_dx2acc_bf16<64, 1>), dim3(grid_blocks(n)), dim3(kRgThreads), 0, st, a);
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// ---- backward of the GCN layer's dense half: BatchNorm backward + both input gradients, dx0 accumulated ------------------
extern "C" int32_t sgf_gcn_bn_bwd_dx_supported(int32_t d, int32_t dtype) {
  return dtype == SGF_BF16 && (d == 64 || d == 128 || d == 256) ? 1 : 0;
}

extern "C" size_t sgf_gcn_bn_bwd_dx_workspace_bytes(int64_t n, int32_t d) {
  (void)n; (void)d;
  return static_cast<size_t>(kNumCU) * kRgWaves * sizeof(uint32_t);
}

extern "C" int sgf_gcn_bn_bwd_dx(const void* gy, int64_t ldg, const void* z, int64_t ldz, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int32_t relu, const float* stats,
                                 float inv_n, int32_t training, const void* w, int64_t ldw, int64_t n, int32_t d,
                                 int32_t dtype, void* dz, int64_t lddz, void* dy, int64_t lddy, const void* acc_in,
                                 void* acc_out, size_t acc_bytes, void* dx0, int64_t lddx0, int32_t add_gy, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_gcn_bn_bwd_dx";
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", fn);
  SGF_REQUIRE(sgf_gcn_bn_bwd_dx_supported(d, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, d in {64, 128, 256} (got d = %d, dtype %d)", fn, d, dtype);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(gy && z && mean && rstd && w && dz && dy && (!training || stats), SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE((dx0 != nullptr) != (acc_out != nullptr), SGF_E_INVALID,
              "%s: exactly one of dx0 (row-major result) and acc_out (running sum) must be given", fn);
  SGF_REQUIRE(ldg >= d && ldz >= d && lddz >= d && lddy >= d && ldw >= 2 * d && (!dx0 || lddx0 >= d), SGF_E_INVALID,
              "%s: leading dimension smaller than the width", fn);
  SGF_REQUIRE(rows16(gy, ldg) && rows16(z, ldz) && rows16(w, ldw) && rows16(dz, lddz) && rows16(dy, lddy) &&
                  (!dx0 || rows16(dx0, lddx0)),
              SGF_E_INVALID, "%s: rows must be 16-byte aligned (pointers %% 16, leading dims %% 8 elements)", fn);
  SGF_REQUIRE(static_cast<uint64_t>(n) * static_cast<uint64_t>(lddz) * 2 < 0xffffff00ull &&
                  static_cast<uint64_t>(n) * static_cast<uint64_t>(ldg) * 2 < 0xffffff00ull &&
                  static_cast<uint64_t>(n) * static_cast<uint64_t>(ldz) * 2 < 0xffffff00ull &&
                  static_cast<uint64_t>(n) * static_cast<uint64_t>(lddy) * 2 < 0xffffff00ull &&
                  (!dx0 || static_cast<uint64_t>(n) * static_cast<uint64_t>(lddx0) * 2 < 0xffffff00ull),
              SGF_E_UNSUPPORTED, "%s: an operand beyond 4 GiB (32-bit offsets)", fn);
  const size_t need = sgf_gcn_epilogue_partial_bytes(n, d);
  SGF_REQUIRE((!acc_in && !acc_out) || acc_bytes >= need, SGF_E_WORKSPACE, "%s: running-sum buffer %zu < %zu", fn, acc_bytes, need);
  SGF_REQUIRE((!acc_in || reinterpret_cast<uintptr_t>(acc_in) % 16 == 0) && (!acc_out || reinterpret_cast<uintptr_t>(acc_out) % 16 == 0),
              SGF_E_INVALID, "%s: running-sum buffers must be 16-byte aligned", fn);
  const int roles = d / 64;
  int vblocks = (grid_blocks(n) + roles - 1) / roles;
  if (roles > 1) {
    vblocks = (vblocks + 7) / 8 * 8;                   // whole groups of 8 tiles' worth of blocks: b, b + 8, ... share an XCD
    if (vblocks > kNumCU / roles) vblocks = kNumCU / roles;
  }
  BnBwdDxArgs a{static_cast<const uint16_t*>(gy), ldg, static_cast<const uint16_t*>(z), ldz, mean, rstd, gamma, beta,
                stats, inv_n, training, relu, static_cast<const uint16_t*>(w), ldw, static_cast<uint16_t*>(dz), lddz,
                static_cast<uint16_t*>(dy), lddy, static_cast<const uint4*>(acc_in), static_cast<uint4*>(acc_out),
                static_cast<uint16_t*>(dx0), lddx0, add_gy, n, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the per-tile rendezvous of a group's blocks (optional: without a workspace, or under SGF_GCN_BWD_SYNC=0, they run free)
  static EnvInt sync_env{"SGF_GCN_BWD_SYNC", 1};
  if (roles > 1 && workspace && workspace_bytes >= sgf_gcn_bn_bwd_dx_workspace_bytes(n, d) && sync_env.get() != 0 &&
      reinterpret_cast<uintptr_t>(workspace) % 4 == 0) {
    a.sync = static_cast<uint32_t*>(workspace);
    SGF_CHECK_HIP(hipMemsetAsync(workspace, 0, sgf_gcn_bn_bwd_dx_workspace_bytes(n, d), st));
  }
  if (d == 64) hipLaunchKernelGGL((k_bn_bwd_dx_bf16<64, 1>), dim3(vblocks), dim3(kRgThreads), 0, st, a);
  else if (d == 128) hipLaunchKernelGGL((k_bn_bwd_dx_bf16<128, 2>), dim3(vblocks * 2), dim3(kRgThreads), 0, st, a);
  else {
#ifdef SGF_PROBES   // timing ablations (make PROBES=1): SGF_GCN_BWD_DEBUG; not compiled into the release library
    static EnvInt dbg_env{"SGF_GCN_BWD_DEBUG", 0};
    switch (dbg_env.get()) {
#define SGF_BWD_DBG(X) case X: hipLaunchKernelGGL((k_bn_bwd_dx_bf16<256, 4, X>), dim3(vblocks * 4), dim3(kRgThreads), 0, st, a); break;
      SGF_BWD_DBG(1) SGF_BWD_DBG(2) SGF_BWD_DBG(3) SGF_BWD_DBG(4) SGF_BWD_DBG(8) SGF_BWD_DBG(12) SGF_BWD_DBG(15)
#undef SGF_BWD_DBG
      default: hipLaunchKernelGGL((k_bn_bwd_dx_bf16<256, 4>), dim3(vblocks * 4), dim3(kRgThreads), 0, st, a); break;
    }
#else
    hipLaunchKernelGGL((k_bn_bwd_dx_bf16<256, 4>), dim3(vblocks * 4), dim3(kRgThreads), 0, st, a);
#endif
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// ---- the two-operand Linear in ONE pass (paired launch at d = 256) ---------------------------------------------------
extern "C" int32_t sgf_gcn_epilogue_cat_supported(int32_t d, int32_t dtype) {
  return dtype == SGF_BF16 && (d == 64 || d == 128 || d == 256) ? 1 : 0;
}

extern "C" int sgf_gcn_epilogue_cat(const void* a1, int64_t lda1, const void* a2, int64_t lda2, const void* w, int64_t ldw,
                                    const float* bias, int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy,
                                    const float* shift, float* stats, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  const char* fn = "sgf_gcn_epilogue_cat";
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", fn);
  SGF_REQUIRE(sgf_gcn_epilogue_cat_supported(d, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, d in {64, 128, 256} (got d = %d, dtype %d)", fn, d, dtype);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (stats) SGF_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * static_cast<size_t>(d) * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(a1 && a2 && w && y, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(lda1 >= d && lda2 >= d && ldy >= d && ldw >= 2 * d, SGF_E_INVALID, "%s: leading dimension smaller than the width", fn);
  SGF_REQUIRE(rows16(a1, lda1) && rows16(a2, lda2) && rows16(w, ldw) && rows16(y, ldy), SGF_E_INVALID,
              "%s: rows must be 16-byte aligned (pointers %% 16, leading dims %% 8 elements)", fn);
  const int roles = d == 256 ? 2 : 1;
  int vblocks = grid_blocks(n);
  if (roles == 2) {                                    // whole groups of 8 pairs = 16 consecutive blocks
    vblocks = (vblocks + 1) / 2;
    vblocks = (vblocks + 7) / 8 * 8;
    if (vblocks > kNumCU / 2) vblocks = kNumCU / 2;
  }
  RowGemm2Args args{static_cast<const uint16_t*>(a1), lda1, static_cast<const uint16_t*>(a2), lda2,
                    static_cast<const uint16_t*>(w), ldw, bias, shift, nullptr, static_cast<uint16_t*>(y), ldy, n};
  if (stats) {
    SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d), SGF_E_WORKSPACE,
                "%s: workspace %zu < %zu", fn, workspace_bytes, sgf_gcn_epilogue_workspace_bytes(n, d));
    args.spart = static_cast<float*>(workspace);
  }
#define SGF_RG2(D_, R_)                                                                                                 \
  if (stats) hipLaunchKernelGGL((k_rowgemm2_bf16<D_, R_, true>), dim3(vblocks * R_), dim3(kRgThreads), 0, st, args);     \
  else hipLaunchKernelGGL((k_rowgemm2_bf16<D_, R_, false>), dim3(vblocks * R_), dim3(kRgThreads), 0, st, args)
  if (d == 64) { SGF_RG2(64, 1); }
  else if (d == 128) { SGF_RG2(128, 1); }
  else { SGF_RG2(256, 2); }
#undef SGF_RG2
  SGF_LAUNCH_CHECK();
  if (stats) {
    hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d + 63) / 64), dim3(256), 0, st, args.spart, vblocks, 2 * d, stats);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

// ---- two-operand Linear  y = [a1 | a2] W^T + bias  (GraphConvLayer with use_init, large/ours.py:36-38) -------------
extern "C" size_t sgf_gcn_epilogue_partial_bytes(int64_t n, int32_t d_out) {
  if (n < 0 || d_out <= 0) return 0;
  return static_cast<size_t>((n + 31) / 32) * 32 * static_cast<size_t>(d_out) * 2;
}

// fp32 storage parks the first operand's product as a plain [n, d_out] fp32 matrix
extern "C" size_t sgf_gcn_epilogue_dtype_partial_bytes(int64_t n, int32_t d_out, int32_t dtype) {
  if (dtype == SGF_F32) return n < 0 || d_out <= 0 ? 0 : static_cast<size_t>(n) * static_cast<size_t>(d_out) * 4;
  return sgf_gcn_epilogue_partial_bytes(n, d_out);
}

extern "C" int sgf_gcn_epilogue_partial(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                                        int64_t n, int32_t d_in, int32_t d_out, int32_t dtype, void* partial,
                                        size_t partial_bytes, void* stream) {
  int rc = check_common("sgf_gcn_epilogue_partial", a, lda, w, ldw, n, d_in, d_out, dtype, partial, d_out);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(partial_bytes >= sgf_gcn_epilogue_dtype_partial_bytes(n, d_out, dtype), SGF_E_WORKSPACE,
              "sgf_gcn_epilogue_partial: partial buffer %zu < %zu", partial_bytes,
              sgf_gcn_epilogue_dtype_partial_bytes(n, d_out, dtype));
  if (dtype == SGF_F32)
    return linear_f32(static_cast<const float*>(a), lda, n, d_in, d_out, static_cast<const float*>(w), ldw, 1, bias, nullptr,
                      0, nullptr, static_cast<float*>(partial), d_out, nullptr, static_cast<hipStream_t>(stream));
  RowGemmArgs args{static_cast<const uint16_t*>(a), lda, static_cast<const uint16_t*>(w), ldw, 0, bias, nullptr,
                   nullptr, nullptr, 0, n, static_cast<uint4*>(partial)};
  return launch_rowgemm<false, 1>(args, d_out, grid_blocks(n), static_cast<hipStream_t>(stream));
}

extern "C" int sgf_gcn_epilogue_stats_add(const void* a, int64_t lda, const void* w, int64_t ldw, const void* partial,
                                          size_t partial_bytes, int64_t n, int32_t d_in, int32_t d_out, int32_t dtype,
                                          void* y, int64_t ldy, const float* shift, float* stats, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_gcn_epilogue_stats_add", a, lda, w, ldw, n, d_in, d_out, dtype, y, ldy);
  if (rc != SGF_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (stats) SGF_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * static_cast<size_t>(d_out) * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(partial && reinterpret_cast<uintptr_t>(partial) % 16 == 0 &&
                  partial_bytes >= sgf_gcn_epilogue_dtype_partial_bytes(n, d_out, dtype),
              SGF_E_INVALID, "sgf_gcn_epilogue_stats_add: partial buffer missing, misaligned or too small");
  if (dtype == SGF_F32) {
    float* spart = nullptr;
    if (stats) {
      SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d_out), SGF_E_WORKSPACE,
                  "sgf_gcn_epilogue_stats_add: workspace too small");
      spart = static_cast<float*>(workspace);
    }
    rc = linear_f32(static_cast<const float*>(a), lda, n, d_in, d_out, static_cast<const float*>(w), ldw, 1, nullptr,
                    static_cast<const float*>(partial), d_out, shift, static_cast<float*>(y), ldy, spart, st);
    if (rc != SGF_OK || !stats) return rc;
    hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d_out + 63) / 64), dim3(256), 0, st, spart, linear_f32_blocks(n),
                       2 * d_out, stats);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int blocks = grid_blocks(n);
  RowGemmArgs args{static_cast<const uint16_t*>(a), lda, static_cast<const uint16_t*>(w), ldw, 0, nullptr, shift,
                   nullptr, static_cast<uint16_t*>(y), ldy, n,
                   const_cast<uint4*>(static_cast<const uint4*>(partial))};
  if (!stats) return launch_rowgemm<false, 2>(args, d_out, blocks, st);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d_out), SGF_E_WORKSPACE,
              "sgf_gcn_epilogue_stats_add: workspace %zu < %zu", workspace_bytes,
              sgf_gcn_epilogue_workspace_bytes(n, d_out));
  args.spart = static_cast<float*>(workspace);
  rc = launch_rowgemm<true, 2>(args, d_out, blocks, st);
  if (rc != SGF_OK) return rc;
  hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d_out + 63) / 64), dim3(256), 0, st, args.spart, blocks, 2 * d_out,
                     stats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_gcn_epilogue_apply(const void* y, int64_t ldy, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, const void* res, int64_t ldr,
                                      int32_t relu, int64_t n, int32_t d, int32_t dtype, void* out, int64_t ldo,
                                      void* stream) {
  SGF_REQUIRE(n <= 0 || (y && mean && rstd && out), SGF_E_INVALID, "sgf_gcn_epilogue_apply: null pointer");
  return sgf_bn_apply(y, ldy, mean, rstd, gamma, beta, res, ldr, relu, n, d, dtype, out, ldo, stream);
}

// ---- K10: both input stems from one read of the node features ----------------------------------------------------
extern "C" int32_t sgf_stem_pair_supported(int32_t d_in, int32_t d_out, int32_t dtype) {
  return dtype == SGF_BF16 && d_in > 0 && d_in % 4 == 0 && d_in <= kStemMaxK && (d_out == 64 || d_out == 128 || d_out == 256)
             ? 1 : 0;
}

extern "C" int sgf_stem_pair(const void* x, int64_t ldx, int64_t n, int32_t d_in, const void* w0, int64_t ldw0,
                             const float* bias0, const void* w1, int64_t ldw1, const float* bias1, int32_t d_out,
                             int32_t dtype, void* y0, int64_t ldy0, void* y1, int64_t ldy1, const float* shift0,
                             float* stats0, void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "sgf_stem_pair: negative n");
  SGF_REQUIRE(sgf_stem_pair_supported(d_in, d_out, dtype), SGF_E_UNSUPPORTED,
              "sgf_stem_pair: bf16 storage, d_in %% 4 == 0, d_in <= %d, d_out in {64, 128, 256} (got %d -> %d, dtype %d)",
              kStemMaxK, d_in, d_out, dtype);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (stats0) SGF_CHECK_HIP(hipMemsetAsync(stats0, 0, 2 * static_cast<size_t>(d_out) * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(x && w0 && y0 && (!w1 || y1), SGF_E_INVALID, "sgf_stem_pair: null pointer");
  SGF_REQUIRE(ldx >= d_in && ldx % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0 && ldw0 >= d_in && ldw0 % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(w0) % 8 == 0 &&
                  (!w1 || (ldw1 >= d_in && ldw1 % 4 == 0 && reinterpret_cast<uintptr_t>(w1) % 8 == 0)),
              SGF_E_INVALID, "sgf_stem_pair: rows of x / W must be 8-byte aligned (pointers %% 8, leading dims %% 4 elements)");
  SGF_REQUIRE(ldy0 >= d_out && ldy0 % 8 == 0 && reinterpret_cast<uintptr_t>(y0) % 16 == 0 &&
                  (!w1 || (ldy1 >= d_out && ldy1 % 8 == 0 && reinterpret_cast<uintptr_t>(y1) % 16 == 0)),
              SGF_E_INVALID, "sgf_stem_pair: rows of y must be 16-byte aligned");
  const int blocks = grid_blocks(n);
  StemArgs a{static_cast<const uint16_t*>(x), ldx, d_in, static_cast<const uint16_t*>(w0), ldw0, bias0,
             static_cast<const uint16_t*>(w1), ldw1, bias1, shift0, nullptr, static_cast<uint16_t*>(y0), ldy0,
             static_cast<uint16_t*>(y1), ldy1, n};
  if (stats0) {
    SGF_REQUIRE(workspace && workspace_bytes >= sgf_gcn_epilogue_workspace_bytes(n, d_out), SGF_E_WORKSPACE,
                "sgf_stem_pair: workspace %zu < %zu", workspace_bytes, sgf_gcn_epilogue_workspace_bytes(n, d_out));
    a.spart = static_cast<float*>(workspace);
  }
#define SGF_STEM(D_)                                                                                              \
  if (stats0) hipLaunchKernelGGL((k_stem_bf16<D_, true>), dim3(blocks), dim3(kRgThreads), 0, st, a);               \
  else hipLaunchKernelGGL((k_stem_bf16<D_, false>), dim3(blocks), dim3(kRgThreads), 0, st, a)
  if (d_out == 64) { SGF_STEM(64); }
  else if (d_out == 128) { SGF_STEM(128); }
  else { SGF_STEM(256); }
#undef SGF_STEM
  SGF_LAUNCH_CHECK();
  if (stats0) {
    hipLaunchKernelGGL(k_rowgemm_stats, dim3((2 * d_out + 63) / 64), dim3(256), 0, st, a.spart, blocks, 2 * d_out, stats0);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}
