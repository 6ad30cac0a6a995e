// attn.hip — T3: SGFormer linear global attention, forward and hand-derived backward.
//
// Reference arithmetic (large/ours.py:130-149,157; medium/ours.py:14-46; 100M/ours.py:12-53),
// per head, with  c = 1/(||Q||_F ||K||_F)  (norms over the whole [N,H,d] tensors):
//     S0 = K^T V   z0 = sum_l K_l        num = c Q S0 + N V      den = c Q.z0 + N      o = num/den
// Backward (re-derived with un-normalised partials so that every global
// reduction is a plain sum a node-sharded run can all-reduce):
//     dnum = g/den   dden = -(g.o)/den    dS0 = sum_n Q_n^T dnum_n    dz0 = sum_n Q_n dden_n
//     s  = c (<S0,dS0> + <z0,dz0>)                       ( = <qn,dqn> = <kn,dkn> )
//     dQ = c (dnum S0^T + dden z0) - s Q/||Q||^2
//     dK = c (V dS0^T + dz0)       - s K/||K||^2
//     dV = N dnum + c K dS0
//
// Two kernel skeletons, both on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32; D = A B with
// A lane l -> A[i = l & 31][k = l >> 5], B lane l -> B[k = l >> 5][j = l & 31]):
//
//  k_attn_reduce  (d x d  <-  [N x d]^T [N x d];  1024 threads = 16 waves, one block per CU)
//     streams 4096/DP-row tiles of the two operands through double-buffered LDS (global -> regs
//     issued a tile ahead, written to LDS after the MFMA phase), each wave owns a 64x64 block of the
//     d x d result as 2x2 MFMA tiles in 64 accumulator registers; for DP < 256 the spare waves take
//     disjoint row groups.  Column sums / sums of squares ride along on the VALU during staging.
//     Every block writes its partial; k_attn_finalize adds the partials in a fixed order
//     (deterministic two-stage reduction, no atomics).
//  k_attn_apply   ([N x d]  <-  [N x d] [d x d];  1024 threads = 16 waves, one block per CU)
//     each wave keeps a [d/2 x 32] piece (column strip x K-half) of the d x d matrix resident in
//     DP/4 registers for the whole kernel and streams row tiles of the left operand through LDS
//     (stride DP+4 floats: conflict-free ds_read_b128, one read feeds 4 MFMAs); the two K-halves
//     swap half of their accumulators through LDS.  Row-local scalars (den, dden, ...) are computed
//     while staging; the epilogue applies  out = ar[n]*acc + br[n]*cvec[j] + gr[n]*E[n][j].
//
// DP = d rounded up to 64 / 128 / 256; columns >= d and rows >= n are zero-filled in LDS.
#include "common.h"
#include "reduce_shared.h"

namespace sgf {
namespace {

// ------------------------------------------------------------------------------------------------
// shared geometry
// ------------------------------------------------------------------------------------------------
constexpr int kRedThreads = 1024;
// the per-block partial layout is shared with csrc/gramx.hip (reduce_shared.h)
constexpr int kTileElems = kRedTileElems;          // RG * DP * DP, identical for every DP
constexpr int kVecB = kRedVecB;                    // second column-sum vector (kModeBwdH) + 1 scalar
constexpr int kVecC = kRedVecC;                    // third column-sum vector (kModeGramLN)
constexpr int kPartialStride = kRedPartialStride;  // + [DP colsum | ssq_a | ssq_q | pad][DP colsum_b | s | pad][DP colsum_c | pad]
constexpr int kMaxBlocks = kRedMaxBlocks;          // persistent: one block per CU

constexpr int kModeFwd = 0;  // reduce: A=K, B=V          apply: out
constexpr int kModeBwd = 1;  // reduce: A=Q, B=dnum
constexpr int kModeGram = 2; // reduce: C = A^T B plus column sums of A (sgf_gram); no third stream
constexpr int kModeBwdH = 3; // reduce: A=h, B=dnum; vecA = sum h*dden, vecB = sum dnum, scalar = sum dden
constexpr int kModeBwdHS = 4; // the same sums with the per-row scalars (1/den, dden) READ (p.den = float2 per row,
                              // written by k_hrow_bf16<B1>): two streams, no row dot (bf16 only)
constexpr int kModeGramBN = 5; // Gram whose A operand is formed on the fly: A = BatchNorm'(relu'(g1 [+ g2])) from g1, g2, z and
                               // the reduced statistics (sgf_gram_bn_bwd: the stem's dW without a materialised dz; bf16 only)
constexpr int kModeGramLN = 6; // Gram whose A operand is the LayerNorm backward of g, formed on the fly: A = LN'(relu'(g)) from g, the
                               // pre-LayerNorm input and the saved row statistics (sgf_gram_ln_bwd: TransConv's stem; bf16 only);
                               // colsum = sum A (the Linear's bias gradient), colsum_b = sum g' (d beta), colsum_c = sum g' xhat
constexpr int kApplyFwd = 0, kApplyDQ = 1, kApplyDK = 2, kApplyDV = 3;
// attention from the un-projected input (sgf_attn_h_*): no E operand, no global scalars
constexpr int kApplyHFwd = 4;   // out = (h M + m) / (h.w + beta)
constexpr int kApplyHBwd1 = 5;  // dh  = dnum M^T + dden w
constexpr int kApplyHBwd2 = 6;  // dh += h D + ds

static inline int padded_dim(int d) { return d <= 64 ? 64 : (d <= 128 ? 128 : 256); }

struct ReduceArgs {
  const void* a;   // fwd: K      bwd: Q
  const void* b;   // fwd: V      bwd: g
  const void* q;   // fwd: Q (norm only)   bwd: o
  const float* den;  // bwd only: [n, heads]
  int64_t lda, ldb, ldq;
  int64_t n;
  int32_t d, heads, b_heads;  // b_heads: 1 -> operand b is shared across heads (use_weight=False)
  int32_t db;                 // valid columns of operand b (= d except for sgf_gram)
  float gscale;               // bwd: 1/H
  float* partial;             // [gridDim.x * heads][kPartialStride]
  // kModeGram, PAIRED launch (pair != 0, bf16): blocks b and b + 8 (one XCD) walk the same row tiles, one multiplying with
  // b, the other with b2 — A leaves HBM once, its second read is served by the XCD's L2.  Partials: [role][virtual block].
  const void* b2; int64_t ldb2; int32_t pair;
  // kModeGramBN: a = g1, q = z, a2 = g2 (or null); per-column coefficients of the BatchNorm backward
  const void* a2; int64_t lda2;
  const float *bn_mean, *bn_rstd, *bn_gamma, *bn_beta, *bn_stats;
  float bn_inv_n; int32_t bn_training, bn_relu;
  // kModeGramLN: a = g, q = the LayerNorm's input; bn_mean / bn_rstd are then PER-ROW [n] (the forward's saved statistics),
  // bn_gamma / bn_beta per column (null: no affine), bn_relu the activation behind the LayerNorm
};

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <typename T, int DP, int MODE>
__global__ __launch_bounds__(kRedThreads) void k_attn_reduce(ReduceArgs p) {
  constexpr int F4 = DP / 4;             // float4 chunks per row
  constexpr int R = kRedThreads / F4;    // rows per tile (16 / 32 / 64)
  constexpr int NB = DP / 64;            // 64x64 blocks per dimension
  constexpr int RG = 16 / (NB * NB);     // row groups
  constexpr int STEPS = R / (2 * RG);    // MFMA k-steps (2 rows each) per tile per wave

  __shared__ float lds[2 * 2 * R * DP];  // [buf][A|B][R][DP]  = 64 KiB
  float* const ldsA = lds;
  float* const ldsB = lds + 2 * R * DP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int head = blockIdx.y;
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  const int blk = wave % (NB * NB);
  const int grp = wave / (NB * NB);
  const int wm = blk / NB;
  const int wd = blk % NB;

  const int srow = tid / F4;
  const int col = (tid % F4) * 4;
  const bool col_ok = col < p.d;
  const bool colb_ok = col < p.db;

  const T* pa = static_cast<const T*>(p.a) + static_cast<int64_t>(head) * p.d + col;
  const T* pb = static_cast<const T*>(p.b) + (p.b_heads == 1 ? 0 : static_cast<int64_t>(head) * p.d) + col;
  const T* pq = static_cast<const T*>(p.q) + static_cast<int64_t>(head) * p.d + col;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 colsum = zero4();  // fwd: sum K        bwd: sum Q * dden
  float4 colsumb = zero4(); // bwdH: sum dnum
  float ssq_a = 0.f;        // fwd: sum K^2      bwdH: sum dden
  float ssq_q = 0.f;        // fwd: sum Q^2

  const int64_t ntiles = (p.n + R - 1) / R;
  float4 ra = zero4(), rb = zero4(), rq = zero4();
  float rden = 1.f;

  auto issue = [&](int64_t tile) {
    const int64_t row = tile * R + srow;
    const bool ok = col_ok && row < p.n;
    ra = ok ? load4<T>(pa + row * p.lda) : zero4();
    rb = (colb_ok && row < p.n) ? load4<T>(pb + row * p.ldb) : zero4();
    if (MODE != kModeGram) rq = ok ? load4<T>(pq + row * p.ldq) : zero4();
    if (MODE == kModeBwd || MODE == kModeBwdH) rden = (row < p.n) ? p.den[row * p.heads + head] : 1.f;
  };
  auto commit = [&](int buf) {
    float4 wa = ra, wb = rb;
    if (MODE == kModeFwd) {
      colsum.x += ra.x; colsum.y += ra.y; colsum.z += ra.z; colsum.w += ra.w;
      ssq_a += dot4(ra, ra);
      ssq_q += dot4(rq, rq);
    } else if (MODE == kModeGram) {
      colsum.x += ra.x; colsum.y += ra.y; colsum.z += ra.z; colsum.w += ra.w;
    } else {
      // rb = g, rq = o:  dnum = (g/H)/den ; dden = -((g/H).o)/den
      const float gdo = group_sum<F4>(dot4(rb, rq));
      const float inv = p.gscale / rden;
      const float dden = -gdo * inv;
      wb = make_float4(rb.x * inv, rb.y * inv, rb.z * inv, rb.w * inv);
      colsum.x += ra.x * dden; colsum.y += ra.y * dden;
      colsum.z += ra.z * dden; colsum.w += ra.w * dden;
      if (MODE == kModeBwdH) {
        colsumb.x += wb.x; colsumb.y += wb.y; colsumb.z += wb.z; colsumb.w += wb.w;
        if (col == 0) ssq_a += dden;   // one lane per row
      }
    }
    *reinterpret_cast<float4*>(&ldsA[(buf * R + srow) * DP + col]) = wa;
    *reinterpret_cast<float4*>(&ldsB[(buf * R + srow) * DP + col]) = wb;
  };

  int64_t tile = blockIdx.x;
  int buf = 0;
  if (tile < ntiles) {
    issue(tile);
    commit(0);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) issue(next);
    const float* A = ldsA + buf * R * DP + 64 * wm + 2 * i31;
    const float* B = ldsB + buf * R * DP + 64 * wd + 2 * i31;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int lrow = 2 * (grp + RG * s) + hi;
      const float2 a2 = *reinterpret_cast<const float2*>(A + lrow * DP);
      const float2 b2 = *reinterpret_cast<const float2*>(B + lrow * DP);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2.y, acc[1][1], 0, 0, 0);
    }
    if (has_next) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- write this block's partial: [grp][m][dd] (DP-padded), then column sums and ssq ----
  float* part = p.partial + (static_cast<int64_t>(head) * gridDim.x + blockIdx.x) * kPartialStride;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 64 * wm + 2 * mfma32_row(r, lane) + tm;
      const int dd = 64 * wd + 2 * i31;
      float2 v = make_float2(acc[tm][0][r], acc[tm][1][r]);
      *reinterpret_cast<float2*>(&part[(grp * DP + m) * DP + dd]) = v;
    }
  // column sums: threads with equal `col` differ in srow -> reduce over R rows through LDS
  __syncthreads();
  *reinterpret_cast<float4*>(&lds[srow * DP + col]) = colsum;
  const float wa = group_sum<64>(ssq_a);
  const float wq = group_sum<64>(ssq_q);
  if (lane == 0) {
    lds[R * DP + wave] = wa;
    lds[R * DP + 16 + wave] = wq;
  }
  __syncthreads();
  if (tid < DP) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += lds[r * DP + tid];
    part[kTileElems + tid] = s;
  }
  if (tid == 0) {
    float sa = 0.f, sq = 0.f;
    for (int w = 0; w < 16; ++w) {
      sa += lds[R * DP + w];
      sq += lds[R * DP + 16 + w];
    }
    part[kTileElems + DP] = sa;
    part[kTileElems + DP + 1] = sq;
  }
  if (MODE == kModeBwdH) {
    __syncthreads();
    *reinterpret_cast<float4*>(&lds[srow * DP + col]) = colsumb;
    __syncthreads();
    if (tid < DP) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += lds[r * DP + tid];
      part[kVecB + tid] = s;
    }
  }
}

// out layout: [ M (heads*d*d) | vec (heads*d) | extra0 | extra1 ]; extras written iff n_extra > 0
// (fwd: ssq_q, ssq_k ; bwd: one zeroed slot that sgf_attn_bwd_apply fills with <S0,dS0>+<z0,dz0>).
__global__ void k_attn_finalize(const float* __restrict__ partial, int nblk, int heads, int d,
                                int DP, int RG, int mode, float* __restrict__ out) {
  const int64_t nmat = static_cast<int64_t>(heads) * d * d;
  const int64_t nvec = static_cast<int64_t>(heads) * d;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx < nmat) {
    const int h = static_cast<int>(idx / (d * d));
    const int m = static_cast<int>((idx / d) % d);
    const int dd = static_cast<int>(idx % d);
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) {
      const float* part = partial + (static_cast<int64_t>(h) * nblk + b) * kPartialStride;
      for (int g = 0; g < RG; ++g) s += part[(g * DP + m) * DP + dd];
    }
    out[idx] = s;
  } else if (idx < nmat + nvec) {
    const int64_t j = idx - nmat;
    const int h = static_cast<int>(j / d);
    const int c = static_cast<int>(j % d);
    float s = 0.f;
    for (int b = 0; b < nblk; ++b)
      s += partial[(static_cast<int64_t>(h) * nblk + b) * kPartialStride + kTileElems + c];
    out[idx] = s;
  } else if (idx < nmat + nvec + 2) {
    const int which = static_cast<int>(idx - nmat - nvec);  // 0 -> ssq_q, 1 -> ssq_k
    if (mode == kModeFwd) {
      float s = 0.f;
      for (int h = 0; h < heads; ++h)
        for (int b = 0; b < nblk; ++b)
          s += partial[(static_cast<int64_t>(h) * nblk + b) * kPartialStride + kTileElems + DP +
                       (which == 0 ? 1 : 0)];
      out[idx] = s;
    } else if (which == 0) {
      out[idx] = 0.f;
    }
  }
}

// sgf_gram: C[mb x kb] block (row stride ldc) and column sums of A from the per-block partials,
// summed in a fixed order.  kFinChains threads per output element (blocks b = q, q + kFinChains, ...; the chains are
// added by a fixed shuffle tree): one thread per element walked 256 partial tiles (64 MiB) in a single dependent chain and
// ran 91 us per call, 11 calls per step; four chains 23 us, sixteen 19 us: what is left is reading nblk x 264 KB of
// partials (64 MB at 256 blocks, 3.5 TB/s), not the chain.
constexpr int kFinChains = 16;
__device__ __forceinline__ float fin_chain_sum(float s) {   // lanes 16 k .. 16 k + 15 hold the chains of one element
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 8, 64);
  return s;
}
__global__ void k_gram_finalize(const float* __restrict__ partial, int nblk, int mb, int kb, int DP,
                                int RG, float* __restrict__ c, int64_t ldc,
                                float* __restrict__ colsum) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t idx = tid / kFinChains;
  const int q = static_cast<int>(tid % kFinChains);
  const int64_t nmat = static_cast<int64_t>(mb) * kb;
  float s = 0.f;
  if (idx < nmat) {
    const int m = static_cast<int>(idx / kb);
    const int dd = static_cast<int>(idx % kb);
    for (int b = q; b < nblk; b += kFinChains) {
      const float* part = partial + static_cast<int64_t>(b) * kPartialStride;
      for (int g = 0; g < RG; ++g) s += part[(g * DP + m) * DP + dd];
    }
  } else if (idx < nmat + mb && colsum != nullptr) {
    const int j = static_cast<int>(idx - nmat);
    for (int b = q; b < nblk; b += kFinChains) s += partial[static_cast<int64_t>(b) * kPartialStride + kTileElems + j];
  }
  const float t = fin_chain_sum(s);
  if (q == 0) {
    if (idx < nmat) {
      const int m = static_cast<int>(idx / kb);
      const int dd = static_cast<int>(idx % kb);
      c[static_cast<int64_t>(m) * ldc + dd] = t;
    } else if (idx < nmat + mb && colsum != nullptr) {
      colsum[idx - nmat] = t;
    }
  }
}

// sgf_attn_h_bwd_reduce: hstats = [ dM (d*d) | dw (d) | dm (d) | dbeta ] from the per-block partials.  Four threads per
// element, each walking every fourth partial, added in the fixed order (s0 + s1) + (s2 + s3) — as k_gram_finalize (r05: one
// thread per element walked all 256 partials in a single dependent chain: 91 us per call).
__global__ void k_hbwd_finalize(const float* __restrict__ partial, int nblk, int d, int DP, int RG,
                                float* __restrict__ out) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t idx = tid / kFinChains;
  const int q = static_cast<int>(tid % kFinChains);
  const int64_t nmat = static_cast<int64_t>(d) * d;
  float s = 0.f;
  if (idx < nmat) {
    const int m = static_cast<int>(idx / d);
    const int dd = static_cast<int>(idx % d);
    for (int b = q; b < nblk; b += kFinChains) {
      const float* part = partial + static_cast<int64_t>(b) * kPartialStride;
      for (int g = 0; g < RG; ++g) s += part[(g * DP + m) * DP + dd];
    }
  } else if (idx < nmat + 2 * d + 1) {
    const int j = static_cast<int>(idx - nmat);
    const int off = j < d ? kTileElems + j : (j < 2 * d ? kVecB + (j - d) : kTileElems + DP);
    for (int b = q; b < nblk; b += kFinChains) s += partial[static_cast<int64_t>(b) * kPartialStride + off];
  }
  const float t = fin_chain_sum(s);
  if (q == 0 && idx < nmat + 2 * d + 1) out[idx] = t;
}

// sdot = <S0,dS0> + <z0,dz0> over all heads: one block, fixed-order tree -> deterministic.
__global__ __launch_bounds__(1024) void k_attn_sdot(const float* __restrict__ stats,
                                                    float* __restrict__ bstats, int64_t len) {
  __shared__ float red[1024];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < len; i += 1024) s += stats[i] * bstats[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) bstats[len] = red[0];
}

template <typename T>
__global__ void k_head_mean(const T* __restrict__ oh, int64_t n, int heads, int d,
                            T* __restrict__ out, int64_t ldo) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n * d) return;
  const int64_t row = idx / d;
  const int c = static_cast<int>(idx % d);
  float s = 0.f;
  for (int h = 0; h < heads; ++h) s += load1<T>(oh + (row * heads + h) * d + c);
  store1<T>(out + row * ldo + c, s / static_cast<float>(heads));
}

// ------------------------------------------------------------------------------------------------
// apply skeleton
// ------------------------------------------------------------------------------------------------
constexpr int kApplyThreads = 1024;

struct ApplyArgs {
  const void* a;    // FWD: Q    DQ: g     DK: V     DV: K
  const void* a2;   // DQ: o
  const void* e;    // FWD: V    DQ: Q     DK: K     DV: g
  void* out;
  int64_t lda, lda2, lde, ldo;
  const float* bmat;   // [d, d] row-major (S0 or dS0 of this head)
  const float* cvec;   // FWD/DQ: z0   DK: dz0   DV: unused   H modes: epilogue vector (m / w / ds)
  const float* dvec;   // HFwd: w (den = h.w + beta)
  const float* beta;   // HFwd: device scalar
  float* den;          // [n, heads] (+head): written by FWD, read by DQ / DV
  const float* stats;  // fwd stats (for ssq_q, ssq_k)
  const float* sdot;   // DQ / DK: pointer to <S0,dS0>+<z0,dz0>
  int64_t stats_len;
  int64_t n;
  int32_t d, heads;
  float ntot, gscale;
  int32_t trans_b;     // B[k][j] = bmat[j*d + k]
  int32_t accumulate;  // out += ...
};

// 16 waves = NS column strips x RS row sub-blocks x 2 K-halves.  A wave keeps the [DP/2 x 32] piece
// of the d x d matrix for its (strip, K-half) in DP/4 registers, so the whole kernel fits the
// 128-VGPR budget of 4 waves/SIMD.  Both K-halves drop their 32x32 accumulator tile into LDS
// (C layout -> row-major), and the epilogue then runs row-wise with the staging thread map: 16 B
// per lane, fully coalesced loads of E and stores of out.
template <typename T, int DP, int MODE>
__global__ __launch_bounds__(kApplyThreads) void k_attn_apply(ApplyArgs p) {
  constexpr int NS = DP / 32;         // 32-column strips
  constexpr int RS = 8 / NS;          // row sub-blocks
  constexpr int RT = 32 * RS;         // rows per tile (32 / 64 / 128)
  constexpr int F4 = DP / 4;
  constexpr int RPP = kApplyThreads / F4;  // rows covered per staging pass (2 passes)
  constexpr int LD = DP + 4;          // LDS row stride (floats)
  constexpr int KSH = DP / 16;        // k-steps of 8 per K-half

  __shared__ float smem[4 * RT * LD + 2 * 3 * RT];
  float* const ldsA = smem;                       // [buf][RT][LD]
  float* const ldsC = smem + 2 * RT * LD;         // [kh][RT][LD]
  float* const ldsR = smem + 4 * RT * LD;         // [buf][ar|br|gr][RT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  const int kh = wave & 1;
  const int ws = (wave >> 1) % NS;
  const int wr = (wave >> 1) / NS;
  const int d = p.d;

  // global scalars (device-side: no host sync anywhere on the path)
  float c = 1.f, gconst = 0.f, hbeta = 0.f;
  if (MODE <= kApplyDV) {
    const float ssq_q = p.stats[p.stats_len - 2];
    const float ssq_k = p.stats[p.stats_len - 1];
    c = 1.0f / (sqrtf(ssq_q) * sqrtf(ssq_k));
    if (MODE == kApplyDQ) gconst = -(c * p.sdot[0]) / ssq_q;
    if (MODE == kApplyDK) gconst = -(c * p.sdot[0]) / ssq_k;
  }
  if (MODE == kApplyHFwd) hbeta = p.beta[0];

  // resident piece of the d x d matrix: breg[4s+t] = B[8(s + kh*KSH) + 4hi + t][32ws + i31]
  float breg[DP / 4];
  {
    const int j = 32 * ws + i31;
#pragma unroll
    for (int s = 0; s < KSH; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 8 * (s + kh * KSH) + 4 * hi + t;
        float v = 0.f;
        if (k < d && j < d) v = p.trans_b ? p.bmat[static_cast<int64_t>(j) * d + k]
                                           : p.bmat[static_cast<int64_t>(k) * d + j];
        breg[4 * s + t] = v;
      }
  }

  // staging / epilogue geometry (same thread map for both)
  const int scol = (tid % F4) * 4;
  const int srow0 = tid / F4;
  const bool scol_ok = scol < d;
  float4 zc = zero4();  // FWD: z0 chunk (den = c Q.z0 + N);  DQ: z0, DK: dz0 chunk (epilogue)
  if (MODE != kApplyDV && scol_ok) zc = *reinterpret_cast<const float4*>(p.cvec + scol);
  float4 zd = zero4();  // HFwd: w chunk (den = h.w + beta)
  if (MODE == kApplyHFwd && scol_ok) zd = *reinterpret_cast<const float4*>(p.dvec + scol);

  const T* pa = static_cast<const T*>(p.a) + scol;
  const T* pa2 = static_cast<const T*>(p.a2) + scol;
  const T* pe = static_cast<const T*>(p.e) + scol;
  T* po = static_cast<T*>(p.out) + scol;

  float4 ra[2], ra2[2];
  float rden[2];

  const int64_t ntiles = (p.n + RT - 1) / RT;

  auto issue = [&](int64_t tile) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = tile * RT + srow0 + i * RPP;
      const bool ok = scol_ok && row < p.n;
      ra[i] = ok ? load4<T>(pa + row * p.lda) : zero4();
      if (MODE == kApplyDQ || MODE == kApplyHBwd1) ra2[i] = ok ? load4<T>(pa2 + row * p.lda2) : zero4();
      if (MODE == kApplyDQ || MODE == kApplyDV || MODE == kApplyHBwd1)
        rden[i] = (row < p.n) ? p.den[row * p.heads] : 1.f;
    }
  };
  auto commit = [&](int buf, int64_t tile) {
    float* rs = ldsR + buf * 3 * RT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lrow = srow0 + i * RPP;
      const int64_t row = tile * RT + lrow;
      float4 w = ra[i];
      float ar = c, br = 0.f, gr = gconst;
      if (MODE == kApplyFwd) {
        const float qz = group_sum<F4>(dot4(ra[i], zc));
        const float den = c * qz + p.ntot;
        ar = c / den;
        gr = p.ntot / den;
        if (scol == 0 && row < p.n) p.den[row * p.heads] = den;
      } else if (MODE == kApplyDQ) {
        const float gdo = group_sum<F4>(dot4(ra[i], ra2[i]));
        const float inv = p.gscale / rden[i];
        w = make_float4(ra[i].x * inv, ra[i].y * inv, ra[i].z * inv, ra[i].w * inv);
        br = c * (-gdo * inv);
      } else if (MODE == kApplyDK) {
        br = c;
      } else if (MODE == kApplyDV) {
        gr = p.ntot * p.gscale / rden[i];
      } else if (MODE == kApplyHFwd) {
        const float den = group_sum<F4>(dot4(ra[i], zd)) + hbeta;
        ar = 1.0f / den;
        br = ar;
        gr = 0.f;
        if (scol == 0 && row < p.n) p.den[row * p.heads] = den;
      } else if (MODE == kApplyHBwd1) {
        const float gdo = group_sum<F4>(dot4(ra[i], ra2[i]));
        const float inv = 1.0f / rden[i];
        w = make_float4(ra[i].x * inv, ra[i].y * inv, ra[i].z * inv, ra[i].w * inv);
        ar = 1.f;
        br = -gdo * inv;
        gr = 0.f;
      } else {  // HBwd2
        ar = 1.f;
        br = 1.f;
        gr = 0.f;
      }
      *reinterpret_cast<float4*>(&ldsA[(buf * RT + lrow) * LD + scol]) = w;
      if (scol == 0) {
        rs[lrow] = ar;
        rs[RT + lrow] = br;
        rs[2 * RT + lrow] = gr;
      }
    }
  };

  int64_t tile = blockIdx.x;
  int buf = 0;
  if (tile < ntiles) {
    issue(tile);
    commit(0, tile);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) issue(next);

    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* A = ldsA + (buf * RT + 32 * wr + i31) * LD + 8 * kh * KSH + 4 * hi;
#pragma unroll
      for (int s = 0; s < KSH; ++s) {
        const float4 a4 = *reinterpret_cast<const float4*>(A + 8 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, breg[4 * s + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, breg[4 * s + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, breg[4 * s + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, breg[4 * s + 3], acc, 0, 0, 0);
      }
      // C layout -> row-major LDS tile of this K-half
      float* C = ldsC + (kh * RT + 32 * wr + 4 * hi) * LD + 32 * ws + i31;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2)) * LD] = acc[r];
    }
    __syncthreads();

    // row-wise epilogue: out = ar*(C0+C1) + br*cvec[j] + gr*E[n][j]
    {
      const float* rs = ldsR + buf * 3 * RT;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int lrow = srow0 + i * RPP;
        const int64_t row = tile * RT + lrow;
        if (scol_ok && row < p.n) {
          const float4 e = MODE <= kApplyDV ? load4<T>(pe + row * p.lde) : zero4();
          const float4 c0 = *reinterpret_cast<const float4*>(&ldsC[lrow * LD + scol]);
          const float4 c1 = *reinterpret_cast<const float4*>(&ldsC[(RT + lrow) * LD + scol]);
          const float ar = rs[lrow], gr = rs[2 * RT + lrow];
          float4 v = make_float4(ar * (c0.x + c1.x) + gr * e.x, ar * (c0.y + c1.y) + gr * e.y,
                                 ar * (c0.z + c1.z) + gr * e.z, ar * (c0.w + c1.w) + gr * e.w);
          if (MODE == kApplyDQ || MODE == kApplyDK || MODE >= kApplyHFwd) {
            const float br = rs[RT + lrow];
            v.x += br * zc.x; v.y += br * zc.y; v.z += br * zc.z; v.w += br * zc.w;
          }
          if (p.accumulate) {
            const float4 o = load4<T>(po + row * p.ldo);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          store4<T>(po + row * p.ldo, v);
        }
      }
    }

    if (has_next) commit(buf ^ 1, next);
    __syncthreads();
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 storage: the same two skeletons on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16,
// 16x the rate of the exact-fp32 MFMA), fp32 accumulation.  A product of two bf16 values is exact
// in fp32, so against the fp32-MFMA kernels above only the summation order differs.
//
//   A operand: lane l holds A[i = l & 31][k = 8 (l >> 5) + 0..7]   (8 bf16 = 4 VGPRs)
//   B operand: lane l holds B[k = 8 (l >> 5) + 0..7][j = l & 31]
//
// k_reduce_bf16 contracts over ROWS (C = A^T B), so both operands need 8 consecutive rows of one
// column per lane.  Tiles are therefore staged TRANSPOSED in LDS, T[col][row]: a thread loads a
// 4-row x 4-column patch (four 8-byte coalesced loads per stream), transposes it in registers and
// writes four 8-byte column segments; an MFMA fragment is then one ds_read_b128.  The 16-byte
// chunk index is XOR-swizzled per column (swz), which makes the fragment reads conflict-free
// (bank = (addr/4) % 64 for b128: the 16 lanes of a service group hit 16 distinct 16-byte slots).
// Any permutation of the rows inside a tile is harmless as long as A and B share it.
// ------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  return static_cast<uint32_t>(f32_to_bf16(lo)) | (static_cast<uint32_t>(f32_to_bf16(hi)) << 16);
}

constexpr int kBfThreads = 512;  // 8 waves = 2 per SIMD: a 256-VGPR budget per wave

// XOR swizzle of the 16-byte chunk index inside an LDS column of the transposed tiles.  With a
// 128-byte column (DP = 256) the ds_read_b128 fragment reads are conflict-free (the 16 lanes of a
// service group land on 16 distinct 16-byte slots of the 256-byte bank row) and the ds_write_b64
// patch stores are 2-way, the minimum for 16 lanes on 8 slots.
__device__ __forceinline__ int swz(int col) { return ((col >> 2) ^ ((col & 2) << 1)) & 7; }

// NW waves per block.  NW = 16 (4 per SIMD, 128 VGPRs): 64x64 wave blocks, one staging pass per
// tile, 64-96 KiB of loads in flight per CU.  NW = 8 (2 per SIMD, 256 VGPRs): 64x128 wave blocks
// (fewer LDS fragment reads per MFMA), two half-tile staging passes.
template <int NW, int DP>
struct ReduceGeom {
  static constexpr int LPQ = DP / 4;             // lanes per patch row (64 / 32 / 16)
  static constexpr int QPW = 64 / LPQ;           // 4-row patches per wave per pass
  static constexpr int NPASS = NW == 16 ? 1 : 2; // staging passes per tile
  static constexpr int R = NW * QPW * 4 * NPASS; // rows per tile (64 / 128 / 256)
  static constexpr int BN = (NW == 8 && DP >= 256) ? 128 : 64;  // a wave owns a 64 x BN result block
  static constexpr int NBM = DP / 64, NBN = DP / BN;
  static constexpr int RG = NW / (NBM * NBN);    // row groups
};

template <int DP, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_reduce_bf16(ReduceArgs p) {
  using G = ReduceGeom<NW, DP>;
  constexpr int LPQ = G::LPQ, QPW = G::QPW, NPASS = G::NPASS, R = G::R, BN = G::BN;
  constexpr int TN = BN / 32;
  constexpr int NBM = G::NBM, NBN = G::NBN, RG = G::RG;
  constexpr int KSTEPS = R / 16;          // MFMA k-steps (16 rows) per tile
  constexpr int SPG = KSTEPS / RG;        // k-steps per row group per tile
  constexpr int CSB = R * 2;              // bytes per LDS column
  constexpr int OPB = DP * CSB;           // bytes per operand tile (32 KiB)
  static_assert(SPG >= 1 && KSTEPS % RG == 0, "tile / row-group mismatch");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * OPB];  // [buf][A|B] = 128 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = DP == 64 ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: `act` below is a scalar branch)
  const int head = blockIdx.y;
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  const int blk = wave % (NBM * NBN);
  const int grp = wave / (NBM * NBN);
  const int wm = blk / NBN;
  const int wd = blk % NBN;

  // staging map: in pass t this thread owns rows 4q..4q+3 (q = q0 + t * NW * QPW), columns c0..c0+3
  const int q0 = wave * QPW + lane / LPQ;
  const int c0 = (lane % LPQ) * 4;
  const bool a_ok = c0 < p.d;
  const bool b_ok = c0 < p.db;

  // paired Gram launch: role 1 multiplies with b2; both roles of a pair walk the same tiles
  const bool paired = MODE == kModeGram && p.pair != 0;
  const int role = paired ? (blockIdx.x >> 3) & 1 : 0;
  const int64_t vblock = paired ? (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3) : blockIdx.x;
  const int64_t vgrid = paired ? gridDim.x / 2 : gridDim.x;
  if (role) {
    p.b = p.b2;
    p.ldb = p.ldb2;
  }
  const uint16_t* pa = static_cast<const uint16_t*>(p.a) + static_cast<int64_t>(head) * p.d + c0;
  const uint16_t* pb = static_cast<const uint16_t*>(p.b) +
                       (p.b_heads == 1 ? 0 : static_cast<int64_t>(head) * p.d) + c0;
  const uint16_t* pq = static_cast<const uint16_t*>(p.q) + static_cast<int64_t>(head) * p.d + c0;

  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 colsum = zero4(), colsumb = zero4();   // colsumb: bwdH only (sum dnum)
  float4 colsumc = zero4();                     // kModeGramLN: sum g' xhat (d gamma)
  float ssq_a = 0.f, ssq_q = 0.f;               // bwdH: ssq_a = sum dden

  const int64_t ntiles = (p.n + R - 1) / R;
  // Only ONE staging pass is in flight at a time (24 VGPRs of loads): a tile iteration is NPASS
  // half-steps, each = issue the loads of pass t of the next tile, run the MFMAs of half of the
  // current tile's k-steps, then transpose / commit pass t into the other LDS buffer.
  uint2 ra[4], rb[4], rq[4], r2[4];
  float rden[4], rden2[4];
  const uint16_t* pa2 = MODE == kModeGramBN && p.a2 ? static_cast<const uint16_t*>(p.a2) + c0 : nullptr;
  // kModeGramBN: this thread's four columns keep their BatchNorm coefficients in registers for the whole kernel
  float bmu[4], brs[4], bga[4], bbe[4], bk0[4], bk1[4];
  if (MODE == kModeGramLN) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j;
      const bool ok = c < p.d;
      bga[j] = ok ? (p.bn_gamma ? p.bn_gamma[c] : 1.f) : 0.f;
      bbe[j] = ok ? (p.bn_beta ? p.bn_beta[c] : 0.f) : 0.f;
      bmu[j] = brs[j] = bk0[j] = bk1[j] = 0.f;
    }
  }
  if (MODE == kModeGramBN) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j;
      const bool ok = c < p.d;
      bmu[j] = ok ? p.bn_mean[c] : 0.f;
      brs[j] = ok ? p.bn_rstd[c] : 0.f;
      bga[j] = ok ? (p.bn_gamma ? p.bn_gamma[c] : 1.f) : 0.f;
      bbe[j] = ok ? (p.bn_beta ? p.bn_beta[c] : 0.f) : 0.f;
      bk0[j] = (ok && p.bn_training) ? p.bn_stats[c] * p.bn_inv_n : 0.f;
      bk1[j] = (ok && p.bn_training) ? p.bn_stats[p.d + c] * p.bn_inv_n : 0.f;
    }
  }

  auto issue = [&](int64_t tile, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = tile * R + 4 * (q0 + t * NW * QPW) + i;
      const bool rok = row < p.n;
      ra[i] = (rok && a_ok) ? *reinterpret_cast<const uint2*>(pa + row * p.lda) : make_uint2(0u, 0u);
      rb[i] = (rok && b_ok) ? *reinterpret_cast<const uint2*>(pb + row * p.ldb) : make_uint2(0u, 0u);
      if (MODE != kModeGram && MODE != kModeBwdHS)
        rq[i] = (rok && a_ok) ? *reinterpret_cast<const uint2*>(pq + row * p.ldq) : make_uint2(0u, 0u);
      if (MODE == kModeGramBN) {
        r2[i] = (rok && a_ok && p.a2) ? *reinterpret_cast<const uint2*>(pa2 + row * p.lda2) : make_uint2(0u, 0u);
        rden[i] = rok ? 1.f : 0.f;                     // rows past the end contribute nothing (their dz is not zero by itself)
      }
      if (MODE == kModeGramLN) {
        int64_t rr = row < p.n ? row : p.n - 1;         // unconditional loads (rows past the end carry g = 0: they add nothing)
        if (LPQ == 64)                                  // one row per wave: a scalar load, the statistics live in SGPRs
          rr = (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rr >> 32))) << 32) |
               static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rr)));
        rden[i] = p.bn_mean[rr];                       // the row's mean / rstd as the forward left them
        rden2[i] = p.bn_rstd[rr];
      }
      if (MODE == kModeBwd || MODE == kModeBwdH) rden[i] = rok ? p.den[row * p.heads + head] : 1.f;
      if (MODE == kModeBwdHS) {
        const float2 rs = rok ? reinterpret_cast<const float2*>(p.den)[row] : make_float2(0.f, 0.f);
        rden[i] = rs.x;      // 1 / den
        rden2[i] = rs.y;     // dden = -(g.o) / den
      }
    }
  };
  // column j of a 4x4 patch as 8 bytes: rows 0..3
  auto column = [](const uint2 (&r)[4], int j) -> uint2 {
    uint32_t x0, x1, x2, x3;
    if (j < 2) { x0 = r[0].x; x1 = r[1].x; x2 = r[2].x; x3 = r[3].x; }
    else       { x0 = r[0].y; x1 = r[1].y; x2 = r[2].y; x3 = r[3].y; }
    if (j & 1) return make_uint2((x0 >> 16) | (x1 & 0xffff0000u), (x2 >> 16) | (x3 & 0xffff0000u));
    return make_uint2((x0 & 0xffffu) | (x1 << 16), (x2 & 0xffffu) | (x3 << 16));
  };
  auto commit = [&](int buf, int t) {
    unsigned char* ta = lds + buf * 2 * OPB;
    unsigned char* tb = ta + OPB;
    if (MODE == kModeFwd) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint2 a = ra[i], qq = rq[i];
        const float a0 = bf_lo(a.x), a1 = bf_hi(a.x), a2 = bf_lo(a.y), a3 = bf_hi(a.y);
        const float q0f = bf_lo(qq.x), q1 = bf_hi(qq.x), q2 = bf_lo(qq.y), q3 = bf_hi(qq.y);
        colsum.x += a0; colsum.y += a1; colsum.z += a2; colsum.w += a3;
        ssq_a += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        ssq_q += q0f * q0f + q1 * q1 + q2 * q2 + q3 * q3;
      }
    } else if (MODE == kModeGram) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        colsum.x += bf_lo(ra[i].x); colsum.y += bf_hi(ra[i].x);
        colsum.z += bf_lo(ra[i].y); colsum.w += bf_hi(ra[i].y);
      }
    } else if (MODE == kModeGramBN) {
      // ra = g1, r2 = g2, rq = z  ->  ra = dz (sgf_bn_bwd_apply's arithmetic; the two gradients are added in fp32),
      // rounded to bf16 once — the tensor sgf_bn_bwd_apply would have written; colsum = sum dz (the bias gradient)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g[4] = {bf_lo(ra[i].x) + bf_lo(r2[i].x), bf_hi(ra[i].x) + bf_hi(r2[i].x),
                            bf_lo(ra[i].y) + bf_lo(r2[i].y), bf_hi(ra[i].y) + bf_hi(r2[i].y)};
        const float zv[4] = {bf_lo(rq[i].x), bf_hi(rq[i].x), bf_lo(rq[i].y), bf_hi(rq[i].y)};
        float dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (zv[j] - bmu[j]) * brs[j];
          float gg = g[j];
          if (p.bn_relu) gg = (xh * bga[j] + bbe[j]) > 0.f ? gg : 0.f;
          gg -= bk0[j] + xh * bk1[j];
          dv[j] = rden[i] != 0.f ? bga[j] * brs[j] * gg : 0.f;
        }
        ra[i] = make_uint2(pack_bf16(dv[0], dv[1]), pack_bf16(dv[2], dv[3]));
        colsum.x += bf_lo(ra[i].x); colsum.y += bf_hi(ra[i].x);
        colsum.z += bf_lo(ra[i].y); colsum.w += bf_hi(ra[i].y);
      }
    } else if (MODE == kModeGramLN) {
      // ra = g (gradient of the LayerNorm's output, behind the activation), rq = the LayerNorm's input  ->  ra = its input
      // gradient (sgf_ln_bwd's arithmetic: row means over the d columns, which one patch row of LPQ lanes holds)
      const float inv_d = 1.0f / static_cast<float>(p.d);
      // three phases so that only the packed operands and eight partial sums live across the lane reductions (the one-phase
      // form kept xhat and dxhat of all four rows in registers and spilled at d = 256): sums, reductions, then xhat / g' again
      float s1[4], s2[4];
      auto masked = [&](int i, int j, float xh) -> float {
        float gm = j == 0 ? bf_lo(ra[i].x) : j == 1 ? bf_hi(ra[i].x) : j == 2 ? bf_lo(ra[i].y) : bf_hi(ra[i].y);
        if (p.bn_relu) gm = fmaf(xh, bga[j], bbe[j]) > 0.f ? gm : 0.f;
        return (c0 + j < p.d) ? gm : 0.f;
      };
      auto xhat = [&](int i, int j) -> float {
        const float xv = j == 0 ? bf_lo(rq[i].x) : j == 1 ? bf_hi(rq[i].x) : j == 2 ? bf_lo(rq[i].y) : bf_hi(rq[i].y);
        return (xv - rden[i]) * rden2[i];
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s1[i] = s2[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = xhat(i, j), gm = masked(i, j, xh), dxh = gm * bga[j];
          s1[i] += dxh;
          s2[i] = fmaf(dxh, xh, s2[i]);
          (&colsumb.x)[j] += gm;
          (&colsumc.x)[j] = fmaf(gm, xh, (&colsumc.x)[j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s1[i] = (LPQ == 64 ? wave_sum_uniform(s1[i]) : group_sum<LPQ>(s1[i])) * inv_d;
        s2[i] = (LPQ == 64 ? wave_sum_uniform(s2[i]) : group_sum<LPQ>(s2[i])) * inv_d;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = xhat(i, j), dxh = masked(i, j, xh) * bga[j];
          dv[j] = (c0 + j < p.d) ? rden2[i] * (dxh - s1[i] - xh * s2[i]) : 0.f;
        }
        ra[i] = make_uint2(pack_bf16(dv[0], dv[1]), pack_bf16(dv[2], dv[3]));
        colsum.x += bf_lo(ra[i].x); colsum.y += bf_hi(ra[i].x);
        colsum.z += bf_lo(ra[i].y); colsum.w += bf_hi(ra[i].y);
      }
    } else if (MODE == kModeBwdHS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint2 g = rb[i];
        const float inv = rden[i], dden = rden2[i];
        const float g0 = bf_lo(g.x) * inv, g1 = bf_hi(g.x) * inv, g2 = bf_lo(g.y) * inv, g3 = bf_hi(g.y) * inv;
        rb[i] = make_uint2(pack_bf16(g0, g1), pack_bf16(g2, g3));
        colsum.x += bf_lo(ra[i].x) * dden; colsum.y += bf_hi(ra[i].x) * dden;
        colsum.z += bf_lo(ra[i].y) * dden; colsum.w += bf_hi(ra[i].y) * dden;
        colsumb.x += g0; colsumb.y += g1; colsumb.z += g2; colsumb.w += g3;
        if (c0 == 0) ssq_a += dden;
      }
    } else {
      // rb = g, rq = o: dnum = (g/H)/den (re-rounded to bf16 for the MFMA); dden = -((g/H).o)/den
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint2 g = rb[i], o = rq[i];
        const float g0 = bf_lo(g.x), g1 = bf_hi(g.x), g2 = bf_lo(g.y), g3 = bf_hi(g.y);
        const float gdo = group_sum<LPQ>(g0 * bf_lo(o.x) + g1 * bf_hi(o.x) + g2 * bf_lo(o.y) +
                                         g3 * bf_hi(o.y));
        const float inv = p.gscale / rden[i];
        const float dden = -gdo * inv;
        rb[i] = make_uint2(pack_bf16(g0 * inv, g1 * inv), pack_bf16(g2 * inv, g3 * inv));
        colsum.x += bf_lo(ra[i].x) * dden; colsum.y += bf_hi(ra[i].x) * dden;
        colsum.z += bf_lo(ra[i].y) * dden; colsum.w += bf_hi(ra[i].y) * dden;
        if (MODE == kModeBwdH) {   // sums of the un-rounded dnum and of dden (one lane per row)
          colsumb.x += g0 * inv; colsumb.y += g1 * inv; colsumb.z += g2 * inv; colsumb.w += g3 * inv;
          if (c0 == 0) ssq_a += dden;
        }
      }
    }
    const int q = q0 + t * NW * QPW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + j;
      const int off = col * CSB + (((q >> 1) ^ swz(col)) << 4) + ((q & 1) << 3);
      *reinterpret_cast<uint2*>(ta + off) = column(ra, j);
      *reinterpret_cast<uint2*>(tb + off) = column(rb, j);
    }
  };

  static_assert(SPG % NPASS == 0, "k-steps must split evenly over the staging passes");
  int64_t tile = vblock;
  int buf = 0;
  if (tile < ntiles) {
#pragma unroll
    for (int t = 0; t < NPASS; ++t) {
      issue(tile, t);
      commit(0, t);
    }
  }
  __syncthreads();
  const int ca0 = 64 * wm + i31, ca1 = ca0 + 32;
  const int cb0 = BN * wd + i31;
  // a wave whose 64 x BN result block lies outside the m x k product (narrow operands: the head's dW with m = 48, the stems'
  // with k = 100) multiplies zeros: it only stages.  Wave-uniform, so the whole MFMA phase and its LDS reads are skipped.
  const bool act = DP == 64 || (64 * wm < p.d && BN * wd < p.db);     // (one block at DP = 64: always inside)
  for (; tile < ntiles; tile += vgrid) {
    const int64_t next = tile + vgrid;
    const bool has_next = next < ntiles;
    const unsigned char* ta = lds + buf * 2 * OPB;
    const unsigned char* tb = ta + OPB;
#pragma unroll
    for (int t = 0; t < NPASS; ++t) {
      if (has_next) issue(next, t);
#pragma unroll 1
      for (int s = t * (SPG / NPASS); act && s < (t + 1) * (SPG / NPASS); ++s) {
        const int chunk = 2 * (grp + RG * s) + hi;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ta + ca0 * CSB + ((chunk ^ swz(ca0)) << 4));
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ta + ca1 * CSB + ((chunk ^ swz(ca1)) << 4));
        bf16x8 bfr[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int cb = cb0 + 32 * tn;
          bfr[tn] = *reinterpret_cast<const bf16x8*>(tb + cb * CSB + ((chunk ^ swz(cb)) << 4));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr[tn], acc[0][tn], 0, 0, 0);
          acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfr[tn], acc[1][tn], 0, 0, 0);
        }
      }
      if (has_next) commit(buf ^ 1, t);
    }
    __syncthreads();
    buf ^= 1;
  }

  // ---- this block's partial, same layout as k_attn_reduce: [grp][m][dd] | colsum[DP] | ssq ----
  float* part = p.partial + ((static_cast<int64_t>(head) + role) * vgrid + vblock) * kPartialStride;
  if (act)                      // (the finalize kernels read the m x k part only)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 64 * wm + 32 * tm + mfma32_row(r, lane);
        const int dd = BN * wd + 32 * tn + i31;
        part[(grp * DP + m) * DP + dd] = acc[tm][tn][r];
      }
  // column sums: threads with equal c0 differ in patch row -> reduce over the 8*QPW slots via LDS
  __syncthreads();
  float* fl = reinterpret_cast<float*>(lds);
  constexpr int SLOTS = NW * QPW;
  *reinterpret_cast<float4*>(&fl[q0 * DP + c0]) = colsum;
  const float wa = group_sum<64>(ssq_a);
  const float wq = group_sum<64>(ssq_q);
  if (lane == 0) {
    fl[SLOTS * DP + wave] = wa;
    fl[SLOTS * DP + NW + wave] = wq;
  }
  __syncthreads();
  if (tid < DP) {
    float s = 0.f;
    for (int r = 0; r < SLOTS; ++r) s += fl[r * DP + tid];
    part[kTileElems + tid] = s;
  }
  if (tid == 0) {
    float sa = 0.f, sq = 0.f;
    for (int w = 0; w < NW; ++w) {
      sa += fl[SLOTS * DP + w];
      sq += fl[SLOTS * DP + NW + w];
    }
    part[kTileElems + DP] = sa;
    part[kTileElems + DP + 1] = sq;
  }
  if (MODE == kModeBwdH || MODE == kModeBwdHS || MODE == kModeGramLN) {
    __syncthreads();
    *reinterpret_cast<float4*>(&fl[q0 * DP + c0]) = colsumb;
    __syncthreads();
    if (tid < DP) {
      float s = 0.f;
      for (int r = 0; r < SLOTS; ++r) s += fl[r * DP + tid];
      part[kVecB + tid] = s;
    }
  }
  if (MODE == kModeGramLN) {
    __syncthreads();
    *reinterpret_cast<float4*>(&fl[q0 * DP + c0]) = colsumc;
    __syncthreads();
    if (tid < DP) {
      float s = 0.f;
      for (int r = 0; r < SLOTS; ++r) s += fl[r * DP + tid];
      part[kVecC + tid] = s;
    }
  }
}

// out[j] = sum over the blocks' partials of one of their column-sum vectors (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_vec_finalize(const float* __restrict__ partial, int nblk, int off, int len,
                                                      float* __restrict__ out) {
  // 32 columns x 8 groups of blocks per workgroup: eight independent chains of loads per column instead of one
  __shared__ float part[8][32];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float s = 0.f;
  if (j < len)
    for (int b = g; b < nblk; b += 8) s += partial[static_cast<int64_t>(b) * kPartialStride + off + j];
  part[g][c] = s;
  __syncthreads();
  if (g == 0 && j < len) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += part[i][c];
    out[j] = t;
  }
}

// k_apply_bf16: out[n x d] = ar[n] (A[n x d] B[d x d]) + br[n] cvec + gr[n] E   on bf16 MFMA.
// 8 waves = NS 32-column strips x RGW row groups; a wave keeps its [DP x 32] strip of B (rounded
// to bf16) in DP/4 registers for the whole kernel and applies it to two 32-row sub-blocks per
// tile.  A tiles are staged row-major (stride DP + 8 bf16: conflict-free ds_read_b128 fragments);
// the accumulators go through an fp32 LDS tile so that E is loaded and `out` stored row-wise,
// 8 bytes per lane, fully coalesced.
template <int DP, int MODE, int RB>
__global__ __launch_bounds__(kBfThreads, RB == 1 ? 4 : 2) void k_apply_bf16(ApplyArgs p) {
  constexpr int NS = DP / 32;             // strips (8 / 4 / 2)
  constexpr int RGW = 8 / NS;             // row groups of waves (1 / 2 / 4)
  constexpr int RT = 32 * RB * RGW;       // rows per tile (RB = 2: 64 / 128 / 256)
  constexpr int F4 = DP / 4;              // lanes per row in the staging / epilogue map
  constexpr int RPP = kBfThreads / F4;    // rows per staging pass (8 / 16 / 32)
  constexpr int NP = RT / RPP;            // staging passes (8)
  constexpr int LDA = DP + 8;             // bf16 elements per LDS row of A
  constexpr int LDC = DP + 4;             // floats per LDS row of C
  constexpr int KS = DP / 16;             // k-steps

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RT * LDA * 2 + RT * LDC * 4 + 2 * 3 * RT * 4];
  uint16_t* const ldsA = reinterpret_cast<uint16_t*>(smem);                       // [buf][RT][LDA]
  float* const ldsC = reinterpret_cast<float*>(smem + 2 * RT * LDA * 2);          // [RT][LDC]
  float* const ldsR = ldsC + RT * LDC;                                            // [buf][3][RT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  const int ws = wave % NS;
  const int wr = wave / NS;
  const int d = p.d;

  float c = 1.f, gconst = 0.f, hbeta = 0.f;
  if (MODE <= kApplyDV) {
    const float ssq_q = p.stats[p.stats_len - 2];
    const float ssq_k = p.stats[p.stats_len - 1];
    c = 1.0f / (sqrtf(ssq_q) * sqrtf(ssq_k));
    if (MODE == kApplyDQ) gconst = -(c * p.sdot[0]) / ssq_q;
    if (MODE == kApplyDK) gconst = -(c * p.sdot[0]) / ssq_k;
  }
  if (MODE == kApplyHFwd) hbeta = p.beta[0];

  // resident strip of B: breg[s][t] = B[16 s + 8 hi + t][32 ws + i31]
  bf16x8 breg[KS];
  {
    const int j = 32 * ws + i31;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = 16 * s + 8 * hi + t;
        float v = 0.f;
        if (k < d && j < d) v = p.trans_b ? p.bmat[static_cast<int64_t>(j) * d + k]
                                           : p.bmat[static_cast<int64_t>(k) * d + j];
        breg[s][t] = static_cast<short>(f32_to_bf16(v));
      }
  }

  const int scol = (tid % F4) * 4;
  const int srow0 = tid / F4;
  const bool scol_ok = scol < d;
  float4 zc = zero4();
  if (MODE != kApplyDV && scol_ok) zc = *reinterpret_cast<const float4*>(p.cvec + scol);
  float4 zd = zero4();  // HFwd: w chunk (den = h.w + beta)
  if (MODE == kApplyHFwd && scol_ok) zd = *reinterpret_cast<const float4*>(p.dvec + scol);

  const uint16_t* pa = static_cast<const uint16_t*>(p.a) + scol;
  const uint16_t* pa2 = static_cast<const uint16_t*>(p.a2) + scol;
  const uint16_t* pe = static_cast<const uint16_t*>(p.e) + scol;
  uint16_t* po = static_cast<uint16_t*>(p.out) + scol;

  uint2 ra[NP], ra2[NP];
  float rden[NP];
  const int64_t ntiles = (p.n + RT - 1) / RT;

  auto issue = [&](int64_t tile) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t row = tile * RT + srow0 + i * RPP;
      const bool ok = scol_ok && row < p.n;
      ra[i] = ok ? *reinterpret_cast<const uint2*>(pa + row * p.lda) : make_uint2(0u, 0u);
      if (MODE == kApplyDQ || MODE == kApplyHBwd1)
        ra2[i] = ok ? *reinterpret_cast<const uint2*>(pa2 + row * p.lda2) : make_uint2(0u, 0u);
      if (MODE == kApplyDQ || MODE == kApplyDV || MODE == kApplyHBwd1)
        rden[i] = (row < p.n) ? p.den[row * p.heads] : 1.f;
    }
  };
  auto commit = [&](int buf, int64_t tile) {
    float* rs = ldsR + buf * 3 * RT;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int lrow = srow0 + i * RPP;
      const int64_t row = tile * RT + lrow;
      uint2 w = ra[i];
      float ar = c, br = 0.f, gr = gconst;
      if (MODE == kApplyFwd) {
        const float qz = group_sum<F4>(bf_lo(w.x) * zc.x + bf_hi(w.x) * zc.y + bf_lo(w.y) * zc.z +
                                       bf_hi(w.y) * zc.w);
        const float den = c * qz + p.ntot;
        ar = c / den;
        gr = p.ntot / den;
        if (scol == 0 && row < p.n) p.den[row * p.heads] = den;
      } else if (MODE == kApplyDQ) {
        const float g0 = bf_lo(w.x), g1 = bf_hi(w.x), g2 = bf_lo(w.y), g3 = bf_hi(w.y);
        const float gdo = group_sum<F4>(g0 * bf_lo(ra2[i].x) + g1 * bf_hi(ra2[i].x) +
                                        g2 * bf_lo(ra2[i].y) + g3 * bf_hi(ra2[i].y));
        const float inv = p.gscale / rden[i];
        w = make_uint2(pack_bf16(g0 * inv, g1 * inv), pack_bf16(g2 * inv, g3 * inv));
        br = c * (-gdo * inv);
      } else if (MODE == kApplyDK) {
        br = c;
      } else if (MODE == kApplyDV) {
        gr = p.ntot * p.gscale / rden[i];
      } else if (MODE == kApplyHFwd) {
        const float hw = group_sum<F4>(bf_lo(w.x) * zd.x + bf_hi(w.x) * zd.y + bf_lo(w.y) * zd.z +
                                       bf_hi(w.y) * zd.w);
        const float den = hw + hbeta;
        ar = 1.0f / den;
        br = ar;
        gr = 0.f;
        if (scol == 0 && row < p.n) p.den[row * p.heads] = den;
      } else if (MODE == kApplyHBwd1) {
        const float g0 = bf_lo(w.x), g1 = bf_hi(w.x), g2 = bf_lo(w.y), g3 = bf_hi(w.y);
        const float gdo = group_sum<F4>(g0 * bf_lo(ra2[i].x) + g1 * bf_hi(ra2[i].x) +
                                        g2 * bf_lo(ra2[i].y) + g3 * bf_hi(ra2[i].y));
        const float inv = 1.0f / rden[i];
        w = make_uint2(pack_bf16(g0 * inv, g1 * inv), pack_bf16(g2 * inv, g3 * inv));
        ar = 1.f;
        br = -gdo * inv;
        gr = 0.f;
      } else {  // HBwd2
        ar = 1.f;
        br = 1.f;
        gr = 0.f;
      }
      *reinterpret_cast<uint2*>(&ldsA[(buf * RT + lrow) * LDA + scol]) = w;
      if (scol == 0) {
        rs[lrow] = ar;
        rs[RT + lrow] = br;
        rs[2 * RT + lrow] = gr;
      }
    }
  };

  int64_t tile = blockIdx.x;
  int buf = 0;
  if (tile < ntiles) {
    issue(tile);
    commit(0, tile);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) issue(next);
    // E rows of THIS tile: issued before the MFMA phase, consumed after it (latency hidden)
    uint2 re[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t row = tile * RT + srow0 + i * RPP;
      re[i] = (MODE <= kApplyDV && scol_ok && row < p.n) ? *reinterpret_cast<const uint2*>(pe + row * p.lde)
                                                          : make_uint2(0u, 0u);
    }
    {
      f32x16 acc[RB];
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      const uint16_t* A0 = ldsA + (buf * RT + 32 * RB * wr + i31) * LDA + 8 * hi;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const bf16x8 xa = *reinterpret_cast<const bf16x8*>(A0 + b * 32 * LDA + 16 * s);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, breg[s], acc[b], 0, 0, 0);
        }
      }
      float* C0 = ldsC + (32 * RB * wr + 4 * hi) * LDC + 32 * ws + i31;
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) C0[(b * 32 + (r & 3) + 8 * (r >> 2)) * LDC] = acc[b][r];
    }
    __syncthreads();
    {
      const float* rs = ldsR + buf * 3 * RT;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int lrow = srow0 + i * RPP;
        const int64_t row = tile * RT + lrow;
        if (scol_ok && row < p.n) {
          const float4 e = make_float4(bf_lo(re[i].x), bf_hi(re[i].x), bf_lo(re[i].y), bf_hi(re[i].y));
          const float4 c0 = *reinterpret_cast<const float4*>(&ldsC[lrow * LDC + scol]);
          const float ar = rs[lrow], gr = rs[2 * RT + lrow];
          float4 v = make_float4(ar * c0.x + gr * e.x, ar * c0.y + gr * e.y, ar * c0.z + gr * e.z,
                                 ar * c0.w + gr * e.w);
          if (MODE == kApplyDQ || MODE == kApplyDK || MODE >= kApplyHFwd) {
            const float br = rs[RT + lrow];
            v.x += br * zc.x; v.y += br * zc.y; v.z += br * zc.z; v.w += br * zc.w;
          }
          if (p.accumulate) {
            const float4 o = load4<uint16_t>(po + row * p.ldo);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          store4<uint16_t>(po + row * p.ldo, v);
        }
      }
    }
    if (has_next) commit(buf ^ 1, next);
    __syncthreads();
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// NW of k_reduce_bf16 per mode: the two-stream Gram mode fits 16 waves x 128 VGPRs without spilling
// (twice the loads in flight); the three-stream attention modes need the 8-wave / 256-VGPR shape.
// (at 16 waves BwdH spills 160 B and runs 2.7x slower, its two-stream form BwdHS spills 84 B: 1.75 ms against 1.10 at 8)
constexpr int bf_reduce_waves(int mode) { return mode == kModeGram ? 16 : 8; }

template <typename T, int MODE>
inline int reduce_rows_per_tile(int DP) {
  // fp32 storage: k_attn_reduce, 1024 / (DP/4) rows;  bf16 storage: ReduceGeom::R
  if (sizeof(T) == 4) return kRedThreads / (DP / 4);
  constexpr int NW = bf_reduce_waves(MODE);
  return DP == 64 ? ReduceGeom<NW, 64>::R : (DP == 128 ? ReduceGeom<NW, 128>::R : ReduceGeom<NW, 256>::R);
}

template <typename T, int MODE>
inline int reduce_row_groups(int DP) {
  // row groups whose partial results the finalize kernels add up (must match the kernels' RG)
  if (sizeof(T) == 4) return 16 / ((DP / 64) * (DP / 64));       // k_attn_reduce: 16 waves, 64x64 blocks
  constexpr int NW = bf_reduce_waves(MODE);
  return DP == 64 ? ReduceGeom<NW, 64>::RG : (DP == 128 ? ReduceGeom<NW, 128>::RG : ReduceGeom<NW, 256>::RG);
}

template <typename T, int MODE>
int launch_reduce(const ReduceArgs& args, int DP, int nblk, hipStream_t st) {
  const dim3 grid(nblk, args.heads), block(sizeof(T) == 2 ? bf_reduce_waves(MODE) * 64 : kRedThreads);
  if (sizeof(T) == 2) {
    switch (DP) {
      case 64: hipLaunchKernelGGL((k_reduce_bf16<64, MODE, bf_reduce_waves(MODE)>), grid, block, 0, st, args); break;
      case 128: hipLaunchKernelGGL((k_reduce_bf16<128, MODE, bf_reduce_waves(MODE)>), grid, block, 0, st, args); break;
      default: hipLaunchKernelGGL((k_reduce_bf16<256, MODE, bf_reduce_waves(MODE)>), grid, block, 0, st, args); break;
    }
  } else {
    switch (DP) {
      case 64: hipLaunchKernelGGL((k_attn_reduce<float, 64, MODE>), grid, block, 0, st, args); break;
      case 128: hipLaunchKernelGGL((k_attn_reduce<float, 128, MODE>), grid, block, 0, st, args); break;
      default: hipLaunchKernelGGL((k_attn_reduce<float, 256, MODE>), grid, block, 0, st, args); break;
    }
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// 32-row blocks per wave in k_apply_bf16: 1 -> ~68 KiB LDS, 128 VGPRs: TWO independent blocks per CU
// whose MFMA / epilogue phases interleave; 2 -> one 135 KiB block per CU with twice the tile.
constexpr int kApplyRB = 2;  // RB = 1 spills at DP = 256 (the resident B strip alone is 64 VGPRs of the 128)

template <typename T, int MODE>
int launch_apply(const ApplyArgs& args, int DP, hipStream_t st) {
  // rows per tile: k_attn_apply 32 * (8 / (DP/32));  k_apply_bf16 32 * kApplyRB * (8 / (DP/32))
  const int RT = (sizeof(T) == 4 ? 32 : 32 * kApplyRB) * (8 / (DP / 32));
  int64_t ntiles = (args.n + RT - 1) / RT;
  const int64_t maxblk = (sizeof(T) == 2 && kApplyRB == 1) ? 2 * kMaxBlocks : kMaxBlocks;
  const int nblk = static_cast<int>(ntiles < maxblk ? ntiles : maxblk);
  const dim3 grid(nblk), block(sizeof(T) == 2 ? kBfThreads : kApplyThreads);
  if (sizeof(T) == 2) {
    switch (DP) {
      case 64: hipLaunchKernelGGL((k_apply_bf16<64, MODE, kApplyRB>), grid, block, 0, st, args); break;
      case 128: hipLaunchKernelGGL((k_apply_bf16<128, MODE, kApplyRB>), grid, block, 0, st, args); break;
      default: hipLaunchKernelGGL((k_apply_bf16<256, MODE, kApplyRB>), grid, block, 0, st, args); break;
    }
  } else {
    switch (DP) {
      case 64: hipLaunchKernelGGL((k_attn_apply<float, 64, MODE>), grid, block, 0, st, args); break;
      case 128: hipLaunchKernelGGL((k_attn_apply<float, 128, MODE>), grid, block, 0, st, args); break;
      default: hipLaunchKernelGGL((k_attn_apply<float, 256, MODE>), grid, block, 0, st, args); break;
    }
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

int check_common(const char* fn, int64_t n, int heads, int d, int dtype) {
  SGF_REQUIRE(n >= 0 && heads >= 1 && d >= 1, SGF_E_INVALID, "%s: bad sizes n=%lld H=%d d=%d", fn,
              static_cast<long long>(n), heads, d);
  SGF_REQUIRE(d % 4 == 0 && d <= 256, SGF_E_UNSUPPORTED,
              "%s: head dim d=%d unsupported (need d %% 4 == 0 and d <= 256)", fn, d);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "%s: unknown dtype %d", fn,
              dtype);
  return SGF_OK;
}

template <typename T>
bool aligned4(const void* p, int64_t ld) {
  return reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T)) == 0 && ld % 4 == 0;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

namespace {
template <typename T>
int gram_t(const void* a, int64_t lda, int m, const void* b, int64_t ldb, int k, int64_t n, float* c,
           int64_t ldc, float* colsum_a, void* ws, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(a, lda) && aligned4<T>(b, ldb), SGF_E_INVALID,
              "sgf_gram: a / b must be 4-element aligned with ld %% 4 == 0");
  for (int mi = 0; mi < m; mi += 256) {
    const int mb = m - mi < 256 ? m - mi : 256;
    for (int ki = 0; ki < k; ki += 256) {
      const int kb = k - ki < 256 ? k - ki : 256;
      const T* ap = static_cast<const T*>(a) + mi;
      const T* bp = static_cast<const T*>(b) + ki;
      if (sizeof(T) == 2 && gramx_supported(ap, lda, mb, bp, ldb, kb, n)) {   // tiles by LDS-DMA (csrc/gramx.hip)
        int nblk = 0;
        int rc = gramx_gram(ap, lda, mb, bp, ldb, nullptr, 0, kb, n, static_cast<float*>(ws), &nblk, st);
        if (rc != SGF_OK) return rc;
        const int64_t len = static_cast<int64_t>(mb) * kb + mb;
        float* cs = (colsum_a != nullptr && ki == 0) ? colsum_a + mi : nullptr;
        hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st,
                           static_cast<const float*>(ws), nblk, mb, kb, 256, 1, c + static_cast<int64_t>(mi) * ldc + ki, ldc, cs);
        SGF_LAUNCH_CHECK();
        continue;
      }
      const int DP = padded_dim(mb > kb ? mb : kb);
      const int R = reduce_rows_per_tile<T, kModeGram>(DP);
      const int64_t ntiles = (n + R - 1) / R;
      const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
      ReduceArgs r{};
      r.a = static_cast<const T*>(a) + mi; r.lda = lda;
      r.b = static_cast<const T*>(b) + ki; r.ldb = ldb;
      r.q = nullptr; r.ldq = 0; r.den = nullptr;
      r.n = n; r.d = mb; r.db = kb; r.heads = 1; r.b_heads = 1; r.gscale = 1.f;
      r.partial = static_cast<float*>(ws);
      int rc = launch_reduce<T, kModeGram>(r, DP, nblk, st);
      if (rc != SGF_OK) return rc;
      const int RG = reduce_row_groups<T, kModeGram>(DP);
      const int64_t len = static_cast<int64_t>(mb) * kb + mb;
      float* cs = (colsum_a != nullptr && ki == 0) ? colsum_a + mi : nullptr;
      hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0,
                         st, r.partial, nblk, mb, kb, DP, RG, c + static_cast<int64_t>(mi) * ldc + ki,
                         ldc, cs);
      SGF_LAUNCH_CHECK();
    }
  }
  return SGF_OK;
}
}  // namespace

// dW / db of a Linear whose output feeds a BatchNorm, WITHOUT a materialised dz (the stem: x is data, nobody else reads dz):
//   c = dz^T b,  colsum = sum_n dz,   dz = BatchNorm'(relu'(g1 [+ g2])) formed per 4 x 4 patch inside the Gram's staging step.
extern "C" int32_t sgf_gram_bn_bwd_supported(int32_t m, int32_t k, int32_t dtype) {
  return dtype == SGF_BF16 && m >= 4 && m <= 256 && m % 4 == 0 && k >= 4 && k <= 256 && k % 4 == 0 ? 1 : 0;
}

extern "C" int sgf_gram_bn_bwd(const void* g1, int64_t ldg1, const void* g2, int64_t ldg2, const void* z, int64_t ldz,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu,
                               const float* stats, float inv_n, int32_t training, int32_t m, const void* b, int64_t ldb,
                               int32_t k, int64_t n, int32_t dtype, float* c, int64_t ldc, float* colsum, void* workspace,
                               size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_gram_bn_bwd";
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", fn);
  SGF_REQUIRE(sgf_gram_bn_bwd_supported(m, k, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, m and k multiples of 4 up to 256 (m=%d k=%d dtype=%d)", fn, m, k, dtype);
  SGF_REQUIRE(c && ldc >= k, SGF_E_INVALID, "%s: null c or ldc < k", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemset2DAsync(c, ldc * sizeof(float), 0, k * sizeof(float), m, st));
    if (colsum) SGF_CHECK_HIP(hipMemsetAsync(colsum, 0, m * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(g1 && z && b && mean && rstd && (!training || stats), SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gram_workspace_bytes(n, m, k), SGF_E_WORKSPACE, "%s: workspace too small", fn);
  SGF_REQUIRE(aligned4<uint16_t>(g1, ldg1) && (!g2 || aligned4<uint16_t>(g2, ldg2)) && aligned4<uint16_t>(z, ldz) &&
                  aligned4<uint16_t>(b, ldb),
              SGF_E_INVALID, "%s: operands must be 4-element aligned with ld %% 4 == 0", fn);
  if (gramt_supported(m, k, n) && gramt_aligned(g1, ldg1) && gramt_aligned(g2, ldg2) && gramt_aligned(z, ldz) &&
      gramt_aligned(b, ldb)) {                          // the operand formed in LDS from DMA-streamed tiles (csrc/gramx.hip)
    int nb = 0;
    float* part = static_cast<float*>(workspace);
    int rc = gramt_bn(g1, ldg1, g2, ldg2, z, ldz, mean, rstd, gamma, beta, relu, stats, inv_n, training, m, b, ldb, k, n, part,
                      &nb, st);
    if (rc != SGF_OK) return rc;
    const int64_t len = static_cast<int64_t>(m) * k + m;
    hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st, part, nb, m, k, 256,
                       1, c, ldc, colsum);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int DP = padded_dim(m > k ? m : k);
  const int R = reduce_rows_per_tile<uint16_t, kModeGramBN>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  ReduceArgs r{};
  r.a = g1; r.lda = ldg1; r.a2 = g2; r.lda2 = ldg2; r.q = z; r.ldq = ldz; r.b = b; r.ldb = ldb; r.den = nullptr;
  r.n = n; r.d = m; r.db = k; r.heads = 1; r.b_heads = 1; r.gscale = 1.f;
  r.bn_mean = mean; r.bn_rstd = rstd; r.bn_gamma = gamma; r.bn_beta = beta; r.bn_stats = stats; r.bn_inv_n = inv_n;
  r.bn_training = training; r.bn_relu = relu;
  r.partial = static_cast<float*>(workspace);
  int rc = launch_reduce<uint16_t, kModeGramBN>(r, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<uint16_t, kModeGramBN>(DP);
  const int64_t len = static_cast<int64_t>(m) * k + m;
  hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st, r.partial, nblk, m,
                     k, DP, RG, c, ldc, colsum);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// dW / db of a Linear whose output feeds a LayerNorm (+ activation), and the LayerNorm's own d gamma / d beta, WITHOUT a
// materialised input gradient (TransConv's stem, large/ours.py:198-201: its input is data):
//   c = dl^T b, colsum = sum dl, dgamma = sum g' xhat, dbeta = sum g',  dl = sgf_ln_bwd(g, ...) formed inside the Gram kernel.
extern "C" int32_t sgf_gram_ln_bwd_supported(int32_t m, int32_t k, int32_t dtype) {
  return dtype == SGF_BF16 && (m == 64 || m == 128 || m == 256) && k >= 4 && k <= 256 && k % 4 == 0 ? 1 : 0;
}

extern "C" int sgf_gram_ln_bwd(const void* g, int64_t ldg, const void* xin, int64_t ldx, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int32_t relu, int32_t m, const void* b, int64_t ldb,
                               int32_t k, int64_t n, int32_t dtype, float* c, int64_t ldc, float* colsum, float* dgamma,
                               float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_gram_ln_bwd";
  SGF_REQUIRE(n >= 0, SGF_E_INVALID, "%s: negative n", fn);
  SGF_REQUIRE(sgf_gram_ln_bwd_supported(m, k, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, m in {64, 128, 256} (a row spans whole lane groups), k %% 4 == 0 up to 256 (m=%d k=%d dtype=%d)",
              fn, m, k, dtype);
  SGF_REQUIRE(c && ldc >= k, SGF_E_INVALID, "%s: null c or ldc < k", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemset2DAsync(c, ldc * sizeof(float), 0, k * sizeof(float), m, st));
    for (float* v : {colsum, dgamma, dbeta})
      if (v) SGF_CHECK_HIP(hipMemsetAsync(v, 0, m * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(g && xin && b && mean && rstd, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gram_workspace_bytes(n, m, k), SGF_E_WORKSPACE, "%s: workspace too small", fn);
  SGF_REQUIRE(aligned4<uint16_t>(g, ldg) && aligned4<uint16_t>(xin, ldx) && aligned4<uint16_t>(b, ldb), SGF_E_INVALID,
              "%s: operands must be 4-element aligned with ld %% 4 == 0", fn);
  if (gramt_supported(m, k, n) && gramt_aligned(g, ldg) && gramt_aligned(xin, ldx) && gramt_aligned(b, ldb)) {   // csrc/gramx.hip
    int nb = 0;
    float* part = static_cast<float*>(workspace);
    int rc = gramt_ln(g, ldg, xin, ldx, mean, rstd, gamma, beta, relu, m, b, ldb, k, n, part, &nb, st);
    if (rc != SGF_OK) return rc;
    const int64_t len = static_cast<int64_t>(m) * k + m;
    hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st, part, nb, m, k, 256,
                       1, c, ldc, colsum);
    if (dbeta) hipLaunchKernelGGL(k_vec_finalize, dim3((m + 31) / 32), dim3(256), 0, st, part, nb, kVecB, m, dbeta);
    if (dgamma) hipLaunchKernelGGL(k_vec_finalize, dim3((m + 31) / 32), dim3(256), 0, st, part, nb, kVecC, m, dgamma);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int DP = m;                                    // a LayerNorm row = exactly one patch row of DP / 4 lanes
  const int R = reduce_rows_per_tile<uint16_t, kModeGramLN>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  ReduceArgs r{};
  r.a = g; r.lda = ldg; r.q = xin; r.ldq = ldx; r.b = b; r.ldb = ldb; r.den = nullptr;
  r.n = n; r.d = m; r.db = k; r.heads = 1; r.b_heads = 1; r.gscale = 1.f;
  r.bn_mean = mean; r.bn_rstd = rstd; r.bn_gamma = gamma; r.bn_beta = beta; r.bn_relu = relu;
  r.partial = static_cast<float*>(workspace);
  int rc = launch_reduce<uint16_t, kModeGramLN>(r, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<uint16_t, kModeGramLN>(DP);
  const int64_t len = static_cast<int64_t>(m) * k + m;
  hipLaunchKernelGGL(k_gram_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st, r.partial, nblk, m,
                     k, DP, RG, c, ldc, colsum);
  if (dbeta) hipLaunchKernelGGL(k_vec_finalize, dim3((m + 31) / 32), dim3(256), 0, st, r.partial, nblk, kVecB, m, dbeta);
  if (dgamma) hipLaunchKernelGGL(k_vec_finalize, dim3((m + 31) / 32), dim3(256), 0, st, r.partial, nblk, kVecC, m, dgamma);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// Two Gram products that share A in ONE paired launch (bf16 storage, m, k <= 256): c1 = a^T b1, c2 = a^T b2.
extern "C" int sgf_gram2(const void* a, int64_t lda, int32_t m, const void* b1, int64_t ldb1, const void* b2, int64_t ldb2,
                         int32_t k, int64_t n, int32_t dtype, float* c1, int64_t ldc1, float* c2, int64_t ldc2,
                         float* colsum_a, void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_gram2";
  SGF_REQUIRE(n >= 0 && m >= 1 && k >= 1, SGF_E_INVALID, "%s: bad sizes n=%lld m=%d k=%d", fn, static_cast<long long>(n), m, k);
  SGF_REQUIRE(m % 4 == 0 && k % 4 == 0, SGF_E_UNSUPPORTED, "%s: m and k must be multiples of 4 (m=%d k=%d)", fn, m, k);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "%s: unknown dtype %d", fn, dtype);
  SGF_REQUIRE(c1 && c2 && ldc1 >= k && ldc2 >= k, SGF_E_INVALID, "%s: null c or ldc < k", fn);
  const int R = dtype == SGF_BF16 && m <= 256 && k <= 256
                    ? reduce_rows_per_tile<uint16_t, kModeGram>(padded_dim(m > k ? m : k)) : 1;
  const int64_t ntiles = (n + R - 1) / R;
  if (dtype != SGF_BF16 || m > 256 || k > 256 || ntiles < 16) {     // nothing to pair: two plain products
    int rc = sgf_gram(a, lda, m, b1, ldb1, k, n, dtype, c1, ldc1, colsum_a, workspace, workspace_bytes, stream);
    if (rc != SGF_OK) return rc;
    return sgf_gram(a, lda, m, b2, ldb2, k, n, dtype, c2, ldc2, nullptr, workspace, workspace_bytes, stream);
  }
  SGF_REQUIRE(a && b1 && b2, SGF_E_INVALID, "%s: null operand", fn);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gram_workspace_bytes(n, m, k), SGF_E_WORKSPACE, "%s: workspace too small", fn);
  SGF_REQUIRE(aligned4<uint16_t>(a, lda) && aligned4<uint16_t>(b1, ldb1) && aligned4<uint16_t>(b2, ldb2), SGF_E_INVALID,
              "%s: a / b must be 4-element aligned with ld %% 4 == 0", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gramx_supported(a, lda, m, b1, ldb1, k, n) && gramx_supported(a, lda, m, b2, ldb2, k, n)) {   // csrc/gramx.hip
    int np = 0;
    float* part = static_cast<float*>(workspace);
    int rc = gramx_gram(a, lda, m, b1, ldb1, b2, ldb2, k, n, part, &np, st);
    if (rc != SGF_OK) return rc;
    const int64_t len = static_cast<int64_t>(m) * k + m;
    const unsigned fb = static_cast<unsigned>((kFinChains * len + 255) / 256);
    hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, part, np, m, k, 256, 1, c1, ldc1, colsum_a);
    hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, part + static_cast<int64_t>(np) * kPartialStride, np, m, k,
                       256, 1, c2, ldc2, static_cast<float*>(nullptr));
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int DP = padded_dim(m > k ? m : k);
  int64_t pairs = ntiles / 2 < kMaxBlocks / 2 ? ntiles / 2 : kMaxBlocks / 2;
  pairs = pairs / 8 * 8;                               // whole groups of 8 pairs = 16 consecutive blocks (>= 8: ntiles >= 16)
  ReduceArgs r{};
  r.a = a; r.lda = lda; r.b = b1; r.ldb = ldb1; r.b2 = b2; r.ldb2 = ldb2; r.pair = 1;
  r.q = nullptr; r.ldq = 0; r.den = nullptr;
  r.n = n; r.d = m; r.db = k; r.heads = 1; r.b_heads = 1; r.gscale = 1.f;
  r.partial = static_cast<float*>(workspace);
  int rc = launch_reduce<uint16_t, kModeGram>(r, DP, static_cast<int>(2 * pairs), st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<uint16_t, kModeGram>(DP);
  const int64_t len = static_cast<int64_t>(m) * k + m;
  const unsigned fb = static_cast<unsigned>((kFinChains * len + 255) / 256);
  hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, r.partial, static_cast<int>(pairs), m, k, DP, RG, c1, ldc1,
                     colsum_a);
  hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, r.partial + pairs * kPartialStride, static_cast<int>(pairs),
                     m, k, DP, RG, c2, ldc2, static_cast<float*>(nullptr));
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int32_t sgf_gram2_bn_bwd_supported(int32_t m, int32_t k, int64_t n, int32_t dtype) {
  return dtype == SGF_BF16 && gramb2_supported(m, k, n) ? 1 : 0;
}

extern "C" int sgf_gram2_bn_bwd(const void* g, int64_t ldg, const void* z, int64_t ldz, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, int32_t relu, const float* stats, float inv_n,
                                int32_t training, int32_t m, const void* b1, int64_t ldb1, const void* b2, int64_t ldb2, int32_t k,
                                int64_t n, int32_t dtype, void* dz, int64_t lddz, float* c1, int64_t ldc1, float* c2, int64_t ldc2,
                                float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_gram2_bn_bwd";
  SGF_REQUIRE(sgf_gram2_bn_bwd_supported(m, k, n, dtype), SGF_E_UNSUPPORTED,
              "%s: bf16 storage, m and k multiples of 8 up to 256, n >= 16384 (m=%d k=%d n=%lld dtype=%d)", fn, m, k,
              static_cast<long long>(n), dtype);
  SGF_REQUIRE(g && z && b1 && b2 && dz && c1 && c2 && mean && rstd && (!training || stats), SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(ldc1 >= k && ldc2 >= k && lddz >= m, SGF_E_INVALID, "%s: ldc < k or lddz < m", fn);
  SGF_REQUIRE(gramt_aligned(g, ldg) && gramt_aligned(z, ldz) && gramt_aligned(b1, ldb1) && gramt_aligned(b2, ldb2) &&
                  gramt_aligned(dz, lddz),
              SGF_E_INVALID, "%s: operands must be 16-byte aligned with ld %% 8 == 0", fn);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gram_workspace_bytes(n, m, k), SGF_E_WORKSPACE, "%s: workspace too small", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int np = 0;
  float* part = static_cast<float*>(workspace);
  int rc = gramb2(g, ldg, z, ldz, mean, rstd, gamma, beta, relu, stats, inv_n, training, m, b1, ldb1, b2, ldb2, k, n, dz, lddz,
                  part, &np, st);
  if (rc != SGF_OK) return rc;
  const int64_t len = static_cast<int64_t>(m) * k + m;
  const unsigned fb = static_cast<unsigned>((kFinChains * len + 255) / 256);
  hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, part, np, m, k, 256, 1, c1, ldc1, colsum);
  hipLaunchKernelGGL(k_gram_finalize, dim3(fb), dim3(256), 0, st, part + static_cast<int64_t>(np) * kPartialStride, np, m, k, 256,
                     1, c2, ldc2, static_cast<float*>(nullptr));
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" size_t sgf_gram_workspace_bytes(int64_t n, int32_t m, int32_t k) {
  (void)n; (void)m; (void)k;
  return static_cast<size_t>(kMaxBlocks) * kPartialStride * sizeof(float);
}

extern "C" int sgf_gram(const void* a, int64_t lda, int32_t m, const void* b, int64_t ldb, int32_t k,
                        int64_t n, int32_t dtype, float* c, int64_t ldc, float* colsum_a,
                        void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(n >= 0 && m >= 1 && k >= 1, SGF_E_INVALID, "sgf_gram: bad sizes n=%lld m=%d k=%d",
              static_cast<long long>(n), m, k);
  SGF_REQUIRE(m % 4 == 0 && k % 4 == 0, SGF_E_UNSUPPORTED,
              "sgf_gram: m and k must be multiples of 4 (m=%d k=%d)", m, k);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "sgf_gram: unknown dtype %d", dtype);
  SGF_REQUIRE(c && ldc >= k, SGF_E_INVALID, "sgf_gram: null c or ldc < k");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemset2DAsync(c, ldc * sizeof(float), 0, k * sizeof(float), m, st));
    if (colsum_a) SGF_CHECK_HIP(hipMemsetAsync(colsum_a, 0, m * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(a && b, SGF_E_INVALID, "sgf_gram: null operand");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_gram_workspace_bytes(n, m, k), SGF_E_WORKSPACE,
              "sgf_gram: workspace too small");
  if (dtype == SGF_F32) return gram_t<float>(a, lda, m, b, ldb, k, n, c, ldc, colsum_a, workspace, st);
  return gram_t<uint16_t>(a, lda, m, b, ldb, k, n, c, ldc, colsum_a, workspace, st);
}

// ------------------------------------------------------------------------------------------------
// attention from the un-projected input (H = 1): see include/sgf.h
// ------------------------------------------------------------------------------------------------
extern "C" int64_t sgf_attn_h_bstats_len(int32_t d) { return static_cast<int64_t>(d) * d + 2 * d + 1; }

namespace {
template <typename T>
int h_fwd_t(const void* h, int64_t ldh, int64_t n, int d, const float* M, const float* m, const float* w,
            const float* beta, void* out, int64_t ldo, float* den, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(h, ldh) && aligned4<T>(out, ldo), SGF_E_INVALID, "sgf_attn_h_fwd: alignment");
  ApplyArgs a{};
  a.a = h; a.lda = ldh;
  a.out = out; a.ldo = ldo;
  a.bmat = M; a.trans_b = 0; a.cvec = m; a.dvec = w; a.beta = beta;
  a.den = den; a.n = n; a.d = d; a.heads = 1; a.gscale = 1.f; a.accumulate = 0;
  return launch_apply<T, kApplyHFwd>(a, padded_dim(d), st);
}

template <typename T>
int h_bwd_reduce_t(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o, int64_t ldo,
                   const float* den, int64_t n, int d, float* hstats, void* ws, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(h, ldh) && aligned4<T>(g, ldg) && aligned4<T>(o, ldo), SGF_E_INVALID,
              "sgf_attn_h_bwd_reduce: h/g/o must be 4-element aligned with ld %% 4 == 0");
  const int DP = padded_dim(d);
  const int R = reduce_rows_per_tile<T, kModeBwdH>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  ReduceArgs a{};
  a.a = h; a.lda = ldh;
  a.b = g; a.ldb = ldg;
  a.q = o; a.ldq = ldo;
  a.den = den;
  a.n = n; a.d = d; a.db = d; a.heads = 1; a.b_heads = 1; a.gscale = 1.f;
  a.partial = static_cast<float*>(ws);
  int rc = launch_reduce<T, kModeBwdH>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<T, kModeBwdH>(DP);
  const int64_t len = sgf_attn_h_bstats_len(d);
  hipLaunchKernelGGL(k_hbwd_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st,
                     a.partial, nblk, d, DP, RG, hstats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

template <typename T>
int h_bwd_apply_t(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o, int64_t ldo,
                  const float* den, int64_t n, int d, const float* M, const float* w, const float* D,
                  const float* ds, void* dh, int64_t lddh, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(h, ldh) && aligned4<T>(g, ldg) && aligned4<T>(o, ldo) && aligned4<T>(dh, lddh),
              SGF_E_INVALID, "sgf_attn_h_bwd_apply: operands must be 4-element aligned");
  const int DP = padded_dim(d);
  ApplyArgs a{};
  a.n = n; a.d = d; a.heads = 1; a.gscale = 1.f;
  a.den = const_cast<float*>(den);
  a.out = dh; a.ldo = lddh;
  // dh = dnum M^T + dden w
  a.a = g; a.lda = ldg;
  a.a2 = o; a.lda2 = ldo;
  a.bmat = M; a.trans_b = 1; a.cvec = w; a.accumulate = 0;
  int rc = launch_apply<T, kApplyHBwd1>(a, DP, st);
  if (rc != SGF_OK) return rc;
  // dh += h D + ds
  a.a = h; a.lda = ldh;
  a.a2 = nullptr; a.lda2 = 0;
  a.bmat = D; a.trans_b = 0; a.cvec = ds; a.accumulate = 1;
  return launch_apply<T, kApplyHBwd2>(a, DP, st);
}
}  // namespace

extern "C" int sgf_attn_h_fwd(const void* h, int64_t ldh, int64_t n, int32_t d, int32_t dtype,
                              const float* M, const float* m, const float* w, const float* beta,
                              void* out, int64_t ldo, float* den, void* stream) {
  int rc = check_common("sgf_attn_h_fwd", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(h && M && m && w && beta && out && den, SGF_E_INVALID, "sgf_attn_h_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32) return h_fwd_t<float>(h, ldh, n, d, M, m, w, beta, out, ldo, den, st);
  // bf16, d in {64, 128, 256}, 16-byte aligned rows: the per-wave streaming kernel of csrc/rowgemm.hip
  if (hrow_supported(d, dtype, h, ldh, nullptr, 0, nullptr, 0, out, ldo) && reinterpret_cast<uintptr_t>(den) % 16 == 0)
    return hrow_fwd(h, ldh, n, d, M, m, w, beta, out, ldo, den, st);
  return h_fwd_t<uint16_t>(h, ldh, n, d, M, m, w, beta, out, ldo, den, st);
}

extern "C" int sgf_attn_h_bwd_reduce(const void* h, int64_t ldh, const void* g, int64_t ldg,
                                     const void* o, int64_t ldo, const float* den, int64_t n,
                                     int32_t d, int32_t dtype, float* hstats, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_h_bwd_reduce", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(hstats, SGF_E_INVALID, "sgf_attn_h_bwd_reduce: null hstats");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(hstats, 0, sgf_attn_h_bstats_len(d) * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(h && g && o && den, SGF_E_INVALID, "sgf_attn_h_bwd_reduce: null pointer");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_attn_workspace_bytes(n, 1, d), SGF_E_WORKSPACE,
              "sgf_attn_h_bwd_reduce: workspace too small");
  if (dtype == SGF_F32)
    return h_bwd_reduce_t<float>(h, ldh, g, ldg, o, ldo, den, n, d, hstats, workspace, st);
  return h_bwd_reduce_t<uint16_t>(h, ldh, g, ldg, o, ldo, den, n, d, hstats, workspace, st);
}

extern "C" int sgf_attn_h_bwd_apply(const void* h, int64_t ldh, const void* g, int64_t ldg,
                                    const void* o, int64_t ldo, const float* den, int64_t n, int32_t d,
                                    int32_t dtype, const float* M, const float* w, const float* D,
                                    const float* ds, void* dh, int64_t lddh, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_h_bwd_apply", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(h && g && o && den && M && w && D && ds && dh, SGF_E_INVALID,
              "sgf_attn_h_bwd_apply: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return h_bwd_apply_t<float>(h, ldh, g, ldg, o, ldo, den, n, d, M, w, D, ds, dh, lddh, st);
  if (hrow_supported(d, dtype, h, ldh, g, ldg, o, ldo, dh, lddh)) {
    SGF_REQUIRE(workspace && reinterpret_cast<uintptr_t>(workspace) % 16 == 0 &&
                    workspace_bytes >= hrow_partial_bytes(n, d),
                SGF_E_WORKSPACE, "sgf_attn_h_bwd_apply: workspace %zu < %zu", workspace_bytes, hrow_partial_bytes(n, d));
    return hrow_bwd(h, ldh, g, ldg, o, ldo, den, n, d, M, w, D, ds, dh, lddh, workspace, st);
  }
  return h_bwd_apply_t<uint16_t>(h, ldh, g, ldg, o, ldo, den, n, d, M, w, D, ds, dh, lddh, st);
}

extern "C" size_t sgf_attn_h_bwd_apply_workspace_bytes(int64_t n, int32_t d, int32_t dtype) {
  if (n <= 0 || dtype != SGF_BF16 || !(d == 64 || d == 128 || d == 256)) return 0;
  return hrow_partial_bytes(n, d);
}

// ---- the same backward as three calls, so that the reduce can use the row scalars the first apply pass computes ----
extern "C" int32_t sgf_attn_h_bwd_split_supported(int32_t d, int32_t dtype) {
  return dtype == SGF_BF16 && (d == 64 || d == 128 || d == 256) ? 1 : 0;
}

extern "C" int sgf_attn_h_bwd_pre(const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n,
                                  int32_t d, int32_t dtype, const float* M, const float* w, void* workspace,
                                  size_t workspace_bytes, float* rowscal, void* stream) {
  int rc = check_common("sgf_attn_h_bwd_pre", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(sgf_attn_h_bwd_split_supported(d, dtype), SGF_E_UNSUPPORTED,
              "sgf_attn_h_bwd_pre: bf16 storage with d in {64, 128, 256} only (d=%d, dtype=%d)", d, dtype);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(g && o && den && M && w && rowscal, SGF_E_INVALID, "sgf_attn_h_bwd_pre: null pointer");
  SGF_REQUIRE(hrow_supported(d, dtype, g, ldg, o, ldo, nullptr, 0, g, ldg) && reinterpret_cast<uintptr_t>(rowscal) % 8 == 0,
              SGF_E_INVALID, "sgf_attn_h_bwd_pre: rows must be 16-byte aligned");
  SGF_REQUIRE(workspace && reinterpret_cast<uintptr_t>(workspace) % 16 == 0 && workspace_bytes >= hrow_partial_bytes(n, d),
              SGF_E_WORKSPACE, "sgf_attn_h_bwd_pre: workspace %zu < %zu", workspace_bytes, hrow_partial_bytes(n, d));
  return hrow_bwd_pre(g, ldg, o, ldo, den, n, d, M, w, workspace, rowscal, static_cast<hipStream_t>(stream));
}

extern "C" int sgf_attn_h_bwd_reduce_scaled(const void* h, int64_t ldh, const void* g, int64_t ldg, const float* rowscal,
                                            int64_t n, int32_t d, int32_t dtype, float* hstats, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_h_bwd_reduce_scaled", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(dtype == SGF_BF16, SGF_E_UNSUPPORTED, "sgf_attn_h_bwd_reduce_scaled: bf16 storage only");
  SGF_REQUIRE(hstats, SGF_E_INVALID, "sgf_attn_h_bwd_reduce_scaled: null hstats");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(hstats, 0, sgf_attn_h_bstats_len(d) * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(h && g && rowscal, SGF_E_INVALID, "sgf_attn_h_bwd_reduce_scaled: null pointer");
  SGF_REQUIRE(aligned4<uint16_t>(h, ldh) && aligned4<uint16_t>(g, ldg) && reinterpret_cast<uintptr_t>(rowscal) % 8 == 0,
              SGF_E_INVALID, "sgf_attn_h_bwd_reduce_scaled: h / g must be 4-element aligned with ld %% 4 == 0");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_attn_workspace_bytes(n, 1, d), SGF_E_WORKSPACE,
              "sgf_attn_h_bwd_reduce_scaled: workspace too small");
  if (gramx_supported(h, ldh, d, g, ldg, d, n)) {       // csrc/gramx.hip
    int nb = 0;
    rc = gramx_bwdhs(h, ldh, g, ldg, rowscal, d, n, static_cast<float*>(workspace), &nb, st);
    if (rc != SGF_OK) return rc;
    const int64_t len = sgf_attn_h_bstats_len(d);
    hipLaunchKernelGGL(k_hbwd_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st,
                       static_cast<const float*>(workspace), nb, d, 256, 1, hstats);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int DP = padded_dim(d);
  const int R = reduce_rows_per_tile<uint16_t, kModeBwdHS>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  ReduceArgs a{};
  a.a = h; a.lda = ldh;
  a.b = g; a.ldb = ldg;
  a.den = rowscal;
  a.n = n; a.d = d; a.db = d; a.heads = 1; a.b_heads = 1; a.gscale = 1.f;
  a.partial = static_cast<float*>(workspace);
  rc = launch_reduce<uint16_t, kModeBwdHS>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<uint16_t, kModeBwdHS>(DP);
  const int64_t len = sgf_attn_h_bstats_len(d);
  hipLaunchKernelGGL(k_hbwd_finalize, dim3(static_cast<unsigned>((kFinChains * len + 255) / 256)), dim3(256), 0, st, a.partial, nblk, d,
                     DP, RG, hstats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_attn_h_bwd_post(const void* h, int64_t ldh, int64_t n, int32_t d, int32_t dtype, const float* D,
                                   const float* ds, const void* workspace, size_t workspace_bytes, const void* addend,
                                   int64_t ldadd, void* dh, int64_t lddh, void* stream) {
  int rc = check_common("sgf_attn_h_bwd_post", n, 1, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(sgf_attn_h_bwd_split_supported(d, dtype), SGF_E_UNSUPPORTED,
              "sgf_attn_h_bwd_post: bf16 storage with d in {64, 128, 256} only (d=%d, dtype=%d)", d, dtype);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(h && D && ds && dh, SGF_E_INVALID, "sgf_attn_h_bwd_post: null pointer");
  SGF_REQUIRE(hrow_supported(d, dtype, h, ldh, nullptr, 0, nullptr, 0, dh, lddh), SGF_E_INVALID,
              "sgf_attn_h_bwd_post: rows must be 16-byte aligned");
  SGF_REQUIRE(workspace && reinterpret_cast<uintptr_t>(workspace) % 16 == 0 && workspace_bytes >= hrow_partial_bytes(n, d),
              SGF_E_WORKSPACE, "sgf_attn_h_bwd_post: workspace %zu < %zu", workspace_bytes, hrow_partial_bytes(n, d));
  SGF_REQUIRE(!addend || (reinterpret_cast<uintptr_t>(addend) % 16 == 0 && ldadd % 8 == 0 && ldadd >= d), SGF_E_INVALID,
              "sgf_attn_h_bwd_post: addend rows must be 16-byte aligned");
  return hrow_bwd_post(h, ldh, n, d, D, ds, workspace, addend, ldadd, dh, lddh, static_cast<hipStream_t>(stream));
}

extern "C" int64_t sgf_attn_stats_len(int32_t heads, int32_t d) {
  return static_cast<int64_t>(heads) * d * d + static_cast<int64_t>(heads) * d + 2;
}
extern "C" int64_t sgf_attn_bstats_len(int32_t heads, int32_t d) {
  return static_cast<int64_t>(heads) * d * d + static_cast<int64_t>(heads) * d + 1;
}
extern "C" size_t sgf_attn_workspace_bytes(int64_t n, int32_t heads, int32_t d) {
  (void)n;
  (void)d;
  if (heads < 1) return 0;
  return static_cast<size_t>(kMaxBlocks) * heads * kPartialStride * sizeof(float);
}

namespace {
template <typename T>
int fwd_reduce_t(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                 int64_t n, int heads, int v_heads, int d, float* stats, void* ws, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(q, ldq) && aligned4<T>(k, ldk) && aligned4<T>(v, ldv), SGF_E_INVALID,
              "sgf_attn_fwd_reduce: q/k/v must be 4-element aligned with ld %% 4 == 0");
  const int DP = padded_dim(d);
  const int R = reduce_rows_per_tile<T, kModeFwd>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  const int64_t len = sgf_attn_stats_len(heads, d);
  if (nblk == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(stats, 0, len * sizeof(float), st));
    return SGF_OK;
  }
  ReduceArgs a{};
  a.a = k; a.lda = ldk;
  a.b = v; a.ldb = ldv;
  a.q = q; a.ldq = ldq;
  a.den = nullptr;
  a.n = n; a.d = d; a.db = d; a.heads = heads; a.b_heads = v_heads; a.gscale = 1.f;
  a.partial = static_cast<float*>(ws);
  int rc = launch_reduce<T, kModeFwd>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<T, kModeFwd>(DP);
  const int fb = static_cast<int>((len + 255) / 256);
  hipLaunchKernelGGL(k_attn_finalize, dim3(fb), dim3(256), 0, st, a.partial, nblk, heads, d, DP, RG,
                     kModeFwd, stats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

template <typename T>
int fwd_apply_t(const void* q, int64_t ldq, const void* v, int64_t ldv, int64_t n, double n_total,
                int heads, int v_heads, int d, const float* stats, void* out, int64_t ldo,
                float* den, void* o_heads, hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(q, ldq), SGF_E_INVALID, "sgf_attn_fwd_apply: q alignment");
  if (n == 0) return SGF_OK;
  const int DP = padded_dim(d);
  const int64_t slen = sgf_attn_stats_len(heads, d);
  for (int h = 0; h < heads; ++h) {
    ApplyArgs a{};
    a.a = static_cast<const T*>(q) + static_cast<int64_t>(h) * d; a.lda = ldq;
    a.e = static_cast<const T*>(v) + (v_heads == 1 ? 0 : static_cast<int64_t>(h) * d); a.lde = ldv;
    if (heads == 1) { a.out = out; a.ldo = ldo; }
    else { a.out = static_cast<T*>(o_heads) + static_cast<int64_t>(h) * d; a.ldo = static_cast<int64_t>(heads) * d; }
    a.bmat = stats + static_cast<int64_t>(h) * d * d;
    a.cvec = stats + static_cast<int64_t>(heads) * d * d + static_cast<int64_t>(h) * d;
    a.den = den + h;
    a.stats = stats; a.stats_len = slen; a.sdot = nullptr;
    a.n = n; a.d = d; a.heads = heads;
    a.ntot = static_cast<float>(n_total); a.gscale = 1.f / heads;
    a.trans_b = 0; a.accumulate = 0;
    int rc = launch_apply<T, kApplyFwd>(a, DP, st);
    if (rc != SGF_OK) return rc;
  }
  if (heads > 1) {
    const int64_t tot = n * d;
    hipLaunchKernelGGL((k_head_mean<T>), dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256),
                       0, st, static_cast<const T*>(o_heads), n, heads, d, static_cast<T*>(out), ldo);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

template <typename T>
int bwd_reduce_t(const void* q, int64_t ldq, const void* g, int64_t ldg, const void* o, int64_t ldo,
                 const float* den, int64_t n, int heads, int d, float* bstats, void* ws,
                 hipStream_t st, bool g_per_head = false) {
  SGF_REQUIRE(aligned4<T>(q, ldq) && aligned4<T>(g, ldg) && aligned4<T>(o, ldo), SGF_E_INVALID,
              "sgf_attn_bwd_reduce: q/g/o must be 4-element aligned with ld %% 4 == 0");
  const int DP = padded_dim(d);
  const int R = reduce_rows_per_tile<T, kModeBwd>(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  const int64_t len = sgf_attn_bstats_len(heads, d);
  if (nblk == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(bstats, 0, len * sizeof(float), st));
    return SGF_OK;
  }
  ReduceArgs a{};
  a.a = q; a.lda = ldq;
  a.b = g; a.ldb = ldg;     // g is [n, d]: the gradient of the head MEAN, shared by all heads — or [n, H, d] (g_per_head)
  a.q = o; a.ldq = ldo;     // o is [n, H, d] (or out when H == 1)
  a.den = den;
  a.n = n; a.d = d; a.db = d; a.heads = heads;
  a.b_heads = g_per_head ? heads : 1; a.gscale = g_per_head ? 1.f : 1.f / heads;
  a.partial = static_cast<float*>(ws);
  int rc = launch_reduce<T, kModeBwd>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = reduce_row_groups<T, kModeBwd>(DP);
  const int fb = static_cast<int>((len + 1 + 255) / 256);
  hipLaunchKernelGGL(k_attn_finalize, dim3(fb), dim3(256), 0, st, a.partial, nblk, heads, d, DP, RG,
                     kModeBwd, bstats);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

template <typename T>
int bwd_apply_t(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n,
                double n_total, int heads, int v_heads, int d, const float* stats, float* bstats,
                void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                hipStream_t st, bool g_per_head = false) {
  SGF_REQUIRE(aligned4<T>(q, ldq) && aligned4<T>(k, ldk) && aligned4<T>(v, ldv) &&
                  aligned4<T>(g, ldg) && aligned4<T>(o, ldo),
              SGF_E_INVALID, "sgf_attn_bwd_apply: operands must be 4-element aligned");
  if (n == 0) return SGF_OK;
  const int DP = padded_dim(d);
  const int64_t slen = sgf_attn_stats_len(heads, d);
  const int64_t blen = sgf_attn_bstats_len(heads, d);
  // s_raw = <S0,dS0> + <z0,dz0>  (bstats[blen-1]); stats and bstats share the [M | vec] prefix
  hipLaunchKernelGGL(k_attn_sdot, dim3(1), dim3(1024), 0, st, stats, bstats, blen - 1);
  SGF_LAUNCH_CHECK();
  const int64_t mat = static_cast<int64_t>(heads) * d * d;
  for (int h = 0; h < heads; ++h) {
    const int64_t ho = static_cast<int64_t>(h) * d;
    ApplyArgs a{};
    a.stats = stats; a.stats_len = slen; a.sdot = bstats + (blen - 1);
    a.n = n; a.d = d; a.heads = heads;
    a.ntot = static_cast<float>(n_total); a.gscale = g_per_head ? 1.f : 1.f / heads;
    a.den = const_cast<float*>(den) + h;
    const T* gh = static_cast<const T*>(g) + (g_per_head ? ho : 0);     // this head's gradient (or the shared mean gradient)
    // dQ_h = c (dnum S0^T + dden z0) - s Q / ||Q||^2
    a.a = gh; a.lda = ldg;
    a.a2 = static_cast<const T*>(o) + ho; a.lda2 = ldo;
    a.e = static_cast<const T*>(q) + ho; a.lde = ldq;
    a.out = static_cast<T*>(dq) + ho; a.ldo = lddq;
    a.bmat = stats + static_cast<int64_t>(h) * d * d; a.trans_b = 1;
    a.cvec = stats + mat + ho; a.accumulate = 0;
    int rc = launch_apply<T, kApplyDQ>(a, DP, st);
    if (rc != SGF_OK) return rc;
    // dK_h = c (V dS0^T + dz0) - s K / ||K||^2
    a.a = static_cast<const T*>(v) + (v_heads == 1 ? 0 : ho); a.lda = ldv;
    a.a2 = nullptr; a.lda2 = 0;
    a.e = static_cast<const T*>(k) + ho; a.lde = ldk;
    a.out = static_cast<T*>(dk) + ho; a.ldo = lddk;
    a.bmat = bstats + static_cast<int64_t>(h) * d * d; a.trans_b = 1;
    a.cvec = bstats + mat + ho;
    rc = launch_apply<T, kApplyDK>(a, DP, st);
    if (rc != SGF_OK) return rc;
    // dV_h = N dnum + c K dS0       (accumulated over heads when V is shared)
    a.a = static_cast<const T*>(k) + ho; a.lda = ldk;
    a.e = gh; a.lde = ldg;
    a.out = static_cast<T*>(dv) + (v_heads == 1 ? 0 : ho); a.ldo = lddv;
    a.bmat = bstats + static_cast<int64_t>(h) * d * d; a.trans_b = 0;
    a.cvec = nullptr;
    a.accumulate = (v_heads == 1 && h > 0) ? 1 : 0;
    rc = launch_apply<T, kApplyDV>(a, DP, st);
    if (rc != SGF_OK) return rc;
  }
  return SGF_OK;
}
}  // namespace

extern "C" int sgf_attn_fwd_reduce(const void* q, int64_t ldq, const void* k, int64_t ldk,
                                   const void* v, int64_t ldv, int64_t n, int32_t heads,
                                   int32_t v_heads, int32_t d, int32_t dtype, float* stats,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_fwd_reduce", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(v_heads == heads || v_heads == 1, SGF_E_INVALID,
              "sgf_attn_fwd_reduce: v_heads must be H or 1");
  SGF_REQUIRE(stats && (n == 0 || (q && k && v)), SGF_E_INVALID, "sgf_attn_fwd_reduce: null pointer");
  SGF_REQUIRE(workspace_bytes >= sgf_attn_workspace_bytes(n, heads, d) && workspace, SGF_E_WORKSPACE,
              "sgf_attn_fwd_reduce: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return fwd_reduce_t<float>(q, ldq, k, ldk, v, ldv, n, heads, v_heads, d, stats, workspace, st);
  return fwd_reduce_t<uint16_t>(q, ldq, k, ldk, v, ldv, n, heads, v_heads, d, stats, workspace, st);
}

extern "C" int sgf_attn_fwd_apply(const void* q, int64_t ldq, const void* v, int64_t ldv, int64_t n,
                                  double n_total, int32_t heads, int32_t v_heads, int32_t d,
                                  int32_t dtype, const float* stats, void* out, int64_t ldo,
                                  float* den, void* o_heads, void* stream) {
  int rc = check_common("sgf_attn_fwd_apply", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(v_heads == heads || v_heads == 1, SGF_E_INVALID,
              "sgf_attn_fwd_apply: v_heads must be H or 1");
  SGF_REQUIRE(stats && (n == 0 || (q && v && out && den)), SGF_E_INVALID,
              "sgf_attn_fwd_apply: null pointer");
  SGF_REQUIRE(heads == 1 || o_heads || n == 0, SGF_E_INVALID,
              "sgf_attn_fwd_apply: o_heads required when heads > 1");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return fwd_apply_t<float>(q, ldq, v, ldv, n, n_total, heads, v_heads, d, stats, out, ldo, den,
                              o_heads, st);
  return fwd_apply_t<uint16_t>(q, ldq, v, ldv, n, n_total, heads, v_heads, d, stats, out, ldo, den,
                               o_heads, st);
}

extern "C" int sgf_attn_bwd_reduce(const void* q, int64_t ldq, const void* g, int64_t ldg,
                                   const void* o, int64_t ldo, const float* den, int64_t n,
                                   int32_t heads, int32_t d, int32_t dtype, float* bstats,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_bwd_reduce", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(bstats && (n == 0 || (q && g && o && den)), SGF_E_INVALID,
              "sgf_attn_bwd_reduce: null pointer");
  SGF_REQUIRE(workspace_bytes >= sgf_attn_workspace_bytes(n, heads, d) && workspace, SGF_E_WORKSPACE,
              "sgf_attn_bwd_reduce: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return bwd_reduce_t<float>(q, ldq, g, ldg, o, ldo, den, n, heads, d, bstats, workspace, st);
  return bwd_reduce_t<uint16_t>(q, ldq, g, ldg, o, ldo, den, n, heads, d, bstats, workspace, st);
}

extern "C" int sgf_attn_bwd_apply(const void* q, int64_t ldq, const void* k, int64_t ldk,
                                  const void* v, int64_t ldv, const void* g, int64_t ldg,
                                  const void* o, int64_t ldo, const float* den, int64_t n,
                                  double n_total, int32_t heads, int32_t v_heads, int32_t d,
                                  int32_t dtype, const float* stats, float* bstats, void* dq,
                                  int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                  void* stream) {
  int rc = check_common("sgf_attn_bwd_apply", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(v_heads == heads || v_heads == 1, SGF_E_INVALID,
              "sgf_attn_bwd_apply: v_heads must be H or 1");
  SGF_REQUIRE(stats && bstats && (n == 0 || (q && k && v && g && o && den && dq && dk && dv)),
              SGF_E_INVALID, "sgf_attn_bwd_apply: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return bwd_apply_t<float>(q, ldq, k, ldk, v, ldv, g, ldg, o, ldo, den, n, n_total, heads,
                              v_heads, d, stats, bstats, dq, lddq, dk, lddk, dv, lddv, st);
  return bwd_apply_t<uint16_t>(q, ldq, k, ldk, v, ldv, g, ldg, o, ldo, den, n, n_total, heads,
                               v_heads, d, stats, bstats, dq, lddq, dk, lddk, dv, lddv, st);
}

// The same backward for PER-HEAD output gradients (full_attention_conv returns [N, H, D], medium/ours.py:14-46, 100M/ours.py:12-53):
// g is [n, H, d] (ldg >= H * d), head h's gradient enters without the 1/H of the head mean.  H = 1: identical to the above.
extern "C" int sgf_attn_bwd_reduce_heads(const void* q, int64_t ldq, const void* g, int64_t ldg, const void* o, int64_t ldo,
                                         const float* den, int64_t n, int32_t heads, int32_t d, int32_t dtype, float* bstats,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common("sgf_attn_bwd_reduce_heads", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(bstats && (n == 0 || (q && g && o && den)), SGF_E_INVALID, "sgf_attn_bwd_reduce_heads: null pointer");
  SGF_REQUIRE(n == 0 || ldg >= static_cast<int64_t>(heads) * d, SGF_E_INVALID, "sgf_attn_bwd_reduce_heads: ldg < H * d");
  SGF_REQUIRE(workspace_bytes >= sgf_attn_workspace_bytes(n, heads, d) && workspace, SGF_E_WORKSPACE,
              "sgf_attn_bwd_reduce_heads: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return bwd_reduce_t<float>(q, ldq, g, ldg, o, ldo, den, n, heads, d, bstats, workspace, st, true);
  return bwd_reduce_t<uint16_t>(q, ldq, g, ldg, o, ldo, den, n, heads, d, bstats, workspace, st, true);
}

extern "C" int sgf_attn_bwd_apply_heads(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                        const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n,
                                        double n_total, int32_t heads, int32_t v_heads, int32_t d, int32_t dtype,
                                        const float* stats, float* bstats, void* dq, int64_t lddq, void* dk, int64_t lddk,
                                        void* dv, int64_t lddv, void* stream) {
  int rc = check_common("sgf_attn_bwd_apply_heads", n, heads, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(v_heads == heads || v_heads == 1, SGF_E_INVALID, "sgf_attn_bwd_apply_heads: v_heads must be H or 1");
  SGF_REQUIRE(stats && bstats && (n == 0 || (q && k && v && g && o && den && dq && dk && dv)), SGF_E_INVALID,
              "sgf_attn_bwd_apply_heads: null pointer");
  SGF_REQUIRE(n == 0 || ldg >= static_cast<int64_t>(heads) * d, SGF_E_INVALID, "sgf_attn_bwd_apply_heads: ldg < H * d");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return bwd_apply_t<float>(q, ldq, k, ldk, v, ldv, g, ldg, o, ldo, den, n, n_total, heads, v_heads, d, stats, bstats, dq,
                              lddq, dk, lddk, dv, lddv, st, true);
  return bwd_apply_t<uint16_t>(q, ldq, k, ldk, v, ldv, g, ldg, o, ldo, den, n, n_total, heads, v_heads, d, stats, bstats, dq,
                               lddq, dk, lddk, dv, lddv, st, true);
}
