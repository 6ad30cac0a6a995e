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

namespace sgf {
namespace {

// ------------------------------------------------------------------------------------------------
// shared geometry
// ------------------------------------------------------------------------------------------------
constexpr int kRedThreads = 1024;
constexpr int kTileElems = 65536;                  // RG * DP * DP, identical for every DP
constexpr int kPartialStride = kTileElems + 264;   // + [DP colsum | ssq_a | ssq_q], padded
constexpr int kMaxBlocks = kNumCU;                 // persistent: one block per CU

constexpr int kModeFwd = 0;  // reduce: A=K, B=V          apply: out
constexpr int kModeBwd = 1;  // reduce: A=Q, B=dnum
constexpr int kApplyFwd = 0, kApplyDQ = 1, kApplyDK = 2, kApplyDV = 3;

static inline int padded_dim(int d) { return d <= 64 ? 64 : (d <= 128 ? 128 : 256); }

struct ReduceArgs {
  const void* a;   // fwd: K      bwd: Q
  const void* b;   // fwd: V      bwd: g
  const void* q;   // fwd: Q (norm only)   bwd: o
  const float* den;  // bwd only: [n, heads]
  int64_t lda, ldb, ldq;
  int64_t n;
  int32_t d, heads, b_heads;  // b_heads: 1 -> operand b is shared across heads (use_weight=False)
  float gscale;               // bwd: 1/H
  float* partial;             // [gridDim.x * heads][kPartialStride]
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
  float ssq_a = 0.f;        // fwd: sum K^2
  float ssq_q = 0.f;        // fwd: sum Q^2

  const int64_t ntiles = (p.n + R - 1) / R;
  float4 ra = zero4(), rb = zero4(), rq = zero4();
  float rden = 1.f;

  auto issue = [&](int64_t tile) {
    const int64_t row = tile * R + srow;
    const bool ok = col_ok && row < p.n;
    ra = ok ? load4<T>(pa + row * p.lda) : zero4();
    rb = ok ? load4<T>(pb + row * p.ldb) : zero4();
    rq = ok ? load4<T>(pq + row * p.ldq) : zero4();
    if (MODE == kModeBwd) rden = (row < p.n) ? p.den[row * p.heads + head] : 1.f;
  };
  auto commit = [&](int buf) {
    float4 wa = ra, wb = rb;
    if (MODE == kModeFwd) {
      colsum.x += ra.x; colsum.y += ra.y; colsum.z += ra.z; colsum.w += ra.w;
      ssq_a += dot4(ra, ra);
      ssq_q += dot4(rq, rq);
    } else {
      // rb = g, rq = o:  dnum = (g/H)/den ; dden = -((g/H).o)/den
      const float gdo = group_sum<F4>(dot4(rb, rq));
      const float inv = p.gscale / rden;
      const float dden = -gdo * inv;
      wb = make_float4(rb.x * inv, rb.y * inv, rb.z * inv, rb.w * inv);
      colsum.x += ra.x * dden; colsum.y += ra.y * dden;
      colsum.z += ra.z * dden; colsum.w += ra.w * dden;
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
  const float* cvec;   // FWD/DQ: z0   DK: dz0   DV: unused
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
  const float ssq_q = p.stats[p.stats_len - 2];
  const float ssq_k = p.stats[p.stats_len - 1];
  const float c = 1.0f / (sqrtf(ssq_q) * sqrtf(ssq_k));
  float gconst = 0.f;
  if (MODE == kApplyDQ) gconst = -(c * p.sdot[0]) / ssq_q;
  if (MODE == kApplyDK) gconst = -(c * p.sdot[0]) / ssq_k;

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
      if (MODE == kApplyDQ) ra2[i] = ok ? load4<T>(pa2 + row * p.lda2) : zero4();
      if (MODE == kApplyDQ || MODE == kApplyDV)
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
      } else {  // DV
        gr = p.ntot * p.gscale / rden[i];
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
          const float4 e = load4<T>(pe + row * p.lde);
          const float4 c0 = *reinterpret_cast<const float4*>(&ldsC[lrow * LD + scol]);
          const float4 c1 = *reinterpret_cast<const float4*>(&ldsC[(RT + lrow) * LD + scol]);
          const float ar = rs[lrow], gr = rs[2 * RT + lrow];
          float4 v = make_float4(ar * (c0.x + c1.x) + gr * e.x, ar * (c0.y + c1.y) + gr * e.y,
                                 ar * (c0.z + c1.z) + gr * e.z, ar * (c0.w + c1.w) + gr * e.w);
          if (MODE == kApplyDQ || MODE == kApplyDK) {
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
// host side
// ------------------------------------------------------------------------------------------------
inline int reduce_rows_per_tile(int DP) { return kRedThreads / (DP / 4); }

template <typename T, int MODE>
int launch_reduce(const ReduceArgs& args, int DP, int nblk, hipStream_t st) {
  const dim3 grid(nblk, args.heads), block(kRedThreads);
  switch (DP) {
    case 64: hipLaunchKernelGGL((k_attn_reduce<T, 64, MODE>), grid, block, 0, st, args); break;
    case 128: hipLaunchKernelGGL((k_attn_reduce<T, 128, MODE>), grid, block, 0, st, args); break;
    default: hipLaunchKernelGGL((k_attn_reduce<T, 256, MODE>), grid, block, 0, st, args); break;
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

template <typename T, int MODE>
int launch_apply(const ApplyArgs& args, int DP, hipStream_t st) {
  const int RT = 32 * (8 / (DP / 32));
  int64_t ntiles = (args.n + RT - 1) / RT;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  const dim3 grid(nblk), block(kApplyThreads);
  switch (DP) {
    case 64: hipLaunchKernelGGL((k_attn_apply<T, 64, MODE>), grid, block, 0, st, args); break;
    case 128: hipLaunchKernelGGL((k_attn_apply<T, 128, MODE>), grid, block, 0, st, args); break;
    default: hipLaunchKernelGGL((k_attn_apply<T, 256, MODE>), grid, block, 0, st, args); break;
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
  const int R = reduce_rows_per_tile(DP);
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
  a.n = n; a.d = d; a.heads = heads; a.b_heads = v_heads; a.gscale = 1.f;
  a.partial = static_cast<float*>(ws);
  int rc = launch_reduce<T, kModeFwd>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = 16 / ((DP / 64) * (DP / 64));
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
                 hipStream_t st) {
  SGF_REQUIRE(aligned4<T>(q, ldq) && aligned4<T>(g, ldg) && aligned4<T>(o, ldo), SGF_E_INVALID,
              "sgf_attn_bwd_reduce: q/g/o must be 4-element aligned with ld %% 4 == 0");
  const int DP = padded_dim(d);
  const int R = reduce_rows_per_tile(DP);
  const int64_t ntiles = (n + R - 1) / R;
  const int nblk = static_cast<int>(ntiles < kMaxBlocks ? ntiles : kMaxBlocks);
  const int64_t len = sgf_attn_bstats_len(heads, d);
  if (nblk == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(bstats, 0, len * sizeof(float), st));
    return SGF_OK;
  }
  ReduceArgs a{};
  a.a = q; a.lda = ldq;
  a.b = g; a.ldb = ldg;     // g is [n, d]: shared by all heads
  a.q = o; a.ldq = ldo;     // o is [n, H, d] (or out when H == 1)
  a.den = den;
  a.n = n; a.d = d; a.heads = heads; a.b_heads = 1; a.gscale = 1.f / heads;
  a.partial = static_cast<float*>(ws);
  int rc = launch_reduce<T, kModeBwd>(a, DP, nblk, st);
  if (rc != SGF_OK) return rc;
  const int RG = 16 / ((DP / 64) * (DP / 64));
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
                hipStream_t st) {
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
    a.ntot = static_cast<float>(n_total); a.gscale = 1.f / heads;
    a.den = const_cast<float*>(den) + h;
    // dQ_h = c (dnum S0^T + dden z0) - s Q / ||Q||^2
    a.a = g; a.lda = ldg;
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
    a.e = g; a.lde = ldg;
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
