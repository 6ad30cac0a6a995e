// fused.hip — T5/T6/T7: the HBM-bound glue around the two branches, one pass each.
//
//   sgf_ln_fwd / sgf_ln_bwd : y = [relu](LayerNorm(a*x + b*res))    large/ours.py:198-202,210-216
//   sgf_colstats / sgf_bn_* : y = [relu](BatchNorm1d(x)) [+ res]    large/ours.py:77-81,87-93
//   sgf_axpby               : y = a*x1 + b*x2                        large/ours.py:269-270
//
// The reference runs each of these as 3-6 separate ATen passes over [N, d]; here each is one read
// of its inputs and one write of its output, 16 B per lane, fp32 arithmetic whatever the storage
// dtype.  Cross-row reductions (BatchNorm statistics, dgamma / dbeta) are two-stage and
// deterministic: per-block partials in a caller-provided workspace, then a fixed-order sum.
#include "common.h"

namespace sgf {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxD = 1024;
constexpr int kMaxStatBlocks = 2048;  // 8 blocks (32 waves) per CU

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ld4f(const float* p) { return *reinterpret_cast<const float4*>(p); }

inline int lanes_per_row(int d) {
  int l = 16;
  while (l * 4 < d && l < 64) l <<= 1;
  return l;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward:  LPR lanes per row, NC float4 chunks per lane (d <= 4*LPR*NC)
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR, int NC>
__global__ __launch_bounds__(kThreads) void k_ln_fwd(
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ res, int64_t ldr, float a, float b,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, float eps, int64_t n,
    int d, T* __restrict__ y, int64_t ldy, float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int RPB = kThreads / LPR;
  const int sl = threadIdx.x % LPR;
  const int sr = threadIdx.x / LPR;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * RPB; row0 < n;
       row0 += static_cast<int64_t>(gridDim.x) * RPB) {
    const int64_t row = row0 + sr;
    const bool rok = row < n;
    float4 pre[NC];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (c * LPR + sl) * 4;
      const bool ok = rok && col < d;
      float4 v = ok ? load4<T>(x + row * ldx + col) : zero4();
      v.x *= a; v.y *= a; v.z *= a; v.w *= a;
      if (res != nullptr && ok) {
        const float4 r = load4<T>(res + row * ldr + col);
        v.x = fmaf(b, r.x, v.x); v.y = fmaf(b, r.y, v.y);
        v.z = fmaf(b, r.z, v.z); v.w = fmaf(b, r.w, v.w);
      }
      pre[c] = v;
      sum += v.x + v.y + v.z + v.w;
    }
    float mu = 0.f, rs = 1.f;
    if (gamma != nullptr) {
      mu = group_sum<LPR>(sum) / static_cast<float>(d);
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = (c * LPR + sl) * 4;
        if (col < d) {
          const float dx = pre[c].x - mu, dy = pre[c].y - mu, dz = pre[c].z - mu, dw = pre[c].w - mu;
          sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
      }
      const float var = group_sum<LPR>(sq) / static_cast<float>(d);
      rs = 1.0f / sqrtf(var + eps);
      if (sl == 0 && rok) {
        mean[row] = mu;
        rstd[row] = rs;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (c * LPR + sl) * 4;
      if (rok && col < d) {
        float4 o = pre[c];
        if (gamma != nullptr) {
          const float4 g = ld4f(gamma + col);
          const float4 be = ld4f(beta + col);
          o.x = (o.x - mu) * rs * g.x + be.x;
          o.y = (o.y - mu) * rs * g.y + be.y;
          o.z = (o.z - mu) * rs * g.z + be.z;
          o.w = (o.w - mu) * rs * g.w + be.w;
        }
        if (relu) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        store4<T>(y + row * ldy + col, o);
      }
    }
  }
}

// bf16 rows of d = 8 * LPR elements, 16-byte aligned: 16 bytes per lane and load, R independent rows per lane in flight
// (the generic kernel above has one 8-byte load per lane outstanding while it reduces a row: latency-bound, 3.1-3.9 TB/s
// at d = 256).  Column of a lane is fixed, so gamma / beta live in registers.
template <int LPR, int R>
__global__ __launch_bounds__(kThreads) void k_ln_fwd_bf16x8(
    const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ res, int64_t ldr, float a, float b,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, float eps, int64_t n,
    uint16_t* __restrict__ y, int64_t ldy, float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int RPS = kThreads / LPR;                 // rows per slot of the block
  constexpr int RPB = RPS * R;
  constexpr float inv_d = 1.0f / static_cast<float>(8 * LPR);
  const int sl = threadIdx.x % LPR;
  const int sr = threadIdx.x / LPR;
  const int col = sl * 8;
  float g[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    g[e] = gamma ? gamma[col + e] : 1.f;
    be[e] = gamma ? beta[col + e] : 0.f;
  }
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * RPB; row0 < n; row0 += static_cast<int64_t>(gridDim.x) * RPB) {
    uint4 xv[R], rv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int64_t row = row0 + q * RPS + sr;
      const int64_t rr = row < n ? row : n - 1;
      xv[q] = *reinterpret_cast<const uint4*>(x + rr * ldx + col);
      if (res) rv[q] = *reinterpret_cast<const uint4*>(res + rr * ldr + col);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int64_t row = row0 + q * RPS + sr;
      const uint32_t xu[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = a * __uint_as_float(xu[e] << 16);
        v[2 * e + 1] = a * __uint_as_float(xu[e] & 0xffff0000u);
      }
      if (res) {
        const uint32_t ru[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = fmaf(b, __uint_as_float(ru[e] << 16), v[2 * e]);
          v[2 * e + 1] = fmaf(b, __uint_as_float(ru[e] & 0xffff0000u), v[2 * e + 1]);
        }
      }
      float mu = 0.f, rs = 1.f;
      if (gamma) {
        float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        mu = group_sum<LPR>(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sq = fmaf(v[e] - mu, v[e] - mu, sq);
        rs = 1.0f / sqrtf(group_sum<LPR>(sq) * inv_d + eps);
        if (sl == 0 && row < n) {
          mean[row] = mu;
          rstd[row] = rs;
        }
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = gamma ? (v[e] - mu) * rs * g[e] + be[e] : v[e];
        if (relu) o[e] = fmaxf(o[e], 0.f);
      }
      if (row < n) {
        uint4 ov;
        ov.x = static_cast<uint32_t>(f32_to_bf16(o[0])) | (static_cast<uint32_t>(f32_to_bf16(o[1])) << 16);
        ov.y = static_cast<uint32_t>(f32_to_bf16(o[2])) | (static_cast<uint32_t>(f32_to_bf16(o[3])) << 16);
        ov.z = static_cast<uint32_t>(f32_to_bf16(o[4])) | (static_cast<uint32_t>(f32_to_bf16(o[5])) << 16);
        ov.w = static_cast<uint32_t>(f32_to_bf16(o[6])) | (static_cast<uint32_t>(f32_to_bf16(o[7])) << 16);
        *reinterpret_cast<uint4*>(y + row * ldy + col) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  Per-block partial dgamma/dbeta: part[blk][2][d].
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR, int NC>
__global__ __launch_bounds__(kThreads) void k_ln_bwd(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ y, int64_t ldy,
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ res, int64_t ldr, float a, float b,
    const float* __restrict__ gamma, int relu, const float* __restrict__ mean,
    const float* __restrict__ rstd, int64_t n, int d, T* __restrict__ dxo, int64_t lddx,
    T* __restrict__ dro, int64_t lddr, float* __restrict__ part) {
  constexpr int RPB = kThreads / LPR;
  __shared__ float red[2 * RPB * LPR * NC * 4];
  const int sl = threadIdx.x % LPR;
  const int sr = threadIdx.x / LPR;
  float4 dg[NC], db[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    dg[c] = zero4();
    db[c] = zero4();
  }
  const bool ln = gamma != nullptr;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * RPB; row0 < n;
       row0 += static_cast<int64_t>(gridDim.x) * RPB) {
    const int64_t row = row0 + sr;
    const bool rok = row < n;
    const float mu = (ln && rok) ? mean[row] : 0.f;
    const float rs = (ln && rok) ? rstd[row] : 1.f;
    float4 dz[NC], xh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (c * LPR + sl) * 4;
      const bool ok = rok && col < d;
      float4 g = ok ? load4<T>(dy + row * lddy + col) : zero4();
      if (relu && ok) {
        const float4 yy = load4<T>(y + row * ldy + col);
        g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
        g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
      }
      float4 h = zero4();
      if (ln && ok) {
        float4 v = load4<T>(x + row * ldx + col);
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        if (res != nullptr) {
          const float4 r = load4<T>(res + row * ldr + col);
          v.x = fmaf(b, r.x, v.x); v.y = fmaf(b, r.y, v.y);
          v.z = fmaf(b, r.z, v.z); v.w = fmaf(b, r.w, v.w);
        }
        h = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
        const float4 gm = ld4f(gamma + col);
        dg[c].x += g.x * h.x; dg[c].y += g.y * h.y; dg[c].z += g.z * h.z; dg[c].w += g.w * h.w;
        db[c].x += g.x; db[c].y += g.y; db[c].z += g.z; db[c].w += g.w;
        g.x *= gm.x; g.y *= gm.y; g.z *= gm.z; g.w *= gm.w;  // dxhat
        s1 += g.x + g.y + g.z + g.w;
        s2 += g.x * h.x + g.y * h.y + g.z * h.z + g.w * h.w;
      }
      dz[c] = g;
      xh[c] = h;
    }
    float m1 = 0.f, m2 = 0.f;
    if (ln) {
      m1 = group_sum<LPR>(s1) / static_cast<float>(d);
      m2 = group_sum<LPR>(s2) / static_cast<float>(d);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = (c * LPR + sl) * 4;
      if (rok && col < d) {
        float4 dp = dz[c];
        if (ln) {
          dp.x = rs * (dp.x - m1 - xh[c].x * m2);
          dp.y = rs * (dp.y - m1 - xh[c].y * m2);
          dp.z = rs * (dp.z - m1 - xh[c].z * m2);
          dp.w = rs * (dp.w - m1 - xh[c].w * m2);
        }
        store4<T>(dxo + row * lddx + col, make_float4(a * dp.x, a * dp.y, a * dp.z, a * dp.w));
        if (dro != nullptr)
          store4<T>(dro + row * lddr + col, make_float4(b * dp.x, b * dp.y, b * dp.z, b * dp.w));
      }
    }
  }
  if (!ln) return;
  // block reduce of dg / db over the RPB row-slots (fixed order)
  constexpr int W = LPR * NC * 4;  // padded width
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = (c * LPR + sl) * 4;
    *reinterpret_cast<float4*>(&red[(0 * RPB + sr) * W + col]) = dg[c];
    *reinterpret_cast<float4*>(&red[(1 * RPB + sr) * W + col]) = db[c];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * d; j += kThreads) {
    const int which = j / d, col = j % d;
    float s = 0.f;
    for (int r = 0; r < RPB; ++r) s += red[(which * RPB + r) * W + col];
    part[static_cast<int64_t>(blockIdx.x) * 2 * d + j] = s;
  }
}

// out[j] = sum_b part[b][j], j < width.  Fixed summation order (deterministic): 32 interleaved partial
// chains per column (b = g, g+32, ...) combined in order g = 0..31 through LDS.  8 columns per block so that a
// 512-wide statistics vector spreads over 64 CUs (16 blocks of 32 columns ran 64 us on 4 MiB of partials:
// latency-bound; this is a per-BatchNorm-layer, per-LayerNorm kernel, 14 launches per step).
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, int nblk,
                                                      int width, float* __restrict__ out0,
                                                      float* __restrict__ out1, int split) {
  __shared__ float red[256];
  const int c = threadIdx.x & 7;
  const int g = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 independent loads in flight per thread
  if (j < width) {
    int b = g;
    for (; b + 96 < nblk; b += 128) {
      s0 += part[static_cast<int64_t>(b) * width + j];
      s1 += part[static_cast<int64_t>(b + 32) * width + j];
      s2 += part[static_cast<int64_t>(b + 64) * width + j];
      s3 += part[static_cast<int64_t>(b + 96) * width + j];
    }
    for (; b < nblk; b += 32) s0 += part[static_cast<int64_t>(b) * width + j];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && j < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k * 8 + c];
    if (j < split) {
      if (out0) out0[j] = t;
    } else {
      if (out1) out1[j - split] = t;
    }
  }
}

// generic column sum for ANY d (e.g. the [N, C] logits gradient, C = 47): part[blk][d]
template <typename T>
__global__ __launch_bounds__(kThreads) void k_colsum_any(const T* __restrict__ x, int64_t ldx,
                                                         int64_t n, int d, float* __restrict__ part) {
  __shared__ float red[kThreads];
  const int rpp = kThreads / d;            // row slots per pass (d <= 256)
  const int c = threadIdx.x % d;
  const int sr = threadIdx.x / d;
  float s = 0.f;
  if (sr < rpp) {
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    for (int64_t row = r0 + sr; row < r1; row += rpp) s += load1<T>(x + row * ldx + c);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < d) {
    float t = 0.f;
    for (int r = 0; r < rpp; ++r) t += red[r * d + threadIdx.x];
    part[static_cast<int64_t>(blockIdx.x) * d + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// column statistics skeleton: each thread owns one float4 column chunk and a row slot; functor F
// maps (row, col) -> two float4 contributions.  part[blk][2][d].
// ------------------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(kThreads) void k_colreduce(F f, int64_t n, int d,
                                                        float* __restrict__ part) {
  __shared__ float red[2 * kThreads * 4];
  const int f4 = d / 4;
  const int rpp = kThreads / f4;  // row slots per pass (>= 1 since d <= 1024)
  const int c4 = threadIdx.x % f4;
  const int sr = threadIdx.x / f4;
  const bool act = sr < rpp;
  const int col = c4 * 4;
  float4 s0 = zero4(), s1 = zero4();
  if (act) {
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    int64_t row = r0 + sr;
    for (; row + 3 * static_cast<int64_t>(rpp) < r1; row += 4 * static_cast<int64_t>(rpp)) {
      float4 a0[4], a1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) f(row + u * static_cast<int64_t>(rpp), col, a0[u], a1[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // same order as the scalar loop: bitwise the same sums
        s0.x += a0[u].x; s0.y += a0[u].y; s0.z += a0[u].z; s0.w += a0[u].w;
        s1.x += a1[u].x; s1.y += a1[u].y; s1.z += a1[u].z; s1.w += a1[u].w;
      }
    }
    for (; row < r1; row += rpp) {
      float4 v0, v1;
      f(row, col, v0, v1);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    }
  }
  *reinterpret_cast<float4*>(&red[threadIdx.x * 4]) = s0;
  *reinterpret_cast<float4*>(&red[(kThreads + threadIdx.x) * 4]) = s1;
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * d; j += kThreads) {
    const int which = j / d, cc = j % d;
    float s = 0.f;
    for (int r = 0; r < rpp; ++r) s += red[(which * kThreads + r * f4 + cc / 4) * 4 + (cc & 3)];
    part[static_cast<int64_t>(blockIdx.x) * 2 * d + j] = s;
  }
}

template <typename T>
struct ColStatsF {
  const T* x; int64_t ldx; const float* shift;
  __device__ void operator()(int64_t row, int col, float4& v0, float4& v1) const {
    float4 v = load4<T>(x + row * ldx + col);
    if (shift) {
      const float4 s = ld4f(shift + col);
      v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w;
    }
    v0 = v;
    v1 = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
  }
};

struct BnParams {
  const float *mean, *rstd, *gamma, *beta;
};
__device__ __forceinline__ void bn_coeffs(const BnParams& p, int col, float4& mu, float4& rs,
                                          float4& ga, float4& be) {
  mu = ld4f(p.mean + col);
  rs = ld4f(p.rstd + col);
  ga = p.gamma ? ld4f(p.gamma + col) : make_float4(1.f, 1.f, 1.f, 1.f);
  be = p.beta ? ld4f(p.beta + col) : zero4();
}

template <typename T>
struct BnBwdStatsF {
  const T* dy; int64_t lddy; const T* x; int64_t ldx; BnParams p; int relu;
  const T* dy2 = nullptr; int64_t lddy2 = 0;           // a second gradient of the same tensor, added in fp32 (or null)
  __device__ void operator()(int64_t row, int col, float4& v0, float4& v1) const {
    float4 mu, rs, ga, be;
    bn_coeffs(p, col, mu, rs, ga, be);
    const float4 xv = load4<T>(x + row * ldx + col);
    float4 g = load4<T>(dy + row * lddy + col);
    if (dy2 != nullptr) {
      const float4 g2 = load4<T>(dy2 + row * lddy2 + col);
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
    }
    const float4 xh = make_float4((xv.x - mu.x) * rs.x, (xv.y - mu.y) * rs.y, (xv.z - mu.z) * rs.z,
                                  (xv.w - mu.w) * rs.w);
    if (relu) {
      g.x = (xh.x * ga.x + be.x) > 0.f ? g.x : 0.f;
      g.y = (xh.y * ga.y + be.y) > 0.f ? g.y : 0.f;
      g.z = (xh.z * ga.z + be.z) > 0.f ? g.z : 0.f;
      g.w = (xh.w * ga.w + be.w) > 0.f ? g.w : 0.f;
    }
    v0 = g;
    v1 = make_float4(g.x * xh.x, g.y * xh.y, g.z * xh.z, g.w * xh.w);
  }
};

// ------------------------------------------------------------------------------------------------
// elementwise: BN apply / BN backward apply / axpby.  One float4 per thread per step.
// ------------------------------------------------------------------------------------------------
// Thread map of the BatchNorm element-wise kernels: a thread keeps ONE 4-column chunk for the whole
// kernel (its mean / rstd / gamma / beta / stats coefficients live in registers) and walks down the
// rows, kRowUnroll independent row loads in flight per thread.  f4 = d/4 chunks per row, rpp =
// 256 / f4 row slots per block pass; threads beyond rpp * f4 idle (only when 256 % f4 != 0).
constexpr int kRowUnroll = 4;

template <typename T>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const T* __restrict__ x, int64_t ldx,
                                                       BnParams p, const T* __restrict__ res,
                                                       int64_t ldr, int relu, int64_t n, int d,
                                                       T* __restrict__ y, int64_t ldy) {
  const int f4 = d / 4;
  const int rpp = kThreads / f4;
  const int sr = threadIdx.x / f4;
  if (sr >= rpp) return;
  const int col = (threadIdx.x % f4) * 4;
  float4 mu, rs, ga, be;
  bn_coeffs(p, col, mu, rs, ga, be);
  // y = x * sc + sh  with sc = rstd*gamma, sh = beta - mean*sc  would round differently from the
  // reference's (x - mean) * rstd * gamma + beta; keep the reference order of operations.
  const int64_t step = static_cast<int64_t>(gridDim.x) * rpp;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * rpp + sr; row0 < n; row0 += step * kRowUnroll) {
    float4 xv[kRowUnroll], rv[kRowUnroll];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const int64_t row = row0 + u * step;
      if (row < n) {
        xv[u] = load4<T>(x + row * ldx + col);
        if (res != nullptr) rv[u] = load4<T>(res + row * ldr + col);
      }
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const int64_t row = row0 + u * step;
      if (row < n) {
        float4 o = make_float4((xv[u].x - mu.x) * rs.x * ga.x + be.x, (xv[u].y - mu.y) * rs.y * ga.y + be.y,
                               (xv[u].z - mu.z) * rs.z * ga.z + be.z, (xv[u].w - mu.w) * rs.w * ga.w + be.w);
        if (relu) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (res != nullptr) {
          o.x += rv[u].x; o.y += rv[u].y; o.z += rv[u].z; o.w += rv[u].w;
        }
        store4<T>(y + row * ldy + col, o);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, BnParams p,
    int relu, const float* __restrict__ stats, float inv_n, int training, int64_t n, int d,
    T* __restrict__ dx, int64_t lddx) {
  const int f4 = d / 4;
  const int rpp = kThreads / f4;
  const int sr = threadIdx.x / f4;
  if (sr >= rpp) return;
  const int col = (threadIdx.x % f4) * 4;
  float4 mu, rs, ga, be;
  bn_coeffs(p, col, mu, rs, ga, be);
  float4 s0 = zero4(), s1 = zero4();
  if (training) {
    s0 = ld4f(stats + col);
    s1 = ld4f(stats + d + col);
  }
  const int64_t step = static_cast<int64_t>(gridDim.x) * rpp;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * rpp + sr; row0 < n; row0 += step * kRowUnroll) {
    float4 xv[kRowUnroll], gv[kRowUnroll];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const int64_t row = row0 + u * step;
      if (row < n) {
        xv[u] = load4<T>(x + row * ldx + col);
        gv[u] = load4<T>(dy + row * lddy + col);
      }
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const int64_t row = row0 + u * step;
      if (row < n) {
        float4 g = gv[u];
        const float4 xh = make_float4((xv[u].x - mu.x) * rs.x, (xv[u].y - mu.y) * rs.y,
                                      (xv[u].z - mu.z) * rs.z, (xv[u].w - mu.w) * rs.w);
        if (relu) {
          g.x = (xh.x * ga.x + be.x) > 0.f ? g.x : 0.f;
          g.y = (xh.y * ga.y + be.y) > 0.f ? g.y : 0.f;
          g.z = (xh.z * ga.z + be.z) > 0.f ? g.z : 0.f;
          g.w = (xh.w * ga.w + be.w) > 0.f ? g.w : 0.f;
        }
        if (training) {
          g.x -= s0.x * inv_n + xh.x * s1.x * inv_n;
          g.y -= s0.y * inv_n + xh.y * s1.y * inv_n;
          g.z -= s0.z * inv_n + xh.z * s1.z * inv_n;
          g.w -= s0.w * inv_n + xh.w * s1.w * inv_n;
        }
        store4<T>(dx + row * lddx + col,
                  make_float4(ga.x * rs.x * g.x, ga.y * rs.y * g.y, ga.z * rs.z * g.z, ga.w * rs.w * g.w));
      }
    }
  }
}

// ---- bf16 rows of d = 8 k elements, 16-byte aligned: 16 bytes per lane and load ---------------------------------------
// The kernels above move a bf16 row as 8 bytes per lane (4 columns): 512 bytes per wave instruction.  On MI355X that form
// streams at 4.5-4.8 TB/s where ATen's 16-byte elementwise kernels reach 5.7-6.0 on the same read : write mix
// (scripts/hbm_mix_probe.py).  Same thread map — a thread keeps its column chunk, so the coefficients stay in registers —
// with 8 columns per thread and a CONTIGUOUS run of rpp * kEw8Unroll rows per block pass.  Per-element arithmetic is the
// x4 kernels' (the reference's order of operations); only the vector width and, for the statistics, the order in which
// rows are added differ.  (Rounding is the hardware's v_cvt_pk_bf16_f32: the same round-to-nearest-even as f32_to_bf16 on every
// finite value and infinity; a NaN stays a NaN, its payload bits may differ from the software rounding's.)
typedef __bf16 ew_bf16v2 __attribute__((ext_vector_type(2)));
typedef float ew_f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ew_pack(float a, float b) {      // v_cvt_pk_bf16_f32: round-to-nearest-even
  const ew_f32v2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ew_bf16v2));
}
__device__ __forceinline__ void ew_unpack(const uint4& r, float (&v)[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 ew_pack8(const float (&o)[8]) {
  return make_uint4(ew_pack(o[0], o[1]), ew_pack(o[2], o[3]), ew_pack(o[4], o[5]), ew_pack(o[6], o[7]));
}
struct BnCoeffs8 {
  float mu[8], rs[8], ga[8], be[8];
};
__device__ __forceinline__ void bn_coeffs8(const BnParams& p, int col, BnCoeffs8& c) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c.mu[e] = p.mean[col + e];
    c.rs[e] = p.rstd[col + e];
    c.ga[e] = p.gamma ? p.gamma[col + e] : 1.f;
    c.be[e] = p.beta ? p.beta[col + e] : 0.f;
  }
}

template <bool kRes, int U>
__global__ __launch_bounds__(kThreads) void k_bn_apply_bf16x8(const uint16_t* __restrict__ x, int64_t ldx, BnParams p,
                                                              const uint16_t* __restrict__ res, int64_t ldr, int relu,
                                                              int64_t n, int d, uint16_t* __restrict__ y, int64_t ldy) {
  const int f8 = d / 8;
  const int rpp = kThreads / f8;
  const int sr = threadIdx.x / f8;
  if (sr >= rpp) return;
  const int col = (threadIdx.x % f8) * 8;
  BnCoeffs8 c;
  bn_coeffs8(p, col, c);
  const int64_t rpb = static_cast<int64_t>(rpp) * U;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * rpb + sr; row0 < n; row0 += static_cast<int64_t>(gridDim.x) * rpb) {
    uint4 xv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * rpp;
      if (row < n) {
        xv[u] = *reinterpret_cast<const uint4*>(x + row * ldx + col);
        if (kRes) rv[u] = *reinterpret_cast<const uint4*>(res + row * ldr + col);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * rpp;
      if (row < n) {
        float v[8], r[8], o[8];
        ew_unpack(xv[u], v);
        if (kRes) ew_unpack(rv[u], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = fmaf((v[e] - c.mu[e]) * c.rs[e], c.ga[e], c.be[e]);   // (explicitly what hipcc contracts the x4 kernel to)
          if (relu) o[e] = fmaxf(o[e], 0.f);
          if (kRes) o[e] += r[e];
        }
        *reinterpret_cast<uint4*>(y + row * ldy + col) = ew_pack8(o);
      }
    }
  }
}

template <int U>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply_bf16x8(
    const uint16_t* __restrict__ dy, int64_t lddy, const uint16_t* __restrict__ x, int64_t ldx, BnParams p, int relu,
    const float* __restrict__ stats, float inv_n, int training, int64_t n, int d, uint16_t* __restrict__ dx, int64_t lddx) {
  const int f8 = d / 8;
  const int rpp = kThreads / f8;
  const int sr = threadIdx.x / f8;
  if (sr >= rpp) return;
  const int col = (threadIdx.x % f8) * 8;
  BnCoeffs8 c;
  bn_coeffs8(p, col, c);
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s0[e] = training ? stats[col + e] : 0.f;
    s1[e] = training ? stats[d + col + e] : 0.f;
  }
  const int64_t rpb = static_cast<int64_t>(rpp) * U;
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * rpb + sr; row0 < n; row0 += static_cast<int64_t>(gridDim.x) * rpb) {
    uint4 xv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * rpp;
      if (row < n) {
        xv[u] = *reinterpret_cast<const uint4*>(x + row * ldx + col);
        gv[u] = *reinterpret_cast<const uint4*>(dy + row * lddy + col);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * rpp;
      if (row < n) {
        float v[8], g[8], o[8];
        ew_unpack(xv[u], v);
        ew_unpack(gv[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] - c.mu[e]) * c.rs[e];
          float ge = g[e];
          if (relu) ge = fmaf(xh, c.ga[e], c.be[e]) > 0.f ? ge : 0.f;
          if (training) ge -= s0[e] * inv_n + xh * s1[e] * inv_n;
          o[e] = c.ga[e] * c.rs[e] * ge;
        }
        *reinterpret_cast<uint4*>(dx + row * lddx + col) = ew_pack8(o);
      }
    }
  }
}

// k_ln_bwd for bf16 rows of d = 8 * LPR elements, 16-byte aligned: 16 bytes per lane and load, R rows per lane in flight.
// Per-element arithmetic as in k_ln_bwd; a row's two sums are taken 8 per lane, then over the LPR lanes.
template <int LPR, int R>
__global__ __launch_bounds__(kThreads) void k_ln_bwd_bf16x8(
    const uint16_t* __restrict__ dy, int64_t lddy, const uint16_t* __restrict__ y, int64_t ldy, const uint16_t* __restrict__ x,
    int64_t ldx, const uint16_t* __restrict__ res, int64_t ldr, float a, float b, const float* __restrict__ gamma, int relu,
    const float* __restrict__ mean, const float* __restrict__ rstd, int64_t n, uint16_t* __restrict__ dxo, int64_t lddx,
    uint16_t* __restrict__ dro, int64_t lddr, float* __restrict__ part) {
  constexpr int RPS = kThreads / LPR;
  constexpr int RPB = RPS * R;
  constexpr int D = 8 * LPR;
  constexpr float inv_d = 1.0f / static_cast<float>(D);
  __shared__ float red[2 * RPS * D];
  const int sl = threadIdx.x % LPR;
  const int sr = threadIdx.x / LPR;
  const int col = sl * 8;
  const bool ln = gamma != nullptr;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gm[e] = ln ? gamma[col + e] : 1.f;
    dg[e] = db[e] = 0.f;
  }
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * RPB; row0 < n; row0 += static_cast<int64_t>(gridDim.x) * RPB) {
    uint4 gv[R], yv[R], xv[R], rv[R];
    float mu[R], rs[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int64_t row = row0 + q * RPS + sr;
      const int64_t rr = row < n ? row : n - 1;
      gv[q] = *reinterpret_cast<const uint4*>(dy + rr * lddy + col);
      if (relu) yv[q] = *reinterpret_cast<const uint4*>(y + rr * ldy + col);
      if (ln) {
        xv[q] = *reinterpret_cast<const uint4*>(x + rr * ldx + col);
        if (res) rv[q] = *reinterpret_cast<const uint4*>(res + rr * ldr + col);
      }
      mu[q] = ln ? mean[rr] : 0.f;
      rs[q] = ln ? rstd[rr] : 1.f;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int64_t row = row0 + q * RPS + sr;
      const bool rok = row < n;
      float g[8], h[8];
      ew_unpack(gv[q], g);
      if (relu) {
        float yy[8];
        ew_unpack(yv[q], yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
      }
      float s1 = 0.f, s2 = 0.f;
      if (ln) {
        float v[8];
        ew_unpack(xv[q], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a;
        if (res) {
          float r[8];
          ew_unpack(rv[q], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(b, r[e], v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          h[e] = (v[e] - mu[q]) * rs[q];
          if (rok) {
            dg[e] += g[e] * h[e];
            db[e] += g[e];
          }
          g[e] *= gm[e];                                   // dxhat
          s1 += g[e];
          s2 += g[e] * h[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = 0.f;
      }
      float m1 = 0.f, m2 = 0.f;
      if (ln) {
        m1 = group_sum<LPR>(s1) * inv_d;
        m2 = group_sum<LPR>(s2) * inv_d;
      }
      if (rok) {
        float o[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dp = ln ? rs[q] * (g[e] - m1 - h[e] * m2) : g[e];
          o[e] = a * dp;
          o2[e] = b * dp;
        }
        *reinterpret_cast<uint4*>(dxo + row * lddx + col) = ew_pack8(o);
        if (dro != nullptr) *reinterpret_cast<uint4*>(dro + row * lddr + col) = ew_pack8(o2);
      }
    }
  }
  if (!ln) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(0 * RPS + sr) * D + col + e] = dg[e];
    red[(1 * RPS + sr) * D + col + e] = db[e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * D; j += kThreads) {
    const int which = j / D, cc = j % D;
    float s = 0.f;
    for (int r = 0; r < RPS; ++r) s += red[(which * RPS + r) * D + cc];
    part[static_cast<int64_t>(blockIdx.x) * 2 * D + j] = s;
  }
}

// BnBwdStatsF over bf16 rows, 8 columns per thread: part[blk][2][d] like k_colreduce
template <bool kTwo, int U>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_stats_bf16x8(
    const uint16_t* __restrict__ dy, int64_t lddy, const uint16_t* __restrict__ dy2, int64_t lddy2,
    const uint16_t* __restrict__ x, int64_t ldx, BnParams p, int relu, int64_t n, int d, float* __restrict__ part) {
  __shared__ float red[2 * kThreads * 8];
  const int f8 = d / 8;
  const int rpp = kThreads / f8;
  const int c8 = threadIdx.x % f8;
  const int sr = threadIdx.x / f8;
  const bool act = sr < rpp;
  const int col = c8 * 8;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (act) {
    BnCoeffs8 c;
    bn_coeffs8(p, col, c);
    const int64_t rows_per_blk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > n) r1 = n;
    auto add_row = [&](const uint4& xr, const uint4& gr, const uint4& g2r) {
      float v[8], g[8], g2[8];
      ew_unpack(xr, v);
      ew_unpack(gr, g);
      if (kTwo) ew_unpack(g2r, g2);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float ge = kTwo ? g[e] + g2[e] : g[e];
        const float xh = (v[e] - c.mu[e]) * c.rs[e];
        if (relu) ge = fmaf(xh, c.ga[e], c.be[e]) > 0.f ? ge : 0.f;
        s0[e] += ge;
        s1[e] += ge * xh;
      }
    };
    int64_t row = r0 + sr;
    for (; row + (U - 1) * static_cast<int64_t>(rpp) < r1; row += U * static_cast<int64_t>(rpp)) {
      uint4 xv[U], gv[U], hv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t rr = row + u * static_cast<int64_t>(rpp);
        xv[u] = *reinterpret_cast<const uint4*>(x + rr * ldx + col);
        gv[u] = *reinterpret_cast<const uint4*>(dy + rr * lddy + col);
        if (kTwo) hv[u] = *reinterpret_cast<const uint4*>(dy2 + rr * lddy2 + col);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) add_row(xv[u], gv[u], kTwo ? hv[u] : gv[u]);
    }
    for (; row < r1; row += rpp) {
      const uint4 xr = *reinterpret_cast<const uint4*>(x + row * ldx + col);
      const uint4 gr = *reinterpret_cast<const uint4*>(dy + row * lddy + col);
      const uint4 hr = kTwo ? *reinterpret_cast<const uint4*>(dy2 + row * lddy2 + col) : gr;
      add_row(xr, gr, hr);
    }
  }
  // red[which][slot sr][column]: the same fixed-order sum over the row slots as k_colreduce
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(0 * kThreads + threadIdx.x) * 8 + e] = s0[e];
    red[(1 * kThreads + threadIdx.x) * 8 + e] = s1[e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * d; j += kThreads) {
    const int which = j / d, cc = j % d;
    float s = 0.f;
    for (int r = 0; r < rpp; ++r) s += red[(which * kThreads + r * f8 + cc / 8) * 8 + (cc & 7)];
    part[static_cast<int64_t>(blockIdx.x) * 2 * d + j] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_axpby(const T* __restrict__ x1, int64_t ld1, float a,
                                                    const T* __restrict__ x2, int64_t ld2, float b,
                                                    int64_t n, int d, T* __restrict__ y,
                                                    int64_t ldy) {
  const int f4 = d / 4;
  const int64_t total = n * f4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t row = i / f4;
    const int col = static_cast<int>(i % f4) * 4;
    const float4 u = load4<T>(x1 + row * ld1 + col);
    const float4 v = load4<T>(x2 + row * ld2 + col);
    store4<T>(y + row * ldy + col,
              make_float4(a * u.x + b * v.x, a * u.y + b * v.y, a * u.z + b * v.z, a * u.w + b * v.w));
  }
}

// out[i, :] = cast(src[idx[i], :])  — row gather with an optional storage-dtype change on the way
// (x[perm] at the module boundary when the graph is re-ordered; x[idx_i] of a mini-batch,
// large/main-batch.py:138).  TI int32 / int64 indices; any d (vector path when d % 4 == 0).
template <typename TS, typename TD, typename TI>
__global__ __launch_bounds__(kThreads) void k_gather_rows(const TS* __restrict__ src, int64_t lds,
                                                          const TI* __restrict__ idx, int64_t n_out, int d,
                                                          int64_t n_src, TD* __restrict__ dst, int64_t ldd,
                                                          int vec) {
  if (vec) {
    const int f4 = d / 4;
    const int64_t total = n_out * f4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
      const int64_t row = i / f4;
      const int col = static_cast<int>(i % f4) * 4;
      const int64_t r = static_cast<int64_t>(idx[row]);
      const float4 v = (r >= 0 && r < n_src) ? load4<TS>(src + r * lds + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      store4<TD>(dst + row * ldd + col, v);
    }
  } else {
    const int64_t total = n_out * d;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
      const int64_t row = i / d;
      const int col = static_cast<int>(i % d);
      const int64_t r = static_cast<int64_t>(idx[row]);
      store1<TD>(dst + row * ldd + col, (r >= 0 && r < n_src) ? load1<TS>(src + r * lds + col) : 0.f);
    }
  }
}

// y = sum_i x_i for up to 8 equally shaped operands: the fused form of the pairwise gradient
// accumulation autograd performs for a tensor with many consumers (GraphConv's x0 feeds every
// layer's [.|x0] Linear and residual, large/ours.py:86-93: 7 gradients -> one 8-stream pass
// instead of six 3-stream adds).  fp32 accumulation in operand order.
struct SumArgs {
  const void* x[8];
  int64_t ld[8];
  int32_t k;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void k_sum_n(SumArgs a, int64_t n, int d, T* __restrict__ y,
                                                    int64_t ldy) {
  const int f4 = d / 4;
  const int64_t total = n * f4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t row = i / f4;
    const int col = static_cast<int>(i % f4) * 4;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.k) v[j] = load4<T>(static_cast<const T*>(a.x[j]) + row * a.ld[j] + col);
    float4 s = v[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < a.k) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    store4<T>(y + row * ldy + col, s);
  }
}

// bf16 operands whose rows are 16-byte aligned: 8 elements per lane and load (an 8-byte-per-lane load runs at 0.54-0.70x
// the rate of a 16-byte one on this part, MI355X_MICROARCH.md), all K loads in flight before the first add.  Same
// summation order as k_sum_n: bitwise the same result.
template <int K>
__global__ __launch_bounds__(kThreads) void k_sum_n_bf16x8(SumArgs a, int64_t n, int d, uint16_t* __restrict__ y,
                                                           int64_t ldy) {
  const int f8 = d / 8;
  const int64_t total = n * f8;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t row = i / f8;
    const int col = static_cast<int>(i % f8) * 8;
    uint4 v[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
      v[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(a.x[j]) + row * a.ld[j] + col);
    float s[8];
    {
      const uint32_t u[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] = __uint_as_float(u[e] << 16);
        s[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int j = 1; j < K; ++j) {
      const uint32_t u[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] += __uint_as_float(u[e] << 16);
        s[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
      }
    }
    uint4 o;
    o.x = static_cast<uint32_t>(f32_to_bf16(s[0])) | (static_cast<uint32_t>(f32_to_bf16(s[1])) << 16);
    o.y = static_cast<uint32_t>(f32_to_bf16(s[2])) | (static_cast<uint32_t>(f32_to_bf16(s[3])) << 16);
    o.z = static_cast<uint32_t>(f32_to_bf16(s[4])) | (static_cast<uint32_t>(f32_to_bf16(s[5])) << 16);
    o.w = static_cast<uint32_t>(f32_to_bf16(s[6])) | (static_cast<uint32_t>(f32_to_bf16(s[7])) << 16);
    *reinterpret_cast<uint4*>(y + row * ldy + col) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Dropout (F.dropout in large/ours.py:81,92,202,216), fused with the residual add that follows it in
// GraphConv (:92-93) and WITHOUT a stored mask: the keep decisions are a pure function of
// (seed, element index) — Philox4x32-10, one counter per 4 consecutive elements of a row — so the
// backward recomputes them.  ATen's native_dropout writes and re-reads an [N, d] mask.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
  const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
  const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
  const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// 4 keep flags (bit j = element 4*chunk + j is kept) for chunk index `chunk` under `seed`
__device__ __forceinline__ uint32_t dropout_keep4(uint64_t seed, uint64_t chunk, float p) {
  uint32_t c[4] = {static_cast<uint32_t>(chunk), static_cast<uint32_t>(chunk >> 32), 0x5347464du, 0u};
  uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = static_cast<float>(c[j] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
    bits |= (u >= p ? 1u : 0u) << j;
  }
  return bits;
}

// y = x * keep / (1 - p) [+ res];  MODE 1 (backward): y = x * keep / (1 - p) with x = dL/dy
template <typename T>
__global__ __launch_bounds__(kThreads) void k_dropout(const T* __restrict__ x, int64_t ldx,
                                                      const T* __restrict__ res, int64_t ldr, float p,
                                                      float scale, uint64_t seed, int64_t n, int d,
                                                      T* __restrict__ y, int64_t ldy) {
  const int f4 = d / 4;
  const int64_t total = n * f4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t row = i / f4;
    const int col = static_cast<int>(i % f4) * 4;
    const uint32_t keep = dropout_keep4(seed, static_cast<uint64_t>(i), p);
    const float4 v = load4<T>(x + row * ldx + col);
    float4 o = make_float4((keep & 1u) ? v.x * scale : 0.f, (keep & 2u) ? v.y * scale : 0.f,
                           (keep & 4u) ? v.z * scale : 0.f, (keep & 8u) ? v.w * scale : 0.f);
    if (res != nullptr) {
      const float4 r = load4<T>(res + row * ldr + col);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    store4<T>(y + row * ldy + col, o);
  }
}

// ------------------------------------------------------------------------------------------------
// N4: log_softmax + NLLLoss on the training rows (large/main.py:139-141), one wave per row.
// ------------------------------------------------------------------------------------------------
constexpr int kNllMaxBlocks = 1024;

// lse of row `r` (C classes), every lane returns it; lane-strided loads, fp32 math
template <typename T>
__device__ __forceinline__ float row_lse(const T* __restrict__ row, int c, int lane) {
  float mx = -3.402823466e+38f;
  for (int j = lane; j < c; j += 64) mx = fmaxf(mx, load1<T>(row + j));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float se = 0.f;
  for (int j = lane; j < c; j += 64) se += expf(load1<T>(row + j) - mx);
  se = group_sum<64>(se);
  return mx + logf(se);
}

// part[blk] = - sum over this block's training rows of log_softmax(logits[row])[label[row]]
template <typename T>
__global__ __launch_bounds__(kThreads) void k_nll_fwd(const T* __restrict__ logits, int64_t ldl, int c,
                                                      const int64_t* __restrict__ labels,
                                                      const int64_t* __restrict__ idx, int64_t m,
                                                      float* __restrict__ part) {
  __shared__ float red[kThreads / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t j0 = static_cast<int64_t>(blockIdx.x) * per;
  int64_t j1 = j0 + per;
  if (j1 > m) j1 = m;
  float acc = 0.f;  // lane 0 of each wave accumulates, rows in index order
  for (int64_t j = j0 + wave; j < j1; j += kThreads / 64) {
    const int64_t r = idx[j];
    const T* row = logits + r * ldl;
    const float lse = row_lse<T>(row, c, lane);
    const int64_t y = labels[r];
    if (lane == 0 && y >= 0 && y < c) acc += lse - load1<T>(row + y);
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    part[blockIdx.x] = s;
  }
}

// C <= 64 classes: 16 lanes per training row, 4 elements per lane held in registers (one load each), so a wave has 4
// rows in flight and a 256-thread block 16 — the wave-per-row kernels above wait one random-row latency per row
// (0.62 ms for 1.2 M training rows at C = 47; these: see profiles).  Rows accumulate per 16-lane group in index order,
// the 16 groups of a block are added in fixed order: deterministic.
template <typename T>
__device__ __forceinline__ void row16_load(const T* __restrict__ row, int c, int sub, float (&v)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = sub + 16 * e < c ? load1<T>(row + sub + 16 * e) : -3.402823466e+38f;
}
__device__ __forceinline__ float row16_lse(const float (&v)[4], int c, int sub) {
  float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float se = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (sub + 16 * e < c) se += expf(v[e] - mx);
  se = group_sum<16>(se);
  return mx + logf(se);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_nll_fwd16(const T* __restrict__ logits, int64_t ldl, int c,
                                                        const int64_t* __restrict__ labels,
                                                        const int64_t* __restrict__ idx, int64_t m,
                                                        float* __restrict__ part) {
  constexpr int SLOTS = kThreads / 16;
  __shared__ float red[SLOTS];
  const int sub = threadIdx.x & 15;
  const int slot = threadIdx.x >> 4;
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t j0 = static_cast<int64_t>(blockIdx.x) * per;
  int64_t j1 = j0 + per;
  if (j1 > m) j1 = m;
  float acc = 0.f;
  for (int64_t jb = j0; jb < j1; jb += SLOTS) {        // uniform trip count: the shuffles below need whole waves
    const int64_t j = jb + slot;
    const bool ok = j < j1;
    const int64_t r = ok ? idx[j] : idx[j0];
    const T* row = logits + r * ldl;
    float v[4];
    row16_load<T>(row, c, sub, v);
    const float lse = row16_lse(v, c, sub);
    const int64_t y = labels[r];
    if (ok && sub == 0 && y >= 0 && y < c) acc += lse - load1<T>(row + y);
  }
  if (sub == 0) red[slot] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < SLOTS; ++w) s += red[w];
    part[blockIdx.x] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_nll_bwd16(const T* __restrict__ logits, int64_t ldl, int c,
                                                        const int64_t* __restrict__ labels,
                                                        const int64_t* __restrict__ idx, int64_t m,
                                                        const float* __restrict__ gout, float inv_denom,
                                                        T* __restrict__ dlogits, int64_t ldd) {
  const int sub = threadIdx.x & 15;
  const int64_t gid = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) >> 4;
  const int64_t ng = (static_cast<int64_t>(gridDim.x) * kThreads) >> 4;
  const float scale = gout[0] * inv_denom;
  const int64_t trips = (m + ng - 1) / ng;             // uniform trip count
  for (int64_t t = 0; t < trips; ++t) {
    const int64_t j = t * ng + gid;
    const bool ok = j < m;
    const int64_t r = ok ? idx[j] : idx[0];
    const T* row = logits + r * ldl;
    float v[4];
    row16_load<T>(row, c, sub, v);
    const float lse = row16_lse(v, c, sub);
    const int64_t y = labels[r];
    if (ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = sub + 16 * e;
        // a label outside [0, c) (nn.NLLLoss's ignore_index, an unlabeled node) adds nothing to the loss: zero gradient
        if (k < c) store1<T>(dlogits + r * ldd + k,
                             (y >= 0 && y < c) ? scale * (expf(v[e] - lse) - ((k == y) ? 1.f : 0.f)) : 0.f);
      }
    }
  }
}

// one wave: lane l adds partials l, l + 64, ... in order, then a fixed shuffle tree (deterministic).  (One thread walking
// all partials ran 38 us — longer than the loss kernel it follows.)
__global__ void k_nll_sum(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) s += part[b];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) out[0] = s;
}

// dlogits[row] = scale * (softmax(logits[row]) - onehot(label[row])) on the training rows (the rest of
// dlogits was zeroed by the caller)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_nll_bwd(const T* __restrict__ logits, int64_t ldl, int c,
                                                      const int64_t* __restrict__ labels,
                                                      const int64_t* __restrict__ idx, int64_t m,
                                                      const float* __restrict__ gout, float inv_denom,
                                                      T* __restrict__ dlogits, int64_t ldd) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) >> 6;
  const int64_t nw = (static_cast<int64_t>(gridDim.x) * kThreads) >> 6;
  const float scale = gout[0] * inv_denom;
  for (int64_t j = wid; j < m; j += nw) {
    const int64_t r = idx[j];
    const T* row = logits + r * ldl;
    const float lse = row_lse<T>(row, c, lane);
    const int64_t y = labels[r];
    for (int k = lane; k < c; k += 64) {
      const float p = expf(load1<T>(row + k) - lse);
      store1<T>(dlogits + r * ldd + k, (y >= 0 && y < c) ? scale * (p - ((k == y) ? 1.f : 0.f)) : 0.f);
    }
  }
}

inline int ew_grid(int64_t total_vec) {
  int64_t b = (total_vec + kThreads - 1) / kThreads;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

// blocks for the column-fixed row-walk kernels: enough to fill the chip (8 blocks / CU), never more
// than one unrolled pass of rows per block
inline int rowwalk_grid(int64_t n, int d) {
  const int rpp = kThreads / (d / 4);
  int64_t b = (n + static_cast<int64_t>(rpp) * kRowUnroll - 1) / (static_cast<int64_t>(rpp) * kRowUnroll);
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

inline int stat_blocks(int64_t n) {
  int64_t b = (n + 63) / 64;  // >= 64 rows per block
  if (b > kMaxStatBlocks) b = kMaxStatBlocks;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

int check_ew(const char* fn, int64_t n, int d, int dtype) {
  SGF_REQUIRE(n >= 0 && d >= 1, SGF_E_INVALID, "%s: bad sizes n=%lld d=%d", fn,
              static_cast<long long>(n), d);
  SGF_REQUIRE(d % 4 == 0 && d <= kMaxD, SGF_E_UNSUPPORTED,
              "%s: d=%d unsupported (need d %% 4 == 0 and d <= %d)", fn, d, kMaxD);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "%s: unknown dtype %d", fn,
              dtype);
  return SGF_OK;
}

template <typename T, typename... Args>
int launch_ln_fwd(int lpr, int nc, dim3 grid, hipStream_t st, Args... args) {
#define SGF_LN(L, C)                                                                   \
  if (lpr == L && nc == C) {                                                           \
    hipLaunchKernelGGL((k_ln_fwd<T, L, C>), grid, dim3(kThreads), 0, st, args...);     \
    SGF_LAUNCH_CHECK();                                                                \
    return SGF_OK;                                                                     \
  }
  SGF_LN(16, 1) SGF_LN(32, 1) SGF_LN(64, 1) SGF_LN(64, 2) SGF_LN(64, 3) SGF_LN(64, 4)
#undef SGF_LN
  set_error("sgf_ln_fwd: no kernel for lpr=%d nc=%d", lpr, nc);
  return SGF_E_UNSUPPORTED;
}
template <typename T, typename... Args>
int launch_ln_bwd(int lpr, int nc, dim3 grid, hipStream_t st, Args... args) {
#define SGF_LN(L, C)                                                                   \
  if (lpr == L && nc == C) {                                                           \
    hipLaunchKernelGGL((k_ln_bwd<T, L, C>), grid, dim3(kThreads), 0, st, args...);     \
    SGF_LAUNCH_CHECK();                                                                \
    return SGF_OK;                                                                     \
  }
  SGF_LN(16, 1) SGF_LN(32, 1) SGF_LN(64, 1) SGF_LN(64, 2) SGF_LN(64, 3) SGF_LN(64, 4)
#undef SGF_LN
  set_error("sgf_ln_bwd: no kernel for lpr=%d nc=%d", lpr, nc);
  return SGF_E_UNSUPPORTED;
}

inline int ln_grid(int64_t n, int lpr) {
  const int rpb = kThreads / lpr;
  int64_t b = (n + rpb - 1) / rpb;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}
inline int ln_bwd_grid(int64_t n, int lpr) {
  const int rpb = kThreads / lpr;
  int64_t b = (n + rpb - 1) / rpb;
  if (b > kMaxStatBlocks) b = kMaxStatBlocks;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

// bf16 rows the 16-byte-per-lane kernels take: d a multiple of 8 with at most kThreads chunks, every row 16-byte aligned
// (SGF_EW8=0: the 8-byte kernels, for A/B)
inline bool ew8_rows(int d, std::initializer_list<std::pair<const void*, int64_t>> ops) {
  static EnvInt ew8{"SGF_EW8", 1};
  if (ew8.get() == 0 || d % 8 != 0 || d / 8 > kThreads) return false;
  for (const auto& o : ops)
    if (o.first && (reinterpret_cast<uintptr_t>(o.first) % 16 != 0 || o.second % 8 != 0)) return false;
  return true;
}
constexpr int kEw8Unroll = 4;   // rows in flight per lane (2 / 4 / 8 measured 0.77 / 0.71 / 0.70 ms on the 2R:1W apply; 8 loses on the statistics)
inline int rowwalk8_grid(int64_t n, int d) {
  const int64_t rpb = static_cast<int64_t>(kThreads / (d / 8)) * kEw8Unroll;
  int64_t b = (n + rpb - 1) / rpb;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  return static_cast<int>(b > cap ? cap : (b < 1 ? 1 : b));
}

template <typename T>
int ln_fwd_t(const void* x, int64_t ldx, const void* res, int64_t ldr, float a, float b,
             const float* gamma, const float* beta, int relu, float eps, int64_t n, int d, void* y,
             int64_t ldy, float* mean, float* rstd, hipStream_t st) {
  if (sizeof(T) == 2 && (d == 64 || d == 128 || d == 256 || d == 512) && ldx % 8 == 0 && ldy % 8 == 0 &&
      (!res || ldr % 8 == 0) && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
      (!res || reinterpret_cast<uintptr_t>(res) % 16 == 0)) {
    constexpr int R = 4;
    const int l8 = d / 8;
    const int rpb = kThreads / l8 * R;
    int64_t nb = (n + rpb - 1) / rpb;
    if (nb > static_cast<int64_t>(kNumCU) * 8) nb = static_cast<int64_t>(kNumCU) * 8;
    const dim3 grid(static_cast<unsigned>(nb < 1 ? 1 : nb));
    const uint16_t* x16 = static_cast<const uint16_t*>(x);
    const uint16_t* r16 = static_cast<const uint16_t*>(res);
    uint16_t* y16 = static_cast<uint16_t*>(y);
#define SGF_LN8(L) hipLaunchKernelGGL((k_ln_fwd_bf16x8<L, R>), grid, dim3(kThreads), 0, st, x16, ldx, r16, ldr, a, b, \
                                      gamma, beta, relu, eps, n, y16, ldy, mean, rstd)
    if (l8 == 8) SGF_LN8(8);
    else if (l8 == 16) SGF_LN8(16);
    else if (l8 == 32) SGF_LN8(32);
    else SGF_LN8(64);
#undef SGF_LN8
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const int lpr = lanes_per_row(d);
  const int nc = (d + 4 * lpr - 1) / (4 * lpr);
  return launch_ln_fwd<T>(lpr, nc, dim3(ln_grid(n, lpr)), st, static_cast<const T*>(x), ldx,
                          static_cast<const T*>(res), ldr, a, b, gamma, beta, relu, eps, n, d,
                          static_cast<T*>(y), ldy, mean, rstd);
}

template <typename T>
int ln_bwd_t(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x, int64_t ldx,
             const void* res, int64_t ldr, float a, float b, const float* gamma, int relu,
             const float* mean, const float* rstd, int64_t n, int d, void* dx, int64_t lddx,
             void* dres, int64_t lddres, float* dgamma, float* dbeta, void* ws, hipStream_t st) {
  float* part = static_cast<float*>(ws);
  if (sizeof(T) == 2 && (d == 64 || d == 128 || d == 256 || d == 512) &&
      ew8_rows(d, {{dy, lddy}, {relu ? y : nullptr, ldy}, {gamma ? x : nullptr, ldx}, {gamma ? res : nullptr, ldr}, {dx, lddx},
                   {dres, lddres}})) {
    constexpr int R = 4;                                   // (2 / 4 / 8 rows in flight per lane measured 1.26 / 1.20 / 1.22 ms)
    const int l8 = d / 8;
    const int rpb = kThreads / l8 * R;
    int64_t nb = (n + rpb - 1) / rpb;
    if (nb > kMaxStatBlocks) nb = kMaxStatBlocks;
    const int nblk8 = static_cast<int>(nb < 1 ? 1 : nb);
#define SGF_LNB8R(L, R_)                                                                                               \
  hipLaunchKernelGGL((k_ln_bwd_bf16x8<L, R_>), dim3(nblk8), dim3(kThreads), 0, st, static_cast<const uint16_t*>(dy), lddy, \
                     static_cast<const uint16_t*>(y), ldy, static_cast<const uint16_t*>(x), ldx,                        \
                     static_cast<const uint16_t*>(res), ldr, a, b, gamma, relu, mean, rstd, n, static_cast<uint16_t*>(dx), \
                     lddx, static_cast<uint16_t*>(dres), lddres, part)
#define SGF_LNB8(L) SGF_LNB8R(L, R)
    if (l8 == 8) SGF_LNB8(8);
    else if (l8 == 16) SGF_LNB8(16);
    else if (l8 == 32) SGF_LNB8(32);
    else SGF_LNB8(64);
#undef SGF_LNB8R
#undef SGF_LNB8
    SGF_LAUNCH_CHECK();
    if (gamma != nullptr && (dgamma || dbeta)) {
      hipLaunchKernelGGL(k_sum_partials, dim3((2 * d + 7) / 8), dim3(256), 0, st, part, nblk8, 2 * d, dgamma, dbeta, d);
      SGF_LAUNCH_CHECK();
    }
    return SGF_OK;
  }
  const int lpr = lanes_per_row(d);
  const int nc = (d + 4 * lpr - 1) / (4 * lpr);
  const int nblk = ln_bwd_grid(n, lpr);
  int rc = launch_ln_bwd<T>(lpr, nc, dim3(nblk), st, static_cast<const T*>(dy), lddy,
                            static_cast<const T*>(y), ldy, static_cast<const T*>(x), ldx,
                            static_cast<const T*>(res), ldr, a, b, gamma, relu, mean, rstd, n, d,
                            static_cast<T*>(dx), lddx, static_cast<T*>(dres), lddres, part);
  if (rc != SGF_OK) return rc;
  if (gamma != nullptr && (dgamma || dbeta)) {
    hipLaunchKernelGGL(k_sum_partials, dim3((2 * d + 7) / 8), dim3(256), 0, st, part, nblk,
                       2 * d, dgamma, dbeta, d);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

template <typename F>
int colreduce(F f, int64_t n, int d, float* stats, void* ws, hipStream_t st) {
  const int nblk = stat_blocks(n);
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL((k_colreduce<F>), dim3(nblk), dim3(kThreads), 0, st, f, n, d, part);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3((2 * d + 7) / 8), dim3(256), 0, st, part, nblk, 2 * d,
                     stats, stats + d, d);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" int sgf_ln_fwd(const void* x, int64_t ldx, const void* res, int64_t ldr, float a,
                          float b, const float* gamma, const float* beta, int32_t relu, float eps,
                          int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy, float* mean,
                          float* rstd, void* stream) {
  int rc = check_ew("sgf_ln_fwd", n, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(x && y && (!gamma || (beta && mean && rstd)), SGF_E_INVALID, "sgf_ln_fwd: null pointer");
  SGF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && (!res || ldr % 4 == 0), SGF_E_INVALID,
              "sgf_ln_fwd: leading dims must be multiples of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == SGF_F32)
    return ln_fwd_t<float>(x, ldx, res, ldr, a, b, gamma, beta, relu, eps, n, d, y, ldy, mean, rstd, st);
  return ln_fwd_t<uint16_t>(x, ldx, res, ldr, a, b, gamma, beta, relu, eps, n, d, y, ldy, mean, rstd, st);
}

extern "C" size_t sgf_ln_bwd_workspace_bytes(int64_t n, int32_t d) {
  (void)n;
  if (d < 1) return 0;
  return static_cast<size_t>(kMaxStatBlocks) * 2 * d * sizeof(float);
}

extern "C" int sgf_ln_bwd(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x,
                          int64_t ldx, const void* res, int64_t ldr, float a, float b,
                          const float* gamma, int32_t relu, const float* mean, const float* rstd,
                          int64_t n, int32_t d, int32_t dtype, void* dx, int64_t lddx, void* dres,
                          int64_t lddres, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream) {
  int rc = check_ew("sgf_ln_bwd", n, d, dtype);
  if (rc != SGF_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (gamma && dgamma) SGF_CHECK_HIP(hipMemsetAsync(dgamma, 0, d * sizeof(float), st));
    if (gamma && dbeta) SGF_CHECK_HIP(hipMemsetAsync(dbeta, 0, d * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(dy && dx && (!relu || y) && (!gamma || (x && mean && rstd)), SGF_E_INVALID,
              "sgf_ln_bwd: null pointer");
  SGF_REQUIRE(!gamma || (workspace && workspace_bytes >= sgf_ln_bwd_workspace_bytes(n, d)),
              SGF_E_WORKSPACE, "sgf_ln_bwd: workspace too small");
  if (dtype == SGF_F32)
    return ln_bwd_t<float>(dy, lddy, y, ldy, x, ldx, res, ldr, a, b, gamma, relu, mean, rstd, n, d,
                           dx, lddx, dres, lddres, dgamma, dbeta, workspace, st);
  return ln_bwd_t<uint16_t>(dy, lddy, y, ldy, x, ldx, res, ldr, a, b, gamma, relu, mean, rstd, n, d,
                            dx, lddx, dres, lddres, dgamma, dbeta, workspace, st);
}

extern "C" size_t sgf_colstats_workspace_bytes(int64_t n, int32_t d) {
  (void)n;
  if (d < 1) return 0;
  return static_cast<size_t>(kMaxStatBlocks) * 2 * d * sizeof(float);
}

extern "C" int sgf_colstats(const void* x, int64_t ldx, const float* shift, int64_t n, int32_t d,
                            int32_t dtype, float* stats, void* workspace, size_t workspace_bytes,
                            void* stream) {
  int rc = check_ew("sgf_colstats", n, d, dtype);
  if (rc != SGF_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  SGF_REQUIRE(stats, SGF_E_INVALID, "sgf_colstats: null stats");
  if (n == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * d * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(x && ldx % 4 == 0, SGF_E_INVALID, "sgf_colstats: bad x / ldx");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_colstats_workspace_bytes(n, d), SGF_E_WORKSPACE,
              "sgf_colstats: workspace too small");
  if (dtype == SGF_F32)
    return colreduce(ColStatsF<float>{static_cast<const float*>(x), ldx, shift}, n, d, stats,
                     workspace, st);
  return colreduce(ColStatsF<uint16_t>{static_cast<const uint16_t*>(x), ldx, shift}, n, d, stats,
                   workspace, st);
}

// BatchNorm's per-column bookkeeping between its two passes, in ONE launch (it was ~14 tiny ATen launches per BatchNorm
// and step): mean / biased variance from the shifted sums, rstd, and the running-statistics update of nn.BatchNorm1d.
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ sums, const float* __restrict__ shift,
                                                     float inv_n, float unbias, float eps, float momentum,
                                                     float* __restrict__ rmean, float* __restrict__ rvar, int d,
                                                     float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float m1 = sums[c] * inv_n;
  const float mu = (shift ? shift[c] : 0.f) + m1;
  float var = sums[d + c] * inv_n - m1 * m1;
  var = var > 0.f ? var : 0.f;
  mean[c] = mu;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (rmean) rmean[c] = rmean[c] * (1.0f - momentum) + momentum * mu;
  if (rvar) rvar[c] = rvar[c] * (1.0f - momentum) + momentum * (var * unbias);
}

extern "C" int sgf_bn_finalize(const float* sums, const float* shift, double n_total, float eps, float momentum,
                               float* running_mean, float* running_var, int32_t d, float* mean, float* rstd, void* stream) {
  SGF_REQUIRE(d >= 1 && n_total >= 0, SGF_E_INVALID, "sgf_bn_finalize: bad sizes");
  SGF_REQUIRE(sums && mean && rstd, SGF_E_INVALID, "sgf_bn_finalize: null pointer");
  const double nn = n_total > 1.0 ? n_total : 1.0;
  const float unbias = static_cast<float>(n_total / (n_total - 1.0 > 1.0 ? n_total - 1.0 : 1.0));
  hipLaunchKernelGGL(k_bn_finalize, dim3((d + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums, shift,
                     static_cast<float>(1.0 / nn), unbias, eps, momentum, running_mean, running_var, d, mean, rstd);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_bn_apply(const void* x, int64_t ldx, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, const void* res, int64_t ldr,
                            int32_t relu, int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy,
                            void* stream) {
  int rc = check_ew("sgf_bn_apply", n, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(x && y && mean && rstd, SGF_E_INVALID, "sgf_bn_apply: null pointer");
  SGF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && (!res || ldr % 4 == 0), SGF_E_INVALID,
              "sgf_bn_apply: leading dims must be multiples of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BnParams p{mean, rstd, gamma, beta};
  if (dtype == SGF_BF16 && ew8_rows(d, {{x, ldx}, {res, ldr}, {y, ldy}})) {
    const dim3 g8(rowwalk8_grid(n, d));
#define SGF_BNA(RES_, U_)                                                                                             \
  hipLaunchKernelGGL((k_bn_apply_bf16x8<RES_, U_>), g8, dim3(kThreads), 0, st, static_cast<const uint16_t*>(x), ldx, p, \
                     static_cast<const uint16_t*>(res), ldr, relu, n, d, static_cast<uint16_t*>(y), ldy)
    if (res) SGF_BNA(true, kEw8Unroll);
    else SGF_BNA(false, kEw8Unroll);
#undef SGF_BNA
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const dim3 grid(rowwalk_grid(n, d));
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_bn_apply<float>), grid, dim3(kThreads), 0, st,
                       static_cast<const float*>(x), ldx, p, static_cast<const float*>(res), ldr,
                       relu, n, d, static_cast<float*>(y), ldy);
  else
    hipLaunchKernelGGL((k_bn_apply<uint16_t>), grid, dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(x), ldx, p, static_cast<const uint16_t*>(res),
                       ldr, relu, n, d, static_cast<uint16_t*>(y), ldy);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_bn_bwd_stats2(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* x, int64_t ldx,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 int32_t relu, int64_t n, int32_t d, int32_t dtype, float* stats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  int rc = check_ew("sgf_bn_bwd_stats2", n, d, dtype);
  if (rc != SGF_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  SGF_REQUIRE(stats, SGF_E_INVALID, "sgf_bn_bwd_stats2: null stats");
  if (n == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * d * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(dy && x && mean && rstd && lddy % 4 == 0 && ldx % 4 == 0 && (!dy2 || lddy2 % 4 == 0), SGF_E_INVALID,
              "sgf_bn_bwd_stats2: bad pointer / ld");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_colstats_workspace_bytes(n, d), SGF_E_WORKSPACE,
              "sgf_bn_bwd_stats2: workspace too small");
  const BnParams p{mean, rstd, gamma, beta};
  if (dtype == SGF_BF16 && ew8_rows(d, {{dy, lddy}, {dy2, lddy2}, {x, ldx}})) {
    const int nblk = stat_blocks(n);
    float* part = static_cast<float*>(workspace);
#define SGF_BNS(TWO_, U_)                                                                                             \
  hipLaunchKernelGGL((k_bn_bwd_stats_bf16x8<TWO_, U_>), dim3(nblk), dim3(kThreads), 0, st, static_cast<const uint16_t*>(dy), \
                     lddy, static_cast<const uint16_t*>(dy2), lddy2, static_cast<const uint16_t*>(x), ldx, p, relu, n, d, part)
    if (dy2) SGF_BNS(true, kEw8Unroll);
    else SGF_BNS(false, kEw8Unroll);
#undef SGF_BNS
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3((2 * d + 7) / 8), dim3(256), 0, st, part, nblk, 2 * d, stats, stats + d, d);
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  if (dtype == SGF_F32)
    return colreduce(BnBwdStatsF<float>{static_cast<const float*>(dy), lddy, static_cast<const float*>(x), ldx, p, relu,
                                        static_cast<const float*>(dy2), lddy2},
                     n, d, stats, workspace, st);
  return colreduce(BnBwdStatsF<uint16_t>{static_cast<const uint16_t*>(dy), lddy, static_cast<const uint16_t*>(x), ldx, p,
                                         relu, static_cast<const uint16_t*>(dy2), lddy2},
                   n, d, stats, workspace, st);
}

extern "C" int sgf_bn_bwd_stats(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                                const float* mean, const float* rstd, const float* gamma,
                                const float* beta, int32_t relu, int64_t n, int32_t d,
                                int32_t dtype, float* stats, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return sgf_bn_bwd_stats2(dy, lddy, nullptr, 0, x, ldx, mean, rstd, gamma, beta, relu, n, d, dtype, stats, workspace,
                           workspace_bytes, stream);
}

extern "C" int sgf_bn_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                                const float* mean, const float* rstd, const float* gamma,
                                const float* beta, int32_t relu, const float* stats, float inv_n,
                                int32_t training, int64_t n, int32_t d, int32_t dtype, void* dx,
                                int64_t lddx, void* stream) {
  int rc = check_ew("sgf_bn_bwd_apply", n, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(dy && x && dx && mean && rstd && (!training || stats), SGF_E_INVALID,
              "sgf_bn_bwd_apply: null pointer");
  SGF_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, SGF_E_INVALID,
              "sgf_bn_bwd_apply: leading dims must be multiples of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const BnParams p{mean, rstd, gamma, beta};
  if (dtype == SGF_BF16 && ew8_rows(d, {{dy, lddy}, {x, ldx}, {dx, lddx}})) {
#define SGF_BNB(U_)                                                                                                   \
  hipLaunchKernelGGL((k_bn_bwd_apply_bf16x8<U_>), dim3(rowwalk8_grid(n, d)), dim3(kThreads), 0, st,                     \
                     static_cast<const uint16_t*>(dy), lddy, static_cast<const uint16_t*>(x), ldx, p, relu, stats, inv_n, \
                     training, n, d, static_cast<uint16_t*>(dx), lddx)
    SGF_BNB(kEw8Unroll);
#undef SGF_BNB
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const dim3 grid(rowwalk_grid(n, d));
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_bn_bwd_apply<float>), grid, dim3(kThreads), 0, st,
                       static_cast<const float*>(dy), lddy, static_cast<const float*>(x), ldx, p,
                       relu, stats, inv_n, training, n, d, static_cast<float*>(dx), lddx);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<uint16_t>), grid, dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(dy), lddy, static_cast<const uint16_t*>(x), ldx,
                       p, relu, stats, inv_n, training, n, d, static_cast<uint16_t*>(dx), lddx);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_axpby(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
                         int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy, void* stream) {
  int rc = check_ew("sgf_axpby", n, d, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(x1 && x2 && y && ld1 % 4 == 0 && ld2 % 4 == 0 && ldy % 4 == 0, SGF_E_INVALID,
              "sgf_axpby: bad pointer / ld");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(ew_grid(n * (d / 4)));
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_axpby<float>), grid, dim3(kThreads), 0, st, static_cast<const float*>(x1),
                       ld1, a, static_cast<const float*>(x2), ld2, b, n, d, static_cast<float*>(y), ldy);
  else
    hipLaunchKernelGGL((k_axpby<uint16_t>), grid, dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(x1), ld1, a, static_cast<const uint16_t*>(x2),
                       ld2, b, n, d, static_cast<uint16_t*>(y), ldy);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_gather_rows(const void* src, int64_t lds, int32_t src_dtype, int64_t n_src, const void* idx,
                               int32_t idx_is_int64, int64_t n_out, int32_t d, void* dst, int64_t ldd,
                               int32_t dst_dtype, void* stream) {
  SGF_REQUIRE(n_out >= 0 && d >= 0 && n_src >= 0, SGF_E_INVALID, "sgf_gather_rows: negative size");
  SGF_REQUIRE((src_dtype == SGF_F32 || src_dtype == SGF_BF16) && (dst_dtype == SGF_F32 || dst_dtype == SGF_BF16),
              SGF_E_INVALID, "sgf_gather_rows: unknown dtype");
  if (n_out == 0 || d == 0) return SGF_OK;
  SGF_REQUIRE(src && idx && dst && lds >= d && ldd >= d, SGF_E_INVALID, "sgf_gather_rows: bad pointer / ld");
  const size_t es = src_dtype == SGF_BF16 ? 2 : 4, ed = dst_dtype == SGF_BF16 ? 2 : 4;
  const int vec = d % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && reinterpret_cast<uintptr_t>(src) % (4 * es) == 0 &&
                  reinterpret_cast<uintptr_t>(dst) % (4 * ed) == 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(ew_grid(vec ? n_out * (d / 4) : n_out * d));
#define SGF_GATHER(TS_, TD_, TI_)                                                                            \
  hipLaunchKernelGGL((k_gather_rows<TS_, TD_, TI_>), grid, dim3(kThreads), 0, st, static_cast<const TS_*>(src), \
                     lds, static_cast<const TI_*>(idx), n_out, d, n_src, static_cast<TD_*>(dst), ldd, vec)
#define SGF_GATHER_I(TS_, TD_) \
  do { if (idx_is_int64) SGF_GATHER(TS_, TD_, int64_t); else SGF_GATHER(TS_, TD_, int32_t); } while (0)
  if (src_dtype == SGF_F32 && dst_dtype == SGF_F32) SGF_GATHER_I(float, float);
  else if (src_dtype == SGF_F32) SGF_GATHER_I(float, uint16_t);
  else if (dst_dtype == SGF_F32) SGF_GATHER_I(uint16_t, float);
  else SGF_GATHER_I(uint16_t, uint16_t);
#undef SGF_GATHER_I
#undef SGF_GATHER
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// dst[i, :d] = src[idx ? idx[i] : i, :d] (storage change allowed), dst[i, d:d_pad] = 0 — the module-entry copy of node
// features whose width is not a multiple of 4 (pokec: f = 65, Cora: f = 1433): one scalar-granular pass ONCE per feature
// tensor, after which every kernel sees 8- / 16-byte aligned rows.  Lanes walk the destination row, so stores coalesce and
// the (unaligned) source rows are read contiguously.
namespace sgf {
namespace {
template <typename TS, typename TD, typename TI>
__global__ __launch_bounds__(kThreads) void k_pad_rows(const TS* __restrict__ src, int64_t lds, const TI* __restrict__ idx,
                                                       int64_t n_src, int64_t n_out, int d, int d_pad, TD* __restrict__ dst,
                                                       int64_t ldd) {
  const int64_t total = n_out * d_pad;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t row = i / d_pad;
    const int col = static_cast<int>(i % d_pad);
    float v = 0.f;
    if (col < d) {
      const int64_t r = idx ? static_cast<int64_t>(idx[row]) : row;
      if (r >= 0 && r < n_src) v = load1<TS>(src + r * lds + col);
    }
    store1<TD>(dst + row * ldd + col, v);
  }
}
}  // namespace
}  // namespace sgf

extern "C" int sgf_pad_rows(const void* src, int64_t lds, int32_t src_dtype, int64_t n_src, const void* idx,
                            int32_t idx_is_int64, int64_t n_out, int32_t d, int32_t d_pad, void* dst, int64_t ldd,
                            int32_t dst_dtype, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(n_out >= 0 && d >= 0 && d_pad >= d && n_src >= 0, SGF_E_INVALID, "sgf_pad_rows: bad sizes");
  SGF_REQUIRE((src_dtype == SGF_F32 || src_dtype == SGF_BF16) && (dst_dtype == SGF_F32 || dst_dtype == SGF_BF16),
              SGF_E_INVALID, "sgf_pad_rows: unknown dtype");
  if (n_out == 0 || d_pad == 0) return SGF_OK;
  SGF_REQUIRE(src && dst && lds >= d && ldd >= d_pad, SGF_E_INVALID, "sgf_pad_rows: bad pointer / ld");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(ew_grid(n_out * d_pad));
#define SGF_PAD(TS_, TD_, TI_)                                                                                      \
  hipLaunchKernelGGL((k_pad_rows<TS_, TD_, TI_>), grid, dim3(kThreads), 0, st, static_cast<const TS_*>(src), lds, \
                     static_cast<const TI_*>(idx), n_src, n_out, d, d_pad, static_cast<TD_*>(dst), ldd)
#define SGF_PAD_I(TS_, TD_) \
  do { if (idx && idx_is_int64) SGF_PAD(TS_, TD_, int64_t); else SGF_PAD(TS_, TD_, int32_t); } while (0)
  if (src_dtype == SGF_F32 && dst_dtype == SGF_F32) SGF_PAD_I(float, float);
  else if (src_dtype == SGF_F32) SGF_PAD_I(float, uint16_t);
  else if (dst_dtype == SGF_F32) SGF_PAD_I(uint16_t, float);
  else SGF_PAD_I(uint16_t, uint16_t);
#undef SGF_PAD_I
#undef SGF_PAD
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_sum_n(const void* const* xs, const int64_t* lds, int32_t k, int64_t n, int32_t d,
                         int32_t dtype, void* y, int64_t ldy, void* stream) {
  int rc = check_ew("sgf_sum_n", n, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(k >= 1 && k <= 8 && xs && lds && y, SGF_E_INVALID, "sgf_sum_n: need 1 <= k <= 8 operands");
  if (n == 0) return SGF_OK;
  SumArgs a{};
  a.k = k;
  for (int j = 0; j < k; ++j) {
    SGF_REQUIRE(xs[j] && lds[j] % 4 == 0, SGF_E_INVALID, "sgf_sum_n: bad operand %d", j);
    a.x[j] = xs[j];
    a.ld[j] = lds[j];
  }
  SGF_REQUIRE(ldy % 4 == 0, SGF_E_INVALID, "sgf_sum_n: bad ldy");
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool wide = dtype == SGF_BF16 && d % 8 == 0 && ldy % 8 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  for (int j = 0; j < k && wide; ++j) wide = lds[j] % 8 == 0 && reinterpret_cast<uintptr_t>(xs[j]) % 16 == 0;
  if (wide) {
    const dim3 grid8(ew_grid(n * (d / 8)));
    uint16_t* y16 = static_cast<uint16_t*>(y);
    switch (k) {
#define SGF_SUM_CASE(K_) case K_: hipLaunchKernelGGL((k_sum_n_bf16x8<K_>), grid8, dim3(kThreads), 0, st, a, n, d, y16, ldy); break;
      SGF_SUM_CASE(1) SGF_SUM_CASE(2) SGF_SUM_CASE(3) SGF_SUM_CASE(4) SGF_SUM_CASE(5) SGF_SUM_CASE(6) SGF_SUM_CASE(7)
      SGF_SUM_CASE(8)
#undef SGF_SUM_CASE
    }
    SGF_LAUNCH_CHECK();
    return SGF_OK;
  }
  const dim3 grid(ew_grid(n * (d / 4)));
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_sum_n<float>), grid, dim3(kThreads), 0, st, a, n, d, static_cast<float*>(y), ldy);
  else
    hipLaunchKernelGGL((k_sum_n<uint16_t>), grid, dim3(kThreads), 0, st, a, n, d,
                       static_cast<uint16_t*>(y), ldy);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_dropout(const void* x, int64_t ldx, const void* res, int64_t ldr, float p,
                           uint64_t seed, int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy,
                           void* stream) {
  int rc = check_ew("sgf_dropout", n, d, dtype);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(p >= 0.f && p <= 1.f, SGF_E_INVALID, "sgf_dropout: p must be in [0, 1] (p=%f)", p);
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(x && y && ldx % 4 == 0 && ldy % 4 == 0 && (!res || ldr % 4 == 0), SGF_E_INVALID,
              "sgf_dropout: bad pointer / ld");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
  const dim3 grid(ew_grid(n * (d / 4)));
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_dropout<float>), grid, dim3(kThreads), 0, st, static_cast<const float*>(x), ldx,
                       static_cast<const float*>(res), ldr, p, scale, seed, n, d, static_cast<float*>(y), ldy);
  else
    hipLaunchKernelGGL((k_dropout<uint16_t>), grid, dim3(kThreads), 0, st, static_cast<const uint16_t*>(x),
                       ldx, static_cast<const uint16_t*>(res), ldr, p, scale, seed, n, d,
                       static_cast<uint16_t*>(y), ldy);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" size_t sgf_nll_workspace_bytes(int64_t m) {
  (void)m;
  return static_cast<size_t>(kNllMaxBlocks) * sizeof(float);
}

extern "C" int sgf_nll_fwd(const void* logits, int64_t ldl, int64_t n, int32_t c, int32_t dtype,
                           const int64_t* labels, const int64_t* idx, int64_t m, float* loss_sum,
                           void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(n >= 0 && m >= 0 && c >= 1 && ldl >= c, SGF_E_INVALID, "sgf_nll_fwd: bad sizes");
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "sgf_nll_fwd: unknown dtype");
  SGF_REQUIRE(loss_sum, SGF_E_INVALID, "sgf_nll_fwd: null loss_sum");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (m == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(logits && labels && idx, SGF_E_INVALID, "sgf_nll_fwd: null pointer");
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_nll_workspace_bytes(m), SGF_E_WORKSPACE,
              "sgf_nll_fwd: workspace too small");
  int64_t b = (m + 63) / 64;
  if (b > kNllMaxBlocks) b = kNllMaxBlocks;
  const int nblk = static_cast<int>(b);
  float* part = static_cast<float*>(workspace);
  if (c <= 64) {
    if (dtype == SGF_F32)
      hipLaunchKernelGGL((k_nll_fwd16<float>), dim3(nblk), dim3(kThreads), 0, st,
                         static_cast<const float*>(logits), ldl, c, labels, idx, m, part);
    else
      hipLaunchKernelGGL((k_nll_fwd16<uint16_t>), dim3(nblk), dim3(kThreads), 0, st,
                         static_cast<const uint16_t*>(logits), ldl, c, labels, idx, m, part);
  } else if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_nll_fwd<float>), dim3(nblk), dim3(kThreads), 0, st,
                       static_cast<const float*>(logits), ldl, c, labels, idx, m, part);
  else
    hipLaunchKernelGGL((k_nll_fwd<uint16_t>), dim3(nblk), dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(logits), ldl, c, labels, idx, m, part);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nll_sum, dim3(1), dim3(64), 0, st, part, nblk, loss_sum);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_nll_bwd(const void* logits, int64_t ldl, int64_t n, int32_t c, int32_t dtype,
                           const int64_t* labels, const int64_t* idx, int64_t m, const float* gout,
                           float inv_denom, void* dlogits, int64_t ldd, void* stream) {
  SGF_REQUIRE(n >= 0 && m >= 0 && c >= 1 && ldl >= c && ldd >= c, SGF_E_INVALID, "sgf_nll_bwd: bad sizes");
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "sgf_nll_bwd: unknown dtype");
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(dlogits && gout, SGF_E_INVALID, "sgf_nll_bwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = dtype == SGF_BF16 ? 2 : 4;
  SGF_CHECK_HIP(hipMemset2DAsync(dlogits, static_cast<size_t>(ldd) * esz, 0, static_cast<size_t>(c) * esz,
                                 static_cast<size_t>(n), st));
  if (m == 0) return SGF_OK;
  SGF_REQUIRE(logits && labels && idx, SGF_E_INVALID, "sgf_nll_bwd: null pointer");
  int64_t b = (m + 3) / 4;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (c <= 64) {
    int64_t b16 = (m + 15) / 16;
    if (b16 > cap) b16 = cap;
    if (dtype == SGF_F32)
      hipLaunchKernelGGL((k_nll_bwd16<float>), dim3(static_cast<unsigned>(b16)), dim3(kThreads), 0, st,
                         static_cast<const float*>(logits), ldl, c, labels, idx, m, gout, inv_denom,
                         static_cast<float*>(dlogits), ldd);
    else
      hipLaunchKernelGGL((k_nll_bwd16<uint16_t>), dim3(static_cast<unsigned>(b16)), dim3(kThreads), 0, st,
                         static_cast<const uint16_t*>(logits), ldl, c, labels, idx, m, gout, inv_denom,
                         static_cast<uint16_t*>(dlogits), ldd);
  } else if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_nll_bwd<float>), dim3(static_cast<unsigned>(b)), dim3(kThreads), 0, st,
                       static_cast<const float*>(logits), ldl, c, labels, idx, m, gout, inv_denom,
                       static_cast<float*>(dlogits), ldd);
  else
    hipLaunchKernelGGL((k_nll_bwd<uint16_t>), dim3(static_cast<unsigned>(b)), dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(logits), ldl, c, labels, idx, m, gout, inv_denom,
                       static_cast<uint16_t*>(dlogits), ldd);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" size_t sgf_colsum_workspace_bytes(int64_t n, int32_t d) {
  (void)n;
  if (d < 1) return 0;
  return static_cast<size_t>(kMaxStatBlocks) * d * sizeof(float);
}

extern "C" int sgf_colsum(const void* x, int64_t ldx, int64_t n, int32_t d, int32_t dtype,
                          float* out, void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(n >= 0 && d >= 1 && d <= kThreads, SGF_E_UNSUPPORTED,
              "sgf_colsum: need 1 <= d <= %d (d=%d)", kThreads, d);
  SGF_REQUIRE(dtype == SGF_F32 || dtype == SGF_BF16, SGF_E_INVALID, "sgf_colsum: unknown dtype");
  SGF_REQUIRE(out, SGF_E_INVALID, "sgf_colsum: null out");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(out, 0, d * sizeof(float), st));
    return SGF_OK;
  }
  SGF_REQUIRE(x && workspace && workspace_bytes >= sgf_colsum_workspace_bytes(n, d), SGF_E_WORKSPACE,
              "sgf_colsum: null x or workspace too small");
  const int nblk = stat_blocks(n);
  float* part = static_cast<float*>(workspace);
  if (dtype == SGF_F32)
    hipLaunchKernelGGL((k_colsum_any<float>), dim3(nblk), dim3(kThreads), 0, st,
                       static_cast<const float*>(x), ldx, n, d, part);
  else
    hipLaunchKernelGGL((k_colsum_any<uint16_t>), dim3(nblk), dim3(kThreads), 0, st,
                       static_cast<const uint16_t*>(x), ldx, n, d, part);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3((d + 7) / 8), dim3(256), 0, st, part, nblk, d, out,
                     static_cast<float*>(nullptr), d);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
