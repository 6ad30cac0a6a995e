// head.hip — T7: the two branches fused into ONE residual + MLP kernel.
//
// Reference (large/ours.py:265-276, `aggregate == 'add'`):
//     x = graph_weight * x2 + (1 - graph_weight) * x1          # [N, d] elementwise
//     x = self.fc(x)                                           # [N, d] -> [N, C], C = 7 ... 47 classes
// As separate ops that is an axpby pass (read 2, write 1 [N, d] tensors), a GEMM that reads the result
// again, and — under autograd — the same again backwards plus two scaled copies of the [N, d] gradient.
//
//   sgf_combine_fc_fwd : logits = (a x1 + b x2) W^T + bias        one pass over x1, x2; fp32 logits
//   sgf_combine_fc_bwd : dX = dlogits W;  dx1 = a dX, dx2 = b dX  one pass over dlogits, two stores
// (dW = a dlogits^T x1 + b dlogits^T x2 and db are reductions over all nodes: sgf_gram, as for every
// other Linear of the path.)
//
// gfx950 mapping (bf16 activations — BASELINE.json config 3; fp32 runs keep sgf_axpby + the library GEMM):
//   * [N, d] x [d, C] with C <= 64 is GEMM-shaped but skinny: v_mfma_f32_32x32x16_bf16, a wave owns 32 rows
//     and both 32-class tiles, so one A fragment feeds two MFMAs; W (bf16, <= 64 x 256 = 32 KiB) lives in LDS
//     for the whole block, rows padded by 16 B so the 16 lanes of a ds_read_b128 group hit 16 distinct slots.
//   * the A operand needs, per lane, 8 consecutive k of ONE row (lane l: row l & 31, k = 8 (l >> 5) ...):
//     16 contiguous bytes of x1 and of x2 — read straight from global memory (each 128-byte line of a row is
//     consumed by 4 consecutive k-steps of the same wave), combined in fp32 and rounded to bf16 ONCE — the same
//     rounding point as the unfused axpby.  No LDS staging of activations, no barrier in the row loop.
//   * backward: A = dlogits (fp32 in memory, rounded to bf16 in registers: the rounding the unfused path
//     applies when it casts the logits gradient to the activation dtype), B = W^T fragments from a transposed
//     LDS copy; a wave keeps the 32 x 256 result in 8 accumulator tiles and stores both scaled copies.
// HBM-bound: forward 2 d s + 4 C bytes per node, backward 4 C + 2 d s.
#include "common.h"

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kHeadThreads = 256;                 // 4 waves, 32 rows each per trip
constexpr int kMaxClasses = 64;                   // two 32-class tiles

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return static_cast<uint32_t>(f32_to_bf16(lo)) | (static_cast<uint32_t>(f32_to_bf16(hi)) << 16);
}
__device__ __forceinline__ float lo16(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

union Frag {
  bf16x8 v;
  uint4 u;
};

// logits[N, C] = (a x1 + b x2)[N, d] W[C, d]^T + bias.   d % 16 == 0, d <= 256, C <= 64.
template <int DP>   // d rounded up to 64 / 128 / 256 (LDS pitch, k-steps)
__global__ __launch_bounds__(kHeadThreads) void k_head_fwd_bf16(
    const uint16_t* __restrict__ x1, int64_t ld1, float a, const uint16_t* __restrict__ x2, int64_t ld2, float b,
    const float* __restrict__ w, const float* __restrict__ bias, int64_t n, int d, int c,
    float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ rmap) {
  constexpr int PITCH = DP + 8;                    // bf16 elements per LDS row of W (+16 B)
  __shared__ __align__(16) uint16_t wl[kMaxClasses * PITCH];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  // W -> bf16 in LDS, classes >= c and columns >= d zero
  for (int i = threadIdx.x; i < kMaxClasses * (DP / 4); i += kHeadThreads) {
    const int row = i / (DP / 4), col = (i % (DP / 4)) * 4;
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < c && col < d) f = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(row) * d + col);
    uint2 pk;
    pk.x = pack2(f.x, f.y);
    pk.y = pack2(f.z, f.w);
    *reinterpret_cast<uint2*>(&wl[row * PITCH + col]) = pk;
  }
  __syncthreads();
  const int i31 = lane & 31, hi = lane >> 5;
  const int ks = d / 16;
  const float bias0 = i31 < c ? bias[i31] : 0.f;
  const float bias1 = 32 + i31 < c ? bias[32 + i31] : 0.f;
  const int64_t ntiles = (n + 31) / 32;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wid; tile < ntiles; tile += static_cast<int64_t>(gridDim.x) * 4) {
    int64_t row = tile * 32 + i31;
    if (row >= n) row = n - 1;                     // clamp: loads stay in bounds, stores are masked
    const uint16_t* p1 = x1 + row * ld1 + 8 * hi;
    const uint16_t* p2 = x2 + row * ld2 + 8 * hi;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    // k-steps in batches of 4 (8 loads), the NEXT batch requested before the current one is multiplied
    constexpr int NB = DP / 64;
    uint4 r1[2][4], r2[2][4];
    auto fetch = [&](int bi, uint4 (&q1)[4], uint4 (&q2)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = 4 * bi + u < ks ? 4 * bi + u : ks - 1;
        q1[u] = *reinterpret_cast<const uint4*>(p1 + 16 * s);
        q2[u] = *reinterpret_cast<const uint4*>(p2 + 16 * s);
      }
    };
    fetch(0, r1[0], r2[0]);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (4 * bi < ks) {
        if (bi + 1 < NB && 4 * (bi + 1) < ks) fetch(bi + 1, r1[(bi + 1) & 1], r2[(bi + 1) & 1]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (4 * bi + u < ks) {
            const uint4 v1 = r1[bi & 1][u], v2 = r2[bi & 1][u];
            Frag fa;
            fa.u.x = pack2(a * lo16(v1.x) + b * lo16(v2.x), a * hi16(v1.x) + b * hi16(v2.x));
            fa.u.y = pack2(a * lo16(v1.y) + b * lo16(v2.y), a * hi16(v1.y) + b * hi16(v2.y));
            fa.u.z = pack2(a * lo16(v1.z) + b * lo16(v2.z), a * hi16(v1.z) + b * hi16(v2.z));
            fa.u.w = pack2(a * lo16(v1.w) + b * lo16(v2.w), a * hi16(v1.w) + b * hi16(v2.w));
            const int k0 = 16 * (4 * bi + u) + 8 * hi;
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&wl[i31 * PITCH + k0]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&wl[(32 + i31) * PITCH + k0]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, b0, acc0, 0, 0, 0);
            if (c > 32) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, b1, acc1, 0, 0, 0);
          }
        }
      }
    }
    // C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    // rmap (nullable): row j of the product is row rmap[j] of `logits` — the module's un-permutation of a re-ordered graph
    // done by the stores (lane i31 holds the target of tile row i31; the owner of a register's row is fetched by bpermute)
    const int mrow = rmap ? rmap[row] : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = mfma32_row(r, lane);
      const int64_t orow = tile * 32 + rl;
      const int64_t trow = rmap ? static_cast<int64_t>(__builtin_amdgcn_ds_bpermute(rl << 2, mrow)) : orow;
      if (orow < n) {
        if (i31 < c) logits[trow * ldl + i31] = acc0[r] + bias0;
        if (32 + i31 < c) logits[trow * ldl + 32 + i31] = acc1[r] + bias1;
      }
    }
  }
}

// dx1 = a (dlogits W), dx2 = b (dlogits W).   d % 32 == 0, d <= 256, C <= 64.
// 2 [n, d] tensors written, almost nothing read: a store kernel.  Product D[node][feature] (lane = feature, registers =
// 16 nodes); each wave parks its whole 32 x d result tile in a private LDS patch and writes it back out as complete
// rows, two rows (1 KiB) per store instruction.  (Storing straight from the accumulator layout — 8- or 16-byte pieces
// of 32 different rows per instruction — measured 1.2-1.5 ms at N = 2.45 M, d = 256: bound by partial-line writes.)
constexpr int kHeadBwdThreads = 512;             // 8 waves, one 32-node tile each per trip

template <int DP>
__global__ __launch_bounds__(kHeadBwdThreads) void k_head_bwd_bf16(
    const float* __restrict__ dl, int64_t lddl, const float* __restrict__ w, int64_t n, int d, int c, float a, float b,
    uint16_t* __restrict__ dx1, int64_t ld1, uint16_t* __restrict__ dx2, int64_t ld2, const int32_t* __restrict__ rmap,
    uint16_t* __restrict__ gout, int64_t ldgo) {
  constexpr int PITCH = kMaxClasses + 8;           // bf16 elements per LDS row of W^T (+16 B)
  constexpr int HW = DP >= 128 ? DP / 2 : DP;      // features per pass: the tile leaves in two column halves
  constexpr int NH = DP / HW;                      // passes
  constexpr int NT = HW / 32;                      // 32-feature strips per pass
  constexpr int PROW = HW * 2 + 16;                // bytes per row of a wave's result patch (16-byte slots rotate)
  constexpr int LPR = HW / 8;                      // lanes per row in the read-back (16 B each)
  constexpr int RPI = 64 / LPR;                    // rows per read-back instruction
  constexpr int KSMAX = kMaxClasses / 16;
  __shared__ __align__(16) uint16_t wt[DP * PITCH];   // wt[feature][class]
  __shared__ __align__(16) unsigned char patch_all[(kHeadBwdThreads / 64) * 32 * PROW];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < DP * kMaxClasses; i += kHeadBwdThreads) {
    const int f = i % DP, cl = i / DP;             // consecutive threads read consecutive features of one class
    const float v = (cl < c && f < d) ? w[static_cast<int64_t>(cl) * d + f] : 0.f;
    wt[f * PITCH + cl] = f32_to_bf16(v);
  }
  __syncthreads();
  const int i31 = lane & 31, hi = lane >> 5;
  unsigned char* const patch = patch_all + wid * 32 * PROW;
  unsigned char* const pw = patch + 4 * hi * PROW + 2 * i31;                       // + row * PROW + 64 t
  const unsigned char* const pr = patch + (lane / LPR) * PROW + 16 * (lane % LPR); // + RPI j * PROW
  const int ks = (c + 15) / 16;
  const int64_t ntiles = (n + 31) / 32;
  constexpr int W = kHeadBwdThreads / 64;

  // A operand of one tile: node i31, classes 16 s + 8 hi .. + 7 (rows of dlogits are C floats: unaligned, scalar loads).
  // Software pipeline over the trips of a wave — the row index (through rmap) is requested TWO tiles ahead, the raw fp32
  // classes ONE tile ahead, and they are packed to bf16 only when their tile starts: every load is unconditional (clamped
  // address, zero selected afterwards) and nothing is waited for in the trip that requests it.  (The first version packed
  // inside the prefetch: four `s_waitcnt vmcnt(0)` per trip, each of them also a wait for the previous tile's 32 row stores —
  // 12 us per tile, 0.64 of the copy rate.)
  struct Raw {
    float g[KSMAX][8];
  };
  // (32-bit and widened only where it is used: a conversion next to the load would be a wait for it)
  auto row_of = [&](int64_t t) -> int32_t {
    int64_t row = t * 32 + i31;
    if (row >= n) row = n - 1;
    if (t >= ntiles) return 0;
    return rmap ? rmap[row] : static_cast<int32_t>(row);     // row j of the product reads row rmap[j] of dlogits
  };
  auto load_raw = [&](int32_t row, Raw& r) {
    const float* pg = dl + static_cast<int64_t>(row) * lddl;
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) {
      if (s < ks) {
        const int k0 = 16 * s + 8 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.g[s][j] = pg[k0 + j < c ? k0 + j : c - 1];
      }
    }
  };
  auto pack_frags = [&](const Raw& r, Frag (&fa)[KSMAX]) {
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) {
      if (s < ks) {
        const int k0 = 16 * s + 8 * hi;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = k0 + j < c ? r.g[s][j] : 0.f;
        fa[s].u.x = pack2(g[0], g[1]);
        fa[s].u.y = pack2(g[2], g[3]);
        fa[s].u.z = pack2(g[4], g[5]);
        fa[s].u.w = pack2(g[6], g[7]);
      }
    }
  };

  Frag cur[KSMAX];
  Raw raw_cur, raw_nxt;
  int64_t tile = static_cast<int64_t>(blockIdx.x) * W + wid;
  const int64_t step = static_cast<int64_t>(gridDim.x) * W;
  int32_t row_nxt = 0;
  if (tile < ntiles) {
    load_raw(row_of(tile), raw_cur);
    row_nxt = row_of(tile + step);
  }
  for (; tile < ntiles; tile += step) {
    const bool more = tile + step < ntiles;                          // (wave-uniform)
    if (more) load_raw(row_nxt, raw_nxt);                            // in flight while this tile is multiplied and stored
    const int32_t row_nn = row_of(tile + 2 * step);
    pack_frags(raw_cur, cur);
    if (gout != nullptr && tile * 32 + i31 < n) {
      // the logits' gradient in the storage dtype, zero-padded to 16 ks classes, in the MODULE's row order: the operand the
      // weight gradient's node reduction (sgf_gram) reads — straight from the A fragments (lane = node, 8 classes each)
      // instead of a cast and a pad pass over dlogits
#pragma unroll
      for (int s = 0; s < KSMAX; ++s)
        if (s < ks) *reinterpret_cast<uint4*>(gout + (tile * 32 + i31) * ldgo + 16 * s + 8 * hi) = cur[s].u;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (HW * h >= d) break;
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KSMAX; ++s) {
        if (s < ks) {
          const int k0 = 16 * s + 8 * hi;
#pragma unroll
          for (int t = 0; t < NT; ++t) {                            // B operand: feature HW h + 32 t + i31, the same classes
            const bf16x8 bt = *reinterpret_cast<const bf16x8*>(&wt[(HW * h + 32 * t + i31) * PITCH + k0]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[s].v, bt, acc[t], 0, 0, 0);
          }
        }
      }
      // accumulator register r of lane (i31, hi): node (r & 3) + 8 (r >> 2) + 4 hi, feature HW h + 32 t + i31
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float sc = which == 0 ? a : b;
        uint16_t* const dst = which == 0 ? dx1 : dx2;
        const int64_t ld = which == 0 ? ld1 : ld2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const uint32_t v = pack2(sc * acc[t][r], sc * acc[t][r + 1]);
            const int rl = (r & 3) + 8 * (r >> 2);
            *reinterpret_cast<uint16_t*>(pw + rl * PROW + 64 * t) = static_cast<uint16_t>(v & 0xffffu);
            *reinterpret_cast<uint16_t*>(pw + (rl + 1) * PROW + 64 * t) = static_cast<uint16_t>(v >> 16);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int fcol = HW * h + 8 * (lane % LPR);
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j) {
          const uint4 v = *reinterpret_cast<const uint4*>(pr + RPI * j * PROW);
          const int64_t orow = tile * 32 + RPI * j + lane / LPR;
          if (fcol < d && orow < n) *reinterpret_cast<uint4*>(dst + orow * ld + fcol) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    raw_cur = raw_nxt;
    row_nxt = row_nn;
  }
}

inline int head_grid(int64_t n) {
  int64_t b = (n + 127) / 128;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 4;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

int check_head(const char* fn, int64_t n, int d, int c, int dtype) {
  SGF_REQUIRE(n >= 0 && d > 0 && c > 0, SGF_E_INVALID, "%s: bad size", fn);
  if (dtype == SGF_F32) {
    SGF_REQUIRE(linear_f32_supported(d, c), SGF_E_UNSUPPORTED,
                "%s: fp32 storage needs d and classes multiples of 4 up to 256 (d=%d, classes=%d; pad the class rows of W)", fn, d, c);
    return SGF_OK;
  }
  SGF_REQUIRE(dtype == SGF_BF16, SGF_E_UNSUPPORTED, "%s: unknown dtype %d", fn, dtype);
  if (c > kMaxClasses) {                               // bf16 rows, many classes: the exact-fp32 kernel with bf16 on the wire
    SGF_REQUIRE(linear_f32_supported(d, c), SGF_E_UNSUPPORTED,
                "%s: more than %d classes need d and classes multiples of 4 up to 256 (d=%d, classes=%d)", fn, kMaxClasses, d, c);
    return SGF_OK;
  }
  SGF_REQUIRE(d % 32 == 0 && d <= 256 && c <= kMaxClasses, SGF_E_UNSUPPORTED,
              "%s: needs d %% 32 == 0, d <= 256, classes <= %d (d=%d, classes=%d)", fn, kMaxClasses, d, c);
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" int32_t sgf_combine_fc_supported(int32_t d, int32_t classes, int32_t dtype) {
  if (dtype == SGF_F32) return linear_f32_supported(d, classes) ? 1 : 0;   // exact-fp32 matrix cores (csrc/linear_f32.hip)
  if (dtype != SGF_BF16 || d <= 0 || classes <= 0) return 0;
  if (classes > kMaxClasses) return linear_f32_supported(d, classes) ? 1 : 0;   // the same kernel, bf16 rows in / out
  return d % 32 == 0 && d <= 256 ? 1 : 0;
}

static int combine_fc_fwd_impl(const char* fn, const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
                               const float* w, const float* bias, int64_t n, int32_t d, int32_t classes, int32_t dtype,
                               float* logits, int64_t ldl, const int32_t* rmap, void* stream) {
  int rc = check_head(fn, n, d, classes, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(!rmap || (dtype == SGF_BF16 && classes <= kMaxClasses), SGF_E_UNSUPPORTED,
              "%s: a row map needs bf16 storage and at most %d classes", fn, kMaxClasses);
  if (dtype == SGF_F32) {
    SGF_REQUIRE(x1 && x2 && w && logits && ld1 >= d && ld2 >= d && ldl >= classes, SGF_E_INVALID,
                "%s: bad pointer / ld", fn);
    return linear_f32_dual(x1, ld1, x2, ld2, a, b, n, d, classes, w, d, 1, bias, logits, ldl, nullptr, 0, 1.f, 1.f, 0, 0,
                           static_cast<hipStream_t>(stream));
  }
  if (classes > kMaxClasses) {
    SGF_REQUIRE(x1 && x2 && w && logits && ld1 >= d && ld2 >= d && ldl >= classes, SGF_E_INVALID,
                "%s: bad pointer / ld", fn);
    return linear_f32_dual(x1, ld1, x2, ld2, a, b, n, d, classes, w, d, 1, bias, logits, ldl, nullptr, 0, 1.f, 1.f, 1, 0,
                           static_cast<hipStream_t>(stream));
  }
  SGF_REQUIRE(x1 && x2 && w && bias && logits && ld1 % 8 == 0 && ld2 % 8 == 0 && ld1 >= d && ld2 >= d && ldl >= classes,
              SGF_E_INVALID, "%s: bad pointer / ld", fn);
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(x1) % 16 == 0 && reinterpret_cast<uintptr_t>(x2) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(w) % 16 == 0,
              SGF_E_INVALID, "%s: x1 / x2 / w must be 16-byte aligned", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(head_grid(n)), block(kHeadThreads);
#define SGF_HEAD_FWD(DP_)                                                                                        \
  hipLaunchKernelGGL((k_head_fwd_bf16<DP_>), grid, block, 0, st, static_cast<const uint16_t*>(x1), ld1, a,       \
                     static_cast<const uint16_t*>(x2), ld2, b, w, bias, n, d, classes, logits, ldl, rmap)
  if (d <= 64) SGF_HEAD_FWD(64);
  else if (d <= 128) SGF_HEAD_FWD(128);
  else SGF_HEAD_FWD(256);
#undef SGF_HEAD_FWD
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_combine_fc_fwd(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
                                  const float* w, const float* bias, int64_t n, int32_t d, int32_t classes,
                                  int32_t dtype, float* logits, int64_t ldl, void* stream) {
  return combine_fc_fwd_impl("sgf_combine_fc_fwd", x1, ld1, a, x2, ld2, b, w, bias, n, d, classes, dtype, logits, ldl, nullptr,
                             stream);
}

// the same with the rows of the result scattered: logits[row_map[j]] = row j (row_map: a permutation of [0, n))
extern "C" int sgf_combine_fc_fwd_mapped(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
                                         const float* w, const float* bias, int64_t n, int32_t d, int32_t classes,
                                         int32_t dtype, float* logits, int64_t ldl, const int32_t* row_map, void* stream) {
  SGF_REQUIRE(row_map || n == 0, SGF_E_INVALID, "sgf_combine_fc_fwd_mapped: null row_map");
  return combine_fc_fwd_impl("sgf_combine_fc_fwd_mapped", x1, ld1, a, x2, ld2, b, w, bias, n, d, classes, dtype, logits, ldl,
                             row_map, stream);
}

static int combine_fc_bwd_impl(const char* fn, const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d,
                               int32_t classes, float a, float b, int32_t dtype, void* dx1, int64_t ld1, void* dx2,
                               int64_t ld2, const int32_t* rmap, void* stream, void* gout = nullptr, int64_t ldgo = 0) {
  int rc = check_head(fn, n, d, classes, dtype);
  if (rc != SGF_OK) return rc;
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(!rmap || (dtype == SGF_BF16 && classes <= kMaxClasses), SGF_E_UNSUPPORTED,
              "%s: a row map needs bf16 storage and at most %d classes", fn, kMaxClasses);
  if (dtype == SGF_F32) {
    SGF_REQUIRE(dlogits && w && dx1 && dx2 && lddl >= classes && ld1 >= d && ld2 >= d, SGF_E_INVALID,
                "%s: bad pointer / ld", fn);
    return linear_f32_dual(dlogits, lddl, nullptr, 0, 1.f, 0.f, n, classes, d, w, d, 0, nullptr, dx1, ld1, dx2, ld2, a, b, 0, 0,
                           static_cast<hipStream_t>(stream));
  }
  if (classes > kMaxClasses) {
    SGF_REQUIRE(dlogits && w && dx1 && dx2 && lddl >= classes && ld1 >= d && ld2 >= d, SGF_E_INVALID,
                "%s: bad pointer / ld", fn);
    return linear_f32_dual(dlogits, lddl, nullptr, 0, 1.f, 0.f, n, classes, d, w, d, 0, nullptr, dx1, ld1, dx2, ld2, a, b, 0, 1,
                           static_cast<hipStream_t>(stream));
  }
  SGF_REQUIRE(dlogits && w && dx1 && dx2 && lddl >= classes && ld1 >= d && ld2 >= d && ld1 % 8 == 0 && ld2 % 8 == 0 &&
                  reinterpret_cast<uintptr_t>(dx1) % 16 == 0 && reinterpret_cast<uintptr_t>(dx2) % 16 == 0,
              SGF_E_INVALID, "%s: bad pointer / ld (dx1 / dx2: 16-byte aligned, ld %% 8 == 0)", fn);
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED, "%s: more than 2^31 - 1 rows", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t nb = ((n + 31) / 32 + kHeadBwdThreads / 64 - 1) / (kHeadBwdThreads / 64);
  if (nb > kNumCU) nb = kNumCU;
  const dim3 grid(static_cast<unsigned>(nb)), block(kHeadBwdThreads);
#define SGF_HEAD_BWD(DP_)                                                                                   \
  hipLaunchKernelGGL((k_head_bwd_bf16<DP_>), grid, block, 0, st, dlogits, lddl, w, n, d, classes, a, b,       \
                     static_cast<uint16_t*>(dx1), ld1, static_cast<uint16_t*>(dx2), ld2, rmap,                \
                     static_cast<uint16_t*>(gout), ldgo)
  if (d <= 64) SGF_HEAD_BWD(64);
  else if (d <= 128) SGF_HEAD_BWD(128);
  else SGF_HEAD_BWD(256);
#undef SGF_HEAD_BWD
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_combine_fc_bwd(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d,
                                  int32_t classes, float a, float b, int32_t dtype, void* dx1, int64_t ld1,
                                  void* dx2, int64_t ld2, void* stream) {
  return combine_fc_bwd_impl("sgf_combine_fc_bwd", dlogits, lddl, w, n, d, classes, a, b, dtype, dx1, ld1, dx2, ld2, nullptr,
                             stream);
}

// the same (row_map may be null) that ALSO leaves the logits' gradient in the storage dtype: g_out[n, 16 ceil(classes / 16)]
// (zero-padded, the module's row order) for the weight gradient's node reductions — bf16 storage, classes <= 64
extern "C" int sgf_combine_fc_bwd_g(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d, int32_t classes,
                                    float a, float b, int32_t dtype, void* dx1, int64_t ld1, void* dx2, int64_t ld2,
                                    const int32_t* row_map, void* g_out, int64_t ldg, void* stream) {
  const char* fn = "sgf_combine_fc_bwd_g";
  SGF_REQUIRE(dtype == SGF_BF16 && classes >= 1 && classes <= kMaxClasses, SGF_E_UNSUPPORTED,
              "%s: bf16 storage and at most %d classes", fn, kMaxClasses);
  SGF_REQUIRE(n == 0 || (g_out && ldg >= (classes + 15) / 16 * 16 && ldg % 8 == 0 && reinterpret_cast<uintptr_t>(g_out) % 16 == 0),
              SGF_E_INVALID, "%s: g_out must be 16-byte aligned with ldg %% 8 == 0 and ldg >= classes rounded up to 16", fn);
  return combine_fc_bwd_impl(fn, dlogits, lddl, w, n, d, classes, a, b, dtype, dx1, ld1, dx2, ld2, row_map, stream, g_out, ldg);
}

// the same reading row row_map[j] of dlogits for row j of the gradients
extern "C" int sgf_combine_fc_bwd_mapped(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d,
                                         int32_t classes, float a, float b, int32_t dtype, void* dx1, int64_t ld1,
                                         void* dx2, int64_t ld2, const int32_t* row_map, void* stream) {
  SGF_REQUIRE(row_map || n == 0, SGF_E_INVALID, "sgf_combine_fc_bwd_mapped: null row_map");
  return combine_fc_bwd_impl("sgf_combine_fc_bwd_mapped", dlogits, lddl, w, n, d, classes, a, b, dtype, dx1, ld1, dx2, ld2,
                             row_map, stream);
}
