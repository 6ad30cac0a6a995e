// linear_f32.hip — T4 / T6 for fp32 storage (BASELINE.json configs 2 and 4): the Linear layers
//     y = [x_1 | x_2] W^T + b   (large/ours.py:36-40, :77, :198, :275)   and   dX = dY W
// as streaming passes on the exact-fp32 matrix cores, with the BatchNorm column sums of large/ours.py:87-88 taken in
// the same pass — so that the fp32 configurations, like the bf16 ones, run without a library GEMM and without a
// separate statistics pass.
//
// Skeleton of k_attn_apply<float> (attn.hip): [n, dk] x [dk, dj], dk, dj <= 256.  d = 256 in fp32 means a 256 KiB
// weight matrix — it does not fit the 160 KiB of LDS the bf16 kernels (rowgemm.hip) park theirs in — so the matrix lives
// in REGISTERS: 16 waves = column strips x row sub-blocks x 2 K-halves, a wave keeps the [DP/2 x 32] piece for its
// (strip, K-half) in DP/4 VGPRs for the whole kernel (persistent, one block per CU) and the row tiles stream through a
// double-buffered LDS tile; v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain, so against a CPU loop only the
// summation order differs.  Both K-halves drop their accumulator tile into LDS and the epilogue runs row-wise with 16 B
// per lane: + bias, + addend (the first operand's product of a two-operand Linear), shifted column sums.
// MFMA-bound at d = 256 (2 n d^2 flop at 155 TF: 2.1 ms at ogbn-products size, against 0.8 ms of HBM time).
#include "common.h"

namespace sgf {
namespace {

constexpr int kLinThreads = 1024;

struct LinArgs {
  const float* a;
  int64_t lda;
  const float* w;          // trans_w = 1: B[k][j] = w[j * ldw + k] (y = x W^T);  0: B[k][j] = w[k * ldw + j] (dx = dy W)
  int64_t ldw;
  int32_t trans_w;
  const float* bias;       // [dj] or null
  const float* addend;     // [n, dj] or null
  int64_t ldadd;
  const float* shift;      // [dj] or null (statistics are of out - shift)
  float* out;
  int64_t ldo;
  int64_t n;
  int32_t dk, dj;
  float* spart;            // [gridDim.x][2 * dj] per-block column sums / sums of squares, or null
  // DUAL form (T7 for fp32 storage, large/ours.py:269-275): the A operand is ca * a + cb * a2, formed while the row tile is
  // staged (a2 != null), and / or the result leaves twice, co * v -> out and co2 * v -> out2 (out2 != null)
  const float* a2;
  int64_t lda2;
  float ca, cb;
  float* out2;
  int64_t ldo2;
  float co, co2;
};

__device__ __forceinline__ float4 zero4f() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// IN16 / OUT16 (DUAL only): the A operand(s) / the result(s) are bf16 in memory — the fused head of bf16 runs whose class count
// exceeds the bf16 head kernel's 64 (C = 172 at the papers100M recipe): same kernel, fp32 arithmetic, bf16 only on the wire.
template <int DP, bool STATS, bool DUAL = false, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(kLinThreads) void k_linear_f32(LinArgs p) {
  constexpr int NS = DP / 32;         // 32-column strips
  constexpr int RS = 8 / NS;          // row sub-blocks
  constexpr int RT = 32 * RS;         // rows per tile (32 / 64 / 128)
  constexpr int F4 = DP / 4;
  constexpr int RPP = kLinThreads / F4;   // rows covered per staging pass (2 passes)
  constexpr int LD = DP + 4;
  constexpr int KSH = DP / 16;        // k-steps of 8 per K-half
  __shared__ float smem[4 * RT * LD];
  float* const ldsA = smem;                       // [buf][RT][LD]
  float* const ldsC = smem + 2 * RT * LD;         // [kh][RT][LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i31 = lane & 31;
  const int hi = lane >> 5;
  const int kh = wave & 1;
  const int ws = (wave >> 1) % NS;
  const int wr = (wave >> 1) / NS;

  // resident piece of the matrix: breg[4 s + t] = B[8 (s + kh KSH) + 4 hi + t][32 ws + i31]
  float breg[DP / 4];
  {
    const int j = 32 * ws + i31;
#pragma unroll
    for (int s = 0; s < KSH; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 8 * (s + kh * KSH) + 4 * hi + t;
        float v = 0.f;
        if (k < p.dk && j < p.dj) v = p.trans_w ? p.w[static_cast<int64_t>(j) * p.ldw + k] : p.w[static_cast<int64_t>(k) * p.ldw + j];
        breg[4 * s + t] = v;
      }
  }
  const int scol = (tid % F4) * 4;
  const int srow0 = tid / F4;
  const bool in_ok = scol < p.dk, out_ok = scol < p.dj;
  // (bias / shift chunks are re-read from the L1-resident vectors in the epilogue: the 256-wide kernel sits exactly at
  // the 128-register budget of 4 waves per SIMD)
  float4 s1 = zero4f(), s2 = zero4f();
  const float* pa = p.a + (IN16 ? scol / 2 : scol);     // (bf16 operands: the pointer is typed float, offsets in bf16 pairs)
  const float* pa2 = (DUAL && p.a2) ? p.a2 + (IN16 ? scol / 2 : scol) : nullptr;
  float4 ra[2];
  const int64_t ntiles = (p.n + RT - 1) / RT;
  auto load_a = [&](const float* base, int64_t row, int64_t ld) -> float4 {
    if (IN16) return load4<uint16_t>(reinterpret_cast<const uint16_t*>(base) + row * ld);
    return *reinterpret_cast<const float4*>(base + row * ld);
  };
  auto issue = [&](int64_t tile) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t row = tile * RT + srow0 + i * RPP;
      ra[i] = (in_ok && row < p.n) ? load_a(pa, row, p.lda) : zero4f();
      if (DUAL && pa2 != nullptr) {                     // the combination a * x1 + b * x2, in fp32, never written
        const float4 r2 = (in_ok && row < p.n) ? load_a(pa2, row, p.lda2) : zero4f();
        ra[i] = make_float4(p.ca * ra[i].x + p.cb * r2.x, p.ca * ra[i].y + p.cb * r2.y, p.ca * ra[i].z + p.cb * r2.z,
                            p.ca * ra[i].w + p.cb * r2.w);
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(&ldsA[(buf * RT + srow0 + i * RPP) * LD + scol]) = ra[i];
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
      float* C = ldsC + (kh * RT + 32 * wr + 4 * hi) * LD + 32 * ws + i31;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2)) * LD] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lrow = srow0 + i * RPP;
      const int64_t row = tile * RT + lrow;
      if (out_ok && row < p.n) {
        const float4 c0 = *reinterpret_cast<const float4*>(&ldsC[lrow * LD + scol]);
        const float4 c1 = *reinterpret_cast<const float4*>(&ldsC[(RT + lrow) * LD + scol]);
        const float4 bz = p.bias ? *reinterpret_cast<const float4*>(p.bias + scol) : zero4f();
        float4 v = make_float4((c0.x + c1.x) + bz.x, (c0.y + c1.y) + bz.y, (c0.z + c1.z) + bz.z, (c0.w + c1.w) + bz.w);
        if (p.addend) {
          const float4 o = *reinterpret_cast<const float4*>(p.addend + row * p.ldadd + scol);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (DUAL && p.out2 != nullptr) {                // dx1 = a (g W), dx2 = b (g W): one product, two scaled copies
          const float4 v1 = make_float4(p.co * v.x, p.co * v.y, p.co * v.z, p.co * v.w);
          const float4 v2 = make_float4(p.co2 * v.x, p.co2 * v.y, p.co2 * v.z, p.co2 * v.w);
          if (OUT16) {
            store4<uint16_t>(reinterpret_cast<uint16_t*>(p.out) + row * p.ldo + scol, v1);
            store4<uint16_t>(reinterpret_cast<uint16_t*>(p.out2) + row * p.ldo2 + scol, v2);
          } else {
            *reinterpret_cast<float4*>(p.out + row * p.ldo + scol) = v1;
            *reinterpret_cast<float4*>(p.out2 + row * p.ldo2 + scol) = v2;
          }
        } else {
          *reinterpret_cast<float4*>(p.out + row * p.ldo + scol) = v;
        }
        if (STATS) {
          const float4 sh = p.shift ? *reinterpret_cast<const float4*>(p.shift + scol) : zero4f();
          const float4 u = make_float4(v.x - sh.x, v.y - sh.y, v.z - sh.z, v.w - sh.w);
          s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w;
          s2.x = fmaf(u.x, u.x, s2.x); s2.y = fmaf(u.y, u.y, s2.y); s2.z = fmaf(u.z, u.z, s2.z); s2.w = fmaf(u.w, u.w, s2.w);
        }
      }
    }
    if (has_next) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (STATS) {
    // the RPP threads that share a column chunk add their sums in thread order (deterministic)
    float* red = smem;                                   // [RPP][2][DP]
    *reinterpret_cast<float4*>(&red[(srow0 * 2 + 0) * DP + scol]) = s1;
    *reinterpret_cast<float4*>(&red[(srow0 * 2 + 1) * DP + scol]) = s2;
    __syncthreads();
    if (srow0 < 2 && out_ok) {
      float4 t = zero4f();
#pragma unroll 4
      for (int r = 0; r < RPP; ++r) {
        const float4 q = *reinterpret_cast<const float4*>(&red[(r * 2 + srow0) * DP + scol]);
        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
      }
      *reinterpret_cast<float4*>(p.spart + (static_cast<int64_t>(blockIdx.x) * 2 + srow0) * p.dj + scol) = t;
    }
  }
}

}  // namespace

bool linear_f32_supported(int d_in, int d_out) {
  return d_in > 0 && d_out > 0 && d_in <= 256 && d_out <= 256 && d_in % 4 == 0 && d_out % 4 == 0;
}

int linear_f32_blocks(int64_t n) {
  int64_t b = (n + 31) / 32;
  if (b > kNumCU) b = kNumCU;
  return b < 1 ? 1 : static_cast<int>(b);
}

// out [n, dj] = a [n, dk] B (+ bias) (+ addend);  spart != null: per-block shifted column sums [blocks][2 * dj]
int linear_f32(const float* a, int64_t lda, int64_t n, int dk, int dj, const float* w, int64_t ldw, int trans_w,
               const float* bias, const float* addend, int64_t ldadd, const float* shift, float* out, int64_t ldo,
               float* spart, hipStream_t st) {
  SGF_REQUIRE(linear_f32_supported(dk, dj), SGF_E_UNSUPPORTED, "linear_f32: widths %d -> %d (multiples of 4, <= 256)", dk, dj);
  SGF_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && (!addend || ldadd % 4 == 0) && reinterpret_cast<uintptr_t>(a) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0 && (!addend || reinterpret_cast<uintptr_t>(addend) % 16 == 0) &&
                  (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0) && (!shift || reinterpret_cast<uintptr_t>(shift) % 16 == 0),
              SGF_E_INVALID, "linear_f32: rows must be 16-byte aligned");
  LinArgs p{a, lda, w, ldw, trans_w, bias, addend, ldadd, shift, out, ldo, n, dk, dj, spart, nullptr, 0, 1.f, 0.f, nullptr, 0,
            1.f, 1.f};
  const int blocks = linear_f32_blocks(n);
  const int dmax = dk > dj ? dk : dj;
  const int DP = dmax <= 64 ? 64 : (dmax <= 128 ? 128 : 256);
#define SGF_LIN(DP_)                                                                                              \
  do {                                                                                                            \
    if (spart) hipLaunchKernelGGL((k_linear_f32<DP_, true>), dim3(blocks), dim3(kLinThreads), 0, st, p);          \
    else hipLaunchKernelGGL((k_linear_f32<DP_, false>), dim3(blocks), dim3(kLinThreads), 0, st, p);               \
  } while (0)
  if (DP == 64) SGF_LIN(64);
  else if (DP == 128) SGF_LIN(128);
  else SGF_LIN(256);
#undef SGF_LIN
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// T7 for fp32 storage: out = (ca a + cb a2) B (+ bias)   [a2 != null],   or   out = co (a B), out2 = co2 (a B)   [out2 != null]
// in16: a / a2 point at bf16 rows (lda, lda2 in bf16 elements); out16: out / out2 receive bf16 (ldo, ldo2 in bf16 elements)
int linear_f32_dual(const void* a, int64_t lda, const void* a2, int64_t lda2, float ca, float cb, int64_t n, int dk, int dj,
                    const float* w, int64_t ldw, int trans_w, const float* bias, void* out, int64_t ldo, void* out2,
                    int64_t ldo2, float co, float co2, int in16, int out16, hipStream_t st) {
  SGF_REQUIRE(linear_f32_supported(dk, dj), SGF_E_UNSUPPORTED, "linear_f32: widths %d -> %d (multiples of 4, <= 256)", dk, dj);
  const uintptr_t ain = in16 ? 8 : 16, aout = out16 ? 8 : 16;
  SGF_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && (!a2 || lda2 % 4 == 0) && (!out2 || ldo2 % 4 == 0) &&
                  reinterpret_cast<uintptr_t>(a) % ain == 0 && reinterpret_cast<uintptr_t>(out) % aout == 0 &&
                  (!a2 || reinterpret_cast<uintptr_t>(a2) % ain == 0) && (!out2 || reinterpret_cast<uintptr_t>(out2) % aout == 0) &&
                  (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0),
              SGF_E_INVALID, "linear_f32: rows must be 16-byte aligned (8-byte for bf16 rows)");
  SGF_REQUIRE(!(in16 && out16) && (!out16 || out2), SGF_E_UNSUPPORTED, "linear_f32_dual: unsupported storage combination");
  LinArgs p{static_cast<const float*>(a), lda, w, ldw, trans_w, bias, nullptr, 0, nullptr, static_cast<float*>(out), ldo, n, dk,
            dj, nullptr, static_cast<const float*>(a2), lda2, ca, cb, static_cast<float*>(out2), ldo2, co, co2};
  const int blocks = linear_f32_blocks(n);
  const int dmax = dk > dj ? dk : dj;
#define SGF_LIN_DUAL(DP_)                                                                                                  \
  do {                                                                                                                     \
    if (in16) hipLaunchKernelGGL((k_linear_f32<DP_, false, true, true, false>), dim3(blocks), dim3(kLinThreads), 0, st, p);  \
    else if (out16) hipLaunchKernelGGL((k_linear_f32<DP_, false, true, false, true>), dim3(blocks), dim3(kLinThreads), 0, st, p); \
    else hipLaunchKernelGGL((k_linear_f32<DP_, false, true>), dim3(blocks), dim3(kLinThreads), 0, st, p);                  \
  } while (0)
  if (dmax <= 64) SGF_LIN_DUAL(64);
  else if (dmax <= 128) SGF_LIN_DUAL(128);
  else SGF_LIN_DUAL(256);
#undef SGF_LIN_DUAL
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace sgf
