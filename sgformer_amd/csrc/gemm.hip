// gemm.hip — T4 catch-all: the matrix product behind every Linear shape the streaming kernels do not take, and the
// d x d algebra of the attention (sgf_attn_h_small_*, attn_small.hip).
//
//     c[i, j] = alpha * sum_k A(i, k) B(k, j) + bias[j] + beta * addend[i, j]
//     A(i, k) = a[i * a_rs + k * a_cs],   B(k, j) = b[k * b_rs + j * b_cs]      (element strides: any transposition,
//                                                                                any alignment, any m / n / k)
// Replaces, for the shapes outside rowgemm.hip / linear_f32.hip (large/ours.py:77,198 with f = 1433 or f not a multiple of 4,
// odd hidden widths, multi-head projections :123-126, medium/models.py GCNConv's x @ weight), the library GEMMs
// F.linear / addmm / matmul reached before — so that no Linear of the path, whatever its shape, leaves libsgf.so.
//
// One skeleton, two matrix-core forms: both operands bf16 -> v_mfma_f32_32x32x16_bf16 (exact products, fp32 sums);
// otherwise v_mfma_f32_32x32x2f32 (an exact fp32 FMA chain — against a CPU loop only the summation order differs), a bf16
// operand widened while it is staged.  64 x 64 output tile per 4-wave workgroup, K in steps of 32 or 128 through LDS; both tiles
// are stored K-CONTIGUOUS in LDS whatever the operand's layout in memory (the staging loop walks the operand along its
// contiguous dimension — chosen per operand at launch — so global loads coalesce either way), next K-step's elements
// are requested into registers before the current step is multiplied.  Output-tile index: column tiles fastest, so the
// workgroups that share a row tile of A run together and its re-reads are L2 hits.
// This is the general-shape kernel, not the fast path: the recipe shapes run on the streaming kernels.
#include "common.h"

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGemmThreads = 256;
constexpr int GBM = 64, GBN = 64;      // output tile; the K step is a template parameter: 32, or 128 once k >= 128 (a product of
                                       // a few hundred in every dimension — the d x d algebra — is bound by the LATENCY of its
                                       // serial K steps, not by bytes: 23 us at K step 32 for 512 x 257 x 257)

struct GemmArgs {
  const void* a;
  int64_t a_rs, a_cs;
  const void* b;
  int64_t b_rs, b_cs;
  int32_t a_bf16, b_bf16;
  int32_t a_kc, b_kc;          // the operand's k stride is 1: stage it walking k; else walk the other dimension
  int64_t m;
  int32_t n;
  int64_t k;
  float alpha;
  const float* alpha_dev;      // nullable: alpha *= *alpha_dev (a scalar computed on the device, e.g. 1 / (||Q|| ||K||))
  const float* bias;           // [n] fp32 or null
  float beta;
  const void* addend;          // [m, n] or null
  int64_t ldadd;
  int32_t add_bf16;
  void* c;
  int64_t ldc;
  int32_t c_bf16;
  int32_t nb_n;                // column tiles
};

template <bool MX16>
struct Stage;
template <>
struct Stage<true> {
  using T = uint16_t;
  static constexpr int kPad = 8;              // 80- / 272-byte rows: 16 lanes' 16-byte fragment reads fall on disjoint banks
  static __device__ __forceinline__ T fetch(const void* p, int64_t i, int) { return static_cast<const uint16_t*>(p)[i]; }
  static __device__ __forceinline__ T zero() { return 0; }
};
template <>
struct Stage<false> {
  using T = float;
  static constexpr int kPad = 4;              // 144- / 528-byte rows: likewise for the float4 fragment reads
  static __device__ __forceinline__ T fetch(const void* p, int64_t i, int is_bf16) {
    return is_bf16 ? bf16_to_f32(static_cast<const uint16_t*>(p)[i]) : static_cast<const float*>(p)[i];
  }
  static __device__ __forceinline__ T zero() { return 0.f; }
};

template <bool MX16, int GBK>
__global__ __launch_bounds__(kGemmThreads) void k_gemm(GemmArgs p) {
  using S = Stage<MX16>;
  using T = typename S::T;
  constexpr int P = GBK + S::kPad;
  constexpr int E = GBM * GBK / kGemmThreads;       // elements of each tile a thread stages per K step
  __shared__ __attribute__((aligned(16))) T As[GBM * P];
  __shared__ __attribute__((aligned(16))) T Bs[GBN * P];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i31 = lane & 31, hi = lane >> 5;
  const int64_t tm = blockIdx.x / p.nb_n;
  const int tn = blockIdx.x % p.nb_n;
  const int64_t m0 = tm * GBM;
  const int n0 = tn * GBN;

  // staging coordinates of this thread's E elements of each tile: (r, kk) for element e
  auto coord = [&](int kc, int e, int& r, int& kk) {
    if (kc) { kk = tid % GBK; r = tid / GBK + (kGemmThreads / GBK) * e; }
    else    { r = tid & 63; kk = (tid >> 6) + 4 * e; }
  };
  T ra[E], rb[E];
  auto issue = [&](int64_t k0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int r, kk;
      coord(p.a_kc, e, r, kk);
      const int64_t i = m0 + r, k = k0 + kk;
      ra[e] = (i < p.m && k < p.k) ? S::fetch(p.a, i * p.a_rs + k * p.a_cs, p.a_bf16) : S::zero();
      coord(p.b_kc, e, r, kk);
      const int64_t j = n0 + r, kb = k0 + kk;
      rb[e] = (j < p.n && kb < p.k) ? S::fetch(p.b, kb * p.b_rs + j * p.b_cs, p.b_bf16) : S::zero();
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int r, kk;
      coord(p.a_kc, e, r, kk);
      As[r * P + kk] = ra[e];
      coord(p.b_kc, e, r, kk);
      Bs[r * P + kk] = rb[e];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int64_t nk = (p.k + GBK - 1) / GBK;
  if (nk > 0) {
    issue(0);
    commit();
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) issue((kt + 1) * GBK);
    const T* A = As + (32 * wm + i31) * P;
    const T* B = Bs + (32 * wn + i31) * P;
    if constexpr (MX16) {
#pragma unroll
      for (int s = 0; s < GBK / 16; ++s) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(A + 16 * s + 8 * hi);
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(B + 16 * s + 8 * hi);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < GBK / 8; ++s) {
        const float4 a4 = *reinterpret_cast<const float4*>(A + 8 * s + 4 * hi);
        const float4 b4 = *reinterpret_cast<const float4*>(B + 8 * s + 4 * hi);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }

  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= *p.alpha_dev;
  const int j = n0 + 32 * wn + i31;
  if (j < p.n) {
    const float bj = p.bias ? p.bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = m0 + 32 * wm + mfma32_row(r, lane);
      if (i < p.m) {
        float v = alpha * acc[r] + bj;
        if (p.addend) {
          const float o = p.add_bf16 ? bf16_to_f32(static_cast<const uint16_t*>(p.addend)[i * p.ldadd + j])
                                     : static_cast<const float*>(p.addend)[i * p.ldadd + j];
          v = fmaf(p.beta, o, v);
        }
        if (p.c_bf16) static_cast<uint16_t*>(p.c)[i * p.ldc + j] = f32_to_bf16(v);
        else static_cast<float*>(p.c)[i * p.ldc + j] = v;
      }
    }
  }
}

}  // namespace

int gemm_launch(const void* a, int64_t a_rs, int64_t a_cs, int a_dtype, const void* b, int64_t b_rs, int64_t b_cs, int b_dtype,
                int64_t m, int n, int64_t k, float alpha, const float* alpha_dev, const float* bias, float beta,
                const void* addend, int64_t ldadd, int add_dtype, void* c, int64_t ldc, int c_dtype, hipStream_t st) {
  if (m == 0 || n == 0) return SGF_OK;
  const int64_t nb_m = (m + GBM - 1) / GBM;
  const int64_t nb_n = (n + GBN - 1) / GBN;
  SGF_REQUIRE(nb_m * nb_n < (int64_t{1} << 31), SGF_E_UNSUPPORTED, "sgf_gemm: %lld x %lld output tiles exceed one launch",
              static_cast<long long>(nb_m), static_cast<long long>(nb_n));
  GemmArgs p{a, a_rs, a_cs, b, b_rs, b_cs, a_dtype == SGF_BF16, b_dtype == SGF_BF16, a_cs == 1, b_rs == 1, m, n, k, alpha,
             alpha_dev, bias, beta, addend, ldadd, add_dtype == SGF_BF16, c, ldc, c_dtype == SGF_BF16, static_cast<int32_t>(nb_n)};
  const dim3 grid(static_cast<unsigned>(nb_m * nb_n));
  const bool mx16 = p.a_bf16 && p.b_bf16;
  if (k >= 128) {
    if (mx16) hipLaunchKernelGGL((k_gemm<true, 128>), grid, dim3(kGemmThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm<false, 128>), grid, dim3(kGemmThreads), 0, st, p);
  } else {
    if (mx16) hipLaunchKernelGGL((k_gemm<true, 32>), grid, dim3(kGemmThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm<false, 32>), grid, dim3(kGemmThreads), 0, st, p);
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace sgf

extern "C" int sgf_gemm(const void* a, int64_t a_rs, int64_t a_cs, int32_t a_dtype, const void* b, int64_t b_rs, int64_t b_cs,
                        int32_t b_dtype, int64_t m, int32_t n, int64_t k, float alpha, const float* alpha_dev, const float* bias,
                        float beta, const void* addend, int64_t ldadd, int32_t add_dtype, void* c, int64_t ldc, int32_t c_dtype,
                        void* stream) {
  using namespace sgf;
  SGF_REQUIRE(m >= 0 && n >= 0 && k >= 0, SGF_E_INVALID, "sgf_gemm: negative size");
  if (m == 0 || n == 0) return SGF_OK;
  SGF_REQUIRE(c && (k == 0 || (a && b)), SGF_E_INVALID, "sgf_gemm: null operand");
  SGF_REQUIRE((a_dtype == SGF_F32 || a_dtype == SGF_BF16) && (b_dtype == SGF_F32 || b_dtype == SGF_BF16) &&
                  (c_dtype == SGF_F32 || c_dtype == SGF_BF16) && (!addend || add_dtype == SGF_F32 || add_dtype == SGF_BF16),
              SGF_E_INVALID, "sgf_gemm: unknown dtype code");
  SGF_REQUIRE(ldc >= n && (!addend || ldadd >= n), SGF_E_INVALID, "sgf_gemm: leading dimension smaller than n = %d", n);
  return gemm_launch(a, a_rs, a_cs, a_dtype, b, b_rs, b_cs, b_dtype, m, n, k, alpha, alpha_dev, bias, beta, addend, ldadd,
                     add_dtype, c, ldc, c_dtype, static_cast<hipStream_t>(stream));
}
