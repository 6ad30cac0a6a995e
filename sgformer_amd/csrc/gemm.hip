// gemm.hip — T4 catch-all: the matrix product behind every Linear shape the streaming kernels do not take, and the
// d x d algebra of the attention (sgf_attn_h_small_*, attn_small.hip).
//
//     c[i, j] = alpha * sum_k A(i, k) B(k, j) + bias[j] + beta * addend[i, j]
//     A(i, k) = a[i * a_rs + k * a_cs],   B(k, j) = b[k * b_rs + j * b_cs]      (element strides: any transposition,
//                                                                                any alignment, any m / n / k)
// Replaces, for the shapes outside rowgemm.hip / linear_f32.hip (large/ours.py:77,198 with f = 1433 or f not a multiple of 4,
// odd hidden widths, multi-head projections :123-126, the 100M recipe's 128 -> 172 head :275, medium/models.py GCNConv's
// x @ weight), the library GEMMs F.linear / addmm / matmul reached before — no Linear of the path, whatever its shape,
// leaves libsgf.so.
//
// One skeleton, two matrix-core forms: both operands bf16 -> v_mfma_f32_32x32x16_bf16 (exact products, fp32 sums);
// otherwise v_mfma_f32_32x32x2f32 (an exact fp32 FMA chain — against a CPU loop only the summation order differs), a bf16
// operand widened while it is staged.  Output tile BM x 64 per 4-wave workgroup (BM = 128 for the node-sized products:
// a wave owns 32 rows and both 32-column tiles; BM = 64 for small matrices: more workgroups), K in steps of 128 BYTES per
// row (64 bf16 / 32 fp32) through LDS.  Both tiles sit K-CONTIGUOUS in LDS whatever the operand's layout in memory:
//   * an operand whose rows are 16-byte aligned along its contiguous dimension is staged with 16-byte loads — along K
//     (x of y = x W^T: one LDS store per load) or along the other dimension (W of dx = g W, A^T: VEC scalar LDS stores);
//   * anything else element by element, the staging loop walking the contiguous dimension (coalesced either way).
// The register prefetch runs TWO K steps ahead: a product of a few hundred in every dimension (the d x d algebra) has a
// handful of workgroups and is bound by the latency of its serial K steps, not by bytes.  Output-tile index: column tiles
// fastest, so the workgroups that share a row tile of A run together and its re-reads are L2 hits.
#include "common.h"

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGemmThreads = 256;
constexpr int GBN = 64;

struct GemmArgs {
  const void* a;
  int64_t a_rs, a_cs;
  const void* b;
  int64_t b_rs, b_cs;
  int32_t a_bf16, b_bf16;
  int32_t a_mode, b_mode;      // 0: element-wise walking k, 1: 16-byte loads along k, 2: 16-byte loads along the other
                               // dimension, 3: element-wise walking the other dimension
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
struct Stage<true> {           // both operands bf16 in memory: staged as they are
  using T = uint16_t;
  static constexpr int kBK = 64, kVec = 8, kPad = 8;      // 144-byte rows: 16 lanes' 16-byte fragment reads on disjoint banks
  static __device__ __forceinline__ T fetch(const void* p, int64_t i, int) { return static_cast<const uint16_t*>(p)[i]; }
  static __device__ __forceinline__ T zero() { return 0; }
  static __device__ __forceinline__ T get(const uint4& q, int i) {       // i: compile-time after unrolling
    const uint32_t w = (i >> 1) == 0 ? q.x : (i >> 1) == 1 ? q.y : (i >> 1) == 2 ? q.z : q.w;
    return static_cast<T>(w >> (16 * (i & 1)));
  }
};
template <>
struct Stage<false> {          // fp32 staging: an fp32 operand as it is, a bf16 operand widened
  using T = float;
  static constexpr int kBK = 32, kVec = 4, kPad = 4;      // 144-byte rows
  static __device__ __forceinline__ T fetch(const void* p, int64_t i, int is_bf16) {
    return is_bf16 ? bf16_to_f32(static_cast<const uint16_t*>(p)[i]) : static_cast<const float*>(p)[i];
  }
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T get(const uint4& q, int i) {
    return __uint_as_float(i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w);
  }
};

// ---- the multiply-accumulate of one staged K step and the epilogue, shared by both kernels --------------------------------
template <bool MX16, int BK, int P, int NACC, typename T>
__device__ __forceinline__ void mma_step(const T* As, const T* Bs, int wm, int wn0, int i31, int hi, f32x16 (&acc)[NACC]) {
  const T* A = As + (32 * wm + i31) * P;
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    const T* B = Bs + (32 * (wn0 + t) + i31) * P;
    if constexpr (MX16) {
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(A + 16 * s + 8 * hi);
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(B + 16 * s + 8 * hi);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        const float4 a4 = *reinterpret_cast<const float4*>(A + 8 * s + 4 * hi);
        const float4 b4 = *reinterpret_cast<const float4*>(B + 8 * s + 4 * hi);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[t], 0, 0, 0);
      }
    }
  }
}

template <int NACC>
__device__ __forceinline__ void epilogue(const GemmArgs& p, int64_t m0, int n0, int wm, int wn0, int lane, const f32x16 (&acc)[NACC]) {
  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= *p.alpha_dev;
  const int i31 = lane & 31;
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    const int j = n0 + 32 * (wn0 + t) + i31;
    if (j >= p.n) continue;
    const float bj = p.bias ? p.bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = m0 + 32 * wm + mfma32_row(r, lane);
      if (i < p.m) {
        float v = alpha * acc[t][r] + bj;
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

// ---- the fast kernel: both operands staged with 16-byte loads ----------------------------------------------------------------
// One operand's tile [ROWS][BK] (row = the non-k index) in flight: CH 16-byte chunks per thread.  ALONG_K: the operand's k
// stride is 1 (chunk = VEC consecutive k of one row, one 16-byte LDS store); else its other stride is 1 (chunk = VEC
// consecutive rows at one k, VEC scalar LDS stores).  The launcher guarantees that a chunk is entirely inside or entirely
// outside the operand (extent along the chunk direction % VEC == 0), so there is no element-wise path here.
template <bool MX16, int ROWS>
struct VecTile {
  using S = Stage<MX16>;
  using T = typename S::T;
  static constexpr int BK = S::kBK, VEC = S::kVec, P = BK + S::kPad;
  static constexpr int CH = ROWS * BK / VEC / kGemmThreads;
  static_assert(CH >= 1, "tile too small for the workgroup");
  uint4 v[CH];

  static __device__ __forceinline__ void coord(bool along_k, int c, int tid, int& row, int& kk) {
    const int id = tid + c * kGemmThreads;
    if (along_k) { row = id / (BK / VEC); kk = (id % (BK / VEC)) * VEC; }
    else         { kk = id / (ROWS / VEC); row = (id % (ROWS / VEC)) * VEC; }
  }
  __device__ __forceinline__ void issue(const void* p, int64_t ld, bool along_k, int tid, int64_t r0, int64_t nrow, int64_t k0,
                                        int64_t nk) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int row, kk;
      coord(along_k, c, tid, row, kk);
      const int64_t r = r0 + row, k = k0 + kk;
      v[c] = (r < nrow && k < nk) ? *reinterpret_cast<const uint4*>(static_cast<const T*>(p) + (along_k ? r * ld + k : k * ld + r))
                                  : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __device__ __forceinline__ void commit(T* lds, bool along_k, int tid) const {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int row, kk;
      coord(along_k, c, tid, row, kk);
      if (along_k) {
        *reinterpret_cast<uint4*>(lds + row * P + kk) = v[c];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) lds[(row + i) * P + kk] = S::get(v[c], i);
      }
    }
  }
};

template <bool MX16, int BM>
__global__ __launch_bounds__(kGemmThreads) void k_gemm_vec(GemmArgs p) {
  using S = Stage<MX16>;
  using T = typename S::T;
  constexpr int BK = S::kBK, P = BK + S::kPad;
  constexpr int NACC = BM / 64;                     // 32 x 32 accumulator tiles per wave
  __shared__ __attribute__((aligned(16))) T As[BM * P];
  __shared__ __attribute__((aligned(16))) T Bs[GBN * P];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = BM == 128 ? wave : wave >> 1;      // this wave's 32-row tile
  const int wn0 = BM == 128 ? 0 : (wave & 1);       // its first 32-column tile
  const int64_t m0 = static_cast<int64_t>(blockIdx.x / p.nb_n) * BM;
  const int n0 = static_cast<int>(blockIdx.x % p.nb_n) * GBN;
  const bool a_k = p.a_mode == 1, b_k = p.b_mode == 1;
  const int64_t lda = a_k ? p.a_rs : p.a_cs, ldb = b_k ? p.b_cs : p.b_rs;

  VecTile<MX16, BM> ra0, ra1;       // two K steps in flight (named, not indexed: they must stay in registers)
  VecTile<MX16, GBN> rb0, rb1;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int64_t nk = (p.k + BK - 1) / BK;
#define SGF_ISSUE(RA, RB, KT)                                          \
  do {                                                                 \
    RA.issue(p.a, lda, a_k, tid, m0, p.m, (KT) * BK, p.k);             \
    RB.issue(p.b, ldb, b_k, tid, n0, p.n, (KT) * BK, p.k);             \
  } while (0)
#define SGF_COMMIT(RA, RB)         \
  do {                             \
    RA.commit(As, a_k, tid);       \
    RB.commit(Bs, b_k, tid);       \
  } while (0)
  if (nk > 0) {
    SGF_ISSUE(ra0, rb0, 0);
    SGF_COMMIT(ra0, rb0);                           // tile 0 -> LDS
    if (nk > 1) SGF_ISSUE(ra0, rb0, 1);             // tiles 1, 2 -> registers
    if (nk > 2) SGF_ISSUE(ra1, rb1, 2);
  }
  __syncthreads();
  // multiply tile kt; the named register set holds tile kt + 1; two K steps of multiplication lie between a request and its use
#define SGF_STEP(RA, RB, KT)                                                            \
  do {                                                                                  \
    mma_step<MX16, BK, P, NACC, T>(As, Bs, wm, wn0, lane & 31, lane >> 5, acc);         \
    __syncthreads();                                                                    \
    if ((KT) + 1 < nk) SGF_COMMIT(RA, RB);                                              \
    __syncthreads();                                                                    \
    if ((KT) + 3 < nk) SGF_ISSUE(RA, RB, (KT) + 3);                                     \
  } while (0)
  for (int64_t kt = 0; kt < nk; kt += 2) {
    SGF_STEP(ra0, rb0, kt);
    if (kt + 1 < nk) SGF_STEP(ra1, rb1, kt + 1);
  }
#undef SGF_STEP
#undef SGF_COMMIT
#undef SGF_ISSUE
  epilogue<NACC>(p, m0, n0, wm, wn0, lane, acc);
}

// ---- the general kernel: any strides, any alignment, element by element ----------------------------------------------------
// 64 x 64 tile, K steps of 32, next step's elements requested before the current one is multiplied; the staging loop walks
// the operand along its contiguous dimension (mode 0: k, mode 3: the other one), so global loads coalesce either way.
template <bool MX16>
__global__ __launch_bounds__(kGemmThreads) void k_gemm_any(GemmArgs p) {
  using S = Stage<MX16>;
  using T = typename S::T;
  constexpr int BM = 64, BK = 32, P = BK + S::kPad, E = BM * BK / kGemmThreads;
  __shared__ __attribute__((aligned(16))) T As[BM * P];
  __shared__ __attribute__((aligned(16))) T Bs[GBN * P];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn0 = wave & 1;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x / p.nb_n) * BM;
  const int n0 = static_cast<int>(blockIdx.x % p.nb_n) * GBN;
  auto coord = [&](int mode, int e, int& r, int& kk) {
    if (mode <= 1) { kk = tid & 31; r = (tid >> 5) + 8 * e; }
    else           { r = tid & 63; kk = (tid >> 6) + 4 * e; }
  };
  T ra[E], rb[E];
  auto issue = [&](int64_t k0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int r, kk;
      coord(p.a_mode, e, r, kk);
      const int64_t i = m0 + r, k = k0 + kk;
      ra[e] = (i < p.m && k < p.k) ? S::fetch(p.a, i * p.a_rs + k * p.a_cs, p.a_bf16) : S::zero();
      coord(p.b_mode, e, r, kk);
      const int64_t j = n0 + r, kb = k0 + kk;
      rb[e] = (j < p.n && kb < p.k) ? S::fetch(p.b, kb * p.b_rs + j * p.b_cs, p.b_bf16) : S::zero();
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int r, kk;
      coord(p.a_mode, e, r, kk);
      As[r * P + kk] = ra[e];
      coord(p.b_mode, e, r, kk);
      Bs[r * P + kk] = rb[e];
    }
  };
  f32x16 acc[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  const int64_t nk = (p.k + BK - 1) / BK;
  if (nk > 0) {
    issue(0);
    commit();
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) issue((kt + 1) * BK);
    mma_step<MX16, BK, P, 1, T>(As, Bs, wm, wn0, lane & 31, lane >> 5, acc);
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }
  epilogue<1>(p, m0, n0, wm, wn0, lane, acc);
}

// staging mode of one operand: `ks` = its stride along k, `os` = along the other index (elements of `esz` bytes), `kext` /
// `oext` the extents: 1 / 2 = 16-byte loads along k / along the other dimension (rows 16-byte aligned, whole chunks only),
// 0 / 3 = element-wise walking k / the other dimension
int stage_mode(const void* ptr, int64_t os, int64_t ks, size_t esz, bool same_type, int64_t oext, int64_t kext) {
  const bool base16 = reinterpret_cast<uintptr_t>(ptr) % 16 == 0;
  const int64_t vec = static_cast<int64_t>(16 / esz);
  if (ks == 1) return (same_type && base16 && (os * static_cast<int64_t>(esz)) % 16 == 0 && kext % vec == 0) ? 1 : 0;
  if (os == 1) return (same_type && base16 && (ks * static_cast<int64_t>(esz)) % 16 == 0 && oext % vec == 0) ? 2 : 3;
  return 0;     // neither dimension contiguous: element-wise, walking k
}

}  // namespace

int gemm_launch(const void* a, int64_t a_rs, int64_t a_cs, int a_dtype, const void* b, int64_t b_rs, int64_t b_cs, int b_dtype,
                int64_t m, int n, int64_t k, float alpha, const float* alpha_dev, const float* bias, float beta,
                const void* addend, int64_t ldadd, int add_dtype, void* c, int64_t ldc, int c_dtype, hipStream_t st) {
  if (m == 0 || n == 0) return SGF_OK;
  const bool mx16 = a_dtype == SGF_BF16 && b_dtype == SGF_BF16;
  // 16-byte staging needs the operand in the staging type: bf16 operands in the bf16 form, fp32 operands in the fp32 form
  const size_t ea = a_dtype == SGF_BF16 ? 2 : 4, eb = b_dtype == SGF_BF16 ? 2 : 4;
  const int a_mode = stage_mode(a, a_rs, a_cs, ea, mx16 || a_dtype == SGF_F32, m, k);
  const int b_mode = stage_mode(b, b_cs, b_rs, eb, mx16 || b_dtype == SGF_F32, n, k);
  const bool vec = (a_mode == 1 || a_mode == 2) && (b_mode == 1 || b_mode == 2);
  const bool big = vec && m >= 8192;               // node-sized products: 128-row tiles; small matrices: more workgroups
  const int bm = big ? 128 : 64;
  const int64_t nb_m = (m + bm - 1) / bm;
  const int64_t nb_n = (n + GBN - 1) / GBN;
  SGF_REQUIRE(nb_m * nb_n < (int64_t{1} << 31), SGF_E_UNSUPPORTED, "sgf_gemm: %lld x %lld output tiles exceed one launch",
              static_cast<long long>(nb_m), static_cast<long long>(nb_n));
  GemmArgs p{a, a_rs, a_cs, b, b_rs, b_cs, a_dtype == SGF_BF16, b_dtype == SGF_BF16, a_mode, b_mode, m, n, k, alpha,
             alpha_dev, bias, beta, addend, ldadd, add_dtype == SGF_BF16, c, ldc, c_dtype == SGF_BF16, static_cast<int32_t>(nb_n)};
  const dim3 grid(static_cast<unsigned>(nb_m * nb_n));
  if (!vec) {
    if (mx16) hipLaunchKernelGGL((k_gemm_any<true>), grid, dim3(kGemmThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_any<false>), grid, dim3(kGemmThreads), 0, st, p);
  } else if (big) {
    if (mx16) hipLaunchKernelGGL((k_gemm_vec<true, 128>), grid, dim3(kGemmThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_vec<false, 128>), grid, dim3(kGemmThreads), 0, st, p);
  } else {
    if (mx16) hipLaunchKernelGGL((k_gemm_vec<true, 64>), grid, dim3(kGemmThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_vec<false, 64>), grid, dim3(kGemmThreads), 0, st, p);
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
