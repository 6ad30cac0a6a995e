// common.h — shared helpers for the gfx950 kernels of libsgf (not part of the C ABI).
#pragma once
#include <cstring>  // must precede rocprim (host memset)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <type_traits>

#include "../../include/sgf.h"

namespace sgf {

// ---- error reporting -----------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define SGF_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::sgf::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SGF_E_HIP;                                                                  \
    }                                                                                    \
  } while (0)

#define SGF_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      ::sgf::set_error(__VA_ARGS__);      \
      return (code);                      \
    }                                     \
  } while (0)

#define SGF_LAUNCH_CHECK()                                                          \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) {                                                         \
      ::sgf::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,            \
                       hipGetErrorString(_e));                                      \
      return SGF_E_HIP;                                                             \
    }                                                                               \
  } while (0)

// ---- experiment switches: environment variables read ONCE per process, not per launch --------------------------------
// sgf_reload_env() (include/sgf.h) makes every switch re-read its variable at its next use (probe scripts that flip a
// switch between launches call it after changing os.environ).
extern int g_env_epoch;
struct EnvInt {
  const char* name;
  int dflt;
  int value = 0;
  int epoch = -1;
  int get() {
    if (epoch != g_env_epoch) {
      const char* e = getenv(name);
      value = (e && *e) ? atoi(e) : dflt;
      epoch = g_env_epoch;
    }
    return value;
  }
};

// ---- chip constants (MI355X / gfx950) ------------------------------------------------------
constexpr int kWave = 64;      // wavefront width
constexpr int kNumCU = 256;    // 8 XCDs x 32 CUs
constexpr int kNumXCD = 8;

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- bf16 <-> fp32 (storage format only; all arithmetic is fp32) ----------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
// round-to-nearest-even, NaN preserved (same rule as torch.bfloat16 casts)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

struct alignas(8) bf16x4 {
  uint16_t v[4];
};

// 4-element vector load/store in the storage dtype, fp32 in registers.
template <typename T>
__device__ __forceinline__ float4 load4(const T* p);
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 load4<uint16_t>(const uint16_t* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  float4 f;
  f.x = __uint_as_float(r.x << 16);
  f.y = __uint_as_float(r.x & 0xffff0000u);
  f.z = __uint_as_float(r.y << 16);
  f.w = __uint_as_float(r.y & 0xffff0000u);
  return f;
}
template <typename T>
__device__ __forceinline__ void store4(T* p, float4 f);
template <>
__device__ __forceinline__ void store4<float>(float* p, float4 f) {
  *reinterpret_cast<float4*>(p) = f;
}
template <>
__device__ __forceinline__ void store4<uint16_t>(uint16_t* p, float4 f) {
  uint2 r;
  r.x = static_cast<uint32_t>(f32_to_bf16(f.x)) | (static_cast<uint32_t>(f32_to_bf16(f.y)) << 16);
  r.y = static_cast<uint32_t>(f32_to_bf16(f.z)) | (static_cast<uint32_t>(f32_to_bf16(f.w)) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<uint16_t>(const uint16_t* p) { return bf16_to_f32(*p); }
template <typename T>
__device__ __forceinline__ void store1(T* p, float f);
template <>
__device__ __forceinline__ void store1<float>(float* p, float f) { *p = f; }
template <>
__device__ __forceinline__ void store1<uint16_t>(uint16_t* p, float f) { *p = f32_to_bf16(f); }

// ---- T1: value of one stored entry of the normalised adjacency (large/ours.py:28-31), shared by csr.hip / subgraph_csr.hip ----
__device__ __forceinline__ float norm_value(int32_t deg_tgt, int32_t deg_src) {
#pragma clang fp contract(off)
  // (1. / d[col]).sqrt() and (1. / d[row]).sqrt(): correctly rounded IEEE div and sqrt
  // (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), then one rounded product.
  const float a = sqrtf(1.0f / static_cast<float>(deg_tgt));
  const float b = sqrtf(1.0f / static_cast<float>(deg_src));
  float v = a * b;
  // torch.nan_to_num(value, nan=0, posinf=0, neginf=0): zero in-degree of the source gives inf
  if (!(fabsf(v) <= 3.402823466e+38f)) v = 0.0f;
  return v;
}

// ---- wave-level reductions (64 lanes) -------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over all 64 lanes on the VALU's DPP path (no LDS traffic, no latency to hide), result uniform — the compiler keeps
// it in an SGPR.  Hillis-Steele inside each row of 16 lanes (row_shr 1/2/4/8, zero fill), then row_bcast:15 into rows 1, 3
// and row_bcast:31 into rows 2, 3: lane 63 holds the total.
__device__ __forceinline__ float wave_sum_uniform(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                 decltype(row_mask)::value, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- MFMA (exact-fp32 matrix cores; cdna_hip_programming.md §3) ------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C/D layout of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int mfma32_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// ---- per-wave streaming row kernels of csrc/rowgemm.hip, used by csrc/attn.hip for the attention-from-input passes ----
bool hrow_supported(int d, int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc,
                    const void* o, int64_t ldo);
size_t hrow_partial_bytes(int64_t n, int d);
int hrow_fwd(const void* h, int64_t ldh, int64_t n, int d, const float* M, const float* m, const float* w,
             const float* beta, void* out, int64_t ldo, float* den, hipStream_t st);
int hrow_bwd_pre(const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n, int d,
                 const float* M, const float* w, void* partial, float* rowscal, hipStream_t st);
int hrow_bwd_post(const void* h, int64_t ldh, int64_t n, int d, const float* Dm, const float* ds, const void* partial,
                  const void* addend, int64_t ldadd, void* dh, int64_t lddh, hipStream_t st);
int hrow_bwd(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den,
             int64_t n, int d, const float* M, const float* w, const float* Dm, const float* ds, void* dh, int64_t lddh,
             void* partial, hipStream_t st);

// ---- fp32-storage Linear layers on the exact-fp32 matrix cores (csrc/linear_f32.hip), used by csrc/rowgemm.hip ----
bool linear_f32_supported(int d_in, int d_out);
int linear_f32_blocks(int64_t n);
int linear_f32(const float* a, int64_t lda, int64_t n, int dk, int dj, const float* w, int64_t ldw, int trans_w,
               const float* bias, const float* addend, int64_t ldadd, const float* shift, float* out, int64_t ldo,
               float* spart, hipStream_t st);

int linear_f32_dual(const void* a, int64_t lda, const void* a2, int64_t lda2, float ca, float cb, int64_t n, int dk, int dj,
                    const float* w, int64_t ldw, int trans_w, const float* bias, void* out, int64_t ldo, void* out2,
                    int64_t ldo2, float co, float co2, int in16, int out16, hipStream_t st);

}  // namespace sgf
