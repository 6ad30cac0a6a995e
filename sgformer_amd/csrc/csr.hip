// csr.hip — T1: symmetric-normalised adjacency in CSR, built once on the GPU.
//
// Reference arithmetic (large/ours.py:26-33, identical in 100M/ours.py:72-79):
//     row, col = edge_index
//     d = degree(col, N).float()                       # in-degree, counts every stored edge
//     value = 1 * sqrt(1/d[col]) * sqrt(1/d[row]);  nan_to_num(value, 0, 0, 0)
//     adj = SparseTensor(row=col, col=row, value=value)   # A[col_e,row_e], sorted by (col,row)
// The reference redoes this (an argsort over nnz edges) in every GraphConvLayer.forward; here the
// result is a plain CSR the caller caches per edge_index.
//
// Pipeline (all on `stream`, no host sync):
//   1. in-degree histogram (int32 atomics; L2 atomics on gfx950 are device-scope)
//   2. rowptr = exclusive scan of the histogram (int64)
//   3. 64-bit keys (tgt << 32 | src), LSD radix sort on the 32 + ceil(log2 n) significant bits.
//      Equal keys are identical edges with identical values, so stability is immaterial and the
//      output equals the stable sort by tgt*N+src bit for bit.
//   4. colind = low word; val from the two degrees with IEEE div / sqrt / mul.
// The sort and the scan use rocPRIM device primitives (the ROCm-native building blocks);
// everything else is hand-written below.
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sgf {
namespace {

constexpr int kThreads = 256;

__global__ void k_in_degree(const int64_t* __restrict__ tgt, int64_t nnz, int64_t n,
                            int32_t* __restrict__ deg) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < nnz; i += stride) {
    const int64_t t = tgt[i];
    if (t >= 0 && t < n) atomicAdd(&deg[t], 1);
  }
}

// key = (hi << 32) | lo
__global__ void k_make_keys(const int64_t* __restrict__ hi, const int64_t* __restrict__ lo,
                            int64_t nnz, uint64_t* __restrict__ keys) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < nnz; i += stride) {
    keys[i] = (static_cast<uint64_t>(hi[i]) << 32) | (static_cast<uint64_t>(lo[i]) & 0xffffffffull);
  }
}

// sorted key = (row_of_A << 32) | col_of_A ; deg is always the IN-degree array.
// kTransposed = false: row_of_A = tgt, col_of_A = src.   true: row = src, col = tgt.
template <bool kTransposed>
__global__ void k_finalize(const uint64_t* __restrict__ keys, int64_t nnz, int64_t n,
                           const int32_t* __restrict__ deg, int32_t* __restrict__ colind,
                           float* __restrict__ val) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint64_t k = keys[i];
    const int64_t r = static_cast<int64_t>(k >> 32);
    const int64_t c = static_cast<int64_t>(k & 0xffffffffull);
    colind[i] = static_cast<int32_t>(c);
    const int64_t t = kTransposed ? c : r;
    const int64_t s = kTransposed ? r : c;
    const bool ok = (t >= 0 && t < n && s >= 0 && s < n);
    val[i] = ok ? norm_value(deg[t], deg[s]) : 0.0f;
  }
}

__global__ void k_set_last(int64_t* rowptr, int64_t n, int64_t nnz) {
  if (blockIdx.x == 0 && threadIdx.x == 0) rowptr[n] = nnz;
}

__global__ void k_out_degree(const int64_t* __restrict__ src, int64_t nnz, int64_t n,
                             int32_t* __restrict__ odeg) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < nnz; i += stride) {
    const int64_t s = src[i];
    if (s >= 0 && s < n) atomicAdd(&odeg[s], 1);
  }
}

__global__ void k_set_flag(int32_t* flag, int32_t v) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag = v;
}

// A^T == A  <=>  the two sorted key arrays coincide  <=>  rowptr and colind coincide.
__global__ void k_compare(const int64_t* __restrict__ rp_a, const int64_t* __restrict__ rp_b,
                          int64_t n1, const int32_t* __restrict__ ci_a,
                          const int32_t* __restrict__ ci_b, int64_t nnz,
                          int32_t* __restrict__ flag) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool diff = false;
  for (int64_t j = i; j < n1; j += stride) diff |= (rp_a[j] != rp_b[j]);
  for (int64_t j = i; j < nnz; j += stride) diff |= (ci_a[j] != ci_b[j]);
  if (diff) *flag = 0;  // benign race: every writer stores the same value
}

struct I32ToI64 {
  __host__ __device__ int64_t operator()(int32_t v) const { return static_cast<int64_t>(v); }
};

inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;  // grid-stride the rest
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

inline unsigned key_bits(int64_t n) {
  unsigned b = 1;
  while ((static_cast<int64_t>(1) << b) < n && b < 31) ++b;
  return 32u + b;
}

struct Plan {
  size_t keys_a, keys_b, tmp, scratch_deg, total, tmp_bytes;
};

// Workspace = [keys_a | keys_b | scratch_deg (n int32, transpose only) | rocprim temp]
int make_plan(int64_t nnz, int64_t n, Plan* p) {
  size_t sort_bytes = 0, scan_bytes = 0;
  if (nnz > 0) {
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, static_cast<uint64_t*>(nullptr),
                                            static_cast<uint64_t*>(nullptr),
                                            static_cast<size_t>(nnz), 0u, key_bits(n));
    if (e != hipSuccess) {
      set_error("rocprim::radix_sort_keys size query failed: %s", hipGetErrorString(e));
      return SGF_E_HIP;
    }
  }
  if (n > 0) {
    auto in = rocprim::make_transform_iterator(static_cast<const int32_t*>(nullptr), I32ToI64());
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, in, static_cast<int64_t*>(nullptr),
                                           static_cast<int64_t>(0), static_cast<size_t>(n),
                                           rocprim::plus<int64_t>());
    if (e != hipSuccess) {
      set_error("rocprim::exclusive_scan size query failed: %s", hipGetErrorString(e));
      return SGF_E_HIP;
    }
  }
  p->tmp_bytes = align_up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes, 256) + 256;
  size_t off = 0;
  p->keys_a = off;
  off += align_up(static_cast<size_t>(nnz) * 8, 256);
  p->keys_b = off;
  off += align_up(static_cast<size_t>(nnz) * 8, 256);
  p->scratch_deg = off;
  off += align_up(static_cast<size_t>(n) * 4, 256);
  p->tmp = off;
  off += p->tmp_bytes;
  p->total = off;
  return SGF_OK;
}

// histogram already in `counts`; writes ptr[0..n], sorted keys left in keys_b.
int scan_and_sort(const int64_t* hi, const int64_t* lo, int64_t nnz, int64_t n,
                  const int32_t* counts, int64_t* ptr, char* ws, const Plan& p, hipStream_t st) {
  if (n > 0) {
    size_t bytes = p.tmp_bytes;
    auto in = rocprim::make_transform_iterator(counts, I32ToI64());
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + p.tmp, bytes, in, ptr, static_cast<int64_t>(0),
                                          static_cast<size_t>(n), rocprim::plus<int64_t>(), st));
  }
  hipLaunchKernelGGL(k_set_last, dim3(1), dim3(64), 0, st, ptr, n, nnz);
  SGF_LAUNCH_CHECK();
  if (nnz > 0) {
    uint64_t* ka = reinterpret_cast<uint64_t*>(ws + p.keys_a);
    uint64_t* kb = reinterpret_cast<uint64_t*>(ws + p.keys_b);
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(nnz)), dim3(kThreads), 0, st, hi, lo, nnz, ka);
    SGF_LAUNCH_CHECK();
    size_t bytes = p.tmp_bytes;
    SGF_CHECK_HIP(rocprim::radix_sort_keys(ws + p.tmp, bytes, ka, kb, static_cast<size_t>(nnz),
                                           0u, key_bits(n), st));
  }
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_csr_workspace_bytes(int64_t nnz, int64_t n) {
  if (nnz < 0 || n < 0) return 0;
  Plan p;
  if (make_plan(nnz, n, &p) != SGF_OK) return 0;
  return p.total;
}

extern "C" int sgf_csr_build(const int64_t* edge_index, int64_t nnz, int64_t n, int64_t* rowptr,
                             int32_t* colind, float* val, int32_t* deg, void* workspace,
                             size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(nnz >= 0 && n >= 0, SGF_E_INVALID, "sgf_csr_build: negative size");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED,
              "sgf_csr_build: n=%lld needs 64-bit column indices (int32 colind only)",
              static_cast<long long>(n));
  SGF_REQUIRE(rowptr && (nnz == 0 || (edge_index && colind && val)) && (n == 0 || deg),
              SGF_E_INVALID, "sgf_csr_build: null pointer");
  Plan p;
  int rc = make_plan(nnz, n, &p);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace_bytes >= p.total && (workspace || p.total == 0), SGF_E_WORKSPACE,
              "sgf_csr_build: workspace %zu < %zu", workspace_bytes, p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const int64_t* src = edge_index;
  const int64_t* tgt = edge_index + nnz;

  if (n > 0) SGF_CHECK_HIP(hipMemsetAsync(deg, 0, static_cast<size_t>(n) * 4, st));
  if (nnz > 0) {
    hipLaunchKernelGGL(k_in_degree, dim3(grid_for(nnz)), dim3(kThreads), 0, st, tgt, nnz, n, deg);
    SGF_LAUNCH_CHECK();
  }
  rc = scan_and_sort(tgt, src, nnz, n, deg, rowptr, ws, p, st);
  if (rc != SGF_OK) return rc;
  if (nnz > 0) {
    const uint64_t* kb = reinterpret_cast<const uint64_t*>(ws + p.keys_b);
    hipLaunchKernelGGL(k_finalize<false>, dim3(grid_for(nnz)), dim3(kThreads), 0, st, kb, nnz, n,
                       deg, colind, val);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}

extern "C" int sgf_csr_transpose(const int64_t* edge_index, int64_t nnz, int64_t n,
                                 const int32_t* deg, const int64_t* rowptr, const int32_t* colind,
                                 int64_t* t_rowptr, int32_t* t_colind, float* t_val,
                                 int32_t* is_symmetric, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  SGF_REQUIRE(nnz >= 0 && n >= 0, SGF_E_INVALID, "sgf_csr_transpose: negative size");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED,
              "sgf_csr_transpose: n too large for int32 colind");
  SGF_REQUIRE(t_rowptr && is_symmetric && rowptr &&
                  (nnz == 0 || (edge_index && t_colind && t_val && colind)) && (n == 0 || deg),
              SGF_E_INVALID, "sgf_csr_transpose: null pointer");
  Plan p;
  int rc = make_plan(nnz, n, &p);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace_bytes >= p.total && (workspace || p.total == 0), SGF_E_WORKSPACE,
              "sgf_csr_transpose: workspace %zu < %zu", workspace_bytes, p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const int64_t* src = edge_index;
  const int64_t* tgt = edge_index + nnz;
  int32_t* odeg = reinterpret_cast<int32_t*>(ws + p.scratch_deg);

  if (n > 0) SGF_CHECK_HIP(hipMemsetAsync(odeg, 0, static_cast<size_t>(n) * 4, st));
  if (nnz > 0) {
    hipLaunchKernelGGL(k_out_degree, dim3(grid_for(nnz)), dim3(kThreads), 0, st, src, nnz, n, odeg);
    SGF_LAUNCH_CHECK();
  }
  rc = scan_and_sort(src, tgt, nnz, n, odeg, t_rowptr, ws, p, st);
  if (rc != SGF_OK) return rc;
  if (nnz > 0) {
    const uint64_t* kb = reinterpret_cast<const uint64_t*>(ws + p.keys_b);
    hipLaunchKernelGGL(k_finalize<true>, dim3(grid_for(nnz)), dim3(kThreads), 0, st, kb, nnz, n,
                       deg, t_colind, t_val);
    SGF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, st, is_symmetric, 1);
  SGF_LAUNCH_CHECK();
  const int64_t work = (n + 1 > nnz) ? n + 1 : nnz;
  hipLaunchKernelGGL(k_compare, dim3(grid_for(work)), dim3(kThreads), 0, st, rowptr, t_rowptr,
                     n + 1, colind, t_colind, nnz, is_symmetric);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
