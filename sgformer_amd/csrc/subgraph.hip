// subgraph.hip — row N1 of SURVEY.md §8f: the induced-subgraph step of the reference's random-partition
// mini-batch trainer, on the GPU.
//
// Reference (large/main-batch.py:134-141, evaluator twin large/eval.py:80-96), per mini-batch, on the
// HOST:   idx_i = randperm(n)[...];  edge_index_i, _ = subgraph(idx_i, edge_index, num_nodes=n,
// relabel_nodes=True)   — torch_geometric 1.7.2 utils.subgraph (third party, not in the tree):
//     n_mask[subset] = 1;  n_idx[subset] = arange(len(subset))
//     mask = n_mask[row] & n_mask[col];  edge_index = edge_index[:, mask];  edge_index = n_idx[edge_index]
// i.e. an O(E) mask + filter over ALL edges for every batch (126 M edges per batch at ogbn-products
// scale), which dominates the epoch once the model itself is fast.
//
// Here: integer / byte work, HBM-bound, no MFMA.
//   k_mark  : relabel[subset[j]] = j (int32, -1 elsewhere) and a membership BITMASK (n/8 bytes: 306 KB
//             at products scale, L2-resident) so that the per-edge membership test never touches HBM.
//   k_count : each block owns a contiguous chunk of edges, reads (src, tgt) once, 16 B per edge fully
//             coalesced, tests both bits, reduces its kept count.
//   scan    : rocPRIM exclusive scan over the block counts (deterministic), total to a device scalar.
//   k_emit  : same chunks; kept edges are written at block offset + in-block rank (wave ballot +
//             popcount), so the output keeps the ORIGINAL EDGE ORDER exactly like the boolean-mask
//             indexing of the reference; node ids relabelled through `relabel` (an int32 gather that
//             only the ~(|subset|/n)^2 kept edges pay).  Optionally the kept edge positions.
// The result is bit-identical to the reference semantics (tests/test_gpu_kernels.py).
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sgf {
namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 2048;  // fixed grid: 8 blocks per CU; chunk = ceil(nnz / kBlocks) rounded to 256

struct SgPlan {
  size_t bits, counts, offsets, tmp, total, tmp_bytes;
};

int sg_plan(int64_t n, SgPlan* p) {
  size_t scan_bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<const int64_t*>(nullptr),
                                         static_cast<int64_t*>(nullptr), static_cast<int64_t>(0),
                                         static_cast<size_t>(kBlocks), rocprim::plus<int64_t>());
  if (e != hipSuccess) {
    set_error("rocprim::exclusive_scan size query failed: %s", hipGetErrorString(e));
    return SGF_E_HIP;
  }
  p->tmp_bytes = align_up(scan_bytes, 256) + 256;
  size_t off = 0;
  p->bits = off;
  off += align_up(static_cast<size_t>((n + 31) / 32) * 4, 256);
  p->counts = off;
  off += align_up(static_cast<size_t>(kBlocks) * 8, 256);
  p->offsets = off;
  off += align_up(static_cast<size_t>(kBlocks) * 8, 256);
  p->tmp = off;
  off += p->tmp_bytes;
  p->total = off;
  return SGF_OK;
}

inline int64_t chunk_of(int64_t nnz) {
  int64_t c = (nnz + kBlocks - 1) / kBlocks;
  return (c + kThreads - 1) / kThreads * kThreads;
}

__global__ void k_mark(const int64_t* __restrict__ subset, int64_t m, int64_t n,
                       int32_t* __restrict__ relabel, uint32_t* __restrict__ bits) {
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; j < m; j += stride) {
    const int64_t v = subset[j];
    if (v >= 0 && v < n) {
      relabel[v] = static_cast<int32_t>(j);
      atomicOr(&bits[v >> 5], 1u << (v & 31));
    }
  }
}

__device__ __forceinline__ bool member(const uint32_t* __restrict__ bits, int64_t v, int64_t n) {
  return v >= 0 && v < n && ((bits[v >> 5] >> (v & 31)) & 1u);
}

__global__ __launch_bounds__(kThreads) void k_count(const int64_t* __restrict__ src,
                                                    const int64_t* __restrict__ tgt, int64_t nnz,
                                                    int64_t n, int64_t chunk,
                                                    const uint32_t* __restrict__ bits,
                                                    int64_t* __restrict__ counts) {
  __shared__ int red[kThreads / 64];
  const int64_t e0 = static_cast<int64_t>(blockIdx.x) * chunk;
  int64_t e1 = e0 + chunk;
  if (e1 > nnz) e1 = nnz;
  int c = 0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += kThreads)
    c += (member(bits, src[e], n) && member(bits, tgt[e], n)) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t s = 0;
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    counts[blockIdx.x] = s;
  }
}

__global__ void k_total(const int64_t* __restrict__ counts, const int64_t* __restrict__ offsets,
                        int64_t* __restrict__ total) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = offsets[kBlocks - 1] + counts[kBlocks - 1];
}

__global__ __launch_bounds__(kThreads) void k_emit(const int64_t* __restrict__ src,
                                                   const int64_t* __restrict__ tgt, int64_t nnz,
                                                   int64_t n, int64_t chunk,
                                                   const uint32_t* __restrict__ bits,
                                                   const int32_t* __restrict__ relabel, int do_relabel,
                                                   const int64_t* __restrict__ offsets, int64_t total,
                                                   int64_t* __restrict__ out_src,
                                                   int64_t* __restrict__ out_tgt,
                                                   int64_t* __restrict__ out_eid) {
  __shared__ int wave_cnt[kThreads / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t e0 = static_cast<int64_t>(blockIdx.x) * chunk;
  int64_t e1 = e0 + chunk;
  if (e1 > nnz) e1 = nnz;
  int64_t base = offsets[blockIdx.x];
  for (int64_t r = e0; r < e1; r += kThreads) {   // uniform trip count per block (chunk % 256 == 0)
    const int64_t e = r + threadIdx.x;
    int64_t s = 0, t = 0;
    bool keep = false;
    if (e < e1) {
      s = src[e];
      t = tgt[e];
      keep = member(bits, s, n) && member(bits, t, n);
    }
    const unsigned long long ballot = __ballot(keep);
    const int rank = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(ballot);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
      const int c = wave_cnt[w];
      before += (w < wave) ? c : 0;
      all += c;
    }
    if (keep) {
      const int64_t pos = base + before + rank;
      if (pos < total) {
        out_src[pos] = do_relabel ? static_cast<int64_t>(relabel[s]) : s;
        out_tgt[pos] = do_relabel ? static_cast<int64_t>(relabel[t]) : t;
        if (out_eid != nullptr) out_eid[pos] = e;
      }
    }
    base += all;
    __syncthreads();
  }
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_subgraph_workspace_bytes(int64_t nnz, int64_t n) {
  (void)nnz;
  if (n < 0) return 0;
  SgPlan p;
  if (sg_plan(n, &p) != SGF_OK) return 0;
  return p.total;
}

extern "C" int sgf_subgraph_plan(const int64_t* edge_index, int64_t nnz, int64_t n,
                                 const int64_t* subset, int64_t m, int32_t* relabel, int64_t* total,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(nnz >= 0 && n >= 0 && m >= 0, SGF_E_INVALID, "sgf_subgraph_plan: negative size");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED, "sgf_subgraph_plan: n must be < 2^31");
  SGF_REQUIRE(total && (n == 0 || relabel) && (nnz == 0 || edge_index) && (m == 0 || subset), SGF_E_INVALID,
              "sgf_subgraph_plan: null pointer");
  SgPlan p;
  int rc = sg_plan(n, &p);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= p.total, SGF_E_WORKSPACE,
              "sgf_subgraph_plan: workspace %zu < %zu", workspace_bytes, p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  uint32_t* bits = reinterpret_cast<uint32_t*>(ws + p.bits);
  int64_t* counts = reinterpret_cast<int64_t*>(ws + p.counts);
  int64_t* offsets = reinterpret_cast<int64_t*>(ws + p.offsets);
  SGF_CHECK_HIP(hipMemsetAsync(bits, 0, p.counts - p.bits, st));
  if (n > 0) SGF_CHECK_HIP(hipMemsetAsync(relabel, 0xff, static_cast<size_t>(n) * 4, st));   // -1
  if (m > 0) {
    int64_t b = (m + kThreads - 1) / kThreads;
    if (b > kBlocks) b = kBlocks;
    hipLaunchKernelGGL(k_mark, dim3(static_cast<unsigned>(b)), dim3(kThreads), 0, st, subset, m, n, relabel,
                       bits);
    SGF_LAUNCH_CHECK();
  }
  const int64_t chunk = chunk_of(nnz);
  hipLaunchKernelGGL(k_count, dim3(kBlocks), dim3(kThreads), 0, st, edge_index, edge_index + nnz, nnz, n,
                     chunk, bits, counts);
  SGF_LAUNCH_CHECK();
  size_t bytes = p.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + p.tmp, bytes, counts, offsets, static_cast<int64_t>(0),
                                        static_cast<size_t>(kBlocks), rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(k_total, dim3(1), dim3(64), 0, st, counts, offsets, total);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_subgraph_emit(const int64_t* edge_index, int64_t nnz, int64_t n,
                                 const int32_t* relabel, int32_t relabel_nodes, int64_t total,
                                 int64_t* out, int64_t* out_eid, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(nnz >= 0 && n >= 0 && total >= 0, SGF_E_INVALID, "sgf_subgraph_emit: negative size");
  if (total == 0 || nnz == 0) return SGF_OK;
  SGF_REQUIRE(edge_index && relabel && out, SGF_E_INVALID, "sgf_subgraph_emit: null pointer");
  SgPlan p;
  int rc = sg_plan(n, &p);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= p.total, SGF_E_WORKSPACE,
              "sgf_subgraph_emit: workspace %zu < %zu", workspace_bytes, p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const uint32_t* bits = reinterpret_cast<const uint32_t*>(ws + p.bits);
  const int64_t* offsets = reinterpret_cast<const int64_t*>(ws + p.offsets);
  hipLaunchKernelGGL(k_emit, dim3(kBlocks), dim3(kThreads), 0, st, edge_index, edge_index + nnz, nnz, n,
                     chunk_of(nnz), bits, relabel, relabel_nodes, offsets, total, out, out + total, out_eid);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
