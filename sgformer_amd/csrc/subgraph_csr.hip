// subgraph_csr.hip — rows N1 + T1 together: the induced subgraph of a mini-batch AND its normalised CSR, straight from the
// cached CSR of the parent graph (r05).
//
// Reference, per mini-batch (large/main-batch.py:134-143): `subgraph(idx_i, edge_index, relabel_nodes=True)` on the host
// (an O(E) mask over all 126 M edges of ogbn-products), then — inside every GraphConvLayer.forward — degree + argsort of the
// induced edge list (large/ours.py:26-33).  sgf_subgraph_* (subgraph.hip) already runs the first on the device as two
// streaming passes over ALL parent edges (2 x 2 GB per batch: 1.0 ms) and sgf_csr_build sorts the result again.
// A batch of m nodes only touches m rows of the parent CSR (m x 51 entries = 20 MB at products size), so here:
//   k_mark   : local_of[subset[j]] = j (int32 table over the parent's nodes, -1 elsewhere; atomicCAS, a repeated node raises
//              the duplicate flag and the caller falls back to sgf_subgraph_*);
//   k_count  : one wavefront per batch row walks that node's parent row, counts the sources that are in the batch
//              (ballot + popcount) = the row length AND the in-degree of the induced graph;
//   scan     : rocPRIM exclusive scan -> rowptr of the batch CSR, total to a device scalar (the one host read of the batch);
//   k_fill   : same walk, kept sources written as LOCAL ids at row offset + rank (ballot prefix), then
//   sort     : rocPRIM segmented radix sort of the local ids inside each row — the order (target, source) in LOCAL ids is what
//              sgf_csr_build produces for the induced edge list, duplicates kept;
//   k_values : val = sqrt(1 / d_t) * sqrt(1 / d_s) with the INDUCED in-degrees (norm_value, the same IEEE expression as
//              csr.hip), the edge list [2, total] in that order (row 0 sources, row 1 targets), and local_of reset to -1.
// Integer / byte work, latency-bound on short rows; no MFMA.  Output = bit for bit the arrays sgf_csr_build gives for
// torch_geometric's subgraph(idx, edge_index, relabel_nodes=True) (tests/test_gpu_r05.py); the edge LIST comes out in
// (target, source) order instead of the parent's edge order — the same multiset of edges, and the order no consumer of the
// path depends on (the model sorts it anyway, large/ours.py:33).
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sgf {
namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / kWave;

struct ScPlan {          // plan workspace: [dup flag | scan temp];  emit workspace: [off32 | tmpcol | sort temp]
  size_t flag, tmp, total, tmp_bytes;
};
struct ScEmit {
  size_t off32, tmpcol, tmp, total, tmp_bytes;
};

int sc_plan(int64_t m, ScPlan* p) {
  size_t scan_bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<const int32_t*>(nullptr), static_cast<int64_t*>(nullptr),
                                         static_cast<int64_t>(0), static_cast<size_t>(m + 1), rocprim::plus<int64_t>());
  if (e != hipSuccess) {
    set_error("rocprim::exclusive_scan size query failed: %s", hipGetErrorString(e));
    return SGF_E_HIP;
  }
  p->tmp_bytes = align_up(scan_bytes, 256) + 256;
  p->flag = 0;
  p->tmp = 256;
  p->total = 256 + p->tmp_bytes;
  return SGF_OK;
}

int sc_emit(int64_t m, int64_t total, ScEmit* p) {
  size_t sort_bytes = 0;
  hipError_t e = rocprim::segmented_radix_sort_keys(nullptr, sort_bytes, static_cast<const int32_t*>(nullptr),
                                                    static_cast<int32_t*>(nullptr), static_cast<unsigned int>(total),
                                                    static_cast<unsigned int>(m), static_cast<const int32_t*>(nullptr),
                                                    static_cast<const int32_t*>(nullptr), 0, 32);
  if (e != hipSuccess) {
    set_error("rocprim::segmented_radix_sort_keys size query failed: %s", hipGetErrorString(e));
    return SGF_E_HIP;
  }
  p->tmp_bytes = align_up(sort_bytes, 256) + 256;
  size_t off = 0;
  p->off32 = off;   off += align_up(static_cast<size_t>(m + 1) * 4, 256);
  p->tmpcol = off;  off += align_up(static_cast<size_t>(total) * 4, 256) + 256;
  p->tmp = off;     off += p->tmp_bytes;
  p->total = off;
  return SGF_OK;
}

__global__ __launch_bounds__(kThreads) void k_mark(const int64_t* __restrict__ subset, int64_t m, int64_t n,
                                                   int32_t* __restrict__ local_of, int32_t* __restrict__ dup) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < m; j += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t v = subset[j];
    if (v < 0 || v >= n) { *dup = 1; continue; }         // an id outside the graph: not this path's case either
    if (atomicCAS(&local_of[v], -1, static_cast<int32_t>(j)) != -1) *dup = 1;
  }
}

__global__ __launch_bounds__(kThreads) void k_unmark(const int64_t* __restrict__ subset, int64_t m, int64_t n,
                                                     int32_t* __restrict__ local_of) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < m; j += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t v = subset[j];
    if (v >= 0 && v < n) local_of[v] = -1;
  }
}

// one wavefront per batch row: count (FILL = false) or write (FILL = true) the parent row's sources that are in the batch
template <bool FILL>
__global__ __launch_bounds__(kThreads) void k_walk(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                   const int64_t* __restrict__ subset, int64_t m, int64_t n,
                                                   const int32_t* __restrict__ local_of, int32_t* __restrict__ cnt,
                                                   const int64_t* __restrict__ rowptr_b, int32_t* __restrict__ out,
                                                   int32_t* __restrict__ longest) {
  const int lane = threadIdx.x & (kWave - 1);
  int wave_max = 0;
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x / kWave);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t j = wave0; j < m; j += nwaves) {
    const int64_t v = subset[j];
    int64_t b = 0, e = 0;
    if (v >= 0 && v < n) { b = rowptr[v]; e = rowptr[v + 1]; }
    int64_t base = FILL ? rowptr_b[j] : 0;
    int total = 0;
    for (int64_t i = b; i < e; i += kWave) {
      const int64_t k = i + lane;
      int32_t loc = -1;
      if (k < e) {
        const int32_t s = colind[k];
        loc = local_of[s];
      }
      const unsigned long long mask = __ballot(loc >= 0);
      if (FILL) {
        if (loc >= 0) out[base + total + __popcll(mask & ((1ull << lane) - 1ull))] = loc;
      }
      total += __popcll(mask);
    }
    if (!FILL && lane == 0) cnt[j] = total;
    if (!FILL) wave_max = max(wave_max, total);
  }
  if (!FILL && lane == 0 && wave_max > 0) atomicMax(longest, wave_max);      // one per wave: the longest row of the batch CSR
}

__global__ __launch_bounds__(kThreads) void k_off32(const int64_t* __restrict__ rowptr_b, int64_t m1, int32_t* __restrict__ off32) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < m1; j += static_cast<int64_t>(gridDim.x) * kThreads)
    off32[j] = static_cast<int32_t>(rowptr_b[j]);
}

// total[0] = number of kept entries, total[1] = duplicate / out-of-range flag, total[2] = the longest row of the batch CSR
// (its largest induced in-degree: what the SpMM needs to know to skip its long-row path): the caller's one host read covers all
__global__ void k_total(const int64_t* __restrict__ rowptr_b, int64_t m, const int32_t* __restrict__ flags, int64_t* __restrict__ total) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    total[0] = rowptr_b[m];
    total[1] = flags[0];
    total[2] = flags[1];
  }
}

// per row: values from the induced in-degrees, the edge list in (target, source) order
__global__ __launch_bounds__(kThreads) void k_values(const int64_t* __restrict__ rowptr_b, const int32_t* __restrict__ colind_b,
                                                     const int32_t* __restrict__ deg, int64_t m, int64_t total,
                                                     float* __restrict__ val, int64_t* __restrict__ ei) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; j < m; j += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t b = rowptr_b[j], e = rowptr_b[j + 1];
    const int32_t dt = deg[j];
    for (int64_t i = b; i < e; ++i) {
      const int32_t s = colind_b[i];
      val[i] = norm_value(dt, deg[s]);
      if (ei) {
        ei[i] = s;
        ei[total + i] = j;
      }
    }
  }
}

inline int grid_rows(int64_t m, int per_block) {
  int64_t b = (m + per_block - 1) / per_block;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 16;
  if (b > cap) b = cap;
  return b < 1 ? 1 : static_cast<int>(b);
}

}  // namespace
}  // namespace sgf

extern "C" size_t sgf_subgraph_csr_plan_workspace_bytes(int64_t m) {
  if (m < 0) return 0;
  sgf::ScPlan p;
  if (sgf::sc_plan(m, &p) != SGF_OK) return 0;
  return p.total;
}

extern "C" size_t sgf_subgraph_csr_emit_workspace_bytes(int64_t m, int64_t total) {
  if (m < 0 || total < 0) return 0;
  sgf::ScEmit p;
  if (sgf::sc_emit(m, total, &p) != SGF_OK) return 0;
  return p.total;
}

extern "C" int sgf_subgraph_csr_plan(const int64_t* rowptr, const int32_t* colind, int64_t n, const int64_t* subset, int64_t m,
                                     int32_t* local_of, int64_t* rowptr_b, int32_t* deg_b, int64_t* total, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(n >= 0 && m >= 0, SGF_E_INVALID, "sgf_subgraph_csr_plan: negative size");
  SGF_REQUIRE(n < (int64_t{1} << 31) && m < (int64_t{1} << 31), SGF_E_UNSUPPORTED, "sgf_subgraph_csr_plan: sizes must be < 2^31");
  SGF_REQUIRE(rowptr && colind && local_of && rowptr_b && deg_b && total && (m == 0 || subset), SGF_E_INVALID,
              "sgf_subgraph_csr_plan: null pointer");
  ScPlan p;
  int rc = sc_plan(m, &p);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= p.total, SGF_E_WORKSPACE, "sgf_subgraph_csr_plan: workspace %zu < %zu", workspace_bytes,
              p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  int32_t* dup = reinterpret_cast<int32_t*>(ws + p.flag);
  SGF_CHECK_HIP(hipMemsetAsync(dup, 0, 8, st));               // [duplicate flag | longest row]
  SGF_CHECK_HIP(hipMemsetAsync(deg_b + m, 0, 4, st));          // deg_b has m + 1 slots: the scan's last input is 0
  if (m > 0) {
    hipLaunchKernelGGL(k_mark, dim3(grid_rows(m, kThreads)), dim3(kThreads), 0, st, subset, m, n, local_of, dup);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_walk<false>), dim3(grid_rows(m, kWavesPerBlock)), dim3(kThreads), 0, st, rowptr, colind, subset, m, n,
                       local_of, deg_b, static_cast<const int64_t*>(nullptr), static_cast<int32_t*>(nullptr), dup + 1);
    SGF_LAUNCH_CHECK();
  }
  size_t tb = p.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + p.tmp, tb, deg_b, rowptr_b, static_cast<int64_t>(0), static_cast<size_t>(m + 1),
                                        rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(k_total, dim3(1), dim3(64), 0, st, rowptr_b, m, dup, total);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_subgraph_csr_emit(const int64_t* rowptr, const int32_t* colind, int64_t n, const int64_t* subset, int64_t m,
                                     int32_t* local_of, const int64_t* rowptr_b, const int32_t* deg_b, int64_t total,
                                     int32_t* colind_b, float* val_b, int64_t* edge_index_b, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(n >= 0 && m >= 0 && total >= 0 && total < (int64_t{1} << 31), SGF_E_INVALID, "sgf_subgraph_csr_emit: bad sizes (total %lld)",
              static_cast<long long>(total));
  SGF_REQUIRE(rowptr && colind && local_of && rowptr_b && deg_b && (m == 0 || subset) && (total == 0 || (colind_b && val_b)),
              SGF_E_INVALID, "sgf_subgraph_csr_emit: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (m > 0 && total > 0) {
    ScEmit p;
    int rc = sc_emit(m, total, &p);
    if (rc != SGF_OK) return rc;
    SGF_REQUIRE(workspace && workspace_bytes >= p.total, SGF_E_WORKSPACE, "sgf_subgraph_csr_emit: workspace %zu < %zu", workspace_bytes,
                p.total);
    char* ws = static_cast<char*>(workspace);
    int32_t* off32 = reinterpret_cast<int32_t*>(ws + p.off32);
    int32_t* tmpcol = reinterpret_cast<int32_t*>(ws + p.tmpcol);
    hipLaunchKernelGGL(k_off32, dim3(grid_rows(m + 1, kThreads)), dim3(kThreads), 0, st, rowptr_b, m + 1, off32);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_walk<true>), dim3(grid_rows(m, kWavesPerBlock)), dim3(kThreads), 0, st, rowptr, colind, subset, m, n,
                       local_of, static_cast<int32_t*>(nullptr), rowptr_b, tmpcol, static_cast<int32_t*>(nullptr));
    SGF_LAUNCH_CHECK();
    int bits = 1;
    while ((int64_t{1} << bits) < m) ++bits;
    size_t tb = p.tmp_bytes;
    SGF_CHECK_HIP(rocprim::segmented_radix_sort_keys(ws + p.tmp, tb, tmpcol, colind_b, static_cast<unsigned int>(total),
                                                     static_cast<unsigned int>(m), off32, off32 + 1, 0, bits, st));
    hipLaunchKernelGGL(k_values, dim3(grid_rows(m, kThreads)), dim3(kThreads), 0, st, rowptr_b, colind_b, deg_b, m, total, val_b,
                       edge_index_b);
    SGF_LAUNCH_CHECK();
  }
  if (m > 0) {
    hipLaunchKernelGGL(k_unmark, dim3(grid_rows(m, kThreads)), dim3(kThreads), 0, st, subset, m, n, local_of);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}
