// prologue.hip — N2: the trainer's graph prologue on the GPU.
//
// Reference (large/main.py:75-79; 100M/nb-sample.py:79-80 is the same minus remove_self_loops), all three
// third-party torch_geometric 1.7.2 utilities run on the HOST by the trainers, before `.to(device)`:
//     if not args.directed: edge_index = to_undirected(edge_index)     # cat both directions, coalesce:
//                                                                      #   sort by row*N+col, drop duplicates
//     edge_index, _ = remove_self_loops(edge_index)                    # mask row != col, order kept
//     edge_index, _ = add_self_loops(edge_index, num_nodes=n)          # append (i, i) for i in range(n)
// O(E log E) on the host for ogbn-products (62 M pairs) or papers100M (1.6 B); here one radix sort of
// 64-bit (row << 32 | col) keys + a flag scan + one scatter, on the device.
//
// Output size is data dependent, so — like sgf_subgraph_* — there are two calls around one host read:
//   sgf_graph_prologue_plan : *total (device int64) = number of output edges; sorted keys / scan stay in
//                             the workspace, which must be handed unchanged to
//   sgf_graph_prologue_emit : out int64 [2, total].
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sgf {
namespace {

constexpr int kThreads = 256;

inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

inline unsigned bits_for(int64_t n) {
  unsigned b = 1;
  while ((static_cast<int64_t>(1) << b) < n && b < 31) ++b;
  return b;
}

// keys[i] = (row << 32) | col for i < m; with `sym` also keys[m + i] = (col << 32) | row
__global__ void k_pro_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t m, int sym,
                           uint64_t* __restrict__ keys) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < m; i += stride) {
    const uint64_t r = static_cast<uint64_t>(row[i]) & 0xffffffffull, c = static_cast<uint64_t>(col[i]) & 0xffffffffull;
    keys[i] = (r << 32) | c;
    if (sym) keys[m + i] = (c << 32) | r;
  }
}

// flag[i] = keep entry i: (first of its run, if `dedup`) and (not a self-loop, if `drop_loops`)
__global__ void k_pro_flags(const uint64_t* __restrict__ keys, int64_t k, int dedup, int drop_loops,
                            uint32_t* __restrict__ flag) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < k; i += stride) {
    const uint64_t key = keys[i];
    bool keep = true;
    if (dedup && i > 0 && keys[i - 1] == key) keep = false;
    if (drop_loops && (key >> 32) == (key & 0xffffffffull)) keep = false;
    flag[i] = keep ? 1u : 0u;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) flag[k] = 0u;   // scan runs over k + 1 entries: scan[k] = count
}

__global__ void k_pro_total(const uint32_t* __restrict__ scan, int64_t k, int64_t loops, int64_t* total) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = static_cast<int64_t>(scan[k]) + loops;
}

__global__ void k_pro_emit(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flag,
                           const uint32_t* __restrict__ scan, int64_t k, int64_t total, int64_t n_loops,
                           int64_t* __restrict__ out) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t kept = total - n_loops;
  for (int64_t j = i; j < k; j += stride) {
    if (flag[j]) {
      const int64_t p = scan[j];
      out[p] = static_cast<int64_t>(keys[j] >> 32);
      out[total + p] = static_cast<int64_t>(keys[j] & 0xffffffffull);
    }
  }
  for (int64_t j = i; j < n_loops; j += stride) {
    out[kept + j] = j;
    out[total + kept + j] = j;
  }
}

struct Layout {
  size_t keys_a, keys_b, flag, scan, tmp, total, tmp_bytes;
};

int make_layout(int64_t m, Layout* L) {
  const size_t k = static_cast<size_t>(2 * m);
  size_t sort_b = 0, scan_b = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, sort_b, static_cast<uint64_t*>(nullptr),
                                          static_cast<uint64_t*>(nullptr), k, 0u, 64u);
  if (e != hipSuccess) { set_error("rocprim sort size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::exclusive_scan(nullptr, scan_b, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u,
                              k + 1, rocprim::plus<uint32_t>());
  if (e != hipSuccess) { set_error("rocprim scan size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  L->tmp_bytes = align_up(sort_b > scan_b ? sort_b : scan_b, 256) + 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L->keys_a = take(k * 8);
  L->keys_b = take(k * 8);
  L->flag = take((k + 1) * 4);
  L->scan = take((k + 1) * 4);
  L->tmp = take(L->tmp_bytes);
  L->total = off;
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_graph_prologue_workspace_bytes(int64_t m, int64_t n) {
  if (m < 0 || n < 0) return 0;
  Layout L;
  if (make_layout(m, &L) != SGF_OK) return 0;
  return L.total;
}

extern "C" int sgf_graph_prologue_plan(const int64_t* edge_index, int64_t m, int64_t n, int32_t to_undirected,
                                       int32_t remove_self_loops, int32_t add_self_loops, int64_t* total,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(m >= 0 && n >= 0, SGF_E_INVALID, "sgf_graph_prologue_plan: negative size");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED, "sgf_graph_prologue_plan: n >= 2^31");
  SGF_REQUIRE(2 * m < (static_cast<int64_t>(1) << 32) - 1, SGF_E_UNSUPPORTED,
              "sgf_graph_prologue_plan: more than 2^32 directed entries (32-bit positions)");
  SGF_REQUIRE(total && (m == 0 || edge_index), SGF_E_INVALID, "sgf_graph_prologue_plan: null pointer");
  Layout L;
  int rc = make_layout(m, &L);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= L.total, SGF_E_WORKSPACE,
              "sgf_graph_prologue_plan: workspace %zu < %zu", workspace_bytes, L.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  uint64_t* ka = reinterpret_cast<uint64_t*>(ws + L.keys_a);
  uint64_t* kb = reinterpret_cast<uint64_t*>(ws + L.keys_b);
  uint32_t* flag = reinterpret_cast<uint32_t*>(ws + L.flag);
  uint32_t* scan = reinterpret_cast<uint32_t*>(ws + L.scan);
  const int sym = to_undirected ? 1 : 0;
  const int64_t k = sym ? 2 * m : m;
  if (m > 0) {
    hipLaunchKernelGGL(k_pro_keys, dim3(grid_for(m)), dim3(kThreads), 0, st, edge_index, edge_index + m, m, sym, ka);
    SGF_LAUNCH_CHECK();
  }
  // coalesce = sort by (row, col) + drop duplicates; without to_undirected the order is the caller's
  const uint64_t* keys = ka;
  if (sym && k > 0) {
    size_t bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::radix_sort_keys(ws + L.tmp, bytes, ka, kb, static_cast<size_t>(k), 0u, 32u + bits_for(n), st));
    keys = kb;
  }
  hipLaunchKernelGGL(k_pro_flags, dim3(grid_for(k > 0 ? k : 1)), dim3(kThreads), 0, st, keys, k, sym,
                     remove_self_loops ? 1 : 0, flag);
  SGF_LAUNCH_CHECK();
  size_t bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, flag, scan, 0u, static_cast<size_t>(k + 1),
                                        rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_pro_total, dim3(1), dim3(64), 0, st, scan, k, add_self_loops ? n : 0, total);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_graph_prologue_emit(int64_t m, int64_t n, int32_t to_undirected, int32_t add_self_loops,
                                       int64_t total, int64_t* out, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  SGF_REQUIRE(m >= 0 && n >= 0 && total >= 0, SGF_E_INVALID, "sgf_graph_prologue_emit: negative size");
  if (total == 0) return SGF_OK;
  SGF_REQUIRE(out, SGF_E_INVALID, "sgf_graph_prologue_emit: null pointer");
  Layout L;
  int rc = make_layout(m, &L);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= L.total, SGF_E_WORKSPACE,
              "sgf_graph_prologue_emit: workspace %zu < %zu", workspace_bytes, L.total);
  const int64_t loops = add_self_loops ? n : 0;
  SGF_REQUIRE(total >= loops, SGF_E_INVALID, "sgf_graph_prologue_emit: total smaller than the self-loops");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const int64_t k = to_undirected ? 2 * m : m;
  const uint64_t* keys = reinterpret_cast<const uint64_t*>(ws + (to_undirected && k > 0 ? L.keys_b : L.keys_a));
  const uint32_t* flag = reinterpret_cast<const uint32_t*>(ws + L.flag);
  const uint32_t* scan = reinterpret_cast<const uint32_t*>(ws + L.scan);
  const int64_t work = k > loops ? k : loops;
  hipLaunchKernelGGL(k_pro_emit, dim3(grid_for(work)), dim3(kThreads), 0, st, keys, flag, scan, k, total, loops, out);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
