// reorder.hip — a locality-restoring node order for the cached CSR (T2's on-chip reuse).
//
// The SpMM of large/ours.py:34 gathers one d-wide row of X per stored entry.  How many of those
// gathers hit the XCD's L2 — or can be served from LDS by the row-block kernel of spmm_blocked.hip —
// depends on the node numbering, which for a dataset is arbitrary (ogbn ids, a random split, ...).
// This file computes a numbering that puts the nodes of one community next to each other and related
// communities next to each other, from the graph alone:
//
//   level 1  synchronous label propagation on the nodes: every node adopts the label carried by most
//            of its in-neighbours (ties -> the smallest label), `iters1` rounds from label[v] = v;
//   level 2  the same on the communities (nodes = level-1 labels, edges = the inter-community
//            edges, so a community adopts the label most of its OUTSIDE edges lead to), `iters2` rounds;
//   order    nodes sorted by (level-2 label, level-1 community, node id).
//
// One round = one radix sort of nnz packed (target, label-of-source) keys + one reduce-by-key (vote
// counts) + a 64-bit atomicMax per (target, label) pair — sort-based so that hub rows cost nothing
// special and the result is deterministic (max is order-independent).  rocPRIM supplies the sort /
// reduce-by-key / scan; everything else is below.  ~30 ms per round at ogbn-products scale, once per graph.
//
// The caller (sgformer_amd/ops.py) relabels edge_index with `inv`, rebuilds the CSR with
// sgf_csr_build (the ORIGINAL-order CSR stays bit-exact and is what the T1 tests check), permutes the
// rows of x once at the module boundary and un-permutes the logits: everything in between is
// permutation-equivariant (SpMM, both attention reductions, BatchNorm statistics).
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

inline unsigned bits_for(int64_t n) {   // smallest B with 2^B - 1 >= n  (so that 2^B - 1 is a free sentinel)
  unsigned b = 1;
  while (((static_cast<int64_t>(1) << b) - 1) < n && b < 31) ++b;
  return b;
}

// key = (target' << B) | label[source'];  x' = map ? map[x] : x.  Edges with ids outside [0, n), and
// (exclude_self) edges with source' == target', get the all-ones sentinel and sort to the end.
__global__ void k_vote_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t m,
                            int64_t n, const int32_t* __restrict__ map, const uint32_t* __restrict__ label,
                            int exclude_self, unsigned B, uint64_t* __restrict__ keys) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t sentinel = (static_cast<uint64_t>(1) << (2 * B)) - 1;
  for (; i < m; i += stride) {
    const int64_t s = src[i], t = dst[i];
    uint64_t k = sentinel;
    if (s >= 0 && s < n && t >= 0 && t < n) {
      const uint32_t su = map ? static_cast<uint32_t>(map[s]) : static_cast<uint32_t>(s);
      const uint32_t tu = map ? static_cast<uint32_t>(map[t]) : static_cast<uint32_t>(t);
      if (!(exclude_self && su == tu)) k = (static_cast<uint64_t>(tu) << B) | label[su];
    }
    keys[i] = k;
  }
}

// best[target] = max over its (label, votes) pairs of (votes << 32) | ~label  -> most votes, ties -> smallest label
__global__ void k_vote_argmax(const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ ucnt,
                              const uint32_t* __restrict__ n_unique, int64_t n, unsigned B,
                              unsigned long long* __restrict__ best) {
  const int64_t u = *n_unique;
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t mask = (static_cast<uint64_t>(1) << B) - 1;
  for (; j < u; j += stride) {
    const uint64_t k = ukeys[j];
    const uint64_t t = k >> B;
    if (t >= static_cast<uint64_t>(n)) continue;   // sentinel run
    const uint32_t lab = static_cast<uint32_t>(k & mask);
    atomicMax(&best[t], (static_cast<unsigned long long>(ucnt[j]) << 32) | (0xffffffffu - lab));
  }
}

__global__ void k_vote_assign(const unsigned long long* __restrict__ best, int64_t n,
                              const uint32_t* __restrict__ label, uint32_t* __restrict__ out) {
  int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; v < n; v += stride) {
    const unsigned long long b = best[v];
    out[v] = (b >> 32) ? 0xffffffffu - static_cast<uint32_t>(b & 0xffffffffull) : label[v];
  }
}

__global__ void k_iota(uint32_t* p, int64_t n) {
  int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; v < n; v += stride) p[v] = static_cast<uint32_t>(v);
}

__global__ void k_mark(const uint32_t* __restrict__ label, int64_t n, uint32_t* __restrict__ used) {
  int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; v < n; v += stride) used[label[v]] = 1u;   // benign race: every writer stores 1
}

// cid[v] = rank of label[v] among the labels in use
__global__ void k_compact(const uint32_t* __restrict__ label, const uint32_t* __restrict__ rank_of_label,
                          int64_t n, int32_t* __restrict__ cid) {
  int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; v < n; v += stride) cid[v] = static_cast<int32_t>(rank_of_label[label[v]]);
}

// community order key: (level-2 label of c) << B | c   for c < n (communities beyond the count in use are
// harmless: they sort by their own id and no node refers to them)
__global__ void k_comm_keys(const uint32_t* __restrict__ label2, int64_t n, unsigned B,
                            uint64_t* __restrict__ keys) {
  int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; c < n; c += stride) keys[c] = (static_cast<uint64_t>(label2[c]) << B) | static_cast<uint64_t>(c);
}

__global__ void k_comm_rank(const uint64_t* __restrict__ sorted, int64_t n, unsigned B,
                            uint32_t* __restrict__ rank) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t mask = (static_cast<uint64_t>(1) << B) - 1;
  for (; p < n; p += stride) rank[sorted[p] & mask] = static_cast<uint32_t>(p);
}

__global__ void k_node_keys(const int32_t* __restrict__ cid, const uint32_t* __restrict__ comm_rank,
                            int64_t n, unsigned B, uint64_t* __restrict__ keys) {
  int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; v < n; v += stride)
    keys[v] = (static_cast<uint64_t>(comm_rank[cid[v]]) << B) | static_cast<uint64_t>(v);
}

__global__ void k_perm(const uint64_t* __restrict__ sorted, int64_t n, unsigned B, int32_t* __restrict__ perm,
                       int32_t* __restrict__ inv) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t mask = (static_cast<uint64_t>(1) << B) - 1;
  for (; p < n; p += stride) {
    const int32_t v = static_cast<int32_t>(sorted[p] & mask);
    perm[p] = v;
    inv[v] = static_cast<int32_t>(p);
  }
}

struct Layout {
  size_t keys_a, keys_b, ukeys, ucnt, best, lab_a, lab_b, lab2_a, lab2_b, cid, used, rank, count, tmp, total;
  size_t tmp_bytes;
};

int make_layout(int64_t m, int64_t n, Layout* L) {
  const size_t big = static_cast<size_t>(m > n ? m : n);
  size_t sort_b = 0, rbk_b = 0, scan_b = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, sort_b, static_cast<uint64_t*>(nullptr),
                                          static_cast<uint64_t*>(nullptr), big, 0u, 64u);
  if (e != hipSuccess) { set_error("rocprim sort size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::reduce_by_key(nullptr, rbk_b, static_cast<uint64_t*>(nullptr),
                             rocprim::constant_iterator<uint32_t>(1u), big, static_cast<uint64_t*>(nullptr),
                             static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                             rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>());
  if (e != hipSuccess) { set_error("rocprim reduce_by_key size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::exclusive_scan(nullptr, scan_b, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                              0u, static_cast<size_t>(n), rocprim::plus<uint32_t>());
  if (e != hipSuccess) { set_error("rocprim scan size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  size_t t = sort_b > rbk_b ? sort_b : rbk_b;
  if (scan_b > t) t = scan_b;
  L->tmp_bytes = align_up(t, 256) + 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L->keys_a = take(big * 8);
  L->keys_b = take(big * 8);
  L->ukeys = take(big * 8);
  L->ucnt = take(big * 4);
  L->best = take(static_cast<size_t>(n) * 8);
  L->lab_a = take(static_cast<size_t>(n) * 4);
  L->lab_b = take(static_cast<size_t>(n) * 4);
  L->lab2_a = take(static_cast<size_t>(n) * 4);
  L->lab2_b = take(static_cast<size_t>(n) * 4);
  L->cid = take(static_cast<size_t>(n) * 4);
  L->used = take(static_cast<size_t>(n) * 4);
  L->rank = take(static_cast<size_t>(n) * 4);
  L->count = take(256);
  L->tmp = take(L->tmp_bytes);
  L->total = off;
  return SGF_OK;
}

// `iters` synchronous rounds of label propagation; labels in `la` (in), result pointer returned via *out
// (la or lb).  map / exclude_self select the level (see k_vote_keys).
int propagate(const int64_t* src, const int64_t* dst, int64_t m, int64_t n, const int32_t* map,
              int exclude_self, int iters, uint32_t* la, uint32_t* lb, char* ws, const Layout& L,
              hipStream_t st, uint32_t** out) {
  const unsigned B = bits_for(n);
  uint64_t* ka = reinterpret_cast<uint64_t*>(ws + L.keys_a);
  uint64_t* kb = reinterpret_cast<uint64_t*>(ws + L.keys_b);
  uint64_t* uk = reinterpret_cast<uint64_t*>(ws + L.ukeys);
  uint32_t* uc = reinterpret_cast<uint32_t*>(ws + L.ucnt);
  unsigned long long* best = reinterpret_cast<unsigned long long*>(ws + L.best);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + L.count);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(k_vote_keys, dim3(grid_for(m)), dim3(kThreads), 0, st, src, dst, m, n, map, la,
                       exclude_self, B, ka);
    SGF_LAUNCH_CHECK();
    size_t bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::radix_sort_keys(ws + L.tmp, bytes, ka, kb, static_cast<size_t>(m), 0u, 2 * B, st));
    bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::reduce_by_key(ws + L.tmp, bytes, kb, rocprim::constant_iterator<uint32_t>(1u),
                                         static_cast<size_t>(m), uk, uc, cnt, rocprim::plus<uint32_t>(),
                                         rocprim::equal_to<uint64_t>(), st));
    SGF_CHECK_HIP(hipMemsetAsync(best, 0, static_cast<size_t>(n) * 8, st));
    hipLaunchKernelGGL(k_vote_argmax, dim3(grid_for(m)), dim3(kThreads), 0, st, uk, uc, cnt, n, B, best);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vote_assign, dim3(grid_for(n)), dim3(kThreads), 0, st, best, n, la, lb);
    SGF_LAUNCH_CHECK();
    uint32_t* t = la; la = lb; lb = t;
  }
  *out = la;
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_reorder_workspace_bytes(int64_t nnz, int64_t n) {
  if (nnz < 0 || n < 0) return 0;
  Layout L;
  if (make_layout(nnz, n, &L) != SGF_OK) return 0;
  return L.total;
}

extern "C" int sgf_reorder(const int64_t* edge_index, int64_t nnz, int64_t n, int32_t iters1, int32_t iters2,
                           int32_t* perm, int32_t* inv, int32_t* community, void* workspace,
                           size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(nnz >= 0 && n >= 0 && iters1 >= 0 && iters2 >= 0, SGF_E_INVALID, "sgf_reorder: negative argument");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31) - 1, SGF_E_UNSUPPORTED, "sgf_reorder: n too large");
  SGF_REQUIRE(nnz < (static_cast<int64_t>(1) << 32), SGF_E_UNSUPPORTED, "sgf_reorder: nnz >= 2^32 (32-bit vote counts)");
  if (n == 0) return SGF_OK;
  SGF_REQUIRE(perm && inv && (nnz == 0 || edge_index), SGF_E_INVALID, "sgf_reorder: null pointer");
  Layout L;
  int rc = make_layout(nnz, n, &L);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= L.total, SGF_E_WORKSPACE, "sgf_reorder: workspace %zu < %zu",
              workspace_bytes, L.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const int64_t* src = edge_index;
  const int64_t* dst = edge_index + nnz;
  const unsigned B = bits_for(n);
  uint32_t* la = reinterpret_cast<uint32_t*>(ws + L.lab_a);
  uint32_t* lb = reinterpret_cast<uint32_t*>(ws + L.lab_b);
  uint32_t* l2a = reinterpret_cast<uint32_t*>(ws + L.lab2_a);
  uint32_t* l2b = reinterpret_cast<uint32_t*>(ws + L.lab2_b);
  int32_t* cid = reinterpret_cast<int32_t*>(ws + L.cid);
  uint32_t* used = reinterpret_cast<uint32_t*>(ws + L.used);
  uint32_t* rank = reinterpret_cast<uint32_t*>(ws + L.rank);
  uint64_t* ka = reinterpret_cast<uint64_t*>(ws + L.keys_a);
  uint64_t* kb = reinterpret_cast<uint64_t*>(ws + L.keys_b);

  // level 1: nodes
  hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kThreads), 0, st, la, n);
  SGF_LAUNCH_CHECK();
  uint32_t* lab1 = la;
  if (nnz > 0 && iters1 > 0) {
    rc = propagate(src, dst, nnz, n, nullptr, 0, iters1, la, lb, ws, L, st, &lab1);
    if (rc != SGF_OK) return rc;
  }
  // compact the labels in use to community ids 0 .. nc-1 (ascending label)
  SGF_CHECK_HIP(hipMemsetAsync(used, 0, static_cast<size_t>(n) * 4, st));
  hipLaunchKernelGGL(k_mark, dim3(grid_for(n)), dim3(kThreads), 0, st, lab1, n, used);
  SGF_LAUNCH_CHECK();
  {
    size_t bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, used, rank, 0u, static_cast<size_t>(n),
                                          rocprim::plus<uint32_t>(), st));
  }
  hipLaunchKernelGGL(k_compact, dim3(grid_for(n)), dim3(kThreads), 0, st, lab1, rank, n, cid);
  SGF_LAUNCH_CHECK();
  if (community) SGF_CHECK_HIP(hipMemcpyAsync(community, cid, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, st));

  // level 2: communities, inter-community edges only
  hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kThreads), 0, st, l2a, n);
  SGF_LAUNCH_CHECK();
  uint32_t* lab2 = l2a;
  if (nnz > 0 && iters2 > 0) {
    rc = propagate(src, dst, nnz, n, cid, 1, iters2, l2a, l2b, ws, L, st, &lab2);
    if (rc != SGF_OK) return rc;
  }
  // communities by (level-2 label, id) -> rank; nodes by (community rank, id) -> perm
  hipLaunchKernelGGL(k_comm_keys, dim3(grid_for(n)), dim3(kThreads), 0, st, lab2, n, B, ka);
  SGF_LAUNCH_CHECK();
  size_t bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::radix_sort_keys(ws + L.tmp, bytes, ka, kb, static_cast<size_t>(n), 0u, 2 * B, st));
  hipLaunchKernelGGL(k_comm_rank, dim3(grid_for(n)), dim3(kThreads), 0, st, kb, n, B, rank);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_node_keys, dim3(grid_for(n)), dim3(kThreads), 0, st, cid, rank, n, B, ka);
  SGF_LAUNCH_CHECK();
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::radix_sort_keys(ws + L.tmp, bytes, ka, kb, static_cast<size_t>(n), 0u, 2 * B, st));
  hipLaunchKernelGGL(k_perm, dim3(grid_for(n)), dim3(kThreads), 0, st, kb, n, B, perm, inv);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
