// sampler.hip — N2 (second half): layer-wise neighbour sampling on the device.
//
// Replaces, for the 100M recipe, what `NeighborLoader(data, num_neighbors=[15, 10, 5], ...)` does on 12 host
// workers (100M/nb-sample.py:125-151; third-party pyg-lib / torch_sparse `neighbor_sample`, replace=False,
// directed=True): starting from the seed nodes, hop h gives every node that ENTERED the batch in hop h - 1 (the seeds
// for h = 0) min(in-degree, fanout_h) of its in-neighbours, drawn without replacement; a neighbour that is not in the
// batch yet gets the next local id, in order of first appearance (seeds first: the trainer slices [:batch_size],
// 100M/nb-sample.py:29-30,41-42); the sampled edges point neighbour -> node, local ids, hop after hop.
// The reference's random stream (std::mt19937 per worker) cannot be matched — nor does the reference match itself
// across worker counts — so parity is structural (tests/test_gpu_sampler.py) and the draw here is a counter-based
// hash keyed on (seed, batch, hop, node): reproducible, order-independent, restated bit for bit in
// oracle/graph_oracle.py::neighbor_sample.
//
// Graph: the CSR over TARGET nodes sgf_csr_build makes (rowptr int64, colind int32 = the in-neighbours of a node).
// State: local_of int32[n_nodes], INT32_MIN = not in the batch (444 MB at papers100M scale; reset per batch by the ids
// the batch touched, not by a full fill).  The order  not-in-batch < claim of a later position < claim of an earlier
// position < local id  is what lets ONE atomicMax per sampled entry elect the first appearance of every new node.
// One hop = counts -> scan -> draw (one thread per frontier node: Floyd's subset sampling up to fan-out 32, selection
// sampling above) -> claim
// (atomicMax of -(position + 2): the FIRST position of every new id wins, whatever order threads run in) -> flag ->
// scan -> assign -> edges.  Deterministic.
#include "common.h"

#include <rocprim/rocprim.hpp>

#include <climits>

namespace sgf {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxFanout = 32;

inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 16;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

__host__ __device__ inline uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// edges this frontier node contributes: min(in-degree, fanout); fanout < 0 = all of them
__global__ void k_counts(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ frontier, int64_t m,
                         int32_t fanout, int32_t* __restrict__ cnt) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i <= m; i += stride) {
    if (i == m) { cnt[i] = 0; continue; }
    const int64_t f = frontier[i];
    const int64_t deg = rowptr[f + 1] - rowptr[f];
    cnt[i] = static_cast<int32_t>(fanout >= 0 && deg > fanout ? fanout : deg);
  }
}

// A uniform k-subset of {0 .. deg-1}, k < deg, emitted as (slot, position) pairs.
//   k <= 32: Floyd's algorithm, k draws — the t-th draw is r = hash mod (t + 1), replaced by t itself when r was drawn
//            before; emitted in draw order.
//   k  > 32: selection sampling (Knuth's algorithm S), one pass over the row, no memory: position j is taken with
//            probability (k - taken) / (deg - j); emitted in position order.
template <class Emit>
__device__ __forceinline__ void draw_subset(int64_t deg, int32_t k, uint64_t nk, Emit emit) {
  if (k <= kMaxFanout) {
    int64_t chosen[kMaxFanout];
    int j = 0;
    for (int64_t t = deg - k; t < deg; ++t, ++j) {
      int64_t r = static_cast<int64_t>(mix64(nk + static_cast<uint64_t>(j)) % static_cast<uint64_t>(t + 1));
      for (int q = 0; q < j; ++q)
        if (chosen[q] == r) { r = t; break; }
      chosen[j] = r;
      emit(j, r);
    }
  } else {
    int32_t taken = 0;
    for (int64_t j = 0; j < deg && taken < k; ++j) {
      const uint64_t u = mix64(nk + static_cast<uint64_t>(j)) % static_cast<uint64_t>(deg - j);
      if (u < static_cast<uint64_t>(k - taken)) emit(taken++, j);
    }
  }
}

__global__ void k_draw(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                       const int32_t* __restrict__ frontier, int64_t m, int32_t fanout, int32_t local0,
                       const int32_t* __restrict__ off, uint64_t key, int32_t* __restrict__ src_global,
                       int32_t* __restrict__ dst_local) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < m; i += stride) {
    const int64_t f = frontier[i];
    const int64_t base = rowptr[f];
    const int64_t deg = rowptr[f + 1] - base;
    const int32_t o = off[i];
    const int32_t dl = local0 + static_cast<int32_t>(i);
    if (fanout < 0 || deg <= fanout) {
      for (int64_t j = 0; j < deg; ++j) {
        src_global[o + j] = colind[base + j];
        dst_local[o + j] = dl;
      }
      continue;
    }
    const uint64_t nk = mix64(key ^ (static_cast<uint64_t>(f) * 0xd6e8feb86659fd93ull));
    draw_subset(deg, fanout, nk, [&](int32_t slot, int64_t r) {
      src_global[o + slot] = colind[base + r];
      dst_local[o + slot] = dl;
    });
  }
}

__global__ void k_claim(const int32_t* __restrict__ src_global, int64_t total, int32_t* __restrict__ local_of) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < total; p += stride) atomicMax(&local_of[src_global[p]], -static_cast<int32_t>(p) - 2);
}

__global__ void k_flag(const int32_t* __restrict__ src_global, int64_t total, const int32_t* __restrict__ local_of,
                       int32_t* __restrict__ flag) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p <= total; p += stride)
    flag[p] = p < total && local_of[src_global[p]] == -static_cast<int32_t>(p) - 2 ? 1 : 0;
}

__global__ void k_assign(const int32_t* __restrict__ src_global, int64_t total, const int32_t* __restrict__ flag,
                         const int32_t* __restrict__ fscan, int32_t n_known, int32_t* __restrict__ local_of,
                         int32_t* __restrict__ new_nodes) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < total; p += stride)
    if (flag[p]) {
      local_of[src_global[p]] = n_known + fscan[p];
      new_nodes[fscan[p]] = src_global[p];
    }
}

__global__ void k_edges(const int32_t* __restrict__ src_global, int64_t total, const int32_t* __restrict__ local_of,
                        const int32_t* __restrict__ fscan, int32_t* __restrict__ src_local, int64_t* __restrict__ counts) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  if (p == 0) {
    counts[0] = total;
    counts[1] = fscan[total];
  }
  for (; p < total; p += stride) src_local[p] = local_of[src_global[p]];
}

__global__ void k_mark(int32_t* __restrict__ local_of, const int32_t* __restrict__ ids, int64_t count, int32_t base) {
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; j < count; j += stride) local_of[ids[j]] = base < 0 ? INT32_MIN : base + static_cast<int32_t>(j);
}

// ---- a whole batch without a host read: the frontier size, the local-id base and the output offsets live in `st` ----
struct BatchState {
  int32_t m, local0, n_known, e_known;
};

__global__ void k_batch_init(const int32_t* __restrict__ seeds, int64_t bs, int32_t* __restrict__ local_of,
                             int32_t* __restrict__ nodes, BatchState* __restrict__ st, int64_t* __restrict__ counts) {
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  if (j == 0) {
    *st = BatchState{static_cast<int32_t>(bs), 0, static_cast<int32_t>(bs), 0};
    counts[0] = bs;
  }
  for (; j < bs; j += stride) {
    nodes[j] = seeds[j];
    local_of[seeds[j]] = static_cast<int32_t>(j);
  }
}

__global__ void k_counts_d(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ nodes,
                           const BatchState* __restrict__ st, int64_t cap_m, int32_t fanout, int32_t* __restrict__ cnt) {
  const int64_t m = st->m;
  const int32_t* frontier = nodes + st->local0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i <= cap_m; i += stride) {
    if (i >= m) { cnt[i] = 0; continue; }
    const int64_t f = frontier[i];
    const int64_t deg = rowptr[f + 1] - rowptr[f];
    cnt[i] = static_cast<int32_t>(deg > fanout ? fanout : deg);
  }
}

__global__ void k_draw_d(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                         const int32_t* __restrict__ nodes, const BatchState* __restrict__ st, int32_t fanout,
                         const int32_t* __restrict__ off, uint64_t key, int32_t* __restrict__ src_global,
                         int32_t* __restrict__ edge_dst) {
  const int64_t m = st->m;
  const int32_t local0 = st->local0;
  const int32_t* frontier = nodes + local0;
  int32_t* dst_local = edge_dst + st->e_known;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; i < m; i += stride) {
    const int64_t f = frontier[i];
    const int64_t base = rowptr[f];
    const int64_t deg = rowptr[f + 1] - base;
    const int32_t o = off[i];
    const int32_t dl = local0 + static_cast<int32_t>(i);
    if (deg <= fanout) {
      for (int64_t j = 0; j < deg; ++j) {
        src_global[o + j] = colind[base + j];
        dst_local[o + j] = dl;
      }
      continue;
    }
    const uint64_t nk = mix64(key ^ (static_cast<uint64_t>(f) * 0xd6e8feb86659fd93ull));
    draw_subset(deg, fanout, nk, [&](int32_t slot, int64_t r) {
      src_global[o + slot] = colind[base + r];
      dst_local[o + slot] = dl;
    });
  }
}

// total = off[cap_m] (the counts beyond the frontier are zero)
__global__ void k_claim_d(const int32_t* __restrict__ src_global, const int32_t* __restrict__ total_p,
                          int32_t* __restrict__ local_of) {
  const int64_t total = *total_p;
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < total; p += stride) atomicMax(&local_of[src_global[p]], -static_cast<int32_t>(p) - 2);
}

__global__ void k_flag_d(const int32_t* __restrict__ src_global, const int32_t* __restrict__ total_p, int64_t cap_e,
                         const int32_t* __restrict__ local_of, int32_t* __restrict__ flag) {
  const int64_t total = *total_p;
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p <= cap_e; p += stride)
    flag[p] = p < total && local_of[src_global[p]] == -static_cast<int32_t>(p) - 2 ? 1 : 0;
}

__global__ void k_assign_d(const int32_t* __restrict__ src_global, const int32_t* __restrict__ total_p,
                           const int32_t* __restrict__ flag, const int32_t* __restrict__ fscan,
                           const BatchState* __restrict__ st, int32_t* __restrict__ local_of, int32_t* __restrict__ nodes) {
  const int64_t total = *total_p;
  const int32_t n_known = st->n_known;
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < total; p += stride)
    if (flag[p]) {
      local_of[src_global[p]] = n_known + fscan[p];
      nodes[n_known + fscan[p]] = src_global[p];
    }
}

__global__ void k_edges_d(const int32_t* __restrict__ src_global, const int32_t* __restrict__ total_p,
                          const int32_t* __restrict__ local_of, const BatchState* __restrict__ st,
                          int32_t* __restrict__ edge_src) {
  const int64_t total = *total_p;
  int32_t* src_local = edge_src + st->e_known;
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < total; p += stride) src_local[p] = local_of[src_global[p]];
}

// after the hop's last reader of `st`: the nodes that entered are the next frontier
__global__ void k_advance(BatchState* __restrict__ st, const int32_t* __restrict__ total_p, const int32_t* __restrict__ new_p,
                          int64_t* __restrict__ counts, int32_t hop) {
  const int32_t total = *total_p, nn = *new_p;
  st->local0 = st->n_known;
  st->m = nn;
  st->n_known += nn;
  st->e_known += total;
  counts[2 + 2 * hop] = total;
  counts[3 + 2 * hop] = nn;
  counts[0] = st->n_known;
  counts[1] = st->e_known;
}

__global__ void k_unmark_d(int32_t* __restrict__ local_of, const int32_t* __restrict__ nodes,
                           const BatchState* __restrict__ st) {
  const int64_t count = st->n_known;
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; j < count; j += stride) local_of[nodes[j]] = INT32_MIN;
}

// the capacities of a batch with non-negative fan-outs: frontier of hop h <= edges of hop h - 1 <= its frontier x fan-out
struct BatchCaps {
  int64_t m_max, e_max, nodes, edges;
};

inline bool batch_caps(int64_t bs, const int32_t* fanouts, int32_t hops, BatchCaps* c) {
  int64_t m = bs;
  c->m_max = bs; c->e_max = 0; c->nodes = bs; c->edges = 0;
  for (int h = 0; h < hops; ++h) {
    if (fanouts[h] < 0) return false;
    const int64_t e = m * fanouts[h];
    if (e >= (static_cast<int64_t>(1) << 31) - 2) return false;
    c->m_max = m > c->m_max ? m : c->m_max;
    c->e_max = e > c->e_max ? e : c->e_max;
    c->nodes += e;
    c->edges += e;
    m = e;
  }
  return c->nodes < (static_cast<int64_t>(1) << 31) - 2;
}

struct Layout {
  size_t cnt, off, flag, fscan, total_slot, tmp, total, tmp_bytes;
};

int make_layout(int64_t m, int64_t cap, Layout* L) {
  size_t a = 0, b = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, a, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                                         static_cast<size_t>(m + 1), rocprim::plus<int32_t>());
  if (e != hipSuccess) { set_error("rocprim scan size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::exclusive_scan(nullptr, b, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), 0,
                              static_cast<size_t>(cap + 1), rocprim::plus<int32_t>());
  if (e != hipSuccess) { set_error("rocprim scan size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  L->tmp_bytes = align_up(a > b ? a : b, 256) + 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L->cnt = take(static_cast<size_t>(m + 1) * 4);
  L->off = take(static_cast<size_t>(m + 1) * 4);
  L->flag = take(static_cast<size_t>(cap + 1) * 4);
  L->fscan = take(static_cast<size_t>(cap + 1) * 4);
  L->total_slot = take(256);
  L->tmp = take(L->tmp_bytes);
  L->total = off;
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_neighbor_sample_workspace_bytes(int64_t m, int64_t edge_cap) {
  if (m < 0 || edge_cap < 0) return 0;
  Layout L;
  if (make_layout(m, edge_cap, &L) != SGF_OK) return 0;
  return L.total;
}

extern "C" int sgf_neighbor_sample_mark(int32_t* local_of, const int32_t* ids, int64_t count, int32_t base, void* stream) {
  SGF_REQUIRE(count >= 0, SGF_E_INVALID, "sgf_neighbor_sample_mark: negative count");
  if (count == 0) return SGF_OK;
  SGF_REQUIRE(local_of && ids, SGF_E_INVALID, "sgf_neighbor_sample_mark: null pointer");
  hipLaunchKernelGGL(k_mark, dim3(grid_for(count)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), local_of, ids,
                     count, base);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_neighbor_sample_hop(const int64_t* rowptr, const int32_t* colind, const int32_t* frontier, int64_t m,
                                       int32_t frontier_local0, int32_t fanout, uint64_t seed, uint64_t batch, int32_t hop,
                                       int32_t* local_of, int32_t n_known, int64_t edge_cap, int32_t* edge_src_local,
                                       int32_t* edge_dst_local, int32_t* src_global, int32_t* new_nodes, int64_t* counts,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_neighbor_sample_hop";
  SGF_REQUIRE(m >= 0 && edge_cap >= 0, SGF_E_INVALID, "%s: bad size", fn);
  SGF_REQUIRE(counts, SGF_E_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (m == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int64_t), st));
    return SGF_OK;
  }
  SGF_REQUIRE(rowptr && colind && frontier && local_of && edge_src_local && edge_dst_local && src_global && new_nodes,
              SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(edge_cap < (static_cast<int64_t>(1) << 31) - 2, SGF_E_UNSUPPORTED, "%s: edge_cap >= 2^31", fn);
  SGF_REQUIRE(fanout < 0 || m * static_cast<int64_t>(fanout) <= edge_cap, SGF_E_WORKSPACE,
              "%s: edge_cap %lld < m * fanout", fn, static_cast<long long>(edge_cap));
  Layout L;
  int rc = make_layout(m, edge_cap, &L);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= L.total, SGF_E_WORKSPACE, "%s: workspace %zu < %zu", fn, workspace_bytes,
              L.total);
  char* ws = static_cast<char*>(workspace);
  int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
  int32_t* off = reinterpret_cast<int32_t*>(ws + L.off);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + L.flag);
  int32_t* fscan = reinterpret_cast<int32_t*>(ws + L.fscan);
  hipLaunchKernelGGL(k_counts, dim3(grid_for(m + 1)), dim3(kThreads), 0, st, rowptr, frontier, m, fanout, cnt);
  SGF_LAUNCH_CHECK();
  size_t bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, cnt, off, 0, static_cast<size_t>(m + 1), rocprim::plus<int32_t>(), st));
  // one host read per hop: the number of sampled edges sizes the launches below (and must fit edge_cap when
  // fanout < 0 takes every in-neighbour)
  int32_t total_h = 0;
  SGF_CHECK_HIP(hipMemcpyAsync(&total_h, off + m, 4, hipMemcpyDeviceToHost, st));
  SGF_CHECK_HIP(hipStreamSynchronize(st));
  SGF_REQUIRE(total_h <= edge_cap, SGF_E_WORKSPACE, "%s: %d edges, room for %lld", fn, total_h,
              static_cast<long long>(edge_cap));
  const int64_t total = total_h;
  const uint64_t key = mix64(seed ^ mix64(batch * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(hop)));
  hipLaunchKernelGGL(k_draw, dim3(grid_for(m)), dim3(kThreads), 0, st, rowptr, colind, frontier, m, fanout,
                     frontier_local0, off, key, src_global, edge_dst_local);
  SGF_LAUNCH_CHECK();
  if (total == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int64_t), st));
    return SGF_OK;
  }
  hipLaunchKernelGGL(k_claim, dim3(grid_for(total)), dim3(kThreads), 0, st, src_global, total, local_of);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_flag, dim3(grid_for(total + 1)), dim3(kThreads), 0, st, src_global, total, local_of, flag);
  SGF_LAUNCH_CHECK();
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, flag, fscan, 0, static_cast<size_t>(total + 1),
                                        rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(k_assign, dim3(grid_for(total)), dim3(kThreads), 0, st, src_global, total, flag, fscan, n_known,
                     local_of, new_nodes);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_edges, dim3(grid_for(total)), dim3(kThreads), 0, st, src_global, total, local_of, fscan,
                     edge_src_local, counts);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

// ---- a whole batch, every hop, no host read (fan-outs >= 0) ----------------------------------------------------------
extern "C" size_t sgf_neighbor_sample_batch_workspace_bytes(int64_t batch_size, const int32_t* fanouts, int32_t hops,
                                                            int64_t* node_cap, int64_t* edge_cap) {
  BatchCaps c;
  if (batch_size < 0 || hops < 0 || (hops > 0 && !fanouts) || !batch_caps(batch_size, fanouts, hops, &c)) return 0;
  if (node_cap) *node_cap = c.nodes;
  if (edge_cap) *edge_cap = c.edges;
  Layout L;
  if (make_layout(c.m_max, c.e_max, &L) != SGF_OK) return 0;
  return L.total + align_up(static_cast<size_t>(c.e_max + 1) * 4, 256);
}

extern "C" int sgf_neighbor_sample_batch(const int64_t* rowptr, const int32_t* colind, const int32_t* seeds,
                                         int64_t batch_size, const int32_t* fanouts, int32_t hops, uint64_t seed,
                                         uint64_t batch, int32_t* local_of, int32_t* nodes, int64_t node_cap,
                                         int32_t* edge_src_local, int32_t* edge_dst_local, int64_t edge_cap, int64_t* counts,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_neighbor_sample_batch";
  SGF_REQUIRE(batch_size >= 0 && hops >= 0 && hops <= 16, SGF_E_INVALID, "%s: bad sizes (hops <= 16)", fn);
  SGF_REQUIRE(counts && (hops == 0 || fanouts), SGF_E_INVALID, "%s: null pointer", fn);
  BatchCaps c;
  SGF_REQUIRE(batch_caps(batch_size, fanouts, hops, &c), SGF_E_UNSUPPORTED,
              "%s: negative fan-out (all neighbours: use sgf_neighbor_sample_hop) or more than 2^31 sampled entries", fn);
  SGF_REQUIRE(node_cap >= c.nodes && edge_cap >= c.edges, SGF_E_WORKSPACE, "%s: node_cap %lld < %lld or edge_cap %lld < %lld",
              fn, static_cast<long long>(node_cap), static_cast<long long>(c.nodes), static_cast<long long>(edge_cap),
              static_cast<long long>(c.edges));
  hipStream_t st = static_cast<hipStream_t>(stream);
  SGF_CHECK_HIP(hipMemsetAsync(counts, 0, static_cast<size_t>(2 + 2 * hops) * sizeof(int64_t), st));
  if (batch_size == 0) return SGF_OK;
  SGF_REQUIRE(rowptr && colind && seeds && local_of && nodes && (c.edges == 0 || (edge_src_local && edge_dst_local)),
              SGF_E_INVALID, "%s: null pointer", fn);
  Layout L;
  int rc = make_layout(c.m_max, c.e_max, &L);
  if (rc != SGF_OK) return rc;
  const size_t src_off = L.total;
  const size_t need = L.total + align_up(static_cast<size_t>(c.e_max + 1) * 4, 256);
  SGF_REQUIRE(workspace && workspace_bytes >= need, SGF_E_WORKSPACE, "%s: workspace %zu < %zu", fn, workspace_bytes, need);
  char* ws = static_cast<char*>(workspace);
  int32_t* cnt = reinterpret_cast<int32_t*>(ws + L.cnt);
  int32_t* off = reinterpret_cast<int32_t*>(ws + L.off);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + L.flag);
  int32_t* fscan = reinterpret_cast<int32_t*>(ws + L.fscan);
  BatchState* state = reinterpret_cast<BatchState*>(ws + L.total_slot);
  int32_t* src_global = reinterpret_cast<int32_t*>(ws + src_off);
  hipLaunchKernelGGL(k_batch_init, dim3(grid_for(batch_size)), dim3(kThreads), 0, st, seeds, batch_size, local_of, nodes, state, counts);
  SGF_LAUNCH_CHECK();
  int64_t cap_m = batch_size;
  for (int h = 0; h < hops; ++h) {
    const int32_t k = fanouts[h];
    const int64_t cap_e = cap_m * k;
    hipLaunchKernelGGL(k_counts_d, dim3(grid_for(cap_m + 1)), dim3(kThreads), 0, st, rowptr, nodes, state, cap_m, k, cnt);
    SGF_LAUNCH_CHECK();
    size_t bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, cnt, off, 0, static_cast<size_t>(cap_m + 1),
                                          rocprim::plus<int32_t>(), st));
    const int32_t* total_p = off + cap_m;
    const uint64_t key = mix64(seed ^ mix64(batch * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(h)));
    hipLaunchKernelGGL(k_draw_d, dim3(grid_for(cap_m)), dim3(kThreads), 0, st, rowptr, colind, nodes, state, k, off, key,
                       src_global, edge_dst_local);
    SGF_LAUNCH_CHECK();
    if (cap_e > 0) {
      hipLaunchKernelGGL(k_claim_d, dim3(grid_for(cap_e)), dim3(kThreads), 0, st, src_global, total_p, local_of);
      SGF_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_flag_d, dim3(grid_for(cap_e + 1)), dim3(kThreads), 0, st, src_global, total_p, cap_e, local_of, flag);
      SGF_LAUNCH_CHECK();
      bytes = L.tmp_bytes;
      SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, flag, fscan, 0, static_cast<size_t>(cap_e + 1),
                                            rocprim::plus<int32_t>(), st));
      hipLaunchKernelGGL(k_assign_d, dim3(grid_for(cap_e)), dim3(kThreads), 0, st, src_global, total_p, flag, fscan, state,
                         local_of, nodes);
      SGF_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_edges_d, dim3(grid_for(cap_e)), dim3(kThreads), 0, st, src_global, total_p, local_of, state,
                         edge_src_local);
      SGF_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, state, total_p, fscan + cap_e, counts, h);
      SGF_LAUNCH_CHECK();
    }
    cap_m = cap_e;
    if (cap_m == 0) break;
  }
  hipLaunchKernelGGL(k_unmark_d, dim3(grid_for(c.nodes)), dim3(kThreads), 0, st, local_of, nodes, state);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
