// spmm_plan.hip — the row-block plan of the LDS-staged SpMM (k_spmm_blk in spmm.hip).
//
// T2, large/ours.py:34: Y[i,:] = sum_e val[e] X[colind[e],:].  The wave-per-row kernel fetches every
// neighbour row from L2 / HBM once per stored entry.  When consecutive rows share neighbours (any graph
// with community structure, after reorder.hip has put communities next to each other), a block of R
// target rows can fetch each SHARED neighbour row once into LDS and serve all its uses from there.
// Which rows are shared is a property of the graph, so it is computed once per CSR, here:
//
//   for every block b of R consecutive target rows
//     unique (b, source) pairs and their multiplicity        radix sort of (block, source) keys + reduce-by-key
//     the `cap` most-referenced sources with multiplicity >= 2  sort of (block, -count, source) keys
//        -> sh_cols[sh_ptr[b] .. sh_ptr[b+1])  : the rows block b stages in LDS, slot = position
//   every stored entry gets a code: 0x80000000 | slot  (served from LDS)   or   the source id (gathered)
//   inside a row the LDS entries are moved in front of the gathered ones (stable), nlds[row] = their count,
//   values permuted alongside.
// Rows longer than `long_len` keep plain source ids (they go to the long-row path of spmm.hip).
// Summation order inside a row therefore differs from the plain kernel (LDS entries first): results
// agree to fp32 rounding, not bit for bit; the order is fixed, so the kernel stays deterministic.
#include "common.h"

#include <rocprim/rocprim.hpp>

#include <vector>

namespace sgf {
namespace {

constexpr int kThreads = 256;
constexpr unsigned kCountBits = 10;                 // multiplicities are clipped to 1023 for the ranking
constexpr uint32_t kCountMax = (1u << kCountBits) - 1;

inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  const int64_t cap = static_cast<int64_t>(kNumCU) * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

inline unsigned bits_for(int64_t n) {   // smallest B with 2^B - 1 >= n
  unsigned b = 1;
  while (((static_cast<int64_t>(1) << b) - 1) < n && b < 40) ++b;
  return b;
}

// row of entry e: largest r with rowptr[r] <= e
__device__ __forceinline__ int64_t row_of(const int64_t* __restrict__ rowptr, int64_t n, int64_t e) {
  int64_t lo = 0, hi = n;   // invariant: rowptr[lo] <= e < rowptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// key1 = (block << B) | source  (all-ones sentinel for entries of long rows), value = entry index
__global__ void k_keys1(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t n,
                        int64_t nnz, int32_t R, const int32_t* __restrict__ row_block, int64_t long_len, unsigned B,
                        unsigned Bb, uint64_t* __restrict__ keys, uint32_t* __restrict__ eidx,
                        int32_t* __restrict__ rowid) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t sentinel = (static_cast<uint64_t>(1) << (B + Bb)) - 1;
  for (; e < nnz; e += stride) {
    const int64_t r = row_of(rowptr, n, e);
    rowid[e] = static_cast<int32_t>(r);
    const bool is_long = rowptr[r + 1] - rowptr[r] > long_len;
    const uint64_t blk = row_block ? static_cast<uint64_t>(row_block[r]) : static_cast<uint64_t>(r / R);
    keys[e] = is_long ? sentinel : ((blk << B) | static_cast<uint32_t>(colind[e]));
    eidx[e] = static_cast<uint32_t>(e);
  }
}

// key2 = (block << (B + C)) | ((CMAX - min(count, CMAX)) << B) | source ; value = unique index j
__global__ void k_keys2(const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ ucnt,
                        const uint32_t* __restrict__ n_unique, int64_t m, unsigned B, unsigned Bb,
                        uint64_t* __restrict__ keys, uint32_t* __restrict__ jidx) {
  const int64_t u = *n_unique;
  int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t smask = (static_cast<uint64_t>(1) << B) - 1;
  const uint64_t sentinel1 = (static_cast<uint64_t>(1) << (B + Bb)) - 1;
  for (; j < m; j += stride) {
    uint64_t k = ~static_cast<uint64_t>(0);
    if (j < u && ukeys[j] != sentinel1) {
      const uint64_t blk = ukeys[j] >> B, s = ukeys[j] & smask;
      const uint32_t c = ucnt[j] < kCountMax ? ucnt[j] : kCountMax;
      k = (blk << (B + kCountBits)) | (static_cast<uint64_t>(kCountMax - c) << B) | s;
    }
    keys[j] = k;
    jidx[j] = static_cast<uint32_t>(j);
  }
}

// first sorted position of every block
__global__ void k_block_heads(const uint64_t* __restrict__ keys2, int64_t m, unsigned B,
                              uint32_t* __restrict__ head) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < m; p += stride) {
    const uint64_t k = keys2[p];
    if (k == ~static_cast<uint64_t>(0)) continue;
    const uint64_t blk = k >> (B + kCountBits);
    if (p == 0 || (keys2[p - 1] >> (B + kCountBits)) != blk) head[blk] = static_cast<uint32_t>(p);
  }
}

// rank inside the block; the first `cap` uniques with count >= 2 get LDS slots
__global__ void k_slots(const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ jidx, int64_t m,
                        unsigned B, const uint32_t* __restrict__ head, int32_t cap, uint32_t min_count,
                        int32_t* __restrict__ slot_of_unique, int32_t* __restrict__ nsh) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < m; p += stride) {
    const uint64_t k = keys2[p];
    if (k == ~static_cast<uint64_t>(0)) continue;
    const uint64_t blk = k >> (B + kCountBits);
    const uint32_t cfield = static_cast<uint32_t>((k >> B) & kCountMax);   // CMAX - count
    const int64_t rank = p - head[blk];
    const bool ok = rank < cap && cfield <= kCountMax - min_count;
    slot_of_unique[jidx[p]] = ok ? static_cast<int32_t>(rank) : -1;
    if (ok) atomicAdd(&nsh[blk], 1);
  }
}

// Tile plans only: the selected sources of a block take their slots in ASCENDING ID order instead of by count.  The two
// or three blocks a community is cut into stage nearly the same sources, start within a microsecond of each other on the
// same XCD and then request them in the same order: the second block's staging hits L2.  (Which sources are selected — the
// `cap` most-referenced — does not change.)  Uniques are numbered in (block, source) order, so the new slot of unique u is
// the number of selected uniques of its block before it.
__global__ void k_reslot_mark(const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ jidx, int64_t m, unsigned B,
                              const int32_t* __restrict__ slot_of_unique, uint32_t* __restrict__ blk_of_unique,
                              uint32_t* __restrict__ sel) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; p < m; p += stride) {
    const uint64_t k = keys2[p];
    if (k == ~static_cast<uint64_t>(0)) continue;
    const uint32_t u = jidx[p];
    blk_of_unique[u] = static_cast<uint32_t>(k >> (B + kCountBits));
    sel[u] = slot_of_unique[u] >= 0 ? 1u : 0u;
  }
}

__global__ void k_reslot_first(const uint32_t* __restrict__ blk_of_unique, const uint32_t* __restrict__ pos,
                               const uint32_t* __restrict__ n_unique, uint32_t* __restrict__ first) {
  const int64_t nu = *n_unique;                          // (the unique of the long rows carries block 0xffffffff: skipped)
  int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; u < nu; u += stride) {
    const uint32_t b = blk_of_unique[u];
    if (b != 0xffffffffu && (u == 0 || blk_of_unique[u - 1] != b)) first[b] = pos[u];
  }
}

__global__ void k_reslot_apply(const uint32_t* __restrict__ blk_of_unique, const uint32_t* __restrict__ pos,
                               const uint32_t* __restrict__ sel, const uint32_t* __restrict__ first,
                               const uint32_t* __restrict__ n_unique, int32_t* __restrict__ slot_of_unique) {
  const int64_t nu = *n_unique;
  int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; u < nu; u += stride)
    if (sel[u]) slot_of_unique[u] = static_cast<int32_t>(pos[u] - first[blk_of_unique[u]]);
}

__global__ void k_sh_cols(const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ jidx, int64_t m,
                          unsigned B, const int32_t* __restrict__ slot_of_unique,
                          const int32_t* __restrict__ sh_ptr, int32_t* __restrict__ sh_cols) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t smask = (static_cast<uint64_t>(1) << B) - 1;
  for (; p < m; p += stride) {
    const uint64_t k = keys2[p];
    if (k == ~static_cast<uint64_t>(0)) continue;
    const int32_t slot = slot_of_unique[jidx[p]];
    if (slot >= 0) sh_cols[sh_ptr[k >> (B + kCountBits)] + slot] = static_cast<int32_t>(k & smask);
  }
}

// run index of sorted position i: largest j < u with ustart[j] <= i
__global__ void k_codes(const uint64_t* __restrict__ keys1, const uint32_t* __restrict__ eidx, int64_t nnz,
                        const uint32_t* __restrict__ ustart, const uint32_t* __restrict__ n_unique,
                        const int32_t* __restrict__ slot_of_unique, const int32_t* __restrict__ colind,
                        unsigned B, unsigned Bb, int32_t* __restrict__ code_tmp, uint32_t* __restrict__ flag) {
  const int64_t u = *n_unique;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t sentinel = (static_cast<uint64_t>(1) << (B + Bb)) - 1;
  for (; i < nnz; i += stride) {
    const uint32_t e = eidx[i];
    int32_t code = colind[e];
    uint32_t f = 0;
    if (keys1[i] != sentinel) {
      int64_t lo = 0, hi = u;   // ustart[lo] <= i < ustart[hi] (ustart[u] = nnz conceptually)
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (static_cast<int64_t>(ustart[mid]) <= i) lo = mid; else hi = mid;
      }
      const int32_t slot = slot_of_unique[lo];
      if (slot >= 0) {
        code = static_cast<int32_t>(0x80000000u | static_cast<uint32_t>(slot));
        f = 1;
      }
    }
    code_tmp[e] = code;
    flag[e] = f;
  }
}

// stable partition inside each row: LDS entries first
__global__ void k_partition(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ rowid,
                            const int32_t* __restrict__ code_tmp, const uint32_t* __restrict__ flag,
                            const uint32_t* __restrict__ fscan, const float* __restrict__ val, int64_t nnz,
                            int32_t* __restrict__ ecode, float* __restrict__ eval, int32_t* __restrict__ nlds) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; e < nnz; e += stride) {
    const int32_t r = rowid[e];
    const int64_t rs = rowptr[r], re = rowptr[r + 1];
    const uint32_t before = fscan[e] - fscan[rs];
    const uint32_t nl = fscan[re] - fscan[rs];          // fscan has nnz + 1 entries
    const int64_t pos = flag[e] ? rs + before : rs + nl + (e - rs - before);
    ecode[pos] = code_tmp[e];
    eval[pos] = val[e];
    if (e == rs) nlds[r] = static_cast<int32_t>(nl);
  }
}

__global__ void k_stats(const uint32_t* __restrict__ fscan, int64_t nnz, const int32_t* __restrict__ sh_ptr,
                        int64_t nb, const uint32_t* __restrict__ n_unique, int64_t* __restrict__ stats) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[0] = fscan[nnz];      // entries served from LDS
    stats[1] = sh_ptr[nb];      // rows staged into LDS, summed over blocks
    stats[2] = *n_unique;       // distinct (block, source) pairs (incl. one run for long-row entries, if any)
    stats[3] = nnz;
  }
}

struct Layout {
  size_t keys_a, keys_b, idx_a, idx_b, ukeys, ucnt, ustart, slot, rowid, j2s, flag, fscan, head, nsh,
      nshp, rowblk, i64, count, tmp, total, tmp_bytes;
};

int make_layout(int64_t nnz, int64_t nb, int64_t n, Layout* L) {
  const size_t m = static_cast<size_t>(nnz);
  size_t sort_b = 0, rbk_b = 0, scan_b = 0, scan64_b = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_b, static_cast<uint64_t*>(nullptr),
                                           static_cast<uint64_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                           static_cast<uint32_t*>(nullptr), m, 0u, 64u);
  if (e != hipSuccess) { set_error("rocprim sort size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::reduce_by_key(nullptr, rbk_b, static_cast<uint64_t*>(nullptr),
                             rocprim::constant_iterator<uint32_t>(1u), m, static_cast<uint64_t*>(nullptr),
                             static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                             rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>());
  if (e != hipSuccess) { set_error("rocprim reduce_by_key size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::exclusive_scan(nullptr, scan_b, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                              0u, m + 1, rocprim::plus<uint32_t>());
  if (e != hipSuccess) { set_error("rocprim scan size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  e = rocprim::exclusive_scan(nullptr, scan64_b, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr),
                              static_cast<int64_t>(0), static_cast<size_t>(n > nb ? n : nb) + 1, rocprim::plus<int64_t>());
  if (e != hipSuccess) { set_error("rocprim scan64 size query: %s", hipGetErrorString(e)); return SGF_E_HIP; }
  size_t t = sort_b > rbk_b ? sort_b : rbk_b;
  if (scan_b > t) t = scan_b;
  if (scan64_b > t) t = scan64_b;
  L->tmp_bytes = align_up(t, 256) + 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L->keys_a = take(m * 8);
  L->keys_b = take(m * 8);
  L->idx_a = take(m * 4);
  L->idx_b = take(m * 4);
  L->ukeys = take(m * 8);
  L->ucnt = take((m + 1) * 4);
  L->ustart = take((m + 1) * 4);
  L->slot = take(m * 4);
  L->rowid = take(m * 4);
  L->j2s = take(m * 4);
  L->flag = take((m + 1) * 4);
  L->fscan = take((m + 1) * 4);
  L->head = take(static_cast<size_t>(nb + 1) * 4);
  L->nsh = take(static_cast<size_t>(nb + 1) * 4);
  L->nshp = take(static_cast<size_t>(nb + 1) * 4);
  L->rowblk = take(static_cast<size_t>(n + 1) * 4);
  L->i64 = take(static_cast<size_t>((n > nb ? n : nb) + 1) * 8);
  L->count = take(256);
  L->tmp = take(L->tmp_bytes);
  L->total = off;
  return SGF_OK;
}

// ---- variable row blocks + dense tiles (the plan of k_spmm_tile_bf16 in spmm_tile.hip) ------------------------------
// row_block[r] = b with blk_row[b] <= r < blk_row[b + 1]
__global__ void k_row_block(const int32_t* __restrict__ blk_row, int64_t nb, int64_t n, int32_t* __restrict__ row_block) {
  int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; r < n; r += stride) {
    int64_t lo = 0, hi = nb;   // blk_row[lo] <= r < blk_row[hi]
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (blk_row[mid] <= r) lo = mid; else hi = mid;
    }
    row_block[r] = static_cast<int32_t>(lo);
  }
}

// staged rows per block rounded up to whole 32-source chunks; fragments per block = row tiles x k-steps;
// *bad = 1 if a block has no rows or more than 256
__global__ void k_pad_counts(const int32_t* __restrict__ nsh, const int32_t* __restrict__ blk_row, int64_t nb,
                             int32_t pad, int32_t* __restrict__ nshp, int64_t* __restrict__ frags,
                             int32_t* __restrict__ bad) {
  int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; b <= nb; b += stride) {
    if (b == nb) { nshp[b] = 0; frags[b] = 0; continue; }
    const int32_t rows = blk_row[b + 1] - blk_row[b];
    if (rows <= 0 || rows > 256) *bad = 1;
    const int32_t s = (nsh[b] + pad - 1) / pad * pad;
    nshp[b] = s;
    frags[b] = static_cast<int64_t>((rows + 31) / 32) * (s / 16);
  }
}

// padding slots of a block's staged list point at the block's first row (a valid row; its tile column is all zero)
__global__ void k_pad_fill(const int32_t* __restrict__ nsh, const int32_t* __restrict__ sh_ptr,
                           const int32_t* __restrict__ blk_row, int64_t nb, int32_t* __restrict__ sh_cols) {
  int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; b < nb; b += stride)
    for (int32_t s = sh_ptr[b] + nsh[b]; s < sh_ptr[b + 1]; ++s) sh_cols[s] = blk_row[b];
}

// gathered entries per row, rounded up to an even count (see k_rem_copy)
__global__ void k_rem_len(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ nlds, int64_t n,
                          int64_t* __restrict__ len) {
  int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; r <= n; r += stride) len[r] = r < n ? ((rowptr[r + 1] - rowptr[r] - nlds[r] + 1) & ~static_cast<int64_t>(1)) : 0;
}

__global__ void k_tile_stats(const uint32_t* __restrict__ fscan, int64_t nnz, const int32_t* __restrict__ sh_ptr,
                             const int64_t* __restrict__ tile_ptr, const int64_t* __restrict__ rem_rowptr, int64_t nb,
                             int64_t n, const uint32_t* __restrict__ n_unique, const int32_t* __restrict__ bad,
                             int64_t* __restrict__ stats) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[0] = fscan[nnz];        // stored entries inside the dense tiles
    stats[1] = sh_ptr[nb];        // staged rows, summed over blocks (incl. padding)
    stats[2] = *n_unique;
    stats[3] = nnz;
    stats[4] = tile_ptr[nb];      // 2 KiB fragments
    stats[5] = rem_rowptr[n];     // entries left on the gather path
    stats[6] = 0;
    stats[7] = *bad;
  }
}

// one thread per row: its tile entries (the first nlds[r] of the row, codes = 0x80000000 | slot) are ADDED, in stored
// order, into the fp32 staging image of the block's tile; a (row, slot) cell belongs to this thread alone, so
// duplicate edges are summed sequentially like the plain kernel sums them
__global__ void k_tile_rows(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ ecode,
                            const float* __restrict__ eval, const int32_t* __restrict__ nlds,
                            const int32_t* __restrict__ blk_row, int64_t nb, const int64_t* __restrict__ tile_ptr,
                            int64_t n, float* __restrict__ tiles) {
  int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; r < n; r += stride) {
    const int32_t nl = nlds[r];
    if (nl == 0) continue;
    int64_t lo = 0, hi = nb;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (blk_row[mid] <= r) lo = mid; else hi = mid;
    }
    const int32_t m = static_cast<int32_t>(r - blk_row[lo]);
    const int32_t rt = (blk_row[lo + 1] - blk_row[lo] + 31) / 32;
    const int64_t base = tile_ptr[lo];
    const int64_t e0 = rowptr[r];
    for (int32_t k = 0; k < nl; ++k) {
      const uint32_t c = static_cast<uint32_t>(ecode[e0 + k]) & 0x7fffffffu;
      const int64_t f = base + (static_cast<int64_t>(c >> 5) * rt + (m >> 5)) * 2 + ((c >> 4) & 1);
      const uint32_t kk = c & 15u;
      const uint32_t lane = static_cast<uint32_t>(m & 31) + 32u * (kk >> 3);
      tiles[f * 512 + lane * 8 + (kk & 7u)] += eval[e0 + k];
    }
  }
}

// fp32 staging image -> matrix-core A fragments, in place: fragment = [hi: 64 lanes x 8 bf16][lo: 64 lanes x 8 bf16],
// value = hi + lo with hi = bf16(v), lo = bf16(v - hi) (relative error <= 2^-17).  One wave per fragment.
__global__ __launch_bounds__(256) void k_tile_convert(float* __restrict__ tiles, int64_t nfrag) {
  const int lane = threadIdx.x & 63;
  int64_t f = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
  for (; f < nfrag; f += stride) {
    float* p = tiles + f * 512;
    const float4 a = *reinterpret_cast<const float4*>(p + lane * 8);
    const float4 b = *reinterpret_cast<const float4*>(p + lane * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint16_t h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = f32_to_bf16(v[j]);
      l[j] = f32_to_bf16(v[j] - bf16_to_f32(h[j]));
    }
    uint4 oh, ol;
    oh.x = h[0] | (static_cast<uint32_t>(h[1]) << 16); oh.y = h[2] | (static_cast<uint32_t>(h[3]) << 16);
    oh.z = h[4] | (static_cast<uint32_t>(h[5]) << 16); oh.w = h[6] | (static_cast<uint32_t>(h[7]) << 16);
    ol.x = l[0] | (static_cast<uint32_t>(l[1]) << 16); ol.y = l[2] | (static_cast<uint32_t>(l[3]) << 16);
    ol.z = l[4] | (static_cast<uint32_t>(l[5]) << 16); ol.w = l[6] | (static_cast<uint32_t>(l[7]) << 16);
    __builtin_amdgcn_s_waitcnt(0);                      // every lane's 32 bytes are in registers before any store
    __builtin_amdgcn_wave_barrier();
    uint4* q = reinterpret_cast<uint4*>(p);
    q[lane] = oh;
    q[64 + lane] = ol;
  }
}

__global__ void k_rem_copy(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ ecode,
                           const float* __restrict__ eval, const int32_t* __restrict__ nlds,
                           const int64_t* __restrict__ rem_rowptr, int64_t n, int64_t nnz,
                           int32_t* __restrict__ rem_col, float* __restrict__ rem_val) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (; e < nnz; e += stride) {
    const int64_t r = row_of(rowptr, n, e);
    const int64_t k = e - rowptr[r];
    if (k >= nlds[r]) {
      const int64_t pos = rem_rowptr[r] + k - nlds[r];
      rem_col[pos] = ecode[e];
      rem_val[pos] = eval[e];
      // a row's share is padded to an even count (the kernel fetches two rows of X per instruction and never
      // lets a pair straddle two target rows): the padding repeats the row's last source with value 0
      if (e + 1 == rowptr[r + 1] && ((k - nlds[r]) & 1) == 0) {
        rem_col[pos + 1] = ecode[e];
        rem_val[pos + 1] = 0.f;
      }
    }
  }
}

// The plan proper.  blk_row == nullptr: fixed blocks of rows_per_block rows (sgf_spmm_plan); otherwise nb variable
// blocks, staged counts padded to `pad` and the tile / remainder offsets computed as well (sgf_spmm_tile_plan).
int plan_core(const int64_t* rowptr, const int32_t* colind, const float* val, int64_t n, int64_t nnz,
              int32_t rows_per_block, const int32_t* blk_row, int64_t nb, int32_t lds_rows, uint32_t min_count,
              int32_t pad, int64_t long_len, int32_t* ecode, float* eval, int32_t* nlds, int32_t* sh_ptr,
              int32_t* sh_cols, int64_t* tile_ptr, int64_t* rem_rowptr, int64_t* stats, void* workspace,
              size_t workspace_bytes, hipStream_t st, const char* fn) {
  const unsigned B = bits_for(n), Bb = bits_for(nb);
  SGF_REQUIRE(B + Bb + kCountBits <= 63, SGF_E_UNSUPPORTED, "%s: key does not fit 64 bits", fn);
  Layout L;
  int rc = make_layout(nnz, nb, n, &L);
  if (rc != SGF_OK) return rc;
  SGF_REQUIRE(workspace && workspace_bytes >= L.total, SGF_E_WORKSPACE, "%s: workspace %zu < %zu", fn,
              workspace_bytes, L.total);
  char* ws = static_cast<char*>(workspace);
  uint64_t* ka = reinterpret_cast<uint64_t*>(ws + L.keys_a);
  uint64_t* kb = reinterpret_cast<uint64_t*>(ws + L.keys_b);
  uint32_t* ia = reinterpret_cast<uint32_t*>(ws + L.idx_a);
  uint32_t* ib = reinterpret_cast<uint32_t*>(ws + L.idx_b);
  uint64_t* uk = reinterpret_cast<uint64_t*>(ws + L.ukeys);
  uint32_t* uc = reinterpret_cast<uint32_t*>(ws + L.ucnt);
  uint32_t* us = reinterpret_cast<uint32_t*>(ws + L.ustart);
  int32_t* slot = reinterpret_cast<int32_t*>(ws + L.slot);
  int32_t* rowid = reinterpret_cast<int32_t*>(ws + L.rowid);
  uint32_t* j2s = reinterpret_cast<uint32_t*>(ws + L.j2s);
  uint32_t* flag = reinterpret_cast<uint32_t*>(ws + L.flag);
  uint32_t* fscan = reinterpret_cast<uint32_t*>(ws + L.fscan);
  uint32_t* head = reinterpret_cast<uint32_t*>(ws + L.head);
  int32_t* nsh = reinterpret_cast<int32_t*>(ws + L.nsh);
  int32_t* nshp = reinterpret_cast<int32_t*>(ws + L.nshp);
  int32_t* rowblk = reinterpret_cast<int32_t*>(ws + L.rowblk);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + L.count);
  int32_t* bad = reinterpret_cast<int32_t*>(cnt + 8);
  const size_t m = static_cast<size_t>(nnz);
  const bool tiles = blk_row != nullptr;

  if (n > 0) SGF_CHECK_HIP(hipMemsetAsync(nlds, 0, static_cast<size_t>(n) * 4, st));
  if (nnz > 0) SGF_CHECK_HIP(hipMemsetAsync(slot, 0xff, m * 4, st));   // -1: no LDS slot
  SGF_CHECK_HIP(hipMemsetAsync(nsh, 0, static_cast<size_t>(nb + 1) * 4, st));
  SGF_CHECK_HIP(hipMemsetAsync(head, 0, static_cast<size_t>(nb + 1) * 4, st));
  SGF_CHECK_HIP(hipMemsetAsync(cnt, 0, 256, st));
  if (nnz == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(sh_ptr, 0, static_cast<size_t>(nb + 1) * 4, st));
    SGF_CHECK_HIP(hipMemsetAsync(stats, 0, (tiles ? 8 : 4) * sizeof(int64_t), st));
    if (tiles) {
      SGF_CHECK_HIP(hipMemsetAsync(tile_ptr, 0, static_cast<size_t>(nb + 1) * 8, st));
      SGF_CHECK_HIP(hipMemsetAsync(rem_rowptr, 0, static_cast<size_t>(n + 1) * 8, st));
    }
    return SGF_OK;
  }
  if (tiles) {
    hipLaunchKernelGGL(k_row_block, dim3(grid_for(n)), dim3(kThreads), 0, st, blk_row, nb, n, rowblk);
    SGF_LAUNCH_CHECK();
  }
  // 1. unique (block, source) pairs and their multiplicities
  hipLaunchKernelGGL(k_keys1, dim3(grid_for(nnz)), dim3(kThreads), 0, st, rowptr, colind, n, nnz, rows_per_block,
                     tiles ? rowblk : static_cast<const int32_t*>(nullptr), long_len, B, Bb, ka, ia, rowid);
  SGF_LAUNCH_CHECK();
  size_t bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::radix_sort_pairs(ws + L.tmp, bytes, ka, kb, ia, ib, m, 0u, B + Bb, st));
  SGF_CHECK_HIP(hipMemsetAsync(uc, 0, (m + 1) * 4, st));
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::reduce_by_key(ws + L.tmp, bytes, kb, rocprim::constant_iterator<uint32_t>(1u), m, uk, uc,
                                       cnt, rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), st));
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, uc, us, 0u, m + 1, rocprim::plus<uint32_t>(), st));
  // 2. per block: the `lds_rows` most-referenced sources with multiplicity >= min_count get slots.
  //    Buffer reuse: keys_a / idx_a are free again (the sorted key1 / entry indices live in keys_b / idx_b);
  //    the sorted key2 overwrites ukeys (last read by k_keys2), its values go to j2s.
  hipLaunchKernelGGL(k_keys2, dim3(grid_for(nnz)), dim3(kThreads), 0, st, uk, uc, cnt, nnz, B, Bb, ka, ia);
  SGF_LAUNCH_CHECK();
  uint64_t* k2s = uk;
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::radix_sort_pairs(ws + L.tmp, bytes, ka, k2s, ia, j2s, m, 0u, B + Bb + kCountBits, st));
  hipLaunchKernelGGL(k_block_heads, dim3(grid_for(nnz)), dim3(kThreads), 0, st, k2s, nnz, B, head);
  SGF_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_slots, dim3(grid_for(nnz)), dim3(kThreads), 0, st, k2s, j2s, nnz, B, head, lds_rows, min_count,
                     slot, nsh);
  SGF_LAUNCH_CHECK();
  const int32_t* counts = nsh;
  if (tiles) {
    // slots in ascending source order (see k_reslot_*); scratch: idx_a (free since the sort), flag / fscan (cleared again
    // in step 3), head (free since k_slots)
    uint32_t* blk_u = ia;
    SGF_CHECK_HIP(hipMemsetAsync(blk_u, 0xff, m * 4, st));
    SGF_CHECK_HIP(hipMemsetAsync(flag, 0, (m + 1) * 4, st));
    hipLaunchKernelGGL(k_reslot_mark, dim3(grid_for(nnz)), dim3(kThreads), 0, st, k2s, j2s, nnz, B, slot, blk_u, flag);
    SGF_LAUNCH_CHECK();
    bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, flag, fscan, 0u, m + 1, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(k_reslot_first, dim3(grid_for(nnz)), dim3(kThreads), 0, st, blk_u, fscan, cnt, head);
    SGF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reslot_apply, dim3(grid_for(nnz)), dim3(kThreads), 0, st, blk_u, fscan, flag, head, cnt, slot);
    SGF_LAUNCH_CHECK();
    int64_t* frags = reinterpret_cast<int64_t*>(ws + L.i64);
    hipLaunchKernelGGL(k_pad_counts, dim3(grid_for(nb + 1)), dim3(kThreads), 0, st, nsh, blk_row, nb, pad, nshp, frags,
                       bad);
    SGF_LAUNCH_CHECK();
    bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, frags, tile_ptr, static_cast<int64_t>(0),
                                          static_cast<size_t>(nb + 1), rocprim::plus<int64_t>(), st));
    counts = nshp;
  }
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, counts, sh_ptr, 0, static_cast<size_t>(nb + 1),
                                        rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(k_sh_cols, dim3(grid_for(nnz)), dim3(kThreads), 0, st, k2s, j2s, nnz, B, slot, sh_ptr, sh_cols);
  SGF_LAUNCH_CHECK();
  if (tiles) {
    hipLaunchKernelGGL(k_pad_fill, dim3(grid_for(nb)), dim3(kThreads), 0, st, nsh, sh_ptr, blk_row, nb, sh_cols);
    SGF_LAUNCH_CHECK();
  }
  // 3. entry codes (in CSR position; written into keys_a, free again), then the stable LDS-first partition
  //    inside every row
  int32_t* code_tmp = reinterpret_cast<int32_t*>(ka);
  SGF_CHECK_HIP(hipMemsetAsync(flag, 0, (m + 1) * 4, st));
  hipLaunchKernelGGL(k_codes, dim3(grid_for(nnz)), dim3(kThreads), 0, st, kb, ib, nnz, us, cnt, slot, colind, B, Bb,
                     code_tmp, flag);
  SGF_LAUNCH_CHECK();
  bytes = L.tmp_bytes;
  SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, flag, fscan, 0u, m + 1, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_partition, dim3(grid_for(nnz)), dim3(kThreads), 0, st, rowptr, rowid, code_tmp, flag, fscan,
                     val, nnz, ecode, eval, nlds);
  SGF_LAUNCH_CHECK();
  if (tiles) {
    int64_t* rlen = reinterpret_cast<int64_t*>(ws + L.i64);
    hipLaunchKernelGGL(k_rem_len, dim3(grid_for(n + 1)), dim3(kThreads), 0, st, rowptr, nlds, n, rlen);
    SGF_LAUNCH_CHECK();
    bytes = L.tmp_bytes;
    SGF_CHECK_HIP(rocprim::exclusive_scan(ws + L.tmp, bytes, rlen, rem_rowptr, static_cast<int64_t>(0),
                                          static_cast<size_t>(n + 1), rocprim::plus<int64_t>(), st));
    hipLaunchKernelGGL(k_tile_stats, dim3(1), dim3(64), 0, st, fscan, nnz, sh_ptr, tile_ptr, rem_rowptr, nb, n, cnt, bad,
                       stats);
  } else {
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, st, fscan, nnz, sh_ptr, nb, cnt, stats);
  }
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" size_t sgf_spmm_plan_workspace_bytes(int64_t nnz, int64_t n, int32_t rows_per_block) {
  if (nnz < 0 || n < 0 || rows_per_block <= 0) return 0;
  Layout L;
  if (make_layout(nnz, (n + rows_per_block - 1) / rows_per_block, n, &L) != SGF_OK) return 0;
  return L.total;
}

extern "C" int sgf_spmm_plan(const int64_t* rowptr, const int32_t* colind, const float* val, int64_t n,
                             int64_t nnz, int32_t rows_per_block, int32_t lds_rows, int64_t long_len,
                             int32_t* ecode, float* eval, int32_t* nlds, int32_t* sh_ptr, int32_t* sh_cols,
                             int64_t* stats, void* workspace, size_t workspace_bytes, void* stream) {
  SGF_REQUIRE(n >= 0 && nnz >= 0 && rows_per_block > 0 && lds_rows > 0 && long_len >= 1, SGF_E_INVALID,
              "sgf_spmm_plan: bad size argument");
  SGF_REQUIRE(nnz < (static_cast<int64_t>(1) << 32) - 1, SGF_E_UNSUPPORTED, "sgf_spmm_plan: nnz >= 2^32");
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31) - 1, SGF_E_UNSUPPORTED, "sgf_spmm_plan: n too large");
  const int64_t nb = (n + rows_per_block - 1) / rows_per_block;
  SGF_REQUIRE(nb * static_cast<int64_t>(lds_rows) < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED,
              "sgf_spmm_plan: blocks x lds_rows overflows int32");
  SGF_REQUIRE(rowptr && nlds && sh_ptr && stats && (nnz == 0 || (colind && val && ecode && eval && sh_cols)),
              SGF_E_INVALID, "sgf_spmm_plan: null pointer");
  return plan_core(rowptr, colind, val, n, nnz, rows_per_block, nullptr, nb, lds_rows, 2u, 1, long_len, ecode, eval,
                   nlds, sh_ptr, sh_cols, nullptr, nullptr, stats, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream), "sgf_spmm_plan");
}

// ---- sgf_spmm_tile_*: community-aligned row blocks whose shared sources are multiplied as dense matrix-core tiles ----
extern "C" int sgf_spmm_tile_blocks(const int32_t* comm_sorted, int64_t n, int32_t max_rows, int32_t* blk_row,
                                    int64_t blk_cap, int64_t* nb_out, void* stream) {
  const char* fn = "sgf_spmm_tile_blocks";
  SGF_REQUIRE(n >= 0 && max_rows >= 32 && max_rows <= 256 && max_rows % 32 == 0 && blk_cap >= 0 && nb_out && blk_row,
              SGF_E_INVALID, "%s: bad argument (max_rows must be a multiple of 32 in [32, 256])", fn);
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31) - 1, SGF_E_UNSUPPORTED, "%s: n too large", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<int32_t> out;
  out.reserve(static_cast<size_t>(n / 64 + 16));
  out.push_back(0);
  if (!comm_sorted) {
    for (int64_t r = max_rows; r < n; r += max_rows) out.push_back(static_cast<int32_t>(r));
    if (n > 0) out.push_back(static_cast<int32_t>(n));
  } else if (n > 0) {
    std::vector<int32_t> comm(static_cast<size_t>(n));
    SGF_CHECK_HIP(hipMemcpyAsync(comm.data(), comm_sorted, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, st));
    SGF_CHECK_HIP(hipStreamSynchronize(st));
    int64_t cur = 0;                              // rows of the open block, which starts at out.back()
    for (int64_t i = 0; i < n;) {
      int64_t j = i + 1;
      while (j < n && comm[j] == comm[i]) ++j;
      const int64_t len = j - i;
      if (len > max_rows) {
        if (cur > 0) { out.push_back(static_cast<int32_t>(i)); cur = 0; }
        const int64_t k = (len + max_rows - 1) / max_rows;
        int64_t piece = ((len + k - 1) / k + 31) / 32 * 32;
        if (piece > max_rows) piece = max_rows;
        for (int64_t pos = i; pos < j;) {
          pos = pos + piece < j ? pos + piece : j;
          out.push_back(static_cast<int32_t>(pos));
        }
      } else {
        if (cur + len > max_rows) { out.push_back(static_cast<int32_t>(i)); cur = 0; }
        cur += len;
      }
      i = j;
    }
    if (cur > 0) out.push_back(static_cast<int32_t>(n));
  }
  const int64_t nb = static_cast<int64_t>(out.size()) - 1;
  *nb_out = nb;
  SGF_REQUIRE(nb <= blk_cap, SGF_E_WORKSPACE, "%s: %lld blocks, room for %lld", fn, static_cast<long long>(nb),
              static_cast<long long>(blk_cap));
  SGF_CHECK_HIP(hipMemcpyAsync(blk_row, out.data(), out.size() * 4, hipMemcpyHostToDevice, st));
  SGF_CHECK_HIP(hipStreamSynchronize(st));        // `out` dies with this call
  return SGF_OK;
}

extern "C" size_t sgf_spmm_tile_plan_workspace_bytes(int64_t nnz, int64_t n, int64_t nb) {
  if (nnz < 0 || n < 0 || nb < 0) return 0;
  Layout L;
  if (make_layout(nnz, nb, n, &L) != SGF_OK) return 0;
  return L.total;
}

extern "C" int sgf_spmm_tile_plan(const int64_t* rowptr, const int32_t* colind, const float* val, int64_t n,
                                  int64_t nnz, const int32_t* blk_row, int64_t nb, int32_t cap, int32_t min_count,
                                  int64_t long_len, int32_t* ecode, float* eval, int32_t* nlds, int32_t* sh_ptr,
                                  int32_t* sh_cols, int64_t* tile_ptr, int64_t* rem_rowptr, int64_t* stats,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_spmm_tile_plan";
  SGF_REQUIRE(n >= 0 && nnz >= 0 && nb >= 0 && long_len >= 1, SGF_E_INVALID, "%s: bad size argument", fn);
  SGF_REQUIRE(cap >= 32 && cap <= 1024 && cap % 32 == 0, SGF_E_INVALID, "%s: cap must be a multiple of 32 in [32, 1024]", fn);
  SGF_REQUIRE(min_count >= 1 && min_count <= 1000, SGF_E_INVALID, "%s: min_count outside [1, 1000]", fn);
  SGF_REQUIRE(nnz < (static_cast<int64_t>(1) << 32) - 1, SGF_E_UNSUPPORTED, "%s: nnz >= 2^32", fn);
  SGF_REQUIRE(n < (static_cast<int64_t>(1) << 31) - 1, SGF_E_UNSUPPORTED, "%s: n too large", fn);
  SGF_REQUIRE(nb * static_cast<int64_t>(cap) < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED,
              "%s: blocks x cap overflows int32", fn);
  SGF_REQUIRE(rowptr && nlds && sh_ptr && tile_ptr && rem_rowptr && stats && blk_row &&
                  (nnz == 0 || (colind && val && ecode && eval && sh_cols)),
              SGF_E_INVALID, "%s: null pointer", fn);
  return plan_core(rowptr, colind, val, n, nnz, 1, blk_row, nb, cap, static_cast<uint32_t>(min_count), 32, long_len,
                   ecode, eval, nlds, sh_ptr, sh_cols, tile_ptr, rem_rowptr, stats, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream), fn);
}

extern "C" int sgf_spmm_tile_fill(const int64_t* rowptr, const int32_t* ecode, const float* eval, const int32_t* nlds,
                                  int64_t n, int64_t nnz, const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr,
                                  int64_t n_frag, const int64_t* rem_rowptr, void* tiles, int32_t* rem_col,
                                  float* rem_val, void* stream) {
  const char* fn = "sgf_spmm_tile_fill";
  SGF_REQUIRE(n >= 0 && nnz >= 0 && nb >= 0 && n_frag >= 0, SGF_E_INVALID, "%s: bad size argument", fn);
  if (n == 0 || nnz == 0) return SGF_OK;
  SGF_REQUIRE(rowptr && ecode && eval && nlds && blk_row && tile_ptr && rem_rowptr && (n_frag == 0 || tiles),
              SGF_E_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_frag > 0) {
    SGF_CHECK_HIP(hipMemsetAsync(tiles, 0, static_cast<size_t>(n_frag) * 2048, st));
    hipLaunchKernelGGL(k_tile_rows, dim3(grid_for(n)), dim3(kThreads), 0, st, rowptr, ecode, eval, nlds, blk_row, nb,
                       tile_ptr, n, static_cast<float*>(tiles));
    SGF_LAUNCH_CHECK();
    int64_t g = (n_frag + 3) / 4;
    if (g > kNumCU * 16) g = kNumCU * 16;
    hipLaunchKernelGGL(k_tile_convert, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, static_cast<float*>(tiles),
                       n_frag);
    SGF_LAUNCH_CHECK();
  }
  if (rem_col && rem_val) {
    hipLaunchKernelGGL(k_rem_copy, dim3(grid_for(nnz)), dim3(kThreads), 0, st, rowptr, ecode, eval, nlds, rem_rowptr, n,
                       nnz, rem_col, rem_val);
    SGF_LAUNCH_CHECK();
  }
  return SGF_OK;
}
