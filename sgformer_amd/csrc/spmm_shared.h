// spmm_shared.h — pieces shared by the SpMM kernels of spmm.hip and spmm_tile.hip (not part of the C ABI).
#pragma once
#include "common.h"

#include <climits>

namespace sgf {

// Blocks are dealt to the 8 XCDs in STRIPES: the hardware places block b on XCD b % 8; virtual block
// v(b) is chosen so that each XCD walks chunks of kChunkBlocks consecutive virtual blocks (4096 rows: its
// private 4 MiB L2 caches the X rows of one graph neighbourhood instead of 1/8 of everybody's), and
// consecutive chunks go round-robin over the XCDs.  Round-robin rather than 8 contiguous ranges: when the
// degree correlates with the node id (datasets sorted by popularity or time) contiguous ranges put most of
// the work on one XCD — a power-law graph ran 1.5x slower that way.  Bijective for any nblocks.
static constexpr int64_t kChunkBlocks = 1024;

static __device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nblocks, int64_t chunk = kChunkBlocks) {
  const int64_t stripe = kNumXCD * chunk;
  const int64_t full = nblocks / stripe * stripe;
  if (b >= full) return b;   // ragged tail: identity
  const int64_t xcd = b % kNumXCD;
  const int64_t j = b / kNumXCD;              // arrival order inside this XCD
  return ((j / chunk) * kNumXCD + xcd) * chunk + j % chunk;
}

// ---- long rows (power-law graphs: ogbn-products has rows of 17 k entries) ----------------------------
// One wave walks a row serially with UNROLL gathers in flight, so a hub row would be a long latency-bound
// tail.  Rows longer than `long_len` are therefore not processed by the row kernels: the wave that meets
// one reserves ceil(len / kSegLen) consecutive queue slots with ONE atomic and enqueues (row, seg, k);
// k_spmm_long_seg then reduces each segment with a whole workgroup (wave-strided slices, fixed-order
// LDS combine) into an fp32 partial, and k_spmm_long_fin adds a row's partials in segment order.  The
// slot reservation order is arbitrary, the arithmetic is not: results are deterministic.
static constexpr int kSegLen = 1024;

struct LongEntry {
  int32_t row, seg, k, pad;
};

struct LongQueue {
  int32_t* count;       // [1]
  LongEntry* entries;   // [cap]
  int32_t cap;
  int64_t long_len;     // rows with more stored entries than this are queued (INT64_MAX: never)
};

static __device__ __forceinline__ void push_long_row(const LongQueue& q, int64_t row, int64_t len, int lane_in_group,
                                              int group_width) {
  const int k = static_cast<int>((len + kSegLen - 1) / kSegLen);
  int base = 0;
  if (lane_in_group == 0) base = atomicAdd(q.count, k);
  base = __shfl(base, (threadIdx.x & 63) - lane_in_group, 64);
  for (int s = lane_in_group; s < k; s += group_width)
    if (base + s < q.cap) q.entries[base + s] = LongEntry{static_cast<int32_t>(row), s, k, 0};
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


// Rows queued by a SpMM kernel (LongQueue) reduced by whole workgroups and written to y (spmm.hip).
int spmm_long_rows(int dtype, const int64_t* rowptr, const int32_t* colind, const float* val, const void* x, int64_t ldx,
                   int32_t d, const LongQueue& lq, float* partial, void* y, int64_t ldy, hipStream_t st);

}  // namespace sgf
