// spmm_pack.hip — the A tiles of sgf_spmm_tile_plan / _fill in the form the kernel streams them (csrc/spmm_tile.hip).
//
// The plan's tiles are dense MFMA A fragments: 2 KiB (hi + lo bf16 of 32 rows x 16 staged sources) whether 5 or 500 of
// the 512 cells are occupied.  On a community graph the cells are 10-25 % occupied, so the fragments were 2.4 GB of the
// kernel's ~14.5 GB of fabric traffic per launch (ogbn-products shape; profiles/r03_spmm_tile.md) — three times the
// 8 bytes per entry the same entries cost in CSR form.  The kernel is bound by that traffic, so the fragments are
// PACKED: per GROUP = (row block b, row tile w, chunk q of 32 staged sources) = the two fragments one wave multiplies
// per chunk,
//   * sparse (<= kSparseMax occupied cells): the occupied cells as 8-byte entries {byte offset of the cell's hi element
//     inside the wave's 4 KiB fragment area, hi | lo << 16}, in cell order, padded to an even count — the wave DMAs them
//     into LDS, clears its fragment area and scatters them (2 x ds_write_b16 per entry);
//   * dense: the 4 KiB as they are (DMA straight into the fragment area).
// Groups are numbered wave-major — g = tile_ptr[b] / 2 + w * NQ_b + q — and described by grp[2 g] = offset into the
// pool in 16-byte units, grp[2 g + 1] = occupied cells (sparse) or -1 (dense).  Same values, same arithmetic as the
// dense fragments: the kernel rebuilds them bit for bit.  Deterministic.
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sgf {
namespace {

constexpr int kSparseMax = 384;                 // 3 KiB of entries against 4 KiB dense (and <= 6 scatter rounds per wave)

struct GroupRef {
  int64_t g;          // group index (wave-major)
  int64_t f0;         // first of its two fragments in the plan's [chunk][row tile] order
};

// groups of block b: local index l = w * NQ + q
__device__ __forceinline__ GroupRef group_of(const int32_t* blk_row, const int64_t* tile_ptr, int64_t b, int l, int* count) {
  const int rows = blk_row[b + 1] - blk_row[b];
  const int rt = (rows + 31) >> 5;
  const int64_t base = tile_ptr[b];
  const int ng = static_cast<int>((tile_ptr[b + 1] - base) >> 1);
  *count = ng;
  const int nq = rt > 0 ? ng / rt : 0;
  const int w = nq > 0 ? l / nq : 0, q = nq > 0 ? l % nq : 0;
  return GroupRef{(base >> 1) + l, base + (static_cast<int64_t>(q) * rt + w) * 2};
}

// occupied cells of this lane in fragment pair f0 (k-step s = 0, 1): bit 8 s + j = cell (lane, j) of k-step s
__device__ __forceinline__ uint32_t lane_cells(const uint4* tiles, int64_t f0, int lane, uint4 (&h)[2], uint4 (&l)[2]) {
  uint32_t m = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    h[s] = tiles[(f0 + s) * 128 + lane];
    l[s] = tiles[(f0 + s) * 128 + 64 + lane];
    const uint32_t hw[4] = {h[s].x, h[s].y, h[s].z, h[s].w}, lw[4] = {l[s].x, l[s].y, l[s].z, l[s].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t hv = (hw[j >> 1] >> (16 * (j & 1))) & 0xffffu, lv = (lw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      if ((hv | lv) != 0) m |= 1u << (8 * s + j);
    }
  }
  return m;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_pack_count(const int32_t* __restrict__ blk_row, const int64_t* __restrict__ tile_ptr,
                                                    int64_t nb, const uint4* __restrict__ tiles, int64_t* __restrict__ units,
                                                    int32_t* __restrict__ grp) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
    int ng;
    group_of(blk_row, tile_ptr, b, 0, &ng);
    for (int l = wv; l < ng; l += 4) {
      const GroupRef r = group_of(blk_row, tile_ptr, b, l, &ng);
      uint4 h[2], lo[2];
      const int cells = wave_sum(__popc(lane_cells(tiles, r.f0, lane, h, lo)));
      if (lane == 0) {
        const bool sparse = cells <= kSparseMax;
        units[r.g] = sparse ? (cells + 1) / 2 : 256;
        grp[2 * r.g + 1] = sparse ? cells : -1;
      }
    }
  }
}

__global__ void k_pack_offsets(const int64_t* __restrict__ offs, int64_t ng, int32_t* __restrict__ grp,
                               int64_t* __restrict__ pool_units) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g < ng) grp[2 * g] = static_cast<int32_t>(static_cast<uint32_t>(offs[g]));
  if (g == 0) *pool_units = offs[ng];
}

__global__ __launch_bounds__(256) void k_pack_write(const int32_t* __restrict__ blk_row, const int64_t* __restrict__ tile_ptr,
                                                    int64_t nb, const uint4* __restrict__ tiles,
                                                    const int32_t* __restrict__ grp, uint4* __restrict__ pool) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
    int ng;
    group_of(blk_row, tile_ptr, b, 0, &ng);
    for (int l = wv; l < ng; l += 4) {
      const GroupRef r = group_of(blk_row, tile_ptr, b, l, &ng);
      uint4* dst = pool + static_cast<uint32_t>(grp[2 * r.g]);
      const int cells = grp[2 * r.g + 1];
      uint4 h[2], lo[2];
      const uint32_t m = lane_cells(tiles, r.f0, lane, h, lo);
      if (cells < 0) {                                   // dense: [k-step][hi, lo][lane] x 16 bytes, as the kernel reads it
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          dst[s * 128 + lane] = h[s];
          dst[s * 128 + 64 + lane] = lo[s];
        }
        continue;
      }
      // sparse: cell order = (k-step, lane, j); this lane's first entry = occupied cells of the lanes before it
      const int c0 = __popc(m & 0xffu), c1 = __popc(m >> 8);
      int p0 = c0, p1 = c1;                              // inclusive prefix sums over the lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t0 = __shfl_up(p0, off, 64), t1 = __shfl_up(p1, off, 64);
        if (lane >= off) { p0 += t0; p1 += t1; }
      }
      const int tot0 = __shfl(p0, 63, 64);
      uint2* ent = reinterpret_cast<uint2*>(dst);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        int at = s == 0 ? p0 - c0 : tot0 + p1 - c1;
        const uint32_t hw[4] = {h[s].x, h[s].y, h[s].z, h[s].w}, lw[4] = {lo[s].x, lo[s].y, lo[s].z, lo[s].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (m & (1u << (8 * s + j))) {
            const uint32_t hv = (hw[j >> 1] >> (16 * (j & 1))) & 0xffffu, lv = (lw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            ent[at++] = make_uint2(static_cast<uint32_t>(s * 2048 + lane * 16 + j * 2), hv | (lv << 16));
          }
        }
      }
      if ((cells & 1) && lane == 0) ent[cells] = make_uint2(0u, 0u);   // padding to 16 bytes (never scattered)
    }
  }
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" int32_t sgf_spmm_tile_sparse_len(void) { return kSparseMax; }

extern "C" size_t sgf_spmm_tile_pack_workspace_bytes(int64_t n_frag) {
  if (n_frag < 0) return 0;
  const size_t ng = static_cast<size_t>(n_frag / 2);
  size_t scan_b = 0;
  if (rocprim::exclusive_scan(nullptr, scan_b, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr),
                              static_cast<int64_t>(0), ng + 1, rocprim::plus<int64_t>()) != hipSuccess)
    return 0;
  return 2 * align_up((ng + 1) * sizeof(int64_t), 256) + align_up(scan_b, 256) + 256;
}

extern "C" int sgf_spmm_tile_pack_layout(const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr, const void* tiles,
                                         int64_t n_frag, int32_t* grp, int64_t* pool_units, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_spmm_tile_pack_layout";
  SGF_REQUIRE(nb >= 0 && n_frag >= 0 && n_frag % 2 == 0, SGF_E_INVALID, "%s: bad size argument", fn);
  SGF_REQUIRE(pool_units, SGF_E_INVALID, "%s: null pointer", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t ng = n_frag / 2;
  if (nb == 0 || ng == 0) {
    SGF_CHECK_HIP(hipMemsetAsync(pool_units, 0, sizeof(int64_t), st));
    return SGF_OK;
  }
  SGF_REQUIRE(blk_row && tile_ptr && tiles && grp, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(ng < (static_cast<int64_t>(1) << 24), SGF_E_UNSUPPORTED, "%s: pool beyond 2^32 x 16 bytes", fn);
  SGF_REQUIRE(workspace && workspace_bytes >= sgf_spmm_tile_pack_workspace_bytes(n_frag), SGF_E_WORKSPACE,
              "%s: workspace too small", fn);
  char* ws = static_cast<char*>(workspace);
  const size_t arr = align_up(static_cast<size_t>(ng + 1) * sizeof(int64_t), 256);
  int64_t* units = reinterpret_cast<int64_t*>(ws);
  int64_t* offs = reinterpret_cast<int64_t*>(ws + arr);
  void* tmp = ws + 2 * arr;
  size_t tmp_b = workspace_bytes - 2 * arr;
  SGF_CHECK_HIP(hipMemsetAsync(units + ng, 0, sizeof(int64_t), st));
  const unsigned grid = static_cast<unsigned>(nb < kNumCU * 8 ? nb : kNumCU * 8);
  hipLaunchKernelGGL(k_pack_count, dim3(grid), dim3(256), 0, st, blk_row, tile_ptr, nb, static_cast<const uint4*>(tiles),
                     units, grp);
  SGF_LAUNCH_CHECK();
  SGF_CHECK_HIP(rocprim::exclusive_scan(tmp, tmp_b, units, offs, static_cast<int64_t>(0), static_cast<size_t>(ng + 1),
                                        rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(k_pack_offsets, dim3(static_cast<unsigned>((ng + 255) / 256)), dim3(256), 0, st, offs, ng, grp,
                     pool_units);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_spmm_tile_pack(const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr, const void* tiles,
                                  int64_t n_frag, const int32_t* grp, void* pool, int64_t pool_units, void* stream) {
  const char* fn = "sgf_spmm_tile_pack";
  SGF_REQUIRE(nb >= 0 && n_frag >= 0 && n_frag % 2 == 0 && pool_units >= 0, SGF_E_INVALID, "%s: bad size argument", fn);
  if (nb == 0 || n_frag == 0 || pool_units == 0) return SGF_OK;
  SGF_REQUIRE(blk_row && tile_ptr && tiles && grp && pool, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(pool_units < (static_cast<int64_t>(1) << 32), SGF_E_UNSUPPORTED, "%s: pool beyond 2^32 x 16 bytes", fn);
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(pool) % 16 == 0, SGF_E_INVALID, "%s: pool must be 16-byte aligned", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = static_cast<unsigned>(nb < kNumCU * 8 ? nb : kNumCU * 8);
  hipLaunchKernelGGL(k_pack_write, dim3(grid), dim3(256), 0, st, blk_row, tile_ptr, nb, static_cast<const uint4*>(tiles), grp,
                     static_cast<uint4*>(pool));
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
