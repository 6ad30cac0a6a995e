// spmm_tile.hip — T2 on a graph with community structure: the shared part of every row block as a DENSE tile on the
// matrix cores, the rest as a gather stream, one rounding.   (large/ours.py:34, torch_sparse.matmul, sum-reduce)
//
// Why (profiles/r02_spmm_structured.md): a kernel that fetches one 512-byte row of X per stored entry is
// capped by the per-CU vector-memory path (58 GB of rows per launch at ogbn-products size = 2.2 ms even at a 100 % L2 hit
// rate), and r02's LDS-staged row blocks paid as many issue slots for an LDS-served entry as for a gathered one.  After
// sgf_reorder a community is a run of consecutive rows whose stored entries mostly point back into the same run, i.e.
// the diagonal blocks of A are 15-60 % dense.  So, per block of <= 128 rows (plan: spmm_plan.hip, sgf_spmm_tile_*):
//
//   staged sources   the S sources at least `min_count` of the block's entries reference (own community first), S padded to
//                    whole chunks of 32; their rows of X stream ONCE per block through a 2 x 16 KiB LDS ring, chunk by chunk,
//                    by LDS-DMA (global_load_lds_dwordx4: no staging registers), written directly in the image the
//                    transposing LDS read wants (512-byte units of four [4 keys][16 columns] subtiles);
//   dense tile       A[rows of the block, staged sources] as matrix-core A fragments, value = hi + lo bf16 (|error| <= 2^-17
//                    relative), built once per graph and PACKED in HBM (spmm_pack.hip: sparse groups as 8-byte entries the
//                    wave scatters over its cleared 4 KiB fragment area in LDS, dense groups copied there by DMA); wave t
//                    owns the 32-row tile t and all 256 feature columns: per chunk 4 ds_read_b128 of fragments, 32
//                    ds_read_b64_tr_b16 (B fragments: k = source, n = feature, straight out of the row-major staged rows)
//                    and 32 v_mfma_f32_32x32x16_bf16 into 8 x 16 accumulator registers — products of bf16 values are
//                    exact in fp32, accumulation is fp32; nothing but DMA is loaded inside the chunk loop;
//   remainder        the entries that are not in the tile (other communities, 20-40 %) keep their CSR form (rem_*) and
//                    are gathered two per 16-byte-per-lane load as in k_spmm_seg_bf16x2 (spmm.hip), 16 rows per pass,
//                    three sets of 8 pair loads in flight; the tile's fp32 partial sums wait in a per-wave LDS patch (the
//                    ring's memory, free by then) and are added when a row is finished: ONE rounding to bf16.  Hub rows go
//                    to the long-row queue; their wave steps over them.
//
// A staged row is read from L2 / HBM once per block instead of once per entry, the dense part costs 0.2 ms of matrix-core
// time at products size instead of 2 ms of gathers, and what is left on the gather path is the part no blocking can
// serve.  Deterministic: tile order and stream order are fixed by the plan.  Non-finite values of X spread inside a block
// (0 x inf = NaN through the tile's zero cells) — the plain kernels confine them to actual neighbours.
#include "common.h"
#include "spmm_shared.h"

#include <type_traits>

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define SGF_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define SGF_GLB(T, p) ((const __attribute__((address_space(1))) T*)(p))

typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32v2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2));
}

// lane l and lane l ^ 32 added (lower half first): one v_permlane32_swap instead of an LDS round trip (ds_bpermute)
__device__ __forceinline__ float sum_halves(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = {lo, lo}, r[1] = {hi, hi}
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr int kChunk = 32;                    // staged sources per ring slot
constexpr int kRingPairs = 8;                 // epilogue: pair loads (two rows of X each) per register set
constexpr int kStashPad = 96;                 // positions of padding behind an epoch (>= 10 kRingPairs: see the gather loop)
constexpr uint32_t kOobOffset = 0xfffffc00u;  // a gather offset beyond any x (x_bytes < 2^32 - 2048): the load returns zeros
constexpr int kStash = 224;                   // epilogue: {source, value} pairs parked in LDS per epoch
constexpr int kMaxStaged = 1024;

struct TilePlanArgs {
  const int32_t* blk_row;        // [nb + 1]
  const int32_t* sh_ptr;         // [nb + 1], multiples of 32
  const int32_t* sh_cols;        // staged source ids
  const int64_t* tile_ptr;       // [nb + 1], in 2 KiB fragments (two per group)
  const int32_t* grp;            // packed tiles (spmm_pack.hip): per group {pool offset in 16-byte units, cells or -1}
  const uint4* pool;
  const int64_t* rem_rowptr;     // [n + 1]
  const int32_t* rem_col;
  const float* rem_val;
};

// NCT column tiles of 32 features: d = 32 * NCT
// NW waves per block (one 32-row tile each): 4 (blocks of <= 128 rows, two blocks per CU) or 8 (<= 256 rows, one)
template <int NCT, int NW, bool kDebug>
__global__ __launch_bounds__(NW * 64, 2) void k_spmm_tile_bf16(
    TilePlanArgs P, const uint16_t* __restrict__ x, uint32_t pitch, uint32_t x_bytes, uint16_t* __restrict__ y,
    int64_t ldy, int32_t nb, int32_t chunk_blocks, LongQueue lq, int dbg_arg) {
  const int dbg = kDebug ? dbg_arg : 0;                  // timing experiments (SGF_SPMM_TILE_DEBUG) compile away otherwise
  constexpr int D = 32 * NCT;
  constexpr int kRowBytes = D * 2;                       // one staged row
  constexpr int kSlotBytes = kChunk * kRowBytes;         // one ring slot (16 KiB at d = 256)
  constexpr int kPatchRows = 16;                         // rows of fp32 partial sums parked per wave
  constexpr int kPatchBytes = kPatchRows * D * 4;
  constexpr int kStashW = kStash + kStashPad;
  // (three ring slots and X chunks staged two iterations ahead measured the same as two slots and one: the tile phase
  // is bound by the fabric's bandwidth, not by its latency)
  constexpr int kSlots = 2;
  constexpr int kRingBytes = kSlots * kSlotBytes;
  constexpr int kFragBytes = 4096;                       // a wave's A fragments of one chunk: 2 k-steps x {hi, lo} x 1 KiB
  // [ K-loop ring | the waves' A fragments | their packed entries, later the NW patches ][ NW stashes ]: only the patches
  // alias the K-loop buffers, so the first stash is filled before the tile phase runs.  The packed entries of chunk 0 land
  // in ring slot 1 (free until chunk 1 is staged) when the buffers leave no room for a second entry area.
  constexpr int kRawBase = kRingBytes + NW * kFragBytes;
  constexpr bool kRaw0InRing = NW * kFragBytes <= kSlotBytes;
  constexpr int kLoopBytes = kRawBase + NW * kFragBytes * (kRaw0InRing ? 1 : 2);
  constexpr int kRaw0Base = kRaw0InRing ? kSlotBytes : kRawBase + NW * kFragBytes;
  constexpr int kBase = NW * kPatchBytes > kLoopBytes ? NW * kPatchBytes : kLoopBytes;
  constexpr int kDataBytes = kBase + NW * (kStashW * 8);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kDataBytes];
  __shared__ int32_t cols_lds[kMaxStaged];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int b = static_cast<int>(xcd_remap(blockIdx.x, nb, chunk_blocks));
  const int row0 = P.blk_row[b];
  const int nrows = P.blk_row[b + 1] - row0;
  const int RT = (nrows + 31) >> 5;
  const int s0 = P.sh_ptr[b];
  const int S = P.sh_ptr[b + 1] - s0;
  const int NQ = S / kChunk;
  const int64_t g0 = (P.tile_ptr[b] >> 1) + static_cast<int64_t>(wid) * NQ;   // this wave's first group
  const bool mine = wid < RT;                            // this wave owns rows [row0 + 32 wid, ...)


  // ---- staging: chunk q -> ring slot ------------------------------------------------------------------------------
  // Image of a chunk (32 sources x D features): 512-byte UNITS [k-step s][read r][column tile t], each the four
  // [4 keys][16 columns] subtiles ONE ds_read_b64_tr_b16 hands to its four 16-lane groups g = 2 khalf + nhalf
  // (source s 16 + 8 khalf + 4 r + key, columns 32 t + 16 nhalf ...), so that a transposing read covers 512 contiguous
  // bytes, lane l at l * 8 — the layout the instruction serves without bank conflicts (cdna_hip_programming.md T10;
  // the first image of this kernel, [key quad][subtile], was 4-way conflicted: 25 % of all LDS cycles).
  // Staging moves 1 KiB pieces = the units (s, r, 2 p) and (s, r, 2 p + 1): lane l carries the 16 bytes
  //   t' = l >> 5, khalf = (l >> 4) & 1, nhalf = (l >> 3) & 1, key = (l >> 1) & 3, 8 columns at 8 (l & 1):
  // 8 source rows x 128 contiguous bytes per piece.
  // A chunk has 2 NCT pieces, numbered (2 s + r) (NCT / 2) + p; wave w stages pieces w PPW .. w PPW + PPW - 1.
  constexpr int kPiecesPerWave = 2 * NCT / NW;
  static_assert(kPiecesPerWave >= 1 && (NCT / 2) % kPiecesPerWave == 0, "a wave's pieces share (s, r)");
  const int sr_w = wid * kPiecesPerWave / (NCT / 2);                 // 2 s + r of this wave's pieces
  const int p0_w = wid * kPiecesPerWave % (NCT / 2);                 // its first p
  const int src_slot = (sr_w >> 1) * 16 + 8 * ((lane >> 4) & 1) + 4 * (sr_w & 1) + ((lane >> 1) & 3);
  const uint32_t col_byte =
      static_cast<uint32_t>((64 * p0_w + 32 * (lane >> 5) + 16 * ((lane >> 3) & 1) + 8 * (lane & 1)) * 2);
  const int unit0_w = sr_w * NCT + 2 * p0_w;                          // first 512-byte unit of this wave's pieces
  // The DMA is issued from inline assembly on purpose: through the builtin, hipcc orders every later LDS read behind it
  // with s_waitcnt vmcnt(0) — the chunk being staged would be waited for before the chunk being multiplied is read.
  // Opaque to the compiler, the only waits are the explicit vmcnt(0) + barrier at the top of the chunk loop.
  const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(SGF_LDS(unsigned char, smem)));
  auto stage = [&](int q, int slot) {
    const int32_t src = cols_lds[q * kChunk + src_slot];
    const unsigned char* g = reinterpret_cast<const unsigned char*>(x) + static_cast<size_t>(src) * pitch + col_byte;
#pragma unroll
    for (int pc = 0; pc < kPiecesPerWave; ++pc) {
      const uint32_t dst = smem_lds + static_cast<uint32_t>(slot * kSlotBytes + (unit0_w + 2 * pc) * 512);
      const unsigned char* gp = g + pc * 128;
      uint32_t keep;
      if (dbg & 4)                                         // timing experiment: the staged rows as a non-temporal stream
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gp), "s"(dst)
            : "memory");
      else
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(gp), "s"(dst)
          : "memory");
    }
  };

  // ---- this wave's share of the gather stream: requested now, needed after the tile phase ---------------------------
  const int nr = mine ? (nrows - 32 * wid < 32 ? nrows - 32 * wid : 32) : 1;
  const int64_t r_base = mine ? static_cast<int64_t>(row0) + 32 * wid : static_cast<int64_t>(row0);
  float* patch = reinterpret_cast<float*>(smem + wid * kPatchBytes);                // 16 rows x D fp32
  uint32_t* stash_off = reinterpret_cast<uint32_t*>(smem + kBase + wid * (kStashW * 8));   // byte offsets of source rows
  float* stash_val = reinterpret_cast<float*>(stash_off + kStashW);
  const int half = lane >> 5;
  const bool hi = lane >= 32;
  const int fc = (lane & 31) * 8;
  const bool active = fc < D;
  const uint32_t lanebase = active ? static_cast<uint32_t>(fc) * 2u : 0u;
  auto lane64 = [&](int64_t v, int i) -> int64_t {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffff), i);
    const uint32_t hi32 = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), i);
    return static_cast<int64_t>((static_cast<uint64_t>(hi32) << 32) | lo);
  };
  const int64_t e_first = P.rem_rowptr[r_base];
  const int64_t end_abs = P.rem_rowptr[r_base + 1 + (lane < nr ? lane : nr - 1)];
  // (the row pointers above are requested before the staged list: one round trip for both)
  for (int i = threadIdx.x; i < S; i += NW * 64) cols_lds[i] = P.sh_cols[s0 + i];
  __syncthreads();
  const int64_t begin_abs = __shfl_up(end_abs, 1, 64);
  const int64_t len_i = end_abs - (lane == 0 ? e_first : begin_abs);       // (0 for lanes >= nr)
  // Rows longer than long_len (hubs) are reduced by whole workgroups afterwards (the long-row queue of sgf_spmm_split);
  // here they count as rows WITHOUT gathered entries: stream positions are numbered over the other rows only, and the
  // loads of the stash skip the hubs' entries.  (A first version walked such a wave's rows one entry at a time: on a
  // power-law graph, where rows just below the threshold sit next to the hubs, a few waves then took 3 ms.)
  const bool long_i = lane < nr && len_i > lq.long_len;
  const uint64_t long_mask = __ballot(long_i);
  const int32_t* __restrict__ ci = P.rem_col + e_first;
  const float* __restrict__ va = P.rem_val + e_first;
  int rel_v = long_i ? 0 : static_cast<int>(len_i);        // -> lane i: end of local row i in stream positions
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(rel_v, off, 64);
    if (lane >= off) rel_v += t;
  }
  // entries of hub rows that lie in front of stream position p
  auto hub_skip = [&](int p) -> int64_t {
    int64_t sk = 0;
    for (uint64_t mk = long_mask; mk != 0; mk &= mk - 1) {
      const int j = __builtin_ctzll(mk);
      if (__builtin_amdgcn_readlane(rel_v, j) <= p) sk += lane64(len_i, j);
    }
    return sk;
  };

  // The first kStash stream positions of a half (16 rows): {byte offset of the source row, value}, followed by kStashPad
  // positions of {out-of-range offset, 0} — the unrolled steps of the gather loop read up to 3 batches past the end without
  // a bound test (such a gather returns zeros without touching memory: buffer range check).  Loaded into registers
  // (coalesced) and parked in LDS later, so that the loads' latency is hidden: half 0 behind the first X chunk, half 1
  // behind the rounding + store pass of half 0.
  constexpr int kStashRegs = (kStash + kStashPad) / 64;
  struct StashRegs {
    int32_t col[kStashRegs];
    float val[kStashRegs];
  };
  auto stash_load = [&](int start, int ne, StashRegs& sr) {
    if (long_mask == 0) {
      // range-checked loads (positions >= ne return 0 without touching memory): unconditional, so all of them are in
      // flight before the first is used
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(ci + start), 0, ne * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(va + start), 0, ne * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < kStashRegs; ++i) {
        sr.col[i] = __builtin_amdgcn_raw_buffer_load_b32(rc, (64 * i + lane) * 4, 0, 0);
        sr.val[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, (64 * i + lane) * 4, 0, 0));
      }
    } else {                                               // (rare) a hub among these rows: its entries are stepped over
#pragma unroll
      for (int i = 0; i < kStashRegs; ++i) {
        const int idx = 64 * i + lane;
        const int64_t g = static_cast<int64_t>(start) + idx + hub_skip(start + idx);
        sr.col[i] = idx < ne ? ci[g] : 0;
        sr.val[i] = idx < ne ? va[g] : 0.f;
      }
    }
  };
  auto stash_park = [&](const StashRegs& sr, int ne) {
    __builtin_amdgcn_sched_barrier(0);                     // the first use of the loads stays behind what was issued since
#pragma unroll
    for (int i = 0; i < kStashRegs; ++i) {
      int32_t c = sr.col[i];
      if (dbg & 16) c &= 4095;                             // timing experiment: every gather an L2 hit
      stash_off[64 * i + lane] = 64 * i + lane < ne ? static_cast<uint32_t>(c) * pitch : kOobOffset;
      stash_val[64 * i + lane] = sr.val[i];
    }
  };
  auto half_rows = [&](int h) { return nr - 16 * h < 16 ? nr - 16 * h : 16; };
  // stream positions [start, stop) of half h
  auto half_start = [&](int h) { return h == 0 ? 0 : __builtin_amdgcn_readlane(rel_v, 15); };
  auto half_stop = [&](int h) { return __builtin_amdgcn_readlane(rel_v, 16 * h + half_rows(h) - 1); };

  f32x16 acc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // A fragments of chunk q for this wave's row tile (k-steps 0, 1 x {hi, lo}) live in the wave's own 4 KiB of LDS and are
  // read from there (lane l: 16 bytes at l * 16 of every 1 KiB fragment half).  A dense group is copied there by LDS-DMA;
  // a sparse one arrives as 8-byte entries {offset in the area, hi | lo << 16} in the wave's entry area, ONE chunk
  // earlier, and is scattered over the cleared area (spmm_pack.hip).  Nothing in the loop is loaded into registers from
  // global memory: hipcc's wait counts do not see the DMA, so its wait for a register load would be a wait for every
  // DMA requested before it — the only waits are the explicit ones at the top of the chunk loop.
  const uint32_t frag_lds = smem_lds + static_cast<uint32_t>(kRingBytes + wid * kFragBytes);
  const uint32_t raw_lds = smem_lds + static_cast<uint32_t>(kRawBase + wid * kFragBytes);
  const uint32_t raw0_lds = smem_lds + static_cast<uint32_t>(kRaw0Base + wid * kFragBytes);
  unsigned char* frag = smem + kRingBytes + wid * kFragBytes;
  // lane q: {pool offset, cells} of this wave's group of chunk q (NQ <= 32)
  int32_t grp_off = 0, grp_cnt = 0;
  if (mine && lane < NQ && !(dbg & 1)) {
    const int2 gv = *reinterpret_cast<const int2*>(P.grp + 2 * (g0 + lane));
    grp_off = gv.x;
    grp_cnt = gv.y;
  }
  auto group_cells = [&](int q) { return __builtin_amdgcn_readlane(grp_cnt, q); };
  // n_ops x 1 KiB from the pool into LDS at `dst`
  auto dma_pool = [&](int q, uint32_t dst, int n_ops) {
    const unsigned char* g = reinterpret_cast<const unsigned char*>(
        P.pool + static_cast<uint32_t>(__builtin_amdgcn_readlane(grp_off, q)) + lane);
    for (int i = 0; i < n_ops; ++i) {
      const uint32_t d = dst + static_cast<uint32_t>(i * 1024);
      const unsigned char* gp = g + i * 1024;
      uint32_t keep;
      if (dbg & 512)                                       // timing experiment: the packed tiles as a non-temporal stream
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gp), "s"(d)
            : "memory");
      else
      asm volatile(                                        // (LDS reads of what is being replaced have completed)
          "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(gp), "s"(d)
          : "memory");
    }
  };
  // entries of a sparse group of chunk q -> the entry area at `dst`; dense groups are copied when their chunk is next
  auto fetch_entries = [&](int q, uint32_t dst) {
    const int c = group_cells(q);
    if (c > 0) dma_pool(q, dst, (c * 8 + 1023) >> 10);
  };
  // the fragment area becomes the fragments of chunk q: entries at `raw` (sparse) or a 4 KiB copy (dense)
  auto build_a = [&](int q, const unsigned char* raw) {
    const int c = group_cells(q);
    if (c < 0) {
      dma_pool(q, frag_lds, 4);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(frag + i * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
    for (int r = 0; r < c; r += 64) {
      const uint2 e = *reinterpret_cast<const uint2*>(raw + (r + lane) * 8);
      if (r + lane < c) {
        *reinterpret_cast<uint16_t*>(frag + e.x) = static_cast<uint16_t>(e.y & 0xffffu);
        *reinterpret_cast<uint16_t*>(frag + e.x + 1024) = static_cast<uint16_t>(e.y >> 16);
      }
    }
  };
  auto read_a = [&](uint4 (&a)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4*>(frag + i * 1024 + lane * 16);
  };
  // B fragment address of this lane inside a unit: the 16-lane groups read one [4 keys][16 columns] subtile each
  const uint32_t b_lane = static_cast<uint32_t>(lane * 8);

  const bool tiles_on = NQ > 0 && !(dbg & 1);
  const bool fast = mine;
  {
    // requested together: [stash of half 0] [A of chunk 0: entries or fragments] [entries of chunk 1] [X chunk 0]
    StashRegs sr;
    const int tot0 = (fast && !(dbg & 2)) ? half_stop(0) : 0;
    const int ne0 = tot0 < kStash ? tot0 : kStash;
    if (fast) stash_load(0, ne0, sr);
    if (tiles_on) {
      if (mine) {
        if (group_cells(0) < 0) build_a(0, nullptr);
        else fetch_entries(0, raw0_lds);
        if (NQ > 1) fetch_entries(1, raw_lds);
      }
      stage(0, 0);
    }
    if (fast) stash_park(sr, ne0);
  }
  if (tiles_on) {
    for (int q = 0; q < NQ; ++q) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): everything this wave requested has landed
      __syncthreads();                                   // chunk q is complete and every wave is done with chunk q - 1
      uint4 a_cur[4];
      if (mine) {
        if (q == 0 && group_cells(0) >= 0) build_a(0, smem + kRaw0Base + wid * kFragBytes);
        read_a(a_cur);
      }
      if (q == 0 && kRaw0InRing && NQ > 1) __syncthreads();   // slot 1 held the entries of chunk 0 until here
      if (q + 1 < NQ) {
        if (mine) {
          build_a(q + 1, smem + kRawBase + wid * kFragBytes);
          if (q + 2 < NQ) fetch_entries(q + 2, raw_lds);
        }
        stage(q + 1, (q + 1) & 1);
      }
      if (mine) {
        const unsigned char* slot = smem + (q & 1) * kSlotBytes + b_lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, a_cur[2 * s]);
          const bf16x8 al = __builtin_bit_cast(bf16x8, a_cur[2 * s + 1]);
          if (dbg & 128) continue;                         // timing experiment: staging + fragment loads only
#pragma unroll
          for (int t = 0; t < NCT; ++t) {
            const unsigned char* p = slot + (2 * s * NCT + t) * 512;
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, p));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, p + NCT * 512));
            bf16x8 bb;
            bb[0] = b0[0]; bb[1] = b0[1]; bb[2] = b0[2]; bb[3] = b0[3];
            bb[4] = b1[0]; bb[5] = b1[1]; bb[6] = b1[2]; bb[7] = b1[3];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bb, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bb, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();                                       // the ring is dead: its memory becomes the waves' patches
  if (!mine) return;

  // ---- epilogue: this wave's <= 32 rows ------------------------------------------------------------------------------
  // The entries left on the gather path form ONE stream per wave (rem_* is contiguous over its rows; every row's share is
  // padded to an even count by the plan, so a pair of stream positions never straddles two rows).  Rows of X are fetched
  // two per instruction (lanes 0-31: the row of position 2 k, lanes 32-63: of 2 k + 1; 16 bytes per lane) in batches of
  // kRingPairs loads; TWO batches are in flight per wave (with 128 accumulator registers there are only 8 waves per CU
  // to cover the gathers' latency, so it has to be covered by loads in flight per wave).
  // the tile's partial sums of rows 16 q ..: accumulator register i holds row (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
  auto refill_q = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int kRegs = kPatchRows / 2;                // accumulator registers per column tile and group
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
      for (int i = kRegs * q; i < kRegs * q + kRegs; ++i)
        patch[(((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) & (kPatchRows - 1)) * D + 32 * t + (lane & 31)] = acc[t][i];
  };

  float racc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) racc[k] = 0.f;
  auto fma8 = [&](float v, const u32x4& r) {
    racc[0] = fmaf(v, __uint_as_float(r.x << 16), racc[0]);
    racc[1] = fmaf(v, __uint_as_float(r.x & 0xffff0000u), racc[1]);
    racc[2] = fmaf(v, __uint_as_float(r.y << 16), racc[2]);
    racc[3] = fmaf(v, __uint_as_float(r.y & 0xffff0000u), racc[3]);
    racc[4] = fmaf(v, __uint_as_float(r.z << 16), racc[4]);
    racc[5] = fmaf(v, __uint_as_float(r.z & 0xffff0000u), racc[5]);
    racc[6] = fmaf(v, __uint_as_float(r.w << 16), racc[6]);
    racc[7] = fmaf(v, __uint_as_float(r.w & 0xffff0000u), racc[7]);
  };
  uint16_t* __restrict__ yl = y + r_base * ldy + (active ? fc : 0);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs =
      __builtin_amdgcn_make_buffer_rsrc(y + r_base * ldy, 0, static_cast<int>(32 * ldy * 2), 0x00020000);

  for (uint64_t mk = long_mask; mk != 0; mk &= mk - 1) {   // hubs: queued for the workgroup-per-segment kernels
    const int j = __builtin_ctzll(mk);
    push_long_row(lq, r_base + j, lane64(len_i, j), lane, 64);
  }

  // ---- gathers consumed straight from registers -----------------------------------------------------------------------
  // Register sets of B pair loads each: while one set is consumed the others are in flight, and a set is re-issued as soon as
  // its last pair is consumed.  Steps are unrolled, registers and LDS offsets static: per pair one ds_read_b32 of the
  // source offset (for the re-issue), one of the value, the multiply-adds, and a wave-uniform compare for a row end.
  // A row end COMMITS the row — (even + odd positions) added into the row's slot of the patch, which already holds the
  // tile's partial sum: ~35 instructions, no vector-memory operation, so the unrolled steps stay small (the first
  // version inlined rounding, the store and the patch refill at every step: 100-390 KiB of code).  Rounding to bf16 and
  // the stores are a compact pass over the patch afterwards, 16 rows at a time: two halves per wave.
  constexpr int B = kRingPairs;
  auto run_half = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
    const int nrh = half_rows(h);
    const int start = half_start(h);                       // stream positions of this half
    const int tot = (dbg & 2) ? 0 : half_stop(h) - start;
    int row = 16 * h;                                      // first row of the half; rows without entries are skipped
    int row_end = __builtin_amdgcn_readlane(rel_v, row) - start;
    auto commit = [&](int pos) {                           // the row that ends at `pos` (relative to the half)
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = sum_halves(racc[k]);
      if (active && !hi) {
        float* pr = patch + (row & 15) * D + fc;
        float4 p0 = *reinterpret_cast<const float4*>(pr), p1 = *reinterpret_cast<const float4*>(pr + 4);
        p0.x += t[0]; p0.y += t[1]; p0.z += t[2]; p0.w += t[3];
        p1.x += t[4]; p1.y += t[5]; p1.z += t[6]; p1.w += t[7];
        *reinterpret_cast<float4*>(pr) = p0;
        *reinterpret_cast<float4*>(pr + 4) = p1;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) racc[k] = 0.f;
      do {
        ++row;
        row_end = row < 16 * h + nrh ? __builtin_amdgcn_readlane(rel_v, row) - start : 0x7fffffff;
      } while (row_end == pos);
    };
    while (row_end == 0) {                                 // leading rows without entries
      ++row;
      row_end = row < 16 * h + nrh ? __builtin_amdgcn_readlane(rel_v, row) - start : 0x7fffffff;
    }
    // pairs b0 .. b0 + B - 1 of the epoch: per pair one LDS read, one add, one load
    auto issue = [&](int b0, u32x4 (&rg)[B]) {
      const uint32_t* so = stash_off + 2 * b0 + half;
#pragma unroll
      for (int j = 0; j < B; ++j) {
        if (dbg & 4096) {                                  // traffic experiment: far sources (> 16384 rows away) non-temporal
          const uint32_t o0 = __builtin_amdgcn_readfirstlane(stash_off[2 * (b0 + j)]);
          const uint32_t here = static_cast<uint32_t>(row0) * pitch;
          const uint32_t dist = o0 > here ? o0 - here : here - o0;
          if (dist > 16384u * pitch)
            rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(so[2 * j] + lanebase), 0, 2);
          else
            rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(so[2 * j] + lanebase), 0, 0);
        } else
        if (dbg & 1024)                                    // timing experiment: every gather a non-temporal load
          rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(so[2 * j] + lanebase), 0, 2);
        else if (dbg & 2048)                               // ... an sc1 load
          rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(so[2 * j] + lanebase), 0, 16);
        else
          rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(so[2 * j] + lanebase), 0, 0);
      }
    };
    // THREE register sets of B pair loads: the fabric answers a gather in ~1.3 us under load while a set is consumed in
    // ~0.45 us, so with two sets (16 KiB in flight per wave, 8 waves per CU) the loop ran at the memory latency:
    // ~2 sets per (latency + one consumption).  The third set is what the register file has left at two waves per SIMD.
    u32x4 ra[B], rb[B], rc[B];
    // epoch 0 is parked already: its first batches are requested before the patch is written (64 LDS stores per lane)
    if (tot > 0) {
      issue(0, ra);
      issue(B, rb);
    }
    refill_q(hc);
    for (int base = 0; base < tot; base += kStash) {
      const int ne = tot - base < kStash ? tot - base : kStash;     // stream positions of this epoch (even)
      const int np = ne >> 1;
      if (base > 0) {                                      // (rare: more than kStash positions in 16 rows)
        StashRegs sr;
        stash_load(start + base, ne, sr);
        stash_park(sr, ne);
        issue(0, ra);
        issue(B, rb);
      }
      // distance (in stream positions) from the end of pair b0's first position to the current row's end
      // (a row end beyond this epoch must not be matched by the padding positions)
      auto rel_end = [&]() -> int { return row_end - base <= ne ? row_end - base : 0x3fffffff; };
      int to_end = rel_end();
      auto consume = [&](int b0, const u32x4 (&rg)[B]) {
        const float* sv = stash_val + 2 * b0 + half;
        float vv[B];
#pragma unroll
        for (int j = 0; j < B; ++j) vv[j] = sv[2 * j];
        int d0 = to_end - 2 * b0;
#pragma unroll
        for (int j = 0; j < B; ++j) {
          if (!(dbg & 32)) fma8(vv[j], rg[j]);  // (dbg & 32: timing experiment without the multiply-adds)
          if (d0 == 2 * j + 2) {                           // wave-uniform: this pair closes its row
            commit(row_end);
            to_end = rel_end();
            d0 = to_end - 2 * b0;
          }
        }
      };
      // Sets A and B cross the loop's back edge in flight, A the older one — the same queue as on entry, so hipcc's
      // wait-count pass keeps exact counts at the loop header (a queue that differs between entry and back edge makes
      // it wait for everything there).  The steps read up to 5 B pairs past the epoch's end: kStashPad.
      __builtin_amdgcn_sched_barrier(0);
      for (int b0 = 0; b0 < np; b0 += 3 * B) {
        issue(b0 + 2 * B, rc);
        __builtin_amdgcn_sched_barrier(0);
        consume(b0, ra);
        __builtin_amdgcn_sched_barrier(0);
        issue(b0 + 3 * B, ra);
        __builtin_amdgcn_sched_barrier(0);
        consume(b0 + B, rb);
        __builtin_amdgcn_sched_barrier(0);
        issue(b0 + 4 * B, rb);
        __builtin_amdgcn_sched_barrier(0);
        consume(b0 + 2 * B, rc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next half's first epoch: requested now, parked after the stores below
    StashRegs nx;
    const bool more = h == 0 && nr > 16;
    int ne1 = 0;
    if (more) {
      const int s1 = half_start(1);
      const int t1 = (dbg & 2) ? 0 : half_stop(1) - s1;
      ne1 = t1 < kStash ? t1 : kStash;
      stash_load(s1, ne1, nx);
      __builtin_amdgcn_sched_barrier(0);
    }
    // rounding + store of the half's rows (a row without gathered entries holds the tile's partial sum as it is)
    for (int lr = 0; lr < nrh; ++lr) {
      if (active && !hi) {
        const float* pr = patch + lr * D + fc;
        const float4 p0 = *reinterpret_cast<const float4*>(pr), p1 = *reinterpret_cast<const float4*>(pr + 4);
        uint4 o;
        o.x = pack_bf16(p0.x, p0.y);
        o.y = pack_bf16(p0.z, p0.w);
        o.z = pack_bf16(p1.x, p1.y);
        o.w = pack_bf16(p1.z, p1.w);
        if (dbg & 8) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(yl + (16 * h + lr) * ldy));
        else if (dbg & 256)                                // timing experiment: write-through stores, the line leaves the L2
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs,
                                                 static_cast<int>(((16 * h + lr) * ldy + fc) * 2), 0, 16);
        else *reinterpret_cast<uint4*>(yl + (16 * h + lr) * ldy) = o;
      }
    }
    if (more) stash_park(nx, ne1);
  };
  run_half(std::integral_constant<int, 0>{});
  if (nr > 16) run_half(std::integral_constant<int, 1>{});
}

}  // namespace
}  // namespace sgf

using namespace sgf;

extern "C" int sgf_spmm_tile_supported(int32_t d, int32_t dtype) {
  return dtype == SGF_BF16 && (d == 256 || d == 128) ? 1 : 0;
}

extern "C" int sgf_spmm_tile(const int32_t* blk_row, int64_t nb, int32_t block_rows, const int32_t* sh_ptr, const int32_t* sh_cols,
                             const int64_t* tile_ptr, const int32_t* grp, const void* pool, const int64_t* rem_rowptr,
                             const int32_t* rem_col, const float* rem_val, const void* x, int64_t ldx, int64_t n_cols,
                             void* y, int64_t ldy, int64_t n_rows, int32_t d, int32_t dtype, int64_t long_len,
                             int64_t long_segments, void* workspace, size_t workspace_bytes, void* stream) {
  const char* fn = "sgf_spmm_tile";
  SGF_REQUIRE(n_rows >= 0 && nb >= 0 && n_cols >= 0, SGF_E_INVALID, "%s: negative size", fn);
  if (n_rows == 0 || nb == 0) return SGF_OK;
  SGF_REQUIRE(block_rows >= 1 && block_rows <= 256, SGF_E_INVALID, "%s: block_rows outside [1, 256]", fn);
  SGF_REQUIRE(sgf_spmm_tile_supported(d, dtype), SGF_E_UNSUPPORTED, "%s: bf16 storage with d = 128 or 256 only (d=%d dtype=%d)",
              fn, d, dtype);
  SGF_REQUIRE(blk_row && sh_ptr && sh_cols && tile_ptr && grp && pool && rem_rowptr && x && y, SGF_E_INVALID, "%s: null pointer", fn);
  SGF_REQUIRE(reinterpret_cast<uintptr_t>(pool) % 16 == 0 && reinterpret_cast<uintptr_t>(grp) % 8 == 0, SGF_E_INVALID,
              "%s: pool / grp alignment", fn);
  SGF_REQUIRE(nb < (static_cast<int64_t>(1) << 31), SGF_E_UNSUPPORTED, "%s: too many blocks", fn);
  SGF_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= d && ldy >= d && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(y) % 16 == 0,
              SGF_E_INVALID, "%s: x / y rows must be 16-byte aligned (ld %% 8 == 0)", fn);
  const uint64_t x_bytes = static_cast<uint64_t>(n_cols) * static_cast<uint64_t>(ldx) * 2;
  SGF_REQUIRE(x_bytes < (static_cast<uint64_t>(1) << 32) - 2048, SGF_E_UNSUPPORTED, "%s: x beyond 4 GiB (32-bit gather offsets)", fn);
  SGF_REQUIRE(long_len >= 1 && long_segments >= 0 && long_segments < (static_cast<int64_t>(1) << 31), SGF_E_INVALID,
              "%s: bad long_len / long_segments", fn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  LongQueue lq{nullptr, nullptr, 0, INT64_MAX};
  float* partial = nullptr;
  if (long_segments > 0) {
    SGF_REQUIRE(workspace && workspace_bytes >= sgf_spmm_split_workspace_bytes(long_segments, d), SGF_E_WORKSPACE,
                "%s: workspace too small", fn);
    char* ws = static_cast<char*>(workspace);
    lq.count = reinterpret_cast<int32_t*>(ws);
    lq.entries = reinterpret_cast<LongEntry*>(ws + 256);
    lq.cap = static_cast<int32_t>(long_segments);
    lq.long_len = long_len;
    partial = reinterpret_cast<float*>(ws + 256 + align_up(static_cast<size_t>(long_segments) * sizeof(LongEntry), 256));
    SGF_CHECK_HIP(hipMemsetAsync(lq.count, 0, sizeof(int32_t), st));
  }
  TilePlanArgs P{blk_row, sh_ptr, sh_cols, tile_ptr, grp, static_cast<const uint4*>(pool), rem_rowptr, rem_col, rem_val};
  // SGF_SPMM_TILE_DEBUG (timing experiments only, results are then wrong): 1 = skip the tile phase, 2 = skip the gathers,
  // 4 / 512 = nt staged rows / packed tiles, 8 / 256 = nt / write-through (sc1) y stores, 16 = gathers clamped to 4096 rows (L2 hits), 32 = no multiply-adds in the gather
  // loop, 128 = no matrix-core work in the tile phase
  // (both switches are cached: a launch costs no environment look-up; sgf_reload_env() re-reads them)
  static EnvInt chunk_env{"SGF_SPMM_TILE_CHUNK", 64};
#ifdef SGF_PROBES   // (make PROBES=1; the release library takes no debug mask and does not contain the masked kernels)
  static EnvInt dbg_env{"SGF_SPMM_TILE_DEBUG", 0};
  const int dbg = dbg_env.get();
#else
  const int dbg = 0;
#endif
  const int chunk = chunk_env.get() > 0 ? chunk_env.get() : 64;   // an XCD walks 64 consecutive blocks (<= 8192 rows) at a time
  const uint16_t* xs = static_cast<const uint16_t*>(x);
  uint16_t* ys = static_cast<uint16_t*>(y);
#define SGF_TILE_LAUNCH(NCT_, NW_, DBG_)                                                                               \
  hipLaunchKernelGGL((k_spmm_tile_bf16<NCT_, NW_, DBG_>), dim3(static_cast<unsigned>(nb)), dim3(NW_ * 64), 0, st, P, xs, \
                     static_cast<uint32_t>(ldx * 2), static_cast<uint32_t>(x_bytes), ys, ldy, static_cast<int32_t>(nb), \
                     chunk, lq, dbg)
#ifdef SGF_PROBES
#define SGF_TILE_LAUNCH2(NCT_, NW_)                                               \
  do {                                                                            \
    if (dbg) SGF_TILE_LAUNCH(NCT_, NW_, true); else SGF_TILE_LAUNCH(NCT_, NW_, false); \
  } while (0)
#else
#define SGF_TILE_LAUNCH2(NCT_, NW_) SGF_TILE_LAUNCH(NCT_, NW_, false)
#endif
  if (d == 256) {
    if (block_rows > 128) SGF_TILE_LAUNCH2(8, 8); else SGF_TILE_LAUNCH2(8, 4);
  } else {
    if (block_rows > 128) SGF_TILE_LAUNCH2(4, 8); else SGF_TILE_LAUNCH2(4, 4);
  }
#undef SGF_TILE_LAUNCH2
#undef SGF_TILE_LAUNCH
  SGF_LAUNCH_CHECK();
  return spmm_long_rows(dtype, rem_rowptr, rem_col, rem_val, x, ldx, d, lq, partial, y, ldy, st);
}
