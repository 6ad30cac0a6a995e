// gramx.hip — node reductions  C[m x k] = A^T B  over the rows of two bf16 operands ([n, m], [n, k], m, k <= 256) with the
// row tiles streamed by LDS-DMA and the matrix-core fragments taken by transposing LDS reads.
//
// What it replaces and why (profiles/r05_products_bf16_kernel_roofline.md): k_reduce_bf16 (attn.hip) stages a tile through
// REGISTERS (8-byte loads, a 4 x 4 register transpose, ds_write_b64) and its 256 x 256 fp32 accumulator takes half of the
// CU's register file, so one staging set is all it can hold: no load is in flight while a tile is transposed and committed,
// and with every CU doing the same the tile's 64 KiB take 2.7 us to arrive — 4.7 us per tile against 2.7 at the copy rate
// (0.54 of it; dW of every Linear, G = h^T h, the attention backward reduce: 6 + 1 launches per training step).  Here the
// tiles never touch a register on their way in:
//   * stage = 32 rows of A and of B; a ring of 4 stages (128 KiB of LDS) is filled by global_load_lds_dwordx4, three stages
//     (96 KiB per CU) in flight while the fourth is multiplied; the only waits are COUNTED s_waitcnt vmcnt(N) (the DMA is
//     issued from inline assembly: invisible to hipcc's wait counts, cdna_hip_programming.md §5.7);
//   * the LDS image of an operand is the one ds_read_b64_tr_b16 serves without bank conflicts (as csrc/spmm_tile.hip):
//     512-byte units [k-step s][read r][column tile t] of four [4 rows][16 columns] subtiles, lane l of a read at l * 8;
//     a DMA piece (1 KiB = the units (s, r, 2 p), (s, r, 2 p + 1)) is 8 rows x 128 contiguous bytes of the operand;
//   * 8 waves, wave (wm, wd) owns the 64 x 128 block of C: per k-step 12 transposing reads and 8 v_mfma_f32_32x32x16_bf16;
//   * column sums (the bias gradients) and the row-scaled operand of the attention backward are formed by a VALU pass over
//     the landed stage (16-byte linear LDS reads: a thread's 8 columns are the same in every stage).
// Results are per-block partials in the layout of reduce_shared.h, added in a fixed order by attn.hip's finalize kernels:
// deterministic, no atomics.  Products of bf16 values are exact in fp32 and accumulation is fp32: against k_reduce_bf16
// only the summation order differs.
#include "reduce_shared.h"

namespace sgf {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define SGF_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int kGxThreads = 512;                    // 8 waves = 2 per SIMD: 256 VGPRs each
constexpr int kGxRows = 32;                        // rows per stage (two MFMA k-steps)
constexpr int kGxStages = 4;                       // ring depth
constexpr int kGxOpBytes = kGxRows * 512;          // one operand's image of a stage: 32 rows x 256 columns = 16 KiB
constexpr int kGxStageBytes = 2 * kGxOpBytes;
constexpr int kGxRingBytes = kGxStages * kGxStageBytes;
constexpr int kGxScalBytes = kGxRows * 8;          // per-row scalars of a stage (float2)

constexpr int kGxGram = 0;    // C = A^T B, column sums of A; paired launch: roles share A
constexpr int kGxBwdHS = 1;   // A = h, B = g / den (formed in LDS); colsum = sum h dden, vecB = sum g / den, scalar = sum dden

struct GramxArgs {
  const void* a;
  const void* b;
  const void* b2;
  int64_t lda, ldb, ldb2;   // elements
  int64_t n;
  int32_t m, k;             // valid columns of a / b
  int32_t pair;
  int32_t same;             // b is a (G = h^T h): one image serves both operands
  const float* rowscal;     // kGxBwdHS: float2 per row
  float* partial;
};

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// two fp32 -> packed bf16, round to nearest even: v_cvt_pk_bf16_f32 (the same rounding as common.h's f32_to_bf16)
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32v2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2));
}
// sum over the 32 lanes of this lane's half of the wave, the same value in all of them, on the VALU alone (four DPP butterfly
// steps inside the rows of 16, then one v_permlane16_swap): no LDS round trips (a __shfl_xor chain is five dependent
// ds_bpermute, and with 2 waves per SIMD nothing hides them)
__device__ __forceinline__ float half_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf,
                                                                 false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  const uint32_t u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // r[0] = rows {0, 0, 2, 2}, r[1] = rows {1, 1, 3, 3}
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 64 lanes x 16 bytes -> LDS [dst, dst + 1024), lane l at l * 16 (dst wave-uniform).  M0 is written and restored in the
// statement that uses it (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void dma16(const unsigned char* gp, uint32_t dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gp), "s"(dst)
               : "memory");
}
// 64 lanes x 4 bytes -> LDS [dst, dst + 256)
__device__ __forceinline__ void dma4(const unsigned char* gp, uint32_t dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gp), "s"(dst)
               : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform n that is only known at run time (the immediate has to be a constant)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int MODE>
__global__ __launch_bounds__(kGxThreads) void k_gramx(GramxArgs p) {
  // ONE LDS object (a second one makes hipcc wait vmcnt(0) in front of LDS reads, cdna_hip_programming.md §5): the ring,
  // then the per-row scalars of the four stages
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kGxRingBytes + kGxStages * kGxScalBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wd = wave & 1;               // this wave's 64 x 128 block of C

  // paired launch (kGxGram): blocks b and b + 8 (one XCD) walk the same stages, role 1 multiplies with b2
  const bool paired = MODE == kGxGram && p.pair != 0;
  const int role = paired ? (blockIdx.x >> 3) & 1 : 0;
  const int64_t vblock = paired ? (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3) : blockIdx.x;
  const int64_t vgrid = paired ? gridDim.x / 2 : gridDim.x;
  const unsigned char* ga = static_cast<const unsigned char*>(p.a);
  const unsigned char* gb = static_cast<const unsigned char*>(role ? p.b2 : p.b);
  const int64_t pitch_a = p.lda * 2, pitch_b = (role ? p.ldb2 : p.ldb) * 2;

  const int64_t total = (p.n + kGxRows - 1) / kGxRows;
  const int nq = vblock < total ? static_cast<int>((total - vblock + vgrid - 1) / vgrid) : 0;

  // ---- staging: wave w moves pieces 2 w, 2 w + 1 of both operands (piece = (2 s + r) * 4 + p) -----------------------
  const int row_l = 16 * (wave >> 2) + 4 * ((wave >> 1) & 1) + 8 * ((lane >> 4) & 1) + ((lane >> 1) & 3);
  const int col_l = 32 * (lane >> 5) + 16 * ((lane >> 3) & 1) + 8 * (lane & 1);
  int64_t off_a[2], off_b[2];                            // this lane's byte offset inside a stage's rows
  int col_a[2], col_b[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = 64 * (2 * (wave & 1) + j) + col_l;
    col_a[j] = c + 8 <= p.m ? c : p.m - 8;               // columns past the operand: a valid address, the product is dropped
    col_b[j] = c + 8 <= p.k ? c : p.k - 8;
    off_a[j] = row_l * pitch_a + col_a[j] * 2;
    off_b[j] = row_l * pitch_b + col_b[j] * 2;
  }
  const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(SGF_LDS(unsigned char, smem)));
  const bool same = MODE == kGxGram && p.same != 0;
  const int nps = same ? 2 : ((MODE == kGxBwdHS && wave == 0) ? 5 : 4);   // DMA instructions of this wave per stage

  auto issue = [&](int q) {
    const int64_t row0 = (vblock + static_cast<int64_t>(q) * vgrid) * kGxRows;
    const uint32_t dst = smem_lds + static_cast<uint32_t>((q & (kGxStages - 1)) * kGxStageBytes + 2 * wave * 1024);
    if (row0 + kGxRows <= p.n) {
      const unsigned char* sa = ga + row0 * pitch_a;
      const unsigned char* sb = gb + row0 * pitch_b;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dma16(sa + off_a[j], dst + j * 1024);
        if (!same) dma16(sb + off_b[j], dst + kGxOpBytes + j * 1024);
      }
    } else {                                             // the tensor's last, ragged stage: rows past the end read row n - 1
      int64_t r = row0 + row_l;                          // (their LDS image is cleared before it is used)
      r = r < p.n ? r : p.n - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dma16(ga + r * pitch_a + col_a[j] * 2, dst + j * 1024);
        if (!same) dma16(gb + r * pitch_b + col_b[j] * 2, dst + kGxOpBytes + j * 1024);
      }
    }
    if (MODE == kGxBwdHS && wave == 0) {                 // 32 rows x (1 / den, dden) = 64 dwords
      int64_t e = row0 * 2 + lane;
      e = e < 2 * p.n ? e : 2 * p.n - 1;
      dma4(reinterpret_cast<const unsigned char*>(p.rowscal + e),
           smem_lds + static_cast<uint32_t>(kGxRingBytes + (q & (kGxStages - 1)) * kGxScalBytes));
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float csa[8], csb[8];                                  // this thread's 8 columns: sums over the rows it sees
#pragma unroll
  for (int e = 0; e < 8; ++e) csa[e] = csb[e] = 0.f;
  float sden = 0.f;

  // column tiles of this wave that lie inside the product (narrow operands: the head's dW has m = 48, the stems' k = 100)
  const int nta = p.m - 64 * wm <= 0 ? 0 : (p.m - 64 * wm > 32 ? 2 : 1);
  const int ntb = p.k - 128 * wd <= 0 ? 0 : (p.k - 128 * wd + 31) / 32 > 4 ? 4 : (p.k - 128 * wd + 31) / 32;
  const bool act = nta > 0 && ntb > 0;

  // VALU pass: thread tid owns the 16-byte slots tid and tid + 512 of an operand image (k-steps 0 and 1)
  const int vrow = 8 * ((tid >> 4) & 1) + 4 * ((tid >> 8) & 1) + ((tid >> 1) & 3);   // + 16 s
  const int vcol = 32 * ((tid >> 5) & 7) + 16 * ((tid >> 3) & 1) + 8 * (tid & 1);

#pragma unroll 1
  for (int q = 0; q < 3 && q < nq; ++q) issue(q);

#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const int later = nq - 1 - q < 2 ? nq - 1 - q : 2;   // stages requested after this one and still allowed in flight
    wait_vm(nps * later);
    __syncthreads();                                     // stage q has landed for every wave; every wave is done with q - 1
    if (q + 3 < nq) issue(q + 3);                        // into the slot of stage q - 1
    const int slot = q & (kGxStages - 1);
    unsigned char* sa = smem + slot * kGxStageBytes;
    unsigned char* sb = same ? sa : sa + kGxOpBytes;
    float2* scal = reinterpret_cast<float2*>(smem + kGxRingBytes + slot * kGxScalBytes);
    const int64_t row0 = (vblock + static_cast<int64_t>(q) * vgrid) * kGxRows;
    if (row0 + kGxRows > p.n) {                          // ragged: clear what lies past the end (wave-uniform, once per launch)
      const int valid = static_cast<int>(p.n - row0);
      if (MODE == kGxBwdHS) {
        if (tid < kGxRows && tid >= valid) scal[tid] = make_float2(0.f, 0.f);
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
          if (16 * s + vrow >= valid) {
            *reinterpret_cast<uint4*>(sa + (tid + 512 * s) * 16) = make_uint4(0u, 0u, 0u, 0u);
            if (!same) *reinterpret_cast<uint4*>(sb + (tid + 512 * s) * 16) = make_uint4(0u, 0u, 0u, 0u);
          }
      }
      __syncthreads();
    }

    // ---- VALU pass over the landed stage ----
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint4 va = *reinterpret_cast<const uint4*>(sa + (tid + 512 * s) * 16);
      const uint32_t ua[4] = {va.x, va.y, va.z, va.w};
      if (MODE == kGxGram) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          csa[2 * e] += bf_lo(ua[e]);
          csa[2 * e + 1] += bf_hi(ua[e]);
        }
      } else {
        const float2 rs = scal[16 * s + vrow];           // (1 / den, dden) of this slot's row
        uint4* pb = reinterpret_cast<uint4*>(sb + (tid + 512 * s) * 16);
        const uint4 vb = *pb;
        const uint32_t ub[4] = {vb.x, vb.y, vb.z, vb.w};
        uint32_t ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          csa[2 * e] += bf_lo(ua[e]) * rs.y;
          csa[2 * e + 1] += bf_hi(ua[e]) * rs.y;
          const float g0 = bf_lo(ub[e]) * rs.x, g1 = bf_hi(ub[e]) * rs.x;
          csb[2 * e] += g0;
          csb[2 * e + 1] += g1;
          ob[e] = pack_bf16(g0, g1);                     // dnum = g / den, re-rounded to bf16 for the matrix cores
        }
        *pb = make_uint4(ob[0], ob[1], ob[2], ob[3]);
        if (vcol == 0) sden += rs.y;
      }
    }
    if (MODE == kGxBwdHS) __syncthreads();               // the scaled B image is complete

    // ---- matrix cores: two k-steps of 16 rows ----
    if (act) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const unsigned char* u = sa + ((2 * s) * 8 + 2 * wm + tm) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 8 * 512));
          af[tm][0] = r0[0]; af[tm][1] = r0[1]; af[tm][2] = r0[2]; af[tm][3] = r0[3];
          af[tm][4] = r1[0]; af[tm][5] = r1[1]; af[tm][6] = r1[2]; af[tm][7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          const unsigned char* u = sb + ((2 * s) * 8 + 4 * wd + tn) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 8 * 512));
          bfr[tn][0] = r0[0]; bfr[tn][1] = r0[1]; bfr[tn][2] = r0[2]; bfr[tn][3] = r0[3];
          bfr[tn][4] = r1[0]; bfr[tn][5] = r1[1]; bfr[tn][6] = r1[2]; bfr[tn][7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          if (tn < ntb) {
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[tn], acc[0][tn], 0, 0, 0);
            if (nta > 1) acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[tn], acc[1][tn], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- this block's partial: [m][dd] (DP = 256, one row group) | column sums | scalars ---------------------------------
  float* part = p.partial + (role * vgrid + vblock) * kRedPartialStride;
  if (act) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = 64 * wm + 32 * tm + mfma32_row(r, lane);
          const int dd = 128 * wd + 32 * tn + (lane & 31);
          part[mm * 256 + dd] = acc[tm][tn][r];
        }
  }
  // column sums: the 16 threads that share vcol differ in the rows they saw
  __syncthreads();
  float* fl = reinterpret_cast<float*>(smem);
  const int ridx = ((tid >> 1) & 3) | (((tid >> 4) & 1) << 2) | (((tid >> 8) & 1) << 3);
  *reinterpret_cast<float4*>(&fl[ridx * 256 + vcol]) = make_float4(csa[0], csa[1], csa[2], csa[3]);
  *reinterpret_cast<float4*>(&fl[ridx * 256 + vcol + 4]) = make_float4(csa[4], csa[5], csa[6], csa[7]);
  if (MODE == kGxBwdHS) {
    *reinterpret_cast<float4*>(&fl[4096 + ridx * 256 + vcol]) = make_float4(csb[0], csb[1], csb[2], csb[3]);
    *reinterpret_cast<float4*>(&fl[4096 + ridx * 256 + vcol + 4]) = make_float4(csb[4], csb[5], csb[6], csb[7]);
    if (vcol == 0) fl[8192 + ridx] = sden;
  }
  __syncthreads();
  if (tid < 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += fl[r * 256 + tid];
    part[kRedTileElems + tid] = s;
    if (MODE == kGxBwdHS) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += fl[4096 + r * 256 + tid];
      part[kRedVecB + tid] = t;
    }
  }
  if (tid == 0) {
    float s = 0.f;
    if (MODE == kGxBwdHS)
      for (int r = 0; r < 16; ++r) s += fl[8192 + r];
    part[kRedTileElems + 256] = s;
    part[kRedTileElems + 257] = 0.f;
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// k_gramt: the Gram whose A operand is FORMED from streamed tensors — the stems' dW / db without a materialised input
// gradient (sgf_gram_bn_bwd: A = BatchNorm'(relu'(g1 + g2)) from g1, g2, z; sgf_gram_ln_bwd: A = LayerNorm'(relu'(g)) from g,
// the LayerNorm's input and its saved row statistics), B = x [n, k <= 128] (the features).
//   * the raw streams land ROW-MAJOR (a DMA piece = two whole 512-byte rows: wave w requests rows 4 w .. 4 w + 3 of every
//     stage and is the wave that transforms them: half a wave per row, lane = a group of 8 columns, so a LayerNorm's row sums
//     are 5 shuffles inside the half-wave and the per-column coefficients of this lane sit in registers for the whole kernel);
//   * the formed operand is rounded to bf16 once (the tensor sgf_bn_bwd_apply / sgf_ln_bwd would have written) and stored into
//     a separate A image in the transposing-read layout (16-byte scattered stores, 4-way conflicted: 2 per thread and stage);
//   * x arrives in the transposing-read layout, 128 columns wide (one piece per wave and stage);
//   * wave w owns the 32 x 128 block of C = A^T x below A's columns [32 w, 32 w + 32): one A fragment, four B fragments and
//     four MFMAs per k-step.
// Ring: 2 stages of 56 KiB (three raw streams) or 3 stages of 40 KiB (two).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kGtBN = 0, kGtLN = 1;
constexpr int kGtBBytes = kGxRows * 256;           // x image: 32 rows x 128 columns = 8 KiB

struct GramtArgs {
  const void* r0;   // BN: g1      LN: g
  const void* r1;   // BN: g2 (or null)
  const void* r2;   // BN: z       LN: the LayerNorm's input
  const void* b;    // x
  int64_t ld0, ld1, ld2, ldb;
  int64_t n;
  int32_t m, k;
  const float *mean, *rstd;      // BN: per column   LN: per row
  const float *gamma, *beta;     // per column (null: 1 / 0)
  const float* stats;            // BN, training: [sum dz' | sum dz' xhat] over the rows
  float inv_n;
  int32_t training, relu;
  float* partial;
};

template <int MODE>
__global__ __launch_bounds__(kGxThreads) void k_gramt(GramtArgs p) {
  constexpr int NR = MODE == kGtBN ? 3 : 2;                        // raw streams
  constexpr int NST = MODE == kGtBN ? 2 : 3;                       // ring depth
  constexpr int kStage = NR * kGxOpBytes + kGtBBytes;
  constexpr int kRing = NST * kStage;
  constexpr int kAimg = kRing;                                     // the formed operand: one image of 16 KiB
  constexpr int kScal = kAimg + kGxOpBytes;                        // LN: [mean 64 | rstd 64] floats per stage
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kScal + NST * 512];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = lane & 31;                                        // this lane's columns: 8 cg .. 8 cg + 7
  const int hw = lane >> 5;
  const bool col_ok = 8 * cg < p.m;

  const int64_t total = (p.n + kGxRows - 1) / kGxRows;
  const int nq = blockIdx.x < total ? static_cast<int>((total - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
  const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(SGF_LDS(unsigned char, smem)));
  const bool has1 = MODE == kGtBN && p.r1 != nullptr;

  // ---- staging ----
  const unsigned char* gr[3] = {static_cast<const unsigned char*>(p.r0), static_cast<const unsigned char*>(p.r1),
                                static_cast<const unsigned char*>(p.r2)};
  const int64_t pitch[3] = {p.ld0 * 2, p.ld1 * 2, p.ld2 * 2};
  const int64_t pitch_b = p.ldb * 2;
  const int rcol = (col_ok ? 8 * cg : p.m - 8) * 2;                // byte offset of this lane's columns in a raw row
  // x piece of this wave: (2 s + r) = wave >> 1, p = wave & 1
  const int brow = 16 * (wave >> 2) + 4 * ((wave >> 1) & 1) + 8 * ((lane >> 4) & 1) + ((lane >> 1) & 3);
  int bcol = 64 * (wave & 1) + 32 * (lane >> 5) + 16 * ((lane >> 3) & 1) + 8 * (lane & 1);
  bcol = bcol + 8 <= p.k ? bcol : p.k - 8;
  const int nps = (MODE == kGtBN ? (has1 ? 7 : 5) : 5) + ((MODE == kGtLN && wave < 2) ? 1 : 0);

  auto issue = [&](int q) {
    const int64_t row0 = (blockIdx.x + static_cast<int64_t>(q) * gridDim.x) * kGxRows;
    const uint32_t base = smem_lds + static_cast<uint32_t>((q % NST) * kStage);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t r = row0 + 4 * wave + 2 * j + hw;
      r = r < p.n ? r : p.n - 1;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int img = MODE == kGtBN ? i : (i == 0 ? 0 : 1);      // LN: r0 -> image 0, r2 -> image 1
        if (MODE == kGtLN && i == 1) continue;
        if (MODE == kGtBN && i == 1 && !has1) continue;
        dma16(gr[i] + r * pitch[i] + rcol, base + img * kGxOpBytes + (2 * wave + j) * 1024);
      }
    }
    {
      int64_t r = row0 + brow;
      r = r < p.n ? r : p.n - 1;
      dma16(static_cast<const unsigned char*>(p.b) + r * pitch_b + bcol * 2, base + NR * kGxOpBytes + wave * 1024);
    }
    if (MODE == kGtLN && wave < 2) {                               // 64 row statistics (the stage's 32 and the next 32)
      int64_t r = row0 + lane;
      r = r < p.n ? r : p.n - 1;
      dma4(reinterpret_cast<const unsigned char*>((wave == 0 ? p.mean : p.rstd) + r),
           smem_lds + static_cast<uint32_t>(kScal + (q % NST) * 512 + wave * 256));
    }
  };

  // ---- this lane's per-column coefficients ----
  float bmu[8], brs[8], bga[8], bbe[8], bk0[8], bk1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * cg + e;
    const bool ok = c < p.m;
    bga[e] = ok ? (p.gamma ? p.gamma[c] : 1.f) : 0.f;
    bbe[e] = ok ? (p.beta ? p.beta[c] : 0.f) : 0.f;
    if (MODE == kGtBN) {
      bmu[e] = ok ? p.mean[c] : 0.f;
      brs[e] = ok ? p.rstd[c] : 0.f;
      bk0[e] = (ok && p.training) ? p.stats[c] * p.inv_n : 0.f;
      bk1[e] = (ok && p.training) ? p.stats[p.m + c] * p.inv_n : 0.f;
    } else {
      bmu[e] = brs[e] = bk0[e] = bk1[e] = 0.f;
    }
  }
  // the coefficient loads are CONSUMED here: hipcc places its wait for a load at the first use, and a first use inside the
  // stage loop would be a vmcnt(0) there — every DMA in flight waited for, every stage
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    asm volatile("" : "+v"(bga[e]), "+v"(bbe[e]));
    if (MODE == kGtBN) asm volatile("" : "+v"(bmu[e]), "+v"(brs[e]), "+v"(bk0[e]), "+v"(bk1[e]));
  }

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float csa[8], csb[8], csc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csa[e] = csb[e] = csc[e] = 0.f;
  const float inv_d = 1.0f / static_cast<float>(p.m);
  const int ntb = (p.k + 31) / 32 > 4 ? 4 : (p.k + 31) / 32;
  const bool act = 32 * wave < p.m;

#pragma unroll 1
  for (int q = 0; q < NST - 1 && q < nq; ++q) issue(q);

#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    const int later = nq - 1 - q < NST - 2 ? nq - 1 - q : NST - 2;
    wait_vm(nps * later);
    __syncthreads();                                     // stage q has landed; every wave is done with stage q - 1 and the A image
    if (q + NST - 1 < nq) issue(q + NST - 1);
    unsigned char* st = smem + (q % NST) * kStage;
    const float* scal = reinterpret_cast<const float*>(smem + kScal + (q % NST) * 512);
    const int64_t row0 = (blockIdx.x + static_cast<int64_t>(q) * gridDim.x) * kGxRows;
    const int valid = p.n - row0 < kGxRows ? static_cast<int>(p.n - row0) : kGxRows;

    // ---- form the A operand: rows 4 wave .. 4 wave + 3, half a wave per row ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = 4 * wave + 2 * j + hw;
      const bool rok = rr < valid;
      const int slot = (2 * wave + j) * 1024 + lane * 16;
      const uint4 v0 = *reinterpret_cast<const uint4*>(st + slot);
      const uint4 v2 = *reinterpret_cast<const uint4*>(st + (NR - 1) * kGxOpBytes + slot);
      const uint32_t u0[4] = {v0.x, v0.y, v0.z, v0.w}, u2[4] = {v2.x, v2.y, v2.z, v2.w};
      float gv[8], zv[8], dv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gv[2 * e] = bf_lo(u0[e]); gv[2 * e + 1] = bf_hi(u0[e]);
        zv[2 * e] = bf_lo(u2[e]); zv[2 * e + 1] = bf_hi(u2[e]);
      }
      if (MODE == kGtBN) {
        if (has1) {
          const uint4 v1 = *reinterpret_cast<const uint4*>(st + kGxOpBytes + slot);
          const uint32_t u1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { gv[2 * e] += bf_lo(u1[e]); gv[2 * e + 1] += bf_hi(u1[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {                   // sgf_bn_bwd_apply's arithmetic
          const float xh = (zv[e] - bmu[e]) * brs[e];
          float gg = gv[e];
          if (p.relu) gg = (xh * bga[e] + bbe[e]) > 0.f ? gg : 0.f;
          gg -= bk0[e] + xh * bk1[e];
          dv[e] = rok ? bga[e] * brs[e] * gg : 0.f;
        }
      } else {                                           // sgf_ln_bwd's arithmetic
        const float mu = scal[rr], rs = scal[64 + rr];
        float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[e] = (zv[e] - mu) * rs;
          float gm = gv[e];
          if (p.relu) gm = fmaf(xh[e], bga[e], bbe[e]) > 0.f ? gm : 0.f;
          gm = (rok && 8 * cg + e < p.m) ? gm : 0.f;
          dxh[e] = gm * bga[e];
          s1 += dxh[e];
          s2 = fmaf(dxh[e], xh[e], s2);
          csb[e] += gm;
          csc[e] = fmaf(gm, xh[e], csc[e]);
        }
        s1 = half_sum(s1) * inv_d;
        s2 = half_sum(s2) * inv_d;
#pragma unroll
        for (int e = 0; e < 8; ++e) dv[e] = (rok && 8 * cg + e < p.m) ? rs * (dxh[e] - s1 - xh[e] * s2) : 0.f;
      }
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = pack_bf16(dv[2 * e], dv[2 * e + 1]);
        csa[2 * e] += bf_lo(o[e]);
        csa[2 * e + 1] += bf_hi(o[e]);
      }
      // the slot of (row rr, column group cg) in the transposing-read image
      const int r16 = rr & 15;
      const int unit = (2 * (rr >> 4) + ((r16 >> 2) & 1)) * 8 + (cg >> 2);
      const int u = 16 * (r16 >> 3) + 8 * ((cg >> 1) & 1) + 2 * (r16 & 3) + (cg & 1);
      *reinterpret_cast<uint4*>(smem + kAimg + unit * 512 + u * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();                                     // the A image is complete

    // ---- matrix cores ----
    if (act) {
      const unsigned char* sa = smem + kAimg;
      const unsigned char* sb = st + NR * kGxOpBytes;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af, bfr[4];
        {
          const unsigned char* u = sa + ((2 * s) * 8 + wave) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 8 * 512));
          af[0] = r0[0]; af[1] = r0[1]; af[2] = r0[2]; af[3] = r0[3];
          af[4] = r1[0]; af[5] = r1[1]; af[6] = r1[2]; af[7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          const unsigned char* u = sb + ((2 * s) * 4 + tn) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 4 * 512));
          bfr[tn][0] = r0[0]; bfr[tn][1] = r0[1]; bfr[tn][2] = r0[2]; bfr[tn][3] = r0[3];
          bfr[tn][4] = r1[0]; bfr[tn][5] = r1[1]; bfr[tn][6] = r1[2]; bfr[tn][7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          if (tn < ntb) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[tn], acc[tn], 0, 0, 0);
      }
    }
  }

  // ---- partial ----
  float* part = p.partial + static_cast<int64_t>(blockIdx.x) * kRedPartialStride;
  if (act) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
      if (tn < ntb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          part[(32 * wave + mfma32_row(r, lane)) * 256 + 32 * tn + (lane & 31)] = acc[tn][r];
  }
  __syncthreads();
  float* fl = reinterpret_cast<float*>(smem);
  const int ridx = 2 * wave + hw;                        // 16 threads share a column group
  auto park = [&](const float (&v)[8], int off) {
    *reinterpret_cast<float4*>(&fl[off + ridx * 256 + 8 * cg]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&fl[off + ridx * 256 + 8 * cg + 4]) = make_float4(v[4], v[5], v[6], v[7]);
  };
  park(csa, 0);
  if (MODE == kGtLN) {
    park(csb, 4096);
    park(csc, 8192);
  }
  __syncthreads();
  if (tid < 256) {
    constexpr int NV = MODE == kGtLN ? 3 : 1;
    const int dst[3] = {kRedTileElems, kRedVecB, kRedVecC};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += fl[4096 * v + r * 256 + tid];
      part[dst[v] + tid] = s;
    }
  }
  if (tid == 0) part[kRedTileElems + 256] = part[kRedTileElems + 257] = 0.f;
}


// ------------------------------------------------------------------------------------------------------------------------
// k_gramb2: a GraphConv layer's BatchNorm backward AND its weight gradients in one pass over (g, z) — replaces
// sgf_bn_bwd_apply (reads g, z, writes dz) + sgf_gram2 (reads dz, y, x0): dz = BatchNorm'(relu'(g)) is formed in LDS as in
// k_gramt<BN>, stored to HBM on the way (sgf_gcn_epilogue_dx2_acc needs it next) and multiplied with y (role 0) / x0 (role 1)
// from the LDS image: 5 [n, d] tensors of traffic instead of 6, one launch instead of two.  Paired as k_gramx: blocks b and
// b + 8 (one XCD) walk the same stages; both form dz (role 1 reads g, z through the XCD's L2), role 0 stores it.
// Ring of 3 stages x 48 KiB (g, z row-major + the role's B image) + the A image = all 160 KiB of the CU's LDS.
// ------------------------------------------------------------------------------------------------------------------------
struct Gramb2Args {
  const void *g, *z, *b, *b2;
  void* dz;
  int64_t ldg, ldz, ldb, ldb2, lddz;
  int64_t n;
  int32_t m, k;                  // columns of g / z / dz, columns of b / b2
  const float *mean, *rstd, *gamma, *beta, *stats;
  float inv_n;
  int32_t training, relu;
  float* partial;
};

__global__ __launch_bounds__(kGxThreads) void k_gramb2(Gramb2Args p) {
  constexpr int NST = 3;
  constexpr int kStage = 3 * kGxOpBytes;                           // g, z (row-major), B image
  constexpr int kAimg = NST * kStage;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kAimg + kGxOpBytes];     // 160 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wd = wave & 1;
  const int cg = lane & 31;
  const int hw = lane >> 5;
  const bool col_ok = 8 * cg < p.m;
  const int role = (blockIdx.x >> 3) & 1;
  const int64_t vblock = (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3);
  const int64_t vgrid = gridDim.x / 2;

  const int64_t total = (p.n + kGxRows - 1) / kGxRows;
  const int nq = vblock < total ? static_cast<int>((total - vblock + vgrid - 1) / vgrid) : 0;
  const uint32_t smem_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(SGF_LDS(unsigned char, smem)));

  const unsigned char* gg = static_cast<const unsigned char*>(p.g);
  const unsigned char* gz = static_cast<const unsigned char*>(p.z);
  const unsigned char* gb = static_cast<const unsigned char*>(role ? p.b2 : p.b);
  const int64_t pitch_g = p.ldg * 2, pitch_z = p.ldz * 2, pitch_b = (role ? p.ldb2 : p.ldb) * 2, pitch_dz = p.lddz * 2;
  const int rcol = (col_ok ? 8 * cg : p.m - 8) * 2;
  const int brow = 16 * (wave >> 2) + 4 * ((wave >> 1) & 1) + 8 * ((lane >> 4) & 1) + ((lane >> 1) & 3);
  int bcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = 64 * (2 * (wave & 1) + j) + 32 * (lane >> 5) + 16 * ((lane >> 3) & 1) + 8 * (lane & 1);
    bcol[j] = c + 8 <= p.k ? c : p.k - 8;
  }

  auto issue = [&](int q) {
    const int64_t row0 = (vblock + static_cast<int64_t>(q) * vgrid) * kGxRows;
    const uint32_t base = smem_lds + static_cast<uint32_t>((q % NST) * kStage);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t r = row0 + 4 * wave + 2 * j + hw;
      r = r < p.n ? r : p.n - 1;
      dma16(gg + r * pitch_g + rcol, base + (2 * wave + j) * 1024);
      dma16(gz + r * pitch_z + rcol, base + kGxOpBytes + (2 * wave + j) * 1024);
    }
    int64_t r = row0 + brow;
    r = r < p.n ? r : p.n - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(gb + r * pitch_b + bcol[j] * 2, base + 2 * kGxOpBytes + (2 * wave + j) * 1024);
  };

  float bmu[8], brs[8], bga[8], bbe[8], bk0[8], bk1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * cg + e;
    const bool ok = c < p.m;
    bga[e] = ok ? (p.gamma ? p.gamma[c] : 1.f) : 0.f;
    bbe[e] = ok ? (p.beta ? p.beta[c] : 0.f) : 0.f;
    bmu[e] = ok ? p.mean[c] : 0.f;
    brs[e] = ok ? p.rstd[c] : 0.f;
    bk0[e] = (ok && p.training) ? p.stats[c] * p.inv_n : 0.f;
    bk1[e] = (ok && p.training) ? p.stats[p.m + c] * p.inv_n : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)        // consumed here: hipcc's wait for these loads must not land inside the stage loop
    asm volatile("" : "+v"(bga[e]), "+v"(bbe[e]), "+v"(bmu[e]), "+v"(brs[e]), "+v"(bk0[e]), "+v"(bk1[e]));

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float csa[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csa[e] = 0.f;
  const int nta = p.m - 64 * wm <= 0 ? 0 : (p.m - 64 * wm > 32 ? 2 : 1);
  const int ntb = p.k - 128 * wd <= 0 ? 0 : ((p.k - 128 * wd + 31) / 32 > 4 ? 4 : (p.k - 128 * wd + 31) / 32);
  const bool act = nta > 0 && ntb > 0;

#pragma unroll 1
  for (int q = 0; q < NST - 1 && q < nq; ++q) issue(q);

#pragma unroll 1
  for (int q = 0; q < nq; ++q) {
    // issued after stage q's requests: [stores of q - 2] [requests of q + 1] [stores of q - 1]; at the tail everything is waited for
    if (q + 1 < nq) wait_vm(role == 0 ? (q >= 2 ? 10 : (q == 1 ? 8 : 6)) : 6);
    else wait_vm(0);
    __syncthreads();
    if (q + NST - 1 < nq) issue(q + NST - 1);
    unsigned char* st = smem + (q % NST) * kStage;
    const int64_t row0 = (vblock + static_cast<int64_t>(q) * vgrid) * kGxRows;
    const int valid = p.n - row0 < kGxRows ? static_cast<int>(p.n - row0) : kGxRows;

#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = 4 * wave + 2 * j + hw;
      const bool rok = rr < valid;
      const int slot = (2 * wave + j) * 1024 + lane * 16;
      const uint4 v0 = *reinterpret_cast<const uint4*>(st + slot);
      const uint4 v2 = *reinterpret_cast<const uint4*>(st + kGxOpBytes + slot);
      const uint32_t u0[4] = {v0.x, v0.y, v0.z, v0.w}, u2[4] = {v2.x, v2.y, v2.z, v2.w};
      float dv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {                     // sgf_bn_bwd_apply's arithmetic
        const float gval = (e & 1) ? bf_hi(u0[e >> 1]) : bf_lo(u0[e >> 1]);
        const float zval = (e & 1) ? bf_hi(u2[e >> 1]) : bf_lo(u2[e >> 1]);
        const float xh = (zval - bmu[e]) * brs[e];
        float gm = gval;
        if (p.relu) gm = (xh * bga[e] + bbe[e]) > 0.f ? gm : 0.f;
        gm -= bk0[e] + xh * bk1[e];
        dv[e] = rok ? bga[e] * brs[e] * gm : 0.f;
      }
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = pack_bf16(dv[2 * e], dv[2 * e + 1]);
        csa[2 * e] += bf_lo(o[e]);
        csa[2 * e + 1] += bf_hi(o[e]);
      }
      const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
      if (role == 0 && rok && col_ok)
        *reinterpret_cast<uint4*>(static_cast<unsigned char*>(p.dz) + (row0 + rr) * pitch_dz + 16 * cg) = ov;
      const int r16 = rr & 15;
      const int unit = (2 * (rr >> 4) + ((r16 >> 2) & 1)) * 8 + (cg >> 2);
      const int u = 16 * (r16 >> 3) + 8 * ((cg >> 1) & 1) + 2 * (r16 & 3) + (cg & 1);
      *reinterpret_cast<uint4*>(smem + kAimg + unit * 512 + u * 16) = ov;
    }
    __syncthreads();

    if (act) {
      const unsigned char* sa = smem + kAimg;
      const unsigned char* sb = st + 2 * kGxOpBytes;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const unsigned char* u = sa + ((2 * s) * 8 + 2 * wm + tm) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 8 * 512));
          af[tm][0] = r0[0]; af[tm][1] = r0[1]; af[tm][2] = r0[2]; af[tm][3] = r0[3];
          af[tm][4] = r1[0]; af[tm][5] = r1[1]; af[tm][6] = r1[2]; af[tm][7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          const unsigned char* u = sb + ((2 * s) * 8 + 4 * wd + tn) * 512 + lane * 8;
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SGF_LDS(s16x4, u + 8 * 512));
          bfr[tn][0] = r0[0]; bfr[tn][1] = r0[1]; bfr[tn][2] = r0[2]; bfr[tn][3] = r0[3];
          bfr[tn][4] = r1[0]; bfr[tn][5] = r1[1]; bfr[tn][6] = r1[2]; bfr[tn][7] = r1[3];
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          if (tn < ntb) {
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[tn], acc[0][tn], 0, 0, 0);
            if (nta > 1) acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[tn], acc[1][tn], 0, 0, 0);
          }
        }
      }
    }
  }

  float* part = p.partial + (role * vgrid + vblock) * kRedPartialStride;
  if (act) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          part[(64 * wm + 32 * tm + mfma32_row(r, lane)) * 256 + 128 * wd + 32 * tn + (lane & 31)] = acc[tm][tn][r];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* fl = reinterpret_cast<float*>(smem);
  const int ridx = 2 * wave + hw;
  *reinterpret_cast<float4*>(&fl[ridx * 256 + 8 * cg]) = make_float4(csa[0], csa[1], csa[2], csa[3]);
  *reinterpret_cast<float4*>(&fl[ridx * 256 + 8 * cg + 4]) = make_float4(csa[4], csa[5], csa[6], csa[7]);
  __syncthreads();
  if (tid < 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += fl[r * 256 + tid];
    part[kRedTileElems + tid] = s;
  }
  if (tid == 0) part[kRedTileElems + 256] = part[kRedTileElems + 257] = 0.f;
}

bool aligned16(const void* p, int64_t ld) { return reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld % 8 == 0; }

}  // namespace

bool gramx_supported(const void* a, int64_t lda, int m, const void* b, int64_t ldb, int k, int64_t n) {
  static EnvInt on{"SGF_GRAMX", 1};
  return on.get() != 0 && m >= 8 && k >= 8 && m <= 256 && k <= 256 && m % 8 == 0 && k % 8 == 0 && n >= 4096 &&
         aligned16(a, lda) && aligned16(b, ldb);
}

int gramx_gram(const void* a, int64_t lda, int m, const void* b, int64_t ldb, const void* b2, int64_t ldb2, int k, int64_t n,
               float* partial, int* nblk, hipStream_t st) {
  const int64_t total = (n + kGxRows - 1) / kGxRows;
  GramxArgs g{};
  g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.b2 = b2; g.ldb2 = ldb2; g.n = n; g.m = m; g.k = k; g.partial = partial;
  g.same = (b2 == nullptr && a == b && lda == ldb && m == k) ? 1 : 0;
  int grid;
  if (b2 != nullptr) {
    int64_t pairs = total / 2 < kRedMaxBlocks / 2 ? total / 2 : kRedMaxBlocks / 2;
    pairs = pairs / 8 * 8;                               // whole groups of 8 pairs = 16 consecutive blocks (n >= 4096: >= 64)
    g.pair = 1;
    grid = static_cast<int>(2 * pairs);
    *nblk = static_cast<int>(pairs);
  } else {
    grid = static_cast<int>(total < kRedMaxBlocks ? total : kRedMaxBlocks);
    *nblk = grid;
  }
  hipLaunchKernelGGL((k_gramx<kGxGram>), dim3(grid), dim3(kGxThreads), 0, st, g);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

int gramx_bwdhs(const void* h, int64_t ldh, const void* g, int64_t ldg, const float* rowscal, int d, int64_t n, float* partial,
                int* nblk, hipStream_t st) {
  const int64_t total = (n + kGxRows - 1) / kGxRows;
  GramxArgs x{};
  x.a = h; x.lda = ldh; x.b = g; x.ldb = ldg; x.n = n; x.m = d; x.k = d; x.rowscal = rowscal; x.partial = partial;
  const int grid = static_cast<int>(total < kRedMaxBlocks ? total : kRedMaxBlocks);
  *nblk = grid;
  hipLaunchKernelGGL((k_gramx<kGxBwdHS>), dim3(grid), dim3(kGxThreads), 0, st, x);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

bool gramt_supported(int m, int k, int64_t n) {
  static EnvInt on{"SGF_GRAMX", 1};
  return on.get() != 0 && m >= 8 && m <= 256 && m % 8 == 0 && k >= 8 && k <= 128 && k % 8 == 0 && n >= 4096;
}
bool gramt_aligned(const void* p, int64_t ld) { return p == nullptr || aligned16(p, ld); }

int gramt_bn(const void* g1, int64_t ldg1, const void* g2, int64_t ldg2, const void* z, int64_t ldz, const float* mean,
             const float* rstd, const float* gamma, const float* beta, int relu, const float* stats, float inv_n, int training,
             int m, const void* b, int64_t ldb, int k, int64_t n, float* partial, int* nblk, hipStream_t st) {
  const int64_t total = (n + kGxRows - 1) / kGxRows;
  GramtArgs a{};
  a.r0 = g1; a.ld0 = ldg1; a.r1 = g2; a.ld1 = ldg2; a.r2 = z; a.ld2 = ldz; a.b = b; a.ldb = ldb; a.n = n; a.m = m; a.k = k;
  a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.stats = stats; a.inv_n = inv_n; a.training = training;
  a.relu = relu; a.partial = partial;
  const int grid = static_cast<int>(total < kRedMaxBlocks ? total : kRedMaxBlocks);
  *nblk = grid;
  hipLaunchKernelGGL((k_gramt<kGtBN>), dim3(grid), dim3(kGxThreads), 0, st, a);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

int gramt_ln(const void* g, int64_t ldg, const void* xin, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
             const float* beta, int relu, int m, const void* b, int64_t ldb, int k, int64_t n, float* partial, int* nblk,
             hipStream_t st) {
  const int64_t total = (n + kGxRows - 1) / kGxRows;
  GramtArgs a{};
  a.r0 = g; a.ld0 = ldg; a.r2 = xin; a.ld2 = ldx; a.b = b; a.ldb = ldb; a.n = n; a.m = m; a.k = k;
  a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.relu = relu; a.partial = partial;
  const int grid = static_cast<int>(total < kRedMaxBlocks ? total : kRedMaxBlocks);
  *nblk = grid;
  hipLaunchKernelGGL((k_gramt<kGtLN>), dim3(grid), dim3(kGxThreads), 0, st, a);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

bool gramb2_supported(int m, int k, int64_t n) {
  static EnvInt on{"SGF_GRAMX", 1};
  // OPT-IN (SGF_GRAM_BN2=1).  Measured on MI355X, products size (scripts/gramx_probe.py): sgf_bn_bwd_apply + sgf_gram2 1.519 ms,
  // this kernel 1.512 ms — level.  The 256 x 512 fp32 result needs two CUs' register files, so the pair fetches (g, z) twice:
  // unless the partner's copy is still in the XCD's L2 the launch moves 7 [n, d] tensors, not 5, and it runs at the
  // fabric's rate on those (5.8 TB/s).  Kept: correct (tests/test_gpu_gramx.py), one launch less, and the shape to start from
  // if the pair is ever made to run in step.
  static EnvInt fuse{"SGF_GRAM_BN2", 0};
  return on.get() != 0 && fuse.get() != 0 && m >= 8 && m <= 256 && m % 8 == 0 && k >= 8 && k <= 256 && k % 8 == 0 && n >= 16384;
}

int gramb2(const void* g, int64_t ldg, const void* z, int64_t ldz, const float* mean, const float* rstd, const float* gamma,
           const float* beta, int relu, const float* stats, float inv_n, int training, int m, const void* b1, int64_t ldb1,
           const void* b2, int64_t ldb2, int k, int64_t n, void* dz, int64_t lddz, float* partial, int* nblk, hipStream_t st) {
  const int64_t total = (n + kGxRows - 1) / kGxRows;
  int64_t pairs = total / 2 < kRedMaxBlocks / 2 ? total / 2 : kRedMaxBlocks / 2;
  pairs = pairs / 8 * 8;                                 // whole groups of 8 pairs (n >= 16384: >= 128)
  Gramb2Args a{};
  a.g = g; a.ldg = ldg; a.z = z; a.ldz = ldz; a.b = b1; a.ldb = ldb1; a.b2 = b2; a.ldb2 = ldb2; a.dz = dz; a.lddz = lddz;
  a.n = n; a.m = m; a.k = k; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.stats = stats; a.inv_n = inv_n;
  a.training = training; a.relu = relu; a.partial = partial;
  *nblk = static_cast<int>(pairs);
  hipLaunchKernelGGL(k_gramb2, dim3(static_cast<unsigned>(2 * pairs)), dim3(kGxThreads), 0, st, a);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

}  // namespace sgf
