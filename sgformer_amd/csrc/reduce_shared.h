// reduce_shared.h — the per-block partial layout of the node reductions (csrc/attn.hip, csrc/gramx.hip) and the entry points
// of the LDS-DMA reduce kernels (not part of the C ABI).
#pragma once
#include "common.h"

namespace sgf {

// One partial per workgroup: [RG x DP x DP result | DP column sums | 2 scalars | pad][DP second vector | scalar | pad]
// [DP third vector | pad]; the finalize kernels of attn.hip add the partials in a fixed order (no atomics anywhere).
constexpr int kRedTileElems = 65536;                       // RG * DP * DP, identical for every DP
constexpr int kRedVecB = kRedTileElems + 264;              // second column-sum vector + 1 scalar
constexpr int kRedVecC = kRedTileElems + 528;              // third column-sum vector
constexpr int kRedPartialStride = kRedTileElems + 792;
constexpr int kRedMaxBlocks = kNumCU;                      // persistent: one block per CU

// ---- csrc/gramx.hip: C = A^T B over the rows of two bf16 operands, tiles by LDS-DMA, fragments by transposing LDS reads ----
// (partials in the layout above with DP = 256, RG = 1)
bool gramx_supported(const void* a, int64_t lda, int m, const void* b, int64_t ldb, int k, int64_t n);
// plain (b2 == nullptr) or paired (b2 != nullptr: partials [role][pair]) Gram; *nblk = partials per product
int gramx_gram(const void* a, int64_t lda, int m, const void* b, int64_t ldb, const void* b2, int64_t ldb2, int k, int64_t n,
               float* partial, int* nblk, hipStream_t st);
// the attention backward reduce with the per-row scalars (1/den, dden) read: A = h, B = g / den; colsum = sum h dden,
// second vector = sum g / den, scalar = sum dden
int gramx_bwdhs(const void* h, int64_t ldh, const void* g, int64_t ldg, const float* rowscal, int d, int64_t n, float* partial,
                int* nblk, hipStream_t st);

// the stems' dW / db with the A operand formed in LDS from the streamed tensors (k_gramt): BatchNorm / LayerNorm backward;
// every tensor operand 16-byte aligned with ld % 8 == 0 (gramt_aligned), k <= 128.  LN: second / third vector = dbeta / dgamma
bool gramt_supported(int m, int k, int64_t n);
bool gramt_aligned(const void* p, int64_t ld);
int gramt_bn(const void* g1, int64_t ldg1, const void* g2, int64_t ldg2, const void* z, int64_t ldz, const float* mean,
             const float* rstd, const float* gamma, const float* beta, int relu, const float* stats, float inv_n, int training,
             int m, const void* b, int64_t ldb, int k, int64_t n, float* partial, int* nblk, hipStream_t st);
int gramt_ln(const void* g, int64_t ldg, const void* xin, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
             const float* beta, int relu, int m, const void* b, int64_t ldb, int k, int64_t n, float* partial, int* nblk,
             hipStream_t st);

// a GraphConv layer's BatchNorm backward + both weight-gradient blocks in one pass (k_gramb2): dz written to [n, m], partials
// [role][pair]; every tensor operand 16-byte aligned with ld % 8 == 0
bool gramb2_supported(int m, int k, int64_t n);
int gramb2(const void* g, int64_t ldg, const void* z, int64_t ldz, const float* mean, const float* rstd, const float* gamma,
           const float* beta, int relu, const float* stats, float inv_n, int training, int m, const void* b1, int64_t ldb1,
           const void* b2, int64_t ldb2, int k, int64_t n, void* dz, int64_t lddz, float* partial, int* nblk, hipStream_t st);

}  // namespace sgf
