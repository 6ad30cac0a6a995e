/*
 * sgf.h — C ABI of libsgf.so: the MI355X (gfx950) hot path of SGFormer forward/backward.
 *
 * This is the drop-in boundary.  The reference (qitianwu/SGFormer) has no FFI of
 * its own: its hot path is a chain of stock PyTorch / torch_sparse calls inside large/ours.py,
 * medium/ours.py and 100M/ours.py.  Every entry point below names the reference lines whose
 * arithmetic it replaces.  The host-side mirror of the reference's nn.Module surface
 * (sgformer_amd/ours.py) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / HIP types in any signature
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the caller allocates and owns every buffer, including outputs and workspaces
 *     (sizes from the *_workspace_bytes queries).  No hidden allocation, no host sync,
 *     no per-thread compute state: calls are re-entrant and may come from the autograd thread.
 *   - return value: 0 = ok, negative = error (SGF_E_*); sgf_last_error() gives the text of the
 *     most recent failure ON THE CALLING THREAD (errno-style; valid until that thread's next failure).
 *   - matrices are row-major with an explicit leading dimension `ld*` counted in ELEMENTS.
 *   - dtype codes: SGF_F32 = 0 (fp32 storage), SGF_BF16 = 1 (bf16 storage, fp32 accumulate).
 */
#ifndef SGF_H_
#define SGF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGF_VERSION 600 /* 0.6.0: sgf_gram2_bn_bwd; the node reductions (sgf_gram, sgf_gram2, sgf_gram_bn_bwd, sgf_gram_ln_bwd, sgf_attn_h_bwd_reduce_scaled) stream their tiles by LDS-DMA (csrc/gramx.hip) */

#define SGF_F32 0
#define SGF_BF16 1

#define SGF_OK 0
#define SGF_E_INVALID (-1)   /* bad argument (shape, alignment, dtype)            */
#define SGF_E_WORKSPACE (-2) /* workspace too small                                */
#define SGF_E_HIP (-3)       /* a HIP runtime call or kernel launch failed         */
#define SGF_E_UNSUPPORTED (-4) /* shape outside what the gfx950 kernels implement  */

int sgf_version(void);
const char* sgf_last_error(void);
/* Experiment switches (SGF_* environment variables read by the library) are cached per process; call this after changing
 * one so that it is read again at its next use. */
int sgf_reload_env(void);

/* ------------------------------------------------------------------------------------------
 * T1 — adjacency normalisation + CSR build.   Replaces large/ours.py:26-33
 * (= 100M/ours.py:72-79): degree(col, N); value = sqrt(1/d[col]) * sqrt(1/d[row]);
 * nan_to_num(->0); SparseTensor(row=col, col=row, value) i.e. A[col_e, row_e] = v_e with entries
 * ordered by (target, source), duplicates kept.  In the reference this (an argsort over all nnz
 * edges) is redone in every layer of every forward; here it is built once and cached by the
 * caller.
 *
 *   edge_index : int64 [2, nnz] row-major (edge_index[0] = source "row", edge_index[1] = target "col")
 *   rowptr     : int64 [n+1]   out — CSR row pointer over TARGET nodes
 *   colind     : int32 [nnz]   out — source node of each stored entry, ascending inside a row
 *   val        : fp32  [nnz]   out — sqrtf(1.0f/deg[tgt]) * sqrtf(1.0f/deg[src]), non-finite -> 0
 *   deg        : int32 [n]     out — in-degree (count of edges whose target is i)
 * Bit-exact contract: rowptr / colind equal a stable sort of the edges by key tgt*N+src; val is the
 * IEEE fp32 product of two correctly rounded sqrt(1/d) terms.  Requires 0 <= node id < n < 2^31.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_csr_workspace_bytes(int64_t nnz, int64_t n);
int sgf_csr_build(const int64_t* edge_index, int64_t nnz, int64_t n, int64_t* rowptr,
                  int32_t* colind, float* val, int32_t* deg, void* workspace,
                  size_t workspace_bytes, void* stream);

/* CSR of A^T (needed by the SpMM backward, dX = A^T dY, large/ours.py:34 under autograd, whenever
 * A is not symmetric: neighbour-sampled batches 100M/nb-sample.py:125-133, --directed runs,
 * ogbn-proteins).  `deg` is the in-degree array produced by sgf_csr_build.  `is_symmetric`
 * (int32[1], device) is set to 1 when A^T == A entry for entry, in which case the caller may
 * reuse the forward CSR and drop the transposed arrays. */
int sgf_csr_transpose(const int64_t* edge_index, int64_t nnz, int64_t n, const int32_t* deg,
                      const int64_t* rowptr, const int32_t* colind, int64_t* t_rowptr,
                      int32_t* t_colind, float* t_val, int32_t* is_symmetric, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * N1 — induced subgraph of a node subset, the per-mini-batch step of the reference's
 * random-partition trainer.   Replaces torch_geometric.utils.subgraph(idx_i, edge_index, num_nodes=n,
 * relabel_nodes=True) at large/main-batch.py:139 and large/eval.py:94 (third-party PyG 1.7.2, run on
 * the HOST by the reference: an O(E) mask + filter over all edges for every batch):
 *     keep edge e iff both endpoints are in `subset`; kept edges stay in their ORIGINAL ORDER;
 *     with relabel_nodes, node subset[j] becomes j.
 * Two calls around one host read of the output size:
 *     sgf_subgraph_plan : relabel[v] = position of v in subset (-1 = absent; int32 [n], out),
 *                         *total (device int64) = number of kept edges; per-block offsets stay in
 *                         `workspace`, which must be passed unchanged to
 *     sgf_subgraph_emit : out int64 [2, total] (row 0 sources, row 1 targets); out_eid int64 [total]
 *                         (optional, NULL to skip) = position of each kept edge in edge_index, for
 *                         filtering edge attributes.
 * `subset` must hold distinct node ids (the trainer's are slices of a permutation).  Ids outside
 * [0, n) never match.  n < 2^31.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_subgraph_workspace_bytes(int64_t nnz, int64_t n);
int sgf_subgraph_plan(const int64_t* edge_index, int64_t nnz, int64_t n, const int64_t* subset,
                      int64_t m, int32_t* relabel, int64_t* total, void* workspace,
                      size_t workspace_bytes, void* stream);
int sgf_subgraph_emit(const int64_t* edge_index, int64_t nnz, int64_t n, const int32_t* relabel,
                      int32_t relabel_nodes, int64_t total, int64_t* out, int64_t* out_eid,
                      void* workspace, size_t workspace_bytes, void* stream);

/* N1 + T1 in one step (r05; csrc/subgraph_csr.hip): the induced subgraph of `subset` AND its normalised CSR, straight from
 * the CSR of the PARENT graph (sgf_csr_build of the full edge list, cached by the caller) — a batch of m nodes reads its m
 * parent rows (20 MB per 100 k-node batch at ogbn-products size) instead of all parent edges twice (4 GB) plus a radix
 * sort of the result.  Replaces large/main-batch.py:139 + large/ours.py:26-33 for the batch:
 *     rowptr_b int64 [m + 1], colind_b int32 [total], val_b fp32 [total], deg_b int32 [m]  == bit for bit what
 *     sgf_csr_build returns for torch_geometric's subgraph(subset, edge_index, relabel_nodes=True);
 *     edge_index_b int64 [2, total] (optional): that edge list in (target, source) order — the same multiset of edges as
 *     the reference's, whose order (the parent's edge order) no consumer on the path depends on.
 * Two calls around ONE host read:  _plan -> total (int64 [3]): total[0] = number of kept entries, total[1] = 1 if `subset`
 * repeats a node or holds an id outside [0, n) (then call _emit with total = 0, which only resets the table, and use
 * sgf_subgraph_* instead), total[2] = the longest row of the batch CSR (<= sgf_spmm's long-row threshold: sgf_spmm_split is
 * not needed for it); _emit with the value read.  local_of: int32 [n] owned by the caller, all -1 between calls (the library restores it).
 * deg_b has m + 1 slots (the last one is scratch).  Workspaces: _plan_workspace_bytes(m), _emit_workspace_bytes(m, total). */
size_t sgf_subgraph_csr_plan_workspace_bytes(int64_t m);
size_t sgf_subgraph_csr_emit_workspace_bytes(int64_t m, int64_t total);
int sgf_subgraph_csr_plan(const int64_t* rowptr, const int32_t* colind, int64_t n, const int64_t* subset, int64_t m,
                          int32_t* local_of, int64_t* rowptr_b, int32_t* deg_b, int64_t* total, void* workspace,
                          size_t workspace_bytes, void* stream);
int sgf_subgraph_csr_emit(const int64_t* rowptr, const int32_t* colind, int64_t n, const int64_t* subset, int64_t m,
                          int32_t* local_of, const int64_t* rowptr_b, const int32_t* deg_b, int64_t total, int32_t* colind_b,
                          float* val_b, int64_t* edge_index_b, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * N2 — the trainer's graph prologue.   Replaces large/main.py:75-79 (and
 * 100M/nb-sample.py:79-80), three torch_geometric 1.7.2 utilities the trainers run on the HOST:
 *     to_undirected      : concatenate both directions, coalesce = sort by row*N+col, drop duplicates
 *     remove_self_loops  : keep entries with row != col, order preserved
 *     add_self_loops     : append (i, i) for i in [0, n)
 * Any subset of the three, applied in that order, selected by the flags.  Two calls around one host
 * read of the output size (as sgf_subgraph_*): _plan leaves the sorted keys / scan in `workspace`
 * and writes *total (device int64); _emit writes out int64 [2, total] (row 0 = edge_index[0]).
 * Without to_undirected the entries keep the caller's order.  n < 2^31, 2 m < 2^32 - 1.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_graph_prologue_workspace_bytes(int64_t m, int64_t n);
int sgf_graph_prologue_plan(const int64_t* edge_index, int64_t m, int64_t n, int32_t to_undirected,
                            int32_t remove_self_loops, int32_t add_self_loops, int64_t* total,
                            void* workspace, size_t workspace_bytes, void* stream);
int sgf_graph_prologue_emit(int64_t m, int64_t n, int32_t to_undirected, int32_t add_self_loops,
                            int64_t total, int64_t* out, void* workspace, size_t workspace_bytes,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * T2 — sum-reduce CSR SpMM.   Replaces torch_sparse.matmul(adj, x) at large/ours.py:34
 * (third-party torch_sparse 0.6.10 spmm, reduce="sum"):  Y[i,:] = sum_e val[e] * X[colind[e],:]
 * for e in [rowptr[i], rowptr[i+1]), accumulated in fp32 in stored order.
 * Backward (dX = A^T dY) is the same call on the transposed CSR (or the same CSR if symmetric).
 *   x : [n_cols, d] dtype, leading dim ldx (n_cols = columns of A = rows of x; when n_cols * ldx * elsize
 *       < 2^32 the row kernel addresses x with 32-bit buffer offsets: 8 instead of 21 instructions per
 *       stored entry);  y : [n_rows, d] dtype, leading dim ldy.
 * d, ldx, ldy must be multiples of 4 elements; x and y aligned to 4 elements.
 * ------------------------------------------------------------------------------------------ */
int sgf_spmm(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x,
             int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d,
             int32_t dtype, void* stream);

/* sgf_spmm with long rows split across workgroups (power-law graphs: a hub row of 17 k entries walked
 * by ONE wave is a latency-bound tail).  Rows with more than `long_len` stored entries are cut into
 * segments of sgf_spmm_segment_len() entries, each reduced by a whole workgroup into an fp32 partial;
 * a row's partials are then added in segment order (deterministic).  `long_segments` = the number of
 * such segments, sum over rows with len > long_len of ceil(len / segment_len), computed by the caller
 * once per CSR (0 = plain sgf_spmm).  Same result as sgf_spmm up to fp32 summation order. */
int32_t sgf_spmm_segment_len(void);
size_t sgf_spmm_split_workspace_bytes(int64_t long_segments, int32_t d);
int sgf_spmm_split(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x,
                   int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d, int32_t dtype,
                   int64_t long_len, int64_t long_segments, void* workspace, size_t workspace_bytes,
                   void* stream);

/* sgf_spmm_split for a CSR whose gathers mostly hit in L2 (a locality-restoring node order, sgf_reorder
 * below).  The vector memory pipe's address rate is per LANE, so bf16 rows (512 B) are then fetched TWO per
 * 16-byte-per-lane load: lanes 0-31 take the row of stream entry 2j, lanes 32-63 the row of entry 2j+1; a
 * wave walks the stored entries of 8 consecutive rows as one stream, the two half-waves keep partial sums of
 * the even / odd positions and are added when a row ends (k_spmm_seg_bf16x2).  Equal to sgf_spmm up to fp32
 * summation order, deterministic.  fp32 storage (already 16 B per lane) and operands beyond 4 GiB run the
 * sgf_spmm kernels.  long_segments = 0: no row is split. */
int sgf_spmm_stream(const int64_t* rowptr, const int32_t* colind, const float* val, const void* x,
                    int64_t ldx, int64_t n_cols, void* y, int64_t ldy, int64_t n_rows, int32_t d, int32_t dtype,
                    int64_t long_len, int64_t long_segments, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * T2 with on-chip reuse: a locality-restoring node order + an LDS-staged row-block SpMM.
 * Same arithmetic as sgf_spmm (large/ours.py:34, torch_sparse.matmul, sum-reduce) — what changes is
 * how often a neighbour row crosses the L2 / HBM boundary.
 *
 * sgf_reorder — a numbering that puts communities next to each other, from the graph alone: two
 *   levels of synchronous label propagation (nodes, then communities over the inter-community edges;
 *   a node adopts the label most of its in-neighbours carry, ties to the smallest), then nodes sorted
 *   by (level-2 label, level-1 community, id).  perm[p] = old id at new position p, inv[v] = new
 *   position of old node v, community[v] (optional, NULL to skip) = level-1 community index of v.
 *   Deterministic.  The caller relabels edge_index with inv and calls sgf_csr_build again; the module
 *   permutes x once on entry and the logits once on exit (everything in between is
 *   permutation-equivariant).  nnz < 2^32.
 *
 * sgf_spmm_plan — for a CSR (rowptr, colind, val) and blocks of `rows_per_block` consecutive rows: the
 *   sources referenced by >= 2 stored entries of a block, at most `lds_rows` of them (most-referenced
 *   first), are listed in sh_cols[sh_ptr[b] .. sh_ptr[b+1]) — the rows block b stages in LDS, slot =
 *   position in that list.  ecode / eval are colind / val with, inside every row, the entries served
 *   from LDS moved to the front (stable) and their code set to 0x80000000 | slot; nlds[row] = their
 *   number.  Rows longer than `long_len` keep plain source ids.  stats (int64[4], device): entries
 *   served from LDS, rows staged (sum over blocks), distinct (block, source) pairs, nnz.
 *   sh_cols must hold ceil(n / rows_per_block) * lds_rows entries.  nnz < 2^32 - 1.
 *
 * sgf_spmm_blocked — Y = A X with that plan: one workgroup per block (rows_per_block / 8 waves), the
 *   staged rows fetched once per block into LDS (144 KiB: lds_rows <= sgf_spmm_lds_rows_len(dtype)
 *   = 288 bf16 / 144 fp32), every other entry gathered as in sgf_spmm.  rows_per_block: multiple of 8,
 *   <= 128; d <= 256.  long_len / long_segments / workspace as for sgf_spmm_split (0 segments = none).
 *   Per row the LDS entries are accumulated first, then the gathered ones, each in ascending source
 *   order: equal to sgf_spmm up to fp32 summation order, deterministic.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_reorder_workspace_bytes(int64_t nnz, int64_t n);
int sgf_reorder(const int64_t* edge_index, int64_t nnz, int64_t n, int32_t iters1, int32_t iters2,
                int32_t* perm, int32_t* inv, int32_t* community, void* workspace, size_t workspace_bytes,
                void* stream);
size_t sgf_spmm_plan_workspace_bytes(int64_t nnz, int64_t n, int32_t rows_per_block);
int sgf_spmm_plan(const int64_t* rowptr, const int32_t* colind, const float* val, int64_t n, int64_t nnz,
                  int32_t rows_per_block, int32_t lds_rows, int64_t long_len, int32_t* ecode, float* eval,
                  int32_t* nlds, int32_t* sh_ptr, int32_t* sh_cols, int64_t* stats, void* workspace,
                  size_t workspace_bytes, void* stream);
int32_t sgf_spmm_lds_rows_len(int32_t dtype);
int sgf_spmm_blocked(const int64_t* rowptr, const int32_t* ecode, const float* eval, const int32_t* nlds,
                     const int32_t* sh_ptr, const int32_t* sh_cols, const void* x, int64_t ldx, void* y,
                     int64_t ldy, int64_t n_rows, int32_t d, int32_t dtype, int32_t rows_per_block,
                     int32_t lds_rows, int64_t long_len, int64_t long_segments, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T2 on a re-ordered graph, the matrix-core form (csrc/spmm_tile.hip).   Same product as sgf_spmm
 * (large/ours.py:34, torch_sparse.matmul, sum-reduce); bf16 storage, d = 128 or 256.
 * After sgf_reorder a community is a run of consecutive rows whose entries mostly point back into
 * the run: the diagonal blocks of A are 15-60 % dense.  Per block of <= 256 rows the sources that at
 * least `min_count` of the block's entries reference are STAGED (their rows of X pass once per block
 * through LDS) and the block's entries towards them are multiplied as a dense tile on the matrix
 * cores (value = hi + lo bf16, relative error <= 2^-17; products of bf16 values are exact in fp32,
 * fp32 accumulation); every other entry keeps its CSR form and is gathered; a row's two parts are
 * added in fp32 and rounded to bf16 once.  Deterministic.  Non-finite values of X spread inside a
 * block (0 x inf through the tile's zero cells).
 *
 * sgf_spmm_tile_blocks — block boundaries from the level-1 communities of sgf_reorder:
 *   comm_sorted[p] = community of the row at NEW position p (device; NULL = fixed blocks of max_rows).
 *   Runs of one community are packed greedily: consecutive runs share a block while they fit
 *   max_rows (a multiple of 32, <= 256); a longer run is cut into ceil(len / max_rows) pieces of
 *   min(max_rows, round_up(ceil(len / pieces), 32)) rows.  blk_row (device, int32[blk_cap + 1]) gets
 *   nb + 1 boundaries, *nb_out (host) the count (SGF_E_WORKSPACE with *nb_out set when blk_cap is too
 *   small; 4 n / max_rows + 4 always suffices).  Synchronises the stream (one device -> host copy of n ints).
 * sgf_spmm_tile_plan — for a CSR and those blocks: per block the <= cap (multiple of 32, <= 1024)
 *   most-referenced sources with >= min_count references, most-referenced first, ties to the smaller id,
 *   padded to a multiple of 32 with the block's first row: sh_cols[sh_ptr[b] .. sh_ptr[b+1]), slot =
 *   position (sh_cols must hold nb * cap entries).  ecode / eval / nlds as sgf_spmm_plan (tile entries
 *   first in every row, code 0x80000000 | slot).  tile_ptr[b] (int64[nb + 1]) = first 2 KiB fragment of
 *   block b, ceil(rows / 32) * (slots / 16) fragments per block; rem_rowptr (int64[n + 1]) = row
 *   pointers of the entries left on the gather path, every row's share padded to an EVEN count (the
 *   padding repeats the row's last source with value 0: the kernel fetches two rows of X per instruction
 *   and no pair straddles two target rows).  Rows longer than long_len stay entirely on it.
 *   stats (int64[8], device): tile entries, staged rows incl. padding, distinct (block, source) pairs,
 *   nnz, fragments, gathered entries incl. padding, 0, 1 if a block has 0 or more than 256 rows (plan unusable).
 * sgf_spmm_tile_fill — the arrays whose size the plan determines: tiles (n_frag * 2048 bytes; fragment
 *   of chunk q = slot / 32, row tile t, k-step s = (slot / 16) % 2 at tile_ptr[b] + (q * T + t) * 2 + s;
 *   inside: [hi: 64 lanes x 8 bf16][lo: 64 lanes x 8 bf16], lane = (row % 32) + 32 * ((slot % 16) / 8),
 *   element slot % 8 — the A operand of v_mfma_f32_32x32x16_bf16; duplicate edges are summed in fp32 in
 *   stored order), rem_col / rem_val (stats[5] entries, stored order + padding).
 * sgf_spmm_tile_pack_layout / sgf_spmm_tile_pack — the fragments in the form the kernel streams them (the kernel is
 *   bound by fabric traffic, and a 2 KiB fragment costs its 2 KiB whether 5 or 500 of its 512 cells are occupied;
 *   on a community graph 10-25 % are).  A GROUP = the two fragments (k-steps) one wave multiplies per chunk, numbered
 *   wave-major: g = tile_ptr[b] / 2 + t * Q_b + q (row tile t, chunk q, Q_b chunks in block b).  grp[2 g + 1] = its
 *   occupied cells if there are at most sgf_spmm_tile_sparse_len() (sparse), else -1 (dense); grp[2 g] = its offset
 *   in the pool in 16-byte units.  Sparse: the occupied cells as 8-byte entries {byte offset of the cell's hi element
 *   in the 4 KiB [k-step][hi, lo][lane][8] image, hi | lo << 16}, in image order, padded with a zero entry to an even
 *   count; dense: the 4 KiB image.  _layout writes grp and *pool_units (device); the caller allocates
 *   pool_units * 16 + 4096 bytes (the kernel fetches whole KiB) and calls _pack.  n_frag = tile_ptr[nb].
 *   The dense fragments are not needed afterwards.
 * sgf_spmm_tile — Y = A X with that plan; block_rows = the max_rows the blocks were made with (<= 128: four
 *   waves per block and two blocks per CU, else eight waves and one).  x: [n_cols, d] bf16, n_cols * ldx * 2 < 2^32; ldx, ldy
 *   multiples of 8, x / y 16-byte aligned.  long_len / long_segments / workspace as for sgf_spmm_split,
 *   counted on the REMAINDER row lengths (0 segments = no row is split).
 * ------------------------------------------------------------------------------------------ */
int sgf_spmm_tile_supported(int32_t d, int32_t dtype);
int sgf_spmm_tile_blocks(const int32_t* comm_sorted, int64_t n, int32_t max_rows, int32_t* blk_row,
                         int64_t blk_cap, int64_t* nb_out, void* stream);
size_t sgf_spmm_tile_plan_workspace_bytes(int64_t nnz, int64_t n, int64_t nb);
int sgf_spmm_tile_plan(const int64_t* rowptr, const int32_t* colind, const float* val, int64_t n, int64_t nnz,
                       const int32_t* blk_row, int64_t nb, int32_t cap, int32_t min_count, int64_t long_len,
                       int32_t* ecode, float* eval, int32_t* nlds, int32_t* sh_ptr, int32_t* sh_cols,
                       int64_t* tile_ptr, int64_t* rem_rowptr, int64_t* stats, void* workspace,
                       size_t workspace_bytes, void* stream);
int sgf_spmm_tile_fill(const int64_t* rowptr, const int32_t* ecode, const float* eval, const int32_t* nlds,
                       int64_t n, int64_t nnz, const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr,
                       int64_t n_frag, const int64_t* rem_rowptr, void* tiles, int32_t* rem_col, float* rem_val,
                       void* stream);
int32_t sgf_spmm_tile_sparse_len(void);
size_t sgf_spmm_tile_pack_workspace_bytes(int64_t n_frag);
int sgf_spmm_tile_pack_layout(const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr, const void* tiles,
                              int64_t n_frag, int32_t* grp, int64_t* pool_units, void* workspace,
                              size_t workspace_bytes, void* stream);
int sgf_spmm_tile_pack(const int32_t* blk_row, int64_t nb, const int64_t* tile_ptr, const void* tiles, int64_t n_frag,
                       const int32_t* grp, void* pool, int64_t pool_units, void* stream);
int sgf_spmm_tile(const int32_t* blk_row, int64_t nb, int32_t block_rows, const int32_t* sh_ptr, const int32_t* sh_cols,
                  const int64_t* tile_ptr, const int32_t* grp, const void* pool, const int64_t* rem_rowptr,
                  const int32_t* rem_col, const float* rem_val, const void* x, int64_t ldx, int64_t n_cols, void* y, int64_t ldy,
                  int64_t n_rows, int32_t d, int32_t dtype, int64_t long_len, int64_t long_segments,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * N2 — layer-wise neighbour sampling on the device.   Replaces NeighborLoader(data,
 * num_neighbors=[15, 10, 5], ...) of 100M/nb-sample.py:125-151 (third-party pyg-lib / torch_sparse
 * neighbor_sample, replace=False, directed=True).  Hop h gives every node that ENTERED the batch in
 * hop h-1 (the seeds for h = 0) min(in-degree, fanout) of its in-neighbours, drawn without
 * replacement (fanout < 0: all of them; Floyd's subset sampling up to 32, selection sampling above); a neighbour not yet in the batch gets
 * the next local id in order of first appearance (so the seeds are rows [0, batch_size): the trainer
 * slices [:batch_size], 100M/nb-sample.py:29-30); sampled edges point neighbour -> node, in local ids.
 * The reference's random stream cannot be matched; the draw is a counter-based hash of
 * (seed, batch, hop, node id): reproducible and independent of the launch geometry.
 *   rowptr / colind : CSR over TARGET nodes (sgf_csr_build): colind[rowptr[v] .. rowptr[v+1]) = the
 *                     in-neighbours of v
 *   local_of        : int32[n_nodes] state, INT32_MIN = node not in the batch (fill it once);
 *                     sgf_neighbor_sample_mark sets local_of[ids[j]] = base + j (seeds: base 0) or INT32_MIN
 *                     (base < 0: reset after the batch)
 *   frontier        : the m nodes that entered in the previous hop, global ids, local ids
 *                     frontier_local0 .. frontier_local0 + m - 1;  n_known = nodes in the batch so far
 *   out             : edge_src_local / edge_dst_local / src_global [edge_cap >= m * fanout], new_nodes
 *                     (global ids of the nodes that entered, in local-id order), counts (device int64[2]:
 *                     edges, new nodes).  Synchronises the stream once (the edge count sizes the launches).
 * ------------------------------------------------------------------------------------------ */
size_t sgf_neighbor_sample_workspace_bytes(int64_t m, int64_t edge_cap);
int sgf_neighbor_sample_mark(int32_t* local_of, const int32_t* ids, int64_t count, int32_t base, void* stream);
int sgf_neighbor_sample_hop(const int64_t* rowptr, const int32_t* colind, const int32_t* frontier, int64_t m,
                            int32_t frontier_local0, int32_t fanout, uint64_t seed, uint64_t batch, int32_t hop,
                            int32_t* local_of, int32_t n_known, int64_t edge_cap, int32_t* edge_src_local,
                            int32_t* edge_dst_local, int32_t* src_global, int32_t* new_nodes, int64_t* counts,
                            void* workspace, size_t workspace_bytes, void* stream);

/* A whole batch — every hop of it — WITHOUT a host read (fan-outs >= 0): the frontier size, the local-id base and the output
 * offsets of every hop stay in device memory, the launches are sized for the capacities
 *   frontier_0 = batch_size, edges_h = frontier_h * fanout_h, frontier_{h+1} = edges_h
 * (sgf_neighbor_sample_batch_workspace_bytes also returns them: node_cap = batch_size + sum edges_h, edge_cap = sum edges_h;
 * 0 bytes = a negative fan-out or more than 2^31 - 2 entries: not supported).  Same draws, same local ids, same edge order as
 * sgf_neighbor_sample_mark + one sgf_neighbor_sample_hop per hop + the un-marking, bit for bit.
 *   nodes  : out, int32[node_cap] global ids in local-id order (seeds first)
 *   edge_* : out, int32[edge_cap] local ids, hop after hop
 *   counts : out, device int64[2 + 2 * hops] = { nodes, edges, edges of hop 0, nodes entered in hop 0, ... } — the ONE read the
 *            caller needs to size its views.  local_of is back to INT32_MIN for every node of the batch on return. */
size_t sgf_neighbor_sample_batch_workspace_bytes(int64_t batch_size, const int32_t* fanouts, int32_t hops, int64_t* node_cap,
                                                  int64_t* edge_cap);
int sgf_neighbor_sample_batch(const int64_t* rowptr, const int32_t* colind, const int32_t* seeds, int64_t batch_size,
                              const int32_t* fanouts, int32_t hops, uint64_t seed, uint64_t batch, int32_t* local_of,
                              int32_t* nodes, int64_t node_cap, int32_t* edge_src_local, int32_t* edge_dst_local,
                              int64_t edge_cap, int64_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* dst[i, :] = src[idx[i], :] with an optional fp32 <-> bf16 storage change.  Replaces the row
 * gathers at the module boundary: x[idx_i] of a mini-batch (large/main-batch.py:138) and the
 * x[perm] / logits[inv] pair around a re-ordered graph.  idx: int32 or int64 (idx_is_int64) device
 * array of n_out row numbers; numbers outside [0, n_src) give zero rows.  Any d. */
int sgf_gather_rows(const void* src, int64_t lds, int32_t src_dtype, int64_t n_src, const void* idx,
                    int32_t idx_is_int64, int64_t n_out, int32_t d, void* dst, int64_t ldd,
                    int32_t dst_dtype, void* stream);
/* The same copy for feature widths that are not a multiple of 4 (pokec f = 65, Cora f = 1433; large/ours.py:77,:198 read x
 * as it comes): dst[i, :d] = src[idx ? idx[i] : i, :d], dst[i, d:d_pad] = 0, with the same optional storage change.  idx may
 * be NULL (identity).  Done once per feature tensor at the module entry; the stems then run on the aligned kernels with
 * zero-padded weight columns. */
int sgf_pad_rows(const void* src, int64_t lds, int32_t src_dtype, int64_t n_src, const void* idx, int32_t idx_is_int64,
                 int64_t n_out, int32_t d, int32_t d_pad, void* dst, int64_t ldd, int32_t dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * T3 — linear global attention core.   Replaces large/ours.py:130-149,157
 * (= medium/ours.py:14-46 full_attention_conv, 100M/ours.py:12-53), per head h:
 *     qn = Q/||Q||_F ; kn = K/||K||_F            (Frobenius norms over the WHOLE [N,H,d] tensors)
 *     S = kn^T V ; z = sum_l kn_l ; num = qn S + N V ; den = qn z + N ; out = mean_h(num/den)
 * The library never writes normalised copies of Q/K: it reduces the un-normalised partials
 *     stats = [ S0 = K^T V  (H*d*d) | z0 = sum_l K_l (H*d) | ssq_q | ssq_k ]       (fp32)
 * in one streaming pass (sgf_attn_fwd_reduce), which is also exactly the buffer a node-sharded
 * multi-GPU run all-reduces, and applies them in a second pass
 * (sgf_attn_fwd_apply) with  c = 1/(||Q|| ||K||):
 *     den_n = c * Q_n.z0 + Ntot ;  out_n = (1/H) sum_h (c * Q_n S0 + Ntot * V_n) / den_n
 *
 *   q,k  : [n, H, d] (leading dims ldq/ldk >= H*d);   v : [n, Hv, d], Hv = H or 1 (Hv = 1 is the
 *          use_weight=False broadcast of large/ours.py:128,138)
 *   stats: fp32 [H*d*d + H*d + 2]  (sgf_attn_stats_len)
 *   n_total : the N that appears in num/den (global node count when node-sharded)
 *   den  : fp32 [n, H] out (saved for backward);  o_heads : [n, H, d] per-head outputs, required
 *          (non-NULL) when H > 1, ignored when H == 1 (then o == out).
 * Supported: d % 4 == 0 and d <= 256 (all reference recipes use 64 / 128 / 256).
 * ------------------------------------------------------------------------------------------ */
int64_t sgf_attn_stats_len(int32_t heads, int32_t d);
size_t sgf_attn_workspace_bytes(int64_t n, int32_t heads, int32_t d);

int sgf_attn_fwd_reduce(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                        int64_t ldv, int64_t n, int32_t heads, int32_t v_heads, int32_t d,
                        int32_t dtype, float* stats, void* workspace, size_t workspace_bytes,
                        void* stream);

int sgf_attn_fwd_apply(const void* q, int64_t ldq, const void* v, int64_t ldv, int64_t n,
                       double n_total, int32_t heads, int32_t v_heads, int32_t d, int32_t dtype,
                       const float* stats, void* out, int64_t ldo, float* den, void* o_heads,
                       void* stream);

/* Backward of the above (hand-derived), given g = dL/dout [n, d]:
 *     dnum = (g/H)/den ; dden = -((g/H).o)/den                              (row-local)
 *     bstats = [ dS0 = sum_n Q_n^T dnum_n (H*d*d) | dz0 = sum_n Q_n dden_n (H*d) ]   (fp32)
 * (sgf_attn_bwd_reduce; again the all-reduce payload when node-sharded), then with
 *     s = c * (<S0,dS0> + <z0,dz0>)        (= <qn,dqn> = <kn,dkn>, both radial terms)
 *     dQ = c (dnum S0^T + dden z0) - s Q/||Q||^2
 *     dK = c (V dS0^T + dz0)       - s K/||K||^2
 *     dV = Ntot dnum + c K dS0                       (summed over heads when Hv == 1)
 * (sgf_attn_bwd_apply).  `o` is o_heads when H > 1 and `out` when H == 1.
 * bstats has H*d*d + H*d + 1 floats: the last slot is zeroed by sgf_attn_bwd_reduce (so the whole
 * buffer can be all-reduced) and filled with <S0,dS0> + <z0,dz0> by sgf_attn_bwd_apply. */
int64_t sgf_attn_bstats_len(int32_t heads, int32_t d);

int sgf_attn_bwd_reduce(const void* q, int64_t ldq, const void* g, int64_t ldg, const void* o,
                        int64_t ldo, const float* den, int64_t n, int32_t heads, int32_t d,
                        int32_t dtype, float* bstats, void* workspace, size_t workspace_bytes,
                        void* stream);

int sgf_attn_bwd_apply(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                       int64_t ldv, const void* g, int64_t ldg, const void* o, int64_t ldo,
                       const float* den, int64_t n, double n_total, int32_t heads,
                       int32_t v_heads, int32_t d, int32_t dtype, const float* stats,
                       float* bstats, void* dq, int64_t lddq, void* dk, int64_t lddk,
                       void* dv, int64_t lddv, void* stream);
/* The same two calls for PER-HEAD output gradients: full_attention_conv returns [N, H, D] (medium/ours.py:14-46,
 * 100M/ours.py:12-53) and a caller may differentiate through the heads before any mean.  g: [n, H, d] (ldg >= H * d); head h's
 * gradient enters as it is (no 1/H).  H = 1 coincides with sgf_attn_bwd_reduce / _apply. */
int sgf_attn_bwd_reduce_heads(const void* q, int64_t ldq, const void* g, int64_t ldg, const void* o, int64_t ldo,
                              const float* den, int64_t n, int32_t heads, int32_t d, int32_t dtype, float* bstats,
                              void* workspace, size_t workspace_bytes, void* stream);
int sgf_attn_bwd_apply_heads(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* g,
                             int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n, double n_total,
                             int32_t heads, int32_t v_heads, int32_t d, int32_t dtype, const float* stats, float* bstats,
                             void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* stream);

/* ------------------------------------------------------------------------------------------
 * T3 + T4 fused — attention straight from the un-projected layer input (H = 1, query == source:
 * every recipe of the reference).  Replaces large/ours.py:123-157 as a whole: the projections
 * Q = h Wq^T + bq, K = h Wk^T + bk, V = h Wv^T + bv are never materialised.  Because the attention
 * only needs K^T V, sum K, ||Q||, ||K|| and Q applied to d x d matrices, everything global follows
 * from the Gram matrix G = h^T h and s = sum_n h_n (sgf_gram(h, h) with its column sums):
 *     S0 = Wk G Wv^T + (Wk s) bv^T + bk (Wv s)^T + N bk bv^T        z0 = Wk s + N bk
 *     ||Q||^2 = tr(Wq G Wq^T) + 2 bq.(Wq s) + N |bq|^2   (same for K)    c = 1 / (||Q|| ||K||)
 *     num = h M + m,  M = c Wq^T S0 + Ntot Wv^T,  m = c bq S0 + Ntot bv
 *     den = h.w + beta,  w = c Wq^T z0,  beta = c bq.z0 + Ntot            out = num / den
 * The d x d algebra (M, m, w, beta and its backward) is a handful of tiny fp32 GEMMs done by the
 * caller (sgformer_amd/ops.py, on device, autograd); the library streams the [N, d] operands:
 *     sgf_attn_h_fwd        : out = (h M + m) / (h.w + beta), den saved            (reads h once)
 *     sgf_attn_h_bwd_reduce : hstats = [ dM = sum h^T dnum | dw = sum h dden | dm = sum dnum |
 *                             dbeta = sum dden ],  dnum = g/den, dden = -(g.out)/den
 *     sgf_attn_h_bwd_apply  : dh = dnum M^T + dden w + h D + ds   with D = dG + dG^T, ds from the
 *                             caller's backward through the d x d algebra
 * HBM traffic per layer: 3 [N,d] passes forward, 9 backward — against ~31 for the materialised
 * Q/K/V form (projection GEMM + its dX / dW, two reduce and four apply passes).  Node-sharded runs
 * all-reduce [G | s] and hstats (d^2 + d and d^2 + 2d + 1 floats).
 * M, D: fp32 [d, d] row-major (M[k][j]: input feature k -> output feature j); m, w, ds: fp32 [d];
 * beta: fp32 [1]; all DEVICE pointers.  d % 4 == 0, d <= 256.
 * ------------------------------------------------------------------------------------------ */
int64_t sgf_attn_h_bstats_len(int32_t d);
int sgf_attn_h_fwd(const void* h, int64_t ldh, int64_t n, int32_t d, int32_t dtype, const float* M,
                   const float* m, const float* w, const float* beta, void* out, int64_t ldo,
                   float* den, void* stream);
int sgf_attn_h_bwd_reduce(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o,
                          int64_t ldo, const float* den, int64_t n, int32_t d, int32_t dtype,
                          float* hstats, void* workspace, size_t workspace_bytes, void* stream);
int sgf_attn_h_bwd_apply(const void* h, int64_t ldh, const void* g, int64_t ldg, const void* o,
                         int64_t ldo, const float* den, int64_t n, int32_t d, int32_t dtype,
                         const float* M, const float* w, const float* D, const float* ds, void* dh,
                         int64_t lddh, void* workspace, size_t workspace_bytes, void* stream);
/* scratch of sgf_attn_h_bwd_apply: bf16 storage with d in {64, 128, 256} runs as two per-wave streaming passes
 * (csrc/rowgemm.hip) with the first product parked in the matrix cores' accumulator layout; 0 otherwise */
size_t sgf_attn_h_bwd_apply_workspace_bytes(int64_t n, int32_t d, int32_t dtype);
/* The same backward as three calls (bf16 storage, d in {64, 128, 256}: sgf_attn_h_bwd_split_supported), ordered so that
 * the node reduction can use what the first apply pass computes anyway:
 *   sgf_attn_h_bwd_pre            : partial (in `workspace`, sgf_attn_h_bwd_apply_workspace_bytes) = dnum M^T + dden w,
 *                                   and rowscal[n][2] = (1 / den, dden = -(g . out) / den) per node;
 *   sgf_attn_h_bwd_reduce_scaled  : hstats as sgf_attn_h_bwd_reduce, from h, g and rowscal — two streams instead of
 *                                   three (no `out`, no per-row dot);
 *   sgf_attn_h_bwd_post           : dh = h D + ds + partial [+ addend].  `addend` (nullable, storage dtype, leading
 *                                   dim ldadd): a second gradient of h — the residual branch's, large/ours.py:206-208 uses
 *                                   the layer input twice — added to the rounded result here instead of in a separate
 *                                   three-tensor pass.
 * sgf_attn_h_bwd_reduce + sgf_attn_h_bwd_apply remain for everything else. */
int32_t sgf_attn_h_bwd_split_supported(int32_t d, int32_t dtype);
int sgf_attn_h_bwd_pre(const void* g, int64_t ldg, const void* o, int64_t ldo, const float* den, int64_t n, int32_t d,
                       int32_t dtype, const float* M, const float* w, void* workspace, size_t workspace_bytes,
                       float* rowscal, void* stream);
int sgf_attn_h_bwd_reduce_scaled(const void* h, int64_t ldh, const void* g, int64_t ldg, const float* rowscal, int64_t n,
                                 int32_t d, int32_t dtype, float* hstats, void* workspace, size_t workspace_bytes,
                                 void* stream);
int sgf_attn_h_bwd_post(const void* h, int64_t ldh, int64_t n, int32_t d, int32_t dtype, const float* D, const float* ds,
                        const void* workspace, size_t workspace_bytes, const void* addend, int64_t ldadd, void* dh,
                        int64_t lddh, void* stream);

/* The d x d algebra between the two node passes, as ONE entry each way (csrc/attn_small.hip; r05 — r04 issued it from
 * Python as 12 + 20 ATen / rocBLAS launches).  Replaces the small dense part of large/ours.py:123-149 with the projections
 * folded in (formulas above), written on augmented operands:
 *   sgf_attn_h_small_fwd : (G [d_in, d_in], s [d_in], n_rows, n_total, wq / wk / wv [d_out, d_in] with leading dim ldw,
 *                           bq / bk / bv [d_out]; wv = NULL: V is h itself, use_weight=False, needs d_in == d_out)
 *                          -> M [d_in, d_out] (ldm), m [d_out], w [d_in], beta [1]   — the operands of sgf_attn_h_fwd;
 *                          `saved` (sgf_attn_h_small_saved_bytes, caller-owned, opaque) keeps what the backward needs.
 *                          6 launches: pack, 3 products on the exact-fp32 matrix cores (csrc/gemm.hip), ||Q||^2 / ||K||^2
 *                          (one block, fixed summation order), scale + unpack.
 *   sgf_attn_h_small_bwd : (dM [d_in, d_out] (lddm), dw [d_in], dm [d_out], dbeta [1] = the blocks of hstats, after the
 *                          ranks' all-reduce) -> dG2 = dG + dG^T [d_in, d_in] (lddg) and ds [d_in] — the D / ds operands
 *                          of sgf_attn_h_bwd_post / _apply — and the six parameter gradients gwq / gwk / gwv [d_out, d_in]
 *                          (ldgw), gbq / gbk / gbv [d_out] (any of them NULL to skip).  9 launches; workspace:
 *                          sgf_attn_h_small_workspace_bytes.
 * n_rows = the number of rows behind G and s (global count when sharded); n_total = the N of large/ours.py:133.
 * Everything fp32 and on the device: no host synchronisation, no host read of ||Q|| or ||K||. */
size_t sgf_attn_h_small_saved_bytes(int32_t d_in, int32_t d_out);
size_t sgf_attn_h_small_workspace_bytes(int32_t d_in, int32_t d_out);
int sgf_attn_h_small_fwd(const float* G, int64_t ldg, const float* s, float n_rows, float n_total, const float* wq,
                         const float* bq, const float* wk, const float* bk, const float* wv, const float* bv, int64_t ldw,
                         int32_t d_in, int32_t d_out, float* M, int64_t ldm, float* m, float* w, float* beta, void* saved,
                         size_t saved_bytes, void* stream);
int sgf_attn_h_small_bwd(const float* dM, int64_t lddm, const float* dw, const float* dm, const float* dbeta, float n_total,
                         int32_t d_in, int32_t d_out, const void* saved, size_t saved_bytes, float* dG2, int64_t lddg,
                         float* ds, float* gwq, float* gbq, float* gwk, float* gbk, float* gwv, float* gbv, int64_t ldgw,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T4/T6/T7 — weight and bias gradients of the Linear layers.   Replaces what autograd does for
 * every nn.Linear on the path under loss.backward() (large/main.py:142): the projections
 * large/ours.py:123-126, the GraphConvLayer weight :36-40, the stems :77,:198 and the head :275:
 *     dW = dY^T X   ([N, m]^T [N, k] -> [m, k]),    db = sum_n dY[n, :]
 * i.e. a d x d <- [N x d]^T [N x d] contraction over ALL nodes: the same streaming skeleton as
 * sgf_attn_fwd_reduce (K^T V), on the bf16 matrix cores for SGF_BF16 and the exact-fp32 MFMA for
 * SGF_F32, fp32 accumulation, deterministic two-stage reduction.  hipBLASLt's kernels for this
 * shape ran at 0.7 TB/s (3.6 ms per call at ogbn-products scale, profiles/); this one is HBM-bound.
 *     c[i, j] = sum_n a[n, i] * b[n, j]       c: fp32 [m, k] row-major, leading dim ldc
 *     colsum_a[i] = sum_n a[n, i]             fp32 [m], optional (NULL to skip)
 * m, k multiples of 4 (any size: tiled in 256 x 256 blocks).  Node-sharded runs all-reduce c.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_gram_workspace_bytes(int64_t n, int32_t m, int32_t k);
int sgf_gram(const void* a, int64_t lda, int32_t m, const void* b, int64_t ldb, int32_t k,
             int64_t n, int32_t dtype, float* c, int64_t ldc, float* colsum_a, void* workspace,
             size_t workspace_bytes, void* stream);
/* Two such products that share a — c1 = a^T b1, c2 = a^T b2 (dW of the two-operand Linear: a = dz, b1 = Ax, b2 = x0) — in ONE
 * paired launch (bf16 storage, m, k <= 256): workgroups b and b + 8 (one XCD) walk the same row tiles, a's second read is an
 * L2 hit.  Other shapes / fp32: two sgf_gram launches.  Same workspace as sgf_gram; colsum_a as there. */
int sgf_gram2(const void* a, int64_t lda, int32_t m, const void* b1, int64_t ldb1, const void* b2, int64_t ldb2, int32_t k,
              int64_t n, int32_t dtype, float* c1, int64_t ldc1, float* c2, int64_t ldc2, float* colsum_a, void* workspace,
              size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T5 — TransConv glue.   Replaces large/ours.py:198-202 and :210-216 (medium/ours.py:150-156,
 * 100M/ours.py:262-268):   y = [relu]( LayerNorm( a * x + b * res ) )   row-wise
 * (large: a = b = 1/2; medium / 100M: a = alpha, b = 1 - alpha; input stem: a = 1, res = NULL).
 * res NULL = no residual; gamma NULL = no LayerNorm (use_bn False; beta then ignored).
 * Saves mean / rstd per row (fp32 [n] each, may be NULL when gamma is NULL) for the backward.
 * d % 4 == 0, d <= 1024.
 * ------------------------------------------------------------------------------------------ */
int sgf_ln_fwd(const void* x, int64_t ldx, const void* res, int64_t ldr, float a, float b,
               const float* gamma, const float* beta, int32_t relu, float eps, int64_t n,
               int32_t d, int32_t dtype, void* y, int64_t ldy, float* mean, float* rstd,
               void* stream);

/* Backward: dz = dy masked by relu (y > 0);  LayerNorm backward to dpre;  dx = a * dpre,
 * dres = b * dpre (dres may be NULL);  dgamma / dbeta (fp32 [d], may be NULL when gamma is NULL)
 * are reduced deterministically through per-block partials in `workspace`. */
size_t sgf_ln_bwd_workspace_bytes(int64_t n, int32_t d);
int sgf_ln_bwd(const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* x,
               int64_t ldx, const void* res, int64_t ldr, float a, float b, const float* gamma,
               int32_t relu, const float* mean, const float* rstd, int64_t n, int32_t d,
               int32_t dtype, void* dx, int64_t lddx, void* dres, int64_t lddres, float* dgamma,
               float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T6 — GraphConv glue.   Replaces large/ours.py:77-81 and :87-93:
 *     y = [relu]( BatchNorm1d(x) ) [+ res]          (dropout, when p > 0, stays in the caller)
 * BatchNorm1d statistics in training mode are over ALL rows — a hidden global reduction — so the
 * op is split at the reduction, which is also where a node-sharded run all-reduces:
 *     sgf_colstats     : stats = [ sum_n (x[n,j]-shift[j]) | sum_n (x[n,j]-shift[j])^2 ]  fp32 [2*d]
 *                        (shift NULL = 0).  Two calls give a two-pass mean / variance.
 *     sgf_bn_apply     : y = [relu]((x - mean[j]) * rstd[j] * gamma[j] + beta[j]) [+ res]
 *     sgf_bn_bwd_stats : stats = [ sum_n dz[n,j] | sum_n dz[n,j] * xhat[n,j] ]   (= dbeta | dgamma)
 *                        dz = dy masked by the recomputed relu, xhat = (x - mean) * rstd
 *     sgf_bn_bwd_apply : dx = gamma*rstd * (dz - [training](stats0*inv_n + xhat*stats1*inv_n))
 * The residual gradient is dy itself.  gamma / beta NULL = 1 / 0.  d % 4 == 0, d <= 1024.
 * ------------------------------------------------------------------------------------------ */
size_t sgf_colstats_workspace_bytes(int64_t n, int32_t d);
int sgf_colstats(const void* x, int64_t ldx, const float* shift, int64_t n, int32_t d,
                 int32_t dtype, float* stats, void* workspace, size_t workspace_bytes,
                 void* stream);
/* BatchNorm's bookkeeping between its statistics pass and its apply pass, one launch: from the shifted column sums
 * sums = [sum (x - shift) | sum (x - shift)^2] over n_total rows (shift may be null):  mean, rstd = 1 / sqrt(var + eps) with
 * the biased variance, and — when the pointers are given — nn.BatchNorm1d's running update
 *   running_mean = (1 - momentum) running_mean + momentum mean,  running_var likewise with the UNBIASED variance. */
int sgf_bn_finalize(const float* sums, const float* shift, double n_total, float eps, float momentum, float* running_mean,
                    float* running_var, int32_t d, float* mean, float* rstd, void* stream);
int sgf_bn_apply(const void* x, int64_t ldx, const float* mean, const float* rstd,
                 const float* gamma, const float* beta, const void* res, int64_t ldr,
                 int32_t relu, int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy,
                 void* stream);
int sgf_bn_bwd_stats(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int32_t relu,
                     int64_t n, int32_t d, int32_t dtype, float* stats, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same sums for a tensor with TWO incoming gradients (dy + dy2, added in fp32; dy2 may be null): the stem's output feeds
 * the first SpMM and the layers' residuals / Linear blocks, whose gradients arrive separately. */
int sgf_bn_bwd_stats2(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* x, int64_t ldx,
                      const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, int64_t n,
                      int32_t d, int32_t dtype, float* stats, void* workspace, size_t workspace_bytes, void* stream);
/* dW / db of a Linear whose output z feeds a BatchNorm, without a materialised dz (the stem of GraphConv, large/ours.py:77-81:
 * its input is data, nobody else needs dz):  c[m, k] = dz^T b,  colsum[m] = sum_n dz,  where
 * dz = sgf_bn_bwd_apply(g1 [+ g2], z, ...) is formed per 4 x 4 patch inside the Gram kernel's staging step (bf16 storage, m and
 * k multiples of 4 up to 256).  stats as returned by sgf_bn_bwd_stats2; workspace: sgf_gram_workspace_bytes. */
/* The LayerNorm form (TransConv's stem, large/ours.py:198-201: Linear -> LayerNorm -> relu, input = data): from g = d(output),
 * the LayerNorm's input xin and the forward's per-row mean / rstd:
 *   c[m, k] = dl^T b, colsum[m] = sum_n dl (the Linear's dW / db), dgamma[m] = sum g' xhat, dbeta[m] = sum g' (the LayerNorm's),
 * dl = sgf_ln_bwd's input gradient, formed per 4 x 4 patch inside the Gram kernel (row means over the m columns across the
 * lanes of a patch row) and never written.  bf16 storage, m in {64, 128, 256}, k % 4 == 0 up to 256.  gamma / beta null: a
 * LayerNorm without affine terms (NOT sgf_ln_fwd's "no LayerNorm"); dgamma / dbeta may be null.  workspace:
 * sgf_gram_workspace_bytes. */
int32_t sgf_gram_ln_bwd_supported(int32_t m, int32_t k, int32_t dtype);
int sgf_gram_ln_bwd(const void* g, int64_t ldg, const void* xin, int64_t ldx, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, int32_t relu, int32_t m, const void* b, int64_t ldb, int32_t k,
                    int64_t n, int32_t dtype, float* c, int64_t ldc, float* colsum, float* dgamma, float* dbeta, void* workspace,
                    size_t workspace_bytes, void* stream);
int32_t sgf_gram_bn_bwd_supported(int32_t m, int32_t k, int32_t dtype);
int sgf_gram_bn_bwd(const void* g1, int64_t ldg1, const void* g2, int64_t ldg2, const void* z, int64_t ldz, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, int32_t relu, const float* stats, float inv_n,
                    int32_t training, int32_t m, const void* b, int64_t ldb, int32_t k, int64_t n, int32_t dtype, float* c,
                    int64_t ldc, float* colsum, void* workspace, size_t workspace_bytes, void* stream);
/* A GraphConv layer's BatchNorm backward AND both blocks of its weight gradient in ONE pass over (g, z) (large/ours.py:36-40,
 * 87-93 differentiated; replaces sgf_bn_bwd_apply + sgf_gram2 — five [n, d] tensors of traffic instead of six):
 *   dz[n, m] = sgf_bn_bwd_apply(g, z, ...)  (written: sgf_gcn_epilogue_dx2_acc reads it next),
 *   c1[m, k] = dz^T b1, c2[m, k] = dz^T b2, colsum[m] = sum_n dz   (b1 = A x, b2 = x0: dW of W [A x | x0], db).
 * bf16 storage, m, k multiples of 8 up to 256, n >= 16384, every tensor operand 16-byte aligned with ld % 8 == 0
 * (sgf_gram2_bn_bwd_supported checks the sizes; a misaligned operand is SGF_E_INVALID).  workspace: sgf_gram_workspace_bytes. */
int32_t sgf_gram2_bn_bwd_supported(int32_t m, int32_t k, int64_t n, int32_t dtype);
int sgf_gram2_bn_bwd(const void* g, int64_t ldg, const void* z, int64_t ldz, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int32_t relu, const float* stats, float inv_n, int32_t training,
                     int32_t m, const void* b1, int64_t ldb1, const void* b2, int64_t ldb2, int32_t k, int64_t n, int32_t dtype,
                     void* dz, int64_t lddz, float* c1, int64_t ldc1, float* c2, int64_t ldc2, float* colsum, void* workspace,
                     size_t workspace_bytes, void* stream);
int sgf_bn_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int32_t relu,
                     const float* stats, float inv_n, int32_t training, int64_t n, int32_t d,
                     int32_t dtype, void* dx, int64_t lddx, void* stream);

/* ------------------------------------------------------------------------------------------
 * T7 — branch combine + output layer, fused (the "residual + MLP" kernel).   Replaces
 * large/ours.py:269-270,275 (aggregate == 'add'):
 *     x = graph_weight * x2 + (1 - graph_weight) * x1;   logits = fc(x) = x W^T + bias
 *   sgf_combine_fc_fwd : logits[n, classes] (fp32, leading dim ldl) = (a x1 + b x2) W^T + bias in ONE pass
 *                        over x1, x2 — the combined activations are rounded to the storage dtype once (where
 *                        the unfused axpby rounds them) and fed to the bf16 matrix cores, never written.
 *   sgf_combine_fc_bwd : dx1 = a (dlogits W), dx2 = b (dlogits W), storage dtype, one pass over dlogits
 *                        (fp32 [n, classes], leading dim lddl; rounded to the storage dtype in registers).
 *   dW = a dlogits^T x1 + b dlogits^T x2 and db are node reductions: sgf_gram.
 * W: fp32 [classes, d] row-major, bias fp32 [classes].  Implemented for bf16 storage, d % 32 == 0,
 * d <= 256, classes <= 64 (sgf_combine_fc_supported); anything else: sgf_axpby + a library GEMM.
 * bf16 storage, backward: n < 2^31 (SGF_E_UNSUPPORTED beyond).
 * ------------------------------------------------------------------------------------------ */
int32_t sgf_combine_fc_supported(int32_t d, int32_t classes, int32_t dtype);
int sgf_combine_fc_fwd(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
                       const float* w, const float* bias, int64_t n, int32_t d, int32_t classes,
                       int32_t dtype, float* logits, int64_t ldl, void* stream);
int sgf_combine_fc_bwd(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d,
                       int32_t classes, float a, float b, int32_t dtype, void* dx1, int64_t ld1, void* dx2,
                       int64_t ld2, void* stream);
/* The same pair with the module's row permutation folded in (bf16 storage, classes <= 64): on a graph the library has
 * re-ordered (sgf_reorder) the model works on permuted rows and its logits must go back to the caller's order
 * (sgformer_amd/ours.py; large/ours.py:275 returns them in the order of x).  _fwd_mapped stores row j of the product as row
 * row_map[j] of `logits`; _bwd_mapped reads row row_map[j] of `dlogits` for row j of dx1 / dx2.  row_map: int32[n], a
 * permutation of [0, n).  Saves the two [N, C] fp32 gather passes around the head. */
int sgf_combine_fc_fwd_mapped(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b, const float* w,
                              const float* bias, int64_t n, int32_t d, int32_t classes, int32_t dtype, float* logits,
                              int64_t ldl, const int32_t* row_map, void* stream);
int sgf_combine_fc_bwd_mapped(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d, int32_t classes,
                              float a, float b, int32_t dtype, void* dx1, int64_t ld1, void* dx2, int64_t ld2,
                              const int32_t* row_map, void* stream);
/* sgf_combine_fc_bwd / _bwd_mapped (row_map may be null) that also leaves the logits' gradient in the storage dtype:
 * g_out[n, 16 ceil(classes / 16)] bf16, zero-padded, in the module's row order (row j = row row_map[j] of dlogits) — the operand
 * of dW = a g^T x1 + b g^T x2 (sgf_gram) — written from the matrix-core fragments the kernel holds anyway instead of by a
 * cast and a pad pass over dlogits.  bf16 storage, classes <= 64; g_out 16-byte aligned, ldg % 8 == 0. */
int sgf_combine_fc_bwd_g(const float* dlogits, int64_t lddl, const float* w, int64_t n, int32_t d, int32_t classes, float a,
                         float b, int32_t dtype, void* dx1, int64_t ld1, void* dx2, int64_t ld2, const int32_t* row_map,
                         void* g_out, int64_t ldg, void* stream);

/* ------------------------------------------------------------------------------------------
 * T6 / K8 — the dense half of a GCN layer as one streaming pass.   Replaces, for large/ours.py:36-40,87-88
 *     x = self.W(gcn_conv(...));  x = self.bns[i](x)        (the Linear and BatchNorm1d's batch statistics)
 *   sgf_gcn_epilogue_stats : y[n, d_out] = a[n, d_in] W^T + bias, W [d_out, d_in] row-major in the storage
 *                            dtype, bias fp32 or null; fp32 accumulation on the matrix cores, y rounded to the
 *                            storage dtype.  With stats != null also, in the same pass,
 *                              stats[j] = sum_i (y_ij - shift_j),  stats[d_out + j] = sum_i (y_ij - shift_j)^2
 *                            of the ROUNDED y (what sgf_colstats(y, shift) returns, without re-reading y);
 *                            shift fp32 [d_out] or null.  workspace: sgf_gcn_epilogue_workspace_bytes (only
 *                            used with stats).  The normalise / ReLU / dropout / residual half of the layer
 *                            is sgf_bn_apply (+ sgf_dropout_*).
 *   sgf_gcn_epilogue_dx    : dx[n, d_in] = dy[n, d_out] W   (the Linear's input gradient; dW, db: sgf_gram).
 * W stays resident in LDS (one 8-wave block per CU), every wave streams 32-row tiles on its own; see
 * csrc/rowgemm.hip.  Implemented for bf16 storage and d_in == d_out in {64, 128, 256}
 * (sgf_gcn_epilogue_supported); rows of a / y / dy / dx / W must be 16-byte aligned.  Anything else: a
 * library GEMM + sgf_colstats.
 * ------------------------------------------------------------------------------------------ */
int32_t sgf_gcn_epilogue_supported(int32_t d_in, int32_t d_out, int32_t dtype);
size_t sgf_gcn_epilogue_workspace_bytes(int64_t n, int32_t d_out);
int sgf_gcn_epilogue_stats(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                           int64_t n, int32_t d_in, int32_t d_out, int32_t dtype, void* y, int64_t ldy,
                           const float* shift, float* stats, void* workspace, size_t workspace_bytes,
                           void* stream);
/* Both input gradients of the two-operand Linear W [a1 | a2] (large/ours.py:36-38) from one read of dy out of HBM:
 * dx1 = dy w1, dx2 = dy w2 with w1 = W[:, :d], w2 = W[:, d:] (same leading dimension ldw).  pair != 0: ONE launch whose
 * workgroups come in pairs on one XCD walking the same row tiles (the second read of a tile is an L2 hit); pair = 0 or
 * fp32 storage: two sgf_gcn_epilogue_dx launches.  Results are identical either way. */
int sgf_gcn_epilogue_dx2(const void* dy, int64_t lddy, const void* w1, const void* w2, int64_t ldw, int64_t n, int32_t d,
                         int32_t dtype, void* dx1, int64_t lddx1, void* dx2, int64_t lddx2, int32_t pair, void* stream);

/* The same two gradients with the gradient of x0 = layer_[0] ACCUMULATED IN PLACE across the layers (large/ours.py:83-93: x0
 * feeds every layer's Linear and its residual; the layers' backward nodes run last-to-first):
 *     dy      = dz W[:, :d]
 *     acc_out = dz W[:, d:] + gadd + acc_in        gadd (nullable): this layer's residual gradient of x0;
 *                                                  acc_in (nullable): the running sum the layers before left
 * One launch, w = [W1 | W2] as the Linear stores it (ldw >= 2 d).  d = 256 runs as PAIRS of workgroups that each produce
 * half of the columns of BOTH results (balanced: the pair reads dz once through L2 and equal shares of the addends).  bf16
 * storage, d in {64, 128, 256}; acc_out may alias acc_in (same rows, same columns are read before they are written).
 * Replaces sgf_gcn_epilogue_dx2 + the final k-operand sgf_sum_n: 14 T instead of 16 T of traffic for three layers. */
int32_t sgf_gcn_epilogue_dx2_acc_supported(int32_t d, int32_t dtype);
int sgf_gcn_epilogue_dx2_acc(const void* dz, int64_t lddz, const void* w, int64_t ldw, int64_t n, int32_t d, int32_t dtype,
                             void* dy, int64_t lddy, const void* gadd, int64_t ldg, const void* acc_in, int64_t ldai,
                             void* acc_out, int64_t ldao, void* stream);
int sgf_gcn_epilogue_dx(const void* dy, int64_t lddy, const void* w, int64_t ldw, int64_t n, int32_t d_in,
                        int32_t d_out, int32_t dtype, void* dx, int64_t lddx, void* stream);
/* GraphConvLayer with use_init (large/ours.py:36-38):  y = [a1 | a2] W^T + bias,  W = [W1 | W2] of width 2 d.
 *   sgf_gcn_epilogue_partial   : partial = a1 W1^T + bias, rounded to the storage dtype and kept in the matrix
 *                                cores' accumulator layout (an opaque buffer of sgf_gcn_epilogue_partial_bytes,
 *                                16-byte aligned; lane-contiguous 16-byte stores, no transposition);
 *   sgf_gcn_epilogue_stats_add : y = a2 W2^T + partial (+ the column sums, as sgf_gcn_epilogue_stats) — the addend
 *                                returns to the registers it left from and the sum is rounded once: the arithmetic
 *                                of a library GEMM with beta = 1 on the rounded first product.
 * W1 / W2 are passed as pointers into W with ldw = 2 d.  5 [n, d] passes in total where
 * GEMM + GEMM(beta = 1) + sgf_colstats take 6. */
/* the other half of the layer (SURVEY.md §7 K8 "gcn_epilogue_apply"): BN normalise -> ReLU -> + residual, one
 * elementwise pass; the same entry as sgf_bn_apply under the name the layer-level API uses */
int sgf_gcn_epilogue_apply(const void* y, int64_t ldy, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const void* res, int64_t ldr, int32_t relu, int64_t n, int32_t d,
                           int32_t dtype, void* out, int64_t ldo, void* stream);
/* Backward of the GCN layer's dense half in ONE launch (large/ours.py:36-40 + :87-93 differentiated; bf16 storage,
 * d in {64, 128, 256}).  Forward:  z = [Ax | x0] W^T + b,  out = [relu](BatchNorm(z)) [+ x0].  Given gy = d out, z, the
 * BatchNorm coefficients and the reduced statistics stats = [sum g' | sum g' xhat] of sgf_bn_bwd_stats:
 *     dz    = sgf_bn_bwd_apply(gy, z, ...)                 written row-major (the weight gradients read it: sgf_gram)
 *     dy    = dz W[:, :d]                                  gradient of Ax, row-major
 *     dx0'  = dx0 + dz W[:, d:] (+ gy when add_gy)         gradient of x0, ACCUMULATED over the layers of the branch:
 * acc_in (null for the first call) / acc_out are opaque buffers of sgf_gcn_epilogue_partial_bytes(n, d) holding the running
 * sum in the matrix cores' accumulator layout (bf16); the LAST call passes dx0 != null (and acc_out = null) and gets the
 * total row-major.  Exactly one of acc_out / dx0 is non-null.  gy and z leave HBM once: the workgroups that produce the
 * 2 d virtual output columns of one row tile (128 columns each) are launched 8 apart, i.e. on one XCD, and share the tiles
 * through its L2 (placement affects traffic only); with a workspace (sgf_gcn_bn_bwd_dx_workspace_bytes, may be null) those
 * workgroups meet once per tile — a bounded wait on device-scope counters, timing only — so that they stay inside the L2's
 * window.  inv_n = 1 / (global row count); training = 0: running statistics (no mean terms). */
int32_t sgf_gcn_bn_bwd_dx_supported(int32_t d, int32_t dtype);
size_t sgf_gcn_bn_bwd_dx_workspace_bytes(int64_t n, int32_t d);   /* arrival counters of the per-tile rendezvous (optional) */
int sgf_gcn_bn_bwd_dx(const void* gy, int64_t ldg, const void* z, int64_t ldz, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int32_t relu, const float* stats, float inv_n,
                      int32_t training, const void* w, int64_t ldw, int64_t n, int32_t d, int32_t dtype, void* dz,
                      int64_t lddz, void* dy, int64_t lddy, const void* acc_in, void* acc_out, size_t acc_bytes,
                      void* dx0, int64_t lddx0, int32_t add_gy, void* workspace, size_t workspace_bytes, void* stream);
/* The same two-operand Linear in ONE pass over a1 and a2 (bf16 storage, d = d_in = d_out in {64, 128, 256}; W [d, 2 d]):
 *   y = [a1 | a2] W^T + bias  [+ stats, as sgf_gcn_epilogue_stats].   d <= 128: W (64 KiB) is resident in LDS whole.
 *   d = 256: W is 256 KiB, more than a CU's LDS, so the launch is PAIRED — workgroups b and b + 8 (one XCD) walk the same
 *   row tiles, each producing one half of the output columns from its 128 rows of W; the second read of a tile is served
 *   by the XCD's L2.  Placement and timing affect only the traffic, never the result.  Rounding: the sum of both products
 *   is rounded ONCE (sgf_gcn_epilogue_partial / _stats_add round the first product to the storage dtype first).
 * workspace: sgf_gcn_epilogue_workspace_bytes(n, d), only when stats != null. */
int32_t sgf_gcn_epilogue_cat_supported(int32_t d, int32_t dtype);
int sgf_gcn_epilogue_cat(const void* a1, int64_t lda1, const void* a2, int64_t lda2, const void* w, int64_t ldw,
                         const float* bias, int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy, const float* shift,
                         float* stats, void* workspace, size_t workspace_bytes, void* stream);
size_t sgf_gcn_epilogue_partial_bytes(int64_t n, int32_t d_out);
size_t sgf_gcn_epilogue_dtype_partial_bytes(int64_t n, int32_t d_out, int32_t dtype);   /* fp32 storage: n * d_out * 4 */
int sgf_gcn_epilogue_partial(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t n,
                             int32_t d_in, int32_t d_out, int32_t dtype, void* partial, size_t partial_bytes,
                             void* stream);
int sgf_gcn_epilogue_stats_add(const void* a, int64_t lda, const void* w, int64_t ldw, const void* partial,
                               size_t partial_bytes, int64_t n, int32_t d_in, int32_t d_out, int32_t dtype,
                               void* y, int64_t ldy, const float* shift, float* stats, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T4 / K10 — both input stems from ONE read of the node features.   Replaces the first line of each branch,
 * large/ours.py:77 (GraphConv: x = self.fcs[0](x), followed by BatchNorm1d :79) and :198 (TransConv):
 *     y0 = x W0^T + b0  [+ stats0 = column sums of the rounded y0, as sgf_gcn_epilogue_stats],   y1 = x W1^T + b1
 * x [n, d_in] in the storage dtype with d_in % 4 == 0, d_in <= 128 (rows 8-byte aligned: 100 features = 200 bytes);
 * W0, W1 [d_out, d_in] in the storage dtype, biases fp32 or null; w1 / y1 null = one output.  Same per-wave
 * streaming skeleton as sgf_gcn_epilogue_stats with both weight matrices resident in LDS (csrc/rowgemm.hip).
 * bf16 storage, d_out in {64, 128, 256} (sgf_stem_pair_supported); workspace (with stats0 only):
 * sgf_gcn_epilogue_workspace_bytes(n, d_out).
 * ------------------------------------------------------------------------------------------ */
int32_t sgf_stem_pair_supported(int32_t d_in, int32_t d_out, int32_t dtype);
int sgf_stem_pair(const void* x, int64_t ldx, int64_t n, int32_t d_in, const void* w0, int64_t ldw0,
                  const float* bias0, const void* w1, int64_t ldw1, const float* bias1, int32_t d_out, int32_t dtype,
                  void* y0, int64_t ldy0, void* y1, int64_t ldy1, const float* shift0, float* stats0, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * T4 — the general Linear: every shape the streaming kernels above do not take (csrc/gemm.hip; r05).   Replaces
 * F.linear / addmm / matmul for large/ours.py:77,:198 with input widths such as f = 1433 (Cora, medium/ours.py:133-145) or
 * widths that are not multiples of 4, the multi-head projections :123-126, medium/models.py GCNConv's x @ weight, the dX of
 * any of them — no Linear of the path leaves this library, whatever its shape:
 *     c[i, j] = alpha * sum_k A(i, k) B(k, j) + bias[j] + beta * addend[i, j]          i < m, j < n
 *     A(i, k) = a[i * a_rs + k * a_cs],   B(k, j) = b[k * b_rs + j * b_cs]             (element strides: any transposition)
 * y = x W^T + b : a = x (a_rs = ldx, a_cs = 1), b = W (b_rs = 1, b_cs = ldw);   dx = g W : b = W (b_rs = ldw, b_cs = 1).
 * Any m, n, k, any alignment.  Both operands bf16: bf16 matrix cores (exact products, fp32 sums); otherwise the exact-fp32
 * matrix cores (a bf16 operand is widened).  alpha_dev (nullable): a DEVICE scalar multiplied into alpha; bias fp32 [n] or
 * NULL; addend [m, n] (ldadd, add_dtype) or NULL, may alias c.  64 x 64 output tile per workgroup; not split over k
 * (node reductions such as dW = g^T x belong on sgf_gram).
 * ------------------------------------------------------------------------------------------ */
int sgf_gemm(const void* a, int64_t a_rs, int64_t a_cs, int32_t a_dtype, const void* b, int64_t b_rs, int64_t b_cs,
             int32_t b_dtype, int64_t m, int32_t n, int64_t k, float alpha, const float* alpha_dev, const float* bias,
             float beta, const void* addend, int64_t ldadd, int32_t add_dtype, void* c, int64_t ldc, int32_t c_dtype,
             void* stream);

/* ------------------------------------------------------------------------------------------
 * T7 — branch combine alone.   large/ours.py:269-270:  y = gw * x2 + (1 - gw) * x1.
 * (generic axpby: y = a * x1 + b * x2; the 'cat' aggregate is a plain copy done by the caller.)
 * ------------------------------------------------------------------------------------------ */
int sgf_axpby(const void* x1, int64_t ld1, float a, const void* x2, int64_t ld2, float b,
              int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dropout.   Replaces F.dropout(x, p, training=True) at large/ours.py:81,92,202,216 (and the
 * residual add that follows it in GraphConv, :92-93):   y = x * keep / (1 - p) [+ res]
 * keep ~ Bernoulli(1 - p) is a pure function of (seed, logical element index row*d + col) —
 * Philox4x32-10 — so NO mask is stored: the backward is the same call on dL/dy with the same seed
 * and res = NULL.  (The RNG stream differs from ATen's, as the CPU and CUDA streams of the
 * reference differ from each other; parity with dropout on is statistical.)  d % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
int sgf_dropout(const void* x, int64_t ldx, const void* res, int64_t ldr, float p, uint64_t seed,
                int64_t n, int32_t d, int32_t dtype, void* y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------------------------------
 * N4 — the trainer's loss.   Replaces large/main.py:139-141
 *     out = F.log_softmax(out, dim=1);  loss = NLLLoss()(out[train_idx], label.squeeze(1)[train_idx])
 * (5 ATen kernels over [N, C] and [M, C] temporaries; its nll_loss forward / backward kernels alone
 * cost 4.2 ms per step at ogbn-products scale) as one pass over the M training rows:
 *     sgf_nll_fwd : loss_sum[0] = - sum_j log_softmax(logits[idx[j]])[labels[idx[j]]]   (fp32; the
 *                   caller divides by M, or by the GLOBAL count when node-sharded)
 *     sgf_nll_bwd : dlogits = 0 everywhere except rows idx[j], where
 *                   dlogits = gout[0] * inv_denom * (softmax(logits[row]) - onehot(label))
 * logits / dlogits: [n, c] storage dtype with leading dims ldl / ldd; labels int64 [n] (indexed by
 * NODE id); idx int64 [m] distinct rows; gout fp32 [1] on the device.  Labels outside [0, c) add
 * nothing to the loss and get a zero gradient row (nn.NLLLoss's ignore_index).  Deterministic (per-block
 * partials, fixed-order sum).
 * ------------------------------------------------------------------------------------------ */
size_t sgf_nll_workspace_bytes(int64_t m);
int sgf_nll_fwd(const void* logits, int64_t ldl, int64_t n, int32_t c, int32_t dtype,
                const int64_t* labels, const int64_t* idx, int64_t m, float* loss_sum,
                void* workspace, size_t workspace_bytes, void* stream);
int sgf_nll_bwd(const void* logits, int64_t ldl, int64_t n, int32_t c, int32_t dtype,
                const int64_t* labels, const int64_t* idx, int64_t m, const float* gout,
                float inv_denom, void* dlogits, int64_t ldd, void* stream);

/* y = sum_i xs[i] for 1 <= k <= 8 equally shaped [n, d] operands (fp32 accumulation in operand
 * order).  Replaces the pairwise gradient accumulation autograd performs for a tensor with several
 * consumers — GraphConv's x0 (large/ours.py:83-93) receives one gradient per layer from the
 * [. | x0] Linear, one per layer from the residual and one from the first SpMM.
 * xs / lds are HOST arrays of k device pointers / leading dimensions. */
int sgf_sum_n(const void* const* xs_host, const int64_t* lds_host, int32_t k, int64_t n, int32_t d,
              int32_t dtype, void* y, int64_t ldy, void* stream);

/* Column sum out[j] = sum_n x[n,j] for ANY 1 <= d <= 256 (no multiple-of-4 requirement): the bias
 * gradient of the output layer, large/ours.py:275 under autograd, where d = number of classes
 * (47 for ogbn-products).  fp32 result, deterministic two-stage reduction. */
size_t sgf_colsum_workspace_bytes(int64_t n, int32_t d);
int sgf_colsum(const void* x, int64_t ldx, int64_t n, int32_t d, int32_t dtype, float* out,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY.md §8b / §8e — the collectives of the node-sharded run (csrc/comm.hip): thin wrappers over RCCL, which is loaded
 * with dlopen() at the first call (libsgf.so has no link-time dependency on librccl.so).  The reference has no multi-GPU
 * path; WHAT is exchanged follows from the kernels' partial-sum layouts above: the attention statistics and BatchNorm sums
 * and the parameter gradients (all-reduce, fp32, in place), the SpMM operand's rows — whole shards (all-gather) or the
 * cut-edge rows of a halo plan (all-to-all with per-peer byte ranges).  One process per GPU; the communicator is created
 * on the calling thread's CURRENT device.  The *_host arrays are host memory (per-peer offsets / sizes in bytes, length =
 * world); everything else is device memory.  The Python host side (sgformer_amd/dist.py) issues the same collectives
 * through torch.distributed (backend `nccl` = RCCL); these entries are for a consumer binding the library from C.
 *   sgf_comm_unique_id  : rank 0 fills id_host (sgf_comm_unique_id_bytes() = 128 bytes) and hands it to the other ranks
 *                         by any out-of-band means (the launcher's rendezvous);
 *   sgf_comm_create     : blocks until all `world` ranks have called it with the same id.
 * ------------------------------------------------------------------------------------------ */
int32_t sgf_comm_available(void);                 /* 1 if librccl.so and its entry points could be loaded */
int32_t sgf_comm_unique_id_bytes(void);
int sgf_comm_unique_id(void* id_host);
int sgf_comm_create(void** comm_out, int32_t world, int32_t rank, const void* id_host);
int sgf_comm_destroy(void* comm);
int sgf_comm_all_reduce_f32(void* comm, float* buf, int64_t count, void* stream);
int sgf_comm_all_gather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int sgf_comm_all_to_all(void* comm, const void* send, const int64_t* send_offset_host, const int64_t* send_bytes_host,
                        void* recv, const int64_t* recv_offset_host, const int64_t* recv_bytes_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGF_H_ */
