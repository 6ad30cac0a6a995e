// attn_small.hip — T3: the d x d algebra between the attention's two node passes, as ONE library entry each way.
//
// The one-head attention straight from the layer input (include/sgf.h, sgf_attn_h_*) needs, between its Gram pass
// G = h^T h, s = sum_n h_n and its apply pass out = (h M + m) / (h w + beta), the small dense algebra of
// large/ours.py:131-149 with the projections Q = h Wq^T + bq, K = ..., V = ... folded in:
//     K^T V, K^T 1, ||Q||_F, ||K||_F  ->  M = c Wq^T (K^T V) + N Wv^T,  m = c bq (K^T V) + N bv,
//                                         w = c Wq^T (K^T 1),           beta = c bq . (K^T 1) + N,   c = 1 / (||Q|| ||K||)
// Written on AUGMENTED operands (ht = [h | 1]):  Gt = [[G, s], [s^T, n]],  Wqk = [[wq | bq], [wk | bk]],  Vx = [[wv^T, 0], [bv^T, 1]]:
//     PG = Wqk Gt,  (||Q||^2, ||K||^2) = row-block sums of PG * Wqk,  SZ = PG[d:] Vx = [K^T V | K^T 1],  T = Wq~^T SZ,
//     Out = c T + N Vx      (M, m, w, beta are the blocks of Out)
// and its hand-derived backward (Gt symmetric):
//     gc = <gOut, T>,  g_s = -gc c / (2 ssq),  gSZ = c Wq~ gOut,  gPG = g_s (.) Wqk;  gPG[d:] += gSZ Vx^T,
//     gWqk = g_s (.) PG + gPG Gt;  gWqk[:d] += c SZ gOut^T,  gGt = Wqk^T gPG,  gVx = N gOut + PG[d:]^T gSZ.
// Forward = 6 launches, backward = 9, all from this file: three / six products on the exact-fp32 matrix cores (gemm.hip) and
// small element-wise / one-block reduction kernels — where r04 issued 12 + 20 ATen / rocBLAS launches from Python.
// Everything is fp32; sums inside one block in a fixed order (deterministic).
#include "common.h"

namespace sgf {

int gemm_launch(const void* a, int64_t a_rs, int64_t a_cs, int a_dtype, const void* b, int64_t b_rs, int64_t b_cs, int b_dtype,
                int64_t m, int n, int64_t k, float alpha, const float* alpha_dev, const float* bias, float beta,
                const void* addend, int64_t ldadd, int add_dtype, void* c, int64_t ldc, int c_dtype, hipStream_t st);

namespace {

constexpr int kSmThreads = 256;
constexpr int kRedThreads = 1024;

struct SmallLayout {        // offsets in floats into the `saved` buffer
  int64_t gt, wqk, vx, pg, sz, t, scal, total;
};
struct SmallWork {          // offsets in floats into the backward workspace
  int64_t gout, gsz, gpg, gwqk, ggt, gvx, gs, total;
};

// leading dimensions: rows padded to a multiple of 4 floats, so that sgf_gemm stages these operands with 16-byte loads
__host__ __device__ inline int64_t ld4(int64_t n) { return (n + 3) / 4 * 4; }

SmallLayout small_layout(int D, int d) {
  const int64_t E = ld4(D + 1), d1 = ld4(d + 1);      // (sizes in the PADDED leading dimensions; the pad columns stay zero)
  SmallLayout L;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
  L.gt = take(E * E);
  L.wqk = take(2 * d * E);
  L.vx = take(E * d1);
  L.pg = take(2 * d * E);
  L.sz = take(d * d1);
  L.t = take(E * d1);
  L.scal = take(4);          // ssq_q, ssq_k, c
  L.total = o;
  return L;
}
SmallWork small_work(int D, int d) {
  const int64_t E = ld4(D + 1), d1 = ld4(d + 1);
  SmallWork W;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
  W.gout = take(E * d1);
  W.gsz = take(d * d1);
  W.gpg = take(2 * d * E);
  W.gwqk = take(2 * d * E);
  W.ggt = take(E * E);
  W.gvx = take(E * d1);
  W.gs = take(4);
  W.total = o;
  return W;
}

struct PackArgs {
  const float* G; int64_t ldg;
  const float* s;
  float n_rows;
  const float *wq, *bq, *wk, *bk, *wv, *bv;     // wv null: V = h (identity, zero bias; needs D == d)
  int64_t ldwq, ldwk, ldwv;
  int D, d;
  float *Gt, *Wqk, *Vx;
};

__global__ __launch_bounds__(kSmThreads) void k_small_pack(PackArgs p) {
  const int E = p.D + 1, d1 = p.d + 1;
  const int LE = static_cast<int>(ld4(E)), L1 = static_cast<int>(ld4(d1));
  const int64_t nG = static_cast<int64_t>(LE) * LE, nW = static_cast<int64_t>(2 * p.d) * LE, nV = static_cast<int64_t>(LE) * L1;
  for (int64_t x = static_cast<int64_t>(blockIdx.x) * kSmThreads + threadIdx.x; x < nG + nW + nV;
       x += static_cast<int64_t>(gridDim.x) * kSmThreads) {
    if (x < nG) {
      const int i = static_cast<int>(x / LE), j = static_cast<int>(x % LE);
      float v = 0.f;                                   // (pad rows / columns: zero)
      if (i < E && j < E) {
        if (i < p.D && j < p.D) v = p.G[i * p.ldg + j];
        else if (i < p.D) v = p.s[i];
        else if (j < p.D) v = p.s[j];
        else v = p.n_rows;
      }
      p.Gt[x] = v;
    } else if (x < nG + nW) {
      const int64_t y = x - nG;
      const int r = static_cast<int>(y / LE), c = static_cast<int>(y % LE);
      float v = 0.f;
      if (c < E) {
        if (r < p.d) v = c < p.D ? p.wq[r * p.ldwq + c] : (p.bq ? p.bq[r] : 0.f);
        else v = c < p.D ? p.wk[(r - p.d) * p.ldwk + c] : (p.bk ? p.bk[r - p.d] : 0.f);
      }
      p.Wqk[y] = v;
    } else {
      const int64_t y = x - nG - nW;
      const int i = static_cast<int>(y / L1), j = static_cast<int>(y % L1);
      float v = 0.f;
      if (i < E && j < d1) {
        if (j < p.d) {
          if (i < p.D) v = p.wv ? p.wv[j * p.ldwv + i] : (i == j ? 1.f : 0.f);
          else v = (p.wv && p.bv) ? p.bv[j] : 0.f;
        } else {
          v = i == p.D ? 1.f : 0.f;
        }
      }
      p.Vx[y] = v;
    }
  }
}

// fixed-order sum over one block of kRedThreads threads (each holds its partial): result in every thread
__device__ __forceinline__ float block_sum_fixed(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = group_sum<64>(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}

// scal = (||Q||_F^2, ||K||_F^2, 1 / (||Q|| ||K||)) from PG (.) Wqk, one block
__global__ __launch_bounds__(kRedThreads) void k_small_ssq(const float* __restrict__ PG, const float* __restrict__ Wqk,
                                                            int64_t half, float* __restrict__ scal) {
  __shared__ float red[kRedThreads / 64];
  float q = 0.f, k = 0.f;
  for (int64_t x = threadIdx.x; x < half; x += kRedThreads) {
    q = fmaf(PG[x], Wqk[x], q);
    k = fmaf(PG[half + x], Wqk[half + x], k);
  }
  const float sq = block_sum_fixed(q, red);
  const float sk = block_sum_fixed(k, red);
  if (threadIdx.x == 0) {
    scal[0] = sq;
    scal[1] = sk;
    scal[2] = 1.0f / sqrtf(sq * sk);
  }
}

// Out = c T + n_total Vx, handed out in blocks: M [D, d] (ldm), m [d], w [D], beta [1]
__global__ __launch_bounds__(kSmThreads) void k_small_out(const float* __restrict__ T, const float* __restrict__ Vx,
                                                          const float* __restrict__ scal, float n_total, int D, int d,
                                                          float* __restrict__ M, int64_t ldm, float* __restrict__ m,
                                                          float* __restrict__ w, float* __restrict__ beta) {
  const int d1 = d + 1, L1 = static_cast<int>(ld4(d1));
  const int64_t total = static_cast<int64_t>(D + 1) * d1;
  const float c = scal[2];
  for (int64_t x = static_cast<int64_t>(blockIdx.x) * kSmThreads + threadIdx.x; x < total;
       x += static_cast<int64_t>(gridDim.x) * kSmThreads) {
    const int i = static_cast<int>(x / d1), j = static_cast<int>(x % d1);
    const int64_t q = static_cast<int64_t>(i) * L1 + j;
    const float v = fmaf(c, T[q], n_total * Vx[q]);
    if (i < D && j < d) M[i * ldm + j] = v;
    else if (i == D && j < d) m[j] = v;
    else if (i < D) w[i] = v;
    else beta[0] = v;
  }
}

// gOut from the reduced gradients (dM [D, d], dw [D], dm [d], dbeta [1]); gc = <gOut, T>; gs = -gc c / (2 ssq).  One block:
// a thread owns a column (and, with threads to spare, every R-th row of it) and walks the rows (coalesced).
__global__ __launch_bounds__(kRedThreads) void k_small_bwd_prep(const float* __restrict__ dM, int64_t lddm,
                                                                const float* __restrict__ dw, const float* __restrict__ dm,
                                                                const float* __restrict__ dbeta, const float* __restrict__ T,
                                                                const float* __restrict__ scal, int D, int d,
                                                                float* __restrict__ gOut, float* __restrict__ gs) {
  __shared__ float red[kRedThreads / 64];
  const int d1 = d + 1, L1 = static_cast<int>(ld4(d1)), LE = static_cast<int>(ld4(D + 1));
  float acc = 0.f;
  // R row groups share a column when the block has threads to spare (d = 256: 260 columns, 3 groups): thread (q, j) walks
  // rows q, q + R, ... — independent loads, four in flight; the block sum below adds the threads in a fixed order
  const int R = L1 < kRedThreads ? kRedThreads / L1 : 1;
  const int q0 = static_cast<int>(threadIdx.x) / L1, j0 = static_cast<int>(threadIdx.x) % L1;
  if (q0 < R) {
    for (int j = j0; j < L1; j += (R > 1 ? L1 : kRedThreads)) {
#pragma unroll 4
      for (int i = q0; i < LE; i += R) {
        float v = 0.f;
        if (i <= D && j < d1) {
          if (i < D && j < d) v = dM[i * lddm + j];
          else if (i == D && j < d) v = dm[j];
          else if (i < D) v = dw[i];
          else v = dbeta[0];
        }
        const int64_t q = static_cast<int64_t>(i) * L1 + j;
        gOut[q] = v;
        acc = fmaf(v, T[q], acc);
      }
    }
  }
  const float gc = block_sum_fixed(acc, red);
  if (threadIdx.x == 0) {
    const float c = scal[2];
    gs[0] = (gc * c * -0.5f) / scal[0];
    gs[1] = (gc * c * -0.5f) / scal[1];
  }
}

// gPG = g_s (.) Wqk,  gWqk = g_s (.) PG   (g_s = gs[0] on the Wq rows, gs[1] on the Wk rows)
__global__ __launch_bounds__(kSmThreads) void k_small_bwd_scale(const float* __restrict__ Wqk, const float* __restrict__ PG,
                                                                const float* __restrict__ gs, int64_t half,
                                                                float* __restrict__ gPG, float* __restrict__ gWqk) {
  const float g0 = gs[0], g1 = gs[1];
  for (int64_t x = static_cast<int64_t>(blockIdx.x) * kSmThreads + threadIdx.x; x < 2 * half;
       x += static_cast<int64_t>(gridDim.x) * kSmThreads) {
    const float g = x < half ? g0 : g1;
    gPG[x] = g * Wqk[x];
    gWqk[x] = g * PG[x];
  }
}

struct UnpackArgs {
  const float *gGt, *gWqk, *gVx;
  int D, d;
  float* Dsym; int64_t lddd;     // [D, D] = gG + gG^T
  float* ds;                     // [D]    = gGt[:D, D] + gGt[D, :D]
  float *gwq, *gbq, *gwk, *gbk, *gwv, *gbv;   // [d, D] / [d]; any may be null
  int64_t ldgq, ldgk, ldgv;
};

__global__ __launch_bounds__(kSmThreads) void k_small_bwd_unpack(UnpackArgs p) {
  const int E = p.D + 1, d1 = p.d + 1;
  const int LE = static_cast<int>(ld4(E)), L1 = static_cast<int>(ld4(d1));
  const int64_t nD = static_cast<int64_t>(p.D) * E;            // Dsym rows + the ds column
  const int64_t nW = static_cast<int64_t>(2 * p.d) * E;
  const int64_t nV = static_cast<int64_t>(E) * p.d;
  for (int64_t x = static_cast<int64_t>(blockIdx.x) * kSmThreads + threadIdx.x; x < nD + nW + nV;
       x += static_cast<int64_t>(gridDim.x) * kSmThreads) {
    if (x < nD) {
      const int i = static_cast<int>(x / E), j = static_cast<int>(x % E);
      const float v = p.gGt[static_cast<int64_t>(i) * LE + j] + p.gGt[static_cast<int64_t>(j) * LE + i];
      if (j < p.D) p.Dsym[i * p.lddd + j] = v;
      else p.ds[i] = v;
    } else if (x < nD + nW) {
      const int64_t y = x - nD;
      const int r = static_cast<int>(y / E), c = static_cast<int>(y % E);
      const float v = p.gWqk[static_cast<int64_t>(r) * LE + c];
      if (r < p.d) {
        if (c < p.D) { if (p.gwq) p.gwq[r * p.ldgq + c] = v; }
        else if (p.gbq) p.gbq[r] = v;
      } else {
        if (c < p.D) { if (p.gwk) p.gwk[(r - p.d) * p.ldgk + c] = v; }
        else if (p.gbk) p.gbk[r - p.d] = v;
      }
    } else {
      const int64_t y = x - nD - nW;
      const int i = static_cast<int>(y / p.d), j = static_cast<int>(y % p.d);     // gVx[i, j] -> gwv[j, i] / gbv[j]
      const float v = p.gVx[static_cast<int64_t>(i) * L1 + j];
      if (i < p.D) { if (p.gwv) p.gwv[j * p.ldgv + i] = v; }
      else if (p.gbv) p.gbv[j] = v;
    }
  }
}

int small_grid(int64_t total) {
  int64_t b = (total + kSmThreads - 1) / kSmThreads;
  if (b > 4 * kNumCU) b = 4 * kNumCU;
  return b < 1 ? 1 : static_cast<int>(b);
}

}  // namespace
}  // namespace sgf

extern "C" size_t sgf_attn_h_small_saved_bytes(int32_t d_in, int32_t d_out) {
  if (d_in <= 0 || d_out <= 0) return 0;
  return static_cast<size_t>(sgf::small_layout(d_in, d_out).total) * sizeof(float);
}

extern "C" size_t sgf_attn_h_small_workspace_bytes(int32_t d_in, int32_t d_out) {
  if (d_in <= 0 || d_out <= 0) return 0;
  return static_cast<size_t>(sgf::small_work(d_in, d_out).total) * sizeof(float);
}

extern "C" int sgf_attn_h_small_fwd(const float* G, int64_t ldg, const float* s, float n_rows, float n_total, const float* wq,
                                    const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                                    int64_t ldw, int32_t d_in, int32_t d_out, float* M, int64_t ldm, float* m, float* w,
                                    float* beta, void* saved, size_t saved_bytes, void* stream) {
  using namespace sgf;
  SGF_REQUIRE(d_in > 0 && d_out > 0 && d_in <= 4096 && d_out <= 4096, SGF_E_INVALID, "sgf_attn_h_small_fwd: widths %d -> %d", d_in,
              d_out);
  SGF_REQUIRE(G && s && wq && wk && M && m && w && beta && saved, SGF_E_INVALID, "sgf_attn_h_small_fwd: null pointer");
  SGF_REQUIRE(wv || d_in == d_out, SGF_E_INVALID, "sgf_attn_h_small_fwd: V = h (wv null) needs d_in == d_out");
  SGF_REQUIRE(ldg >= d_in && ldw >= d_in && ldm >= d_out, SGF_E_INVALID, "sgf_attn_h_small_fwd: leading dimension too small");
  const SmallLayout L = small_layout(d_in, d_out);
  SGF_REQUIRE(saved_bytes >= static_cast<size_t>(L.total) * sizeof(float), SGF_E_WORKSPACE,
              "sgf_attn_h_small_fwd: saved buffer of %zu bytes, need %zu", saved_bytes, static_cast<size_t>(L.total) * sizeof(float));
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* base = static_cast<float*>(saved);
  // E, d1: the PADDED extents (multiples of 4).  Pad rows / columns of every operand are zero (k_small_pack, k_small_bwd_prep)
  // and every product below maps zero pads to zero pads, so the padded products equal the logical ones exactly.
  const int D = d_in, d = d_out, E = static_cast<int>(ld4(D + 1)), d1 = static_cast<int>(ld4(d + 1));
  float *Gt = base + L.gt, *Wqk = base + L.wqk, *Vx = base + L.vx, *PG = base + L.pg, *SZ = base + L.sz, *T = base + L.t,
        *scal = base + L.scal;
  PackArgs pa{G, ldg, s, n_rows, wq, bq, wk, bk, wv, bv, ldw, ldw, ldw, D, d, Gt, Wqk, Vx};
  const int64_t npack = static_cast<int64_t>(E) * E + static_cast<int64_t>(2 * d) * E + static_cast<int64_t>(E) * d1;
  hipLaunchKernelGGL(k_small_pack, dim3(small_grid(npack)), dim3(kSmThreads), 0, st, pa);
  SGF_LAUNCH_CHECK();
  int rc = gemm_launch(Wqk, E, 1, SGF_F32, Gt, E, 1, SGF_F32, 2 * d, E, E, 1.f, nullptr, nullptr, 0.f, nullptr, 0, SGF_F32, PG, E,
                       SGF_F32, st);
  if (rc != SGF_OK) return rc;
  hipLaunchKernelGGL(k_small_ssq, dim3(1), dim3(kRedThreads), 0, st, PG, Wqk, static_cast<int64_t>(d) * E, scal);
  SGF_LAUNCH_CHECK();
  rc = gemm_launch(PG + static_cast<int64_t>(d) * E, E, 1, SGF_F32, Vx, d1, 1, SGF_F32, d, d1, E, 1.f, nullptr, nullptr, 0.f,
                   nullptr, 0, SGF_F32, SZ, d1, SGF_F32, st);
  if (rc != SGF_OK) return rc;
  rc = gemm_launch(Wqk, 1, E, SGF_F32, SZ, d1, 1, SGF_F32, E, d1, d, 1.f, nullptr, nullptr, 0.f, nullptr, 0, SGF_F32, T, d1,
                   SGF_F32, st);
  if (rc != SGF_OK) return rc;
  hipLaunchKernelGGL(k_small_out, dim3(small_grid(static_cast<int64_t>(E) * d1)), dim3(kSmThreads), 0, st, T, Vx, scal, n_total,
                     D, d, M, ldm, m, w, beta);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}

extern "C" int sgf_attn_h_small_bwd(const float* dM, int64_t lddm, const float* dw, const float* dm, const float* dbeta,
                                    float n_total, int32_t d_in, int32_t d_out, const void* saved, size_t saved_bytes,
                                    float* dG2, int64_t lddg, float* ds, float* gwq, float* gbq, float* gwk, float* gbk,
                                    float* gwv, float* gbv, int64_t ldgw, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  using namespace sgf;
  SGF_REQUIRE(d_in > 0 && d_out > 0 && d_in <= 4096 && d_out <= 4096, SGF_E_INVALID, "sgf_attn_h_small_bwd: widths %d -> %d", d_in,
              d_out);
  SGF_REQUIRE(dM && dw && dm && dbeta && saved && dG2 && ds && workspace, SGF_E_INVALID, "sgf_attn_h_small_bwd: null pointer");
  SGF_REQUIRE(lddm >= d_out && lddg >= d_in && ldgw >= d_in, SGF_E_INVALID, "sgf_attn_h_small_bwd: leading dimension too small");
  const SmallLayout L = small_layout(d_in, d_out);
  const SmallWork W = small_work(d_in, d_out);
  SGF_REQUIRE(saved_bytes >= static_cast<size_t>(L.total) * sizeof(float), SGF_E_WORKSPACE,
              "sgf_attn_h_small_bwd: saved buffer of %zu bytes, need %zu", saved_bytes, static_cast<size_t>(L.total) * sizeof(float));
  SGF_REQUIRE(workspace_bytes >= static_cast<size_t>(W.total) * sizeof(float), SGF_E_WORKSPACE,
              "sgf_attn_h_small_bwd: workspace of %zu bytes, need %zu", workspace_bytes, static_cast<size_t>(W.total) * sizeof(float));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* base = static_cast<const float*>(saved);
  float* wsp = static_cast<float*>(workspace);
  // E, d1: the PADDED extents (multiples of 4).  Pad rows / columns of every operand are zero (k_small_pack, k_small_bwd_prep)
  // and every product below maps zero pads to zero pads, so the padded products equal the logical ones exactly.
  const int D = d_in, d = d_out, E = static_cast<int>(ld4(D + 1)), d1 = static_cast<int>(ld4(d + 1));
  const float *Gt = base + L.gt, *Wqk = base + L.wqk, *Vx = base + L.vx, *PG = base + L.pg, *SZ = base + L.sz, *T = base + L.t,
              *scal = base + L.scal;
  float *gOut = wsp + W.gout, *gSZ = wsp + W.gsz, *gPG = wsp + W.gpg, *gWqk = wsp + W.gwqk, *gGt = wsp + W.ggt, *gVx = wsp + W.gvx,
        *gs = wsp + W.gs;
  const int64_t dE = static_cast<int64_t>(d) * E;
  hipLaunchKernelGGL(k_small_bwd_prep, dim3(1), dim3(kRedThreads), 0, st, dM, lddm, dw, dm, dbeta, T, scal, D, d, gOut, gs);
  SGF_LAUNCH_CHECK();
  // gSZ = c Wq~ gOut
  int rc = gemm_launch(Wqk, E, 1, SGF_F32, gOut, d1, 1, SGF_F32, d, d1, E, 1.f, scal + 2, nullptr, 0.f, nullptr, 0, SGF_F32, gSZ, d1,
                       SGF_F32, st);
  if (rc != SGF_OK) return rc;
  hipLaunchKernelGGL(k_small_bwd_scale, dim3(small_grid(2 * dE)), dim3(kSmThreads), 0, st, Wqk, PG, gs, dE, gPG, gWqk);
  SGF_LAUNCH_CHECK();
  // gPG[d:] += gSZ Vx^T
  rc = gemm_launch(gSZ, d1, 1, SGF_F32, Vx, 1, d1, SGF_F32, d, E, d1, 1.f, nullptr, nullptr, 1.f, gPG + dE, E, SGF_F32, gPG + dE, E,
                   SGF_F32, st);
  if (rc != SGF_OK) return rc;
  // gWqk += gPG Gt
  rc = gemm_launch(gPG, E, 1, SGF_F32, Gt, E, 1, SGF_F32, 2 * d, E, E, 1.f, nullptr, nullptr, 1.f, gWqk, E, SGF_F32, gWqk, E, SGF_F32,
                   st);
  if (rc != SGF_OK) return rc;
  // gWqk[:d] += c SZ gOut^T
  rc = gemm_launch(SZ, d1, 1, SGF_F32, gOut, 1, d1, SGF_F32, d, E, d1, 1.f, scal + 2, nullptr, 1.f, gWqk, E, SGF_F32, gWqk, E, SGF_F32,
                   st);
  if (rc != SGF_OK) return rc;
  // gGt = Wqk^T gPG
  rc = gemm_launch(Wqk, 1, E, SGF_F32, gPG, E, 1, SGF_F32, E, E, 2 * d, 1.f, nullptr, nullptr, 0.f, nullptr, 0, SGF_F32, gGt, E,
                   SGF_F32, st);
  if (rc != SGF_OK) return rc;
  // gVx = n_total gOut + PG[d:]^T gSZ
  rc = gemm_launch(PG + dE, 1, E, SGF_F32, gSZ, d1, 1, SGF_F32, E, d1, d, 1.f, nullptr, nullptr, n_total, gOut, d1, SGF_F32, gVx, d1,
                   SGF_F32, st);
  if (rc != SGF_OK) return rc;
  UnpackArgs ua{gGt, gWqk, gVx, D, d, dG2, lddg, ds, gwq, gbq, gwk, gbk, gwv, gbv, ldgw, ldgw, ldgw};
  const int64_t nun = static_cast<int64_t>(D) * E + 2 * dE + static_cast<int64_t>(E) * d;
  hipLaunchKernelGGL(k_small_bwd_unpack, dim3(small_grid(nun)), dim3(kSmThreads), 0, st, ua);
  SGF_LAUNCH_CHECK();
  return SGF_OK;
}
