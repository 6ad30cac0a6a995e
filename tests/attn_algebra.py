"""Reference formulations of the attention's d x d algebra — TEST ONLY (moved out of sgformer_amd/ops.py in r05, when the
training step's algebra became sgf_attn_h_small_fwd / _bwd inside libsgf.so).

  * attn_h_small        : include/sgf.h's formulas term by term (large/ours.py:131-149 with the projections folded in);
  * attn_h_packed_fwd / _bwd : the packed forward and its hand-derived backward, the arithmetic csrc/attn_small.hip issues
                               launch by launch (same names for the intermediates).
Plain torch on whatever dtype the operands have (the tests run them in fp64)."""
import torch


def attn_h_small(G, s, n_rows: float, n_total: float, wq, bq, wk, bk, wv, bv, sum_v: bool = False):
    """The d x d algebra of include/sgf.h (sgf_attn_h_*): fp32, tiny, differentiable torch ops.
    G = h^T h, s = sum_n h_n over ALL rows (n_rows of them); weights [d, d_in], biases [d].
    sum_v=False: SGFormer's numerator  q S + N V_n   (large/ours.py:137-138);
    sum_v=True : DIFFormer's numerator q S + sum_l V_l (medium/difformer.py:26-29)."""
    wk_s, wv_s, wq_s = wk @ s, wv @ s, wq @ s
    s0 = wk @ G @ wv.t() + torch.outer(wk_s, bv) + torch.outer(bk, wv_s) + n_rows * torch.outer(bk, bv)
    z0 = wk_s + n_rows * bk
    ssq_q = ((wq @ G) * wq).sum() + 2.0 * torch.dot(bq, wq_s) + n_rows * torch.dot(bq, bq)
    ssq_k = ((wk @ G) * wk).sum() + 2.0 * torch.dot(bk, wk_s) + n_rows * torch.dot(bk, bk)
    c = 1.0 / (torch.sqrt(ssq_q) * torch.sqrt(ssq_k))
    if sum_v:
        M = c * (wq.t() @ s0)
        m = c * (bq @ s0) + wv_s + n_rows * bv
    else:
        M = c * (wq.t() @ s0) + n_total * wv.t()
        m = c * (bq @ s0) + n_total * bv
    w = c * (wq.t() @ z0)
    beta = (c * torch.dot(bq, z0) + n_total).reshape(1)
    return M.contiguous(), m.contiguous(), w.contiguous(), beta.contiguous()


def attn_h_packed_fwd(Gt, Wqk, Wv, n_total: float):
    """ops._attn_h_small_packed(sum_v=False) WITHOUT autograd, keeping what its hand-written backward needs:
    Out = c Wq~^T (Wk~ Gt) Vx + n_total Vx with Vx = [Wv~^T | e_D]; M, m, w, beta are the blocks of Out."""
    d = Wv.shape[0]
    E = Gt.shape[0]
    D = E - 1
    PG = Wqk @ Gt
    ssq = (PG * Wqk).view(2, -1).sum(1)
    c = torch.rsqrt(ssq[0] * ssq[1])
    Vx = torch.zeros((E, d + 1), dtype=Gt.dtype, device=Gt.device)
    Vx[:, :d] = Wv.t()
    Vx[D, d:].fill_(1.0)                                 # (fill_, not `= 1.0`: a Python scalar assignment is a syncing H2D copy)
    SZ = PG[d:] @ Vx                                     # [s0 | z0]
    T = Wqk[:d].t() @ SZ
    Out = torch.addcmul(Vx * n_total, T, c)
    return Out, (PG, ssq, c, Vx, SZ, T)


def attn_h_packed_bwd(Gt, Wqk, Wv, n_total: float, saved, gOut):
    """Gradients of the packed operands from gOut [(D + 1) x (d + 1)] (the gradients of M, m, w, beta in Out's blocks):
    20 launches, none of autograd's zero-filled slice gradients.  With Gt symmetric:
        gT = c gOut, gc = <gOut, T>, (g_sq, g_sk) = -gc c / (2 ssq)
        gWq~ = SZ gT^T + 2 g_sq PQ... (PQ = Wq~ Gt enters twice: through ||Q||^2 directly and through Gt's symmetry)"""
    PG, ssq, c, Vx, SZ, T = saved
    d = Wv.shape[0]
    gc = torch.dot(gOut.reshape(-1), T.reshape(-1))
    gs = (gc * c * -0.5) / ssq                           # [g_sq, g_sk]
    gT = gOut * c
    Wq = Wqk[:d]
    gSZ = Wq @ gT                                        # [d, d + 1]
    gs2 = gs.repeat_interleave(d)[:, None]               # [2 d, 1]
    gPG = gs2 * Wqk                                      # [g_sq Wq~ ; g_sk Wk~]
    gPG[d:] += gSZ @ Vx.t()                              # + the path through K^T V and K^T 1
    gWqk = torch.addmm(gs2 * PG, gPG, Gt)                # PG = Wqk Gt: gPG Gt^T (Gt symmetric) + the direct <PG, Wqk> term
    gWqk[:d] += SZ @ gT.t()
    gGt = Wqk.t() @ gPG
    gVx = torch.addmm(gOut * n_total, PG[d:].t(), gSZ)   # Vx enters twice: SZ = PK Vx and the n_total Vx term
    gWv = gVx[:, :d].t()
    return gGt, gWqk, gWv


