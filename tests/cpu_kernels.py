"""A CPU kernel table with the exact contract of sgformer_amd.ops.HipKernels — TEST ONLY.

Installed with `ops.set_kernels(CpuKernels())` by the CPU tests so that the host logic above the
C ABI (ops.py autograd wiring, ours.py module surface, dist.py sharding and collectives under
gloo) can be exercised without a GPU.  Every method restates what the corresponding libsgf entry
point computes (same decomposition into un-normalised partials, same buffer layouts), on top of
the oracle package; it is therefore also an executable specification of include/sgf.h.  It lives
under tests/ and is never imported by the product.
"""
import numpy as np
import torch

from oracle import sgformer_oracle as O


class CpuKernels:
    name = "cpu-oracle"

    @staticmethod
    def check(*tensors):
        return None

    # ---- T1 ----
    @staticmethod
    def csr_build(ei, n):
        rowptr, colind, val, deg = O.csr_build(ei.cpu().numpy(), n)
        return (torch.from_numpy(rowptr), torch.from_numpy(colind.astype(np.int32)),
                torch.from_numpy(val), torch.from_numpy(deg.astype(np.int32)))

    @staticmethod
    def csr_transpose(ei, n, deg, rowptr, colind):
        t_rowptr, t_colind, t_val, sym = O.csr_transpose(ei.cpu().numpy(), n)
        return (torch.from_numpy(t_rowptr), torch.from_numpy(t_colind.astype(np.int32)),
                torch.from_numpy(t_val), sym)

    # ---- N1 ----
    @staticmethod
    def subgraph(ei, n, subset, relabel_nodes, want_eid):
        mask_n = torch.zeros(n, dtype=torch.bool)
        mask_n[subset] = True
        keep = mask_n[ei[0]] & mask_n[ei[1]]
        out = ei[:, keep]
        if relabel_nodes:
            idx = torch.zeros(n, dtype=torch.int64)
            idx[subset] = torch.arange(subset.numel())
            out = idx[out]
        return out, (keep.nonzero().view(-1) if want_eid else None)

    # ---- T7 fused ----
    @staticmethod
    def combine_fc_supported(d, classes, dtype):
        return dtype == torch.bfloat16 and d % 32 == 0 and d <= 256 and classes <= 64

    @staticmethod
    def combine_fc_mapped_supported(d, classes, dtype):
        return dtype == torch.bfloat16 and classes <= 64 and d % 32 == 0 and d <= 256

    @staticmethod
    def combine_fc_fwd(x1, a, x2, b, w, bias, row_map=None):
        xc = (a * x1.float() + b * x2.float()).to(x1.dtype)          # rounded once, as the kernel does
        out = xc.float() @ w.to(x1.dtype).float().t() + bias
        if row_map is not None:
            res = torch.empty_like(out)
            res[row_map.long()] = out
            return res
        return out

    @staticmethod
    def combine_fc_bwd(g, w, a, b, dtype, row_map=None):
        if row_map is not None:
            g = g[row_map.long()]
        dx = g.to(dtype).float() @ w.to(dtype).float()
        return (a * dx).to(dtype), (b * dx).to(dtype)

    # ---- streaming row-GEMM (csrc/rowgemm.hip) ----
    @staticmethod
    def gcn_epilogue_supported(d_in, d_out, dtype):
        return dtype == torch.bfloat16 and d_in == d_out and d_in in (64, 128, 256)

    @staticmethod
    def gcn_epilogue_stats(a, w, bias, shift=None, want_stats=False):
        y = a.float() @ w.float().t()
        if bias is not None:
            y = y + bias.float()
        y = y.to(a.dtype)
        if not want_stats:
            return y, None
        v = y.float() - (shift.float() if shift is not None else 0.0)
        return y, torch.cat([v.sum(0), (v * v).sum(0)])

    @staticmethod
    def gcn_epilogue_cat(a1, a2, w, bias, shift=None, want_stats=False):
        d = a1.shape[1]
        part, _ = CpuKernels.gcn_epilogue_stats(a1, w[:, :d], bias)          # rounded first product
        y = (a2.float() @ w[:, d:].float().t() + part.float()).to(a1.dtype)
        if not want_stats:
            return y, None
        v = y.float() - (shift.float() if shift is not None else 0.0)
        return y, torch.cat([v.sum(0), (v * v).sum(0)])

    @staticmethod
    def gcn_epilogue_dx(dy, w):
        return (dy.float() @ w.float()).to(dy.dtype)

    # ---- backward of the GCN layer's dense half (sgf_gcn_bn_bwd_dx); the opaque running sum is a plain bf16 matrix here ----
    @staticmethod
    def gcn_bn_bwd_dx_supported(d, dtype):
        return dtype == torch.bfloat16 and d in (64, 128, 256)

    @staticmethod
    def gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, w, acc_in, last, add_gy):
        d = z.shape[1]
        dz = CpuKernels.bn_bwd_apply(gy, z, mean, rstd, gamma, beta, relu, stats, inv_n, training)
        dy = (dz.float() @ w[:, :d].float()).to(z.dtype)
        acc = dz.float() @ w[:, d:].float()
        if add_gy:
            acc = acc + gy.float()
        if acc_in is not None:
            acc = acc + acc_in.float()
        return dz, dy, acc.to(z.dtype)

    @staticmethod
    def gcn_epilogue_dx2_acc_supported(d, dtype):
        return dtype == torch.bfloat16 and d in (64, 128, 256)

    @staticmethod
    def gcn_epilogue_dx2_acc(dz, w, gadd, acc_in):
        d = dz.shape[1]
        dy = (dz.float() @ w[:, :d].float()).to(dz.dtype)
        acc = (dz.float() @ w[:, d:].float()).to(dz.dtype).float()            # rounded once before the addends, as the kernel
        if gadd is not None:
            acc = acc + gadd.float()
        if acc_in is not None:
            acc = acc + acc_in.float()
        return dy, acc.to(dz.dtype)

    @staticmethod
    def gcn_epilogue_dx2(dy, w1, w2, pair=True):
        return CpuKernels.gcn_epilogue_dx(dy, w1), CpuKernels.gcn_epilogue_dx(dy, w2)

    @staticmethod
    def stem_pair_supported(d_in, d_out, dtype):
        return dtype == torch.bfloat16 and d_in % 4 == 0 and d_in <= 128 and d_out in (64, 128, 256)

    @staticmethod
    def stem_pair(x, w0, b0, w1, b1, shift0=None, want_stats0=False):
        y0, st = CpuKernels.gcn_epilogue_stats(x, w0, b0, shift0, want_stats0)
        y1 = None if w1 is None else CpuKernels.gcn_epilogue_stats(x, w1, b1)[0]
        return y0, y1, st

    # ---- graph-side planning (oracle/graph_oracle.py) ----
    @staticmethod
    def graph_prologue(ei, n, undirected, remove_loops, add_loops):
        from oracle import graph_oracle as G
        e = ei.cpu().numpy()
        if undirected and remove_loops and add_loops:
            return torch.from_numpy(G.graph_prologue(e, n, undirected=True))
        src, dst = e[0], e[1]
        if undirected:
            key = np.unique(np.concatenate([src, dst]) * n + np.concatenate([dst, src]))
            src, dst = key // n, key % n
        if remove_loops:
            keep = src != dst
            src, dst = src[keep], dst[keep]
        if add_loops:
            loops = np.arange(n, dtype=np.int64)
            src, dst = np.concatenate([src, loops]), np.concatenate([dst, loops])
        return torch.from_numpy(np.stack([src, dst]))

    @staticmethod
    def reorder(ei, n, iters1, iters2):
        from oracle import graph_oracle as G
        return tuple(torch.from_numpy(a) for a in G.reorder(ei.cpu().numpy(), n, iters1, iters2))

    @staticmethod
    def gather_rows(src, idx, out_dtype=None):
        return src[idx.long()].to(out_dtype or src.dtype)

    @staticmethod
    def pad_rows(src, idx, d_pad, out_dtype=None):
        rows = src if idx is None else src[idx.long()]
        out = torch.zeros((rows.shape[0], d_pad), dtype=out_dtype or src.dtype)
        out[:, :src.shape[1]] = rows.to(out.dtype)
        return out

    # ---- the general Linear (csrc/gemm.hip): fp32 accumulation of exact products, one rounding on the way out ----
    @staticmethod
    def gemm(a, b, bias=None, out=None, out_dtype=None, alpha=1.0, alpha_dev=None, beta=0.0, addend=None):
        al = float(alpha) * (float(alpha_dev) if alpha_dev is not None else 1.0)
        acc = torch.float64 if a.dtype == torch.float64 else torch.float32     # (fp64: the formulation tests)
        v = al * (a.to(acc) @ b.to(acc))
        if bias is not None:
            v = v + bias.to(acc)
        if addend is not None:
            v = v + beta * addend.to(acc)
        if out is None:
            return v.to(out_dtype or a.dtype)
        out.copy_(v.to(out.dtype))
        return out

    # ---- the d x d algebra of the attention (csrc/attn_small.hip): the packed formulation itself ----
    @staticmethod
    def attn_h_small_fwd(G, s, n_rows, n_total, wq, bq, wk, bk, wv, bv):
        from sgformer_amd import ops
        from tests import attn_algebra as A
        d, D = wq.shape
        if wv is None:
            wv, bv = torch.eye(D, dtype=torch.float32), torch.zeros(D, dtype=torch.float32)
        packed = ops._attn_h_pack(G.float(), s.float(), float(n_rows), wq.float(), bq.float(), wk.float(), bk.float(),
                                  wv.float(), bv.float())
        Out, saved = A.attn_h_packed_fwd(*packed, float(n_total))
        return (Out[:D, :d].contiguous(), Out[D, :d].contiguous(), Out[:D, d].contiguous(), Out[D:, d].contiguous(),
                (packed, saved))

    @staticmethod
    def attn_h_small_bwd(dM, dw, dm, dbeta, n_total, saved, d_in, d_out, want_v=True):
        from tests import attn_algebra as A
        packed, sv = saved
        D, d = d_in, d_out
        gOut = torch.empty((D + 1, d + 1), dtype=torch.float32)
        gOut[:D, :d], gOut[D, :d], gOut[:D, d], gOut[D, d:] = dM, dm, dw, dbeta
        gGt, gWqk, gWv = A.attn_h_packed_bwd(*packed, float(n_total), sv, gOut)
        Dm = gGt[:D, :D]
        return ((Dm + Dm.t()).contiguous(), (gGt[:D, D] + gGt[D, :D]).contiguous(), gWqk[:d, :D].contiguous(),
                gWqk[:d, D].contiguous(), gWqk[d:, :D].contiguous(), gWqk[d:, D].contiguous(),
                gWv[:, :D].contiguous() if want_v else None, gWv[:, D].contiguous() if want_v else None)

    @staticmethod
    def lds_rows_max(dtype):
        return 288 if dtype == torch.bfloat16 else 144

    # ---- T2 ----
    @staticmethod
    def spmm(rowptr, colind, val, x, n_rows, out=None, long_segments=0, stream_hint=False):
        y = torch.zeros((n_rows, x.shape[1]), dtype=x.dtype)
        nnz = int(rowptr[n_rows]) if n_rows > 0 else 0      # the kernels bound their work by rowptr: arrays may be longer
        if nnz > 0:
            counts = (rowptr[1:n_rows + 1] - rowptr[:n_rows])
            rows = torch.repeat_interleave(torch.arange(n_rows), counts)
            y.index_add_(0, rows, val[:nnz].to(x.dtype).unsqueeze(1) * x[colind[:nnz].long()])
        if out is not None:
            out.copy_(y)
            return out
        return y

    # ---- T3+T4 fused ----
    @staticmethod
    def attn_h_fwd(h, M, m, w, beta):
        hf = h.float()
        den = (hf @ w + beta).unsqueeze(1)
        return ((hf @ M + m) / den).to(h.dtype), den.contiguous()

    @staticmethod
    def attn_h_bwd_reduce(h, g, o, den):
        hf, gf, of = h.float(), g.float(), o.float()
        dnum = gf / den
        dden = -(gf * of).sum(1, keepdim=True) / den
        return torch.cat([(hf.t() @ dnum).reshape(-1), (hf * dden).sum(0), dnum.sum(0), dden.sum().reshape(1)])

    @staticmethod
    def attn_h_bwd_apply(h, g, o, den, M, w, D, ds):
        hf, gf, of = h.float(), g.float(), o.float()
        dnum = gf / den
        dden = -(gf * of).sum(1, keepdim=True) / den
        return (dnum @ M.t() + dden * w + hf @ D + ds).to(h.dtype)

    # the three-call form of the same backward (bf16 storage): state of the first pass kept on the table
    _h_partial = None

    @staticmethod
    def attn_h_bwd_split_supported(h, g, o):
        return h.dtype == torch.bfloat16 and h.shape[1] in (64, 128, 256)

    @staticmethod
    def attn_h_bwd_pre(g, o, den, M, w):
        gf, of = g.float(), o.float()
        inv = 1.0 / den.reshape(-1, 1)
        dden = -(gf * of).sum(1, keepdim=True) * inv
        CpuKernels._h_partial = ((gf @ M.t()) * inv + dden * w).to(g.dtype)   # rounded scratch
        return torch.cat([inv, dden], 1).contiguous()

    @staticmethod
    def attn_h_bwd_reduce_scaled(h, g, rowscal):
        hf = h.float()
        dnum = (g.float() * rowscal[:, :1]).to(g.dtype).float()      # re-rounded for the matrix cores, as the kernel does
        dden = rowscal[:, 1:2]
        return torch.cat([(hf.t() @ dnum).reshape(-1), (hf * dden).sum(0), (g.float() * rowscal[:, :1]).sum(0),
                          dden.sum().reshape(1)])

    @staticmethod
    def attn_h_bwd_post(h, D, ds, addend=None):
        dh = (h.float() @ D + ds + CpuKernels._h_partial.float()).to(h.dtype)
        return dh if addend is None else (dh.float() + addend.float()).to(h.dtype)

    @staticmethod
    def dropout(x, res, p, seed):
        g = torch.Generator().manual_seed(seed % (2 ** 31))
        keep = (torch.rand(x.shape, generator=g) >= p).to(torch.float32)
        y = x.float() * keep * (1.0 / (1.0 - p) if p < 1.0 else 0.0)
        if res is not None:
            y = y + res.float()
        return y.to(x.dtype)

    @staticmethod
    def nll_fwd(logits, labels, idx):
        lp = torch.log_softmax(logits.float(), dim=1)
        lab = labels[idx]
        ok = (lab >= 0) & (lab < logits.shape[1])            # include/sgf.h: labels outside [0, c) add nothing
        return -(lp[idx, lab.clamp(0, logits.shape[1] - 1)] * ok).sum().reshape(1)

    @staticmethod
    def nll_bwd(logits, labels, idx, gout, inv_denom):
        d = torch.zeros_like(logits, dtype=torch.float32)
        p = torch.softmax(logits.float()[idx], dim=1)
        lab = labels[idx]
        ok = (lab >= 0) & (lab < logits.shape[1])
        p[torch.arange(idx.numel()), lab.clamp(0, logits.shape[1] - 1)] -= 1.0
        d[idx] = p * ok[:, None] * (gout[0] * inv_denom)
        return d.to(logits.dtype)

    @staticmethod
    def sum_n(xs):
        out = xs[0].float()
        for x in xs[1:]:
            out = out + x.float()
        return out.to(xs[0].dtype)

    # ---- T4 ----
    @staticmethod
    def gram(a, b, out=None, want_colsum=True):
        c = a.float().t() @ b.float()
        if out is None:
            out = c
        else:
            out.copy_(c)
        return out, (a.float().sum(0) if want_colsum else None)

    @staticmethod
    def gram2(a, b1, b2, out1, out2, want_colsum=True):
        out1.copy_(a.float().t() @ b1.float())
        out2.copy_(a.float().t() @ b2.float())
        return a.float().sum(0) if want_colsum else None

    # ---- T3 ----
    @staticmethod
    def _unpack(stats, heads, d):
        nm = heads * d * d
        return stats[:nm].reshape(heads, d, d), stats[nm:nm + heads * d].reshape(heads, d)

    @staticmethod
    def attn_fwd_reduce(q, k, v, heads, v_heads, d):
        n = q.shape[0]
        return O.attention_raw_stats(q.reshape(n, heads, d).float(), k.reshape(n, heads, d).float(),
                                     v.reshape(n, v_heads, d).float())

    @staticmethod
    def attn_fwd_apply(q, v, stats, n_total, heads, v_heads, d):
        n = q.shape[0]
        s0, z0 = CpuKernels._unpack(stats, heads, d)
        c = 1.0 / (torch.sqrt(stats[-2]) * torch.sqrt(stats[-1]))
        qh = q.reshape(n, heads, d).float()
        vh = v.reshape(n, v_heads, d).float().expand(-1, heads, -1)
        den = c * torch.einsum("nhm,hm->nh", qh, z0) + n_total
        num = c * torch.einsum("nhm,hmd->nhd", qh, s0) + n_total * vh
        o = num / den.unsqueeze(-1)
        out = o.mean(dim=1).to(q.dtype)
        return out, den.contiguous(), (o.reshape(n, heads * d).to(q.dtype) if heads > 1 else None)

    @staticmethod
    def attn_bwd_reduce(q, g, o, den, heads, d, per_head=False):
        n = q.shape[0]
        qh = q.reshape(n, heads, d).float()
        oh = o.reshape(n, heads, d).float()
        gh = g.float().reshape(n, heads, d) if per_head else (g.float() / heads).unsqueeze(1).expand(-1, heads, -1)
        dnum = gh / den.unsqueeze(-1)
        dden = -(gh * oh).sum(-1) / den
        ds0 = torch.einsum("nhm,nhd->hmd", qh, dnum)
        dz0 = torch.einsum("nhm,nh->hm", qh, dden)
        return torch.cat([ds0.reshape(-1), dz0.reshape(-1), torch.zeros(1)])

    @staticmethod
    def attn_bwd_apply(q, k, v, g, o, den, stats, bstats, n_total, heads, v_heads, d, dq, dk, dv, per_head=False):
        n = q.shape[0]
        s0, z0 = CpuKernels._unpack(stats, heads, d)
        ds0, dz0 = CpuKernels._unpack(bstats, heads, d)
        ssq_q, ssq_k = stats[-2], stats[-1]
        c = 1.0 / (torch.sqrt(ssq_q) * torch.sqrt(ssq_k))
        sdot = (s0 * ds0).sum() + (z0 * dz0).sum()
        bstats[-1] = sdot
        s = c * sdot
        qh, kh = q.reshape(n, heads, d).float(), k.reshape(n, heads, d).float()
        vh = v.reshape(n, v_heads, d).float().expand(-1, heads, -1)
        oh = o.reshape(n, heads, d).float()
        gh = g.float().reshape(n, heads, d) if per_head else (g.float() / heads).unsqueeze(1).expand(-1, heads, -1)
        dnum = gh / den.unsqueeze(-1)
        dden = -(gh * oh).sum(-1) / den
        gq = c * (torch.einsum("nhd,hmd->nhm", dnum, s0) + dden.unsqueeze(-1) * z0) - s * qh / ssq_q
        gk = c * (torch.einsum("nhd,hmd->nhm", vh, ds0) + dz0) - s * kh / ssq_k
        gv = n_total * dnum + c * torch.einsum("nhm,hmd->nhd", kh, ds0)
        if v_heads == 1 and heads > 1:
            gv = gv.sum(dim=1, keepdim=True)
        dq.copy_(gq.reshape(n, -1).to(dq.dtype))
        dk.copy_(gk.reshape(n, -1).to(dk.dtype))
        dv.copy_(gv.reshape(n, -1).to(dv.dtype))

    # ---- T5 ----
    @staticmethod
    def ln_fwd(x, res, a, b, gamma, beta, relu, eps):
        pre = a * x.float() + (b * res.float() if res is not None else 0.0)
        mean = rstd = None
        if gamma is not None:
            mean = pre.mean(dim=1)
            rstd = 1.0 / torch.sqrt(((pre - mean[:, None]) ** 2).mean(dim=1) + eps)
            pre = (pre - mean[:, None]) * rstd[:, None] * gamma + beta
        y = torch.relu(pre) if relu else pre
        return y.to(x.dtype), mean, rstd

    @staticmethod
    def ln_bwd(gy, y, x, res, a, b, gamma, relu, mean, rstd):
        dz = gy.float()
        if relu:
            dz = dz * (y.float() > 0)
        dgamma = dbeta = None
        if gamma is not None:
            pre = a * x.float() + (b * res.float() if res is not None else 0.0)
            xh = (pre - mean[:, None]) * rstd[:, None]
            dgamma, dbeta = (dz * xh).sum(0), dz.sum(0)
            dxh = dz * gamma
            dz = rstd[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
        dx = (a * dz).to(x.dtype)
        dres = (b * dz).to(x.dtype) if res is not None else None
        return dx, dres, dgamma, dbeta

    # ---- T6 ----
    @staticmethod
    def colstats(x, shift):
        v = x.float() - (shift if shift is not None else 0.0)
        return torch.cat([v.sum(0), (v * v).sum(0)])

    @staticmethod
    def _bn(x, mean, rstd, gamma, beta):
        xh = (x.float() - mean) * rstd
        return xh, xh * (gamma if gamma is not None else 1.0) + (beta if beta is not None else 0.0)

    @staticmethod
    def bn_finalize(sums, shift, n_total, eps, momentum, running_mean, running_var):
        d = sums.numel() // 2
        n = max(float(n_total), 1.0)
        m1 = sums[:d] / n
        mean = (shift if shift is not None else 0.0) + m1
        var = (sums[d:] / n - m1 * m1).clamp_min(0.0)
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
        if running_var is not None:
            running_var.mul_(1.0 - momentum).add_(var * (float(n_total) / max(float(n_total) - 1.0, 1.0)), alpha=momentum)
        return mean, torch.rsqrt(var + eps)

    @staticmethod
    def bn_apply(x, mean, rstd, gamma, beta, res, relu):
        _, y = CpuKernels._bn(x, mean, rstd, gamma, beta)
        if relu:
            y = torch.relu(y)
        if res is not None:
            y = y + res.float()
        return y.to(x.dtype)

    @staticmethod
    def bn_bwd_stats(gy, x, mean, rstd, gamma, beta, relu):
        xh, y = CpuKernels._bn(x, mean, rstd, gamma, beta)
        dz = gy.float() * (y > 0) if relu else gy.float()
        return torch.cat([dz.sum(0), (dz * xh).sum(0)])

    @staticmethod
    def bn_bwd_stats2(gy, gy2, x, mean, rstd, gamma, beta, relu):
        g = gy if gy2 is None else (gy.float() + gy2.float())
        return CpuKernels.bn_bwd_stats(g, x, mean, rstd, gamma, beta, relu)

    @staticmethod
    def gram_ln_bwd_supported(m, k, dtype):
        return dtype == torch.bfloat16 and m in (64, 128, 256) and k % 4 == 0 and k <= 256

    @staticmethod
    def gram_ln_bwd(g, xin, mean, rstd, gamma, beta, relu, b):
        xh = (xin.float() - mean[:, None]) * rstd[:, None]
        ga = gamma if gamma is not None else 1.0
        gm = g.float()
        if relu:
            gm = gm * ((xh * ga + (beta if beta is not None else 0.0)) > 0)
        dxh = gm * ga
        dl = rstd[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
        dl = dl.to(xin.dtype).float()
        return dl.t() @ b.float(), dl.sum(0), (gm * xh).sum(0), gm.sum(0)

    @staticmethod
    def gram_bn_bwd_supported(m, k, dtype):
        return dtype == torch.bfloat16 and m % 4 == 0 and k % 4 == 0 and m <= 256 and k <= 256

    @staticmethod
    def gram_bn_bwd(gy, gy2, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, b):
        g = gy.float() if gy2 is None else (gy.float() + gy2.float())
        dz = CpuKernels.bn_bwd_apply(g, z, mean, rstd, gamma, beta, relu, stats, inv_n, training).float()
        return dz.t() @ b.float(), dz.sum(0)

    @staticmethod
    def bn_bwd_apply(gy, x, mean, rstd, gamma, beta, relu, stats, inv_n, training):
        d = x.shape[1]
        xh, y = CpuKernels._bn(x, mean, rstd, gamma, beta)
        dz = gy.float() * (y > 0) if relu else gy.float()
        if training:
            dz = dz - stats[:d] * inv_n - xh * stats[d:] * inv_n
        return ((gamma if gamma is not None else 1.0) * rstd * dz).to(x.dtype)

    @staticmethod
    def colsum(x):
        return x.float().sum(0)

    # ---- T7 ----
    @staticmethod
    def axpby(x1, a, x2, b):
        return (a * x1.float() + b * x2.float()).to(x1.dtype)
