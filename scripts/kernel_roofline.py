"""Per-kernel achieved bandwidth from a `rocprofv3 --kernel-trace --stats` CSV of bench.py.

    python scripts/kernel_roofline.py profiles/r02_products_bf16_kernel_stats.csv > profiles/r02_products_bf16_kernel_roofline.md

Every kernel of the step streams whole [N, d] activation tensors; its ALGORITHMIC bytes are the number
of such tensors it must read + write (DESIGN.md §3 column "algorithmic bytes") times T = N*d*s.  The
table divides that by the profile's average duration.  Two ceilings are quoted: the 8 TB/s HBM3E
spec figure bench.py's `roofline.peak` uses, and the 6.3 TB/s a plain device copy reaches on this
part (/opt/skills/guides/MI355X_MICROARCH.md), which is the practical bound of a streaming kernel.
"""
from __future__ import annotations

import csv
import sys

N, NNZ, D = 2_449_029, 126_123_675, 256      # synth.SHAPES['ogbn-products']
SPEC, COPY = 8000.0, 6300.0                  # GB/s


def main(path: str, elem: int):
    tname = "unsigned short" if elem == 2 else "float"
    T = N * D * elem / 1e9
    rows = list(csv.DictReader(open(path)))

    def find(*subs):
        for r in rows:
            if all(s in r["Name"] for s in subs):
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
        return None, 0

    def find_full(*subs):
        """Kernels that ops.linear_bn_stats also launches on a 1024-row sample (for BatchNorm's shift) appear twice
        per layer: the full-size launch is (total - the ~12 us sample launches) / (calls / 2)."""
        for r in rows:
            if all(s in r["Name"] for s in subs):
                calls = int(r["Calls"]) // 2
                return (float(r["TotalDurationNs"]) / 1e3 - 12.0 * calls) / calls, calls
        return None, 0

    # r04 on: the GCN layer's forward is ONE two-operand pass (k_rowgemm2_bf16), its input gradients one PAIRED launch, its
    # dW one paired Gram; the stems' backward forms dz / dl inside the Gram (modes 5, 6).  Calls per step change the mix.
    r04 = find("k_rowgemm2_bf16<")[0] is not None
    XF = N * 100 * elem / 1e9                    # the [N, 100] feature matrix
    GC = N * 48 * elem / 1e9                     # the logits' gradient, padded to 48 columns
    csr = (NNZ * 8 + (N + 1) * 8) / 1e9
    gather = (NNZ * (8 + D * elem) + (N + 1) * 8) / 1e9 + T     # every stored entry fetches one X row
    spmm_name = "k_spmm_row<" if find("k_spmm_row<" + tname)[0] else "k_spmm_wave<"
    spmm_us = find(spmm_name + tname)[0]
    table = [
        (spmm_name.rstrip("<"), (spmm_name + tname,), csr + 2 * T,
         "gather-bound: {:.1f} GB of {}-B row fetches per launch = {:.1f} TB/s".format(
             gather, D * elem, gather / (spmm_us * 1e-6) / 1e3) if spmm_us else "", find),
        ("k_reduce_bf16<256,Gram,16>  (sgf_gram: G, dW)", ("k_reduce_bf16<256, 2, 16>",),
         ((T + 2 * (T + GC) + 3 * 3 * T) / 6) if r04 else 2 * T,
         "6 per step: G = h^T h (1T), the head's two dW (T + g each), three PAIRED dW of a GCN layer (dz, y, x0: 3T)" if r04
         else "reads 2 tensors", find),
        # r06: the node reductions on csrc/gramx.hip (tiles by LDS-DMA, fragments by transposing LDS reads)
        ("k_gramx<Gram>  (sgf_gram / sgf_gram2: G, dW)", ("k_gramx<0>",), (T + 2 * (T + GC) + 3 * 3 * T) / 6,
         "6 per step: G = h^T h (1T, one image serves both operands), the head's two dW (T + g each), three PAIRED dW of a "
         "GCN layer (dz, y, x0: 3T)", find),
        ("k_gramx<BwdHS>  (sgf_attn_h_bwd_reduce_scaled)", ("k_gramx<1>",), 2 * T + N * 8 / 1e9,
         "reads h, dout and 8 B of row scalars per node; dnum formed in LDS", find),
        ("k_gramt<BN>  (sgf_gram_bn_bwd)", ("k_gramt<0>",), 3 * T + N * 104 * elem / 1e9,
         "GraphConv stem: g1, g2, z -> dz formed in LDS, x [N,104]: dW, db without a stored dz", find),
        ("k_gramt<LN>  (sgf_gram_ln_bwd)", ("k_gramt<1>",), 2 * T + N * 104 * elem / 1e9 + N * 8 / 1e9,
         "TransConv stem: g, LayerNorm input, row statistics -> dl formed in LDS, x [N,104]: dW, db, dgamma, dbeta", find),
        ("k_reduce_bf16<256,GramBN,8>  (sgf_gram_bn_bwd)", ("k_reduce_bf16<256, 5, 8>",), 3 * T + XF,
         "GraphConv stem: g1, g2, z -> dz on the fly, x [N,100]: dW, db without a stored dz", find),
        ("k_reduce_bf16<256,GramLN,8>  (sgf_gram_ln_bwd)", ("k_reduce_bf16<256, 6, 8>",), 2 * T + XF,
         "TransConv stem: g, LayerNorm input -> its input gradient on the fly, x [N,100]: dW, db, dgamma, dbeta", find),
        ("k_reduce_bf16<256,BwdH,8>", ("k_reduce_bf16<256, 3, 8>",), 3 * T, "reads h, out, dout", find),
        ("k_reduce_bf16<256,BwdHS,8>  (sgf_attn_h_bwd_reduce_scaled)", ("k_reduce_bf16<256, 4, 8>",), 2 * T + N * 8 / 1e9,
         "reads h, dout and 8 B of row scalars per node", find),
        ("k_stem_bf16<256>  (sgf_stem_pair)", ("k_stem_bf16<256, true>",), N * 100 * elem / 1e9 + 2 * T,
         "x [N,100] -> both stems + BatchNorm sums; full-size launches only", find_full),
        ("k_apply_bf16<256,HFwd>", ("k_apply_bf16<256, 4, 2>",), 2 * T, "h -> out", find),
        ("k_apply_bf16<256,HBwd1>", ("k_apply_bf16<256, 5, 2>",), 3 * T, "", find),
        ("k_apply_bf16<256,HBwd2>", ("k_apply_bf16<256, 6, 2>",), 3 * T, "", find),
        ("k_hrow_bf16<256,F>  (sgf_attn_h_fwd)", ("k_hrow_bf16<256, 0>",), 2 * T, "h -> out, den", find),
        ("k_hrow_bf16<256,B1>  (sgf_attn_h_bwd_pre)", ("k_hrow_bf16<256, 1>",), 3 * T,
         "reads g, out; writes the partial and the row scalars", find),
        ("k_hrow_bf16<256,B2>  (sgf_attn_h_bwd_post)", ("k_hrow_bf16<256, 2>",), 4 * T,
         "reads h, the partial, the residual's gradient; writes dh", find),
        ("k_rowgemm_bf16<256,IO 0>  (sgf_gcn_epilogue_dx2, paired)" if r04 else "k_rowgemm_bf16<256,IO 0>  (sgf_gcn_epilogue_dx)",
         ("k_rowgemm_bf16<256, false, 0, 0>",), 3 * T if r04 else 2 * T,
         "dz -> dy AND dx0 from one HBM read of dz (workgroups b, b + 8 share the tile in L2)" if r04 else "dy -> dx", find),
        ("k_dx2acc_bf16<256,2>  (sgf_gcn_epilogue_dx2_acc, balanced pairs)", ("k_dx2acc_bf16<256, 2>",), 14 * T / 3,
         "dz -> dy AND acc_out = dz W2 + residual gradient + acc_in: 4T for the first layer of the chain, 5T for the other two",
         find),
        ("k_rowgemm2_bf16<256,2,stats>  (sgf_gcn_epilogue_cat)", ("k_rowgemm2_bf16<256, 2, true>",), 3 * T,
         "[y | x0] W^T + b + BatchNorm sums in one pass (paired column halves); full-size launches only", find_full),
        ("k_rowgemm_bf16<256,IO 1>  (sgf_gcn_epilogue_partial)", ("k_rowgemm_bf16<256, false, 1, 0>",), 2 * T,
         "a1 -> partial; full-size launches only", find_full),
        ("k_rowgemm_bf16<256,stats,IO 2>  (sgf_gcn_epilogue_stats_add)", ("k_rowgemm_bf16<256, true, 2, 0>",), 3 * T,
         "a2, partial -> y + BatchNorm sums; full-size launches only", find_full),
        ("k_attn_reduce<float,256,Gram>  (sgf_gram: G, dW; exact-fp32 MFMA)", ("k_attn_reduce<float, 256, 2>",), 2 * T,
         "MFMA-bound: 2*N*d^2 flop at the 157 TF fp32 MFMA peak = 2.0 ms", find),
        ("k_attn_reduce<float,256,BwdH>", ("k_attn_reduce<float, 256, 3>",), 3 * T, "MFMA-bound", find),
        ("k_attn_apply<float,256,HFwd>", ("k_attn_apply<float, 256, 4>",), 2 * T, "MFMA-bound", find),
        ("k_attn_apply<float,256,HBwd1>", ("k_attn_apply<float, 256, 5>",), 3 * T, "MFMA-bound", find),
        ("k_attn_apply<float,256,HBwd2>", ("k_attn_apply<float, 256, 6>",), 3 * T, "MFMA-bound", find),
        ("k_ln_fwd", ("k_ln_fwd<" + tname,), 2.5 * T, "mean of stem (2T) and post-attention (3T) calls", find),
        ("k_ln_fwd_bf16x8", ("k_ln_fwd_bf16x8<",), 2.5 * T, "mean of stem (2T) and post-attention (3T) calls", find),
        ("k_ln_bwd", ("k_ln_bwd<" + tname,), 5 * T if r04 else 4.5 * T,
         "post-attention only (g, y, x, res in; one shared dx out)" if r04 else "mean of stem (4T) and post-attention (5T) calls", find),
        ("k_bn_apply", ("k_bn_apply<" + tname,), 2.75 * T, "stem 2T, layers 3T (residual)", find),
        ("k_colreduce<BnBwdStats>", ("BnBwdStatsF<" + tname,), 2.25 * T if r04 else 2 * T,
         "three layers (g, z) and the stem (g1, g2, z)" if r04 else "", find),
        ("k_bn_bwd_apply", ("k_bn_bwd_apply<" + tname,), 3 * T, "", find),
        ("k_ln_bwd_bf16x8", ("k_ln_bwd_bf16x8<",), 4 * T, "post-attention only (g, x, res in — no activation follows, so y is not read; one shared dx out: 4T, as the PMC pass measures); 16 B per lane (r06)", find),
        ("k_bn_apply_bf16x8<residual>", ("k_bn_apply_bf16x8<true",), 3 * T, "the three layers: z, residual -> x'; 16 B per lane (r06)", find),
        ("k_bn_apply_bf16x8<no residual>", ("k_bn_apply_bf16x8<false",), 2 * T, "the stem", find),
        ("k_bn_bwd_stats_bf16x8<one gradient>", ("k_bn_bwd_stats_bf16x8<false",), 2 * T, "the three layers (g, z)", find),
        ("k_bn_bwd_stats_bf16x8<two gradients>", ("k_bn_bwd_stats_bf16x8<true",), 3 * T, "the stem (g1, g2, z)", find),
        ("k_bn_bwd_apply_bf16x8", ("k_bn_bwd_apply_bf16x8<",), 3 * T, "g, z -> dz", find),
        ("k_sum_n (7 operands)", ("k_sum_n<" + tname,), 8 * T, "fan-out hub gradient", find),
        ("k_sum_n_bf16x8<7>", ("k_sum_n_bf16x8<7>",), 8 * T, "fan-out hub gradient", find),
        ("k_sum_n_bf16x8<6>", ("k_sum_n_bf16x8<6>",), 7 * T, "x0's gradient: three dz W2 and three residual gradients", find),
        ("k_head_fwd_bf16  (sgf_combine_fc_fwd)", ("k_head_fwd_bf16<256>",), 2 * T + N * 47 * 4 / 1e9, "x1, x2 -> logits", find),
        ("k_head_bwd_bf16  (sgf_combine_fc_bwd_g)", ("k_head_bwd_bf16<256>",), 2 * T + N * 47 * 4 / 1e9 + (GC if find("k_gramx<0>")[0] else 0.0),
         "dlogits -> dx1, dx2" + (" and (r06) the bf16 [N, 48] operand of the head's dW" if find("k_gramx<0>")[0] else ""), find),
        ("k_axpby", ("k_axpby<" + tname,), 3 * T, "", find),
    ]
    if find("k_rowgemm_bf16<")[0]:    # r02 on: the square layers run on k_rowgemm_bf16, the library keeps the two input stems
        table.append(("hipBLASLt [N,100]x[100,256]  (the two input stems)", ("Cijk_Alik_Bljk", "MT256x256x32"),
                      N * (100 + 256) * elem / 1e9, "library GEMM (rows of 100 elements are not 16-byte aligned)", find))
    else:
        table.append(("hipBLASLt [N,256]x[256,256] (Y = XW^T)", ("Cijk_Alik_Bljk", "MT256x256x32"), 2 * T,
                      "library GEMM, HBM-bound shape", find))
        table.append(("hipBLASLt [N,256]x[256,256] (dX = dY W)", ("Cijk_Ailk_Bljk", "MT256x256x32"), 2 * T, "library GEMM", find))
    print(f"# Per-kernel achieved bandwidth — {path.split('/')[-1]}\n")
    print(f"ogbn-products shape: N = {N:,}, d = {D}, {elem}-byte activations, T = N*d*s = {T:.3f} GB.  "
          f"Generated by `python scripts/kernel_roofline.py {path}`.\n")
    print("| kernel | calls | avg us | algorithmic GB | GB/s | of 8 TB/s spec | of 6.3 TB/s copy | note |")
    print("|---|---|---|---|---|---|---|---|")
    for name, subs, gb, note, finder in table:
        us, calls = finder(*subs)
        if us is None:
            continue
        bw = gb / (us * 1e-6)
        print(f"| `{name}` | {calls} | {us:.0f} | {gb:.2f} | {bw:.0f} | {bw / SPEC:.2f} | {bw / COPY:.2f} | {note} |")


if __name__ == "__main__":
    p = sys.argv[1]
    main(p, 4 if "f32" in p else 2)
