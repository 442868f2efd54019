"""PyTorch-CPU restatements used ONLY as checkers / CPU baseline (tests/, bench.py cpu_baseline leg).

* ``feature_match_index_conv`` -- the reference's own algorithm shape (ref patches as conv2d filters, chunked,
  running arg-max; ref_map_util.py:26-86) written against stock torch ops.  This is what "the reference's
  PyTorch-CPU path" costs on a host; bench.py times it next to the GPU number (kind="port").
* ``dcn_v2_reference`` -- differentiable gather-based DCNv2 forward following dcn_v2_im2col_cuda.cu:25-54,125-195
  and dcn_v2_cuda.cu:123-163.  Autograd through it (fp64) is the independent check of the analytic backward
  (dcn_v2_im2col_cuda.cu:56-123,197-327) restated in oracle/c2m_oracle.c.

Test infrastructure: nothing under c2-matching_amd/ imports this file.
"""
import torch
import torch.nn.functional as F


def feature_match_index_conv(feat_input, feat_ref, patch_size=3, input_stride=1, ref_stride=1, is_norm=True,
                             norm_input=False, chunk_elems=2 ** 29):
    c, h, w = feat_input.shape
    # every ref patch becomes one conv filter [n, c, p, p]; patches enumerated row-major (ref_map_util.py:19-22)
    filt = F.unfold(feat_ref[None], patch_size, stride=ref_stride)[0]           # [c*p*p, n]
    n = filt.shape[1]
    filt = filt.t().reshape(n, c, patch_size, patch_size)
    per = max(1, int(chunk_elems / (h * w)))                                      # :56 memory-bounded chunks
    best_v = best_i = None
    for s in range(0, n, per):
        f = filt[s:s + per]
        if is_norm:
            f = f / (f.flatten(1).norm(dim=1).view(-1, 1, 1, 1) + 1e-5)           # :62-63
        v, i = F.conv2d(feat_input[None], f, stride=input_stride)[0].max(dim=0)  # :64-69
        if best_v is None:
            best_v, best_i = v, i
        else:
            upd = v > best_v                                                      # strict: earlier chunk wins ties
            best_v = torch.where(upd, v, best_v)
            best_i = torch.where(upd, i + s, best_i)
    if norm_input:
        q = F.unfold(feat_input[None], patch_size, stride=input_stride)[0].norm(dim=0) + 1e-5
        best_v = best_v / q.view(best_v.shape)                                    # :78-84
    return best_i, best_v


def dcn_v2_reference(inp, weight, bias, offset, mask, stride=(1, 1), padding=(1, 1), dilation=(1, 1), dg=1):
    """Differentiable DCNv2 forward (any float dtype).  Shapes as dcn_v2.py:16-30."""
    B, C, H, W = inp.shape
    Co, _, kh, kw = weight.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    K, cpg = kh * kw, C // dg
    dt, dev = inp.dtype, inp.device
    base_y = (torch.arange(Ho, device=dev, dtype=dt) * sh - ph).view(1, 1, Ho, 1)
    base_x = (torch.arange(Wo, device=dev, dtype=dt) * sw - pw).view(1, 1, 1, Wo)
    off = offset.view(B, dg, K, 2, Ho, Wo)
    msk = mask.view(B, dg, K, Ho, Wo)
    flat = inp.reshape(B, dg, cpg, H * W)
    cols = []
    for t in range(K):
        i, j = divmod(t, kw)
        hy = base_y + i * dh + off[:, :, t, 0]            # [B,dg,Ho,Wo]
        wx = base_x + j * dw + off[:, :, t, 1]
        inside = (hy > -1) & (wx > -1) & (hy < H) & (wx < W)
        y0, x0 = torch.floor(hy), torch.floor(wx)
        ly, lx = hy - y0, wx - x0
        y0, x0 = y0.long(), x0.long()
        acc = 0
        for (yy, xx, wgt) in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx),
                              (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            lin = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(B, dg, 1, Ho * Wo).expand(B, dg, cpg, Ho * Wo)
            v = torch.gather(flat, 3, lin).view(B, dg, cpg, Ho, Wo)
            acc = acc + v * (wgt * ok.to(dt)).unsqueeze(2)
        cols.append(acc * msk[:, :, t].unsqueeze(2))      # [B,dg,cpg,Ho,Wo]
    col = torch.stack(cols, dim=3).reshape(B, C * K, Ho * Wo)   # column row index = c*K + t
    out = torch.einsum("ok,bkp->bop", weight.reshape(Co, C * K), col).view(B, Co, Ho, Wo)
    return out + bias.view(1, Co, 1, 1)
