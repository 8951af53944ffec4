"""Torch-tensor front end of the dense C-ABI operators (include/bdm_hip.h, section 2).

Tensors are (B, C, L) or (B, C, M, U) fp32 on a HIP device; any view whose innermost dims
are contiguous (e.g. a channel slice of a concat buffer) is passed by (batch stride,
row stride) without a copy.  Torch supplies memory and the current stream only.
"""
import ctypes
import os

import torch

from . import _lib as L
from . import tape


def _bcl(t):
    """(tensor, B, C, L, batch_stride, row_stride) of a (B, C, *rest) tensor.

    Views whose trailing dims are densely packed (channel slices of a concat buffer) pass
    through untouched; anything else (expanded / permuted views) is made contiguous first."""
    if t.dtype != torch.float32:
        raise L.BdmHipError(f"expected float32, got {t.dtype}")
    B, C = t.shape[0], t.shape[1]
    if B * C > 0 and t.is_contiguous():  # the common case, without the stride walk below
        l = t.numel() // (B * C)
        return t, B, C, l, C * l, l
    l = 1
    for s in t.shape[2:]:
        l *= s
    dense, exp = True, 1
    for d in range(t.dim() - 1, 1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            dense = False
        exp *= t.shape[d]
    if dense and C > 1 and t.stride(1) < l:
        dense = False
    if not dense:
        t = t.contiguous()
    ld = t.stride(1) if C > 1 else l
    bs = t.stride(0) if B > 1 else C * ld
    return t, B, C, l, bs, ld


_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """Cached scratch buffer per (tag, device, current stream): kernels of different streams never share one."""
    dev = torch.device(device)
    raw = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch._C._cuda_getDevice()) if dev.type == "cuda" else 0
    key = (tag, str(device), raw)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# 1x1 convolutions: "fp32" (default) = fp32-input MFMA (dense_ops.hip); "bf16x6" = exact 3-way bf16 operand split, six partial
# products on the 16-bit matrix pipe (csrc/experimental/pointwise_s3.hip, EXPERIMENTAL=1 builds only).  Measured at B = 16: no
# faster (the GEMMs are bound by operand staging, not by the matrix pipe), so it stays opt-in.  The skinny shapes (<= 64 columns,
# K >= 128) keep their K-split fp32 kernel either way.
PW_IMPL = "fp32"  # "bf16x6": experimental build only (tests / tools/pw_bench.py set it)
PW_S3_MIN_COLUMNS = 65   # shapes with fewer columns per shape stay on the fp32 kernels (latency-bound: nothing to gain)
_pw_s3_packs = {}


def _pw_s3_weights(w, weight):
    """bf16x6 records of a (M, K) weight matrix, packed once per (tensor, version).  Only for views of a long-lived tensor (a
    module parameter or a cached concatenation): a temporary copy would be re-packed on every call -> None (fp32 kernel)."""
    if PW_IMPL != "bf16x6" or not w.is_cuda or w.data_ptr() != weight.data_ptr():
        return None
    L.experimental("bdm_pointwise_s3_pack_weights")  # a clear error in a default build (never a silent fallback)
    M, K = w.shape
    key = (w.data_ptr(), M, K)
    hit = _pw_s3_packs.get(key)
    if hit is None or hit[0] != weight._version or hit[2] is not weight:
        if len(_pw_s3_packs) > 1024:  # (a model has ~50 such matrices; tests create many throw-away ones)
            _pw_s3_packs.clear()
        lib = L.lib()
        packed = torch.empty(lib.bdm_pointwise_s3_weight_elems(M, K), dtype=torch.bfloat16, device=w.device)
        L.check(lib.bdm_pointwise_s3_pack_weights(M, K, L.ptr(w), K, L.ptr(packed), L.stream()), "pointwise_s3_pack_weights")
        hit = (weight._version, packed, weight)  # holds the weight tensor: its address cannot be recycled while the pack lives
        _pw_s3_packs[key] = hit
    return hit[1]


def pointwise_conv(x, weight, bias=None, out=None, batch_bias=None, act=0, slope=0.0, residual=None):
    """y = W x + b  over the channel axis; weight (M,K[,1[,1]]) as in nn.Conv1d/Conv2d(k=1)/nn.Linear."""
    x, B, K, n, bs_x, ld_x = _bcl(x)
    M = weight.shape[0]
    w = weight.reshape(M, -1)
    assert w.shape[1] == K, f"weight expects {w.shape[1]} channels, input has {K}"
    w = w if w.is_contiguous() else w.contiguous()
    if out is None:
        out = torch.empty((B, M) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    o, Bo, Mo, no, bs_y, ld_y = _bcl(out)
    assert o.data_ptr() == out.data_ptr() and (Bo, Mo, no) == (B, M, n), "out must be a dense-row view of matching shape"
    if residual is not None:
        rr, _, _, _, bs_r, ld_r = _bcl(residual)
        assert rr.data_ptr() == residual.data_ptr() and residual.shape[1] == M
    else:
        bs_r, ld_r = 0, 0
    packed = _pw_s3_weights(w, weight) if (n >= PW_S3_MIN_COLUMNS and int(act) in (0, 2, 3)) else None
    if packed is not None:
        L.check(L.lib().bdm_pointwise_conv_s3(B, M, K, n, L.ptr(packed), L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(bias),
                                              L.ptr(batch_bias), M, L.ptr(residual), L.c_ll(bs_r), ld_r, L.ptr(out),
                                              L.c_ll(bs_y), ld_y, int(act), L.c_float(slope), L.stream()), "pointwise_conv_s3")
        return out
    L.check(L.lib().bdm_pointwise_conv(B, M, K, n, L.ptr(w), K, L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(bias),
                                       L.ptr(batch_bias), M, L.ptr(residual), L.c_ll(bs_r), ld_r, L.ptr(out),
                                       L.c_ll(bs_y), ld_y, int(act), L.c_float(slope), L.stream()), "pointwise_conv")
    return out


def gn_foldable(channels, groups):
    """Can bdm_pointwise_conv_gn leave / take GroupNorm(groups) statistics for a `channels`-wide tensor?"""
    cg = channels // groups if groups and channels % groups == 0 else 0
    return groups <= 8 and cg >= 4 and (cg & (cg - 1)) == 0 and channels <= 1024


def pointwise_conv_gn(x, weight, bias=None, out=None, fold_in=None, out_groups=None, x2=None, amax=None, amax_rows=0, add=None, batch_bias=None):
    """1x1 convolution with GroupNorm folding (bdm_pointwise_conv_gn).  fold_in = (stats, gn) of a previous call: x is that
    call's raw output and Swish(GroupNorm(x)) is applied on the fly.  out_groups: also return the statistics of the output
    -> (y, (partial, slices, groups)).  add (B, M, n): per-element addend, part of y before the statistics (the hoisted share of
    the layer: ops.Conditioning).  batch_bias (B, M[, 1]): per-shape bias, likewise (the time embedding's share).  shared_mlp.py:25-30."""
    x, B, K, n, bs_x, ld_x = _bcl(x)
    k1, x2p, bs_x2, ld_x2 = 0, None, 0, 0
    if x2 is not None:  # the operand is cat([x, x2], dim=1), read in place
        x2p, B2, K2, n2, bs_x2, ld_x2 = _bcl(x2)
        assert (B2, n2) == (B, n)
        k1, K = K, K + K2
    M = weight.shape[0]
    w = weight.reshape(M, -1)
    assert w.shape[1] == K, f"weight expects {w.shape[1]} channels, input has {K}"
    w = w if w.is_contiguous() else w.contiguous()
    if out is None:
        out = torch.empty((B, M) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    o, Bo, Mo, no, bs_y, ld_y = _bcl(out)
    assert o.data_ptr() == out.data_ptr() and (Bo, Mo, no) == (B, M, n), "out must be a dense-row view of matching shape"
    lib = L.lib()
    in_p, in_s, in_g, in_gamma, in_beta, in_eps = None, 0, 0, None, None, 0.0
    if fold_in is not None:
        (in_p, in_s, in_g), gn = fold_in
        assert gn.num_groups == in_g and gn.weight.shape[0] == K
        in_gamma, in_beta, in_eps = gn.weight, gn.bias, gn.eps
    stats, out_p, og = None, None, 0
    if out_groups:
        og = int(out_groups)
        slices = lib.bdm_pointwise_conv_gn_slices(B, M, K, n, og)
        assert slices > 0
        out_p = torch.empty(B * og * slices * 2, dtype=torch.float64, device=x.device)
        stats = (out_p, slices, og)
    if batch_bias is not None:
        bb = batch_bias.reshape(B, -1)
        assert bb.shape[1] == M and bb.stride(1) == 1, "batch_bias must be (B, M) with unit channel stride"
        ap, bs_a, ld_a = None, 0, 0
        if add is not None:
            aa, Ba, Ma, na, bs_a, ld_a = _bcl(add)
            assert aa.data_ptr() == add.data_ptr() and (Ba, Ma, na) == (B, M, n), "add must be a dense-row (B, M, n) view"
            ap = add
        L.check(lib.bdm_pointwise_conv_gn_bb(B, M, K, n, L.ptr(w), K, L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(x2p), L.c_ll(bs_x2), ld_x2, k1,
                                             L.ptr(bias), L.ptr(out), L.c_ll(bs_y), ld_y, L.ptr(in_p), in_s, in_g, L.ptr(in_gamma), L.ptr(in_beta),
                                             L.c_float(in_eps), og, L.ptr(out_p), L.ptr(amax), int(amax_rows), L.ptr(bb), bb.stride(0),
                                             L.ptr(ap), L.c_ll(bs_a), ld_a, L.stream()), "pointwise_conv_gn_bb")
        return (out, stats) if out_groups else out
    if add is not None:
        aa, Ba, Ma, na, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr() and (Ba, Ma, na) == (B, M, n), "add must be a dense-row (B, M, n) view"
        L.check(lib.bdm_pointwise_conv_gn_add(B, M, K, n, L.ptr(w), K, L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(x2p), L.c_ll(bs_x2), ld_x2, k1,
                                              L.ptr(bias), L.ptr(out), L.c_ll(bs_y), ld_y, L.ptr(in_p), in_s, in_g, L.ptr(in_gamma), L.ptr(in_beta),
                                              L.c_float(in_eps), og, L.ptr(out_p), L.ptr(amax), int(amax_rows), L.ptr(add), L.c_ll(bs_a), ld_a,
                                              L.stream()), "pointwise_conv_gn_add")
        return (out, stats) if out_groups else out
    packed = _pw_s3_weights(w, weight) if n >= PW_S3_MIN_COLUMNS else None
    if packed is not None:
        L.check(lib.bdm_pointwise_conv_gn_s3(B, M, K, n, L.ptr(packed), L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(x2p), L.c_ll(bs_x2), ld_x2, k1,
                                             L.ptr(bias), L.ptr(out), L.c_ll(bs_y), ld_y, L.ptr(in_p), in_s, in_g, L.ptr(in_gamma), L.ptr(in_beta),
                                             L.c_float(in_eps), og, L.ptr(out_p), L.ptr(amax), int(amax_rows), L.stream()), "pointwise_conv_gn_s3")
        return (out, stats) if out_groups else out
    L.check(lib.bdm_pointwise_conv_gn(B, M, K, n, L.ptr(w), K, L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(x2p), L.c_ll(bs_x2), ld_x2, k1,
                                      L.ptr(bias), L.ptr(out), L.c_ll(bs_y), ld_y, L.ptr(in_p), in_s, in_g, L.ptr(in_gamma), L.ptr(in_beta), L.c_float(in_eps), og,
                                      L.ptr(out_p), L.ptr(amax), int(amax_rows), L.stream()), "pointwise_conv_gn")
    return (out, stats) if out_groups else out


def group_norm_(x, gamma, beta, groups=8, eps=1e-5, swish=False, residual=None, out=None):
    """GroupNorm(groups) [+residual first] [+Swish]; in place unless `out` is given."""
    xx, B, C, l, bs_x, ld_x = _bcl(x)
    assert xx.data_ptr() == x.data_ptr(), "group_norm_ needs a dense-row view"
    if out is None:
        out = x
    _, _, _, _, bs_y, ld_y = _bcl(out)
    if residual is not None:
        rr, _, _, _, bs_r, ld_r = _bcl(residual)
        assert rr.data_ptr() == residual.data_ptr()
    else:
        bs_r, ld_r = 0, 0
    ws = workspace(L.lib().bdm_group_norm_workspace_bytes(B, groups), x.device, "gn")
    L.check(L.lib().bdm_group_norm(B, C, l, groups, L.ptr(x), L.c_ll(bs_x), ld_x, L.ptr(residual), L.c_ll(bs_r), ld_r,
                                   L.ptr(gamma), L.ptr(beta), L.c_float(eps), 1 if swish else 0, L.ptr(out),
                                   L.c_ll(bs_y), ld_y, L.ptr(ws), L.stream()), "group_norm")
    return out


def max_over_neighbors(x, out=None, fold=None):
    """max over the neighbour axis; fold = (stats, gn): x is a raw convolution output, Swish(GroupNorm(x)) is applied on the fly."""
    B, C, M, U = x.shape
    x = x.contiguous()
    if out is None:
        out = torch.empty(B, C, M, dtype=torch.float32, device=x.device)
    _, _, _, _, bs_y, ld_y = _bcl(out)
    if fold is not None:
        (p, slices, groups), gn = fold
        L.check(L.lib().bdm_max_over_neighbors_gn(B, C, M, U, L.ptr(x), L.ptr(p), slices, groups, L.ptr(gn.weight), L.ptr(gn.bias),
                                                  L.c_float(gn.eps), L.ptr(out), L.c_ll(bs_y), ld_y, L.stream()),
                "max_over_neighbors_gn")
        return out
    L.check(L.lib().bdm_max_over_neighbors(B, C, M, U, L.ptr(x), L.ptr(out), L.c_ll(bs_y), ld_y, L.stream()),
            "max_over_neighbors")
    return out


def sa_group(coords, centers, features, indices, point_major=None):
    """cat[grouping(coords) - centers, grouping(features)] -> (B, 3+C, M, U)."""
    B, _, n = coords.shape
    m, u = indices.shape[1], indices.shape[2]
    f, _, C, _, bs_f, ld_f = _bcl(features)
    out = torch.empty(B, 3 + C, m, u, dtype=torch.float32, device=coords.device)
    if point_major is None:  # the repack pays off only for the big first level (measured: tools/query_bench.py)
        point_major = B * m * u >= (1 << 18) and C <= 64
    ws = workspace(L.lib().bdm_sa_group_workspace_bytes(B, C, n), coords.device, "sa_group") if point_major else None
    L.check(L.lib().bdm_sa_group(B, C, n, m, u, L.ptr(coords), L.ptr(centers), L.ptr(f), L.c_ll(bs_f), ld_f,
                                 L.ptr(indices), L.ptr(out), L.ptr(ws), L.stream()), "sa_group")
    return out


def sa_mlp2_fusable(mlp, c_in, u):
    """Does bdm_sa_mlp2_fused cover this SharedMLP (3 + c_in inputs, 1 <= c_in <= 32, -> 32 -> 64 channels, GroupNorm(8), 32 neighbours)?"""
    layers = mlp.layers
    if len(layers) != 6 or u != 32 or not 1 <= c_in <= 32:
        return False
    (c1, g1), (c2, g2) = (layers[0], layers[1]), (layers[3], layers[4])
    return (c1.out_channels == 32 and c2.out_channels == 64 and c1.in_channels == c_in + 3 and g1.num_groups == 8 and g2.num_groups == 8
            and c1.bias is not None and c2.bias is not None)


def sa_mlp2_fused(coords, centers, features, indices, mlp):
    """max over neighbours of SharedMLP(cat[grouping(coords) - centers, grouping(features)]) -> (B, 64, M), nothing of size M x U in
    memory (bdm_sa_mlp2_fused: three recompute passes; pointnet.py:80-90 at the first level)."""
    B, _, n = coords.shape
    m, u = indices.shape[1], indices.shape[2]
    f, _, C, _, bs_f, ld_f = _bcl(features)
    dev = coords.device
    c1, g1, c2, g2 = mlp.layers[0], mlp.layers[1], mlp.layers[3], mlp.layers[4]
    rows = workspace(L.lib().bdm_sa_mlp2_fused_rows_bytes(B, C, n), dev, "sa_rows")
    S = L.lib().bdm_sa_mlp2_fused_slices(B, m)
    partial = torch.empty(2, B, 8, S, 2, dtype=torch.float64, device=dev)
    out = torch.empty(B, c2.out_channels, m, dtype=torch.float32, device=dev)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    L.check(L.lib().bdm_sa_mlp2_fused(B, C, n, m, u, c1.out_channels, c2.out_channels, L.ptr(coords), L.ptr(f), L.c_ll(bs_f), ld_f,
                                      L.ptr(centers), L.ptr(indices), L.ptr(c1.weight), L.ptr(c1.bias), L.ptr(g1.weight), L.ptr(g1.bias),
                                      L.c_float(g1.eps), L.ptr(c2.weight), L.ptr(c2.bias), L.ptr(g2.weight), L.ptr(g2.bias),
                                      L.c_float(g2.eps), 8, L.ptr(rows), L.ptr(partial[0]), L.ptr(partial[1]), L.ptr(out), L.c_ll(bs_o),
                                      ld_o, L.stream()), "sa_mlp2_fused")
    return out


def broadcast_rows(v, l, out=None):
    """v (B,C) -> (B,C,l) materialised (or written into the view `out`)."""
    B, C = v.shape
    v = v.contiguous()
    if out is None:
        out = torch.empty(B, C, l, dtype=torch.float32, device=v.device)
    _, _, _, _, bs_y, ld_y = _bcl(out)
    L.check(L.lib().bdm_broadcast_rows(B, C, l, L.ptr(v), C, L.ptr(out), L.c_ll(bs_y), ld_y, L.stream()),
            "broadcast_rows")
    return out


def xyz_rows(inputs):
    """inputs[:, :3, :] of a channel-first (B, C, N) tensor as a contiguous (B, 3, N) tensor: one library launch (a torch
    `.contiguous()` of the strided view is an elementwise kernel + a copy and a Python entry on a recorded step)."""
    v = inputs[:, :3, :]
    if v.is_contiguous() or not inputs.is_cuda:
        return v.contiguous()
    out = torch.empty(inputs.shape[0], 3, inputs.shape[2], dtype=torch.float32, device=inputs.device)
    return copy_rows(v, out)


def copy_rows(x, out):
    xx, B, C, l, bs_x, ld_x = _bcl(x)
    _, _, _, _, bs_y, ld_y = _bcl(out)
    L.check(L.lib().bdm_copy_rows(B, C, l, L.ptr(xx), L.c_ll(bs_x), ld_x, L.ptr(out), L.c_ll(bs_y), ld_y, L.stream()),
            "copy_rows")
    return out


def is_point_invariant(t):
    """True for `v[:, :, None].expand(-1, -1, N)` views (the time embedding, pvcnn.py:88)."""
    return t.dim() == 3 and t.stride(2) == 0


def materialize(t):
    if is_point_invariant(t):
        return broadcast_rows(t[:, :, 0], t.shape[2])
    return t


def cat_channels(parts):
    """torch.cat(parts, dim=1) for (B, C_i, L) parts, built with copy/broadcast kernels."""
    parts = [p for p in parts if p.shape[1] > 0]
    B, l = parts[0].shape[0], parts[0].shape[2]
    out = torch.empty(B, sum(p.shape[1] for p in parts), l, dtype=torch.float32, device=parts[0].device)
    if len(parts) == 2:  # the denoisers' cat([features, temb]): one launch
        desc = []
        for p in parts:
            if is_point_invariant(p):
                v = p[:, :, 0]
                v = v if v.stride(1) == 1 else v.contiguous()
                desc.append((v, p.shape[1], v.stride(0), 0))
            else:
                x, _, C, _, bs_x, ld_x = _bcl(p)
                desc.append((x, C, bs_x, ld_x))
        _, _, _, _, bs_y, ld_y = _bcl(out)
        (x0, c0_, bs0, ld0), (x1, c1_, bs1, ld1) = desc
        L.check(L.lib().bdm_concat2_rows(B, l, c0_, L.ptr(x0), L.c_ll(bs0), ld0, c1_, L.ptr(x1), L.c_ll(bs1), ld1, L.ptr(out),
                                         L.c_ll(bs_y), ld_y, L.stream()), "concat2_rows")
        return out
    c0 = 0
    for p in parts:
        dst = out[:, c0:c0 + p.shape[1], :]
        if is_point_invariant(p):
            broadcast_rows(p[:, :, 0], l, out=dst)
        else:
            copy_rows(p, dst)
        c0 += p.shape[1]
    return out


def transpose12(x):
    """(B, R, C) -> (B, C, R) contiguous."""
    if x.dim() == 3 and not x.is_contiguous() and x.transpose(1, 2).is_contiguous():
        return x.transpose(1, 2)  # already stored the other way round (model.get_input_with_conditioning): no copy
    x = x.contiguous()
    B, R, C = x.shape
    out = torch.empty(B, C, R, dtype=torch.float32, device=x.device)
    L.check(L.lib().bdm_transpose(B, R, C, L.ptr(x), L.ptr(out), L.stream()), "transpose")
    return out


def is_transposed_view_of(x, x_cf):
    """Is the (B, R, C) tensor `x` the no-copy transposed view of the channel-first (B, C, R) tensor `x_cf` (the case transpose12
    answers without a copy)?"""
    return (x.dim() == 3 and not x.is_contiguous() and x.transpose(1, 2).is_contiguous() and x.data_ptr() == x_cf.data_ptr()
            and tuple(x.transpose(1, 2).shape) == tuple(x_cf.shape))


def time_embedding(t, w0, b0, w2, b2):
    B, dim = t.shape[0], w0.shape[0]
    out = torch.empty(B, dim, dtype=torch.float32, device=t.device)
    if t.dtype == torch.int64 and t.is_contiguous():   # the schedulers' timesteps: cast inside the kernel (no torch operator in a recorded step)
        L.check(L.lib().bdm_time_embedding_i64(B, dim, L.ptr(t), L.ptr(w0), L.ptr(b0), L.ptr(w2), L.ptr(b2), L.ptr(out), L.stream()),
                "time_embedding_i64")
        return out
    tf = t.to(torch.float32).contiguous()
    L.check(L.lib().bdm_time_embedding(B, dim, L.ptr(tf), L.ptr(w0), L.ptr(b0), L.ptr(w2), L.ptr(b2), L.ptr(out),
                                       L.stream()), "time_embedding")
    return out


def voxel_coords(coords, r, eps=0.0):
    B, _, n = coords.shape
    coords = coords.contiguous()
    nc = torch.empty(B, 3, n, dtype=torch.float32, device=coords.device)
    vc = torch.empty(B, 3, n, dtype=torch.int32, device=coords.device)
    L.check(L.lib().bdm_voxel_coords(B, n, int(r), L.c_float(eps), L.ptr(coords), L.ptr(nc), L.ptr(vc), L.stream()),
            "voxel_coords")
    return nc, vc


def avg_voxelize(features, vox_coords, r, with_row_occupancy=False):
    f = features.contiguous()
    B, C, n = f.shape
    dev = f.device
    out = torch.empty(B, C, r ** 3, dtype=torch.float32, device=dev)
    ind = torch.empty(B, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, r ** 3, dtype=torch.int32, device=dev)
    ws = workspace(L.lib().bdm_voxelize_workspace_bytes(B, n, r), dev, "vox")
    L.check(L.lib().bdm_avg_voxelize_forward(B, C, n, r, L.ptr(f), L.ptr(vox_coords), L.ptr(out), L.ptr(ind), L.ptr(cnt),
                                             L.ptr(ws), L.stream()), "avg_voxelize")
    if with_row_occupancy:
        rowocc = torch.empty(B, r * r, dtype=torch.uint8, device=dev)
        L.check(L.lib().bdm_voxel_row_occupancy(B, r, L.ptr(cnt), L.ptr(rowocc), L.stream()), "voxel_row_occupancy")
        return out, rowocc
    return out


def se_gate(x, w1, w2, fused=False):
    B, C = x.shape[:2]
    x = x.contiguous()
    l = x.numel() // (B * C)
    mean = torch.empty(B, C, dtype=torch.float32, device=x.device)
    gate = torch.empty(B, C, dtype=torch.float32, device=x.device)
    # fused=True is the one-launch form (last workgroup of a shape runs the FC layers).  Measured on MI355X: it makes the
    # forward 0.9 ms SLOWER -- its device-scope release fence writes back the XCD's whole L2 (full of the convolution's
    # dirty output) once per workgroup -- so the two-launch form stays the default.
    counters = _zero_counters(B, x.device) if fused else None
    L.check(L.lib().bdm_se_gate(B, C, w1.shape[0], l, L.ptr(x), L.ptr(w1), L.ptr(w2), L.ptr(mean), L.ptr(gate),
                                L.ptr(counters), L.stream()), "se_gate")
    return gate


_counter_cache = {}


def _zero_counters(n, device):
    """int32 counters that kernels find zero and leave zero (one buffer per device and stream)."""
    dev = torch.device(device)
    key = (str(device), torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch._C._cuda_getDevice()))
    buf = _counter_cache.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 256), dtype=torch.int32, device=device)
        _counter_cache[key] = buf
    return buf


def devoxelize_gate_add(norm_coords, grid, r, gate=None, add=None, out=None):
    B, C = grid.shape[:2]
    n = norm_coords.shape[2]
    grid = grid.contiguous()
    if out is None:
        out = torch.empty(B, C, n, dtype=torch.float32, device=grid.device)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    if add is not None:
        aa, _, _, _, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr()
    else:
        bs_a, ld_a = 0, 0
    L.check(L.lib().bdm_devoxelize_gate_add(B, C, n, int(r), L.ptr(norm_coords), L.ptr(grid), L.ptr(gate), L.ptr(add),
                                            L.c_ll(bs_a), ld_a, L.ptr(out), L.c_ll(bs_o), ld_o, L.stream()),
            "devoxelize_gate_add")
    return out


ATTENTION_IMPL = os.environ.get("BDM_ATTENTION", "fp16x3")  # "bf16x6": six-product kernel; "fp32": the fp32-input MFMA flash kernel


def amax_slots(device, n):
    """n zeroed floats (consecutive slots of the per-stream ring of _amax_slot; a buffer of their own for large n: the zero-fill
    is then part of the step, which a recorded step replays like any other launch)"""
    if n > 64:
        return torch.zeros(n, dtype=torch.float32, device=device)
    while True:
        first = _amax_slot(device)
        ring = _amax_rings[(str(device), torch._C._cuda_getCurrentRawStream(first.device.index))]
        if ring[1] - 1 + n <= ring[0].shape[0]:
            start = ring[1] - 1
            ring[1] = start + n
            return ring[0][start:start + n]
        ring[1] = ring[0].shape[0]  # not enough room left in this ring: take a fresh one


def attention_h2_ok(C, l):
    return ATTENTION_IMPL == "fp16x3" and l > 64 and C <= 64 and C % 32 == 0


# Key ranges of the fp16x3 attention (csrc/attention_h2.hip): None = the library's recommendation for the launch's (shapes, positions) -- 4 / 2 / 1
# ranges at 4096 positions for < 4 / < 16 / more shapes, i.e. like the convolutions' tile forms a shape's bits depend on the per-launch batch
# (tests pin the count where they compare batch sizes) -- or a fixed count 1 .. 8 from BDM_ATTN_KSPLIT, read ONCE here.
ATTN_KSPLIT = int(os.environ["BDM_ATTN_KSPLIT"]) if os.environ.get("BDM_ATTN_KSPLIT", "") in tuple("12345678") else None


def attention_key_slices(B, l):
    return int(ATTN_KSPLIT) if ATTN_KSPLIT else int(L.lib().bdm_attention_h2_key_slices(B, l))


def attention_core(qkv, C, impl=None, amax=None):
    """qkv (B, 3C, L): rows [0,C) = q, [C,2C) = k, [2C,3C) = v  ->  (B, C, L).  amax (3 floats per shape: max |q|, |k|, |v|,
    from pointwise_conv_gn) selects the fp16x3 kernel."""
    B, _, l = qkv.shape
    out = torch.empty(B, C, l, dtype=torch.float32, device=qkv.device)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    lib = L.lib()
    if amax is not None:
        ks = attention_key_slices(B, l)
        nbytes = lib.bdm_attention_h2_workspace_bytes(B, C, l, ks)
        ws = workspace(nbytes, qkv.device, "attention_h2")
        L.check(lib.bdm_attention_core_h2(B, C, l, L.ptr(q), L.ptr(k), L.ptr(v), L.c_ll(qkv.stride(0)), qkv.stride(1), L.ptr(amax),
                                          L.ptr(out), L.c_ll(C * l), l, L.ptr(ws), ctypes.c_size_t(ws.numel() * ws.element_size()), ks, L.stream()),
                "attention_core_h2")
        return out
    nbytes = lib.bdm_attention_workspace_bytes(B, C, l) if (impl or ATTENTION_IMPL) in ("bf16x6", "fp16x3") else 0
    ws = workspace(nbytes, qkv.device, "attention") if nbytes else None
    L.check(lib.bdm_attention_core(B, C, l, L.ptr(q), L.ptr(k), L.ptr(v), L.c_ll(qkv.stride(0)), qkv.stride(1),
                                   L.ptr(out), L.c_ll(C * l), l, L.ptr(ws), L.stream()), "attention_core")
    return out


def conv3d_pack(weight):
    cout, cin = weight.shape[:2]
    w = weight.contiguous()
    packed = torch.empty(27, cin, cout, dtype=torch.float32, device=w.device)
    L.check(L.experimental("bdm_conv3d_pack_weights")(cout, cin, L.ptr(w), L.ptr(packed), L.stream()), "conv3d_pack_weights")
    return packed


def conv3d(x, packed_w, bias, r, rowocc=None):
    """x (B, Cin, r^3) contiguous -> (B, Cout, r^3).  rowocc: row-occupancy flags of a freshly voxelised input."""
    x = x.contiguous()
    B, cin = x.shape[:2]
    cout = packed_w.shape[2]
    y = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=x.device)
    L.check(L.experimental("bdm_conv3d_3x3x3_sparse")(B, cin, cout, int(r), L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(rowocc),
                                            L.ptr(y), L.stream()), "conv3d")
    return y


# ---- bf16x6 ("S3") convolution path: fp32 accuracy on the bf16 matrix cores (csrc/conv3d_s3.hip) -------------------
def conv3d_s3_pack(weight):
    import ctypes
    cout, cin = weight.shape[:2]
    w = weight.contiguous()
    packed = torch.empty(L.lib().bdm_conv3d_s3_weight_elems(cout, cin), dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().bdm_conv3d_s3_pack_weights(cout, cin, L.ptr(w), L.ptr(packed), L.stream()), "conv3d_s3_pack_weights")
    return packed


def to_s3(x, gn=None, swish=False):
    """x (B, C, V) fp32 contiguous -> S3 (B, ceil(C/8), 3, V, 8) bf16, optionally through GroupNorm(+Swish)."""
    x = x.contiguous()
    B, C = x.shape[:2]
    V = x.numel() // (B * C)
    out = torch.empty(B, (C + 7) // 8, 3, V, 8, dtype=torch.bfloat16, device=x.device)
    if gn is not None:
        ws = workspace(L.lib().bdm_group_norm_workspace_bytes(B, gn.num_groups), x.device, "gn")
        L.check(L.lib().bdm_group_norm_to_s3(B, C, V, gn.num_groups, L.ptr(x), L.ptr(gn.weight), L.ptr(gn.bias),
                                             L.c_float(gn.eps), 1 if swish else 0, L.ptr(out), L.ptr(ws), L.stream()),
                "group_norm_to_s3")
    else:
        L.check(L.lib().bdm_group_norm_to_s3(B, C, V, 0, L.ptr(x), L.ptr(None), L.ptr(None), L.c_float(0.0), 0, L.ptr(out),
                                             L.ptr(None), L.stream()), "to_s3")
    return out


def avg_voxelize_s3(features, vox_coords, r):
    f, B, C, n, bs_f, ld_f = _bcl(features)
    dev = f.device
    out = torch.empty(B, (C + 7) // 8, 3, r ** 3, 8, dtype=torch.bfloat16, device=dev)
    ind = torch.empty(B, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, r ** 3, dtype=torch.int32, device=dev)
    ws = workspace(L.lib().bdm_voxelize_workspace_bytes(B, n, r), dev, "vox")
    L.check(L.lib().bdm_avg_voxelize_s3(B, C, n, r, L.ptr(f), L.c_ll(bs_f), ld_f, L.ptr(vox_coords), L.ptr(out), L.ptr(ind),
                                        L.ptr(cnt), L.ptr(ws), L.stream()), "avg_voxelize_s3")
    return out


def conv3d_s3(x_s3, packed_w, bias, cin, cout, r):
    """x_s3 (B, ceil(cin/8), 3, r^3, 8) bf16 -> (B, cout, r^3) fp32."""
    B = x_s3.shape[0]
    y = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=x_s3.device)
    L.check(L.lib().bdm_conv3d_3x3x3_s3(B, cin, cout, int(r), L.ptr(x_s3), L.ptr(packed_w), L.ptr(bias), L.ptr(y), L.stream()),
            "conv3d_s3")
    return y


# ---- fp16x3 ("H2") convolution path: two fp16 terms per operand, three MFMA products (csrc/conv3d_h2.hip) ------------
def conv3d_h2_pack(weight):
    """-> (packed fp16 weights, inv_scale (Cout,) fp32)."""
    cout, cin = weight.shape[:2]
    w = weight.contiguous()
    lib = L.lib()
    packed = torch.empty(lib.bdm_conv3d_h2_weight_elems(cout, cin), dtype=torch.float16, device=w.device)
    scale = torch.empty(cout, dtype=torch.float32, device=w.device)
    inv_scale = torch.empty(cout, dtype=torch.float32, device=w.device)
    L.check(lib.bdm_conv3d_h2_pack_weights(cout, cin, L.ptr(w), L.ptr(packed), L.ptr(scale), L.ptr(inv_scale), L.stream()),
            "conv3d_h2_pack_weights")
    return packed, inv_scale


def _pow2_below(v):
    import math
    return 2.0 ** math.floor(math.log2(v))


def h2_activation_scale(gn, sigmas=64.0):
    """Power-of-two scale for GroupNorm(+Swish) outputs: |y| <= |gamma| |z| + |beta|, so that a `sigmas`-sigma value of
    the widest channel still fits fp16 (beyond that the split saturates).  Depends on the parameters only: cached per
    (module, parameter version), one small device->host read when the weights change."""
    sig = (gn.weight._version, gn.bias._version, gn.weight.data_ptr(), sigmas)
    hit = getattr(gn, "_bdm_h2_scale", None)  # lives on the module: no id() reuse after garbage collection
    if hit is None or hit[0] != sig:
        bound = float((gn.weight.detach().abs() * sigmas + gn.bias.detach().abs()).max())
        hit = (sig, _pow2_below(32768.0 / max(bound, 1e-30)))
        gn._bdm_h2_scale = hit
    return hit[1]


_sat_registry = {}  # device -> [flags tensor (int32, 256 slots), [weakref(owner) per slot]]
_sat_epoch = 0      # bumped whenever a layer is switched to the bf16x6 route: recorded steps (launch tape, hipGraph) key on it


def saturation_epoch():
    """Changes when poll_h2_saturation() re-routes a layer.  A recorded reverse step has the fp16x3 kernels of that layer baked
    in, so every cache of recorded steps carries this number in its key and re-records on the new route."""
    return _sat_epoch


def saturation_slot(owner, device):
    """One sticky device word per fp16x3 layer (`owner` = its PVConv): set by to_h2 when a value left fp16's range."""
    import weakref
    key = str(device)
    reg = _sat_registry.get(key)
    if reg is None:
        reg = _sat_registry[key] = [torch.zeros(1024, dtype=torch.int32, device=device), []]
    slots = getattr(owner, "_bdm_sat_slots", None)
    if slots is None:
        slots = owner._bdm_sat_slots = {}
    idx = slots.get(key)
    if idx is None:
        if len(reg[1]) < reg[0].numel():
            idx = len(reg[1])
            reg[1].append(weakref.ref(owner))
        else:  # recycle the slot of a layer that no longer exists
            idx = next((i for i, r in enumerate(reg[1]) if r() is None), None)
            if idx is None:
                raise L.BdmHipError(f"more than {reg[0].numel()} live fp16x3 layers on one device")
            reg[0][idx:idx + 1].zero_()
            reg[1][idx] = weakref.ref(owner)
        slots[key] = idx
    return reg[0][idx:idx + 1]


def poll_h2_saturation():
    """Host check of the saturation words (ONE small device->host copy; the samplers call it once per trajectory, never per
    step).  Every flagged layer is switched to the bf16x6 kernels (exact split, no range limit) for all later calls and a
    warning names it: the trajectory that just finished used clamped activations in that layer."""
    import warnings
    global _sat_epoch
    hit = []
    for key, (flags, owners) in _sat_registry.items():
        if not owners:
            continue
        host = flags[:len(owners)].cpu()
        for i, ref in enumerate(owners):
            m = ref()
            if m is not None and int(host[i]) != 0 and not getattr(m, "h2_saturated", False):
                m.h2_saturated = True
                hit.append(m)
        if hit:
            flags.zero_()
    if hit:
        _sat_epoch += 1
        warnings.warn(f"fp16x3 convolution input saturated (|scale * y| > 65504) in {len(hit)} layer(s): "
                      f"{[getattr(m, 'bdm_name', type(m).__name__) for m in hit]}; the trajectory that just finished used clamped "
                      "activations there.  These layers now run the bf16x6 kernels (no range limit).", stacklevel=2)
    return hit


def to_h2(x, gn=None, swish=False, scale=None, saturated=None, stats=None):
    """x (B, C, V) fp32 contiguous -> (H2 tensor (B, ceil(C/8), 2, V, 8) fp16 of scale * [swish(group_norm(x))], 1 / scale).
    scale: power of two; default from the GroupNorm parameters, or (no GroupNorm: test path, host sync) from max |x|.
    saturated: optional 1-element int32 device tensor, OR-ed with 1 when a scaled value left fp16's range."""
    if isinstance(x, CompactGrid):            # first convolution in compact form: rows of the dilated voxels + bias everywhere else
        assert gn is not None and stats is not None, "a compact grid comes with its producer's statistics"
        B, C, V = x.rows.shape[0], x.channels, x.plan.r ** 3
        s = float(scale if scale is not None else h2_activation_scale(gn))
        out = torch.empty(B, (C + 7) // 8, 2, V, 8, dtype=torch.float16, device=x.rows.device)
        partial, slices, groups = stats
        L.check(L.lib().bdm_group_norm_to_h2_stats_compact(B, C, V, groups, L.ptr(x.rows), x.plan.n_dil_max, L.ptr(x.plan.dil_index),
                                                           L.ptr(x.bias), L.ptr(gn.weight), L.ptr(gn.bias), L.c_float(gn.eps),
                                                           1 if swish else 0, L.c_float(s), L.ptr(out), L.ptr(partial), slices,
                                                           L.ptr(saturated), L.stream()), "group_norm_to_h2_stats_compact")
        return out, 1.0 / s
    x = x.contiguous()
    B, C = x.shape[:2]
    V = x.numel() // (B * C)
    out = torch.empty(B, (C + 7) // 8, 2, V, 8, dtype=torch.float16, device=x.device)
    if gn is not None and stats is not None:  # statistics left by the producer (sparse gather): no pass over x for them
        scale = h2_activation_scale(gn) if scale is None else scale
        partial, slices, groups = stats
        assert groups == gn.num_groups
        L.check(L.lib().bdm_group_norm_to_h2_stats(B, C, V, groups, L.ptr(x), L.ptr(gn.weight), L.ptr(gn.bias), L.c_float(gn.eps),
                                                   1 if swish else 0, L.c_float(scale), L.ptr(out), L.ptr(partial), slices,
                                                   L.ptr(saturated), L.stream()), "group_norm_to_h2_stats")
    elif gn is not None:
        scale = h2_activation_scale(gn) if scale is None else scale
        ws = workspace(L.lib().bdm_group_norm_workspace_bytes(B, gn.num_groups), x.device, "gn")
        L.check(L.lib().bdm_group_norm_to_h2(B, C, V, gn.num_groups, L.ptr(x), L.ptr(gn.weight), L.ptr(gn.bias),
                                             L.c_float(gn.eps), 1 if swish else 0, L.c_float(scale), L.ptr(out), L.ptr(ws),
                                             L.ptr(saturated), L.stream()), "group_norm_to_h2")
    else:
        scale = _pow2_below(32768.0 / max(float(x.abs().max()), 1e-30)) if scale is None else scale
        L.check(L.lib().bdm_group_norm_to_h2(B, C, V, 0, L.ptr(x), L.ptr(None), L.ptr(None), L.c_float(0.0), 0,
                                             L.c_float(scale), L.ptr(out), L.ptr(None), L.ptr(saturated), L.stream()), "to_h2")
    return out, 1.0 / scale


def conv3d_h2(x_h2, packed, bias, cin, cout, r):
    """x_h2 = to_h2(...) -> ((B, ceil(cin/8), 2, r^3, 8) fp16, 1/scale); packed = conv3d_h2_pack(w) -> (B, cout, r^3) fp32."""
    (xh, x_inv_scale), (packed_w, inv_scale) = x_h2, packed
    B = xh.shape[0]
    y = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=xh.device)
    L.check(L.lib().bdm_conv3d_3x3x3_h2(B, cin, cout, int(r), L.ptr(xh), L.c_float(x_inv_scale), L.ptr(packed_w),
                                        L.ptr(inv_scale), L.ptr(bias), L.ptr(y), L.stream()), "conv3d_h2")
    return y


def conv3d_h2_gn(x_h2, packed, bias, cin, cout, r, groups=8):
    """conv3d_h2 that also leaves the GroupNorm(groups) statistics of its output: (y raw, (partials workspace, slices))."""
    (xh, x_inv_scale), (packed_w, inv_scale) = x_h2, packed
    B = xh.shape[0]
    y = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=xh.device)
    ws = torch.empty(L.lib().bdm_group_norm_workspace_bytes(B, groups), dtype=torch.uint8, device=xh.device)
    slices = ctypes.c_int(0)
    L.check(L.lib().bdm_conv3d_3x3x3_h2_gn(B, cin, cout, int(r), L.ptr(xh), L.c_float(x_inv_scale), L.ptr(packed_w),
                                           L.ptr(inv_scale), L.ptr(bias), L.ptr(y), groups, L.ptr(ws), ctypes.byref(slices),
                                           L.stream()), "conv3d_h2_gn")
    return y, (ws, slices.value)


def se_gate_gn(x, stats, gn, w1, w2, pf=None, n_points=0):
    """SE gate of swish(group_norm(x)) evaluated from the raw grid x and the producer's statistics: (gate (B,C), coef (B,C,2)).
    pf = ((partial, slices, groups), gn) of the PVConv's point branch: also returns that GroupNorm's affine forms (B,C,2)."""
    ws, slices = stats
    B, C = x.shape[:2]
    l = x.numel() // (B * C)
    mean = torch.empty(B, C, dtype=torch.float32, device=x.device)
    coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    gate = torch.empty(B, C, dtype=torch.float32, device=x.device)
    if pf is not None:
        (pp, ps, pg), pgn = pf
        pf_coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        L.check(L.lib().bdm_se_gate_gn_pf(B, C, w1.shape[0], l, gn.num_groups, L.ptr(x), L.ptr(ws), slices, L.ptr(gn.weight),
                                          L.ptr(gn.bias), L.c_float(gn.eps), L.ptr(w1), L.ptr(w2), L.ptr(mean), L.ptr(coef), L.ptr(gate),
                                          L.ptr(pp), ps, pg, int(n_points), L.ptr(pgn.weight), L.ptr(pgn.bias), L.c_float(pgn.eps),
                                          L.ptr(pf_coef), L.stream()), "se_gate_gn_pf")
        return gate, coef, pf_coef
    L.check(L.lib().bdm_se_gate_gn(B, C, w1.shape[0], l, gn.num_groups, L.ptr(x), L.ptr(ws), slices, L.ptr(gn.weight), L.ptr(gn.bias),
                                   L.c_float(gn.eps), L.ptr(w1), L.ptr(w2), L.ptr(mean), L.ptr(coef), L.ptr(gate), L.stream()),
            "se_gate_gn")
    return gate, coef


def se_means_gn(x, stats, gn, pf=None, n_points=0):
    """Per-channel means of swish(group_norm(x)) + the rows' affine forms, from the raw grid and the producer's statistics:
    (mean (B,C), coef (B,C,2)).  The SE block's FC layers are then evaluated inside the devoxelisation kernel.
    pf = ((partial, slices, groups), gn) of the PVConv's point branch: also returns that GroupNorm's affine forms (B,C,2)."""
    ws, slices = stats
    B, C = x.shape[:2]
    l = x.numel() // (B * C)
    mean = torch.empty(B, C, dtype=torch.float32, device=x.device)
    coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    if pf is not None:
        (pp, ps, pg), pgn = pf
        pf_coef = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        L.check(L.lib().bdm_se_gate_gn_pf(B, C, 1, l, gn.num_groups, L.ptr(x), L.ptr(ws), slices, L.ptr(gn.weight), L.ptr(gn.bias),
                                          L.c_float(gn.eps), L.ptr(None), L.ptr(None), L.ptr(mean), L.ptr(coef), L.ptr(None),
                                          L.ptr(pp), ps, pg, int(n_points), L.ptr(pgn.weight), L.ptr(pgn.bias), L.c_float(pgn.eps),
                                          L.ptr(pf_coef), L.stream()), "se_means_gn_pf")
        return mean, coef, pf_coef
    L.check(L.lib().bdm_se_gate_gn(B, C, 1, l, gn.num_groups, L.ptr(x), L.ptr(ws), slices, L.ptr(gn.weight), L.ptr(gn.bias),
                                   L.c_float(gn.eps), L.ptr(None), L.ptr(None), L.ptr(mean), L.ptr(coef), L.ptr(None), L.stream()),
            "se_means_gn")
    return mean, coef


def devoxelize_gn_se_add(norm_coords, grid, coef, r, mean, w1, w2, add=None):
    B, C = grid.shape[:2]
    n = norm_coords.shape[2]
    out = torch.empty(B, C, n, dtype=torch.float32, device=grid.device)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    if add is not None:
        aa, _, _, _, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr()
    else:
        bs_a, ld_a = 0, 0
    L.check(L.lib().bdm_devoxelize_gn_se_add(B, C, n, int(r), L.ptr(norm_coords), L.ptr(grid), L.ptr(coef), L.ptr(mean), w1.shape[0],
                                             L.ptr(w1), L.ptr(w2), L.ptr(add), bs_a, ld_a, L.ptr(out), bs_o, ld_o, L.stream()),
            "devoxelize_gn_se_add")
    return out


def devoxelize_gn_gate_add(norm_coords, grid, coef, r, gate=None, add=None, add_coef=None):
    B, C = grid.shape[:2]
    n = norm_coords.shape[2]
    out = torch.empty(B, C, n, dtype=torch.float32, device=grid.device)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    if add is not None:
        aa, _, _, _, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr()
    else:
        bs_a, ld_a = 0, 0
    if add_coef is not None:  # add is the point branch's raw convolution output; its GroupNorm + Swish is applied in the kernel
        L.check(L.lib().bdm_devoxelize_gn_gate_add_pf(B, C, n, int(r), L.ptr(norm_coords), L.ptr(grid), L.ptr(coef), L.ptr(gate),
                                                      L.ptr(add), bs_a, ld_a, L.ptr(add_coef), L.ptr(out), bs_o, ld_o, L.stream()),
                "devoxelize_gn_gate_add_pf")
        return out
    L.check(L.lib().bdm_devoxelize_gn_gate_add(B, C, n, int(r), L.ptr(norm_coords), L.ptr(grid), L.ptr(coef), L.ptr(gate), L.ptr(add),
                                               bs_a, ld_a, L.ptr(out), bs_o, ld_o, L.stream()), "devoxelize_gn_gate_add")
    return out


# ---- PVConv glue on the small voxel grids (csrc/pvconv_small.hip) ------------------------------------------------------------------
# Small voxel grids (8^3 levels; round 5):
#   "tail" (default): where the NEXT PVConv of a stage shares the voxel plan, the tail of a PVConv (SE FC layers + GroupNorm-2 + Swish + gate
#            + devoxelisation + point branch) also leaves that PVConv's first-convolution operand (csrc/pvconv_small.hip): -13 us per module;
#   "1":     the tail kernel everywhere + the gather + GroupNorm-1 + split kernel (csrc/experimental/, EXPERIMENTAL=1 builds): measured slower;
#   "0":     the operator chain.
_SMALL_GLUE_MODE = os.environ.get("BDM_SMALL_GLUE", "tail")
SMALL_GLUE = _SMALL_GLUE_MODE in ("1", "tail", "tail_all")
SMALL_GLUE_TAIL_ONLY = _SMALL_GLUE_MODE == "tail"   # only the tail, and only where a next PVConv takes its operand
SMALL_GLUE_NO_GATHER = _SMALL_GLUE_MODE in ("tail", "tail_all")   # ("tail_all": the tail kernel on every small-grid PVConv, A/B only)


def small_grid_tail_ok(r, c, n):
    """bdm_pvconv_tail_small covers it: a slab's cells (8 channels x r^3) + its features (8 x n) fit LDS comfortably"""
    return SMALL_GLUE and c % 8 == 0 and (r ** 3) % 4 == 0 and (8 * r ** 3 + 8 * n + c + 64) * 4 <= 64 * 1024


def small_grid_gather_ok(r, cout, groups):
    cg = cout // groups if groups and cout % groups == 0 else 0
    return SMALL_GLUE and not SMALL_GLUE_NO_GATHER and cg >= 8 and cg % 8 == 0 and cg <= 64 and 256 % (cg // 4) == 0 and (cg * (r ** 3 + 1) + r ** 3) * 4 <= 150 * 1024


def h2_sum_scale(gns, sigmas=64.0):
    """Power-of-two scale for a SUM of GroupNorm(+Swish) outputs (|Swish(y)| <= |y|; convex combinations and gates in [0, 1] do not
    raise the bound): the fused features of a PVConv = devoxelised Swish(GroupNorm-2) * gate + Swish(GroupNorm(point branch)), and every
    mean of them over the points of a cell.  From the parameters only (cached per parameter version), as h2_activation_scale."""
    sig = tuple((g.weight._version, g.bias._version, g.weight.data_ptr()) for g in gns) + (sigmas,)
    owner = gns[0]
    hit = getattr(owner, "_bdm_h2_sum_scale", None)
    if hit is None or hit[0] != sig:
        bound = sum(float((g.weight.detach().abs() * sigmas + g.bias.detach().abs()).max()) for g in gns)
        hit = (sig, _pow2_below(32768.0 / max(bound, 1e-30)))
        owner._bdm_h2_sum_scale = hit
    return hit[1]


class VoxelRows:
    """First-convolution operand of a PVConv, formed by the PREVIOUS PVConv's tail on the same voxel plan (bdm_pvconv_tail_small):
    xh (B, C/8, 2, n_max, 8) fp16 records + amax (B) for bdm_sparse_conv_gemm_h2; valid for `plan` and `features` only."""
    __slots__ = ("plan", "xh", "amax", "channels")

    def __init__(self, plan, xh, amax, channels):
        self.plan, self.xh, self.amax, self.channels = plan, xh, amax, channels


def pvconv_tail_small(norm_coords, grid, coef, mean, w1, w2, r, add=None, add_coef=None, head=None):
    """SE gate + Swish(GroupNorm-2) + devoxelisation + point branch in one launch (-> out (B, C, n)); head = (plan, x_scale, saturated):
    also the next PVConv's operand on `plan` -> (out, VoxelRows)."""
    B, C = grid.shape[:2]
    n = norm_coords.shape[2]
    out = torch.empty(B, C, n, dtype=torch.float32, device=grid.device)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    if add is not None:
        aa, _, _, _, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr()
    else:
        bs_a, ld_a = 0, 0
    rows = None
    if head is not None:
        plan, x_scale, saturated = head
        xh = torch.empty(B, C // 8, 2, plan.n_max, 8, dtype=torch.float16, device=grid.device)
        amax = torch.empty(B, dtype=torch.float32, device=grid.device)
        rows = VoxelRows(plan, xh, amax, C)
        L.check(L.lib().bdm_pvconv_tail_small(B, C, n, int(r), w1.shape[0], L.ptr(norm_coords), L.ptr(grid), L.ptr(coef), L.ptr(mean),
                                              L.ptr(w1), L.ptr(w2), L.ptr(add), bs_a, ld_a, L.ptr(add_coef), L.ptr(out), bs_o, ld_o,
                                              L.ptr(plan.cnt), L.ptr(plan.ws), L.ptr(plan.occ_list), L.ptr(plan.n_occ), plan.n_max,
                                              L.c_float(x_scale), L.ptr(xh), L.ptr(amax), L.ptr(saturated), L.stream()), "pvconv_tail_small")
        return out, rows
    L.check(L.lib().bdm_pvconv_tail_small(B, C, n, int(r), w1.shape[0], L.ptr(norm_coords), L.ptr(grid), L.ptr(coef), L.ptr(mean),
                                          L.ptr(w1), L.ptr(w2), L.ptr(add), bs_a, ld_a, L.ptr(add_coef), L.ptr(out), bs_o, ld_o,
                                          L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), 0, L.c_float(0.0), L.ptr(None), L.ptr(None),
                                          L.ptr(None), L.stream()), "pvconv_tail_small")
    return out, None


# ---- sparse first convolution of a PVConv (csrc/sparse_conv.hip) ----------------------------------------------------
def sparse_conv_pack(weight):
    cout, cin = weight.shape[:2]
    wt = torch.empty(cin, 27 * cout, dtype=torch.float32, device=weight.device)
    L.check(L.lib().bdm_sparse_conv_pack_weights(cout, cin, L.ptr(weight.contiguous()), L.ptr(wt), L.stream()),
            "sparse_conv_pack_weights")
    return wt


def sparse_conv_pack_s3(weight):
    """(Cout, Cin, 3,3,3) fp32 -> pre-split bf16 GEMM operand (ceil(Cin/8), 3, 27*Cout, 8)."""
    cout, cin = weight.shape[:2]
    lib = L.lib()
    ws = torch.empty(lib.bdm_sparse_conv_s3_weight_elems(cout, cin), dtype=torch.bfloat16, device=weight.device)
    L.check(lib.bdm_sparse_conv_pack_weights_s3(cout, cin, L.ptr(weight.contiguous()), L.ptr(ws), L.stream()),
            "sparse_conv_pack_weights_s3")
    return ws


def sparse_conv_pack_fused(weight):
    """(Cout, Cin, 3,3,3) fp32 -> (fp16 records [ceil(Cin/8)][27][2][Cout][8], inv_scale (Cout,)) for bdm_sparse_conv_fused."""
    cout, cin = weight.shape[:2]
    lib = L.lib()
    packed = torch.empty(L.experimental("bdm_sparse_conv_fused_weight_elems")(cout, cin), dtype=torch.float16, device=weight.device)
    scale = torch.empty(cout, dtype=torch.float32, device=weight.device)
    inv_scale = torch.empty(cout, dtype=torch.float32, device=weight.device)
    L.check(L.experimental("bdm_sparse_conv_fused_pack_weights")(cout, cin, L.ptr(weight.contiguous()), L.ptr(packed), L.ptr(scale),
                                                   L.ptr(inv_scale), L.stream()), "sparse_conv_fused_pack_weights")
    return "fused", packed, inv_scale


def sparse_conv_pack_h2(weight):
    """(Cout, Cin, 3,3,3) fp32 -> ("h2", fp16 records [ceil(Cin/8)][2][27*Cout][8], inv_scale (Cout,)) for bdm_sparse_conv_gemm_h2."""
    cout, cin = weight.shape[:2]
    lib = L.lib()
    packed = torch.empty(lib.bdm_sparse_conv_h2_weight_elems(cout, cin), dtype=torch.float16, device=weight.device)
    scale = torch.empty(cout, dtype=torch.float32, device=weight.device)
    inv_scale = torch.empty(cout, dtype=torch.float32, device=weight.device)
    L.check(lib.bdm_sparse_conv_pack_weights_h2(cout, cin, L.ptr(weight.contiguous()), L.ptr(packed), L.ptr(scale), L.ptr(inv_scale),
                                                L.stream()), "sparse_conv_pack_weights_h2")
    return "h2", packed, inv_scale


SPARSE_Y_BYTES = 256 * 2 ** 20
_amax_rings = {}
# != 0 while a step is being recorded for replay (hipGraph capture or launch tape: model.static_step).  A replayed step finds
# its amax slots as the previous replay left them, so the recording must contain the zero-fill: it never takes slots of a
# ring that was zeroed outside it (and the eager path never takes slots of a recorded ring).
_static_epoch = 0
_static_epochs = 0


class static_step:
    """`with ops.static_step():` around the capture / recording of a step that will be replayed on the same buffers."""

    def __enter__(self):
        global _static_epoch, _static_epochs
        _static_epochs += 1
        _static_epoch = _static_epochs

    def __exit__(self, *exc):
        global _static_epoch
        _static_epoch = 0
        return False


def _amax_slot(device):
    """One zeroed float per sparse convolution call (its activation-scale maximum): slots of a per-(device, stream) ring that
    is refilled with zeros in ONE launch every 256 calls, so the convolution itself needs no memset."""
    dev = torch.device(device)
    key = (str(device), torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch._C._cuda_getDevice()))
    ring = _amax_rings.get(key)
    if ring is None or ring[1] >= ring[0].shape[0] or ring[2] != _static_epoch:
        # a NEW buffer (not an in-place refill): slots handed out earlier may still be read by enqueued kernels
        ring = [torch.zeros(256, dtype=torch.float32, device=device), 0, _static_epoch]
        _amax_rings[key] = ring
    slot = ring[0][ring[1]:ring[1] + 1]
    ring[1] += 1
    return slot


class VoxelPlan:
    """Everything that depends on (coords, r) only: normalised / integer voxel coordinates, per-voxel point lists,
    occupied-cell compaction and row occupancy.  PVConvs of one level share it (exactly the same values)."""
    __slots__ = ("r", "n", "n_max", "coords", "norm_coords", "vox_coords", "ind", "cnt", "ws", "occ_index", "occ_list", "n_occ", "rowocc",
                 "ready", "stream", "dil_list", "dil_index", "plane_start", "tile_start", "n_dil_max",
                 "d2_list", "d2_index", "d2_tiles", "d2_class_count", "tile", "d2_tile")


_plan_cache = {}


def clear_plan_cache():
    _plan_cache.clear()


def voxel_plan(coords, r, eps=0.0, dilate=False):
    """dilate: also build the dilated voxel list of the compact first convolution now, on this stream (plan_dilation; a consumer that
    finds it missing builds it on ITS stream)."""
    key = (coords.data_ptr(), coords._version, tuple(coords.shape), int(r))
    p = _plan_cache.get(key)
    if p is not None:
        ready = getattr(p, "ready", None)
        if ready is not None:  # planned ahead on a side stream (pvcnn.plan_sampling_chain): the current stream waits for it
            if torch.cuda.current_stream(coords.device) != getattr(p, "stream", None):
                tape.wait_event(ready)
                p.ready = None
        return p
    lib = L.lib()
    B, _, n = coords.shape
    dev = coords.device
    r3 = r ** 3
    p = VoxelPlan()
    p.ready = None
    p.r, p.n, p.n_max = int(r), n, min(n, r3)
    p.coords = coords   # (the contiguous (B, 3, n) tensor the plan describes: consumers that need the xyz rows again take them from here)
    p.norm_coords, p.vox_coords = voxel_coords(coords, r, eps)
    p.ind = torch.empty(B, n, dtype=torch.int32, device=dev)
    p.cnt = torch.empty(B, r3, dtype=torch.int32, device=dev)
    p.ws = torch.empty(lib.bdm_voxelize_workspace_bytes(B, n, r), dtype=torch.uint8, device=dev)
    p.occ_index = torch.empty(B, r3, dtype=torch.int32, device=dev)
    p.occ_list = torch.empty(B, p.n_max, dtype=torch.int32, device=dev)
    p.n_occ = torch.empty(B, dtype=torch.int32, device=dev)
    p.rowocc = torch.empty(B, r * r, dtype=torch.uint8, device=dev)
    L.check(lib.bdm_voxelize_plan_full(B, n, r, p.n_max, L.ptr(p.vox_coords), L.ptr(p.ind), L.ptr(p.cnt), L.ptr(p.ws),
                                       L.ptr(p.occ_index), L.ptr(p.occ_list), L.ptr(p.n_occ), L.ptr(p.rowocc), L.stream()),
            "voxelize_plan_full")
    p.dil_list = p.d2_list = None
    if dilate:
        plan_dilation(p)
        if dilate == 2:
            plan_dilation2(p)
    p.stream = torch.cuda.current_stream(dev) if coords.is_cuda else None
    _plan_cache[key] = p
    return p


def has_voxel_plan(coords, r):
    """Is the plan of (this very coordinate tensor, r) in the pass's cache?"""
    return (coords.data_ptr(), coords._version, tuple(coords.shape), int(r)) in _plan_cache


TILE_REC = 16         # ints per tile record of a dilated plan (include/bdm_hip.h)
# Tile form of the list convolutions (csrc/sparse_conv_os.hip): "0" (default) = FULL tiles, one eight-wave workgroup per CU (round 4's
# kernel); "64" / "128" / "256" = HALF tiles of that many entries, two four-wave workgroups per CU, weights straight from L2 (round 6).
# The half form was built to hide a tile's look-up chain / store burst behind a co-resident tile's matrix phase and
# MEASURED: two workgroups do co-reside, give the full form's bits, and a 256-entry tile then lives exactly as long as a 512-entry one
# (75 vs 78 us at 64 -> 64 channels, 32^3: the chunk loop is bound by the CU's matrix pipe, which the partners share, and with ~1 tile per
# slot both run the same phase at the same time) -- kernel 94 vs 84 us, replayed C2 step 5.16 vs 4.95 ms; at 8^3 it loses to GEMM + gather
# by 1.4 - 2.7x (profiles/r06_sparse_dil_half_tiles.txt).  Kept with its tests in the EXPERIMENTAL=1 build (as the other negative-result kernel
# families: csrc/experimental/); the default library refuses a non-zero tile.
DIL_TILE = os.environ.get("BDM_DIL_TILE", "0")


def dil_tile(batch, n_points, r, second):
    """Entries per tile of the (once- / twice-) dilated list of a level: 0 = the full-tile form (the default, see DIL_TILE).  One value
    per (batch, points, resolution) -- not per channel count -- because the PVConvs of a level share the plan."""
    if r not in (8, 16, 32) or DIL_TILE in ("0", "", "auto"):
        return 0
    if not L.has_experimental():
        raise L.BdmHipError(f"BDM_DIL_TILE={DIL_TILE}: the half-tile list convolution belongs to the experimental kernel families; "
                            "rebuild with `make -C bdm_amd/csrc EXPERIMENTAL=1`")
    return min(int(DIL_TILE), 128) if r == 8 else int(DIL_TILE)


DILATED_PLAN = True   # voxel plans carry the once-dilated voxel list + tile table of the compact first convolution (sparse_conv_os.hip)


def plan_dilation(p):
    """dil_list / dil_index / plane_start / tile_start of a plan (bdm_voxel_dilate): the output voxels of the first convolution that
    can differ from the bias, in voxel order, cut into tiles.  One launch on the plan's stream; depends on (coords, r) only."""
    if getattr(p, "dil_list", None) is not None:
        return p
    p.dil_list = p.dil_index = p.plane_start = p.tile_start = None
    p.n_dil_max = 0
    if not DILATED_PLAN or p.r not in (8, 16, 32):
        return p
    lib, B, r = L.lib(), p.cnt.shape[0], p.r
    dev = p.cnt.device
    p.n_dil_max = r ** 3
    p.tile = dil_tile(B, p.n, r, second=False)
    p.dil_list = torch.empty(B, p.n_dil_max, dtype=torch.int32, device=dev)
    p.dil_index = torch.empty(B, r ** 3, dtype=torch.int32, device=dev)
    p.plane_start = torch.empty(B, r + 2, dtype=torch.int32, device=dev)
    p.tile_start = torch.empty(B, lib.bdm_voxel_dilate_slices(r, p.tile), TILE_REC, dtype=torch.int32, device=dev)
    L.check(lib.bdm_voxel_dilate(B, r, p.n_dil_max, L.ptr(p.cnt), L.ptr(p.dil_list), L.ptr(p.dil_index), L.ptr(p.plane_start),
                                 L.ptr(p.tile_start), p.tile, L.stream()), "voxel_dilate")
    return p


def plan_dilation2(p):
    """The TWICE-dilated voxel list of a plan (bdm_voxel_dilate_again): where the second convolution of a PVConv can differ from its
    per-class constants; tiles whose input ranges are rows of the first list; voxels outside it counted per boundary class."""
    if getattr(p, "d2_list", None) is not None:
        return p
    plan_dilation(p)
    p.d2_list = p.d2_index = p.d2_tiles = p.d2_class_count = None
    if p.dil_list is None:
        return p
    lib, B, r = L.lib(), p.cnt.shape[0], p.r
    dev = p.cnt.device
    p.d2_tile = dil_tile(B, p.n, r, second=True)
    p.d2_list = torch.empty(B, p.n_dil_max, dtype=torch.int32, device=dev)
    p.d2_index = torch.empty(B, r ** 3, dtype=torch.int32, device=dev)
    ps = torch.empty(B, r + 2, dtype=torch.int32, device=dev)
    p.d2_tiles = torch.empty(B, lib.bdm_voxel_dilate_slices(r, p.d2_tile), TILE_REC, dtype=torch.int32, device=dev)
    p.d2_class_count = torch.empty(B, 27, dtype=torch.int32, device=dev)
    L.check(lib.bdm_voxel_dilate_again(B, r, p.n_dil_max, L.ptr(p.dil_index), L.ptr(p.d2_list), L.ptr(p.d2_index), L.ptr(ps),
                                       L.ptr(p.d2_tiles), L.ptr(p.d2_class_count), p.d2_tile, L.stream()), "voxel_dilate_again")
    return p


# ---- projection conditioning in factored form ---------------------------------------------------------------------------------
# x_in[i] = [xyz_i, F[pix_i]] (projection_model.py:179-231): F, the pixel-major conditioning image, is fixed for a trajectory; only
# xyz and the owning pixel change per step.  Every LINEAR first-layer map of x_in therefore commutes with the gather:
#     W . x_in[i] = Wx . xyz_i + (F . Wf^T)[pix_i]
# and F . Wf^T is computed once per (image batch, weight) -- hoisted out of the reverse loop like the image encoder itself.  Per step
# the consumers gather rows of 32 / 128 / 864 channels instead of convolving 390: SA0's point branch, the first sparse
# convolution and the last FP module's first MLP layer (pvcnn.py:90-127).  Same sums up to the reassociation of each dot product.
HOIST_CONDITIONING = os.environ.get("BDM_HOIST", "1") == "1"


def three_nn_search(points_coords, centers_coords):
    """(idx (B, 3, n) int32, w (B, 3, n)): the three nearest centres of every point and their inverse-distance weights
    (neighbor_interpolate.cu:19-92), searched ONCE for features and t_emb (the reference searches twice: same result)."""
    pc, cc = points_coords.contiguous(), centers_coords.contiguous()
    B, _, n = pc.shape
    m = cc.shape[2]
    idx = torch.empty(B, 3, n, dtype=torch.int32, device=pc.device)
    w = torch.empty(B, 3, n, dtype=torch.float32, device=pc.device)
    L.check(L.lib().bdm_three_nn_search(B, m, n, L.ptr(pc), L.ptr(cc), L.ptr(idx), L.ptr(w), L.stream()), "three_nn_search")
    return idx, w


def _fill_conditioning_map(feat, wf, out):
    xt = wf.t().contiguous()[None]                                     # (1, C, M): the "activation" operand of the GEMM below
    for b in range(feat.shape[0]):                                      # y (HW x M) = F[b] (HW x C) . Wf^T: pixels on the GEMM's row axis
        pointwise_conv(xt, feat[b], out=out[b:b + 1])


def refresh_conditioning_maps(feat, maps):
    """New image features in the same buffer (model.conditioning_image): every hoisted map is recomputed into ITS buffer, so the
    addresses a recorded step holds stay valid.  Maps of rewritten weights are dropped (rebuilt on first use)."""
    for key, ent in list(maps.items()):
        if len(ent) != 4:
            continue                                                    # weight-only entries (conv1_wx): image-independent
        if ent[0] != ent[2]._version:
            del maps[key]
        else:
            _fill_conditioning_map(feat, ent[3]().contiguous(), ent[1])


class Conditioning:
    """Handle that travels with the denoiser input of ONE reverse step (attribute `_bdm_cond` of the x_in tensor)."""

    def __init__(self, feat, hw, pix, x_t, x_cf, maps):
        # feat (B, HW, C) pixel-major image; pix (B, N) int32 owning pixel or -1; x_t (B, N, 3) point-major and x_cf (B, 3 + C, N)
        # channel-first (rows 0..2 = xyz) forms of the step's cloud; maps: the per-image cache of hoisted maps
        self.feat, self.hw, self.pix, self.x_t, self.x_cf, self.maps = feat, hw, pix, x_t, x_cf, maps
        self.C = feat.shape[2]
        self.early = None   # (centres of the first set-abstraction level, sampled from x_t while the conditioning ran; pvcnn.early_first_sampler)
        # LAZY form (model.get_input_with_conditioning(lazy=True), the reverse loops): only the coordinate rows of x_cf are written; the
        # denoiser completes the tensor (ensure_features) unless every reader of rows 3.. takes its share from a hoisted map
        self.features_ready = True

    def ensure_features(self):
        """Write the feature rows of x_cf if the lazy form left them out (one launch of the full conditioning gather)."""
        if not self.features_ready:
            B, _, N = self.x_cf.shape
            L.check(L.lib().bdm_condition_gather_cf(B, N, self.C, self.feat.shape[1], L.ptr(self.x_t), L.ptr(self.feat), L.ptr(self.pix),
                                                    L.ptr(self.x_cf), L.stream()), "condition_gather_cf")
            self.features_ready = True

    def map(self, kind, weight, build):
        """(B, HW, M) = F . Wf^T for the (M, C) matrix `build()` returns; cached per (kind, weight tensor, version) for the image batch."""
        key = (kind, weight.data_ptr())
        hit = self.maps.get(key)
        if hit is None or hit[0] != weight._version or hit[2] is not weight:
            wf = build().contiguous()                                  # (M, C)
            B, HW, _ = self.feat.shape
            out = hit[1] if hit is not None and tuple(hit[1].shape) == (B, HW, wf.shape[0]) else \
                torch.empty(B, HW, wf.shape[0], dtype=torch.float32, device=self.feat.device)
            _fill_conditioning_map(self.feat, wf, out)
            hit = (weight._version, out, weight, build)
            self.maps[key] = hit
        return hit[1]

    def gather(self, fmap):
        """(B, 3 + M, N) = [xyz, fmap[pix]] channel-first (zeros where a point owns no pixel)."""
        B, HW, M = fmap.shape
        N = self.pix.shape[1]
        out = torch.empty(B, 3 + M, N, dtype=torch.float32, device=fmap.device)
        L.check(L.lib().bdm_condition_gather_cf(B, N, M, HW, L.ptr(self.x_t), L.ptr(fmap), L.ptr(self.pix), L.ptr(out), L.stream()),
                "condition_gather_cf")
        return out


def sparse_first_conv_from_map(cond, plan, conv, cout, gn_groups=None):
    """Conv3d(k3, p1)(avg_voxelize(x_in)) for x_in = [xyz, F[pix]] via the hoisted map (bdm_sparse_conv_rows_from_map) + the usual
    gather: (B, cout, r^3) fp32 (+ GroupNorm statistics as sparse_first_conv_planned)."""
    w = conv.weight
    n27 = 27 * cout
    C = cond.C
    hmap = cond.map("conv1", w, lambda w=w, C=C, cout=cout, n27=n27: w.detach()[:, 3:3 + C].reshape(cout, C, 27).permute(2, 0, 1).reshape(n27, C))  # row tap*cout+co
    key = ("conv1_wx", w.data_ptr())
    hit = cond.maps.get(key)
    if hit is None or hit[0] != w._version:
        hit = (w._version, w.detach()[:, :3].reshape(cout, 3, 27).permute(2, 0, 1).reshape(n27, 3).contiguous())
        cond.maps[key] = hit
    wx = hit[1]
    B, _, n = cond.x_cf.shape
    dev, lib, r = cond.x_cf.device, L.lib(), plan.r
    xyz = cond.x_cf[:, :3]
    if not xyz.is_contiguous():
        # the plan was built from a contiguous copy of exactly these rows (pvcnn.encode): no second copy inside the step
        pc = getattr(plan, "coords", None)
        xyz = pc if (pc is not None and pc.is_contiguous() and tuple(pc.shape) == tuple(xyz.shape)) else xyz_rows(cond.x_cf)
    y = torch.empty(B, plan.n_max, n27, dtype=torch.float32, device=dev)
    L.check(lib.bdm_sparse_conv_rows_from_map(B, n, r, plan.n_max, n27, hmap.shape[1], L.ptr(hmap), L.ptr(cond.pix), L.ptr(xyz), L.ptr(wx),
                                              L.ptr(plan.cnt), L.ptr(plan.ws), L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(y), L.stream()),
            "sparse_conv_rows_from_map")
    out = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=dev)
    gn, stats = None, None
    if gn_groups and gather_gn_ok(cout, gn_groups):
        partial = torch.empty(B, gn_groups, r * r, 2, dtype=torch.float64, device=dev)
        gn, stats = (partial, int(gn_groups)), (partial, r * r, int(gn_groups))
    _gather(lib, B, cout, r, plan, 0, y, conv.bias, out, gn)
    return (out, stats) if gn_groups else out


def gather_gn_ok(cout, groups):
    cg = cout // groups if groups and cout % groups == 0 else 0
    return cg >= 4 and cg % 4 == 0 and cout % 4 == 0 and 256 % (cout // 4) == 0


def _gather(lib, nb, cout, r, plan, b0, y, bias, out, gn):
    """sparse gather of one shape group; gn = (partial (B, G, r*r, 2) fp64, groups) also leaves the output's GroupNorm partials"""
    if gn is None:
        L.check(lib.bdm_sparse_conv_gather(nb, cout, r, plan.n_max, L.ptr(y), L.ptr(plan.occ_index[b0:]), L.ptr(plan.rowocc[b0:]),
                                           L.ptr(bias), L.ptr(out[b0:]), L.stream()), "sparse_conv_gather")
    else:
        partial, groups = gn
        L.check(lib.bdm_sparse_conv_gather_gn(nb, cout, r, plan.n_max, L.ptr(y), L.ptr(plan.occ_index[b0:]), L.ptr(plan.rowocc[b0:]),
                                              L.ptr(bias), L.ptr(out[b0:]), groups, L.ptr(partial[b0:]), L.stream()),
                "sparse_conv_gather_gn")


def sparse_first_conv_planned(features, plan, wt, bias, cout, gn_groups=None, rows=None, h2_out=None, col_bias=None):
    """Conv3d(k3, p1)(avg_voxelize(features)) on the occupied voxels of `plan`: (B, cout, r^3) fp32.
    gn_groups: also return the GroupNorm(gn_groups) statistics of the output as (partial, slices = r*r, groups) -> (out, stats).
    rows (fp16x3 GEMM only): a VoxelRows the previous PVConv's tail left for this very plan -- the feature pass and the split are skipped.
    h2_out = (gn, act_scale, saturated) (fp16x3 GEMM on a small grid, small_grid_gather_ok): GroupNorm + Swish + the second convolution's
    operand split in the gather's epilogue -> ((B, cout/8, 2, r^3, 8) fp16, 1 / act_scale); the dense fp32 grid is not written.
    col_bias (B, 27 * cout) (fp16x3 / bf16x6 GEMM): per-shape addend of every occupied row's columns -- the share of input channels that are
    constant over a shape (the time embedding; `features` and `wt` then hold the other channels only: bdm_sparse_conv_gemm_s3_cb)."""
    f, B, C, n, bs_f, ld_f = _bcl(features)
    dev, lib, r = f.device, L.lib(), plan.r
    if col_bias is not None:
        assert tuple(col_bias.shape) == (B, 27 * cout) and col_bias.stride(1) == 1 and h2_out is None

    def gemm_h2(nb, b0, xh, amax, packed, inv_scale, y):
        if col_bias is None:
            L.check(lib.bdm_sparse_conv_gemm_h2(nb, plan.n_max, C, cout, L.ptr(xh[b0:]), L.ptr(amax[b0:]), L.ptr(packed), L.ptr(inv_scale),
                                                L.ptr(plan.n_occ[b0:]), L.ptr(y), L.stream()), "sparse_conv_gemm_h2")
        else:
            L.check(lib.bdm_sparse_conv_gemm_h2_cb(nb, plan.n_max, C, cout, L.ptr(xh[b0:]), L.ptr(amax[b0:]), L.ptr(packed), L.ptr(inv_scale),
                                                   L.ptr(plan.n_occ[b0:]), L.ptr(col_bias[b0:]), L.c_ll(col_bias.stride(0)), L.ptr(y), L.stream()), "sparse_conv_gemm_h2_cb")
    if isinstance(wt, tuple) and wt[0] == "h2":  # fp16x3 GEMM (sparse_conv_pack_h2) + gather: the default
        _, packed, inv_scale = wt
        if rows is not None:
            assert rows.plan is plan and rows.channels == C
            xh, amax = rows.xh, rows.amax
        else:
            xr = torch.empty(B, (C + 7) // 8, plan.n_max, 8, dtype=torch.float32, device=dev)
            amax = amax_slots(dev, B)  # one activation scale per shape: a shape's result does not depend on its batch-mates
            L.check(lib.bdm_sparse_voxel_features_f32(B, C, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt), L.ptr(plan.ws),
                                                      L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()),
                    "sparse_voxel_features_f32")
            xh = torch.empty(B, (C + 7) // 8, 2, plan.n_max, 8, dtype=torch.float16, device=dev)
            L.check(lib.bdm_sparse_split_h2(B, C, plan.n_max, L.ptr(xr), L.ptr(amax), L.ptr(xh), L.stream()), "sparse_split_h2")
        per_shape = plan.n_max * 27 * cout * 4
        gb = max(1, min(B, SPARSE_Y_BYTES // max(per_shape, 1)))
        y = torch.empty(gb, plan.n_max, 27 * cout, dtype=torch.float32, device=dev)
        if h2_out is not None:
            gn1, act_scale, saturated = h2_out
            x2 = torch.empty(B, cout // 8, 2, r ** 3, 8, dtype=torch.float16, device=dev)
            for b0 in range(0, B, gb):
                nb = min(gb, B - b0)
                L.check(lib.bdm_sparse_conv_gemm_h2(nb, plan.n_max, C, cout, L.ptr(xh[b0:]), L.ptr(amax[b0:]), L.ptr(packed), L.ptr(inv_scale),
                                                    L.ptr(plan.n_occ[b0:]), L.ptr(y), L.stream()), "sparse_conv_gemm_h2")
                L.check(L.experimental("bdm_sparse_conv_gather_h2_small")(nb, cout, r, plan.n_max, L.ptr(y), L.ptr(plan.occ_index[b0:]), L.ptr(bias),
                                                            gn1.num_groups, L.ptr(gn1.weight), L.ptr(gn1.bias), L.c_float(gn1.eps),
                                                            L.c_float(act_scale), L.ptr(x2[b0:]), L.ptr(saturated), L.stream()),
                        "sparse_conv_gather_h2_small")
            return x2, 1.0 / act_scale
        out = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=dev)
        gn, stats = None, None
        if gn_groups and gather_gn_ok(cout, gn_groups):
            partial = torch.empty(B, gn_groups, r * r, 2, dtype=torch.float64, device=dev)
            gn, stats = (partial, int(gn_groups)), (partial, r * r, int(gn_groups))
        for b0 in range(0, B, gb):
            nb = min(gb, B - b0)
            gemm_h2(nb, b0, xh, amax, packed, inv_scale, y)
            _gather(lib, nb, cout, r, plan, b0, y, bias, out, gn)
        return (out, stats) if gn_groups else out
    if isinstance(wt, tuple):  # fused kernel (sparse_conv_pack_fused): GEMM + scatter in one launch, no intermediate
        assert col_bias is None
        _, packed, inv_scale = wt
        xr = torch.empty(B, (C + 7) // 8, plan.n_max, 8, dtype=torch.float32, device=dev)
        amax = amax_slots(dev, B)  # one activation scale per shape: a shape's result does not depend on its batch-mates
        L.check(lib.bdm_sparse_voxel_features_f32(B, C, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt), L.ptr(plan.ws),
                                                  L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()),
                "sparse_voxel_features_f32")
        out = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=dev)
        L.check(L.experimental("bdm_sparse_conv_fused")(B, C, cout, r, plan.n_max, L.ptr(xr), L.ptr(amax), L.ptr(packed), L.ptr(inv_scale),
                                          L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(bias), L.ptr(out), L.stream()),
                "sparse_conv_fused")
        return (out, None) if gn_groups else out
    out = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=dev)
    # The 27x-expanded intermediate Y (n_occ x 27*cout fp32 per shape) is written by the GEMM and read once by the gather.
    # Shapes are processed in groups whose Y stays within SPARSE_Y_BYTES (256 MiB, about the memory-side cache),
    # reusing ONE Y buffer.  Measured (tools/sparse_bench.py): two groups of 8 at the 64 -> 64 / 32^3 layer: 287 -> 261 us;
    # smaller groups lose more to the extra launches than they gain.
    per_shape = plan.n_max * 27 * cout * 4
    gb = max(1, min(B, SPARSE_Y_BYTES // max(per_shape, 1)))
    y = torch.empty(gb, plan.n_max, 27 * cout, dtype=torch.float32, device=dev)
    s3 = wt.dtype == torch.bfloat16  # bf16x6: pre-split operands (sparse_conv_pack_s3)
    G8 = (C + 7) // 8
    if s3:
        xs = torch.empty(B, G8, 3, plan.n_max, 8, dtype=torch.bfloat16, device=dev)
        L.check(lib.bdm_sparse_voxel_features_s3(B, C, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt),
                                                 L.ptr(plan.ws), L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xs), L.stream()),
                "sparse_voxel_features_s3")
    else:
        xs = torch.empty(B, C, plan.n_max, dtype=torch.float32, device=dev)
        L.check(lib.bdm_sparse_voxel_features(B, C, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt),
                                              L.ptr(plan.ws), L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xs), L.stream()),
                "sparse_voxel_features")
    gn, stats = None, None
    if gn_groups and gather_gn_ok(cout, gn_groups):
        partial = torch.empty(B, gn_groups, r * r, 2, dtype=torch.float64, device=dev)
        gn, stats = (partial, int(gn_groups)), (partial, r * r, int(gn_groups))
    for b0 in range(0, B, gb):
        nb = min(gb, B - b0)
        if s3 and col_bias is not None:
            L.check(lib.bdm_sparse_conv_gemm_s3_cb(nb, plan.n_max, C, 27 * cout, L.ptr(xs[b0:]), L.ptr(wt), L.ptr(plan.n_occ[b0:]),
                                                   L.ptr(col_bias[b0:]), L.c_ll(col_bias.stride(0)), L.ptr(y), L.stream()), "sparse_conv_gemm_s3_cb")
        elif s3:
            L.check(lib.bdm_sparse_conv_gemm_s3(nb, plan.n_max, C, 27 * cout, L.ptr(xs[b0:]), L.ptr(wt), L.ptr(plan.n_occ[b0:]),
                                                L.ptr(y), L.stream()), "sparse_conv_gemm_s3")
        else:
            assert col_bias is None
            L.check(lib.bdm_sparse_conv_gemm(nb, plan.n_max, C, 27 * cout, L.ptr(xs[b0:]), L.ptr(wt), L.ptr(plan.n_occ[b0:]),
                                             L.ptr(y), L.stream()), "sparse_conv_gemm")
        _gather(lib, nb, cout, r, plan, b0, y, bias, out, gn)
    return (out, stats) if gn_groups else out


def sparse_conv_pack_os(weight, form="dil"):
    """(Cout, Cin, 3,3,3) fp32 -> (form, fp16 weight image of the dense fp16x3 convolution, inv_scale) for sparse_first_conv_os."""
    return (form,) + tuple(conv3d_h2_pack(weight))


class CompactGrid:
    """Output of the first convolution in compact form: rows (B, n_dil_max, C), one per entry of the plan's dilated voxel list; every
    other voxel of the (B, C, r^3) grid equals bias.  to_h2 consumes it directly; dense() materialises the grid (tests)."""

    def __init__(self, rows, plan, bias, channels):
        self.rows, self.plan, self.bias, self.channels = rows, plan, bias, channels

    is_cuda = property(lambda self: self.rows.is_cuda)
    device = property(lambda self: self.rows.device)

    def dense(self):
        B, r3, C = self.rows.shape[0], self.plan.r ** 3, self.channels
        out = self.bias.detach().view(1, C, 1).expand(B, C, r3).contiguous()
        idx = self.plan.dil_index.long()
        for b in range(B):
            v = torch.nonzero(idx[b] >= 0).squeeze(1)
            out[b, :, v] = self.rows[b, idx[b, v]].t()
        return out


def sparse_first_conv_os(features, plan, packed, bias, cout, gn_groups=None, compact=False):
    """Conv3d(k3, p1)(avg_voxelize(features)) as ONE output-stationary implicit GEMM with tap skipping over the once-dilated voxel
    list of the plan (csrc/sparse_conv_os.hip): occupied cells' fp32 feature records + per-shape maximum (one launch), then the
    convolution (one launch) -- no 27x intermediate.  packed = conv3d_h2_pack(weight).
    compact: -> CompactGrid (rows per list entry; the dense grid is never written) instead of the (B, cout, r^3) tensor.
    gn_groups: -> (out, (partials, slices, groups)) with the GroupNorm statistics of the (dense) output."""
    plan_dilation(plan)
    if plan.dil_list is None:
        raise L.BdmHipError(f"sparse_first_conv_os: resolution {plan.r} has no dilated plan (8, 16, 32)")
    f, B, C, n, bs_f, ld_f = _bcl(features)
    dev, lib, r = f.device, L.lib(), plan.r
    packed_w, inv_scale = packed
    xr = torch.empty(B, (C + 7) // 8, plan.n_max, 8, dtype=torch.float32, device=dev)
    slots = amax_slots(dev, B + 1)            # B activation-scale maxima + the convolution's work counter (zero bits = int 0)
    amax, counter = slots[:B], slots[B:]
    L.check(lib.bdm_sparse_voxel_features_f32(B, C, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt), L.ptr(plan.ws),
                                              L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()),
            "sparse_voxel_features_f32")
    if compact:
        y = torch.empty(B, plan.n_dil_max, cout, dtype=torch.float32, device=dev)
        out = CompactGrid(y, plan, bias, cout)
    else:
        y = out = torch.empty(B, cout, r ** 3, dtype=torch.float32, device=dev)
    args = (B, C, cout, r, plan.n_max, plan.n_dil_max, L.ptr(xr), L.ptr(amax), L.ptr(plan.occ_index), L.ptr(plan.dil_list),
            L.ptr(plan.dil_index), L.ptr(plan.tile_start), L.ptr(packed_w), L.ptr(inv_scale), L.ptr(bias), L.ptr(y), 1 if compact else 0)
    if gn_groups:
        tiles = plan.tile_start.shape[1]
        partial = torch.empty(B, gn_groups, tiles, 2, dtype=torch.float64, device=dev)
        slices = ctypes.c_int(0)
        L.check(lib.bdm_sparse_conv_dil_gn(*args, int(gn_groups), L.ptr(partial), ctypes.byref(slices), plan.tile, L.ptr(counter), L.stream()),
                "sparse_conv_dil_gn")
        return out, (partial, tiles, int(gn_groups))
    L.check(lib.bdm_sparse_conv_dil(*args, plan.tile, L.ptr(counter), L.stream()), "sparse_conv_dil")
    return out


# ---- the PVConv voxel branch on voxel lists only (csrc/pvconv_compact.hip) ---------------------------------------------------------
def to_h2_rows(x, plan, gn, stats, swish=True, saturated=None, bias=None):
    """GroupNorm + Swish + fp16 split of the first convolution's output on the rows of the plan's dilated list.  x: CompactGrid or the
    dense (B, C, r^3) grid (then `bias` = the convolution's bias).  -> (rows_h2, const_h2, const_f32, 1 / scale)."""
    plan_dilation(plan)
    lib = L.lib()
    compact = isinstance(x, CompactGrid)
    src = x.rows if compact else x.contiguous()
    bias = x.bias if compact else bias
    B, C, V = src.shape[0], (x.channels if compact else src.shape[1]), plan.r ** 3
    dev = src.device
    s = float(h2_activation_scale(gn))
    C8 = (C + 7) // 8
    rows = torch.empty(B, C8, 2, plan.n_dil_max, 8, dtype=torch.float16, device=dev)
    const_h2 = torch.empty(B, C8, 2, 8, dtype=torch.float16, device=dev)
    const_f32 = torch.empty(B, C, dtype=torch.float32, device=dev)
    partial, slices, groups = stats
    L.check(lib.bdm_group_norm_to_h2_rows(B, C, V, groups, L.ptr(src), 0 if compact else 1, plan.n_dil_max, L.ptr(plan.dil_list),
                                          L.ptr(plan.tile_start), plan.tile_start.shape[1], L.ptr(bias), L.ptr(gn.weight), L.ptr(gn.bias),
                                          L.c_float(gn.eps), 1 if swish else 0, L.c_float(s), L.ptr(rows), L.ptr(const_h2), L.ptr(const_f32),
                                          L.ptr(partial), slices, L.ptr(saturated), L.stream()), "group_norm_to_h2_rows")
    return rows, const_h2, const_f32, 1.0 / s


def conv_class_pack(weight):
    """(Cout, Cin, 3,3,3) fp32 -> (27, Cin, Cout) fp64: per boundary class the sum of the taps that stay inside the grid."""
    cout, cin = weight.shape[:2]
    lib = L.lib()
    wsum = torch.empty(lib.bdm_conv3d_class_weight_elems(cout, cin), dtype=torch.float64, device=weight.device)
    L.check(lib.bdm_conv3d_class_weight_sums(cout, cin, L.ptr(weight.contiguous()), L.ptr(wsum), L.stream()), "conv3d_class_weight_sums")
    return wsum


def second_conv_rows(rows_h2, const_h2, const_f32, x_inv_scale, plan, packed, wsum, bias, cin, cout, groups=8):
    """The second convolution of a PVConv on the plan's twice-dilated list: -> (rows (B, n_dil_max, cout), class_vals (B, 27, cout),
    (partials, slices)) -- every voxel outside the list has the value of its boundary class."""
    plan_dilation2(plan)
    lib, B, r = L.lib(), rows_h2.shape[0], plan.r
    dev = rows_h2.device
    packed_w, inv_scale = packed
    y = torch.empty(B, plan.n_dil_max, cout, dtype=torch.float32, device=dev)
    tiles = plan.d2_tiles.shape[1]
    partial = torch.empty(B, groups, tiles + 27, 2, dtype=torch.float64, device=dev)
    counter = amax_slots(dev, 1)
    slices = ctypes.c_int(0)
    L.check(lib.bdm_sparse_conv_dil_h2_gn(B, cin, cout, r, plan.n_dil_max, plan.n_dil_max, L.ptr(rows_h2), L.ptr(const_h2), L.c_float(x_inv_scale),
                                          L.ptr(plan.dil_index), L.ptr(plan.d2_list), L.ptr(plan.d2_index), L.ptr(plan.d2_tiles), L.ptr(packed_w),
                                          L.ptr(inv_scale), L.ptr(bias), L.ptr(y), int(groups), L.ptr(partial), ctypes.byref(slices), plan.d2_tile,
                                          L.ptr(counter), L.stream()), "sparse_conv_dil_h2_gn")
    assert slices.value == tiles + 27
    class_vals = torch.empty(B, 27, cout, dtype=torch.float32, device=dev)
    L.check(lib.bdm_conv3d_class_constants(B, cin, cout, L.ptr(wsum), L.ptr(bias), L.ptr(const_f32), L.ptr(plan.d2_class_count), L.ptr(class_vals),
                                           int(groups), L.ptr(partial), tiles + 27, tiles, L.stream()), "conv3d_class_constants")
    return y, class_vals, (partial, tiles + 27)


def se_gate_gn_rows(rows, class_vals, plan, stats, gn, w1, w2, pf=None, n_points=0):
    """se_gate_gn from the rows of the twice-dilated list + counts x class constants: (gate, coef[, pf_coef])."""
    partial, slices = stats
    B, C = rows.shape[0], rows.shape[2]
    dev, lib = rows.device, L.lib()
    V = plan.r ** 3
    mean = torch.empty(B, C, dtype=torch.float32, device=dev)
    coef = torch.empty(B, C, 2, dtype=torch.float32, device=dev)
    gate = torch.empty(B, C, dtype=torch.float32, device=dev)
    part = torch.empty(lib.bdm_se_gate_gn_rows_workspace_elems(B, C), dtype=torch.float32, device=dev)
    args = (B, C, w1.shape[0], V, gn.num_groups, L.ptr(rows), plan.n_dil_max, L.ptr(plan.d2_tiles), plan.d2_tiles.shape[1], L.ptr(class_vals),
            L.ptr(plan.d2_class_count), L.ptr(partial), slices, L.ptr(gn.weight), L.ptr(gn.bias), L.c_float(gn.eps), L.ptr(w1), L.ptr(w2), L.ptr(part),
            L.ptr(mean), L.ptr(coef), L.ptr(gate))
    if pf is not None:
        (pp, ps, pg), pgn = pf
        pf_coef = torch.empty(B, C, 2, dtype=torch.float32, device=dev)
        L.check(lib.bdm_se_gate_gn_rows_pf(*args, L.ptr(pp), ps, pg, int(n_points), L.ptr(pgn.weight), L.ptr(pgn.bias), L.c_float(pgn.eps),
                                           L.ptr(pf_coef), L.stream()), "se_gate_gn_rows_pf")
        return gate, coef, pf_coef
    L.check(lib.bdm_se_gate_gn_rows(*args, L.stream()), "se_gate_gn_rows")
    return gate, coef


def devoxelize_gn_gate_add_rows(norm_coords, rows, class_vals, plan, coef, gate=None, add=None, add_coef=None):
    B, C = rows.shape[0], rows.shape[2]
    n = norm_coords.shape[2]
    out = torch.empty(B, C, n, dtype=torch.float32, device=rows.device)
    _, _, _, _, bs_o, ld_o = _bcl(out)
    if add is not None:
        aa, _, _, _, bs_a, ld_a = _bcl(add)
        assert aa.data_ptr() == add.data_ptr()
    else:
        bs_a, ld_a = 0, 0
    lib = L.lib()
    args = (B, C, n, plan.r, L.ptr(norm_coords), L.ptr(rows), plan.n_dil_max, L.ptr(plan.d2_index), L.ptr(class_vals), L.ptr(coef), L.ptr(gate),
            L.ptr(add), bs_a, ld_a)
    if add_coef is not None:
        L.check(lib.bdm_devoxelize_gn_gate_add_rows_pf(*args, L.ptr(add_coef), L.ptr(out), bs_o, ld_o, L.stream()), "devoxelize_gn_gate_add_rows_pf")
    else:
        L.check(lib.bdm_devoxelize_gn_gate_add_rows(*args, L.ptr(out), bs_o, ld_o, L.stream()), "devoxelize_gn_gate_add_rows")
    return out


def densify_rows(rows, class_vals, plan):
    """(B, C, r^3) grid of a second convolution's compact result (tests): rows at the list's voxels, the class constant elsewhere."""
    B, _, C = rows.shape
    r = plan.r
    ax = torch.arange(r, device=rows.device)
    cls1 = torch.where(ax == 0, 0, torch.where(ax == r - 1, 2, 1))
    cls = (cls1[:, None, None] * 9 + cls1[None, :, None] * 3 + cls1[None, None, :]).reshape(-1)
    out = class_vals[:, cls].permute(0, 2, 1).contiguous()
    idx = plan.d2_index.long()
    for b in range(B):
        v = torch.nonzero(idx[b] >= 0).squeeze(1)
        out[b, :, v] = rows[b, idx[b, v]].t()
    return out


COMPACT_16_MIN_ITEMS = 336    # the same for the whole voxel branch on lists at 16^3 (12 tiles per 1024-point shape: 28 shapes)
SPARSE_DIL_MIN_ITEMS = 160   # tiles x channel blocks x shapes below which the compact convolution cannot fill the chip


def sparse_dil_pays(batch, n_points, r, cout):
    """Does the compact output-stationary convolution beat GEMM + gather for this layer?  Its persistent workgroups take one tile
    (<= 512 / 256 / 128 dilated voxels x 64 or 32 channels) through ALL channel chunks: ~85 us per tile at 64 -> 64 channels whatever
    the batch, so it needs about a tile per CU to pay (measured on one box, B = 16, N = 4096: step 6.39 -> 6.13 ms; B = 4: 3.84 ->
    4.23 ms, C1 0.279 -> 0.306 s; profiles/r04_sparse_dil_small_batch.txt).  Tiles per shape are estimated from the sizes alone
    (a Gaussian-like cloud of n points dilates to ~2 n voxels at 32^3, ~1.8 n at 16^3), so the choice depends on the configuration,
    never on the data."""
    if r == 32:
        tiles = min(64, max(1, (2 * n_points) // 512))
    elif r == 16:
        tiles = min(16, max(1, (2 * n_points) // 256))
    else:
        return False          # 8^3: the GEMM over <= 256 occupied rows per shape wins at every batch measured
    return batch * tiles * (2 if cout > 64 else 1) >= SPARSE_DIL_MIN_ITEMS


def compact_tail_pays(batch, n_points, r, cout):
    """Does the voxel branch on voxel lists (second convolution on the twice-dilated list, no dense grids: pvconv_compact.hip) beat the
    dense-grid path?  Measured at B = 16 (round 4's pricing tool, since replaced by tools/sparse_os_probe.py): 64 channels at 32^3 187 vs 358 us, 32 channels 84 vs 117,
    64 channels at 16^3 56 vs 75, 128 channels at 16^3 176 vs 157 (the list covers 72 % of that grid: no).  As sparse_dil_pays: decided
    from the sizes, never from the data."""
    if r == 32:
        tiles = min(64, max(1, (3 * n_points) // 512))
    elif r == 16 and cout <= 64:
        # (round 5: the dense 16^3 convolution in its small tiling runs 50 us at B = 16, not 72: the lists win from ~28 shapes on --
        # replayed step with / without them at 16^3: B = 16 4.925 / 4.885 ms, B = 24 7.336 / 7.301, B = 32 (N = 16384) 11.90 / 11.98)
        tiles = min(16, max(1, (3 * n_points) // 256))
        return batch * tiles >= COMPACT_16_MIN_ITEMS
    else:
        return False
    return batch * tiles * (2 if cout > 64 else 1) >= SPARSE_DIL_MIN_ITEMS


def sparse_os_gn_ok(cout, groups, r):
    cg = cout // groups if groups and cout % groups == 0 else 0
    tile = 64 if cout > 32 else 32
    return cg >= 4 and (cg & (cg - 1)) == 0 and tile % cg == 0


def sparse_first_conv(features, vox_coords, r, wt, bias, cout):
    """Conv3d(k3, p1)(avg_voxelize(features, vox_coords, r)) evaluated on the occupied voxels: (B, cout, r^3) fp32.
    Builds a one-off plan from integer voxel coordinates (the modules share cached plans: voxel_plan)."""
    B, _, n = vox_coords.shape
    dev, lib, r3 = vox_coords.device, L.lib(), r ** 3
    p = VoxelPlan()
    p.ready = None
    p.r, p.n, p.n_max = int(r), n, min(n, r3)
    p.norm_coords, p.vox_coords = None, vox_coords
    p.ind = torch.empty(B, n, dtype=torch.int32, device=dev)
    p.cnt = torch.empty(B, r3, dtype=torch.int32, device=dev)
    p.ws = torch.empty(lib.bdm_voxelize_workspace_bytes(B, n, r), dtype=torch.uint8, device=dev)
    L.check(lib.bdm_voxelize_plan(B, n, r, L.ptr(vox_coords), L.ptr(p.ind), L.ptr(p.cnt), L.ptr(p.ws), L.stream()), "voxelize_plan")
    p.occ_index = torch.empty(B, r3, dtype=torch.int32, device=dev)
    p.occ_list = torch.empty(B, p.n_max, dtype=torch.int32, device=dev)
    p.n_occ = torch.empty(B, dtype=torch.int32, device=dev)
    p.rowocc = torch.empty(B, r * r, dtype=torch.uint8, device=dev)
    L.check(lib.bdm_voxel_compact(B, r, p.n_max, L.ptr(p.cnt), L.ptr(p.occ_index), L.ptr(p.occ_list), L.ptr(p.n_occ), L.stream()),
            "voxel_compact")
    L.check(lib.bdm_voxel_row_occupancy(B, r, L.ptr(p.cnt), L.ptr(p.rowocc), L.stream()), "voxel_row_occupancy")
    p.dil_list = None
    if isinstance(wt, tuple) and wt[0] in ("dil", "dil_compact"):
        out = sparse_first_conv_os(features, plan_dilation(p), wt[1:], bias, cout, compact=wt[0] == "dil_compact")
        return out.dense() if wt[0] == "dil_compact" else out
    return sparse_first_conv_planned(features, p, wt, bias, cout)
