"""The training half of the plugin (gradient operators + training-mode devoxelisation; reference: grouping.cu:58-77,
neighbor_interpolate.cu:145-170, trilinear_devox.cu:21-162, sampling.cu:52-66, vox.cu:86-110).
  CPU: the oracle's C restatement equals torch autograd through a pure-torch statement of each forward;
  GPU: the HIP kernels equal the oracle, and gradients flow through bdm_amd.functional's autograd Functions."""
import pytest
import torch


def _inputs():
    g = torch.Generator().manual_seed(7)
    B, C, N, M, U, R = 2, 5, 200, 40, 8, 4
    x = {"B": B, "C": C, "N": N, "M": M, "U": U, "R": R}
    x["feat"] = torch.randn(B, C, N, generator=g)
    x["idx"] = torch.randint(0, N, (B, M), generator=g, dtype=torch.int32)
    x["nbr"] = torch.randint(0, N, (B, M, U), generator=g, dtype=torch.int32)
    x["pts"] = torch.randn(B, 3, N, generator=g) * 0.3
    x["ctr"] = x["pts"][:, :, :M].contiguous() + 0.01
    x["cfeat"] = torch.randn(B, C, M, generator=g)
    x["vc"] = torch.randint(0, R, (B, 3, N), generator=g, dtype=torch.int32)
    x["grid"] = torch.randn(B, C, R ** 3, generator=g)
    x["nc"] = torch.rand(B, 3, N, generator=g) * (R - 1)
    x["nc"][0, :, 0] = torch.tensor([1.0, 2.0, float(R - 1)])     # integer coordinates incl. the upper boundary
    return x


def test_oracle_backward_equals_autograd(oracle_ops):
    O, x = oracle_ops, _inputs()
    B, C, N, M, U, R = (x[k] for k in "BCNMUR")
    g = torch.Generator().manual_seed(1)
    # gather / grouping: forward = index_select
    f = x["feat"].clone().requires_grad_()
    y = torch.stack([f[b][:, x["idx"][b].long()] for b in range(B)])
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    assert torch.allclose(O.gather_features_backward(gy, x["idx"], N), f.grad, atol=1e-5)
    f = x["feat"].clone().requires_grad_()
    y = torch.stack([f[b][:, x["nbr"][b].long().reshape(-1)].reshape(C, M, U) for b in range(B)])
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    assert torch.allclose(O.grouping_backward(gy, x["nbr"], N), f.grad, atol=1e-5)
    # three-NN interpolation: out = sum_k w_k f[idx_k]
    out, idx, w = O.three_nearest_neighbors_interpolate_forward(x["pts"], x["ctr"], x["cfeat"])
    cf = x["cfeat"].clone().requires_grad_()
    y = sum(torch.stack([cf[b][:, idx[b, k].long()] for b in range(B)]) * w[:, k:k + 1] for k in range(3))
    assert torch.allclose(y, out, atol=1e-6)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    assert torch.allclose(O.three_nearest_neighbors_interpolate_backward(gy, idx, w, M), cf.grad, atol=1e-5)
    # average voxelisation
    vox, ind, cnt = O.avg_voxelize_forward(x["feat"], x["vc"], R)
    f = x["feat"].clone().requires_grad_()
    y = torch.zeros(B, C, R ** 3)
    y = y.index_put((torch.arange(B)[:, None, None], torch.arange(C)[None, :, None], ind.long()[:, None, :]),
                    f / cnt.gather(1, ind.long()).float()[:, None, :], accumulate=True)
    assert torch.allclose(y, vox, atol=1e-5)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    assert torch.allclose(O.avg_voxelize_backward(gy, ind, cnt), f.grad, atol=1e-5)
    # trilinear devoxelisation: training mode saves (inds, wgts); eval and training forwards agree
    out_t, inds, wgts = O.trilinear_devoxelize_forward(R, True, x["nc"], x["grid"])
    assert torch.equal(out_t, O.trilinear_devoxelize_forward(R, False, x["nc"], x["grid"])[0])
    assert inds.shape == (B, 8, N) and int(inds.max()) < R ** 3 and torch.allclose(wgts.sum(1), torch.ones(B, N), atol=1e-6)
    gr = x["grid"].clone().requires_grad_()
    y = sum(torch.stack([gr[b][:, inds[b, k].long()] for b in range(B)]) * wgts[:, k:k + 1] for k in range(8))
    assert torch.allclose(y, out_t, atol=1e-5)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    assert torch.allclose(O.trilinear_devoxelize_backward(gy, inds, wgts, R), gr.grad, atol=1e-5)


@pytest.mark.gpu
def test_hip_backward_equals_oracle(hip, oracle_ops):
    O, x = oracle_ops, _inputs()
    B, C, N, M, U, R = (x[k] for k in "BCNMUR")
    g = torch.Generator().manual_seed(2)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()}
    gy = torch.randn(B, C, M, generator=g)
    assert torch.allclose(hip.gather_features_backward(gy.cuda(), d["idx"], N).cpu(), O.gather_features_backward(gy, x["idx"], N), atol=1e-5)
    gy = torch.randn(B, C, M, U, generator=g)
    assert torch.allclose(hip.grouping_backward(gy.cuda(), d["nbr"], N).cpu(), O.grouping_backward(gy, x["nbr"], N), atol=1e-5)
    out, idx, w = O.three_nearest_neighbors_interpolate_forward(x["pts"], x["ctr"], x["cfeat"])
    gy = torch.randn(B, C, N, generator=g)
    got = hip.three_nearest_neighbors_interpolate_backward(gy.cuda(), idx.cuda(), w.cuda(), M).cpu()
    assert torch.allclose(got, O.three_nearest_neighbors_interpolate_backward(gy, idx, w, M), atol=1e-5)
    vox, ind, cnt = O.avg_voxelize_forward(x["feat"], x["vc"], R)
    gy = torch.randn(B, C, R ** 3, generator=g)
    assert torch.equal(hip.avg_voxelize_backward(gy.cuda(), ind.cuda(), cnt.cuda()).cpu(), O.avg_voxelize_backward(gy, ind, cnt))
    out_t, inds, wgts = O.trilinear_devoxelize_forward(R, True, x["nc"], x["grid"])
    h_out, h_inds, h_wgts = hip.trilinear_devoxelize_forward(R, True, d["nc"], d["grid"])
    assert torch.equal(h_out.cpu(), out_t) and torch.equal(h_inds.cpu(), inds) and torch.equal(h_wgts.cpu(), wgts)
    gy = torch.randn(B, C, N, generator=g)
    assert torch.allclose(hip.trilinear_devoxelize_backward(gy.cuda(), h_inds, h_wgts, R).cpu(),
                          O.trilinear_devoxelize_backward(gy, inds, wgts, R), atol=1e-5)


@pytest.mark.gpu
def test_autograd_flows_through_the_functional_api(hip):
    """reference-style use: F.grouping / F.avg_voxelize / F.trilinear_devoxelize / F.nearest_neighbor_interpolate in a graph."""
    from bdm_amd import functional as F
    x = _inputs()
    R = x["R"]
    f = x["feat"].cuda().requires_grad_()
    vox = F.avg_voxelize(f, x["vc"].cuda(), R)                                   # (B, C, R, R, R)
    back = F.trilinear_devoxelize(vox, x["nc"].cuda(), R, True)                  # (B, C, N)
    grp = F.grouping(back, x["nbr"].cuda())                                      # (B, C, M, U)
    ctr_feat = grp.max(dim=-1).values                                            # (B, C, M)
    up = F.nearest_neighbor_interpolate(x["pts"].cuda(), x["ctr"].cuda(), ctr_feat)
    loss = (up * up).sum() + F.gather(f, x["idx"].cuda()).sum()
    loss.backward()
    assert f.grad is not None and f.grad.shape == f.shape and bool(torch.isfinite(f.grad).all()) and float(f.grad.abs().sum()) > 0
