"""oracle/ref_net.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-PyTorch (CPU, fp32) functional restatement of the reference's point-voxel denoisers,
driven by a state dict that uses the reference's own key names, so that the very same
weights can be fed to (i) the reference's nn.Modules imported in the build container
(oracle/gen_golden.py), (ii) this oracle, and (iii) the HIP product path (bdm_amd/).

Follows, line by line:
  experiments/model/pvcnn/pvcnn.py:78-150             (PVCNN2_PC2 wiring, block tables)
  experiments/pvd/model/pvcnn_generation.py:225-245   (PVD denoiser: same wiring)
  experiments/pvd/__init__.py:301-312                 (PVD block tables)
  experiments/model/pvcnn/pvcnn_utils.py:71-185       (builders: which blocks exist, time embedding)
  experiments/model/pvcnn/pvcnn_fuse.py:125-237       (Merging network)
  experiments/model/pvcnn/modules/{pvconv,pointnet,ball_query,shared_mlp,se,voxelization}.py

Parity status: the module arithmetic is PINNED against the reference's own modules
(tests/golden/*.npz, produced by oracle/gen_golden.py with the reference imported);
the seven native ops inside it are the oracle's C restatement ("parity unpinned", see
oracle/pvcnn_ops_ref.c).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O

# block tables shared by PVCNN2_PC2 (pvcnn.py:130-143), PVCNN2_PVD (pvd/__init__.py:300-312)
# and PVCNN_fuse (pvcnn_fuse.py:240-252)
SA_BLOCKS = [
    ((32, 2, 32), (1024, 0.1, 32, (32, 64))),
    ((64, 3, 16), (256, 0.2, 32, (64, 128))),
    ((128, 3, 8), (64, 0.4, 32, (128, 256))),
    (None, (16, 0.8, 32, (256, 256, 512))),
]
FP_BLOCKS = [
    ((256, 256), (256, 3, 8)),
    ((256, 256), (256, 3, 8)),
    ((256, 128), (128, 2, 16)),
    ((128, 128, 64), (64, 2, 32)),
]


def swish(x):
    return x * torch.sigmoid(x)


def timestep_embedding(t, dim):
    """pvcnn_utils.py:171-185 (PC2) / pvcnn_generation.py:203-216 (PVD)."""
    half = dim // 2
    e = np.log(10000) / (half - 1)
    freq = torch.from_numpy(np.exp(np.arange(0, half) * -e)).float()
    arg = t[:, None] * freq[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1).float()


def embedf(sd, pre, t, dim):
    h = F.linear(timestep_embedding(t, dim), sd[pre + "0.weight"], sd[pre + "0.bias"])
    h = F.leaky_relu(h, 0.1)
    return F.linear(h, sd[pre + "2.weight"], sd[pre + "2.bias"])


def shared_mlp(sd, pre, x):
    """shared_mlp.py:25-37: [conv k1 -> GroupNorm(8) -> Swish] * L on (B,C,N) or (B,C,M,U)."""
    i = 0
    while f"{pre}layers.{3 * i}.weight" in sd:
        w, b = sd[f"{pre}layers.{3 * i}.weight"], sd[f"{pre}layers.{3 * i}.bias"]
        x = F.conv2d(x, w, b) if x.dim() == 4 else F.conv1d(x, w, b)
        x = F.group_norm(x, 8, sd[f"{pre}layers.{3 * i + 1}.weight"], sd[f"{pre}layers.{3 * i + 1}.bias"])
        x = swish(x)
        i += 1
    return x


def attention(sd, pre, x):
    """pvconv.py:40-63 (no 1/sqrt(C) scale)."""
    B, C = x.shape[:2]
    flat = x.reshape(B, C, -1)

    def proj(n):
        return F.conv1d(flat, sd[pre + n + ".weight"].reshape(C, C, 1), sd[pre + n + ".bias"])

    q, k, v = proj("q"), proj("k"), proj("v")
    w = torch.softmax(torch.matmul(q.permute(0, 2, 1), k), -1)
    h = torch.matmul(v, w.permute(0, 2, 1))
    h = F.conv1d(h, sd[pre + "out.weight"].reshape(C, C, 1), sd[pre + "out.bias"]).reshape(x.shape)
    y = F.group_norm(h + x, 8, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return swish(y)


def voxel_coords(coords, r):
    """voxelization.py:16-25 with normalize=True, eps=0."""
    nc = coords - coords.mean(2, keepdim=True)
    nc = nc / (nc.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0 + 0) + 0.5
    nc = torch.clamp(nc * r, 0, r - 1)
    return nc, torch.round(nc).to(torch.int32)


def pvconv(sd, pre, features, coords, r, with_attention):
    """pvconv.py:74-97 (eval mode: Dropout is the identity)."""
    B = features.shape[0]
    nc, vc = voxel_coords(coords, r)
    vox = O.avg_voxelize_forward(features.contiguous(), vc.contiguous(), r)[0].view(B, -1, r, r, r)
    v = F.conv3d(vox, sd[pre + "voxel_layers.0.weight"], sd[pre + "voxel_layers.0.bias"], padding=1)
    v = swish(F.group_norm(v, 8, sd[pre + "voxel_layers.1.weight"], sd[pre + "voxel_layers.1.bias"]))
    v = F.conv3d(v, sd[pre + "voxel_layers.4.weight"], sd[pre + "voxel_layers.4.bias"], padding=1)
    v = F.group_norm(v, 8, sd[pre + "voxel_layers.5.weight"], sd[pre + "voxel_layers.5.bias"])
    v = attention(sd, pre + "voxel_layers.6.", v) if with_attention else swish(v)
    # SE3d (se.py:8-19), with_se_relu=True
    s = v.mean(-1).mean(-1).mean(-1)
    s = torch.sigmoid(F.linear(F.relu(F.linear(s, sd[pre + "voxel_layers.7.fc.0.weight"])),
                               sd[pre + "voxel_layers.7.fc.2.weight"]))
    v = v * s.view(B, -1, 1, 1, 1)
    dv = O.trilinear_devoxelize_forward(r, False, nc.contiguous(), v.reshape(B, v.shape[1], -1).contiguous())[0]
    return dv + shared_mlp(sd, pre + "point_features.", features)


def sa_module(sd, pre, features, coords, temb, num_centers, radius, num_neighbors):
    """pointnet.py:80-90 + ball_query.py:16-30 (single-radius form)."""
    coords = coords.contiguous()
    idx = O.furthest_point_sampling(coords, num_centers)
    centers = O.gather_features_forward(coords, idx)
    nb = O.ball_query(centers, coords, radius, num_neighbors)
    g_xyz = O.grouping_forward(coords, nb) - centers.unsqueeze(-1)
    g = torch.cat([g_xyz, O.grouping_forward(features.contiguous(), nb)], dim=1)
    g_t = O.grouping_forward(temb.contiguous(), nb)
    out = shared_mlp(sd, pre + "mlps.0.", g).max(dim=-1).values
    return out, centers, g_t.max(dim=-1).values


def fp_module(sd, pre, points_coords, centers_coords, centers_features, points_features, temb):
    """pointnet.py:101-113."""
    pc, cc = points_coords.contiguous(), centers_coords.contiguous()
    interp = O.three_nearest_neighbors_interpolate_forward(pc, cc, centers_features.contiguous())[0]
    interp_t = O.three_nearest_neighbors_interpolate_forward(pc, cc, temb.contiguous())[0]
    if points_features is not None:
        interp = torch.cat([interp, points_features], dim=1)
    return shared_mlp(sd, pre + "mlp.", interp), interp_t


def _scaled(v, wm):
    return int(wm * v)


def encoder(sd, pre, inputs, temb, embed_dim=64, wm=1, vrm=1, use_att=True, sa_name="sa_layers.", att_name="global_att."):
    """Down path + global attention (pvcnn.py:90-110).  Returns features, coords, temb, skips."""
    coords = inputs[:, :3, :].contiguous()
    features = inputs
    coords_list, feats_list = [], []
    for i, (conv_cfg, sa_cfg) in enumerate(SA_BLOCKS):
        feats_list.append(features)
        coords_list.append(coords)
        x = features if i == 0 else torch.cat([features, temb], dim=1)
        nblk = 0
        if conv_cfg is not None:
            _, num_blocks, res = conv_cfg
            # pvcnn_utils.py:98-101: only the FIRST PVConv exists at levels > 0
            nblk = num_blocks if i == 0 else 1
            for p in range(nblk):
                att = ((i + 1) % 2 == 0) and use_att and p == 0
                x = pvconv(sd, f"{pre}{sa_name}{i}.{p}.", x, coords, int(vrm * res), att)
        m, radius, u, _ = sa_cfg
        sa_pre = f"{pre}{sa_name}{i}.{nblk}." if nblk > 0 else f"{pre}{sa_name}{i}."
        features, coords, temb = sa_module(sd, sa_pre, x, coords, temb, m, radius, u)
    if use_att:
        features = attention(sd, pre + att_name, features)
    return features, coords, temb, coords_list, feats_list


def decoder(sd, pre, features, coords, temb, coords_list, skips, vrm=1, fp_name="fp_layers.", cls_name="classifier."):
    """Up path + head (pvcnn.py:112-127)."""
    for k, (_, conv_cfg) in enumerate(FP_BLOCKS):
        features, temb = fp_module(sd, f"{pre}{fp_name}{k}.0.", coords_list[-1 - k], coords,
                                   torch.cat([features, temb], dim=1), skips[-1 - k], temb)
        coords = coords_list[-1 - k]
        _, num_blocks, res = conv_cfg
        for p in range(num_blocks):
            # pvcnn_utils.py:139,150: attention flag is always False here (shadowed variable)
            features = pvconv(sd, f"{pre}{fp_name}{k}.{p + 1}.", features, coords, int(vrm * res), False)
    h = shared_mlp(sd, pre + cls_name + "0.", features)
    return F.conv1d(h, sd[pre + cls_name + "2.weight"], sd[pre + cls_name + "2.bias"])


def pvcnn_forward(sd, inputs, t, prefix="", embed_dim=64, wm=1, vrm=1, use_att=True):
    """PVCNN2Base_PC2.forward / PVCNN2Base_PVD.forward.  inputs (B, 3+S, N) channel-first; t (B,)."""
    N = inputs.shape[-1]
    temb = embedf(sd, prefix + "embedf.", t, embed_dim)[:, :, None].expand(-1, -1, N)
    features, coords, temb, coords_list, feats_list = encoder(sd, prefix, inputs, temb, embed_dim, wm, vrm, use_att)
    feats_list[0] = inputs[:, 3:, :].contiguous()
    return decoder(sd, prefix, features, coords, temb, coords_list, feats_list, vrm)


def point_cloud_model_forward(sd, x, t, prefix="", **kw):
    """point_cloud_model.py:61-65: (B,N,C) in, (B,N,3) out."""
    return pvcnn_forward(sd, x.transpose(1, 2), t, prefix=prefix, **kw).transpose(1, 2)


def projs(sd, pre, x):
    """pvcnn_fuse.py:111-123: Conv1d - LeakyReLU(0.02) - Conv1d - zero-Conv1d."""
    h = F.conv1d(x, sd[pre + "0.weight"], sd[pre + "0.bias"])
    h = F.leaky_relu(h, 0.02)
    h = F.conv1d(h, sd[pre + "2.weight"], sd[pre + "2.bias"])
    return F.conv1d(h, sd[pre + "3.weight"], sd[pre + "3.bias"])


def pvcnn_fuse_forward(sd, recon_with_cond, from_prior, t, prefix="", mode="fusion_nstep", embed_dim=64, vrm=1):
    """PVCNNBase_fuse.forward (pvcnn_fuse.py:125-237).

    DEFINED SEMANTIC for the reference's undefined behaviour: the
    reference feeds the 16-point t_emb left over from the PC2 encoder into the PVD encoder
    at N points (out-of-bounds gather in grouping.cu:33).  Here, and in the HIP path, the
    PVD encoder receives the time embedding re-broadcast to its own N points; the decoder
    receives the t_emb produced by the PVD encoder pass (as the reference's control flow
    does), which is point-invariant and therefore equal to the broadcast embedding.
    """
    N = recon_with_cond.shape[-1]
    te = embedf(sd, prefix + "embedf.", t, embed_dim)
    temb = te[:, :, None].expand(-1, -1, N)
    f_pc2, c_pc2, temb_pc2, coords_list, skips_pc2 = encoder(
        sd, prefix, recon_with_cond, temb, embed_dim, vrm=vrm,
        sa_name="pc2_model_sa_layers.", att_name="pc2_model_global_att.")
    skips_pc2[0] = recon_with_cond[:, 3:, :].contiguous()
    pvd_in = (from_prior if mode == "fusion_nstep" else recon_with_cond[:, :3, :]).contiguous()
    temb_pvd = te[:, :, None].expand(-1, -1, pvd_in.shape[-1])
    f_pvd, _, temb_out, _, skips_pvd = encoder(
        sd, prefix, pvd_in, temb_pvd, embed_dim, vrm=vrm,
        sa_name="pvd_model_sa_layers.", att_name="pvd_model_global_att.")
    features = projs(sd, prefix + "projs.3.", f_pvd) + f_pc2
    fused = [skips_pc2[0]]
    for i in range(3):
        fused.append(projs(sd, f"{prefix}projs.{i}.", skips_pvd[i + 1]) + skips_pc2[i + 1])
    return decoder(sd, prefix, features, c_pc2, temb_out, coords_list, fused, vrm,
                   fp_name="fusion_decoder_fp_layers.", cls_name="classifier.")
