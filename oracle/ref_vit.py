"""oracle/ref_vit.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Torch-CPU restatement of the image encoder (experiments/model/feature_model.py:85-132 with timm 0.9.7's
VisionTransformer: patch embed, CLS + positional embedding, pre-norm blocks with LayerNorm eps 1e-6, scaled dot
product attention, exact GELU, final norm) and of get_local_conditioning (projection_model.py:110-125).
"parity unpinned": timm is a third-party dependency absent from /root/reference and from this image."""
import math

import torch
import torch.nn.functional as F

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vit_tokens(sd, img, prefix="feature_model.model.", heads=6):
    x = (img - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
    w = sd[prefix + "patch_embed.proj.weight"]
    x = F.conv2d(x, w, sd[prefix + "patch_embed.proj.bias"], stride=w.shape[-1]).flatten(2).transpose(1, 2)
    x = torch.cat([sd[prefix + "cls_token"].expand(x.shape[0], -1, -1), x], dim=1) + sd[prefix + "pos_embed"]
    D = x.shape[-1]
    i = 0
    while f"{prefix}blocks.{i}.norm1.weight" in sd:
        p = f"{prefix}blocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        B, T, _ = h.shape
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, T, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        a = torch.softmax((qkv[0] * (D // heads) ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1)
        h = (a @ qkv[2]).transpose(1, 2).reshape(B, T, D)
        x = x + F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + h
        i += 1
    return F.layer_norm(x, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)


def feature_model(sd, img, prefix="feature_model.model.", heads=6):
    """FeatureModel.forward(return_type='features'): (B, D, H, W)."""
    B, _, H, W = img.shape
    feats = vit_tokens(sd, img, prefix, heads)
    hw = int(math.sqrt(feats.shape[1] - 1))
    out = feats[:, 1:, :].reshape(B, hw, hw, -1).permute(0, 3, 1, 2)
    return F.interpolate(out, size=(H, W), mode="bilinear", align_corners=False)


def local_conditioning(sd, img, prefix="feature_model.model.", colors_mean=0.5, colors_std=0.5, heads=6):
    """get_local_conditioning (local colours + local features): (B, 3 + D, H, W)."""
    return torch.cat([(img - colors_mean) / colors_std, feature_model(sd, img, prefix, heads)], dim=1)
