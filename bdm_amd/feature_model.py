"""Image encoder of the projection conditioning: ViT-S/16 (MSN) with timm-compatible state-dict keys
(experiments/model/feature_model.py:41-132; timm 0.9.7 VisionTransformer, third-party, not installed).

SCOPE NOTE: the encoder output depends only on the image, so it is HOISTED out of the
per-step loop -- the reference recomputes it at every one of the 1000 steps (projection_model.py:199)
with bit-identical results.  It runs once per batch, outside the per-step hot path, and is written with
stock torch tensor ops for now (SURVEY.md 8f row 1 "next": a HIP ViT).  Parity status: UNPINNED (timm
absent; no reference test pins it); the pretrained MSN weights are not reachable offline, so weights are
procedural unless a checkpoint provides `feature_model.model.*`.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
MODEL_KWARGS = {
    "vit_base_patch16_224_mae": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12),
    "vit_small_patch16_224_msn": dict(patch_size=16, embed_dim=384, depth=12, num_heads=6),
    "vit_large_patch7_224_msn": dict(patch_size=7, embed_dim=1024, depth=24, num_heads=16),
}


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, D = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q * (D // self.num_heads) ** -0.5) @ k.transpose(-2, -1), dim=-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, T, D))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, dim * 4)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class VisionTransformer(nn.Module):
    """Key layout of timm.models.vision_transformer.VisionTransformer(num_classes=0, global_pool='')."""

    def __init__(self, img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        x = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        return self.norm(self.blocks(x))


class FeatureModel(nn.Module):
    def __init__(self, image_size=224, model_name="vit_small_patch16_224_msn", global_pool=""):
        super().__init__()
        self.model_name = model_name
        if model_name == "identity":
            return
        self.model = VisionTransformer(img_size=image_size, **MODEL_KWARGS[model_name])
        self.feature_dim = self.model.embed_dim
        self.mean, self.std = IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
        self.fc = nn.Identity()

    def normalize(self, img):
        mean = torch.tensor(self.mean, device=img.device).view(1, 3, 1, 1)
        std = torch.tensor(self.std, device=img.device).view(1, 3, 1, 1)
        return (img - mean) / std

    @torch.no_grad()
    def forward(self, x, return_type="features", return_upscaled_features=True):
        """feature_model.py:85-132: (B,3,H,W) in [0,1] -> (B, D, H, W) bilinearly upsampled patch tokens."""
        assert return_type in {"cls_token", "features", "all"}
        if self.model_name == "identity":
            return x
        B, C, H, W = x.shape
        feats = self.model(self.normalize(x))
        if return_type == "cls_token":
            return feats[:, 0]
        B, T, D = feats.shape
        hw = int(math.sqrt(T - 1))
        out = feats[:, 1:, :].reshape(B, hw, hw, D).permute(0, 3, 1, 2)
        if return_upscaled_features:
            out = F.interpolate(out, size=(H, W), mode="bilinear", align_corners=False)
        return out if return_type == "features" else (feats[:, 0], out)
