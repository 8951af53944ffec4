"""The point-voxel denoisers of BDM on the HIP path, state-dict compatible with the reference:

  PVCNN2_PC2   experiments/model/pvcnn/pvcnn.py:10-150          (reconstruction denoiser, 390 input channels)
  PVCNN2_PVD   experiments/pvd/model/pvcnn_generation.py:172-245 + experiments/pvd/__init__.py:299-333 (prior)
  PVCNN_fuse   experiments/model/pvcnn/pvcnn_fuse.py:14-276      (BDM-Merging network)

Topology quirks reproduced on purpose: set-abstraction levels 1-2 hold ONE PVConv
although their tables say 3 (pvcnn_utils.py:98-101); voxel attention exists only in sa_layers.1.0;
feature-propagation PVConvs never get attention (pvcnn_utils.py:139,150).
"""
import os

import torch
import torch.nn as nn

from . import ops
from . import tape
from .modules import Attention, PointNetFPModule, PointNetSAModule, PVConv, SharedMLP

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


def create_pointnet2_sa_components(sa_blocks, extra_feature_channels, embed_dim=64, use_att=False, dropout=0.1,
                                   with_se=False, normalize=True, eps=0, width_multiplier=1,
                                   voxel_resolution_multiplier=1):
    """pvcnn_utils.py:71-127."""
    r, vr = width_multiplier, voxel_resolution_multiplier
    in_channels = extra_feature_channels + 3
    sa_layers, sa_in_channels = [], []
    num_centers = None
    for c, (conv_configs, sa_configs) in enumerate(sa_blocks):
        k = 0
        sa_in_channels.append(in_channels)
        blocks = []
        if conv_configs is not None:
            out_channels, num_blocks, voxel_resolution = conv_configs
            out_channels = int(r * out_channels)
            for p in range(num_blocks):
                attention = (c + 1) % 2 == 0 and use_att and p == 0
                if c == 0 or k == 0:  # levels > 0 keep only their first block (pvcnn_utils.py:98-101)
                    blocks.append(PVConv(in_channels if c == 0 else in_channels + embed_dim, out_channels,
                                         kernel_size=3, resolution=int(vr * voxel_resolution), attention=attention,
                                         dropout=dropout, with_se=with_se, with_se_relu=True, normalize=normalize, eps=eps))
                in_channels = out_channels
                k += 1
            extra_feature_channels = in_channels
        num_centers, radius, num_neighbors, out_channels = sa_configs
        out_channels = [int(r * oc) for oc in out_channels]
        blocks.append(PointNetSAModule(num_centers=num_centers, radius=radius, num_neighbors=num_neighbors,
                                       in_channels=extra_feature_channels + (embed_dim if k == 0 else 0),
                                       out_channels=out_channels, include_coordinates=True))
        in_channels = extra_feature_channels = blocks[-1].out_channels
        sa_layers.append(blocks[0] if len(blocks) == 1 else nn.Sequential(*blocks))
    return sa_layers, sa_in_channels, in_channels, 1 if num_centers is None else num_centers


def create_pointnet2_fp_modules(fp_blocks, in_channels, sa_in_channels, embed_dim=64, use_att=False, dropout=0.1,
                                with_se=False, normalize=True, eps=0, width_multiplier=1,
                                voxel_resolution_multiplier=1):
    """pvcnn_utils.py:130-168 (attention is never enabled here: shadowed variable in the reference)."""
    r, vr = width_multiplier, voxel_resolution_multiplier
    fp_layers = []
    for fp_idx, (fp_configs, conv_configs) in enumerate(fp_blocks):
        blocks = []
        out_channels = tuple(int(r * oc) for oc in fp_configs)
        blocks.append(PointNetFPModule(in_channels=in_channels + sa_in_channels[-1 - fp_idx] + embed_dim,
                                       out_channels=out_channels))
        in_channels = out_channels[-1]
        if conv_configs is not None:
            out_channels, num_blocks, voxel_resolution = conv_configs
            out_channels = int(r * out_channels)
            for _ in range(num_blocks):
                blocks.append(PVConv(in_channels, out_channels, kernel_size=3, resolution=int(vr * voxel_resolution),
                                     attention=False, dropout=dropout, with_se=with_se, with_se_relu=True,
                                     normalize=normalize, eps=eps))
                in_channels = out_channels
        fp_layers.append(blocks[0] if len(blocks) == 1 else nn.Sequential(*blocks))
    return fp_layers, in_channels


def create_classifier(in_channels, dropout, num_classes, width_multiplier=1):
    """create_mlp_components(in, [128, dropout, num_classes], classifier=True, dim=2) (pvcnn_utils.py:13-44)."""
    hidden = int(width_multiplier * 128)
    return nn.Sequential(SharedMLP(in_channels, hidden), nn.Dropout(dropout), nn.Conv1d(hidden, num_classes, 1))


def run_blocks(blocks, inputs):
    """nn.Sequential protocol of the reference: every block maps a tuple to a tuple."""
    if isinstance(blocks, nn.Sequential):
        mods = list(blocks)
        for i, blk in enumerate(mods):
            if isinstance(blk, PVConv):
                # (not a submodule registration: the successor is only looked at, PVConv._head_for_next)
                blk.__dict__["_next_pv"] = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], PVConv) else None
            inputs = blk(inputs)
        return inputs
    return blocks(inputs)


def run_classifier(classifier, features):
    h, pending = classifier[0].run(features, fold_last=True)
    last = classifier[-1]
    if pending is not None:  # GroupNorm + Swish of the hidden layer applied inside the last convolution (dropout is identity)
        return ops.pointwise_conv_gn(h, last.weight, last.bias, fold_in=pending)
    return ops.pointwise_conv(h, last.weight, last.bias)


def embed_time(embedf, t, embed_dim, n):
    """get_timestep_embedding + embedf, as a point-invariant (B, D, n) view (pvcnn.py:87-88)."""
    te = ops.time_embedding(t, embedf[0].weight, embedf[0].bias, embedf[2].weight, embedf[2].bias)
    return te[:, :, None].expand(-1, -1, n)


_side_streams = {}
SIDE_STREAM = os.environ.get("BDM_SIDE_STREAM", "1") == "1"  # sampler chain on its own stream (0: inline, for experiments)
# voxel plans of levels 1.. on the sampler's side stream (tools/coresidency/two_proc_race.py flips it)
SIDE_PLAN = os.environ.get("BDM_SIDE_PLAN", "1") == "1"
DECODER_PLAN = os.environ.get("BDM_DECODER_PLAN", "1") == "1"   # ... and those of the decoder's PVConvs no encoder level has made (FP0)
# B * N below which the sampler chain stays on the main stream.  0: always on its own stream -- since the launch count and the
# per-launch host cost came down it pays even for one small shape (B=1, N=1024: 3.48 -> 3.30 ms; B=4: 3.86 -> 3.42 ms), where
# furthest point sampling is a fifth of the forward.  (The PVConv point branch keeps its 8192-point threshold: measured slower below.)
SIDE_STREAM_MIN_POINTS = 0
SIDE_STREAM_PRIORITY = int(os.environ.get("BDM_SIDE_PRIORITY", "0"))  # torch: lower = higher priority; -1 measured no different (5.52 vs 5.53 ms per step)
DEFER_CHAIN = True  # levels 1.. of the sampler chain enqueued when the first SA module is reached
SIDE_NN = True      # 3-NN searches of the FP modules on the sampler's stream (False: each FP module searches on the main stream)
NN_PLANS = {}       # (points ptr, centres ptr) -> (points, centres, idx, w, event): both tensors are held, so a key cannot alias


EARLY_SAMPLER = os.environ.get("BDM_EARLY_SAMPLER", "1") == "1"


def _side_stream(device):
    cur = torch.cuda.current_stream(device)
    key = (device, cur.cuda_stream)  # one sampler stream per main stream: concurrent lanes do not queue behind each other
    side = _side_streams.get(key)
    if side is None:
        # (priority: see SIDE_STREAM_PRIORITY)
        side = _side_streams[key] = torch.cuda.Stream(device=device, priority=SIDE_STREAM_PRIORITY)
    return cur, side


def early_first_sampler(net, x_t):
    """The first level's furthest point sampling needs the step's cloud only: started on the sampler's stream BEFORE the projection
    conditioning of the step (raster + gather, ~140 us at B = 16), whose output the rest of the denoiser waits for.  x_t (B, N, 3)
    point-major, as the conditioning receives it; -> the centres (B, 3, M), for plan_sampling_chain of THIS step's forward (the handle
    travels with the conditioned input: ops.Conditioning.early), or None."""
    sa_layers = getattr(net, "sa_layers", None)
    if not (EARLY_SAMPLER and SIDE_STREAM and sa_layers is not None and x_t.is_cuda and x_t.dim() == 3 and x_t.shape[2] == 3
            and x_t.shape[0] * x_t.shape[1] >= SIDE_STREAM_MIN_POINTS):
        return None
    first = sa_layers[0][-1] if isinstance(sa_layers[0], nn.Sequential) else sa_layers[0]
    cur, side = _side_stream(x_t.device)
    tape.wait_stream(side, cur)
    with torch.cuda.stream(side):
        c0 = ops.transpose12(x_t) if x_t.is_contiguous() else x_t.transpose(1, 2).contiguous()   # the values of the denoiser's coordinate rows (model.get_input_with_conditioning); one library launch
        centers0 = first.sample(c0)
    return (c0, centers0, side)


def plan_sampling_chain(sa_layers, coords, early=None, fp_layers=None):
    """Furthest point sampling + ball query of ALL set-abstraction levels depend on the input coordinates only
    (1 356 strictly sequential sampler rounds on 16 CUs).  They are enqueued on a side stream so that they overlap
    the level-0 PVConvs; each SA module waits on its own event.

    Host order matters when the host, not the GPU, paces the step (one small shape): only the first level's sampler -- the
    long pole -- is enqueued up front; the other ~19 launches of the chain follow when the first SA module is reached, i.e.
    AFTER the host has fed the main stream its first PVConvs (they could not start before that sampler finishes anyway)."""
    cur, side = _side_stream(coords.device)
    tape.wait_stream(side, cur)
    first = sa_layers[0][-1] if isinstance(sa_layers[0], nn.Sequential) else sa_layers[0]
    c0 = coords.contiguous()
    if early is not None and early[2] is side and tuple(early[0].shape) == tuple(c0.shape) and early[1].shape[2] == first.num_centers:
        centers0 = early[1]   # sampled from the same cloud on this very stream, ahead of the conditioning (early_first_sampler)
    else:
        with torch.cuda.stream(side):
            centers0 = first.sample(c0)

    def level(li, c, centers):
        """ball query of level li (+ its sampler for li > 0) and the next level's voxel plan, on the side stream; -> its centres"""
        sa = sa_layers[li][-1] if isinstance(sa_layers[li], nn.Sequential) else sa_layers[li]
        if li > 0:
            centers = sa.sample(c)
        idx = sa.query(c, centers)
        ev = torch.cuda.Event()
        tape.record_event(ev, side)
        sa._planned = (centers, idx, ev, c if li > 0 else coords)
        # the voxel plan of the NEXT level's PVConvs (sort of the centres into cells, occupied-cell lists) is geometry
        # too: one single-workgroup-per-shape kernel that would otherwise sit on the main stream's critical path
        nxt = sa_layers[li + 1] if li + 1 < len(sa_layers) else None
        pv = nxt[0] if isinstance(nxt, nn.Sequential) and hasattr(nxt[0], "voxel_plan_args") else None
        args = pv.voxel_plan_args() if (pv is not None and SIDE_PLAN) else None
        if args is not None:
            plan = ops.voxel_plan(centers, *args, dilate=2 if pv.wants_compact_tail(centers.shape[0], centers.shape[2]) else
                                  (1 if pv.wants_dilated_plan(centers.shape[0], centers.shape[2]) else 0))
            plan.ready = torch.cuda.Event()
            tape.record_event(plan.ready, side)
        return centers

    # voxel plans of the DECODER's PVConvs: an FP stage's points are an encoder level's points (coords_list), so its plan is usually the
    # one an encoder PVConv of the same resolution already made; where there is none (FP0: the 64 points SA3 reads, which has no PVConv)
    # the two launches (voxel coordinates + plan) would sit on the main stream's critical path: they are geometry, planned here
    wanted = {}
    for i, blk in enumerate(fp_layers if fp_layers is not None else ()):
        pv = next((m for m in blk if hasattr(m, "voxel_plan_args")), None) if isinstance(blk, nn.Sequential) else None
        args = pv.voxel_plan_args() if (pv is not None and SIDE_PLAN and DECODER_PLAN) else None
        if args is not None:
            wanted[i] = (pv, args)

    def decoder_plan(points, fp_index):
        """(side stream) the plan of FP stage fp_index, whose points these are, unless the cache holds it already"""
        pv, args = wanted.get(fp_index, (None, None))
        if pv is not None and not ops.has_voxel_plan(points, args[0]):
            plan = ops.voxel_plan(points, *args, dilate=2 if pv.wants_compact_tail(points.shape[0], points.shape[2]) else
                                  (1 if pv.wants_dilated_plan(points.shape[0], points.shape[2]) else 0))
            plan.ready = torch.cuda.Event()
            tape.record_event(plan.ready, side)

    def remaining(c):
        with torch.cuda.stream(side):
            cents = [c]
            for li in range(1, len(sa_layers)):
                c = level(li, c, None)
                cents.append(c)
                if li + 1 < len(sa_layers):   # these centres are level li + 1's points = FP stage (L - 2 - li)'s points (decode: coords_list[-1 - i])
                    decoder_plan(c, len(sa_layers) - 2 - li)
            # the 3-NN searches of the feature-propagation modules are geometry too (level i's points against level i + 1's
            # centres): four launches that leave the main stream's critical path; an FP module finds its pair in NN_PLANS
            if SIDE_NN:
                chain = [c0] + cents
                for pts, ctr in zip(chain[:-1], chain[1:]):
                    idx, w = ops.three_nn_search(pts, ctr)
                    ev = torch.cuda.Event()
                    tape.record_event(ev, side)
                    NN_PLANS[(pts.data_ptr(), ctr.data_ptr())] = (pts, ctr, idx, w, ev)

    def rest():
        # Level 0's ball query first; the REST of the chain is enqueued by the SA module after the main stream's wait on level 0's
        # event is in its queue (first._more).  With the whole chain enqueued before that wait, the replayed step's main queue resumed
        # ~160 us after the ball query had finished -- when the side queue reached its third sampler (profiles/r04_replayed_steps.txt).
        with torch.cuda.stream(side):
            c1 = level(0, c0, centers0)
        first._more = lambda: remaining(c1)
        return first._planned

    if DEFER_CHAIN:
        first._planned = (rest, coords)
    else:
        rest()
        more, first._more = first._more, None
        more()


def encode(sa_layers, global_att, inputs, t_emb, early=None, fp_layers=None):
    """Down path (pvcnn.py:90-110)."""
    coords = ops.xyz_rows(inputs)
    ops.clear_plan_cache()  # voxel plans are valid within one encoder/decoder pass
    NN_PLANS.clear()
    for blocks in sa_layers:  # sampler plans too: never inherit one from an aborted or foreign forward
        (blocks[-1] if isinstance(blocks, nn.Sequential) else blocks)._planned = None
        (blocks[-1] if isinstance(blocks, nn.Sequential) else blocks)._more = None
        first = blocks[0] if isinstance(blocks, nn.Sequential) else blocks
        if getattr(first, "_temb_terms", None) is not None:   # ... nor the time-embedding terms of another timestep
            first._temb_terms = None
    # also inside a hipGraph capture: the side stream forks from and joins the capturing stream.  Small problems (one
    # small shape) are bound by kernel-to-kernel dispatch latency, where the extra events cost more than the overlap gains
    if coords.is_cuda and SIDE_STREAM and coords.shape[0] * coords.shape[2] >= SIDE_STREAM_MIN_POINTS:
        plan_sampling_chain(sa_layers, coords, early, fp_layers)
    features = inputs
    coords_list, in_features_list = [], []
    for i, sa_blocks in enumerate(sa_layers):
        in_features_list.append(features)
        coords_list.append(coords)
        if i == 0:
            features, coords, t_emb = run_blocks(sa_blocks, (features, coords, t_emb))
        else:
            # (split: the level's first module takes the embedding's share of its first layers as per-shape terms; the module's OWN
            # predicate decides, so encode and forward cannot disagree about whether the concatenation happened)
            first = sa_blocks[0] if isinstance(sa_blocks, nn.Sequential) else sa_blocks
            can = hasattr(first, "can_split_temb") and first.can_split_temb(features, t_emb)
            if can and i == 1:
                _enc_temb_terms(sa_layers, t_emb)
            if not can and hasattr(first, "_temb_terms"):
                first._temb_terms = None
            features, coords, t_emb = run_blocks(sa_blocks, (features if can else ops.cat_channels([features, t_emb]), coords, t_emb))
    if global_att is not None:
        features = global_att(features)
    return features, coords, t_emb, coords_list, in_features_list


def _enc_temb_terms(sa_layers, t_emb):
    """The per-shape time-embedding terms of every encoder level's first module (PVConv.temb_rows / PointNetSAModule.temb_rows) in ONE
    launch: module i gets its (B, rows_i) view as _temb_terms (a module without one computes its own)."""
    mods = [(b[0] if isinstance(b, nn.Sequential) else b) for b in list(sa_layers)[1:]]
    mods = [m for m in mods if hasattr(m, "temb_rows")]
    if not mods:
        return
    c_t = t_emb.shape[1]
    parts = [m.temb_rows(c_t) for m in mods]
    key = tuple((p.data_ptr(), p._version) for p in parts)
    owner = mods[0]
    hit = getattr(owner, "_temb_rows_all", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.cat(parts, dim=0).contiguous())
        owner._temb_rows_all = hit
    tvec = t_emb[:, :, 0].contiguous()
    bb = ops.pointwise_conv(tvec[:, :, None], hit[1])[:, :, 0]   # (B, sum rows_i)
    lo = 0
    for m, p in zip(mods, parts):
        m._temb_terms = bb[:, lo:lo + p.shape[0]]
        lo += p.shape[0]


FP_TEMB_SPLIT = os.environ.get("BDM_FP_TEMB_SPLIT", "1") == "1"  # FP modules take the point-invariant time embedding as a per-shape bias


def _fp_temb_biases(fp_layers, in_features_list, c_feat0, t_emb):
    """All FP modules' W[:, t columns] . t in ONE launch (the embedding is the same for all of them): module i gets its (B, M_i) view."""
    mods = [(b[0] if isinstance(b, nn.Sequential) else b) for b in fp_layers]
    c_t, c_feat, parts = t_emb.shape[1], c_feat0, []
    for i, fp in enumerate(mods):
        conv0 = fp.mlp.layers[0]
        c_skip = in_features_list[-1 - i].shape[1]
        if c_feat + c_t + c_skip != conv0.in_channels:
            return  # not the denoisers' layout: every module computes its own
        parts.append(fp.temb_weight(c_feat, c_t))
        c_feat = fp.mlp.layers[-3].out_channels   # (this module's output feeds the next one; its PVConvs keep the width)
        nxt = fp_layers[i]
        if isinstance(nxt, nn.Sequential) and len(nxt) > 1:
            last = nxt[-1]
            c_feat = getattr(last, "out_channels", c_feat)
    key = tuple((p.data_ptr(), p._version) for p in parts)
    owner = mods[0]
    hit = getattr(owner, "_wt_all", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.cat(parts, dim=0).contiguous())
        owner._wt_all = hit
    tvec = t_emb[:, :, 0].contiguous()
    bb = ops.pointwise_conv(tvec[:, :, None], hit[1])[:, :, 0]   # (B, sum M_i), row stride sum M_i
    lo = 0
    for fp, p in zip(mods, parts):
        fp._temb_bias = bb[:, lo:lo + p.shape[0]]
        lo += p.shape[0]


def decode(fp_layers, classifier, features, coords, t_emb, coords_list, in_features_list):
    """Up path + head (pvcnn.py:112-127)."""
    split = FP_TEMB_SPLIT and features.is_cuda and ops.is_point_invariant(t_emb)
    if split:
        _fp_temb_biases(fp_layers, in_features_list, features.shape[1], t_emb)
    for fp_idx, fp_blocks in enumerate(fp_layers):
        # (split: the module takes the embedding's share of its first layer as a per-shape bias and hands the embedding on unchanged;
        # the module's OWN predicate decides, so decode and forward cannot disagree about whether the concatenation happened)
        fp = fp_blocks[0] if isinstance(fp_blocks, nn.Sequential) else fp_blocks
        can = split and hasattr(fp, "can_split") and fp.can_split(features, in_features_list[-1 - fp_idx], t_emb)
        if not can:
            fp._temb_bias = None
        cf = features if can else ops.cat_channels([features, t_emb])
        features, coords, t_emb = run_blocks(fp_blocks, (coords_list[-1 - fp_idx], coords, cf, in_features_list[-1 - fp_idx], t_emb))
    return run_classifier(classifier, features)


class PVCNN2Base(nn.Module):
    """Shared body of PVCNN2Base_PC2 (pvcnn.py:10-127) and PVCNN2Base_PVD (pvcnn_generation.py:172-245)."""
    sa_blocks = SA_BLOCKS
    fp_blocks = FP_BLOCKS

    def __init__(self, num_classes, embed_dim, use_att=True, dropout=0.1, extra_feature_channels=3,
                 width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        assert extra_feature_channels >= 0
        self.embed_dim = embed_dim
        self.dropout = dropout
        self.width_multiplier = width_multiplier
        self.in_channels = extra_feature_channels + 3
        sa_layers, sa_in_channels, channels_sa_features, _ = create_pointnet2_sa_components(
            sa_blocks=self.sa_blocks, extra_feature_channels=extra_feature_channels, with_se=True, embed_dim=embed_dim,
            use_att=use_att, dropout=dropout, width_multiplier=width_multiplier,
            voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.sa_layers = nn.ModuleList(sa_layers)
        self.global_att = None if not use_att else Attention(channels_sa_features, 8, D=1)
        sa_in_channels[0] = extra_feature_channels  # only the last FP module sees the extra features
        fp_layers, channels_fp_features = create_pointnet2_fp_modules(
            fp_blocks=self.fp_blocks, in_channels=channels_sa_features, sa_in_channels=sa_in_channels, with_se=True,
            embed_dim=embed_dim, use_att=use_att, dropout=dropout, width_multiplier=width_multiplier,
            voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.fp_layers = nn.ModuleList(fp_layers)
        self.channels_fp_features = channels_fp_features
        self.classifier = create_classifier(channels_fp_features, dropout, num_classes, width_multiplier)
        self.embedf = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.LeakyReLU(0.1, inplace=True),
                                    nn.Linear(embed_dim, embed_dim))
        for name, m in self.named_modules():
            if isinstance(m, PVConv):
                m.bdm_name = name  # for diagnostics (saturation guard)

    @torch.no_grad()
    def forward(self, inputs, t):
        """inputs (B, 3+S, N) channel-first fp32 on the GPU; t (B,).  Returns (B, num_classes, N)."""
        cond = getattr(inputs, "_bdm_cond", None)
        inputs = inputs.contiguous()
        t_emb = embed_time(self.embedf, t, self.embed_dim, inputs.shape[-1])
        hoisted = self._hoist_targets(cond, inputs)
        if cond is not None and not cond.features_ready and not self._hoist_complete(hoisted, cond, inputs):
            cond.ensure_features()   # a layer will read the feature rows of the lazily built input: write them now
        try:
            for m in hoisted:
                m._cond = cond
            # (the first level's centres, sampled ahead of the conditioning from this very cloud: the handle vouches for the tensor)
            early = cond.early if (cond is not None and cond.x_cf.data_ptr() == inputs.data_ptr()) else None
            features, coords, t_emb, coords_list, in_features_list = encode(self.sa_layers, self.global_att, inputs, t_emb, early, self.fp_layers)
            in_features_list[0] = inputs[:, 3:, :]
            return decode(self.fp_layers, self.classifier, features, coords, t_emb, coords_list, in_features_list)
        finally:
            for m in hoisted:
                m._cond = None

    def _hoist_complete(self, hoisted, cond, inputs):
        """True when NO layer of this forward reads rows 3.. of the conditioned input: its three readers -- the point branch and the
        first convolution of the first PVConv, the skip input of the last FP module -- all take the hoisted maps (the conditions of
        PVConv._point_branch, PVConv.forward and PointNetFPModule.forward, restated)."""
        if len(hoisted) != 2 or cond.x_cf.data_ptr() != inputs.data_ptr():
            return False
        first, last = hoisted
        conv1 = first.voxel_layers[0]
        return (len(first.point_features.layers) == 3 and isinstance(conv1, nn.Conv3d) and 27 * conv1.out_channels <= 1024
                and first.conv_impl in ("bf16x6", "fp16x3") and first.sparse_first_conv and first.resolution in first.sparse_resolutions
                and getattr(last, "two_source", False) and inputs.shape[1] - 3 == cond.C)

    def _hoist_targets(self, cond, inputs):
        """Modules whose first linear map reads the raw conditioned input (ops.Conditioning): the first PVConv of the first
        set-abstraction level and the last feature-propagation module.  Empty unless this forward's input carries a handle."""
        if cond is None or not ops.HOIST_CONDITIONING or inputs.shape[1] != 3 + cond.C or cond.x_cf.data_ptr() != inputs.data_ptr():
            return []
        first = self.sa_layers[0][0] if isinstance(self.sa_layers[0], nn.Sequential) else None
        last = self.fp_layers[-1][0] if isinstance(self.fp_layers[-1], nn.Sequential) else self.fp_layers[-1]
        return [m for m in (first, last) if m is not None and hasattr(m, "_cond")]


class PVCNN2_PC2(PVCNN2Base):
    """pvcnn.py:130-150."""


class PVCNN2_PVD(PVCNN2Base):
    """pvd/__init__.py:299-333."""


class PVCNN_fuse(nn.Module):
    """pvcnn_fuse.py:14-276: PC2 encoder + PVD encoder + zero-conv projections + fine-tuned PC2 decoder copy.

    Defined semantic for the reference's out-of-bounds t_emb gather (pvcnn_fuse.py:163-165 -> 184-186,
    grouping.cu:33): the PVD encoder receives the time embedding
    re-broadcast to its own N points."""
    sa_blocks = SA_BLOCKS
    fp_blocks = FP_BLOCKS

    def __init__(self, pvd_model, pc2_model, num_classes, embed_dim, use_att=True, dropout=0.1,
                 extra_feature_channels=3, width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        assert extra_feature_channels >= 0
        self.pvd_model_sa_layers = pvd_model.model.module.sa_layers
        self.pvd_model_global_att = pvd_model.model.module.global_att
        pc2 = pc2_model.point_cloud_model.model
        self.pc2_model_sa_layers = pc2.sa_layers
        self.pc2_model_global_att = pc2.global_att
        self.pc2_model_fp_layers = pc2.fp_layers
        self.pc2_model_classiifier = pc2.classifier  # (sic) the reference's attribute name is a state-dict key
        self.pc2_model_embedf = pc2.embedf
        self.embed_dim = embed_dim
        self.dropout = dropout
        self.width_multiplier = width_multiplier
        self.in_channels = extra_feature_channels + 3
        _, sa_in_channels, channels_sa_features, _ = create_pointnet2_sa_components(
            sa_blocks=self.sa_blocks, extra_feature_channels=extra_feature_channels, with_se=True, embed_dim=embed_dim,
            use_att=use_att, dropout=dropout, width_multiplier=width_multiplier,
            voxel_resolution_multiplier=voxel_resolution_multiplier)
        sa_in_channels[0] = extra_feature_channels
        fp_layers, channels_fp_features = create_pointnet2_fp_modules(
            fp_blocks=self.fp_blocks, in_channels=channels_sa_features, sa_in_channels=sa_in_channels, with_se=True,
            embed_dim=embed_dim, use_att=use_att, dropout=dropout, width_multiplier=width_multiplier,
            voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.fusion_decoder_fp_layers = nn.ModuleList(fp_layers)
        self.channels_fp_features = channels_fp_features
        self.classifier = create_classifier(channels_fp_features, dropout, num_classes, width_multiplier)
        self.embedf = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.LeakyReLU(0.1, inplace=True),
                                    nn.Linear(embed_dim, embed_dim))
        # initialise the trainable copies from the PC2 weights (pvcnn_fuse.py:100-107)
        self.embedf.load_state_dict(self.pc2_model_embedf.state_dict())
        self.fusion_decoder_fp_layers.load_state_dict(self.pc2_model_fp_layers.state_dict())
        self.classifier.load_state_dict(self.pc2_model_classiifier.state_dict())
        projs = []
        for dim in [64, 128, 256, 512]:  # hard-coded in the reference (pvcnn_fuse.py:111-123)
            conv1, conv2, zero_conv = nn.Conv1d(dim, dim, 1), nn.Conv1d(dim, dim, 1), nn.Conv1d(dim, dim, 1)
            for p in (conv1, conv2):
                nn.init.normal_(p.weight, mean=0.0, std=(2 / dim) ** 0.5)
                nn.init.constant_(p.bias, 0)
            for p in zero_conv.parameters():
                p.detach().zero_()
            projs.append(nn.Sequential(conv1, nn.LeakyReLU(0.02, inplace=True), conv2, zero_conv))
        self.projs = nn.ModuleList(projs)

    @staticmethod
    def _proj(seq, x, add):
        h = ops.pointwise_conv(x, seq[0].weight, seq[0].bias, act=2, slope=0.02)
        h = ops.pointwise_conv(h, seq[2].weight, seq[2].bias)
        return ops.pointwise_conv(h, seq[3].weight, seq[3].bias, residual=add)

    @torch.no_grad()
    def forward(self, recon_inputs_with_cond, input_from_prior, t, mode="fusion_nstep"):
        x = recon_inputs_with_cond.contiguous()
        n = x.shape[-1]
        te = ops.time_embedding(t, self.embedf[0].weight, self.embedf[0].bias, self.embedf[2].weight, self.embedf[2].bias)
        t_emb = te[:, :, None].expand(-1, -1, n)
        f_pc2, c_pc2, _, coords_pc2_list, skips_pc2 = encode(self.pc2_model_sa_layers, self.pc2_model_global_att, x, t_emb)
        skips_pc2[0] = x[:, 3:, :]
        pvd_in = (input_from_prior if mode == "fusion_nstep" else x[:, :3, :]).contiguous()
        t_emb_pvd = te[:, :, None].expand(-1, -1, pvd_in.shape[-1])
        f_pvd, _, t_emb_out, _, skips_pvd = encode(self.pvd_model_sa_layers, self.pvd_model_global_att, pvd_in, t_emb_pvd)
        features = self._proj(self.projs[-1], f_pvd, f_pc2)
        fused = [skips_pc2[0]]
        for i in range(3):
            fused.append(self._proj(self.projs[i], skips_pvd[i + 1], skips_pc2[i + 1]))
        return decode(self.fusion_decoder_fp_layers, self.classifier, features, c_pc2, t_emb_out, coords_pc2_list, fused)
