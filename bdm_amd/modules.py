"""Point-voxel building blocks with the reference's class names, constructor signatures,
tuple-in / tuple-out forward protocol and state-dict keys
(experiments/model/pvcnn/modules/{pvconv,pointnet,ball_query,shared_mlp,se,voxelization}.py;
identical copy under experiments/pvd/modules/), so published checkpoints load unchanged.

The torch.nn layers held inside (nn.Conv1d, nn.Conv3d, nn.GroupNorm, nn.Linear) are PARAMETER
CONTAINERS only: their forward is never called.  Every forward below runs hand-written gfx950
kernels through the C ABI (bdm_amd/ops.py, bdm_amd/functional).  Inference only.
"""
import os

import torch
import torch.nn as nn

from . import functional as F
from . import ops
from . import tape

__all__ = ["Swish", "SharedMLP", "SE3d", "Voxelization", "Attention", "PVConv", "BallQuery",
           "PointNetSAModule", "PointNetFPModule"]


class Swish(nn.Module):
    """pvconv.py:12-14.  Placeholder inside nn.Sequential containers (keeps layer indices = key names);
    the activation itself is fused into the GroupNorm kernel."""

    def forward(self, x):  # pragma: no cover - never on the HIP path
        raise RuntimeError("Swish is fused into bdm_group_norm on the HIP path")


class SharedMLP(nn.Module):
    """shared_mlp.py:11-37: [Conv k=1 -> GroupNorm(8) -> Swish] per output width."""

    def __init__(self, in_channels, out_channels, dim=1):
        super().__init__()
        if dim == 1:
            conv = nn.Conv1d
        elif dim == 2:
            conv = nn.Conv2d
        else:
            raise ValueError
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [out_channels]
        layers = []
        for oc in out_channels:
            layers.extend([conv(in_channels, oc, 1), nn.GroupNorm(8, oc), Swish()])
            in_channels = oc
        self.layers = nn.Sequential(*layers)

    # GroupNorm folding (class attribute; the equality tests flip it): a layer's convolution leaves the GroupNorm statistics of its output
    # and the NEXT consumer (the next layer's convolution, or the caller's max over neighbours with fold_last) normalises +
    # Swishes on the fly -- no GroupNorm pass, the normalised tensor is never written
    fold_gn = True

    def run(self, x, out_last=None, fold_last=False, x2=None, first_weight=None, first_add=None, first_bias=None):
        """-> activations; with fold_last -> (raw output of the last convolution, (stats, gn) | None): the caller applies the
        last GroupNorm + Swish inside its own consumer kernel (None: already applied).  x2: the input is cat([x, x2], dim=1),
        read in place by the first convolution.  first_weight / first_add: the first convolution's weight restricted to the columns
        of `x` and the per-element addend that stands for its other columns (the hoisted projection conditioning:
        W . [x, F[pix]] = W_x . x + (F . W_f^T)[pix], ops.Conditioning).  first_bias (B, M): a per-shape bias of the first convolution
        (the share of input columns that are constant along a shape's points: the time embedding)."""
        n = len(self.layers) // 3
        pending = None
        for i in range(n):
            conv, gn = self.layers[3 * i], self.layers[3 * i + 1]
            last = i == n - 1
            dst = out_last if last else None
            defer = (self.fold_gn and x.is_cuda and (not last or fold_last) and ops.gn_foldable(conv.out_channels, gn.num_groups))
            second = x2 if i == 0 else None
            weight = first_weight if (i == 0 and first_weight is not None) else conv.weight
            addend = first_add if i == 0 else None
            bb = first_bias if i == 0 else None
            if defer or pending is not None or second is not None or addend is not None or bb is not None:
                r = ops.pointwise_conv_gn(x, weight, conv.bias, out=dst, fold_in=pending,
                                          out_groups=gn.num_groups if defer else None, x2=second, add=addend, batch_bias=bb)
                if defer:
                    x, stats = r
                    pending = (stats, gn)
                else:
                    x, pending = r, None
                    ops.group_norm_(x, gn.weight, gn.bias, gn.num_groups, gn.eps, swish=True)
            else:
                x = ops.pointwise_conv(x, weight, conv.bias, out=dst)
                ops.group_norm_(x, gn.weight, gn.bias, gn.num_groups, gn.eps, swish=True)
        return (x, pending) if fold_last else x

    def forward(self, inputs):
        if isinstance(inputs, (list, tuple)):
            return (self.run(inputs[0]), *inputs[1:])
        return self.run(inputs)


def hoisted_first_weight(owner, conv, keep_cols):
    """W[:, :keep_cols] (contiguous copy) of a 1x1 convolution whose remaining input columns were hoisted into a per-image map
    (ops.Conditioning); the gathered map rows enter the layer as its addend.  keep_cols may also be a tuple of (lo, hi) column ranges
    (the columns that stay, in order).  Cached on `owner` per weight version."""
    w = conv.weight
    hit = getattr(owner, "_hoist_w", None)
    if hit is None or hit[0] != (w._version, w.data_ptr(), keep_cols):
        w2 = w.detach().reshape(w.shape[0], -1)
        kept = w2[:, :keep_cols] if isinstance(keep_cols, int) else torch.cat([w2[:, lo:hi] for lo, hi in keep_cols], dim=1)
        hit = ((w._version, w.data_ptr(), keep_cols), kept.contiguous())
        owner._hoist_w = hit
    return hit[1]


class SE3d(nn.Module):
    """se.py:8-19.  forward returns the per-(shape, channel) gate; the multiplication is fused
    into the devoxelisation gather (PVConv.forward)."""

    def __init__(self, channel, reduction=8, use_relu=False):
        super().__init__()
        if not use_relu:
            raise NotImplementedError("the denoisers are built with with_se_relu=True (pvcnn_utils.py:93)")
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def gate(self, grid):
        return ops.se_gate(grid, self.fc[0].weight, self.fc[2].weight)


class Voxelization(nn.Module):
    """voxelization.py:10-25."""

    def __init__(self, resolution, normalize=True, eps=0):
        super().__init__()
        self.r = int(resolution)
        self.normalize = normalize
        self.eps = eps
        if not normalize:
            raise NotImplementedError("normalize=False is not used by the denoisers")

    def forward(self, features, coords, with_row_occupancy=False):
        norm_coords, vox_coords = ops.voxel_coords(coords, self.r, self.eps)
        return ops.avg_voxelize(features, vox_coords, self.r, with_row_occupancy), norm_coords


class Attention(nn.Module):
    """pvconv.py:17-63 (D=3 voxel attention / D=1 global attention); no 1/sqrt(C) scale."""

    def __init__(self, in_ch, num_groups, D=3):
        super().__init__()
        assert in_ch % num_groups == 0
        conv = nn.Conv3d if D == 3 else nn.Conv1d
        self.q = conv(in_ch, in_ch, 1)
        self.k = conv(in_ch, in_ch, 1)
        self.v = conv(in_ch, in_ch, 1)
        self.out = conv(in_ch, in_ch, 1)
        self.norm = nn.GroupNorm(num_groups, in_ch)
        self.nonlin = Swish()
        self._qkv = None

    def _qkv_params(self):
        """q, k, v projections as ONE (3C x C) GEMM (cached; rebuilt when a weight changes)."""
        ver = tuple(p._version for p in (self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias))
        ptr = self.q.weight.data_ptr()
        if self._qkv is None or self._qkv[0] != (ver, ptr):
            C = self.q.weight.shape[0]
            w = torch.cat([self.q.weight.reshape(C, C), self.k.weight.reshape(C, C), self.v.weight.reshape(C, C)], 0).contiguous()
            b = torch.cat([self.q.bias, self.k.bias, self.v.bias], 0).contiguous()
            self._qkv = ((ver, ptr), w, b)
        return self._qkv[1], self._qkv[2]

    def forward(self, x):
        B, C = x.shape[:2]
        flat = x.reshape(B, C, -1)
        w, b = self._qkv_params()
        if flat.is_cuda and ops.attention_h2_ok(C, flat.shape[2]):
            # fp16x3 attention: the projection GEMM leaves max |q|, |k|, |v| (the operands' power-of-two scales)
            amax = ops.amax_slots(flat.device, 3 * flat.shape[0])  # per shape: max |q|, |k|, |v|
            qkv = ops.pointwise_conv_gn(flat, w, b, amax=amax, amax_rows=C)
            h = ops.attention_core(qkv, C, amax=amax)
        else:
            qkv = ops.pointwise_conv(flat, w, b)
            h = ops.attention_core(qkv, C)
        h = ops.pointwise_conv(h, self.out.weight, self.out.bias)
        ops.group_norm_(h, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, swish=True,
                        residual=flat)
        return h.reshape(x.shape)


class PVConv(nn.Module):
    """pvconv.py:65-97."""

    def __init__(self, in_channels, out_channels, kernel_size, resolution, attention=False,
                 dropout=0.1, with_se=False, with_se_relu=False, normalize=True, eps=0):
        super().__init__()
        assert kernel_size == 3, "the denoisers use 3x3x3 voxel convolutions only"
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.resolution = resolution
        self.voxelization = Voxelization(resolution, normalize=normalize, eps=eps)
        voxel_layers = [nn.Conv3d(in_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
                        nn.GroupNorm(num_groups=8, num_channels=out_channels), Swish()]
        voxel_layers += [nn.Dropout(dropout)] if dropout is not None else []
        voxel_layers += [nn.Conv3d(out_channels, out_channels, kernel_size, stride=1, padding=kernel_size // 2),
                         nn.GroupNorm(num_groups=8, num_channels=out_channels),
                         Attention(out_channels, 8) if attention else Swish()]
        if with_se:
            voxel_layers.append(SE3d(out_channels, use_relu=with_se_relu))
        self.voxel_layers = nn.Sequential(*voxel_layers)
        self.point_features = SharedMLP(in_channels, out_channels)
        self._packed = {}

    # Voxel-convolution arithmetic (both are fp32-accurate, tests/test_hip_dense.py):
    #   "fp16x3": (default) second convolution on two-term fp16 operands, three partial products (conv3d_h2.hip); the
    #             first convolution (raw point features, unbounded range) stays on the bf16x6 kernels;
    #   "bf16x6": exact 3-way bf16 split of both operands, six partial products on the bf16 matrix cores;
    #   "fp32"  : v_mfma_f32_32x32x2_f32 kernels of conv3d.hip (BDM_CONV=fp32).
    conv_impl = os.environ.get("BDM_CONV", "fp16x3")
    fold_gn2 = True  # second GroupNorm folded into its consumers (fp16x3 path)
    sparse_first_conv = True
    sparse_resolutions = {8, 16, 32}
    # first convolution on the occupied voxels: "fp16x3" (default: batched GEMM on two-term fp16 operands, activation scale
    # from a device-side max, + gather) | "bf16x6" (same structure, exact 3-way bf16 split, twice the matrix work) | "fp32"
    # (fp32 MFMA) | "fused" (sparse_conv_fused.hip: one kernel, accumulators in LDS; correct and deterministic but measured
    # SLOWER on MI355X -- DESIGN.md section 7 -- so it is opt-in)
    sparse_gemm = {"fp32": "sparse", "bf16x6": "sparse_s3", "fused": "sparse_fused"}.get(os.environ.get("BDM_SPARSE_GEMM", "fp16x3"), "sparse_h2")

    # formulation of the first convolution: "dil" = output-stationary implicit GEMM over tiles of the once-dilated voxel list, tap
    # skipping, compact output (sparse_conv_os.hip, one launch after the feature records) |
    # "gemm" = batched GEMM over the occupied rows + output-stationary gather (27x intermediate)
    # The output-stationary form computes every (dilated voxel, tap) pair -- ~3.5x the matrix work of the GEMM over the occupied
    # (cell, tap) pairs -- and wins where the GEMM's 27x intermediate is what costs: the 32^3 levels (measured at B = 16,
    # profiles/r04_sparse_dil_probe.txt: 64 -> 64: 188 vs 235 us with the operand split, 32 -> 32: 85 vs 128; 16^3: 127 vs 127;
    # 8^3: 114 vs 62 -- the small grids stay on GEMM + gather)
    sparse_conv = os.environ.get("BDM_SPARSE_CONV", "dil")
    sparse_dil_resolutions = {int(v) for v in os.environ.get("BDM_SPARSE_DIL_R", "32,16").split(",") if v}
    sparse_dil_always = os.environ.get("BDM_SPARSE_DIL_ALWAYS", "0") == "1"   # ignore ops.sparse_dil_pays (tests, A/B timing)

    def _packed_weight(self, conv, impl, cin=None):
        """cin: pack the first `cin` input channels only (the others enter as a per-shape addend: forward's time-embedding split)"""
        key = (id(conv), impl, cin)
        sig = (conv.weight._version, conv.weight.data_ptr())
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            pack = {"bf16x6": ops.conv3d_s3_pack, "fp16x3": ops.conv3d_h2_pack, "sparse": ops.sparse_conv_pack, "class": ops.conv_class_pack,
                    "sparse_s3": ops.sparse_conv_pack_s3, "sparse_fused": ops.sparse_conv_pack_fused, "sparse_h2": ops.sparse_conv_pack_h2,
                    "fp32": ops.conv3d_pack}[impl]
            w = conv.weight.detach()
            hit = (sig, pack(w if cin is None else w[:, :cin].contiguous()))
            self._packed[key] = hit
        return hit[1]

    # The time embedding as a per-shape term: the reference concatenates the point-invariant embedding to the features before
    # the first PVConv of set-abstraction levels 1.. (pvcnn.py:103).  Its voxel mean is the same constant on every occupied cell, so its share of
    # the first convolution is u[b][tap][co] = W[co][c_feat:, tap] . t[b], added once per occupied neighbour: a column addend of the occupied-row
    # GEMM (bdm_sparse_conv_gemm_*_cb); its share of the point branch is a per-shape bias.  No concatenation, K = the real feature channels.
    temb_split = os.environ.get("BDM_ENC_TEMB_SPLIT", "1") == "1"
    _temb_terms = None   # (B, 28 * cout) = [u | point-branch bias] of this forward when pvcnn.encode computed all levels' in one launch

    def temb_rows(self, c_t):
        """(27 * cout + cout, c_t): the embedding's columns of the first voxel convolution (row tap * cout + co) and of the point branch"""
        conv1, pconv = self.voxel_layers[0], self.point_features.layers[0]
        key = (conv1.weight._version, conv1.weight.data_ptr(), pconv.weight._version, pconv.weight.data_ptr(), c_t)
        hit = getattr(self, "_temb_rows", None)
        if hit is None or hit[0] != key:
            cout, c_feat = conv1.out_channels, self.in_channels - c_t
            wv = conv1.weight.detach()[:, c_feat:].reshape(cout, c_t, 27).permute(2, 0, 1).reshape(27 * cout, c_t)
            wp = pconv.weight.detach().reshape(pconv.out_channels, -1)[:, c_feat:]
            hit = (key, torch.cat([wv, wp], dim=0).contiguous())
            self._temb_rows = hit
        return hit[1]

    def can_split_temb(self, features, temb):
        """May this module take `features` WITHOUT the concatenated embedding?  The one predicate of pvcnn.encode and of forward."""
        c_feat = features.shape[1]
        return (self.temb_split and features.is_cuda and ops.is_point_invariant(temb) and not ops.is_point_invariant(features)
                and c_feat + temb.shape[1] == self.in_channels and c_feat % 8 == 0
                and self.conv_impl in ("bf16x6", "fp16x3") and self.sparse_first_conv and self.resolution in self.sparse_resolutions
                and self.sparse_gemm in ("sparse_h2", "sparse_s3") and len(self.point_features.layers) == 3
                and self.point_features.layers[0].out_channels == self.voxel_layers[0].out_channels
                and not self.wants_dilated_plan(features.shape[0], features.shape[2]))

    def _temb_split_terms(self, features, temb):
        """-> (col_bias (B, 27 * cout), point-branch bias (B, cout)) views"""
        cout, B = self.voxel_layers[0].out_channels, features.shape[0]
        terms, self._temb_terms = self._temb_terms, None
        if terms is None or tuple(terms.shape) != (B, 28 * cout):
            tvec = temb[:, :, 0].contiguous()
            terms = ops.pointwise_conv(tvec[:, :, None], self.temb_rows(temb.shape[1]))[:, :, 0]
        return terms[:, :27 * cout], terms[:, 27 * cout:]

    # The point branch (1x1 conv + GroupNorm + Swish on the N points) does not depend on the voxel branch: it is enqueued on
    # a second stream and joined before the devoxelisation adds it, so its small, latency-bound kernels run beside the
    # voxel convolutions instead of after them (point_stream = False: serial).
    se_in_devox = False  # SE block's FC layers inside the devoxelisation kernel: measured slower (DESIGN.md negative results)
    fold_gn1 = True  # GroupNorm-1 statistics from the sparse gather's epilogue
    fold_pf = True  # point branch's GroupNorm folded into the devoxelisation kernel
    point_stream = True  # the point branch of a PVConv on its own stream (False: inline; tests and tools/coresidency/two_proc_race.py flip it)
    point_stream_min = int(os.environ.get("BDM_POINT_STREAM_MIN", "8192"))  # B * N below which the branch stays on the main stream
    _streams = {}

    def voxel_plan_args(self):
        """(resolution, eps) of the ops.voxel_plan this module's forward will ask for, or None when it voxelises densely."""
        if self.conv_impl in ("bf16x6", "fp16x3") and self.sparse_first_conv and self.resolution in self.sparse_resolutions:
            return self.resolution, self.voxelization.eps
        return None

    def wants_dilated_plan(self, batch, n_points):
        """Will this module's first convolution run in the compact output-stationary form for (batch, n_points)?  (the side-stream
        planner then builds the dilated list with the plan, off the main stream)"""
        conv1 = self.voxel_layers[0]
        return (self.sparse_conv == "dil" and self.resolution in self.sparse_dil_resolutions and self.conv_impl == "fp16x3"
                and self.sparse_gemm == "sparse_h2" and not getattr(self, "h2_saturated", False) and self.sparse_first_conv
                and (self.sparse_dil_always or ops.sparse_dil_pays(batch, n_points, self.resolution, conv1.out_channels)))

    # The whole voxel branch on voxel lists (pvconv_compact.hip): second convolution on the twice-dilated list, SE means and
    # devoxelisation from its rows + 27 class constants -- no dense grid is written.  "1": where ops.compact_tail_pays; "always"; "0".
    compact_tail = os.environ.get("BDM_COMPACT_TAIL", "1")
    compact_tail_resolutions = {int(v) for v in os.environ.get("BDM_COMPACT_TAIL_R", "32,16").split(",") if v}

    def wants_compact_tail(self, batch, n_points):
        conv2 = self.voxel_layers[4] if isinstance(self.voxel_layers[3], nn.Dropout) else self.voxel_layers[3]
        if not isinstance(conv2, nn.Conv3d):
            conv2 = next(m for m in list(self.voxel_layers)[1:] if isinstance(m, nn.Conv3d))
        c = conv2.out_channels
        has_att = any(isinstance(m, Attention) for m in self.voxel_layers)
        has_se = any(isinstance(m, SE3d) for m in self.voxel_layers)
        return (self.compact_tail != "0" and self.resolution in self.compact_tail_resolutions and self.conv_impl == "fp16x3"
                and self.sparse_first_conv and self.fold_gn1 and self.fold_gn2 and not self.se_in_devox and not has_att and has_se
                and not getattr(self, "h2_saturated", False) and c % 4 == 0 and c <= 256 and (c // 8) in (4, 8, 16, 32)
                and (self.compact_tail == "always" or ops.compact_tail_pays(batch, n_points, self.resolution, c)))

    _cond = None  # ops.Conditioning of this forward when the input is the raw conditioned cloud (set by PVCNN2Base.forward)
    _next_pv = None  # the PVConv that consumes this one's output inside the same nn.Sequential (pvcnn.run_blocks), else None

    def accepts_rows(self, plan, channels, batch):
        """Will this module's first convolution take the fp16x3 GEMM over the occupied rows of `plan` (so that the previous PVConv's
        tail may leave it the operand, ops.VoxelRows)?  Mirrors the route choice in forward."""
        conv1 = self.voxel_layers[0]
        return (ops.SMALL_GLUE and self.conv_impl == "fp16x3" and self.sparse_first_conv and self.resolution == plan.r
                and self.resolution in self.sparse_resolutions and self.sparse_gemm == "sparse_h2" and plan.n_max <= 256
                and conv1.in_channels == channels and channels % 8 == 0 and not getattr(self, "h2_saturated", False)
                and not self.wants_dilated_plan(batch, plan.n))

    def _head_for_next(self, plan, gn2, batch, device):
        """(plan, power-of-two scale, saturation word) for bdm_pvconv_tail_small's head, or None: the next PVConv of the Sequential runs on
        the same voxel plan (same coordinates, same resolution) and takes the operand."""
        nxt = self._next_pv
        if nxt is None or plan is None or not nxt.accepts_rows(plan, self.out_channels, batch):
            return None
        pf_gn = self.point_features.layers[-2]
        return plan, ops.h2_sum_scale([gn2, pf_gn]), ops.saturation_slot(nxt, device)

    def _hoisted(self, features):
        """The handle, if `features` IS the conditioned input it describes (first PVConv of the PC^2 denoiser)."""
        cond = self._cond
        if cond is None or features.data_ptr() != cond.x_cf.data_ptr() or features.shape[1] != 3 + cond.C:
            return None
        return cond

    def _need_features(self, features):
        """A reader that does NOT take the hoisted map is about to read rows 3.. of the conditioned input: if this step's input was
        built lazily (coordinate rows only, ops.Conditioning), complete it now, on the CURRENT stream (callers call this before they
        fork a branch stream).  PVCNN2Base._hoist_complete is only the fast path that spares the launch; the readers protect themselves."""
        cond = self._cond
        if cond is not None and not cond.features_ready and features.data_ptr() == cond.x_cf.data_ptr():
            cond.ensure_features()

    def _point_branch(self, features, fold=False, temb_bias=None):
        """-> (activations, event | None, pending): with fold the LAST GroupNorm + Swish of the branch is left to the caller
        (pending = (stats, gn), activations = raw convolution output; None when the layer cannot be folded).
        temb_bias (B, cout): `features` arrive without the concatenated time embedding, whose share is this per-shape bias."""
        cond = self._hoisted(features) if temb_bias is None else None
        if cond is not None and len(self.point_features.layers) != 3:
            cond = None
        if cond is None:
            self._need_features(features)   # (on the caller's stream, before the branch stream forks from it)

        def run():
            x, first_weight, first_add = features, None, None
            if temb_bias is not None:
                first_weight = hoisted_first_weight(self.point_features, self.point_features.layers[0], features.shape[1])
                if fold:
                    return self.point_features.run(x, fold_last=True, first_weight=first_weight, first_bias=temb_bias)
                return self.point_features.run(x, first_weight=first_weight, first_bias=temb_bias), None
            if cond is not None:
                # W . [xyz, F[pix]] = Wx . xyz + (F . Wf^T)[pix]: gather 32 map channels instead of convolving 390 (ops.Conditioning).
                # The gather runs HERE, on the branch's stream: its output is allocated on the stream that consumes it (a block of
                # the main stream freed at return could be handed out again while the branch still reads it)
                conv = self.point_features.layers[0]
                fmap = cond.map("point_branch", conv.weight,
                                lambda conv=conv, C=cond.C: conv.weight.detach().reshape(conv.out_channels, -1)[:, 3:3 + C])  # (holds no step tensors: the closure is cached)
                gathered = cond.gather(fmap)                         # (B, 3 + M, N): rows 0..2 = xyz, the rest = map rows
                x, first_add = gathered[:, :3], gathered[:, 3:]
                first_weight = hoisted_first_weight(self.point_features, conv, 3)
            if fold:
                return self.point_features.run(x, fold_last=True, first_weight=first_weight, first_add=first_add)
            return self.point_features.run(x, first_weight=first_weight, first_add=first_add), None
        if not (self.point_stream and features.is_cuda and features.shape[0] * features.shape[2] >= self.point_stream_min):
            pf, pending = run()
            return pf, None, pending
        dev = features.device
        cur = torch.cuda.current_stream(dev)
        side = PVConv._streams.get((dev, cur.cuda_stream))  # one branch stream per main stream (concurrent lanes stay independent)
        if side is None:
            side = PVConv._streams[(dev, cur.cuda_stream)] = torch.cuda.Stream(device=dev)
        tape.wait_stream(side, cur)
        with torch.cuda.stream(side):
            pf, pending = run()
            ev = torch.cuda.Event()
            tape.record_event(ev, side)
        return pf, ev, pending

    def forward(self, inputs):
        features, coords, temb = inputs
        r = self.resolution
        layers = list(self.voxel_layers)
        conv1, gn1 = layers[0], layers[1]
        rest = [m for m in layers[2:] if isinstance(m, (nn.Conv3d, nn.GroupNorm, Attention, SE3d))]
        conv2, gn2 = rest[0], rest[1]
        att = next((m for m in rest if isinstance(m, Attention)), None)
        se = next((m for m in rest if isinstance(m, SE3d)), None)

        rows_in = getattr(features, "_bdm_rows", None)   # first-convolution operand left by the previous PVConv's tail (ops.VoxelRows)
        features = ops.materialize(features)
        col_bias, pb_bias, cin1 = None, None, None
        if features.shape[1] != self.in_channels:   # the caller (pvcnn.encode) left the time embedding out: it enters as per-shape terms
            if not self.can_split_temb(features, temb):
                raise ValueError(f"PVConv expects {self.in_channels} input channels, got {features.shape[1]} "
                                 f"(and the time embedding cannot enter as a per-shape term here)")
            col_bias, pb_bias = self._temb_split_terms(features, temb)
            cin1 = features.shape[1]
        gn1_stats, plan, xh_ready = None, None, None
        cg2_, tile_ = conv2.out_channels // gn2.num_groups, (64 if (conv2.out_channels > 32 and r != 8) else 32)
        folded_tail = (self.conv_impl == "fp16x3" and not getattr(self, "h2_saturated", False) and self.fold_gn2 and att is None
                       and se is not None and cg2_ in (4, 8, 16, 32) and tile_ % cg2_ == 0)
        # with the folded tail the branch's GroupNorm + Swish is applied by the devoxelisation kernel (one launch less)
        pf, pf_ready, pf_pending = self._point_branch(features, fold=folded_tail and self.fold_pf and not self.se_in_devox, temb_bias=pb_bias)
        if self.conv_impl in ("bf16x6", "fp16x3"):
            if self.sparse_first_conv and r in self.sparse_resolutions:
                # conv1 sees the freshly voxelised cloud: evaluate it on the occupied cells only (sparse_conv.hip);
                # the (coords, r) plan is shared by the PVConvs of one level
                plan = ops.voxel_plan(coords, r, self.voxelization.eps)
                norm_coords = plan.norm_coords
                # the fp16x3 GEMM (half the matrix work) wins on the small grids, where the GEMM is matrix-bound and
                # 64-row tiles cut the padding; on the 16^3 / 32^3 levels the batched GEMM is bound by its 27x-expanded
                # output and the extra operand split costs more than it saves (measured: tools/sparse_bench.py)
                impl = self.sparse_gemm
                wide16 = plan.n_max <= 1024 and (cin1 or conv1.in_channels) >= 128 and conv1.out_channels >= 128  # 108 vs 121 us at 16^3
                if impl == "sparse_h2" and plan.n_max > 256 and not wide16:
                    impl = "sparse_s3"
                # with the fp16x3 second convolution the gather also leaves GroupNorm-1's statistics (no pass over the grid for them)
                want_stats = (self.fold_gn1 and self.conv_impl == "fp16x3" and not getattr(self, "h2_saturated", False)
                              and impl != "sparse_fused")
                cond = self._hoisted(features) if col_bias is None else None
                if not (cond is not None and 27 * conv1.out_channels <= 1024):
                    self._need_features(features)
                if cond is not None and 27 * conv1.out_channels <= 1024:  # hoisted map instead of feature gather + K = 390 GEMM
                    v = ops.sparse_first_conv_from_map(cond, plan, conv1, conv1.out_channels, gn_groups=gn1.num_groups if want_stats else None)
                elif col_bias is None and self.wants_dilated_plan(features.shape[0], plan.n):
                    # one output-stationary implicit GEMM with tap skipping over the dilated voxel list: no 27x intermediate; with the
                    # statistics in its epilogue the output stays COMPACT (rows of the dilated voxels) and the operand split of the
                    # second convolution reads it through the plan's index -- the dense fp32 grid is never written (sparse_conv_os.hip)
                    want_stats = want_stats and ops.sparse_os_gn_ok(conv1.out_channels, gn1.num_groups, r)
                    v = ops.sparse_first_conv_os(features, plan, self._packed_weight(conv1, "fp16x3"), conv1.bias, conv1.out_channels,
                                                 gn_groups=gn1.num_groups if want_stats else None, compact=want_stats)
                else:
                    if rows_in is not None and not (impl == "sparse_h2" and rows_in.plan is plan and rows_in.channels == conv1.in_channels
                                                    and col_bias is None):
                        rows_in = None
                    if (impl == "sparse_h2" and want_stats and folded_tail and plan.n_max <= 256 and col_bias is None
                            and ops.small_grid_gather_ok(r, conv1.out_channels, gn1.num_groups)):
                        # small grid: GroupNorm-1 + Swish + the second convolution's operand split in the gather's epilogue (one
                        # workgroup per (shape, group)): no dense fp32 grid, no statistics hand-off, no to_h2 launch (pvconv_small.hip)
                        xh_ready = ops.sparse_first_conv_planned(features, plan, self._packed_weight(conv1, impl), conv1.bias, conv1.out_channels,
                                                                 rows=rows_in, h2_out=(gn1, ops.h2_activation_scale(gn1),
                                                                                       ops.saturation_slot(self, features.device)))
                        v, want_stats = None, False
                    else:
                        v = ops.sparse_first_conv_planned(features, plan, self._packed_weight(conv1, impl, cin=cin1), conv1.bias, conv1.out_channels,
                                                          gn_groups=gn1.num_groups if want_stats else None, rows=rows_in, col_bias=col_bias)
                if want_stats:
                    v, gn1_stats = v
            else:
                assert col_bias is None
                self._need_features(features)
                norm_coords, vox_coords = ops.voxel_coords(coords, r, self.voxelization.eps)
                x3 = ops.avg_voxelize_s3(features, vox_coords, r)
                v = ops.conv3d_s3(x3, self._packed_weight(conv1, "bf16x6"), conv1.bias, conv1.in_channels,
                                  conv1.out_channels, r)
            if gn1_stats is not None and folded_tail and self.wants_compact_tail(features.shape[0], features.shape[2]):
                # the rest of the branch on voxel lists: operand split on the once-dilated rows, second convolution on the
                # twice-dilated list, SE means and devoxelisation from its rows + the 27 class constants (pvconv_compact.hip)
                sat = ops.saturation_slot(self, features.device)
                rows_h2, const_h2, const_f32, inv_s = ops.to_h2_rows(v, plan, gn1, gn1_stats, saturated=sat, bias=conv1.bias)
                y2, cvals, st2 = ops.second_conv_rows(rows_h2, const_h2, const_f32, inv_s, plan, self._packed_weight(conv2, "fp16x3"),
                                                      self._packed_weight(conv2, "class"), conv2.bias, conv2.in_channels,
                                                      conv2.out_channels, gn2.num_groups)
                w1, w2 = se.fc[0].weight, se.fc[2].weight
                if pf_pending is not None:
                    if pf_ready is not None:
                        tape.wait_event(pf_ready)  # the branch's statistics are read by the SE kernel
                    gate, coef, pf_coef = ops.se_gate_gn_rows(y2, cvals, plan, st2, gn2, w1, w2, pf=pf_pending, n_points=pf.shape[2])
                    return ops.devoxelize_gn_gate_add_rows(norm_coords, y2, cvals, plan, coef, gate=gate, add=pf, add_coef=pf_coef), coords, temb
                gate, coef = ops.se_gate_gn_rows(y2, cvals, plan, st2, gn2, w1, w2)
                if pf_ready is not None:
                    tape.wait_event(pf_ready)
                return ops.devoxelize_gn_gate_add_rows(norm_coords, y2, cvals, plan, coef, gate=gate, add=pf), coords, temb
            # GroupNorm + Swish fused into the operand split of the second conv
            if self.conv_impl == "fp16x3" and not getattr(self, "h2_saturated", False):
                # saturation guard: to_h2 raises this layer's sticky device word when a scaled activation leaves fp16's
                # range; ops.poll_h2_saturation() (once per trajectory) then routes the layer to bf16x6 and warns
                if xh_ready is not None:
                    xh = xh_ready
                else:
                    sat = ops.saturation_slot(self, v.device) if v.is_cuda else None
                    xh = ops.to_h2(v, gn1, swish=True, saturated=sat, stats=gn1_stats)
                cg2, tile = conv2.out_channels // gn2.num_groups, (64 if (conv2.out_channels > 32 and r != 8) else 32)
                if self.fold_gn2 and att is None and se is not None and cg2 in (4, 8, 16, 32) and tile % cg2 == 0:
                    # GroupNorm-folded tail: the convolution leaves the statistics of its output, SE and the devoxelisation
                    # normalise + Swish on the fly -- the grid is written once (by the convolution) and never rewritten
                    v, stats = ops.conv3d_h2_gn(xh, self._packed_weight(conv2, "fp16x3"), conv2.bias, conv2.in_channels,
                                                conv2.out_channels, r, gn2.num_groups)
                    w1, w2 = se.fc[0].weight, se.fc[2].weight
                    head = self._head_for_next(plan, gn2, features.shape[0], features.device) if ops.SMALL_GLUE else None
                    if ops.small_grid_tail_ok(r, conv2.out_channels, features.shape[2]) and w1.shape[0] <= 256 and (head is not None or not ops.SMALL_GLUE_TAIL_ONLY):
                        # small grid: SE's FC layers + GroupNorm-2 + Swish + gate + devoxelisation + point branch in one launch of
                        # per-shape workgroups, which also leave the NEXT PVConv's first-convolution operand when it shares the plan
                        if pf_ready is not None:
                            tape.wait_event(pf_ready)
                        pf_coef = None
                        if pf_pending is not None:
                            mean, coef, pf_coef = ops.se_means_gn(v, stats, gn2, pf=pf_pending, n_points=pf.shape[2])
                        else:
                            mean, coef = ops.se_means_gn(v, stats, gn2)
                        out, rows = ops.pvconv_tail_small(norm_coords, v, coef, mean, w1, w2, r, add=pf, add_coef=pf_coef, head=head)
                        if rows is not None:
                            out._bdm_rows = rows
                        return out, coords, temb
                    if self.se_in_devox and w1.shape[0] <= 64:
                        # SE block's FC layers inside the devoxelisation kernel: one launch less, but every workgroup re-reads
                        # w1 / w2 (measured at B=16: devoxelisation 325 -> 650 us per forward for 100 us of se_fc saved), so
                        # this is only worth it where the forward is launch-bound (PVConv.se_in_devox, off)
                        mean, coef = ops.se_means_gn(v, stats, gn2)
                        if pf_ready is not None:
                            tape.wait_event(pf_ready)
                        return ops.devoxelize_gn_se_add(norm_coords, v, coef, r, mean, w1, w2, add=pf), coords, temb
                    if pf_pending is not None:
                        if pf_ready is not None:
                            tape.wait_event(pf_ready)  # the branch's statistics are read by the SE kernel
                        gate, coef, pf_coef = ops.se_gate_gn(v, stats, gn2, w1, w2, pf=pf_pending, n_points=pf.shape[2])
                        return ops.devoxelize_gn_gate_add(norm_coords, v, coef, r, gate=gate, add=pf, add_coef=pf_coef), coords, temb
                    gate, coef = ops.se_gate_gn(v, stats, gn2, w1, w2)
                    if pf_ready is not None:
                        tape.wait_event(pf_ready)
                    return ops.devoxelize_gn_gate_add(norm_coords, v, coef, r, gate=gate, add=pf), coords, temb
                v = ops.conv3d_h2(xh, self._packed_weight(conv2, "fp16x3"), conv2.bias, conv2.in_channels, conv2.out_channels, r)
            else:
                v = ops.conv3d_s3(ops.to_s3(v, gn1, swish=True), self._packed_weight(conv2, "bf16x6"), conv2.bias,
                                  conv2.in_channels, conv2.out_channels, r)
            ops.group_norm_(v, gn2.weight, gn2.bias, 8, gn2.eps, swish=(att is None))
            if att is not None:
                v = att(v)
            gate = se.gate(v) if se is not None else None
            if pf_ready is not None:
                tape.wait_event(pf_ready)  # the current stream waits for the point branch
            return ops.devoxelize_gate_add(norm_coords, v, r, gate=gate, add=pf), coords, temb
        # the first conv's input is the freshly voxelised cloud: on the 32^3 grids (<= 12.5 % occupied cells) the
        # occupancy-skipping variant wins (measured 1.3-1.4x); on 16^3 / 8^3 the dense kernel is as fast or faster
        sparse = r >= 32
        self._need_features(features)
        vox, norm_coords = self.voxelization(features, coords, with_row_occupancy=sparse)
        rowocc = None
        if sparse:
            vox, rowocc = vox
        v = ops.conv3d(vox, self._packed_weight(conv1, "fp32"), conv1.bias, r, rowocc=rowocc)
        ops.group_norm_(v, gn1.weight, gn1.bias, 8, gn1.eps, swish=True)
        v = ops.conv3d(v, self._packed_weight(conv2, "fp32"), conv2.bias, r)
        ops.group_norm_(v, gn2.weight, gn2.bias, 8, gn2.eps, swish=(att is None))
        if att is not None:
            v = att(v)
        gate = se.gate(v) if se is not None else None
        if pf_ready is not None:
            tape.wait_event(pf_ready)
        fused = ops.devoxelize_gate_add(norm_coords, v, r, gate=gate, add=pf)
        return fused, coords, temb


class BallQuery(nn.Module):
    """ball_query.py:9-35."""

    def __init__(self, radius, num_neighbors, include_coordinates=True):
        super().__init__()
        self.radius = radius
        self.num_neighbors = num_neighbors
        self.include_coordinates = include_coordinates

    def forward(self, points_coords, centers_coords, temb, points_features=None, neighbor_indices=None):
        points_coords = points_coords.contiguous()
        centers_coords = centers_coords.contiguous()
        idx = neighbor_indices
        if idx is None:
            idx = F.ball_query(centers_coords, points_coords, self.radius, self.num_neighbors)
        if points_features is None:
            assert self.include_coordinates, "No Features For Grouping"
            points_features = points_coords[:, :0]
        assert self.include_coordinates
        grouped = ops.sa_group(points_coords, centers_coords, points_features, idx)
        if ops.is_point_invariant(temb):
            # grouping a point-invariant tensor is the identity on its values (exact)
            g_t = temb[:, :, :1, None].expand(-1, -1, idx.shape[1], idx.shape[2])
        else:
            g_t = F.grouping(temb, idx)
        return grouped, g_t


class PointNetSAModule(nn.Module):
    """pointnet.py:49-94 (single-radius form used by the denoisers)."""

    def __init__(self, num_centers, radius, num_neighbors, in_channels, out_channels, include_coordinates=True):
        super().__init__()
        if not isinstance(radius, (list, tuple)):
            radius = [radius]
        if not isinstance(num_neighbors, (list, tuple)):
            num_neighbors = [num_neighbors] * len(radius)
        assert len(radius) == len(num_neighbors)
        if not isinstance(out_channels, (list, tuple)):
            out_channels = [[out_channels]] * len(radius)
        elif not isinstance(out_channels[0], (list, tuple)):
            out_channels = [out_channels] * len(radius)
        assert len(radius) == len(out_channels)
        groupers, mlps = [], []
        total_out_channels = 0
        for _radius, _out_channels, _num_neighbors in zip(radius, out_channels, num_neighbors):
            groupers.append(BallQuery(radius=_radius, num_neighbors=_num_neighbors, include_coordinates=include_coordinates))
            mlps.append(SharedMLP(in_channels=in_channels + (3 if include_coordinates else 0),
                                  out_channels=_out_channels, dim=2))
            total_out_channels += _out_channels[-1]
        self.num_centers = num_centers
        self.out_channels = total_out_channels
        self.groupers = nn.ModuleList(groupers)
        self.mlps = nn.ModuleList(mlps)

    # grouped MLP by recomputation where bdm_sa_mlp2_fused covers it (the first level); the equality tests flip it
    fuse_mlp = os.environ.get("BDM_SA_FUSED", "1") != "0"

    _temb_terms = None   # (B, M) = W[:, t columns] . t of this forward, when pvcnn.encode computed all levels' in one launch

    def temb_rows(self, c_t):
        """(M, c_t): the time embedding's columns (the LAST c_t: [xyz diff, features, t], ball_query.py:31-33 after pvcnn.py:103) of the
        grouped MLP's first layer"""
        w = self.mlps[0].layers[0].weight
        hit = getattr(self, "_temb_rows", None)
        if hit is None or hit[0] != (w._version, w.data_ptr(), c_t):
            w2 = w.detach().reshape(w.shape[0], -1)
            hit = ((w._version, w.data_ptr(), c_t), w2[:, w2.shape[1] - c_t:].contiguous())
            self._temb_rows = hit
        return hit[1]

    def can_split_temb(self, features, temb):
        """May this module take `features` WITHOUT the concatenated time embedding (it then enters the grouped MLP's first layer as a
        per-shape bias: grouping a point-invariant channel is the identity)?  The one predicate of pvcnn.encode and of forward."""
        conv0 = self.mlps[0].layers[0]
        return (PVConv.temb_split and features is not None and features.is_cuda and len(self.groupers) == 1
                and self.groupers[0].include_coordinates and ops.is_point_invariant(temb) and not ops.is_point_invariant(features)
                and 3 + features.shape[1] + temb.shape[1] == conv0.in_channels)

    def plan(self, coords):
        """Geometry-only part of the module (furthest point sampling + ball query): depends on the coordinates
        alone, so the denoiser's encoder runs it for all levels on a side stream while the first PVConvs compute."""
        coords = coords.contiguous()
        centers_coords = self.sample(coords)
        return centers_coords, self.query(coords, centers_coords)

    def sample(self, coords):
        return F.furthest_point_sample(coords, self.num_centers)

    def query(self, coords, centers_coords):
        g = self.groupers[0]
        return F.ball_query(centers_coords, coords, g.radius, g.num_neighbors)

    def forward(self, inputs):
        features, coords, temb = inputs
        coords = coords.contiguous()
        assert len(self.groupers) == 1, "multi-radius grouping is not used by the denoisers"
        planned = getattr(self, "_planned", None)
        self._planned = None
        if planned is not None and callable(planned[0]):  # deferred rest of the sampler chain (pvcnn.plan_sampling_chain)
            planned = planned[0]() if planned[1] is coords else None
            self._planned = None
            if planned is None:
                self._more = None
        # a plan is valid for the very coordinate tensor it was computed from (a forward that aborted midway, or
        # sa_layers shared between networks, must never leave a plan behind for other coordinates)
        if planned is not None and planned[3] is coords:
            centers_coords, idx, event, _ = planned
            tape.wait_event(event)  # the current stream waits for the side stream's sampler
            more, self._more = getattr(self, "_more", None), None
            if more is not None:    # the rest of the sampler chain goes to the side stream AFTER this wait is queued (pvcnn.plan_sampling_chain)
                more()
        else:
            centers_coords, idx = self.plan(coords)
        if (self.fuse_mlp and coords.is_cuda and features is not None and ops.is_point_invariant(temb) and idx.shape[1] <= 8192
                and ops.sa_mlp2_fusable(self.mlps[0], features.shape[1], idx.shape[2])):
            # first level: grouping, both MLP layers and the max in three recompute passes over the packed points -- neither the
            # grouped tensor nor a layer's output is written (sa_mlp_fused.hip)
            out = ops.sa_mlp2_fused(coords, centers_coords.contiguous(), features, idx, self.mlps[0])
            return out, centers_coords, temb[:, :, :1].expand(-1, -1, self.num_centers)
        grouped, g_t = self.groupers[0](coords, centers_coords, temb, features, neighbor_indices=idx)
        conv0 = self.mlps[0].layers[0]
        if grouped.shape[1] != conv0.in_channels:   # the caller (pvcnn.encode) left the time embedding out
            if not self.can_split_temb(features, temb):
                raise ValueError(f"PointNetSAModule's MLP expects {conv0.in_channels} grouped channels, got {grouped.shape[1]}")
            bb, self._temb_terms = self._temb_terms, None
            if bb is None or tuple(bb.shape) != (grouped.shape[0], conv0.out_channels):
                tvec = temb[:, :, 0].contiguous()
                bb = ops.pointwise_conv(tvec[:, :, None], self.temb_rows(temb.shape[1]))[:, :, 0]
            h, pending = self.mlps[0].run(grouped, fold_last=True, first_bias=bb,
                                          first_weight=hoisted_first_weight(self.mlps[0], conv0, grouped.shape[1]))
        else:
            h, pending = self.mlps[0].run(grouped, fold_last=True)
        out = ops.max_over_neighbors(h, fold=pending)
        if g_t.stride(2) == 0 and g_t.stride(3) == 0:
            temb_out = g_t[:, :, 0, 0][:, :, None].expand(-1, -1, self.num_centers)
        else:
            temb_out = ops.max_over_neighbors(g_t.contiguous())
        return out, centers_coords, temb_out


class PointNetFPModule(nn.Module):
    """pointnet.py:96-113."""
    two_source = True  # skip features read in place by the MLP's first convolution
    _cond = None       # ops.Conditioning of this forward when the skip features are the raw conditioning channels
    _temb_bias = None  # (B, M) = W[:, t columns] . t of this forward, when pvcnn.decode has computed all modules' in one launch

    def temb_weight(self, c_feat, c_t):
        """W[:, c_feat : c_feat + c_t] of the first MLP layer (contiguous, cached per weight version): the time embedding's columns"""
        w = self.mlp.layers[0].weight
        hit = getattr(self, "_wt", None)
        if hit is None or hit[0] != (w._version, w.data_ptr(), c_feat, c_t):
            hit = ((w._version, w.data_ptr(), c_feat, c_t), w.detach().reshape(w.shape[0], -1)[:, c_feat:c_feat + c_t].contiguous())
            self._wt = hit
        return hit[1]

    def _need_skip(self, points_features):
        """The skip rows are about to be READ (no hoisted map took their place): complete a lazily conditioned input first."""
        cond = self._cond
        if (cond is not None and not cond.features_ready and points_features is not None
                and points_features.data_ptr() == cond.x_cf[:, 3:].data_ptr()):
            cond.ensure_features()

    def can_split(self, centers_features, points_features, temb):
        """May this module take the time embedding as a per-shape bias (centers_features arriving WITHOUT the embedding)?  The ONE
        predicate of pvcnn.decode (which then does not concatenate) and of forward (which then does not expect the concatenation)."""
        conv0 = self.mlp.layers[0]
        c_skip = 0 if points_features is None else points_features.shape[1]
        return (centers_features.is_cuda and ops.is_point_invariant(temb) and not ops.is_point_invariant(centers_features)
                and centers_features.shape[1] + temb.shape[1] + c_skip == conv0.in_channels)

    def _forward_split(self, pc, cc, idx, w, features, points_features, temb, points_coords):
        from . import _lib as L
        B, _, n = pc.shape
        m, dev = cc.shape[2], pc.device
        conv0 = self.mlp.layers[0]
        c_feat, c_t = features.shape[1], temb.shape[1]
        c_skip = 0 if points_features is None else points_features.shape[1]
        bb, self._temb_bias = self._temb_bias, None
        if bb is None or tuple(bb.shape) != (B, conv0.out_channels):
            tvec = temb[:, :, 0].contiguous()
            bb = ops.pointwise_conv(tvec[:, :, None], self.temb_weight(c_feat, c_t))[:, :, 0]
        buf = torch.empty(B, c_feat, n, dtype=torch.float32, device=dev)
        fa, _, _, _, bs_a, ld_a = ops._bcl(features)
        _, _, _, _, bs_0, ld_0 = ops._bcl(buf)
        L.check(L.lib().bdm_fp_assemble(B, m, n, L.ptr(idx), L.ptr(w), c_feat, L.ptr(fa), L.c_ll(bs_a), ld_a, 0, None, L.c_ll(0), 0,
                                        0, None, L.c_ll(0), 0, L.ptr(buf), L.c_ll(bs_0), ld_0, None, L.c_ll(0), 0, L.stream()), "fp_assemble")
        temb = temb[:, :, :1].expand(-1, -1, n)   # the reference's interpolated_temb: (B, c_t, n) (pointnet.py:108)
        if c_skip == 0:
            return self.mlp.run(buf, first_weight=hoisted_first_weight(self.mlp, conv0, c_feat), first_bias=bb), points_coords, temb
        cond = self._cond
        if cond is not None and points_features.shape[1] == cond.C and points_features.data_ptr() == cond.x_cf[:, 3:].data_ptr():
            # the skip channels are F[pix]: their share of the first layer, (F . W_skip^T)[pix], is gathered from the hoisted map
            fmap = cond.map("fp_skip", conv0.weight,
                            lambda conv=conv0, lo=c_feat + c_t, C=cond.C: conv.weight.detach().reshape(conv.out_channels, -1)[:, lo:lo + C])
            g = cond.gather(fmap)
            return (self.mlp.run(buf, first_weight=hoisted_first_weight(self.mlp, conv0, c_feat), first_add=g[:, 3:], first_bias=bb),
                    points_coords, temb)
        self._need_skip(points_features)
        keep = ((0, c_feat), (c_feat + c_t, c_feat + c_t + c_skip))
        return self.mlp.run(buf, x2=points_features, first_weight=hoisted_first_weight(self.mlp, conv0, keep), first_bias=bb), points_coords, temb

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.mlp = SharedMLP(in_channels=in_channels, out_channels=out_channels, dim=1)

    def forward(self, inputs):
        if len(inputs) == 3:
            points_coords, centers_coords, centers_features, temb = inputs
            points_features = None
        else:
            points_coords, centers_coords, centers_features, points_features, temb = inputs
        from . import _lib as L
        pc, cc = points_coords.contiguous(), centers_coords.contiguous()
        B, _, n = pc.shape
        m = cc.shape[2]
        dev = pc.device
        # one search serves both tensors (the reference searches twice, pointnet.py:107-108: same result); when the sampler chain
        # has already run it on its own stream for this very pair of coordinate tensors (pvcnn.plan_sampling_chain), wait for that
        from . import pvcnn
        planned = pvcnn.NN_PLANS.pop((pc.data_ptr(), cc.data_ptr()), None)
        if planned is not None and planned[0].shape == pc.shape and planned[1].shape == cc.shape:
            idx, w = planned[2], planned[3]
            tape.wait_event(planned[4])
        else:
            idx, w = ops.three_nn_search(pc, cc)

        # SPLIT form (pvcnn.decode): centers_features arrive WITHOUT the time embedding, which is constant along a shape's points
        # (t_emb[:, :, None].expand): its interpolation is itself (the weights sum to one) and its share of the first MLP layer is a
        # per-shape bias W[:, t columns] . t -- no concatenation, 64 channels less to interpolate (twice: the reference interpolates
        # cat([features, t_emb]) AND t_emb) and 64 columns less in the first GEMM
        c_t = temb.shape[1]
        if self.can_split(centers_features, points_features, temb):
            return self._forward_split(pc, cc, idx, w, centers_features, points_features, temb, points_coords)
        cf = ops.materialize(centers_features)
        c_int = cf.shape[1]
        c_skip = 0 if points_features is None else points_features.shape[1]
        two_source = self.two_source and c_skip > 0 and dev.type == "cuda"
        if two_source:  # the skip rows stay where they are: the MLP's first convolution reads cat([interpolated, skip]) in place
            skip_src, c_skip = points_features, 0
        buf = torch.empty(B, c_int + c_skip, n, dtype=torch.float32, device=dev)
        t_src = ops.materialize(temb)
        interpolated_temb = torch.empty(B, t_src.shape[1], n, dtype=torch.float32, device=dev)
        # interpolated features, skip features and interpolated t_emb written by ONE launch
        fa, _, _, _, bs_a, ld_a = ops._bcl(cf)
        ft, _, c_t, _, bs_t, ld_t = ops._bcl(t_src)
        if c_skip:
            self._need_skip(points_features)
            fs, _, _, _, bs_s, ld_s = ops._bcl(points_features)
        else:
            fs, bs_s, ld_s = None, 0, 0
        _, _, _, _, bs_0, ld_0 = ops._bcl(buf)
        _, _, _, _, bs_1, ld_1 = ops._bcl(interpolated_temb)
        L.check(L.lib().bdm_fp_assemble(B, m, n, L.ptr(idx), L.ptr(w), c_int, L.ptr(fa), L.c_ll(bs_a), ld_a, c_skip, L.ptr(fs),
                                        L.c_ll(bs_s), ld_s, c_t, L.ptr(ft), L.c_ll(bs_t), ld_t, L.ptr(buf), L.c_ll(bs_0), ld_0,
                                        L.ptr(interpolated_temb), L.c_ll(bs_1), ld_1, L.stream()), "fp_assemble")
        if two_source:
            cond = self._cond
            if (cond is not None and skip_src.shape[1] == cond.C and skip_src.data_ptr() == cond.x_cf[:, 3:].data_ptr()):
                # the skip channels are F[pix]: their share of the first layer, (F . W_skip^T)[pix], is gathered from the hoisted map
                conv = self.mlp.layers[0]
                fmap = cond.map("fp_skip", conv.weight,
                                lambda conv=conv, lo=c_int, C=cond.C: conv.weight.detach().reshape(conv.out_channels, -1)[:, lo:lo + C])
                g = cond.gather(fmap)
                return self.mlp.run(buf, first_weight=hoisted_first_weight(self.mlp, conv, c_int), first_add=g[:, 3:]), points_coords, interpolated_temb
            self._need_skip(skip_src)
            return self.mlp.run(buf, x2=skip_src), points_coords, interpolated_temb
        return self.mlp.run(buf), points_coords, interpolated_temb
