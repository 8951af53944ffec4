"""Per-shape random streams of the sampler (SURVEY.md 8e "Partitioning"; kernels in csrc/rng_ops.hip).

Every shape owns a Philox4x32-10 stream keyed by (run seed, GLOBAL shape index); a draw is addressed by
(purpose, draw index, element), so a shape's initial cloud, DDPM / PVD noise and blend masks do not depend on the rank
count, the batch size or the position of the shape in its batch.  The reference instead seeds one generator per process
with seed + rank (experiments/training_utils.py:373-378) and draws at main_blending.py:228,330-338, model/model.py:286 and
pvd/__init__.py:213,232 -- kept available as the default "reference" mode of the samplers (global torch generators).
"""
import torch

from . import _lib as L

MASK64 = (1 << 64) - 1
# purposes (word 3 of the Philox counter): independent sub-streams of one shape
INIT, PC2, PVD, MASK, FUSE = 0, 1, 2, 3, 4


def shape_key(seed: int, shape_index: int) -> int:
    """64-bit Philox key of a shape: splitmix64 finaliser over (seed, global shape index)."""
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(shape_index) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


class ShapeStreams:
    """Noise source of one batch: `shape_indices` are the GLOBAL indices of the batch's shapes, in batch order."""

    def __init__(self, seed, shape_indices, device):
        self.seed, self.shape_indices = int(seed), [int(i) for i in shape_indices]
        keys = [shape_key(seed, i) for i in self.shape_indices]
        signed = [k - (1 << 64) if k >= (1 << 63) else k for k in keys]  # same 64 bits in an int64 tensor
        self.keys = torch.tensor(signed, dtype=torch.int64).to(device)
        self.device = self.keys.device
        self.counters = {}

    def _next(self, purpose):
        d = self.counters.get(purpose, 0)
        self.counters[purpose] = d + 1
        return d

    def _check(self, shape):
        if shape[0] != len(self.shape_indices):
            raise ValueError(f"draw of {tuple(shape)} for a stream set of {len(self.shape_indices)} shapes")
        per = 1
        for s in shape[1:]:
            per *= int(s)
        return per

    def normal(self, shape, purpose):
        """(B, ...) float32 standard normals; row b comes from shape b's stream."""
        per = self._check(shape)
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        L.check(L.lib().bdm_philox_normal(shape[0], per, L.ptr(self.keys), self._next(purpose), purpose, L.ptr(out), L.stream()),
                "philox_normal")
        return out

    def bits(self, shape, purpose=MASK):
        """(B, ...) int64 in {0, 1}: Bernoulli(1/2), the blend masks."""
        per = self._check(shape)
        out = torch.empty(tuple(shape), dtype=torch.int64, device=self.device)
        L.check(L.lib().bdm_philox_bits(shape[0], per, L.ptr(self.keys), self._next(purpose), purpose, L.ptr(out), L.stream()),
                "philox_bits")
        return out

    # fused scheduler steps: the noise never touches memory -------------------------------------------------------
    def ddpm_step(self, x, eps, c, purpose=PC2, out=None):
        """DDPMScheduler.step arithmetic (schedulers.py) with this batch's noise generated in the kernel.  out: where the result goes
        (may be `x` itself: the kernel is elementwise)."""
        x, eps = x.contiguous(), eps.contiguous()
        per = self._check(x.shape)
        if out is None:
            out = torch.empty_like(x)
        if not (out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype and out.device == x.device):
            raise ValueError("ShapeStreams.ddpm_step: `out` must be a contiguous tensor of x's shape, dtype and device")
        sigma = c["sigma"] if c.get("noise", True) else 0.0
        draw = self._next(purpose) if sigma != 0.0 else 0
        L.check(L.lib().bdm_ddpm_step_philox(x.shape[0], per, L.ptr(x), L.ptr(eps), L.ptr(self.keys), draw, purpose,
                                             c["sqrt_beta_prod"], c["sqrt_alpha_prod"], c["coef_x0"], c["coef_x"], sigma,
                                             L.ptr(out), L.stream()), "ddpm_step_philox")
        return out

    def pvd_step(self, x, eps, c, purpose=PVD, out=None):
        """out: may be `x` itself (contiguous): the step is elementwise (pvd.Model's replayed loop steps in place)"""
        x, eps = x.contiguous(), eps.contiguous()
        per = self._check(x.shape)
        if out is None:
            out = torch.empty_like(x)
        assert out.is_contiguous() and out.shape == x.shape
        draw = self._next(purpose)  # the reference draws at t == 0 too (pvd/__init__.py:213): the index advances
        L.check(L.lib().bdm_pvd_step_philox(x.shape[0], per, L.ptr(x), L.ptr(eps), L.ptr(self.keys), draw, purpose,
                                            c["a"], c["b"], c["c1"], c["c2"], c["sigma"], L.ptr(out), L.stream()),
                "pvd_step_philox")
        return out
