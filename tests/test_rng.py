"""Per-shape Philox streams: the oracle restatement against the published Random123 known-answer vectors (CPU), the HIP
kernels against the oracle (GPU, bit-exact integers), and the fused scheduler steps against the two-launch form."""
import numpy as np
import pytest
import torch


def test_philox_known_answer_vectors():
    """Random123 kat_vectors, philox4x32-10."""
    from oracle.ref_rng import philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want


def test_key_derivation_matches_product_and_separates_shapes():
    from bdm_amd.rng import shape_key
    from oracle import ref_rng
    keys = {shape_key(42, i) for i in range(4096)} | {shape_key(43, i) for i in range(4096)}
    assert len(keys) == 8192
    for s, i in [(0, 0), (42, 7), (2 ** 31, 12345), (7, 2 ** 40)]:
        assert shape_key(s, i) == ref_rng.shape_key(s, i)


def test_oracle_normals_are_standard():
    from oracle import ref_rng
    z = ref_rng.normal(ref_rng.shape_key(1, 2), 200_000, 3, 1)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3.0) < 0.06
    b = ref_rng.bits(ref_rng.shape_key(1, 2), 200_000, 0, 3)
    assert abs(b.mean() - 0.5) < 0.005


@pytest.mark.gpu
def test_hip_streams_match_the_oracle(hip):
    from bdm_amd import rng
    from oracle import ref_rng
    idx = [5, 0, 1000003]
    st = rng.ShapeStreams(42, idx, "cuda")
    for per_shape, shape in [(3 * 4096, (3, 4096, 3)), (3 * 1023 + 0, (3, 3, 1023)), (5, (3, 5))]:
        st.counters.clear()
        a = st.normal(shape, rng.PC2).cpu().numpy().reshape(3, -1)
        b = st.normal(shape, rng.PC2).cpu().numpy().reshape(3, -1)   # draw 1
        m = st.bits(shape, rng.MASK).cpu().numpy().reshape(3, -1)
        for r, gi in enumerate(idx):
            key = ref_rng.shape_key(42, gi)
            assert np.abs(a[r] - ref_rng.normal(key, per_shape, 0, rng.PC2)).max() < 2e-5
            assert np.abs(b[r] - ref_rng.normal(key, per_shape, 1, rng.PC2)).max() < 2e-5
            assert np.array_equal(m[r], ref_rng.bits(key, per_shape, 0, rng.MASK))
    # a shape's draws do not depend on its batch: same shape index in another batch, other slot
    st.counters.clear()
    full = st.normal((3, 4096, 3), rng.PC2)
    other = rng.ShapeStreams(42, [9, 1000003], "cuda").normal((2, 4096, 3), rng.PC2)
    assert torch.equal(other[1], full[2]) and not torch.equal(other[0], full[0])


@pytest.mark.gpu
def test_fused_steps_equal_two_launch_form(hip):
    from bdm_amd import rng
    from bdm_amd.pvd import GaussianDiffusion, get_betas
    from bdm_amd.schedulers import DDPMScheduler
    from helpers import seeded
    B, N = 3, 1001
    x, eps = seeded((B, N, 3), 1).cuda(), seeded((B, N, 3), 2).cuda()
    s = DDPMScheduler(beta_start=1e-5, beta_end=8e-3, clip_sample=False)
    s.set_timesteps(1000)
    for t in (999, 1, 0):
        a, b = rng.ShapeStreams(7, [4, 5, 6], "cuda"), rng.ShapeStreams(7, [4, 5, 6], "cuda")
        fused = a.ddpm_step(x, eps, dict(s.step_coefficients(t), noise=t > 0))
        s.noise_source = lambda shape, dev: b.normal(shape, rng.PC2)
        two = s.step(eps, t, x).prev_sample
        assert torch.equal(fused, two), t
        assert a.counters == b.counters
    gd = GaussianDiffusion(get_betas("linear", 0.0001, 0.02, 1000), "mse", "eps", "fixedsmall")
    xp, ep = x.permute(0, 2, 1).contiguous(), eps.permute(0, 2, 1).contiguous()
    for t in (999, 0):
        a, b = rng.ShapeStreams(7, [4, 5, 6], "cuda"), rng.ShapeStreams(7, [4, 5, 6], "cuda")
        fused = a.pvd_step(xp, ep, gd.step_coefficients(t))
        gd.noise_source = lambda shape, dev: b.normal(shape, rng.PVD)
        two = gd.p_sample(lambda d, t_: ep, xp, torch.full((B,), t, device="cuda"), t_int=t)
        assert torch.equal(fused, two), t
