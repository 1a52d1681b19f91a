"""Data preparation: coarse-graining and z-score scaling.  These are the only functions of the path's
neighbourhood the reference's own test-suite covers, so the CPU tests below restate ITS checks — test/test_coarse_graining.jl
(linear profile: constant spacing after coarse-graining, mean preserved; lengths) and test/test_feature_scaling.jl
(μ, σ equal mean/std; scaled data has mean 0, std 1; inv∘scale = identity) — against the oracle, and the GPU tests run the
same checks and an oracle comparison through the C ABI."""
import numpy as np
import pytest

from oracle import nde_oracle as orc
from colnde import synthetic

RNG = np.random.default_rng(20261004)


def linear(x, m, c): return m * x + c
def quadratic(x, a, b, c): return a * x ** 2 + b * x + c


# ------------------------------------------------------------------ the reference's tests, on the oracle (CPU)
def test_reference_coarse_graining_center_checks():               # test/test_coarse_graining.jl:5-17
    y = linear(np.arange(1, 101, dtype=np.float64), 0.5, -3)
    yc = orc.coarse_grain_center(y, 20)
    assert yc.shape == (20,)
    assert np.allclose(np.diff(yc), np.diff(yc)[0])
    assert np.isclose(y.mean(), yc.mean())


def test_reference_coarse_graining_face_checks():                 # test/test_coarse_graining.jl:19-38
    y = linear(np.arange(1, 101, dtype=np.float64), 0.5, -3)
    yq = quadratic(np.arange(1, 101, dtype=np.float64), 0.5, -3, 5)
    yc = orc.coarse_grain_linear_interpolation_face(y, 20)
    yqc = orc.coarse_grain_linear_interpolation_face(yq, 20)
    assert yc.shape == (20,) and yqc.shape == (20,)
    assert np.allclose(np.diff(yc), np.diff(yc)[0])
    assert np.isclose(y.mean(), yc.mean())
    assert yc[0] == y[0] and yc[-1] == y[-1]


@pytest.mark.parametrize("shape", [(10,), (5, 5), (3, 3, 3)])     # test/test_feature_scaling.jl:1-12
def test_reference_zero_mean_unit_variance_checks(shape):
    data = RNG.random(shape)
    scaled, mu, sigma = orc.zero_mean_unit_variance(data)
    assert mu == data.mean() and sigma == data.std(ddof=1)
    assert abs(scaled.mean()) < 1e-10
    assert np.isclose(scaled.std(ddof=1), 1.0)
    assert np.allclose(sigma * scaled + mu, data)


def test_data_container_shapes_128_to_32_and_129_to_33():         # data_containers.jl:343-372: gap = 4 -> every 4th face
    u = RNG.standard_normal((7, 128))
    w = RNG.standard_normal((7, 129))
    assert np.allclose(orc.coarse_grain_center(u, 32), u.reshape(7, 32, 4).mean(-1))
    assert np.allclose(orc.coarse_grain_linear_interpolation_face(w, 33), w[:, ::4])


# ------------------------------------------------------------------ HIP kernels through the C ABI
@pytest.mark.gpu
def test_gpu_coarse_grain_matches_oracle_and_reference_checks():
    import torch
    import colnde
    cfg = synthetic.wind_mixing_problem(1, n_frames=2).cfg
    with colnde.ColumnNDE(cfg, 1) as nde:
        for rows, N, n in [(1153, 128, 32), (3, 100, 20), (1, 64, 64)]:
            x = RNG.standard_normal((rows, N)).astype(np.float32)
            got = nde.coarse_grain(torch.from_numpy(x).cuda(), n, "center").cpu().numpy()
            np.testing.assert_allclose(got, orc.coarse_grain_center(x, n), rtol=2e-6, atol=2e-6)
        for rows, N, n in [(1153, 129, 33), (3, 100, 20), (2, 50, 7)]:
            x = RNG.standard_normal((rows, N)).astype(np.float32)
            got = nde.coarse_grain(torch.from_numpy(x).cuda(), n, "face").cpu().numpy()
            np.testing.assert_allclose(got, orc.coarse_grain_linear_interpolation_face(x, n), rtol=2e-6, atol=2e-6)
            assert np.array_equal(got[:, 0], x[:, 0]) and np.array_equal(got[:, -1], x[:, -1])
        # the reference's own checks, on the device results
        y = linear(np.arange(1, 101, dtype=np.float32), 0.5, -3)[None, :]
        for loc in ("center", "face"):
            yc = nde.coarse_grain(torch.from_numpy(y).cuda(), 20, loc).cpu().numpy()[0]
            assert yc.shape == (20,)
            assert np.allclose(np.diff(yc), np.diff(yc)[0], rtol=1e-5)
            assert np.isclose(y.mean(), yc.mean(), rtol=1e-6)
        with pytest.raises(colnde.ColndeError):
            nde.coarse_grain(torch.zeros(2, 100, device="cuda"), 30, "center")       # 30 does not divide 100


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(10,), (5, 5), (3, 3, 3), (96, 1153)])
def test_gpu_zscore_matches_oracle_and_reference_checks(shape):
    import torch
    import colnde
    cfg = synthetic.wind_mixing_problem(1, n_frames=2).cfg
    data = (RNG.random(shape) * 3.0 + 17.0).astype(np.float32)
    want, mu, sigma = orc.zero_mean_unit_variance(data)
    with colnde.ColumnNDE(cfg, 1) as nde:
        scaled, ms = nde.zscore(torch.from_numpy(data).cuda())
    scaled, ms = scaled.cpu().numpy(), ms.cpu().numpy()
    assert np.isclose(ms[0], mu, rtol=1e-6) and np.isclose(ms[1], sigma, rtol=1e-6)
    np.testing.assert_allclose(scaled, want, rtol=1e-4, atol=2e-5)
    assert abs(scaled.astype(np.float64).mean()) < 1e-5                  # float32 data: 1e-10 of the float64 reference test relaxed
    assert np.isclose(scaled.astype(np.float64).std(ddof=1), 1.0, rtol=1e-5)
    assert np.allclose(ms[1] * scaled + ms[0], data, rtol=1e-6)
