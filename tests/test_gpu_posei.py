"""GPU parity of osh_posei_optimize (Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame, src/Optimizer.cc:4499-5299) against
the CPU oracle on the same frames: state, outlier decisions, counts and the Hessian handed to ConstraintPoseImu."""
import numpy as np
import pytest

from orb_slam3_study_kr_amd import lba
from orb_slam3_study_kr_amd import synth_inertial as si

from helpers import POSEI_FIXTURES, load_posei_fixture
from test_oracle_posei import check_against_posei_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(hip_lib):
    s = lba.LbaSolver(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _check(got, ref, f, tol=1e-7, chi_tol=1e-5):
    assert (got.rounds, got.status) == (ref.rounds, 0)
    # float32 sinf / cosf of the preintegration getters differ by an ulp between device and host libm (see tests/test_gpu_liba.py)
    assert np.abs(got.Rwb - ref.Rwb).max() < tol and np.abs(got.twb - ref.twb).max() < tol * max(1.0, np.abs(ref.twb).max())
    assert np.abs(got.Rcw - ref.Rcw).max() < tol and np.abs(got.tcw - ref.tcw).max() < tol * max(1.0, np.abs(ref.tcw).max())
    np.testing.assert_allclose(got.vel, ref.vel, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got.bias_g, ref.bias_g, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(got.bias_a, ref.bias_a, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got.edge_chi2, ref.edge_chi2, rtol=chi_tol, atol=chi_tol)
    # identical decisions except within rounding of a threshold
    thr = np.where(f.edge_kind == 1, f.chi2_stereo[3], np.where(f.edge_close == 1, 1.5 * f.chi2_mono[3], f.chi2_mono[3]))
    near = np.abs(ref.edge_chi2 - thr) < 10 * chi_tol * thr
    assert np.array_equal(got.outlier[~near], ref.outlier[~near])
    if not near.any():
        assert (got.n_bad, got.n_inliers) == (ref.n_bad, ref.n_inliers)
    np.testing.assert_allclose(got.H, ref.H, rtol=1e-6, atol=1e-8 * np.abs(ref.H).max())


@pytest.mark.parametrize("mode", [0, 1])
def test_stereo_frames_both_variants(solver, ob, mode):
    frames = [si.make_posei_frame(10 + k, mode=mode, n_points=300 + 100 * k) for k in range(3)]
    got = solver.optimize_poses_inertial(frames)
    for g, f in zip(got, frames):
        _check(g, ob.posei_optimize(f), f)


def test_mixed_batch_mono_fisheye_rig_and_small_frames(solver, ob):
    frames = [si.make_posei_frame(20, mode=0, stereo=False, n_points=250), si.make_posei_frame(21, mode=1, fisheye=True, n_points=250),
              si.make_posei_frame(22, mode=1, rig=True, n_points=300), si.make_posei_frame(23, mode=0, rig=True, n_points=300),
              si.make_posei_frame(24, mode=0, n_points=25, outlier_frac=0.3),                    # < 30 inliers: recovery pass
              si.make_posei_frame(25, mode=0, n_points=25, outlier_frac=0.3, rec_init=True),
              si.make_posei_frame(26, mode=0, n_points=5, outlier_frac=0.0)]                     # < 10 graph edges: one round
    got = solver.optimize_poses_inertial(frames)
    for g, f in zip(got, frames):
        # fisheye residuals are a float32 staircase (tests/test_gpu_liba.py): chi2 of an edge agrees to 1e-3, the state to 1e-6
        fish = f.kb8 is not None
        _check(g, ob.posei_optimize(f), f, tol=2e-6 if fish else 1e-7, chi_tol=2e-3 if fish else 1e-5)
    assert got[6].rounds == 1
    again = solver.optimize_poses_inertial(frames)
    for g, a in zip(got, again):
        np.testing.assert_array_equal(g.twb, a.twb)      # fixed reduction orders: bitwise reproducible
        np.testing.assert_array_equal(g.H, a.H)


@pytest.mark.parametrize("name", POSEI_FIXTURES)
def test_device_matches_the_numpy_models_golden_outputs(solver, name):
    """The committed fixtures of tests/golden/make_golden.py (independent numpy model of the whole function): no oracle in the loop."""
    f, z = load_posei_fixture(name)
    fish = f.kb8 is not None
    check_against_posei_fixture(solver.optimize_poses_inertial([f])[0], z, f, state_tol=2e-6 if fish else 1e-7, chi_tol=2e-3 if fish else 2e-5)
