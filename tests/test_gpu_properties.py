"""The portable properties the reference's own native test asserts about this path
(src/chain-supervision-test.hpp), checked on the HIP path itself -- no oracle
involved: finite differences of the objective against the derivative the kernels return, posterior sums,
the shift property, objf <= 0.  Thresholds are the reference's."""
import numpy as np
import pytest

from torchain_amd import synth

from helpers import hip_chain, hip_den, hip_num

pytestmark = pytest.mark.gpu


def approx_equal_vec(a, b, tol):
    """[K] VectorBase::ApproxEqual: ||a - b|| <= tol * ||a||."""
    return np.linalg.norm(a - b) <= tol * np.linalg.norm(a)


GRAPHS = {
    "tied": lambda: synth.random_den_fst(60, 4, 35, seed=13),
    "general": lambda: synth.skewed_den_fst(80, 900, 40, seed=5),
    "tied_hubs": lambda: synth.skewed_tied_den_fst(300, 5000, 90, seed=6),
}


@pytest.mark.parametrize("kind", sorted(GRAPHS))
@pytest.mark.parametrize("zero", [False, True])
def test_chain_denominator_test(kind, zero):
    """ChainDenominatorTest (chain-supervision-test.hpp:388-463): default opts (leaky 1e-5), Backward(1.0),
    sum(deriv) - S*T < 10, finite differences within 0.25 (T < 50)."""
    fst = GRAPHS[kind]()
    rng = np.random.default_rng(0)
    S, T = 4, 17
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=3, zero=zero)
    base = hip_den(fst, y, S, leaky=1e-5, deriv_weight=1.0)
    assert base["status"] == 0
    assert base["deriv"].sum() - S * T < 10.0
    np.testing.assert_allclose(base["deriv"].sum(axis=1, dtype=np.float64), 1.0, atol=1e-4)
    pred, obs = np.zeros(5), np.zeros(5)
    for k in range(5):
        delta = (rng.standard_normal(y.shape) * 1e-4).astype(np.float32)
        pred[k] = float((base["deriv"].astype(np.float64) * delta).sum())
        obs[k] = hip_den(fst, y + delta, S, leaky=1e-5, want_deriv=False, graph=base["graph"])["logprob"] - base["logprob"]
    assert approx_equal_vec(pred, obs, 0.25)


@pytest.mark.parametrize("kind", sorted(GRAPHS))
@pytest.mark.parametrize("leaky", [1e-5, 0.2])
def test_chain_training_test(oracle, kind, leaky):
    """ChainTrainingTest (chain-supervision-test.hpp:239-341): ||row sums of deriv|| < 0.1, sum(deriv) < 0.2,
    objf <= 0 when the numerator carries the denominator's weights, finite differences with mean
    correction within 0.25.  (The oracle fixture is used only for the graph's initial probs that weight
    the synthetic numerator's first arcs.)"""
    fst = GRAPHS[kind]()
    rng = np.random.default_rng(1)
    S, T = 3, 15
    pi = oracle.DenGraph(fst).initial_probs()
    sup = synth.random_supervision(fst, S, T, 3, seed=4, weight=1.0, initial_probs=pi)
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=5)
    out = hip_chain(fst, sup, y, l2=0.0, leaky=leaky)
    objf, weight = out["results"][0], out["results"][2]
    assert np.linalg.norm(out["deriv"].sum(axis=1, dtype=np.float64)) < 0.1
    assert out["deriv"].sum(dtype=np.float64) < 0.2
    assert objf <= 0.0 and weight == S * T
    pred, obs = np.zeros(5), np.zeros(5)
    for k in range(5):
        delta = (rng.standard_normal(y.shape) * 1e-4).astype(np.float32)
        pred[k] = float((out["deriv"].astype(np.float64) * delta).sum())
        obs[k] = hip_chain(fst, sup, y + delta, l2=0.0, leaky=leaky, want_deriv=False, graph=out["graph"])["results"][0] - objf
    obs = obs + (pred.sum() - obs.sum()) / len(pred)
    if np.linalg.norm(pred) > 0.1 * 1e-4:
        assert approx_equal_vec(pred, obs, 0.25)


def test_supervision_numerator(oracle):
    """TestSupervisionNumerator (chain-supervision-test.hpp:92-152): finite differences within 0.1 and the
    shift property: adding r[row] to every column of a row changes Forward() by sum(r)."""
    fst = GRAPHS["tied"]()
    rng = np.random.default_rng(2)
    S, T = 3, 12
    sup = synth.random_supervision(fst, S, T, 3, seed=6, initial_probs=oracle.DenGraph(fst).initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=7)
    base = hip_num(sup, y)
    np.testing.assert_allclose(base["deriv"].sum(axis=1), sup.weight, atol=1e-5)
    pred, obs = np.zeros(3), np.zeros(3)
    for k in range(3):
        delta = (rng.standard_normal(y.shape) * 1e-4).astype(np.float32)
        pred[k] = float((base["deriv"].astype(np.float64) * delta).sum())
        obs[k] = hip_num(sup, y + delta, want_deriv=False)["logprob_weighted"] - base["logprob_weighted"]
    obs = obs + (pred.sum() - obs.sum()) / 3
    assert approx_equal_vec(pred, obs, 0.1)
    r = rng.standard_normal(S * T).astype(np.float32)
    mod = hip_num(sup, y + r[:, None], want_deriv=False)
    assert abs(float(r.sum()) - (mod["logprob_weighted"] - base["logprob_weighted"])) < 0.1
