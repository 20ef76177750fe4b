"""The committed golden vectors (tests/golden/*.npz: inputs + float64 outputs of the independent formulation)
through the HIP path itself -- tc_chain_objf_and_deriv, tc_den_forward_backward, tc_num_forward_backward over the
C ABI.  Round 1 only checked them against the oracle on the CPU."""
import os

import numpy as np
import pytest

from helpers import hip_chain, hip_den, hip_num, rel_err
from fixtures import GOLDEN, load_golden as load

pytestmark = pytest.mark.gpu
REL = 1e-4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(path):
    z, fst, sup = load(path)
    y = np.ascontiguousarray(z["nnet_output"], np.float32)
    out = hip_chain(fst, sup, y, l2=float(z["l2_regularize"]), leaky=float(z["leaky"]), xent=True)
    res = out["results"]
    np.testing.assert_allclose(out["graph"].initial_probs(), z["initial_probs"], rtol=1e-5, atol=1e-9)
    assert abs(res[0] - float(z["objf"])) <= REL * abs(float(z["objf"]))
    assert abs(res[1] - float(z["l2_term"])) <= REL * abs(float(z["l2_term"])) + 1e-12
    assert res[2] == float(z["weight"])
    assert rel_err(out["deriv"], z["deriv"], floor=sup.weight) <= REL
    assert rel_err(out["xent_deriv"], z["xent_deriv"], floor=sup.weight) <= REL
    frames = sup.num_sequences * sup.frames_per_sequence
    den = hip_den(fst, y, sup.num_sequences, leaky=float(z["leaky"]), deriv_weight=1.0, graph=out["graph"])
    assert den["status"] == 0
    assert abs(den["logprob"] - float(z["den_logprob"])) <= REL * max(abs(float(z["den_logprob"])), frames)
    assert np.abs(den["deriv"] - z["den_deriv"]).max() <= REL
    num = hip_num(sup, y)
    assert abs(num["logprob_weighted"] - sup.weight * float(z["num_logprob"])) <= REL * max(abs(float(z["num_logprob"])), frames)
