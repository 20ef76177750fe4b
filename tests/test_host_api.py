"""Host-side mirror of the reference's Python interface (torchain/functions.py, torchain/io.py):
names, signatures, defaults and the ChainResults arithmetic, checked without a GPU."""
import inspect

import pytest
import torch

import torchain_amd
from torchain_amd import functions, io, parallel


def test_chain_loss_signature_matches_reference():
    """torchain/functions.py:128-130."""
    sig = inspect.signature(functions.chain_loss)
    assert list(sig.parameters) == ["input", "den_graph", "supervision", "l2_regularize",
                                    "leaky_hmm_coefficient", "xent_regularize", "xent_input", "kaldi_way"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == dict(l2_regularize=0.0, leaky_hmm_coefficient=1e-5, xent_regularize=0.0, xent_input=None,
                     kaldi_way=False)


def test_chain_results_semantics():
    """functions.py:9-19: CPU float[3], loss = -objf/weight (l2 not included), repr format."""
    r = functions.ChainResults()
    assert r.data.shape == (3,) and r.data.dtype == torch.float32 and not r.data.is_cuda
    r.data[:] = torch.tensor([-120.0, -3.5, 48.0])
    assert float(r.loss) == pytest.approx(2.5)
    assert repr(r) == "ChainResults(loss=2.500000, objf=-120.000000, l2_term=-3.500000, weight=48.000000)"
    # callers sum .data across steps (train.py:145)
    acc = functions.ChainResults()
    acc.data += r.data
    acc.data += r.data
    assert float(acc.loss) == pytest.approx(2.5)


def test_to2d_is_frame_major():
    """functions.py:118-125: (B, C, T) -> rows t*B + b."""
    B, Cc, T = 3, 5, 4
    x = torch.arange(B * Cc * T, dtype=torch.float32).reshape(B, Cc, T)
    y = functions.to2d(x)
    assert y.shape == (T * B, Cc)
    for t in range(T):
        for b in range(B):
            assert torch.equal(y[t * B + b], x[b, :, t])
    assert functions.to2d(y) is y


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors (reference: functions.py:67)."""
    x = torch.zeros(4, 3, requires_grad=True)
    with pytest.raises(AssertionError):
        functions.chain_loss(x, None, None)


def test_reference_import_paths_work():
    """example/chime5/train.py:11-12 imports these names."""
    from torchain import io as ref_io
    from torchain.functions import ChainResults, chain_loss

    assert chain_loss is functions.chain_loss and ChainResults is functions.ChainResults
    assert ref_io.DenominatorGraph is io.DenominatorGraph and ref_io.Supervision is io.Supervision
    assert callable(ref_io.set_kaldi_device)
    assert torchain_amd.chain_loss is functions.chain_loss


def test_shard_range_partitions_exactly():
    for S, W in [(256, 8), (2048, 8), (7, 3), (5, 8)]:
        spans = [parallel.shard_range(S, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == S
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_torch_library_registration_and_fake_implementation():
    """``torchain_amd::chain_step`` is a registered operator with a schema and a fake implementation:
    shapes of its three outputs without a GPU, for a training and an evaluation step."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from torchain_amd import ops  # noqa: F401  (registers the operator)
    op = torch.ops.torchain_amd.chain_step
    assert "Tensor input, Tensor xent_input" in str(op.default._schema)
    with FakeTensorMode():
        x, xe = torch.empty(4, 7, 5), torch.empty(4, 7, 5)
        out, g, xg = op(x, xe, 1, 2, 0.0, 0.1, 0.1, True, True)
        assert out.shape == (6,) and g.shape == x.shape and xg.shape == xe.shape
        out, g, xg = op(x, torch.empty(0), 1, 2, 0.0, 0.1, 0.0, True, False)
        assert out.shape == (6,) and g.numel() == 0 and xg.numel() == 0
