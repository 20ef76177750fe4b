"""``torch.library`` registration of the training-side step: ``torchain_amd::chain_step``.

The product path (``functions.chain_loss``) is a ``torch.autograd.Function`` over ctypes calls, which is what the reference's is
(``torchain/functions.py:62-115`` over cffi) and is opaque to ``torch.compile`` / ``torch.export``.  This module registers the same
one-call step (``tc_chain_step``) as a custom operator with a fake (meta) implementation and an autograd formula, so that a model
whose loss is ``chain_loss_op`` traces as ONE node:

    torch.ops.torchain_amd.chain_step(input, xent_input, den_graph_ptr, supervision_ptr, l2, leaky, xent, kaldi_way, want_grad)
        -> (out6, grad, xent_grad)
        out6      float32[6]: objf, l2_term, weight | loss = -objf / weight | xent objective (float64 in two floats)
        grad      what backward() returns for ``input`` (-deriv; ``grad_output`` is ignored, nothing is divided by weight:
                  ``functions.py:106-115``); empty when ``want_grad`` is false (the evaluation step)
        xent_grad -xent_regularize * xent_deriv, or empty

Handles cross the operator boundary as integers (the ``.ptr`` of ``io.DenominatorGraph`` / ``io.Supervision``: plain C pointers,
as in the C ABI); the caller keeps the Python objects alive.  Same values, bit for bit, as ``functions.chain_loss`` (tested)."""
import torch

from . import functions, io

_EMPTY = (0,)


@torch.library.custom_op("torchain_amd::chain_step", mutates_args=())
def chain_step(input: torch.Tensor, xent_input: torch.Tensor, den_graph_ptr: int, supervision_ptr: int, l2_regularize: float,
               leaky_hmm_coefficient: float, xent_regularize: float, kaldi_way: bool, want_grad: bool
               ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    use_xent = xent_input.numel() > 0 and xent_regularize != 0.0
    results = functions.ChainResults()
    out, grad, xgrad = functions._run_step(input, xent_input if use_xent else None, results, den_graph_ptr, supervision_ptr,
                                           l2_regularize, leaky_hmm_coefficient, xent_regularize, kaldi_way, want_grad=want_grad)
    return out, (grad if grad is not None else input.new_empty(_EMPTY)), (xgrad if xgrad is not None else input.new_empty(_EMPTY))


@chain_step.register_fake
def _(input, xent_input, den_graph_ptr, supervision_ptr, l2_regularize, leaky_hmm_coefficient, xent_regularize, kaldi_way, want_grad):
    use_xent = xent_input.numel() > 0 and xent_regularize != 0.0
    return (input.new_empty((6,), dtype=torch.float32),
            torch.empty_like(input, memory_format=torch.contiguous_format) if want_grad else input.new_empty(_EMPTY),
            torch.empty_like(xent_input, memory_format=torch.contiguous_format) if want_grad and use_xent else input.new_empty(_EMPTY))


def _setup(ctx, inputs, output):
    _out, grad, xgrad = output
    ctx.save_for_backward(grad, xgrad)
    ctx.has_xent = xgrad.numel() > 0


def _backward(ctx, g_out, g_grad, g_xgrad):
    grad, xgrad = ctx.saved_tensors
    if grad.numel() == 0:
        raise RuntimeError("torchain_amd::chain_step was run as an evaluation step (want_grad=False): it has no gradient")
    # the reference's backward: the stored matrices whatever grad_output is (torchain/functions.py:106-115)
    return grad, (xgrad if ctx.has_xent else None), None, None, None, None, None, None, None


chain_step.register_autograd(_backward, setup_context=_setup)


def chain_loss_op(input, den_graph, supervision, l2_regularize=0.0, leaky_hmm_coefficient=1e-5, xent_regularize=0.0, xent_input=None,
                  kaldi_way=False):
    """``functions.chain_loss`` through the registered operator: ``(loss, results)`` with the same values and gradients.  Inputs
    must be what ``tc_chain_step`` takes as they are (CUDA float32, (B, C, T) contiguous or 2-D with unit column stride)."""
    if not (functions._one_call(input) and functions._one_call(xent_input)):
        raise ValueError("chain_loss_op needs CUDA float32 tensors, (B, C, T) contiguous or 2-D with unit column stride")
    if isinstance(den_graph, io.DenominatorGraph):
        den_graph.prepare(input.device)
    want_grad = functions._needs_grad(input, xent_input)
    xe = xent_input if xent_input is not None else input.new_empty(_EMPTY)
    out, _grad, _xgrad = torch.ops.torchain_amd.chain_step(
        input, xe, int(den_graph.ptr.value if hasattr(den_graph.ptr, "value") else den_graph.ptr),
        int(supervision.ptr.value if hasattr(supervision.ptr, "value") else supervision.ptr), float(l2_regularize),
        float(leaky_hmm_coefficient), float(xent_regularize), bool(kaldi_way), bool(want_grad))
    results = functions.ChainResults()
    results._dev = out.detach()[:3]
    results._stale = True
    results._ready = torch.cuda.Event()
    results._ready.record(torch.cuda.current_stream(input.device))
    if xent_input is not None and xent_regularize != 0.0:
        results._xent_dev = out.detach()[4:6].view(torch.float64)
        results._xent_ready = results._ready
        results._xent_scale = 1.0 / -float(xent_regularize)
    return out[3:4], results
