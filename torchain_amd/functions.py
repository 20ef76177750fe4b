"""``chain_loss`` / ``ChainResults``: the autograd wrapper of the reference
(``torchain/functions.py:9-138``) on top of the HIP C ABI.

Same names, argument meaning, defaults and quirks as the reference:

* ``ChainResults.data`` is a CPU float tensor ``[objf, l2_term, weight]``; ``loss = -objf/weight``
  (l2 not included, ``functions.py:19``).
* ``chain_loss`` accepts ``(B, C, T)`` or ``(T*B, C)`` input; 3-D input is permuted to frame-major
  rows ``t*B + b`` (``functions.py:118-125``).  For CUDA float32 3-D inputs the permutation, its inverse
  and the backward's negation / xent scale are single fused passes (``tc_to2d`` / ``tc_from2d``) with
  identical values.
* backward returns ``-mmi_grad`` for the input and ``-xent_regularize * xent_grad`` for
  ``xent_input``; ``grad_output`` is ignored and nothing is divided by ``weight``
  (``functions.py:106-115``).
* ``kaldi_way=False`` re-runs the whole objective on ``xent_input`` and overwrites ``results`` and
  the MMI gradient with that second call's outputs, as the reference does (``functions.py:96-103``).
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import io
from ._lib import check, lib


class ChainResults:
    def __init__(self):
        # ``data``: the CPU float tensor [objf, l2_term, weight] of the reference (functions.py:11-21).  It is read from
        # the device when it is first looked at (one 12-byte D2H), not inside ``chain_loss``: a training loop that
        # logs every few steps enqueues its steps without waiting for the GPU in between -- which is what lets the
        # reader's hand-over of the next minibatch (io.RandExample) hide behind the running step.
        self._host = torch.zeros(3)
        self._stale = False
        # Device-side originals: [objf, l2_term, weight] as the kernels left them (float32[3]) and Kaldi's
        # cross-entropy objective sum(xent_output * xent_deriv) (float64[1]; [K] nnet-chain-training.cc, a TODO in
        # the reference, functions.py:88-89).  The data-parallel wrapper reduces THESE (one collective, no host
        # round trip); ``xent_objf`` is read from the device only when somebody asks for it.
        self._dev = None
        self._ready = None       # event recorded on the stream that produced _dev: the lazy reads wait for it
        self._xent_dev = None
        self._xent_ready = None
        self._xent_scale = 1.0
        self._xent_host = None
        self._defer_host_copy = False  # set by chain_loss_data_parallel: the one D2H follows the all-reduce

    @property
    def data(self):
        if self._stale and not self._defer_host_copy and self._dev is not None:
            # the copy runs on whatever stream is current NOW, which need not be the one chain_loss ran on
            if self._ready is not None:
                torch.cuda.current_stream(self._dev.device).wait_event(self._ready)
            self._host.copy_(self._dev)
            self._stale = False
        return self._host

    def invalidate(self):
        """The device-side values changed behind this object -- a captured graph holding the step was replayed: the next
        read of ``data`` / ``xent_objf`` copies them again (on the current stream, behind the replay)."""
        if self._dev is not None:
            self._stale = True
        if self._xent_dev is not None:
            self._xent_host = None

    @data.setter
    def data(self, value):
        self._host = value
        self._stale = False
        self._dev = None  # (callers that sum ChainResults.data over steps own the numbers from here on)
        self._ready = None

    @property
    def xent_objf(self):
        """Kaldi's cross-entropy objective, or None without a xent branch.  Lazy: the value stays on the device
        until it is read (one 8-byte D2H), so a training step has no second host sync."""
        if self._xent_host is None and self._xent_dev is not None:
            if self._xent_ready is not None:
                torch.cuda.current_stream(self._xent_dev.device).wait_event(self._xent_ready)
            self._xent_host = float(self._xent_dev.item()) * self._xent_scale
        return self._xent_host

    @xent_objf.setter
    def xent_objf(self, value):
        self._xent_host = value
        self._xent_dev = None

    def __repr__(self):
        return "ChainResults(loss=%f, objf=%f, l2_term=%f, weight=%lf)" % (
            self.loss, self.data[0], self.data[1], self.data[2])

    @property
    def loss(self):
        return -self.data[0] / self.data[2]

    @property
    def xent_loss(self):
        """-xent_objf / weight, the per-frame cross-entropy Kaldi logs as 'output-xent'; None without a xent branch."""
        return None if self.xent_objf is None else -self.xent_objf / float(self.data[2])


_workspaces = {}  # (device, stream) -> workspace, most recently used last
_MAX_WORKSPACES = 4  # per process: a C3 workspace is 1.3 GB, and streams come and go


def _workspace(device, stream, nbytes):
    """The caller's workspace for (device, stream): reused from step to step, at most ``_MAX_WORKSPACES`` alive (least
    recently used dropped; the caching allocator keeps a dropped block away from other streams until the work
    queued on its own stream has passed)."""
    key = (device.index, stream)
    ws = _workspaces.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    _workspaces[key] = ws
    while len(_workspaces) > _MAX_WORKSPACES:
        _workspaces.pop(next(iter(_workspaces)))
    return ws


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def compute_chain_objf_and_deriv(den_graph, supervision, nnet_output, results, nnet_output_deriv,
                                 xent_output_deriv, l2_regularize, leaky_hmm_coefficient, xent_regularize,
                                 as_gradients=False, holder=None):
    """The hot call: replaces ``my_lib.my_lib_ComputeChainObjfAndDeriv`` (``src/my_lib.h:33-42``).
    ``results`` is the CPU float[3] tensor of ``ChainResults`` and is filled on return.  ``as_gradients``: the two
    matrices come back as the reference's backward returns them (``functions.py:106-115``), ``-deriv`` and
    ``-xent_regularize * xent_deriv`` (``tc_chain_objf_and_grad``).  ``holder``: the ``ChainResults`` that keeps the
    device-side float[3]; when it asks for it (data-parallel use) the host copy is left to the caller."""
    assert nnet_output.is_cuda, "Only the HIP (ROCm) implementation is available"
    if nnet_output.dim() != 2 or nnet_output.stride(1) != 1 or nnet_output.dtype != torch.float32:
        raise ValueError("nnet_output must be a 2-D float32 tensor with unit column stride")
    if isinstance(den_graph, io.DenominatorGraph):
        den_graph.prepare(nnet_output.device)  # (a dictionary look-up after the first call on a device)
    den_ptr = den_graph.ptr if isinstance(den_graph, io.DenominatorGraph) else den_graph
    sup_ptr = supervision.ptr if isinstance(supervision, io.Supervision) else supervision
    rows, cols = nnet_output.shape
    device = nnet_output.device
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        n_seq = lib.tc_supervision_num_sequence(sup_ptr)
        n_frame = lib.tc_supervision_num_frame(sup_ptr)
        nbytes = lib.tc_chain_workspace_bytes(den_ptr, n_seq, n_frame)
        if nbytes < 0:
            check(int(nbytes), "tc_chain_workspace_bytes")
        ws = _workspace(device, stream, nbytes)
        res_dev = torch.empty(3, dtype=torch.float32, device=device)
        for t in (nnet_output_deriv, xent_output_deriv):
            if t is not None and (t.dim() != 2 or t.stride(1) != 1 or t.shape != nnet_output.shape):
                raise ValueError("derivative tensors must match nnet_output and have unit column stride")
        entry = lib.tc_chain_objf_and_grad if as_gradients else lib.tc_chain_objf_and_deriv
        rc = entry(
            den_ptr, sup_ptr, _ptr(nnet_output), rows, cols, nnet_output.stride(0), _ptr(res_dev),
            _ptr(nnet_output_deriv), nnet_output_deriv.stride(0) if nnet_output_deriv is not None else 0,
            _ptr(xent_output_deriv), xent_output_deriv.stride(0) if xent_output_deriv is not None else 0,
            float(l2_regularize), float(leaky_hmm_coefficient), float(xent_regularize), _ptr(ws), ws.numel(),
            device.index, C.c_void_p(stream))
        check(rc, "tc_chain_objf_and_grad" if as_gradients else "tc_chain_objf_and_deriv")
        if holder is not None:
            holder._dev = res_dev  # (ChainResults.data copies it to the host when it is read)
            holder._stale = True
            holder._ready = torch.cuda.Event()
            holder._ready.record(torch.cuda.current_stream(device))
        else:
            results.copy_(res_dev)  # 12-byte D2H, the one host sync of the call (reference: >= 4)
    return results


def xent_objective(xent_output, xent_output_deriv):
    """sum(xent_output * xent_output_deriv) on the device (``tc_xent_objf``), as a float64[1] DEVICE tensor: no host
    sync (``ChainResults.xent_objf`` reads it when asked)."""
    for t in (xent_output, xent_output_deriv):
        if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
            raise ValueError("xent tensors must be 2-D float32 CUDA tensors with unit column stride")
    if xent_output.shape != xent_output_deriv.shape:
        raise ValueError("xent_input and its derivative must have the same shape")
    device = xent_output.device
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        ws = _workspace(device, stream, 4096)
        out = torch.empty(1, dtype=torch.float64, device=device)
        rc = lib.tc_xent_objf(_ptr(xent_output), xent_output.shape[0], xent_output.shape[1], xent_output.stride(0),
                              _ptr(xent_output_deriv), xent_output_deriv.stride(0), _ptr(out), _ptr(ws), ws.numel(),
                              device.index, C.c_void_p(stream))
        check(rc, "tc_xent_objf")
        return out


def _device_loss(results, like):
    """-objf / weight as a one-element tensor on the input's device (the reference's ``input.new([results.loss])``,
    functions.py:104) computed from the device-side results: no host round trip."""
    r = results._dev
    return (-(r[0] / r[2])).reshape(1).to(like.dtype)


class _ChainLoss(Function):
    """Lattice-free MMI loss; see the reference docstring ``torchain/functions.py:23-60`` for the
    meaning of every argument (identical here)."""

    @staticmethod
    def forward(ctx, input, xent_input, results, den_graph, supervision,
                l2_regularize, leaky_hmm_coefficient, xent_regularize=0.0, kaldi_way=False):
        assert input.is_cuda, "Only CUDA implementation is available"
        # The kernels write the two matrices as backward() returns them (-deriv, -xent_regularize * xent_deriv):
        # the reference scales one and negates both in three more passes over (T*B, C) tensors.
        mmi_grad = torch.empty_like(input, memory_format=torch.contiguous_format)
        use_xent = xent_input is not None and xent_regularize != 0.0
        xent_grad = torch.empty_like(xent_input, memory_format=torch.contiguous_format) if use_xent else None
        compute_chain_objf_and_deriv(den_graph, supervision, input.detach(), results._host, mmi_grad, xent_grad,
                                     l2_regularize, leaky_hmm_coefficient, xent_regularize, as_gradients=True,
                                     holder=results)
        ctx.mmi_grad = mmi_grad
        if use_xent:
            # sum(xent_output * xent_deriv) from the scaled matrix: the scale is one factor of every term
            results._xent_dev = xent_objective(xent_input.detach(), xent_grad)
            results._xent_ready = torch.cuda.Event()
            results._xent_ready.record(torch.cuda.current_stream(xent_input.device))
            results._xent_scale = 1.0 / -float(xent_regularize)
            results._xent_host = None
            if not kaldi_way:  # the reference's second call (functions.py:96-103)
                compute_chain_objf_and_deriv(den_graph, supervision, xent_input.detach(), results._host, mmi_grad,
                                             xent_grad, l2_regularize, leaky_hmm_coefficient, xent_regularize,
                                             as_gradients=True, holder=results)
            ctx.xent_grad = xent_grad
        if results._defer_host_copy:
            return input.new_zeros(1)  # (filled in behind the all-reduce: parallel.chain_loss_data_parallel)
        return _device_loss(results, input)

    @staticmethod
    def backward(ctx, grad_output):
        return (ctx.mmi_grad, getattr(ctx, "xent_grad", None), None, None, None, None, None, None, None)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def to2d_hip(x):
    """(B, C, T) -> (T*B, C) in one pass (``tc_to2d``); same values as ``to2d``."""
    B, Cn, T = x.shape
    x = x.contiguous()
    out = torch.empty(T * B, Cn, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.tc_to2d(_ptr(x), B, Cn, T, _ptr(out), out.stride(0), x.device.index, _stream(x.device)), "tc_to2d")
    return out


def from2d_hip(g2d, shape, scale):
    """(T*B, C) -> scale * (B, C, T) in one pass (``tc_from2d``): inverse permutation fused with the
    negation / xent scale of the reference's backward (``functions.py:108-112``)."""
    B, Cn, T = shape
    out = torch.empty(B, Cn, T, dtype=torch.float32, device=g2d.device)
    with torch.cuda.device(g2d.device):
        check(lib.tc_from2d(_ptr(g2d), g2d.stride(0), B, Cn, T, float(scale), _ptr(out), g2d.device.index,
                            _stream(g2d.device)), "tc_from2d")
    return out


class _ChainLoss3d(Function):
    """``_ChainLoss`` for ``(B, C, T)`` inputs with the layout conversions fused in: the
    value and the gradients are those of ``_ChainLoss.apply(to2d(input), ...)`` followed by autograd's
    inverse permute, without the three extra passes over the activation tensor."""

    @staticmethod
    def forward(ctx, input, xent_input, results, den_graph, supervision,
                l2_regularize, leaky_hmm_coefficient, xent_regularize=0.0, kaldi_way=False):
        assert input.is_cuda, "Only CUDA implementation is available"
        x2d = to2d_hip(input.detach())
        mmi_grad = torch.empty_like(x2d)
        use_xent = xent_input is not None and xent_regularize != 0.0
        xe2d = to2d_hip(xent_input.detach()) if use_xent else None
        xent_grad = torch.empty_like(xe2d) if use_xent else None
        compute_chain_objf_and_deriv(den_graph, supervision, x2d, results._host, mmi_grad, xent_grad,
                                     l2_regularize, leaky_hmm_coefficient, xent_regularize, holder=results)
        if use_xent:
            results._xent_dev = xent_objective(xe2d, xent_grad)
            results._xent_scale = 1.0
            results._xent_host = None
        if use_xent and not kaldi_way:  # the reference's second call (functions.py:96-103)
            compute_chain_objf_and_deriv(den_graph, supervision, xe2d, results._host, mmi_grad, xent_grad,
                                         l2_regularize, leaky_hmm_coefficient, xent_regularize, holder=results)
        ctx.mmi_grad = mmi_grad
        ctx.in_shape = tuple(input.shape)
        if use_xent:
            ctx.xent_grad = xent_grad
            ctx.xent_shape = tuple(xent_input.shape)
            ctx.xent_scale = float(xent_regularize)
        if results._defer_host_copy:
            return input.new_zeros(1)
        return _device_loss(results, input)

    @staticmethod
    def backward(ctx, grad_output):
        g = from2d_hip(ctx.mmi_grad, ctx.in_shape, -1.0)
        xg = from2d_hip(ctx.xent_grad, ctx.xent_shape, -ctx.xent_scale) if hasattr(ctx, "xent_grad") else None
        return (g, xg, None, None, None, None, None, None, None)


class _ChainStep(Function):
    """The whole training-side step behind ONE library call (``tc_chain_step``): what ``_ChainLoss`` / ``_ChainLoss3d``
    above do with four to six calls and as many tensor allocations -- layout copies, objective, the reference's second
    call for ``kaldi_way=False``, cross-entropy objective, loss value, gradients in the input's own layout.  The
    reference's step is one FFI call too (``torchain/functions.py:82-86``).  Same values as the classes above (tested)."""

    @staticmethod
    def forward(ctx, input, xent_input, results, den_graph, supervision,
                l2_regularize, leaky_hmm_coefficient, xent_regularize=0.0, kaldi_way=False):
        out, grad, xgrad = _run_step(input, xent_input, results, den_graph, supervision, l2_regularize,
                                     leaky_hmm_coefficient, xent_regularize, kaldi_way, want_grad=True)
        ctx.grads = (grad, xgrad)
        if results._defer_host_copy:
            return input.new_zeros(1)
        return out[3:4]

    @staticmethod
    def backward(ctx, grad_output):
        return (ctx.grads[0], ctx.grads[1], None, None, None, None, None, None, None)


def _run_step(input, xent_input, results, den_graph, supervision, l2_regularize, leaky_hmm_coefficient,
              xent_regularize, kaldi_way, want_grad):
    """One ``tc_chain_step``.  ``want_grad=False`` is the evaluation step (``grad == NULL``): forward recursions only,
    no gradient tensors -- [K] ``ComputeChainObjfAndDeriv`` with ``nnet_output_deriv == NULL``."""
    device = input.device
    three_d = input.dim() == 3
    use_xent = xent_input is not None and xent_regularize != 0.0
    if isinstance(den_graph, io.DenominatorGraph):
        den_graph.prepare(device)
    den_ptr = den_graph.ptr if isinstance(den_graph, io.DenominatorGraph) else den_graph
    if isinstance(supervision, io.Supervision):
        sup_ptr, S, T, P = supervision.ptr, supervision.n_batch, supervision.n_frame, supervision.n_pdf
    else:
        sup_ptr = supervision
        S, T, P = (lib.tc_supervision_num_sequence(sup_ptr), lib.tc_supervision_num_frame(sup_ptr),
                   lib.tc_supervision_num_pdf(sup_ptr))
    want = (S, P, T) if three_d else (S * T, P)
    if tuple(input.shape) != want or (use_xent and tuple(xent_input.shape) != want):
        raise ValueError("nnet output of shape %s does not match the supervision (%s expected)" % (tuple(input.shape), want))
    x = input.detach()
    xe = xent_input.detach() if use_xent else None
    grad = torch.empty_like(x, memory_format=torch.contiguous_format) if want_grad else None
    xgrad = torch.empty_like(xe, memory_format=torch.contiguous_format) if use_xent and want_grad else None
    with torch.cuda.device(device):
        cur = torch.cuda.current_stream(device)
        key = (S, T, three_d, use_xent)
        sizes = den_graph.__dict__.setdefault("_step_bytes", {}) if isinstance(den_graph, io.DenominatorGraph) else {}
        nbytes = sizes.get(key)
        if nbytes is None:
            nbytes = lib.tc_chain_step_workspace_bytes(den_ptr, S, T, int(three_d), int(use_xent))
            if nbytes < 0:
                check(int(nbytes), "tc_chain_step_workspace_bytes")
            sizes[key] = nbytes
        capturing = torch.cuda.is_current_stream_capturing()
        # (inside a graph capture the workspace comes from the graph's own pool and lives as long as the graph does)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device) if capturing else _workspace(device, cur.cuda_stream, nbytes)
        out = torch.empty(6, dtype=torch.float32, device=device)  # objf, l2_term, weight | loss | xent objective (f64)
        rc = lib.tc_chain_step(den_ptr, sup_ptr, _ptr(x), _ptr(xe), int(three_d), 0 if three_d else x.stride(0),
                               float(l2_regularize), float(leaky_hmm_coefficient), float(xent_regularize),
                               int(bool(kaldi_way)), _ptr(grad), _ptr(xgrad), C.c_void_p(out.data_ptr()),
                               C.c_void_p(out.data_ptr() + 12), C.c_void_p(out.data_ptr() + 16) if use_xent else None,
                               _ptr(ws), ws.numel(), device.index, C.c_void_p(cur.cuda_stream))
        check(rc, "tc_chain_step")
        results._dev = out[:3]
        results._stale = True
        results._ready = None
        if capturing:
            results._graph_keeps = ws  # replays write through it; read the values after a replay with ``invalidate()``
        else:
            results._ready = torch.cuda.Event()
            results._ready.record(cur)
        if use_xent:
            results._xent_dev = out[4:6].view(torch.float64)
            results._xent_ready = results._ready
            results._xent_scale = 1.0 / -float(xent_regularize)  # (the library's sum is of the scaled gradient, both layouts)
            results._xent_host = None
    return out, grad, xgrad


def _one_call(x):
    """Tensors ``tc_chain_step`` takes as they are: CUDA float32, (B, C, T) contiguous or 2-D with unit column stride."""
    if x is None:
        return True
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    return (x.dim() == 3 and x.is_contiguous()) or (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1])


def _needs_grad(input, xent_input):
    if not torch.is_grad_enabled():
        return False
    return input.requires_grad or (xent_input is not None and xent_input.requires_grad)


def _as_step_input(x):
    """A tensor ``tc_chain_step`` takes as it is (evaluation steps only: nothing flows back, so a copy is harmless)."""
    assert x.is_cuda, "Only CUDA implementation is available"
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    return x if _one_call(x) else x.contiguous()


def to2d(x):
    if x.dim() == 3:  # (B, C, T)
        n_pdf = x.shape[1]
        x = x.permute(2, 0, 1).contiguous().view(-1, n_pdf)  # (T * B, C)
    assert x.dim() == 2
    return x


def _fusable(x):
    return x.dim() == 3 and x.is_cuda and x.dtype == torch.float32


def chain_loss(input, den_graph, supervision,
               l2_regularize=0.0, leaky_hmm_coefficient=1e-5,
               xent_regularize=0.0, xent_input=None, kaldi_way=False):
    return _chain_loss_into(ChainResults(), input, den_graph, supervision, l2_regularize, leaky_hmm_coefficient,
                            xent_regularize, xent_input, kaldi_way)


def _chain_loss_into(results, input, den_graph, supervision, l2_regularize, leaky_hmm_coefficient, xent_regularize,
                     xent_input, kaldi_way):
    """``chain_loss`` with the ``ChainResults`` given (parallel.chain_loss_data_parallel passes one that defers the
    host copy of the three floats until after its all-reduce)."""
    if not _needs_grad(input, xent_input):
        # Evaluation (the recipe's validation loop runs under torch.no_grad(), example/chime5/train.py:150-171): the
        # forward recursions only.  The reference has no such check (functions.py:74,82) and pays for a training step.
        x, xe = _as_step_input(input), (_as_step_input(xent_input) if xent_input is not None else None)
        out, _, _ = _run_step(x, xe, results, den_graph, supervision, l2_regularize, leaky_hmm_coefficient,
                              xent_regularize, kaldi_way, want_grad=False)
        return (input.new_zeros(1) if results._defer_host_copy else out[3:4].to(input.dtype)), results
    if (_one_call(input) and _one_call(xent_input) and (xent_input is None or xent_input.dim() == input.dim())
            and (xent_input is None or xent_input.dim() == 3 or xent_input.stride(0) == input.stride(0))):
        loss = _ChainStep.apply(input, xent_input, results, den_graph, supervision,
                                l2_regularize, leaky_hmm_coefficient, xent_regularize, kaldi_way)
        return loss, results
    if _fusable(input) and (xent_input is None or _fusable(xent_input)):
        loss = _ChainLoss3d.apply(input, xent_input, results, den_graph, supervision,
                                  l2_regularize, leaky_hmm_coefficient, xent_regularize, kaldi_way)
        return loss, results
    input = to2d(input)
    if xent_input is not None:
        xent_input = to2d(xent_input)

    loss = _ChainLoss.apply(input, xent_input, results, den_graph, supervision,
                            l2_regularize, leaky_hmm_coefficient, xent_regularize, kaldi_way)
    return loss, results
