"""Independent float64 formulation of the chain objective -- TEST INFRASTRUCTURE ONLY.

A second, structurally different statement of the same mathematics as oracle/chain_oracle.c, used to
pin that restatement (parity is otherwise unpinned: see chain_oracle.c's header) and to generate
tests/golden/*.npz.  Differences from the Kaldi-style restatement, on purpose:

* everything in float64 and in the log semiring (``logsumexp`` over arcs), no per-frame
  renormalisation, no "arbitrary scale" bookkeeping;
* derivatives come from torch autograd of the scalar objective, not from a hand-written
  backward (beta / occupation) recursion;
* the numerator runs level-by-level on the merged supervision FST.

Mathematics:
  alpha_0 = pi;  alpha'_t = alpha_t + c * pi * sum_h alpha_t(h)
  alpha_{t+1}(g) = sum_{(h->g, pdf, w)} alpha'_t(h) * w * exp(y[t, s, pdf])
  den = sum_s log sum_h alpha'_T(h, s)
  num = log-partition of the supervision acceptor with arc scores y[row(tau), pdf] - arc.weight
  objf = w_sup * num - w_sup * den;  l2_term = -0.5 * w_sup * l2 * sum y^2
  deriv = d(objf + l2_term) / dy;  xent_deriv = w_sup * d num / dy
"""
import numpy as np
import torch


def _segment_logsumexp(term, index, size):
    """logsumexp of term[..., a] grouped by index[a] -> [..., size]; empty groups give -inf."""
    idx = index.expand_as(term)
    mx = torch.full(term.shape[:-1] + (size,), -np.inf, dtype=term.dtype)
    mx = mx.scatter_reduce(-1, idx, term.detach(), "amax", include_self=True)
    safe = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    ex = torch.exp(term - safe.gather(-1, idx))
    sm = torch.zeros_like(mx).scatter_add(-1, idx, ex)
    return torch.where(sm > 0, safe + torch.log(torch.where(sm > 0, sm, torch.ones_like(sm))),
                       torch.full_like(sm, -np.inf))


def den_logprob(fst, initial_probs, y, num_sequences, leaky):
    """y: torch float64 (T*S, P), row = t*S + s.  Returns sum_s log Z_den(s)."""
    S = num_sequences
    T = y.shape[0] // S
    H = fst.num_states
    src = torch.as_tensor(np.asarray(fst.src, np.int64))
    dst = torch.as_tensor(np.asarray(fst.dst, np.int64))
    pdf = torch.as_tensor(np.asarray(fst.ilabel, np.int64) - 1)
    logw = -torch.as_tensor(np.asarray(fst.weight, np.float64))
    # pi(h) = 0 (state not reached in 100 iterations) is floored at 1e-300 so that autograd never
    # meets logaddexp(-inf, -inf); the floor is ~250 orders of magnitude below anything measurable.
    logpi = torch.as_tensor(np.log(np.maximum(np.asarray(initial_probs, np.float64), 1e-300)))
    logc = float(np.log(leaky))
    yy = y.view(T, S, -1)

    def dash(la):
        tot = torch.logsumexp(la, dim=1, keepdim=True)
        return torch.logaddexp(la, logc + logpi.unsqueeze(0) + tot)

    la = logpi.unsqueeze(0).expand(S, H)
    for t in range(T):
        lad = dash(la)
        term = lad[:, src] + logw.unsqueeze(0) + yy[t][:, pdf]
        la = _segment_logsumexp(term, dst.unsqueeze(0), H)
    return torch.logsumexp(dash(la), dim=1).sum()


def _state_times(sup):
    times = np.full(sup.num_states, -1, np.int64)
    times[0] = 0
    for s in range(sup.num_states):
        for a in range(sup.arc_begin[s], sup.arc_begin[s + 1]):
            times[sup.nextstate[a]] = times[s] + 1
    return times


def num_logprob(sup, y):
    """log-partition of the merged supervision acceptor (unweighted by sup.weight)."""
    S, T = sup.num_sequences, sup.frames_per_sequence
    times = _state_times(sup)
    nst = sup.num_states
    arc_src = np.repeat(np.arange(nst), np.diff(sup.arc_begin))
    arc_t = times[arc_src]
    la = torch.full((nst,), -np.inf, dtype=torch.float64)
    la[0] = 0.0
    src_all = torch.as_tensor(arc_src.astype(np.int64))
    dst_all = torch.as_tensor(np.asarray(sup.nextstate, np.int64))
    pdf_all = torch.as_tensor(np.asarray(sup.ilabel, np.int64) - 1)
    logw_all = -torch.as_tensor(np.asarray(sup.arc_weight, np.float64))
    order = np.argsort(arc_t, kind="stable")
    bounds = np.searchsorted(arc_t[order], np.arange(S * T + 1))
    for tau in range(S * T):
        sel = torch.as_tensor(order[bounds[tau]:bounds[tau + 1]].astype(np.int64))
        row = tau // T + S * (tau % T)
        term = la[src_all[sel]] + logw_all[sel] + y[row, pdf_all[sel]]
        upd = _segment_logsumexp(term, dst_all[sel], nst)
        la = torch.where(torch.isinf(upd), la, upd)  # every state has exactly one time
    fin = torch.as_tensor(np.asarray(sup.final, np.float64))
    mask = ~torch.isinf(fin)
    return torch.logsumexp(la[mask] - fin[mask], dim=0)


def chain_objf_and_deriv(fst, initial_probs, sup, y, l2_regularize=0.0, leaky=1e-5):
    """Returns dict(objf, l2_term, weight, num, den, deriv, xent_deriv) in float64."""
    yt = torch.tensor(np.asarray(y, np.float64), requires_grad=True)
    w = float(sup.weight)
    num = num_logprob(sup, yt)
    (gnum,) = torch.autograd.grad(num, yt, retain_graph=False)
    yt2 = torch.tensor(np.asarray(y, np.float64), requires_grad=True)
    den = den_logprob(fst, initial_probs, yt2, sup.num_sequences, leaky)
    (gden,) = torch.autograd.grad(den, yt2)
    objf = w * float(num.detach()) - w * float(den.detach())
    ynp = np.asarray(y, np.float64)
    l2_term = -0.5 * w * l2_regularize * float((ynp * ynp).sum())
    deriv = w * gnum.numpy() - w * gden.numpy() - w * l2_regularize * ynp
    return dict(objf=objf, l2_term=l2_term, weight=w * sup.num_sequences * sup.frames_per_sequence,
                num=float(num.detach()), den=float(den.detach()), deriv=deriv, xent_deriv=w * gnum.numpy(),
                den_deriv=gden.numpy())


def den_logprob_and_deriv(fst, initial_probs, y, num_sequences, leaky):
    yt = torch.tensor(np.asarray(y, np.float64), requires_grad=True)
    den = den_logprob(fst, initial_probs, yt, num_sequences, leaky)
    (g,) = torch.autograd.grad(den, yt)
    return float(den.detach()), g.numpy()
