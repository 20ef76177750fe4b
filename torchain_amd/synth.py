"""Seeded synthetic denominator graphs, supervisions and nnet outputs.

The reference's real inputs (CHiME-5 ``den.fst`` and chain egs, ``test/test.py:36-38,83-86``) are
private, so tests and ``bench.py`` drive the path with generators that mimic their structure.
Everything here is plain numpy on the host; nothing in this file computes the loss.

FST conventions are Kaldi's (``src/my_lib_example.cpp:129-134``, [K] chain-den-graph.cc /
chain-supervision.h): ilabel = pdf_id + 1, weights are tropical (-log prob), final weight +inf marks
a non-final state, arcs are listed state-major.
"""
from collections import namedtuple

import numpy as np

DenFst = namedtuple("DenFst", "num_states src dst ilabel weight final start num_pdfs")
SupFst = namedtuple(
    "SupFst", "weight num_sequences frames_per_sequence label_dim num_states arc_begin ilabel arc_weight nextstate final"
)


def left_to_right_den_fst(num_pdfs=200, seed=42):
    """Config C1: random 3-state left-to-right den.fst (0->0,0->1,1->1,1->2,2->2,2->0)."""
    rng = np.random.default_rng(seed)
    pairs = [(0, 0), (0, 1), (1, 1), (1, 2), (2, 2), (2, 0)]
    src = np.array([p[0] for p in pairs], np.int32)
    dst = np.array([p[1] for p in pairs], np.int32)
    pdf = rng.integers(0, num_pdfs, size=len(pairs)).astype(np.int32)
    prob = rng.uniform(0.1, 1.0, size=len(pairs))
    for s in range(3):
        m = src == s
        prob[m] /= prob[m].sum()
    return DenFst(3, src, dst, (pdf + 1).astype(np.int32), (-np.log(prob)).astype(np.float32),
                  np.zeros(3, np.float32), 0, num_pdfs)


def random_den_fst(num_states, out_degree, num_pdfs, seed=42):
    """CHiME5-like den.fst: per state one self-loop labelled with its self-loop pdf and
    ``out_degree - 1`` arcs to uniformly random states labelled with the destination's forward pdf
    (chain topology: two pdfs per tied phone state).  Every pdf is used at least once when
    2*num_states >= num_pdfs.  Arc probs ~ U(0.1, 1) normalised per state; all states final (0)."""
    rng = np.random.default_rng(seed)
    H, d, P = int(num_states), int(out_degree), int(num_pdfs)
    n = 2 * H
    if n >= P:
        slots = rng.permutation(np.concatenate([rng.permutation(P), rng.integers(0, P, size=n - P)]))
    else:
        slots = rng.permutation(P)[:n]
    self_pdf = slots[:H].astype(np.int32)
    fwd_pdf = slots[H:].astype(np.int32)
    src = np.repeat(np.arange(H, dtype=np.int32), d)
    dst = rng.integers(0, H, size=(H, d)).astype(np.int32)
    dst[:, 0] = np.arange(H)
    pdf = fwd_pdf[dst]
    pdf[:, 0] = self_pdf
    prob = rng.uniform(0.1, 1.0, size=(H, d))
    prob /= prob.sum(axis=1, keepdims=True)
    return DenFst(H, src, dst.reshape(-1), (pdf.reshape(-1) + 1).astype(np.int32),
                  (-np.log(prob)).reshape(-1).astype(np.float32), np.zeros(H, np.float32), 0, P)


def skewed_den_fst(num_states, num_arcs, num_pdfs, seed=3, hub_fraction=0.02):
    """A graph with heavily skewed in/out degrees and arbitrary arc->pdf labelling (no chain
    structure), some non-final states and parallel arcs: exercises row splitting and the general
    case of the transition schedule."""
    rng = np.random.default_rng(seed)
    H, A, P = int(num_states), int(num_arcs), int(num_pdfs)
    hubs = max(1, int(H * hub_fraction))
    w_state = np.ones(H)
    w_state[rng.choice(H, hubs, replace=False)] = H / hubs / 2.0
    w_state /= w_state.sum()
    src = np.sort(np.concatenate([np.arange(H), rng.choice(H, A - H, p=w_state)])).astype(np.int32)
    dst = rng.choice(H, A, p=rng.permutation(w_state)).astype(np.int32)
    pdf = rng.integers(0, P, size=A).astype(np.int32)
    prob = rng.uniform(0.05, 1.0, size=A)
    tot = np.zeros(H)
    np.add.at(tot, src, prob)
    prob /= tot[src]
    final = np.where(rng.uniform(size=H) < 0.7, 0.0, np.inf).astype(np.float32)
    final[0] = 0.0
    return DenFst(H, src, dst, (pdf + 1).astype(np.int32), (-np.log(prob)).astype(np.float32), final, 0, P)


def skewed_tied_den_fst(num_states, num_arcs, num_pdfs, seed=5, hub_fraction=0.02):
    """Chain-structured ("tied") graph with heavily skewed degrees: every non-self-loop arc carries the
    forward pdf of its destination, most states have one self-loop with their own self-loop pdf, a few
    hub states have hundreds of in- and out-arcs (longer than one schedule row), some states are
    non-final.  Exercises the secondary rows (and their fold) of the owner-computes schedules."""
    rng = np.random.default_rng(seed)
    H, A, P = int(num_states), int(num_arcs), int(num_pdfs)
    hubs = max(1, int(H * hub_fraction))
    w_state = np.ones(H)
    w_state[rng.choice(H, hubs, replace=False)] = H / hubs / 2.0
    w_state /= w_state.sum()
    has_loop = rng.uniform(size=H) < 0.8
    n_loop = int(has_loop.sum())
    n_other = A - n_loop
    o_src = np.concatenate([np.arange(H), rng.choice(H, n_other - H, p=w_state)]).astype(np.int32)
    o_dst = rng.choice(H, n_other, p=rng.permutation(w_state)).astype(np.int32)
    same = o_src == o_dst  # keep the non-self-loop class free of accidental self-loops
    o_dst[same] = (o_dst[same] + 1) % H
    fwd_pdf = rng.integers(0, P, size=H).astype(np.int32)
    self_pdf = rng.integers(0, P, size=H).astype(np.int32)
    l_src = np.nonzero(has_loop)[0].astype(np.int32)
    src = np.concatenate([o_src, l_src])
    dst = np.concatenate([o_dst, l_src])
    pdf = np.concatenate([fwd_pdf[o_dst], self_pdf[l_src]])
    order = np.argsort(src, kind="stable")
    src, dst, pdf = src[order], dst[order], pdf[order]
    prob = rng.uniform(0.05, 1.0, size=len(src))
    tot = np.zeros(H)
    np.add.at(tot, src, prob)
    prob /= tot[src]
    final = np.where(rng.uniform(size=H) < 0.7, 0.0, np.inf).astype(np.float32)
    final[0] = 0.0
    return DenFst(H, src, dst, (pdf + 1).astype(np.int32), (-np.log(prob)).astype(np.float32), final, 0, P)


def phone_lm_den_fst(num_phones=42, num_histories=600, branching=12, num_pdfs=2928, seed=7, backoff_fraction=0.1,
                     unigram_fraction=0.0):
    """A den.fst with the STRUCTURE Kaldi's chain recipe produces (the real ones are private): a pruned n-gram phone
    LM composed with the one-state-per-phone chain topology and a left-biphone tree.

    LM histories h = 0..num_histories-1 each remember their last phone; from h a (Zipf-weighted) subset of
    ``branching`` phones may follow, each leading to a history that ends in that phone -- or, for a
    ``backoff_fraction`` of the LM arcs, to a shared low-order history (0..num_phones-1, one per phone), which is what
    makes a few states very popular.  A graph state is one phone instance (h, p): entered by arcs that carry the
    forward pdf of p in the context of the previous phone, it loops on its self-loop pdf and leaves to every phone
    instance (h', p') of the history h' = next(h, p).  So: chain-structured by construction (every arc into a state
    carries that state's forward pdf), out-degrees ~``branching``, in-degrees from 1 to hundreds, pdfs shared by all
    instances of a biphone.  States: ~num_histories * branching; arcs: ~states * (branching + 1).

    ``unigram_fraction`` > 0 adds the LM's empty history: a fraction of the LM arcs of every history lead to it, it
    remembers no phone, so its phone instances are entered through arcs of as many different forward pdfs as there
    are left contexts -- the graph is then only NEARLY chain-structured (the library splits such states)."""
    rng = np.random.default_rng(seed)
    NP, NH, B, P = int(num_phones), int(num_histories), int(branching), int(num_pdfs)
    assert NH >= NP and B <= NP
    last = np.concatenate([np.arange(NP), rng.integers(0, NP, size=NH - NP)])  # last phone of every history
    empty = NH if unigram_fraction > 0 else -1  # index of the empty history (appended below)
    by_phone = [np.nonzero(last == p)[0] for p in range(NP)]
    zipf = 1.0 / np.arange(1, NP + 1) ** 0.8
    zipf /= zipf.sum()
    # biphone tree: (left phone, phone) -> forward pdf, self-loop pdf; leaves shared between similar contexts
    n_leaf = max(2, P // 2)
    leaf = rng.integers(0, n_leaf, size=(NP, NP))
    fwd_of = (2 * leaf) % P
    self_of = (2 * leaf + 1) % P
    inst = {}        # (h, p) -> state id
    nxt = []         # per state: history reached after the phone
    lm_prob = []     # per state: P_LM(p | h)
    for h in range(NH + (1 if empty >= 0 else 0)):
        nb = NP if h == empty else B  # the empty history allows every phone
        phones = rng.choice(NP, size=nb, replace=False, p=zipf)
        pr = rng.dirichlet(np.ones(nb) * 0.7)
        for p, q in zip(phones, pr):
            inst[(h, int(p))] = len(nxt)
            cand = by_phone[int(p)]
            u = rng.uniform()
            if empty >= 0 and u < unigram_fraction:
                tgt = empty
            elif u < unigram_fraction + backoff_fraction:
                tgt = int(p)
            else:
                tgt = int(cand[rng.integers(0, len(cand))])
            nxt.append(tgt)
            lm_prob.append(float(q))
    H = len(nxt)
    members = [[] for _ in range(NH + (1 if empty >= 0 else 0))]  # phone instances of every history
    for (h, p), g in inst.items():
        members[h].append((p, g))
    src, dst, pdf, prob = [], [], [], []
    for (h, p), g in sorted(inst.items(), key=lambda kv: kv[1]):
        stay = float(rng.uniform(0.3, 0.7))
        left = p if h == empty else int(last[h])  # (the empty history has no left context of its own)
        src.append(g), dst.append(g), pdf.append(int(self_of[left, p])), prob.append(stay)
        for p2, g2 in members[nxt[g]]:
            src.append(g), dst.append(g2), pdf.append(int(fwd_of[p, p2])), prob.append((1.0 - stay) * lm_prob[g2])
    src, dst, pdf = np.array(src, np.int32), np.array(dst, np.int32), np.array(pdf, np.int32)
    prob = np.array(prob)
    tot = np.zeros(H)
    np.add.at(tot, src, prob)
    prob /= tot[src]
    return DenFst(H, src, dst, (pdf + 1).astype(np.int32), (-np.log(prob)).astype(np.float32), np.zeros(H, np.float32), 0, P)


def nearly_tied_den_fst(num_states, out_degree, num_pdfs, seed=42, fraction=0.03):
    """A chain-structured graph (``random_den_fst``) in which a few states are entered through arcs of two
    or three different pdfs and a few carry a second self-loop -- what minimisation produces in a real
    den.fst when two phone instances with equal futures but different forward pdfs are merged.  Exercises
    the state splitting ("tied-ification") of the schedule builder."""
    fst = random_den_fst(num_states, out_degree, num_pdfs, seed=seed)
    rng = np.random.default_rng(seed + 1)
    lab = fst.ilabel.copy()
    H = fst.num_states
    odd = rng.choice(H, max(1, int(H * fraction)), replace=False)
    for g in odd:
        arcs = np.nonzero((fst.dst == g) & (fst.src != g))[0]
        if len(arcs) >= 2:
            k = rng.integers(1, min(3, len(arcs)))
            pick = rng.choice(arcs, k, replace=False)
            lab[pick] = rng.integers(1, num_pdfs + 1, size=k)
    return DenFst(H, fst.src, fst.dst, lab.astype(np.int32), fst.weight, fst.final, 0, num_pdfs)


def initial_probs_f64(fst, num_iters=100):
    """Plain float64 numpy version of the 100-iteration initial-prob estimate (used only to weight
    the synthetic numerator's first arcs and to cross-check the oracle)."""
    H = fst.num_states
    p_arc = np.exp(-fst.weight.astype(np.float64))
    tot = np.exp(-fst.final.astype(np.float64))
    tot = tot + np.bincount(fst.src, weights=p_arc, minlength=H)
    norm = 1.0 / tot
    cur = np.zeros(H)
    cur[fst.start] = 1.0
    avg = np.zeros(H)
    for _ in range(num_iters):
        avg += cur / num_iters
        nxt = np.bincount(fst.dst, weights=cur[fst.src] * norm[fst.src] * p_arc, minlength=H)
        cur = nxt / nxt.sum()
    return avg


def random_supervision(fst, num_sequences, frames_per_sequence, paths_per_sequence=3, seed=7, weight=1.0,
                       initial_probs=None, final_weights=False):
    """Numerator supervision: per sequence the union of k random length-T paths through ``fst``
    stored as a time-sorted epsilon-free acceptor trie; arc weight = -log(den arc prob), the first
    arc of each path also carrying -log pi(start state).  The S per-sequence FSTs are then merged
    the way [K] AppendSupervision does (fst::Concat + RmEpsilon + breadth-first renumbering): the
    final states of sequence k-1 receive copies of sequence k's start arcs.  Because the numerator
    is a weighted subset of denominator paths, objf <= 0 must hold
    (``src/chain-supervision-test.hpp:285``).

    ``final_weights=True`` gives every sequence's final states distinct non-zero final weights, as real
    supervisions have after [K] AddWeightToSupervisionFst: Concat + RmEpsilon then folds final weight f_i of
    boundary state i into its copies of the next sequence's start arcs (the f_i - f_0 re-weighting branch of
    the per-sequence split in ``csrc/supervision.cpp``)."""
    rng = np.random.default_rng(seed)
    S, T, k = int(num_sequences), int(frames_per_sequence), int(paths_per_sequence)
    H = fst.num_states
    pi = initial_probs_f64(fst) if initial_probs is None else np.asarray(initial_probs, np.float64)
    order = np.argsort(fst.src, kind="stable")
    first = np.searchsorted(fst.src[order], np.arange(H + 1))
    start_pool = np.flatnonzero(pi > 1e-8)

    # per sequence: levels[t] = dict(path-prefix-key -> local node id at time t)
    seq_arcs = []  # per sequence: list over t of list of (src_node, dst_node, ilabel, weight)
    seq_nodes = []  # per sequence: number of nodes per level (level 0 has the single root)
    for _ in range(S):
        level_nodes = [1]
        arcs_by_t = [[] for _ in range(T)]
        tries = [dict() for _ in range(T + 1)]
        for _ in range(k):
            h = int(rng.choice(start_pool))
            node, h0 = 0, h
            for t in range(T):
                lo, hi = first[h], first[h + 1]
                a = order[lo + rng.integers(0, hi - lo)]
                nh = int(fst.dst[a])
                key = (node, int(a), h0 if t == 0 else -1)  # parent node + arc identifies the prefix
                nxt = tries[t + 1].get(key)
                if nxt is None:
                    nxt = len(tries[t + 1])
                    tries[t + 1][key] = nxt
                    w = float(fst.weight[a])
                    if t == 0:
                        w += float(-np.log(pi[h0]))
                    arcs_by_t[t].append((node, nxt, int(fst.ilabel[a]), w))
                node, h = nxt, nh
        for t in range(1, T + 1):
            level_nodes.append(len(tries[t]))
        seq_arcs.append(arcs_by_t)
        seq_nodes.append(level_nodes)

    # merge: global state ids in time order; level T of sequence q is level 0 of sequence q+1
    # (one copy of q+1's start arcs per final state of q).
    state_base = []  # state_base[q][t] = global id of local node 0 at level t of sequence q
    n = 0
    for q in range(S):
        bases = []
        for t in range(T + 1):
            if t == 0 and q > 0:
                bases.append(state_base[q - 1][T])
                continue
            bases.append(n)
            n += seq_nodes[q][t]
        state_base.append(bases)
    num_states = n
    fw = [rng.uniform(0.1, 2.0, size=seq_nodes[q][T]) if final_weights else np.zeros(seq_nodes[q][T]) for q in range(S)]
    out = [[] for _ in range(num_states)]
    for q in range(S):
        for t in range(T):
            for (a, b, il, w) in seq_arcs[q][t]:
                dst = state_base[q][t + 1] + b
                if t == 0 and q > 0:
                    for f in range(seq_nodes[q - 1][T]):  # copies on every final state of q-1
                        out[state_base[q][0] + f].append((il, w + float(fw[q - 1][f]), dst))
                else:
                    out[state_base[q][t] + a].append((il, w, dst))
    arc_begin = np.zeros(num_states + 1, np.int32)
    ilabel, aw, nxt = [], [], []
    for i, lst in enumerate(out):
        arc_begin[i + 1] = arc_begin[i] + len(lst)
        for (il, w, d) in lst:
            ilabel.append(il)
            aw.append(w)
            nxt.append(d)
    final = np.full(num_states, np.inf, np.float32)
    final[state_base[S - 1][T]: state_base[S - 1][T] + seq_nodes[S - 1][T]] = fw[S - 1]
    return SupFst(float(weight), S, T, fst.num_pdfs, num_states, arc_begin, np.array(ilabel, np.int32),
                  np.array(aw, np.float32), np.array(nxt, np.int32), final)


def random_nnet_output(num_sequences, frames_per_sequence, num_pdfs, seed=1234, scale=1.0, zero=False):
    """y ~ N(0, scale^2) fp32, shape (T*S, P) with row = t*S + s (``torchain/functions.py:27-30``);
    ``zero=True`` gives the all-zero matrix the reference's tests use with p = 1/4
    (``src/chain-supervision-test.hpp:397-399``)."""
    rows = int(num_sequences) * int(frames_per_sequence)
    if zero:
        return np.zeros((rows, num_pdfs), np.float32)
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((rows, num_pdfs), dtype=np.float32) * np.float32(scale)).astype(np.float32)


# the five BASELINE.json configs as concrete shapes
CONFIGS = {
    "C1": dict(S=16, T=50, P=200, H=3, degree=2, leaky=1e-5),
    "C2": dict(S=64, T=150, P=4096, H=8192, degree=8, leaky=0.1, l2=5e-5),
    "C3": dict(S=256, T=150, P=4096, H=8192, degree=8, leaky=0.1, l2=5e-5),
    "C4": dict(S=2048, T=150, P=4096, H=8192, degree=8, leaky=0.1, l2=5e-5),
    "C5": dict(S=128, T=150, P=10240, H=8192, degree=7.5, leaky=0.1, l2=5e-5),
    # not in BASELINE.json: a den graph of the size Kaldi recipes produce for a few-thousand-leaf tree
    # (robustness / timing of the <JV=4> instantiation only)
    # phone-LM-structured graphs (phone_lm_den_fst): 42 phones, 2928 pdfs as the CHiME-5 tree of test/test.py:54
    "R1": dict(S=256, T=150, P=2928, H=None, leaky=0.1, l2=5e-5, phone_lm=dict(num_histories=640, branching=12)),
    "R2": dict(S=256, T=150, P=2928, H=None, leaky=0.1, l2=5e-5, phone_lm=dict(num_histories=1150, branching=12)),
    # R1 plus the LM's empty history: 7722 states, split by the library into 9681 chain-structured ones
    "R3": dict(S=256, T=150, P=2928, H=None, leaky=0.1, l2=5e-5,
               phone_lm=dict(num_histories=640, branching=12, unigram_fraction=0.03)),
    "X1": dict(S=256, T=150, P=2928, H=14000, degree=15, leaky=0.1, l2=5e-5),
    # beyond the on-chip layouts: the streamed (sequence-minor) kernels
    "X2": dict(S=256, T=150, P=4096, H=40000, degree=10, leaky=0.1, l2=5e-5),
    # ... and a phone-LM-structured graph of that size class: 24000 states, 312000 arcs, in-degrees up to the hundreds
    "R4": dict(S=256, T=150, P=2928, H=None, leaky=0.1, l2=5e-5, phone_lm=dict(num_histories=2000, branching=12)),
}


def config_den_fst(name):
    c = CONFIGS[name]
    if name == "C1":
        return left_to_right_den_fst(c["P"], seed=42)
    if "phone_lm" in c:
        return phone_lm_den_fst(num_pdfs=c["P"], seed=42, **c["phone_lm"])
    if name == "C5":
        # 61440 arcs over 8192 states: half the states have 8 out-arcs, half 7
        fa = random_den_fst(c["H"], 8, c["P"], seed=42)
        keep = np.ones(fa.src.shape[0], bool)
        keep[np.arange(c["H"] // 2) * 8 + 7] = False
        w = np.exp(-fa.weight.astype(np.float64))
        w[~keep] = 0
        tot = np.bincount(fa.src, weights=w, minlength=fa.num_states)
        w = w / tot[fa.src]
        return DenFst(fa.num_states, fa.src[keep], fa.dst[keep], fa.ilabel[keep],
                      (-np.log(w[keep])).astype(np.float32), fa.final, 0, fa.num_pdfs)
    return random_den_fst(c["H"], c["degree"], c["P"], seed=42)
