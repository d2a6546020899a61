"""An INDEPENDENT derivation of what the oracle computes — not another forward-backward: for chains of T <= 7 windows all
4^T state paths of the model of SURVEY.md Appendix A are enumerated in 50-digit arithmetic (mpmath),

    w(path) = start[s_0]·e_{s_0}(x_0) · prod_{t>=1} T_t(s_{t-1}, s_t) · e_{s_t}(x_t | x_{t-1}, alpha[s_{t-1}][s_t], beta_t)

and from the definitions
    log-likelihood            log sum_paths w                                   (hmm.c:428: sum of log scale)
    posterior of window t     sum_{paths, s_t = s} w·M[s_{T-1}][End], normalised (hmm.c:671-685)
    pair count (t, t+1)       sum_{paths, s_t = pre, s_{t+1} = s} w·M[s_{T-1}][End] / (Z · terminationProb)      (hmm.c:563-614)
    estimator sums            the updates of hmm_utils.c:812-839, 1027-1034, 2010-2015 applied to those counts, t = 1..T-2
with beta_t, the emission densities and the conditional transition written here from Appendix A.2-A.4 in mpmath.
The oracle's log-likelihood, posteriors and statistics must agree to 1e-12 relative (double rounding of ~10^2 operations).
The reference itself cannot be built here: this pins the oracle's MATH, the fixtures pin its plumbing."""
import itertools

import mpmath as mp
import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import synth
from oracle_py import Oracle

mp.mp.dps = 50
MAXC = 16
PI = mp.mpf("3.14159")           # common.h:15 (sic)


def _int(x):                      # common.c:142-148: min / max take int arguments: truncation toward zero
    return int(x)


def beta_of(store, c, t, adjust, min_frac):
    """Appendix A.2 (hmm.c:301-316)."""
    if not adjust:
        return mp.mpf(1)
    s, e, W, L = int(store.chunk_s[c]), int(store.chunk_e[c]), store.window_len, store.avg_alignment_len
    ctg_len = int(store.chunk_ctg_len[c])
    mid = min(_int(s + float(W) * (t + 0.5)), _int((s + float(W) * t + e) / 2))
    lo = max(mid - L + 1, _int(-(1 - min_frac) * L))
    hi = min(mid, _int(ctg_len - min_frac * L))
    if L == 0:
        return mp.mpf("0.25")
    b = mp.mpf(hi - lo) / L
    return b if b > mp.mpf("0.25") else mp.mpf("0.25")


class Params:
    def __init__(self, vec, R, K, model_type, alpha):
        v = np.asarray(vec, dtype=np.float64).reshape(R, -1)
        self.R, self.K, self.model_type = R, K, model_type
        self.trans = [[[mp.mpf(float(x)) for x in row] for row in v[r, :25].reshape(5, 5)] for r in range(R)]
        self.lam = [mp.mpf(float(v[r, 25])) for r in range(R)]
        self.trunc = [mp.mpf(float(v[r, 26])) for r in range(R)]
        o = 27
        self.mean, self.var, self.weight = [], [], []
        for dst in (self.mean, self.var, self.weight):
            for r in range(R):
                dst.append([[mp.mpf(float(x)) for x in row] for row in v[r, o:o + 4 * MAXC].reshape(4, MAXC)])
            o += 4 * MAXC
        self.alpha = [[mp.mpf(float(a)) for a in row] for row in alpha]
        self.ncomp = [1, 1, 1, K]

    def comp_probs(self, r, s, x, px, alpha, beta):
        """p_c of Appendix A.3 (hmm_utils.c:768-793)."""
        out = []
        for c in range(self.ncomp[s]):
            mean = ((1 - alpha) * self.mean[r][s][c] + alpha * px) * beta
            var = self.var[r][s][c] * beta
            p = self.weight[r][s][c] / mp.sqrt(var * 2 * PI) * mp.e ** (mp.mpf("-0.5") * (x - mean) ** 2 / var)
            out.append(p if p >= mp.mpf("1e-40") else mp.mpf("1e-40"))
        return out

    def emit(self, r, s, x, px, alpha, beta):
        if s == 0 and self.model_type == N.HF_MODEL_TRUNC_EXP_GAUSSIAN:     # hmm_utils.c:941-947
            lam, b = self.lam[r] / beta, beta * self.trunc[r]
            if self.trunc[r] < x:
                return mp.mpf(0)
            return lam * mp.e ** (-lam * x) / (1 - mp.e ** (-lam * b))
        return mp.fsum(self.comp_probs(r, s, x, px, alpha, beta))

    def tcond(self, r, pre, s, cov, mapq, clip, max_mapq, min_mapq, min_clip=1.0):
        """Appendix A.4 (hmm_utils.c:2229-2292)."""
        ratio_m, ratio_c = mapq / (0.1 + cov), clip / (0.1 + cov)
        valid = [True, not (ratio_m > max_mapq), True, not (ratio_m < min_mapq), not (ratio_c < min_clip)]
        tot = mp.fsum(self.trans[r][pre][j] for j in range(5) if valid[j])
        return self.trans[r][pre][s] / tot if valid[s] else mp.mpf(0)


def brute_force(store, c, P, adjust, min_frac, max_mapq, min_mapq):
    """LL, posteriors [T][4], and the estimator sums of one chunk, from the definitions."""
    t0, T = int(store.chunk_off[c]), int(store.chunk_off[c + 1] - store.chunk_off[c])
    x = [mp.mpf(int(store.cov[t0 + t]) & 0xff) for t in range(T)]
    reg = [int(store.annot[t0 + t] >> np.uint64(58)) for t in range(T)]
    beta = [beta_of(store, c, t, adjust, min_frac) for t in range(T)]
    # step weights A[t][pre][s], t >= 1 (hmm.c:380-406), and the first column (hmm.c:338-352)
    first = [P.trans[reg[0]][4][s] * P.emit(reg[0], s, x[0], mp.mpf(0), mp.mpf(0), beta[0]) for s in range(4)]
    A = [None]
    for t in range(1, T):
        cov, mapq, clip = float(int(store.cov[t0 + t])), float(int(store.mapq[t0 + t])), float(int(store.clip[t0 + t]))
        At = [[None] * 4 for _ in range(4)]
        for pre in range(4):
            for s in range(4):
                tp = mp.mpf(1) / 5 if reg[t] != reg[t - 1] else P.tcond(reg[t], pre, s, cov, mapq, clip, max_mapq, min_mapq)
                At[pre][s] = tp * P.emit(reg[t], s, x[t], x[t - 1], P.alpha[pre][s], beta[t])
        A.append(At)
    end = [P.trans[reg[T - 1]][s][4] for s in range(4)]
    Z = mp.mpf(0)
    post = [[mp.mpf(0)] * 4 for _ in range(T)]
    pair = [[[mp.mpf(0)] * 4 for _ in range(4)] for _ in range(T)]      # pair[t][pre][s]: windows (t, t+1)
    for path in itertools.product(range(4), repeat=T):
        w = first[path[0]]
        for t in range(1, T):
            w = w * A[t][path[t - 1]][path[t]]
            if w == 0:
                break
        if w == 0:
            continue
        Z += w
        we = w * end[path[-1]]
        for t in range(T):
            post[t][path[t]] += we
        for t in range(T - 1):
            pair[t][path[t]][path[t + 1]] += we
    term = mp.mpf("1e-4")                                                 # hmm_utils.c:2112
    K = P.K
    stride = 24 * K + 16
    stats = [mp.mpf(0)] * (1 + P.R * stride)
    stats[0] = mp.log(Z)
    te = P.model_type == N.HF_MODEL_TRUNC_EXP_GAUSSIAN
    for t in range(1, T - 1):                                             # pairs (t, t+1), t = 1..T-2 (hmm.c:638-642)
        r = reg[t + 1]
        base = 1 + r * stride
        for s in range(4):
            for pre in range(4):
                cnt = pair[t][pre][s] / Z / term
                stats[base + 24 * K + pre * 4 + s] += cnt
                if s == 0 and te:
                    stats[base + 0] += cnt * x[t + 1]
                    stats[base + K] += cnt
                    continue
                al = P.alpha[pre][s]
                x_adj = (x[t + 1] - al * x[t]) / (1 - al)
                pc = P.comp_probs(r, s, x[t + 1], x[t], al, beta[t + 1])
                tot = mp.fsum(pc)
                for cc in range(P.ncomp[s]):
                    w = cnt * pc[cc] / tot
                    z = (x_adj - P.mean[r][s][cc]) * (1 - al)
                    stats[base + ((s * 3 + 0) * 2 + 0) * K + cc] += w * x_adj
                    stats[base + ((s * 3 + 0) * 2 + 1) * K + cc] += w
                    stats[base + ((s * 3 + 1) * 2 + 0) * K + cc] += w * z * z
                    stats[base + ((s * 3 + 1) * 2 + 1) * K + cc] += w
                    stats[base + ((s * 3 + 2) * 2 + 0) * K + cc] += w
                    for c2 in range(P.ncomp[s]):
                        stats[base + ((s * 3 + 2) * 2 + 1) * K + c2] += w
    posterior = [[p / mp.fsum(row) for p in row] for row in post]
    return stats, posterior


def _tiny_store(rng, lengths, regions, window_len=1000, avg_len=2500):
    """A few contigs of a handful of windows each (one chunk per contig), random coverage / mapq / clip, region switches."""
    store = synth.synthesize([n * window_len for n in lengths], window_len, 10 ** 9, regions, seed=int(rng.integers(1 << 30)),
                             avg_alignment_len=avg_len)
    n = store.n_windows
    store.cov = rng.integers(0, 70, size=n).astype(np.uint16)
    store.mapq = np.where(rng.random(n) < 0.5, store.cov, (store.cov * rng.random(n)).astype(np.uint16)).astype(np.uint16)
    store.clip = np.where(rng.random(n) < 0.2, store.cov, 0).astype(np.uint16)
    if len(regions) > 1:
        reg = rng.integers(0, len(regions), size=n).astype(np.uint64)
        store.annot = (store.annot & np.uint64((1 << 58) - 1)) | (reg << np.uint64(58))
    return store


@pytest.mark.parametrize("model_type", [N.HF_MODEL_TRUNC_EXP_GAUSSIAN, N.HF_MODEL_GAUSSIAN])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_against_path_enumeration(model_type, seed):
    rng = np.random.default_rng(100 + seed)
    regions = [20] if seed == 0 else [20, 31]
    alpha = synth.HIFI_ALPHA if seed != 2 else np.zeros((4, 4))
    adjust = seed != 1
    K = 2 + seed
    store = _tiny_store(rng, [7, 5, 3, 6], regions)
    orc = Oracle(store, model_type, K, alpha, max_mapq=0.25, min_mapq=0.75, adjust=adjust, min_read_frac=0.9, threads=1)
    try:
        # perturb the initial model so that nothing is symmetric: a few EM-like nudges of means / variances / transitions
        v = orc.param_vector().reshape(len(regions), -1)
        v[:, 27:27 + 4 * MAXC] *= rng.uniform(0.8, 1.2, size=(len(regions), 4 * MAXC))
        v[:, 27 + 4 * MAXC:27 + 8 * MAXC] *= rng.uniform(0.8, 1.5, size=(len(regions), 4 * MAXC))
        for r in range(len(regions)):
            t = v[r, :25].reshape(5, 5)
            t[:4, :4] *= rng.uniform(0.5, 2.0, size=(4, 4))
            t[:4, :4] *= ((1 - 1e-4) / t[:4, :4].sum(axis=1))[:, None]
        orc.set_param_vector(v.ravel())
        assert orc.run_iteration() == 0
        got = orc.stats_vector(K)
        f, b, sc = orc.forward_backward()
        P = Params(orc.param_vector(), len(regions), K, model_type, alpha)
        want = [mp.mpf(0)] * got.size
        posts = []
        for c in range(store.n_chunks):
            st, post = brute_force(store, c, P, adjust, 0.9, 0.25, 0.75)
            want = [a + b_ for a, b_ in zip(want, st)]
            posts.extend(post)
        want = np.array([float(w) for w in want])
        # log-likelihood and every statistic
        assert abs(got[0] - want[0]) <= 1e-12 * abs(want[0]), (got[0], want[0])
        scale = np.maximum(np.abs(want), 1e-9 * np.abs(want).max())
        bad = np.abs(got - want) > 1e-11 * scale
        assert not bad.any(), (np.flatnonzero(bad)[:8], got[bad][:8], want[bad][:8])
        assert (want[1:] != 0).sum() > 16 * len(regions)
        # posteriors (hmm.c:671-685) and labels
        post = f * b * sc[:, None]
        post /= post.sum(axis=1, keepdims=True)
        wp = np.array([[float(p) for p in row] for row in posts])
        assert np.allclose(post, wp, rtol=1e-11, atol=1e-300)
        top2 = np.sort(wp, axis=1)[:, -2:]
        clear = top2[:, 1] - top2[:, 0] > 1e-9
        assert np.array_equal(orc.labels()[clear], wp.argmax(axis=1)[clear].astype(np.int8))
    finally:
        orc.close()
