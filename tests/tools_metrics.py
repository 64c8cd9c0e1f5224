import numpy as np


def matched_iou(a, b, K):
    """Mean IoU between two clusterings after greedily matching cluster ids by overlap (K-means ids are arbitrary
    up to a permutation when features differ in the last bits)."""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    la, lb = np.unique(a), np.unique(b)
    conf = np.array([[np.sum((a == x) & (b == y)) for y in lb] for x in la])
    used, ious, agree = set(), [], 0
    for i in np.argsort(-conf.max(axis=1)):
        order = np.argsort(-conf[i])
        j = next((j for j in order if j not in used), None)
        if j is None:
            ious.append(0.0)
            continue
        used.add(j)
        inter = conf[i, j]
        union = np.sum(a == la[i]) + np.sum(b == lb[j]) - inter
        ious.append(inter / union)
        agree += inter
    return float(np.mean(ious)), float(agree / a.size)


def clustering_objective(flat, labels):
    """K-means objective of a labelling on given features: sum over tokens of the squared distance to the mean of the token's
    cluster (float64).  Two labelings of the same features with (nearly) the same objective are equally good clusterings: which
    one best-of-n_init K-means returns is decided by rounding."""
    X = np.asarray(flat, dtype=np.float64)
    lab = np.asarray(labels).reshape(-1)
    tot = 0.0
    for l in np.unique(lab):
        m = X[lab == l]
        tot += float(((m - m.mean(axis=0)) ** 2).sum())
    return tot


# ---- the bar: a rate with an interval, not "all but two" --------------------------------------------------------------------------
# What is known about the problem (profiles/r03_mask_knee_study_1e-6.txt: white noise of normalised rms 1e-6 -- another fp32
# summation order -- on the REFERENCE's own fp32 taps, then the reference's own sklearn call): 46 of 48 runs keep the reference's
# masks (IoU >= 0.99), i.e. p0 = 0.958 per window.  An fp32-accurate device evaluation is one more such summation order, so the
# number of windows that keep the reference's masks must be compatible with Binomial(n, P0); the test rejects at the 1 % level.
P0 = 46.0 / 48.0


def binom_min_successes(n, p0=P0, alpha=0.01):
    """Smallest k with P(K <= k - 1 | Binomial(n, p0)) < alpha <= P(K <= k): observing fewer than k successes rejects p >= p0."""
    from math import comb
    cdf = 0.0
    for k in range(n + 1):
        cdf += comb(n, k) * p0 ** k * (1.0 - p0) ** (n - k)
        if cdf >= alpha:
            return k
    return n


def wilson(k, n, z=1.96):
    p = k / n
    d = 1.0 + z * z / n
    c = (p + z * z / (2 * n)) / d
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / d
    return max(0.0, c - h), min(1.0, c + h)
