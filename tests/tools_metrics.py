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
