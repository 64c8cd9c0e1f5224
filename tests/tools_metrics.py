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
