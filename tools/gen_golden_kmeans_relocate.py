"""tests/golden/kmeans_relocate.npz: scikit-learn's own KMeans(init=ndarray, n_init=1) on data whose initial centres leave
clusters EMPTY, so that _relocate_empty_clusters_dense (cluster/_k_means_common.pyx:167-211) runs -- the reference reaches
that code through KMeans(...).fit (feature_extraction.py:562-570) whenever a cluster loses all members.  Build-container only."""
import os
import sys

import numpy as np
from sklearn.cluster import KMeans

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_case(seed, n_per=200, C=16, far=2):
    g = np.random.Generator(np.random.PCG64(seed))
    blobs = g.standard_normal((3, C)) * 0.5
    X = np.concatenate([b + 0.05 * g.standard_normal((n_per, C)) for b in blobs]).astype(np.float16)
    init = np.concatenate([blobs, 5.0 + g.standard_normal((far, C))]).astype(np.float64)     # `far` centres nobody is close to
    return X, init


def main():
    rec = {}
    for name, seed, far in (("a", 3, 2), ("b", 4, 1), ("c", 5, 3)):
        X, init = make_case(seed, far=far)
        km = KMeans(n_clusters=init.shape[0], init=init.copy(), n_init=1).fit(X)
        rec[f"{name}_X"] = X
        rec[f"{name}_init"] = init
        rec[f"{name}_labels"] = km.labels_.astype(np.int32)
        rec[f"{name}_centers"] = km.cluster_centers_
        rec[f"{name}_n_iter"] = np.int32(km.n_iter_)
        print(name, "n_iter", km.n_iter_, "cluster sizes", np.bincount(km.labels_, minlength=init.shape[0]))
    path = os.path.join(ROOT, "tests", "golden", "kmeans_relocate.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    sys.exit(main())
