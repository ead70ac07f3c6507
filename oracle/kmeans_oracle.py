"""CPU restatement of the Lloyd k-means loop the reference reaches through scikit-learn (TEST INFRASTRUCTURE ONLY).

The reference initialises inducing points with ``sklearn.cluster.KMeans(n_clusters=K).fit(X).cluster_centers_``
(/root/reference/oak/model_utils.py:31-41, oak/utils.py:533-574).  scikit-learn (pinned in the reference's setup.py,
1.7.2 in this image) is a third-party dependency; the loop below restates its published single-run algorithm
``sklearn/cluster/_kmeans.py::_kmeans_single_lloyd`` with explicit seeds:

    labels_old = -1
    for i in range(max_iter):
        E-step: labels = argmin_k |x - c_k|^2 (first minimum); M-step: c_new = cluster means, empty clusters relocated
        centres <- c_new
        if labels == labels_old: strict convergence, stop
        if sum_k |c_new - c_old|^2 <= tol: stop
        labels_old = labels
    if not strictly converged: E-step once more against the final centres
    inertia = sum_i |x_i - c_{labels_i}|^2

Pinned by tests/test_oracle_kmeans.py against scikit-learn itself (KMeans(init=seeds, n_init=1, algorithm="lloyd")).
Only tests/ may import this module.
"""
import numpy as np


def _e_step(X, C):
    # direct form, chunked so the N x K x D temporary stays small
    N = X.shape[0]
    labels = np.empty(N, dtype=np.int32)
    mind = np.empty(N)
    step = max(1, (1 << 22) // max(1, C.shape[0] * X.shape[1]))
    for a in range(0, N, step):
        d2 = ((X[a:a + step, None, :] - C[None, :, :]) ** 2).sum(axis=2)
        labels[a:a + step] = np.argmin(d2, axis=1)          # first minimum wins, as scikit-learn
        mind[a:a + step] = d2[np.arange(d2.shape[0]), labels[a:a + step]]
    return labels, mind


def lloyd(X, init_centres, max_iter=300, tol=0.0):
    """Returns (centres, labels, inertia, n_iter).  ``tol`` is absolute (sklearn: tol * mean(var(X, axis=0)))."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    C = np.array(init_centres, dtype=np.float64)
    K = C.shape[0]
    labels_old = np.full(X.shape[0], -1, dtype=np.int32)
    strict = False
    n_iter = 0
    for i in range(max_iter):
        labels, mind = _e_step(X, C)
        sums = np.zeros_like(C)
        np.add.at(sums, labels, X)
        counts = np.bincount(labels, minlength=K).astype(np.int64)
        empty = np.flatnonzero(counts == 0)
        if empty.size:
            # sklearn _relocate_empty_clusters_dense: farthest points first (ties by lower index here)
            order = np.lexsort((np.arange(X.shape[0]), -mind))[: empty.size]
            for tgt, far in zip(empty, order):
                donor = labels[far]
                sums[donor] -= X[far]
                sums[tgt] = X[far]
                counts[tgt] = 1
                counts[donor] -= 1
        C_new = np.where(counts[:, None] > 0, sums / np.maximum(counts, 1)[:, None], C)
        shift_tot = float(((C_new - C) ** 2).sum())
        C = C_new
        n_iter = i + 1
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift_tot <= tol:
            labels_old = labels
            break
        labels_old = labels
    if not strict:
        labels, mind = _e_step(X, C)
    return C, labels, float(mind.sum()), n_iter


def sklearn_tolerance(X, tol=1e-4):
    """scikit-learn's absolute tolerance (``_tolerance``): tol * mean of the per-feature variances."""
    return float(np.mean(np.var(np.asarray(X, dtype=np.float64), axis=0)) * tol)
