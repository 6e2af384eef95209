"""Synthetic XR-Linear models and query matrices in the reference's on-disk layout.

Used by tests/ and bench.py (there is no network for datasets or checkpoints).  The tree shapes
follow the reference's defaults (nr_splits=16, max_leaf_size=100; pecos/xmc/base.py:107-131,
pecos/utils/cluster_util.py:133-175): leaf clusters = 2^ceil(log2(L/100)), then /16 per level
until <=16, then a root.  The model folder is what ``XLinearModel.save`` writes
(pecos/xmc/xlinear/model.py:94-103, pecos/xmc/base.py:807-830,1371-1395):

    <folder>/param.json
    <folder>/ranker/param.json                 {model, depth, nr_features, nr_codes, nr_labels}
    <folder>/ranker/{d}.model/param.json       {model, bias, pred_kwargs{only_topk, post_processor}, ...}
    <folder>/ranker/{d}.model/W.npz            CSC f32 (D+1) x K_d, uncompressed npz
    <folder>/ranker/{d}.model/C.npz            CSC f32 K_d x K_{d-1}

Shapes: feature ids are drawn from a Zipf(1.1) popularity over D (through a
fixed random relabelling) for both W columns and X rows so that intersections are non-trivial.
"""
import json
import math
import os

import numpy as np
import scipy.sparse as smat

CONFIGS = {
    # name: (N, D, L, nnz_per_row, w_nnz_per_col top->leaf, beam)
    "toy": dict(N=64, D=64, L=40, x_nnz=8, w_nnz=[24, 12], beam=10),
    "eurlex-4k": dict(N=15000, D=5000, L=3956, x_nnz=240, w_nnz=[2000, 1000, 150], beam=10),
    "wiki10-31k": dict(N=14000, D=100000, L=30938, x_nnz=670, w_nnz=[20000, 8000, 3000, 300], beam=20),
    "amazon-670k": dict(N=490000, D=135000, L=670091, x_nnz=76, w_nnz=[20000, 8000, 3000, 800, 100], beam=10),
    # dense-input config (BASELINE.json configs[4]); N is per-bench adjustable
    "dense-768": dict(N=1000000, D=768, L=3000000, x_nnz=None, w_nnz=[768, 768, 768, 768, 256], beam=10),
}


def tree_shape(L, nr_splits=16, max_leaf_size=100):
    """[K_0, ..., K_{T-1}=L] under the reference's default hierarchical clustering."""
    if max_leaf_size >= L:
        return [L]
    depth = max(1, int(math.ceil(math.log2(L / max_leaf_size))))
    ks = [L, 1 << depth]
    while ks[-1] > nr_splits:
        ks.append((ks[-1] + nr_splits - 1) // nr_splits)
    return ks[::-1]


def _zipf_cdf(D, a=1.1):
    w = 1.0 / np.arange(1, D + 1, dtype=np.float64) ** a
    return np.cumsum(w / w.sum())


def _draw_sorted_unique(rng, cdf, relabel, n_cols, per_col, oversample=1.35):
    """per-column sets of ~per_col distinct feature ids -> (indptr, indices) with sorted columns."""
    D = len(cdf)
    per_col = min(per_col, D)
    if per_col > D // 8 or n_cols <= 1024:
        # dense-ish columns: exact sampling without replacement from the popularity weights
        w = np.diff(np.concatenate([[0.0], cdf]))
        cols = []
        for _ in range(n_cols):
            if per_col >= D:
                ids = np.arange(D)
            else:
                ids = rng.choice(D, size=per_col, replace=False, p=w)
            cols.append(np.sort(relabel[ids]))
        indptr = np.zeros(n_cols + 1, np.int64)
        np.cumsum([len(c) for c in cols], out=indptr[1:])
        return indptr, np.concatenate(cols).astype(np.int64)
    m = int(per_col * oversample)
    out_ptr = [np.zeros(1, np.int64)]
    out_idx = []
    base = 0
    step = max(1, (1 << 24) // m)
    for c0 in range(0, n_cols, step):
        c1 = min(n_cols, c0 + step)
        u = rng.random((c1 - c0) * m)
        ids = relabel[np.searchsorted(cdf, u).clip(0, D - 1)]
        key = np.repeat(np.arange(c1 - c0, dtype=np.int64), m) * D + ids
        key = np.unique(key)
        col = key // D
        # keep at most per_col ids per column (drop a random-ish tail deterministically)
        cnt = np.bincount(col, minlength=c1 - c0)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        rank = np.arange(len(key)) - np.repeat(start, cnt)
        keep = rank < per_col
        key = key[keep]; col = col[keep]
        cnt = np.bincount(col, minlength=c1 - c0)
        out_ptr.append(base + np.cumsum(cnt))
        base += int(cnt.sum())
        out_idx.append(key - col * D)
    return np.concatenate(out_ptr), np.concatenate(out_idx)


def make_model(folder, D, L, w_nnz, bias=1.0, post_processor="l3-hinge", only_topk=20,
               seed=0, permute_leaf=True, nr_splits=16, max_leaf_size=100, shape=None, prune=0.0):
    """Write a synthetic model folder; returns the list of layer sizes."""
    rng = np.random.default_rng(seed)
    ks = shape or tree_shape(L, nr_splits, max_leaf_size)
    assert len(w_nnz) == len(ks), (w_nnz, ks)
    cdf = _zipf_cdf(D)
    relabel = rng.permutation(D)
    os.makedirs(os.path.join(folder, "ranker"), exist_ok=True)
    prev_k = 1
    for d, (K, c) in enumerate(zip(ks, w_nnz)):
        lf = os.path.join(folder, "ranker", f"{d}.model")
        os.makedirs(lf, exist_ok=True)
        indptr, idx = _draw_sorted_unique(rng, cdf, relabel, K, c)
        val = rng.standard_normal(len(idx)).astype(np.float32)
        if bias > 0:  # explicit bias row D at the end of (almost) every column
            has_b = rng.random(K) < 0.97
            cnt = np.diff(indptr)
            new_ptr = np.zeros(K + 1, np.int64)
            np.cumsum(cnt + has_b, out=new_ptr[1:])
            new_idx = np.empty(new_ptr[-1], np.int64); new_val = np.empty(new_ptr[-1], np.float32)
            dst = np.arange(len(idx)) + np.repeat(np.cumsum(np.concatenate([[0], has_b[:-1]])), cnt)
            new_idx[dst] = idx; new_val[dst] = val
            bpos = new_ptr[1:][has_b] - 1
            new_idx[bpos] = D; new_val[bpos] = rng.standard_normal(len(bpos)).astype(np.float32)
            indptr, idx, val = new_ptr, new_idx, new_val
        rows = D + 1 if bias > 0 else D
        W = smat.csc_matrix((val, idx.astype(np.int32 if rows < 2**31 else np.int64), indptr), shape=(rows, K))
        # C: child j -> parent floor(j * prev_k / K) (contiguous); the leaf layer optionally through a
        # permutation (real k-means leaves have indices = argsort(codes), xmc/base.py:232-237)
        parent = (np.arange(K, dtype=np.int64) * prev_k) // K
        child_ids = np.arange(K, dtype=np.int64)
        if permute_leaf and d == len(ks) - 1 and d > 0:
            child_ids = rng.permutation(K)
        if prune > 0 and d == len(ks) - 1 and d > 0:
            keep = rng.random(K) >= prune
            parent, child_ids = parent[keep], child_ids[keep]
        cptr = np.zeros(prev_k + 1, np.int64)
        np.cumsum(np.bincount(parent, minlength=prev_k), out=cptr[1:])
        Cm = smat.csc_matrix((np.ones(len(child_ids), np.float32), child_ids.astype(np.int32), cptr), shape=(K, prev_k))
        smat.save_npz(os.path.join(lf, "W.npz"), W, compressed=False)
        smat.save_npz(os.path.join(lf, "C.npz"), Cm, compressed=False)
        json.dump({"__meta__": {"class_fullname": "pecos.xmc.base###MLModel"}, "model": "MLModel",
                   "bias": bias, "nr_labels": K, "nr_codes": prev_k, "nr_features": D,
                   "pred_kwargs": {"only_topk": only_topk, "post_processor": post_processor}},
                  open(os.path.join(lf, "param.json"), "w"), indent=True)
        prev_k = K
    json.dump({"__meta__": {"class_fullname": "pecos.xmc.base###HierarchicalMLModel"},
               "model": "HierarchicalMLModel", "depth": len(ks), "nr_features": D,
               "nr_codes": ks[-2] if len(ks) > 1 else 1, "nr_labels": ks[-1]},
              open(os.path.join(folder, "ranker", "param.json"), "w"), indent=True)
    json.dump({"__meta__": {"class_fullname": "pecos.xmc.xlinear.model###XLinearModel"}, "model": "XLinearModel"},
              open(os.path.join(folder, "param.json"), "w"), indent=True)
    return ks


def make_queries(N, D, x_nnz, seed=1, relabel_seed=0):
    """CSR f32, sorted unique indices, L2-normalised rows (examples/pecos-xrlinear-jmlr22/xrl_predict.py:143).
    x_nnz=None -> dense standard-normal rows (the dense-input config)."""
    rng = np.random.default_rng(seed)
    if x_nnz is None:
        X = rng.standard_normal((N, D), dtype=np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        return X
    relabel = np.random.default_rng(relabel_seed).permutation(D)
    cdf = _zipf_cdf(D)
    # per-row nnz ~ lognormal with MEAN x_nnz (real TF-IDF rows are ragged), at least 1.  Zipf draws repeat the popular features,
    # so rows are topped up with further draws until every row holds its target number of DISTINCT features (SURVEY.md 8d:
    # 76 / 240 / 670 per row; round 1 stopped after one pass and came out 12-38 % lighter).
    tgt = np.clip(rng.lognormal(np.log(x_nnz) - 0.125, 0.5, N).astype(np.int64), 1, min(D, 8 * x_nnz))
    work = np.zeros(0, np.int64)                   # keys (row * D + feature) of the rows still short of their target
    done = []                                      # keys of finished rows
    need = tgt.copy()
    for it in range(64):
        rows_short = np.nonzero(need > 0)[0]
        if len(rows_short) == 0:
            break
        draws = np.ceil(need[rows_short] * (1.3 + 0.5 * it)).astype(np.int64) + 1
        row = np.repeat(rows_short, draws)
        ids = relabel[np.searchsorted(cdf, rng.random(len(row))).clip(0, D - 1)]
        work = np.unique(np.concatenate([work, row * D + ids]))
        # keep at most tgt ids per row; which ones does not matter for the arithmetic contract, the count does
        row = work // D
        cnt = np.bincount(row, minlength=N)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        work = work[(np.arange(len(work)) - np.repeat(start, cnt)) < np.repeat(tgt, cnt)]
        row = work // D
        need = np.where(need > 0, tgt - np.bincount(row, minlength=N), 0)
        fin = need[row] <= 0
        done.append(work[fin]); work = work[~fin]
    key = np.sort(np.concatenate(done + [work]))
    row = key // D
    cnt = np.bincount(row, minlength=N)
    indptr = np.zeros(N + 1, np.int64); np.cumsum(cnt, out=indptr[1:])
    val = np.abs(rng.standard_normal(len(key))).astype(np.float32) + 0.05
    X = smat.csr_matrix((val, (key - row * D).astype(np.int32), indptr), shape=(N, D))
    nrm = np.sqrt(np.asarray(X.multiply(X).sum(axis=1)).ravel())
    nrm[nrm == 0] = 1.0
    X.data /= np.repeat(nrm, cnt).astype(np.float32)
    X.has_sorted_indices = True
    return X


def make_config(name, folder, scale=1.0, seed=0, **kw):
    """Materialise CONFIGS[name] (optionally scaled down for tests): returns (ks, X, cfg)."""
    cfg = dict(CONFIGS[name])
    if scale != 1.0:
        cfg["N"] = max(8, int(cfg["N"] * scale))
        cfg["L"] = max(40, int(cfg["L"] * scale))
        ks = tree_shape(cfg["L"])
        cfg["w_nnz"] = cfg["w_nnz"][len(cfg["w_nnz"]) - len(ks):] if len(ks) <= len(cfg["w_nnz"]) else \
            [cfg["w_nnz"][0]] * (len(ks) - len(cfg["w_nnz"])) + cfg["w_nnz"]
    ks = make_model(folder, cfg["D"], cfg["L"], cfg["w_nnz"], seed=seed, **kw)
    X = make_queries(cfg["N"], cfg["D"], cfg["x_nnz"], seed=seed + 1, relabel_seed=seed)
    return ks, X, cfg
