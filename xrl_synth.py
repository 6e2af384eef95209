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
    # the SAME shape on a model that does not flatter bound pruning: every level-3 cluster owns a topic (the
    # supports are nested down the tree), a query draws 60 % of its features from the topic of ONE uniformly chosen cluster (routing is
    # query-dependent), small bias-row weights, every level's weights scaled so that the BEST margin of a query reaches 1 -- where
    # l3-hinge saturates -- for 5 % of the queries only (make_model_hard / make_queries_hard)
    "amazon-670k-hard": dict(N=490000, D=135000, L=670091, x_nnz=76, w_nnz=[20000, 8000, 3000, 800, 100], beam=10, hard=True),
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


def _write_layer(folder, d, W, parent, child_ids, prev_k, K, D, bias, only_topk, post_processor):
    lf = os.path.join(folder, "ranker", f"{d}.model")
    os.makedirs(lf, exist_ok=True)
    cptr = np.zeros(prev_k + 1, np.int64)
    np.cumsum(np.bincount(parent, minlength=prev_k), out=cptr[1:])
    Cm = smat.csc_matrix((np.ones(len(child_ids), np.float32), child_ids.astype(np.int32), cptr), shape=(K, prev_k))
    smat.save_npz(os.path.join(lf, "W.npz"), W, compressed=False)
    smat.save_npz(os.path.join(lf, "C.npz"), Cm, compressed=False)
    json.dump({"__meta__": {"class_fullname": "pecos.xmc.base###MLModel"}, "model": "MLModel",
               "bias": bias, "nr_labels": K, "nr_codes": prev_k, "nr_features": D,
               "pred_kwargs": {"only_topk": only_topk, "post_processor": post_processor}},
              open(os.path.join(lf, "param.json"), "w"), indent=True)


def _nested_supports(rng, cdf, relabel, par_ptr, par_idx, parent, per_col, share):
    """Per child column: ~share * per_col distinct features sampled from ITS PARENT's support (par_ptr / par_idx, parent[c]) plus
    fresh Zipf draws up to per_col distinct features in all -> (indptr, sorted indices)."""
    D = len(cdf)
    K = len(parent)
    n_sh = max(1, int(round(per_col * share)))
    m_sh = int(n_sh * 1.25) + 2
    m_fr = int((per_col - n_sh) * 1.5) + 2
    out_ptr = [np.zeros(1, np.int64)]
    out_idx = []
    base = 0
    step = max(1, (1 << 23) // (m_sh + m_fr))
    for c0 in range(0, K, step):
        c1 = min(K, c0 + step)
        n = c1 - c0
        par = parent[c0:c1]
        plen = (par_ptr[par + 1] - par_ptr[par]).astype(np.int64)
        pos = (rng.random((n, m_sh)) * plen[:, None]).astype(np.int64)
        ids = par_idx[(par_ptr[par][:, None] + pos).ravel()]
        col = np.repeat(np.arange(n, dtype=np.int64), m_sh)
        key = np.unique(col * D + ids)
        col = key // D
        cnt = np.bincount(col, minlength=n)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        # (the features are relabelled at random, so keeping the n_sh smallest ids of a column is a uniform choice)
        key = key[(np.arange(len(key)) - np.repeat(start, cnt)) < n_sh]
        ids2 = relabel[np.searchsorted(cdf, rng.random(n * m_fr)).clip(0, D - 1)]
        key = np.unique(np.concatenate([key, np.repeat(np.arange(n, dtype=np.int64), m_fr) * D + ids2]))
        col = key // D
        cnt = np.bincount(col, minlength=n)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        # cap at per_col: drop a random-ish subset (hash of the key), not the largest ids, so that shared features are not favoured
        over = np.maximum(cnt - per_col, 0)
        if over.any():
            h = (key.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(11)
            order = np.lexsort((h, col))
            rank = np.empty(len(key), np.int64); rank[order] = np.arange(len(key)) - np.repeat(start, cnt)
            key = key[rank < per_col]
            col = key // D
            cnt = np.bincount(col, minlength=n)
        out_ptr.append(base + np.cumsum(cnt)); base += int(cnt.sum())
        out_idx.append(key - col * D)
    return np.concatenate(out_ptr), np.concatenate(out_idx)


def _rank_of_feature(D, relabel):
    r = np.empty(D, np.int64); r[relabel] = np.arange(D)
    return r


def _idf_of_feature(D, relabel):
    return np.log(2.0 + _rank_of_feature(D, relabel))


def _informativeness(D, relabel, knee=5000.0):
    """Weight scale per feature: the most popular features carry almost no weight (what a regularised linear ranker learns for
    features every document holds), features past popularity rank `knee` full weight."""
    return np.minimum(1.0, np.log(1.0 + _rank_of_feature(D, relabel)) / np.log(knee)) ** 4


def make_queries_hard(N, D, x_nnz, topic_ptr, topic_idx, seed=1, relabel_seed=0, topic_share=0.6, return_topics=False):
    """Queries of the "hard" configs: every row picks ONE topic (a column of the given CSC support -- the level above the leaves) uniformly
    and draws ~topic_share of its features from that topic's support, the rest from the global Zipf popularity.  CSR f32, sorted unique
    indices, L2-normalised rows, ragged row lengths like make_queries."""
    rng = np.random.default_rng(seed)
    relabel = np.random.default_rng(relabel_seed).permutation(D)
    cdf = _zipf_cdf(D)
    n_topics = len(topic_ptr) - 1
    tgt = np.clip(rng.lognormal(np.log(x_nnz) - 0.125, 0.5, N).astype(np.int64), 1, min(D, 8 * x_nnz))
    topic = rng.integers(0, n_topics, N)
    keys = []
    for r0 in range(0, N, 1 << 16):
        r1 = min(N, r0 + (1 << 16))
        n = r1 - r0
        t = tgt[r0:r1]
        n_top = np.maximum(1, np.round(t * topic_share).astype(np.int64))
        tl = (topic_ptr[topic[r0:r1] + 1] - topic_ptr[topic[r0:r1]]).astype(np.int64)
        m = int(n_top.max())
        pos = (rng.random((n, m)) * tl[:, None]).astype(np.int64)
        ids = topic_idx[(topic_ptr[topic[r0:r1]][:, None] + pos)]
        use = np.arange(m)[None, :] < n_top[:, None]
        row = np.broadcast_to(np.arange(r0, r1, dtype=np.int64)[:, None], (n, m))
        key = np.unique(row[use] * D + ids[use])
        # Zipf draws for the rest, topped up until every row holds its target number of DISTINCT features
        for it in range(64):
            rows_k = key // D - r0
            have = np.bincount(rows_k, minlength=n)
            need = t - have
            short = np.nonzero(need > 0)[0]
            if len(short) == 0:
                break
            draws = np.ceil(need[short] * (1.2 + 0.4 * it)).astype(np.int64) + 1
            rr = np.repeat(short, draws)
            ids2 = relabel[np.searchsorted(cdf, rng.random(len(rr))).clip(0, D - 1)]
            new = np.setdiff1d(np.unique((rr + r0) * D + ids2), key, assume_unique=True)
            # keep at most `need` new features per row
            nr = new // D - r0
            cnt = np.bincount(nr, minlength=n)
            start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
            new = new[(np.arange(len(new)) - np.repeat(start, cnt)) < np.repeat(np.maximum(need, 0), cnt)]
            key = np.union1d(key, new)
        keys.append(key)
    key = np.concatenate(keys)
    row = key // D
    cnt = np.bincount(row, minlength=N)
    indptr = np.zeros(N + 1, np.int64); np.cumsum(cnt, out=indptr[1:])
    # tf x idf: |N(0,1)| + 0.05 term weight times log(2 + popularity rank) (Zipf popularity: idf grows with the log of the rank)
    val = ((np.abs(rng.standard_normal(len(key))) + 0.05) * _idf_of_feature(D, relabel)[key - row * D]).astype(np.float32)
    X = smat.csr_matrix((val, (key - row * D).astype(np.int32), indptr), shape=(N, D))
    nrm = np.sqrt(np.asarray(X.multiply(X).sum(axis=1)).ravel())
    nrm[nrm == 0] = 1.0
    X.data /= np.repeat(nrm, cnt).astype(np.float32)
    X.has_sorted_indices = True
    return (X, topic) if return_topics else X


def hard_query_topics(N, x_nnz, n_topics, seed=1):
    """The topic (cluster of the level above the leaves) every row of make_queries_hard(N, ..., seed) was drawn from."""
    rng = np.random.default_rng(seed)
    rng.lognormal(np.log(x_nnz) - 0.125, 0.5, N)
    return rng.integers(0, n_topics, N)


def make_model_hard(folder, D, L, w_nnz, x_nnz, bias=1.0, post_processor="l3-hinge", only_topk=20, seed=0, permute_leaf=True,
                    nr_splits=16, max_leaf_size=100, shape=None, share=(0.5, 0.5, 0.5, 0.5, 0.7), sat_quantile=0.95, calib_rows=384, bias_std=0.03):
    """A synthetic model with query-dependent routing and an unsaturated post-processor (CONFIGS["amazon-670k-hard"]).

    Supports are nested down the tree: a node's column takes `share` of its features from its parent's support and the rest from
    the global Zipf popularity, so a query drawn from the topic of one level-(T-2) cluster matches that cluster's whole ancestor
    path better than their siblings.  Weights ~ N(0.5, 0.6) (a matched feature raises the margin on average) times the feature's
    informativeness (the most popular features carry almost no weight); bias-row weights ~ N(0, bias_std) -- 0.03: with 0.1 the tail of the bias weights
    outranks the margins of unmatched columns (median 0.015) and the ten most common leaf parents hold 6.4 % of the final labels.  Every level is then scaled so that the LARGEST margin a calibration query reaches over ALL columns of the level
    is >= 1 for a fraction 1 - sat_quantile of the queries: P(margin >= 1) <= 5 % for any candidate at every level, i.e. l{p}-hinge
    saturates rarely and children do not tie with their parents' scores.  Returns (ks, topic_ptr, topic_idx) -- the supports of
    the level above the leaves, which make_queries_hard draws the queries from."""
    rng = np.random.default_rng(seed)
    ks = shape or tree_shape(L, nr_splits, max_leaf_size)
    assert len(w_nnz) == len(ks), (w_nnz, ks)
    T = len(ks)
    share = list(share)[-T:] if len(share) >= T else [share[0]] * (T - len(share)) + list(share)
    cdf = _zipf_cdf(D)
    relabel = rng.permutation(D)
    os.makedirs(os.path.join(folder, "ranker"), exist_ok=True)
    # ---- supports, top down
    sup = []
    prev_k = 1
    for d, (K, c) in enumerate(zip(ks, w_nnz)):
        parent = (np.arange(K, dtype=np.int64) * prev_k) // K
        if d == 0:
            ptr, idx = _draw_sorted_unique(rng, cdf, relabel, K, c)
        else:
            ptr, idx = _nested_supports(rng, cdf, relabel, sup[-1][0], sup[-1][1], parent, min(c, D), share[d])
        sup.append((ptr, idx, parent))
        prev_k = K
    topic_ptr, topic_idx = (sup[-2][0], sup[-2][1]) if T > 1 else (sup[-1][0], sup[-1][1])
    # ---- calibration queries (the generator the benchmark's queries come from, another seed)
    info = _informativeness(D, relabel)
    Xc = make_queries_hard(calib_rows, D, x_nnz, topic_ptr, topic_idx, seed=seed + 7919, relabel_seed=seed)
    prev_k = 1
    for d, (K, c) in enumerate(zip(ks, w_nnz)):
        ptr, idx, parent = sup[d]
        val = ((0.5 + 0.6 * rng.standard_normal(len(idx))) * info[idx]).astype(np.float32)
        Wf = smat.csc_matrix((val, idx.astype(np.int32), ptr), shape=(D, K))
        # largest margin per calibration query over all columns of the level, in column blocks
        best = np.full(Xc.shape[0], -np.inf)
        blk = max(1, (1 << 26) // max(1, Xc.shape[0]))
        for c0 in range(0, K, blk):
            best = np.maximum(best, np.asarray((Xc @ Wf[:, c0:c0 + blk]).max(axis=1).todense()).ravel())
        q = float(np.quantile(best, sat_quantile))
        scale = np.float32(1.0 / q) if q > 0 else np.float32(1.0)
        val *= scale
        if bias > 0:   # explicit bias row D at the end of (almost) every column, small weights
            has_b = rng.random(K) < 0.97
            cnt = np.diff(ptr)
            new_ptr = np.zeros(K + 1, np.int64)
            np.cumsum(cnt + has_b, out=new_ptr[1:])
            new_idx = np.empty(new_ptr[-1], np.int64); new_val = np.empty(new_ptr[-1], np.float32)
            dst = np.arange(len(idx)) + np.repeat(np.cumsum(np.concatenate([[0], has_b[:-1]])), cnt)
            new_idx[dst] = idx; new_val[dst] = val
            bpos = new_ptr[1:][has_b] - 1
            new_idx[bpos] = D; new_val[bpos] = (bias_std * rng.standard_normal(len(bpos))).astype(np.float32)
            ptr, idx, val = new_ptr, new_idx, new_val
        rows = D + 1 if bias > 0 else D
        W = smat.csc_matrix((val, idx.astype(np.int32), ptr), shape=(rows, K))
        child_ids = np.arange(K, dtype=np.int64)
        if permute_leaf and d == T - 1 and d > 0:
            child_ids = rng.permutation(K)
        _write_layer(folder, d, W, parent, child_ids, prev_k, K, D, bias, only_topk, post_processor)
        prev_k = K
    json.dump({"__meta__": {"class_fullname": "pecos.xmc.base###HierarchicalMLModel"},
               "model": "HierarchicalMLModel", "depth": T, "nr_features": D,
               "nr_codes": ks[-2] if T > 1 else 1, "nr_labels": ks[-1]},
              open(os.path.join(folder, "ranker", "param.json"), "w"), indent=True)
    json.dump({"__meta__": {"class_fullname": "pecos.xmc.xlinear.model###XLinearModel"}, "model": "XLinearModel"},
              open(os.path.join(folder, "param.json"), "w"), indent=True)
    return ks, topic_ptr, topic_idx


def make_config(name, folder, scale=1.0, seed=0, **kw):
    """Materialise CONFIGS[name] (optionally scaled down for tests): returns (ks, X, cfg)."""
    cfg = dict(CONFIGS[name])
    if scale != 1.0:
        cfg["N"] = max(8, int(cfg["N"] * scale))
        cfg["L"] = max(40, int(cfg["L"] * scale))
        ks = tree_shape(cfg["L"])
        cfg["w_nnz"] = cfg["w_nnz"][len(cfg["w_nnz"]) - len(ks):] if len(ks) <= len(cfg["w_nnz"]) else \
            [cfg["w_nnz"][0]] * (len(ks) - len(cfg["w_nnz"])) + cfg["w_nnz"]
    if cfg.get("hard"):
        ks, tp, ti = make_model_hard(folder, cfg["D"], cfg["L"], cfg["w_nnz"], cfg["x_nnz"], seed=seed, **kw)
        X = make_queries_hard(cfg["N"], cfg["D"], cfg["x_nnz"], tp, ti, seed=seed + 1, relabel_seed=seed)
        return ks, X, cfg
    ks = make_model(folder, cfg["D"], cfg["L"], cfg["w_nnz"], seed=seed, **kw)
    X = make_queries(cfg["N"], cfg["D"], cfg["x_nnz"], seed=seed + 1, relabel_seed=seed)
    return ks, X, cfg
