"""GPU (-m gpu): randomized parity fuzz.  Irregular trees (empty parents, pruned children, permuted child
order, chunks wider than a tile), ragged queries (empty rows, explicit zeros, one very long row), random
beam / top-k / post-processor, sparse and dense X -- every case bit-exact against the oracle (which is
pinned on the real reference)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import assert_same_topk

pytestmark = pytest.mark.gpu


def random_model(folder, rng, D, depth, bias):
    os.makedirs(os.path.join(folder, "ranker"), exist_ok=True)
    prev_k = 1
    sizes = []
    for d in range(depth):
        K = int(rng.integers(max(2, prev_k), max(3, prev_k * int(rng.integers(2, 40)))))
        if d == depth - 1 and rng.random() < 0.3:
            K = int(rng.integers(150, 400))          # wide chunks -> column tiles
        sizes.append(K)
        lf = os.path.join(folder, "ranker", f"{d}.model"); os.makedirs(lf, exist_ok=True)
        # children -> parents: random, some parents empty; child order inside a parent random; some pruned
        parent = rng.integers(0, prev_k, K)
        if prev_k > 2:
            parent[parent == rng.integers(0, prev_k)] = (rng.integers(0, prev_k))   # likely empties one parent
        keep = rng.random(K) >= (0.15 if (d > 0 and rng.random() < 0.4) else 0.0)
        order = rng.permutation(K) if rng.random() < 0.7 else np.arange(K)
        order = order[keep[order]]
        order = order[np.argsort(parent[order], kind="stable")]
        cptr = np.zeros(prev_k + 1, np.int64); np.cumsum(np.bincount(parent[order], minlength=prev_k), out=cptr[1:])
        C = smat.csc_matrix((np.ones(len(order), np.float32), order.astype(np.int32), cptr), shape=(K, prev_k))
        rows = D + 1 if bias > 0 else D
        dens = rng.choice([0.02, 0.1, 0.4])
        W = smat.random(rows, K, density=dens, format="csc", dtype=np.float32, random_state=int(rng.integers(1 << 30)),
                        data_rvs=lambda n: rng.standard_normal(n).astype(np.float32))
        W.sort_indices()
        if rng.random() < 0.3:
            W.data[rng.random(W.nnz) < 0.05] = 0.0   # explicit zero weights
        smat.save_npz(os.path.join(lf, "W.npz"), W, compressed=False)
        smat.save_npz(os.path.join(lf, "C.npz"), C, compressed=False)
        json.dump({"model": "MLModel", "bias": bias, "pred_kwargs": {"only_topk": int(rng.integers(1, 30)),
                   "post_processor": str(rng.choice(["l3-hinge", "noop", "log-l2-hinge", "l1-hinge"]))}},
                  open(os.path.join(lf, "param.json"), "w"))
        prev_k = K
    json.dump({"model": "HierarchicalMLModel", "depth": depth}, open(os.path.join(folder, "ranker", "param.json"), "w"))
    json.dump({"model": "XLinearModel"}, open(os.path.join(folder, "param.json"), "w"))
    return sizes


def random_queries(rng, N, D):
    dens = rng.choice([0.01, 0.08, 0.5])
    X = smat.random(N, D, density=dens, format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
    X = X.tolil()
    X[0, :] = 0
    if N > 2:
        X[1, :] = rng.standard_normal(D).astype(np.float32)      # one fully dense row (forces queue overflow drains)
    X = X.tocsr().astype(np.float32)
    if X.nnz:
        X.data[rng.random(X.nnz) < 0.03] = 0.0                   # explicit zeros stay stored
    X.sort_indices()
    return X


SEED0 = int(os.environ.get("XRL_FUZZ_SEED0", "0"))      # XRL_FUZZ_SEED0=<n>: another 48 random cases (deeper fuzz runs; default: the fixed set)


@pytest.mark.parametrize("seed", range(48))
def test_fuzz(seed, tmp_path, oracle_mod):
    from pecos_amd import XLinearModel, clib
    seed = SEED0 + seed
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([37, 300, 2500]))
    depth = int(rng.integers(1, 5))
    bias = float(rng.choice([1.0, -1.0, 0.5]))
    folder = str(tmp_path / "m")
    sizes = random_model(folder, rng, D, depth, bias)
    # row lookup structure, rotated over the models: 32-feature rank-bitmap words, the bucket table + binary search
    # that layers too large for bitmaps fall back to, 64-feature words that carry the first row's extent (sparse tiles)
    mode = ("bucket", "bitmap", "bitmap64")[seed % 3]
    os.environ["XRL_LOOKUP"] = mode
    try:
        m = XLinearModel.load(folder)
    finally:
        os.environ.pop("XRL_LOOKUP", None)
    assert clib.xlinear_get_int_attr(m.model.model_chain, "nr_bucket_layers") == (depth if mode == "bucket" else 0)
    assert clib.xlinear_get_int_attr(m.model.model_chain, "nr_bitmap64_layers") == (depth if mode == "bitmap64" else 0)
    om = oracle_mod.OracleModel.load(folder)
    X = random_queries(rng, int(rng.integers(1, 70)), D)
    for trial in range(4):
        beam = int(rng.choice([1, 2, 5, 10, 33, 70, 200]))
        topk = int(rng.choice([1, 3, 10, 64, 65, 150]))
        pp = rng.choice([None, "noop", "l2-hinge", "log-l4-hinge", "l3-hinge"])
        kw = dict(beam_size=beam, only_topk=topk)
        if pp is not None:
            kw["post_processor"] = str(pp)
        # layers that carry the dense row format: fused kernel K1Q (when the beam's candidates fit its registers) / tile-format kernels
        clib.set_option(m.model.model_chain, "dense_layers", (2, 0, 1, 0)[trial])
        clib.set_option(m.model.model_chain, "k1g_min_items", 1 if trial == 0 else 16)   # dense X: tiled SGEMM forced / by batch size
        clib.set_option(m.model.model_chain, "k1g_variant", int(rng.integers(0, 2)))      # its alternative tile shapes
        clib.set_option(m.model.model_chain, "k1_group", int(rng.choice([0, 0, 1, 4, 16, 64])))
        clib.set_option(m.model.model_chain, "presence", int(rng.choice([0, 1, 2, 2])))           # K1Q presence words: never / unstaged layers / always
        clib.set_option(m.model.model_chain, "sort_min_tiles", int(rng.choice([0, 1, 1])))       # tile-format layers: items in natural order / tile-sorted
        clib.set_option(m.model.model_chain, "k2_big_min_k", int(rng.choice([0, 0, 1, 40])))      # the segmented-sort K2 that serves k > 20 480, forced on small k
        clib.set_option(m.model.model_chain, "tile_rows", (2, 0, 1, 2)[trial])                    # tile-format layers, sparse X: K1T (densely held tile rows) in every launch / never / query-order launches
        # K1Q's sorted launch (queries counting-sorted by the best parent of their beam, XCD-contiguous): forced on these small batches / off
        qs = int(rng.choice([0, 1, 1]))
        clib.set_option(m.model.model_chain, "qsort", qs)
        clib.set_option(m.model.model_chain, "qsort_min_rows", 1); clib.set_option(m.model.model_chain, "qsort_min_parents", 2)
        os.environ["XRL_K1Q_FUSE01"] = str(int(rng.choice([0, 1, 2])))                          # levels 0 + 1 separately / in one feature walk, a load per level / one load (merged rows)
        clib.set_option(m.model.model_chain, "prune", int(rng.choice([0, 1, 1])))                  # exact bound pruning on / off: same bits
        # one or two row batches in flight (two streams), whole or ragged batches
        clib.set_option(m.model.model_chain, "overlap_min_rows", int(rng.choice([0, 2])))
        clib.set_option(m.model.model_chain, "max_batch_rows", int(rng.choice([0, 0, 7, 32])))
        for Xq in (X, np.ascontiguousarray(X.toarray())):
            a = m.predict(Xq, **kw)
            b = om.predict(Xq, **kw)
            assert a.shape == b.shape
            assert_same_topk(a, b, exact_scores=True, what=f"seed={seed} sizes={sizes} D={D} bias={bias} {kw} dense={not smat.issparse(Xq)}")
    clib.set_option(m.model.model_chain, "k1_group", 0)
    clib.set_option(m.model.model_chain, "k1g_min_items", 16)
    clib.set_option(m.model.model_chain, "k1g_variant", 0)
    clib.set_option(m.model.model_chain, "dense_layers", 1)
    clib.set_option(m.model.model_chain, "sort_min_tiles", 0)
    clib.set_option(m.model.model_chain, "tile_rows", 1)
    clib.set_option(m.model.model_chain, "k2_big_min_k", 0)
    clib.set_option(m.model.model_chain, "presence", 1)
    clib.set_option(m.model.model_chain, "prune", 1)
    os.environ.pop("XRL_K1Q_FUSE01", None)
    clib.set_option(m.model.model_chain, "max_batch_rows", 0)
    # model defaults (per-layer only_topk / post-processor from param.json)
    assert_same_topk(m.predict(X), om.predict(X), exact_scores=True, what=f"seed={seed} defaults")
