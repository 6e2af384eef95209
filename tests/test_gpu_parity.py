"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against
 (1) the committed golden fixtures made by the real reference (tests/golden/),
 (2) the CPU oracle (and oracle/_ref when present) on seeded synthetic models,
 (3) size-independent properties at BASELINE.json's full sizes.
Bar (BASELINE.json): label indices and their order bit-exact; scores within 1e-5 relative fp32
(bit-exact for noop / l{p}-hinge / log-l{p}-hinge; sigmoid variants differ from glibc's expf in
the last ulp)."""
import os

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import GOLDEN, assert_same_topk, load_raw_csr, load_X

pytestmark = pytest.mark.gpu

EXACT_PP = lambda pp: pp is None or "sigmoid" not in pp


@pytest.fixture(scope="module")
def clib():
    from pecos_amd import clib
    assert clib.device_count() > 0, "no GPU visible"
    return clib


@pytest.fixture(scope="module")
def XLM():
    from pecos_amd import XLinearModel
    return XLinearModel


def test_reference_cli_goldens(manifest, XLM):
    # test/pecos/xmc/xlinear/test_xlinear.py:314-640 (abs=1e-6 like the reference's own assertion)
    Xt = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"))
    for c in manifest["cli"]:
        m = XLM.load(os.path.join(GOLDEN, "models", c["model"]))
        P = m.predict(Xt, **c["kwargs"])
        G = smat.load_npz(os.path.join(GOLDEN, "ref_fixtures", c["golden"]))
        assert np.allclose(P.toarray(), G.toarray(), atol=1e-6), c


def test_reference_toy_matrix(manifest, XLM):
    # test_xlinear.py:106-245: 3 trained models x 11 post-processors x {batch, realtime} x {sparse, dense}
    models = {}
    for c in manifest["toy"]:
        if c["model"] not in models:
            models[c["model"]] = XLM.load(os.path.join(GOLDEN, "models", c["model"]))
        m = models[c["model"]]
        X = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"), c["x"])
        G = smat.load_npz(os.path.join(GOLDEN, "preds", c["pred"])).toarray()
        kw = dict(beam_size=c["beam_size"], post_processor=c["post_processor"])
        assert np.allclose(m.predict(X, **kw).toarray(), G, atol=1e-6), c
        for i in range(X.shape[0]):  # realtime mode
            q = X[[i], :] if c["x"] == "sparse" else np.ascontiguousarray(X[[i], :])
            if c["x"] == "sparse":
                q.sort_indices()
            assert np.allclose(m.predict(q, **kw).toarray(), G[[i]], atol=1e-6), (c, i)


def test_synthetic_goldens_bit_exact(manifest, XLM, clib):
    # reference outputs on seeded synthetic models: contiguous / permuted / pruned leaves, no-bias,
    # deep tree, flat (one 300-column chunk -> column tiles), wide chunks, empty query row,
    # beams up to 70 and only_topk up to 100 (LDS top-k path), every lanes-per-item variant
    models = {}
    for c in manifest["synth"]:
        if c["model"] not in models:
            models[c["model"]] = XLM.load(os.path.join(GOLDEN, "synth", c["model"]))
        m = models[c["model"]]
        X = load_X(os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"), c["x"])
        G = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"]))
        assert clib.xlinear_get_int_attr(m.model.model_chain, "nr_dense_layers") > 0
        for dl in (1, 2):                      # default policy / K1Q wherever the candidates fit its registers
            clib.set_option(m.model.model_chain, "dense_layers", dl)
            for pres in (1, 2, 0):             # presence words: unstaged layers only (default) / every layer that has them / never
                clib.set_option(m.model.model_chain, "presence", pres)
                P = m.predict(X, **c["kwargs"])
                assert_same_topk(P, G, exact_scores=EXACT_PP(c["kwargs"].get("post_processor")), what=f"{c} dense format (K1Q), dense_layers={dl} presence={pres}")
            clib.set_option(m.model.model_chain, "presence", 1)
        clib.set_option(m.model.model_chain, "dense_layers", 0)     # tile format: K0 -> K1 -> K2
        for g in (0, 1, 2, 8, 64):
            clib.set_option(m.model.model_chain, "k1_group", g)
            P = m.predict(X, **c["kwargs"])
            assert_same_topk(P, G, exact_scores=EXACT_PP(c["kwargs"].get("post_processor")), what=f"{c} G={g}")
        clib.set_option(m.model.model_chain, "k1_group", 0)
        # tile-sorted items (counting sort by tile; both phases of a pruned layer) instead of the natural order
        clib.set_option(m.model.model_chain, "sort_min_tiles", 1)
        P = m.predict(X, **c["kwargs"])
        assert_same_topk(P, G, exact_scores=EXACT_PP(c["kwargs"].get("post_processor")), what=f"{c} tile-sorted items")
        clib.set_option(m.model.model_chain, "sort_min_tiles", 0)
        clib.set_option(m.model.model_chain, "dense_layers", 1)


@pytest.mark.parametrize("name,scale", [("eurlex-4k", 0.5), ("wiki10-31k", 0.1), ("amazon-670k", 0.02), ("amazon-670k-hard", 0.02)])
def test_scaled_configs_vs_oracle(name, scale, XLM, clib, oracle_mod, tmp_path):
    import xrl_synth
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale)
    X = X[:400]
    m = XLM.load(folder)
    ref = oracle_mod.RefModel(folder) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(folder)
    for pp in (None, "log-l1-hinge", "sigmoid"):
        kw = dict(beam_size=cfg["beam"], only_topk=10)
        if pp:
            kw["post_processor"] = pp
        assert_same_topk(m.predict(X, **kw), ref.predict(X, **kw), exact_scores=EXACT_PP(pp), what=f"{name} {pp}")
    # model defaults (no overrides), max_pred_chunk slicing, dense queries
    assert_same_topk(m.predict(X), ref.predict(X), exact_scores=True, what="defaults")
    # dense row format (default, K1Q) vs tile format: same bits, sparse and dense queries, beams wider than K1Q's registers
    for kw in (dict(beam_size=cfg["beam"], only_topk=10), dict(beam_size=3, only_topk=64), dict(beam_size=70, only_topk=100),
               dict(beam_size=200, only_topk=5, post_processor="log-sigmoid")):
        a = m.predict(X, **kw)
        clib.set_option(m.model.model_chain, "dense_layers", 2)
        a2 = m.predict(X, **kw)
        clib.set_option(m.model.model_chain, "dense_layers", 0)
        b = m.predict(X, **kw)
        clib.set_option(m.model.model_chain, "dense_layers", 1)
        assert_same_topk(a, b, exact_scores=True, what=f"{name} dense vs tile format {kw}")
        assert_same_topk(a2, b, exact_scores=True, what=f"{name} dense (forced) vs tile format {kw}")
    # K1Q: levels 0 and 1 in one walk over the query's features (default where the root keeps all its children) vs separate walks
    for kw in (dict(beam_size=cfg["beam"], only_topk=10), dict(beam_size=64, only_topk=64, post_processor="log-l3-hinge"), dict(beam_size=2, only_topk=5, post_processor="sigmoid")):
        os.environ["XRL_K1Q_FUSE01"] = "0"
        a0 = m.predict(X, **kw)
        os.environ["XRL_K1Q_FUSE01"] = "1"                       # one walk, a load per level
        assert_same_topk(m.predict(X, **kw), a0, exact_scores=True, what=f"{name} levels 0+1 fused (two loads) vs separate {kw}")
        os.environ.pop("XRL_K1Q_FUSE01")                         # default: one walk, ONE load per feature (merged rows) where the model carries them
        assert_same_topk(m.predict(X, **kw), a0, exact_scores=True, what=f"{name} levels 0+1 fused vs separate {kw}")
        assert_same_topk(a0, ref.predict(X, **kw), exact_scores=EXACT_PP(kw.get("post_processor")), what=f"{name} separate walks vs reference {kw}")
    # exact bound pruning off / on (default on): the same bits on every kernel family, and the profile shows both phases of a pruned layer
    for kw in (dict(beam_size=cfg["beam"], only_topk=10), dict(beam_size=cfg["beam"], only_topk=10, post_processor="log-sigmoid"),
               dict(beam_size=4, only_topk=40), dict(beam_size=cfg["beam"], only_topk=3, post_processor="noop")):
        for dl in (1, 0):
            clib.set_option(m.model.model_chain, "dense_layers", dl)
            clib.set_option(m.model.model_chain, "prune", 0)
            a0 = m.predict(X, **kw)
            clib.set_option(m.model.model_chain, "prune", 1)
            assert_same_topk(m.predict(X, **kw), a0, exact_scores=True, what=f"{name} prune on vs off {kw} dense_layers={dl}")
    clib.set_option(m.model.model_chain, "dense_layers", 0)
    clib.set_option(m.model.model_chain, "adaptive", 0)         # (the pruning feedback may have switched a layer to one unstaged pass by now)
    clib.profile_enable(m.model.model_chain, True); clib.profile_reset(m.model.model_chain)
    m.predict(X, beam_size=cfg["beam"], only_topk=10)
    pnames = {r["name"] for r in clib.profile_get(m.model.model_chain)}
    clib.profile_enable(m.model.model_chain, False)
    clib.set_option(m.model.model_chain, "adaptive", 1)
    assert {"k0b_remaining", "k1_sparse_rest", "k2_topk_rest"} <= pnames, pnames
    clib.set_option(m.model.model_chain, "dense_layers", 0)     # the remaining checks are about the tile-format kernels
    # two row batches in flight on two streams (default only for large X): same results, also with a ragged tail batch
    for rows in (2, 150):
        clib.set_option(m.model.model_chain, "overlap_min_rows", 2)
        clib.set_option(m.model.model_chain, "max_batch_rows", rows if rows > 2 else 0)
        assert_same_topk(m.predict(X, beam_size=cfg["beam"], only_topk=10), ref.predict(X, beam_size=cfg["beam"], only_topk=10),
                         exact_scores=True, what=f"{name} two lanes, max_batch_rows={rows}")
    clib.set_option(m.model.model_chain, "max_batch_rows", 0)
    clib.set_option(m.model.model_chain, "overlap_min_rows", 0)
    for srt in (0, 1):   # items in natural order / tile-sorted
        clib.set_option(m.model.model_chain, "sort_min_tiles", srt)
        for pp in (None, "sigmoid"):
            kw = dict(beam_size=cfg["beam"], only_topk=10, **({"post_processor": pp} if pp else {}))
            assert_same_topk(m.predict(X, **kw), ref.predict(X, **kw), exact_scores=EXACT_PP(pp), what=f"{name} sort_min_tiles={srt} {pp}")
    clib.set_option(m.model.model_chain, "sort_min_tiles", 0)
    clib.set_option(m.model.model_chain, "dense_layers", 1)
    assert_same_topk(m.predict(X, beam_size=5, only_topk=3, max_pred_chunk=37), ref.predict(X, beam_size=5, only_topk=3),
                     exact_scores=True, what="max_pred_chunk")
    # the same model with every other row lookup structure: bucket table (what layers too large for rank-bitmaps use),
    # 32-feature words and 64-feature words with the first row's extent (the default picks per layer by tile sparsity)
    for mode, attr in (("bucket", "nr_bucket_layers"), ("bitmap64", "nr_bitmap64_layers"), ("bitmap", None)):
        os.environ["XRL_LOOKUP"] = mode
        try:
            mb = XLM.load(folder)
        finally:
            os.environ.pop("XRL_LOOKUP", None)
        clib.set_option(mb.model.model_chain, "dense_layers", 0)   # this loop is about the tile format's row lookups
        if attr:
            assert clib.xlinear_get_int_attr(mb.model.model_chain, attr) == len(ks)
        for pp in (None, "sigmoid"):
            kw = dict(beam_size=cfg["beam"], only_topk=10, **({"post_processor": pp} if pp else {}))
            assert_same_topk(mb.predict(X, **kw), ref.predict(X, **kw), exact_scores=EXACT_PP(pp), what=f"{name} lookup={mode} {pp}")
        del mb
    if X.shape[1] <= 6000:
        Xd = np.ascontiguousarray(X[:64].toarray())
        for dl, g, gv in ((1, 0, 0), (1, 1, 0), (1, 1, 1), (0, 0, 0)):     # K1Q / the tiled SGEMM K1G forced on every eligible layer (each tile shape) / tile format
            clib.set_option(m.model.model_chain, "dense_layers", dl)
            clib.set_option(m.model.model_chain, "k1g_min_items", g)
            clib.set_option(m.model.model_chain, "k1g_variant", gv)
            for kw in (dict(beam_size=4, only_topk=6), dict(beam_size=cfg["beam"], only_topk=10, post_processor="sigmoid"), dict(beam_size=70, only_topk=100)):
                assert_same_topk(m.predict(Xd, **kw), ref.predict(Xd, **kw), exact_scores=EXACT_PP(kw.get("post_processor")),
                                 what=f"dense X, dense_layers={dl} k1g_min_items={g} {kw}")
        clib.set_option(m.model.model_chain, "dense_layers", 1)
        clib.set_option(m.model.model_chain, "k1g_min_items", 16)
        clib.set_option(m.model.model_chain, "k1g_variant", 0)


def test_full_width_rows_every_lookup(XLM, clib, oracle_mod, tmp_path):
    # one 128-column chunk whose weight rows are completely dense: every tile row holds 128 entries, the value at which
    # the packed extent's length field is all ones (the 64-feature bitmap words route such rows through the extent table);
    # plus a second layer with narrow chunks, under each row lookup structure
    import json
    D, K0, K1 = 70, 128, 256
    folder = str(tmp_path / "m")
    rng = np.random.default_rng(5)
    for d, (K, Kp) in enumerate(((K0, 1), (K1, K0))):
        lf = os.path.join(folder, "ranker", f"{d}.model"); os.makedirs(lf, exist_ok=True)
        W = rng.standard_normal((D + 1, K)).astype(np.float32)
        if d == 1:
            W[rng.random(W.shape) < 0.7] = 0.0
        smat.save_npz(os.path.join(lf, "W.npz"), smat.csc_matrix(W), compressed=False)
        par = np.arange(K) * Kp // K
        C = smat.csc_matrix((np.ones(K, np.float32), (np.arange(K), par)), shape=(K, Kp))
        smat.save_npz(os.path.join(lf, "C.npz"), C, compressed=False)
        json.dump({"model": "MLModel", "bias": 1.0, "pred_kwargs": {"only_topk": 20, "post_processor": "l3-hinge"}},
                  open(os.path.join(lf, "param.json"), "w"))
    json.dump({"model": "HierarchicalMLModel", "depth": 2}, open(os.path.join(folder, "ranker", "param.json"), "w"))
    json.dump({"model": "XLinearModel"}, open(os.path.join(folder, "param.json"), "w"))
    X = smat.random(40, D, density=0.3, format="csr", dtype=np.float32, random_state=3); X.sort_indices()
    om = oracle_mod.OracleModel.load(folder)
    want = om.predict(X, beam_size=128, only_topk=30)
    for mode in ("bitmap", "bitmap64", "bucket", None):
        if mode:
            os.environ["XRL_LOOKUP"] = mode
        try:
            m = XLM.load(folder)
        finally:
            os.environ.pop("XRL_LOOKUP", None)
        assert_same_topk(m.predict(X, beam_size=128, only_topk=30), want, exact_scores=True, what=f"full-width rows, lookup={mode}")
        clib.set_option(m.model.model_chain, "dense_layers", 0)
        assert_same_topk(m.predict(X, beam_size=128, only_topk=30), want, exact_scores=True, what=f"full-width rows, tile format, lookup={mode}")
        assert_same_topk(m.predict(np.ascontiguousarray(X.toarray()), beam_size=128, only_topk=30),
                         om.predict(np.ascontiguousarray(X.toarray()), beam_size=128, only_topk=30), exact_scores=True, what=f"dense X, lookup={mode}")


def test_edge_cases(XLM, clib, oracle_mod, tmp_path):
    import xrl_synth
    folder = str(tmp_path / "m")
    xrl_synth.make_model(folder, 300, 400, [80, 50, 15], seed=11, shape=[3, 20, 400])
    m = XLM.load(folder)
    om = oracle_mod.OracleModel.load(folder)
    X = xrl_synth.make_queries(33, 300, 20, seed=12, relabel_seed=11)
    # all-empty matrix, zero rows, k larger than the number of labels, beam larger than any layer
    E = smat.csr_matrix((5, 300), dtype=np.float32)
    assert_same_topk(m.predict(E, beam_size=2, only_topk=4), om.predict(E, beam_size=2, only_topk=4), exact_scores=True, what="empty rows")
    Z = smat.csr_matrix((0, 300), dtype=np.float32)
    assert m.predict(Z).shape == (0, 400)
    assert_same_topk(m.predict(X, beam_size=1000, only_topk=1000), om.predict(X, beam_size=1000, only_topk=1000), exact_scores=True, what="k > L")
    assert_same_topk(m.predict(X, beam_size=1, only_topk=1), om.predict(X, beam_size=1, only_topk=1), exact_scores=True, what="k = 1")
    # explicit zeros stored in X still "match" (adds 0*w) and unsorted input is rejected like the reference
    Xz = X.copy(); Xz.data[::7] = 0.0
    assert_same_topk(m.predict(Xz, beam_size=3, only_topk=5), om.predict(Xz, beam_size=3, only_topk=5), exact_scores=True, what="explicit zeros")
    Xu = X.copy(); Xu.indices[:2] = Xu.indices[:2][::-1].copy(); Xu.has_sorted_indices = False
    with pytest.raises(ValueError, match="sorted"):
        m.predict(Xu)
    with pytest.raises(AssertionError):
        m.predict(X.astype(np.float64))
    with pytest.raises(AssertionError):
        m.predict(smat.csr_matrix((3, 299), dtype=np.float32))
    # a NON-FINITE query value on a feature no weight column uses: the reference never multiplies it (the row is in no chunk), so the
    # results stay finite -- the dense-format kernels must SKIP cells without a weight, not multiply by zero (K1G's exact loop, K1Q's
    # kMissing select), sparse and dense X
    Wcols = [smat.load_npz(os.path.join(folder, "ranker", f"{d}.model", "W.npz")).tocsr() for d in range(3)]
    unused = [f for f in range(300) if all(W.indptr[f + 1] == W.indptr[f] for W in Wcols)]
    if unused:
        Xi = np.ascontiguousarray(X.toarray()); Xi[::3, unused[0]] = np.inf; Xi[1::3, unused[0]] = np.nan
        Xs_inf = smat.csr_matrix(Xi); Xs_inf.sort_indices()
        want = om.predict(Xi, beam_size=4, only_topk=6)
        assert np.all(np.isfinite(want.data))
        for g in (1, 0):
            clib.set_option(m.model.model_chain, "k1g_min_items", g)
            assert_same_topk(m.predict(Xi, beam_size=4, only_topk=6), want, exact_scores=True, what=f"non-finite x on an unused feature, dense X, k1g_min_items={g}")
        clib.set_option(m.model.model_chain, "k1g_min_items", 16)
        assert_same_topk(m.predict(Xs_inf, beam_size=4, only_topk=6), om.predict(Xs_inf, beam_size=4, only_topk=6), exact_scores=True, what="non-finite x, sparse X")
    # all scores tie (zero weights): order must follow candidate position, not label id
    folder2 = str(tmp_path / "ties")
    xrl_synth.make_model(folder2, 50, 120, [10, 5], seed=13, shape=[6, 120])
    for d in range(2):
        p = os.path.join(folder2, "ranker", f"{d}.model", "W.npz")
        W = smat.load_npz(p); W.data[:] = 0.0; smat.save_npz(p, W, compressed=False)
    m2 = XLM.load(folder2); o2 = oracle_mod.OracleModel.load(folder2)
    Xq = xrl_synth.make_queries(9, 50, 6, seed=14, relabel_seed=13)
    assert_same_topk(m2.predict(Xq, beam_size=3, only_topk=7), o2.predict(Xq, beam_size=3, only_topk=7), exact_scores=True, what="ties")


def test_attributes_and_errors(XLM, clib, tmp_path):
    m = XLM.load(os.path.join(GOLDEN, "synth", "s_pruned"))
    info = clib.inspect_model(os.path.join(GOLDEN, "synth", "s_pruned", "ranker"))
    assert m.depth == 3 and m.nr_features == 300
    assert m.nr_labels == info[-1]["c_nnz"] and m.nr_pred_cols == info[-1]["c_rows"]   # pruned: fewer kept children
    assert m.nr_codes == info[-1]["c_cols"]
    assert clib.xlinear_get_layer_type(m.model.model_chain, 0) == 2
    with pytest.raises((RuntimeError, FileNotFoundError)):
        XLM.load(str(tmp_path))            # no param.json
    with pytest.raises(NotImplementedError):
        XLM.load(os.path.join(GOLDEN, "synth", "s_pruned"), is_predict_only=False)
    with pytest.raises(NotImplementedError):   # non-uniform per-layer override is not expressible natively
        pp = m.get_pred_params(); pp.hlm_args.model_chain[0].only_topk = 3; pp.hlm_args.model_chain[1].only_topk = 4
        m.predict(smat.csr_matrix((1, 300), dtype=np.float32), pred_params=pp)


def test_sparse_inner_products(clib, oracle_mod):
    # KAT of test/pecos/core/test_clib.py:39-69 + random pairs vs the oracle, 4 layout combos
    X = smat.csr_matrix([[1.0, 0.0], [0.5, 0.5], [0.0, 1.0]], dtype=np.float32)
    Y = smat.csr_matrix([[0.5, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 0.5]], dtype=np.float32)
    gt = np.array([[0.50, 0.00, 1.00, 0.00], [0.25, 0.50, 0.50, 0.25], [0.00, 1.00, 0.00, 0.50]], dtype=np.float32)
    r = np.array([0, 1, 2], dtype=np.uint32); c = np.array([1, 2, 3], dtype=np.uint32)
    true = np.array([gt[i, j] for i, j in zip(r, c)], dtype=np.float32)
    W = Y.T.tocsc()
    assert np.allclose(clib.sparse_inner_products(X, W, r, c), true, atol=1e-9)
    assert np.allclose(clib.sparse_inner_products(X.toarray(), np.asfortranarray(Y.toarray().T), r, c), true, atol=1e-9)
    rng = np.random.default_rng(5)
    A = smat.random(200, 500, density=0.05, format="csr", dtype=np.float32, random_state=6); A.sort_indices()
    B = smat.random(500, 300, density=0.08, format="csc", dtype=np.float32, random_state=7); B.sort_indices()
    rr = rng.integers(0, 200, 5000).astype(np.uint32); cc = rng.integers(0, 300, 5000).astype(np.uint32)
    for Aq, Bq in [(A, B), (np.ascontiguousarray(A.toarray()), B), (A, np.asfortranarray(B.toarray())),
                   (np.ascontiguousarray(A.toarray()), np.asfortranarray(B.toarray()))]:
        got = clib.sparse_inner_products(Aq, Bq, rr, cc)
        exp = oracle_mod.sparse_inner_products(Aq, Bq, rr, cc)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        if oracle_mod.ref_available():   # the function this entry point replaces (libpecos.cpp:337-355)
            assert np.array_equal(got.view(np.uint32), oracle_mod.ref_sparse_inner_products(Aq, Bq, rr, cc).view(np.uint32))
    # long index lists on both sides (several 16-entry steps, binary searches in either direction), empty rows / columns
    A2 = smat.random(50, 20000, density=0.02, format="csr", dtype=np.float32, random_state=8); A2.sort_indices()
    B2 = smat.random(20000, 60, density=0.3, format="csc", dtype=np.float32, random_state=9); B2.sort_indices()
    B2 = B2.tolil(); B2[:, 5] = 0; B2 = B2.tocsc().astype(np.float32); B2.sort_indices()
    r2 = rng.integers(0, 50, 3000).astype(np.uint32); c2 = rng.integers(0, 60, 3000).astype(np.uint32)
    got = clib.sparse_inner_products(A2, B2, r2, c2)
    assert np.array_equal(got.view(np.uint32), oracle_mod.sparse_inner_products(A2, B2, r2, c2).view(np.uint32))


def test_single_layer_predict(clib, oracle_mod):
    # N2: c_xlinear_single_layer_predict_{csr,drm}_f32 against the function it replaces -- the REAL reference's entry point
    # (libpecos.cpp:201-235: MLModel<csc_t>, i.e. the CSC arithmetic: bias first, dot product summed separately) -- bit for
    # bit: without codes, with codes (combine), sparse and dense X, a codes row that lists a parent twice
    from pecos_amd.core import ScipyCompressedSparseAllocator
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    layers = oracle_mod.load_model_folder(folder)
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    L0, L1, L2 = layers[0], layers[1], layers[2]

    def ours(Xq, codes, L, pp, k):
        alloc = ScipyCompressedSparseAllocator()
        clib.xlinear_single_layer_predict(Xq, codes, L["W"], L["C"], pp, k, -1, L["bias"], alloc)
        return alloc.get()

    have_ref = oracle_mod.ref_available()
    clib.single_layer_cache_clear()
    st0 = clib.single_layer_cache_stats()
    for Xq in (X, np.ascontiguousarray(X.toarray())):
        for pp in ("l3-hinge", "noop", "sigmoid"):
            P0 = ours(Xq, None, L0, pp, 3)
            codes = smat.csr_matrix(P0, dtype=np.float32)
            P1 = ours(Xq, codes, L1, pp, 6)
            codes1 = smat.csr_matrix(P1, dtype=np.float32)
            P2 = ours(Xq, codes1, L2, pp, 9)
            if have_ref:
                R0 = oracle_mod.ref_single_layer_predict(Xq, None, L0["W"], L0["C"], pp, 3, L0["bias"])
                R1 = oracle_mod.ref_single_layer_predict(Xq, codes, L1["W"], L1["C"], pp, 6, L1["bias"])
                R2 = oracle_mod.ref_single_layer_predict(Xq, codes1, L2["W"], L2["C"], pp, 9, L2["bias"])
                for a, b, what in ((P0, R0, "layer 0"), (P1, R1, "layer 1 with codes"), (P2, R2, "leaf with codes (permuted children)")):
                    assert_same_topk(a, b, exact_scores=EXACT_PP(pp), what=f"single layer vs reference: {what} {pp} dense={not smat.issparse(Xq)}")
            else:   # chunked restatement: same labels, scores within the reference's own cross-layout tolerance
                om = oracle_mod.OracleModel(layers[:1])
                assert np.allclose(P0.toarray(), om.predict(Xq, only_topk=3, post_processor=pp).toarray(), atol=1e-6)
    st1 = clib.single_layer_cache_stats()
    assert st1["misses"] - st0["misses"] == 3 and st1["hits"] - st0["hits"] == 15, (st0, st1)   # 3 layers compiled once, reused 15 times
    # a codes row listing the same parent twice (non-canonical CSR): the reference prolongates both occurrences
    P0 = ours(X, None, L0, "l3-hinge", 3)
    c = smat.csr_matrix(P0, dtype=np.float32)
    d_idx = np.concatenate([np.tile(c.indices[c.indptr[i]:c.indptr[i + 1]], 2) for i in range(c.shape[0])]).astype(np.int32)
    d_val = np.concatenate([np.tile(c.data[c.indptr[i]:c.indptr[i + 1]], 2) for i in range(c.shape[0])]).astype(np.float32)
    d_ptr = np.concatenate([[0], np.cumsum(2 * np.diff(c.indptr))]).astype(np.int32)
    dup = smat.csr_matrix(c.shape, dtype=np.float32)
    dup.indices, dup.data, dup.indptr = d_idx, d_val, d_ptr          # assigned directly: the constructor would merge duplicates
    Pd = ours(X, dup, L1, "l3-hinge", 12)
    if have_ref:
        assert_same_topk(Pd, oracle_mod.ref_single_layer_predict(X, dup, L1["W"], L1["C"], "l3-hinge", 12, L1["bias"]), exact_scores=True, what="duplicate parents")
    # an IN-PLACE edit of one weight (same arrays, same addresses) must be seen: the cache key covers every byte of W and C
    Wm = L0["W"]
    old = float(Wm.data[len(Wm.data) // 3])
    Wm.data[len(Wm.data) // 3] = old + 7.5
    Pe = ours(X, None, L0, "noop", 3)
    if have_ref:
        assert_same_topk(Pe, oracle_mod.ref_single_layer_predict(X, None, L0["W"], L0["C"], "noop", 3, L0["bias"]), exact_scores=True, what="W edited in place")
    Wm.data[len(Wm.data) // 3] = old
    assert_same_topk(ours(X, None, L0, "noop", 3), P0n := ours(X, None, L0, "noop", 3), exact_scores=True, what="restored")
    assert not np.array_equal(Pe.toarray(), P0n.toarray())
    bad = smat.csr_matrix((np.ones(1, np.float32), np.array([L1["C"].shape[1]]), np.array([0, 1] + [1] * (X.shape[0] - 1))), shape=(X.shape[0], L1["C"].shape[1] + 1))
    with pytest.raises(RuntimeError):
        ours(X, bad, L1, "l3-hinge", 3)
    clib.single_layer_cache_clear()
    assert clib.single_layer_cache_stats()["entries"] == 0


def test_csc_weight_matrix_type(XLM, clib, oracle_mod, tmp_path):
    # weight_matrix_type="CSC" (pecos/core/base.py:49): the reference runs w_ops<csc_t> (inference.hpp:1081-1149) on every
    # layer; so does this library (K0 -> K1C -> K2), bit-identical to RefModel(folder, "CSC"), sparse and dense X
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    import xrl_synth
    cases = [(os.path.join(GOLDEN, "synth", n), load_X(os.path.join(GOLDEN, "synth", n + "__X.npz"))) for n in ("s_eurlex", "s_pruned", "s_nobias", "s_flat", "s_deep")]
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config("eurlex-4k", folder, scale=0.5)
    cases.append((folder, X[:300]))
    for f, Xs in cases:
        m = XLM.load(f, weight_matrix_type="CSC")
        assert clib.xlinear_get_layer_type(m.model.model_chain, 0) == 0
        rm = oracle_mod.RefModel(f, "CSC")
        for kw in (dict(beam_size=5, only_topk=7), dict(beam_size=10, only_topk=10, post_processor="log-l2-hinge"), dict()):
            for Xq in (Xs, np.ascontiguousarray(Xs[:48].toarray())):
                assert_same_topk(m.predict(Xq, **kw), rm.predict(Xq, **kw), exact_scores=True, what=f"CSC type {os.path.basename(f)} {kw}")


def test_predict_on_selected_outputs(manifest, XLM, oracle_mod):
    # N1: c_xlinear_predict_on_selected_outputs (libpecos.cpp:179-198).  (1) the reference's own -so golden
    # (test_xlinear.py:368-383); (2) bit-exact vs the CSC-route restatement (pinned on the real reference),
    # output ORDER included, sparse and dense queries.
    Xt = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"))
    G = smat.load_npz(os.path.join(GOLDEN, "ref_fixtures", "Yt_pred.npz")).tocsr()
    m = XLM.load(os.path.join(GOLDEN, "models", "mls10"))
    P = m.predict(Xt, selected_outputs_csr=G)
    assert np.allclose(P.toarray(), G.toarray(), atol=1e-6)
    for name in ["s_eurlex", "s_contig", "s_deep", "s_nobias", "s_flat", "s_wide"]:
        folder = os.path.join(GOLDEN, "synth", name)
        X = load_X(os.path.join(GOLDEN, "synth", name + "__X.npz"))
        m = XLM.load(folder)
        om = oracle_mod.OracleModel.load(folder)
        rm = oracle_mod.RefModel(folder, "CSC") if oracle_mod.ref_available() else None
        S = om.predict(X, beam_size=6, only_topk=8)
        S = smat.csr_matrix((S.data, S.indices, S.indptr), shape=(S.shape[0], m.nr_pred_cols))
        for pp in (None, "sigmoid", "log-l2-hinge", "noop"):
            for Xq in (X, np.ascontiguousarray(X.toarray())):
                kw = {"post_processor": pp} if pp else {}
                a = m.predict(Xq, selected_outputs_csr=S, **kw)
                b = om.predict_on_selected_outputs(Xq, S, pp)
                assert_same_topk(a, b, exact_scores=EXACT_PP(pp), what=f"{name} {pp}")
                if rm is not None:
                    assert_same_topk(a, rm.predict_on_selected_outputs(Xq, S, pp), exact_scores=EXACT_PP(pp), what=f"ref {name} {pp}")
        # top-k and selected-outputs agree on the top-k pattern (test_xlinear.py:1059-1137)
        T = m.predict(X, beam_size=6, only_topk=8)
        Ssel = m.predict(X, selected_outputs_csr=T, beam_size=6)
        assert np.allclose(Ssel.toarray(), T.toarray(), atol=1e-6)
    with pytest.raises(ValueError):
        m.predict(X, selected_outputs_csr=smat.csr_matrix((X.shape[0], 3), dtype=np.float32))
    dup = smat.csr_matrix((np.ones(2, np.float32), np.array([1, 1]), np.array([0, 2] + [2] * (X.shape[0] - 1))), shape=(X.shape[0], m.nr_pred_cols))
    with pytest.raises(RuntimeError, match="twice"):
        m.predict(X, selected_outputs_csr=dup)


def test_single_layer_predict_on_selected_outputs(clib, oracle_mod):
    # libpecos.cpp:237-274: layer-by-layer selected prediction chained through csr_codes equals the
    # whole-model selected prediction (values and order), which is pinned on the real reference
    from pecos_amd.core import ScipyCompressedSparseAllocator
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    layers = oracle_mod.load_model_folder(folder)
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    om = oracle_mod.OracleModel.load(folder)
    S2 = om.predict(X, beam_size=5, only_topk=6)
    full = om.predict_on_selected_outputs(X, S2)
    # per-layer patterns bottom-up (parents of the selected nodes)
    pats = [None, None, smat.csr_matrix(S2)]
    for l in (2, 1):
        Cl = smat.csr_matrix(layers[l]["C"])                 # child x parent
        P = (smat.csr_matrix((np.ones_like(pats[l].data), pats[l].indices, pats[l].indptr), shape=pats[l].shape) @ Cl).tocsr()
        P.sort_indices(); pats[l - 1] = P
    codes = None
    for l in range(3):
        alloc = ScipyCompressedSparseAllocator()
        clib.xlinear_single_layer_predict_on_selected_outputs(X, pats[l].astype(np.float32), codes, layers[l]["W"], layers[l]["C"],
                                                              "l3-hinge", -1, layers[l]["bias"], alloc)
        codes = smat.csr_matrix(alloc.get(), dtype=np.float32)
    assert_same_topk(codes, full, exact_scores=True, what="chained single-layer selected == whole-model selected")


def test_mmap_models(manifest, XLM, clib, tmp_path):
    # N3: memory-mapped model folders compiled BY THE REFERENCE load through c_xlinear_load_mmap_model_from_disk
    # and predict bit-identically to the npz model; our own compile -> load round trip does too
    for c in manifest["mmap"]:
        npz = os.path.join(GOLDEN, c["kind"], c["model"])
        X = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz") if c["kind"] == "models" else os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"))
        m_npz = XLM.load(npz)
        m_map = XLM.load(os.path.join(GOLDEN, "mmap", c["model"]))
        out = str(tmp_path / c["model"])
        XLM.compile_mmap_model(npz, out)
        m_rt = XLM.load(out)
        for kw in (dict(), dict(beam_size=3, only_topk=5, post_processor="sigmoid")):
            for Xq in (X, np.ascontiguousarray(X.toarray())):
                a = m_npz.predict(Xq, **kw)
                for other in (m_map, m_rt):
                    b = other.predict(Xq, **kw)
                    assert a.shape == b.shape
                    assert_same_topk(b, a, exact_scores=True, what=f"{c} {kw}")
        assert (m_map.depth, m_map.nr_features, m_map.nr_labels, m_map.nr_codes, m_map.nr_pred_cols) == \
               (m_npz.depth, m_npz.nr_features, m_npz.nr_labels, m_npz.nr_codes, m_npz.nr_pred_cols)
    with pytest.raises(RuntimeError, match="npz model"):
        clib.xlinear_load_mmap(os.path.join(GOLDEN, "synth", "s_eurlex", "ranker"))
    with pytest.raises(RuntimeError, match="mmap model"):
        clib.xlinear_load_predict_only(os.path.join(GOLDEN, "mmap", "s_eurlex", "ranker"))


def test_full_size_properties(XLM, clib, tmp_path):
    # BASELINE.json configs[1] (Eurlex-4K shape) at FULL size through size-independent properties
    import xrl_synth
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config("eurlex-4k", folder)
    m = XLM.load(folder)
    P = m.predict(X, beam_size=10, only_topk=10)
    assert P.shape == (X.shape[0], ks[-1])
    cnt = np.diff(P.indptr)
    assert np.all(cnt == 10)                                                  # L >> k: every row is full
    D = P.data.reshape(-1, 10)
    assert np.all(D[:, :-1] >= D[:, 1:])                                      # rows score-sorted, descending
    I = P.indices.reshape(-1, 10)
    assert np.all(np.sort(I, axis=1)[:, 1:] != np.sort(I, axis=1)[:, :-1])    # no duplicate label in a row
    assert np.all((D > 0) & (D <= 1))                                         # l3-hinge products live in (0, 1]
    # idempotence + batch-composition independence (rows are independent): permuted / chunked batches
    perm = np.random.default_rng(0).permutation(X.shape[0])
    Pp = m.predict(X[perm], beam_size=10, only_topk=10)
    assert np.array_equal(Pp.indices.reshape(-1, 10), I[perm]) and np.array_equal(Pp.data.reshape(-1, 10), D[perm])
    Pc = m.predict(X, beam_size=10, only_topk=10, max_pred_chunk=4099)
    assert np.array_equal(Pc.indices, P.indices) and np.array_equal(Pc.data, P.data)
    # top-k prefix property: only_topk=5 is the prefix of only_topk=10 at equal beam
    P5 = m.predict(X, beam_size=10, only_topk=5)
    assert np.array_equal(P5.indices.reshape(-1, 5), I[:, :5]) and np.array_equal(P5.data.reshape(-1, 5), D[:, :5])
    # every lanes-per-item variant of the tile-format kernels gives the same bits as the dense-format kernel
    clib.set_option(m.model.model_chain, "dense_layers", 0)
    for g in (0, 1, 4, 16, 64):
        clib.set_option(m.model.model_chain, "k1_group", g)
        Pg = m.predict(X, beam_size=10, only_topk=10)
        assert np.array_equal(Pg.indices, P.indices) and np.array_equal(Pg.data.view(np.uint32), P.data.view(np.uint32))
    # device-resident path == host-ABI path; batching of rows does not change results
    clib.set_option(m.model.model_chain, "k1_group", 0)
    clib.set_option(m.model.model_chain, "dense_layers", 1)
    clib.set_option(m.model.model_chain, "max_batch_rows", 1777)
    Pb = m.predict(X, beam_size=10, only_topk=10)
    clib.set_option(m.model.model_chain, "max_batch_rows", 0)
    assert np.array_equal(Pb.indices, P.indices) and np.array_equal(Pb.data, P.data)


def _reference_model(oracle_mod, folder):
    return oracle_mod.RefModel(folder) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(folder)


@pytest.mark.parametrize("name", ["eurlex-4k", "wiki10-31k"])
def test_full_size_vs_reference(name, XLM, clib, oracle_mod, tmp_path):
    # BASELINE.json configs[1] and configs[2] at FULL size, every row against the reference (oracle/_ref when built,
    # else the C restatement): label ids, order and fp32 scores bit-identical, for the dense row format (K1Q where a
    # layer's candidates fit its registers) and for the tile format (K0 -> K1 -> K2)
    import xrl_synth
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config(name, folder)
    m = XLM.load(folder)
    ref = _reference_model(oracle_mod, folder)
    kw = dict(beam_size=cfg["beam"], only_topk=10)
    want = ref.predict(X, threads=32, **kw) if oracle_mod.ref_available() else ref.predict(X, **kw)
    assert want.shape == (X.shape[0], ks[-1])
    for dl in (1, 2, 0):
        clib.set_option(m.model.model_chain, "dense_layers", dl)
        assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{name} full size, dense_layers={dl}")
    clib.set_option(m.model.model_chain, "dense_layers", 1)
    if name == "wiki10-31k":
        # host ABI with >= 3 row batches (75 MB of CSR, 4 MB first batch) while predict_device's OWN two-lane split is on (the two
        # schemes used to share scratch lanes and the auxiliary stream without an ordering between them); repeated, the race was timing-dependent
        h = m.model.model_chain
        clib.set_option(h, "host_batch_mb", 4)
        for omr in (2, 0):
            clib.set_option(h, "overlap_min_rows", omr)
            for rep in range(3):
                assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{name} full size, host batches + overlap_min_rows={omr}, call {rep}")
        clib.set_option(h, "host_batch_mb", 12)


def test_hash_chunked_sparse_queries_bit_exact(manifest, XLM, clib, oracle_mod, tmp_path):
    # weight_matrix_type="HASH_CHUNKED" with CSR queries: the reference adds the bias row FIRST and then the query's features in
    # ascending order (chunk_ops<csr, hash_chunked>, inference.hpp:705-735) -- deterministic, so it is matched bit for bit:
    # reference-generated goldens (make_golden_r03.py) through every kernel family, then a fresh model against the live reference
    models = {}
    for c in manifest["synth_hash"]:
        if c["model"] not in models:
            models[c["model"]] = XLM.load(os.path.join(GOLDEN, "synth", c["model"]), weight_matrix_type="HASH_CHUNKED")
        m = models[c["model"]]
        h = m.model.model_chain
        X = load_X(os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"))
        G = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"]))
        ex = EXACT_PP(c["kwargs"].get("post_processor"))
        for dl, srt in ((1, 0), (2, 0), (0, 0), (0, 1)):
            clib.set_option(h, "dense_layers", dl); clib.set_option(h, "sort_min_tiles", srt)
            assert_same_topk(m.predict(X, **c["kwargs"]), G, exact_scores=ex, what=f"HASH_CHUNKED {c} dense_layers={dl} sort_min_tiles={srt}")
        clib.set_option(h, "dense_layers", 1); clib.set_option(h, "sort_min_tiles", 0)
    if oracle_mod.ref_available():
        import xrl_synth
        folder = str(tmp_path / "m")
        ks, X, cfg = xrl_synth.make_config("eurlex-4k", folder, scale=0.5)
        X = X[:2000]
        m = XLM.load(folder, weight_matrix_type="HASH_CHUNKED")
        ref = oracle_mod.RefModel(folder, "HASH_CHUNKED")
        ref_bs = oracle_mod.RefModel(folder)
        for kw in (dict(beam_size=10, only_topk=10), dict(beam_size=4, only_topk=20, post_processor="log-l1-hinge")):
            want = ref.predict(X, **kw)
            for dl in (1, 0):
                clib.set_option(m.model.model_chain, "dense_layers", dl)
                assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"HASH_CHUNKED vs live reference {kw} dense_layers={dl}")
            other = ref_bs.predict(X, **kw)
            assert not np.array_equal(other.data.view(np.uint32), want.data.view(np.uint32))     # the two layouts do differ in the last bits
        # DENSE queries under HASH_CHUNKED: the reference sums a chunk's rows in the iteration order of its robin-hood hash table
        # (chunk_ops<drm, hash>, inference.hpp:737-768) -- not reproduced: this library sums in ascending feature order (the BINARY_SEARCH_CHUNKED
        # arithmetic).  The two orders differ by a few ulps of the ACCUMULATOR (margins ~1: <= 5e-7 absolute); an additive
        # post-processor turns that into the same ABSOLUTE error of a score that may itself be small (log-l1-hinge: 1 - margin cancels: 2.4e-5
        # relative observed on scores of -0.02), a multiplicative one keeps it relative.  What must hold: the same labels up to swaps of
        # near-ties, scores within 1e-5 relative + 2e-6 absolute.
        from conftest import assert_topk_close
        Xd = np.ascontiguousarray(X[:400].toarray())
        for kw in (dict(beam_size=10, only_topk=10), dict(beam_size=4, only_topk=20, post_processor="log-l1-hinge")):
            want = ref.predict(Xd, **kw)
            for dl in (1, 0):
                clib.set_option(m.model.model_chain, "dense_layers", dl)
                assert_topk_close(m.predict(Xd, **kw), want, rel=1e-5, atol=2e-6, what=f"HASH_CHUNKED dense X vs live reference {kw} dense_layers={dl}")
        clib.set_option(m.model.model_chain, "dense_layers", 1)


def test_topk_beyond_the_lds_limit(XLM, clib, oracle_mod, tmp_path):
    # only_topk / beam_size > 20 480 (what a workgroup's LDS can rank): the reference's sorted_csr has no cap (inference.hpp:1223-1298); here the
    # segmented-sort K2 (csrc/xrl_topk_big.hip) takes over.  A flat model of 30 000 labels, only_topk 25 000; and a two-layer one with beam 21 000.
    import xrl_synth
    for shape, kw in (([30000], dict(only_topk=25000)), ([24000, 26000], dict(beam_size=21000, only_topk=22000))):
        folder = str(tmp_path / f"m{len(shape)}")
        xrl_synth.make_model(folder, 60, shape[-1], [8] * len(shape), seed=77, shape=shape)
        X = xrl_synth.make_queries(6, 60, 10, seed=78, relabel_seed=77)
        m = XLM.load(folder)
        om = oracle_mod.OracleModel.load(folder)
        for Xq in (X, np.ascontiguousarray(X.toarray())):
            assert_same_topk(m.predict(Xq, **kw), om.predict(Xq, **kw), exact_scores=True, what=f"big k {shape} {kw} dense={not smat.issparse(Xq)}")


@pytest.mark.parametrize("variant", ["all_saturated", "bias_only", "descending"])
def test_bound_pruning_with_massive_ties(variant, XLM, clib, oracle_mod, tmp_path):
    # exact bound pruning at its boundary conditions.  all_saturated: every weight is +2 and x >= 0, so every accumulator is >= 1 wherever a
    # feature matches and l3-hinge returns exactly 1.0 -- parents tie, children tie with their parents, and the whole top-k is decided by
    # candidate POSITION; bias_only: no query feature matches anything (empty rows), scores come from the bias alone and tie within a
    # layer; descending: weights shrink with the child id so that scores strictly decrease and the k-th best sits just above / below the
    # next parent's score.  Pruning on must equal pruning off and the oracle for every beam / top-k, every kernel family, sparse and dense X.
    import xrl_synth
    folder = str(tmp_path / "m")
    D = 120
    xrl_synth.make_model(folder, D, 700, [60, 40, 12], seed=41, shape=[5, 40, 700], permute_leaf=True)
    rng = np.random.default_rng(7)
    for d in range(3):
        f = os.path.join(folder, "ranker", f"{d}.model", "W.npz")
        W = smat.load_npz(f).tocsc().astype(np.float32)
        if variant == "descending":
            col = np.repeat(np.arange(W.shape[1]), np.diff(W.indptr))
            W.data[:] = (1.5 / (1.0 + 0.01 * col)).astype(np.float32)
        else:
            W.data[:] = 2.0
        smat.save_npz(f, W, compressed=False)
    X = xrl_synth.make_queries(50, D, 12, seed=43, relabel_seed=41)
    if variant == "bias_only":
        X = smat.csr_matrix(X.shape, dtype=np.float32)
    m = XLM.load(folder)
    h = m.model.model_chain
    om = oracle_mod.OracleModel.load(folder)
    for Xq in (X, np.ascontiguousarray(X.toarray())):
        for kw in (dict(beam_size=10, only_topk=10), dict(beam_size=3, only_topk=30), dict(beam_size=2, only_topk=1), dict(beam_size=25, only_topk=64),
                   dict(beam_size=10, only_topk=10, post_processor="log-l2-hinge"), dict(beam_size=7, only_topk=12, post_processor="sigmoid")):
            want = om.predict(Xq, **kw)
            for dl in (1, 2, 0):
                clib.set_option(h, "dense_layers", dl)
                for pr in (1, 0):
                    clib.set_option(h, "prune", pr)
                    assert_same_topk(m.predict(Xq, **kw), want, exact_scores=EXACT_PP(kw.get("post_processor")),
                                     what=f"{variant} {kw} dense_layers={dl} prune={pr} dense_x={not smat.issparse(Xq)}")
    clib.set_option(h, "dense_layers", 1); clib.set_option(h, "prune", 1)


@pytest.mark.parametrize("variant", ["x_nonfinite", "x_huge", "w_inf", "w_nan"])
def test_bound_pruning_guard_nonfinite(variant, XLM, clib, oracle_mod, tmp_path):
    # exact bound pruning assumes a child's score is at most its parent's bound, which fails when a child's score is NaN
    # (non-finite x, inf * 0, inf - inf after an overflow, non-finite weights) -- a positive NaN ranks above +inf in the top-k order, so a
    # later beam slot's NaN candidate belongs at the top of the reference's list.  The guard (prune_guard_ok, csrc/xrl_device.h) never prunes
    # a query whose largest |x| times the model's largest |weight| could leave the fp32 range.  What must hold: prune = 1 gives the bits
    # prune = 0 gives, on every kernel family, sparse and dense X -- and where no NaN can arise (all-positive weights, +inf / huge x) both equal
    # the oracle.
    import xrl_synth
    folder = str(tmp_path / "m")
    D = 120
    xrl_synth.make_model(folder, D, 700, [60, 40, 12], seed=51, shape=[5, 40, 700], permute_leaf=True)
    rng = np.random.default_rng(11)
    positive = variant in ("x_huge", "w_inf")
    for d in range(3):
        f = os.path.join(folder, "ranker", f"{d}.model", "W.npz")
        W = smat.load_npz(f).tocsc().astype(np.float32)
        if positive:
            W.data[:] = np.abs(W.data) + 0.01
        if variant == "w_inf" and d >= 1:
            W.data[rng.integers(0, len(W.data), 6)] = np.inf
        if variant == "w_nan" and d >= 1:
            W.data[rng.integers(0, len(W.data), 6)] = np.nan
            W.data[rng.integers(0, len(W.data), 6)] = -np.inf
        smat.save_npz(f, W, compressed=False)
    X = xrl_synth.make_queries(64, D, 14, seed=53, relabel_seed=51).tocsr()
    X.data = np.abs(X.data)
    if variant == "x_nonfinite":      # NaN, +inf, -inf and explicit zeros on features the model uses
        for r in range(0, 64, 2):
            lo, hi = X.indptr[r], X.indptr[r + 1]
            if hi - lo >= 3:
                X.data[lo + int(rng.integers(0, hi - lo))] = [np.nan, np.inf, -np.inf, 0.0][(r // 2) % 4]
    if variant == "x_huge":           # products overflow to +inf (weights are positive: no inf - inf)
        for r in range(0, 64, 3):
            lo, hi = X.indptr[r], X.indptr[r + 1]
            if hi > lo:
                X.data[lo + int(rng.integers(0, hi - lo))] = [3.0e38, np.inf, 1.0e30][(r // 3) % 3]
    m = XLM.load(folder)
    h = m.model.model_chain
    om = oracle_mod.RefModel(folder) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(folder)
    for Xq in (X, np.ascontiguousarray(X.toarray())):
        for kw in (dict(beam_size=10, only_topk=10), dict(beam_size=3, only_topk=30), dict(beam_size=25, only_topk=5, post_processor="log-l2-hinge"),
                   dict(beam_size=7, only_topk=12, post_processor="sigmoid")):
            base = None
            for dl, tr in ((1, 1), (2, 1), (0, 0), (0, 2)):   # tr: tile-format layers on the entry-list kernel K1 / on the densely held tile rows (K1T) in every launch
                clib.set_option(h, "dense_layers", dl); clib.set_option(h, "tile_rows", tr)
                for pr in (0, 1):
                    clib.set_option(h, "prune", pr)
                    got = m.predict(Xq, **kw)
                    what = f"{variant} {kw} dense_layers={dl} tile_rows={tr} prune={pr} dense_x={not smat.issparse(Xq)}"
                    if pr == 0 and not (dl == 0 and tr == 2):
                        base = got
                    else:   # pruning must not change a bit, NaN scores included -- nor must the kernel that serves the tile format (K1T's exact loop on a non-finite x)
                        assert np.array_equal(got.indptr, base.indptr) and np.array_equal(got.indices, base.indices), what
                        assert np.array_equal(got.data.view(np.uint32), base.data.view(np.uint32)), what
                    # (dense X multiplies EVERY chunk row: 0 * inf is a NaN in the reference too, and where a NaN lands in its std::sort is
                    #  not a defined order -- the oracle is consulted only where no NaN can arise)
                    if positive and (smat.issparse(Xq) or variant == "x_huge"):
                        assert_same_topk(got, om.predict(Xq, **kw), exact_scores=EXACT_PP(kw.get("post_processor")), what=what)
    clib.set_option(h, "dense_layers", 1); clib.set_option(h, "prune", 1); clib.set_option(h, "tile_rows", 1)


def _bench_workload(name, cache=None):
    """The folder bench.py generates / re-uses for a workload at scale 1.0 (so that the driver's pytest and bench runs build it once)."""
    import json
    import xrl_synth
    folder = os.path.join(cache or os.environ.get("XRL_BENCH_CACHE", "/tmp/xrl_bench"), f"{name}_1.0")
    if not os.path.exists(os.path.join(folder, ".done")):
        os.makedirs(folder, exist_ok=True)
        ks, X, cfg = xrl_synth.make_config(name, folder, scale=1.0)
        if smat.issparse(X):
            smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
        else:
            np.save(os.path.join(folder, "X.npy"), X)
        json.dump({"ks": ks, "cfg": cfg}, open(os.path.join(folder, "meta.json"), "w"))
        open(os.path.join(folder, ".done"), "w").write("ok")
    meta = json.load(open(os.path.join(folder, "meta.json")))
    return folder, meta["ks"], meta["cfg"]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("workload", ["amazon-670k", "amazon-670k-hard"])
def test_headline_config_full_size_all_rows_vs_reference(workload, XLM, clib, oracle_mod):
    # BASELINE.json configs[3], the configuration the target is quoted on, at FULL size: Amazon-670K shape, N = 490 000 queries,
    # D = 135 000, L = 670 091, tree [2, 32, 512, 8192, 670091], beam 10, top-k 10 -- EVERY row against the compiled reference
    # (test/pecos/xmc/xlinear/test_xlinear.py:106-245 in spirit): label ids, order and fp32 score bits.  The default policy runs the
    # kernel instantiations bench.py times (fused k1q_kernel<3,0,false,true> over levels 0-3, leaf k1_kernel<32,3,0,false,2>);
    # dense_layers=0 sends every level through the tile-format kernels.
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) is not built: 490 k rows are out of reach of the single-threaded restatement")
    if os.environ.get("XRL_SKIP_FULLSIZE") == "1":
        pytest.skip("XRL_SKIP_FULLSIZE=1 (builder iterations on a metered GPU: the two workloads take ~4 minutes of host time to generate)")
    # "amazon-670k-hard": the same shape on the model with query-dependent routing and an unsaturated post-processor (xrl_synth.make_model_hard),
    # where bound pruning stops almost nothing -- the second phases of every kernel family do the bulk of the work.
    folder, ks, cfg = _bench_workload(workload)
    X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32); X.sort_indices()
    assert X.shape == (490000, 135000) and ks == [2, 32, 512, 8192, 670091]
    kw = dict(beam_size=10, only_topk=10)
    want = oracle_mod.RefModel(folder).predict(X, threads=min(os.cpu_count() or 1, 64), **kw)
    m = XLM.load(folder)
    h = m.model.model_chain
    clib.profile_enable(h, True); clib.profile_reset(h)
    assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{workload} full size, default policy")
    names = {r["name"] for r in clib.profile_get(h)}
    clib.profile_enable(h, False)
    assert "k1q_fused_0_3" in names and "k1_sparse" in names, names
    # ---- the launch shape bench.py TIMES (VERDICT r5 missing #3): X resident in HBM, ONE xrl_predict_device_rows call over all 490 000
    #      rows with default options -- which is what reaches the sorted launch (qsort_min_rows = 131 072): levels 0-2 fused, queries
    #      counting-sorted by the best beam parent, level 3 as its own single-layer launch on the permutation -- every row against the reference
    import torch
    from pecos_amd.distributed import rows_to_csr
    k = clib.effective_topk(h, 10)
    q = clib.queries_upload(h, X)
    N = X.shape[0]
    dev = torch.device("cuda", 0)
    t_idx = torch.zeros((N, k), dtype=torch.int32, device=dev); t_val = torch.zeros((N, k), dtype=torch.float32, device=dev)
    t_cnt = torch.zeros((N,), dtype=torch.int32, device=dev)
    for rep in range(7):   # repeats: the pruning feedback switches layers to their unstaged (presence-word) instantiations on the hard model
        t_idx.fill_(-1); t_val.fill_(float("nan")); t_cnt.fill_(-1)
        clib.profile_enable(h, True); clib.profile_reset(h)
        clib.predict_device_rows(h, q, 10, None, 10, t_idx.data_ptr(), t_val.data_ptr(), t_cnt.data_ptr(), k, 0, N, sync=True)
        names = {r["name"] for r in clib.profile_get(h)}
        clib.profile_enable(h, False)
        assert {"k1q_fused_0_2", "k1_sort_queries", "k1q_dense"} <= names and "k1q_fused_0_3" not in names, names
        got = rows_to_csr(t_idx.cpu().numpy().view(np.uint32), t_val.cpu().numpy(), t_cnt.cpu().numpy().view(np.uint32), m.nr_pred_cols)
        assert_same_topk(got, want, exact_scores=True, what=f"{workload} full size, device-resident X, one 490000-row call (sorted launch), predict #{rep}")
    clib.queries_free(q)
    del t_idx, t_val, t_cnt
    clib.set_option(h, "dense_layers", 0)
    assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{workload} full size, tile format everywhere")
    clib.set_option(h, "dense_layers", 1)
    clib.set_option(h, "prune", 0)                    # every candidate of every beam parent scored (no exact bound pruning)
    assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{workload} full size, prune=0")
    clib.set_option(h, "prune", 1)
    # pruning feedback (option adaptive, default on): after a few predicts the handle runs the layers whose first stage settled nothing
    # unstaged -- on the hard model the leaf switches to one pass over tile-sorted items and levels 2-3 drop their staged walk; same bits
    for rep in range(4):
        clib.profile_enable(h, True); clib.profile_reset(h)
        got = m.predict(X, **kw)
        prof = clib.profile_get(h)
        names = {r["name"] for r in prof}
        clib.profile_enable(h, False)
        assert_same_topk(got, want, exact_scores=True, what=f"{workload} full size, predict #{rep + 4} (pruning feedback)")
    if workload.endswith("-hard"):
        # the leaf runs unstaged by now -- in (nearly) every row batch of the call: the feedback stages a layer again once in 32 predict_device
        # calls to look whether the data has changed, and a host-ABI call is ~10 of them, so a single re-probed batch may show up
        launches = {}
        for r in prof:
            launches[r["name"]] = launches.get(r["name"], 0) + r["launches"]
        assert "k1_sort_items" in names and launches.get("k1_sparse_rest", 0) <= 2 <= launches["k1_sort_items"], launches
    else:
        assert "k1_sparse_rest" in names, names                                           # ... and stays staged where pruning works
    clib.set_option(h, "adaptive", 0)
    assert_same_topk(m.predict(X, **kw), want, exact_scores=True, what=f"{workload} full size, adaptive=0")
    clib.set_option(h, "adaptive", 1)


@pytest.mark.timeout(2400)
def test_dense768_full_size_model_vs_reference(XLM, clib, oracle_mod, tmp_path):
    # BASELINE.json configs[4] with the FULL-size model (D = 768 dense fp32 queries, L = 3 000 000, tree [8, 128, 2048, 32768, 3 M],
    # 17 GB on the GPU): 32 768 queries, every one against the compiled reference -- the tiled SGEMM K1G on every level by default,
    # the query-stationary kernel on a slice.  XRL_SKIP_HUGE=1 skips it (builder iterations: the model takes minutes to generate).
    if os.environ.get("XRL_SKIP_HUGE") == "1":
        pytest.skip("XRL_SKIP_HUGE=1")
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) is not built")
    import xrl_synth
    cfg = xrl_synth.CONFIGS["dense-768"]
    folder = os.path.join(os.environ.get("XRL_BENCH_CACHE", "/tmp/xrl_bench"), "dense-768_1.0")
    if os.path.exists(os.path.join(folder, ".done")):
        ks = xrl_synth.tree_shape(cfg["L"])
    else:
        folder = str(tmp_path / "m")
        ks = xrl_synth.make_model(folder, cfg["D"], cfg["L"], cfg["w_nnz"], seed=0)
    assert ks == [8, 128, 2048, 32768, 3000000]
    X = xrl_synth.make_queries(32768, cfg["D"], None, seed=1)
    kw = dict(beam_size=10, only_topk=10)
    m = XLM.load(folder)
    h = m.model.model_chain
    clib.profile_enable(h, True); clib.profile_reset(h)
    P = m.predict(X, **kw)
    names = {r["name"] for r in clib.profile_get(h)}
    clib.profile_enable(h, False)
    assert "k1g_dense_x" in names, names
    want = oracle_mod.RefModel(folder).predict(X, threads=min(os.cpu_count() or 1, 64), **kw)
    assert_same_topk(P, want, exact_scores=True, what="dense-768 full-size model, 32768 queries (K1G)")
    clib.set_option(h, "k1g_min_items", 0)
    assert_same_topk(m.predict(X[:2048], **kw), want[:2048], exact_scores=True, what="dense-768 full-size model, query-stationary kernels")
    clib.set_option(h, "k1g_min_items", 16)


def test_dense_input_config_vs_reference(XLM, clib, oracle_mod, tmp_path):
    # BASELINE.json configs[4] (dense fp32 X, D=768) at a tenth of its label count: N=50k x 768, L=300k, tree
    # [16, 256, 4096, 300000].  Every layer fits the dense row format, so the whole beam search runs in K1Q with dense
    # queries; 2048 rows are compared with the reference (its dense path costs ~1 M multiply-adds per query), all rows
    # with the tile-format kernels.
    import xrl_synth
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config("dense-768", folder, scale=0.1)
    X = np.ascontiguousarray(X[:50000])
    m = XLM.load(folder)
    assert clib.xlinear_get_int_attr(m.model.model_chain, "nr_dense_layers") == len(ks)
    kw = dict(beam_size=10, only_topk=10)
    P = m.predict(X, **kw)
    ref = _reference_model(oracle_mod, folder)
    ns = 2048
    want = ref.predict(X[:ns], threads=32, **kw) if oracle_mod.ref_available() else ref.predict(X[:ns], **kw)
    got = smat.csr_matrix((P.data[:P.indptr[ns]], P.indices[:P.indptr[ns]], P.indptr[:ns + 1]), shape=(ns, P.shape[1]))
    assert_same_topk(got, want, exact_scores=True, what="dense-768 vs reference")
    # the default above ran the tiled SGEMM K1G on every layer (50k dense queries share few parents); the query-stationary
    # kernel K1Q and the tile-format kernels must give the same bits
    clib.profile_enable(m.model.model_chain, True); clib.profile_reset(m.model.model_chain)
    m.predict(X[:8192], **kw)
    names = {r["name"] for r in clib.profile_get(m.model.model_chain)}
    clib.profile_enable(m.model.model_chain, False)
    assert "k1g_dense_x" in names, names
    for opt, val in (("k1g_variant", 1), ("k1g_min_items", 0), ("dense_layers", 0)):
        clib.set_option(m.model.model_chain, opt, val)
        Pt = m.predict(X[:8192], **kw)
        assert np.array_equal(Pt.indices, P.indices[:P.indptr[8192]]) and np.array_equal(Pt.data.view(np.uint32), P.data[:P.indptr[8192]].view(np.uint32)), opt
