"""CPU: pin the oracle (oracle/xrl_oracle.c) against the reference's golden vectors and against
outputs the real reference produced in this container (tests/golden/, made by make_golden.py)."""
import os

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import GOLDEN, assert_same_topk, load_raw_csr, load_X


def test_oracle_vs_reference_cli_goldens(manifest, oracle_mod):
    # test/pecos/xmc/xlinear/test_xlinear.py:314-640 golden prediction files, abs=1e-6 like the reference
    Xt = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"))
    for c in manifest["cli"]:
        m = oracle_mod.OracleModel.load(os.path.join(GOLDEN, "models", c["model"]))
        kw = {k: v for k, v in c["kwargs"].items() if k != "max_pred_chunk"}
        P = m.predict(Xt, **kw)
        G = smat.load_npz(os.path.join(GOLDEN, "ref_fixtures", c["golden"]))
        assert np.allclose(P.toarray(), G.toarray(), atol=1e-6), c


def test_oracle_vs_reference_toy_matrix(manifest, oracle_mod):
    # every post-processor x {sparse, dense} x 3 trained toy models (test_xlinear.py:106-245)
    models = {}
    for c in manifest["toy"]:
        if c["model"] not in models:
            models[c["model"]] = oracle_mod.OracleModel.load(os.path.join(GOLDEN, "models", c["model"]))
        X = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"), c["x"])
        P = models[c["model"]].predict(X, beam_size=c["beam_size"], post_processor=c["post_processor"])
        G = smat.load_npz(os.path.join(GOLDEN, "preds", c["pred"]))
        assert np.allclose(P.toarray(), G.toarray(), atol=1e-6), c


def test_oracle_bit_exact_vs_reference_on_synthetic(manifest, oracle_mod):
    # seeded synthetic models: label sequence identical, fp32 scores BIT-identical to the compiled reference
    models = {}
    for c in manifest["synth"]:
        if c["model"] not in models:
            models[c["model"]] = oracle_mod.OracleModel.load(os.path.join(GOLDEN, "synth", c["model"]))
        X = load_X(os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"), c["x"])
        P = models[c["model"]].predict(X, **c["kwargs"])
        G = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"]))
        assert_same_topk(P, G, exact_scores=True, what=str(c))


def test_oracle_vs_live_reference_if_present(oracle_mod, tmp_path):
    # when oracle/_ref is available (it travels to the GPU box), compare on a fresh seeded model too
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    import xrl_synth
    folder = str(tmp_path / "m")
    xrl_synth.make_model(folder, 400, 700, [120, 80, 25], seed=3, shape=[3, 20, 700])
    X = xrl_synth.make_queries(40, 400, 30, seed=4, relabel_seed=3)
    om = oracle_mod.OracleModel.load(folder)
    rm = oracle_mod.RefModel(folder)
    for pp in [None, "sigmoid", "log-sigmoid", "l1-hinge", "log-l4-hinge", "noop"]:
        for Xq in (X, np.ascontiguousarray(X.toarray())):
            assert_same_topk(om.predict(Xq, beam_size=5, only_topk=7, post_processor=pp),
                             rm.predict(Xq, beam_size=5, only_topk=7, post_processor=pp), exact_scores=True, what=str(pp))


def test_oracle_sparse_inner_products_kat(oracle_mod):
    # test/pecos/core/test_clib.py:39-69 known-answer test
    X = smat.csr_matrix([[1.0, 0.0], [0.5, 0.5], [0.0, 1.0]], dtype=np.float32)
    Y = smat.csr_matrix([[0.5, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 0.5]], dtype=np.float32)
    gt = np.array([[0.50, 0.00, 1.00, 0.00], [0.25, 0.50, 0.50, 0.25], [0.00, 1.00, 0.00, 0.50]], dtype=np.float32)
    r = np.array([0, 1, 2], dtype=np.uint32); c = np.array([1, 2, 3], dtype=np.uint32)
    true = np.array([gt[i, j] for i, j in zip(r, c)], dtype=np.float32)
    W = Y.T.tocsc()
    for Xq, Wq in [(X, W), (X.toarray(), W), (X, np.asfortranarray(W.toarray())), (X.toarray(), np.asfortranarray(W.toarray()))]:
        assert np.allclose(oracle_mod.sparse_inner_products(Xq, Wq, r, c), true, atol=1e-9)


def test_oracle_sparse_inner_products_pinned_on_reference(oracle_mod):
    # the restatement of do_dot_product (matrix.hpp:836-877) against the REAL reference's c_sparse_inner_products_*
    # on random pairs, all four layout combinations, bit for bit (the KAT above only holds 3 pairs)
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    A = smat.random(300, 700, density=0.05, format="csr", dtype=np.float32, random_state=6); A.sort_indices()
    B = smat.random(700, 400, density=0.08, format="csc", dtype=np.float32, random_state=7); B.sort_indices()
    A = A.tolil(); A[3, :] = 0; A[4, :] = rng.standard_normal(700).astype(np.float32); A = A.tocsr().astype(np.float32); A.sort_indices()
    rr = rng.integers(0, 300, 8000).astype(np.uint32); cc = rng.integers(0, 400, 8000).astype(np.uint32)
    for Aq, Bq in [(A, B), (np.ascontiguousarray(A.toarray()), B), (A, np.asfortranarray(B.toarray())),
                   (np.ascontiguousarray(A.toarray()), np.asfortranarray(B.toarray()))]:
        got = oracle_mod.sparse_inner_products(Aq, Bq, rr, cc)
        ref = oracle_mod.ref_sparse_inner_products(Aq, Bq, rr, cc)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_oracle_selected_outputs(manifest, oracle_mod):
    # reference golden: `predict -so Yt_pred.npz` reproduces Yt_pred (test_xlinear.py:368-383), and the
    # CSC-route restatement is bit-identical to the real reference (values AND output order) when present
    Xt = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"))
    G = smat.load_npz(os.path.join(GOLDEN, "ref_fixtures", "Yt_pred.npz")).tocsr()
    om = oracle_mod.OracleModel.load(os.path.join(GOLDEN, "models", "mls10"))
    assert np.allclose(om.predict_on_selected_outputs(Xt, G).toarray(), G.toarray(), atol=1e-6)
    if not oracle_mod.ref_available():
        return
    for name in ["s_eurlex", "s_deep", "s_nobias", "s_flat"]:
        folder = os.path.join(GOLDEN, "synth", name)
        X = load_X(os.path.join(GOLDEN, "synth", name + "__X.npz"))
        om = oracle_mod.OracleModel.load(folder); rm = oracle_mod.RefModel(folder, "CSC")
        S = om.predict(X, beam_size=6, only_topk=8)
        for pp in (None, "sigmoid", "log-l2-hinge"):
            for Xq in (X, np.ascontiguousarray(X.toarray())):
                assert_same_topk(om.predict_on_selected_outputs(Xq, S, pp), rm.predict_on_selected_outputs(Xq, S, pp),
                                 exact_scores=True, what=f"{name} {pp}")


def test_mmap_writer_is_readable_by_the_reference(oracle_mod, tmp_path):
    # c_xlinear_compile_mmap_model of libxrl_amd.so (host-only) writes the reference's byte layout: the REAL
    # reference loads our folders and predicts exactly what it predicts from the npz model; file sizes equal the
    # reference's own compile output, C/perm stores are byte-identical (W holds stale pointers in the reference's)
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    from pecos_amd import clib
    for kind, name in (("models", "splits2"), ("synth", "s_eurlex"), ("synth", "s_pruned"), ("synth", "s_flat")):
        src = os.path.join(GOLDEN, kind, name)
        X = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz") if kind == "models" else os.path.join(GOLDEN, "synth", name + "__X.npz"))
        ours, ref = str(tmp_path / (name + "_ours")), str(tmp_path / (name + "_ref"))
        clib.xlinear_compile_mmap_model(src + "/ranker", ours)
        oracle_mod.ref_compile_mmap_model(src + "/ranker", ref)
        a = oracle_mod.RefModel(src).predict(X, beam_size=5, only_topk=6)
        b = oracle_mod.RefModel(ours, mmap=True).predict(X, beam_size=5, only_topk=6)
        assert a.shape == b.shape
        assert_same_topk(b, a, exact_scores=True, what=name)
        for layer in sorted(os.listdir(ref)):
            if not layer.endswith(".model"):
                continue
            for f in os.listdir(os.path.join(ref, layer)):
                fo, fr = os.path.join(ours, layer, f), os.path.join(ref, layer, f)
                assert os.path.getsize(fo) == os.path.getsize(fr), (name, layer, f)
                if f in ("C.mmap_store", "perm.mmap_store"):
                    assert open(fo, "rb").read() == open(fr, "rb").read(), (name, layer, f)
    with pytest.raises(RuntimeError):
        clib.xlinear_compile_mmap_model(str(tmp_path / "nope"), str(tmp_path / "out"))


def test_oracle_hash_chunked_arithmetic_vs_reference(manifest, oracle_mod):
    # weight_matrix_type=HASH_CHUNKED with sparse queries (inference.hpp:705-735: bias first, then the query's features ascending):
    # the restatement's second arithmetic, pinned bit for bit on outputs of the compiled reference (make_golden_r03.py)
    models = {}
    differs = 0
    for c in manifest["synth_hash"]:
        if c["model"] not in models:
            models[c["model"]] = oracle_mod.OracleModel.load(os.path.join(GOLDEN, "synth", c["model"]), "HASH_CHUNKED")
        X = load_X(os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"))
        P = models[c["model"]].predict(X, **c["kwargs"])
        G = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"]))
        assert_same_topk(P, G, exact_scores=True, what=f"HASH_CHUNKED {c}")
        B = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"].replace("__hash__", "__")))
        differs += int(not np.array_equal(B.data.view(np.uint32), G.data.view(np.uint32)))
    assert differs > 10      # the fixtures do tell the two layouts apart
