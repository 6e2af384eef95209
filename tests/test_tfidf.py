"""TF-IDF query producer: c_tfidf_load / c_tfidf_predict with the reference's signatures and the device-resident
variant, pinned on vectorizers the REFERENCE trained, saved and predicted with (tests/golden/make_golden_r04.py) and, where oracle/_ref is
built, on the live reference."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import GOLDEN

TF = os.path.join(GOLDEN, "tfidf_models")


def _case(name):
    d = os.path.join(TF, name)
    z = np.load(os.path.join(d, "X.npz"))
    X = smat.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    return os.path.join(d, "model"), json.load(open(os.path.join(d, "corpus.json"))), X


def _names():
    return [c["name"] for c in json.load(open(os.path.join(GOLDEN, "manifest.json")))["tfidf_models"]]


@pytest.mark.parametrize("name", _names())
def test_term_counts_have_the_reference_pattern(name):
    # host half (no GPU): model files, tokenizer (word / char / char_wb, truncation, unknown tokens), n-gram lookup, ensembles --
    # the rows and feature ids must be exactly those of the reference's output, and the counts whole numbers
    from pecos_amd import clib
    folder, corpus, X = _case(name)
    h = clib.tfidf_load(folder)
    try:
        assert clib.tfidf_nr_features(h) == X.shape[1]
        for threads in (1, 3):
            C = clib.tfidf_counts(h, corpus, threads=threads)
            assert C.shape == X.shape and np.array_equal(C.indptr, X.indptr) and np.array_equal(C.indices, X.indices), name
            assert np.all(C.data >= 1.0) and np.array_equal(C.data, np.round(C.data))
        one = clib.tfidf_counts(h, [corpus[0]])
        assert np.array_equal(one.indices, X.indices[: X.indptr[1]])
    finally:
        clib.tfidf_destruct(h)


def test_tfidf_load_errors(tmp_path):
    from pecos_amd import clib
    from pecos_amd.features import Tfidf
    with pytest.raises(ValueError):
        Tfidf.load(str(tmp_path / "nope"))
    with pytest.raises(RuntimeError):
        clib.tfidf_load(str(tmp_path))               # an empty folder: no tokenizer/config.json
    folder, corpus, X = _case(_names()[0])
    h = clib.tfidf_load(folder)
    with pytest.raises(RuntimeError):                # invalid UTF-8 under a char tokenizer is an error in the reference too; a word tokenizer takes any bytes
        hc = clib.tfidf_load(_case("char_trigram")[0])
        try:
            clib.tfidf_counts(hc, [b"\x80abc"])
        finally:
            clib.tfidf_destruct(hc)
    assert clib.tfidf_counts(h, [b"\x80abc w1"]).nnz >= 0
    clib.tfidf_destruct(h)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_c_tfidf_predict_vs_reference_goldens(name, manifest):
    # the drop-in entry point: host CSR through the allocator, bit-identical to the reference's output (sublinear_tf: the device's logf, <= 1 ulp)
    from pecos_amd.features import Tfidf
    folder, corpus, X = _case(name)
    vec = Tfidf.load(folder)
    P = vec.predict(corpus)
    assert P.shape == X.shape and np.array_equal(P.indptr, X.indptr) and np.array_equal(P.indices, X.indices)
    sub = next(c for c in manifest["tfidf_models"] if c["name"] == name)["sublinear"]
    if sub:
        assert np.allclose(P.data, X.data, rtol=3e-7, atol=0)
    else:
        assert np.array_equal(P.data.view(np.uint32), X.data.astype(np.float32).view(np.uint32)), name
    one = vec.predict([corpus[1]], threads=1)        # nr_doc == 1
    assert np.array_equal(one.indices, X.indices[X.indptr[1]: X.indptr[2]])
    with pytest.raises(RuntimeError):
        vec.predict([])                              # Invalid nr_doc 0 (libpecos.cpp:442-444)


@pytest.mark.gpu
def test_text_to_labels_device_resident(oracle_mod, tmp_path):
    # Text2Text.predict's two lines (pecos/apps/text2text/model.py:416-417) with X device-resident: texts -> tf-idf on the GPU -> beam search in
    # place, against the reference doing both steps on the host (its own vectorizer output fed to the oracle / the compiled reference);
    # then the concat model's form (pecos/xmc/xtransformer/model.py:589-603) with a dense embedding block appended on the device
    import torch
    import xrl_synth
    from pecos_amd import XLinearModel as XLM
    from pecos_amd.features import Tfidf, concat_features, predict_text
    folder, corpus, X = _case("word_bigram_trunc")
    vec = Tfidf.load(folder)
    D = X.shape[1]
    mdir = str(tmp_path / "m")
    xrl_synth.make_model(mdir, D, 600, [120, 60, 20], seed=61, shape=[6, 48, 600])
    m = XLM.load(mdir)
    ref = oracle_mod.RefModel(mdir) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(mdir)
    Xs = X.astype(np.float32).tocsr(); Xs.sort_indices()
    for kw in (dict(beam_size=5, only_topk=7), dict(beam_size=10, only_topk=3, post_processor="log-l2-hinge")):
        got = predict_text(vec, m, corpus, **kw)
        want = ref.predict(Xs, **kw)
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices), kw
        assert np.array_equal(got.data.view(np.uint32), want.data.view(np.uint32)), kw
    # two models: the ensemble average of Text2Text
    got2 = predict_text(vec, [m, m], corpus, beam_size=5, only_topk=7)
    assert np.allclose(got2.toarray(), ref.predict(Xs, beam_size=5, only_topk=7).toarray(), rtol=1e-6)
    # concat model: [tf-idf | embedding]
    H = 16
    mdir2 = str(tmp_path / "m2")
    xrl_synth.make_model(mdir2, D + H, 600, [120, 60, 20], seed=62, shape=[6, 48, 600])
    m2 = XLM.load(mdir2)
    ref2 = oracle_mod.RefModel(mdir2) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(mdir2)
    emb = np.random.default_rng(5).standard_normal((len(corpus), H)).astype(np.float32)
    got3 = predict_text(vec, m2, corpus, X_emb=torch.from_numpy(emb).cuda(), normalize_emb=False, beam_size=6, only_topk=6)
    want3 = ref2.predict(concat_features(Xs, emb, normalize_emb=False), beam_size=6, only_topk=6)
    assert np.array_equal(got3.indices, want3.indices) and np.array_equal(got3.data.view(np.uint32), want3.data.view(np.uint32))


@pytest.mark.gpu
def test_c_tfidf_predict_vs_live_reference(oracle_mod, tmp_path):
    # a larger random corpus against the reference's own c_tfidf_predict, run here (oracle/_ref/refpy); skipped where the reference is not built
    refpy = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "refpy")
    if not (oracle_mod.ref_available() and os.path.isdir(refpy)):
        pytest.skip("oracle/_ref/refpy not built")
    import subprocess
    import sys
    folder, corpus, _ = _case("ensemble_word_char")
    rng = np.random.default_rng(9)
    words = [f"w{i}" for i in range(250)] + ["naïve", "日本", "語", "café", "x", "unk1", "unk2"]
    big = [" ".join(rng.choice(words, size=int(rng.integers(0, 120)))) for _ in range(3000)]
    json.dump(big, open(tmp_path / "big.json", "w"), ensure_ascii=False)
    code = (f"import sys, json, numpy as np; sys.path.insert(0, {os.path.abspath(refpy)!r})\n"
            "from pecos.utils.featurization.text.vectorizers import Tfidf\n"
            f"X = Tfidf.load({folder!r}).predict(json.load(open({str(tmp_path / 'big.json')!r}))).tocsr()\n"
            f"np.savez({str(tmp_path / 'ref.npz')!r}, indptr=X.indptr, indices=X.indices, data=X.data.astype(np.float32), shape=np.asarray(X.shape))\n")
    subprocess.check_call([sys.executable, "-c", code])
    z = np.load(tmp_path / "ref.npz")
    from pecos_amd.features import Tfidf
    P = Tfidf.load(folder).predict(big)
    assert tuple(z["shape"]) == P.shape and np.array_equal(P.indptr, z["indptr"]) and np.array_equal(P.indices, z["indices"])
    assert np.array_equal(P.data.view(np.uint32), z["data"].view(np.uint32))
