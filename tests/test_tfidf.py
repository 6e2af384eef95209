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


_REFPY = os.path.abspath(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "refpy"))


def _live_reference_predict(folder, docs, tmp_path):
    """The reference's own c_tfidf_load + c_tfidf_predict, run here in a subprocess (oracle/_ref/refpy; never on the GPU box)."""
    import subprocess
    import sys
    json.dump(docs, open(tmp_path / "docs.json", "w"), ensure_ascii=False)
    code = (f"import sys, json, numpy as np; sys.path.insert(0, {_REFPY!r})\n"
            "from pecos.utils.featurization.text.vectorizers import Tfidf\n"
            f"X = Tfidf.load({folder!r}).predict(json.load(open({str(tmp_path / 'docs.json')!r}))).tocsr()\n"
            f"np.savez({str(tmp_path / 'live.npz')!r}, indptr=X.indptr, indices=X.indices, data=X.data.astype(np.float32), shape=np.asarray(X.shape))\n")
    subprocess.check_call([sys.executable, "-c", code])
    return np.load(tmp_path / "live.npz")


def _ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    return int(np.max(np.abs(ia - ib))) if len(a) else 0


@pytest.mark.parametrize("name", _names())
def test_oracle_restatement_vs_reference_goldens(name):
    # the CPU restatement (oracle/tfidf_oracle.py: counts + float32 weighting) against what the REFERENCE produced: pattern identical, values
    # bit for bit (with sublinear tf numpy's logf may differ from glibc's in the last bit)
    from oracle.tfidf_oracle import TfidfOracle
    folder, corpus, X = _case(name)
    O = TfidfOracle(folder)
    assert O.nr_features == X.shape[1]
    indptr, idx, val = O.predict(corpus)
    assert np.array_equal(indptr, X.indptr.astype(np.uint64)) and np.array_equal(idx, X.indices.astype(np.uint32)), name
    sub = any(b.sublinear_tf for b in O.base)
    assert _ulp_diff(val, X.data.astype(np.float32)) <= (1 if sub else 0), name


def _fuzz_corpus(rng, tokens, n_docs, max_len):
    """Documents over `tokens` (bytes) plus unknown ones, runs of spaces, empty documents."""
    junk = [b"zzq", b"unknown-token-longer-than-eight", "éè".encode(), "日本".encode(), b"a", b"  ", b""]
    docs = []
    for _ in range(n_docs):
        n = int(rng.integers(0, max_len))
        parts = [tokens[int(i)] if rng.random() < 0.85 else junk[int(rng.integers(len(junk)))] for i in rng.integers(0, len(tokens), size=n)]
        sep = b" " if rng.random() < 0.8 else b"  "
        d = sep.join(parts)
        if rng.random() < 0.2:
            d = b" " + d + b" "
        docs.append(d)
    docs[0] = b""
    docs[1] = b"     "
    return docs


def _check_counts(clib, h, O, corpus, what):
    indptr, idx, val = O.counts(corpus)
    for threads in (1, 4):
        C = clib.tfidf_counts(h, corpus, threads=threads)
        assert C.shape == (len(corpus), O.nr_features), what
        assert np.array_equal(C.indptr.astype(np.uint64), indptr) and np.array_equal(C.indices.astype(np.uint32), idx), what
        assert np.array_equal(C.data.astype(np.float32), val), what


@pytest.mark.parametrize("dense_limit", [None, "0"])
@pytest.mark.parametrize("name", _names())
def test_host_half_counts_vs_oracle_fuzz(name, dense_limit, monkeypatch):
    # the host half's COUNTS (ids and values) against the restatement on fuzzed corpora, on both counting paths: dense counters + bitmap walk
    # (the default) and the per-document sort (XRL_TFIDF_DENSE_LIMIT=0, what a model of > 16 M features uses)
    from oracle.tfidf_oracle import TfidfOracle
    from pecos_amd import clib
    if dense_limit is not None:
        monkeypatch.setenv("XRL_TFIDF_DENSE_LIMIT", dense_limit)
    folder, corpus, X = _case(name)
    O = TfidfOracle(folder)
    h = clib.tfidf_load(folder)
    try:
        _check_counts(clib, h, O, [c.encode("utf-8") for c in corpus], name)
        rng = np.random.default_rng(abs(hash(name)) % 1000 + 17)
        toks = sorted(set().union(*[set(b.vocab) for b in O.base]))
        if O.base[0].tok_type != 10:          # character vocabularies: build documents from words of those characters
            toks = [b"".join(toks[int(i)] for i in rng.integers(0, len(toks), size=int(rng.integers(1, 9)))) for _ in range(200)]
        _check_counts(clib, h, O, _fuzz_corpus(rng, toks, 300, 90), name + " fuzz")
        long_doc = b" ".join(toks[int(i)] for i in rng.integers(0, min(len(toks), 12), size=6000))     # counts in the hundreds, one very long row
        _check_counts(clib, h, O, [long_doc, b"", long_doc[:50]], name + " long")
    finally:
        clib.tfidf_destruct(h)


def _write_base(folder, tok_type, vocab_lines, kwargs, features):
    os.makedirs(os.path.join(folder, "tokenizer")); os.makedirs(os.path.join(folder, "vectorizer"))
    json.dump({"token_type": tok_type}, open(os.path.join(folder, "tokenizer", "config.json"), "w"))
    with open(os.path.join(folder, "tokenizer", "vocab.txt"), "wb") as f:
        f.write(f"{len(vocab_lines)}\n".encode())
        for idx, tok in vocab_lines:
            f.write(f"{idx}\t".encode() + tok + b"\n")
    json.dump({"type": "tfidf", "kwargs": kwargs}, open(os.path.join(folder, "vectorizer", "config.json"), "w"))
    with open(os.path.join(folder, "vectorizer", "tfidf-model.txt"), "w") as f:
        f.write(f"{len(features)}\n")
        for fid, idf, toks in features:
            f.write(f"{fid} {idf!r} {len(toks)}" + "".join(f" {t}" for t in toks) + "\n")


@pytest.mark.parametrize("dense_limit", [None, "0"])
def test_host_half_handmade_models(tmp_path, dense_limit, monkeypatch):
    # model files written by hand to reach what a trained model seldom holds: tokens of exactly 8, 9 and 40 bytes, a token listed twice
    # (the last index wins), unigrams whose token index lies far outside the vocabulary, an n-gram that names the UNKNOWN token (-1),
    # 3- / 4- / 5-grams, an n-gram listed twice (the last feature id wins), truncation in the middle of an n-gram
    from oracle.tfidf_oracle import TfidfOracle
    from pecos_amd import clib
    if dense_limit is not None:
        monkeypatch.setenv("XRL_TFIDF_DENSE_LIMIT", dense_limit)
    rng = np.random.default_rng(3)
    words = [b"a", b"bb", b"ccc", b"12345678", b"123456789", b"x" * 40, b"x" * 41, "日本語".encode(), b"dup", b"e", b"f", b"g"]
    vocab = [(i, w) for i, w in enumerate(words)] + [(77, b"dup"), (5000000, b"far")]
    kw = dict(ngram_range=[1, 5], max_length=-1, binary=False, use_idf=True, sublinear_tf=False, norm_p="l2", min_df_ratio=0.0, max_df_ratio=1.0,
              min_df_cnt=0, max_df_cnt=-1, add_one_idf=False, keep_frequent_feature=True, smooth_idf=True, max_feature=0)
    grams = [(0,), (1,), (2,), (3,), (4,), (5,), (6,), (7,), (77,), (5000000,), (0, 1), (1, 0), (3, 4), (77, 0), (0, -1), (-1,), (-1, -1, 2),
             (0, 1, 2), (2, 1, 0), (0, 0, 0, 0), (1, 2, 3, 4, 5), (9, 10, 11), (0, 1)]
    feats = [(i, 1.0 + 0.25 * i, g) for i, g in enumerate(grams)]
    # ("small": without the far token index every n-gram up to 9 tokens packs into one u64, and the ones naming -1 still need the general table)
    for case, over in (("neg", {}), ("trunc", dict(max_length=7)), ("noneg", None), ("small", {})):
        d = str(tmp_path / f"{case}_{dense_limit}")
        fs = feats if over is not None else [(i, f[1], f[2]) for i, f in enumerate(f for f in feats if all(t >= 0 for t in f[2]))]
        if case == "small":
            fs = [(i, f[1], f[2]) for i, f in enumerate(f for f in feats if 5000000 not in f[2])]
        _write_base(d, 10, vocab, dict(kw, **(over or {})), fs)
        O = TfidfOracle(d)
        h = clib.tfidf_load(d)
        try:
            pool = words + [b"far", b"unk", b"another-unknown-token"]
            docs = [b" ".join(pool[int(i)] for i in rng.integers(0, len(pool), size=int(rng.integers(0, 40)))) for _ in range(400)]
            docs += [b"a bb ccc", b"a a a a a a a a", b"unk ccc", b"unk unk ccc", b"a unk", b"dup a", b"far", b"bb ccc 12345678 123456789 " + b"x" * 40]
            _check_counts(clib, h, O, docs, case)
        finally:
            clib.tfidf_destruct(h)
        if dense_limit is None and os.path.isdir(_REFPY):      # the restatement itself on this odd model, against the live reference
            z = _live_reference_predict(d, [x.decode("utf-8") for x in docs], tmp_path)
            ip, ix, v = O.predict(docs)
            assert np.array_equal(z["indptr"], ip) and np.array_equal(z["indices"], ix) and np.array_equal(z["data"].view(np.uint32), v.view(np.uint32)), case


def test_host_half_vs_live_reference_pattern(oracle_mod, tmp_path):
    # where the reference is built here (oracle/_ref/refpy): a larger random corpus through the reference's own c_tfidf_predict; the host
    # half must produce exactly its rows and feature ids (the weights are the device kernel's half: -m gpu tests)
    refpy = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "refpy")
    if not (oracle_mod.ref_available() and os.path.isdir(refpy)):
        pytest.skip("oracle/_ref/refpy not built")
    import subprocess
    import sys
    from oracle.tfidf_oracle import TfidfOracle
    from pecos_amd import clib
    for name in ("ensemble_word_char", "charwb_bigram", "word_sublinear_addone"):
        folder, corpus, _ = _case(name)
        rng = np.random.default_rng(11)
        words = [f"w{i}" for i in range(250)] + ["naïve", "日本", "語", "café", "x", "unk1", "unk2"]
        big = [" ".join(rng.choice(words, size=int(rng.integers(0, 150)))) for _ in range(1500)]
        json.dump(big, open(tmp_path / "big.json", "w"), ensure_ascii=False)
        code = (f"import sys, json, numpy as np; sys.path.insert(0, {os.path.abspath(refpy)!r})\n"
                "from pecos.utils.featurization.text.vectorizers import Tfidf\n"
                f"X = Tfidf.load({folder!r}).predict(json.load(open({str(tmp_path / 'big.json')!r}))).tocsr()\n"
                f"np.savez({str(tmp_path / 'ref.npz')!r}, indptr=X.indptr, indices=X.indices, data=X.data.astype(np.float32), shape=np.asarray(X.shape))\n")
        subprocess.check_call([sys.executable, "-c", code])
        z = np.load(tmp_path / "ref.npz")
        h = clib.tfidf_load(folder)
        try:
            C = clib.tfidf_counts(h, big, threads=4)
        finally:
            clib.tfidf_destruct(h)
        assert tuple(z["shape"]) == C.shape and np.array_equal(C.indptr, z["indptr"]) and np.array_equal(C.indices, z["indices"]), name
        # and the restatement's weighting of THESE counts reproduces the reference's values
        O = TfidfOracle(folder)
        _, _, val = O.predict(big[:300])
        n = int(z["indptr"][300])
        assert _ulp_diff(val, z["data"][:n]) <= (1 if any(b.sublinear_tf for b in O.base) else 0), name


def test_corpus_packing_forms():
    # the Python side packs the corpus into one buffer + a pointer table (pecos_amd.core._corpus_arrays): str (ASCII fast path and not),
    # bytes, mixed lists, an empty corpus, an embedded NUL (the old c_char_p array cut documents there) all give the same matrix
    from pecos_amd import clib
    folder, corpus, X = _case("ensemble_word_char")
    h = clib.tfidf_load(folder)
    try:
        A = clib.tfidf_counts(h, corpus)
        B = clib.tfidf_counts(h, [c.encode("utf-8") for c in corpus])
        M = clib.tfidf_counts(h, [c.encode("utf-8") if i % 2 else c for i, c in enumerate(corpus)])
        ascii_only = [c for c in corpus if c.isascii()]
        S = clib.tfidf_counts(h, ascii_only)
        for other in (B, M):
            assert np.array_equal(A.indptr, other.indptr) and np.array_equal(A.indices, other.indices) and np.array_equal(A.data, other.data)
        keep = np.array([i for i, c in enumerate(corpus) if c.isascii()])
        assert (S != A[keep]).nnz == 0 and len(ascii_only) < len(corpus)
        E = clib.tfidf_counts(h, [])
        assert E.shape == (0, X.shape[1]) and E.nnz == 0
        two = clib.tfidf_counts(h, ["w1 w2", "w1\x00w2 w3"])
        assert two.shape[0] == 2 and two.indptr[2] >= two.indptr[1]          # the NUL is an ordinary (unknown) byte inside a token, not the end of the document
        ref = clib.tfidf_counts(h, ["w1 w2", "w3"])
        assert set(ref[1].indices) <= set(two[1].indices)
    finally:
        clib.tfidf_destruct(h)


def test_preprocessor_folder_forms(tmp_path):
    # Text2Text keeps a Preprocessor (preprocess.py:22-88): its folder is the vectorizer's files plus config.json {"type": ...}; a folder without
    # config.json is a tfidf one (vectorizers.py:75-79); other vectorizer types have no device path and say so
    import shutil
    from pecos_amd import clib
    from pecos_amd.features import Preprocessor
    folder, corpus, X = _case("word_bigram_trunc")
    p0 = Preprocessor.load(folder)                                           # no config.json
    assert p0.config["type"] == "tfidf" and p0.nr_features == X.shape[1]
    d = str(tmp_path / "pre")
    shutil.copytree(folder, d)
    json.dump({"type": "tfidf", "kwargs": {"ngram_range": [1, 2]}}, open(os.path.join(d, "config.json"), "w"))
    p1 = Preprocessor.load(d)
    assert p1.nr_features == X.shape[1]
    C = clib.tfidf_counts(p1.vectorizer.model, corpus)                       # (host half: no GPU needed)
    assert np.array_equal(C.indptr, X.indptr) and np.array_equal(C.indices, X.indices)
    with pytest.raises(NotImplementedError):
        p1.predict("corpus.txt")
    json.dump({"type": "hashing", "kwargs": {}}, open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(NotImplementedError):
        Preprocessor.load(d)
    json.dump({"kwargs": {}}, open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(ValueError):
        Preprocessor.load(d)


def _ref_sorted_csr(csr, only_topk=None):
    """Restatement of smat_util.sorted_csr_from_coo (smat_util.py:174-210), the per-row loop, as the checker of the vectorised mirror."""
    c = smat.csr_matrix(csr, copy=True); c.sum_duplicates(); c.sort_indices()
    ip, ix, dv = [0], [], []
    for i in range(c.shape[0]):
        a, b = c.indptr[i], c.indptr[i + 1]
        o = np.argsort(-c.data[a:b], kind="mergesort")
        if only_topk is not None:
            o = o[: max(min(1, only_topk), only_topk)]
        ix += list(c.indices[a:b][o]); dv += list(c.data[a:b][o]); ip.append(len(ix))
    return np.array(ip), np.array(ix, dtype=np.int64), np.array(dv, dtype=c.data.dtype)


def test_text2text_finish_matches_the_reference_semantics():
    # Text2Text.predict after the per-model predictions (model.py:418-427): CsrEnsembler.average, threshold, sorted_csr(only_topk) --
    # ties broken by ascending label id, NOT by the beam search's positional order
    from pecos_amd.features import Text2Text, ensemble_average, sorted_csr
    rng = np.random.default_rng(2)
    n, L = 60, 40

    def rand_pred(seed):
        r = np.random.default_rng(seed)
        rows = []
        for _ in range(n):
            k = int(r.integers(0, 9))
            cols = r.choice(L, size=k, replace=False)
            vals = np.round(r.random(k), 1).astype(np.float32)          # many exact ties
            rows.append((cols, vals))
        ip = np.cumsum([0] + [len(c) for c, _ in rows])
        return smat.csr_matrix((np.concatenate([v for _, v in rows]) if ip[-1] else np.zeros(0, np.float32),
                                np.concatenate([c for c, _ in rows]) if ip[-1] else np.zeros(0, np.int64), ip), shape=(n, L))
    A, B = rand_pred(1), rand_pred(2)
    for M in (A, B, (A + B).tocsr()):
        for topk in (None, 1, 3, 100):
            got = sorted_csr(M, only_topk=topk)
            ip, ix, dv = _ref_sorted_csr(M, topk)
            assert np.array_equal(got.indptr, ip) and np.array_equal(got.indices, ix) and np.array_equal(got.data, dv), topk
    avg = ensemble_average([A, B])
    ip, ix, dv = _ref_sorted_csr((A + B).tocsr())
    assert np.array_equal(avg.indices, ix) and np.array_equal(avg.data, dv / np.float32(2))
    out = Text2Text.finish([A, B], threshold=0.3, only_topk=4)
    S = (A + B).tocsr(); S = _ref_sorted_csr(S)
    Y = smat.csr_matrix((S[2] / np.float32(2), S[1], S[0]), shape=(n, L)); Y.data[Y.data <= 0.3] = 0; Y.eliminate_zeros()
    ip, ix, dv = _ref_sorted_csr(Y, 4)
    assert np.array_equal(out.indptr, ip) and np.array_equal(out.indices, ix) and np.array_equal(out.data, dv)
    one = Text2Text.finish([A], only_topk=2)
    ip, ix, dv = _ref_sorted_csr(A, 2)
    assert np.array_equal(one.indptr, ip) and np.array_equal(one.indices, ix) and np.array_equal(one.data, dv)
    assert rng is not None


def test_sorted_csr_and_average_vs_live_reference(tmp_path):
    # where the reference's python package is at hand (oracle/_ref/refpy, this container only): smat_util.sorted_csr / CsrEnsembler.average
    # themselves on random matrices full of ties, with NaN, -0.0 and 0.0 among the values
    if not os.path.isdir(_REFPY):
        pytest.skip("oracle/_ref/refpy not built")
    import subprocess
    import sys
    code = f"""
import sys, numpy as np, scipy.sparse as smat
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {_REFPY!r})
from pecos.utils import smat_util
from pecos_amd.features import sorted_csr, ensemble_average
bad = 0
for t in range(25):
    M = smat.random(50, 30, density=0.3, format='csr', dtype=np.float32, random_state=t); M.data = np.round(M.data, 1).astype(np.float32)
    if t % 5 == 0 and M.nnz > 3: M.data[:3] = [np.nan, -0.0, 0.0]
    for k in (None, 1, 5):
        a = sorted_csr(M, only_topk=k); b = smat_util.sorted_csr(M, only_topk=k)
        bad += not (np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data, equal_nan=True))
    N = smat.random(50, 30, density=0.3, format='csr', dtype=np.float32, random_state=100 + t); N.data = np.round(N.data, 1).astype(np.float32)
    a = ensemble_average([M, N]); b = smat_util.CsrEnsembler.average(M, N)
    bad += not (np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data, equal_nan=True))
sys.exit(1 if bad else 0)
"""
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_host_half_properties_at_scale(tmp_path, monkeypatch):
    # size-independent properties on a corpus far beyond what the Python restatement can check (60 k documents, 4.5 M tokens, a 40 k-word
    # vectorizer written here in the reference's file format): (1) the result does not depend on the thread count or on the counting path;
    # (2) LINEARITY of a unigram vectorizer: counts("a b") = counts("a") + counts("b"); (3) every row's counts sum to the number of known tokens;
    # (4) a permutation of the documents permutes the rows
    from pecos_amd import clib
    rng = np.random.default_rng(5)
    V, n = 40000, 60000
    words = [f"t{i:x}".encode() for i in range(V)]
    d = str(tmp_path / "uni")
    kw = dict(ngram_range=[1, 1], max_length=-1, binary=False, use_idf=True, sublinear_tf=False, norm_p="l2", min_df_ratio=0.0, max_df_ratio=1.0,
              min_df_cnt=0, max_df_cnt=-1, add_one_idf=False, keep_frequent_feature=True, smooth_idf=True, max_feature=0)
    perm = rng.permutation(V)
    _write_base(d, 10, [(i, w) for i, w in enumerate(words)], kw, [(int(perm[i]), 1.0 + (i % 7), (i,)) for i in range(V)])
    p = 1.0 / np.arange(1, V + 1) ** 1.07; p /= p.sum()
    lens = np.maximum(1, rng.poisson(75, size=n))
    ids = rng.choice(V, size=int(lens.sum()), p=p)
    unk = rng.random(len(ids)) < 0.02                                   # 2 % unknown tokens
    toks = np.array(words + [b"zz-unknown"], dtype=object)[np.where(unk, V, ids)]
    off = np.concatenate([[0], np.cumsum(lens)])
    docs = [b" ".join(toks[off[i]:off[i + 1]]) for i in range(n)]
    h = clib.tfidf_load(d)
    try:
        A = clib.tfidf_counts(h, docs, threads=1)
        B = clib.tfidf_counts(h, docs, threads=7)
        monkeypatch.setenv("XRL_TFIDF_DENSE_LIMIT", "0")
        S = clib.tfidf_counts(h, docs, threads=3)
        monkeypatch.delenv("XRL_TFIDF_DENSE_LIMIT")
        for other in (B, S):
            assert np.array_equal(A.indptr, other.indptr) and np.array_equal(A.indices, other.indices) and np.array_equal(A.data, other.data)
        known = np.add.reduceat((~unk).astype(np.int64), off[:-1])
        assert np.array_equal(np.asarray(A.sum(axis=1)).ravel().astype(np.int64), known)
        half = n // 2
        joined = [docs[i] + b" " + docs[half + i] for i in range(half)]
        J = clib.tfidf_counts(h, joined, threads=4)
        want = (A[:half] + A[half:2 * half]).tocsr(); want.sort_indices()
        assert np.array_equal(J.indptr, want.indptr) and np.array_equal(J.indices, want.indices) and np.array_equal(J.data, want.data)
        order = rng.permutation(n)
        P = clib.tfidf_counts(h, [docs[i] for i in order], threads=5)
        assert (P != A[order]).nnz == 0
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
    # calls of <= 4 documents are weighted on the HOST (the reference's nr_doc == 1 path is a host call too, libpecos.cpp:437-439); XRL_TFIDF_HOST_DOCS=0 sends
    # them to the device: both must give the golden rows (host: glibc's logf like the reference -> bit-identical also with sublinear_tf)
    for nd in (1, 3, 4):
        for env in (None, "0"):
            if env is None:
                os.environ.pop("XRL_TFIDF_HOST_DOCS", None)
            else:
                os.environ["XRL_TFIDF_HOST_DOCS"] = env
            try:
                few = vec.predict(corpus[1: 1 + nd], threads=1)
            finally:
                os.environ.pop("XRL_TFIDF_HOST_DOCS", None)
            lo, hi = X.indptr[1], X.indptr[1 + nd]
            assert few.shape == (nd, X.shape[1]) and np.array_equal(few.indptr, X.indptr[1: 2 + nd] - lo) and np.array_equal(few.indices, X.indices[lo:hi]), (name, nd, env)
            if sub and env == "0":
                assert np.allclose(few.data, X.data[lo:hi], rtol=3e-7, atol=0)
            else:
                assert np.array_equal(few.data.view(np.uint32), X.data[lo:hi].astype(np.float32).view(np.uint32)), (name, nd, env)
    with pytest.raises(RuntimeError):
        vec.predict([])                              # Invalid nr_doc 0 (libpecos.cpp:442-444)


def _file_cases():
    d = os.path.join(GOLDEN, "tfidf_files")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".txt")) if os.path.isdir(d) else []


def _file_docs(raw):
    """The documents the reference's reader makes of a file (tfidf.hpp:279-294): every newline ends one, a last line without a newline keeps the NUL its buffer ends with."""
    docs, start = [], 0
    for i, b in enumerate(raw):
        if b == 10:
            docs.append(raw[start:i]); start = i + 1
    if start < len(raw):
        docs.append(raw[start:] + b"\0")
    return docs


@pytest.mark.parametrize("case", _file_cases())
def test_file_reader_documents_have_the_reference_pattern(case):
    # host only: the term-count pattern of the documents cut from the file == the pattern of the reference's predict_from_file output
    from pecos_amd import clib
    name = case.split("__")[0]
    raw = open(os.path.join(GOLDEN, "tfidf_files", case + ".txt"), "rb").read()
    z = np.load(os.path.join(GOLDEN, "tfidf_files", case + ".npz"))
    h = clib.tfidf_load(os.path.join(GOLDEN, "tfidf_models", name, "model"))
    try:
        C = clib.tfidf_counts(h, _file_docs(raw))
    finally:
        clib.tfidf_destruct(h)
    assert C.shape == tuple(z["shape"]) and np.array_equal(C.indptr, z["indptr"]) and np.array_equal(C.indices, z["indices"]), case


@pytest.mark.gpu
@pytest.mark.parametrize("case", _file_cases())
def test_c_tfidf_predict_from_file_vs_reference_goldens(case):
    # libpecos.cpp:413-425 through the drop-in entry point: one document per line, blank lines are documents, a last line without a newline
    from pecos_amd.features import Tfidf
    name = case.split("__")[0]
    z = np.load(os.path.join(GOLDEN, "tfidf_files", case + ".npz"))
    P = Tfidf.load(os.path.join(GOLDEN, "tfidf_models", name, "model")).predict(os.path.join(GOLDEN, "tfidf_files", case + ".txt"))
    assert P.shape == tuple(z["shape"]) and np.array_equal(P.indptr, z["indptr"]) and np.array_equal(P.indices, z["indices"]), case
    assert np.array_equal(P.data.view(np.uint32), z["data"].astype(np.float32).view(np.uint32)), case


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
def test_text2text_mirror_device_resident(oracle_mod, tmp_path):
    # pecos.apps.text2text.model.Text2Text, predict half: the folder layout its save() writes (preprocessor/ + config.json, xlinear_ensemble/{config.json,
    # 0, 1}, output_items.json), loaded and predicted through the device-resident pipeline, against the reference's own steps done on the host:
    # X = the reference vectorizer's output (golden), every model through the oracle / compiled reference, CsrEnsembler.average, threshold, sorted_csr
    import shutil
    import xrl_synth
    from pecos_amd.features import Text2Text
    folder, corpus, X = _case("word_bigram_trunc")          # (no sublinear tf: X is bit-identical to the reference's, so the scores are too)
    root = tmp_path / "t2t"
    shutil.copytree(folder, root / "preprocessor")
    json.dump({"type": "tfidf", "kwargs": {}}, open(root / "preprocessor" / "config.json", "w"))
    os.makedirs(root / "xlinear_ensemble")
    D, L = X.shape[1], 500
    for i, seed in enumerate((71, 72)):
        xrl_synth.make_model(str(root / "xlinear_ensemble" / str(i)), D, L, [100, 60, 20], seed=seed, shape=[5, 50, L])
    json.dump({"nr_ensembles": 2, "kwargs": [{}, {}]}, open(root / "xlinear_ensemble" / "config.json", "w"))
    json.dump([f"item {i}" for i in range(L)], open(root / "output_items.json", "w"))
    t2t = Text2Text.load(str(root))
    assert len(t2t.xlinear_models) == 2 and t2t.get_output_item(3) == "item 3" and t2t.preprocessor.nr_features == D
    Xs = X.astype(np.float32).tocsr(); Xs.sort_indices()
    refs = []
    for i in range(2):
        mdir = str(root / "xlinear_ensemble" / str(i))
        refs.append(oracle_mod.RefModel(mdir) if oracle_mod.ref_available() else oracle_mod.OracleModel.load(mdir))
    for kw, thr in ((dict(beam_size=5, only_topk=6), None), (dict(beam_size=8, only_topk=4, post_processor="l3-hinge"), 0.2)):
        got = t2t.predict(corpus, threshold=thr, **kw)
        want = [r.predict(Xs, **kw) for r in refs]
        S = (want[0] + want[1]).tocsr()
        ip, ix, dv = _ref_sorted_csr(S)
        Y = smat.csr_matrix((dv / np.float32(2), ix, ip), shape=S.shape)
        if thr is not None:
            Y.data[Y.data <= thr] = 0; Y.eliminate_zeros()
        ip, ix, dv = _ref_sorted_csr(Y, kw["only_topk"])
        assert np.array_equal(got.indptr, ip) and np.array_equal(got.indices, ix), kw
        assert np.array_equal(got.data.view(np.uint32), dv.astype(np.float32).view(np.uint32)), kw
    one = Text2Text(t2t.preprocessor, t2t.xlinear_models[:1], t2t.output_items).predict(corpus, beam_size=5, only_topk=6)
    ip, ix, dv = _ref_sorted_csr(refs[0].predict(Xs, beam_size=5, only_topk=6), 6)
    assert np.array_equal(one.indptr, ip) and np.array_equal(one.indices, ix) and np.array_equal(one.data.view(np.uint32), dv.view(np.uint32))


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


def test_bench_text_to_labels_corpus_reproduces_the_query_pattern(tmp_path):
    # bench.py's extra.text_to_labels builds a corpus whose document i names every feature of query row i once and a unigram vectorizer over
    # the model's feature dimension (scripts/n4_producer_bench.write_vectorizer, the reference's file format): the host half's term counts
    # must then have exactly X's sparsity pattern, every count 1 -- so the beam search of that line sees the benchmark's own queries
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "scripts")); sys.path.insert(0, repo)
    import n4_producer_bench as N4
    import xrl_synth
    from pecos_amd import clib
    D, N = 3000, 400
    X = xrl_synth.make_queries(N, D, 40, seed=3, relabel_seed=0)
    words = np.array([f"t{i:x}" for i in range(D)])
    tok = words[X.indices]
    corpus = [" ".join(tok[X.indptr[i]:X.indptr[i + 1]]) for i in range(N)]
    vdir = str(tmp_path / "vec")
    os.makedirs(vdir)
    N4.write_vectorizer(vdir, list(words), [(i,) for i in range(D)], np.random.default_rng(5))
    h = clib.tfidf_load(vdir)
    try:
        C = clib.tfidf_counts(h, corpus, threads=2)
    finally:
        clib.tfidf_destruct(h)
    # feature ids are a permutation of the word ids in that synthetic model file: compare per-row counts and totals
    assert C.shape == (N, D) and np.array_equal(np.diff(C.indptr), np.diff(X.indptr)) and np.all(C.data == 1.0)
