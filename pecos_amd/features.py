"""Query-side hand-offs for callers that produce X on the GPU or in pieces.

* :func:`concat_features` -- host mirror of ``TransformerMatcher.concat_features``
  (pecos/xmc/xtransformer/matcher.py:864-890): [numerical features | (row-normalised) embeddings], the matrix
  XR-Transformer's ``concat_model`` predicts on (pecos/xmc/xtransformer/model.py:589-603).
* :func:`tfidf_weight` / :func:`predict_tfidf_from_torch` -- the weighting half of the reference's TF-IDF vectorizer
  (``BaseVectorizer::get_sorted_feature``, pecos/core/utils/tfidf.hpp:798-822): term counts -> tf -> x idf -> l1/l2 norm, as a host
  mirror (numpy float32, sequential like the reference) and as a device kernel feeding the beam search without a host round trip.
* :func:`predict_from_torch` -- X already in HBM as torch tensors (a GPU TF-IDF featurizer's CSR -- the reference's
  ``c_tfidf_predict`` produces that CSR on the host, pecos/core/libpecos.cpp:427-445 -- optionally with a dense embedding block to
  append on the device): no host round trip of X, results stay on the device.
"""
import numpy as np
import scipy.sparse as smat

from .core import clib


def concat_features(X_feat, X_emb, normalize_emb=True):
    """matcher.py:864-890 on the host, same operations in the same order (sklearn's ``normalize``, ``dense_to_csr``, ``hstack_csr``)."""
    if normalize_emb:
        from sklearn.preprocessing import normalize as sk_normalize
        X_cat = sk_normalize(X_emb)
    else:
        X_cat = X_emb
    if isinstance(X_feat, smat.csr_matrix):
        # smat_util.dense_to_csr keeps EVERY cell of the dense block as a stored entry (zeros included) and hstack_csr appends
        # it after the row's sparse features (pinned on the reference's outputs, tests/golden/concat/)
        X_cat = np.ascontiguousarray(X_cat, dtype=np.float32)
        n, H = X_cat.shape
        E = smat.csr_matrix((X_cat.ravel(), np.tile(np.arange(H, dtype=np.int64), n), np.arange(n + 1, dtype=np.int64) * H), shape=(n, H))
        X_cat = smat.hstack([X_feat.astype(np.float32), E], format="csr", dtype=np.float32)
    elif isinstance(X_feat, np.ndarray):
        X_cat = np.hstack([X_feat, X_cat])
    elif X_feat is None:
        pass
    else:
        raise TypeError(f"Expected CSR or ndarray, got {type(X_feat)}")
    return X_cat


def predict_from_torch(model, crow, col, val, n_cols, beam_size=None, only_topk=None, post_processor=None, emb=None, stream=None,
                       normalize_emb=False):
    """Beam search on queries that are already on the GPU.

    crow: int64 [rows+1], col: int32 [nnz] (sorted inside every row), val: float32 [nnz] -- CUDA tensors of a CSR with
    ``n_cols`` columns; emb: optional float32 [rows, H] CUDA tensor appended as columns n_cols .. n_cols+H-1 on the device
    (``normalize_emb=True`` l2-normalises its rows on the device first, like the reference's concat_features).  Returns CUDA tensors
    (labels int32 [rows, k], scores float32 [rows, k], counts int32 [rows]); row r holds counts[r] valid entries, best first."""
    import torch
    h = model.model.model_chain
    assert crow.is_cuda and col.is_cuda and val.is_cuda and crow.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
    crow, col, val = crow.contiguous(), col.contiguous(), val.contiguous()
    rows = crow.numel() - 1
    nnz = int(val.numel())
    # outputs first: their zero-fills run on torch's current stream and must be complete (like the inputs, produced on that
    # stream) before the predict starts on `stream`
    k = clib.effective_topk(h, only_topk)
    idx = torch.zeros((rows, k), dtype=torch.int32, device=val.device)
    sc = torch.zeros((rows, k), dtype=torch.float32, device=val.device)
    cnt = torch.zeros((rows,), dtype=torch.int32, device=val.device)
    torch.cuda.current_stream().synchronize()
    if emb is not None:
        assert emb.is_cuda and emb.dtype == torch.float32 and emb.shape[0] == rows
        emb = emb.contiguous()
        q = clib.queries_concat_device(h, rows, n_cols, crow.data_ptr(), col.data_ptr(), val.data_ptr(), nnz, emb.shape[1], emb.data_ptr(),
                                       normalize_emb=normalize_emb)
    else:
        q = clib.queries_from_device_csr(h, rows, n_cols, crow.data_ptr(), col.data_ptr(), val.data_ptr(), nnz)
    try:
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if rows:
            clib.predict_device(h, q, beam_size, post_processor, only_topk, idx.data_ptr(), sc.data_ptr(), cnt.data_ptr(), k,
                                stream=s or None, sync=True)
    finally:
        clib.queries_free(q)
    return idx, sc, cnt


def tfidf_weight(counts, idf=None, binary=False, sublinear_tf=False, norm="l2"):
    """Host mirror of tfidf.hpp:798-822 for a CSR of term counts (sorted column ids): float32, the norm accumulated
    sequentially in ascending feature order, exactly the reference's operations (pinned on its outputs, tests/golden/tfidf/)."""
    C = smat.csr_matrix(counts, dtype=np.float32)
    C.sort_indices()
    out = np.empty(C.nnz, dtype=np.float32)
    f32 = np.float32
    for r in range(C.shape[0]):
        b, e = C.indptr[r], C.indptr[r + 1]
        v = np.ones(e - b, f32) if binary else C.data[b:e].astype(f32)
        if sublinear_tf:
            v = (np.log(v).astype(f32).astype(np.float64) + 1.0).astype(f32)
        if idf is not None:
            v = (v * np.asarray(idf, f32)[C.indices[b:e]]).astype(f32)
        denom = f32(0.0)
        for x in v:
            denom = f32(denom + (f32(abs(x)) if norm == "l1" else f32(x * x)))
        if abs(denom) < np.finfo(np.float32).eps:
            denom = f32(1.0)
        elif norm == "l2":
            denom = f32(np.sqrt(denom))
        out[b:e] = (v / denom).astype(f32)
    return smat.csr_matrix((out, C.indices.copy(), C.indptr.copy()), shape=C.shape)


def predict_tfidf_from_torch(model, crow, col, count, n_cols, idf=None, binary=False, sublinear_tf=False, norm="l2", beam_size=None,
                             only_topk=None, post_processor=None, stream=None):
    """Term-count CSR already on the GPU (crow int64 [rows+1], col int32, count float32; idf float32 [n_cols] CUDA tensor or None)
    -> tf-idf weighting on the device -> beam search; returns (labels, scores, counts) CUDA tensors like predict_from_torch."""
    import torch
    h = model.model.model_chain
    assert crow.is_cuda and col.is_cuda and count.is_cuda and crow.dtype == torch.int64 and col.dtype == torch.int32 and count.dtype == torch.float32
    crow, col, count = crow.contiguous(), col.contiguous(), count.contiguous()
    rows = crow.numel() - 1
    k = clib.effective_topk(h, only_topk)
    idx = torch.zeros((rows, k), dtype=torch.int32, device=count.device)
    sc = torch.zeros((rows, k), dtype=torch.float32, device=count.device)
    cnt = torch.zeros((rows,), dtype=torch.int32, device=count.device)
    torch.cuda.current_stream().synchronize()
    q = clib.queries_tfidf_device(h, rows, n_cols, crow.data_ptr(), col.data_ptr(), count.data_ptr(), int(count.numel()),
                                  idf.data_ptr() if idf is not None else None, binary, sublinear_tf, 1 if norm == "l1" else 2)
    try:
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if rows:
            clib.predict_device(h, q, beam_size, post_processor, only_topk, idx.data_ptr(), sc.data_ptr(), cnt.data_ptr(), k, stream=s or None, sync=True)
    finally:
        clib.queries_free(q)
    return idx, sc, cnt


class Tfidf:
    """The PREDICT half of the reference's ``pecos.utils.featurization.text.vectorizers.Tfidf`` (vectorizers.py:163-308): ``load`` a
    folder the reference saved, ``predict`` a list of strings to a scipy CSR -- same names, arguments and result -- with the tokenizer on
    host threads and the weighting / normalisation on the device; plus ``predict_device``, which leaves X in HBM.  Training and saving
    stay the reference's."""

    def __init__(self, model=None):
        self.model = model

    def __del__(self):
        try:
            clib.tfidf_destruct(self.model)
        except Exception:
            pass

    @classmethod
    def load(cls, load_dir):
        import os
        if not os.path.exists(load_dir):
            raise ValueError(f"tfidf model not exist at {load_dir}")
        return cls(clib.tfidf_load(load_dir))

    @property
    def nr_features(self):
        return clib.tfidf_nr_features(self.model)

    def predict(self, corpus, **kwargs):
        return clib.tfidf_predict(self.model, corpus, buffer_size=kwargs.get("buffer_size", 0), threads=kwargs.get("threads", -1))

    def predict_device(self, xlinear_model, corpus, threads=-1):
        """Texts -> X resident on ``xlinear_model``'s GPU: a query handle (``clib.queries_free`` it) for ``clib.predict_device``."""
        return clib.tfidf_predict_device(self.model, xlinear_model.model.model_chain, corpus, threads)


class Preprocessor:
    """The PREDICT half of ``pecos.utils.featurization.text.preprocess.Preprocessor`` (preprocess.py:22-88), the object ``Text2Text`` keeps as
    ``self.preprocessor``: ``load`` the folder its ``save`` wrote (``config.json`` = {"type": ..., "kwargs": ...} beside the vectorizer's own
    files; a folder without it is a tfidf one, vectorizers.py:75-79) and ``predict`` texts.  Only the ``tfidf`` type has a device path; ``hashing``
    and ``sklearntfidf`` folders raise -- they stay the reference's."""

    def __init__(self, vectorizer=None, config=None):
        self.vectorizer = vectorizer
        self.config = config

    @classmethod
    def load(cls, preprocessor_folder):
        import json
        import os
        cfg_path = os.path.join(preprocessor_folder, "config.json")
        config = {"type": "tfidf", "kwargs": {}}
        if os.path.exists(cfg_path):
            with open(cfg_path, "r", encoding="utf-8") as fin:
                config = json.loads(fin.read())
        vtype = config.get("type", None)
        if vtype is None:
            raise ValueError(f"{preprocessor_folder} is not a valid vectorizer folder")
        if vtype != "tfidf":
            raise NotImplementedError(f"vectorizer type {vtype!r}: only 'tfidf' has a device-resident predict path; use the reference's Preprocessor")
        return cls(Tfidf.load(preprocessor_folder), config)

    @property
    def nr_features(self):
        return self.vectorizer.nr_features

    def predict(self, corpus, **kwargs):
        """Texts -> scipy CSR (the reference's result, bit for bit)."""
        if isinstance(corpus, str):
            raise NotImplementedError("predict from a corpus FILE is not offered by pecos_amd: read the lines and pass a list")
        return self.vectorizer.predict(corpus, **kwargs)

    def predict_device(self, xlinear_model, corpus, threads=-1):
        return self.vectorizer.predict_device(xlinear_model, corpus, threads=threads)


def _predict_handle_to_csr(model, q, rows, beam_size=None, only_topk=None, post_processor=None):
    import torch
    from .distributed import rows_to_csr
    h = model.model.model_chain
    k = clib.effective_topk(h, only_topk)
    dev = torch.device("cuda", clib.xlinear_get_int_attr(h, "device"))
    idx = torch.zeros((rows, k), dtype=torch.int32, device=dev)
    sc = torch.zeros((rows, k), dtype=torch.float32, device=dev)
    cnt = torch.zeros((rows,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    if rows:
        clib.predict_device(h, q, beam_size, post_processor, only_topk, idx.data_ptr(), sc.data_ptr(), cnt.data_ptr(), k, stream=None, sync=True)
    return rows_to_csr(idx.cpu().numpy().view(np.uint32), sc.cpu().numpy(), cnt.cpu().numpy(), model.nr_pred_cols)


def predict_text(vectorizer, models, corpus, X_emb=None, normalize_emb=True, threads=-1, **kwargs):
    """The reference's text call sites with X DEVICE-RESIDENT end to end:

    * ``Text2Text.predict`` (pecos/apps/text2text/model.py:416-422): ``X = preprocessor.predict(corpus); Y = [m.predict(X) ...]`` --
      here the texts are tokenised on the host, their term counts uploaded once, weighted on the GPU, and every model of ``models``
      (one XLinearModel or a list: the ensemble is averaged like ``CsrEnsembler.average``) searches that X in place;
    * ``XTransformer.predict`` (pecos/xmc/xtransformer/model.py:589-603): with ``X_emb`` (float32 [rows, H] CUDA tensor, the encoder's
      output) the concat model's input ``[X_feat | normalize(X_emb)]`` is assembled on the device as well.

    kwargs: beam_size, only_topk, post_processor.  Returns the predicted label matrix as scipy CSR (rows score-sorted)."""
    models = list(models) if isinstance(models, (list, tuple)) else [models]
    if isinstance(vectorizer, Preprocessor):
        vectorizer = vectorizer.vectorizer
    outs = []
    for m in models:
        q = vectorizer.predict_device(m, corpus, threads=threads)
        q2 = None
        try:
            if X_emb is not None:
                import torch
                assert X_emb.is_cuda and X_emb.dtype == torch.float32 and X_emb.shape[0] == len(corpus)
                X_emb = X_emb.contiguous()
                torch.cuda.current_stream().synchronize()
                q2 = clib.queries_concat_handle(m.model.model_chain, q, X_emb.shape[1], X_emb.data_ptr(), normalize_emb=normalize_emb)
            outs.append(_predict_handle_to_csr(m, q2 if q2 is not None else q, len(corpus), kwargs.get("beam_size"), kwargs.get("only_topk"), kwargs.get("post_processor")))
        finally:
            clib.queries_free(q)
            if q2 is not None:
                clib.queries_free(q2)
    if len(outs) == 1:
        return outs[0]
    return ensemble_average(outs)                 # CsrEnsembler.average (smat_util.py:828-842): sum, sorted_csr, divide -- rows score-sorted like the reference's


def sorted_csr(csr, only_topk=None):
    """``pecos.utils.smat_util.sorted_csr`` (smat_util.py:174-272): every row ordered by value, descending, ties by ascending column (the
    reference sorts the columns, then mergesorts -value: stable), optionally cut to the first ``only_topk``; duplicates summed like its
    ``csr_matrix((val, (row, col)))``.  One lexsort instead of the reference's Python loop over rows."""
    if not isinstance(csr, smat.csr_matrix):
        raise ValueError("the input matrix must be a csr_matrix.")
    c = smat.csr_matrix(csr, copy=True)
    c.sum_duplicates()                                       # (also sorts the columns inside every row)
    n = c.shape[0]
    counts = np.diff(c.indptr)
    rows = np.repeat(np.arange(n, dtype=np.int64), counts)
    order = np.lexsort((c.indices, -c.data, rows))           # by row, then -value, then column; NaN last like argsort
    idx, val = c.indices[order], c.data[order]
    indptr = c.indptr.astype(np.int64)
    if only_topk is not None:
        assert isinstance(only_topk, int), f"Wrong type: type(only_topk) = {type(only_topk)}"
        only_topk = max(min(1, only_topk), only_topk)        # (the reference's own expression, smat_util.py:198)
        keep = (np.arange(len(val), dtype=np.int64) - indptr[rows]) < only_topk
        idx, val = idx[keep], val[keep]
        indptr = np.concatenate([[0], np.cumsum(np.minimum(counts, only_topk))]).astype(np.int64)
    return smat.csr_matrix((val, idx.astype(np.int64), indptr), shape=c.shape)


def ensemble_average(mats):
    """``CsrEnsembler.average`` (smat_util.py:828-842): sum, ``sorted_csr``, divide by the number of matrices."""
    assert all(m.shape == mats[0].shape for m in mats)
    ret = sorted_csr(sum(mats).tocsr())
    ret.data /= len(mats)
    return ret


class Text2Text:
    """The PREDICT half of ``pecos.apps.text2text.model.Text2Text`` (model.py:136-190 load, :389-427 predict) over the device-resident
    pipeline: ``load`` the folder its ``save`` wrote (``preprocessor/``, ``xlinear_ensemble/{config.json, 0, 1, ...}``, ``output_items.json``),
    ``predict`` a list of strings -- texts -> term counts (host threads) -> X in HBM -> beam search in place, per model; ensemble average,
    threshold and the final ``sorted_csr(only_topk)`` as the reference does them.  Training, saving and ``set_output_constraint`` stay the
    reference's."""

    def __init__(self, preprocessor, xlinear_models, output_items):
        self.preprocessor = preprocessor
        self.xlinear_models = xlinear_models
        self.output_items = output_items

    @classmethod
    def load(cls, model_folder, is_predict_only=True, **kwargs):
        import json
        import os
        from .xlinear import XLinearModel
        preprocessor = Preprocessor.load(os.path.join(model_folder, "preprocessor"))
        xlinear_folder = os.path.join(model_folder, "xlinear_ensemble")
        with open(os.path.join(xlinear_folder, "config.json"), "r", encoding="utf-8") as fin:
            ensemble_config = json.loads(fin.read())
        xlinear_models = []
        for i, model_kwargs in enumerate(ensemble_config["kwargs"]):
            xlinear_models += [(XLinearModel.load(os.path.join(xlinear_folder, str(i)), is_predict_only, **kwargs), model_kwargs)]
        with open(os.path.join(model_folder, "output_items.json"), "r", encoding="utf-8") as fin:
            output_items = json.load(fin)
        if not output_items:
            raise ValueError("Could not read output items saved in json format")
        return cls(preprocessor, xlinear_models, output_items)

    @staticmethod
    def finish(Y_pred, threshold=None, only_topk=None):
        """model.py:418-427 after the per-model predictions: ensemble average, threshold, ``sorted_csr``."""
        Y = ensemble_average(Y_pred) if len(Y_pred) > 1 else Y_pred[0].tocsr()
        if threshold is not None:
            Y = Y.copy()
            Y.data[Y.data <= threshold] = 0
            Y.eliminate_zeros()
        return sorted_csr(Y, only_topk=only_topk)

    def predict(self, corpus, threshold=None, **kwargs):
        """Same arguments and result as the reference's ``Text2Text.predict`` (``threads`` applies to the tokenizer's host threads)."""
        threads = kwargs.pop("threads", -1)
        Y_pred = [predict_text(self.preprocessor, m, corpus, threads=threads, **kwargs) for m, _ in self.xlinear_models]
        return self.finish(Y_pred, threshold=threshold, only_topk=kwargs.get("only_topk", None))

    def get_output_item(self, output_id):
        return self.output_items[output_id]
