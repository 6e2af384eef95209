"""``XLinearModel`` -- the reference's user-facing XR-Linear API for the inference path, running
on MI355X through ``libxrl_amd.so``.

Surface kept from the reference (same names, kwargs and error behaviour):

* ``XLinearModel.load(model_folder, is_predict_only=True, **kwargs)``   pecos/xmc/xlinear/model.py:105-134
* ``XLinearModel.predict(X, pred_params=None, **kwargs)``               pecos/xmc/xlinear/model.py:480-550
  kwargs: beam_size, only_topk, post_processor, threads, max_pred_chunk
* ``HierarchicalMLModel.{load, predict, get_pred_params, PredParams}``  pecos/xmc/base.py:1090-1680
* ``MLModel.PredParams``                                                pecos/xmc/base.py:640-678

Training, pruning and the python (non predict-only) model chain stay in the reference: loading
with ``is_predict_only=False`` raises ``NotImplementedError``.
"""
import copy
import dataclasses as dc
import json
import os
from glob import glob
from os import path

import numpy as np
import scipy.sparse as smat

from .core import ScipyCompressedSparseAllocator, clib

VALID_POST_PROCESSORS = ["noop", "sigmoid", "log-sigmoid"] + \
    [f"l{p}-hinge" for p in range(1, 5)] + [f"log-l{p}-hinge" for p in range(1, 5)]  # xmc/base.py:456-475


class BaseParams:
    """Dataclass <-> dict helper with the reference's semantics (pecos/__init__.py:47-100)."""

    @classmethod
    def from_dict(cls, param=None):
        if param is None:
            return cls()
        if isinstance(param, cls):
            return copy.deepcopy(param)
        if isinstance(param, dict):
            names = {f.name for f in dc.fields(cls)}
            return cls(**{k: v for k, v in param.items() if k in names})
        raise ValueError(f"{type(param)} is not supported for {cls.__name__}.from_dict")

    def to_dict(self, with_meta=True):
        d = {}
        for f in dc.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, BaseParams):
                v = v.to_dict(with_meta)
            elif isinstance(v, (list, tuple)):
                v = [x.to_dict(with_meta) if isinstance(x, BaseParams) else x for x in v]
            d[f.name] = v
        return d


class MLModel:
    @dc.dataclass
    class PredParams(BaseParams):
        only_topk: int = 20
        post_processor: str = "l3-hinge"

        def override_with_kwargs(self, pred_kwargs):
            if pred_kwargs is not None:
                if not isinstance(pred_kwargs, dict):
                    raise TypeError("type(pred_kwargs) must be dict")
                for k, v in pred_kwargs.items():
                    if k in ("only_topk", "post_processor") and v is not None:
                        setattr(self, k, v)
            return self

    @classmethod
    def load_pred_params(cls, folder):
        param = json.loads(open(f"{folder}/param.json", "r", encoding="utf-8").read())
        return cls.PredParams.from_dict(param["pred_kwargs"])


class HierarchicalMLModel:
    """Predict-only hierarchical model: ``model_chain`` is the native handle (an int)."""

    @dc.dataclass
    class PredParams(BaseParams):
        model_chain: MLModel.PredParams = None  # type: ignore

        def __len__(self):
            return len(self.model_chain)

        @classmethod
        def from_dict(cls, param=None):
            if param is None:
                return cls()
            if isinstance(param, cls):
                return copy.deepcopy(param)
            chain = param.get("model_chain") if isinstance(param, dict) else None
            if isinstance(chain, (list, tuple)):
                chain = [MLModel.PredParams.from_dict(c) for c in chain]
            elif chain is not None:
                chain = MLModel.PredParams.from_dict(chain)
            return cls(model_chain=chain)

        def override_with_kwargs(self, pred_kwargs):
            # xmc/base.py:1140-1173: beam_size -> every layer but the last, only_topk -> last layer
            if pred_kwargs is None:
                return self
            if not isinstance(pred_kwargs, dict):
                raise TypeError("type(pred_kwargs) must be dict")
            beam = pred_kwargs.get("beam_size", None)
            topk = pred_kwargs.get("only_topk", None)
            pp = pred_kwargs.get("post_processor", None)
            if isinstance(self.model_chain, (list, tuple)):
                depth = len(self.model_chain)
                for d in range(depth):
                    if beam and d < depth - 1:
                        self.model_chain[d].only_topk = beam
                    if topk and d == depth - 1:
                        self.model_chain[d].only_topk = topk
                    if pp:
                        self.model_chain[d].post_processor = pp
            elif isinstance(self.model_chain, MLModel.PredParams):
                if topk:
                    self.model_chain.only_topk = topk
                if pp:
                    self.model_chain.post_processor = pp
            return self

    def __init__(self, model_chain, pred_params=None, is_predict_only=True, **kwargs):
        if not isinstance(model_chain, int):
            raise NotImplementedError("pecos_amd only wraps native (predict-only) model handles")
        self.model_chain = model_chain
        self.pred_params = self.PredParams.from_dict(pred_params)
        self.pred_params.override_with_kwargs(kwargs.get("pred_kwargs", None))
        self.is_predict_only = True

    def __del__(self):
        try:
            if getattr(self, "model_chain", None):
                clib.xlinear_destruct_model(self.model_chain)
                self.model_chain = None
        except Exception:
            pass

    @property
    def depth(self):
        return clib.xlinear_get_int_attr(self.model_chain, "depth")

    @property
    def nr_features(self):
        return clib.xlinear_get_int_attr(self.model_chain, "nr_features")

    @property
    def nr_codes(self):
        return clib.xlinear_get_int_attr(self.model_chain, "nr_codes")

    @property
    def nr_labels(self):
        return clib.xlinear_get_int_attr(self.model_chain, "nr_labels")

    @property
    def nr_pred_cols(self):
        """Column count of predict()'s CSR (== C.rows of the last layer; differs from nr_labels only
        for pruned trees, inference.hpp:1776-1784)."""
        return clib.xlinear_get_int_attr(self.model_chain, "nr_pred_cols")

    def get_pred_params(self):
        return copy.deepcopy(self.pred_params)

    @classmethod
    def load(cls, model_folder, is_predict_only=True, **kwargs):
        param = json.loads(open(f"{model_folder}/param.json", "r", encoding="utf-8").read())
        assert param["model"] == cls.__name__
        depth = int(param.get("depth", len(glob("{}/*.model".format(model_folder)))))
        if not is_predict_only:
            raise NotImplementedError(
                "pecos_amd accelerates the predict-only path; load with is_predict_only=True "
                "(training / pruning stay in the reference's CPU code)")
        if bool(param.get("is_mmap", False)):
            model = clib.xlinear_load_mmap(model_folder, **kwargs)
        else:
            model = clib.xlinear_load_predict_only(model_folder, **kwargs)
        pred_params = cls.PredParams(
            model_chain=[MLModel.load_pred_params(f"{model_folder}/{d}.model") for d in range(depth)])
        return cls(model, pred_params=pred_params, is_predict_only=True)

    def _resolve_overrides(self, pred_params, kwargs):
        """xmc/base.py:1609-1654: merge kwargs, then only UNIFORM overrides are expressible natively."""
        if pred_params is None:
            pred_params = self.get_pred_params()
        elif isinstance(pred_params, self.PredParams):
            pred_params = self.PredParams.from_dict(pred_params)
            if isinstance(pred_params.model_chain, MLModel.PredParams):
                pred_params.model_chain = [copy.deepcopy(pred_params.model_chain) for _ in range(self.depth)]
            elif len(pred_params.model_chain) != self.depth:
                raise ValueError(f"len(params.model_chain)={len(pred_params.model_chain)} != {self.depth}")
        else:
            raise ValueError("unknown type(pred_params)!!")
        pred_params.override_with_kwargs(kwargs)
        old_chain = self.get_pred_params().model_chain
        new_chain = pred_params.model_chain
        if all(o.post_processor == n.post_processor for o, n in zip(old_chain, new_chain)):
            pp = None
        elif all(new_chain[0].post_processor == n.post_processor for n in new_chain):
            pp = new_chain[0].post_processor
        else:
            raise NotImplementedError("when is_predict_only=True, post_processor is not supported for overriddng")
        if all(o.only_topk == n.only_topk for o, n in zip(old_chain[:-1], new_chain[:-1])):
            beam = None
        elif all(new_chain[0].only_topk == n.only_topk for n in new_chain[:-1]):
            beam = new_chain[0].only_topk
        else:
            raise NotImplementedError("when is_predict_only=True, beam_size is not supported for overriding")
        return beam, pp, new_chain[-1].only_topk

    def predict(self, X, csr_codes=None, pred_params=None, **kwargs):
        assert X.dtype == np.float32
        assert isinstance(X, smat.csr_matrix) or (isinstance(X, np.ndarray) and X.flags["C_CONTIGUOUS"])
        assert X.shape[1] == self.nr_features
        if csr_codes is not None:
            raise NotImplementedError("is_predict_only=True did not support csr_codes being not None")
        beam, pp, topk = self._resolve_overrides(pred_params, kwargs)
        pred_alloc = ScipyCompressedSparseAllocator()
        clib.xlinear_predict(self.model_chain, X, beam, pp, topk, kwargs.get("threads", -1), pred_alloc)
        return pred_alloc.get()


def _selected(self, X, selected_outputs_csr, pred_params=None, **kwargs):
    """HierarchicalMLModel.predict_on_selected_outputs, pecos/xmc/base.py:1682-1790."""
    if X.dtype != np.float32:
        raise ValueError("X.dtype = {} is not supported".format(X.dtype))
    if not isinstance(X, smat.csr_matrix) and not (isinstance(X, np.ndarray) and X.flags["C_CONTIGUOUS"]):
        raise ValueError("type(X) = {} is not supported".format(type(X)))
    if X.shape[1] != self.nr_features:
        raise ValueError("Feature dimension of query matrix does not match weight matrix")
    if not isinstance(selected_outputs_csr, smat.csr_matrix):
        raise ValueError("type(selected_outputs_csr) = {} is not supported".format(type(selected_outputs_csr)))
    if selected_outputs_csr.shape[1] != self.nr_pred_cols:
        raise ValueError("Label dimension of selected output matrix does not match")
    if X.shape[0] != selected_outputs_csr.shape[0]:
        raise ValueError("Instance dimension of query and selected output matrix do not match")
    _, pp, _ = self._resolve_overrides(pred_params, {k: v for k, v in kwargs.items() if k == "post_processor"})
    pred_alloc = ScipyCompressedSparseAllocator()
    clib.xlinear_predict_on_selected_outputs(self.model_chain, X, selected_outputs_csr, pp, kwargs.get("threads", -1), pred_alloc)
    return pred_alloc.get()


HierarchicalMLModel.predict_on_selected_outputs = _selected


def vstack_csr(matrices):
    """Row-stack CSR blocks keeping the (score-sorted) order inside rows (smat_util.py:343-390)."""
    indptr = [np.zeros(1, dtype=np.int64)]
    base = 0
    for m in matrices:
        indptr.append(m.indptr[1:].astype(np.int64) + base)
        base += int(m.indptr[-1])
    return smat.csr_matrix((np.concatenate([m.data for m in matrices]),
                            np.concatenate([m.indices for m in matrices]).astype(np.int64),
                            np.concatenate(indptr)),
                           shape=(sum(m.shape[0] for m in matrices), matrices[0].shape[1]))


class XLinearModel:
    """Drop-in for ``pecos.xmc.xlinear.XLinearModel`` on the predict path."""

    @dc.dataclass
    class PredParams(BaseParams):
        hlm_args: HierarchicalMLModel.PredParams = None  # type: ignore

        def override_with_kwargs(self, pred_kwargs):
            self.hlm_args.override_with_kwargs(pred_kwargs)
            return self

    def __init__(self, model=None):
        self.model = model

    @property
    def depth(self):
        return self.model.depth

    @property
    def nr_features(self):
        return self.model.nr_features

    @property
    def nr_labels(self):
        return self.model.nr_labels

    @property
    def nr_codes(self):
        return self.model.nr_codes

    @property
    def nr_pred_cols(self):
        return self.model.nr_pred_cols

    @property
    def is_predict_only(self):
        return True

    @classmethod
    def load(cls, model_folder, is_predict_only=True, **kwargs):
        """kwargs: weight_matrix_type in {"BINARY_SEARCH_CHUNKED", "HASH_CHUNKED", "CSC"} (pecos/core/base.py:49).
        "CSC" runs the reference's CSC arithmetic (bias first, dot product summed separately, inference.hpp:1018-1149) and is
        bit-identical to the reference loaded with the same type.  The two chunked types share one device layout; with SPARSE
        queries "HASH_CHUNKED" takes that layout's arithmetic (bias first, then the query's features ascending,
        inference.hpp:705-735) and is bit-identical to the reference loaded as HASH_CHUNKED too; with DENSE queries the reference's
        hash layout walks its hash map in the map's own order (:737-768), which no other implementation can reproduce -- dense
        queries get the BINARY_SEARCH_CHUNKED arithmetic there (same labels, scores within ~3e-6 relative of the hash layout's)."""
        model = HierarchicalMLModel.load(path.join(model_folder, "ranker"), is_predict_only, **kwargs)
        return cls(model)

    @classmethod
    def compile_mmap_model(cls, npz_folder, mmap_folder):
        """npz model folder -> memory-mapped model folder (pecos/xmc/xlinear/model.py:136-152)."""
        import shutil
        os.makedirs(mmap_folder, exist_ok=True)
        shutil.copy(path.join(npz_folder, "param.json"), path.join(mmap_folder, "param.json"))
        clib.xlinear_compile_mmap_model(path.join(npz_folder, "ranker"), path.join(mmap_folder, "ranker"))

    # ---- query / label matrix files as the reference's predict CLI reads and writes them (pecos/xmc/xlinear/model.py:424-467,
    #      pecos/utils/smat_util.py:60-150): .npy = dense, scipy .npz = sparse
    @staticmethod
    def _load_matrix(src):
        if not isinstance(src, str):
            raise ValueError("src for load_matrix must be a str")
        mat = np.load(src)
        if isinstance(mat, np.ndarray):
            return mat
        fmt = mat["format"].item()
        fmt = fmt if isinstance(fmt, str) else fmt.decode("ascii")
        if fmt in ("csc", "csr", "bsr"):
            out = getattr(smat, fmt + "_matrix")((mat["data"], mat["indices"], mat["indptr"]), shape=mat["shape"])
            out.sort_indices()
            return out
        if fmt == "coo":
            return smat.coo_matrix((mat["data"], (mat["row"], mat["col"])), shape=mat["shape"])
        if fmt == "dia":
            return smat.dia_matrix((mat["data"], mat["offsets"]), shape=mat["shape"])
        raise ValueError("Unknown matrix format {}".format(fmt))

    @staticmethod
    def save_feature_matrix(tgt, feat_mat):
        """dense -> .npy stream, sparse -> scipy .npz (uncompressed, like the reference's save_matrix)."""
        if isinstance(feat_mat, np.ndarray):
            np.save(tgt, feat_mat, allow_pickle=False)
        elif smat.issparse(feat_mat):
            smat.save_npz(tgt, feat_mat, compressed=False)
        else:
            raise NotImplementedError("Save not implemented for matrix type {}".format(type(feat_mat)))

    @staticmethod
    def load_feature_matrix(src):
        """csr_matrix with sorted indices, or a C-contiguous ndarray (what predict() accepts)."""
        feat_mat = XLinearModel._load_matrix(src)
        if isinstance(feat_mat, np.ndarray):
            return np.ascontiguousarray(feat_mat)
        feat_mat = feat_mat.tocsr()
        feat_mat.sort_indices()
        return feat_mat

    @staticmethod
    def load_label_matrix(src, for_training=False):
        assert isinstance(src, str), "src for load_label_matrix must be a str"
        lab = XLinearModel._load_matrix(src)
        lab = smat.csc_matrix(lab) if for_training else smat.csr_matrix(lab)
        return lab.astype(np.float32)

    def get_pred_params(self):
        return self.PredParams(hlm_args=self.model.get_pred_params())

    def predict(self, X, pred_params=None, selected_outputs_csr=None, **kwargs):
        if (pred_params is not None) and (not isinstance(pred_params, self.PredParams)):
            raise TypeError("type(pred_kwargs) is not supported")
        max_pred_chunk = kwargs.get("max_pred_chunk", 10**7)
        if not (max_pred_chunk is None or isinstance(max_pred_chunk, int)):
            raise TypeError("type(max_pred_chunk) is not supported.")
        if max_pred_chunk is None or max_pred_chunk >= X.shape[0]:
            kw = {k: v for k, v in kwargs.items() if k != "max_pred_chunk"}
            hp = None if pred_params is None else pred_params.hlm_args
            if selected_outputs_csr is None:
                return self.model.predict(X, pred_params=hp, **kw)
            return self.model.predict_on_selected_outputs(X, selected_outputs_csr, pred_params=hp, **kw)
        new_kwargs = kwargs.copy()
        new_kwargs.pop("max_pred_chunk", None)
        Ys = [self.predict(X[i: i + max_pred_chunk, :], pred_params=pred_params,
                           selected_outputs_csr=(selected_outputs_csr[i: i + max_pred_chunk, :]
                                                 if selected_outputs_csr is not None else None), **new_kwargs)
              for i in range(0, X.shape[0], max_pred_chunk)]
        return vstack_csr(Ys)
