"""TEST INFRASTRUCTURE ONLY -- python front-end of the CPU oracle.

Two checkers live here; neither may be imported from ``pecos_amd/`` (the product path):

* :class:`OracleModel`  -- ctypes wrapper of ``oracle/liboracle.so`` (``xrl_oracle.c``, the plain-C
  restatement of pecos/core/xmc/inference.hpp:2446-2488 et al.).
* :class:`RefModel`     -- ctypes wrapper of ``oracle/_ref/libpecos_float32.so``, i.e. the REAL
  reference compiled from /root/reference by ``oracle/Makefile`` (its C ABI is
  pecos/core/libpecos.cpp:116-176; the struct mirrors follow pecos/core/base.py:172-354 and the
  allocator callback pecos/core/base.py:431-464).

Parity status of the restatement: pinned (see header of xrl_oracle.c and tests/test_oracle_*.py).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import scipy.sparse as smat

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libpecos_float32.so")

PP_KINDS = {"noop": 0, "sigmoid": 1, "log-sigmoid": 2}


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference is present, oracle/_ref."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "xrl_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", HERE, os.path.join(HERE, "liboracle.so")])
    if not os.path.exists(REF_SO) and os.path.exists("/root/reference/pecos/core/libpecos.cpp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def parse_post_processor(name):
    """name -> (kind, p) exactly as PostProcessor<T>::get, inference.hpp:192-240."""
    if name in PP_KINDS:
        return PP_KINDS[name], 0
    if name.startswith("log-l") and name.endswith("-hinge"):
        return 4, int(name[len("log-l"):-len("-hinge")] or 0)
    if name.startswith("l") and name.endswith("-hinge"):
        return 3, int(name[1:-len("-hinge")] or 0)
    return 0, 0  # unknown names behave as a default-constructed PostProcessor (identity)


class _Csc(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("col_ptr", C.c_void_p),
                ("row_idx", C.c_void_p), ("val", C.c_void_p)]


class _Layer(C.Structure):
    _fields_ = [("W", _Csc), ("C", _Csc), ("bias", C.c_float), ("only_topk", C.c_uint32),
                ("pp_kind", C.c_int32), ("pp_p", C.c_int32)]


def _as_csc(m, keep):
    m = smat.csc_matrix(m, dtype=np.float32)
    m.sort_indices()
    bufs = (m.indptr.astype(np.uint64), m.indices.astype(np.uint32), m.data.astype(np.float32))
    keep.append(bufs)
    return _Csc(m.shape[0], m.shape[1], bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data)


def load_model_folder(folder):
    """Read the reference's on-disk layout (xmc/base.py:807-830,1371-1395):
    <folder>/ranker/param.json + <folder>/ranker/{d}.model/{param.json,W.npz,C.npz}.
    Accepts either the XLinearModel folder or its ``ranker`` sub-folder."""
    if os.path.isdir(os.path.join(folder, "ranker")):
        folder = os.path.join(folder, "ranker")
    param = json.load(open(os.path.join(folder, "param.json")))
    layers = []
    for d in range(int(param["depth"])):
        lf = os.path.join(folder, f"{d}.model")
        p = json.load(open(os.path.join(lf, "param.json")))
        W = smat.load_npz(os.path.join(lf, "W.npz")).tocsc().astype(np.float32)
        cpath = os.path.join(lf, "C.npz")
        if os.path.exists(cpath):
            Cm = smat.load_npz(cpath).tocsc().astype(np.float32)
        else:  # inference.hpp:1580-1583
            Cm = smat.csc_matrix(np.ones((W.shape[1], 1), dtype=np.float32))
        layers.append(dict(W=W, C=Cm, bias=float(p["bias"]),
                           only_topk=int(p["pred_kwargs"]["only_topk"]),
                           post_processor=p["pred_kwargs"]["post_processor"]))
    return layers


def _dense_rows_to_csr(idx, val, cnt, n_cols):
    n = idx.shape[0]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    mask = np.arange(idx.shape[1])[None, :] < cnt[:, None]
    out = smat.csr_matrix((val[mask], idx[mask].astype(np.int64), indptr), shape=(n, n_cols))
    return out  # rows stay score-sorted (indices NOT sorted), like the reference's output


class OracleModel:
    def __init__(self, layers, weight_matrix_type="BINARY_SEARCH_CHUNKED"):
        """weight_matrix_type "HASH_CHUNKED": sparse X takes that layout's arithmetic (bias first, inference.hpp:705-735)."""
        build()
        self.lib = C.CDLL(ORACLE_SO)
        self.hash_arith = 1 if weight_matrix_type == "HASH_CHUNKED" else 0
        self.layers = layers
        self._keep = []
        arr = (_Layer * len(layers))()
        for i, L in enumerate(layers):
            # NB: C must keep its STORED order (tie-break depends on it) -> no sort_indices on C
            Cm = smat.csc_matrix(L["C"], dtype=np.float32)
            cb = (Cm.indptr.astype(np.uint64), Cm.indices.astype(np.uint32), Cm.data.astype(np.float32))
            self._keep.append(cb)
            kind, p = parse_post_processor(L["post_processor"])
            arr[i] = _Layer(_as_csc(L["W"], self._keep),
                            _Csc(Cm.shape[0], Cm.shape[1], cb[0].ctypes.data, cb[1].ctypes.data, cb[2].ctypes.data),
                            L["bias"], L["only_topk"], kind, p)
        self._arr = arr
        self.nr_labels = layers[-1]["W"].shape[1]
        self.nr_features = layers[0]["W"].shape[0] - (1 if layers[0]["bias"] > 0 else 0)

    @classmethod
    def load(cls, folder, weight_matrix_type="BINARY_SEARCH_CHUNKED"):
        return cls(load_model_folder(folder), weight_matrix_type)

    def predict_arrays(self, X, beam_size=0, only_topk=0, post_processor=None, trace=False):
        n = X.shape[0]
        depth = len(self.layers)
        k = only_topk or self.layers[-1]["only_topk"]
        ks = [(only_topk if l == depth - 1 else beam_size) or self.layers[l]["only_topk"] for l in range(depth)]
        stride = max(ks) if trace else k
        kind, p = parse_post_processor(post_processor) if post_processor else (-1, 0)
        out_idx = np.zeros((n, stride), np.uint32); out_val = np.zeros((n, stride), np.float32)
        out_cnt = np.zeros(n, np.uint32)
        tr = None
        if trace:
            tr = (np.zeros((depth, n, stride), np.uint32), np.zeros((depth, n, stride), np.float32),
                  np.zeros((depth, n), np.uint32))
        if smat.issparse(X):
            X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
            ip, ii, iv = X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32)
            xargs = (ip.ctypes.data, ii.ctypes.data, iv.ctypes.data, None, X.shape[1])
        else:
            Xd = np.ascontiguousarray(X, dtype=np.float32)
            xargs = (None, None, None, Xd.ctypes.data, Xd.shape[1])
        f = self.lib.orc_predict
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.orc_set_hash_arith(self.hash_arith if smat.issparse(X) else 0)
        rc = f(C.addressof(self._arr), depth, n, *xargs, beam_size, only_topk, kind, p,
               out_idx.ctypes.data, out_val.ctypes.data, out_cnt.ctypes.data, stride,
               tr[0].ctypes.data if tr else None, tr[1].ctypes.data if tr else None,
               tr[2].ctypes.data if tr else None)
        if rc != 0:
            raise RuntimeError("oracle failed")
        return (out_idx[:, :k], out_val[:, :k], np.minimum(out_cnt, k), tr)

    def predict(self, X, beam_size=0, only_topk=0, post_processor=None):
        idx, val, cnt, _ = self.predict_arrays(X, beam_size, only_topk, post_processor)
        return _dense_rows_to_csr(idx, val, cnt, self.nr_labels)

    def predict_on_selected_outputs(self, X, selected_outputs_csr, post_processor=None):
        S = smat.csr_matrix(selected_outputs_csr)
        sp, si = S.indptr.astype(np.uint64), S.indices.astype(np.uint32)
        kind, p = parse_post_processor(post_processor) if post_processor else (-1, 0)
        out_idx = np.zeros(S.nnz, np.uint32); out_val = np.zeros(S.nnz, np.float32)
        if smat.issparse(X):
            X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
            ip, ii, iv = X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32)
            xargs = (ip.ctypes.data, ii.ctypes.data, iv.ctypes.data, None, X.shape[1])
        else:
            Xd = np.ascontiguousarray(X, dtype=np.float32)
            xargs = (None, None, None, Xd.ctypes.data, Xd.shape[1])
        f = self.lib.orc_predict_selected
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(C.addressof(self._arr), len(self.layers), X.shape[0], *xargs, sp.ctypes.data, si.ctypes.data, kind, p,
               out_idx.ctypes.data, out_val.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle predict_on_selected_outputs failed ({rc})")
        return smat.csr_matrix((out_val, out_idx.astype(np.int64), S.indptr.astype(np.int64)), shape=S.shape)


def sparse_inner_products(X, W, rows, cols):
    """Restatement of clib.sparse_inner_products (pecos/core/base.py:1536-1589)."""
    build()
    lib = C.CDLL(ORACLE_SO)
    rows = np.ascontiguousarray(rows, np.uint32); cols = np.ascontiguousarray(cols, np.uint32)
    out = np.zeros(len(rows), np.float32)
    keep = []
    if smat.issparse(X):
        X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
        xb = (X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32))
        xa = (xb[0].ctypes.data, xb[1].ctypes.data, xb[2].ctypes.data, None)
    else:
        xb = np.ascontiguousarray(X, np.float32); xa = (None, None, None, xb.ctypes.data)
    if smat.issparse(W):
        W = smat.csc_matrix(W, dtype=np.float32); W.sort_indices()
        wb = (W.indptr.astype(np.uint64), W.indices.astype(np.uint32), W.data.astype(np.float32))
        wa = (wb[0].ctypes.data, wb[1].ctypes.data, wb[2].ctypes.data, None)
    else:
        wb = np.asfortranarray(W, np.float32); wa = (None, None, None, wb.ctypes.data)
    keep += [xb, wb]
    f = lib.orc_sparse_inner_products
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 8 + [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    f(*xa, *wa, X.shape[1], len(rows), rows.ctypes.data, cols.ctypes.data, out.ctypes.data)
    return out


# ---------------------------------------------------------------------------------------------
# The real reference through its own C ABI (oracle/_ref/libpecos_float32.so)
# ---------------------------------------------------------------------------------------------
class _CsrF32(C.Structure):  # pecos/core/utils/matrix.hpp:49-55
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("row_ptr", C.c_void_p),
                ("col_idx", C.c_void_p), ("val", C.c_void_p)]


class _DrmF32(C.Structure):  # pecos/core/utils/matrix.hpp:63-67
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("val", C.c_void_p)]


_ALLOC = C.CFUNCTYPE(None, C.c_bool, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p)
WEIGHT_TYPES = {"CSC": 0, "HASH_CHUNKED": 1, "BINARY_SEARCH_CHUNKED": 2}  # pecos/core/base.py:49


def ref_available():
    return os.path.exists(REF_SO)


def ref_compile_mmap_model(npz_ranker_folder, mmap_ranker_folder):
    """c_xlinear_compile_mmap_model of the real reference (libpecos.cpp:133-138)."""
    lib = C.CDLL(REF_SO)
    lib.c_xlinear_compile_mmap_model.argtypes = [C.c_char_p, C.c_char_p]
    lib.c_xlinear_compile_mmap_model.restype = None
    lib.c_xlinear_compile_mmap_model(npz_ranker_folder.encode(), mmap_ranker_folder.encode())


class RefModel:
    """The reference's predict-only model handle (c_xlinear_load_model_from_disk_ext)."""

    def __init__(self, folder, weight_matrix_type="BINARY_SEARCH_CHUNKED", mmap=False):
        if os.path.isdir(os.path.join(folder, "ranker")):
            folder = os.path.join(folder, "ranker")
        self.lib = C.CDLL(REF_SO)
        self.lib.c_xlinear_load_mmap_model_from_disk.restype = C.c_void_p
        self.lib.c_xlinear_load_mmap_model_from_disk.argtypes = [C.c_char_p, C.c_bool]
        self.lib.c_xlinear_load_model_from_disk_ext.restype = C.c_void_p
        self.lib.c_xlinear_load_model_from_disk_ext.argtypes = [C.c_char_p, C.c_int]
        self.lib.c_xlinear_get_int_attr.restype = C.c_uint32
        self.lib.c_xlinear_get_int_attr.argtypes = [C.c_void_p, C.c_char_p]
        self.lib.c_xlinear_destruct_model.argtypes = [C.c_void_p]
        if mmap:   # libpecos.cpp:128-131
            self.h = self.lib.c_xlinear_load_mmap_model_from_disk(folder.encode(), False)
        else:
            self.h = self.lib.c_xlinear_load_model_from_disk_ext(folder.encode(), WEIGHT_TYPES[weight_matrix_type])
        self.nr_labels = self.lib.c_xlinear_get_int_attr(self.h, b"nr_labels")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.c_xlinear_destruct_model(self.h)
            self.h = None

    def predict(self, X, beam_size=0, only_topk=0, post_processor=None, threads=8):
        # threads: the reference's OpenMP path crawls with hundreds of threads on small batches
        res = {}

        def alloc(is_col_major, rows, cols, nnz, indices_pp, indptr_pp, data_pp):
            res["indptr"] = np.zeros(rows + 1, np.uint64)
            res["indices"] = np.zeros(nnz, np.uint32)
            res["data"] = np.zeros(nnz, np.float32)
            res["shape"] = (rows, cols)
            C.cast(indices_pp, C.POINTER(C.c_uint64)).contents.value = res["indices"].ctypes.data
            C.cast(indptr_pp, C.POINTER(C.c_uint64)).contents.value = res["indptr"].ctypes.data
            C.cast(data_pp, C.POINTER(C.c_uint64)).contents.value = res["data"].ctypes.data

        cb = _ALLOC(alloc)
        pp = post_processor.encode() if post_processor else None
        if smat.issparse(X):
            X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
            bufs = (X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32))
            px = _CsrF32(X.shape[0], X.shape[1], bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data)
            fn = self.lib.c_xlinear_predict_csr_f32
        else:
            bufs = np.ascontiguousarray(X, np.float32)
            px = _DrmF32(bufs.shape[0], bufs.shape[1], bufs.ctypes.data)
            fn = self.lib.c_xlinear_predict_drm_f32
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, _ALLOC]
        fn(self.h, C.byref(px), beam_size, pp, only_topk, threads, cb)
        return smat.csr_matrix((res["data"], res["indices"].astype(np.int64), res["indptr"].astype(np.int64)),
                               shape=res["shape"])

    def predict_on_selected_outputs(self, X, selected_outputs_csr, post_processor=None, threads=8):
        """c_xlinear_predict_on_selected_outputs_{csr,drm}_f32 (libpecos.cpp:179-198); CSC layers only."""
        res = {}

        def alloc(is_col_major, rows, cols, nnz, indices_pp, indptr_pp, data_pp):
            res["indptr"] = np.zeros(rows + 1, np.uint64)
            res["indices"] = np.zeros(nnz, np.uint32)
            res["data"] = np.zeros(nnz, np.float32)
            res["shape"] = (rows, cols)
            C.cast(indices_pp, C.POINTER(C.c_uint64)).contents.value = res["indices"].ctypes.data
            C.cast(indptr_pp, C.POINTER(C.c_uint64)).contents.value = res["indptr"].ctypes.data
            C.cast(data_pp, C.POINTER(C.c_uint64)).contents.value = res["data"].ctypes.data

        cb = _ALLOC(alloc)
        pp = post_processor.encode() if post_processor else None
        S = smat.csr_matrix(selected_outputs_csr, dtype=np.float32)
        sb = (S.indptr.astype(np.uint64), S.indices.astype(np.uint32), S.data.astype(np.float32))
        ps = _CsrF32(S.shape[0], S.shape[1], sb[0].ctypes.data, sb[1].ctypes.data, sb[2].ctypes.data)
        if smat.issparse(X):
            X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
            bufs = (X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32))
            px = _CsrF32(X.shape[0], X.shape[1], bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data)
            fn = self.lib.c_xlinear_predict_on_selected_outputs_csr_f32
        else:
            bufs = np.ascontiguousarray(X, np.float32)
            px = _DrmF32(bufs.shape[0], bufs.shape[1], bufs.ctypes.data)
            fn = self.lib.c_xlinear_predict_on_selected_outputs_drm_f32
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, _ALLOC]
        fn(self.h, C.byref(px), C.byref(ps), pp, threads, cb)
        return smat.csr_matrix((res["data"], res["indices"].astype(np.int64), res["indptr"].astype(np.int64)),
                               shape=res["shape"])


# ---------------------------------------------------------------------------------------------
# The real reference's single-layer and inner-product entry points (libpecos.cpp:201-235, 337-355)
# ---------------------------------------------------------------------------------------------
class _CscF32(C.Structure):  # pecos/core/utils/matrix.hpp:56-62
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("col_ptr", C.c_void_p),
                ("row_idx", C.c_void_p), ("val", C.c_void_p)]


def _ref_x(X, keep):
    if smat.issparse(X):
        X = smat.csr_matrix(X, dtype=np.float32); X.sort_indices()
        b = (X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float32))
        keep.append(b)
        return _CsrF32(X.shape[0], X.shape[1], b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data), "csr"
    b = np.ascontiguousarray(X, np.float32)
    keep.append(b)
    return _DrmF32(b.shape[0], b.shape[1], b.ctypes.data), "drm"


def _ref_csc(M, keep, sort=True):
    M = smat.csc_matrix(M, dtype=np.float32)
    if sort:
        M.sort_indices()
    b = (M.indptr.astype(np.uint64), M.indices.astype(np.uint32), M.data.astype(np.float32))
    keep.append(b)
    return _CscF32(M.shape[0], M.shape[1], b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data)


def ref_single_layer_predict(X, csr_codes, W, Cm, post_processor, only_topk, bias, threads=8):
    """c_xlinear_single_layer_predict_{csr,drm}_f32 of the real reference: MLModel<csc_t> around W / C, i.e. the CSC
    arithmetic (bias first, dot product summed separately; inference.hpp:1018-1149)."""
    lib = C.CDLL(REF_SO)
    keep, res = [], {}

    def alloc(is_col_major, rows, cols, nnz, indices_pp, indptr_pp, data_pp):
        res["indptr"] = np.zeros(rows + 1, np.uint64); res["indices"] = np.zeros(nnz, np.uint32)
        res["data"] = np.zeros(nnz, np.float32); res["shape"] = (rows, cols)
        C.cast(indices_pp, C.POINTER(C.c_uint64)).contents.value = res["indices"].ctypes.data
        C.cast(indptr_pp, C.POINTER(C.c_uint64)).contents.value = res["indptr"].ctypes.data
        C.cast(data_pp, C.POINTER(C.c_uint64)).contents.value = res["data"].ctypes.data

    cb = _ALLOC(alloc)
    px, kind = _ref_x(X, keep)
    fn = getattr(lib, f"c_xlinear_single_layer_predict_{kind}_f32")
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_float, _ALLOC]
    pcodes = None
    if csr_codes is not None:
        S = smat.csr_matrix(csr_codes, dtype=np.float32)      # stored order kept: it is the beam order
        sb = (S.indptr.astype(np.uint64), S.indices.astype(np.uint32), S.data.astype(np.float32)); keep.append(sb)
        pcodes = _CsrF32(S.shape[0], S.shape[1], sb[0].ctypes.data, sb[1].ctypes.data, sb[2].ctypes.data)
    pw = _ref_csc(W, keep)
    pc = _ref_csc(Cm, keep, sort=False)                       # C keeps its STORED child order (tie-break)
    fn(C.byref(px), C.byref(pcodes) if pcodes is not None else None, C.byref(pw), C.byref(pc), post_processor.encode(),
       only_topk, threads, bias, cb)
    return smat.csr_matrix((res["data"], res["indices"].astype(np.int64), res["indptr"].astype(np.int64)), shape=res["shape"])


def ref_sparse_inner_products(X, W, rows, cols, threads=8):
    """c_sparse_inner_products_{csr2csc,drm2csc,csr2dcm,drm2dcm}_f32 of the real reference."""
    lib = C.CDLL(REF_SO)
    keep = []
    px, xk = _ref_x(X, keep)
    if smat.issparse(W):
        pw, wk = _ref_csc(W, keep), "csc"
    else:
        wb = np.asfortranarray(W, np.float32); keep.append(wb)
        pw, wk = _DrmF32(wb.shape[0], wb.shape[1], wb.ctypes.data), "dcm"     # ScipyDcmF32 has the same fields
    rows = np.ascontiguousarray(rows, np.uint32); cols = np.ascontiguousarray(cols, np.uint32)
    out = np.zeros(len(rows), np.float32)
    fn = getattr(lib, f"c_sparse_inner_products_{xk}2{wk}_f32")
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    fn(C.byref(px), C.byref(pw), len(rows), rows.ctypes.data, cols.ctypes.data, out.ctypes.data, threads)
    return out
