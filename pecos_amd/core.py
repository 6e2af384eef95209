"""ctypes binding of ``libxrl_amd.so`` -- the MI355X counterpart of ``pecos.core.clib`` for the
XR-Linear inference path.

Mirrors, for that path only, the reference's binding layer (pecos/core/base.py):

* struct views ``ScipyCsrF32 / ScipyCscF32 / ScipyDrmF32 / ScipyDcmF32``   base.py:172-354
* ``ScipyCompressedSparseAllocator`` (python-side result allocator)        base.py:357-478
* ``corelib`` methods ``xlinear_load_predict_only``, ``xlinear_destruct_model``,
  ``xlinear_get_int_attr``, ``xlinear_get_layer_type``, ``xlinear_predict``,
  ``xlinear_single_layer_predict``, ``sparse_inner_products``             base.py:978-1405, 1536-1589

with the same names, argument order and meaning, so that code written against
``pecos.core.clib`` reads the same against :data:`clib` here.  There is NO CPU fallback: every
compute call needs a visible HIP device and raises ``RuntimeError`` otherwise.
"""
import ctypes
import sys
import os
from ctypes import (CFUNCTYPE, POINTER, byref, c_bool, c_char_p, c_double, c_float, c_int, c_int64,
                    c_uint32, c_uint64, c_void_p, cast)

import numpy as np
import scipy.sparse as smat

XLINEAR_INFERENCE_MODEL_TYPES = {"CSC": 0, "HASH_CHUNKED": 1, "BINARY_SEARCH_CHUNKED": 2}  # base.py:49

# PECOS_XRL_AMD_SO selects another build of the same library (kernel-tuning variants)
_LIB_PATH = os.environ.get("PECOS_XRL_AMD_SO") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libxrl_amd.so")


class _MatView(ctypes.Structure):
    """Common behaviour of the zero-copy matrix views (the buffers are kept alive in py_buf)."""

    @classmethod
    def init_from(cls, A):
        if A is None:
            return None
        if isinstance(A, cls):
            return A
        return cls(A)

    @property
    def dtype(self):
        return self.buf.dtype

    @property
    def shape(self):
        return self.buf.shape


class ScipyCscF32(_MatView):
    _fields_ = [("rows", c_uint32), ("cols", c_uint32), ("col_ptr", POINTER(c_uint64)),
                ("row_idx", POINTER(c_uint32)), ("val", POINTER(c_float))]

    def __init__(self, A):
        assert isinstance(A, smat.csc_matrix)
        assert A.dtype == np.float32
        self.py_buf = {"col_ptr": A.indptr.astype(np.uint64, copy=False),
                       "row_idx": A.indices.astype(np.uint32, copy=False),
                       "val": A.data.astype(np.float32, copy=False)}
        self.rows, self.cols = A.shape
        for name, typ in self._fields_[2:]:
            setattr(self, name, self.py_buf[name].ctypes.data_as(typ))
        self.buf = A


class ScipyCsrF32(_MatView):
    _fields_ = [("rows", c_uint32), ("cols", c_uint32), ("row_ptr", POINTER(c_uint64)),
                ("col_idx", POINTER(c_uint32)), ("val", POINTER(c_float))]

    def __init__(self, A):
        assert isinstance(A, smat.csr_matrix)
        assert A.dtype == np.float32
        self.py_buf = {"row_ptr": A.indptr.astype(np.uint64, copy=False),
                       "col_idx": A.indices.astype(np.uint32, copy=False),
                       "val": A.data.astype(np.float32, copy=False)}
        self.rows, self.cols = A.shape
        for name, typ in self._fields_[2:]:
            setattr(self, name, self.py_buf[name].ctypes.data_as(typ))
        self.buf = A


class ScipyDrmF32(_MatView):
    _fields_ = [("rows", c_uint32), ("cols", c_uint32), ("val", POINTER(c_float))]

    def __init__(self, A):
        assert isinstance(A, np.ndarray)
        assert A.dtype == np.float32
        assert A.flags["C_CONTIGUOUS"] is True
        self.py_buf = {"val": A}
        self.rows, self.cols = A.shape
        self.val = A.ctypes.data_as(POINTER(c_float))
        self.buf = A


class ScipyDcmF32(_MatView):
    _fields_ = [("rows", c_uint32), ("cols", c_uint32), ("val", POINTER(c_float))]

    def __init__(self, A):
        assert isinstance(A, np.ndarray)
        assert A.dtype == np.float32
        assert A.flags["F_CONTIGUOUS"] is True
        self.py_buf = {"val": A}
        self.rows, self.cols = A.shape
        self.val = A.ctypes.data_as(POINTER(c_float))
        self.buf = A


class ScipyCompressedSparseAllocator(object):
    """Python-side allocator handed to the native predict call (base.py:357-478): the callee passes
    the addresses of its three pointers, we allocate numpy arrays and write their addresses back."""

    CFUNCTYPE = CFUNCTYPE(None, c_bool, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p)

    def __init__(self, rows=0, cols=0, dtype=np.float32):
        assert dtype == np.float32
        self.rows, self.cols, self.dtype = rows, cols, dtype
        self.indices = self.indptr = self.data = None
        self.is_col_major = None

    def __call__(self, is_col_major, rows, cols, nnz, indices_ptr, indptr_ptr, data_ptr):
        self.rows, self.cols, self.is_col_major = rows, cols, is_col_major
        self.indptr = np.zeros((cols if is_col_major else rows) + 1, dtype=np.uint64)
        self.indices = np.zeros(nnz, dtype=np.uint32)
        self.data = np.zeros(nnz, dtype=self.dtype)
        for dst, arr in ((indices_ptr, self.indices), (indptr_ptr, self.indptr), (data_ptr, self.data)):
            cast(dst, POINTER(c_uint64)).contents.value = arr.ctypes.data_as(c_void_p).value or 0

    def get(self):
        if self.indptr is None:
            raise RuntimeError("native call did not produce a result")
        mat = smat.csc_matrix if self.is_col_major else smat.csr_matrix
        # int64 index arrays (scipy would otherwise up/down-cast unsigned ones); order is preserved
        return mat((self.data, self.indices.astype(np.int64), self.indptr.astype(np.int64)),
                   shape=(self.rows, self.cols))

    @property
    def cfunc(self):
        return self.CFUNCTYPE(self)


class ProfileRec(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("layer", c_uint32), ("launches", c_uint32),
                ("ms", c_double), ("reserved", c_double)]


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If this library pulls in the
    system copy first and torch is imported LATER, torch finds "No HIP GPUs"; importing torch first avoided that,
    but nothing enforced the order.  When a torch installation is present and not imported yet, its copy of the runtime is loaded here
    (without importing torch), so that libxrl_amd.so binds to the runtime torch will use -- in either import order."""
    if "torch" in sys.modules:
        return
    try:
        import glob
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for d in (spec.submodule_search_locations or []) if spec else []:
            for so in sorted(glob.glob(os.path.join(d, "lib", "libamdhip64.so*"))):
                ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
                return
    except Exception:
        pass                                     # fall back to the system runtime


class corelib(object):
    """The C-ABI library, loaded lazily (so that importing the package works on a CPU-only box)."""

    def __init__(self, so_path=_LIB_PATH):
        self.so_path = so_path
        self._lib = None

    # ------------------------------------------------------------------ loading / errors
    @property
    def clib_float32(self):
        if self._lib is None:
            if not os.path.exists(self.so_path):
                raise RuntimeError(
                    f"{self.so_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  pecos_amd has no CPU fallback.")
            _preload_torch_hip_runtime()
            self._lib = ctypes.CDLL(self.so_path)
            self._link(self._lib)
        return self._lib

    @staticmethod
    def _link(lib):
        alloc_t = ScipyCompressedSparseAllocator.CFUNCTYPE
        sigs = {
            "xrl_last_error": (c_char_p, []),
            "xrl_clear_error": (None, []),
            "xrl_version": (c_char_p, []),
            "xrl_device_count": (c_int, []),
            "xrl_set_device": (c_int, [c_int]),
            "c_xlinear_load_model_from_disk": (c_void_p, [c_char_p]),
            "c_xlinear_load_model_from_disk_ext": (c_void_p, [c_char_p, c_int]),
            "c_xlinear_load_mmap_model_from_disk": (c_void_p, [c_char_p, c_bool]),
            "c_xlinear_compile_mmap_model": (None, [c_char_p, c_char_p]),
            "c_xlinear_destruct_model": (None, [c_void_p]),
            "c_xlinear_get_int_attr": (c_uint32, [c_void_p, c_char_p]),
            "c_xlinear_get_layer_type": (c_int, [c_void_p, c_int]),
            "c_xlinear_predict_csr_f32": (None, [c_void_p, POINTER(ScipyCsrF32), c_uint32, c_char_p, c_uint32, c_int, alloc_t]),
            "c_xlinear_predict_drm_f32": (None, [c_void_p, POINTER(ScipyDrmF32), c_uint32, c_char_p, c_uint32, c_int, alloc_t]),
            "c_xlinear_predict_on_selected_outputs_csr_f32": (None, [c_void_p, POINTER(ScipyCsrF32), POINTER(ScipyCsrF32), c_char_p, c_int, alloc_t]),
            "c_xlinear_predict_on_selected_outputs_drm_f32": (None, [c_void_p, POINTER(ScipyDrmF32), POINTER(ScipyCsrF32), c_char_p, c_int, alloc_t]),
            "c_xlinear_single_layer_predict_csr_f32": (None, [POINTER(ScipyCsrF32), POINTER(ScipyCsrF32), POINTER(ScipyCscF32), POINTER(ScipyCscF32), c_char_p, c_uint32, c_int, c_float, alloc_t]),
            "c_xlinear_single_layer_predict_drm_f32": (None, [POINTER(ScipyDrmF32), POINTER(ScipyCsrF32), POINTER(ScipyCscF32), POINTER(ScipyCscF32), c_char_p, c_uint32, c_int, c_float, alloc_t]),
            "c_xlinear_single_layer_predict_on_selected_outputs_csr_f32": (None, [POINTER(ScipyCsrF32), POINTER(ScipyCsrF32), POINTER(ScipyCsrF32), POINTER(ScipyCscF32), POINTER(ScipyCscF32), c_char_p, c_int, c_float, alloc_t]),
            "c_xlinear_single_layer_predict_on_selected_outputs_drm_f32": (None, [POINTER(ScipyDrmF32), POINTER(ScipyCsrF32), POINTER(ScipyCsrF32), POINTER(ScipyCscF32), POINTER(ScipyCscF32), c_char_p, c_int, c_float, alloc_t]),
            "c_sparse_inner_products_csr2csc_f32": (None, [POINTER(ScipyCsrF32), POINTER(ScipyCscF32), c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_float), c_int]),
            "c_sparse_inner_products_drm2csc_f32": (None, [POINTER(ScipyDrmF32), POINTER(ScipyCscF32), c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_float), c_int]),
            "c_sparse_inner_products_csr2dcm_f32": (None, [POINTER(ScipyCsrF32), POINTER(ScipyDcmF32), c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_float), c_int]),
            "c_sparse_inner_products_drm2dcm_f32": (None, [POINTER(ScipyDrmF32), POINTER(ScipyDcmF32), c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_float), c_int]),
            "xrl_inspect_model": (c_int, [c_char_p, POINTER(c_uint64), c_uint32]),
            "xrl_model_create": (c_void_p, [c_uint32, c_void_p, c_void_p, POINTER(c_float), POINTER(c_uint32), POINTER(c_char_p)]),
            "xrl_queries_upload_csr": (c_void_p, [c_void_p, POINTER(ScipyCsrF32)]),
            "xrl_queries_upload_drm": (c_void_p, [c_void_p, POINTER(ScipyDrmF32)]),
            "xrl_queries_from_device_csr": (c_void_p, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint64]),
            "xrl_queries_from_device_drm": (c_void_p, [c_void_p, c_uint32, c_uint32, c_void_p]),
            "xrl_queries_concat_device": (c_void_p, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_void_p]),
            "xrl_queries_tfidf_device": (c_void_p, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
            "xrl_queries_concat_device_ex": (c_void_p, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_int, c_void_p]),
            "xrl_queries_free": (None, [c_void_p]),
            "c_tfidf_load": (c_void_p, [c_char_p]),
            "c_tfidf_destruct": (None, [c_void_p]),
            "c_tfidf_predict": (None, [c_void_p, c_void_p, POINTER(c_uint64), c_uint64, c_int, ScipyCompressedSparseAllocator.CFUNCTYPE]),
            "c_tfidf_predict_from_file": (None, [c_void_p, c_void_p, c_uint64, c_uint64, c_int, ScipyCompressedSparseAllocator.CFUNCTYPE]),
            "xrl_tfidf_counts": (None, [c_void_p, c_void_p, POINTER(c_uint64), c_uint64, c_int, ScipyCompressedSparseAllocator.CFUNCTYPE]),
            "xrl_tfidf_nr_features": (c_uint32, [c_void_p]),
            "xrl_tfidf_predict_device": (c_void_p, [c_void_p, c_void_p, c_void_p, POINTER(c_uint64), c_uint64, c_int]),
            "xrl_queries_concat_handle": (c_void_p, [c_void_p, c_void_p, c_uint32, c_void_p, c_int, c_void_p]),
            "xrl_predict_device": (c_int, [c_void_p, c_void_p, c_uint32, c_char_p, c_uint32, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_int]),
            "xrl_predict_device_rows": (c_int, [c_void_p, c_void_p, c_uint32, c_char_p, c_uint32, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_int, c_uint32, c_uint32]),
            "xrl_predict_stats": (c_int, [c_void_p, c_void_p, c_uint32, c_char_p, c_uint32, POINTER(c_double), c_uint32]),
            "xrl_effective_topk": (c_uint32, [c_void_p, c_uint32]),
            "xrl_profile_enable": (None, [c_void_p, c_int]),
            "xrl_profile_reset": (None, [c_void_p]),
            "xrl_profile_get": (c_uint32, [c_void_p, POINTER(ProfileRec), c_uint32]),
            "xrl_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
            "xrl_model_device_bytes": (c_uint64, [c_void_p]),
            "xrl_debug_k1_phases": (None, [POINTER(c_uint64), c_int]),
            "xrl_layer_info": (c_uint32, [c_void_p, c_uint32, POINTER(c_uint64), c_uint32]),
            "xrl_single_layer_cache_clear": (None, []),
            "xrl_single_layer_cache_stats": (None, [POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args

    EXPORTED_SYMBOLS = None  # filled below from include/xrl_abi.h by tests

    def _check(self):
        err = self.clib_float32.xrl_last_error()
        if err:
            self.clib_float32.xrl_clear_error()
            raise RuntimeError(err.decode("utf-8", "replace"))

    # ------------------------------------------------------------------ device management (additive)
    def device_count(self):
        return int(self.clib_float32.xrl_device_count())

    def set_device(self, device):
        self.clib_float32.xrl_set_device(int(device))
        self._check()

    def inspect_model(self, folder):
        """Host-only parse of a model folder (no GPU needed): list of per-layer shape dicts."""
        buf = (c_uint64 * (6 * 64))()
        depth = self.clib_float32.xrl_inspect_model(folder.encode("utf-8"), buf, 6 * 64)
        self._check()
        keys = ("w_rows", "w_cols", "w_nnz", "c_rows", "c_cols", "c_nnz")
        return [dict(zip(keys, [int(buf[6 * d + i]) for i in range(6)])) for d in range(depth)]

    # ------------------------------------------------------------------ xlinear (base.py:978-1405)
    def xlinear_load_predict_only(self, folder, weight_matrix_type="BINARY_SEARCH_CHUNKED"):
        """Load a model folder (``<model>/ranker``) onto the GPU; returns the native handle."""
        type_id = XLINEAR_INFERENCE_MODEL_TYPES[weight_matrix_type]
        cmodel = self.clib_float32.c_xlinear_load_model_from_disk_ext(c_char_p(folder.encode("utf-8")), c_int(int(type_id)))
        self._check()
        return cmodel

    def xlinear_load_mmap(self, folder, lazy_load=False):
        """Load a memory-mapped model folder (base.py:990-1009)."""
        cmodel = self.clib_float32.c_xlinear_load_mmap_model_from_disk(c_char_p(folder.encode("utf-8")), c_bool(lazy_load))
        self._check()
        return cmodel

    def xlinear_compile_mmap_model(self, npz_folder, mmap_folder):
        """npz model folder -> mmap model folder (base.py:978-988); host-only."""
        self.clib_float32.c_xlinear_compile_mmap_model(c_char_p(npz_folder.encode("utf-8")), c_char_p(mmap_folder.encode("utf-8")))
        self._check()

    def xlinear_destruct_model(self, c_model):
        if self._lib is not None and c_model:
            self._lib.c_xlinear_destruct_model(c_void_p(c_model))

    def xlinear_get_int_attr(self, c_model, attr):
        assert attr in {"depth", "nr_features", "nr_labels", "nr_codes", "nr_pred_cols", "nr_bucket_layers", "nr_bitmap64_layers", "nr_dense_layers", "device", "nr_devices"}, f"attr {attr} not implemented"
        v = self.clib_float32.c_xlinear_get_int_attr(c_void_p(c_model), c_char_p(attr.encode("utf-8")))
        self._check()
        return v

    def xlinear_get_layer_type(self, c_model, layer_depth):
        v = self.clib_float32.c_xlinear_get_layer_type(c_void_p(c_model), layer_depth)
        self._check()
        return v

    def xlinear_predict(self, c_model, X, overriden_beam_size, overriden_post_processor_str, overriden_only_topk,
                        threads, pred_alloc):
        """Full beam-search prediction (base.py:1041-1095).  ``None``/0 overrides = model defaults."""
        clib = self.clib_float32
        if isinstance(X, smat.csr_matrix):
            if not X.has_sorted_indices:
                raise ValueError("Query matrix does not have sorted indices!")
            X = ScipyCsrF32.init_from(X)
        elif isinstance(X, np.ndarray):
            X = ScipyDrmF32.init_from(X)
        if isinstance(X, ScipyCsrF32):
            c_predict = clib.c_xlinear_predict_csr_f32
        elif isinstance(X, ScipyDrmF32):
            c_predict = clib.c_xlinear_predict_drm_f32
        else:
            raise NotImplementedError("type(X) = {} not implemented".format(type(X)))
        cb = pred_alloc.cfunc
        c_predict(c_void_p(c_model), byref(X), overriden_beam_size if overriden_beam_size else 0,
                  overriden_post_processor_str.encode("utf-8") if overriden_post_processor_str else None,
                  overriden_only_topk if overriden_only_topk else 0, threads, cb)
        self._check()

    def xlinear_predict_on_selected_outputs(self, c_model, X, selected_outputs_csr, overriden_post_processor_str, threads,
                                            pred_alloc):
        """Scores for a given (query, label) pattern (base.py:1097-1141)."""
        clib = self.clib_float32
        if isinstance(X, smat.csr_matrix):
            if not X.has_sorted_indices:
                raise ValueError("Query matrix does not have sorted indices!")
            X = ScipyCsrF32.init_from(X)
        elif isinstance(X, np.ndarray):
            X = ScipyDrmF32.init_from(X)
        if not isinstance(selected_outputs_csr, smat.csr_matrix):
            raise ValueError("type(selected_outputs_csr) = {} not implemented".format(type(selected_outputs_csr)))
        S = ScipyCsrF32.init_from(selected_outputs_csr.astype(np.float32))
        if isinstance(X, ScipyCsrF32):
            c_predict = clib.c_xlinear_predict_on_selected_outputs_csr_f32
        elif isinstance(X, ScipyDrmF32):
            c_predict = clib.c_xlinear_predict_on_selected_outputs_drm_f32
        else:
            raise NotImplementedError("type(X) = {} not implemented".format(type(X)))
        cb = pred_alloc.cfunc
        c_predict(c_void_p(c_model), byref(X), byref(S),
                  overriden_post_processor_str.encode("utf-8") if overriden_post_processor_str else None, threads, cb)
        self._check()

    def xlinear_single_layer_predict(self, X, csr_codes, W, C, post_processor_str, only_topk, num_threads, bias, pred_alloc):
        """One layer from python-owned W / C (base.py:1143-1207)."""
        clib = self.clib_float32
        post_processor_str = post_processor_str.encode("utf-8")
        W = ScipyCscF32.init_from(W)
        C = ScipyCscF32.init_from(C)
        if isinstance(X, smat.csr_matrix):
            if not X.has_sorted_indices:
                raise ValueError("Query matrix does not have sorted indices!")
            X = ScipyCsrF32.init_from(X)
        elif isinstance(X, np.ndarray):
            X = ScipyDrmF32.init_from(X)
        if isinstance(X, ScipyCsrF32):
            c_predict = clib.c_xlinear_single_layer_predict_csr_f32
        elif isinstance(X, ScipyDrmF32):
            c_predict = clib.c_xlinear_single_layer_predict_drm_f32
        else:
            raise NotImplementedError("type(X) = {} not implemented".format(type(X)))
        codes = ScipyCsrF32.init_from(csr_codes)
        cb = pred_alloc.cfunc
        c_predict(byref(X), byref(codes) if codes is not None else None, byref(W),
                  byref(C) if C is not None else None, post_processor_str, only_topk, num_threads, bias, cb)
        self._check()

    def xlinear_single_layer_predict_on_selected_outputs(self, X, selected_outputs_csr, csr_codes, W, C, post_processor_str,
                                                         num_threads, bias, pred_alloc):
        """One layer from python-owned W / C on a given output pattern (base.py:1227-1303)."""
        clib = self.clib_float32
        post_processor_str = post_processor_str.encode("utf-8")
        W = ScipyCscF32.init_from(W)
        S = ScipyCsrF32.init_from(selected_outputs_csr.astype(np.float32))
        if isinstance(X, smat.csr_matrix):
            if not X.has_sorted_indices:
                raise ValueError("Query matrix does not have sorted indices!")
            X = ScipyCsrF32.init_from(X)
        elif isinstance(X, np.ndarray):
            X = ScipyDrmF32.init_from(X)
        if isinstance(X, ScipyCsrF32):
            fn = clib.c_xlinear_single_layer_predict_on_selected_outputs_csr_f32
        elif isinstance(X, ScipyDrmF32):
            fn = clib.c_xlinear_single_layer_predict_on_selected_outputs_drm_f32
        else:
            raise NotImplementedError("type(X) = {} not implemented".format(type(X)))
        if C is None:
            C = smat.csc_matrix(np.ones((W.shape[1], 1), dtype=np.float32))
        C = ScipyCscF32.init_from(C)
        if csr_codes is not None:
            if csr_codes.shape[0] != X.shape[0]:
                raise ValueError("Instance dimension of query and csr_codes matrix do not match")
            if csr_codes.shape[1] != C.shape[1]:
                raise ValueError("Label dimension of csr_codes and C matrix do not match")
            csr_codes = ScipyCsrF32.init_from(csr_codes)
        cb = pred_alloc.cfunc
        fn(byref(X), byref(S), byref(csr_codes) if csr_codes is not None else None, byref(W), byref(C), post_processor_str,
           num_threads, bias, cb)
        self._check()

    def sparse_inner_products(self, X, W, X_row_idx, W_col_idx, pred_values=None, threads=-1):
        """val[i] = <X[X_row_idx[i], :], W[:, W_col_idx[i]]>  (base.py:1536-1589)."""
        clib = self.clib_float32
        nnz = len(X_row_idx)
        assert nnz == len(W_col_idx)
        assert X.shape[1] == W.shape[0]
        if isinstance(X, smat.csr_matrix) and isinstance(W, smat.csc_matrix):
            pX, pW, fn = ScipyCsrF32.init_from(X), ScipyCscF32.init_from(W), clib.c_sparse_inner_products_csr2csc_f32
        elif isinstance(X, np.ndarray) and isinstance(W, smat.csc_matrix):
            pX, pW, fn = ScipyDrmF32.init_from(X), ScipyCscF32.init_from(W), clib.c_sparse_inner_products_drm2csc_f32
        elif isinstance(X, smat.csr_matrix) and isinstance(W, np.ndarray):
            pX, pW, fn = ScipyCsrF32.init_from(X), ScipyDcmF32.init_from(W), clib.c_sparse_inner_products_csr2dcm_f32
        elif isinstance(X, np.ndarray) and isinstance(W, np.ndarray):
            pX, pW, fn = ScipyDrmF32.init_from(X), ScipyDcmF32.init_from(W), clib.c_sparse_inner_products_drm2dcm_f32
        else:
            raise NotImplementedError("type(X)={} and type(W)={} no implemented".format(type(X), type(W)))
        if pred_values is None or len(pred_values) != nnz or pred_values.dtype != np.float32:
            pred_values = np.zeros(nnz, pW.dtype)
        rows = np.ascontiguousarray(X_row_idx, dtype=np.uint32)
        cols = np.ascontiguousarray(W_col_idx, dtype=np.uint32)
        fn(byref(pX), byref(pW), nnz, rows.ctypes.data_as(POINTER(c_uint32)), cols.ctypes.data_as(POINTER(c_uint32)),
           pred_values.ctypes.data_as(POINTER(c_float)), threads)
        self._check()
        return pred_values

    # ------------------------------------------------------------------ device-resident path (additive)
    def queries_upload(self, c_model, X):
        lib = self.clib_float32
        if isinstance(X, smat.csr_matrix):
            if not X.has_sorted_indices:
                raise ValueError("Query matrix does not have sorted indices!")
            h = lib.xrl_queries_upload_csr(c_void_p(c_model), byref(ScipyCsrF32.init_from(X)))
        elif isinstance(X, np.ndarray):
            h = lib.xrl_queries_upload_drm(c_void_p(c_model), byref(ScipyDrmF32.init_from(X)))
        else:
            raise NotImplementedError("type(X) = {} not implemented".format(type(X)))
        self._check()
        return h

    def queries_from_device_csr(self, c_model, rows, cols, row_ptr_addr, col_idx_addr, val_addr, nnz):
        """Wrap a CSR that already lives in HBM (raw device addresses: u64 row_ptr, u32 col_idx, f32 val); non-owning."""
        h = self.clib_float32.xrl_queries_from_device_csr(c_void_p(c_model), rows, cols, c_void_p(row_ptr_addr), c_void_p(col_idx_addr),
                                                          c_void_p(val_addr), nnz)
        self._check()
        return h

    def queries_from_device_drm(self, c_model, rows, cols, val_addr):
        h = self.clib_float32.xrl_queries_from_device_drm(c_void_p(c_model), rows, cols, c_void_p(val_addr))
        self._check()
        return h

    def queries_tfidf_device(self, c_model, rows, cols, row_ptr_addr, col_idx_addr, count_addr, nnz, idf_addr=None, binary=False,
                             sublinear_tf=False, norm_p=2, stream=None, out_addr=None):
        """Device CSR of term counts -> the reference's tf-idf weighting + normalisation on the device (tfidf.hpp:798-822);
        the handle references row_ptr / col_idx (keep them alive); the weighted values go to out_addr (nnz floats) or a buffer it owns."""
        h = self.clib_float32.xrl_queries_tfidf_device(c_void_p(c_model), rows, cols, c_void_p(row_ptr_addr), c_void_p(col_idx_addr),
                                                       c_void_p(count_addr), nnz, c_void_p(idf_addr or 0), 1 if binary else 0,
                                                       1 if sublinear_tf else 0, int(norm_p), c_void_p(out_addr or 0), c_void_p(stream or 0))
        self._check()
        return h

    def queries_concat_device(self, c_model, rows, sparse_cols, row_ptr_addr, col_idx_addr, val_addr, nnz, dense_cols, emb_addr, stream=None,
                              normalize_emb=False):
        """[X_feat (device CSR) | X_emb (device dense)] -> one device CSR (XR-Transformer concat_model's query form);
        normalize_emb: l2-normalise the rows of X_emb on the device first (the reference's default)."""
        h = self.clib_float32.xrl_queries_concat_device_ex(c_void_p(c_model), rows, sparse_cols, c_void_p(row_ptr_addr), c_void_p(col_idx_addr),
                                                           c_void_p(val_addr), nnz, dense_cols, c_void_p(emb_addr), 1 if normalize_emb else 0,
                                                           c_void_p(stream or 0))
        self._check()
        return h

    # ---- TF-IDF query producer (pecos/core/base.py:1696-1725, 1820-1862: tfidf_load / tfidf_destruct / tfidf_predict)
    def tfidf_load(self, load_dir):
        h = self.clib_float32.c_tfidf_load(c_char_p(load_dir.encode("utf-8")))
        self._check()
        return h

    def tfidf_destruct(self, model):
        if self._lib is not None and model:
            self._lib.c_tfidf_destruct(c_void_p(model))

    @staticmethod
    def _corpus_arrays(corpus):
        """(char** as a ctypes pointer, byte lengths u64[n], n) for a list of str / bytes.  The documents are packed into ONE bytes object
        and the pointer table is numpy arithmetic on its address: per-document Python work is what bounds the producer once the native
        half runs at millions of documents a second, so an all-ASCII corpus is encoded in one go (byte length == len(str))."""
        nr_doc = len(corpus)
        if nr_doc == 0:
            return c_void_p(0), np.zeros(1, dtype=np.uint64), 0
        if all(type(d) is str for d in corpus):
            joined = "".join(corpus)
            if joined.isascii():
                buf = joined.encode("ascii")
                lens = np.fromiter(map(len, corpus), dtype=np.uint64, count=nr_doc)
            else:
                enc = [d.encode("utf-8") for d in corpus]
                buf = b"".join(enc)
                lens = np.fromiter(map(len, enc), dtype=np.uint64, count=nr_doc)
        else:
            enc = [d.encode("utf-8") if isinstance(d, str) else bytes(d) for d in corpus]
            buf = b"".join(enc)
            lens = np.fromiter(map(len, enc), dtype=np.uint64, count=nr_doc)
        base = ctypes.cast(c_char_p(buf), c_void_p).value or 0
        ptrs = np.empty(nr_doc, dtype=np.uint64)
        ptrs[0] = base
        if nr_doc > 1:
            np.cumsum(lens[:-1], out=ptrs[1:]); ptrs[1:] += np.uint64(base)
        arr = ptrs.ctypes.data_as(c_void_p)
        arr._keep = (buf, ptrs)               # the table and the text live as long as the pointer object does
        return arr, lens, nr_doc

    def tfidf_predict(self, model, corpus, buffer_size=0, threads=-1):
        """Vectorize a list of strings, or -- corpus given as a path -- the lines of a text file (pecos/core/base.py:1820-1863)."""
        pred_alloc = ScipyCompressedSparseAllocator()
        if isinstance(corpus, str):
            assert os.path.isfile(corpus), "Cannot predict from {}!".format(corpus)
            corpus_utf8 = corpus.encode("utf-8")
            self.clib_float32.c_tfidf_predict_from_file(c_void_p(model), ctypes.cast(c_char_p(corpus_utf8), c_void_p), len(corpus_utf8), buffer_size, threads, pred_alloc.cfunc)
            self._check()
            return pred_alloc.get()
        arr, lens, nr_doc = self._corpus_arrays(corpus)
        self.clib_float32.c_tfidf_predict(c_void_p(model), arr, lens.ctypes.data_as(POINTER(c_uint64)), nr_doc, threads, pred_alloc.cfunc)
        self._check()
        return pred_alloc.get()

    def tfidf_counts(self, model, corpus, threads=-1):
        """Host only: the CSR of term COUNTS (hstacked over an ensemble's base vectorizers) the device weighting starts from."""
        pred_alloc = ScipyCompressedSparseAllocator()
        arr, lens, nr_doc = self._corpus_arrays(corpus)
        self.clib_float32.xrl_tfidf_counts(c_void_p(model), arr, lens.ctypes.data_as(POINTER(c_uint64)), nr_doc, threads, pred_alloc.cfunc)
        self._check()
        return pred_alloc.get()

    def tfidf_nr_features(self, model):
        v = int(self.clib_float32.xrl_tfidf_nr_features(c_void_p(model)))
        self._check()
        return v

    def tfidf_predict_device(self, model, c_model, corpus, threads=-1):
        """Texts -> tf-idf X that STAYS on c_model's device: a query handle for predict_device (free it with queries_free)."""
        arr, lens, nr_doc = self._corpus_arrays(corpus)
        h = self.clib_float32.xrl_tfidf_predict_device(c_void_p(model), c_void_p(c_model), arr, lens.ctypes.data_as(POINTER(c_uint64)), nr_doc, threads)
        self._check()
        return h

    def queries_concat_handle(self, c_model, queries, dense_cols, emb_addr, normalize_emb=False, stream=None):
        h = self.clib_float32.xrl_queries_concat_handle(c_void_p(c_model), c_void_p(queries), dense_cols, c_void_p(emb_addr), 1 if normalize_emb else 0, c_void_p(stream or 0))
        self._check()
        return h

    def queries_download(self, h):
        """Copy of a query handle's matrix back to the host (scipy CSR with the stored order, or ndarray)."""
        info = (ctypes.c_uint64 * 4)()
        fn = self.clib_float32.xrl_queries_info
        fn.restype = ctypes.c_int; fn.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        fn(c_void_p(h), info); self._check()
        rows, cols, nnz, dense = (int(v) for v in info)
        dl = self.clib_float32.xrl_queries_download
        dl.restype = ctypes.c_int; dl.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
        if dense:
            out = np.zeros((rows, cols), dtype=np.float32)
            dl(c_void_p(h), None, None, out.ctypes.data_as(c_void_p)); self._check()
            return out
        ptr = np.zeros(rows + 1, dtype=np.uint64); idx = np.zeros(max(nnz, 1), dtype=np.uint32); val = np.zeros(max(nnz, 1), dtype=np.float32)
        dl(c_void_p(h), ptr.ctypes.data_as(c_void_p), idx.ctypes.data_as(c_void_p), val.ctypes.data_as(c_void_p)); self._check()
        m = smat.csr_matrix((rows, cols), dtype=np.float32)
        m.indptr, m.indices, m.data = ptr.astype(np.int64), idx[:nnz].astype(np.int64), val[:nnz]      # assigned directly: keeps explicit zeros / order
        return m

    def queries_free(self, h):
        if self._lib is not None and h:
            self._lib.xrl_queries_free(c_void_p(h))

    def predict_device(self, c_model, queries, beam_size, post_processor, only_topk, d_idx, d_val, d_cnt, out_stride,
                       stream=None, sync=True):
        """``d_*`` are raw device addresses (e.g. ``tensor.data_ptr()``)."""
        rc = self.clib_float32.xrl_predict_device(
            c_void_p(c_model), c_void_p(queries), beam_size or 0,
            post_processor.encode("utf-8") if post_processor else None, only_topk or 0,
            c_void_p(d_idx), c_void_p(d_val), c_void_p(d_cnt), out_stride, c_void_p(stream or 0), 1 if sync else 0)
        self._check()
        return rc

    def predict_device_rows(self, c_model, queries, beam_size, post_processor, only_topk, d_idx, d_val, d_cnt, out_stride,
                            row_begin, row_count, stream=None, sync=True):
        """predict_device for rows [row_begin, row_begin + row_count) only; results land at the same rows of the output buffers."""
        rc = self.clib_float32.xrl_predict_device_rows(
            c_void_p(c_model), c_void_p(queries), beam_size or 0,
            post_processor.encode("utf-8") if post_processor else None, only_topk or 0,
            c_void_p(d_idx), c_void_p(d_val), c_void_p(d_cnt), out_stride, c_void_p(stream or 0), 1 if sync else 0,
            int(row_begin), int(row_count))
        self._check()
        return rc

    def effective_topk(self, c_model, only_topk):
        return int(self.clib_float32.xrl_effective_topk(c_void_p(c_model), only_topk or 0))

    def debug_split_chunk(self, col_nnz, limit):
        """Host-only: number of even column tiles the model compiler cuts a chunk with these column nnz into."""
        cum = np.concatenate([[0], np.cumsum(np.asarray(col_nnz, dtype=np.uint64))]).astype(np.uint64)
        fn = self.clib_float32.xrl_debug_split_chunk
        fn.restype = ctypes.c_uint32
        fn.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.c_uint64]
        return int(fn(cum.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(col_nnz), int(limit)))

    def debug_layout_rows(self, row_len, align=True):
        """Host-only: (offset, length) of every tile row and the padded entry count, as the model compiler places them."""
        rptr = np.concatenate([[0], np.cumsum(np.asarray(row_len, dtype=np.int64))]).astype(np.uint32)
        ext = np.zeros(len(row_len), dtype=np.uint32)
        fn = self.clib_float32.xrl_debug_layout_rows
        fn.restype = ctypes.c_uint64
        fn.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        total = int(fn(rptr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(row_len), 1 if align else 0,
                       ext.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        self._check()
        return ext & 0x1FFFFFF, (ext >> 25) + 1, total

    def profile_enable(self, c_model, on=True):
        self.clib_float32.xrl_profile_enable(c_void_p(c_model), 1 if on else 0)

    def profile_reset(self, c_model):
        self.clib_float32.xrl_profile_reset(c_void_p(c_model))

    def profile_get(self, c_model):
        n = self.clib_float32.xrl_profile_get(c_void_p(c_model), None, 0)
        arr = (ProfileRec * max(1, n))()
        n = self.clib_float32.xrl_profile_get(c_void_p(c_model), arr, n)
        return [dict(name=arr[i].name.decode(), layer=arr[i].layer, launches=arr[i].launches, ms=arr[i].ms)
                for i in range(n)]

    def predict_stats(self, c_model, queries, beam_size, post_processor, only_topk):
        """Untimed predict (tile-format kernels) returning per layer a dict of work counters: ref_chunk_bytes (SURVEY.md 8d:
        every active reference chunk streamed whole), candidates, items, probes, hit_rows, hit_entries, item_cols, x_cols."""
        depth = self.xlinear_get_int_attr(c_model, "depth")
        out = (c_double * (8 * depth))()
        self.clib_float32.xrl_predict_stats(c_void_p(c_model), c_void_p(queries), beam_size or 0,
                                            post_processor.encode("utf-8") if post_processor else None,
                                            only_topk or 0, out, 8 * depth)
        self._check()
        keys = ("ref_chunk_bytes", "candidates", "items", "probes", "hit_rows", "hit_entries", "item_cols", "x_cols")
        return [dict(zip(keys, [out[8 * l + i] for i in range(8)])) for l in range(depth)]

    def set_option(self, c_model, key, value):
        self.clib_float32.xrl_set_option(c_void_p(c_model), key.encode("utf-8"), int(value))
        self._check()

    def debug_k1_phases(self, reset=True):
        out = (c_uint64 * 8)()
        self.clib_float32.xrl_debug_k1_phases(out, 1 if reset else 0)
        self._check()
        return [int(v) for v in out]

    def single_layer_cache_clear(self):
        """Drop every cached single-layer handle (call after modifying W / C in place)."""
        self.clib_float32.xrl_single_layer_cache_clear()

    def single_layer_cache_stats(self):
        h, m, e = c_uint64(0), c_uint64(0), c_uint64(0)
        self.clib_float32.xrl_single_layer_cache_stats(byref(h), byref(m), byref(e))
        return dict(hits=int(h.value), misses=int(m.value), entries=int(e.value))

    def layer_info(self, c_model, layer):
        out = (c_uint64 * 12)()
        self.clib_float32.xrl_layer_info(c_void_p(c_model), layer, out, 12)
        self._check()
        keys = ("lookup", "bucket_levels", "dense", "dense_tile_width", "dense_ld", "tiles", "entries", "dense_bytes", "w_rows",
                "children", "max_tile_cols", "device_bytes")
        return dict(zip(keys, [int(v) for v in out]))

    def model_device_bytes(self, c_model):
        return int(self.clib_float32.xrl_model_device_bytes(c_void_p(c_model)))


clib = corelib()
