"""The drop-in boundary seen from the REFERENCE's side: this repo's libxrl_amd.so bound through the reference's own ctypes code
(pecos/core/base.py:799-976 ``corelib.link_xlinear_methods`` / ``link_mlmodel_methods`` / ``link_sparse_operations``) and driven by
the reference's own ``XLinearModel`` (pecos/xmc/xlinear/model.py, pecos/xmc/base.py:1090-1680) -- no code of pecos_amd's Python
layer in the call path.  The reference's package is imported from oracle/_ref/refpy (built by ``make -C oracle``; git-ignored,
travels to the GPU box).  Symbols this repo does not provide (training, clustering, tf-idf, ANN) resolve to the compiled
reference, exactly as they would in a deployment that swaps only the inference path."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, assert_same_topk, load_raw_csr, load_X

REFPY = os.path.join(REPO, "oracle", "_ref", "refpy")
OURS = os.environ.get("PECOS_XRL_AMD_SO") or os.path.join(REPO, "pecos_amd", "lib", "libxrl_amd.so")


def _bind():
    if not os.path.isdir(os.path.join(REFPY, "pecos")):
        pytest.skip("oracle/_ref/refpy (the reference's python package) is not built")
    if REFPY not in sys.path:
        sys.path.insert(0, REFPY)
    import pecos.core.base as pcb
    ours, ref = ctypes.CDLL(OURS), pcb.clib.clib_float32

    class Mixed:                       # this repo's symbols first, the compiled reference for everything it does not export
        taken = set()

        def __getattr__(self, name):
            try:
                f = getattr(ours, name); Mixed.taken.add(name)
            except AttributeError:
                f = getattr(ref, name)
            return f

    lib = pcb.corelib.__new__(pcb.corelib)
    lib.clib_float32 = Mixed()
    lib.link_mlmodel_methods()
    lib.link_xlinear_methods()
    lib.link_sparse_operations()
    lib.link_tfidf_vectorizer()
    return pcb, lib, Mixed, ours


def test_reference_prototypes_bind_to_this_library():
    # CPU part: every xlinear / sparse-inner-product entry point the reference's binding links resolves to THIS library
    pcb, lib, Mixed, ours = _bind()
    need = {"c_xlinear_load_model_from_disk", "c_xlinear_load_model_from_disk_ext", "c_xlinear_load_mmap_model_from_disk",
            "c_xlinear_compile_mmap_model", "c_xlinear_destruct_model", "c_xlinear_get_int_attr", "c_xlinear_get_layer_type",
            "c_xlinear_predict_csr_f32", "c_xlinear_predict_drm_f32", "c_xlinear_predict_on_selected_outputs_csr_f32",
            "c_xlinear_predict_on_selected_outputs_drm_f32", "c_xlinear_single_layer_predict_csr_f32", "c_xlinear_single_layer_predict_drm_f32",
            "c_xlinear_single_layer_predict_on_selected_outputs_csr_f32", "c_xlinear_single_layer_predict_on_selected_outputs_drm_f32",
            "c_sparse_inner_products_csr2csc_f32", "c_sparse_inner_products_drm2csc_f32", "c_sparse_inner_products_csr2dcm_f32",
            "c_sparse_inner_products_drm2dcm_f32"}
    need |= {"c_tfidf_load", "c_tfidf_destruct", "c_tfidf_predict"}          # round 4: the vectorizer's predict path (libpecos.cpp:398-445)
    need |= {"c_tfidf_predict_from_file"}                                     # round 6: one document per line of a file (libpecos.cpp:413-425)
    assert need <= Mixed.taken, sorted(need - Mixed.taken)
    # and the training entry points stayed with the reference
    assert "c_xlinear_single_layer_train_csr_f32" not in Mixed.taken
    assert not {"c_tfidf_train", "c_tfidf_save"} & Mixed.taken


@pytest.mark.gpu
def test_reference_xlinear_model_predicts_through_this_library(manifest):
    pcb, lib, Mixed, ours = _bind()
    import pecos.xmc.base as xb
    import pecos.xmc.xlinear.model as xm
    saved = xb.clib
    xb.clib = lib                                  # `from pecos.core import clib` in pecos/xmc/base.py: the one binding object the model classes use
    try:
        seen = 0
        for c in manifest["synth"][:12] + manifest["toy"][:6]:
            toy = "post_processor" in c and "kwargs" not in c
            folder = os.path.join(GOLDEN, "models" if toy else "synth", c["model"])
            m = xm.XLinearModel.load(folder, is_predict_only=True)
            if toy:
                X = load_X(os.path.join(GOLDEN, "ref_fixtures", "Xt.npz"), c["x"])
                P = m.predict(X, post_processor=c["post_processor"], beam_size=c["beam_size"])
                import scipy.sparse as smat
                G = smat.load_npz(os.path.join(GOLDEN, "preds", c["pred"]))
                assert np.allclose(P.toarray(), G.toarray(), atol=1e-6), c
            else:
                X = load_X(os.path.join(GOLDEN, "synth", c["model"] + "__X.npz"), c["x"])
                P = m.predict(X, **c["kwargs"])
                G = load_raw_csr(os.path.join(GOLDEN, "preds", c["pred"]))
                pp = c["kwargs"].get("post_processor")
                assert_same_topk(P, G, exact_scores=pp in (None, "noop") or "hinge" in pp, what=f"reference XLinearModel on libxrl_amd.so: {c}")
            seen += 1
            del m
        assert seen == 18
        # the handle really is this library's: its additive introspection answers
        ours.xrl_version.restype = ctypes.c_char_p
        assert b"gfx950" in ours.xrl_version()
    finally:
        xb.clib = saved


@pytest.mark.gpu
def test_reference_tfidf_vectorizer_predicts_through_this_library(manifest):
    # the reference's own Tfidf class (pecos/utils/featurization/text/vectorizers.py:163-308) on its own ctypes prototypes
    # (corelib.link_tfidf_vectorizer, base.py:1648-1694), with c_tfidf_load / c_tfidf_predict / c_tfidf_destruct resolved to THIS library:
    # the vectorizers the reference trained and saved (tests/golden/tfidf_models), its own predict() outputs as the expectation
    import json
    import scipy.sparse as smat
    pcb, lib, Mixed, ours = _bind()
    import pecos.utils.featurization.text.vectorizers as vz
    saved = vz.clib
    vz.clib = lib
    try:
        for c in manifest["tfidf_models"]:
            d = os.path.join(GOLDEN, "tfidf_models", c["name"])
            corpus = json.load(open(os.path.join(d, "corpus.json")))
            z = np.load(os.path.join(d, "X.npz"))
            vec = vz.Tfidf.load(os.path.join(d, "model"))
            P = vec.predict(corpus).tocsr()
            assert P.shape == tuple(z["shape"]) and np.array_equal(P.indptr, z["indptr"]) and np.array_equal(P.indices, z["indices"]), c
            if c["sublinear"]:
                assert np.allclose(P.data, z["data"], rtol=3e-7, atol=0), c
            else:
                assert np.array_equal(P.data.astype(np.float32).view(np.uint32), z["data"].astype(np.float32).view(np.uint32)), c
            del vec
    finally:
        vz.clib = saved
