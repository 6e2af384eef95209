"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE REFERENCE here.

Needs /root/reference (read-only) and oracle/_ref/libpecos_float32.so (``make -C oracle ref``).
The reference python package is imported from a scratch copy (/tmp) with the compiled .so dropped
into pecos/core/ and two numpy-2 / scipy-1.15 shims in pecos/utils/smat_util.py;
nothing is written to /root/reference and no reference source is copied into this repo.

What it writes (all small):
  ref_fixtures/*.npz          the reference's own test fixtures for this path
                              (test/tst-data/xmc/xlinear/{X,Xt,Y,Yt,Yt_pred,Yt_pred_with_tfn+man,
                              P_nr_splits=2,P_nr_splits=4}.npz) -- data, not code
  models/<name>/              toy models TRAINED BY THE REFERENCE with the exact command lines of
                              test/pecos/xmc/xlinear/test_xlinear.py:106-143 and :314-640
  preds/<name>__<case>.npz    reference predict-only (BINARY_SEARCH_CHUNKED) outputs for every
                              post-processor x {sparse, dense} at beam_size=2 (test_xlinear.py:147-245)
  synth/<name>/ + synth_preds  seeded synthetic models (xrl_synth) and the reference's outputs on them
  manifest.json               index of all cases
"""
import json
import os
import shlex
import shutil
import subprocess
import sys

import numpy as np
import scipy.sparse as smat

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SCRATCH = "/tmp/xrl_refpy"
sys.path.insert(0, REPO)


def prepare_reference_package():
    if os.path.exists(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree(os.path.join(REF, "pecos"), os.path.join(SCRATCH, "pecos"))
    shutil.copy(os.path.join(REPO, "oracle", "_ref", "libpecos_float32.so"),
                os.path.join(SCRATCH, "pecos", "core", "libpecos_float32.so"))
    p = os.path.join(SCRATCH, "pecos", "utils", "smat_util.py")
    s = open(p).read()
    s = s.replace("smat.sputils.get_index_dtype", "smat._sputils.get_index_dtype")
    s = s.replace("smat.sputils.upcast", "smat._sputils.upcast")
    s = s.replace("np.array(X.indices, dtype=idx_dtype, copy=False)", "np.asarray(X.indices, dtype=idx_dtype)")
    s = s.replace("np.array(X.indptr, dtype=idx_dtype, copy=False)", "np.asarray(X.indptr, dtype=idx_dtype)")
    s = s.replace("copy=False)", "copy=None)")
    open(p, "w").write(s)


def run(cmd):
    env = dict(os.environ, PYTHONPATH=SCRATCH)
    r = subprocess.run(shlex.split(cmd), cwd=SCRATCH, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(cmd + "\n" + r.stderr.decode()[-3000:])
    return r.stdout.decode()


def save(path, m):
    smat.save_npz(path, smat.csr_matrix(m), compressed=True)


def main():
    prepare_reference_package()
    fx = os.path.join(REF, "test/tst-data/xmc/xlinear")
    out_fx = os.path.join(HERE, "ref_fixtures")
    os.makedirs(out_fx, exist_ok=True)
    for n in ["X", "Xt", "Y", "Yt", "Yt_pred", "Yt_pred_with_tfn+man", "P_nr_splits=2", "P_nr_splits=4"]:
        shutil.copy(os.path.join(fx, n + ".npz"), os.path.join(out_fx, n + ".npz"))

    train = {  # name -> extra train args (test_xlinear.py)
        "default": "",
        "splits2": "--nr-splits 2 --max-leaf-size 2",
        "splits4": "--nr-splits 4 --max-leaf-size 2",
        "mls10": "--max-leaf-size 10",
        "tfn_man": "--max-leaf-size 10 -pp noop -b 2 -ns tfn+man",
    }
    models_dir = os.path.join(HERE, "models")
    if os.path.exists(models_dir):
        shutil.rmtree(models_dir)
    for name, extra in train.items():
        mf = os.path.join(models_dir, name)
        run(f"python3 -m pecos.xmc.xlinear.train -x {fx}/X.npz -y {fx}/Y.npz -m {mf} {extra}")

    sys.path.insert(0, SCRATCH)
    from pecos.xmc import PostProcessor
    from pecos.xmc.xlinear import XLinearModel as ref_xlm
    from pecos.utils import smat_util

    manifest = {"toy": [], "cli": [], "synth": []}
    preds = os.path.join(HERE, "preds")
    if os.path.exists(preds):
        shutil.rmtree(preds)
    os.makedirs(preds)
    Xt = smat_util.load_matrix(f"{fx}/Xt.npz").tocsr().astype(np.float32)
    Xt.sort_indices()
    for name in ["default", "splits2", "splits4"]:
        mf = os.path.join(models_dir, name)
        m = ref_xlm.load(mf, is_predict_only=True, weight_matrix_type="BINARY_SEARCH_CHUNKED")
        py_m = ref_xlm.load(mf)
        for pp in PostProcessor.valid_list():
            for kind, Xq in (("sparse", Xt), ("dense", np.ascontiguousarray(Xt.toarray()))):
                P = m.predict(Xq, post_processor=pp, beam_size=2)
                P_py = py_m.predict(Xq, post_processor=pp, beam_size=2)
                assert np.allclose(P.toarray(), P_py.toarray(), atol=1e-6)
                fn = f"{name}__{pp}__{kind}.npz"
                save(os.path.join(preds, fn), P)
                manifest["toy"].append(dict(model=name, post_processor=pp, beam_size=2, x=kind, pred=fn))
    # the CLI goldens of test_cli (prediction kwargs as on those command lines)
    manifest["cli"] = [
        dict(model="mls10", kwargs={}, golden="Yt_pred.npz"),
        dict(model="tfn_man", kwargs={"post_processor": "sigmoid", "beam_size": 4}, golden="Yt_pred_with_tfn+man.npz"),
        dict(model="splits2", kwargs={"max_pred_chunk": 2}, golden="P_nr_splits=2.npz"),
        dict(model="splits4", kwargs={"max_pred_chunk": 2}, golden="P_nr_splits=4.npz"),
    ]
    for c in manifest["cli"]:
        m = ref_xlm.load(os.path.join(models_dir, c["model"]), is_predict_only=True)
        P = m.predict(Xt, **c["kwargs"])
        G = smat_util.load_matrix(f"{fx}/{c['golden']}")
        assert np.allclose(P.toarray(), G.toarray(), atol=1e-6), c

    # seeded synthetic models + the reference's outputs (bit-exact targets for the oracle and the GPU)
    import xrl_synth
    synth_dir = os.path.join(HERE, "synth")
    if os.path.exists(synth_dir):
        shutil.rmtree(synth_dir)
    cases = [
        dict(name="s_eurlex", D=600, L=900, w_nnz=[200, 120, 30], x_nnz=40, N=64, kw=dict(shape=[4, 32, 900])),
        dict(name="s_contig", D=300, L=500, w_nnz=[100, 60, 20], x_nnz=25, N=48, kw=dict(shape=[4, 24, 500], permute_leaf=False)),
        dict(name="s_pruned", D=300, L=500, w_nnz=[100, 60, 20], x_nnz=25, N=48, kw=dict(shape=[4, 24, 500], prune=0.2)),
        dict(name="s_deep", D=2000, L=5000, w_nnz=[300, 200, 100, 60, 12], x_nnz=30, N=48, kw=dict(shape=[2, 8, 32, 128, 5000])),
        dict(name="s_nobias", D=300, L=500, w_nnz=[100, 60, 20], x_nnz=25, N=48, kw=dict(shape=[4, 24, 500], bias=-1.0)),
        dict(name="s_flat", D=200, L=300, w_nnz=[30], x_nnz=20, N=32, kw=dict(shape=[300])),
        dict(name="s_wide", D=400, L=1500, w_nnz=[150, 25], x_nnz=30, N=32, kw=dict(shape=[5, 1500])),
    ]
    for c in cases:
        folder = os.path.join(synth_dir, c["name"])
        kw = dict(c["kw"])
        ks = xrl_synth.make_model(folder, c["D"], c["L"], c["w_nnz"], seed=7, **kw)
        X = xrl_synth.make_queries(c["N"], c["D"], c["x_nnz"], seed=8, relabel_seed=7)
        # a few adversarial rows: empty row, single feature, explicit zero value
        X = X.tolil(); X[0, :] = 0; X = X.tocsr().astype(np.float32); X.eliminate_zeros(); X.sort_indices()
        smat.save_npz(os.path.join(synth_dir, c["name"] + "__X.npz"), X, compressed=True)
        m = ref_xlm.load(folder, is_predict_only=True, weight_matrix_type="BINARY_SEARCH_CHUNKED")
        for (beam, topk, pp) in [(10, 10, None), (3, 5, "sigmoid"), (2, 20, "log-l2-hinge"), (40, 64, "noop"), (70, 100, None)]:
            kwargs = dict(beam_size=beam, only_topk=topk)
            if pp:
                kwargs["post_processor"] = pp
            for kind, Xq in (("sparse", X), ("dense", np.ascontiguousarray(X.toarray()))):
                P = m.predict(Xq, **kwargs)
                fn = f"{c['name']}__b{beam}_k{topk}_{pp}__{kind}.npz"
                # keep the score-sorted row order: store raw CSR arrays
                np.savez_compressed(os.path.join(preds, fn), indptr=P.indptr, indices=P.indices, data=P.data, shape=P.shape)
                manifest["synth"].append(dict(model=c["name"], layers=ks, kwargs=kwargs, x=kind, pred=fn))
    # memory-mapped model folders COMPILED BY THE REFERENCE (XLinearModel.compile_mmap_model,
    # pecos/xmc/xlinear/model.py:136-152 -> c_xlinear_compile_mmap_model) for the mmap reader tests
    mmap_dir = os.path.join(HERE, "mmap")
    if os.path.exists(mmap_dir):
        shutil.rmtree(mmap_dir)
    manifest["mmap"] = []
    for kind, name in (("models", "splits2"), ("models", "mls10"), ("synth", "s_eurlex"), ("synth", "s_pruned"), ("synth", "s_contig")):
        ref_xlm.compile_mmap_model(os.path.join(HERE, kind, name), os.path.join(mmap_dir, name))
        manifest["mmap"].append(dict(kind=kind, model=name))
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    os.system(f"du -sh {HERE}")


if __name__ == "__main__":
    main()
