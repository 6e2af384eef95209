import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as smat

try:        # one HIP runtime per process: torch's bundled copy must be the one in place before any test loads libxrl_amd.so directly (ctypes.CDLL)
    import torch  # noqa: F401
except Exception:
    pass

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import xrl_oracle
    xrl_oracle.build()
    return xrl_oracle


def load_raw_csr(path):
    """Golden predictions are stored as raw CSR arrays to keep the score-sorted order inside rows."""
    z = np.load(path)
    if "format" in z.files:  # scipy save_npz (toy goldens; order canonical)
        return smat.load_npz(path).tocsr()
    return smat.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))


def load_X(path, kind="sparse"):
    X = smat.load_npz(path).tocsr().astype(np.float32)
    X.sort_indices()
    return X if kind == "sparse" else np.ascontiguousarray(X.toarray())


def assert_same_topk(a, b, rel=1e-5, exact_scores=False, what=""):
    """Parity bar of BASELINE.json: identical row lengths, identical label sequence (order included),
    scores within `rel` relative (bit-exact when exact_scores)."""
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.array_equal(a.indptr, b.indptr), f"{what}: row lengths differ"
    assert np.array_equal(a.indices, b.indices), f"{what}: label ids / order differ"
    if exact_scores:
        assert np.array_equal(a.data.astype(np.float32).view(np.uint32), b.data.astype(np.float32).view(np.uint32)), \
            f"{what}: scores not bit-identical"
    else:
        err = np.abs(a.data - b.data)
        tol = rel * np.abs(b.data) + 1e-37
        assert np.all(err <= tol), f"{what}: max rel err {np.max(err / np.maximum(np.abs(b.data), 1e-30))}"


def assert_topk_close(a, b, rel=1e-5, what="", atol=1e-37):
    """Where the two sides may legitimately round differently (a summation ORDER the reference itself leaves to a hash table): same row lengths, every
    label both sides return scores within `rel`, the i-th scores agree within `rel`, and a label only one side returns sits within `rel` of that
    side's last (k-th) score -- i.e. the lists differ at most by swaps of near-ties."""
    assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr), f"{what}: shapes / row lengths differ"
    for r in range(a.shape[0]):
        lo, hi = a.indptr[r], a.indptr[r + 1]
        if hi == lo:
            continue
        ia, va, ib, vb = a.indices[lo:hi], a.data[lo:hi], b.indices[lo:hi], b.data[lo:hi]
        tol = rel * np.maximum(np.abs(va), np.abs(vb)) + atol
        assert np.all(np.abs(va - vb) <= tol), f"{what}: row {r}: i-th scores differ by more than {rel} (+ {atol})"
        da, db = dict(zip(ia.tolist(), va.tolist())), dict(zip(ib.tolist(), vb.tolist()))
        for lab in set(da) | set(db):
            if lab in da and lab in db:
                assert abs(da[lab] - db[lab]) <= rel * max(abs(da[lab]), abs(db[lab])) + atol, f"{what}: row {r} label {lab}"
            else:
                v, last = (da[lab], va[-1]) if lab in da else (db[lab], vb[-1])
                assert abs(v - last) <= 4 * (rel * max(abs(v), abs(last)) + atol), f"{what}: row {r}: label {lab} is not a near-tie of the k-th score"
