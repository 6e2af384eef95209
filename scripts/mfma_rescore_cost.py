"""What the exact re-score behind an MFMA pre-score costs with a PRODUCT kernel (VERDICT r4 next #4's closing measurement).

The screen-then-rescore scheme for the dense-input leaf (profiles/r04_mfma_prescore.md) must re-score, with the reference's sequential
arithmetic, every candidate inside the band -- at least the k = 10 winners of a query, measured 10.1-11.4 per query.  This script times exactly
that work with the library's pair kernel K3 (c_sparse_inner_products_drm2csc_f32: one (query row, weight column) pair per 16 lanes, coalesced
reads of the column's entries, products folded in ascending index order = the reference's chain) on the dense-768 workload at a quarter of the
label count (N = 250 000, L = 750 000, leaf columns of 256 entries): the pairs are every query's ten final labels plus one more column.
Run under `rocprofv3 --kernel-trace --stats`; the kernel's duration stands beside K1G's first leaf stage on the same rows (bench line) and
the MFMA pre-score's measured rate (72 TFLOP/s)."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as smat

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth  # noqa: E402
from pecos_amd import XLinearModel, clib  # noqa: E402

name, scale, N = "dense-768", 0.25, 250000
folder = f"/tmp/xrl_bench/{name}_{scale}"
if not os.path.exists(folder + "/.done"):
    t0 = time.time()
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale)
    np.save(folder + "/X.npy", X); json.dump({"ks": ks, "cfg": cfg}, open(folder + "/meta.json", "w")); open(folder + "/.done", "w").write("ok")
    print(f"generated in {time.time() - t0:.0f} s", flush=True)
X = np.ascontiguousarray(np.load(folder + "/X.npy", mmap_mode="r")[:N])
m = XLinearModel.load(folder)
Y = m.predict(X, beam_size=10, only_topk=10)
depth = json.load(open(folder + "/meta.json"))["ks"]
W = smat.load_npz(os.path.join(folder, "ranker", f"{len(depth) - 1}.model", "W.npz")).tocsc().astype(np.float32)
W.sort_indices()
print("leaf W", W.shape, "nnz/col", W.nnz / W.shape[1], "labels per query", Y.nnz / N, flush=True)
Xb = np.ascontiguousarray(np.concatenate([X, np.ones((N, W.shape[0] - X.shape[1]), np.float32)], axis=1)) if W.shape[0] > X.shape[1] else X
cnt = np.diff(Y.indptr)
rows = np.repeat(np.arange(N, dtype=np.uint32), cnt + 1)
extra = np.minimum(Y.indices[np.maximum(Y.indptr[1:] - 1, 0)] + 1, W.shape[1] - 1)          # one more column per query (a neighbour of its last label)
cols = np.empty(len(rows), np.uint32)
pos = np.cumsum(cnt + 1) - 1
mask = np.ones(len(rows), bool); mask[pos] = False
cols[mask] = Y.indices; cols[pos] = extra
for it in range(3):
    t0 = time.perf_counter()
    v = clib.sparse_inner_products(Xb, W, rows, cols)
    print(f"call {it}: {len(rows)} pairs ({len(rows) / N:.1f} per query) in {(time.perf_counter() - t0) * 1e3:.1f} ms wall (incl. the uploads of X and W)", flush=True)
print("sum", float(np.sum(v[:1000])))
