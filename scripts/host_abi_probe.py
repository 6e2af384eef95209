#!/usr/bin/env python3
"""Host-ABI call loop for tracing: N back-to-back c_xlinear_predict_csr_f32 calls on the bench workload with
XRL_HOST_TIMING=1 (one stage-timing line per call on stderr) -- run plain, or under `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace`.

    python scripts/host_abi_probe.py [--config amazon-670k] [--calls 12] [--opt key=int ...]
"""
import argparse, json, os, sys, time
os.environ.setdefault("XRL_HOST_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as smat
import torch  # noqa: F401
from pecos_amd import XLinearModel, clib
from pecos_amd.core import ScipyCompressedSparseAllocator, ScipyCsrF32

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="amazon-670k"); ap.add_argument("--calls", type=int, default=12)
ap.add_argument("--cache", default="/tmp/xrl_bench"); ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--check", action="store_true", help="compare every call's CSR with the first call's, bit for bit (stress test of the pipelined path)")
ap.add_argument("--reuse-alloc", action="store_true", help="hand the SAME output arrays to every call (takes the allocator's page faults out of the picture)")
a = ap.parse_args()
folder = os.path.join(a.cache, f"{a.config}_1.0")
if not os.path.exists(os.path.join(folder, ".done")):
    import xrl_synth
    os.makedirs(folder, exist_ok=True)
    ks, X, cfg = xrl_synth.make_config(a.config, folder)
    smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
    json.dump({"ks": ks, "cfg": cfg}, open(os.path.join(folder, "meta.json"), "w")); open(os.path.join(folder, ".done"), "w").write("ok")
X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32); X.sort_indices()
clib.set_device(0)
m = XLinearModel.load(folder)
h = m.model.model_chain
for kv in a.opt:
    k, v = kv.split("="); clib.set_option(h, k, int(v))
view = ScipyCsrF32.init_from(X)
times = []


class ReusingAllocator(ScipyCompressedSparseAllocator):
    keep = {}

    def __call__(self, is_col_major, rows, cols, nnz, indices_ptr, indptr_ptr, data_ptr):
        import ctypes
        k = (rows, cols, nnz)
        if k not in self.keep:
            self.keep[k] = (np.zeros(nnz, np.uint32), np.zeros(rows + 1, np.uint64), np.zeros(nnz, np.float32))
        self.rows, self.cols, self.is_col_major = rows, cols, is_col_major
        self.indices, self.indptr, self.data = self.keep[k]
        for dst, arr in ((indices_ptr, self.indices), (indptr_ptr, self.indptr), (data_ptr, self.data)):
            ctypes.cast(dst, ctypes.POINTER(ctypes.c_uint64)).contents.value = arr.ctypes.data


for c in range(a.calls):
    alloc = ReusingAllocator() if a.reuse_alloc else ScipyCompressedSparseAllocator()
    t0 = time.perf_counter()
    clib.xlinear_predict(h, view, 10, None, 10, -1, alloc)
    times.append((time.perf_counter() - t0) * 1e3)
    print(f"[probe] call {c}: {times[-1]:.2f} ms", file=sys.stderr, flush=True)
    if a.check:
        cur = (np.array(alloc.indptr, copy=True), np.array(alloc.indices, copy=True), np.array(alloc.data, copy=True).view(np.uint32))
        if c == 0:
            first = cur
        elif not all(np.array_equal(x, y) for x, y in zip(cur, first)):
            print(json.dumps(dict(config=a.config, MISMATCH_at_call=c))); sys.exit(1)
print(json.dumps(dict(config=a.config, calls=a.calls, ms=[round(t, 3) for t in times], median=round(float(np.median(times[1:])), 3), reuse_alloc=a.reuse_alloc, opts=a.opt)))
