"""A/B timing of library options on one workload, one process (the workload is generated / loaded once):
    python scripts/ab.py <config> <scale> <steps> "<opt=val,opt=val>" "<...>" ...        ("" = the defaults)
Per option set: options reset to the defaults, then the set applied; 6 untimed predicts (the pruning feedback settles), `steps` profiled
predicts; prints ms per step, per-kernel ms per step, and whether the outputs (indices, score bits, counts) equal the FIRST set's."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as smat

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth  # noqa: E402
from pecos_amd import XLinearModel, clib  # noqa: E402

name, scale, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
sets = sys.argv[4:] or [""]
rows_limit = int(os.environ.get("AB_ROWS", "0"))
DEFAULTS = dict(qsort=1, qsort_min_parents=64, qsort_min_rows=16384, prune=1, adaptive=1, presence=1, sort_rest=1, prune_mid=1, k1q_fuse=3, dense_layers=1, k1_group=0, sort_min_tiles=0)
folder = f"/tmp/xrl_bench/{name}_{scale}"
if not os.path.exists(folder + "/.done"):
    t0 = time.time()
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale)
    smat.save_npz(folder + "/X.npz", X, compressed=False); json.dump({"ks": ks, "cfg": cfg}, open(folder + "/meta.json", "w")); open(folder + "/.done", "w").write("ok")
    print(f"generated {name} in {time.time() - t0:.0f} s", flush=True)
X = smat.load_npz(folder + "/X.npz").tocsr().astype(np.float32); X.sort_indices()
if rows_limit:
    X = X[:rows_limit]
cfg = xrl_synth.CONFIGS[name]
m = XLinearModel.load(folder); h = m.model.model_chain
q = clib.queries_upload(h, X)
k, N = 10, X.shape[0]
hip = ctypes.CDLL("libamdhip64.so")


def dmalloc(n):
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0; return p.value


def fetch(p, n, dt):
    a = np.empty(n, dt); assert hip.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), ctypes.c_size_t(a.nbytes), 2) == 0; return a


di, dv, dc = dmalloc(N * k * 4), dmalloc(N * k * 4), dmalloc(N * 4)
ref = None
for st in sets:
    for kk, vv in DEFAULTS.items():
        clib.set_option(h, kk, vv)
    for kv in st.split(","):
        if kv:
            clib.set_option(h, kv.split("=")[0], int(kv.split("=")[1]))
    for _ in range(6):
        clib.predict_device(h, q, cfg["beam"], None, k, di, dv, dc, k, sync=True)
    clib.profile_reset(h); clib.profile_enable(h, True)
    t0 = time.perf_counter()
    for _ in range(steps - 1):
        clib.predict_device(h, q, cfg["beam"], None, k, di, dv, dc, k, sync=False)
    clib.predict_device(h, q, cfg["beam"], None, k, di, dv, dc, k, sync=True)
    dt = (time.perf_counter() - t0) / steps
    clib.profile_enable(h, False)
    prof = clib.profile_get(h)
    out = (fetch(di, N * k, np.uint32), fetch(dv, N * k, np.uint32), fetch(dc, N, np.uint32))
    if ref is None:
        ref = out
    mask = (np.arange(k)[None, :] < out[2][:, None]).ravel()
    same = np.array_equal(out[2], ref[2]) and np.array_equal(out[0][mask], ref[0][mask]) and np.array_equal(out[1][mask], ref[1][mask])
    per = " ".join(f"{r['name']}@{r['layer']}={r['ms'] / steps:.3f}" for r in prof)
    print(f"[{st or 'defaults'}] {dt * 1e3:.3f} ms/step ({N / dt / 1e6:.1f} Mq/s) same_as_first={same} :: {per}", flush=True)
