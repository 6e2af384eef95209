"""A/B timing of library options on one workload, one process (the workload is generated / loaded once):
    python scripts/ab.py <config> <scale> <steps> "<opt=val,opt=val>" "<...>" ...        ("" = the defaults)
Per option set: options reset to the defaults, then the set applied; 6 untimed predicts (the pruning feedback settles), `steps` profiled
predicts; prints ms per step, per-kernel ms per step, and whether the outputs (indices, score bits, counts) equal the FIRST set's."""
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
shard_rows = int(os.environ.get("AB_SHARD_ROWS", "0"))   # > 0: predict only the first AB_SHARD_ROWS rows of the (whole, device-resident) X: bench.py's extra.shard8 shape
DEFAULTS = dict(tile_rows=1, k1g_first=0, k1g_variant=0, sort_rest_min=32768, qsort=1, qsort_min_parents=64, qsort_min_rows=131072, prune=1, adaptive=1, presence=1, sort_rest=1, prune_mid=1, k1q_fuse=3, dense_layers=1, k1_group=0, sort_min_tiles=0)
folder = f"/tmp/xrl_bench/{name}_{scale}"
if not os.path.exists(folder + "/.done"):
    t0 = time.time()
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale)
    if smat.issparse(X):
        smat.save_npz(folder + "/X.npz", X, compressed=False)
    else:
        np.save(folder + "/X.npy", X)
    json.dump({"ks": ks, "cfg": cfg}, open(folder + "/meta.json", "w")); open(folder + "/.done", "w").write("ok")
    print(f"generated {name} in {time.time() - t0:.0f} s", flush=True)
if os.path.exists(folder + "/X.npy"):      # dense queries (dense-768)
    X = np.load(folder + "/X.npy", mmap_mode="r")
    X = np.ascontiguousarray(X[:rows_limit] if rows_limit else X, dtype=np.float32)
else:
    X = smat.load_npz(folder + "/X.npz").tocsr().astype(np.float32); X.sort_indices()
    if rows_limit:
        X = X[:rows_limit]
cfg = xrl_synth.CONFIGS[name]
m = XLinearModel.load(folder); h = m.model.model_chain
q = clib.queries_upload(h, X)
k, N = 10, X.shape[0]
import torch  # noqa: E402  (device buffers only)
t_idx = torch.zeros(N * k, dtype=torch.int32, device="cuda"); t_val = torch.zeros(N * k, dtype=torch.int32, device="cuda"); t_cnt = torch.zeros(N, dtype=torch.int32, device="cuda")
di, dv, dc = t_idx.data_ptr(), t_val.data_ptr(), t_cnt.data_ptr()


def fetch(t):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint32)


def run(sync):
    if shard_rows:
        clib.predict_device_rows(h, q, cfg["beam"], None, k, di, dv, dc, k, 0, shard_rows, sync=sync)
    else:
        clib.predict_device(h, q, cfg["beam"], None, k, di, dv, dc, k, sync=sync)


ref = None
for st in sets:
    for kk, vv in DEFAULTS.items():
        clib.set_option(h, kk, vv)
    for kv in st.split(","):
        if kv:
            clib.set_option(h, kv.split("=")[0], int(kv.split("=")[1]))
    for _ in range(6):
        run(True)
    clib.profile_reset(h); clib.profile_enable(h, True)
    t0 = time.perf_counter()
    for _ in range(steps - 1):
        run(False)
    run(True)
    dt = (time.perf_counter() - t0) / steps
    clib.profile_enable(h, False)
    prof = clib.profile_get(h)
    out = (fetch(t_idx), fetch(t_val), fetch(t_cnt))
    if ref is None:
        ref = out
    mask = (np.arange(k)[None, :] < out[2][:, None]).ravel()
    same = np.array_equal(out[2], ref[2]) and np.array_equal(out[0][mask], ref[0][mask]) and np.array_equal(out[1][mask], ref[1][mask])
    per = " ".join(f"{r['name']}@{r['layer']}={r['ms'] / steps:.3f}" for r in prof)
    print(f"[{st or 'defaults'}] {dt * 1e3:.3f} ms/step ({N / dt / 1e6:.1f} Mq/s) same_as_first={same} :: {per}", flush=True)
