"""Per-layer K1 timing for every lanes-per-item variant (device-resident queries, hipEvent pairs)."""
import json, os, sys, time
import numpy as np, scipy.sparse as smat
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth
from pecos_amd import XLinearModel, clib

name = sys.argv[1] if len(sys.argv) > 1 else "amazon-670k"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
groups = [int(g) for g in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1,2,4,8,16".split(","))]
folder = f"/tmp/xrl_bench/{name}_{scale}"
if not os.path.exists(folder + "/.done"):
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale)
    smat.save_npz(folder + "/X.npz", X, compressed=False); json.dump({"ks": ks, "cfg": cfg}, open(folder + "/meta.json", "w")); open(folder + "/.done", "w").write("ok")
X = smat.load_npz(folder + "/X.npz").tocsr().astype(np.float32); X.sort_indices()
cfg = xrl_synth.CONFIGS[name]
m = XLinearModel.load(folder); h = m.model.model_chain
q = clib.queries_upload(h, X)
import ctypes
k = 10
hip = ctypes.CDLL("libamdhip64.so")
def dmalloc(n):
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0; return p.value
di, dv, dc = dmalloc(X.shape[0] * k * 4), dmalloc(X.shape[0] * k * 4), dmalloc(X.shape[0] * 4)
print("stats (chunk bytes, n_eval) per layer:", [(f"{a:.3g}", f"{b:.3g}") for a, b in clib.predict_stats(h, q, cfg["beam"], None, k)])
PP = os.environ.get("XRL_PP") or None
for extra in (sys.argv[4:] or [""]):
    for kv in extra.split(","):
        if kv: clib.set_option(h, kv.split("=")[0], int(kv.split("=")[1]))
    for g in groups:
        clib.set_option(h, "k1_group", g)
        clib.predict_device(h, q, cfg["beam"], PP, k, di, dv, dc, k, sync=True)
        clib.profile_reset(h); clib.profile_enable(h, True)
        t0 = time.perf_counter()
        for _ in range(3): clib.predict_device(h, q, cfg["beam"], PP, k, di, dv, dc, k, sync=False)
        clib.predict_device(h, q, cfg["beam"], PP, k, di, dv, dc, k, sync=True)
        dt = (time.perf_counter() - t0) / 4
        clib.profile_enable(h, False)
        prof = clib.profile_get(h)
        row = {}
        for r in prof: row[(r["name"], r["layer"])] = r["ms"] / r["launches"]
        k1 = [row.get(("k1_sparse", l), 0) for l in range(m.depth)]
        k2 = [row.get(("k2_topk", l), 0) + row.get(("k1_sort_items", l), 0) for l in range(m.depth)]
        print(f"[{extra}] G={g:2d} total {dt*1e3:7.2f} ms ({X.shape[0]/dt/1e6:.2f} Mq/s)  k1/layer " + " ".join(f"{v:7.3f}" for v in k1) + "   k2+sort/layer " + " ".join(f"{v:6.3f}" for v in k2), flush=True)
