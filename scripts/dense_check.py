"""Dense-input config (BASELINE.json configs[4], scaled): parity vs reference + timing."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth
from oracle import xrl_oracle as O
from pecos_amd import XLinearModel, clib
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
folder = f"/tmp/xrl_bench/dense768_{scale}"
L = int(3000000 * scale); N = int(1000000 * scale)
ks = xrl_synth.tree_shape(L)
w_nnz = [768] * (len(ks) - 1) + [256]
t = time.time()
if not os.path.exists(folder + "/.done"):
    xrl_synth.make_model(folder, 768, L, w_nnz, seed=0); open(folder + "/.done", "w").write("ok")
X = xrl_synth.make_queries(N, 768, None, seed=1)
print("synth", ks, X.shape, f"{time.time()-t:.1f}s", flush=True)
m = XLinearModel.load(folder); h = m.model.model_chain
print("model GB", clib.model_device_bytes(h) / 1e9, flush=True)
ref = O.RefModel(folder)
ns = 512
a = m.predict(X[:ns], beam_size=10, only_topk=10); b = ref.predict(X[:ns], beam_size=10, only_topk=10)
print("parity idx", np.array_equal(a.indices, b.indices), "bit", np.array_equal(a.data.view(np.uint32), b.data.view(np.uint32)), flush=True)
q = clib.queries_upload(h, X)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def dmalloc(n):
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0; return p.value
k = 10
di, dv, dc = dmalloc(N * k * 4), dmalloc(N * k * 4), dmalloc(N * 4)
for g in [0, 8, 16, 32, 64]:
    clib.set_option(h, "k1_group", g)
    clib.predict_device(h, q, 10, None, k, di, dv, dc, k, sync=True)
    clib.profile_reset(h); clib.profile_enable(h, True)
    t0 = time.perf_counter()
    for _ in range(3): clib.predict_device(h, q, 10, None, k, di, dv, dc, k, sync=True)
    dt = (time.perf_counter() - t0) / 3
    clib.profile_enable(h, False)
    row = {}
    for r in clib.profile_get(h): row[(r["name"], r["layer"])] = r["ms"] / r["launches"]
    print(f"G={g:2d} {dt*1e3:8.2f} ms  {N/dt/1e6:.3f} Mq/s  k1/layer " + " ".join(f"{row.get(('k1_dense', l), 0):8.3f}" for l in range(m.depth)), flush=True)
t0 = time.perf_counter(); ref.predict(X[:4096], beam_size=10, only_topk=10, threads=32); t1 = time.perf_counter() - t0
print(f"cpu ref 32 thr: {4096/t1:.0f} q/s")
