"""K1 per-phase cycle breakdown (s_memtime instrumentation, debug option k1_ablate=64), per layer."""
import os, sys, ctypes, numpy as np, scipy.sparse as smat
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth
from pecos_amd import XLinearModel, clib
name, scale = "amazon-670k", 1.0
folder = f"/tmp/xrl_bench/{name}_{scale}"
X = smat.load_npz(folder + "/X.npz").tocsr().astype(np.float32); X.sort_indices()
m = XLinearModel.load(folder); h = m.model.model_chain
# one model per prefix depth is not available; instead run the full predict and difference by layer is impossible ->
# use single-layer timing via beam trick: run predict on truncated models? Simplest: whole-predict totals.
q = clib.queries_upload(h, X)
hip = ctypes.CDLL("libamdhip64.so")
def dmalloc(n):
    p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0; return p.value
k = 10; di, dv, dc = dmalloc(X.shape[0]*k*4), dmalloc(X.shape[0]*k*4), dmalloc(X.shape[0]*4)
clib.predict_device(h, q, 10, None, k, di, dv, dc, k, sync=True)
names = ["prologue|copy", "fill|prefetch", "D1|fill", "D3|drain", "epilogue"]   # K1 | K1T phase names
depth = clib.xlinear_get_int_attr(h, "depth")
for extra in sys.argv[1:]:
    key, val = extra.split("="); clib.set_option(h, key, int(val))
for layer in range(depth):
    clib.set_option(h, "k1_ablate", 64 | ((layer + 1) << 8))
    clib.debug_k1_phases(True)
    clib.predict_device(h, q, 10, None, k, di, dv, dc, k, sync=True)
    ph = clib.debug_k1_phases(True)
    tot = sum(ph[:5])
    print(f"layer {layer}: waves {ph[5]} cycles/wave {tot / max(1, ph[5]):.0f}  " + "  ".join(f"{n} {v / max(1, ph[5]):.0f} ({v / tot * 100:.0f}%)" for n, v in zip(names, ph[:5])))
