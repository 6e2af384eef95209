"""First-contact GPU script: parity vs oracle/_ref on a synthetic model + a rough timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth
from oracle import xrl_oracle as O
from pecos_amd import XLinearModel, clib

def compare(a, b, tag):
    same_ptr = np.array_equal(a.indptr, b.indptr)
    same_idx = same_ptr and np.array_equal(a.indices, b.indices)
    bit = same_idx and np.array_equal(a.data.view(np.uint32), b.data.view(np.uint32))
    rel = float(np.max(np.abs(a.data - b.data) / np.maximum(np.abs(b.data), 1e-30))) if same_ptr and a.nnz else -1
    print(f"  [{tag}] indptr={same_ptr} idx={same_idx} bit={bit} maxrel={rel:.3g}", flush=True)
    if not same_idx and same_ptr:
        bad = np.nonzero(a.indices != b.indices)[0]
        rows = np.searchsorted(a.indptr, bad[:5], side='right') - 1
        print("   first mismatching rows", rows, flush=True)
        r = rows[0]; s, e = a.indptr[r], a.indptr[r+1]
        print("   gpu", a.indices[s:e], a.data[s:e]); print("   ref", b.indices[s:e], b.data[s:e])
    return same_idx

name = sys.argv[1] if len(sys.argv) > 1 else "eurlex-4k"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
folder = f"/tmp/xrl_{name}_{scale}"
t = time.time(); ks, X, cfg = xrl_synth.make_config(name, folder, scale=scale); print("synth", ks, X.shape, f"{time.time()-t:.1f}s", flush=True)
print("devices", clib.device_count(), flush=True)
t = time.time(); m = XLinearModel.load(folder); print(f"load {time.time()-t:.2f}s  device MB {clib.model_device_bytes(m.model.model_chain)/1e6:.1f}", flush=True)
beam = cfg["beam"]
ns = min(X.shape[0], 2000)
Xs = X[:ns]
ref = O.RefModel(folder) if O.ref_available() else O.OracleModel.load(folder)
for pp in [None, "sigmoid", "log-sigmoid", "l2-hinge", "log-l3-hinge", "noop"]:
    b = ref.predict(Xs, beam_size=beam, only_topk=10, post_processor=pp)
    for g in [0, 1, 4, 16, 64]:
        clib.set_option(m.model.model_chain, "k1_group", g)
        kw = dict(beam_size=beam, only_topk=10)
        if pp: kw["post_processor"] = pp
        a = m.predict(Xs, **kw)
        compare(a, b, f"pp={pp} G={g}")
# dense path
Xd = np.ascontiguousarray(Xs[:256].toarray()) if hasattr(Xs, "toarray") else Xs[:256]
b = ref.predict(Xd, beam_size=beam, only_topk=10)
for g in [0, 1, 8, 64]:
    clib.set_option(m.model.model_chain, "k1_group", g)
    compare(m.predict(Xd, beam_size=beam, only_topk=10), b, f"dense G={g}")
# big k (LDS top-k path)
clib.set_option(m.model.model_chain, "k1_group", 0)
compare(m.predict(Xs[:300], beam_size=100, only_topk=150), ref.predict(Xs[:300], beam_size=100, only_topk=150), "beam=100 topk=150")
# timing through the host ABI and device-resident
for g in [0, 1, 2, 4, 8, 16]:
    clib.set_option(m.model.model_chain, "k1_group", g)
    m.predict(X, beam_size=beam, only_topk=10)
    t = time.time(); m.predict(X, beam_size=beam, only_topk=10); dt = time.time() - t
    print(f"host-ABI predict G={g}: N={X.shape[0]} {dt*1e3:.1f} ms  {X.shape[0]/dt:.0f} q/s", flush=True)
clib.set_option(m.model.model_chain, "k1_group", 0)
clib.profile_enable(m.model.model_chain, True)
m.predict(X, beam_size=beam, only_topk=10)
for r in clib.profile_get(m.model.model_chain):
    print("  ", r)
