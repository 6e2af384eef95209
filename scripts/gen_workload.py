#!/usr/bin/env python3
"""Materialise a bench workload under /tmp/xrl_bench exactly as bench.py would (host-only): lets a GPU call generate the big synthetic
models in the background while GPU tests run.   python scripts/gen_workload.py <config> [cache]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as smat
import xrl_synth
name = sys.argv[1]; cache = sys.argv[2] if len(sys.argv) > 2 else "/tmp/xrl_bench"
folder = os.path.join(cache, f"{name}_1.0")
if not os.path.exists(os.path.join(folder, ".done")):
    os.makedirs(folder, exist_ok=True)
    ks, X, cfg = xrl_synth.make_config(name, folder, scale=1.0)
    if smat.issparse(X):
        smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
    else:
        np.save(os.path.join(folder, "X.npy"), X)
    json.dump({"ks": ks, "cfg": cfg}, open(os.path.join(folder, "meta.json"), "w"))
    open(os.path.join(folder, ".done"), "w").write("ok")
print("ready", folder)
