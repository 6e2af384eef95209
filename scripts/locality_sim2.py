"""Window model of K1Q's L2 reuse on the Amazon-670K workloads (CPU only; numpy beam search over levels 0-2 of the synthetic model).

One XCD runs G queries at a time (32 CUs x 4 SIMDs x 8 wavefronts = 1024); every query touches, per level, the 64-byte segment
(feature, parent) of each of its features x each beam parent the level evaluates (all of them on an unstaged layer; presence words skip
the empty ones).  Within a window every distinct segment is fetched ONCE if the window's segments fit the XCD's L2: misses >= distinct,
hits <= touches - distinct.  The script prints touches, distinct segments and the bound on the hit rate for several QUERY ORDERS of
levels 2 and 3:  launch order (block b of 4 queries -> XCD b % 8), queries sorted by the best parent of the level with each XCD taking a
contiguous eighth of the sorted order, and finer sort keys.    usage: python scripts/locality_sim2.py <cache folder> [n windows]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as smat

folder = sys.argv[1]
NWIN = int(sys.argv[2]) if len(sys.argv) > 2 else 12
G, BEAM = 1024, 10


def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32)
N, D = X.shape
W = [smat.load_npz(os.path.join(folder, "ranker", f"{d}.model", "W.npz")).tocsc() for d in range(4)]
ks = [w.shape[1] for w in W]
log("levels", ks, "queries", N, "nnz/row", X.nnz / N)


def hinge3(z):
    return np.exp(-np.maximum(1.0 - z, 0.0) ** 3)


def scores(l, rows):
    Wl, bl = W[l][:D].tocsr(), W[l][D].toarray().astype(np.float32)
    return hinge3((X[rows] @ Wl).toarray() + bl).astype(np.float32)


b1 = np.empty((N, BEAM), np.int32); b2 = np.empty((N, BEAM), np.int32)
for r0 in range(0, N, 65536):
    rows = slice(r0, min(N, r0 + 65536)); n = rows.stop - rows.start
    s0 = scores(0, rows)
    s1 = scores(1, rows) * np.repeat(s0, ks[1] // ks[0], axis=1)
    o1 = np.argsort(-s1, axis=1, kind="stable")[:, :BEAM]
    s1b = np.take_along_axis(s1, o1, axis=1)
    ch = ks[2] // ks[1]
    cand = (o1[:, :, None] * ch + np.arange(ch)[None, None, :]).reshape(n, -1)
    s2 = np.take_along_axis(scores(2, rows), cand, axis=1) * np.repeat(s1b, ch, axis=1)
    o2 = np.argsort(-s2, axis=1, kind="stable")[:, :BEAM]
    b1[rows] = o1; b2[rows] = np.take_along_axis(cand, o2, axis=1)
log("beams done; distinct best level-2 nodes", len(np.unique(b2[:, 0])), "share of the 10 most common", np.sort(np.bincount(b2[:, 0]))[-10:].sum() / N)

# non-empty (feature, parent) segments of levels 2 and 3 (parents of 16 children)
present = []
for l in (2, 3):
    Wc = W[l][:D].tocoo()
    present.append(np.unique(Wc.row.astype(np.int64) * (ks[l] // 16) + Wc.col // 16))
    log(f"level {l}: {len(present[-1])} non-empty segments of {D * (ks[l] // 16)}")

ip, ii = X.indptr, X.indices


def window_stats(qs, beam, npar, pres, nstage=BEAM):
    """touches / distinct segments of the queries qs at a level whose parents are beam[q, :nstage]"""
    cnt = np.diff(ip)[qs]
    f = np.concatenate([ii[ip[q]:ip[q + 1]] for q in qs]).astype(np.int64)
    par = np.repeat(beam[qs, :nstage], cnt, axis=0).astype(np.int64)            # [touch rows, nstage]
    keys = (f[:, None] * npar + par).ravel()
    if pres is not None:
        keys = keys[np.isin(keys, pres, assume_unique=False)]
    return len(keys), len(np.unique(keys))


def run(order_of_xcd0, label, level, pres_on=True, nstage=BEAM):
    beam, npar, pres = (b1, ks[1], present[0]) if level == 2 else (b2, ks[2], present[1])
    T = Dn = 0
    nw = min(NWIN, len(order_of_xcd0) // G)
    pick = np.linspace(0, len(order_of_xcd0) // G - 1, nw).astype(int)       # windows spread over the XCD's whole share
    for w in pick:
        t, d = window_stats(order_of_xcd0[w * G:(w + 1) * G], beam, npar, pres if pres_on else None, nstage)
        T += t; Dn += d
    log(f"level {level} {label}: touches/query {T / (nw * G):.0f}  distinct/query {Dn / (nw * G):.0f}  hit bound {1 - Dn / T:.3f}  window set {Dn / nw * 64 / 2**20:.1f} MiB")


launch = np.nonzero((np.arange(N) // 4) % 8 == 0)[0]
eighth = N // 8
for level, beam in ((2, b1), (3, b2)):
    run(launch, "launch order", level)
    o = np.argsort(beam[:, 0], kind="stable")
    run(o[:eighth], "sorted by best parent, XCD 0's eighth", level); run(o[3 * eighth:4 * eighth], "sorted by best parent, XCD 3's eighth", level)
    o = np.lexsort((beam[:, 1], beam[:, 0]))
    run(o[3 * eighth:4 * eighth], "sorted by (best, second), XCD 3's eighth", level)
    o = np.lexsort((beam[:, 2], beam[:, 1], beam[:, 0]))
    run(o[3 * eighth:4 * eighth], "sorted by (best, second, third), XCD 3's eighth", level)
    # interleaved: sorted order dealt to the XCDs in blocks of 4 (what a plain sorted launch without an XCD-aware block map gives)
    o = np.argsort(beam[:, 0], kind="stable")
    run(o[(np.arange(N) // 4) % 8 == 0], "sorted by best parent, NO xcd map", level)
    run(launch, "launch order, staged (4 parents)", level, nstage=4)
    run(np.argsort(beam[:, 0], kind="stable")[3 * eighth:4 * eighth], "sorted, staged (4 parents)", level, nstage=4)
