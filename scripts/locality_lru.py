import os, sys, time, collections
import numpy as np, scipy.sparse as smat
folder = sys.argv[1]; NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
G, BEAM = 1024, 10
def log(*a): print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)
X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32)
N, D = X.shape
W = [smat.load_npz(os.path.join(folder, "ranker", f"{d}.model", "W.npz")).tocsc() for d in range(4)]
ks = [w.shape[1] for w in W]
def hinge3(z): return np.exp(-np.maximum(1.0 - z, 0.0) ** 3)
def scores(l, rows):
    Wl, bl = W[l][:D].tocsr(), W[l][D].toarray().astype(np.float32)
    return hinge3((X[rows] @ Wl).toarray() + bl).astype(np.float32)
cache = "/tmp/sim/beams_" + os.path.basename(folder.rstrip("/")) + ".npz"
if os.path.exists(cache):
    z = np.load(cache); b1, b2 = z["b1"], z["b2"]
else:
    b1 = np.empty((N, BEAM), np.int32); b2 = np.empty((N, BEAM), np.int32)
    for r0 in range(0, N, 65536):
        rows = slice(r0, min(N, r0 + 65536)); n = rows.stop - rows.start
        s0 = scores(0, rows); s1 = scores(1, rows) * np.repeat(s0, ks[1] // ks[0], axis=1)
        o1 = np.argsort(-s1, axis=1, kind="stable")[:, :BEAM]; s1b = np.take_along_axis(s1, o1, axis=1)
        ch = ks[2] // ks[1]
        cand = (o1[:, :, None] * ch + np.arange(ch)[None, None, :]).reshape(n, -1)
        s2 = np.take_along_axis(scores(2, rows), cand, axis=1) * np.repeat(s1b, ch, axis=1)
        o2 = np.argsort(-s2, axis=1, kind="stable")[:, :BEAM]
        b1[rows] = o1; b2[rows] = np.take_along_axis(cand, o2, axis=1)
    np.savez(cache, b1=b1, b2=b2)
log("beams ready")
Wc = W[3][:D].tocoo()
present = set((Wc.row.astype(np.int64) * 512 + Wc.col // 16).tolist())
log("presence set", len(present))
ip, ii = X.indptr, X.indices
LINES = (4 << 20) // 128

def simulate(order, label, nstage=BEAM, use_pres=True, stagger=True):
    lru = collections.OrderedDict()   # line key -> sector mask
    hits = miss = ptouch = pmiss = 0
    order = list(order[:NQ]); nxt = 0
    slots = []
    for s in range(min(G, len(order))):
        q = order[nxt]; nxt += 1
        slots.append([q, ip[q] + (0 if not stagger else 0), ip[q + 1]])
    active = len(slots)
    while active:
        for s in slots:
            if s is None: continue
        for si in range(len(slots)):
            s = slots[si]
            if s is None: continue
            q, pos, end = s
            if pos >= end:
                if nxt < len(order):
                    q = order[nxt]; nxt += 1; slots[si] = [q, ip[q], ip[q + 1]]
                else:
                    slots[si] = None; active -= 1
                continue
            f = int(ii[pos]); s[1] = pos + 1
            if use_pres:
                k = (-1, f); ptouch += 1
                if k in lru: lru.move_to_end(k)
                else:
                    pmiss += 1; lru[k] = 3
                    if len(lru) > LINES: lru.popitem(last=False)
            for p in b2[q, :nstage]:
                p = int(p)
                if use_pres and (f * 512 + p) not in present: continue
                k = (f, p >> 1); bit = 1 << (p & 1)
                m = lru.get(k)
                if m is None:
                    miss += 1; lru[k] = bit
                    if len(lru) > LINES: lru.popitem(last=False)
                else:
                    lru.move_to_end(k)
                    if m & bit: hits += 1
                    else: miss += 1; lru[k] = m | bit
    n = len(order)
    log(f"{label}: weight touches/q {(hits + miss) / n:.0f} misses/q {miss / n:.0f} hit {hits / max(1, hits + miss):.3f}; presence misses/q {pmiss / n:.1f} of {ptouch / n:.0f}")

launch = np.nonzero((np.arange(N) // 4) % 8 == 0)[0]
eighth = N // 8
o = np.argsort(b2[:, 0], kind="stable")
simulate(launch, "launch order")
simulate(o[3 * eighth:4 * eighth], "sorted by best parent, xcd-contiguous")
simulate(o[(np.arange(N) // 4) % 8 == 0], "sorted by best parent, no xcd map")
o2 = np.lexsort((b2[:, 1], b2[:, 0]))
simulate(o2[3 * eighth:4 * eighth], "sorted by (best, second), xcd-contiguous")
simulate(launch, "launch order staged4", nstage=4, use_pres=False)
simulate(o[3 * eighth:4 * eighth], "sorted staged4", nstage=4, use_pres=False)
