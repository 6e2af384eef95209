"""CPU-side estimate for DESIGN.md section 10.1(c): how much of K1Q's L2 traffic on Amazon-670K could a different QUERY ORDER keep in
one XCD's L2?  (No GPU, no oracle: numpy beam search over the upper levels of the synthetic model, then an LRU model of one XCD.)

Model of the kernel: one wavefront per query, queries go to XCDs in blocks of 4 (block b -> XCD b % 8), G queries of an XCD run
concurrently and walk their features in step; per feature a query touches, in the dense row format,
  level 0 + 1: the line holding the 2 + the line holding the 32 columns of the feature's rows,
  level 2 / 3: the 64-byte segments of its 4 best beam parents (stage 0 of the bound pruning) in the feature's row
               (two sibling parents share a 128-byte line).
L2 of one XCD: 4 MiB, 128-byte lines, LRU.  The measured hit rate of the real kernel (TCC_HIT / TCP_TCC_READ_REQ = 46 %,
profiles/r03_pmc_l2.csv) calibrates the model; the variants then say what sorting the queries by their best level-3 parent between
levels 2 and 3 could buy.   usage: python scripts/locality_sim.py [cache folder] [queries per XCD to simulate]"""
import collections
import os
import sys
import time

import numpy as np
import scipy.sparse as smat

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import xrl_synth  # noqa: E402

folder = sys.argv[1] if len(sys.argv) > 1 else "/tmp/xrl_cpu/amazon"
n_sim = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
BEAM, STAGE0, G, LINES = 10, 4, 896, (4 << 20) // 128


def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


if not os.path.exists(os.path.join(folder, "X.npz")):
    log("generating the amazon-670k synthetic workload (a few minutes) ...")
    ks, X, cfg = xrl_synth.make_config("amazon-670k", folder)
    smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32)
N, D = X.shape
W = [smat.load_npz(os.path.join(folder, "ranker", f"{d}.model", "W.npz")).tocsc() for d in range(3)]
ks = [w.shape[1] for w in W]
log("levels", ks, "queries", N)

# the first 8 * n_sim queries; XCD 0's share of them in launch order comes FIRST in `sel` (rows [0, n_sim) of everything below)
blk = np.arange(min(N, 8 * n_sim)) // 4
mine = np.nonzero(blk % 8 == 0)[0]
others = np.nonzero(blk % 8 != 0)[0]
sel = np.concatenate([mine, others])
n_mine = len(mine)
Xs = X[sel]


def hinge3(z):
    return np.exp(-np.maximum(1.0 - z, 0.0) ** 3)


def scores(l):
    out = np.empty((Xs.shape[0], ks[l]), np.float32)
    Wl, bl = W[l][:D].tocsr(), W[l][D].toarray().astype(np.float32)
    for r0 in range(0, Xs.shape[0], 32768):                       # chunks: the level-2 result of all rows at once would be 2 GB in f64
        out[r0:r0 + 32768] = hinge3((Xs[r0:r0 + 32768] @ Wl).toarray() + bl)
    return out


log("beam search over levels 0-2 on", len(sel), "queries ...")
s0 = scores(0)                                                   # [n, 2]: the root keeps both
s1 = scores(1) * np.repeat(s0, ks[1] // ks[0], axis=1)           # [n, 32]: all children of both
b1 = np.argsort(-s1, axis=1, kind="stable")[:, :BEAM]            # level-1 beam, best first
s1b = np.take_along_axis(s1, b1, axis=1)
t2 = scores(2)                                                   # [n, 512]
ch = ks[2] // ks[1]
cand = (b1[:, :, None] * ch + np.arange(ch)[None, None, :]).reshape(len(sel), -1)         # 160 candidates in position order
s2 = np.take_along_axis(t2, cand, axis=1) * np.repeat(s1b, ch, axis=1)
o2 = np.argsort(-s2, axis=1, kind="stable")[:, :BEAM]
b2 = np.take_along_axis(cand, o2, axis=1)                        # level-2 beam (= the parents of level 3), best first
log("share of queries whose best level-3 parent is among the 10 most common:", round(float(np.isin(b2[:, 0], np.argsort(-np.bincount(b2[:, 0], minlength=ks[2]))[:10]).mean()), 3))

ip, ii = Xs.indptr, Xs.indices


def simulate(order3, label):
    """levels 0-2 in launch order, level 3 in `order3` (a permutation of the simulated queries); G queries in flight"""
    lru = collections.OrderedDict()
    hit = collections.Counter(); tot = collections.Counter()

    def touch(key, lvl):
        tot[lvl] += 1
        if key in lru:
            lru.move_to_end(key); hit[lvl] += 1
        else:
            lru[key] = None
            if len(lru) > LINES:
                lru.popitem(last=False)

    def run(qs, levels):
        for g0 in range(0, len(qs), G):
            grp = qs[g0:g0 + G]
            feats = [ii[ip[q]:ip[q + 1]] for q in grp]
            for lv in levels:
                for t in range(max(len(f) for f in feats)):
                    for q, f in zip(grp, feats):
                        if t >= len(f):
                            continue
                        ft = int(f[t])
                        if lv == 1:
                            touch((0, ft), "L0"); touch((1, ft), "L1")
                        elif lv == 2:
                            for p in {int(x) >> 1 for x in b1[q, :STAGE0]}:
                                touch((2, ft, p), "L2")
                        else:
                            for p in {int(x) >> 1 for x in b2[q, :STAGE0]}:
                                touch((3, ft, p), "L3")
    n = n_mine
    if order3 is None:
        run(list(range(n)), (1, 2, 3))                           # the fused kernel: a wavefront carries its query through all levels
    else:
        run(list(range(n)), (1, 2))
        run(list(order3), (3,))
    T, H = sum(tot.values()), sum(hit.values())
    log(f"{label}: L2 hit rate {H / T:.3f} overall;", "  ".join(f"{k} {hit[k] / tot[k]:.3f} ({tot[k] / n:.0f} lines/query)" for k in sorted(tot)),
        f"; misses per query {(T - H) / n:.0f}")


simulate(None, "fused, launch order")
simulate(np.arange(n_mine), "levels 0-2 | level 3, launch order")
simulate(np.argsort(b2[:n_mine, 0], kind="stable"), "level 3: this XCD's queries sorted by best parent")
simulate(np.lexsort((b2[:n_mine, 1], b2[:n_mine, 0])), "level 3: this XCD's queries sorted by (best, second) parent")
# XCD-affine: ALL queries sorted by best parent, every XCD takes a contiguous eighth of that order (so an XCD sees an eighth of the parents)
order_all = np.argsort(b2[:, 0], kind="stable")
for x in (0, 3, 7):
    simulate(order_all[x * n_mine:(x + 1) * n_mine], f"level 3: all queries sorted by best parent, XCD {x} takes its contiguous eighth")
