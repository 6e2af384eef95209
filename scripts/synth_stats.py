#!/usr/bin/env python3
"""Routing / saturation / pruning statistics of a synthetic workload (numpy beam search on a query sample; statistics only --
the parity checkers are oracle/).  Used to accept `amazon-670k-hard`: the ten most common leaf parents must
hold < 5 % of the final labels, P(margin >= 1) <= 5 % at every level.

    python scripts/synth_stats.py --config amazon-670k-hard [--scale 1.0] [--sample 2000] [--cache /tmp/xrl_bench]
"""
import argparse, json, os, sys, time
import numpy as np
import scipy.sparse as smat
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrl_synth
from oracle.xrl_oracle import load_model_folder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="amazon-670k-hard"); ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--sample", type=int, default=2000); ap.add_argument("--cache", default="/tmp/xrl_bench")
    ap.add_argument("--beam", type=int, default=0); ap.add_argument("--topk", type=int, default=10)
    a = ap.parse_args()
    folder = os.path.join(a.cache, f"{a.config}_{a.scale}")
    if not os.path.exists(os.path.join(folder, ".done")):
        os.makedirs(folder, exist_ok=True)
        t0 = time.time()
        ks, X, cfg = xrl_synth.make_config(a.config, folder, scale=a.scale)
        smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
        json.dump({"ks": ks, "cfg": cfg}, open(os.path.join(folder, "meta.json"), "w"))
        open(os.path.join(folder, ".done"), "w").write("ok")
        print(f"generated in {time.time() - t0:.0f}s", file=sys.stderr)
    X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr()
    cfg = xrl_synth.CONFIGS[a.config]
    beam = a.beam or cfg["beam"]
    layers = load_model_folder(folder)
    T = len(layers)
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(X.shape[0], size=min(a.sample, X.shape[0]), replace=False))
    D = X.shape[1]
    Ws = [L["W"].tocsc() for L in layers]
    Cs = [L["C"].tocsc() for L in layers]
    n_cand = np.zeros(T); n_sat = np.zeros(T); n_tie = np.zeros(T); parents_needed = [[] for _ in range(T)]
    final_parent = []; final_labels = []
    topics = xrl_synth.hard_query_topics(X.shape[0], cfg["x_nnz"], Cs[-1].shape[1], seed=1) if cfg.get("hard") and a.scale == 1.0 or cfg.get("hard") else None
    if topics is not None and len(topics) != X.shape[0]: topics = None
    own_in_beam = np.zeros(T); own_top1 = 0
    for r in rows:
        x = X[r]
        xd = np.zeros(D + 1, np.float64); xd[x.indices] = x.data
        prev = [(0, 1.0)]
        for l in range(T):
            W, Cm = Ws[l], Cs[l]
            xd[D] = layers[l]["bias"] if W.shape[0] == D + 1 else 0.0
            k = a.topk if l == T - 1 else beam
            cands = []; pos_parent = []
            per_parent = []
            for (p, s) in prev:
                ch = Cm.indices[Cm.indptr[p]:Cm.indptr[p + 1]]
                sub = W[:, ch]
                v = np.asarray(sub.T @ xd[: W.shape[0]]).ravel()
                sc = np.exp(-np.maximum(0.0, 1.0 - v) ** 3) * (s if l > 0 else 1.0)
                n_cand[l] += len(ch); n_sat[l] += int((v >= 1.0).sum()); n_tie[l] += int((sc == s).sum()) if l > 0 else 0
                per_parent.append(sc)
                cands += [(float(scv), int(c), p) for scv, c in zip(sc, ch)]
            # parents needed by exact bound pruning: smallest j with >= k candidates of parents < j scoring >= score of parent j
            if l > 0 and len(prev) > 1:
                need = len(prev)
                allsc = np.array([])
                for j in range(1, len(prev)):
                    allsc = np.concatenate([allsc, per_parent[j - 1]])
                    if (allsc >= prev[j][1]).sum() >= k:
                        need = j; break
                parents_needed[l].append(need)
            order = sorted(range(len(cands)), key=lambda i: (-cands[i][0], i))[:k]
            if topics is not None and l == T - 2:
                own_in_beam[l] += int(any(cands[i][1] == topics[r] for i in order))
            if topics is not None and l == T - 1:
                own_top1 += int(cands[order[0]][2] == topics[r])
            if l == T - 1:
                final_parent += [cands[i][2] for i in order]; final_labels += [cands[i][1] for i in order]
            prev = [(cands[i][1], cands[i][0]) for i in order]
    fp = np.array(final_parent)
    cnt = np.sort(np.bincount(fp))[::-1]
    out = dict(config=a.config, scale=a.scale, sample=len(rows), beam=beam, topk=a.topk,
               top1_leaf_parent_share=float(cnt[0] / cnt.sum()), top10_leaf_parent_share=float(cnt[:10].sum() / cnt.sum()),
               distinct_leaf_parents=int((cnt > 0).sum()),
               own_topic_in_last_beam=(own_in_beam[T - 2] / len(rows)) if topics is not None else None,
               top1_label_under_own_topic=(own_top1 / len(rows)) if topics is not None else None,
               per_level=[dict(level=l, candidates_per_query=n_cand[l] / len(rows), p_margin_ge_1=n_sat[l] / max(1, n_cand[l]),
                               p_child_ties_parent=n_tie[l] / max(1, n_cand[l]),
                               mean_parents_needed=float(np.mean(parents_needed[l])) if parents_needed[l] else None,
                               frac_done_after_first_parent=float(np.mean(np.array(parents_needed[l]) == 1)) if parents_needed[l] else None)
                          for l in range(T)])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
