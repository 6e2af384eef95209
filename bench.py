#!/usr/bin/env python3
"""Benchmark of the XR-Linear batch-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One STEP = one full beam search (all layers: prolongate -> sparse inner products + post-processor
+ combine -> per-query top-k) over the whole synthetic query batch of the named workload, queries
already resident in HBM, results left in HBM; with N>1 GPUs the batch is split into N contiguous
nnz-balanced row shards (strong scaling: total work fixed) and every step ends with the RCCL
all-gather of the fixed-stride top-k.  Rank 0 prints ONE JSON line.

Workload: BASELINE.json's Amazon-670K shape (the configuration its target is quoted on; it fits
one GPU): N=490,000 queries, D=135,000, L=670,091, tree [2,32,512,8192,670091], beam=10, top-k=10,
post-processor l3-hinge, synthetic CSR (xrl_synth.py, fixed seeds).

Extra objects on the JSON line:
  roofline      dominant kernel (k1_sparse, all layers): ALGORITHMIC bytes per launch (SURVEY.md 8d:
                every active reference chunk streamed once, 8E+4R+4(R+1) bytes, + the query row + the
                scores written) / average launch duration from hipEvent pairs recorded around each
                launch on its own stream during the timed steps; peak = 8 TB/s HBM.
  cpu_baseline  the REAL reference (oracle/_ref, compiled from /root/reference's own sources) timed
                on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="amazon-670k")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--beam", type=int, default=0)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--cache", default="/tmp/xrl_bench")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--k1-group", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="library option key=int (xrl_set_option), repeatable")
    args = ap.parse_args()

    import numpy as np
    import scipy.sparse as smat
    import torch  # first: pecos_amd's HIP library then shares torch's HIP runtime
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        log(f"WARNING: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import xrl_synth
    from pecos_amd import XLinearModel, clib
    from pecos_amd.distributed import shard_bounds, take_rows

    cfg = dict(xrl_synth.CONFIGS[args.config])
    beam = args.beam or cfg["beam"]
    folder = os.path.join(args.cache, f"{args.config}_{args.scale}")
    done = os.path.join(folder, ".done")
    t0 = time.time()
    if local == 0 and not os.path.exists(done):
        os.makedirs(folder, exist_ok=True)
        ks, X, cfg2 = xrl_synth.make_config(args.config, folder, scale=args.scale)
        if smat.issparse(X):
            smat.save_npz(os.path.join(folder, "X.npz"), X, compressed=False)
        else:
            np.save(os.path.join(folder, "X.npy"), X)
        json.dump({"ks": ks, "cfg": cfg2}, open(os.path.join(folder, "meta.json"), "w"))
        open(done, "w").write("ok")
    t_wait = time.time()
    while not os.path.exists(done):
        if time.time() - t_wait > 900:
            raise RuntimeError("timed out waiting for local rank 0 to generate the synthetic workload")
        time.sleep(0.5)
    meta = json.load(open(os.path.join(folder, "meta.json")))
    ks = meta["ks"]
    if os.path.exists(os.path.join(folder, "X.npz")):
        X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32); X.sort_indices()
    else:
        X = np.load(os.path.join(folder, "X.npy"))
    n_total = X.shape[0]
    if rank == 0:
        log(f"workload {args.config} scale={args.scale}: layers={ks} X={X.shape} nnz/row={getattr(X, 'nnz', X.size) / max(1, n_total):.1f} ({time.time() - t0:.1f}s)")

    clib.set_device(local)
    t0 = time.time()
    model = XLinearModel.load(folder)
    h = model.model.model_chain
    if args.k1_group:
        clib.set_option(h, "k1_group", args.k1_group)
    for kv in args.opt:
        key, val = kv.split("=")
        clib.set_option(h, key, int(val))
    if rank == 0:
        log(f"model on GPU: {clib.model_device_bytes(h) / 1e9:.2f} GB in {time.time() - t0:.1f}s")

    bounds = shard_bounds(X, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    Xs = take_rows(X, lo, hi)
    q = clib.queries_upload(h, Xs)
    k = clib.effective_topk(h, args.topk)
    rows = hi - lo
    maxr = int(np.diff(bounds).max())
    dev = torch.device("cuda", local)
    # packed result rows [idx(k) | val(k)] so that ONE all-gather moves both
    packed = torch.zeros((maxr, 2 * k), dtype=torch.int32, device=dev)
    cnt = torch.zeros((maxr,), dtype=torch.int32, device=dev)
    if world > 1:
        g_packed = torch.empty((world, maxr, 2 * k), dtype=torch.int32, device=dev)
        g_cnt = torch.empty((world, maxr), dtype=torch.int32, device=dev)
    # one explicit (non-default) torch stream carries the kernels AND the RCCL all-gathers, so that the gather of a
    # step is ordered after its predict without a host sync (handle 0 would mean "the library's own stream")
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream

    def step():
        with torch.cuda.stream(tstream):
            if rows:
                clib.predict_device(h, q, beam, None, args.topk, packed.data_ptr(), packed.data_ptr() + 4 * k, cnt.data_ptr(),
                                    2 * k, stream=stream, sync=False)
            if world > 1:
                dist.all_gather_into_tensor(g_packed, packed)
                dist.all_gather_into_tensor(g_cnt, cnt)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.synchronize()        # buffers were zero-filled on the default stream
    for _ in range(args.warmup):
        step()
    sync_all()
    clib.profile_reset(h)
    clib.profile_enable(h, True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    clib.profile_enable(h, False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = clib.profile_get(h)
    if rank == 0:
        log("per-launch ms: " + "  ".join(f"{r['name']}[{r['layer']}]={r['ms'] / max(1, r['launches']):.3f}" for r in prof))

    out = None
    if rank == 0:
        ms_per_step = dt / max(1, args.steps) * 1e3
        value = n_total * args.steps / dt
        # ---- roofline of the dominant kernel (k1, summed over layers) on this rank's shard
        st = clib.predict_stats(h, q, beam, None, args.topk) if rows else []
        xbytes = 8.0 * Xs.nnz if smat.issparse(Xs) else 4.0 * Xs.size
        alg_k1 = sum(cb + 4.0 * ne for cb, ne in st) + xbytes            # bytes per predict
        fam = {}
        for r in prof:
            f = fam.setdefault(r["name"], dict(ms=0.0, launches=0)); f["ms"] += r["ms"]; f["launches"] += r["launches"]
        dom = max(fam, key=lambda n: fam[n]["ms"]) if fam else None
        roof = None
        if dom and dom.startswith("k1"):
            launches_per_step = fam[dom]["launches"] / max(1, args.steps)
            avg_ms = fam[dom]["ms"] / max(1, fam[dom]["launches"])
            per_launch = alg_k1 / max(1.0, launches_per_step)
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            traffic, tsrc = None, None
            tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")   # written from separate rocprofv3 --pmc passes
            if os.path.exists(tfile):
                tj = json.load(open(tfile))
                if tj.get("config") == args.config and tj.get("scale") == args.scale and tj.get("n_gpus") == world:
                    traffic, tsrc = tj.get("hbm_bytes_per_launch"), tj.get("source")
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4),
                        traffic=traffic, traffic_source=tsrc, alg_bytes_per_launch=per_launch, avg_launch_ms=round(avg_ms, 4),
                        launches_per_step=launches_per_step,
                        per_kernel_ms_per_step={n: round(v["ms"] / max(1, args.steps), 4) for n, v in fam.items()},
                        per_layer=[dict(layer=r["layer"], ms=round(r["ms"] / max(1, r["launches"]), 4),
                                        ref_chunk_bytes=st[r["layer"]][0], candidates=st[r["layer"]][1])
                                   for r in prof if r["name"] == dom],
                        note="algorithmic bytes assume NO inter-query reuse (SURVEY.md 8d); frac>1 means chunks are served from L2/MALL")
        cfg_out = dict(workload=f"{args.config} synthetic x{args.scale}: N={n_total} D={X.shape[1]} L={ks[-1]} tree={ks} "
                                f"nnz/row={getattr(X, 'nnz', X.size) / max(1, n_total):.1f} beam={beam} topk={k} pp=l3-hinge bias=1.0",
                       parallelism=f"query-shard x{world}" + (" + rccl all-gather(top-k)" if world > 1 else ""),
                       model_hbm_gb=round(clib.model_device_bytes(h) / 1e9, 3))
        out = dict(metric=baseline_metric(), value=round(value, 1), unit="queries/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f32", data="synthetic", config=cfg_out, roofline=roof)

        # ---- CPU baseline + parity on a bounded sample (N=1 only)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(folder, X, model, beam, args.topk, args.cpu_seconds, log)
        print(json.dumps(out), flush=True)

    clib.queries_free(q)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def baseline_metric():
    """The metric string of BASELINE.json (the line is judged against it), with a literal fallback."""
    try:
        return json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"]
    except Exception:
        return "XLinear queries/sec @ beam=10 top-k=10; P@1 vs reference; 1/2/4/8 GPU"


def cpu_baseline(folder, X, model, beam, topk, budget_s, log):
    """Time the real reference (oracle/_ref, built from /root/reference's own sources) on this box's
    host cores on a bounded sample of the workload; fall back to the single-threaded C restatement
    ("port") when oracle/_ref is absent.  The reference's OpenMP path does not scale to hundreds of
    threads on a small batch, so the thread count is swept and the BEST is reported (cores = threads
    used).  Also compares the GPU output with the reference's on that sample."""
    import numpy as np
    from oracle import xrl_oracle as O
    ncpu = os.cpu_count() or 1
    n = X.shape[0]
    parity = None
    if O.ref_available():
        results = {}
        for wtype in ("BINARY_SEARCH_CHUNKED", "HASH_CHUNKED"):
            t0 = time.time()
            rm = O.RefModel(folder, wtype)
            load_s = time.time() - t0
            ns = min(n, 8192)
            rm.predict(X[:ns], beam_size=beam, only_topk=topk, threads=min(ncpu, 32))   # warm-up (page faults)
            best = None
            for th in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
                t0 = time.perf_counter(); rm.predict(X[:ns], beam_size=beam, only_topk=topk, threads=th); t1 = time.perf_counter() - t0
                if best is None or ns / t1 > best[0]:
                    best = (ns / t1, th)
            th = best[1]
            ns = int(min(n, max(ns, best[0] * budget_s / 2)))
            t0 = time.perf_counter(); P = rm.predict(X[:ns], beam_size=beam, only_topk=topk, threads=th); t1 = time.perf_counter() - t0
            results[wtype] = dict(qps=ns / t1, sample=ns, threads=th, load_s=round(load_s, 1))
            log(f"cpu reference {wtype}: {ns} queries in {t1:.2f}s = {ns / t1:.0f} q/s with {th} threads of {ncpu} (load {load_s:.1f}s)")
            if wtype == "BINARY_SEARCH_CHUNKED":
                G = model.predict(X[:ns], beam_size=beam, only_topk=topk)
                same_rows = np.array_equal(G.indptr, P.indptr)
                same_idx = same_rows and np.array_equal(G.indices, P.indices)
                rel = float(np.max(np.abs(G.data - P.data) / np.maximum(np.abs(P.data), 1e-30))) if same_rows and P.nnz else None
                bit = bool(same_idx and np.array_equal(G.data.view(np.uint32), P.data.view(np.uint32)))
                p1 = float(np.mean(G.indices[G.indptr[:-1]] == P.indices[P.indptr[:-1]])) if same_rows else None
                parity = dict(vs="reference BINARY_SEARCH_CHUNKED", sample=ns, indices_identical=bool(same_idx),
                              scores_bit_identical=bit, max_rel_err=rel, top1_agreement=p1)
            del rm
        bw = max(results, key=lambda w: results[w]["qps"])
        base = dict(value=round(results[bw]["qps"], 1), unit="queries/s", cores=results[bw]["threads"], kind="reference",
                    sample=f"first {results[bw]['sample']} queries of the workload, layout {bw}, best of a thread sweep "
                           f"({results[bw]['threads']} OpenMP threads on a {ncpu}-cpu host), 1 warm-up + 1 timed call; other layout: " +
                           "; ".join(f"{w}={results[w]['qps']:.0f} q/s @ {results[w]['threads']} thr" for w in results if w != bw))
    else:
        om = O.OracleModel.load(folder)
        ns = min(n, 256)
        t0 = time.perf_counter(); P = om.predict(X[:ns], beam_size=beam, only_topk=topk); t1 = time.perf_counter() - t0
        base = dict(value=round(ns / t1, 1), unit="queries/s", cores=1, kind="port",
                    sample=f"first {ns} queries, single-threaded C restatement (oracle/_ref absent)")
    return base, parity


if __name__ == "__main__":
    main()
