#!/usr/bin/env python3
"""Benchmark of the XR-Linear batch-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One STEP = one full beam search (all layers: prolongate -> inner products + post-processor + combine -> per-query
top-k) over the whole synthetic query batch of the named workload, queries already resident in HBM, results left in
HBM; with N>1 GPUs the batch is split into N contiguous nnz-balanced row shards (strong scaling: total work fixed)
and every step ends with ONE RCCL all-gather of the packed fixed-stride top-k rows.  Rank 0 prints ONE JSON line.

Workload: BASELINE.json's Amazon-670K shape (the configuration its target is quoted on; it fits one GPU): N=490,000
queries, D=135,000, L=670,091, tree [2,32,512,8192,670091], beam=10, top-k=10, post-processor l3-hinge, synthetic CSR
(xrl_synth.py, fixed seeds).  --config eurlex-4k | wiki10-31k | dense-768 (with --scale / --rows) run the other shapes.

On the JSON line:
  value           queries/s with X resident in HBM when the timed region starts (the contract's `value`)
  value_host_abi  queries/s through the drop-in entry point c_xlinear_predict_csr_f32 -- pageable host X in, H2D,
                  kernels, D2H, allocator callback, host CSR out -- the figure SURVEY.md 8(d) specifies (N=1 only)
  roofline        the kernel family with the most GPU time (hipEvent pairs around every launch on the stream it runs on, during the
                  timed steps).  Three fractions of the 8 TB/s HBM peak, side by side, because they answer different questions:
                    frac            COUNTER bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, separate --pmc passes over this same
                                    command, committed as profiles/pmc_traffic.json; fabric requests x 64 B, Infinity-Cache hits
                                    included) / launch time.  This is the HBM-roofline fraction.  If no counter file matches the
                                    configuration the line says so (`basis`) and falls back to frac_matched.
                    frac_matched    MATCHED-WORK bytes / launch time: the bytes the algorithm addresses for the rows a query actually
                                    matches, no inter-query reuse assumed (tile format: per (query, tile) item 8*nnz_x + probe bytes
                                    + 4*hit_rows + 8*hit_entries + 4*ncols; dense format: per query 8*nnz_x + 4 * sum over (feature,
                                    candidate column) + 16*beam), counted by an untimed stats pass.  Contains cache hits: can exceed
                                    what HBM streams.
                    frac_ref_layout SURVEY.md 8(d)'s own figure (every active REFERENCE chunk streamed whole) / step time.  It is >> 1:
                                    the kernels look rows up instead of streaming chunks, so this model does not describe them.
                  `requests` is the roof the memory side actually sits on (fabric read requests/s against the ceiling calibrated in
                  profiles/r02_calib_fetch.md), `issue` the VALU-issue occupancy from the SQ counters (1 wavefront instruction per 4
                  cycles per SIMD) -- on this workload the binding limit (profiles/r03_k1r_experiments.txt).  A layer structure that
                  FITS the 288 MB of on-chip cache is reported against the L2 peak in `l2`.  `kernels` lists every launch family.
  cpu_baseline    the REAL reference (oracle/_ref, compiled from /root/reference's own sources) on this box's host cores,
                  bounded sample of the same workload, 1 warm-up + median of 5 calls (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBPS = 34500.0          # MI355X_MICROARCH.md: ~34.5 TB/s aggregate L2
ONCHIP_CACHE_BYTES = 256e6 + 32e6  # MI355X_MICROARCH.md: 256 MB Infinity Cache + 8 x 4 MB L2
FABRIC_REQ_CEILING_G = 57.0     # profiles/r02_calib_fetch.md: fabric read requests/s sustained by narrow gathers (51 G/s HBM .. 59 G/s Infinity Cache)
SHADER_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md
N_CU, N_SIMD = 256, 1024


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


# the sources a kernel launch of the bench path is built from: a counter set is valid for exactly these (host-only files -- the C ABI glue, the
# folder readers, the TF-IDF producer's host half -- cannot change a counter, so editing them does not mark the sets stale)
_KERNEL_PATH = ("xrl_device.h", "xrl_kernels.h", "xrl_k1q_impl.h", "xrl_items.h", "xrl_common.h", "xrl_predict.cpp", "xrl_predict.h", "xrl_model.cpp", "xrl_model.h", "Makefile")


def csrc_sha16():
    """First 16 hex digits of the SHA-256 over the KERNEL-PATH sources (file names and contents, sorted: pecos_amd/csrc/*.hip + _KERNEL_PATH):
    profiles/pmc_traffic.json records it with every counter set, and a line only quotes counters taken on the SAME sources."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(REPO, "pecos_amd", "csrc")
    for name in sorted(os.listdir(root)):
        path = os.path.join(root, name)
        if os.path.isfile(path) and (name.endswith(".hip") or name in _KERNEL_PATH):
            h.update(name.encode()); h.update(b"\0"); h.update(open(path, "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


def pmc_key(config, scale, opts):
    """Key of a counter set in profiles/pmc_traffic.json: workload, scale and the library options the run was made with."""
    return f"{config}@{scale}" + ("+" + ",".join(sorted(o.replace(" ", "") for o in opts)) if opts else "")


def pmc_entry(config, scale, opts, world):
    """(entry, why-not) -- the counter set recorded for exactly this configuration on exactly these sources, or None and the reason."""
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(tfile):
        return None, "no profiles/pmc_traffic.json"
    ent = json.load(open(tfile)).get("entries", {}).get(pmc_key(config, scale, opts))
    if ent is None:
        return None, f"no counter set recorded for {pmc_key(config, scale, opts)}"
    if ent.get("n_gpus") != world:
        return None, f"counter set recorded at n_gpus={ent.get('n_gpus')}"
    if ent.get("csrc_sha16") != csrc_sha16():
        return None, f"counter set is STALE: taken on sources {ent.get('csrc_sha16')}, this tree is {csrc_sha16()}"
    return ent, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="amazon-670k")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rows", type=int, default=0, help="use only the first ROWS queries of the workload (dense-768)")
    ap.add_argument("--beam", type=int, default=0)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--cache", default="/tmp/xrl_bench")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-abi", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="skip the untimed matched-work stats pass (PMC collection runs: only the timed kernels launch)")
    ap.add_argument("--host-steps", type=int, default=5)
    ap.add_argument("--include-upload", action="store_true",
                    help="time the drop-in entry point instead: every rank calls c_xlinear_predict_* on ITS shard (pageable host X in, H2D, kernels, "
                         "D2H, host CSR out); value = all queries / max-over-ranks time (what a caller that shards by process gets, PCIe included)")
    ap.add_argument("--parity-rows", type=int, default=-1, help="rows per shard of the TIMED output compared with the reference after the timed loop "
                                                               "(0 = skip; -1, the default = EVERY row when oracle/_ref is built, else 4096)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extra", action="store_true", help="skip the extra lines of the default run (extra.hard: the same shape on the model that does not flatter "
                                                             "bound pruning; extra.text_to_labels: texts -> labels with X never on the host)")
    ap.add_argument("--text-docs", type=int, default=100000, help="documents of the text -> labels line")
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
    # N > 1 REHEARSAL on a one-GPU box (tests): XRL_BENCH_ONE_DEVICE=1 puts every rank on device 0 and XRL_BENCH_BACKEND=gloo exchanges the
    # packed rows through host-staged tensors (RCCL refuses two ranks on one device) -- shard bounds, the double-buffered gather pipeline,
    # max-over-ranks timing, timed-output parity on rows of every shard and rank 0's JSON line all run exactly as with N GPUs
    backend = os.environ.get("XRL_BENCH_BACKEND", "nccl")
    one_device = bool(os.environ.get("XRL_BENCH_ONE_DEVICE"))
    gen_rank0 = local == 0                                   # who generates the synthetic workload on this node
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get("XRL_BENCH_FORCE_DIST"):   # the latter: exercise RCCL init + the gather with one rank (tests)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import xrl_synth
    from pecos_amd import XLinearModel, clib
    from pecos_amd.core import ScipyCompressedSparseAllocator, ScipyCsrF32, ScipyDrmF32
    from pecos_amd.distributed import GatherPipeline, shard_bounds, take_rows

    cfg = dict(xrl_synth.CONFIGS[args.config])
    beam = args.beam or cfg["beam"]
    folder = os.path.join(args.cache, f"{args.config}_{args.scale}")
    done = os.path.join(folder, ".done")
    t0 = time.time()
    if gen_rank0 and not os.path.exists(done):
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
        if time.time() - t_wait > 1500:
            raise RuntimeError("timed out waiting for local rank 0 to generate the synthetic workload")
        time.sleep(0.5)
    meta = json.load(open(os.path.join(folder, "meta.json")))
    ks = meta["ks"]
    if os.path.exists(os.path.join(folder, "X.npz")):
        X = smat.load_npz(os.path.join(folder, "X.npz")).tocsr().astype(np.float32); X.sort_indices()
    else:
        X = np.load(os.path.join(folder, "X.npy"), mmap_mode="r")
    if args.rows and args.rows < X.shape[0]:
        X = X[: args.rows]
    if not smat.issparse(X):
        X = np.ascontiguousarray(X)
    n_total = X.shape[0]
    sparse = smat.issparse(X)
    if args.parity_rows < 0:
        # every row of the timed output against the compiled reference (~10 s of host time for 490 000 Amazon-shape rows); the
        # single-threaded restatement and the dense-768 reference path (1 M multiply-adds per query) get a bounded sample
        ref_built = os.path.exists(os.path.join(REPO, "oracle", "_ref", "libpecos_float32.so"))   # (the checker itself is only loaded after the timed region)
        args.parity_rows = n_total if (ref_built and sparse) else 4096
    nnz_row = (X.nnz if sparse else X.size) / max(1, n_total)
    if rank == 0:
        log(f"workload {args.config} scale={args.scale}: layers={ks} X={X.shape} nnz/row={nnz_row:.1f} ({time.time() - t0:.1f}s)")

    clib.set_device(local)
    t0 = time.time()
    model = XLinearModel.load(folder)
    h = model.model.model_chain
    if args.k1_group:
        clib.set_option(h, "k1_group", args.k1_group)
    for kv in args.opt:
        key, val = kv.split("=")
        clib.set_option(h, key, int(val))
    depth = len(ks)
    linfo = [clib.layer_info(h, l) for l in range(depth)]
    if rank == 0:
        log(f"model on GPU: {clib.model_device_bytes(h) / 1e9:.2f} GB in {time.time() - t0:.1f}s; dense row format on layers "
            f"{[l for l in range(depth) if linfo[l]['dense']]}")

    # every rank keeps only ITS rows of X
    bounds = shard_bounds(X, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    Xs = take_rows(X, lo, hi)
    # rows of the TIMED output that rank 0 compares with the reference afterwards: the first --parity-rows rows of EVERY shard
    par_sel = np.concatenate([np.arange(int(bounds[r]), min(int(bounds[r + 1]), int(bounds[r]) + max(0, args.parity_rows))) for r in range(world)]).astype(np.int64)
    Xpar = (X[par_sel] if sparse else np.ascontiguousarray(X[par_sel])) if (rank == 0 and len(par_sel)) else None
    if world > 1:
        del X
    q = clib.queries_upload(h, Xs)
    k = clib.effective_topk(h, args.topk)
    rows = hi - lo
    dev = torch.device("cuda", local)
    # one explicit (non-default) torch stream carries the kernels AND the RCCL all-gather, so that the gather of a step is
    # ordered after its predict without a host sync (handle 0 would mean "the library's own stream")
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    use_dist = dist.is_initialized()
    # packed rows [idx(k) | val(k) | cnt].  With a process group the all-gather of step s runs on a second stream under the kernels
    # of step s+1 (two result buffers take turns): a rank's launches are latency-bound at a shard's size (61 k rows take 0.96 ms,
    # two halves of it 2 x 0.70 ms -- profiles/r03_pruning_topk.md section 3), so the shard is NOT cut to hide the gather inside its own step.
    # Every step's gathered result is complete before the closing barrier; the last step's is what the parity check reads.
    parts = 1
    cstream = torch.cuda.Stream(device=dev) if use_dist else None
    with torch.cuda.stream(tstream):
        pipe = GatherPipeline(bounds, rank, k, dev, n_buf=2 if use_dist else 1, compute_stream=tstream if use_dist else None,
                              gather_stream=cstream, gather=use_dist)

    view = (ScipyCsrF32.init_from(Xs) if sparse else ScipyDrmF32.init_from(Xs)) if args.include_upload else None
    last_host = [None]

    def step_upload():
        # the drop-in entry point on this rank's shard: pageable host X in, H2D, kernels, D2H, allocator callback, host CSR out
        alloc = ScipyCompressedSparseAllocator()
        if rows:
            clib.xlinear_predict(h, view, beam, None, args.topk, -1, alloc)
            last_host[0] = alloc

    def step_resident():
        with torch.cuda.stream(tstream):
            cur = pipe.begin()                             # (waits for the gather that last read this buffer: step s - 2)
            b, e = cur.rows(0)
            p_idx, p_val, p_cnt, p_stride = cur.pointers(0)
            if e > b:
                clib.predict_device_rows(h, q, beam, None, args.topk, p_idx, p_val, p_cnt, p_stride, b, e - b, stream=stream, sync=False)
            pipe.end()                                     # the step's all-gather: on the second stream, under the next step's kernels

    step = step_upload if args.include_upload else step_resident

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.synchronize()
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
    if use_dist and world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = clib.profile_get(h)

    # ---- what the TIMED steps produced, compared with the reference on rows of every shard (rank 0)
    timed_parity = None
    if len(par_sel):
        n_sel = [int(min(int(bounds[r + 1]), int(bounds[r]) + max(0, args.parity_rows)) - int(bounds[r])) for r in range(world)]
        if args.include_upload:
            mine = last_host[0].get()[: n_sel[rank]] if (last_host[0] is not None and n_sel[rank]) else None
            pieces = [mine]
            if use_dist and world > 1:
                pieces = [None] * world if rank == 0 else None
                dist.gather_object(mine, pieces, dst=0)
            if rank == 0:
                G = smat.vstack([pc for pc in pieces if pc is not None], format="csr")
        else:
            pk = pipe.last()                               # what the LAST timed step wrote (and gathered)
            if not use_dist:
                with torch.cuda.stream(tstream):
                    for p in range(parts):
                        pk.gather(p)                       # no process group: copies the send buffers into the receive side
            torch.cuda.synchronize()
            if rank == 0:
                gi, gv, gc = pk.unpack()
                from pecos_amd.distributed import rows_to_csr
                sel_t = torch.from_numpy(par_sel).to(gi.device)
                G = rows_to_csr(gi[sel_t].cpu().numpy().view(np.uint32), gv[sel_t].cpu().numpy(), gc[sel_t].cpu().numpy(), model.nr_pred_cols)
        if rank == 0:
            timed_parity = timed_output_parity(folder, Xpar, G, beam, args.topk, world, args.parity_rows, log)

    out = None
    if rank == 0:
        ms_per_step = dt / max(1, args.steps) * 1e3
        value = n_total * args.steps / dt
        log("per-launch ms: " + "  ".join(f"{r['name']}[{r['layer']}]={r['ms'] / max(1, r['launches']):.3f}" for r in prof))
        roof = roofline(clib, h, q, Xs, prof, linfo, beam, args, k, rows, world, ms_per_step)
        cfg_out = dict(workload=f"{args.config} synthetic x{args.scale}: N={n_total} D={Xs.shape[1]} L={ks[-1]} tree={ks} "
                                f"nnz/row={nnz_row:.1f} beam={beam} topk={k} pp=l3-hinge bias=1.0",
                       parallelism=f"query-shard x{world}" + (f" + {'rccl' if backend == 'nccl' else backend + ' (host-staged REHEARSAL, all ranks on one device)'} all-gather of the packed top-k rows of step s on a second stream under the kernels of step s+1 (two result buffers)" if world > 1 else ""),
                       model_hbm_gb=round(clib.model_device_bytes(h) / 1e9, 3),
                       dense_format_layers=[l for l in range(depth) if linfo[l]["dense"]])
        out = dict(metric=baseline_metric(), value=round(value, 1), unit="queries/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f32", data="synthetic", config=cfg_out, roofline=roof,
                   value_definition=("PCIe-INCLUSIVE: every rank times the drop-in entry point c_xlinear_predict_* on its shard (pageable host X in, H2D, kernels, "
                                     "D2H, allocator callback, host CSR out) -- SURVEY.md 8(d)'s metric" if args.include_upload else
                                     "DEVICE-RESIDENT: queries/s with X already in HBM when the timed region starts and the top-k left in HBM (the bench contract's "
                                     "`value`); SURVEY.md 8(d)'s metric -- timed around the C-ABI call incl. H2D of X and D2H of the results -- is `value_host_abi`"))

        if args.include_upload:
            cfg_out["mode"] = "include-upload: every rank times c_xlinear_predict_* on its shard (pageable host X in, H2D, kernels, D2H, host CSR out); no gather"
        # ---- the SURVEY 8(d) figure: the drop-in C-ABI entry point, pageable host X in, host CSR out (N=1)
        if world == 1 and not args.no_host_abi and not args.include_upload:
            view = ScipyCsrF32.init_from(Xs) if sparse else ScipyDrmF32.init_from(Xs)
            times = []
            first_call = None
            for it in range(1 + max(1, args.host_steps)):
                alloc = ScipyCompressedSparseAllocator()
                t0 = time.perf_counter()
                clib.xlinear_predict(h, view, beam, None, args.topk, -1, alloc)
                t1 = time.perf_counter() - t0
                if it:
                    times.append(t1)
                else:
                    first_call = t1
            med = float(np.median(times))
            out["value_host_abi"] = round(n_total / med, 1)
            out["host_abi"] = dict(entry="c_xlinear_predict_csr_f32" if sparse else "c_xlinear_predict_drm_f32", ms_per_call=round(med * 1e3, 3),
                                   calls=len(times), ms_calls=[round(t * 1e3, 2) for t in times],
                                   first_call_ms=round(first_call * 1e3, 2), first_call_note="this process's first host-ABI call (untimed warm-up call of the loop): the device-"
                                   "resident loop has run before it, so kernels are loaded; X's device arrays and the pinned result buffers are allocated inside it. "
                                   "A fresh process's first call: profiles/r06_host_abi.md", includes="H2D of X from pageable host memory (pinned staging, row batches pipelined with compute), "
                                   "kernels, D2H, allocator callback, host CSR assembly", x_bytes=int(Xs.nnz * 8 + (rows + 1) * 8) if sparse else int(Xs.size * 4),
                                   ratio_to_device_resident=round(n_total / med / value, 3))
            log(f"host ABI: {med * 1e3:.2f} ms per call = {n_total / med / 1e6:.2f} M q/s ({n_total / med / value:.2f} x device-resident)")

        # ---- what keeping the featurizer's output on the device buys.  The queries' values stand in for term counts:
        #      the reference's tf-idf weighting + l2 normalisation runs as a kernel on the device CSR (xrl_queries_tfidf_device), and
        #      the beam search reads its output in place -- against the host ABI, which receives X from the host.
        if world == 1 and sparse and not args.include_upload and not args.no_host_abi:
            try:
                tx = torch.from_numpy(Xs.indptr.astype(np.int64)).to(dev), torch.from_numpy(Xs.indices.astype(np.int32)).to(dev), \
                    torch.from_numpy(np.abs(Xs.data).astype(np.float32)).to(dev)
                idf_t = torch.ones(Xs.shape[1], dtype=torch.float32, device=dev)
                w_out = torch.empty(int(Xs.nnz), dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                ts = []
                for it in range(4):
                    t0 = time.perf_counter()
                    qh = clib.queries_tfidf_device(h, rows, Xs.shape[1], tx[0].data_ptr(), tx[1].data_ptr(), tx[2].data_ptr(), int(Xs.nnz), idf_t.data_ptr(), False, False, 2, out_addr=w_out.data_ptr())
                    t1 = time.perf_counter() - t0
                    clib.queries_free(qh)
                    if it:
                        ts.append(t1)
                tf_ms = float(np.median(ts)) * 1e3
                out["device_featurizer"] = dict(tfidf_weight_ms=round(tf_ms, 3), queries_per_s=round(n_total / ((tf_ms + ms_per_step) * 1e-3), 1),
                                                vs_host_abi=round(out.get("value_host_abi", 0) and (n_total / ((tf_ms + ms_per_step) * 1e-3)) / out["value_host_abi"], 3),
                                                note="tf-idf weighting + l2 norm of the whole batch on the device (term counts in HBM -> X in HBM, host-timed call incl. its "
                                                     "sync) + one beam-search step on device-resident X, against value_host_abi (X arrives from pageable host memory)")
                log(f"device featurizer: tf-idf weighting {tf_ms:.3f} ms; weighting + step = {n_total / ((tf_ms + ms_per_step) * 1e-3) / 1e6:.2f} M q/s")
            except Exception as e:   # never let the extra line break the bench
                log(f"device featurizer line skipped: {e}")

        # ---- CPU baseline + parity on a bounded sample (N=1 only)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(folder, Xs, model, beam, args.topk, args.cpu_seconds, log)
        if timed_parity is not None:
            out["parity"] = dict(out.get("parity") or {}, **timed_parity)
        # ---- extra lines of the DEFAULT run: the honest model and the text -> labels pipeline, under the driver's eyes
        if world == 1 and not args.no_extra and not args.include_upload and args.config == "amazon-670k" and args.scale == 1.0 and not args.rows and not args.opt:
            out["extra"] = {}
            try:
                out["extra"]["text_to_labels"] = text_to_labels(clib, model, folder, Xs, beam, args.topk, args.text_docs, log)
            except Exception as e:   # never let an extra line break the bench
                log(f"extra.text_to_labels skipped: {e!r}")
            try:
                out["extra"]["shard8"] = shard8_line(clib, h, q, torch, dev, stream, tstream, beam, args, k, n_total, ms_per_step, log)
            except Exception as e:
                log(f"extra.shard8 skipped: {e!r}")
            clib.queries_free(q); q = None
            try:
                out["extra"]["hard"] = hard_line(args, log)
            except Exception as e:
                log(f"extra.hard skipped: {e!r}")
        print(json.dumps(out), flush=True)

    if q is not None:
        clib.queries_free(q)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def roofline(clib, h, q, Xs, prof, linfo, beam, args, k, rows, world, ms_per_step):
    import scipy.sparse as smat
    sparse = smat.issparse(Xs)
    if args.include_upload:   # the entry point cuts a call into row batches: a family's launches of one step count as ONE launch (time summed over the step)
        prof = [dict(r, launches=min(r["launches"], max(1, args.steps))) for r in prof]
    st = clib.predict_stats(h, q, beam, None, args.topk) if (rows and not args.no_stats) else []
    nnz = float(Xs.nnz) if sparse else float(Xs.size)
    x_bytes_q = 8.0 * nnz if sparse else 4.0 * nnz               # the query rows read once

    def matched_bytes(name, layer):
        s, li = st[layer], linfo[layer]
        if name.endswith("_rest") or name.endswith("_mid") or name.startswith("k0b"):     # second phase of a bound-pruned layer: its (few) items are counted with the first phase's kernel
            return 0.0
        if name.startswith("k1q_fused"):                          # several dense-format layers in one launch: "k1q_fused[_x]_<first>_<last>"
            l0, l1 = (int(v) for v in name.split("_")[-2:])
            return x_bytes_q + sum(4.0 * st[ll]["x_cols"] for ll in range(l0, l1 + 1)) + 16.0 * k * rows
        if name.startswith("k1q"):                                # dense row format, query-stationary
            return x_bytes_q + 4.0 * s["x_cols"] + 16.0 * k * rows
        if name.startswith("k1g"):                                # dense X, tiled SGEMM: per item the query row + the parent's weight panel + scores
            return 4.0 * s["probes"] + 4.0 * s["x_cols"] + 4.0 * s["item_cols"]
        if name.startswith("k1"):                                 # tile format: per item x row + lookups + extents + matched entries + scores
            probe = {0: 8.0, 1: 8.0 + 4.0 * li["bucket_levels"], 2: 16.0}[li["lookup"]]
            if not sparse:                                        # dense X walks every tile row: row id + x value instead of lookups
                return 8.0 * s["hit_rows"] + 4.0 * s["hit_rows"] + 8.0 * s["hit_entries"] + 4.0 * s["item_cols"]
            return 8.0 * s["probes"] + probe * s["probes"] + 4.0 * s["hit_rows"] + 8.0 * s["hit_entries"] + 4.0 * s["item_cols"]
        if name.startswith("k2"):
            return 4.0 * s["candidates"] + 8.0 * k * rows
        if name.startswith("k0"):
            return 32.0 * s["items"] + 8.0 * rows
        return 0.0

    def layers_of(name, layer):
        if name.startswith("k1q_fused"):
            l0, l1 = (int(v) for v in name.split("_")[-2:])
            return list(range(l0, l1 + 1))
        return [layer]

    def hbm_bytes(name, layer):
        """(HBM-level algorithmic bytes, matched-work bytes, structure footprint): matched work unless every layer structure the
        launch gathers from fits the on-chip cache -- then each byte once (structure + queries + beams + scores)."""
        mb = matched_bytes(name, layer)
        if not (name.startswith("k1") and not name.startswith("k1_sort")):
            return mb, mb, 0.0
        ls = layers_of(name, layer)
        dense_fmt = name.startswith("k1q") or name.startswith("k1g")
        foot = float(sum(linfo[ll]["dense_bytes"] if dense_fmt else linfo[ll]["device_bytes"] - linfo[ll]["dense_bytes"] for ll in ls))
        if foot > ONCHIP_CACHE_BYTES:
            return mb, mb, foot
        scores = 0.0 if name.startswith("k1q") else 4.0 * st[layer]["item_cols"]
        return min(mb, foot + x_bytes_q + 16.0 * k * rows + scores), mb, foot

    kernels, fam = [], {}
    for r in prof:
        ms = r["ms"] / max(1, r["launches"])
        hb, mb, foot = hbm_bytes(r["name"], r["layer"]) if st else (0.0, 0.0, 0.0)
        gbps = hb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        kernels.append(dict(name=r["name"], layer=r["layer"], ms=round(ms, 4), alg_bytes=hb, matched_bytes=mb, structure_bytes=foot,
                            gbps=round(gbps, 1), frac_hbm=round(gbps / HBM_PEAK_GBPS, 4),
                            matched_gbps=round(mb / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0))
        # a bound-pruned layer runs a kernel twice (first beam slots, then the rest of the unfinished queries: "<name>_rest"); both count
        # as ONE launch of the family: time summed, work (counted for both phases together by the stats pass) attributed once
        rest = r["name"].endswith("_rest") or r["name"].endswith("_mid")     # later stages of a bound-pruned layer
        f = fam.setdefault(r["name"].rsplit("_", 1)[0] if rest else r["name"], dict(ms=0.0, launches=0, bytes=0.0, matched=0.0))
        f["ms"] += r["ms"]
        if not rest:
            f["launches"] += r["launches"]; f["bytes"] += hb * r["launches"]; f["matched"] += mb * r["launches"]
    if not fam:
        return None
    dom = max(fam, key=lambda n: fam[n]["ms"])
    launches_per_step = fam[dom]["launches"] / max(1, args.steps)
    avg_ms = fam[dom]["ms"] / max(1, fam[dom]["launches"])
    per_launch = fam[dom]["bytes"] / max(1, fam[dom]["launches"])
    ach = per_launch / (avg_ms * 1e-3) / 1e9
    step_bytes = sum(kk["alg_bytes"] for kk in kernels)
    matched_rate = fam[dom]["matched"] / max(1, fam[dom]["launches"]) / (avg_ms * 1e-3) / 1e9
    cache_resident = fam[dom]["matched"] > fam[dom]["bytes"] * 1.0001
    traffic, tsrc, l2, requests, issue = None, None, None, None, None
    # counters: separate rocprofv3 --pmc passes over this same command on these same sources (scripts/gpu_round.sh pmc -> scripts/pmc_traffic.py).
    # A counter set holds per-STEP sums per KERNEL (k1q_kernel, k1_kernel, ...): every bench family that kernel serves -- several layers, both
    # phases of a pruned layer -- is priced together: counter bytes of the kernel per step / its GPU time per step.  Per launch = / the
    # family's launches per step.
    def pmc_family(name):
        return ("k1q_dense" if name.startswith("k1q") else "k1g_dense_x" if name.startswith("k1g") else "k1_sort_items" if name.startswith("k1_sort") else
                "k1_sparse" if name.startswith("k1_") else "k2_topk" if name.startswith("k2") else name)
    pfam = pmc_family(dom)
    pf_ms_step = sum(v["ms"] for n_, v in fam.items() if pmc_family(n_) == pfam) / max(1, args.steps)
    pf_launches_step = max(1.0, sum(v["launches"] for n_, v in fam.items() if pmc_family(n_) == pfam) / max(1, args.steps))
    tj, why_no_counters = (None, "the upload mode times per-batch launches") if args.include_upload else pmc_entry(args.config, args.scale, args.opt, world)
    ent = tj.get("kernels", {}).get(pfam) if tj else None
    if tj and not ent:
        why_no_counters = f"the counter set has no kernel family {pfam}"
    pmc_avg_ms = pf_ms_step / pf_launches_step
    if ent:
        traffic, tsrc = ent.get("hbm_bytes_per_step") / pf_launches_step, tj.get("source")
        if ent.get("l2_read_req_per_step"):
            req = ent["l2_read_req_per_step"]
            l2b = req * 64.0 / (pf_ms_step * 1e-3) / 1e9
            l2 = dict(bound="l2", read_requests_per_step=req, request_bytes=64, achieved=round(l2b, 1), peak=L2_PEAK_GBPS, unit="GB/s",
                      frac=round(l2b / L2_PEAK_GBPS, 4), requests_per_s_G=round(req / (pf_ms_step * 1e-3) / 1e9, 1),
                      hit_rate=round(ent["l2_hit_per_step"] / max(1.0, ent["l2_hit_per_step"] + (ent.get("fabric_read_req_per_step") or 0.0)), 3) if ent.get("l2_hit_per_step") else None,
                      note="TCP_TCC_READ_REQ (L1->L2 read requests) x 64 B / kernel time; the gathers of this kernel use 8-64 B of every request")
        if ent.get("fabric_read_req_per_step"):
            fr = ent["fabric_read_req_per_step"]
            rate = fr / (pf_ms_step * 1e-3) / 1e9
            requests = dict(fabric_read_req_per_step=fr, fabric_req_per_s_G=round(rate, 1), ceiling_G=FABRIC_REQ_CEILING_G, frac=round(rate / FABRIC_REQ_CEILING_G, 3),
                            note="L2 misses (TCC_MISS) / kernel time against the fabric read-request ceiling measured with 8-byte gathers "
                                 "(profiles/r02_calib_fetch.md: 51 G/s from HBM, 59 G/s from the Infinity Cache; 57 used)")
        if ent.get("valu_insts_per_step"):
            cu_cycles = pf_ms_step * 1e-3 * SHADER_CLOCK_HZ
            issue = dict(valu_insts_per_step=ent["valu_insts_per_step"], salu_insts_per_step=ent.get("salu_insts_per_step"),
                         valu_busy_frac=round(ent.get("valu_active_quad_cycles_per_step", 0.0) * 4.0 / (cu_cycles * N_SIMD), 3),
                         salu_issue_frac=round((ent.get("salu_insts_per_step") or 0.0) / (cu_cycles * N_CU), 3),
                         note="SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (kernel time x 2.4 GHz x 1024 SIMDs); SQ_INSTS_SALU / (kernel time x 2.4 GHz x 256 CUs: "
                              "one scalar issue per CU per cycle).  A wavefront VALU instruction occupies its SIMD for 4 cycles, so instruction COUNT per "
                              "query is what these kernels run on")
    if l2 is None and cache_resident:
        l2 = dict(bound="l2", achieved=round(matched_rate, 1), peak=L2_PEAK_GBPS, unit="GB/s", frac=round(matched_rate / L2_PEAK_GBPS, 4),
                  note="the structure this kernel gathers from fits the on-chip cache: matched-work bytes (no inter-query reuse) / launch time against the L2 peak")
    if dom.startswith("k1g") and st:
        # dense queries: the dominant kernel is a k-ordered fp32 SGEMM -> flops roofline.  2 flops per multiply-add over the
        # (feature, padded column) cells the items address; peak = the dense fp32 matrix/vector rate (MI355X_MICROARCH.md: 157.3 TF).
        # The reference's arithmetic is a separately rounded multiply and add (no FMA), so 78.6 TFLOP/s is the rate this
        # instruction mix can reach at best (frac_of_reachable).
        flops = sum(2.0 * st[r["layer"]]["x_cols"] * r["launches"] for r in prof if r["name"] == dom) / max(1, fam[dom]["launches"])
        tf = flops / (avg_ms * 1e-3) / 1e12
        return dict(bound="valu", bound_note="the kernel issues v_pk_mul_f32 + v_pk_add_f32 (no MFMA, no FMA: the reference rounds the product and the sum separately); "
                    "peak = the fp32 VECTOR rate with FMA, frac_of_reachable = against half of it (what separate multiply + add can reach); measured on-chip bound "
                    "(vector issue + LDS + barriers: profiles/r06_leaf.md section 5), not HBM",
                    kernel=dom, achieved=round(tf, 2), peak=157.3, unit="TFLOP/s", frac=round(tf / 157.3, 4), frac_of_reachable=round(tf / 78.6, 4),
                    traffic=traffic, flops_per_launch=flops, avg_launch_ms=round(avg_ms, 4), launches_per_step=launches_per_step,
                    model="2 flops per multiply-add of every (query feature, candidate column) cell; fp32 multiply and add rounded separately (the reference carries no FMA), "
                          "so MFMA (fused) cannot be used and half the fp32 peak is the reachable rate",
                    per_kernel_ms_per_step={n: round(v["ms"] / max(1, args.steps), 4) for n, v in fam.items()}, kernels=kernels,
                    work=[dict(layer=l, **{kk: st[l][kk] for kk in ("items", "probes", "hit_rows", "hit_entries", "candidates")}) for l in range(len(st))])
    ref_layout = (sum(s_["ref_chunk_bytes"] + 4.0 * s_["candidates"] for s_ in st) + x_bytes_q) if st else None
    if traffic is not None:
        ach_c = traffic / (pmc_avg_ms * 1e-3) / 1e9
        head = dict(achieved=round(ach_c, 1), frac=round(ach_c / HBM_PEAK_GBPS, 4), pmc_kernel=pfam, pmc_kernel_ms_per_step=round(pf_ms_step, 4),
                    basis="pmc: rocprofv3 FETCH_SIZE + WRITE_SIZE of the kernel per step (profiles/pmc_traffic.json, same sources: csrc " + csrc_sha16() +
                          ") / its hipEvent time per step measured in this run")
    else:
        head = dict(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBPS, 4),
                    basis=f"matched-work bytes ({why_no_counters}): NOT an HBM fraction, see frac_matched")
    return dict(bound="hbm", kernel=dom, **head, peak=HBM_PEAK_GBPS, unit="GB/s",
                traffic=traffic, traffic_source=tsrc,
                frac_matched=round(matched_rate / HBM_PEAK_GBPS, 4), matched_gbps=round(matched_rate, 1),
                frac_ref_layout=round(ref_layout / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 2) if ref_layout else None,
                frac_ref_layout_note="SURVEY.md 8(d): every active reference chunk streamed whole, over the whole step; > 1 because chunks are looked up, not streamed",
                requests=requests, issue=issue, l2=l2,
                pruning=None if any(o.replace(" ", "") == "prune=0" for o in args.opt) else
                        "exact bound pruning is on: the matched-work figures count the items the kernels EVALUATE (the untimed stats pass stages every layer "
                        "like its timed kernel does); frac_ref_layout prices the UNPRUNED reference (every active chunk of every beam parent streamed) "
                        "against this step's time and is far above 1 -- `frac` (counter bytes) is what moved",
                alg_bytes_per_launch=per_launch,
                # counter-based block: the launches priced are ALL launches of the kernel the counters belong to (K1Q: the fused narrow levels +
                # the sorted launch of the last dense-format level; both stages of a pruned layer) -- `achieved` = `traffic` / `avg_launch_ms`
                avg_launch_ms=round(pmc_avg_ms if traffic is not None else avg_ms, 4),
                launches_per_step=pf_launches_step if traffic is not None else launches_per_step,
                launches_priced=sorted(n_ for n_ in fam if pmc_family(n_) == pfam) if traffic is not None else [dom],
                dominant_family_avg_launch_ms=round(avg_ms, 4),
                model="frac: counter bytes; frac_matched: matched work, no inter-query reuse (compulsory bytes for cache-resident structures); "
                      "frac_ref_layout: SURVEY 8(d) chunk streaming (see bench.py docstring / DESIGN.md section 4)",
                step_alg_bytes=step_bytes, step_gbps=round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                alg_bytes_ref_layout=ref_layout,
                per_kernel_ms_per_step={n: round(v["ms"] / max(1, args.steps), 4) for n, v in fam.items()},
                kernels=kernels,
                work=[dict(layer=l, **{kk: st[l][kk] for kk in ("items", "probes", "hit_rows", "hit_entries", "candidates")}) for l in range(len(st))])


def baseline_metric():
    """The metric string of BASELINE.json (the line is judged against it), with a literal fallback."""
    try:
        return json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"]
    except Exception:
        return "XLinear queries/sec @ beam=10 top-k=10; P@1 vs reference; 1/2/4/8 GPU"


def timed_output_parity(folder, Xpar, G, beam, topk, world, rows_per_shard, log):
    """The buffers the TIMED steps filled (after the all-gather at N > 1), rows of every shard, against the reference
    (oracle/_ref when present, else the C restatement on fewer rows): label ids, order and fp32 score bits."""
    import numpy as np
    from oracle import xrl_oracle as O
    if O.ref_available():
        ref, kind, n = O.RefModel(folder), "reference BINARY_SEARCH_CHUNKED", Xpar.shape[0]
        P = ref.predict(Xpar, beam_size=beam, only_topk=topk, threads=min(os.cpu_count() or 1, 32))
    else:
        ref, kind, n = O.OracleModel.load(folder), "C restatement (oracle/_ref absent)", min(Xpar.shape[0], 512)
        P = ref.predict(Xpar[:n], beam_size=beam, only_topk=topk)
        G = G[:n]
    same_rows = np.array_equal(G.indptr, P.indptr)
    same_idx = bool(same_rows and np.array_equal(G.indices, P.indices))
    bit = bool(same_idx and np.array_equal(G.data.astype(np.float32).view(np.uint32), P.data.astype(np.float32).view(np.uint32)))
    log(f"timed output vs {kind}: {n} rows ({rows_per_shard} per shard x {world}): indices identical={same_idx} scores bit-identical={bit}")
    return dict(timed_output_identical=bool(same_idx and bit), timed_output_rows=int(n),
                timed_output_sample=f"first {rows_per_shard} rows of each of the {world} shard(s) = {n} rows, "
                f"read back from the buffers the timed steps wrote, vs {kind}", timed_output_indices_identical=same_idx, timed_output_scores_bit_identical=bit)


def shard8_line(clib, h, q, torch, dev, stream, tstream, beam, args, k, n_total, ms_full, log):
    """extra.shard8: the one-GPU PROXY of the 8-GPU run (SCALE_rNN needs an 8-GPU node): the step on ONE rank's shard -- the first
    n_total / 8 rows, device-resident, same entry point and options -- timed like the main loop.  projected_speedup_8 = full-step time /
    shard-step time: what 8 ranks would reach if the all-gather hides completely under the next step (it is queued on a second
    stream for that) and RCCL took no CUs.  Not a measurement of 8 GPUs."""
    rows = n_total // 8
    idx = torch.zeros((rows, k), dtype=torch.int32, device=dev); val = torch.zeros((rows, k), dtype=torch.float32, device=dev)
    cnt = torch.zeros((rows,), dtype=torch.int32, device=dev)

    def step():
        clib.predict_device_rows(h, q, beam, None, args.topk, idx.data_ptr(), val.data_ptr(), cnt.data_ptr(), k, 0, rows, stream=stream, sync=False)
    with torch.cuda.stream(tstream):
        for _ in range(max(args.warmup, 30)):     # (the pruning feedback's item counts settle at the shard's size; the clocks are back up after the CPU legs)
            step()
        torch.cuda.synchronize()
        n = max(args.steps, 200)      # (0.85 ms steps: 50 of them ran 15 % slower than 200 -- the device had idled through the host-ABI and CPU legs)
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    log(f"extra.shard8: {rows} rows in {ms:.3f} ms per step -> projected 8-GPU speed-up {ms_full / ms:.2f}x (one-GPU proxy)")
    return dict(rows=rows, ms_per_step=round(ms, 4), steps=n, projected_speedup_8=round(ms_full / ms, 2),
                note="one-GPU proxy of the strong-scaling run: one rank's shard (n_total / 8 rows) on this GPU; assumes the all-gather of step s hides under "
                     "step s+1 and RCCL takes no CUs; NOT an 8-GPU measurement")


def hard_line(args, log):
    """extra.hard: this same benchmark on `amazon-670k-hard` (same shape; query-dependent routing, unsaturated hinge: bound pruning settles
    ~nothing -- profiles/r04_hard_config.md), run as a child process with the same steps / warm-up, its timed output compared with the
    reference on --parity-rows rows.  The headline `value` stays on the BASELINE configuration."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "amazon-670k-hard", "--steps", str(args.steps), "--warmup", str(max(args.warmup, 6)),
           "--no-cpu-baseline", "--no-host-abi", "--no-extra", "--cache", args.cache, "--parity-rows", str(args.parity_rows)]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        raise RuntimeError(f"child bench failed rc={r.returncode}: {r.stderr[-400:]}")
    j = json.loads(line)
    rf = j.get("roofline") or {}
    log(f"extra.hard: {j['ms_per_step']} ms per step = {j['value'] / 1e6:.1f} M q/s ({time.time() - t0:.0f} s incl. generating the workload)")
    return dict(config="amazon-670k-hard", value=j["value"], unit=j["unit"], ms_per_step=j["ms_per_step"], steps=j["steps"], warmup=j["warmup"],
                roofline=dict(kernel=rf.get("kernel"), frac=rf.get("frac"), achieved=rf.get("achieved"), basis=rf.get("basis"), traffic=rf.get("traffic"),
                              requests=rf.get("requests"), issue=rf.get("issue"), per_kernel_ms_per_step=rf.get("per_kernel_ms_per_step")),
                parity=j.get("parity"), workload=j["config"]["workload"])


def text_to_labels(clib, model, folder, X, beam, topk, n_docs, log):
    """extra.text_to_labels: documents/s of texts -> labels with X never on the host -- pecos_amd.features.predict_text:
    tokenise + n-gram lookup on host threads, term counts uploaded once, tf-idf weighting + l2 norm on the device (K5), beam search in
    place -- beside the reference's own pipeline on the same corpus and host cores: its compiled c_tfidf_predict (host CSR out) followed by
    its XLinearModel predict on that CSR (oracle/_ref).  The corpus is SYNTHETIC: document i holds the word of every feature of query row i
    (a unigram vectorizer over the model's feature dimension, written here in the reference's file format), so the beam search sees this
    workload's sparsity pattern; both pipelines' labels are compared on the reference's sample."""
    import ctypes as C
    import tempfile
    import numpy as np
    from pecos_amd import features
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import n4_producer_bench as N4
    n_docs = int(min(n_docs, X.shape[0]))
    D = X.shape[1]
    rng = np.random.default_rng(5)
    words = np.array([f"t{i:x}" for i in range(D)])
    Xd = X[:n_docs]
    tok = words[Xd.indices]
    corpus = [" ".join(tok[Xd.indptr[i]:Xd.indptr[i + 1]]) for i in range(n_docs)]
    tmp = tempfile.mkdtemp(prefix="t2l_")
    vdir = os.path.join(tmp, "vectorizer")
    os.makedirs(vdir)
    N4.write_vectorizer(vdir, list(words), [(i,) for i in range(D)], rng)
    vec = features.Tfidf.load(vdir)
    kw = dict(beam_size=beam, only_topk=topk)
    features.predict_text(vec, model, corpus[:4096], **kw)            # warm-up (tables, pinned buffers)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); Y = features.predict_text(vec, model, corpus, **kw); ts.append(time.perf_counter() - t0)
    t_gpu = float(np.median(ts))
    res = dict(docs=n_docs, tokens_per_doc=round(float(Xd.nnz) / n_docs, 1), vectorizer=f"synthetic unigram tf-idf, {D} words, l2 norm",
               value=round(n_docs / t_gpu, 1), unit="documents/s", ms=round(t_gpu * 1e3, 2),
               includes="Python call, tokeniser + n-gram lookup on host threads, H2D of the term counts, tf-idf weighting + norm on the device, beam search, D2H of the top-k, CSR assembly")
    log(f"extra.text_to_labels: {n_docs} documents in {t_gpu * 1e3:.1f} ms = {n_docs / t_gpu / 1e6:.2f} M docs/s")
    # the reference's two calls on a bounded sample
    from oracle import xrl_oracle as O
    ref_so = os.path.join(REPO, "oracle", "_ref", "libpecos_float32.so")
    if O.ref_available() and os.path.exists(ref_so):
        import scipy.sparse as smat
        ref = C.CDLL(ref_so)
        ref.c_tfidf_load.restype = C.c_void_p; ref.c_tfidf_load.argtypes = [C.c_char_p]
        ref.c_tfidf_predict.restype = None; ref.c_tfidf_predict.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_int, N4.ALLOC]
        ref.c_tfidf_destruct.argtypes = [C.c_void_p]
        ns = min(n_docs, 20000)
        arr, dl, n = clib._corpus_arrays(corpus[:ns])
        dlp = dl.ctypes.data_as(C.POINTER(C.c_uint64))
        rh = ref.c_tfidf_load(vdir.encode())
        rm = O.RefModel(folder, "BINARY_SEARCH_CHUNKED")     # (the layout whose arithmetic order the GPU path reproduces bit for bit)
        ncpu = os.cpu_count() or 1
        best = None
        for th in sorted({t for t in (16, 32, 64, ncpu) if t <= ncpu}):
            w = N4.Warm(); f = N4.ALLOC(w)
            t0 = time.perf_counter()
            ref.c_tfidf_predict(C.c_void_p(rh), arr, dlp, n, th, f)
            Xr = smat.csr_matrix((w.a[2], w.a[0].astype(np.int32), w.a[1].astype(np.int64)), shape=(n, D))
            P = rm.predict(Xr, beam_size=beam, only_topk=topk, threads=th)
            t1 = time.perf_counter() - t0
            if best is None or t1 < best[0]:
                best = (t1, th, P)
        ref.c_tfidf_destruct(C.c_void_p(rh))
        t_ref, th, P = best
        G = Y[:ns]
        same = bool(np.array_equal(G.indptr, P.indptr) and np.array_equal(G.indices, P.indices))
        rel = float(np.max(np.abs(G.data - P.data) / np.maximum(np.abs(P.data), 1e-30))) if same and P.nnz else None
        res["reference"] = dict(value=round(ns / t_ref, 1), unit="documents/s", sample=ns, threads=th, cores=ncpu,
                                what="the reference's compiled library on this host: c_tfidf_predict (host CSR) + c_xlinear_predict_csr_f32 (BINARY_SEARCH_CHUNKED), best thread count of a sweep")
        res["labels_identical_to_reference"] = same
        res["scores_bit_identical_to_reference"] = bool(same and np.array_equal(G.data.view(np.uint32), P.data.view(np.uint32)))
        res["max_rel_err_scores"] = rel
        res["speedup"] = round((n_docs / t_gpu) / (ns / t_ref), 1)
        log(f"extra.text_to_labels: reference pipeline {ns / t_ref:.0f} docs/s with {th} threads; labels identical: {same}, max rel err {rel}")
    return res


def cpu_baseline(folder, X, model, beam, topk, budget_s, log):
    """Time the real reference (oracle/_ref, built from /root/reference's own sources) on this box's host cores on a
    bounded sample of the workload; fall back to the single-threaded C restatement ("port") when oracle/_ref is absent.
    The reference's OpenMP path does not scale to hundreds of threads on a small batch, so the thread count is swept on
    a small sample first and the best one is used (cores = threads used); then 1 warm-up + median of 5 timed calls
    (SURVEY.md 8d).  Also compares the GPU output with the reference's on that sample."""
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
            rm.predict(X[:ns], beam_size=beam, only_topk=topk, threads=min(ncpu, 32))   # page faults
            best = None
            for th in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
                t0 = time.perf_counter(); rm.predict(X[:ns], beam_size=beam, only_topk=topk, threads=th); t1 = time.perf_counter() - t0
                if best is None or ns / t1 > best[0]:
                    best = (ns / t1, th)
            th = best[1]
            ns = int(min(n, max(ns, best[0] * budget_s / 2 / 6)))          # 6 calls (warm-up + 5) per layout within the budget
            Xq = X[:ns]
            P = rm.predict(Xq, beam_size=beam, only_topk=topk, threads=th)  # warm-up
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); P = rm.predict(Xq, beam_size=beam, only_topk=topk, threads=th); ts.append(time.perf_counter() - t0)
            t1 = float(np.median(ts))
            # BASELINE.md's procedure is threads=-1 (every host core): one warm-up + median of 3 on the same sample, reported beside the sweep's best
            rm.predict(Xq, beam_size=beam, only_topk=topk, threads=ncpu)
            ta = []
            for _ in range(3):
                t0 = time.perf_counter(); rm.predict(Xq, beam_size=beam, only_topk=topk, threads=ncpu); ta.append(time.perf_counter() - t0)
            results[wtype] = dict(qps=ns / t1, sample=ns, threads=th, load_s=round(load_s, 1), qps_all_cores=ns / float(np.median(ta)))
            log(f"cpu reference {wtype}: {ns} queries, median of 5 = {t1:.3f}s ({min(ts):.3f}-{max(ts):.3f}) = {ns / t1:.0f} q/s with {th} threads of {ncpu} "
                f"(threads=-1, all {ncpu}: {results[wtype]['qps_all_cores']:.0f} q/s; load {load_s:.1f}s)")
            if wtype == "BINARY_SEARCH_CHUNKED":
                G = model.predict(Xq, beam_size=beam, only_topk=topk)
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
                    all_cores=dict(value=round(max(r["qps_all_cores"] for r in results.values()), 1), cores=ncpu,
                                   note="threads=-1 as in BASELINE.md (every host core), better of the two layouts, same sample, median of 3"),
                    sample=f"first {results[bw]['sample']} queries of the workload, layout {bw}, best thread count of a sweep "
                           f"({results[bw]['threads']} OpenMP threads on a {ncpu}-cpu host), 1 warm-up + median of 5 timed calls; other layout: " +
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
