"""GPU (-m gpu): queries that are already in HBM and the RCCL step of bench.py with one rank.

* device CSR / dense hand-off (xrl_queries_from_device_*): same bits as the host-ABI predict;
* concat_model's query form [X_feat | X_emb] assembled on the device (xrl_queries_concat_device) == predicting on the
  host-side TransformerMatcher.concat_features matrix;
* bench.py's step -- predict on an explicit stream followed by ONE packed all_gather_into_tensor -- under
  torch.distributed.run with the "nccl" (= RCCL) backend and one rank, so that RCCL initialisation and the collective on a
  non-default stream have executed on hardware at least once before the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import GOLDEN, REPO, assert_same_topk, load_X

pytestmark = pytest.mark.gpu


def _rows_to_csr(idx, val, cnt, n_cols):
    from pecos_amd.distributed import rows_to_csr
    return rows_to_csr(idx.cpu().numpy().view(np.uint32), val.cpu().numpy(), cnt.cpu().numpy(), n_cols)


def test_device_resident_queries_and_concat():
    import torch
    from pecos_amd import XLinearModel, clib
    from pecos_amd.features import concat_features, predict_from_torch
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    m = XLinearModel.load(folder)
    dev = torch.device("cuda", 0)
    kw = dict(beam_size=5, only_topk=7)
    want = m.predict(X, **kw)
    crow = torch.from_numpy(X.indptr.astype(np.int64)).to(dev); col = torch.from_numpy(X.indices.astype(np.int32)).to(dev)
    val = torch.from_numpy(X.data.astype(np.float32)).to(dev)
    got = _rows_to_csr(*predict_from_torch(m, crow, col, val, X.shape[1], **kw), m.nr_pred_cols)
    assert_same_topk(got, want, exact_scores=True, what="device-resident CSR")
    # dense device matrix
    Xd = np.ascontiguousarray(X.toarray())
    td = torch.from_numpy(Xd).to(dev)
    h = m.model.model_chain
    q = clib.queries_from_device_drm(h, Xd.shape[0], Xd.shape[1], td.data_ptr())
    k = clib.effective_topk(h, 7)
    idx = torch.zeros((Xd.shape[0], k), dtype=torch.int32, device=dev); sc = torch.zeros((Xd.shape[0], k), dtype=torch.float32, device=dev)
    cnt = torch.zeros((Xd.shape[0],), dtype=torch.int32, device=dev)
    clib.predict_device(h, q, 5, None, 7, idx.data_ptr(), sc.data_ptr(), cnt.data_ptr(), k, sync=True)
    clib.queries_free(q)
    assert_same_topk(_rows_to_csr(idx, sc, cnt, m.nr_pred_cols), m.predict(Xd, **kw), exact_scores=True, what="device-resident dense X")
    # concat_model form: the last H feature columns play the embedding block
    H = 24
    D = X.shape[1]
    X_feat = X[:, : D - H].tocsr(); X_feat.sort_indices()
    emb = np.ascontiguousarray(X[:, D - H:].toarray()) + np.float32(0.25)      # dense block, no zeros
    host_cat = concat_features(X_feat, emb, normalize_emb=False)
    want_cat = m.predict(host_cat, **kw)
    fcrow = torch.from_numpy(X_feat.indptr.astype(np.int64)).to(dev); fcol = torch.from_numpy(X_feat.indices.astype(np.int32)).to(dev)
    fval = torch.from_numpy(X_feat.data.astype(np.float32)).to(dev)
    got_cat = _rows_to_csr(*predict_from_torch(m, fcrow, fcol, fval, D - H, emb=torch.from_numpy(emb).to(dev), **kw), m.nr_pred_cols)
    assert_same_topk(got_cat, want_cat, exact_scores=True, what="[X_feat | X_emb] assembled on the device")
    # normalised embeddings (the reference's default): host mirror == sklearn on the same data
    nrm = concat_features(X_feat, emb, normalize_emb=True)
    assert nrm.shape == host_cat.shape and np.allclose(np.asarray(nrm[:, D - H:].multiply(nrm[:, D - H:]).sum(axis=1)).ravel(), 1.0, atol=1e-5)


def test_device_concat_vs_reference_concat_features(manifest):
    # xrl_queries_concat_device_ex with the embedding normalisation done ON THE DEVICE, read back and compared with the output of the
    # reference's own TransformerMatcher.concat_features (tests/golden/concat/, made by make_golden_r03.py): pattern and order
    # identical (every cell of the dense block is a stored entry), sparse part bit-identical, normalised block within 1e-6 relative
    # (sklearn sums the squares in numpy's order, the kernel in a wavefront tree)
    import torch
    from pecos_amd import XLinearModel, clib
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    emb = np.load(os.path.join(GOLDEN, "concat", "X_emb.npy"))
    m = XLinearModel.load(folder)
    h = m.model.model_chain
    dev = torch.device("cuda", 0)
    crow = torch.from_numpy(X.indptr.astype(np.int64)).to(dev); col = torch.from_numpy(X.indices.astype(np.int32)).to(dev)
    val = torch.from_numpy(X.data.astype(np.float32)).to(dev); temb = torch.from_numpy(emb).to(dev)
    for c in manifest["concat"]:
        if c["feat"] != "csr":
            continue
        z = np.load(os.path.join(GOLDEN, "concat", c["out"]))
        q = clib.queries_concat_device(h, X.shape[0], X.shape[1], crow.data_ptr(), col.data_ptr(), val.data_ptr(), int(X.nnz), emb.shape[1],
                                       temb.data_ptr(), normalize_emb=c["normalize_emb"])
        got = clib.queries_download(q)
        clib.queries_free(q)
        assert got.shape == tuple(z["shape"]) and np.array_equal(got.indptr, z["indptr"]) and np.array_equal(got.indices, z["indices"]), c
        is_emb = got.indices >= X.shape[1]
        assert np.array_equal(got.data[~is_emb].view(np.uint32), z["data"][~is_emb].view(np.uint32))
        if c["normalize_emb"]:
            assert np.all(np.abs(got.data[is_emb] - z["data"][is_emb]) <= 1e-6 * np.abs(z["data"][is_emb]) + 1e-30)
        else:
            assert np.array_equal(got.data[is_emb].view(np.uint32), z["data"][is_emb].view(np.uint32))


def test_multi_device_behind_the_c_abi(oracle_mod, tmp_path):
    # xrl_set_option(h, "devices", n): the SAME drop-in entry points (c_xlinear_predict_{csr,drm}_f32, here through XLinearModel.predict)
    # then shard the rows over n copies of the compiled model, one host thread + stream + pinned staging per copy, and fill the arrays
    # of the ONE allocator call.  On a one-GPU box the copies share the device ("virtual devices"), which exercises everything but the
    # second PCIe link: results must equal the single-device ones bit for bit -- small X (direct upload), large X (pipelined staged
    # upload per shard), dense X, fewer rows than devices, empty rows at the shard boundaries.
    import xrl_synth
    from pecos_amd import XLinearModel, clib
    folder = str(tmp_path / "m")
    ks, X, cfg = xrl_synth.make_config("eurlex-4k", folder, scale=0.5)
    m = XLinearModel.load(folder)
    h = m.model.model_chain
    om = oracle_mod.OracleModel.load(folder)
    kw = dict(beam_size=10, only_topk=10)
    Xs = X[:600].tolil(); Xs[199] = 0; Xs[200] = 0; Xs[399] = 0; Xs = Xs.tocsr().astype(np.float32); Xs.eliminate_zeros(); Xs.sort_indices()
    Xbig = smat.vstack([X] * 12).tocsr(); Xbig.sort_indices()             # > 32 MB per shard: the staged upload path
    want_s, want_big = m.predict(Xs, **kw), m.predict(Xbig, **kw)
    assert_same_topk(want_s, om.predict(Xs, **kw), exact_scores=True, what="single device vs oracle")
    Xd = np.ascontiguousarray(X[:300].toarray())
    want_d = m.predict(Xd, **kw)
    for n in (3, 2):
        clib.set_option(h, "devices", n)
        assert clib.xlinear_get_int_attr(h, "nr_devices") == n
        for trial in range(2):
            assert_same_topk(m.predict(Xs, **kw), want_s, exact_scores=True, what=f"{n} devices, small X, trial {trial}")
        assert_same_topk(m.predict(Xbig, **kw), want_big, exact_scores=True, what=f"{n} devices, staged upload")
        assert_same_topk(m.predict(Xd, **kw), want_d, exact_scores=True, what=f"{n} devices, dense X")
        assert_same_topk(m.predict(Xs[:n], **kw), want_s[:n], exact_scores=True, what=f"{n} devices, {n} rows")
        assert_same_topk(m.predict(Xs[:1], **kw), want_s[:1], exact_scores=True, what=f"{n} devices, one row")
        clib.set_option(h, "dense_layers", 0)                              # options reach the replicas
        assert_same_topk(m.predict(Xs, **kw), want_s, exact_scores=True, what=f"{n} devices, tile format")
        clib.set_option(h, "dense_layers", 1)
    clib.set_option(h, "devices", 1)
    assert clib.xlinear_get_int_attr(h, "nr_devices") == 1
    assert_same_topk(m.predict(Xs, **kw), want_s, exact_scores=True, what="back to one device")


def test_predict_device_rows_ranges_vs_oracle(oracle_mod):
    # xrl_predict_device_rows (what bench.py's timed step and the sharded path call): arbitrary row ranges with row_begin > 0,
    # results landing at the SAME rows of the caller's buffers (rows outside the range untouched), sparse and dense X, tile-format
    # and dense-format kernels, against the CPU oracle -- label ids, order and score bits
    import torch
    from pecos_amd import XLinearModel, clib
    from pecos_amd.distributed import PackedTopk
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    X = smat.vstack([X, X[::-1]]).tocsr(); X.sort_indices()
    n = X.shape[0]
    m = XLinearModel.load(folder)
    om = oracle_mod.OracleModel.load(folder)
    h = m.model.model_chain
    dev = torch.device("cuda", 0)
    kw = dict(beam_size=6, only_topk=9)
    k = clib.effective_topk(h, 9)
    for Xq in (X, np.ascontiguousarray(X.toarray())):
        want = om.predict(Xq, **kw)
        q = clib.queries_upload(h, Xq)
        for dl in (1, 0):
            clib.set_option(h, "dense_layers", dl)
            for (b, e) in ((0, n), (1, n), (n // 3, n // 3 + 1), (n // 2, n), (7, n - 5), (n - 1, n), (n, n)):
                idx = torch.full((n, k), -1, dtype=torch.int32, device=dev); sc = torch.full((n, k), -7.0, dtype=torch.float32, device=dev)
                cnt = torch.full((n,), -3, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                clib.predict_device_rows(h, q, 6, None, 9, idx.data_ptr(), sc.data_ptr(), cnt.data_ptr(), k, b, e - b, sync=True)
                got = _rows_to_csr(idx[b:e], sc[b:e], cnt[b:e], m.nr_pred_cols)
                assert_same_topk(got, want[b:e], exact_scores=True, what=f"rows [{b},{e}) dense_layers={dl} dense_x={not smat.issparse(Xq)}")
                out = torch.ones(n, dtype=torch.bool, device=dev); out[b:e] = False
                assert bool((cnt[out] == -3).all()) and bool((idx[out] == -1).all()), "rows outside the range were written"
        clib.set_option(h, "dense_layers", 1)
        # the packed two-part layout bench.py uses: absolute local row indexing into per-part buffers
        bounds = np.array([0, n])
        pk = PackedTopk(bounds, 0, k, dev, parts=2)
        for p in range(2):
            b, e = pk.rows(p)
            pi, pv, pc, ps = pk.pointers(p)
            clib.predict_device_rows(h, q, 6, None, 9, pi, pv, pc, ps, b, e - b, sync=True)
            pk.gather(p)
        gi, gv, gc = pk.unpack()
        assert_same_topk(_rows_to_csr(gi, gv, gc, m.nr_pred_cols), want, exact_scores=True, what="PackedTopk two parts")
        clib.queries_free(q)


def test_async_predicts_on_two_streams_share_the_handle():
    # xrl_predict_device(sync=0) on two caller streams: the handle's scratch buffers are shared, so the second predict must be
    # ordered after the first (event recorded at the end of every predict); both results equal the synchronous ones
    import torch
    from pecos_amd import XLinearModel, clib
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))
    reps = 40                                         # enough rows for the kernels of two predicts to overlap if nothing ordered them
    Xa = smat.vstack([X] * reps).tocsr(); Xb = smat.vstack([X[::-1]] * reps).tocsr()
    Xa.sort_indices(); Xb.sort_indices()
    m = XLinearModel.load(folder)
    h = m.model.model_chain
    dev = torch.device("cuda", 0)
    k = clib.effective_topk(h, 7)
    want = [m.predict(Xa, beam_size=5, only_topk=7), m.predict(Xb, beam_size=5, only_topk=7)]
    qs = [clib.queries_upload(h, Xa), clib.queries_upload(h, Xb)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for trial in range(3):
        outs = []
        for q, st, Xq in zip(qs, streams, (Xa, Xb)):
            n = Xq.shape[0]
            idx = torch.zeros((n, k), dtype=torch.int32, device=dev); sc = torch.zeros((n, k), dtype=torch.float32, device=dev)
            cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            outs.append((idx, sc, cnt))
        for q, st, o in zip(qs, streams, outs):
            clib.predict_device(h, q, 5, None, 7, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), k, stream=st.cuda_stream, sync=False)
        torch.cuda.synchronize()
        for o, w in zip(outs, want):
            assert_same_topk(_rows_to_csr(*o, m.nr_pred_cols), w, exact_scores=True, what=f"async predicts on two streams, trial {trial}")
    for q in qs:
        clib.queries_free(q)


@pytest.mark.timeout(600)
def test_bench_step_under_rccl_one_rank(tmp_path):
    env = dict(os.environ, XRL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "eurlex-4k", "--scale", "0.25",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-host-abi", "--cache", str(tmp_path / "cache")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0


@pytest.mark.timeout(900)
def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    # the whole N > 1 control flow of bench.py before the driver ever gets an 8-GPU node -- two ranks launched exactly as
    # the driver launches them, both on device 0, packed rows exchanged through gloo (host-staged): nnz-balanced shard bounds, the
    # double-buffered GatherPipeline, max-over-ranks timing, the timed output of BOTH shards compared with the reference, one JSON line
    # from rank 0
    env = dict(os.environ, XRL_BENCH_BACKEND="gloo", XRL_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "eurlex-4k", "--scale", "0.5",
           "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-host-abi", "--parity-rows", "1500", "--cache", str(tmp_path / "cache")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    p = out["parity"]
    assert p["timed_output_identical"] and p["timed_output_scores_bit_identical"] and "2 shard" in p["timed_output_sample"], p
    assert "gloo" in out["config"]["parallelism"]


def test_device_tfidf_weighting_vs_reference(manifest):
    # xrl_queries_tfidf_device: term counts on the device -> the reference's tf / idf / norm arithmetic -> query handle, read back and
    # compared with the output of the reference's own c_tfidf_predict (tests/golden/tfidf/): bit-identical (sublinear_tf: the
    # device's logf, <= 1 ulp); and a beam search fed from that handle equals the one fed with the reference's matrix
    import torch
    from pecos_amd import XLinearModel, clib
    from pecos_amd.features import predict_tfidf_from_torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_abi_and_host import _tfidf_case
    import xrl_synth
    dev = torch.device("cuda", 0)
    for c in manifest["tfidf"]:
        counts, want, kw = _tfidf_case(c)
        D = counts.shape[1]
        import tempfile
        with tempfile.TemporaryDirectory() as folder:
            xrl_synth.make_model(folder, D, 300, [max(8, D // 6), max(6, D // 10), 12], seed=31, shape=[4, 24, 300])
            m = XLinearModel.load(folder)
            h = m.model.model_chain
            crow = torch.from_numpy(counts.indptr.astype(np.int64)).to(dev); col = torch.from_numpy(counts.indices.astype(np.int32)).to(dev)
            cnt = torch.from_numpy(counts.data.astype(np.float32)).to(dev)
            idf = torch.from_numpy(kw["idf"]).to(dev) if kw["idf"] is not None else None
            q = clib.queries_tfidf_device(h, counts.shape[0], D, crow.data_ptr(), col.data_ptr(), cnt.data_ptr(), int(counts.nnz),
                                          idf.data_ptr() if idf is not None else None, kw["binary"], kw["sublinear_tf"], 1 if kw["norm"] == "l1" else 2)
            got = clib.queries_download(q)
            clib.queries_free(q)
            assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices), c
            if kw["sublinear_tf"]:
                assert np.all(np.abs(got.data - want.data) <= 2.5e-7 * np.abs(want.data)), c
            else:
                assert np.array_equal(got.data.view(np.uint32), want.data.view(np.uint32)), c
                pk = dict(beam_size=5, only_topk=7)
                P = _rows_to_csr(*predict_tfidf_from_torch(m, crow, col, cnt, D, idf=idf, binary=kw["binary"], sublinear_tf=False, norm=kw["norm"], **pk), m.nr_pred_cols)
                assert_same_topk(P, m.predict(want, **pk), exact_scores=True, what=f"beam search on device tf-idf queries, {c}")
