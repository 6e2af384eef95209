"""CPU, world_size 2 over gloo: the N>1 path (nnz-balanced sharding + ONE packed all-gather (PackedTopk, the same
object bench.py's step uses) + CSR assembly) with the oracle standing in for the per-rank GPU compute."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, assert_same_topk, load_X


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import xrl_oracle
    from pecos_amd.distributed import ShardedXLinear
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    folder = os.path.join(GOLDEN, "synth", "s_eurlex")
    om = xrl_oracle.OracleModel.load(folder)
    X = load_X(os.path.join(GOLDEN, "synth", "s_eurlex__X.npz"))

    class Stub:  # what ShardedXLinear needs from a model
        nr_pred_cols = om.nr_labels

    def fn(Xs, beam, topk, pp):
        if Xs.shape[0] == 0:
            k = topk or 20
            return (torch.zeros((0, k), dtype=torch.int32), torch.zeros((0, k)), torch.zeros((0,), dtype=torch.int32))
        idx, val, cnt, _ = om.predict_arrays(Xs, beam or 0, topk or 0, pp)
        return (torch.from_numpy(idx.view(np.int32).copy()), torch.from_numpy(val.copy()), torch.from_numpy(cnt.astype(np.int32)))

    sh = ShardedXLinear(Stub(), predict_shard_fn=fn)
    P = sh.predict(X, beam_size=10, only_topk=10)
    # each rank passing only ITS rows (what bench.py does): same result
    from pecos_amd.distributed import shard_bounds, take_rows
    bounds = shard_bounds(X, world)
    Pl = sh.predict_shard(take_rows(X, int(bounds[rank]), int(bounds[rank + 1])), bounds, beam_size=10, only_topk=10)
    assert_same_topk(Pl, P, exact_scores=True, what=f"rank {rank} predict_shard")
    full = om.predict(X, beam_size=10, only_topk=10)
    assert_same_topk(P, full, exact_scores=True, what=f"rank {rank}")
    # uneven shards around the "two parts" threshold (4 rows): every rank must issue the same collectives
    for n in (7, 5, 9):
        Xn = X[:n]
        for b in ([0, 4, n], [0, n - 4, n], [0, 3, n]):
            bb = np.asarray(b, dtype=np.int64)
            Pn = sh.predict_shard(take_rows(Xn, int(bb[rank]), int(bb[rank + 1])), bb, beam_size=5, only_topk=4)
            assert_same_topk(Pn, om.predict(Xn, beam_size=5, only_topk=4), exact_scores=True, what=f"rank {rank} uneven shards {b}")
    # ragged: more ranks than useful rows on one side (tiny X)
    P2 = sh.predict(X[:1], beam_size=3, only_topk=5)
    assert_same_topk(P2, om.predict(X[:1], beam_size=3, only_topk=5), exact_scores=True, what="one row")
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_predict_world2_gloo(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_packed_topk_uneven_and_empty_shards_two_parts():
    # PackedTopk's cut / padding algebra without a process group: three ranks with uneven shards (one of them EMPTY), two parts
    # per shard; every rank's send buffers are placed into rank 0's receive buffers by hand, unpack must return the rows in
    # global order with labels, scores (bit patterns) and row lengths intact
    import torch
    from pecos_amd.distributed import PackedTopk
    rng = np.random.default_rng(5)
    k = 7
    for sizes in ([5, 0, 9], [1, 1, 1], [0, 0, 4], [8, 3, 2]):
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        n = int(bounds[-1])
        idx = torch.from_numpy(rng.integers(0, 1 << 20, (n, k)).astype(np.int32))
        val = torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32))
        val[0, 0] = float("-0.0")
        cnt = torch.from_numpy(rng.integers(0, k + 1, n).astype(np.int32))
        for parts in (1, 2):
            pks = [PackedTopk(bounds, r, k, torch.device("cpu"), parts=parts) for r in range(3)]
            for r, pk in enumerate(pks):
                lo, hi = int(bounds[r]), int(bounds[r + 1])
                assert [pk.rows(p) for p in range(parts)][0][0] == 0 and pk.rows(parts - 1)[1] == hi - lo
                pk.store(idx[lo:hi], val[lo:hi], cnt[lo:hi])
                for p in range(parts):
                    pk.gather(p)                                   # no process group: folds cnt into the packed rows
            for p in range(parts):
                for r in range(3):
                    pks[0].gathered[p][r].copy_(pks[r].buf[p])      # what the all-gather would deliver
            gi, gv, gc = pks[0].unpack()
            assert torch.equal(gi, idx) and torch.equal(gv.view(torch.int32), val.view(torch.int32)) and torch.equal(gc, cnt), (sizes, parts)


def _pipeline_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from pecos_amd.distributed import GatherPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = 3
    cases = ([5, 2], [0, 4], [3, 3]) if world == 2 else \
            ([7, 0, 3, 5, 1, 0, 9, 2], [0, 0, 0, 4, 0, 0, 0, 0], [6] * 8, [61250] * 7 + [61251])   # N = 8: uneven, empty, even, and the driver's Amazon-670K shard sizes
    cases = [c for c in cases if len(c) == world]
    for sizes in cases:                                         # uneven and empty shards
        bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n_all, lo, hi = int(bounds[-1]), int(bounds[rank]), int(bounds[rank + 1])
        pipe = GatherPipeline(bounds, rank, k, torch.device("cpu"), n_buf=2)       # CPU: no streams, the gather runs inside end()
        seen = []
        for step in range(5):
            pk = pipe.begin()
            seen.append(id(pk))
            # what this rank's kernels would write in this step: labels / scores that encode (step, global row, slot)
            rows = torch.arange(lo, hi, dtype=torch.int32)[:, None]
            idx = (step * 1000 + rows * 10 + torch.arange(k, dtype=torch.int32)[None, :]).to(torch.int32)
            val = idx.to(torch.float32) * 0.5
            cnt = torch.full((hi - lo,), (step % k) + 1, dtype=torch.int32)
            pk.store(idx, val, cnt)
            pipe.end()
            gi, gv, gc = pipe.last().unpack()                  # every rank holds every shard's rows of THIS step
            want = step * 1000 + torch.arange(n_all, dtype=torch.int32)[:, None] * 10 + torch.arange(k, dtype=torch.int32)[None, :]
            assert torch.equal(gi, want.to(torch.int32)) and torch.equal(gv, want.to(torch.float32) * 0.5), (sizes, step, rank)
            assert torch.equal(gc, torch.full((n_all,), (step % k) + 1, dtype=torch.int32))
        assert seen[0] != seen[1] and seen[0] == seen[2] == seen[4] and seen[1] == seen[3]      # two buffers taking turns
    open(os.path.join(out_dir, f"pipe_ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gather_pipeline_world2_gloo(tmp_path):
    # bench.py's N > 1 step: two result buffers take turns, the all-gather of a step is issued when the step's kernels are queued
    # (on the GPU: on a second stream under the next step's kernels; here synchronously) -- same collectives in the same order on every rank
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "pipe_ok0") and os.path.exists(tmp_path / "pipe_ok1")


@pytest.mark.timeout(600)
def test_gather_pipeline_world8_gloo(tmp_path):
    # the exact collective sequence of `bench.py --gpus 8` (SCALE's last point): 8 ranks, nnz-balanced shard bounds of unequal sizes (some
    # empty), double-buffered PackedTopk, one all-gather per step issued in step order on every rank -- executed once here under gloo, since
    # no 8-GPU node is available to the builder (RCCL itself has only run with one rank: DESIGN.md section 6, "unmeasured")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f"pipe_ok{r}") for r in range(8))
