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
