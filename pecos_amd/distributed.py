"""Query-sharded multi-GPU prediction: one process per GPU, model replicated, rows of X split into
contiguous nnz-balanced shards, ONE all-gather of the fixed-stride top-k at the end
(RCCL over xGMI through ``torch.distributed``'s "nccl" backend; "gloo" on CPU for tests).

The reference has no multi-process inference at all (only sequential
``max_pred_chunk`` row chunking, pecos/xmc/xlinear/model.py:532-548); this module is what the
MI355X build adds.  Every stage of the beam search is row-local, so there is no data-path
collective until the final gather: payload = rows x k x 8 B + rows x 4 B.

torch is imported lazily and BEFORE the HIP library is first touched, so that both share one HIP
runtime in the process.
"""
import numpy as np
import scipy.sparse as smat


def shard_bounds(X, world_size):
    """Contiguous row ranges with ~equal work: cost ~ nnz (+1 per row so empty rows still count).
    Returns an int64 array of world_size+1 row boundaries."""
    n = X.shape[0]
    if smat.issparse(X):
        cost = np.diff(X.indptr).astype(np.int64) + 1
    else:
        cost = np.ones(n, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(cost)])
    targets = csum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(csum, targets, side="left")
    return np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)


def take_rows(X, lo, hi):
    if smat.issparse(X):
        S = X[lo:hi]
        S.sort_indices()
        return S
    return np.ascontiguousarray(X[lo:hi])


def rows_to_csr(idx, val, cnt, n_cols):
    """Fixed-stride (rows x k) results -> CSR with score-sorted rows (the reference's output form)."""
    idx = np.asarray(idx); val = np.asarray(val); cnt = np.asarray(cnt).astype(np.int64)
    n, k = idx.shape
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    mask = np.arange(k)[None, :] < cnt[:, None]
    return smat.csr_matrix((val[mask].astype(np.float32), idx[mask].astype(np.int64), indptr), shape=(n, n_cols))


def all_gather_topk(idx, val, cnt, bounds, group=None):
    """All-gather per-rank fixed-stride results (torch tensors on the backend's device).

    idx: int32/uint32-as-int32 [rows_r, k], val: float32 [rows_r, k], cnt: int32 [rows_r];
    ``bounds`` are the global row boundaries, identical on every rank.  Shards are padded to the
    largest shard so that one ``all_gather_into_tensor`` per array suffices.  Returns global
    (idx, val, cnt) tensors in row order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = np.diff(bounds)
    k = idx.shape[1]
    maxr = int(sizes.max()) if len(sizes) else 0
    dev = idx.device

    def pad(t, cols):
        out = torch.zeros((maxr, cols) if cols else (maxr,), dtype=t.dtype, device=dev)
        out[: t.shape[0]] = t
        return out

    outs = []
    for t, cols in ((idx, k), (val, k), (cnt, 0)):
        send = pad(t, cols).contiguous()
        recv = torch.empty((world,) + tuple(send.shape), dtype=t.dtype, device=dev)
        try:
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
        except (RuntimeError, NotImplementedError):  # backends without the fused form
            parts = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(parts, send, group=group)
            recv = torch.stack(parts)
        outs.append(torch.cat([recv[r, : int(sizes[r])] for r in range(world)], dim=0))
    return tuple(outs)


class ShardedXLinear:
    """Wraps a loaded :class:`pecos_amd.XLinearModel` for data-parallel prediction.

    ``predict_shard_fn`` is the per-rank compute; the default runs the HIP library on this rank's
    GPU with inputs resident in HBM.  (tests inject a CPU stand-in to exercise the shard/gather
    algebra under gloo -- the product default has no CPU path.)"""

    def __init__(self, model, group=None, predict_shard_fn=None):
        import torch  # noqa: F401  (first, so the HIP runtime is shared)
        import torch.distributed as dist
        self.model = model
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._fn = predict_shard_fn or self._predict_shard_gpu

    def _predict_shard_gpu(self, Xs, beam_size, only_topk, post_processor):
        import torch
        from .core import clib
        h = self.model.model.model_chain
        k = clib.effective_topk(h, only_topk)
        n = Xs.shape[0]
        dev = torch.device("cuda", torch.cuda.current_device())
        idx = torch.zeros((n, k), dtype=torch.int32, device=dev)
        val = torch.zeros((n, k), dtype=torch.float32, device=dev)
        cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
        if n:
            q = clib.queries_upload(h, Xs)
            try:
                clib.predict_device(h, q, beam_size, post_processor, only_topk, idx.data_ptr(), val.data_ptr(),
                                    cnt.data_ptr(), k, stream=torch.cuda.current_stream().cuda_stream, sync=True)
            finally:
                clib.queries_free(q)
        return idx, val, cnt

    def predict(self, X, beam_size=None, only_topk=None, post_processor=None):
        """X: the FULL query matrix (identical on every rank).  Returns the full CSR on every rank."""
        bounds = shard_bounds(X, self.world)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        idx, val, cnt = self._fn(take_rows(X, lo, hi), beam_size, only_topk, post_processor)
        if self.world > 1:
            idx, val, cnt = all_gather_topk(idx, val, cnt, bounds, self.group)
        return rows_to_csr(idx.cpu().numpy().view(np.uint32), val.cpu().numpy(), cnt.cpu().numpy(), self.model.nr_pred_cols)
