"""Query-sharded multi-GPU prediction: one process per GPU, model replicated, rows of X split into
contiguous nnz-balanced shards, ONE all-gather of the packed fixed-stride top-k rows at the end
(RCCL over xGMI through ``torch.distributed``'s "nccl" backend; "gloo" on CPU for tests).

The reference has no multi-process inference at all (only sequential
``max_pred_chunk`` row chunking, pecos/xmc/xlinear/model.py:532-548); this module is what the
MI355X build adds.  Every stage of the beam search is row-local, so there is no data-path
collective until the final gather: payload = rows x (8k + 4) B.  bench.py and ShardedXLinear share PackedTopk.

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


class PackedTopk:
    """Fixed-stride result rows ``[idx(k) | val(k) | cnt]`` of one rank's shard held as int32 tensors, so that ONE
    ``all_gather_into_tensor`` per PART moves labels, scores and row lengths together (shards are padded to the largest).

    ``parts`` = 1: the whole shard is one part.  ``parts`` = 2: every rank's rows are cut in two halves, each with its own send /
    receive buffer, so that the gather of the first half can run (on another stream) under the kernels of the second half.
    The compute writes labels / scores through :meth:`pointers` (row stride 2k+1, ABSOLUTE local row indexing) and row lengths
    into ``cnt``; :meth:`gather` folds ``cnt`` into column 2k and runs the collective on the CURRENT stream of the backend's
    device, :meth:`unpack` cuts the padding away."""

    def __init__(self, bounds, rank, k, device, parts=1):
        import torch
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.sizes = np.diff(self.bounds)
        self.world, self.rank, self.k, self.parts = len(self.sizes), int(rank), int(k), int(parts)
        # part p of rank r = its local rows [cut[r][p], cut[r][p+1])
        self.cut = [[int(n * p // self.parts) for p in range(self.parts)] + [int(n)] for n in self.sizes]
        self.max_part = [max(max(c[p + 1] - c[p] for c in self.cut), 1) for p in range(self.parts)]
        self.buf = [torch.zeros((self.max_part[p], 2 * self.k + 1), dtype=torch.int32, device=device) for p in range(self.parts)]
        self.gathered = [torch.empty((self.world, self.max_part[p], 2 * self.k + 1), dtype=torch.int32, device=device) for p in range(self.parts)]
        self.cnt = torch.zeros((max(int(self.sizes.max()), 1),), dtype=torch.int32, device=device)

    def rows(self, part):
        """local row range [begin, end) of this rank's part"""
        c = self.cut[self.rank]
        return c[part], c[part + 1]

    def pointers(self, part=0):
        """(idx_ptr, val_ptr, cnt_ptr, row_stride) for ``clib.predict_device[_rows]``: addresses such that ABSOLUTE local row r of the
        part lands in row r - begin of the part's buffer."""
        begin = self.cut[self.rank][part]
        base = self.buf[part].data_ptr() - begin * (2 * self.k + 1) * 4
        return base, base + 4 * self.k, self.cnt.data_ptr(), 2 * self.k + 1

    def store(self, idx, val, cnt):
        """Fill from separate tensors (CPU stand-ins in tests): idx int32 [n,k], val float32 [n,k], cnt int32 [n]."""
        import torch
        for p in range(self.parts):
            b, e = self.rows(p)
            self.buf[p][: e - b, : self.k] = idx[b:e]
            self.buf[p][: e - b, self.k: 2 * self.k] = val[b:e].contiguous().view(torch.int32)
        self.cnt[: idx.shape[0]] = cnt

    def gather(self, part=0, group=None):
        import torch
        import torch.distributed as dist
        b, e = self.rows(part)
        self.buf[part][: e - b, 2 * self.k] = self.cnt[b:e]
        if not dist.is_initialized():          # single process without a process group: nothing to exchange
            self.gathered[part][0].copy_(self.buf[part])
            return
        if self.buf[part].is_cuda and dist.get_backend(group) == "gloo":
            # host-staged exchange (the N > 1 rehearsal of bench.py on one GPU; gloo gathers CPU tensors only): the copy waits for
            # everything queued on the current stream, i.e. for the step's kernels
            host = self.buf[part].cpu()
            chunks = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(chunks, host, group=group)
            self.gathered[part].copy_(torch.stack(chunks))
            return
        try:                                   # (a one-rank group still runs the collective: tests exercise RCCL that way)
            dist.all_gather_into_tensor(self.gathered[part], self.buf[part], group=group)
        except (RuntimeError, NotImplementedError):  # backends without the fused form
            chunks = [torch.empty_like(self.buf[part]) for _ in range(self.world)]
            dist.all_gather(chunks, self.buf[part], group=group)
            self.gathered[part].copy_(torch.stack(chunks))

    def unpack(self):
        """-> global (idx int32 [N,k], val float32 [N,k], cnt int32 [N]) in row order (after every part has been gathered)."""
        import torch
        pieces = []
        for r in range(self.world):
            for p in range(self.parts):
                n = self.cut[r][p + 1] - self.cut[r][p]
                pieces.append(self.gathered[p][r, :n])
        rows = torch.cat(pieces, dim=0)
        return rows[:, : self.k].contiguous(), rows[:, self.k: 2 * self.k].contiguous().view(torch.float32), rows[:, 2 * self.k].contiguous()


class GatherPipeline:
    """Back-to-back predict steps of one rank with the all-gather of step s running under the kernels of step s+1.

    ``n_buf`` PackedTopk buffers take turns: :meth:`begin` hands out the buffer of the coming step (and makes the compute stream wait
    until the gather that last read that buffer has finished), the caller queues its kernels on the compute stream writing through
    ``buffer.pointers()``, :meth:`end` queues the buffer's all-gather on the gather stream behind an event recorded on the compute
    stream.  Every rank must call begin/end the same number of times (the collectives are issued in step order on every rank).
    Why not hide the gather inside its own step by cutting the shard in halves: a rank's launches are latency-bound at a shard's size
    (61 k Amazon-670K rows: 0.96 ms; two halves: 2 x 0.70 ms -- profiles/r03_pruning_topk.md section 3).

    ``compute_stream`` / ``gather_stream``: torch.cuda.Stream objects, or None on CPU (tests under gloo): then the gather runs
    synchronously inside :meth:`end`.  ``gather=False`` (single process without a process group) only rotates the buffers."""

    def __init__(self, bounds, rank, k, device, n_buf=2, compute_stream=None, gather_stream=None, group=None, gather=True):
        import torch
        self.n_buf, self.group, self.gather = int(n_buf), group, bool(gather)
        self.cs, self.gs = compute_stream, gather_stream
        self.bufs = [PackedTopk(bounds, rank, k, device, parts=1) for _ in range(self.n_buf)]
        on_gpu = self.cs is not None and self.gs is not None
        self.ev_done = [torch.cuda.Event() for _ in range(self.n_buf)] if on_gpu else None   # step's kernels queued (compute stream)
        self.ev_free = [torch.cuda.Event() for _ in range(self.n_buf)] if on_gpu else None   # step's gather finished (gather stream)
        self.step = 0

    def begin(self):
        i = self.step % self.n_buf
        if self.ev_free is not None and self.gather and self.step >= self.n_buf:
            self.cs.wait_event(self.ev_free[i])              # the gather of step - n_buf has left this buffer
        return self.bufs[i]

    def end(self):
        import torch
        i = self.step % self.n_buf
        if self.gather:
            if self.ev_done is not None:
                self.ev_done[i].record(self.cs)
                with torch.cuda.stream(self.gs):
                    self.gs.wait_event(self.ev_done[i])
                    self.bufs[i].gather(0, self.group)       # under the next step's kernels
                    self.ev_free[i].record(self.gs)
            else:
                self.bufs[i].gather(0, self.group)
        self.step += 1

    def last(self):
        """the buffer of the most recent finished step (its gathered result is complete once the gather stream has been synchronised)"""
        return self.bufs[(self.step - 1) % self.n_buf] if self.step else self.bufs[0]


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
        model_dev = clib.xlinear_get_int_attr(h, "device")
        if model_dev != dev.index:   # the handle lives on the device that was current when it was loaded (clib.set_device)
            raise RuntimeError(f"model handle is on GPU {model_dev} but torch's current device is {dev.index}: call "
                               "pecos_amd.clib.set_device(local_rank) before XLinearModel.load")
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
        return self.predict_shard(take_rows(X, lo, hi), bounds, beam_size, only_topk, post_processor)

    def predict_shard(self, X_local, bounds, beam_size=None, only_topk=None, post_processor=None):
        """X_local: THIS rank's rows only, ``bounds``: the global row boundaries (world+1 ints, identical on every rank;
        e.g. from :func:`shard_bounds`).  One packed all-gather; returns the full CSR on every rank."""
        bounds = np.asarray(bounds, dtype=np.int64)
        if X_local.shape[0] != int(bounds[self.rank + 1] - bounds[self.rank]):
            raise ValueError("X_local does not hold the rows bounds assign to this rank")
        idx, val, cnt = self._fn(X_local, beam_size, only_topk, post_processor)
        if self.world > 1:
            # `parts` must be the same on every rank (it fixes the number and shapes of the collectives): derive it from the
            # GLOBAL bounds, never from this rank's row count (nnz-balanced shards are uneven)
            pk = PackedTopk(bounds, self.rank, idx.shape[1], idx.device, parts=2 if int(np.diff(bounds).min()) >= 4 else 1)
            pk.store(idx, val, cnt)
            for p in range(pk.parts):
                pk.gather(p, self.group)
            idx, val, cnt = pk.unpack()
        return rows_to_csr(idx.cpu().numpy().view(np.uint32), val.cpu().numpy(), cnt.cpu().numpy(), self.model.nr_pred_cols)
