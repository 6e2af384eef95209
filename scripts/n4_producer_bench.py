"""N4 measurement on the GPU box: the TF-IDF query producer end to end, beside the reference's own c_tfidf_predict on the same host cores.

  texts --(host threads: tokenise + n-gram lookup)--> term counts --H2D--> K5 weighting on the device --> X resident in HBM

Legs (documents/s, all host threads):
  reference      oracle/_ref/libpecos_float32.so (the reference's compiled library, built by oracle/Makefile; it travels to the GPU box)
                 c_tfidf_load + c_tfidf_predict through ctypes -- host CSR out
  host_half      xrl_tfidf_counts alone (warm output arrays)
  c_tfidf_predict  pecos_amd's drop-in: host half + H2D + K5 + D2H + allocator (host CSR out, bit-compared with the reference's)
  device         xrl_tfidf_predict_device: host half + H2D + K5, X stays in HBM (what the beam search consumes)

The vectorizer folders are SYNTHETIC (written here in the reference's file format; no training needed, nothing read from /root/reference):
  unigram   135,909 words = Amazon-670K's feature dimension
  bigram    60,000 words + the 1,000,000 most frequent bigrams of a sample of the corpus
Usage: python scripts/n4_producer_bench.py [--docs 300000] [--out gpurun_out/r04_n4_producer.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ALLOC = C.CFUNCTYPE(None, C.c_bool, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p)


def zipf_corpus(rng, words, n_docs, mean_len, s=1.07):
    p = 1.0 / np.arange(1, len(words) + 1) ** s
    p /= p.sum()
    lens = np.maximum(1, rng.poisson(mean_len, size=n_docs))
    ids = rng.choice(len(words), size=int(lens.sum()), p=p).astype(np.int32)
    flat = np.asarray(words)[ids]
    out, o = [], 0
    for n in lens:
        out.append(" ".join(flat[o:o + n])); o += n
    return out, ids, lens


def write_vectorizer(folder, words, grams, rng, norm="l2"):
    """One BaseVectorizer folder under meta.json, the layout Tfidf.save writes (vectorizers.py:163-300 -> c_tfidf_save)."""
    base = os.path.join(folder, "0.base")
    os.makedirs(os.path.join(base, "tokenizer")); os.makedirs(os.path.join(base, "vectorizer"))
    json.dump({"type": "tfidf", "kwargs": {"norm_p": 2 if norm == "l2" else 1, "num_base_vect": 1}}, open(os.path.join(folder, "meta.json"), "w"))
    json.dump({"token_type": 10}, open(os.path.join(base, "tokenizer", "config.json"), "w"))
    with open(os.path.join(base, "tokenizer", "vocab.txt"), "w") as f:
        f.write(f"{len(words)}\n" + "".join(f"{i}\t{w}\n" for i, w in enumerate(words)))
    max_n = max(len(g) for g in grams)
    kw = dict(ngram_range=[1, max_n], max_length=-1, binary=False, use_idf=True, sublinear_tf=False, norm_p=norm, min_df_ratio=0.0, max_df_ratio=1.0,
              min_df_cnt=0, max_df_cnt=-1, add_one_idf=False, keep_frequent_feature=True, smooth_idf=True, max_feature=0)
    json.dump({"type": "tfidf", "kwargs": kw}, open(os.path.join(base, "vectorizer", "config.json"), "w"))
    idf = 1.0 + 8.0 * rng.random(len(grams))
    order = rng.permutation(len(grams))                     # feature ids are not in n-gram order in a trained model either
    with open(os.path.join(base, "vectorizer", "tfidf-model.txt"), "w") as f:
        f.write(f"{len(grams)}\n")
        f.write("".join(f"{order[i]} {idf[i]:.6f} {len(g)} {' '.join(map(str, g))}\n" for i, g in enumerate(grams)))


class Warm:
    def __init__(self):
        self.a = None

    def __call__(self, is_col_major, rows, cols, nnz, indices_pp, indptr_pp, data_pp):
        if self.a is None or len(self.a[0]) != nnz or len(self.a[1]) != rows + 1:
            self.a = (np.zeros(nnz, np.uint32), np.zeros(rows + 1, np.uint64), np.zeros(nnz, np.float32))
        self.shape = (rows, cols)
        for dst, arr in zip((indices_pp, indptr_pp, data_pp), self.a):
            C.cast(dst, C.POINTER(C.c_uint64)).contents.value = arr.ctypes.data


def best(fn, reps):
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=300000)
    ap.add_argument("--mean-len", type=int, default=76)      # Amazon-670K's average nnz per query
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=-1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from pecos_amd.core import clib as amd
    have_gpu = torch.cuda.is_available()
    ref_so = os.path.join(REPO, "oracle", "_ref", "libpecos_float32.so")
    ref = C.CDLL(ref_so) if os.path.exists(ref_so) else None
    if ref is not None:
        ref.c_tfidf_load.restype = C.c_void_p; ref.c_tfidf_load.argtypes = [C.c_char_p]
        ref.c_tfidf_predict.restype = None; ref.c_tfidf_predict.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_int, ALLOC]
        ref.c_tfidf_destruct.argtypes = [C.c_void_p]
    rng = np.random.default_rng(11)
    res = dict(docs=a.docs, host_cores=os.cpu_count(), threads=a.threads, gpu=torch.cuda.get_device_name(0) if have_gpu else None, rows=[])
    tmp = tempfile.mkdtemp(prefix="n4_")
    dev_model = None
    if have_gpu:                                              # any loaded model gives the producer its device and stream
        import xrl_synth
        from pecos_amd import XLinearModel
        mdir = os.path.join(tmp, "ctx")
        xrl_synth.make_model(mdir, 1000, 600, [120, 60, 20], seed=61, shape=[6, 48, 600])
        dev_model = XLinearModel.load(mdir)
    for case in ("unigram", "bigram"):
        V = 135909 if case == "unigram" else 60000
        words = [f"t{i:x}" for i in range(V)]
        corpus, ids, lens = zipf_corpus(rng, words, a.docs, a.mean_len)
        grams = [(i,) for i in range(V)]
        if case == "bigram":
            take = int(min(len(ids) - 1, 8_000_000))
            pair = ids[:take].astype(np.int64) * V + ids[1:take + 1]          # (document boundaries add a few spurious pairs: harmless, they are just features)
            u, c = np.unique(pair, return_counts=True)
            top = u[np.argsort(-c, kind="stable")[:1_000_000]]
            grams += [(int(p // V), int(p % V)) for p in top]
        folder = os.path.join(tmp, case)
        write_vectorizer(folder, words, grams, rng)
        arr, dl, n = amd._corpus_arrays(corpus)
        dlp = dl.ctypes.data_as(C.POINTER(C.c_uint64))
        row = dict(case=case, features=len(grams), docs=n, text_MB=float(dl.sum()) / 1e6)
        h = amd.tfidf_load(folder)
        wa, wb, wr = Warm(), Warm(), Warm()
        fa, fb, fr = ALLOC(wa), ALLOC(wb), ALLOC(wr)
        t = best(lambda: amd.clib_float32.xrl_tfidf_counts(C.c_void_p(h), arr, dlp, n, a.threads, fa), a.reps)
        row.update(nnz=int(len(wa.a[0])), host_half_docs_per_s=n / t, host_half_MB_per_s=row["text_MB"] / t)
        if ref is not None:
            rh = ref.c_tfidf_load(folder.encode())
            t = best(lambda: ref.c_tfidf_predict(C.c_void_p(rh), arr, dlp, n, a.threads, fr), max(2, a.reps // 2))
            row.update(reference_docs_per_s=n / t)
            ref.c_tfidf_destruct(C.c_void_p(rh))
        if have_gpu:
            t = best(lambda: amd.clib_float32.c_tfidf_predict(C.c_void_p(h), arr, dlp, n, a.threads, fb), a.reps)
            amd._check()
            row.update(c_tfidf_predict_docs_per_s=n / t)
            if ref is not None:                               # the drop-in's output against the reference's, the whole corpus
                row.update(pattern_identical=bool(np.array_equal(wb.a[1], wr.a[1]) and np.array_equal(wb.a[0], wr.a[0])),
                           values_bit_identical=bool(np.array_equal(wb.a[2].view(np.uint32), wr.a[2].view(np.uint32))))
            hs = []
            fn = amd.clib_float32.xrl_tfidf_predict_device          # (the native entry point directly: the corpus is already packed)
            cm = C.c_void_p(dev_model.model.model_chain)

            def dev_native():
                q = fn(C.c_void_p(h), cm, arr, dlp, n, a.threads); amd._check(); hs.append(q)
            t = best(dev_native, a.reps)
            for q in hs:
                amd.queries_free(q)
            row.update(device_docs_per_s=n / t)
        amd.tfidf_destruct(h)
        res["rows"].append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
