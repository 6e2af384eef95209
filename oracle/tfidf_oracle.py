"""TEST INFRASTRUCTURE -- CPU restatement of the reference's TF-IDF *predict* path, the checker for pecos_amd's host half
(tokenizer + n-gram lookup -> term counts) and, in numpy float32, for the weighting the device kernel K5 performs.  Only tests/ may
import this; nothing under pecos_amd/ does.

Restated from pecos/core/utils/tfidf.hpp (behaviour, not code):
  Tokenizer::load              :363-386   vocab.txt: first line = size, then "<idx>\\t<token>" (a repeated token keeps the LAST index)
  Tokenizer::split_into_tokens :389-429   word: pieces between ' ' (0x20 only), empty pieces dropped; char / char_wb: UTF-8 code points by
                                          lead byte (>= 0xf0: 4, >= 0xe0: 3, >= 0xc0: 2, < 0x80: 1, else "not utf-8 encoded")
  Tokenizer::tokenize          :433-448   the first max_length tokens when max_length > 0; unknown token -> -1
  BaseVectorizer::load         :707-745   tfidf-model.txt: "<n>" then per feature "<id> <idf> <len> <tok>*len"
  get_sorted_feature           :775-822   n-grams of min_ngram..min(max_ngram, #tokens) tokens looked up, a float count per feature (+= 1.0),
                                          ascending ids; binary -> 1; sublinear -> log(v) + 1.0; x idf; l1 / l2 norm summed in that order
  Vectorizer::load / predict   :1247-1266, :1405-1430; normalize_csr :1318-1354   ensembles: hstack + one more normalisation

PINNED: tests/test_tfidf.py::test_oracle_restatement_vs_reference_goldens holds it bit for bit (sublinear tf: <= 1 ulp, numpy's logf vs
glibc's) against tests/golden/tfidf_models/*/X.npz, which the reference itself produced (tests/golden/make_golden_r04.py).
"""
import json
import os

import numpy as np

F32 = np.float32


class BaseOracle:
    def __init__(self, folder):
        tc = json.load(open(os.path.join(folder, "tokenizer", "config.json")))
        self.tok_type = int(tc["token_type"])
        self.vocab = {}
        with open(os.path.join(folder, "tokenizer", "vocab.txt"), "rb") as f:
            f.readline()
            for line in f.read().split(b"\n"):
                if not line:
                    continue
                idx, tok = line.split(b"\t", 1)
                self.vocab[tok] = int(idx)
        kw = json.load(open(os.path.join(folder, "vectorizer", "config.json")))["kwargs"]
        self.min_ngram, self.max_ngram = (int(v) for v in kw["ngram_range"])
        self.max_length = int(kw["max_length"])
        self.binary, self.use_idf, self.sublinear_tf = bool(kw["binary"]), bool(kw["use_idf"]), bool(kw["sublinear_tf"])
        self.norm_p = {"l1": 1, "l2": 2}[kw["norm_p"]]
        words = open(os.path.join(folder, "vectorizer", "tfidf-model.txt")).read().split()
        total = int(words[0])
        self.idf = np.zeros(total, dtype=F32)
        self.feature_vocab = {}
        at = 1
        for _ in range(total):
            fid, idf, n = int(words[at]), F32(words[at + 1]), int(words[at + 2])
            self.feature_vocab[tuple(int(w) for w in words[at + 3: at + 3 + n])] = fid
            self.idf[fid] = idf
            at += 3 + n
        self.nr_features = total

    def tokens(self, doc):
        if self.tok_type == 10:
            pieces = [p for p in doc.split(b" ") if p]
        else:
            pieces, i = [], 0
            while i < len(doc):
                c = doc[i]
                n = 4 if c >= 0xF0 else 3 if c >= 0xE0 else 2 if c >= 0xC0 else 1 if c < 0x80 else None
                if n is None:
                    raise ValueError("the string is not utf-8 encoded!")
                pieces.append(doc[i:i + n]); i += n
        if self.max_length > 0:
            pieces = pieces[: self.max_length]
        return [self.vocab.get(p, -1) for p in pieces]

    def counts(self, doc):
        """[(feature id, count)] ascending."""
        t = self.tokens(doc)
        c = {}
        for n in range(self.min_ngram, min(self.max_ngram, len(t)) + 1):
            for i in range(len(t) - n + 1):
                f = self.feature_vocab.get(tuple(t[i:i + n]))
                if f is not None:
                    c[f] = c.get(f, 0) + 1
        return sorted(c.items())

    def weights(self, pairs):
        """float32, operation by operation (:798-822)."""
        vals = []
        denom = F32(0.0)
        for f, cnt in pairs:
            v = F32(1.0) if self.binary else F32(cnt)
            if self.sublinear_tf:
                v = F32(np.float64(np.log(v, dtype=F32)) + 1.0)
            if self.use_idf:
                v = F32(v * self.idf[f])
            denom = F32(denom + (F32(abs(v)) if self.norm_p == 1 else F32(v * v)))
            vals.append(v)
        return _normalise(vals, denom, self.norm_p, already_summed=True)


def _normalise(vals, denom, norm_p, already_summed=False):
    if not already_summed:
        denom = F32(0.0)
        for v in vals:
            denom = F32(denom + (F32(abs(v)) if norm_p == 1 else F32(v * v)))
    if abs(denom) < np.finfo(F32).eps:
        denom = F32(1.0)
    elif norm_p == 2:
        denom = F32(np.sqrt(denom))
    return [F32(v / denom) for v in vals]


class TfidfOracle:
    def __init__(self, folder):
        meta = os.path.join(folder, "meta.json")
        if os.path.exists(meta):
            kw = json.load(open(meta))["kwargs"]
            self.base = [BaseOracle(os.path.join(folder, f"{i}.base")) for i in range(int(kw["num_base_vect"]))]
            self.norm_p = int(kw["norm_p"])
        else:
            self.base = [BaseOracle(folder)]
            self.norm_p = self.base[0].norm_p
        self.nr_features = sum(b.nr_features for b in self.base)

    def _rows(self, corpus, weighted):
        indptr, idx, val = [0], [], []
        for doc in corpus:
            doc = doc.encode("utf-8") if isinstance(doc, str) else bytes(doc)
            off, row_i, row_v = 0, [], []
            for b in self.base:
                pairs = b.counts(doc)
                row_i += [off + f for f, _ in pairs]
                row_v += b.weights(pairs) if weighted else [F32(c) for _, c in pairs]
                off += b.nr_features
            if weighted and (len(self.base) > 1 or self.norm_p != self.base[0].norm_p):
                row_v = _normalise(row_v, None, self.norm_p)
            idx += row_i; val += row_v; indptr.append(len(idx))
        return np.array(indptr, dtype=np.uint64), np.array(idx, dtype=np.uint32), np.array(val, dtype=F32)

    def counts(self, corpus):
        """(indptr, indices, data) of the hstacked term-count CSR."""
        return self._rows(corpus, False)

    def predict(self, corpus):
        """(indptr, indices, data) of the reference's c_tfidf_predict output."""
        return self._rows(corpus, True)
