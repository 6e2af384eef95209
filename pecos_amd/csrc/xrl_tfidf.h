// TF-IDF query producer: the reference's `pecos::tfidf::Vectorizer` predict path (pecos/core/utils/tfidf.hpp:297-492
// tokenizer, :707-745 model files, :775-822 get_sorted_feature, :1212-1466 ensemble) split where the hardware suggests: tokenisation and
// the n-gram -> feature lookup are string work and run on host threads; their output -- a CSR of term COUNTS -- goes to the device
// once, and the weighting / normalisation (K5, xrl_features.hip) leaves X in HBM for the beam search.  No D2H / H2D of X.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "xrl_common.h"

namespace xrl {

// one BaseVectorizer folder (tokenizer/{config.json,vocab.txt}, vectorizer/{config.json,tfidf-model.txt})
struct TfidfBase {
    int tok_type = 10;                       // 10 word, 20 char, 30 char_wb (tfidf.hpp:281-285)
    int min_ngram = 1, max_ngram = 1, max_length = -1, norm_p = 2;
    bool binary = false, use_idf = true, sublinear_tf = false;
    std::unordered_map<std::string, int32_t> vocab;            // token -> token index
    // n-gram of token indices -> feature id; the key is the n-gram's int32 sequence as raw bytes
    std::unordered_map<std::string, uint32_t> feature_vocab;
    std::vector<float> idf;                                    // [nr_features]; features the model file does not list keep 0 (the reference's .at() would throw)
    std::vector<uint8_t> idf_known;
    uint32_t nr_features = 0;                                  // = idx_idf.size() (the reference's column count, :1153)

    void load(const std::string& dir);
    // term counts of one document: ascending feature ids (tfidf.hpp:775-796)
    void count(const char* doc, size_t len, std::vector<std::pair<uint32_t, float>>& out, std::vector<int32_t>& tok_scratch, std::string& key_scratch) const;
};

struct TfidfVectorizer {
    std::vector<TfidfBase> base;
    int norm_p = 2;                          // the ensemble's norm (meta.json); == base[0].norm_p for a folder saved from one BaseVectorizer
    uint32_t nr_features = 0;                // sum over the base vectorizers (hstack)

    void load(const std::string& dir);       // Vectorizer::load, tfidf.hpp:1247-1266
    // Host half for a corpus: per (document, base vectorizer) SEGMENT the term counts, laid out as the hstacked CSR (document-major,
    // base vectorizers side by side, column ids offset).  seg_ptr has nr_doc * base.size() + 1 entries.
    void count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr,
                      std::vector<uint32_t>& col_idx, std::vector<float>& cnt) const;
};

}  // namespace xrl
