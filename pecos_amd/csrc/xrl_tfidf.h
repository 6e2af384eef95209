// TF-IDF query producer: the reference's `pecos::tfidf::Vectorizer` predict path (pecos/core/utils/tfidf.hpp:297-492
// tokenizer, :707-745 model files, :775-822 get_sorted_feature, :1212-1466 ensemble) split where the hardware suggests: tokenisation and
// the n-gram -> feature lookup are string work and run on host threads; their output -- a CSR of term COUNTS -- goes to the device
// once, and the weighting / normalisation (K5, xrl_features.hip) leaves X in HBM for the beam search.  No D2H / H2D of X.
//
// The host half is the slowest stage of text -> labels by two orders of magnitude (10^5..10^6 documents/s against the beam search's
// 10^7..10^8 queries/s), so its containers are built for the lookup loop rather than for generality: flat open-addressing tables with the
// key INLINE where it fits in 8 bytes (tokens of <= 8 bytes; unigrams as a direct array; n-grams packed into one u64), two passes per document
// (hash + prefetch every slot, then probe) so the cache misses of a multi-million-entry feature table overlap, and documents handed to
// the threads in small dynamic chunks (profiles/r04_tfidf_host.md).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "xrl_common.h"

namespace xrl {

// A zero-filled table on its own pages.  Past 1 MiB it is 2 MiB-aligned and marked MADV_HUGEPAGE: the lookups are random reads over tens
// of MB, and with 4 KiB pages every one of them (and every software prefetch) is a TLB miss first.
template <class T> struct HugeArray {
    T* p = nullptr; size_t n = 0;
    HugeArray() = default;
    HugeArray(const HugeArray&) = delete;
    HugeArray& operator=(const HugeArray&) = delete;
    HugeArray(HugeArray&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    HugeArray& operator=(HugeArray&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~HugeArray() { release(); }
    void release();
    void assign_zero(size_t count);
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// token bytes -> token index (tokenizer/vocab.txt)
struct TokenTable {
    struct Short { uint64_t key; uint32_t len; int32_t idx; };                 // tokens of 1..8 bytes: the bytes themselves, zero-extended; len 0 = empty slot
    struct Long { uint64_t hash; uint32_t off, len; int32_t idx; uint32_t pad; };   // longer tokens: hash, bytes in `arena`
    HugeArray<Short> s;
    HugeArray<Long> l;
    std::string arena;
    unsigned s_shift = 63, l_shift = 63;                                       // slot = hash >> shift
    size_t n_short = 0, n_long = 0;

    void build(const std::vector<std::pair<std::string, int32_t>>& items);     // later duplicates overwrite earlier ones (the reference's map assignment)
    static inline uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; return x ^ (x >> 29); }
    static uint64_t hash_long(const char* p, size_t n);
    inline size_t short_slot(uint64_t key, uint32_t len) const { return (size_t)(mix(key ^ ((uint64_t)len << 56) ^ 0x5bd1e995u) * 0xD6E8FEB86659FD93ull >> s_shift); }
    inline int32_t find_short(uint64_t key, uint32_t len, size_t slot) const {
        const size_t mask = s.size() - 1;
        for (;; slot = (slot + 1) & mask) {
            const Short& e = s[slot];
            if (e.len == 0) return -1;
            if (e.key == key && e.len == len) return e.idx;
        }
    }
    int32_t find_long(const char* p, size_t n, uint64_t h) const;
};

// n-gram of token indices -> feature id (vectorizer/tfidf-model.txt).  Three homes, by what the key needs:
//   uni      unigrams, a direct array over the token index
//   packed   n-grams whose token indices are all >= 0 and fit one u64 at pack_bits each (field = index + 1, so the length is implied):
//            EVERY bigram, and e.g. up to 9-grams of a 100-character vocabulary or 4-grams of a 60k-word one -- key compared inline
//   gen      the rest (longer n-grams, n-grams naming the unknown token -1): hash + token ids in an arena
struct NgramTable {
    static constexpr uint32_t kNone = 0;                                       // stored ids are feature id + 1
    std::vector<uint32_t> uni;
    struct Packed { uint64_t key; uint32_t id1; uint32_t pad; };
    HugeArray<Packed> packed;
    struct Gen { uint64_t hash; uint32_t off, n, id1, pad; };
    HugeArray<Gen> gen;
    std::vector<int32_t> arena;
    unsigned p_shift = 63, g_shift = 63;
    size_t n_packed = 0, n_gen = 0;
    int32_t max_tok = -1;                                                      // the largest token index any n-gram names
    unsigned pack_bits = 1;                                                    // bit width of max_tok + 1
    int pack_max_n = 0;                                                        // 64 / pack_bits
    uint64_t packed_n_mask = 0, gen_n_mask = 0;                                // bit min(n, 63): that table holds n-grams of n tokens
    bool negative_keys = false;                                                // some n-gram names a negative token index: unknown tokens cannot be skipped
    int max_n = 0;

    void build(const std::vector<int32_t>& flat, const std::vector<uint64_t>& off, const std::vector<uint32_t>& ids, size_t vocab_hint);
    static inline uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; return x ^ (x >> 29); }
    static inline uint64_t n_bit(int n) { return 1ull << (n < 63 ? n : 63); }
    inline uint64_t pack(const int32_t* t, int n) const {
        uint64_t k = 0;
        for (int i = 0; i < n; ++i) k |= (uint64_t)((uint32_t)t[i] + 1u) << (pack_bits * (unsigned)i);
        return k;
    }
    inline size_t packed_slot(uint64_t key) const { return (size_t)(mix(key) * 0xD6E8FEB86659FD93ull >> p_shift); }
    static inline uint64_t gen_hash(const int32_t* t, int n) {
        uint64_t h = 0x2545F4914F6CDD1Dull ^ (uint64_t)n;
        for (int i = 0; i < n; ++i) h = mix(h ^ (uint32_t)t[i]) + 0x9E3779B97F4A7C15ull;
        return h * 0xD6E8FEB86659FD93ull;
    }
    inline size_t gen_slot(uint64_t h) const { return (size_t)(h >> g_shift); }
    inline uint32_t find_packed(uint64_t key, size_t slot) const {
        const size_t mask = packed.size() - 1;
        for (;; slot = (slot + 1) & mask) {
            const Packed& e = packed[slot];
            if (e.id1 == kNone) return kNone;
            if (e.key == key) return e.id1;
        }
    }
    uint32_t find_gen(const int32_t* t, int n, uint64_t h) const;
};

// per-thread scratch of the counting loop
struct TfidfScratch {
    // (sized for the longest document seen so far, never cleared)
    std::vector<uint64_t> key;      // per token: inline key or hash; per n-gram: key / hash
    std::vector<uint32_t> len;      // per token: byte length
    std::vector<uint64_t> aux;      // per token: the table slot (tokens of <= 8 bytes) or the offset of its first byte in the document (longer ones)
    std::vector<int32_t> tok;       // token indices
    std::vector<int32_t> run;       // tokens in a row from here on that some n-gram of the model names (0 <= index <= max_tok)
    std::vector<uint32_t> feat;     // feature ids found (one per occurrence), then the same sorted
    // counting without a sort (models of up to dense_limit features): a counter per feature and a three-level bitmap of the ids touched
    bool dense_ok = false;
    std::vector<uint32_t> dense;
    std::vector<uint64_t> bits0, bits1, bits2;
    void prepare(uint32_t max_features, size_t dense_limit);
};

// a thread's output: (column, count) pairs appended document after document
struct TfidfOut {
    uint32_t* col = nullptr; float* val = nullptr; size_t n = 0, cap = 0;
    TfidfOut() = default;
    TfidfOut(const TfidfOut&) = delete;
    TfidfOut& operator=(const TfidfOut&) = delete;
    ~TfidfOut();
    void ensure(size_t extra);
};

// one BaseVectorizer folder (tokenizer/{config.json,vocab.txt}, vectorizer/{config.json,tfidf-model.txt})
struct TfidfBase {
    int tok_type = 10;                       // 10 word, 20 char, 30 char_wb (tfidf.hpp:281-285)
    int min_ngram = 1, max_ngram = 1, max_length = -1, norm_p = 2;
    bool binary = false, use_idf = true, sublinear_tf = false;
    TokenTable vocab;
    NgramTable features;
    std::vector<float> idf;                                    // [nr_features]
    uint32_t nr_features = 0;                                  // = idx_idf.size() (the reference's column count, :1153)
    unsigned sort_shift = 0;                                   // feature id >> sort_shift in [0, 256): the bucket pass of the per-document sort

    void load(const std::string& dir);
    // term counts of one document APPENDED to out: ascending feature ids + col_off, counts as floats (tfidf.hpp:775-796); returns how many
    size_t count(const char* doc, size_t len, TfidfScratch& S, uint32_t col_off, TfidfOut& out) const;
};

struct TfidfVectorizer {
    std::vector<TfidfBase> base;
    int norm_p = 2;                          // the ensemble's norm (meta.json); == base[0].norm_p for a folder saved from one BaseVectorizer
    uint32_t nr_features = 0;                // sum over the base vectorizers (hstack)

    void load(const std::string& dir);       // Vectorizer::load, tfidf.hpp:1247-1266
    // Host half for a corpus: per (document, base vectorizer) SEGMENT the term counts, laid out as the hstacked CSR (document-major,
    // base vectorizers side by side, column ids offset).  seg_ptr has nr_doc * base.size() + 1 entries.
    // The CALLER owns the destination: once the total is known `provide(nnz, col, cnt)` is called (on the calling thread) and must hand back
    // two arrays of nnz elements; the worker threads then write their pieces into them in parallel -- first touch included, which is what
    // made a std::vector result the slowest phase of the call (17 of 25 ms for 200 k documents: zero fill of fresh pages on one thread).
    // The allocator callback's numpy arrays and a pinned staging buffer for the H2D copy are such destinations.
    using Provide = std::function<void(uint64_t nnz, uint32_t*& col, float*& cnt)>;
    void count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr, const Provide& provide) const;
    // (the same into vectors: tests, small corpora)
    void count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr,
                      std::vector<uint32_t>& col_idx, std::vector<float>& cnt) const;
};

}  // namespace xrl
