// Host half of the TF-IDF query producer (see xrl_tfidf.h): model files, tokenizer, n-gram lookup, term counts.
//
// Restated from the reference's behaviour (pecos/core/utils/tfidf.hpp), not from its code layout:
//   Tokenizer::load            :363-386   config.json {"token_type"}, vocab.txt "<n>\n<idx>\t<token>\n..."
//   split_into_tokens          :389-429   word: split on ' ' (empty pieces dropped); char / char_wb: UTF-8 code points, a stray
//                                         continuation byte is an error
//   tokenize                   :433-448   truncate to max_length (> 0), unknown token -> -1
//   BaseVectorizer::load       :707-745   tfidf-model.txt "<n>" then per feature "<id> <idf> <len> <tok>..."
//   get_sorted_feature         :775-796   every n-gram (min_ngram..min(max_ngram, #tokens)) looked up, counts per feature id,
//                                         ascending ids -- NOTE the predict path does not pad char_wb words with spaces (only the
//                                         training path's count_ngrams does, :452-487); this mirrors predict
//   Vectorizer::load / predict :1247-1266, 1405-1430   single folder or meta.json + <i>.base; hstack with column offsets
#include "xrl_tfidf.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <system_error>
#include <thread>

#include <sys/mman.h>

#include "xrl_io.h"

namespace xrl {

namespace {
const JsonValue& need(const JsonValue& o, const char* key, const std::string& where) {
    const JsonValue* v = o.get(key);
    if (!v) fail(where + ": missing key \"" + key + "\"");
    return *v;
}
int as_int(const JsonValue& v, const std::string& where) {
    if (v.type != JsonValue::NUMBER) fail(where + ": expected a number");
    return (int)v.num;
}
bool as_bool(const JsonValue& v, const std::string& where) {
    if (v.type == JsonValue::BOOL) return v.b;
    if (v.type == JsonValue::NUMBER) return v.num != 0;
    fail(where + ": expected a boolean");
}
// table of 2^bits slots for n keys at a load factor <= 1/2 (probe chains stay a slot or two long, misses end at once)
unsigned table_bits(size_t n) {
    unsigned bits = 4;
    while (((size_t)1 << bits) < 2 * n + 2) ++bits;
    return bits;
}
// the first n (1..8) bytes at p, zero-extended; one unaligned load when 8 bytes are readable
inline uint64_t load_key(const char* p, size_t n, const char* last) {
    uint64_t v = 0;
    if (p + 8 <= last) {
        std::memcpy(&v, p, 8);
        return n == 8 ? v : v & ((1ull << (8 * n)) - 1);
    }
    std::memcpy(&v, p, n);
    return v;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------- tables
template <class T> void HugeArray<T>::release() { std::free(p); p = nullptr; n = 0; }
template <class T> void HugeArray<T>::assign_zero(size_t count) {
    release();
    if (!count) return;
    const size_t bytes = count * sizeof(T), huge = (size_t)2 << 20;
    const bool big = bytes >= ((size_t)1 << 20);
    const size_t align = big ? huge : 64, padded = (bytes + align - 1) / align * align;
    void* q = nullptr;
    if (posix_memalign(&q, align, padded) != 0 || !q) fail("tfidf: out of host memory");
    if (big) (void)madvise(q, padded, MADV_HUGEPAGE);          // advisory: the table works without it
    std::memset(q, 0, padded);
    p = static_cast<T*>(q); n = count;
}
template struct HugeArray<TokenTable::Short>;
template struct HugeArray<TokenTable::Long>;
template struct HugeArray<NgramTable::Packed>;
template struct HugeArray<NgramTable::Gen>;

uint64_t TokenTable::hash_long(const char* p, size_t n) {
    uint64_t h = 0x2545F4914F6CDD1Dull ^ n;
    while (n >= 8) { uint64_t w; std::memcpy(&w, p, 8); h = mix(h ^ w) + 0x9E3779B97F4A7C15ull; p += 8; n -= 8; }
    if (n) { uint64_t w = 0; std::memcpy(&w, p, n); h = mix(h ^ w) + 0x9E3779B97F4A7C15ull; }
    return h * 0xD6E8FEB86659FD93ull;
}

int32_t TokenTable::find_long(const char* p, size_t n, uint64_t h) const {
    if (l.empty()) return -1;
    const size_t mask = l.size() - 1;
    for (size_t slot = (size_t)(h >> l_shift);; slot = (slot + 1) & mask) {
        const Long& e = l[slot];
        if (e.len == 0) return -1;
        if (e.hash == h && e.len == n && std::memcmp(arena.data() + e.off, p, n) == 0) return e.idx;
    }
}

void TokenTable::build(const std::vector<std::pair<std::string, int32_t>>& items) {
    size_t ns = 0, nl = 0, bytes = 0;
    for (const auto& it : items) { if (it.first.size() > 8) { ++nl; bytes += it.first.size(); } else if (!it.first.empty()) ++ns; }
    if (bytes > 0xFFFFFFFFull) fail("tokenizer vocabulary: more than 4 GiB of token text");
    const unsigned sb = table_bits(ns), lb = table_bits(nl);
    s.assign_zero((size_t)1 << sb); s_shift = 64 - sb;
    l.release();
    if (nl) { l.assign_zero((size_t)1 << lb); l_shift = 64 - lb; }
    arena.clear(); arena.reserve(bytes);
    n_short = n_long = 0;
    for (const auto& it : items) {
        const std::string& t = it.first;
        if (t.empty()) continue;                       // never looked up: a token has at least one byte
        if (t.size() <= 8) {
            const uint64_t key = load_key(t.data(), t.size(), t.data());     // (last = begin: the byte-wise branch)
            const size_t mask = s.size() - 1;
            size_t slot = short_slot(key, (uint32_t)t.size());
            for (;; slot = (slot + 1) & mask) {
                Short& e = s[slot];
                if (e.len == 0) { e = Short{key, (uint32_t)t.size(), it.second}; ++n_short; break; }
                if (e.key == key && e.len == t.size()) { e.idx = it.second; break; }
            }
        } else {
            const uint64_t h = hash_long(t.data(), t.size());
            const size_t mask = l.size() - 1;
            size_t slot = (size_t)(h >> l_shift);
            for (;; slot = (slot + 1) & mask) {
                Long& e = l[slot];
                if (e.len == 0) {
                    e = Long{h, (uint32_t)arena.size(), (uint32_t)t.size(), it.second, 0};
                    arena.append(t); ++n_long; break;
                }
                if (e.hash == h && e.len == t.size() && std::memcmp(arena.data() + e.off, t.data(), t.size()) == 0) { e.idx = it.second; break; }
            }
        }
    }
}

uint32_t NgramTable::find_gen(const int32_t* t, int n, uint64_t h) const {
    if (gen.empty()) return kNone;
    const size_t mask = gen.size() - 1;
    for (size_t slot = gen_slot(h);; slot = (slot + 1) & mask) {
        const Gen& e = gen[slot];
        if (e.id1 == kNone) return kNone;
        if (e.hash == h && e.n == (uint32_t)n && std::memcmp(arena.data() + e.off, t, (size_t)n * 4) == 0) return e.id1;
    }
}

void NgramTable::build(const std::vector<int32_t>& flat, const std::vector<uint64_t>& off, const std::vector<uint32_t>& ids, size_t vocab_hint) {
    const size_t F = ids.size();
    max_tok = -1; max_n = 0; negative_keys = false;
    for (size_t f = 0; f < F; ++f) {
        const size_t n = off[f + 1] - off[f];
        if (n == 0) continue;                          // an empty n-gram is never looked up (min_ngram >= 1)
        max_n = std::max<int>(max_n, (int)std::min<size_t>(n, 1u << 20));
        for (size_t i = off[f]; i < off[f + 1]; ++i) { negative_keys |= flat[i] < 0; max_tok = std::max(max_tok, flat[i]); }
    }
    pack_bits = 1;
    while (pack_bits < 32 && (((uint64_t)(uint32_t)std::max(max_tok, 0) + 1) >> pack_bits) != 0) ++pack_bits;
    pack_max_n = (int)(64 / pack_bits);
    // unigrams go to the direct array when their token index is a plausible one (the tokenizer numbers tokens 0..V-1)
    size_t uni_cap = 0;
    for (size_t f = 0; f < F; ++f)
        if (off[f + 1] - off[f] == 1) { const int32_t t = flat[off[f]]; if (t >= 0 && (size_t)t < 4 * vocab_hint + 1024) uni_cap = std::max(uni_cap, (size_t)t + 1); }
    uni.assign(uni_cap, kNone);
    auto home = [&](size_t f) {                        // 0 direct, 1 packed, 2 general
        const size_t n = off[f + 1] - off[f];
        bool nonneg = true;
        for (size_t i = off[f]; i < off[f + 1]; ++i) nonneg &= flat[i] >= 0;
        if (n == 1 && nonneg && (size_t)flat[off[f]] < uni_cap) return 0;
        return nonneg && n <= (size_t)pack_max_n ? 1 : 2;
    };
    size_t np = 0, ng = 0, ints = 0;
    for (size_t f = 0; f < F; ++f) {
        const size_t n = off[f + 1] - off[f];
        if (n == 0) continue;
        const int hm = home(f);
        if (hm == 1) ++np; else if (hm == 2) { ++ng; ints += n; }
    }
    if (ints > 0xFFFFFFFFull) fail("tfidf model: n-gram arena over 2^32 token ids");
    const unsigned pb = table_bits(np), gb = table_bits(ng);
    packed.release(); gen.release(); arena.clear();
    if (np) { packed.assign_zero((size_t)1 << pb); p_shift = 64 - pb; }
    if (ng) { gen.assign_zero((size_t)1 << gb); g_shift = 64 - gb; arena.reserve(ints); }
    n_packed = n_gen = 0; packed_n_mask = gen_n_mask = 0;
    for (size_t f = 0; f < F; ++f) {                   // file order: a repeated n-gram keeps the LAST id, like the reference's map assignment
        const size_t n = off[f + 1] - off[f];
        if (n == 0) continue;
        const int32_t* t = flat.data() + off[f];
        const uint32_t id1 = ids[f] + 1;
        const int hm = home(f);
        if (hm == 0) { uni[(size_t)t[0]] = id1; continue; }
        if (hm == 1) {
            const uint64_t key = pack(t, (int)n);
            const size_t mask = packed.size() - 1;
            packed_n_mask |= n_bit((int)n);
            for (size_t slot = packed_slot(key);; slot = (slot + 1) & mask) {
                Packed& e = packed[slot];
                if (e.id1 == kNone) { e = Packed{key, id1, 0}; ++n_packed; break; }
                if (e.key == key) { e.id1 = id1; break; }
            }
        } else {
            if (n > 0x7FFFFFFFull) fail("tfidf model: n-gram too long");
            const uint64_t h = gen_hash(t, (int)n);
            const size_t mask = gen.size() - 1;
            gen_n_mask |= n_bit((int)n);
            for (size_t slot = gen_slot(h);; slot = (slot + 1) & mask) {
                Gen& e = gen[slot];
                if (e.id1 == kNone) {
                    e = Gen{h, (uint32_t)arena.size(), (uint32_t)n, id1, 0};
                    arena.insert(arena.end(), t, t + n); ++n_gen; break;
                }
                if (e.hash == h && e.n == n && std::memcmp(arena.data() + e.off, t, n * 4) == 0) { e.id1 = id1; break; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ load
void TfidfBase::load(const std::string& dir) {
    // ---- tokenizer
    size_t vocab_size = 0;
    {
        const std::string cf = dir + "/tokenizer/config.json";
        const JsonValue j = parse_json_file(cf);
        tok_type = as_int(need(j, "token_type", cf), cf);
        if (tok_type != 10 && tok_type != 20 && tok_type != 30) fail("received unknown tok_type: " + std::to_string(tok_type));
        const std::string vf = dir + "/tokenizer/vocab.txt";
        std::ifstream in(vf, std::ios::binary);
        if (!in.is_open()) fail("Unable to open tokenizer vocab file at " + dir + "/tokenizer/vocab.txt");
        std::string line;
        if (!std::getline(in, line)) fail("Corrupted vocab file.");
        std::vector<std::pair<std::string, int32_t>> items;
        items.reserve((size_t)std::strtoull(line.c_str(), nullptr, 10));
        while (std::getline(in, line)) {
            const size_t pos = line.find('\t');
            if (pos == std::string::npos) fail("Corrupted vocab file.");
            items.emplace_back(line.substr(pos + 1), (int32_t)std::strtol(line.substr(0, pos).c_str(), nullptr, 10));
        }
        vocab.build(items);
        vocab_size = items.size();
    }
    // ---- parameters
    {
        const std::string cf = dir + "/vectorizer/config.json";
        const JsonValue j = parse_json_file(cf);
        const JsonValue& ty = need(j, "type", cf);
        if (ty.type != JsonValue::STRING || ty.str != "tfidf") fail("Wrong vectorizer type: " + ty.str);
        const JsonValue& kw = need(j, "kwargs", cf);
        const JsonValue& ng = need(kw, "ngram_range", cf);
        if (ng.type != JsonValue::ARRAY || ng.arr.size() != 2) fail(cf + ": ngram_range must hold two numbers");
        min_ngram = as_int(ng.arr[0], cf); max_ngram = as_int(ng.arr[1], cf);
        if (min_ngram <= 0 || min_ngram > max_ngram) fail("expect 0 < min_ngram <= max_ngram");
        max_length = as_int(need(kw, "max_length", cf), cf);
        binary = as_bool(need(kw, "binary", cf), cf);
        use_idf = as_bool(need(kw, "use_idf", cf), cf);
        sublinear_tf = as_bool(need(kw, "sublinear_tf", cf), cf);
        const JsonValue& np = need(kw, "norm_p", cf);
        if (np.type == JsonValue::STRING && np.str == "l1") norm_p = 1;
        else if (np.type == JsonValue::STRING && np.str == "l2") norm_p = 2;
        else fail("Unknown normalization type");
    }
    // ---- features: id, idf, n-gram
    {
        const std::string mf = dir + "/vectorizer/tfidf-model.txt";
        FILE* fp = std::fopen(mf.c_str(), "rb");
        if (!fp) fail("Unable to load tfidf model file to " + mf);
        std::string buf;
        char tmp[1 << 16];
        size_t n;
        while ((n = std::fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.append(tmp, n);
        std::fclose(fp);
        const char* s = buf.c_str();
        char* e = nullptr;
        const long long total = std::strtoll(s, &e, 10);
        if (e == s || total < 0) fail("Invalid tfidf model file (total_features).");
        s = e;
        idf.assign((size_t)total, 0.0f);
        std::vector<uint8_t> known((size_t)total, 0);
        std::vector<int32_t> flat; std::vector<uint64_t> off; std::vector<uint32_t> ids;
        off.reserve((size_t)total + 1); ids.reserve((size_t)total); flat.reserve((size_t)total * 2);
        off.push_back(0);
        for (long long f = 0; f < total; ++f) {
            const long id = std::strtol(s, &e, 10);
            if (e == s) fail("Invalid tfidf model file (idx, idf, ngram_len).");
            s = e;
            const float v = std::strtof(s, &e);
            if (e == s) fail("Invalid tfidf model file (idx, idf, ngram_len).");
            s = e;
            const long long len = std::strtoll(s, &e, 10);
            if (e == s || len < 0) fail("Invalid tfidf model file (idx, idf, ngram_len).");
            s = e;
            if (id < 0 || id >= total) fail("tfidf model file: feature id " + std::to_string(id) + " outside [0, " + std::to_string(total) + ")");
            idf[(size_t)id] = v; known[(size_t)id] = 1;
            for (long long t = 0; t < len; ++t) {
                const long tok = std::strtol(s, &e, 10);
                if (e == s) fail("Invalid tfidf model file (tok_idx).");
                s = e;
                flat.push_back((int32_t)tok);
            }
            off.push_back(flat.size()); ids.push_back((uint32_t)id);
        }
        nr_features = 0;
        for (uint8_t k : known) nr_features += k;                        // = idx_idf.size(): the reference's number of columns
        if (nr_features != (uint32_t)total) fail("tfidf model file: duplicate feature ids");
        features.build(flat, off, ids, vocab_size);
        sort_shift = 0;
        while (sort_shift < 24 && (((uint64_t)total - (total > 0)) >> sort_shift) > 255) ++sort_shift;      // id >> sort_shift in [0, 256)
    }
}

// ----------------------------------------------------------------------------------------------------------------------------- count
namespace {
// Ascending order of the feature ids of ONE document (tens to a few thousand, spread over [0, nr_features)): one pass into 256 buckets by
// the top bits, then each bucket -- a handful of ids -- by insertion; a crowded bucket falls back to std::sort.  src is overwritten; the
// sorted ids end up in dst.  (std::sort alone was 37 % of the host half: ~26 ns per id in branch mispredictions.)
void sort_ids(uint32_t* src, size_t n, uint32_t* dst, unsigned shift) {
    auto insertion = [](uint32_t* a, size_t m) {
        for (size_t i = 1; i < m; ++i) {
            const uint32_t v = a[i];
            size_t j = i;
            while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; --j; }
            a[j] = v;
        }
    };
    if (n <= 20) { std::memcpy(dst, src, n * 4); insertion(dst, n); return; }
    // about n / 2 .. n buckets (16..256): the fixed cost of a pass (clear, prefix sum, walk) is per bucket, and a document of ~60 ids does not need 256
    unsigned bits = 4;
    while (bits < 8 && ((size_t)2 << bits) <= n) ++bits;
    const unsigned B = 1u << bits;
    shift += 8 - bits;
    uint32_t start[257], cur[256];
    std::memset(start, 0, (B + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) ++start[(src[i] >> shift) + 1];
    for (unsigned b = 1; b <= B; ++b) start[b] += start[b - 1];
    std::memcpy(cur, start, B * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) dst[cur[src[i] >> shift]++] = src[i];
    for (unsigned b = 0; b < B; ++b) {
        const size_t m = start[b + 1] - start[b];
        if (m < 2) continue;
        if (m <= 32) insertion(dst + start[b], m); else std::sort(dst + start[b], dst + start[b + 1]);
    }
}
inline void grow(std::vector<uint64_t>& v, size_t n) { if (v.size() < n) v.resize(n + n / 2 + 64); }
inline void grow(std::vector<uint32_t>& v, size_t n) { if (v.size() < n) v.resize(n + n / 2 + 64); }
inline void grow(std::vector<int32_t>& v, size_t n) { if (v.size() < n) v.resize(n + n / 2 + 64); }
}  // namespace

size_t TfidfBase::count(const char* doc, size_t len, TfidfScratch& S, uint32_t col_off, TfidfOut& O) const {
    const char* p = doc; const char* const last = doc + len;
    // the scratch arrays only ever grow (no per-document clearing): a word document has at most (len + 1) / 2 tokens, a character one len
    const size_t tok_bound = std::min<size_t>(max_length > 0 ? (size_t)max_length : ~(size_t)0, tok_type == 10 ? (len + 1) / 2 : len);
    grow(S.key, tok_bound + 1); grow(S.len, tok_bound + 1); grow(S.aux, tok_bound + 1); grow(S.tok, tok_bound + 1); grow(S.run, tok_bound + 2);
    uint64_t* const K = S.key.data(); uint32_t* const L = S.len.data(); uint64_t* const A = S.aux.data();
    size_t T = 0;
    const TokenTable::Short* const vs = vocab.s.data();
    // pass 1: token boundaries, the lookup key of every token, its table slot on the way into the cache.  A keeps a short token's table slot,
    // a long token's offset in the document.
    auto note_short = [&](uint64_t k, size_t n) {
        const size_t slot = vocab.short_slot(k, (uint32_t)n);
        __builtin_prefetch(vs + slot);
        K[T] = k; L[T] = (uint32_t)n; A[T] = (uint64_t)slot; ++T;
    };
    auto note_long = [&](const char* b, size_t n) {
        const uint64_t h = TokenTable::hash_long(b, n);
        if (!vocab.l.empty()) __builtin_prefetch(&vocab.l[(size_t)(h >> vocab.l_shift)]);
        K[T] = h; L[T] = (uint32_t)std::min<size_t>(n, 0xFFFFFFFFu); A[T] = (uint64_t)(b - doc); ++T;
    };
    if (tok_type == 10) {
        while (p < last) {
            if (p + 8 <= last) {                      // eight bytes at once: where is the first ' '?  (exact for the LOWEST zero byte of x)
                uint64_t v; std::memcpy(&v, p, 8);
                const uint64_t x = v ^ 0x2020202020202020ull;
                const uint64_t m = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
                if (m) {
                    const unsigned n = (unsigned)__builtin_ctzll(m) >> 3;
                    if (n) { if (T >= tok_bound) break; note_short(v & ((1ull << (8 * n)) - 1), n); }
                    p += n + 1;
                    continue;
                }
            }
            const char* q = p;
            while (q < last && *q != ' ') ++q;
            if (q != p) {
                if (T >= tok_bound) break;
                const size_t n = (size_t)(q - p);
                if (n <= 8) note_short(load_key(p, n, last), n); else note_long(p, n);
            }
            p = q + 1;
        }
    } else {
        while (p < last) {
            const uint8_t c = (uint8_t)*p;
            size_t cs;
            if (c >= 0xF0) cs = 4; else if (c >= 0xE0) cs = 3; else if (c >= 0xC0) cs = 2; else if (c < 0x80) cs = 1;
            else fail("the string is not utf-8 encoded!");
            if (T >= tok_bound) break;
            // (a truncated multi-byte character at the end of the buffer: the reference reads past it; here the token is what is left)
            const size_t n = std::min(cs, (size_t)(last - p));
            note_short(load_key(p, n, last), n);
            p += cs;
        }
    }
    // pass 2: token indices (unknown -> -1)
    int32_t* const tok = S.tok.data();
    for (size_t i = 0; i < T; ++i)
        tok[i] = L[i] <= 8 ? vocab.find_short(K[i], L[i], (size_t)A[i]) : vocab.find_long(doc + A[i], L[i], K[i]);
    // run[i] = tokens in a row from i that the model's n-grams name at all (0 <= index <= max_tok): an n-gram with any other token in it can
    // only be a feature through the general table, and only when the model file names negative token indices
    const NgramTable& G = features;
    int32_t* const run = S.run.data();
    run[T] = 0;
    for (size_t i = T; i-- > 0;) run[i] = (tok[i] < 0 || tok[i] > G.max_tok) ? 0 : run[i + 1] + 1;
    const int n_lo = min_ngram, n_hi = (int)std::min<size_t>((size_t)std::min(max_ngram, G.max_n), T);
    size_t f_bound = 0;
    for (int n = n_lo; n <= n_hi; ++n) f_bound += T - (size_t)n + 1;
    grow(S.feat, 2 * f_bound + 2);
    uint32_t* const F = S.feat.data();
    size_t nf = 0;
    const unsigned pbits = G.pack_bits;
    for (int n = n_lo; n <= n_hi; ++n) {
        const size_t cnt = T - (size_t)n + 1;
        const bool packable = n <= G.pack_max_n;
        const bool in_packed = packable && (G.packed_n_mask & NgramTable::n_bit(n)) != 0;
        const bool in_gen = (G.gen_n_mask & NgramTable::n_bit(n)) != 0 && (!packable || G.negative_keys);
        if (n == 1) {
            const size_t U = G.uni.size();
            const uint32_t* const uni = G.uni.data();
            for (size_t i = 0; i < cnt; ++i) {
                const int32_t t = tok[i];
                uint32_t id1 = NgramTable::kNone;
                if (t >= 0 && (size_t)t < U) id1 = uni[t];
                else if (t >= 0) { if (in_packed && t <= G.max_tok) { const uint64_t k = (uint64_t)(uint32_t)t + 1u; id1 = G.find_packed(k, G.packed_slot(k)); } }
                else if (in_gen) id1 = G.find_gen(tok + i, 1, NgramTable::gen_hash(tok + i, 1));
                F[nf] = id1 - 1; nf += id1 != 0;
            }
            continue;
        }
        if (in_packed) {
            // the key of the n-gram at i from the one at i - 1: drop the lowest field, add the new token on top
            const unsigned top = pbits * (unsigned)(n - 1);
            uint64_t prev = 0; bool have = false;
            for (size_t i = 0; i < cnt; ++i) {
                if (run[i] < n) { have = false; continue; }
                const uint64_t k = have ? (prev >> pbits) | ((uint64_t)((uint32_t)tok[i + (size_t)n - 1] + 1u) << top) : G.pack(tok + i, n);
                prev = k; have = true;
                K[i] = k;
                __builtin_prefetch(&G.packed[G.packed_slot(k)]);
            }
            for (size_t i = 0; i < cnt; ++i) {
                if (run[i] < n) continue;
                const uint32_t id1 = G.find_packed(K[i], G.packed_slot(K[i]));
                F[nf] = id1 - 1; nf += id1 != 0;
            }
        }
        if (in_gen) {
            // not packable: every n-gram of named tokens (all of them when the model names negative indices); packable: only those the packed table cannot hold
            auto wanted = [&](size_t i) { const bool good = run[i] >= n; return packable ? !good : (good || G.negative_keys); };
            for (size_t i = 0; i < cnt; ++i) {
                if (!wanted(i)) continue;
                const uint64_t h = NgramTable::gen_hash(tok + i, n);
                K[i] = h;
                __builtin_prefetch(&G.gen[G.gen_slot(h)]);
            }
            for (size_t i = 0; i < cnt; ++i) {
                if (!wanted(i)) continue;
                const uint32_t id1 = G.find_gen(tok + i, n, K[i]);
                F[nf] = id1 - 1; nf += id1 != 0;
            }
        }
    }
    if (!nf) return 0;
    // counts per feature, ascending ids; a count is the reference's float incremented once per occurrence (exact up to 2^24, where += 1.0f stops moving)
    O.ensure(nf);
    uint32_t* const oc = O.col + O.n; float* const ov = O.val + O.n;
    size_t out = 0;
    if (S.dense_ok) {
        // dense counters + a three-level bitmap of the ids touched: walking the bitmap yields the ids in ascending order, no sort; everything
        // is back to zero when the walk ends
        uint32_t* const cnt = S.dense.data();
        uint64_t* const b0 = S.bits0.data(); uint64_t* const b1 = S.bits1.data(); uint64_t* const b2 = S.bits2.data();
        for (size_t i = 0; i < nf; ++i) {
            const uint32_t id = F[i];
            ++cnt[id];
            b0[id >> 6] |= 1ull << (id & 63);
            b1[id >> 12] |= 1ull << ((id >> 6) & 63);
            b2[id >> 18] |= 1ull << ((id >> 12) & 63);
        }
        const size_t n2 = ((((size_t)nr_features + 63) / 64 + 63) / 64 + 63) / 64;
        for (size_t w2 = 0; w2 < n2; ++w2) {
            uint64_t m2 = b2[w2];
            if (!m2) continue;
            b2[w2] = 0;
            do {
                const size_t w1 = (w2 << 6) | (unsigned)__builtin_ctzll(m2);
                m2 &= m2 - 1;
                uint64_t m1 = b1[w1];
                b1[w1] = 0;
                do {
                    const size_t w0 = (w1 << 6) | (unsigned)__builtin_ctzll(m1);
                    m1 &= m1 - 1;
                    uint64_t m0 = b0[w0];
                    b0[w0] = 0;
                    do {
                        const uint32_t id = (uint32_t)((w0 << 6) | (unsigned)__builtin_ctzll(m0));
                        m0 &= m0 - 1;
                        oc[out] = col_off + id;
                        ov[out] = (float)std::min<uint32_t>(cnt[id], 1u << 24);
                        cnt[id] = 0;
                        ++out;
                    } while (m0);
                } while (m1);
            } while (m2);
        }
    } else {
        uint32_t* const sorted = F + f_bound + 1;
        sort_ids(F, nf, sorted, sort_shift);
        // run lengths without a data-dependent branch: every step writes the current run's id and count, a new id moves the cursor
        uint32_t c = 1;
        oc[0] = col_off + sorted[0];
        for (size_t r = 1; r < nf; ++r) {
            const bool same = sorted[r] == sorted[r - 1];
            ov[out] = (float)c;
            out += !same;
            c = same ? std::min<uint32_t>(c + 1, 1u << 24) : 1;
            oc[out] = col_off + sorted[r];
        }
        ov[out++] = (float)c;
    }
    O.n += out;
    return out;
}

void TfidfScratch::prepare(uint32_t max_features, size_t dense_limit) {
    dense_ok = max_features > 0 && max_features <= dense_limit;
    if (!dense_ok) return;
    const size_t n0 = ((size_t)max_features + 63) / 64, n1 = (n0 + 63) / 64, n2 = (n1 + 63) / 64;
    dense.assign(max_features, 0); bits0.assign(n0, 0); bits1.assign(n1, 0); bits2.assign(n2, 0);
}

void TfidfOut::ensure(size_t extra) {
    if (n + extra <= cap) return;
    const size_t want = std::max<size_t>(std::max<size_t>(2 * cap, n + extra), 1u << 16);
    // realloc, not a vector: growing a large block is a page remap (no copy), and nothing is zero-filled
    void* c = std::realloc(col, want * sizeof(uint32_t));
    if (c) col = static_cast<uint32_t*>(c);
    void* v = std::realloc(val, want * sizeof(float));
    if (v) val = static_cast<float*>(v);
    if (!c || !v) fail("tfidf: out of host memory");
    cap = want;
}
TfidfOut::~TfidfOut() { std::free(col); std::free(val); }

void TfidfVectorizer::load(const std::string& dir) {
    base.clear();
    if (!file_exists(dir + "/meta.json")) {          // a folder saved from one BaseVectorizer
        base.resize(1);
        base[0].load(dir);
        norm_p = base[0].norm_p;
    } else {
        const std::string cf = dir + "/meta.json";
        const JsonValue j = parse_json_file(cf);
        const JsonValue& ty = need(j, "type", cf);
        if (ty.type != JsonValue::STRING || ty.str != "tfidf") fail("Wrong vectorizer type: " + ty.str);
        const JsonValue& kw = need(j, "kwargs", cf);
        const int nb = as_int(need(kw, "num_base_vect", cf), cf);
        norm_p = as_int(need(kw, "norm_p", cf), cf);
        if (nb <= 0) fail(cf + ": num_base_vect must be positive");
        base.resize((size_t)nb);
        for (int i = 0; i < nb; ++i) base[(size_t)i].load(dir + "/" + std::to_string(i) + ".base");
    }
    if (norm_p != 1 && norm_p != 2) fail("invalid normalize option, norm_p: [ 1| 2]");
    uint64_t tot = 0;
    for (const auto& b : base) tot += b.nr_features;
    if (tot > 0xFFFFFFFFull) fail("tfidf: too many features");
    nr_features = (uint32_t)tot;
}

// Documents go to the threads in small dynamic chunks (document lengths are far from uniform); every thread appends to its own arrays and
// notes where each chunk's output starts; after the prefix sum over chunks the caller provides the destination and the threads copy their
// pieces to their final position in it.
void TfidfVectorizer::count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr,
                                   std::vector<uint32_t>& col_idx, std::vector<float>& cnt) const {
    count_corpus(corpus, doc_lens, nr_doc, threads, seg_ptr, [&](uint64_t n, uint32_t*& c, float*& v) {
        col_idx.resize(n); cnt.resize(n);
        c = col_idx.data(); v = cnt.data();
    });
}

void TfidfVectorizer::count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr,
                                   const Provide& provide) const {
    // XRL_TFIDF_TIMING=1: one line per call on stderr (count | prefix sum + result pages | copy)
    const bool timing = std::getenv("XRL_TFIDF_TIMING") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = timing ? now_ms() : 0.0;
    const size_t nb = base.size();
    // threads <= 0: every hardware thread up to 128 and one per ~1024 documents -- on the 256-thread host of the MI355X box the counting
    // phase of 200 k documents takes 4.6 ms on 64 threads, 7 ms on 128, 11 ms on 256 (thread creation), profiles/r04_tfidf_host.md
    unsigned nt = threads > 0 ? (unsigned)threads : (unsigned)std::min<size_t>(std::min(128u, std::max(1u, std::thread::hardware_concurrency())), nr_doc / 1024 + 1);
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, (nr_doc + 63) / 64));
    nt = std::min(nt, 256u);
    const size_t chunk = std::max<size_t>(16, std::min<size_t>(512, nr_doc / ((size_t)nt * 16) + 1));
    const size_t n_chunks = (nr_doc + chunk - 1) / chunk;
    std::vector<uint32_t> col_off(nb, 0);
    for (size_t b = 1; b < nb; ++b) col_off[b] = col_off[b - 1] + base[b - 1].nr_features;
    seg_ptr.assign(nr_doc * nb + 1, 0);                         // first the segment LENGTHS at [seg + 1]
    struct Piece { unsigned thread; size_t begin, n; };         // chunk c's entries: parts[thread].col[begin, begin + n)
    std::vector<Piece> piece(n_chunks, Piece{0, 0, 0});
    struct Part { TfidfOut out; std::string err; };
    std::vector<Part> parts(nt);
    // dense per-thread counters (4 B per feature) while all threads together stay under 1 GiB; past that the per-document sort
    uint32_t max_features = 0;
    for (const auto& b : base) max_features = std::max(max_features, b.nr_features);
    size_t dense_limit = std::min<size_t>((size_t)1 << 24, ((size_t)1 << 28) / nt);
    if (const char* e = std::getenv("XRL_TFIDF_DENSE_LIMIT")) dense_limit = (size_t)std::strtoull(e, nullptr, 10);      // (tests: 0 forces the sort path)
    std::atomic<size_t> next{0};
    std::atomic<bool> stop{false};
    auto work = [&](unsigned t) {
        Part& P = parts[t];
        try {
            TfidfScratch S;
            S.prepare(max_features, dense_limit);
            for (;;) {
                const size_t c = next.fetch_add(1, std::memory_order_relaxed);
                if (c >= n_chunks || stop.load(std::memory_order_relaxed)) break;
                const size_t d0 = c * chunk, d1 = std::min(nr_doc, d0 + chunk), begin = P.out.n;
                for (size_t d = d0; d < d1; ++d)
                    for (size_t b = 0; b < nb; ++b) seg_ptr[d * nb + b + 1] = base[b].count(corpus[d], doc_lens[d], S, col_off[b], P.out);
                piece[c] = Piece{t, begin, P.out.n - begin};
            }
        } catch (const std::exception& e) { P.err = e.what(); stop.store(true); }
    };
    // (chunks are handed out dynamically, so the work gets done by however many threads could be started: when the process is at its
    //  thread limit -- std::system_error from the constructor -- the ones that exist, at least the caller's, share it; nothing unwinds
    //  past a joinable thread)
    auto run_all = [&](auto&& fn) {
        std::vector<std::thread> th;
        th.reserve(nt);
        try {
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(fn, t);
        } catch (const std::system_error&) {
        }
        std::exception_ptr ep;
        try { fn(0u); } catch (...) { ep = std::current_exception(); stop.store(true); }
        for (auto& x : th) x.join();
        if (ep) std::rethrow_exception(ep);
    };
    run_all(work);
    const double t_counted = timing ? now_ms() : 0.0;
    for (const auto& P : parts) if (!P.err.empty()) fail(P.err);
    for (size_t i = 1; i < seg_ptr.size(); ++i) seg_ptr[i] += seg_ptr[i - 1];
    const uint64_t total = seg_ptr.back();
    uint32_t* dst_col = nullptr; float* dst_cnt = nullptr;
    provide(total, dst_col, dst_cnt);
    if (total && (!dst_col || !dst_cnt)) fail("tfidf: no destination for the term counts");
    std::vector<uint64_t> at(n_chunks + 1, 0);
    for (size_t c = 0; c < n_chunks; ++c) at[c + 1] = at[c] + piece[c].n;
    if (at[n_chunks] != total) fail("tfidf: internal error (chunk sizes)");
    const double t_sized = timing ? now_ms() : 0.0;
    next.store(0);
    run_all([&](unsigned) {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= n_chunks) break;
            const Piece& pc = piece[c];
            if (!pc.n) continue;
            std::memcpy(dst_col + at[c], parts[pc.thread].out.col + pc.begin, pc.n * 4);
            std::memcpy(dst_cnt + at[c], parts[pc.thread].out.val + pc.begin, pc.n * 4);
        }
    });
    if (timing)
        std::fprintf(stderr, "[xrl tfidf] %zu documents, %u threads, %zu chunks: count %.2f ms, prefix + destination %.2f ms, copy %.2f ms (nnz %llu)\n", nr_doc, nt, n_chunks,
                     t_counted - t_begin, t_sized - t_counted, now_ms() - t_sized, (unsigned long long)total);
}

}  // namespace xrl
