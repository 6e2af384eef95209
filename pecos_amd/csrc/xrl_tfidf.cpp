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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

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
}  // namespace

void TfidfBase::load(const std::string& dir) {
    // ---- tokenizer
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
        vocab.reserve((size_t)std::strtoull(line.c_str(), nullptr, 10));
        while (std::getline(in, line)) {
            const size_t pos = line.find('\t');
            if (pos == std::string::npos) fail("Corrupted vocab file.");
            vocab[line.substr(pos + 1)] = (int32_t)std::strtol(line.substr(0, pos).c_str(), nullptr, 10);
        }
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
        idf.assign((size_t)total, 0.0f); idf_known.assign((size_t)total, 0);
        feature_vocab.reserve((size_t)total);
        std::string key;
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
            idf[(size_t)id] = v; idf_known[(size_t)id] = 1;
            key.clear();
            for (long long t = 0; t < len; ++t) {
                const long tok = std::strtol(s, &e, 10);
                if (e == s) fail("Invalid tfidf model file (tok_idx).");
                s = e;
                const int32_t ti = (int32_t)tok;
                key.append(reinterpret_cast<const char*>(&ti), 4);
            }
            feature_vocab[key] = (uint32_t)id;
        }
        nr_features = 0;
        for (uint8_t k : idf_known) nr_features += k;                    // = idx_idf.size(): the reference's number of columns
        if (nr_features != (uint32_t)total) fail("tfidf model file: duplicate feature ids");
    }
}

void TfidfBase::count(const char* doc, size_t len, std::vector<std::pair<uint32_t, float>>& out, std::vector<int32_t>& tok, std::string& key) const {
    out.clear(); tok.clear();
    const char* p = doc; const char* last = doc + len;
    auto push = [&](const char* b, size_t n) {
        if (max_length > 0 && tok.size() >= (size_t)max_length) return false;
        auto it = vocab.find(std::string(b, n));
        tok.push_back(it == vocab.end() ? -1 : it->second);
        return true;
    };
    if (tok_type == 10) {
        while (p < last) {
            const char* q = static_cast<const char*>(std::memchr(p, ' ', (size_t)(last - p)));
            if (!q) q = last;
            if (q != p && !push(p, (size_t)(q - p))) break;
            p = q + 1;
        }
    } else {
        while (p < last) {
            const uint8_t c = (uint8_t)*p;
            size_t cs;
            if (c >= 0xF0) cs = 4; else if (c >= 0xE0) cs = 3; else if (c >= 0xC0) cs = 2; else if (c < 0x80) cs = 1;
            else fail("the string is not utf-8 encoded!");
            // (a truncated multi-byte character at the end of the buffer: the reference reads past it; here the token is what is left)
            if (!push(p, std::min(cs, (size_t)(last - p)))) break;
            p += cs;
        }
    }
    const int T = (int)tok.size();
    for (int n = min_ngram; n <= std::min(max_ngram, T); ++n) {
        for (int i = 0; i + n <= T; ++i) {
            key.assign(reinterpret_cast<const char*>(tok.data() + i), (size_t)n * 4);
            auto it = feature_vocab.find(key);
            if (it != feature_vocab.end()) out.emplace_back(it->second, 1.0f);
        }
    }
    std::sort(out.begin(), out.end());
    size_t w = 0;
    for (size_t r = 0; r < out.size();) {
        size_t r2 = r; float c = 0.0f;
        while (r2 < out.size() && out[r2].first == out[r].first) { c += 1.0f; ++r2; }       // += 1.0 per occurrence, like the reference's float map
        out[w++] = std::make_pair(out[r].first, c);
        r = r2;
    }
    out.resize(w);
}

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

void TfidfVectorizer::count_corpus(const char* const* corpus, const size_t* doc_lens, size_t nr_doc, int threads, std::vector<uint64_t>& seg_ptr,
                                   std::vector<uint32_t>& col_idx, std::vector<float>& cnt) const {
    const size_t nb = base.size();
    unsigned nt = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, (nr_doc + 63) / 64));
    nt = std::min(nt, 64u);
    struct Part { std::vector<uint64_t> seg_len; std::vector<uint32_t> col; std::vector<float> val; std::string err; };
    std::vector<Part> parts(nt);
    std::vector<uint32_t> col_off(nb, 0);
    for (size_t b = 1; b < nb; ++b) col_off[b] = col_off[b - 1] + base[b - 1].nr_features;
    auto work = [&](unsigned t) {
        Part& P = parts[t];
        const size_t d0 = nr_doc * t / nt, d1 = nr_doc * (t + 1) / nt;
        P.seg_len.reserve((d1 - d0) * nb);
        std::vector<std::pair<uint32_t, float>> feats; std::vector<int32_t> tok; std::string key;
        try {
            for (size_t d = d0; d < d1; ++d)
                for (size_t b = 0; b < nb; ++b) {
                    base[b].count(corpus[d], doc_lens[d], feats, tok, key);
                    P.seg_len.push_back(feats.size());
                    for (const auto& f : feats) { P.col.push_back(col_off[b] + f.first); P.val.push_back(f.second); }
                }
        } catch (const std::exception& e) { P.err = e.what(); }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (const auto& P : parts) if (!P.err.empty()) fail(P.err);
    seg_ptr.assign(nr_doc * nb + 1, 0);
    size_t s = 0; uint64_t run = 0;
    for (const auto& P : parts) for (uint64_t l : P.seg_len) { run += l; seg_ptr[++s] = run; }
    col_idx.resize(run); cnt.resize(run);
    uint64_t at = 0;
    for (const auto& P : parts) {
        if (!P.col.empty()) { std::memcpy(col_idx.data() + at, P.col.data(), P.col.size() * 4); std::memcpy(cnt.data() + at, P.val.data(), P.val.size() * 4); }
        at += P.col.size();
    }
}

}  // namespace xrl
