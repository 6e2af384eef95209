// Model-file readers: the reference's on-disk XR-Linear layout, read natively so that
// c_xlinear_load_model_from_disk{,_ext}(path) is a drop-in (pecos/core/libpecos.cpp:116-126).
//
//   <path>/param.json                 {"model":"HierarchicalMLModel","depth":T,...}   inference.hpp:52-99
//   <path>/{d}.model/param.json       {"bias":..,"pred_kwargs":{"only_topk":..,"post_processor":".."}}
//                                                                                       inference.hpp:101-176
//   <path>/{d}.model/W.npz, C.npz     scipy CSC, float32, UNCOMPRESSED npz             scipy_loader.hpp:311-372
//
// This is an independent implementation: the zip is read through its central directory
// (zip64-aware), .npy headers are parsed generically, integer arrays of any width are cast to
// u32 / u64 like scipy_loader.hpp:152-183 does, and compressed members are rejected with the
// same message class as scipy_loader.hpp:247-249.
#include "xrl_io.h"

#include <algorithm>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cstring>
#include <map>

namespace xrl {

// ------------------------------------------------------------------------------ tiny JSON
namespace {
struct JsonParser {
    const char* s; const char* e;
    void ws() { while (s < e && std::isspace((unsigned char)*s)) ++s; }
    [[noreturn]] void bad(const char* what) { fail(std::string("param.json: ") + what); }
    JsonValue parse() {
        ws();
        if (s >= e) bad("unexpected end");
        JsonValue v;
        if (*s == '{') {
            v.type = JsonValue::OBJECT; ++s; ws();
            if (s < e && *s == '}') { ++s; return v; }
            for (;;) {
                ws();
                JsonValue k = parse();
                if (k.type != JsonValue::STRING) bad("object key must be a string");
                ws();
                if (s >= e || *s != ':') bad("expected ':'");
                ++s;
                v.obj.emplace_back(k.str, parse());
                ws();
                if (s < e && *s == ',') { ++s; continue; }
                if (s < e && *s == '}') { ++s; break; }
                bad("expected ',' or '}'");
            }
        } else if (*s == '[') {
            v.type = JsonValue::ARRAY; ++s; ws();
            if (s < e && *s == ']') { ++s; return v; }
            for (;;) {
                v.arr.push_back(parse());
                ws();
                if (s < e && *s == ',') { ++s; continue; }
                if (s < e && *s == ']') { ++s; break; }
                bad("expected ',' or ']'");
            }
        } else if (*s == '"') {
            v.type = JsonValue::STRING; ++s;
            while (s < e && *s != '"') {
                if (*s == '\\' && s + 1 < e) {
                    ++s;
                    switch (*s) {
                    case 'n': v.str.push_back('\n'); break;
                    case 't': v.str.push_back('\t'); break;
                    case 'u': v.str.push_back('?'); s += (e - s > 4) ? 4 : 0; break;
                    default: v.str.push_back(*s);
                    }
                    ++s;
                } else {
                    v.str.push_back(*s++);
                }
            }
            if (s >= e) bad("unterminated string");
            ++s;
        } else if (!std::strncmp(s, "true", 4)) { v.type = JsonValue::BOOL; v.b = true; s += 4; }
        else if (!std::strncmp(s, "false", 5)) { v.type = JsonValue::BOOL; v.b = false; s += 5; }
        else if (!std::strncmp(s, "null", 4)) { v.type = JsonValue::NUL; s += 4; }
        else {
            char* end = nullptr;
            v.type = JsonValue::NUMBER;
            v.num = std::strtod(s, &end);
            if (end == s) bad("unexpected token");
            s = end;
        }
        return v;
    }
};
}  // namespace

const JsonValue* JsonValue::get(const std::string& key) const {
    for (auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
}

JsonValue parse_json_file(const std::string& path) {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) fail("cannot open " + path);
    std::string buf;
    char tmp[4096];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.append(tmp, n);
    std::fclose(fp);
    JsonParser p{buf.data(), buf.data() + buf.size()};
    return p.parse();
}

// ------------------------------------------------------------------------------ zip / npy
namespace {
struct MappedFile {
    const uint8_t* p = nullptr; size_t n = 0; int fd = -1;
    explicit MappedFile(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) fail("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); fail("cannot stat " + path); }
        n = (size_t)st.st_size;
        if (n) {
            void* m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); fail("cannot mmap " + path); }
            p = (const uint8_t*)m;
        }
    }
    ~MappedFile() { if (p) ::munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};
template <class T> T rd(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }

struct Member { uint64_t data_off, size; };

std::map<std::string, Member> zip_members(const MappedFile& f, const std::string& path) {
    // end-of-central-directory: scan backwards for PK\5\6
    if (f.n < 22) fail(path + ": not a zip archive");
    size_t pos = f.n - 22;
    const size_t stop = f.n > (1u << 16) + 22 ? f.n - (1u << 16) - 22 : 0;
    bool found = false;
    for (;; --pos) {
        if (rd<uint32_t>(f.p + pos) == 0x06054b50u) { found = true; break; }
        if (pos == stop) break;
    }
    if (!found) fail(path + ": zip end-of-central-directory not found");
    uint64_t n_entries = rd<uint16_t>(f.p + pos + 10);
    uint64_t cd_off = rd<uint32_t>(f.p + pos + 16);
    if (pos >= 20 && rd<uint32_t>(f.p + pos - 20) == 0x07064b50u) {  // zip64 locator
        const uint64_t z = rd<uint64_t>(f.p + pos - 20 + 8);
        if (z + 56 <= f.n && rd<uint32_t>(f.p + z) == 0x06064b50u) {
            n_entries = rd<uint64_t>(f.p + z + 32);
            cd_off = rd<uint64_t>(f.p + z + 48);
        }
    }
    std::map<std::string, Member> out;
    uint64_t c = cd_off;
    for (uint64_t i = 0; i < n_entries; ++i) {
        if (c + 46 > f.n || rd<uint32_t>(f.p + c) != 0x02014b50u) fail(path + ": corrupt zip central directory");
        const uint16_t method = rd<uint16_t>(f.p + c + 10);
        uint64_t csize = rd<uint32_t>(f.p + c + 20), usize = rd<uint32_t>(f.p + c + 24);
        const uint16_t nlen = rd<uint16_t>(f.p + c + 28), xlen = rd<uint16_t>(f.p + c + 30), clen = rd<uint16_t>(f.p + c + 32);
        uint64_t lho = rd<uint32_t>(f.p + c + 42);
        if (c + 46 + (uint64_t)nlen + xlen > f.n) fail(path + ": corrupt zip central directory (entry runs past the file)");
        std::string name((const char*)f.p + c + 46, nlen);
        // zip64 extra: fields appear only for values that were 0xFFFFFFFF, in this fixed order
        uint64_t x = c + 46 + nlen; const uint64_t xe = x + xlen;
        while (x + 4 <= xe) {
            const uint16_t id = rd<uint16_t>(f.p + x), sz = rd<uint16_t>(f.p + x + 2);
            if (id == 0x0001) {
                uint64_t y = x + 4;
                const uint64_t ye = std::min<uint64_t>(xe, x + 4 + sz);
                auto take = [&](uint64_t& v) { if (y + 8 > ye) fail(path + ": corrupt zip64 extra field"); v = rd<uint64_t>(f.p + y); y += 8; };
                if (usize == 0xFFFFFFFFu) take(usize);
                if (csize == 0xFFFFFFFFu) take(csize);
                if (lho == 0xFFFFFFFFu) take(lho);
            }
            x += 4 + sz;
        }
        if (method != 0) fail(path + ": only uncompressed npz archives are supported (save with compressed=False)");
        if (lho > f.n || f.n - lho < 30 || rd<uint32_t>(f.p + lho) != 0x04034b50u) fail(path + ": corrupt zip local header");
        const uint64_t data = lho + 30 + rd<uint16_t>(f.p + lho + 26) + rd<uint16_t>(f.p + lho + 28);
        if (data > f.n || usize > f.n - data) fail(path + ": truncated zip member " + name);   // no wrap-around
        out[name] = Member{data, usize};
        c += 46 + nlen + xlen + clen;
    }
    return out;
}

struct Npy { std::string descr; bool fortran = false; std::vector<uint64_t> shape; const uint8_t* data; uint64_t count; uint32_t itemsize; };

Npy parse_npy(const MappedFile& f, const Member& m, const std::string& what) {
    const uint8_t* p = f.p + m.data_off;
    if (m.size < 10 || std::memcmp(p, "\x93NUMPY", 6) != 0) fail(what + ": not an .npy member");
    const uint8_t major = p[6];
    uint64_t hlen, hoff;
    if (major == 1) { hlen = rd<uint16_t>(p + 8); hoff = 10; } else { hlen = rd<uint32_t>(p + 8); hoff = 12; }
    if (hoff + hlen > m.size) fail(what + ": truncated npy header");
    std::string h((const char*)p + hoff, hlen);
    Npy a;
    auto find_val = [&](const char* key) -> size_t {
        size_t k = h.find(key);
        if (k == std::string::npos) fail(what + ": npy header lacks " + key);
        k = h.find(':', k);
        return k + 1;
    };
    {
        size_t k = find_val("'descr'");
        const size_t q0 = h.find('\'', k), q1 = h.find('\'', q0 + 1);
        a.descr = h.substr(q0 + 1, q1 - q0 - 1);
    }
    { size_t k = find_val("'fortran_order'"); while (h[k] == ' ') ++k; a.fortran = h.compare(k, 4, "True") == 0; }
    {
        size_t k = find_val("'shape'");
        const size_t p0 = h.find('(', k), p1 = h.find(')', p0);
        std::string t = h.substr(p0 + 1, p1 - p0 - 1);
        const char* s = t.c_str();
        while (*s) {
            while (*s && !std::isdigit((unsigned char)*s)) ++s;
            if (!*s) break;
            char* end; a.shape.push_back(std::strtoull(s, &end, 10)); s = end;
        }
    }
    a.count = 1;
    for (auto d : a.shape) a.count *= d;
    if (a.descr.size() < 3) fail(what + ": unsupported dtype " + a.descr);
    if (a.descr[0] == '>') fail(what + ": big-endian arrays are not supported");
    a.itemsize = (uint32_t)std::strtoul(a.descr.c_str() + 2, nullptr, 10);
    if (a.descr[1] == 'U') a.itemsize *= 4;
    a.data = p + hoff + hlen;
    if (hoff + hlen + a.count * a.itemsize > m.size) fail(what + ": truncated npy payload");
    return a;
}

template <class OUT> void cast_ints(const Npy& a, std::vector<OUT>& out, const std::string& what) {
    out.resize(a.count);
    const char k = a.descr[1];
    if (k != 'i' && k != 'u') fail(what + ": expected an integer array, got " + a.descr);
#define XRL_CASE(T) { const uint8_t* p = a.data; for (uint64_t i = 0; i < a.count; ++i) out[i] = (OUT)rd<T>(p + i * sizeof(T)); }
    if (k == 'i' && a.itemsize == 4) XRL_CASE(int32_t)
    else if (k == 'i' && a.itemsize == 8) XRL_CASE(int64_t)
    else if (k == 'u' && a.itemsize == 4) XRL_CASE(uint32_t)
    else if (k == 'u' && a.itemsize == 8) XRL_CASE(uint64_t)
    else if (k == 'i' && a.itemsize == 2) XRL_CASE(int16_t)
    else if (k == 'u' && a.itemsize == 2) XRL_CASE(uint16_t)
    else fail(what + ": unsupported integer width in " + a.descr);
#undef XRL_CASE
}
}  // namespace

void load_csc_npz(const std::string& path, HostCsc& out) {
    MappedFile f(path);
    auto mem = zip_members(f, path);
    auto need = [&](const char* n) -> const Member& {
        auto it = mem.find(n);
        if (it == mem.end()) fail(path + ": missing member " + n);
        return it->second;
    };
    {
        Npy fmt = parse_npy(f, need("format.npy"), path + ":format");
        std::string s;
        if (fmt.descr[1] == 'U') for (uint32_t i = 0; i < fmt.itemsize / 4; ++i) { char c = (char)fmt.data[4 * i]; if (c) s.push_back(c); }
        else for (uint32_t i = 0; i < fmt.itemsize; ++i) { char c = (char)fmt.data[i]; if (c) s.push_back(c); }
        if (s != "csc") fail(path + " is not a valid scipy CSC npz (format=" + s + ")");
    }
    std::vector<uint64_t> shape;
    cast_ints(parse_npy(f, need("shape.npy"), path + ":shape"), shape, path + ":shape");
    if (shape.size() != 2) fail(path + ": shape must have 2 entries");
    if (shape[0] > 0xFFFFFFFFull || shape[1] > 0xFFFFFFFFull) fail(path + ": dimensions exceed uint32");
    out.rows = (uint32_t)shape[0]; out.cols = (uint32_t)shape[1];
    cast_ints(parse_npy(f, need("indptr.npy"), path + ":indptr"), out.col_ptr, path + ":indptr");
    cast_ints(parse_npy(f, need("indices.npy"), path + ":indices"), out.row_idx, path + ":indices");
    Npy d = parse_npy(f, need("data.npy"), path + ":data");
    out.val.resize(d.count);
    if (d.descr[1] == 'f' && d.itemsize == 4) std::memcpy(out.val.data(), d.data, d.count * 4);
    else if (d.descr[1] == 'f' && d.itemsize == 8) for (uint64_t i = 0; i < d.count; ++i) out.val[i] = (float)rd<double>(d.data + 8 * i);
    else fail(path + ": data must be float32/float64, got " + d.descr);
    if (out.col_ptr.size() != (size_t)out.cols + 1) fail(path + ": indptr length does not match shape");
    if (out.col_ptr.back() != out.row_idx.size() || out.row_idx.size() != out.val.size()) fail(path + ": inconsistent nnz");
    // indptr must start at 0 and never decrease (a corrupt file would otherwise send the model compiler and the device
    // kernels over bad ranges); row ids must lie inside the matrix
    if (out.col_ptr.front() != 0) fail(path + ": indptr[0] != 0");
    for (size_t c = 0; c < out.cols; ++c) if (out.col_ptr[c + 1] < out.col_ptr[c]) fail(path + ": indptr decreases at column " + std::to_string(c));
    for (size_t i = 0; i < out.row_idx.size(); ++i) if (out.row_idx[i] >= out.rows) fail(path + ": row index out of range at entry " + std::to_string(i));
}

bool file_exists(const std::string& path) { return ::access(path.c_str(), F_OK) == 0; }

}  // namespace xrl
