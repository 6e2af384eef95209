// The reference's memory-mapped model format (*.mmap_store), read AND written natively.
//
// Container (pecos/core/utils/mmap_util.hpp:54-184, 289-337): data blocks, each starting at a
// 16-byte-aligned offset (zero padding), then the metadata `u64 n_blocks; {u64 offset; u64 size}[n]`,
// then a 16-byte trailer `0x93 'P' 'E' 'C' 'O' 'S' | endianness | version(=1) | u64 meta_offset`.
// A scalar is a 1-element block; a MmapableVector<T> is two blocks: `u64 size`, `T[size]` (:526-537).
//
//   W.mmap_store  (bin_search_chunked_matrix_t, inference.hpp:413-455):
//       u32 chunk_count, u32 rows, u32 cols, vec<chunk_t 32 B>, vec<u32 row_idx>, vec<u64 row_ptr>
//       (nnz_rows+1 per NON-EMPTY chunk, absolute entry offsets), vec<{u32 col_offset; f32 val}>
//   C.mmap_store  (csc_t, matrix.hpp:386-407): u32 rows, u32 cols, u64 nnz, u64 col_ptr[cols+1],
//       u32 row_idx[nnz], f32 val[nnz]  -- already in the rearranged (contiguous) child order
//   perm.mmap_store (rearrangement_t, inference.hpp:1720-1728), only if children were re-ordered:
//       vec<u32 perm> (orig -> new, == nnz(C) marks a pruned child), vec<u32 perm_inv> (new -> orig)
//   param.json: {"model":"HierarchicalMLModel","depth":T,"is_mmap":true} / per layer
//       {"model":"MLModel","bias":..,"pred_kwargs":{..},"is_mmap":true}      (inference.hpp:86-99,159-176)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <fstream>

#include "xrl_model.h"

namespace xrl {
namespace {

struct ChunkRec { uint32_t col_begin, col_end, nnz_rows, has_bias; uint64_t p0, p1; };   // 32 bytes on disk
static_assert(sizeof(ChunkRec) == 32, "chunk_t layout");

class MmapReader {
public:
    explicit MmapReader(const std::string& path) : path_(path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) fail("cannot open " + path);
        struct stat st;
        if (::fstat(fd_, &st) != 0 || st.st_size < 24) { ::close(fd_); fail(path + " is not a valid PECOS MMAP file."); }
        n_ = (size_t)st.st_size;
        void* m = ::mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) { ::close(fd_); fail("cannot mmap " + path); }
        p_ = (const uint8_t*)m;
        const uint8_t magic[6] = {0x93u, 'P', 'E', 'C', 'O', 'S'};
        const uint8_t* sig = p_ + n_ - 16;
        if (std::memcmp(sig, magic, 6) != 0) fail(path + ": File is not a valid PECOS MMAP file.");
        if (sig[6] != (uint8_t)'<') fail(path + ": Inconsistent endianness between runtime and the mmap file.");
        if (sig[7] != 1) fail(path + ": Inconsistent version between code and the mmap file.");
        uint64_t meta; std::memcpy(&meta, sig + 8, 8);
        if (meta + 8 > n_) fail(path + ": corrupt mmap metadata");
        uint64_t nb; std::memcpy(&nb, p_ + meta, 8);
        if (meta + 8 + nb * 16 > n_) fail(path + ": corrupt mmap metadata");
        blocks_.resize(nb);
        for (uint64_t i = 0; i < nb; ++i) {
            std::memcpy(&blocks_[i].first, p_ + meta + 8 + i * 16, 8);
            std::memcpy(&blocks_[i].second, p_ + meta + 16 + i * 16, 8);
            if (blocks_[i].first + blocks_[i].second > meta) fail(path + ": mmap block out of range");
        }
    }
    ~MmapReader() { if (p_) ::munmap((void*)p_, n_); if (fd_ >= 0) ::close(fd_); }
    template <class T> T one() { const T* p = many<T>(1); return *p; }
    template <class T> const T* many(uint64_t count) {
        if (it_ >= blocks_.size()) fail(path_ + ": mmap file has fewer blocks than expected");
        const auto b = blocks_[it_++];
        if (b.second != count * sizeof(T)) fail(path_ + ": unexpected mmap block size");
        return reinterpret_cast<const T*>(p_ + b.first);
    }
    template <class T> const T* vec(uint64_t& size) { size = one<uint64_t>(); return many<T>(size); }
private:
    std::string path_; int fd_ = -1; const uint8_t* p_ = nullptr; size_t n_ = 0;
    std::vector<std::pair<uint64_t, uint64_t>> blocks_; size_t it_ = 0;
};

class MmapWriter {
public:
    explicit MmapWriter(const std::string& path) : path_(path), fp_(std::fopen(path.c_str(), "wb")) {
        if (!fp_) fail("MmapStoreSave: Open file failed: " + path);
    }
    ~MmapWriter() { if (fp_) std::fclose(fp_); }
    template <class T> void many(const T* src, uint64_t count) {
        const uint64_t pad = (16 - (end_ % 16)) % 16;
        static const char zeros[16] = {0};
        put(zeros, pad);
        blocks_.emplace_back(end_ + pad, (uint64_t)sizeof(T) * count);
        put(src, sizeof(T) * count);
        end_ = blocks_.back().first + blocks_.back().second;
    }
    template <class T> void one(const T& v) { many<T>(&v, 1); }
    template <class T> void vec(const std::vector<T>& v) { one<uint64_t>(v.size()); many<T>(v.data(), v.size()); }
    void finish() {
        const uint64_t meta = pos_;
        const uint64_t nb = blocks_.size();
        put(&nb, 8);
        for (auto& b : blocks_) { put(&b.first, 8); put(&b.second, 8); }
        const uint8_t sig[8] = {0x93u, 'P', 'E', 'C', 'O', 'S', (uint8_t)'<', 1};
        put(sig, 8); put(&meta, 8);
        if (std::fclose(fp_) != 0) { fp_ = nullptr; fail("MmapStoreSave: Close file failed: " + path_); }
        fp_ = nullptr;
    }
private:
    void put(const void* p, size_t n) { if (n && std::fwrite(p, 1, n, fp_) != n) fail("write failed: " + path_); pos_ += n; }
    std::string path_; FILE* fp_; uint64_t end_ = 0, pos_ = 0;
    std::vector<std::pair<uint64_t, uint64_t>> blocks_;
};

void read_layer_params(const std::string& lp, float& bias, uint32_t& topk, std::string& pp) {
    const JsonValue p = parse_json_file(lp + "/param.json");
    const JsonValue* b = p.get("bias"); const JsonValue* kw = p.get("pred_kwargs");
    if (!b || b->type != JsonValue::NUMBER || !kw) fail(lp + "/param.json: missing bias / pred_kwargs");
    const JsonValue* t = kw->get("only_topk"); const JsonValue* q = kw->get("post_processor");
    if (!t || !q || q->type != JsonValue::STRING) fail(lp + "/param.json: missing pred_kwargs.only_topk / post_processor");
    bias = (float)b->num; topk = (uint32_t)t->num; pp = q->str;
}
}  // namespace

// HierarchicalMLModel(folderpath, lazy_load) / load_mmap, inference.hpp:2597-2614, LayerData::init_mmap :1885-1908
std::unique_ptr<Model> load_mmap_model_from_disk(const std::string& path) {
    const JsonValue meta = parse_json_file(path + "/param.json");
    const JsonValue* dv = meta.get("depth");
    if (!dv || dv->type != JsonValue::NUMBER) fail(path + "/param.json: missing \"depth\"");
    const JsonValue* mm = meta.get("is_mmap");
    if (!mm || mm->type != JsonValue::BOOL || !mm->b) fail("This folder contains npz model. Cannot load in mmap format.");
    auto m = std::make_unique<Model>();
    XRL_HIP(hipGetDevice(&m->device));
    m->weight_matrix_type = 2;
    for (int d = 0; d < (int)dv->num; ++d) {
        const std::string lp = path + "/" + std::to_string(d) + ".model";
        float bias; uint32_t topk; std::string pp;
        read_layer_params(lp, bias, topk, pp);
        // W: chunked -> CSC in the rearranged column space (rows ascending inside a column)
        HostCsc W, C;
        {
            MmapReader r(lp + "/W.mmap_store");
            const uint32_t chunk_count = r.one<uint32_t>();
            W.rows = r.one<uint32_t>(); W.cols = r.one<uint32_t>();
            uint64_t n_chunks, n_ridx, n_rptr, n_ent;
            const ChunkRec* chunks = r.vec<ChunkRec>(n_chunks);
            const uint32_t* ridx = r.vec<uint32_t>(n_ridx);
            const uint64_t* rptr = r.vec<uint64_t>(n_rptr);
            struct E { uint32_t col; float val; };
            const E* ent = r.vec<E>(n_ent);
            if (n_chunks != chunk_count) fail(lp + "/W.mmap_store: chunk count mismatch");
            W.col_ptr.assign((size_t)W.cols + 1, 0);
            size_t io = 0, po = 0;
            for (uint32_t c = 0; c < chunk_count; ++c) {        // count
                const ChunkRec& ch = chunks[c];
                if (ch.nnz_rows == 0) continue;
                if (io + ch.nnz_rows > n_ridx || po + ch.nnz_rows + 1 > n_rptr) fail(lp + "/W.mmap_store: truncated chunk arrays");
                for (uint64_t e = rptr[po]; e < rptr[po + ch.nnz_rows]; ++e) {
                    if (e >= n_ent || ch.col_begin + ent[e].col >= W.cols) fail(lp + "/W.mmap_store: entry out of range");
                    W.col_ptr[ch.col_begin + ent[e].col + 1]++;
                }
                io += ch.nnz_rows; po += ch.nnz_rows + 1;
            }
            for (uint32_t c = 0; c < W.cols; ++c) W.col_ptr[c + 1] += W.col_ptr[c];
            W.row_idx.resize(n_ent); W.val.resize(n_ent);
            std::vector<uint64_t> fill(W.col_ptr.begin(), W.col_ptr.end() - 1);
            io = 0; po = 0;
            for (uint32_t c = 0; c < chunk_count; ++c) {        // fill: chunk rows ascending -> columns stay sorted
                const ChunkRec& ch = chunks[c];
                if (ch.nnz_rows == 0) continue;
                for (uint32_t s = 0; s < ch.nnz_rows; ++s)
                    for (uint64_t e = rptr[po + s]; e < rptr[po + s + 1]; ++e) {
                        const uint64_t dst = fill[ch.col_begin + ent[e].col]++;
                        W.row_idx[dst] = ridx[io + s]; W.val[dst] = ent[e].val;
                    }
                io += ch.nnz_rows; po += ch.nnz_rows + 1;
            }
        }
        {
            MmapReader r(lp + "/C.mmap_store");
            C.rows = r.one<uint32_t>(); C.cols = r.one<uint32_t>();
            const uint64_t nnz = r.one<uint64_t>();
            const uint64_t* cp = r.many<uint64_t>((uint64_t)C.cols + 1);
            const uint32_t* ri = r.many<uint32_t>(nnz);
            const float* va = r.many<float>(nnz);
            C.col_ptr.assign(cp, cp + C.cols + 1); C.row_idx.assign(ri, ri + nnz); C.val.assign(va, va + nnz);
        }
        std::vector<uint32_t> perm_inv; uint32_t orig_rows = C.rows;
        if (file_exists(lp + "/perm.mmap_store")) {
            MmapReader r(lp + "/perm.mmap_store");
            uint64_t np, ni;
            const uint32_t* perm = r.vec<uint32_t>(np); (void)perm;
            const uint32_t* pinv = r.vec<uint32_t>(ni);
            perm_inv.assign(pinv, pinv + ni);
            orig_rows = (uint32_t)np;
            if (ni != C.rows) fail(lp + "/perm.mmap_store: size does not match C");
            for (uint64_t i = 0; i < ni; ++i) if (perm_inv[i] >= np) fail(lp + "/perm.mmap_store: perm_inv entry out of range");
        }
        m->layers.push_back(compile_layer(W, C, bias, topk, pp, perm_inv.empty() ? nullptr : &perm_inv, orig_rows));
    }
    finalize_model(*m);
    return m;
}

// c_xlinear_compile_mmap_model (libpecos.cpp:133-138): npz folder -> mmap folder, byte layout as above
// (make_chunked_from_csc, inference.hpp:557-650; rearrangement_t, :1746-1824; save_mmap, :2575-2595).
static void make_dirs(const std::string& path) {
    // mkdir -p without a shell: the path comes straight from the caller (C ABI / Python)
    if (path.empty()) fail("Cannot create folder: empty path");
    for (size_t i = 1; i <= path.size(); ++i) {
        if (i != path.size() && path[i] != '/') continue;
        const std::string part = path.substr(0, i);
        if (::mkdir(part.c_str(), 0777) != 0 && errno != EEXIST)
            fail("Cannot create folder " + part + ": " + std::strerror(errno));
    }
    struct stat st;
    if (::stat(path.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) fail("Cannot create folder: " + path + " exists and is not a directory");
}

void compile_mmap_model(const std::string& npz_path, const std::string& mmap_path) {
    const JsonValue meta = parse_json_file(npz_path + "/param.json");
    const JsonValue* dv = meta.get("depth");
    if (!dv || dv->type != JsonValue::NUMBER) fail(npz_path + "/param.json: missing \"depth\"");
    if (const JsonValue* mm = meta.get("is_mmap")) if (mm->type == JsonValue::BOOL && mm->b) fail("This folder contains mmap model. Cannot load in npz format.");
    const int depth = (int)dv->num;
    make_dirs(mmap_path);
    { std::ofstream o(mmap_path + "/param.json"); o << "{\n\"model\": \"HierarchicalMLModel\",\n\"depth\": " << depth << ",\n\"is_mmap\": true\n}\n"; }
    for (int d = 0; d < depth; ++d) {
        const std::string lp = npz_path + "/" + std::to_string(d) + ".model", op = mmap_path + "/" + std::to_string(d) + ".model";
        make_dirs(op);
        float bias; uint32_t topk; std::string pp;
        read_layer_params(lp, bias, topk, pp);
        { std::ofstream o(op + "/param.json");
          o << "{\n\"model\": \"MLModel\",\n\"bias\": " << bias << ",\n\"pred_kwargs\": {\n\t\"only_topk\": " << topk
            << ",\n\t\"post_processor\": \"" << pp << "\"\n\t},\n\"is_mmap\": true\n}\n"; }
        HostCsc W, C;
        load_csc_npz(lp + "/W.npz", W);
        if (d == 0 && !file_exists(lp + "/C.npz")) {
            C.rows = W.cols; C.cols = 1; C.col_ptr = {0, W.cols}; C.row_idx.resize(W.cols); C.val.assign(W.cols, 1.f);
            for (uint32_t i = 0; i < W.cols; ++i) C.row_idx[i] = i;
        } else load_csc_npz(lp + "/C.npz", C);
        const uint64_t c_nnz = C.nnz();
        bool contiguous = (c_nnz == C.rows);
        if (contiguous) for (uint64_t i = 0; i < c_nnz; ++i) if (C.row_idx[i] != i) { contiguous = false; break; }
        auto orig_col = [&](uint32_t c) { return contiguous ? c : C.row_idx[c]; };
        const bool has_bias = bias > 0.0f;
        // chunked W
        std::vector<ChunkRec> chunks(C.cols);
        std::vector<uint32_t> ridx; std::vector<uint64_t> rptr;
        struct E { uint32_t col; float val; };
        std::vector<E> ent;
        struct Nz { uint32_t row, col; float val; };
        std::vector<Nz> nz;
        for (uint32_t p = 0; p < C.cols; ++p) {
            ChunkRec& ch = chunks[p];
            ch.col_begin = (uint32_t)C.col_ptr[p]; ch.col_end = (uint32_t)C.col_ptr[p + 1]; ch.p0 = ch.p1 = 0;
            nz.clear();
            for (uint32_t c = ch.col_begin; c < ch.col_end; ++c) {
                const uint32_t oc = orig_col(c);
                for (uint64_t e = W.col_ptr[oc]; e < W.col_ptr[oc + 1]; ++e) nz.push_back(Nz{W.row_idx[e], c - ch.col_begin, W.val[e]});
            }
            std::stable_sort(nz.begin(), nz.end(), [](const Nz& a, const Nz& b) { return a.row < b.row; });
            uint32_t nrows = 0;
            const uint64_t base = ent.size();
            for (size_t i = 0; i < nz.size(); ++i) {
                if (i == 0 || nz[i].row != nz[i - 1].row) { ridx.push_back(nz[i].row); rptr.push_back(base + i); ++nrows; }
                ent.push_back(E{nz[i].col, nz[i].val});
            }
            if (nrows) rptr.push_back(base + nz.size());
            ch.nnz_rows = nrows;
            ch.has_bias = (has_bias && nrows > 0 && ridx.back() == W.rows - 1) ? 1u : 0u;
        }
        {
            MmapWriter w(op + "/W.mmap_store");
            w.one<uint32_t>(C.cols); w.one<uint32_t>(W.rows); w.one<uint32_t>((uint32_t)c_nnz);
            w.vec(chunks); w.vec(ridx); w.vec(rptr); w.vec(ent);
            w.finish();
        }
        {   // C in rearranged order: row_idx[i] = i
            MmapWriter w(op + "/C.mmap_store");
            std::vector<uint32_t> ri(c_nnz); for (uint64_t i = 0; i < c_nnz; ++i) ri[i] = (uint32_t)i;
            w.one<uint32_t>((uint32_t)c_nnz); w.one<uint32_t>(C.cols); w.one<uint64_t>(c_nnz);
            w.many<uint64_t>(C.col_ptr.data(), (uint64_t)C.cols + 1); w.many<uint32_t>(ri.data(), c_nnz); w.many<float>(C.val.data(), c_nnz);
            w.finish();
        }
        if (!contiguous) {
            std::vector<uint32_t> perm(C.rows, (uint32_t)c_nnz), pinv(c_nnz);
            for (uint64_t i = 0; i < c_nnz; ++i) { perm[C.row_idx[i]] = (uint32_t)i; pinv[i] = C.row_idx[i]; }
            MmapWriter w(op + "/perm.mmap_store");
            w.vec(perm); w.vec(pinv);
            w.finish();
        }
    }
}

}  // namespace xrl
