// Host-side model-file readers (npz / param.json).  See xrl_io.cpp.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "xrl_common.h"

namespace xrl {

struct HostCsc {
    uint32_t rows = 0, cols = 0;
    std::vector<uint64_t> col_ptr;
    std::vector<uint32_t> row_idx;
    std::vector<float> val;
    uint64_t nnz() const { return row_idx.size(); }
};

struct JsonValue {
    enum Type { NUL, BOOL, NUMBER, STRING, ARRAY, OBJECT } type = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;
    const JsonValue* get(const std::string& key) const;
};

JsonValue parse_json_file(const std::string& path);
void load_csc_npz(const std::string& path, HostCsc& out);
bool file_exists(const std::string& path);

}  // namespace xrl
