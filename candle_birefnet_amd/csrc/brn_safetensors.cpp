// brn_safetensors.cpp — native safetensors reader for brn_model_create_from_safetensors: the C-ABI face of
// `VarBuilder::from_mmaped_safetensors(&[path], DType::F32, &device)` + `BiRefNet::new(config, vb)` (infer_image.rs:35-40).
// Format (safetensors 0.x): u64 LE header length, a JSON object {name: {"dtype","shape","data_offsets":[b,e]}, "__metadata__"?},
// then the raw little-endian tensor bytes.  F32 is used in place from the mapping; F16 / BF16 are widened to fp32 on the host
// (the reference's VarBuilder converts every dtype to the requested F32 the same way).
#include "brn_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <map>

namespace brn {

namespace {

struct Cursor {
    const char* p; const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    void need(char c) { if (!eat(c)) fail(BRN_ERR_INVALID_ARG, "safetensors header: expected '%c'", c); }
    std::string str() {
        ws();
        if (p >= e || *p != '"') fail(BRN_ERR_INVALID_ARG, "safetensors header: expected a string");
        ++p;
        std::string s;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                if (*p == 'u' && p + 4 < e) { s += '?'; p += 5; continue; }     // names on this path are ASCII
                s += *p++;
            } else s += *p++;
        }
        if (p >= e) fail(BRN_ERR_INVALID_ARG, "safetensors header: unterminated string");
        ++p;
        return s;
    }
    long long num() {
        ws();
        long long v = 0; bool any = false;
        while (p < e && *p >= '0' && *p <= '9') {
            if (v > (1LL << 53) / 10) fail(BRN_ERR_INVALID_ARG, "safetensors header: number out of range");   // sizes beyond 2^53 are not a file
            v = v * 10 + (*p - '0'); ++p; any = true;
        }
        if (!any) fail(BRN_ERR_INVALID_ARG, "safetensors header: expected a number");
        return v;
    }
    void skip_value() {      // any JSON value (used for __metadata__ and unknown keys)
        ws();
        if (p >= e) fail(BRN_ERR_INVALID_ARG, "safetensors header: truncated");
        if (*p == '"') { (void)str(); return; }
        if (*p == '{' || *p == '[') {       // nested containers of either kind, brackets matched on a stack
            std::vector<char> stack;
            do {
                if (*p == '"') { (void)str(); continue; }
                if (*p == '{' || *p == '[') stack.push_back(*p == '{' ? '}' : ']');
                else if (*p == '}' || *p == ']') {
                    if (stack.back() != *p) fail(BRN_ERR_INVALID_ARG, "safetensors header: mismatched brackets");
                    stack.pop_back();
                }
                ++p;
            } while (p < e && !stack.empty());
            if (!stack.empty()) fail(BRN_ERR_INVALID_ARG, "safetensors header: truncated");
            return;
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};

inline float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, ex = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (ex == 0) {
        if (m == 0) u = s;
        else { int e2 = -1; uint32_t mm = m; do { ++e2; mm <<= 1; } while (!(mm & 1024)); u = s | ((uint32_t)(112 - e2) << 23) | ((mm & 1023) << 13); }
    } else if (ex == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((ex + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &u, 4); return f;
}

}  // namespace

SafetensorsFile::~SafetensorsFile() {
    if (map && map != MAP_FAILED) munmap(map, map_len);
}

void SafetensorsFile::open(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) fail(BRN_ERR_INVALID_ARG, "cannot open safetensors file '%s'", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { ::close(fd); fail(BRN_ERR_INVALID_ARG, "'%s' is not a safetensors file", path); }
    map_len = (size_t)st.st_size;
    map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) { map = nullptr; fail(BRN_ERR_INVALID_ARG, "mmap of '%s' failed", path); }
    const unsigned char* b = static_cast<const unsigned char*>(map);
    uint64_t hl = 0;
    for (int i = 7; i >= 0; --i) hl = (hl << 8) | b[i];
    if (hl > map_len - 8) fail(BRN_ERR_INVALID_ARG, "'%s': header length %llu exceeds the file", path, (unsigned long long)hl);
    const char* data = reinterpret_cast<const char*>(b) + 8 + hl;
    const size_t data_len = map_len - 8 - (size_t)hl;
    Cursor c{reinterpret_cast<const char*>(b) + 8, reinterpret_cast<const char*>(b) + 8 + hl};
    auto trailer = [&] {      // after the closing brace only padding (spaces, as the safetensors writers emit) may follow
        c.ws();
        if (c.p != c.e) fail(BRN_ERR_INVALID_ARG, "'%s': bytes after the header's JSON object", path);
    };
    std::map<std::string, int> seen;
    std::vector<std::pair<long long, long long>> spans;     // [b0, b1) of every tensor, for the overlap check
    c.need('{');
    if (c.eat('}')) { trailer(); return; }
    do {
        const std::string name = c.str();
        c.need(':');
        if (seen[name]++) fail(BRN_ERR_INVALID_ARG, "'%s': tensor name '%s' appears twice", path, name.c_str());
        if (name == "__metadata__") { c.skip_value(); continue; }
        std::string dtype; std::vector<int64_t> shape; long long b0 = -1, b1 = -1;
        c.need('{');
        do {
            const std::string key = c.str();
            c.need(':');
            if (key == "dtype") dtype = c.str();
            else if (key == "shape") {
                c.need('[');
                if (!c.eat(']')) { do shape.push_back(c.num()); while (c.eat(',')); c.need(']'); }
            } else if (key == "data_offsets") { c.need('['); b0 = c.num(); c.need(','); b1 = c.num(); c.need(']'); }
            else c.skip_value();
        } while (c.eat(','));
        c.need('}');
        if (b0 < 0 || b1 < b0 || (size_t)b1 > data_len) fail(BRN_ERR_INVALID_ARG, "'%s': tensor '%s' has data_offsets outside the file", path, name.c_str());
        if (b1 > b0) spans.emplace_back(b0, b1);
        const size_t esz = dtype == "F32" ? 4 : (dtype == "F16" || dtype == "BF16") ? 2 : 0;
        if (!esz) continue;                                  // integer / f64 tensors: nothing on this path reads them
        size_t numel = 1;
        for (int64_t d : shape) {                            // checked product: a wrapped shape must not pass the size test below
            if (d < 0 || (d > 0 && numel > (SIZE_MAX / esz) / (size_t)d)) fail(BRN_ERR_SHAPE, "'%s': tensor '%s' has an impossible shape", path, name.c_str());
            numel *= (size_t)d;
        }
        if (numel * esz != (size_t)(b1 - b0)) fail(BRN_ERR_SHAPE, "'%s': tensor '%s' byte size does not match its shape", path, name.c_str());
        Entry en;
        en.name = name; en.shape = shape;
        if (dtype == "F32" && ((uintptr_t)(data + b0) & 3) == 0) {
            en.data = reinterpret_cast<const float*>(data + b0);
        } else {
            converted.emplace_back(numel);
            float* dst = converted.back().data();
            if (dtype == "F32") std::memcpy(dst, data + b0, numel * 4);
            else {
                const uint16_t* src = reinterpret_cast<const uint16_t*>(data + b0);
                for (size_t i = 0; i < numel; ++i) {
                    uint16_t h; std::memcpy(&h, src + i, 2);
                    if (dtype == "BF16") { uint32_t u = (uint32_t)h << 16; std::memcpy(dst + i, &u, 4); }
                    else dst[i] = half_to_float(h);
                }
            }
            en.data = dst;
        }
        entries.push_back(std::move(en));
    } while (c.eat(','));
    c.need('}');
    trailer();
    std::sort(spans.begin(), spans.end());                   // the safetensors crate rejects overlapping tensors; so do we
    for (size_t i = 1; i < spans.size(); ++i)
        if (spans[i].first < spans[i - 1].second) fail(BRN_ERR_INVALID_ARG, "'%s': tensors overlap in the data section", path);
}

std::vector<brn_named_tensor> SafetensorsFile::named(const char* prefix) const {
    const std::string pre = prefix ? prefix : "";
    std::vector<brn_named_tensor> out;
    for (const Entry& e : entries) {
        if (e.name.compare(0, pre.size(), pre) != 0) continue;
        brn_named_tensor t;
        t.name = e.name.c_str() + pre.size();
        t.data = e.data;
        t.shape = e.shape.data();
        t.ndim = (int)e.shape.size();
        out.push_back(t);
    }
    return out;
}

}  // namespace brn
