// gndt_io.cpp — host-side reader for .pcd files (the format pcl::io::loadPCDFile reads at src/publisher.cpp:19),
// written from the published PCD v0.7 description: a text header (VERSION, FIELDS, SIZE, TYPE, COUNT, WIDTH,
// HEIGHT, VIEWPOINT, POINTS, DATA) followed by `DATA ascii` (one point per line) or `DATA binary` (POINTS records
// of sum(SIZE*COUNT) bytes).  Only what the path needs is kept: where x, y, z sit in a record.  The payload is
// handed on as it is in the file (binary) or as packed xyz (ascii); NaN stripping and unpacking happen on the GPU
// (gndt_pack.hpp).  `DATA binary_compressed` is PCL's own container: two uint32 (compressed size, uncompressed size),
// then an LZF stream (Marc Lehmann's liblzf format, restated below from its published description) whose output holds
// the fields one after the other (all x, then all y, ...: structure of arrays); it is de-interleaved into records here.
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <sstream>
#include <string>
#include <vector>

#include "gndt.h"

namespace {

void set_err(char err[256], const std::string& m) {
    if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; }
}

// LZF decompression.  A control byte c < 32 starts a literal run of c + 1 bytes; otherwise it is a back reference of
// length (c >> 5) + 2 (a length field of 7 is extended by the next byte) at distance ((c & 31) << 8 | next byte) + 1.
// Returns the number of bytes produced, or 0 on a malformed stream.
size_t lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return 0;
            memcpy(out + op, in + ip, run);
            ip += run; op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return 0; len += in[ip++]; }
            if (ip >= in_len) return 0;
            const size_t dist = ((size_t)(ctrl & 31u) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return 0;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];     // may overlap: byte by byte
        }
    }
    return op;
}

std::vector<std::string> split(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string t;
    while (ss >> t) out.push_back(t);
    return out;
}

}  // namespace

extern "C" {

void gndt_pcd_free(gndt_pcd* p) {
    if (p && p->data) { free(p->data); p->data = nullptr; }
}

int gndt_pcd_read(const char* path, gndt_pcd* out, char err[256]) {
    if (!path || !out) return GNDT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, std::string("cannot open ") + path + ": " + strerror(errno)); return GNDT_ERR_INVALID; }
    std::vector<std::string> fields, types;
    std::vector<uint32_t> sizes, counts;
    uint64_t width = 0, height = 1, points = 0;
    bool have_points = false;
    std::string data_kind;
    char buf[4096];
    while (fgets(buf, sizeof buf, f)) {
        std::string line(buf);
        if (!line.empty() && line[0] == '#') continue;
        std::vector<std::string> t = split(line);
        if (t.empty()) continue;
        const std::string& k = t[0];
        if (k == "FIELDS" || k == "COLUMNS") fields.assign(t.begin() + 1, t.end());
        else if (k == "SIZE") { sizes.clear(); for (size_t i = 1; i < t.size(); ++i) sizes.push_back((uint32_t)strtoul(t[i].c_str(), nullptr, 10)); }
        else if (k == "TYPE") types.assign(t.begin() + 1, t.end());
        else if (k == "COUNT") { counts.clear(); for (size_t i = 1; i < t.size(); ++i) counts.push_back((uint32_t)strtoul(t[i].c_str(), nullptr, 10)); }
        else if (k == "WIDTH" && t.size() > 1) width = strtoull(t[1].c_str(), nullptr, 10);
        else if (k == "HEIGHT" && t.size() > 1) height = strtoull(t[1].c_str(), nullptr, 10);
        else if (k == "POINTS" && t.size() > 1) { points = strtoull(t[1].c_str(), nullptr, 10); have_points = true; }
        else if (k == "DATA" && t.size() > 1) { data_kind = t[1]; break; }
    }
    auto fail = [&](const std::string& m) { fclose(f); set_err(err, std::string(path) + ": " + m); return GNDT_ERR_INVALID; };
    if (data_kind.empty()) return fail("no DATA line (not a PCD file?)");
    if (fields.empty() || sizes.size() != fields.size()) return fail("FIELDS / SIZE missing or inconsistent");
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (types.empty()) types.assign(fields.size(), "F");
    if (counts.size() != fields.size() || types.size() != fields.size()) return fail("TYPE / COUNT inconsistent with FIELDS");
    // Every size below comes from the header: nothing is multiplied, allocated or copied before it has been bounded
    // (a crafted POINTS / WIDTH x HEIGHT / SIZE x COUNT must not wrap 64 bits or outrun the file).
    constexpr uint64_t kMaxPoints = 0xFFFFFFFEull;          // the build path indexes points with 32 bits
    constexpr uint64_t kMaxStep = 1u << 20;                  // a single record beyond 1 MiB is not a point cloud
    if (!have_points) {
        if (height && width > kMaxPoints / height) return fail("WIDTH x HEIGHT exceeds the 32-bit point index");
        points = width * height;
    }
    if (points == 0) return fail("POINTS is zero");
    if (points > kMaxPoints) return fail("POINTS exceeds the 32-bit point index of the build path");
    uint64_t step64 = 0, column64 = 0;
    uint32_t off[3] = {0, 0, 0}, col[3] = {0, 0, 0};
    bool found[3] = {false, false, false};
    for (size_t i = 0; i < fields.size(); ++i) {
        if (sizes[i] == 0 || sizes[i] > 8 || counts[i] > kMaxStep) return fail("SIZE must be 1..8 and COUNT reasonable");
        for (int a = 0; a < 3; ++a)
            if (fields[i] == std::string(1, "xyz"[a])) {
                if (sizes[i] != 4 || types[i] != "F" || counts[i] != 1) return fail("x/y/z must be single 4-byte floats (pcl::PointXYZ)");
                off[a] = (uint32_t)step64; col[a] = (uint32_t)column64; found[a] = true;
            }
        step64 += (uint64_t)sizes[i] * counts[i];
        column64 += counts[i];
        if (step64 > kMaxStep) return fail("record size (sum of SIZE x COUNT) is implausibly large");
    }
    if (!found[0] || !found[1] || !found[2]) return fail("no x, y, z fields");
    const uint32_t step = (uint32_t)step64, column = (uint32_t)column64;
    // what is left of the file bounds every payload (binary: exactly; ascii: at least 2 bytes per value and line)
    uint64_t remaining = 0;
    {
        struct stat st;
        const long here = ftell(f);
        if (here < 0 || fstat(fileno(f), &st) != 0 || st.st_size < here) return fail("cannot size the file");
        remaining = (uint64_t)st.st_size - (uint64_t)here;
    }
    out->num_points = points;
    if (data_kind == "binary") {
        out->data_kind = 1;
        out->layout.point_step = step; out->layout.offset_x = off[0]; out->layout.offset_y = off[1]; out->layout.offset_z = off[2];
        if (points > remaining / step) return fail("payload shorter than POINTS * record size");
        const size_t bytes = (size_t)points * step;
        out->data = malloc(bytes ? bytes : 1);
        if (!out->data) return fail("out of memory");
        if (fread(out->data, 1, bytes, f) != bytes) { free(out->data); out->data = nullptr; return fail("payload shorter than POINTS * record size"); }
    } else if (data_kind == "ascii") {
        out->data_kind = 0;
        out->layout.point_step = 12; out->layout.offset_x = 0; out->layout.offset_y = 4; out->layout.offset_z = 8;
        if (points > remaining / 2) return fail("fewer lines than POINTS");       // a data line takes at least "0\n"
        float* xyz = (float*)malloc((size_t)points * 12);
        if (!xyz) return fail("out of memory");
        out->data = xyz;
        std::string line;
        std::vector<char> big(1 << 16);
        for (uint64_t i = 0; i < points; ++i) {
            if (!fgets(big.data(), (int)big.size(), f)) { free(xyz); out->data = nullptr; return fail("fewer lines than POINTS"); }
            // columns are separated by blanks; "nan" parses through strtof
            const char* p = big.data();
            for (uint32_t c = 0; c < column; ++c) {
                char* end = nullptr;
                const float v = strtof(p, &end);
                if (end == p) { free(xyz); out->data = nullptr; return fail("unparsable value on a data line"); }
                for (int a = 0; a < 3; ++a)
                    if (c == col[a]) xyz[3 * i + a] = v;
                p = end;
            }
        }
    } else if (data_kind == "binary_compressed") {
        out->data_kind = 2;
        out->layout.point_step = step; out->layout.offset_x = off[0]; out->layout.offset_y = off[1]; out->layout.offset_z = off[2];
        uint32_t sizes2[2];
        if (fread(sizes2, 4, 2, f) != 2) return fail("binary_compressed: missing size words");
        const size_t csize = sizes2[0], usize = sizes2[1];
        if (points > 0xFFFFFFFFull / step || usize != (size_t)points * step) return fail("binary_compressed: uncompressed size is not POINTS * record size");
        if (csize > remaining - 8) return fail("binary_compressed: payload shorter than its size word");
        std::vector<unsigned char> comp(csize ? csize : 1), soa(usize ? usize : 1);
        if (fread(comp.data(), 1, csize, f) != csize) return fail("binary_compressed: payload shorter than its size word");
        if (usize && lzf_decompress(comp.data(), csize, soa.data(), usize) != usize) return fail("binary_compressed: malformed LZF stream");
        unsigned char* rec = (unsigned char*)malloc(usize ? usize : 1);
        if (!rec) return fail("out of memory");
        out->data = rec;
        // structure of arrays (field after field, each with POINTS * SIZE * COUNT bytes) -> array of records
        size_t src = 0, foff = 0;
        for (size_t fi = 0; fi < fields.size(); ++fi) {
            const size_t fbytes = (size_t)sizes[fi] * counts[fi];
            for (uint64_t i = 0; i < points; ++i) memcpy(rec + i * step + foff, soa.data() + src + i * fbytes, fbytes);
            src += (size_t)points * fbytes;
            foff += fbytes;
        }
    } else {
        return fail("DATA " + data_kind + " is not supported (ascii, binary and binary_compressed are)");
    }
    fclose(f);
    return GNDT_OK;
}

}  // extern "C"
