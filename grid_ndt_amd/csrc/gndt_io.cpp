// gndt_io.cpp — host-side reader for .pcd files (the format pcl::io::loadPCDFile reads at src/publisher.cpp:19),
// written from the published PCD v0.7 description: a text header (VERSION, FIELDS, SIZE, TYPE, COUNT, WIDTH,
// HEIGHT, VIEWPOINT, POINTS, DATA) followed by `DATA ascii` (one point per line) or `DATA binary` (POINTS records
// of sum(SIZE*COUNT) bytes).  Only what the path needs is kept: where x, y, z sit in a record.  The payload is
// handed on as it is in the file (binary) or as packed xyz (ascii); NaN stripping and unpacking happen on the GPU
// (gndt_pack.hpp).  `binary_compressed` (LZF) is not read.
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sstream>
#include <string>
#include <vector>

#include "gndt.h"

namespace {

void set_err(char err[256], const std::string& m) {
    if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; }
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
    if (!have_points) points = width * height;
    uint32_t step = 0, off[3] = {0, 0, 0}, col[3] = {0, 0, 0};
    bool found[3] = {false, false, false};
    uint32_t column = 0;
    for (size_t i = 0; i < fields.size(); ++i) {
        for (int a = 0; a < 3; ++a)
            if (fields[i] == std::string(1, "xyz"[a])) {
                if (sizes[i] != 4 || types[i] != "F") return fail("x/y/z must be 4-byte floats (pcl::PointXYZ)");
                off[a] = step; col[a] = column; found[a] = true;
            }
        step += sizes[i] * counts[i];
        column += counts[i];
    }
    if (!found[0] || !found[1] || !found[2]) return fail("no x, y, z fields");
    out->num_points = points;
    if (data_kind == "binary") {
        out->data_kind = 1;
        out->layout.point_step = step; out->layout.offset_x = off[0]; out->layout.offset_y = off[1]; out->layout.offset_z = off[2];
        const size_t bytes = (size_t)points * step;
        out->data = malloc(bytes ? bytes : 1);
        if (!out->data) return fail("out of memory");
        if (fread(out->data, 1, bytes, f) != bytes) { free(out->data); out->data = nullptr; return fail("payload shorter than POINTS * record size"); }
    } else if (data_kind == "ascii") {
        out->data_kind = 0;
        out->layout.point_step = 12; out->layout.offset_x = 0; out->layout.offset_y = 4; out->layout.offset_z = 8;
        float* xyz = (float*)malloc(points ? points * 12 : 1);
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
    } else {
        return fail("DATA " + data_kind + " is not supported (ascii and binary are)");
    }
    fclose(f);
    return GNDT_OK;
}

}  // extern "C"
