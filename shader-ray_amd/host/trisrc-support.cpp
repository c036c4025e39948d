// trisrc-support.cpp -- "trisrc" parser.
//
// Record grammar (trisrc-support.cpp:50-84 upstream):
//     "texture-name" tag  sr sg sb sa shininess
//     3 x ( px py pz  nx ny nz  r g b a  u v )
// Positions are scaled by GEOMETRY_SCALE (default 1), colours are raised to
// the screen gamma 2.63 unless COLORS_ARE_LINEAR is set, normals are
// normalised (trisrc-support.cpp:24-40, :92-101).  Texture name, tag,
// specular block and texture coordinates are read and dropped, as upstream.
//
// The reference scans with fscanf one field at a time; here the file is read
// once and tokenised in memory with strtof (same correctly-rounded decimal ->
// float32 conversion), in parallel pieces since round 4 (ParseTriSrcText).
#include "trisrc-support.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "host-log.h"

namespace {

const float kScreenGamma = 2.63f;

struct cursor {
    const char *p;
    void skip_space()
    {
        while (*p && isspace((unsigned char)*p))
            p++;
    }
    // One %g-style field: optional white space, then a number.
    bool number(float *out)
    {
        skip_space();
        char *end = nullptr;
        const float v = strtof(p, &end);
        if (end == p)
            return false;
        p = end;
        *out = v;
        return true;
    }
    bool numbers(float *out, int n)
    {
        for (int k = 0; k < n; k++)
            if (!number(out + k))
                return false;
        return true;
    }
};

inline float gamma_to_linear(float c) { return (float)pow((double)c, (double)kScreenGamma); }

// Parses records from `begin` on; stops at `end` (when a record ends exactly there) or at the first thing that is not a
// record.  Appends three vertices per triangle to `out`.  *stopped_at = where it stopped; returns false on a malformed
// record (after the message, unless the piece is a speculative one whose failure only means "parse serially"), as upstream.
bool parse_records(const char *begin, const char *end, float geometry_scale, bool colors_are_linear, std::vector<vertex> &out,
                   const char **stopped_at, bool speculative = false)
{
    cursor c{begin};
    for (;;) {
        *stopped_at = c.p;
        if (c.p >= end)
            return true;
        // "name": must start right here, must be non-empty
        if (*c.p != '"')
            return true;
        const char *name_begin = c.p + 1;
        const char *name_end = name_begin;
        while (*name_end && *name_end != '"')
            name_end++;
        if (name_end == name_begin)
            return true;
        c.p = (*name_end == '"') ? name_end + 1 : name_end;

        // tag: one blank-delimited word
        c.skip_space();
        if (!*c.p) {
            if (!speculative)
                fprintf(stderr, "trisrc: record without a tag name\n");
            return false;
        }
        while (*c.p && !isspace((unsigned char)*c.p))
            c.p++;

        float specular[5];
        if (!c.numbers(specular, 5)) {
            if (!speculative)
                fprintf(stderr, "trisrc: record without its 5 specular values\n");
            return false;
        }

        float field[3][12];
        for (int corner = 0; corner < 3; corner++) {
            if (!c.numbers(field[corner], 12)) {
                if (!speculative)
                    fprintf(stderr, "trisrc: vertex %d of a triangle is incomplete\n", corner);
                return false;
            }
        }
        c.skip_space();

        for (int corner = 0; corner < 3; corner++) {
            const float *f = field[corner];
            vertex vtx;
            vtx.v = vec3(f[0], f[1], f[2]) * geometry_scale;
            vtx.n = normalize(vec3(f[3], f[4], f[5]));
            if (colors_are_linear)
                vtx.c.set(f[6], f[7], f[8]);
            else
                vtx.c.set(gamma_to_linear(f[6]), gamma_to_linear(f[7]), gamma_to_linear(f[8]));
            out.push_back(vtx);
        }
    }
}

}   // namespace

// The text is parsed in pieces, one per thread, and the triangles go to triangle_set::add_bulk.  A piece may only start
// where the serial scan would start a record, and the grammar alone does not say where that is (a tag may begin with a
// quote); so pieces are cut in front of a `"` that follows a line end, every piece is parsed up to the next cut, and the
// result is kept only if every piece ended EXACTLY at the next cut on a record boundary -- then, by induction from the
// first piece, every cut was a record start of the serial scan and the pieces' records are its records in order.
// Anything else (a record that straddles a cut, a malformed record, text after the last record) is parsed again serially.
bool ParseTriSrcText(const char *text, triangle_set_ptr triangles)
{
    const char *scale_env = getenv("GEOMETRY_SCALE");
    const float geometry_scale = scale_env ? (float)atof(scale_env) : 1.0f;
    const bool colors_are_linear = getenv("COLORS_ARE_LINEAR") != nullptr;
    const size_t length = strlen(text);
    const char *const text_end = text + length;

    const int threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_load_threads(), length >> 18));
    if (threads > 1) {
        std::vector<const char *> cut(1, text);
        for (int j = 1; j < threads; j++) {
            const char *at = std::max(cut.back(), text + length * (size_t)j / (size_t)threads);
            const char *found = nullptr;
            for (const char *p = at; p + 1 < text_end; p++)
                if (p[0] == '\n' && p[1] == '"') {
                    found = p + 1;
                    break;
                }
            cut.push_back(found ? found : text_end);
        }
        cut.push_back(text_end);
        std::vector<std::vector<vertex>> piece((size_t)threads);
        std::vector<char> clean((size_t)threads, 0);
        host_in_parallel(threads, [&](int j) {
            const char *stopped = nullptr;
            piece[(size_t)j].reserve((size_t)(cut[(size_t)j + 1] - cut[(size_t)j]) / 100);
            const bool ok = parse_records(cut[(size_t)j], cut[(size_t)j + 1], geometry_scale, colors_are_linear, piece[(size_t)j], &stopped, true);
            clean[(size_t)j] = ok && stopped == cut[(size_t)j + 1];
        });
        bool all_clean = true;
        for (char c : clean)
            all_clean = all_clean && c;
        if (all_clean) {
            size_t total = 0;
            for (const auto &p : piece)
                total += p.size();
            std::vector<vertex> corners;
            corners.reserve(total);
            for (const auto &p : piece)
                corners.insert(corners.end(), p.begin(), p.end());
            triangles->add_bulk(corners.data(), corners.size() / 3, threads);
            return true;
        }
    }
    std::vector<vertex> corners;
    const char *stopped = nullptr;
    const bool ok = parse_records(text, text_end, geometry_scale, colors_are_linear, corners, &stopped);
    // (the triangles read before a malformed record stay in the set, as upstream)
    triangles->add_bulk(corners.data(), corners.size() / 3, threads);
    return ok;
}

bool ParseTriSrc(FILE *fp, triangle_set_ptr triangles)
{
    std::string text;
    char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0)
        text.append(chunk, got);
    // an embedded NUL would end the reference's scan too (no field matches it)
    return ParseTriSrcText(text.c_str(), triangles);
}
