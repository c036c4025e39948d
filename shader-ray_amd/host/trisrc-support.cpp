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
// float32 conversion), which parses the 30 MB bunny-class file several times
// faster.
#include "trisrc-support.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <string>

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

}   // namespace

bool ParseTriSrcText(const char *text, triangle_set_ptr triangles)
{
    const char *scale_env = getenv("GEOMETRY_SCALE");
    const float geometry_scale = scale_env ? (float)atof(scale_env) : 1.0f;
    const bool colors_are_linear = getenv("COLORS_ARE_LINEAR") != nullptr;

    cursor c{text};
    for (;;) {
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
            fprintf(stderr, "trisrc: record without a tag name\n");
            return false;
        }
        while (*c.p && !isspace((unsigned char)*c.p))
            c.p++;

        float specular[5];
        if (!c.numbers(specular, 5)) {
            fprintf(stderr, "trisrc: record without its 5 specular values\n");
            return false;
        }

        float field[3][12];
        for (int corner = 0; corner < 3; corner++) {
            if (!c.numbers(field[corner], 12)) {
                fprintf(stderr, "trisrc: vertex %d of a triangle is incomplete\n", corner);
                return false;
            }
        }
        c.skip_space();

        vertex vtx[3];
        for (int corner = 0; corner < 3; corner++) {
            const float *f = field[corner];
            vtx[corner].v = vec3(f[0], f[1], f[2]) * geometry_scale;
            vtx[corner].n = normalize(vec3(f[3], f[4], f[5]));
            if (colors_are_linear)
                vtx[corner].c.set(f[6], f[7], f[8]);
            else
                vtx[corner].c.set(gamma_to_linear(f[6]), gamma_to_linear(f[7]), gamma_to_linear(f[8]));
        }
        triangles->add(vtx[0], vtx[1], vtx[2]);
    }
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
