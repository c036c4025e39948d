// obj-support.cpp -- Wavefront OBJ subset loader.
//
// Line handling follows the reference (obj-support.cpp:226-318): the line
// type is the text before the first blank, the payload starts at the next
// non-blank, payload fields are separated by runs of blanks -- and, as
// upstream, a payload that ENDS in a blank yields one extra empty field
// (split_tuple_fuzzy, obj-support.cpp:61-82).  For an attribute line more
// than three fields is reported and leaves the attribute at 0; for a face the
// empty field becomes an extra corner with index 0 and clears the face's
// attribute mask.  Face corners are `v`, `v/vt`, `v//vn` or `v/vt/vn`, 1-based
// (obj-support.cpp:171-206).  Indices that do not name an existing attribute
// are undefined behaviour upstream; here they fail the load.
//
// Unlike the reference this parser keeps faces in flat arrays and scans the
// file in place, so the 1M-triangle benchmark mesh loads in well under a second.
#include "obj-support.h"

#include "host-log.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

// Fields separated by runs of ' ', reference semantics (see file header).
template <class F>
void for_each_field(const char *begin, const char *end, F &&fn)
{
    const char *p = begin;
    for (;;) {
        const char *q = p;
        while (q < end && *q != ' ')
            q++;
        fn(p, q);
        if (q == end)
            return;
        while (q < end && *q == ' ')
            q++;
        p = q;
        if (p == end) {
            fn(p, p);   // trailing blank -> one empty field
            return;
        }
    }
}

// Decimal text -> float, stream-extraction style: leading junk gives 0.
float field_to_float(const char *begin, const char *end)
{
    char buf[64];
    size_t n = std::min<size_t>(sizeof(buf) - 1, (size_t)(end - begin));
    memcpy(buf, begin, n);
    buf[n] = 0;
    return strtof(buf, nullptr);
}

// Unsigned decimal, 0 when absent / malformed, wrapping like istream >> unsigned.
unsigned int field_to_uint(const char *begin, const char *end)
{
    char buf[32];
    size_t n = std::min<size_t>(sizeof(buf) - 1, (size_t)(end - begin));
    memcpy(buf, begin, n);
    buf[n] = 0;
    return (unsigned int)strtoul(buf, nullptr, 10);
}

}   // namespace

Obj::Obj() {}
Obj::~Obj() {}

void Obj::parse_attribute(const char *begin, const char *end, vec3 &out)
{
    float value[3] = {0, 0, 0};
    int fields = 0;
    for_each_field(begin, end, [&](const char *b, const char *e) {
        if (fields < 3)
            value[fields] = field_to_float(b, e);
        fields++;
    });
    if (fields >= 1 && fields <= 3) {
        // only the fields present are assigned (obj-support.cpp:156-169)
        if (fields >= 1) out.x = value[0];
        if (fields >= 2) out.y = value[1];
        if (fields >= 3) out.z = value[2];
    } else {
        fprintf(stderr, "obj: attribute line with %d fields ignored\n", fields);
    }
}

void Obj::parse_face(const char *begin, const char *end)
{
    const size_t first = corners.size();
    unsigned int mask = 0;
    for_each_field(begin, end, [&](const char *b, const char *e) {
        corner c = {0, 0, 0};
        mask = 0;
        if (b != e) {
            // split on '/', keeping empty pieces between slashes
            const char *piece[3] = {nullptr, nullptr, nullptr};
            const char *piece_end[3] = {nullptr, nullptr, nullptr};
            int pieces = 0;
            const char *p = b;
            while (pieces < 3) {
                const char *q = p;
                while (q < e && *q != '/')
                    q++;
                piece[pieces] = p;
                piece_end[pieces] = q;
                pieces++;
                if (q == e)
                    break;
                p = q + 1;
                if (p == e)
                    break;   // trailing '/' adds no piece
            }
            mask |= HAS_POSITION;
            c.v = field_to_uint(piece[0], piece_end[0]) - 1;
            if (pieces > 1 && piece[1] != piece_end[1]) {
                mask |= HAS_TEXCOORD;
                c.vt = field_to_uint(piece[1], piece_end[1]) - 1;
            }
            if (pieces > 2 && piece[2] != piece_end[2]) {
                mask |= HAS_NORMAL;
                c.vn = field_to_uint(piece[2], piece_end[2]) - 1;
            }
        }
        corners.push_back(c);
    });
    face_first.push_back(first);
    face_size.push_back((unsigned int)(corners.size() - first));
    face_attribs.push_back(mask);   // the last corner's mask stands for the face
}

void Obj::synthesize_normals()
{
    normals.assign(positions.size(), vec3(0.0f));
    for (size_t f = 0; f < face_first.size(); f++) {
        corner *fc = &corners[face_first[f]];
        const unsigned int n = face_size[f];
        for (unsigned int k = 1; k + 1 < n; k++) {
            corner &c0 = fc[0], &c1 = fc[k], &c2 = fc[k + 1];
            const vec3 face_normal = cross(positions[c1.v] - positions[c0.v], positions[c2.v] - positions[c0.v]);
            face_attribs[f] |= HAS_NORMAL;
            c0.vn = c0.v;
            c1.vn = c1.v;
            c2.vn = c2.v;
            normals[c0.vn] = normals[c0.vn] + face_normal;
            normals[c1.vn] = normals[c1.vn] + face_normal;
            normals[c2.vn] = normals[c2.vn] + face_normal;
        }
    }
    for (vec3 &n : normals)
        n = normalize(n);
}

bool Obj::indices_in_range(std::string *why) const
{
    for (size_t f = 0; f < face_first.size(); f++) {
        if (face_size[f] < 3) {
            *why = "a face has fewer than 3 corners";
            return false;
        }
        for (unsigned int k = 0; k < face_size[f]; k++) {
            const corner &c = corners[face_first[f] + k];
            if (c.v >= positions.size()) {
                *why = "a face names a position that does not exist";
                return false;
            }
        }
    }
    return true;
}

bool Obj::load_object_from_text(const char *text, size_t length)
{
    const char *p = text;
    const char *const text_end = text + length;
    while (p < text_end) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(text_end - p));
        if (!eol)
            eol = text_end;
        const char *line = p;
        p = (eol < text_end) ? eol + 1 : text_end;
        if (line == eol || *line == '#')
            continue;

        const char *blank = (const char *)memchr(line, ' ', (size_t)(eol - line));
        const char *type_end = blank ? blank : eol;
        const char *payload = eol;
        if (blank) {
            payload = blank;
            while (payload < eol && *payload == ' ')
                payload++;
        }
        const size_t type_len = (size_t)(type_end - line);

        if (type_len == 1 && line[0] == 'v') {
            vec3 v(0.0f);
            parse_attribute(payload, eol, v);
            positions.push_back(v);
        } else if (type_len == 2 && line[0] == 'v' && line[1] == 'n') {
            vec3 v(0.0f);
            parse_attribute(payload, eol, v);
            normals.push_back(v);
        } else if (type_len == 2 && line[0] == 'v' && line[1] == 't') {
            vec3 v(0.0f);
            parse_attribute(payload, eol, v);
            texcoords.push_back(v);
        } else if (type_len == 1 && line[0] == 'f') {
            parse_face(payload, eol);
        } else if (type_len == 1 && line[0] == 'o') {
            host_info("obj: object '%.*s'\n", (int)(eol - payload), payload);
        }
    }

    host_info("obj: %zu faces, %zu positions, %zu normals, %zu texcoords\n", face_first.size(), positions.size(),
            normals.size(), texcoords.size());

    std::string why;
    if (!indices_in_range(&why)) {
        fprintf(stderr, "obj: %s\n", why.c_str());
        return false;
    }
    if (normals.empty()) {
        host_info("obj: no normals in file, computing area-weighted vertex normals\n");
        synthesize_normals();
    }
    return true;
}

bool Obj::load_object_from_file(const std::string &filename)
{
    FILE *fp = fopen(filename.c_str(), "rb");
    if (!fp)
        return false;
    std::string text;
    char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0)
        text.append(chunk, got);
    fclose(fp);
    return load_object_from_text(text.data(), text.size());
}

bool Obj::fill_triangle_set(triangle_set_ptr triangles)
{
    for (size_t f = 0; f < face_first.size(); f++) {
        const corner *fc = &corners[face_first[f]];
        const bool with_normals = (face_attribs[f] & HAS_NORMAL) != 0;
        for (unsigned int k = 1; k + 1 < face_size[f]; k++) {
            const corner *pick[3] = {&fc[0], &fc[k], &fc[k + 1]};
            vertex vtx[3];
            for (int j = 0; j < 3; j++) {
                vtx[j].v = positions[pick[j]->v];
                if (with_normals) {
                    if (pick[j]->vn >= normals.size()) {
                        fprintf(stderr, "obj: a face names a normal that does not exist\n");
                        return false;
                    }
                    vtx[j].n = normals[pick[j]->vn];
                }
                vtx[j].c = vec3(1.0f, 1.0f, 1.0f);
            }
            triangles->add(vtx[0], vtx[1], vtx[2]);
        }
    }
    return true;
}
