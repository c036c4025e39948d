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
// file in place (mapped read-only, round 6); since round 4 in parallel: the text is cut at line ends into one piece per
// thread, each piece is scanned into arrays of its own and the arrays are copied, every piece by its own thread, one behind
// the other in file order (a face's indices are absolute, so nothing has to be renumbered); the synthesized normals are
// summed per vertex in face order by the thread that owns the vertex (the float sums are those of the serial loop), and the
// triangles go to triangle_set::add_bulk.  Same arrays, same triangle_set, bit for bit (tests/test_loaders.py compares
// with SHRAY_LOAD_THREADS=1; tests/test_host_vs_reference.py with the reference's own loader).
#include "obj-support.h"

#include "host-log.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <new>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
    const int threads = host_load_threads();
    // every corner's normal index is its position index; the faces have normals now
    host_in_parallel(threads, [&](int j) {
        const size_t lo = face_first.size() * (size_t)j / threads, hi = face_first.size() * (size_t)(j + 1) / threads;
        for (size_t f = lo; f < hi; f++) {
            if (face_size[f] >= 3)
                face_attribs[f] |= HAS_NORMAL;
            corner *fc = &corners[face_first[f]];
            for (unsigned int k = 0; k < face_size[f]; k++)
                if (face_size[f] >= 3)
                    fc[k].vn = fc[k].v;
        }
    });
    // a vertex's normal is the sum of the face normals of the triangles that use it, added in face order, fan order and
    // corner order (obj-support.cpp:104-146): thread j adds for the vertices [lo, hi) it owns and walks every face
    host_in_parallel(threads, [&](int j) {
        const size_t lo = positions.size() * (size_t)j / threads, hi = positions.size() * (size_t)(j + 1) / threads;
        for (size_t f = 0; f < face_first.size(); f++) {
            const corner *fc = &corners[face_first[f]];
            const unsigned int n = face_size[f];
            for (unsigned int k = 1; k + 1 < n; k++) {
                const unsigned int v[3] = {fc[0].v, fc[k].v, fc[k + 1].v};
                if (!((v[0] >= lo && v[0] < hi) || (v[1] >= lo && v[1] < hi) || (v[2] >= lo && v[2] < hi)))
                    continue;
                const vec3 face_normal = cross(positions[v[1]] - positions[v[0]], positions[v[2]] - positions[v[0]]);
                for (int c = 0; c < 3; c++)
                    if (v[c] >= lo && v[c] < hi)
                        normals[v[c]] = normals[v[c]] + face_normal;
            }
        }
    });
    host_in_parallel(threads, [&](int j) {
        const size_t lo = normals.size() * (size_t)j / threads, hi = normals.size() * (size_t)(j + 1) / threads;
        for (size_t k = lo; k < hi; k++)
            normals[k] = normalize(normals[k]);
    });
}

bool Obj::indices_in_range(std::string *why) const
{
    const int threads = host_load_threads();
    std::vector<char> found((size_t)threads, 0);     // 1: a short face, 2: a position that does not exist (the first in file order is reported)
    std::vector<size_t> found_at((size_t)threads, 0);
    host_in_parallel(threads, [&](int j) {
        const size_t lo = face_first.size() * (size_t)j / threads, hi = face_first.size() * (size_t)(j + 1) / threads;
        for (size_t f = lo; f < hi && !found[(size_t)j]; f++) {
            if (face_size[f] < 3) {
                found[(size_t)j] = 1;
                found_at[(size_t)j] = f;
                break;
            }
            for (unsigned int k = 0; k < face_size[f]; k++)
                if (corners[face_first[f] + k].v >= positions.size()) {
                    found[(size_t)j] = 2;
                    found_at[(size_t)j] = f;
                    break;
                }
        }
    });
    for (int j = 0; j < threads; j++)
        if (found[(size_t)j]) {
            *why = found[(size_t)j] == 1 ? "a face has fewer than 3 corners" : "a face names a position that does not exist";
            return false;
        }
    return true;
}

void Obj::scan_lines(const char *text, const char *text_end)
{
    const char *p = text;
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
            object_names.emplace_back(payload, (size_t)(eol - payload));
        }
    }
}

// the arrays of the file's pieces one behind the other, in file order (a face's indices are absolute: nothing is renumbered)
void Obj::append_all(std::vector<Obj> &piece)
{
    const size_t n = piece.size();
    std::vector<size_t> at_position(n + 1, positions.size()), at_normal(n + 1, normals.size()), at_texcoord(n + 1, texcoords.size()),
        at_corner(n + 1, corners.size()), at_face(n + 1, face_first.size());
    for (size_t j = 0; j < n; j++) {
        at_position[j + 1] = at_position[j] + piece[j].positions.size();
        at_normal[j + 1] = at_normal[j] + piece[j].normals.size();
        at_texcoord[j + 1] = at_texcoord[j] + piece[j].texcoords.size();
        at_corner[j + 1] = at_corner[j] + piece[j].corners.size();
        at_face[j + 1] = at_face[j] + piece[j].face_first.size();
    }
    positions.resize(at_position[n]);
    normals.resize(at_normal[n]);
    texcoords.resize(at_texcoord[n]);
    corners.resize(at_corner[n]);
    face_first.resize(at_face[n]);
    face_size.resize(at_face[n]);
    face_attribs.resize(at_face[n]);
    host_in_parallel((int)n, [&](int k) {
        const size_t j = (size_t)k;
        Obj &chunk = piece[j];
        std::copy(chunk.positions.begin(), chunk.positions.end(), positions.begin() + (ptrdiff_t)at_position[j]);
        std::copy(chunk.normals.begin(), chunk.normals.end(), normals.begin() + (ptrdiff_t)at_normal[j]);
        std::copy(chunk.texcoords.begin(), chunk.texcoords.end(), texcoords.begin() + (ptrdiff_t)at_texcoord[j]);
        std::copy(chunk.corners.begin(), chunk.corners.end(), corners.begin() + (ptrdiff_t)at_corner[j]);
        for (size_t f = 0; f < chunk.face_first.size(); f++)
            face_first[at_face[j] + f] = chunk.face_first[f] + at_corner[j];
        std::copy(chunk.face_size.begin(), chunk.face_size.end(), face_size.begin() + (ptrdiff_t)at_face[j]);
        std::copy(chunk.face_attribs.begin(), chunk.face_attribs.end(), face_attribs.begin() + (ptrdiff_t)at_face[j]);
    });
    for (Obj &chunk : piece)
        for (std::string &name : chunk.object_names)
            object_names.push_back(std::move(name));
}

bool Obj::load_object_from_text(const char *text, size_t length)
{
    // one piece per thread, cut behind a line end; pieces of less than 256 KB are not worth a thread
    const int threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_load_threads(), length >> 18));
    std::vector<const char *> cut(1, text);
    for (int j = 1; j < threads; j++) {
        const char *at = text + length * (size_t)j / (size_t)threads;
        if (at < cut.back())
            at = cut.back();
        const char *eol = (const char *)memchr(at, '\n', (size_t)(text + length - at));
        cut.push_back(eol ? eol + 1 : text + length);
    }
    cut.push_back(text + length);
    if (threads == 1) {
        scan_lines(text, text + length);
    } else {
        std::vector<Obj> piece((size_t)threads);
        host_in_parallel(threads, [&](int j) { piece[(size_t)j].scan_lines(cut[(size_t)j], cut[(size_t)j + 1]); });
        append_all(piece);
    }
    for (const std::string &name : object_names)
        host_info("obj: object '%s'\n", name.c_str());

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
    // the file mapped read-only and scanned where it lies; what cannot be mapped (a pipe, an empty file) is read
    const int fd = open(filename.c_str(), O_RDONLY);
    if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map != MAP_FAILED) {
                close(fd);
                const bool ok = load_object_from_text((const char *)map, (size_t)st.st_size);
                munmap(map, (size_t)st.st_size);
                return ok;
            }
        }
        close(fd);
    }
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
    // triangle_first[f] = the number of the first fan triangle of face f
    std::vector<size_t> triangle_first(face_first.size() + 1, 0);
    for (size_t f = 0; f < face_first.size(); f++)
        triangle_first[f + 1] = triangle_first[f] + (face_size[f] >= 3 ? face_size[f] - 2 : 0);
    // (raw storage: every corner is constructed by the thread that fills it in, not all of them here first)
    struct raw_free {
        void operator()(vertex *p) const { ::operator delete((void *)p); }
    };
    std::unique_ptr<vertex, raw_free> tri_corners((vertex *)::operator new(sizeof(vertex) * std::max<size_t>(1, 3 * triangle_first.back())));
    const int threads = host_load_threads();
    std::vector<char> missing_normal((size_t)threads, 0);
    host_in_parallel(threads, [&](int j) {
        const size_t lo = face_first.size() * (size_t)j / threads, hi = face_first.size() * (size_t)(j + 1) / threads;
        for (size_t f = lo; f < hi; f++) {
            const corner *fc = &corners[face_first[f]];
            const bool with_normals = (face_attribs[f] & HAS_NORMAL) != 0;
            vertex *out = tri_corners.get() + 3 * triangle_first[f];
            for (unsigned int k = 1; k + 1 < face_size[f]; k++) {
                const corner *pick[3] = {&fc[0], &fc[k], &fc[k + 1]};
                for (int c = 0; c < 3; c++, out++) {
                    new (out) vertex();
                    out->v = positions[pick[c]->v];
                    if (with_normals) {
                        if (pick[c]->vn >= normals.size()) {
                            missing_normal[(size_t)j] = 1;
                            continue;
                        }
                        out->n = normals[pick[c]->vn];
                    }
                    out->c = vec3(1.0f, 1.0f, 1.0f);
                }
            }
        }
    });
    for (char missing : missing_normal)
        if (missing) {
            fprintf(stderr, "obj: a face names a normal that does not exist\n");
            return false;
        }
    triangles->add_bulk(tri_corners.get(), triangle_first.back(), threads);
    return true;
}
