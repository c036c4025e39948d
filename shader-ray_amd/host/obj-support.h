// obj-support.h -- Wavefront OBJ subset loader (reference: obj-support.h:32-78).
//
// Understands `v`, `vn`, `vt`, `f` (and reports `o`); every other line type is
// ignored.  Faces of any size are fan-triangulated around their first corner.
// When the file has no `vn` lines at all, per-vertex normals are the
// normalised sum of the (area-weighted) face normals of every triangle that
// uses the position (obj-support.cpp:104-146 upstream).
#pragma once

#include <string>
#include <vector>

#include "triangle-set.h"
#include "vectormath.h"

class Obj {
public:
    Obj();
    ~Obj();

    bool load_object_from_file(const std::string &filename);
    bool load_object_from_text(const char *text, size_t length);
    bool fill_triangle_set(triangle_set_ptr triangles);

    size_t position_count() const { return positions.size(); }
    size_t face_count() const { return face_first.size(); }

private:
    static const unsigned int HAS_POSITION = 0x1;
    static const unsigned int HAS_NORMAL = 0x2;
    static const unsigned int HAS_TEXCOORD = 0x4;

    struct corner {
        unsigned int v, vn, vt;
    };

    std::vector<vec3> positions;
    std::vector<vec3> normals;
    std::vector<vec3> texcoords;

    // face f owns corners [face_first[f], face_first[f] + face_size[f])
    std::vector<corner> corners;
    std::vector<size_t> face_first;
    std::vector<unsigned int> face_size;
    std::vector<unsigned int> face_attribs;

    std::vector<std::string> object_names;      // the `o` lines, in file order

    void scan_lines(const char *begin, const char *end);
    void append_all(std::vector<Obj> &piece);
    void parse_attribute(const char *begin, const char *end, vec3 &out);
    void parse_face(const char *begin, const char *end);
    void synthesize_normals();
    bool indices_in_range(std::string *why) const;
};
