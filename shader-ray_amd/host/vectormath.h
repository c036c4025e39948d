// vectormath.h -- float32 vector / box / 4x4 matrix helpers for the host side.
//
// API-compatible with the reference's math header (vectormath.h:22-601) for
// everything the kept loader / BVH / flattener / frame-parameter code uses:
// vec3, vec4, box3d, and the column-major float[16] helpers.  Written from
// the behaviour, not the text: every operation is a single IEEE float32
// operation in the order the reference performs it, because BVH splits and
// frame parameters must come out bit-identical.
// Build with -ffp-contract=off and without -ffast-math.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

inline float to_radians(float degrees) { return degrees / 180 * M_PI; }   // vectormath.h:22
inline float to_degrees(float radians) { return radians * 180 / M_PI; }   // vectormath.h:27

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float s) : x(s), y(s), z(s), w(1) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4 &set(float x_, float y_, float z_, float w_) { x = x_; y = y_; z = z_; w = w_; return *this; }
};

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3 &set(float x_, float y_, float z_) { x = x_; y = y_; z = z_; return *this; }
    float operator[](int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
    // Writes this vector as element `index` of a packed float3 array.
    const vec3 &store(float *packed, unsigned int index) const
    {
        float *dst = packed + 3u * index;
        dst[0] = x; dst[1] = y; dst[2] = z;
        return *this;
    }
};

inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 min(const vec3 &a, const vec3 &b) { return vec3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline vec3 max(const vec3 &a, const vec3 &b) { return vec3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3 &a, const vec3 &b)
{
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline vec3 normalize(const vec3 &a) { return a / sqrtf(dot(a, a)); }

struct ray {
    vec3 o;
    vec3 d;
};

// Axis-aligned box.  An empty box is (+FLT_MAX, -FLT_MAX); adding a point
// inflates by an absolute 1e-5 (vectormath.h:189-195), adding a box does not.
struct box3d {
    vec3 boxmin;
    vec3 boxmax;
    box3d()
        : boxmin(std::numeric_limits<float>::max()),
          boxmax(-std::numeric_limits<float>::max())
    {
    }
    box3d(const vec3 &lo, const vec3 &hi) : boxmin(lo), boxmax(hi) {}

    vec3 center() const { return (boxmin + boxmax) * .5f; }
    vec3 dim() const { return max(vec3(0.0f), boxmax - boxmin); }

    box3d &add(const vec3 &point)
    {
        const float bump = .00001f;
        boxmin = min(boxmin, point - vec3(bump));
        boxmax = max(boxmax, point + vec3(bump));
        return *this;
    }
    box3d &add(const vec3 &c, float r)
    {
        const float bump = 1.0001f;
        boxmin = min(boxmin, c - vec3(r * bump));
        boxmax = max(boxmax, c + vec3(r * bump));
        return *this;
    }
    box3d &add(const vec3 &lo, const vec3 &hi)
    {
        boxmin = min(lo, boxmin);
        boxmax = max(hi, boxmax);
        return *this;
    }
    box3d &add(const box3d &other) { return add(other.boxmin, other.boxmax); }
    box3d &add(const vec3 &a, const vec3 &b, const vec3 &c) { add(a); add(b); return add(c); }
};

// ---- column-major 4x4 matrices: element (row r, column c) is m[c * 4 + r] ----

static const float mat4_identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

inline void mat4_make_identity(float m[16]) { std::memcpy(m, mat4_identity, sizeof(mat4_identity)); }

// matrix * column vector (vectormath.h:258-272)
inline vec4 operator*(const float m[16], const vec4 &v)
{
    float out[4];
    for (int r = 0; r < 4; r++)
        out[r] = m[r] * v.x + m[4 + r] * v.y + m[8 + r] * v.z + m[12 + r] * v.w;
    return vec4(out[0], out[1], out[2], out[3]);
}

inline void mat4_make_translation(float x, float y, float z, float m[16])
{
    mat4_make_identity(m);
    m[12] = x; m[13] = y; m[14] = z;
}

inline void mat4_make_scale(float x, float y, float z, float m[16])
{
    mat4_make_identity(m);
    m[0] = x; m[5] = y; m[10] = z;
}

// r[i*4+j] = sum_k a[i*4+k] * b[k*4+j], accumulated k = 0..3 (vectormath.h:459-474).
inline void mat4_mult(const float a[16], const float b[16], float r[16])
{
    float t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            t[i * 4 + j] = a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j] +
                           a[i * 4 + 2] * b[2 * 4 + j] + a[i * 4 + 3] * b[3 * 4 + j];
    std::memcpy(r, t, sizeof(t));
}

inline void mat4_transpose(const float m[16], float r[16])
{
    float t[16];
    std::memcpy(t, m, sizeof(t));
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[i + j * 4] = t[j + i * 4];
}

// Determinant by 2x2 minors, in the reference's pairing (vectormath.h:279-293).
inline float mat4_determinant(const float m[16])
{
    return (m[0] * m[5] - m[1] * m[4]) * (m[10] * m[15] - m[11] * m[14]) +
           (m[2] * m[4] - m[0] * m[6]) * (m[9] * m[15] - m[11] * m[13]) +
           (m[0] * m[7] - m[3] * m[4]) * (m[9] * m[14] - m[10] * m[13]) +
           (m[1] * m[6] - m[2] * m[5]) * (m[8] * m[15] - m[11] * m[12]) +
           (m[3] * m[5] - m[1] * m[7]) * (m[8] * m[14] - m[10] * m[12]) +
           (m[2] * m[7] - m[3] * m[6]) * (m[8] * m[13] - m[9] * m[12]);
}

// Gauss-Jordan inverse working on "lines" L_k = {m[k], m[4+k], m[8+k], m[12+k]}.
// Pivot order 0..3; a pivot below 1e-5 is exchanged with a later line exactly
// the way the reference picks it (vectormath.h:308-457), including that the
// second pivot re-uses the first pivot's choice when it finds no candidate.
// Returns -1 for |det| < 1e-5, else 0.
inline int mat4_invert(const float mat[16], float inv[16])
{
    const float tiny = .00001f;
    float work[16];
    std::memcpy(work, mat, sizeof(work));
    std::memcpy(inv, mat4_identity, sizeof(mat4_identity));
    // evaluated after `inv` is reset, as upstream: an in-place call
    // (inv == mat) therefore never reports a singular matrix
    const float det = mat4_determinant(mat);
    if (std::fabs(det) < tiny)
        return -1;

    int exchange = 0;
    for (int p = 0; p < 4; p++) {
        if (p < 3 && std::fabs(work[p * 5]) < tiny) {
            if (p == 2) {
                exchange = 3;
            } else {
                for (int cand = p + 1; cand < 4; cand++) {
                    if (std::fabs(work[p * 4 + cand]) > tiny) {
                        exchange = cand;
                        break;
                    }
                }
            }
            for (int i = 0; i < 4; i++) {
                std::swap(work[i * 4 + p], work[i * 4 + exchange]);
                std::swap(inv[i * 4 + p], inv[i * 4 + exchange]);
            }
        }
        const float pivot = work[p * 5];
        for (int i = 0; i < 4; i++) {
            work[i * 4 + p] /= pivot;
            inv[i * 4 + p] /= pivot;
        }
        for (int k = 0; k < 4; k++) {
            if (k == p)
                continue;
            const float f = work[p * 4 + k];
            for (int i = 0; i < 4; i++) {
                work[i * 4 + k] -= f * work[i * 4 + p];
                inv[i * 4 + k] -= f * inv[i * 4 + p];
            }
        }
    }
    return 0;
}

// Axis-angle rotation (vectormath.h:529-557); cos/sin evaluated in double and
// rounded once to float, as `(float)cos(a)` does.
inline void mat4_make_rotation(float a, float x, float y, float z, float m[16])
{
    const float c = (float)cos(a);
    const float s = (float)sin(a);
    const float t = 1.0f - c;
    m[0] = t * x * x + c;      m[1] = t * x * y + s * z;  m[2] = t * x * z - s * y;  m[3] = 0;
    m[4] = t * x * y - s * z;  m[5] = t * y * y + c;      m[6] = t * y * z + s * x;  m[7] = 0;
    m[8] = t * x * z + s * y;  m[9] = t * y * z - s * x;  m[10] = t * z * z + c;     m[11] = 0;
    m[12] = 0;                 m[13] = 0;                 m[14] = 0;                 m[15] = 1;
}

// Recovers (angle, axis) from a rotation matrix (vectormath.h:476-527).
inline void mat4_get_rotation(const float m[16], float r[4])
{
    float cosine = (m[0] + m[5] + m[10] - 1.0f) / 2.0f;
    cosine = std::min(1.0f, std::max(-1.0f, cosine));
    r[0] = (float)acos(cosine);
    r[1] = m[6] - m[9];
    r[2] = m[8] - m[2];
    r[3] = m[1] - m[4];
    const float len = sqrt(r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    r[1] /= len; r[2] /= len; r[3] /= len;
}

inline void rotation_mult_rotation(const float first[4], const float second[4], float result[4])
{
    float a[16], b[16], ab[16];
    mat4_make_rotation(first[0], first[1], first[2], first[3], a);
    mat4_make_rotation(second[0], second[1], second[2], second[3], b);
    mat4_mult(b, a, ab);
    mat4_get_rotation(ab, result);
}
