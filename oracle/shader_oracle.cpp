// shader_oracle.cpp -- CPU restatement of shader-ray's per-pixel path.
//
// *** TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
// *** cpu_baseline leg may load this.  The product path never routes through it.
//
// PARITY STATUS: PINNED AGAINST THE REFERENCE ITSELF.  The per-pixel path of the reference exists only as GLSL
// (raytracer.vs, raytracer.es.fs) and the reference ships no tests or vectors for it -- but the shaders run in this
// container: unmodified, as "#version 140" (ray.cpp:401), on Mesa's llvmpipe (CPU) in the 3.2 core context
// ray.cpp:964-967 asks for, behind oracle/glsl_ref/glsl_ref.cpp, which restates ray.cpp's GL calls around them.
// tests/golden/glsl_reference/*.npz are frames rendered that way (tests/golden/make_glsl_reference.py);
// tests/test_reference_shader.py holds this oracle to them: traversal, intersection, shading, shadow rays, caps and
// tone map agree to float rounding (a 256 x 256 frame of the 69k-triangle mesh under a constant environment: largest
// relative difference 4.4e-6), frames with an environment lookup to 1e-4 on >= 97 % of the pixels and 5e-4 on all but
// the rare pixel whose hit / shadow decision falls the other way under the GLSL compiler's own pow / atan / acos; at
// BASELINE's full 1920 x 1080, 99.95 % of configs[1]'s pixels within 1e-4 (profiles/history/r03/reference_shader_agreement.txt).
// What the shader text leaves to the GL implementation is fixed here by rule, and recorded rather than asserted
// against the driver: textureGrad with zero derivatives under 4x anisotropy (level-0 bilinear here: what the driver
// does at anisotropy 1), the which == 1 filter, operations on NaN (IEEE here), pow of a negative base (x^5 here).
// Also pinning it: (1) its inputs -- the flattened arrays and frame parameters -- checked bit for bit against the
// compiled reference host code (oracle/_ref/ref_host, tests/golden); (2) analytic known-answer tests written from
// the shader text (tests/test_oracle_kat.py).
//
// Every function below cites the shader lines it restates (file:line into the
// reference repository; "fs" = raytracer.es.fs, "vs" = raytracer.vs).
//
// Arithmetic contract (shared with the HIP kernel):
//   * IEEE float32 throughout: the host prepends "#version 140" (ray.cpp:401),
//     desktop GLSL, where mediump/highp are all fp32.  One rounding per
//     operation, no FMA contraction (-ffp-contract=off), IEEE divide and sqrt.
//   * dot(a,b)   = a.x*b.x + a.y*b.y + a.z*b.z, left to right
//     cross      = the usual three differences of products
//     normalize  = v / sqrt(dot(v,v))                    (GLSL 1.40 spec 8.4)
//     reflect    = I - (2*dot(N,I))*N                    (GLSL 1.40 spec 8.4)
//     max(x,y)   = x < y ? y : x;  min(x,y) = y < x ? y : x   (spec 8.3)
//     mat4*vec4  = sum over columns, x..w in order
//   * the three transcendental built-ins the path uses are specified as
//     explicit fp32 operation sequences (GLSL only bounds their precision
//     loosely), so that the whole path is bit-reproducible on any IEEE machine:
//       atan(y,x) -> sr_atan2: octant reduction + the odd degree-9 polynomial
//                    derived by oracle/tools/fit_atan.py (max error 2.8 ulp)
//       acos(x)   -> sr_atan2(sqrt((1-x)*(1+x)), x), x clamped to [-1,1]
//                    (GLSL leaves |x| > 1 undefined, fs:130)
//       pow(x,5.) -> ((x*x)*(x*x))*x                       (fs:481)
//     The C library's versions differ by an ulp between glibc and the GPU's
//     ocml, and a sharp environment map (the reference's own `grid`) amplifies
//     that beyond 1e-4 relative; with the explicit forms CPU and GPU agree exactly.
//   * data textures are NEAREST-filtered (ray.cpp:351-352) and index_to_sample
//     (fs:239-245) lands in texel (which mod W, which div W): plain array reads.
//     All indices are float32 values (exact below 2^24).
//   * normals are stored as GL_RGB16F (ray.cpp:474): rounded to binary16
//     (round-to-nearest-even) when params.normals_fp16 is set.
//   * the environment is kept float32 (the reference uploads it with an
//     unsized GL_RGB internal format, ray.cpp:508, which would usually clamp to
//     8 bits; a literal evaluation of the shader sees floats).  Lookup =
//     level-0 bilinear with full float weights, REPEAT wrap in s and t
//     (textureGrad with zero gradients, fs:153; MAG LINEAR ray.cpp:505).
//   * primary rays: the vertex shader (vs:39-60) is evaluated at the pixel
//     centre instead of being interpolated from the quad corners; the two
//     agree up to float rounding because all four corner rays have one length.
//     Row 0 of the output is the BOTTOM row (v = 0, vs:43-44).
//   * `which` (fs:27, ray.cpp:46): 0 = normal.  2 and 3 are the shader's differential debug
//     views (fs:147-149, :642-650), 5 its 5x5 supersampled "reference image" (fs:654-673);
//     all three are restated here.  For 2 the rays carry the differentials of fs:58-63 through
//     ray_transfer / ray_reflect exactly as written, including the vec3 - scalar of fs:92-93.
//     pow(x, 1.5) (fs:623) is evaluated as x * sqrt(x).  For 5 the shader offsets the
//     INTERPOLATED varying world_ray_direction (fs:663), which is the corner rays' common
//     1/length times (ipw(u-.5), ipw(v-.5)aspect, -1): that is what is used here; its first,
//     discarded trace (fs:652 overwritten at :656) is not executed.  Any value other than
//     1, 2, 3, 5 renders like 0, as in the shader.
//   * which == 1 (fs:144-146): textureGrad on a mip-mapped texture (LINEAR_MIPMAP_LINEAR, MAG
//     LINEAR, 4x anisotropy, ray.cpp:503-509).  OpenGL leaves parts of this open; the rule fixed
//     here (and in the kernel) is: mip level k+1 = 2x2 box filter of level k, ((a+b)+(c+d))*0.25,
//     dimensions max(1, n/2), down to 1x1 (glGenerateMipmap's recommended filter);
//     Px = |(du/dx * W, dv/dx * H)|, Py likewise (GL 3.1 section 3.8.9 scale factors);
//     N = min(ceil(Pmax/Pmin), 4) probes, lambda = log2(Pmax / N), probes at
//     P + (i/(N+1) - 1/2) * (major-axis derivative), i = 1..N, averaged
//     (EXT_texture_filter_anisotropic's reference formulas); each probe is LINEAR at level 0 when
//     lambda <= 0, else a blend of the two nearest levels by frac(lambda), clamped to the last
//     level; log2 is the explicit fp32 sequence sr_log2 below; non-finite derivatives (rays
//     straight up or down, fs:135-139 divide by zero there) select the 1x1 level.
//   * spp > 1 (not in the reference except which==5, fs:654-673): sample s of
//     n uses sub-pixel offset ((s+.5)/n, bitreverse32(s)*2^-32 + .5/n);
//     linear radiance is summed in sample order, divided by n, then tone
//     mapped once -- the same order of operations as fs:669-676.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "shader_ray_hip.h"

namespace {

// ------------------------------------------------------------------ GLSL-ish
struct vec3 {
    float x, y, z;
};
inline vec3 V(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 operator+(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return V(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline vec3 normalize(vec3 a) { return a / sqrtf(dot(a, a)); }
inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline float gl_max(float x, float y) { return x < y ? y : x; }
inline float gl_min(float x, float y) { return y < x ? y : x; }

// atan(y, x), see the header: one rounding per operation, nothing fused
inline float sr_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax < ay ? ay : ax;
    const float mn = ax < ay ? ax : ay;
    if (mx == 0.0f)
        return 0.0f;
    const float a = mn / mx;
    const bool upper = a > 0.414213562f;                    // tan(pi/8)
    const float z = upper ? (a - 1.0f) / (a + 1.0f) : a;
    const float s = z * z;
    float p = 0.0803788006f * s;
    p = (p + -0.138722613f) * s;
    p = (p + 0.199771404f) * s;
    p = p + -0.33332932f;
    float r = z + (z * s) * p;
    if (upper)
        r = 0.785398163f + r;
    if (ay > ax)
        r = 1.57079633f - r;
    if (x < 0.0f)
        r = 3.14159265f - r;
    if (y < 0.0f)
        r = -r;
    return r;
}
inline float sr_acos(float x) { return sr_atan2(sqrtf((1.0f - x) * (1.0f + x)), x); }
// pow(x, 5.0) of fs:481.  GLSL leaves pow undefined for x < 0, and the implementations that run the reference evaluate it
// as exp2(5 log2 x): NaN for a negative base (measured on the driver behind the fixtures: the three pixels of a 1080p
// frame where an fp16-rounded normal slightly longer than 1 makes the base -1e-8 come out black -- the NaN reaches
// tonemap_and_gamma's max(0, c - .004), which by GLSL's definition returns 0).  So does this; x^5 for x >= 0 (-0 included).
inline float sr_pow5(float x)
{
    if (x < 0.0f)
        return std::numeric_limits<float>::quiet_NaN();
    const float x2 = x * x;
    return (x2 * x2) * x;
}

// log2(x) for finite x > 0: exponent + 2/ln2 * atanh((m-1)/(m+1)) by its odd series to z^9,
// m in [1, 2); every step one fp32 operation (absolute error < 2e-6, ample for choosing and
// blending mip levels)
inline float sr_log2(float x)
{
    int e;
    const float m = 2.0f * frexpf(x, &e);   // x = m * 2^(e-1), m in [1, 2)
    const float z = (m - 1.0f) / (m + 1.0f);
    const float z2 = z * z;
    float p = 0.111111111f * z2;
    p = (p + 0.142857143f) * z2;
    p = (p + 0.2f) * z2;
    p = (p + 0.333333333f) * z2;
    p = (p + 1.0f) * z;
    return (float)(e - 1) + 2.88539008f * p;
}

// mat4 * vec4(v, w), column-major storage
inline vec3 transform(const float m[16], vec3 v, float w)
{
    return V(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * w,
             m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * w,
             m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * w);
}

// binary32 -> binary16 -> binary32, round to nearest even, IEEE subnormals
float round_through_half(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t mag = u & 0x7fffffffu;
    float out;
    if (mag >= 0x7f800000u) {              // inf / nan stay
        return f;
    } else if (mag >= 0x477ff000u) {       // rounds to >= 65520 -> inf
        mag = 0x7f800000u;
    } else if (mag < 0x38800000u) {        // below 2^-14: half subnormal, quantum 2^-24
        float a;
        memcpy(&a, &mag, 4);
        const float q = a * 16777216.0f;   // exact scaling by 2^24
        const float r = nearbyintf(q);     // default rounding mode = nearest even
        a = r / 16777216.0f;
        memcpy(&mag, &a, 4);
    } else {                               // normal half: keep 10 fraction bits
        const uint32_t lsb = (mag >> 13) & 1u;
        mag += 0xfffu + lsb;
        mag &= ~0x1fffu;
    }
    mag |= sign;
    memcpy(&out, &mag, 4);
    return out;
}

// --------------------------------------------------------------------- scene
struct Scene {
    uint32_t width;            // data_texture_width
    int group_rows;
    float tree_root;
    const float *positions;
    std::vector<float> normals;   // possibly fp16-rounded copy
    const float *boxmin, *boxmax, *hitmiss, *objects;
    const float *env;
    int env_w, env_h;
    // mip pyramid of the environment (level 0 = env itself), built on demand for which == 1
    std::vector<std::vector<float>> mip;
    std::vector<int> mip_w, mip_h;
};

// How the environment is stored (shray_oracle_set_env_storage): 0 = the floats as given; 1 = 8-bit normalized fixed
// point, what the reference's unsized GL_RGB upload (ray.cpp:508) becomes on most drivers -- every texel, and every
// texel of every mip level, passes through c = floor(255 clamp(f, 0, 1) + 0.5), f = c / 255 (GL 3.1 section 2.1.5)
int g_env_storage = 0;
float through_unorm8(float f)
{
    const float clamped = !(f > 0.0f) ? 0.0f : (f > 1.0f ? 1.0f : f);
    return std::floor(clamped * 255.0f + 0.5f) / 255.0f;
}

void build_mips(Scene &sc)
{
    sc.mip.clear();
    sc.mip_w.assign(1, sc.env_w);
    sc.mip_h.assign(1, sc.env_h);
    sc.mip.emplace_back(sc.env, sc.env + 3 * (size_t)sc.env_w * sc.env_h);
    while (sc.mip_w.back() > 1 || sc.mip_h.back() > 1) {
        const int sw = sc.mip_w.back(), sh = sc.mip_h.back();
        const int w = std::max(1, sw / 2), h = std::max(1, sh / 2);
        const std::vector<float> &src = sc.mip.back();
        std::vector<float> dst(3 * (size_t)w * h);
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const int i0 = std::min(2 * i, sw - 1), i1 = std::min(2 * i + 1, sw - 1);
                const int j0 = std::min(2 * j, sh - 1), j1 = std::min(2 * j + 1, sh - 1);
                for (int c = 0; c < 3; c++) {
                    const float a = src[3 * ((size_t)j0 * sw + i0) + c], b = src[3 * ((size_t)j0 * sw + i1) + c];
                    const float cc = src[3 * ((size_t)j1 * sw + i0) + c], d = src[3 * ((size_t)j1 * sw + i1) + c];
                    const float mean = ((a + b) + (cc + d)) * 0.25f;
                    dst[3 * ((size_t)j * w + i) + c] = g_env_storage == 1 ? through_unorm8(mean) : mean;
                }
            }
        sc.mip.push_back(std::move(dst));
        sc.mip_w.push_back(w);
        sc.mip_h.push_back(h);
    }
}

struct Counters {
    uint64_t node_visits = 0, leaf_visits = 0, triangle_tests = 0, shaded_hits = 0, env_lookups = 0, traversals = 0,
             bad_hits = 0;
};

struct Ctx {
    const Scene *scene;
    const shray_frame_params *p;
    Counters c;
    // diagnostics (shray_oracle_set_path_map): what the last trace() did -- see there
    uint32_t path = 0;
    int32_t first_triangle = -1;
    float last_which = -1.0f;
    bool last_lit = false;
    float edge_margin = 1.0f;     // the smallest barycentric coordinate of any closest hit of the path (1: no hit)
    float last_edge_margin = 1.0f;
    float env_dy = 0.0f;          // D.y of the last environment lookup, before the clamp of sample_environment
    // (shray_oracle_set_decision_map) how close any triangle test of the path -- closest-hit AND shadow traversals -- came to
    // deciding the other way: the smallest distance of a tested point from the triangle's boundary, in barycentric units
    // (tests whose distance d lies in the leaf's range and in front of the hit so far), and from the ends of that range
    // relative to d; and the distance of Schlick's pow(x, 5.0) base from zero (negative: NaN, a black pixel)
    bool track_decisions = false;
    float decision_margin = 1.0f;
    // (the quantised-record probe is on: read once per pixel, not from the global in every node visit -- ADVICE round 5: the
    // diagnostics must not weigh on the loops bench.py's cpu_baseline times)
    bool probe_quantised = false;
};

// optional per-pixel cost map: node visits summed over the pixel's samples (diagnostics)
uint32_t *g_visit_map = nullptr;

// optional probe (tests/quantised_record_probe.py; EXPERIMENTS R5.4): would a node record with its six planes quantised to steps of
// `g_quant_step` (scene extent / 65536 for 16-bit planes) decide this visit?  A visit is UNDECIDED when the box grown by one step and
// the box shrunk by one step answer fs:400's question differently.  {visits, undecided visits}, summed over the render's threads.
float g_quant_step = 0.0f;
std::atomic<unsigned long long> g_quant_visits{0}, g_quant_undecided{0};

// optional per-pixel path planes (tests/pixel_classifier.py: where the frame is discontinuous): for spp == 1 plain frames,
// path = bit 2i: bounce i hit a triangle, bit 2i + 1: that hit was lit (its shadow ray escaped; only when diffuse > 0),
// bits 24-27: bounces that hit, bit 30: the iteration-cap marker; first_triangle = the primary ray's triangle or -1
// edge_margin = the smallest barycentric coordinate of any of the path's hits: a ray that passes within a few 1e-7 of an edge
// two triangles share can, in another arithmetic, miss both (the test of fs:333-340 is not watertight) or hit the other one
uint32_t *g_path_map = nullptr;
int32_t *g_first_triangle_map = nullptr;
float *g_edge_margin_map = nullptr;
float *g_env_dy_map = nullptr;   // D.y of the pixel's environment lookup as the shader hands it to acos (|D.y| > 1: undefined there)
// decision_margin (Ctx): a shadow ray that grazes an occluder's silhouette, or slips between two triangles, by a few 1e-6 of a
// barycentric coordinate is lit in one fp32 evaluation and shadowed in another -- round 5's matte frames of the 1M-triangle scene
float *g_decision_map = nullptr;

// optional per-ray event trace (diagnostics, single-threaded renders only; oracle/tools/wave_sim.cpp reads it):
// per sample, per traversal: 0xF0 then one byte per node visit = the number of triangle tests that visit ran
struct VisitTrace {
    uint8_t *bytes;
    uint64_t capacity, length;
    uint64_t *sample_offsets;   // [samples + 1]
    uint64_t samples;
};
VisitTrace *g_trace = nullptr;
inline void trace_byte(uint8_t b)
{
    if (g_trace && g_trace->length < g_trace->capacity)
        g_trace->bytes[g_trace->length] = b;
    if (g_trace)
        g_trace->length++;
}

struct ray {   // fs:58-63
    vec3 P, D;
    vec3 dPdx, dDdx, dPdy, dDdy;
};
inline ray make_ray(vec3 P, vec3 D) { return ray{P, D, V(0, 0, 0), V(0, 0, 0), V(0, 0, 0), V(0, 0, 0)}; }

struct surface_hit {   // fs:108-113
    float t, which;
    vec3 uvw;
};

const float infinitely_far = 10000000.0f;   // fs:115
const float pi = 3.14159265259f;            // fs:116 (rounds to the float nearest pi)
const float tau = 2 * pi;                   // fs:117
const float terminator = 16777215.0f;       // fs:384

struct range {   // fs:168-171
    float t0, t1;
};
inline range make_range(float t0, float t1) { return range{t0, t1}; }             // fs:173-179
inline range range_intersect(range a, range b)                                     // fs:186-191
{
    return make_range(gl_max(a.t0, b.t0), gl_min(a.t1, b.t1));
}
inline bool range_is_empty(range r) { return r.t0 >= r.t1; }                       // fs:193-196

// fs:200-217 -- slab test with a true division per plane
// (always inlined into group_intersect's loop: with the diagnostics' cold callers below as further call sites the compiler stopped
// inlining it there, and the oracle -- bench.py's cpu_baseline -- lost 40 % of its rate)
__attribute__((always_inline)) inline range range_intersect_box(vec3 boxmin, vec3 boxmax, const ray &theray, range prevr)
{
    float t0, t1;
    t0 = (boxmin.x - theray.P.x) / theray.D.x;
    t1 = (boxmax.x - theray.P.x) / theray.D.x;
    range r0 = range_intersect(prevr, (theray.D.x >= 0.0f) ? make_range(t0, t1) : make_range(t1, t0));
    t0 = (boxmin.y - theray.P.y) / theray.D.y;
    t1 = (boxmax.y - theray.P.y) / theray.D.y;
    range r1 = range_intersect(r0, (theray.D.y >= 0.0f) ? make_range(t0, t1) : make_range(t1, t0));
    t0 = (boxmin.z - theray.P.z) / theray.D.z;
    t1 = (boxmax.z - theray.P.z) / theray.D.z;
    range r2 = range_intersect(r1, (theray.D.z >= 0.0f) ? make_range(t0, t1) : make_range(t1, t0));
    return r2;
}

inline vec3 fetch3(const float *a, float index)
{
    const size_t i = (size_t)index;
    return V(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
}

struct group {   // fs:219-227
    bool is_branch;
    float start, count;
    vec3 boxmin, boxmax;
    float hit_next, miss_next;
};

// fs:247-270
group get_group(Ctx &cx, float which, float hitmiss_offset)
{
    const Scene &s = *cx.scene;
    group g;
    g.boxmin = fetch3(s.boxmin, which);
    g.boxmax = fetch3(s.boxmax, which);
    const size_t link = (size_t)(which + hitmiss_offset);
    g.hit_next = s.hitmiss[2 * link];
    g.miss_next = s.hitmiss[2 * link + 1];
    g.is_branch = (g.hit_next != g.miss_next);
    g.start = g.count = 0;
    if (!g.is_branch) {
        const size_t i = (size_t)which;
        g.start = s.objects[2 * i];
        g.count = s.objects[2 * i + 1];
        cx.c.leaf_visits++;
    }
    return g;
}

// Diagnostics (Ctx::decision_margin; tests/pixel_classifier.py), kept OUT of the timed loops: cold, never inlined -- the oracle is
// also bench.py's cpu_baseline, and these bodies inside triangle_intersect / group_intersect cost it a quarter of its rate.
__attribute__((noinline, cold)) void note_determinant(Ctx &cx, vec3 rayP, vec3 rayD, range r, float hit_t, vec3 v0, vec3 e0, vec3 e1,
                                                      vec3 M, float det)
{
    // a triangle the ray WOULD hit, whose determinant lies at the rejection threshold of fs:312 (the small triangles of a
    // 1M-triangle mesh seen at grazing incidence: |det| ~ 1e-7) -- taken by one fp32 evaluation, skipped by another
    const float epsilon = 0.0000001f;
    if (det == 0.0f)
        return;
    const float closeness = fabsf(fabsf(det) - epsilon) / epsilon;
    if (closeness < 1e-2f) {
        const float id = 1.0f / det;
        const vec3 T0 = rayP - v0, Q0 = cross(T0, e0);
        const float d0 = -dot(e1, Q0) * id, u0 = dot(T0, M) * id, w0 = dot(rayD, Q0) * id;
        if (!(d0 > hit_t) && !(d0 < r.t0 || d0 > r.t1) && u0 >= 0.0f && w0 >= 0.0f && u0 + w0 <= 1.0f)
            cx.decision_margin = gl_min(cx.decision_margin, closeness * 1e-3f);
    }
}

__attribute__((noinline, cold)) void note_triangle_test(Ctx &cx, vec3 rayD, range r, float hit_t, vec3 T, vec3 Q, vec3 M, float inv_det, float d)
{
    // the distance of this test from each of its decisions
    if (d > hit_t)
        return;
    const float uu = dot(T, M) * inv_det, vv = dot(rayD, Q) * inv_det;
    const float inside = gl_min(gl_min(uu, vv), 1.0f - uu - vv);            // > 0 inside the triangle
    const float scale = gl_max(fabsf(d), 1e-30f);
    const float ends = gl_min(fabsf(d - r.t0), fabsf(d - r.t1)) / scale;    // relative distance from the range's ends
    const bool in_range = !(d < r.t0 || d > r.t1);
    if (in_range)
        cx.decision_margin = gl_min(cx.decision_margin, fabsf(inside));
    // (boxes are inflated by 1e-5 absolute, vectormath.h:189-195: a hit on a box face sits a few 1e-6 of d inside the range, safely
    // -- d's own rounding is 1e-7 of it; the ends count twenty-fold, so that the classifier's 2e-5 means 1e-6 of d here)
    if (inside >= 0.0f || fabsf(inside) < 1e-3f)
        cx.decision_margin = gl_min(cx.decision_margin, 20.0f * ends);
}

// fs:297-346
void triangle_intersect(Ctx &cx, float which, const ray &theray, range r, surface_hit &hit)
{
    cx.c.triangle_tests++;
    const Scene &s = *cx.scene;
    const vec3 v0 = fetch3(s.positions, which * 3.0f + 0.0f);
    const vec3 v1 = fetch3(s.positions, which * 3.0f + 1.0f);
    const vec3 v2 = fetch3(s.positions, which * 3.0f + 2.0f);

    const vec3 e0 = v1 - v0;
    const vec3 e1 = v0 - v2;
    const vec3 M = cross(e1, theray.D);
    const float det = dot(e0, M);
    const float epsilon = 0.0000001f;
    if (__builtin_expect(cx.track_decisions, 0))
        note_determinant(cx, theray.P, theray.D, r, hit.t, v0, e0, e1, M, det);
    if (det > -epsilon && det < epsilon)
        return;
    const float inv_det = 1.0f / det;

    const vec3 T = theray.P - v0;
    const vec3 Q = cross(T, e0);
    const float d = -dot(e1, Q) * inv_det;
    if (__builtin_expect(cx.track_decisions, 0))
        note_triangle_test(cx, theray.D, r, hit.t, T, Q, M, inv_det, d);
    if (d > hit.t)
        return;
    if (d < r.t0 || d > r.t1)
        return;
    const float u = dot(T, M) * inv_det;
    if (u < 0.0f || u > 1.0f)
        return;
    const float v = dot(theray.D, Q) * inv_det;
    if (v < 0.0f || u + v > 1.0f)
        return;
    hit.which = which;
    hit.t = d;
    hit.uvw.x = 1.0f - u - v;
    hit.uvw.y = u;
    hit.uvw.z = v;
}

__attribute__((noinline, cold)) void note_quantised_visit(vec3 boxmin, vec3 boxmax, vec3 rayP, vec3 rayD, range prevr, float hit_t)
{
    ray theray{};
    theray.P = rayP;
    theray.D = rayD;
    const vec3 q = V(g_quant_step, g_quant_step, g_quant_step);
    const range grown = range_intersect_box(boxmin - q, boxmax + q, theray, prevr);
    const range shrunk = range_intersect_box(boxmin + q, boxmax - q, theray, prevr);
    const bool a = (!range_is_empty(grown)) && (grown.t0 < hit_t), b = (!range_is_empty(shrunk)) && (shrunk.t0 < hit_t);
    g_quant_visits.fetch_add(1, std::memory_order_relaxed);
    if (a != b)
        g_quant_undecided.fetch_add(1, std::memory_order_relaxed);
}

// fs:386-443, CONSTANT_LENGTH_LOOPS branch
void group_intersect(Ctx &cx, float root, const ray &theray, range prevr, surface_hit &hit)
{
    cx.c.traversals++;
    const Scene &s = *cx.scene;
    const int max_bvh_iterations = cx.p->max_bvh_iterations;   // fs:381
    const float max_leaf_tests = (float)cx.p->max_leaf_tests;  // fs:382
    float g = root;
    const float xd = (theray.D.x > 0.0f) ? 1.0f : 0.0f;
    const float yd = (theray.D.y > 0.0f) ? 2.0f : 0.0f;
    const float zd = (theray.D.z > 0.0f) ? 4.0f : 0.0f;
    const float offset = (xd + yd + zd) * float(s.group_rows) * float(s.width);

    trace_byte(0xF0);
    for (int i = 0; i < max_bvh_iterations; i++) {
        cx.c.node_visits++;
        group gg = get_group(cx, g, offset);
        range r = range_intersect_box(gg.boxmin, gg.boxmax, theray, prevr);
        const uint64_t tests_before = cx.c.triangle_tests;
        if (__builtin_expect(cx.probe_quantised, 0))
            note_quantised_visit(gg.boxmin, gg.boxmax, theray.P, theray.D, prevr, hit.t);
        if ((!range_is_empty(r)) && (r.t0 < hit.t)) {
            if (!gg.is_branch) {
                for (float j = 0.0f; j < max_leaf_tests; j++) {
                    if (j >= gg.count)
                        break;
                    triangle_intersect(cx, gg.start + j, theray, r, hit);
                }
            }
            g = gg.hit_next;
        } else {
            g = gg.miss_next;
        }
        trace_byte((uint8_t)(cx.c.triangle_tests - tests_before));
        if (g >= terminator)
            return;
        if (i == max_bvh_iterations - 1) {   // fs:436-438, set_bad_hit(hit, 1, 0, 0) fs:162-166
            hit.t = -1.0f;
            hit.uvw = V(1.0f, 0.0f, 0.0f);
        }
    }
}

inline surface_hit surface_hit_init() { return surface_hit{infinitely_far, -1.0f, V(1, 0, 0)}; }   // fs:157-160

// fs:98-106 (P and D only)
inline ray ray_transform(const ray &r, const float matrix[16], const float normal_matrix[16])
{
    return make_ray(transform(matrix, r.P, 1.0f), transform(normal_matrix, r.D, 0.0f));
}

// fs:65-81
inline ray ray_transfer(const ray &in, float t, vec3 normal)
{
    ray out;
    out.P = in.P + in.D * t;
    out.D = in.D;
    const float dtdx = -dot(in.dPdx + t * in.dDdx, normal) / dot(in.D, normal);
    out.dPdx = in.dPdx + t * in.dDdx + dtdx * in.D;
    out.dDdx = in.dDdx;
    const float dtdy = -dot(in.dPdy + t * in.dDdy, normal) / dot(in.D, normal);
    out.dPdy = in.dPdy + t * in.dDdy + dtdy * in.D;
    out.dDdy = in.dDdy;
    return out;
}

// fs:83-96; the direction differentials lose a SCALAR per component, as written ("do this right")
inline ray ray_reflect(const ray &in, vec3 normal)
{
    ray out;
    out.D = reflect(in.D, normal);
    out.P = in.P + normal * .0001f;
    out.dPdx = in.dPdx;
    out.dPdy = in.dPdy;
    const float sx = 2 * dot(in.dDdx, normal), sy = 2 * dot(in.dDdy, normal);
    out.dDdx = V(in.dDdx.x - sx, in.dDdx.y - sx, in.dDdx.z - sx);
    out.dDdy = V(in.dDdy.x - sy, in.dDdy.y - sy, in.dDdy.z - sy);
    return out;
}

// fs:288-295
vec3 triangle_interpolate_normal(const Scene &s, float which, vec3 uvw)
{
    const vec3 n0 = fetch3(s.normals.data(), which * 3.0f + 0.0f);
    const vec3 n1 = fetch3(s.normals.data(), which * 3.0f + 1.0f);
    const vec3 n2 = fetch3(s.normals.data(), which * 3.0f + 2.0f);
    return n0 * uvw.x + n1 * uvw.y + n2 * uvw.z;
}

// fs:447-472
vec3 approximate_diffuse(Ctx &cx, vec3 point, vec3 normal)
{
    const shray_frame_params &p = *cx.p;
    const vec3 light_dir = V(p.light_dir[0], p.light_dir[1], p.light_dir[2]);
    const float lcos = gl_max(0.0f, dot(normal, light_dir));
    const vec3 light_diffuse = V(1.0f, 1.0f, 1.0f) * lcos;   // light_color fs:25
    vec3 diffuse = V(0.0f, 0.0f, 0.0f);                      // ambient
    if (p.cast_shadows) {
        surface_hit shadow_hit = surface_hit_init();
        const ray world_shadowray = make_ray(point, light_dir);
        const ray object_shadowray = ray_transform(world_shadowray, p.object_matrix, p.object_normal_matrix);
        group_intersect(cx, cx.scene->tree_root, object_shadowray, make_range(0.0f, 100000000.0f), shadow_hit);
        cx.last_lit = shadow_hit.t >= infinitely_far;
        if (shadow_hit.t >= infinitely_far)
            diffuse = diffuse + light_diffuse;
    } else {
        cx.last_lit = true;
        diffuse = diffuse + light_diffuse;
    }
    return diffuse;
}

// fs:479-482
inline vec3 f_schlick_vr(vec3 cspec, vec3 v, vec3 r)
{
    const float w = sr_pow5(dot(v, r) * .5f + .5f);
    return cspec + (V(1.0f, 1.0f, 1.0f) - cspec) * w;
}

// fs:484-522
int intersect_and_shade(Ctx &cx, const ray &worldray, vec3 &object_diffuse, vec3 &object_specular, vec3 &normal,
                        ray &reflected)
{
    const shray_frame_params &p = *cx.p;
    surface_hit shading = surface_hit_init();
    const ray objectray = ray_transform(worldray, p.object_matrix, p.object_normal_matrix);
    group_intersect(cx, cx.scene->tree_root, objectray, make_range(0.0f, 100000000.0f), shading);

    if (shading.t >= infinitely_far)
        return 0;
    if (shading.t == -1.0f) {
        object_diffuse = shading.uvw;
        object_specular = V(0, 0, 0);
        return 2;
    }

    // shade(), fs:362-377; hit.which >= 0 always holds here
    cx.c.shaded_hits++;
    cx.last_which = shading.which;
    cx.last_edge_margin = gl_min(gl_min(shading.uvw.x, shading.uvw.y), shading.uvw.z);
    const vec3 object_normal = triangle_interpolate_normal(*cx.scene, shading.which, shading.uvw);
    const vec3 object_color = V(1.0f, 1.0f, 1.0f);

    vec3 world_normal = transform(p.object_normal_inverse, object_normal, 0.0f);
    if (dot(world_normal, worldray.D) > 0.0f)
        world_normal = world_normal * -1.0f;

    // ray_transfer fs:65-81, ray_reflect fs:83-96.  P and D do not depend on the differentials;
    // the differentials are only carried when a view needs them (which == 2)
    if (p.which == 1 || p.which == 2) {
        reflected = ray_reflect(ray_transfer(worldray, shading.t, world_normal), world_normal);
    } else {
        reflected = make_ray(worldray.P + worldray.D * shading.t + world_normal * .0001f, reflect(worldray.D, world_normal));
    }

    object_specular = f_schlick_vr(V(p.specular_color[0], p.specular_color[1], p.specular_color[2]), worldray.D, reflected.D);
    if (cx.track_decisions) {
        // diagnostics only: the base of fs:481's pow(x, 5.0) is 0 +- 1e-8 at normal incidence, and a negative base is NaN (a black
        // pixel): which side of zero it falls on is a last-bit matter
        const float base = dot(worldray.D, reflected.D) * .5f + .5f;
        cx.decision_margin = gl_min(cx.decision_margin, fabsf(base));
    }
    object_diffuse = V(p.diffuse_color[0], p.diffuse_color[1], p.diffuse_color[2]) * object_color;
    normal = world_normal;
    return 1;
}

// LINEAR lookup with REPEAT wrap in one level of the environment at texture coordinates (s, t)
vec3 bilinear_level(const float *texels, int w, int h, float s, float t)
{
    const float fw = (float)w, fh = (float)h;
    const float u = s * fw - 0.5f;
    const float v = t * fh - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    auto wrap = [](float f, int n) {
        int i = (int)f % n;
        return i < 0 ? i + n : i;
    };
    const int i0 = wrap(fu, w), i1 = wrap(fu + 1.0f, w);
    const int j0 = wrap(fv, h), j1 = wrap(fv + 1.0f, h);
    auto texel = [&](int i, int j) {
        const float *px = texels + 3 * ((size_t)j * w + i);
        return V(px[0], px[1], px[2]);
    };
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return texel(i0, j0) * w00 + texel(i1, j0) * w10 + texel(i0, j1) * w01 + texel(i1, j1) * w11;
}

// one probe of the filtered lookup at level-of-detail lambda (see the header)
vec3 trilinear_probe(const Scene &sc, float s, float t, float lambda)
{
    const int last = (int)sc.mip.size() - 1;
    if (!(lambda > 0.0f))
        return bilinear_level(sc.mip[0].data(), sc.mip_w[0], sc.mip_h[0], s, t);
    if (lambda >= (float)last)
        return bilinear_level(sc.mip[last].data(), sc.mip_w[last], sc.mip_h[last], s, t);
    const float fl = floorf(lambda);
    const int d1 = (int)fl;
    const float f = lambda - fl;
    const vec3 lo = bilinear_level(sc.mip[d1].data(), sc.mip_w[d1], sc.mip_h[d1], s, t);
    const vec3 hi = bilinear_level(sc.mip[d1 + 1].data(), sc.mip_w[d1 + 1], sc.mip_h[d1 + 1], s, t);
    return lo * (1.0f - f) + hi * f;
}

// textureGrad(sampler, (s, t), (dudx, dvdx), (dudy, dvdy)) as fixed in the header
vec3 texture_grad(const Scene &sc, float s, float t, float dudx, float dvdx, float dudy, float dvdy)
{
    const float fw = (float)sc.env_w, fh = (float)sc.env_h;
    const float ax = dudx * fw, ay = dvdx * fh, bx = dudy * fw, by = dvdy * fh;
    const float px = sqrtf(ax * ax + ay * ay), py = sqrtf(bx * bx + by * by);
    const float pmax = gl_max(px, py), pmin = gl_min(px, py);
    const int last = (int)sc.mip.size() - 1;
    if (!(pmax <= 3.0e38f))   // infinite or NaN footprint: the coarsest level
        return bilinear_level(sc.mip[last].data(), sc.mip_w[last], sc.mip_h[last], s, t);
    if (!(pmax > 0.0f))       // zero footprint: plain magnification
        return bilinear_level(sc.mip[0].data(), sc.mip_w[0], sc.mip_h[0], s, t);
    float n = 4.0f;           // GL_TEXTURE_MAX_ANISOTROPY_EXT, ray.cpp:506
    if (pmin > 0.0f)
        n = gl_min(ceilf(pmax / pmin), 4.0f);
    const float lambda = sr_log2(pmax / n);
    const bool along_x = px >= py;
    const float mu = along_x ? dudx : dudy, mv = along_x ? dvdx : dvdy;
    vec3 sum = V(0, 0, 0);
    const int probes = (int)n;
    for (int i = 1; i <= probes; i++) {
        const float o = (float)i / (n + 1.0f) - 0.5f;
        sum = sum + trilinear_probe(sc, s + o * mu, t + o * mv, lambda);
    }
    return sum / n;
}

// fs:127-155
vec3 sample_environment(Ctx &cx, const ray &r)
{
    cx.c.env_lookups++;
    const Scene &sc = *cx.scene;
    cx.env_dy = r.D.y;
    // acos(D.y), fs:130: GLSL leaves acos undefined outside [-1, 1], and D is not a unit vector there (reflect() about a
    // normal that is not renormalized, fs:288-295, :91): |D.y| exceeds 1 by a few 1e-6 in about one pixel of a 1080p frame.
    // The implementations that run the reference return NaN (measured: tests/golden/glsl_reference/driver_functions.json),
    // the lookup's colour is NaN and the pixel ends black through tonemap_and_gamma's max(0, c - .004).  So does this.
    const bool outside_acos = !(fabsf(r.D.y) <= 1.0f);
    const vec3 undefined_colour = V(std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN(),
                                    std::numeric_limits<float>::quiet_NaN());
    const float dy = gl_min(gl_max(r.D.y, -1.0f), 1.0f);
    const float s = 1.0f + sr_atan2(-r.D.z, r.D.x) / tau;
    const float t = 1.0f - sr_acos(dy) / pi;
    if (cx.p->which == 1 || cx.p->which == 2) {   // fs:135-142: derivatives of the lookup coordinates
        const float two_pi_rxz = 2.0f * pi * (r.D.x * r.D.x + r.D.z * r.D.z);
        const float dudx = (r.D.x * r.dDdx.z - r.D.z * r.dDdx.x) / two_pi_rxz;
        const float dudy = (r.D.x * r.dDdy.z - r.D.z * r.dDdy.x) / two_pi_rxz;
        const float pi_ryy = pi * sqrtf(1.0f - r.D.y * r.D.y);
        const float dvdx = r.dDdx.y / pi_ryy;
        const float dvdy = r.dDdy.y / pi_ryy;
        if (cx.p->which == 2)   // fs:147-149: draw the dY differential
            return V(fabsf(dudy) * 1.0f * 100, fabsf(dvdy) * 1.0f * 100, 0.0f);
        if (outside_acos)
            return undefined_colour;
        return texture_grad(sc, s, t, dudx, dvdx, dudy, dvdy);   // fs:144-146
    }
    if (outside_acos)
        return undefined_colour;
    return bilinear_level(sc.env, sc.env_w, sc.env_h, s, t);   // fs:150-154: zero gradients = level 0, LINEAR
}

// fs:552-582
vec3 trace(Ctx &cx, ray worldray)
{
    vec3 accumulated = V(0, 0, 0);
    vec3 modulation = V(1, 1, 1);
    cx.path = 0;
    cx.first_triangle = -1;
    cx.edge_margin = 1.0f;
    cx.decision_margin = 1.0f;
    cx.track_decisions = g_decision_map != nullptr;
    cx.probe_quantised = g_quant_step > 0.0f;
    for (int i = 0; i < cx.p->bounce_count; i++) {
        ray reflected{};
        vec3 object_diffuse{}, object_specular{}, normal{};
        const int hit_something = intersect_and_shade(cx, worldray, object_diffuse, object_specular, normal, reflected);
        if (hit_something == 0)
            break;
        if (hit_something == 2) {
            cx.c.bad_hits++;
            cx.path |= 1u << 30;
            return object_diffuse;
        }
        if (i < 12)
            cx.path |= 1u << (2 * i);
        cx.path += 1u << 24;
        if (i == 0)
            cx.first_triangle = (int32_t)cx.last_which;
        cx.edge_margin = gl_min(cx.edge_margin, cx.last_edge_margin);
        if (object_diffuse.x > 0.0f && object_diffuse.y > 0.0f && object_diffuse.z > 0.0f) {
            const vec3 diffuse_irradiance = approximate_diffuse(cx, reflected.P, normal);
            if (cx.last_lit && i < 12)
                cx.path |= 2u << (2 * i);
            accumulated = accumulated + modulation * object_diffuse * diffuse_irradiance;
        }
        modulation = modulation * object_specular;
        worldray = reflected;
    }
    const vec3 background_color = sample_environment(cx, worldray);
    return accumulated + modulation * background_color;
}

// fs:527-531
inline float filmic(float c)
{
    const float x = gl_max(0.0f, c - 0.004f);
    return (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
}

inline uint32_t bitreverse32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}

// vs:39-60 evaluated at image-plane position (u, v), then fs:617-619
// fs:621-625: direction differentials of a ray through the image plane
inline void set_differentials(const shray_frame_params &p, ray &r)
{
    const vec3 right = V(p.right[0], p.right[1], p.right[2]), up = V(p.up[0], p.up[1], p.up[2]);
    const vec3 d = r.D;
    const float dd = dot(d, d);
    const float dd15 = dd * sqrtf(dd);   // pow(dot(d, d), 1.5)
    r.dPdx = V(0, 0, 0);
    r.dDdx = (dd * right - dot(d, right) * d) / dd15;
    r.dPdy = V(0, 0, 0);
    r.dDdy = (dd * up - dot(d, up) * d) / dd15;
}

ray primary_ray(const shray_frame_params &p, float u, float v)
{
    const vec3 eye_d = normalize(V(p.image_plane_width * (u - 0.5f), p.image_plane_width * (v - 0.5f) * p.aspect, -1.0f));
    ray world = ray_transform(make_ray(V(0, 0, 0), eye_d), p.camera_matrix, p.camera_normal_matrix);
    world.D = normalize(world.D);
    if (p.which == 1 || p.which == 2 || p.which == 3)
        set_differentials(p, world);
    return world;
}

// The varyings of vs:58-59 as the rasteriser interpolates them at (u, v): every corner ray is
// divided by the same length, so the interpolated direction is the unnormalised image-plane
// vector over that length (used by the which == 5 view, fs:663).
inline void interpolated_varyings(const shray_frame_params &p, float u, float v, vec3 &origin, vec3 &direction)
{
    const float hx = p.image_plane_width * (1.0f - 0.5f), hy = p.image_plane_width * (1.0f - 0.5f) * p.aspect;
    const float corner_length = sqrtf(dot(V(hx, hy, -1.0f), V(hx, hy, -1.0f)));
    const vec3 eye = V(p.image_plane_width * (u - 0.5f), p.image_plane_width * (v - 0.5f) * p.aspect, -1.0f) / corner_length;
    origin = transform(p.camera_matrix, V(0, 0, 0), 1.0f);
    direction = transform(p.camera_normal_matrix, eye, 0.0f);
}

// fs:121-125
inline void environment_map_coords(vec3 d, float &s, float &t)
{
    s = 1.0f + sr_atan2(-d.z, d.x) / tau;
    t = 1.0f - sr_acos(gl_min(gl_max(d.y, -1.0f), 1.0f)) / pi;
}

void shade_pixel(Ctx &cx, int px, int py, int width, int height, int spp, float out[4])
{
    const shray_frame_params &p = *cx.p;
    if (p.which == 3) {   // fs:642-650: the pixel's own differentials, no trace, no tone map
        const ray r = primary_ray(p, ((float)px + 0.5f) / (float)width, ((float)py + 0.5f) / (float)height);
        float sb, tb, sa, ta;
        environment_map_coords(r.D - r.dDdy / 2.0f, sb, tb);
        environment_map_coords(r.D + r.dDdy / 2.0f, sa, ta);
        out[0] = fabsf(sa - sb) * 1.0f * 100;
        out[1] = fabsf(ta - tb) * 1.0f * 100;
        out[2] = 0.0f;
        out[3] = 1.0f;
        return;
    }
    if (p.which == 5) {   // fs:654-673: 5 x 5 supersampled reference image
        const vec3 right = V(p.right[0], p.right[1], p.right[2]), up = V(p.up[0], p.up[1], p.up[2]);
        vec3 origin, direction;
        interpolated_varyings(p, ((float)px + 0.5f) / (float)width, ((float)py + 0.5f) / (float)height, origin, direction);
        vec3 result = V(0, 0, 0);
        const int blarg = 5;
        for (int i = 0; i < blarg; i++) {
            for (int j = 0; j < blarg; j++) {
                const float u = ((float)i / float(blarg) - .5f);
                const float v = ((float)j / float(blarg) - .5f);
                const ray r = make_ray(origin, normalize(direction + u * .2f * right + v * .2f * up));
                result = result + trace(cx, r);
            }
        }
        result = result / (float)(blarg * blarg);
        if (p.tonemap)
            result = V(filmic(result.x), filmic(result.y), filmic(result.z));
        out[0] = result.x;
        out[1] = result.y;
        out[2] = result.z;
        out[3] = 1.0f;
        return;
    }
    vec3 sum = V(0, 0, 0);
    const uint64_t visits_before = cx.c.node_visits + (cx.c.triangle_tests << 32);
    for (int s = 0; s < spp; s++) {
        const float ox = ((float)s + 0.5f) / (float)spp;
        const float oy = (float)bitreverse32((uint32_t)s) * 2.3283064365386963e-10f + 0.5f / (float)spp;
        const float u = ((float)px + ox) / (float)width;
        const float v = ((float)py + oy) / (float)height;
        if (g_trace)
            g_trace->sample_offsets[((size_t)py * width + px) * spp + s] = g_trace->length;
        const vec3 radiance = trace(cx, primary_ray(p, u, v));
        sum = (spp == 1) ? radiance : sum + radiance;
    }
    vec3 result = (spp == 1) ? sum : sum / (float)spp;
    if (p.tonemap)   // fs:675-679, use_filmic fs:524
        result = V(filmic(result.x), filmic(result.y), filmic(result.z));
    out[0] = result.x;
    out[1] = result.y;
    out[2] = result.z;
    out[3] = 1.0f;
    if (g_path_map && spp == 1) {
        g_path_map[(size_t)py * width + px] = cx.path;
        if (g_first_triangle_map)
            g_first_triangle_map[(size_t)py * width + px] = cx.first_triangle;
        if (g_edge_margin_map)
            g_edge_margin_map[(size_t)py * width + px] = cx.edge_margin;
        if (g_env_dy_map)
            g_env_dy_map[(size_t)py * width + px] = cx.env_dy;
        if (g_decision_map)
            g_decision_map[(size_t)py * width + px] = cx.decision_margin;
    }
    if (g_visit_map)
    {
        const uint64_t now = cx.c.node_visits + (cx.c.triangle_tests << 32), delta = now - visits_before;
        // low 16 bits: node visits, high 16 bits: triangle tests (both saturating)
        const uint32_t nv = (uint32_t)std::min<uint64_t>(delta & 0xffffffffu, 0xffffu);
        const uint32_t tt = (uint32_t)std::min<uint64_t>(delta >> 32, 0xffffu);
        g_visit_map[(size_t)py * width + px] = nv | (tt << 16);
    }
}

}   // namespace

extern "C" {

// Renders rows [row_begin, row_end) (row 0 = bottom) of a width x height frame
// into rgba_out (full-frame RGBA float32, row-major from the bottom; only the
// requested rows are written).  threads <= 0 = hardware_concurrency().
// Returns 0, or -1 on bad arguments.
int shray_oracle_render(const shray_scene_desc *desc, const float *env_rgb, int env_w, int env_h,
                        const shray_frame_params *params, int width, int height, int spp, int row_begin, int row_end,
                        int threads, float *rgba_out, shray_counters *counters_out)
{
    if (!desc || !env_rgb || !params || !rgba_out || width <= 0 || height <= 0 || spp <= 0 || env_w <= 0 || env_h <= 0)
        return -1;
    if ((params->which == 3 || params->which == 5) && spp != 1)
        return -1;   // the 3 / 5 views are per-pixel, not per-sample
    row_begin = std::max(0, row_begin);
    row_end = std::min(height, row_end);

    Scene scene;
    scene.width = desc->data_texture_width;
    scene.group_rows = desc->group_data_rows;
    scene.tree_root = (float)desc->tree_root;
    scene.positions = desc->vertex_positions;
    scene.normals.assign(desc->vertex_normals, desc->vertex_normals + 3 * (size_t)desc->vertex_count);
    if (params->normals_fp16)
        for (float &n : scene.normals)
            n = round_through_half(n);
    scene.boxmin = desc->group_boxmin;
    scene.boxmax = desc->group_boxmax;
    scene.hitmiss = desc->group_hitmiss;
    scene.objects = desc->group_objects;
    std::vector<float> stored_env;
    if (g_env_storage == 1) {
        stored_env.assign(env_rgb, env_rgb + 3 * (size_t)env_w * env_h);
        for (float &texel : stored_env)
            texel = through_unorm8(texel);
        env_rgb = stored_env.data();
    }
    scene.env = env_rgb;
    scene.env_w = env_w;
    scene.env_h = env_h;
    if (params->which == 1)
        build_mips(scene);

    int nthreads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nthreads = std::max(1, std::min(nthreads, std::max(1, row_end - row_begin)));
    std::vector<Counters> per_thread(nthreads);
    std::atomic<int> next_row(row_begin);
    auto worker = [&](int tid) {
        Ctx cx{&scene, params, Counters()};
        for (;;) {
            const int py = next_row.fetch_add(1);
            if (py >= row_end)
                break;
            for (int px = 0; px < width; px++)
                shade_pixel(cx, px, py, width, height, spp, rgba_out + 4 * ((size_t)py * width + px));
        }
        per_thread[tid] = cx.c;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++)
        pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool)
        t.join();

    if (counters_out) {
        memset(counters_out, 0, sizeof(*counters_out));
        for (const Counters &c : per_thread) {
            counters_out->node_visits += c.node_visits;
            counters_out->leaf_visits += c.leaf_visits;
            counters_out->triangle_tests += c.triangle_tests;
            counters_out->shaded_hits += c.shaded_hits;
            counters_out->env_lookups += c.env_lookups;
            counters_out->traversals += c.traversals;
            counters_out->bad_hits += c.bad_hits;
        }
        counters_out->samples = (uint64_t)width * (uint64_t)(row_end - row_begin) * (uint64_t)spp;
    }
    return 0;
}

// Diagnostics: the quantised-record probe (g_quant_step): step > 0 starts counting from zero, 0 stops; counts[2] = {visits, undecided}.
void shray_oracle_quant_probe(float step, unsigned long long counts[2])
{
    if (counts) {
        counts[0] = g_quant_visits.load();
        counts[1] = g_quant_undecided.load();
    }
    g_quant_visits = 0;
    g_quant_undecided = 0;
    g_quant_step = step;
}

// Diagnostics: when set, shray_oracle_render also writes node visits per pixel (width*height uint32).
void shray_oracle_set_visit_map(uint32_t *map) { g_visit_map = map; }

// Diagnostics: when set, 1 spp plain frames also write the path planes described at g_path_map (width * height each;
// first_triangle and edge_margin may be NULL).  Pass NULLs to stop.
void shray_oracle_set_path_map(uint32_t *path, int32_t *first_triangle, float *edge_margin, float *env_dy)
{
    g_path_map = path;
    g_first_triangle_map = first_triangle;
    g_edge_margin_map = edge_margin;
    g_env_dy_map = env_dy;
}

// Diagnostics: with the path map set, also the decision-margin plane (Ctx::decision_margin; width * height floats).  NULL to stop.
void shray_oracle_set_decision_map(float *decision_margin) { g_decision_map = decision_margin; }

// Diagnostics: records the event trace of the next renders (call with threads = 1, whole frame, in pixel order);
// sample_offsets needs width*height*spp + 1 entries.  shray_oracle_trace_end returns the bytes the trace needed.
void shray_oracle_trace_begin(uint8_t *bytes, uint64_t capacity, uint64_t *sample_offsets, uint64_t samples)
{
    static VisitTrace trace;
    trace = VisitTrace{bytes, capacity, 0, sample_offsets, samples};
    g_trace = &trace;
}
uint64_t shray_oracle_trace_end()
{
    if (!g_trace)
        return 0;
    const uint64_t n = g_trace->length;
    g_trace->sample_offsets[g_trace->samples] = n;
    g_trace = nullptr;
    return n;
}

// Pieces exposed for the known-answer tests.
float shray_oracle_filmic(float c) { return filmic(c); }
float shray_oracle_half(float f) { return round_through_half(f); }
float shray_oracle_atan2(float y, float x) { return sr_atan2(y, x); }
float shray_oracle_acos(float x) { return sr_acos(x); }
float shray_oracle_pow5(float x) { return sr_pow5(x); }
float shray_oracle_log2(float x) { return sr_log2(x); }
// textureGrad of a given image (KATs): out = rgb
void shray_oracle_texture_grad(const float *rgb, int w, int h, float s, float t, float dudx, float dvdx, float dudy,
                               float dvdy, float out[3])
{
    Scene sc;
    sc.env = rgb;
    sc.env_w = w;
    sc.env_h = h;
    build_mips(sc);
    const vec3 v = texture_grad(sc, s, t, dudx, dvdx, dudy, dvdy);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void shray_oracle_set_env_storage(int storage) { g_env_storage = storage; }
void shray_oracle_schlick(const float cspec[3], const float v[3], const float r[3], float out[3])
{
    const vec3 f = f_schlick_vr(V(cspec[0], cspec[1], cspec[2]), V(v[0], v[1], v[2]), V(r[0], r[1], r[2]));
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}
void shray_oracle_primary_ray(const shray_frame_params *p, float u, float v, float origin[3], float direction[3])
{
    const ray r = primary_ray(*p, u, v);
    origin[0] = r.P.x; origin[1] = r.P.y; origin[2] = r.P.z;
    direction[0] = r.D.x; direction[1] = r.D.y; direction[2] = r.D.z;
}

}   // extern "C"
