// glsl_ref.cpp -- TEST INFRASTRUCTURE: runs the REFERENCE'S OWN shaders (raytracer.vs, raytracer.es.fs, read from
// the reference tree at run time, never copied) on a CPU OpenGL implementation that happens to be in this image, so
// that the CPU oracle (oracle/shader_oracle.cpp) can be pinned against outputs of the reference itself.
//
// The reference's host (ray.cpp) needs GLFW and FreeImagePlus and is unbuildable here; what it does around the
// shaders is restated below, call for call, with the line it follows:
//   shader preamble + compile + link      ray.cpp:398-430
//   data textures (formats, filtering)    ray.cpp:348-355, :470-497
//   background texture + mipmaps          ray.cpp:499-510
//   screen quad                           ray.cpp:517-566
//   uniforms + draw                       ray.cpp:599-707
//   readback                              ray.cpp:760
// The GL context comes from Mesa's llvmpipe (the image's /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so, a CPU
// implementation of desktop OpenGL 4.5), driven through the driver's own loader interface
// (/usr/include/GL/internal/dri_interface.h, shipped with the image's mesa-common-dev: DRI_SWRast with a loader whose
// "window" is a 16x16 stub) -- no X server, no EGL, no GPU.  The context is what ray.cpp:964-967 asks GLFW for: 3.2
// core, forward compatible; the version line is ray.cpp:401's "#version 140"; the shader files are compiled
// unmodified.  One deliberate difference: the frame is rendered into an RGBA32F renderbuffer instead of the window, so
// that the pixels can be read back as the floats the shader wrote (the reference displays them on an 8-bit window).
// The background's internal format is either the reference's literal unsized GL_RGB (ray.cpp:508: the driver
// chooses; Mesa stores 8-bit normalized) or the sized GL_RGB32F (float storage, this repository's default contract).
//
// Only tests/golden/make_glsl_reference.py calls this (to write fixtures); nothing in the product does.
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <GL/internal/dri_interface.h>

#include "shader_ray_hip.h"

namespace {

typedef unsigned int GLenum, GLuint, GLbitfield;
typedef int GLint, GLsizei;
typedef float GLfloat;
typedef char GLchar;
typedef unsigned char GLboolean;
typedef void (*proc)(void);

struct GL {
    proc (*GetProcAddress)(const char *) = nullptr;     // libglapi's _glapi_get_proc_address
    template <class F>
    void load(F &f, const char *name)
    {
        f = (F)GetProcAddress(name);
    }
#define GLFN(ret, name, ...) ret (*name)(__VA_ARGS__) = nullptr
    GLFN(GLenum, glGetError, void);
    GLFN(GLuint, glCreateShader, GLenum);
    GLFN(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
    GLFN(void, glCompileShader, GLuint);
    GLFN(void, glGetShaderiv, GLuint, GLenum, GLint *);
    GLFN(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
    GLFN(GLuint, glCreateProgram, void);
    GLFN(void, glAttachShader, GLuint, GLuint);
    GLFN(void, glBindAttribLocation, GLuint, GLuint, const GLchar *);
    GLFN(void, glLinkProgram, GLuint);
    GLFN(void, glGetProgramiv, GLuint, GLenum, GLint *);
    GLFN(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
    GLFN(void, glUseProgram, GLuint);
    GLFN(GLint, glGetUniformLocation, GLuint, const GLchar *);
    GLFN(void, glUniform1i, GLint, GLint);
    GLFN(void, glUniform1f, GLint, GLfloat);
    GLFN(void, glUniform3fv, GLint, GLsizei, const GLfloat *);
    GLFN(void, glUniform3f, GLint, GLfloat, GLfloat, GLfloat);
    GLFN(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *);
    GLFN(void, glGenTextures, GLsizei, GLuint *);
    GLFN(void, glBindTexture, GLenum, GLuint);
    GLFN(void, glTexParameteri, GLenum, GLenum, GLint);
    GLFN(void, glTexParameterf, GLenum, GLenum, GLfloat);
    GLFN(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
    GLFN(void, glGenerateMipmap, GLenum);
    GLFN(void, glActiveTexture, GLenum);
    GLFN(void, glGenBuffers, GLsizei, GLuint *);
    GLFN(void, glBindBuffer, GLenum, GLuint);
    GLFN(void, glBufferData, GLenum, intptr_t, const void *, GLenum);
    GLFN(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
    GLFN(void, glEnableVertexAttribArray, GLuint);
    GLFN(void, glGenVertexArrays, GLsizei, GLuint *);
    GLFN(void, glBindVertexArray, GLuint);
    GLFN(void, glGenFramebuffers, GLsizei, GLuint *);
    GLFN(void, glBindFramebuffer, GLenum, GLuint);
    GLFN(void, glGenRenderbuffers, GLsizei, GLuint *);
    GLFN(void, glBindRenderbuffer, GLenum, GLuint);
    GLFN(void, glRenderbufferStorage, GLenum, GLenum, GLsizei, GLsizei);
    GLFN(void, glFramebufferRenderbuffer, GLenum, GLenum, GLenum, GLuint);
    GLFN(GLenum, glCheckFramebufferStatus, GLenum);
    GLFN(void, glViewport, GLint, GLint, GLsizei, GLsizei);
    GLFN(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
    GLFN(void, glClear, GLbitfield);
    GLFN(void, glDrawArrays, GLenum, GLint, GLsizei);
    GLFN(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
    GLFN(void, glFinish, void);
    GLFN(void, glPixelStorei, GLenum, GLint);
    GLFN(const unsigned char *, glGetString, GLenum);
#undef GLFN
};

enum : GLenum {
    GL_TEXTURE_2D = 0x0DE1, GL_TEXTURE_MIN_FILTER = 0x2801, GL_TEXTURE_MAG_FILTER = 0x2800, GL_NEAREST = 0x2600, GL_LINEAR = 0x2601,
    GL_LINEAR_MIPMAP_LINEAR = 0x2703, GL_RGB = 0x1907, GL_RG = 0x8227, GL_RGBA = 0x1908, GL_FLOAT = 0x1406, GL_UNSIGNED_BYTE = 0x1401,
    GL_RGB32F = 0x8815, GL_RGB16F = 0x881B, GL_RG32F = 0x8230, GL_RGBA32F = 0x8814, GL_RGB8 = 0x8051, GL_TEXTURE0 = 0x84C0,
    GL_FRAGMENT_SHADER = 0x8B30, GL_VERTEX_SHADER = 0x8B31, GL_COMPILE_STATUS = 0x8B81, GL_LINK_STATUS = 0x8B82, GL_ARRAY_BUFFER = 0x8892,
    GL_STATIC_DRAW = 0x88E4, GL_TRIANGLE_STRIP = 0x0005, GL_FRAMEBUFFER = 0x8D40, GL_RENDERBUFFER = 0x8D41, GL_COLOR_ATTACHMENT0 = 0x8CE0,
    GL_FRAMEBUFFER_COMPLETE = 0x8CD5, GL_COLOR_BUFFER_BIT = 0x4000, GL_DEPTH_BUFFER_BIT = 0x100, GL_TEXTURE_MAX_ANISOTROPY_EXT = 0x84FE,
    GL_PACK_ALIGNMENT = 0x0D05, GL_UNPACK_ALIGNMENT = 0x0CF5, GL_VERSION = 0x1F02, GL_RENDERER = 0x1F01
};

// the loader side of DRI_SWRast: a 16 x 16 "window" nobody looks at (the frame goes to a framebuffer object)
void stub_drawable_info(__DRIdrawable *, int *x, int *y, int *w, int *h, void *)
{
    *x = *y = 0;
    *w = *h = 16;
}
void stub_put_image(__DRIdrawable *, int, int, int, int, int, char *, void *) {}
void stub_get_image(__DRIdrawable *, int, int, int width, int height, char *data, void *) { memset(data, 0, (size_t)width * height * 4); }
void stub_put_image2(__DRIdrawable *, int, int, int, int, int, int, char *, void *) {}
void stub_get_image2(__DRIdrawable *, int, int, int, int height, int stride, char *data, void *) { memset(data, 0, (size_t)stride * height); }

std::string read_file(const std::string &path)
{
    std::string text;
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp)
        return text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0)
        text.append(buf, n);
    fclose(fp);
    return text;
}

void say(std::string &log, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void say(std::string &log, const char *fmt, ...)
{
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    log += buf;
    log += "\n";
}

}   // namespace

extern "C" {

// Renders one frame of the reference's shaders.  Returns 0, or a negative code with the reason in `log`.
//   reference_dir   where raytracer.vs / raytracer.es.fs lie (the reference tree)
//   dri_driver      the Mesa software driver, e.g. /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so
//   background_mode 0: background stored as the sized GL_RGB32F; 1: the reference's literal unsized GL_RGB with float data
//                   (ray.cpp:508: the driver chooses the storage)
//   rgba_out        width * height * 4 floats, row 0 = bottom row (glReadPixels order, ray.cpp:760)
int shray_glsl_ref_render(const char *reference_dir, const char *dri_driver, const shray_scene_desc *desc, const float *env_rgb,
                          int env_w, int env_h, int background_mode, const shray_frame_params *p, int width, int height, float *rgba_out,
                          char *log_out, int log_len)
{
    std::string log;
    auto finish = [&](int code) {
        if (log_out && log_len > 0) {
            strncpy(log_out, log.c_str(), (size_t)log_len - 1);
            log_out[log_len - 1] = 0;
        }
        return code;
    };
    GL gl;
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    void *driver = dlopen(dri_driver, RTLD_NOW | RTLD_GLOBAL);
    if (!glapi || !driver) {
        say(log, "cannot load libglapi.so.0 / %s: %s", dri_driver, dlerror());
        return finish(-1);
    }
    gl.GetProcAddress = (proc(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    const __DRIextension **(*driver_extensions)(void) = (const __DRIextension **(*)(void))dlsym(driver, "__driDriverGetExtensions_swrast");
    if (!gl.GetProcAddress || !driver_extensions) {
        say(log, "the driver does not export __driDriverGetExtensions_swrast");
        return finish(-1);
    }
    const __DRIextension **extensions = driver_extensions();
    const __DRIcoreExtension *core = nullptr;
    const __DRIswrastExtension *swrast = nullptr;
    for (int k = 0; extensions[k]; k++) {
        if (!strcmp(extensions[k]->name, __DRI_CORE))
            core = (const __DRIcoreExtension *)extensions[k];
        if (!strcmp(extensions[k]->name, __DRI_SWRAST))
            swrast = (const __DRIswrastExtension *)extensions[k];
    }
    if (!core || !swrast || swrast->base.version < 4) {
        say(log, "the driver lacks DRI_Core / DRI_SWRast version 4");
        return finish(-2);
    }
    static const __DRIswrastLoaderExtension loader = {{__DRI_SWRAST_LOADER, 3}, stub_drawable_info, stub_put_image, stub_get_image,
                                                     stub_put_image2, stub_get_image2, nullptr, nullptr, nullptr, nullptr};
    const __DRIextension *loader_extensions[] = {&loader.base, nullptr};
    const __DRIconfig **configs = nullptr;
    __DRIscreen *screen = swrast->createNewScreen2(0, loader_extensions, extensions, &configs, nullptr);
    if (!screen || !configs || !configs[0]) {
        say(log, "createNewScreen2 failed");
        return finish(-2);
    }
    // the context ray.cpp:964-967 asks for: OpenGL 3.2, core profile, forward compatible
    const uint32_t attributes[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 2, __DRI_CTX_ATTRIB_FLAGS,
                                   __DRI_CTX_FLAG_FORWARD_COMPATIBLE};
    unsigned error = 0;
    __DRIcontext *context = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], nullptr, 3, attributes, &error, nullptr);
    __DRIdrawable *drawable = swrast->createNewDrawable(screen, configs[0], nullptr);
    if (!context || !drawable || !core->bindContext(context, drawable, drawable)) {
        say(log, "cannot make an OpenGL 3.2 core context current (DRI error %u)", error);
        return finish(-2);
    }
#define L(name) gl.load(gl.name, #name)
    L(glGetError); L(glCreateShader); L(glShaderSource); L(glCompileShader); L(glGetShaderiv); L(glGetShaderInfoLog); L(glCreateProgram);
    L(glAttachShader); L(glBindAttribLocation); L(glLinkProgram); L(glGetProgramiv); L(glGetProgramInfoLog); L(glUseProgram);
    L(glGetUniformLocation); L(glUniform1i); L(glUniform1f); L(glUniform3fv); L(glUniform3f); L(glUniformMatrix4fv); L(glGenTextures);
    L(glBindTexture); L(glTexParameteri); L(glTexParameterf); L(glTexImage2D); L(glGenerateMipmap); L(glActiveTexture); L(glGenBuffers);
    L(glBindBuffer); L(glBufferData); L(glVertexAttribPointer); L(glEnableVertexAttribArray); L(glGenVertexArrays); L(glBindVertexArray);
    L(glGenFramebuffers); L(glBindFramebuffer); L(glGenRenderbuffers); L(glBindRenderbuffer); L(glRenderbufferStorage);
    L(glFramebufferRenderbuffer); L(glCheckFramebufferStatus); L(glViewport); L(glClearColor); L(glClear); L(glDrawArrays); L(glReadPixels);
    L(glFinish); L(glPixelStorei); L(glGetString);
#undef L
    say(log, "GL_VERSION %s; GL_RENDERER %s", gl.glGetString(GL_VERSION), gl.glGetString(GL_RENDERER));

    // ---- the shaders, from the reference tree, with ray.cpp:398-404's two prefix strings (version line: see header)
    const std::string fs_text = read_file(std::string(reference_dir) + "/raytracer.es.fs");
    const std::string vs_text = read_file(std::string(reference_dir) + "/raytracer.vs");
    if (fs_text.empty() || vs_text.empty()) {
        say(log, "cannot read raytracer.vs / raytracer.es.fs under %s", reference_dir);
        return finish(-3);
    }
    char preamble[128];
    snprintf(preamble, sizeof preamble, "const int data_texture_width = %u;\n", desc->data_texture_width);
    const char *version = "#version 140\n";      // ray.cpp:401
    auto compile = [&](GLenum kind, const std::string &text, const char *what) -> GLuint {
        const GLuint shader = gl.glCreateShader(kind);
        const char *strings[3] = {version, preamble, text.c_str()};
        gl.glShaderSource(shader, 3, strings, nullptr);
        gl.glCompileShader(shader);
        GLint ok = 0;
        gl.glGetShaderiv(shader, GL_COMPILE_STATUS, &ok);
        char info[16384] = "";
        gl.glGetShaderInfoLog(shader, sizeof info, nullptr, info);
        if (info[0])
            say(log, "%s compile log: %s", what, info);
        return ok ? shader : 0;
    };
    const GLuint fs = compile(GL_FRAGMENT_SHADER, fs_text, "fragment shader");
    const GLuint vs = compile(GL_VERTEX_SHADER, vs_text, "vertex shader");
    if (!fs || !vs)
        return finish(-4);
    const GLuint program = gl.glCreateProgram();
    gl.glAttachShader(program, vs);
    gl.glAttachShader(program, fs);
    gl.glBindAttribLocation(program, 0, "pos");      // ray.cpp:420-421
    gl.glBindAttribLocation(program, 1, "vtex");
    gl.glLinkProgram(program);
    GLint linked = 0;
    gl.glGetProgramiv(program, GL_LINK_STATUS, &linked);
    if (!linked) {
        char info[8192] = "";
        gl.glGetProgramInfoLog(program, sizeof info, nullptr, info);
        say(log, "link log: %s", info);
        return finish(-4);
    }
    gl.glUseProgram(program);

    // ---- render target: RGBA32F, so that the fragment colours come back as the floats the shader wrote
    GLuint fbo = 0, rbo = 0;
    gl.glGenFramebuffers(1, &fbo);
    gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    gl.glGenRenderbuffers(1, &rbo);
    gl.glBindRenderbuffer(GL_RENDERBUFFER, rbo);
    gl.glRenderbufferStorage(GL_RENDERBUFFER, GL_RGBA32F, width, height);
    gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, rbo);
    if (gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
        say(log, "RGBA32F framebuffer incomplete");
        return finish(-5);
    }
    gl.glViewport(0, 0, width, height);
    gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    gl.glPixelStorei(GL_PACK_ALIGNMENT, 1);

    // ---- data textures: NEAREST (ray.cpp:348-355), formats of ray.cpp:470-497
    auto data_texture = [&](GLenum internal, GLenum format, int rows, const float *data) -> GLuint {
        GLuint tex = 0;
        gl.glGenTextures(1, &tex);
        gl.glBindTexture(GL_TEXTURE_2D, tex);
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, (GLint)GL_NEAREST);
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, (GLint)GL_NEAREST);
        gl.glTexImage2D(GL_TEXTURE_2D, 0, (GLint)internal, (GLsizei)desc->data_texture_width, rows, 0, format, GL_FLOAT, data);
        return tex;
    };
    const GLuint positions = data_texture(GL_RGB32F, GL_RGB, (int)desc->vertex_data_rows, desc->vertex_positions);
    const GLuint normals = data_texture(GL_RGB16F, GL_RGB, (int)desc->vertex_data_rows, desc->vertex_normals);
    const GLuint objects = data_texture(GL_RG32F, GL_RG, desc->group_data_rows, desc->group_objects);
    const GLuint hitmiss = data_texture(GL_RG32F, GL_RG, desc->group_data_rows * 8, desc->group_hitmiss);
    const GLuint boxmin = data_texture(GL_RGB32F, GL_RGB, desc->group_data_rows, desc->group_boxmin);
    const GLuint boxmax = data_texture(GL_RGB32F, GL_RGB, desc->group_data_rows, desc->group_boxmax);
    // ---- background (ray.cpp:499-510): trilinear, MAG linear, 4x anisotropy, mipmaps generated by GL
    GLuint background = 0;
    gl.glGenTextures(1, &background);
    gl.glBindTexture(GL_TEXTURE_2D, background);
    gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, (GLint)GL_LINEAR_MIPMAP_LINEAR);
    gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, (GLint)GL_LINEAR);
    // (SHRAY_GLSL_REF_ANISOTROPY: diagnostic override of ray.cpp:506's 4.0, to tell the shader from the driver's filter)
    const char *aniso_override = getenv("SHRAY_GLSL_REF_ANISOTROPY");
    gl.glTexParameterf(GL_TEXTURE_2D, GL_TEXTURE_MAX_ANISOTROPY_EXT, aniso_override ? (float)atof(aniso_override) : 4.0f);
    gl.glTexImage2D(GL_TEXTURE_2D, 0, (GLint)(background_mode == 1 ? GL_RGB : GL_RGB32F), env_w, env_h, 0, GL_RGB, GL_FLOAT, env_rgb);   // ray.cpp:508
    gl.glGenerateMipmap(GL_TEXTURE_2D);
    if (GLenum e = gl.glGetError())
        say(log, "GL error 0x%x after the texture uploads", e);

    // ---- screen quad (ray.cpp:517-566)
    const float verts[4][4] = {{-1, -1, 0, 1}, {1, -1, 0, 1}, {-1, 1, 0, 1}, {1, 1, 0, 1}};
    const float texcoords[4][2] = {{0, 1}, {1, 1}, {0, 0}, {1, 0}};
    GLuint vao = 0, buffers[2] = {0, 0};
    gl.glGenVertexArrays(1, &vao);
    gl.glBindVertexArray(vao);
    gl.glGenBuffers(2, buffers);
    gl.glBindBuffer(GL_ARRAY_BUFFER, buffers[0]);
    gl.glBufferData(GL_ARRAY_BUFFER, sizeof verts, verts, GL_STATIC_DRAW);
    gl.glVertexAttribPointer(0, 4, GL_FLOAT, 0, 0, nullptr);
    gl.glEnableVertexAttribArray(0);
    gl.glBindBuffer(GL_ARRAY_BUFFER, buffers[1]);
    gl.glBufferData(GL_ARRAY_BUFFER, sizeof texcoords, texcoords, GL_STATIC_DRAW);
    gl.glVertexAttribPointer(1, 2, GL_FLOAT, 0, 0, nullptr);
    gl.glEnableVertexAttribArray(1);

    // ---- DrawFrame (ray.cpp:599-707)
    gl.glClearColor(1, 0, 0, 1);
    gl.glClear(GL_COLOR_BUFFER_BIT);
    int unit = 0;
    auto bind = [&](GLuint tex, const char *name) {
        gl.glActiveTexture(GL_TEXTURE0 + (GLenum)unit);
        gl.glBindTexture(GL_TEXTURE_2D, tex);
        gl.glUniform1i(gl.glGetUniformLocation(program, name), unit);
        unit++;
    };
    bind(positions, "vertex_positions");
    bind(normals, "vertex_normals");
    bind(objects, "group_objects");
    bind(hitmiss, "group_hitmiss");
    bind(boxmin, "group_boxmin");
    bind(boxmax, "group_boxmax");
    bind(background, "background");
    auto loc = [&](const char *name) { return gl.glGetUniformLocation(program, name); };
    gl.glUniform1i(loc("which"), p->which);
    gl.glUniform1f(loc("tree_root"), (float)desc->tree_root);
    gl.glUniform1i(loc("vertex_data_rows"), (GLint)desc->vertex_data_rows);
    gl.glUniform1i(loc("group_data_rows"), desc->group_data_rows);
    gl.glUniformMatrix4fv(loc("camera_matrix"), 1, 0, p->camera_matrix);
    gl.glUniformMatrix4fv(loc("camera_normal_matrix"), 1, 0, p->camera_normal_matrix);
    gl.glUniformMatrix4fv(loc("object_matrix"), 1, 0, p->object_matrix);
    gl.glUniformMatrix4fv(loc("object_inverse"), 1, 0, p->object_inverse);
    gl.glUniformMatrix4fv(loc("object_normal_matrix"), 1, 0, p->object_normal_matrix);
    gl.glUniformMatrix4fv(loc("object_normal_inverse"), 1, 0, p->object_normal_inverse);
    gl.glUniform1f(loc("image_plane_width"), p->image_plane_width);
    gl.glUniform1f(loc("aspect"), p->aspect);
    gl.glUniform3fv(loc("right"), 1, p->right);
    gl.glUniform3fv(loc("up"), 1, p->up);
    const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    gl.glUniformMatrix4fv(loc("modelview"), 1, 0, identity);
    gl.glUniform3fv(loc("light_dir"), 1, p->light_dir);
    gl.glUniform3fv(loc("specular_color"), 1, p->specular_color);
    gl.glUniform3fv(loc("diffuse_color"), 1, p->diffuse_color);
    gl.glBindVertexArray(vao);
    gl.glDrawArrays(GL_TRIANGLE_STRIP, 0, 4);
    gl.glFinish();
    if (GLenum e = gl.glGetError())
        say(log, "GL error 0x%x after the draw", e);
    gl.glReadPixels(0, 0, width, height, GL_RGBA, GL_FLOAT, rgba_out);      // ray.cpp:760 (floats instead of bytes)
    if (GLenum e = gl.glGetError()) {
        say(log, "GL error 0x%x from glReadPixels", e);
        return finish(-6);
    }
    core->unbindContext(context);
    core->destroyDrawable(drawable);
    core->destroyContext(context);
    core->destroyScreen(screen);
    return finish(0);
}

}   // extern "C"
