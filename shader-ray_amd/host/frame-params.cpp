// frame-params.cpp -- see frame-params.h.  Operation order is the
// reference's; results are compared bit-for-bit with values produced by the
// compiled reference (tests/golden).
#include "frame-params.h"

#include "frame_params_defaults.h"

const material materials[7] = {
    {{1, .71f, .29f}, true},       // gold
    {{.95f, .95f, 0.88f}, true},   // silver
    {{0.95f, 0.64f, 0.54f}, true}, // copper
    {{0.56f, 0.57f, 0.58f}, true}, // iron
    {{0.91f, 0.92f, 0.92f}, true}, // aluminium
    {{.03f, .03f, .03f}, false},   // plastic / glass (low)
    {{.05f, .05f, .05f}, false},   // plastic (high), the "glazed plaster" look
};
const int material_count = 7;

const vec3 diffuse_colors[4] = {{1, 1, 1}, {1, .5f, .5f}, {.25f, 1, .25f}, {.5f, .5f, 1}};
const int diffuse_color_count = 4;

namespace {
void clear_projective_row(float m[16]) { m[3] = m[7] = m[11] = 0.0f; }
}   // namespace

// Eye -> world: a pure translation to the viewpoint; normals go through the
// inverse transpose with its projective row cleared.
void create_camera_matrix(const vec3 &viewpoint, float matrix[16], float normal_matrix[16])
{
    float to_viewpoint[16];
    mat4_make_identity(matrix);
    mat4_make_translation(viewpoint.x, viewpoint.y, viewpoint.z, to_viewpoint);
    mat4_mult(to_viewpoint, matrix, matrix);

    mat4_invert(matrix, normal_matrix);
    mat4_transpose(normal_matrix, normal_matrix);
    clear_projective_row(normal_matrix);
}

// World -> object: rotation times translation by (center + position).
void create_object_matrix(const vec3 &center, const float rotation[4], const vec3 &position, float matrix[16],
                          float inverse[16], float normal[16], float normal_inverse[16])
{
    float shift[16];
    mat4_make_rotation(rotation[0], rotation[1], rotation[2], rotation[3], matrix);
    mat4_make_translation(center.x + position.x, center.y + position.y, center.z + position.z, shift);
    mat4_mult(matrix, shift, matrix);

    mat4_invert(matrix, inverse);

    mat4_transpose(matrix, normal);
    mat4_invert(normal, normal);
    clear_projective_row(normal);

    mat4_transpose(matrix, normal_inverse);
    clear_projective_row(normal_inverse);
}

vec3 compute_light_dir(const float light_rotation[4])
{
    float rotation[16], transposed[16], for_normals[16];
    mat4_make_rotation(light_rotation[0], light_rotation[1], light_rotation[2], light_rotation[3], rotation);
    mat4_transpose(rotation, transposed);
    mat4_invert(transposed, for_normals);
    clear_projective_row(for_normals);
    const vec4 l = for_normals * vec4(0, 0, 1, 0);
    return vec3(l.x, l.y, l.z);
}

void update_view_params(world_ptr w, float zoom, const float object_rotation[4], const vec3 &object_position)
{
    create_camera_matrix(vec3(0, 0, zoom), w->camera_matrix, w->camera_normal_matrix);
    create_object_matrix(w->scene_center, object_rotation, object_position, w->object_matrix, w->object_inverse,
                         w->object_normal_matrix, w->object_normal_inverse);
}

void drag_to_rotation(float dx, float dy, float rotation[4])
{
    // the reference scales by 10000 before the square root "to decrease chance of underflow"
    const float dist = sqrt(dx * 10000 * dx * 10000 + dy * 10000 * dy * 10000) / 10000;
    rotation[0] = M_PI * dist;
    rotation[1] = dy / dist;
    rotation[2] = dx / dist;
    rotation[3] = 0.0f;
}

void trackball_motion(float prevrotation[4], float dx, float dy, float newrotation[4])
{
    if (dx != 0 || dy != 0) {
        float drag[4];
        drag_to_rotation(dx, dy, drag);
        rotation_mult_rotation(prevrotation, drag, newrotation);
    }
}

view_state default_view_state(const world_ptr &w)
{
    view_state s;
    s.fov = to_radians(40.0);
    s.zoom = w->scene_extent / 2 / sinf(s.fov / 2);
    s.object_rotation[0] = s.object_rotation[1] = s.object_rotation[2] = s.object_rotation[3] = 0;
    s.object_position = vec3(0, 0, 0);
    // 20 degrees about an axis halfway between +X and -Y
    s.light_rotation[0] = to_radians(-20.0);
    s.light_rotation[1] = .707f;
    s.light_rotation[2] = -.707f;
    s.light_rotation[3] = 0;
    s.which = 0;
    s.which_material = 0;
    s.which_diffuse_color = 0;
    return s;
}

void make_frame_params(world_ptr w, const view_state &view, int width, int height, shray_frame_params *out)
{
    shray_frame_params_defaults(out);

    w->cam.fov = view.fov;
    w->xsub = w->ysub = 1;
    update_view_params(w, view.zoom, view.object_rotation, view.object_position);

    out->which = view.which;
    memcpy(out->camera_matrix, w->camera_matrix, sizeof(out->camera_matrix));
    memcpy(out->camera_normal_matrix, w->camera_normal_matrix, sizeof(out->camera_normal_matrix));
    memcpy(out->object_matrix, w->object_matrix, sizeof(out->object_matrix));
    memcpy(out->object_inverse, w->object_inverse, sizeof(out->object_inverse));
    memcpy(out->object_normal_matrix, w->object_normal_matrix, sizeof(out->object_normal_matrix));
    memcpy(out->object_normal_inverse, w->object_normal_inverse, sizeof(out->object_normal_inverse));

    const float image_plane_width = 2 * tanf(view.fov / 2.0);
    const float aspect = height / (1.0f * width);
    out->image_plane_width = image_plane_width;
    out->aspect = aspect;

    const vec4 right = w->camera_normal_matrix * vec4(image_plane_width / width, 0, 0, 0.0f);
    const vec4 up = w->camera_normal_matrix * vec4(0, image_plane_width * aspect / height, 0, 0.0f);
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;

    const vec3 light = compute_light_dir(view.light_rotation);
    out->light_dir[0] = light.x; out->light_dir[1] = light.y; out->light_dir[2] = light.z;

    const material &mtl = materials[view.which_material % material_count];
    out->specular_color[0] = mtl.specular_color.x;
    out->specular_color[1] = mtl.specular_color.y;
    out->specular_color[2] = mtl.specular_color.z;
    const vec3 diffuse = mtl.metal ? vec3(0, 0, 0) : diffuse_colors[view.which_diffuse_color % diffuse_color_count];
    out->diffuse_color[0] = diffuse.x; out->diffuse_color[1] = diffuse.y; out->diffuse_color[2] = diffuse.z;
}
