/*
 * c_abi_demo.c — the drop-in boundary used from plain C, with no Python and no
 * C++ in sight: build a scene by hand as flat tables, upload it, render a frame
 * into host memory and probe one pixel.  This is what a D `extern(C)` caller
 * does (integration/d/rt/gpu.d).
 *
 *   gcc -Iinclude examples/c_abi_demo.c -Lchess2rt_amd -lc2rt -Wl,-rpath,$PWD/chess2rt_amd -lm -o /tmp/c_abi_demo
 *   /tmp/c_abi_demo 320 240 out.ppm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c2rt.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int st_ = (call);                                                                             \
        if (st_ != C2RT_OK) {                                                                         \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, st_, c2rt_status_string(st_), ctx ? c2rt_last_error(ctx) : ""); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

/* The demo scene as flat tables (static storage) and the camera for a W x H frame. */
static void demo_scene(c2rt_scene_desc *sc, c2rt_camera_frame *cam, uint32_t W, uint32_t H)
{
    /* a checkered floor, a Phong sphere and a cube with a spherical bite (CsgDiff) */
    static const int32_t geom_type[] = {C2RT_GEOM_PLANE, C2RT_GEOM_SPHERE, C2RT_GEOM_CUBE, C2RT_GEOM_SPHERE, C2RT_GEOM_CSG_DIFF};
    static const double geom_param[] = {0, NAN, 0, 0, /**/ 30, 20, 120, 20, /**/ -40, 25, 100, 50, /**/ -25, 45, 80, 28, /**/ 0, 0, 0, 0};
    static const int32_t geom_child[] = {-1, -1, -1, -1, -1, -1, -1, -1, 2, 3};
    static const int32_t tex_type[] = {C2RT_TEX_CHECKER};
    static float tex_color[18] = {0.1f, 0.1f, 0.1f, 0.9f, 0.9f, 0.8f};
    static double tex_param[6] = {20};
    static const float tex_scaling[] = {1};
    static const uint32_t tex_wh[] = {0};
    static const uint64_t tex_off[] = {0};
    static const int32_t shader_type[] = {C2RT_SHADER_LAMBERT, C2RT_SHADER_PHONG, C2RT_SHADER_PHONG};
    static const float shader_color[] = {1, 1, 1, 0.1f, 0.2f, 0.8f, 0.8f, 0.6f, 0.1f};
    static const int32_t shader_texture[] = {0, -1, -1};
    static const double shader_exponent[] = {16, 60, 20};
    static const float shader_strength[] = {1, 1, 0.6f};
    static const int32_t light_type[] = {C2RT_LIGHT_POINT};
    static const double light_pos[] = {-80, 200, -20};
    static const float light_color[] = {1, 1, 1}, light_power[] = {60000};
    static const int32_t node_geom[] = {0, 1, 4}, node_shader[] = {0, 1, 2}, node_bump[] = {-1, -1, -1};
    static double node_transform[3 * 30];
    for (int n = 0; n < 3; ++n) {
        double *t = node_transform + 30 * n;
        memset(t, 0, 30 * sizeof(double));
        for (int m = 0; m < 3; ++m) t[9 * m + 0] = t[9 * m + 4] = t[9 * m + 8] = 1.0; /* transform = inverse = transposedInverse = I */
    }
    memset(sc, 0, sizeof *sc);
    sc->abi_version = C2RT_ABI_VERSION;
    sc->n_geoms = 5; sc->geom_type = geom_type; sc->geom_param = geom_param; sc->geom_child = geom_child;
    sc->n_textures = 1; sc->tex_type = tex_type; sc->tex_color = tex_color; sc->tex_param = tex_param; sc->tex_scaling = tex_scaling;
    sc->tex_width = tex_wh; sc->tex_height = tex_wh; sc->tex_offset = tex_off;
    sc->n_shaders = 3; sc->shader_type = shader_type; sc->shader_color = shader_color; sc->shader_texture = shader_texture;
    sc->shader_exponent = shader_exponent; sc->shader_strength = shader_strength;
    sc->n_lights = 1; sc->light_type = light_type; sc->light_pos = light_pos; sc->light_color = light_color; sc->light_power = light_power;
    sc->n_nodes = 3; sc->node_geom = node_geom; sc->node_shader = node_shader; sc->node_bump = node_bump; sc->node_transform = node_transform;
    sc->ambient[0] = sc->ambient[1] = sc->ambient[2] = 0.1f;
    sc->max_trace_depth = 4;
    /* what Camera.beginFrame leaves behind for pos (0,60,-60), pitch -20 deg, fov 80, no yaw/roll */
    memset(cam, 0, sizeof *cam);
    const double aspect = (double)W / H, s = tan(40.0 * M_PI / 180.0) / hypot(aspect, 1.0);
    const double cp = cos(-20.0 * M_PI / 180.0), sp = sin(-20.0 * M_PI / 180.0);
    const double pos[3] = {0, 60, -60};
    const double corners[3][2] = {{-aspect * s, s}, {aspect * s, s}, {-aspect * s, -s}}; /* upLeft, upRight, downLeft (x, y) at z = 1 */
    double *dst[3] = {cam->up_left, cam->up_right, cam->down_left};
    for (int k = 0; k < 3; ++k) { /* row vector (x, y, 1) times rotateX(pitch), plus pos */
        dst[k][0] = corners[k][0] + pos[0];
        dst[k][1] = corners[k][1] * cp + 1.0 * sp + pos[1];
        dst[k][2] = corners[k][1] * -sp + 1.0 * cp + pos[2];
    }
    memcpy(cam->pos, pos, sizeof pos);
    cam->right_dir[0] = 1; cam->up_dir[1] = cp; cam->up_dir[2] = -sp; cam->front_dir[1] = sp; cam->front_dir[2] = cp;
    cam->frame_width = W; cam->frame_height = H; cam->num_samples = 25; cam->focal_plane_dist = 1; cam->disc_multiplier = 10;

}

int main(int argc, char **argv)
{
    const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 320, H = argc > 2 ? (uint32_t)atoi(argv[2]) : 240;
    c2rt_ctx *ctx = NULL;
    CHECK(c2rt_init(-1, &ctx));

    c2rt_scene_desc sc;
    c2rt_camera_frame cam;
    demo_scene(&sc, &cam, W, H);
    CHECK(c2rt_upload_scene(ctx, &sc));

    c2rt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.width = W; opts.height = H; opts.taps = C2RT_TAPS_REF5; opts.count_rays = 1;
    float *frame = (float *)malloc((size_t)W * H * 3 * sizeof(float));
    CHECK(c2rt_render_frame(ctx, &cam, &opts, frame, NULL));
    c2rt_ray_stats rs;
    CHECK(c2rt_get_ray_stats(ctx, &rs));
    c2rt_trace_result tr;
    CHECK(c2rt_render_pixel(ctx, &cam, &opts, (int)W / 2, (int)(H * 2 / 3), &tr));
    double mean = 0;
    for (size_t i = 0; i < (size_t)W * H * 3; ++i) mean += frame[i];
    mean /= (double)W * H * 3;
    printf("frame %ux%u: mean %.6f, %llu primary + %llu shadow rays; probe(%u,%u): node %d leaf %d dist %.6f rgb %.4f %.4f %.4f\n", W, H, mean,
           (unsigned long long)rs.primary_rays, (unsigned long long)rs.shadow_rays, W / 2, H * 2 / 3, tr.closest_node, tr.leaf_geom, tr.dist,
           tr.color[0], tr.color[1], tr.color[2]);
    if (argc > 3) { /* plain PPM dump, clamped, no gamma */
        FILE *f = fopen(argv[3], "wb");
        fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (size_t i = 0; i < (size_t)W * H * 3; ++i) fputc((int)(fminf(fmaxf(frame[i], 0.f), 1.f) * 255.f), f);
        fclose(f);
    }
    free(frame);
    c2rt_destroy(ctx);
    return !(mean > 0.01 && rs.primary_rays == (unsigned long long)W * H * 5);
}
