/*
 * c2rt_host.h — C view of the host-side mirror of the reference's
 * Scene / Camera / Renderer API (C++ implementation in
 * chess2rt_amd/csrc/host/).  The reference is D and cannot be compiled in
 * the build environment, so the host side that would normally stay in D is
 * restated in C++ above the C ABI of c2rt.h; these wrappers let the Python
 * test driver and bench.py reach it through ctypes.
 *
 * Mirrors (paths relative to /root/reference/source):
 *   parseSceneFromFile   rt/scene_loader.d:20-41
 *   Scene.beginFrame     rt/scene.d:55-58 -> Camera.beginFrame rt/camera.d:77-117
 *   updateToWindowSize   gui/raytracer_demo.d:126-143 (frame-size override)
 *   Renderer.renderRT    rt/renderer.d:83-192
 *   renderSceneAsync     rt/renderer.d:23-44
 *   renderPixel          rt/renderer.d:46-57
 *   Camera.move/rotate   rt/camera.d:181-229
 *   Transform            rt/transform.d:24-63
 *   Bitmap.loadImage     rt/bitmap.d:67-80, imageio/bmp.d:60-193
 *   Bitmap.saveImage     rt/bitmap.d:84-103, imageio/bmp.d:195-237
 */
#ifndef C2RT_HOST_H
#define C2RT_HOST_H

#include "c2rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c2rt_host_scene c2rt_host_scene;

/* GlobalSettings as parsed (rt/global_settings.d:5-32). */
typedef struct c2rt_host_settings {
    uint32_t frame_width, frame_height;
    uint32_t fullscreen, allow_resize, dynamic_aspect_ratio, interactive;
    uint32_t bucket_size, thread_count;
    uint32_t prepass_enabled, prepass_only, gi_enabled, aa_enabled;
    double aa_threshold;
    uint32_t paths_per_pixel, max_trace_depth;
    float ambient[3];
    uint32_t debug_enabled;
} c2rt_host_settings;

/* Camera fields (rt/camera.d:29-45). */
typedef struct c2rt_host_camera {
    uint64_t frame_width, frame_height;
    double aspect;
    double pos[3];
    double yaw, pitch, roll, fov;
    double focal_plane_dist, f_number, disc_multiplier;
    uint32_t dof;
    uint64_t num_samples;
    double stereo_separation;
} c2rt_host_camera;

/* parseSceneFromFile.  On failure returns C2RT_ERR_IO (SceneNotFoundException)
 * or C2RT_ERR_PARSE (InvalidSceneException, EntityWithDuplicateName, ...) and
 * writes a message to `err` (nullable). */
int c2rt_host_scene_load(const char *path, c2rt_host_scene **out, char *err, size_t err_len);
void c2rt_host_scene_free(c2rt_host_scene *scene);
const char *c2rt_host_scene_name(const c2rt_host_scene *scene);

/* The flat tables for c2rt_upload_scene; owned by the scene, valid until it
 * is freed or mutated. */
const c2rt_scene_desc *c2rt_host_scene_desc(c2rt_host_scene *scene);

void c2rt_host_scene_get_settings(const c2rt_host_scene *scene, c2rt_host_settings *out);
void c2rt_host_scene_get_camera(const c2rt_host_scene *scene, c2rt_host_camera *out);
void c2rt_host_scene_set_camera(c2rt_host_scene *scene, const c2rt_host_camera *in);

/* settings.frameWidth/Height := w,h and camera.setFrameSize(w,h), as
 * RTDemo.updateToWindowSize does with dynamicAspectRatio. */
void c2rt_host_scene_set_frame_size(c2rt_host_scene *scene, uint32_t width, uint32_t height);
void c2rt_host_scene_set_aa(c2rt_host_scene *scene, uint32_t aa_enabled);
void c2rt_host_scene_set_dof(c2rt_host_scene *scene, uint32_t dof);

/* Scene.beginFrame: computes the camera frame for the current settings. */
void c2rt_host_scene_begin_frame(c2rt_host_scene *scene, c2rt_camera_frame *out);
/* Camera.move / Camera.rotate (need a beginFrame before move). */
void c2rt_host_camera_move(c2rt_host_scene *scene, double dx, double dy, double dz);
void c2rt_host_camera_rotate(c2rt_host_scene *scene, double dyaw, double droll, double dpitch);

/* Renderer.renderRT through the GPU library: uploads the scene if `ctx` has
 * none from this scene yet, beginFrame, taps from AAEnabled, blocking.
 * `out_rgb` is frameWidth*frameHeight*3 floats (Image!Color). */
int c2rt_host_render_rt(c2rt_ctx *ctx, c2rt_host_scene *scene, float *out_rgb,
                        const volatile uint8_t *stop_flag);
/* renderSceneAsync: beginFrame on the caller's thread, then one render
 * thread; `*is_rendering` is cleared when the frame is complete,
 * `needs_rendering` is the stop request. */
int c2rt_host_render_scene_async(c2rt_ctx *ctx, c2rt_host_scene *scene, float *out_rgb,
                                 volatile uint8_t *is_rendering,
                                 const volatile uint8_t *needs_rendering);
/* joins the render thread of the last async call (test helper) */
int c2rt_host_render_wait(c2rt_host_scene *scene);
/* renderPixel */
int c2rt_host_render_pixel(c2rt_ctx *ctx, c2rt_host_scene *scene, int x, int y,
                           c2rt_trace_result *out);

/* loadBmpImage!Color: malloc'd width*height*3 floats (free with
 * c2rt_host_free); y = 0 is the top row.  No gamma decode. */
int c2rt_host_bmp_decode(const uint8_t *bytes, size_t len, uint32_t *width, uint32_t *height,
                         float **out_rgb);
/* BitmapTexture gamma step (rt/texture.d:137-141, rt/bitmap.d:116-136). */
void c2rt_host_texture_gamma(float *texels, size_t n_floats, float assumed_gamma);
/* Color.toRGB32 + saveBmp 24-bpp: malloc'd file image. */
int c2rt_host_bmp_encode(const float *rgb, uint32_t width, uint32_t height, uint8_t **out_bytes,
                         size_t *out_len);
uint32_t c2rt_host_color_to_rgb32(const float rgb[3]);
void c2rt_host_free(void *p);

/* Transform (rt/transform.d:24-63) as the loader's mirror evaluates it.  `t` holds transform[9],
 * inverseTransform[9], transposedInverse[9] (row-major) and offset[3] — the layout of
 * c2rt_scene_desc::node_transform.  The REAL rotate (Rx(pitch) * Ry(yaw) * Rz(roll), :41-50) is reachable only
 * here: the scene loader's "rotate" key scales (reference bug). */
void c2rt_host_transform_reset(double t[30]);
void c2rt_host_transform_scale(double t[30], double x, double y, double z);
void c2rt_host_transform_rotate(double t[30], double yaw, double pitch, double roll);
void c2rt_host_transform_translate(double t[30], const double v[3]);
/* point(P) = mul(P, transform) + offset (:57-63) */
void c2rt_host_transform_point(const double t[30], const double p[3], double out[3]);

#ifdef __cplusplus
}
#endif
#endif
