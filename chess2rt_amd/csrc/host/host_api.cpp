// host_api.cpp — Renderer (the GPU drop-in for rt/renderer.d's Renderer /
// renderSceneAsync / renderPixel) and the C wrappers of include/c2rt_host.h.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../../include/c2rt_host.h"
#include "scene.hpp"

using namespace c2rt::host;

struct c2rt_host_scene {
    std::unique_ptr<Scene> scene;
    const c2rt_scene_desc *desc = nullptr; // cached flat view
    c2rt_ctx *uploaded_to = nullptr;       // context these tables were last uploaded to ...
    uint64_t uploaded_gen = 0;             // ... and the generation that upload got (c2rt_scene_generation)
    std::thread render_thread;
    int async_status = C2RT_OK;
};

namespace {

void set_err(char *err, size_t len, const std::string &msg)
{
    if (err && len) {
        std::snprintf(err, len, "%s", msg.c_str());
    }
}

c2rt_render_opts opts_of(const Scene &s)
{
    c2rt_render_opts o;
    std::memset(&o, 0, sizeof o);
    o.width = s.settings.frameWidth;
    o.height = s.settings.frameHeight;
    o.taps = s.settings.AAEnabled ? C2RT_TAPS_REF5 : C2RT_TAPS_1; // rt/renderer.d:147,183-186
    return o;
}

// struct Renderer — rt/renderer.d:59-192, with renderRT's body replaced by
// the C-ABI calls (upload once per scene, then one blocking frame).
struct Renderer {
    c2rt_ctx *ctx;
    c2rt_host_scene *hs;
    float *outputImage;
    volatile uint8_t *isRendering;
    const volatile uint8_t *isStopRequested;

    int ensureUploaded()
    {
        // A context holds ONE scene.  Skip the upload only if what the context holds now is this
        // scene's own upload: the generation is process-unique, so another scene uploaded to the same
        // context since (a second Renderer, a direct c2rt_upload_scene), or a new context allocated at
        // the address of a destroyed one, both show up as a different generation.
        if (hs->desc && hs->uploaded_to == ctx && hs->uploaded_gen != 0 && c2rt_scene_generation(ctx) == hs->uploaded_gen)
            return C2RT_OK;
        hs->desc = hs->scene->flatten();
        const int st = c2rt_upload_scene(ctx, hs->desc);
        hs->uploaded_to = st == C2RT_OK ? ctx : nullptr;
        hs->uploaded_gen = st == C2RT_OK ? c2rt_scene_generation(ctx) : 0;
        return st;
    }

    int renderRT(const c2rt_camera_frame &cam)
    {
        int st = ensureUploaded();
        const GlobalSettings &gs = hs->scene->settings;
        if (st == C2RT_OK && gs.prepassOnly && !gs.prepassEnabled) {
            // rt/renderer.d:110,129: no prepass and an immediate return — the frame is left untouched
        } else if (st == C2RT_OK) {
            // the prepass (rt/renderer.d:110-127) only paints a preview that pass 2
            // overwrites, unless prepassOnly stops there
            c2rt_render_opts o = opts_of(*hs->scene);
            if (gs.prepassOnly) o.prepass_bucket = gs.bucketSize;
            st = c2rt_render_frame(ctx, &cam, &o, outputImage, isStopRequested);
        }
        if (isRendering) *isRendering = 0; // end(): atomicStore(*isRendering, false) — rt/renderer.d:87-91
        return st;
    }
};

} // namespace

extern "C" {

int c2rt_host_scene_load(const char *path, c2rt_host_scene **out, char *err, size_t err_len)
{
    if (!path || !out) return C2RT_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        std::unique_ptr<Scene> s = parseSceneFromFile(path);
        c2rt_host_scene *h = new c2rt_host_scene();
        h->scene = std::move(s);
        *out = h;
        return C2RT_OK;
    } catch (const SceneError &e) {
        set_err(err, err_len, e.what());
        return e.status;
    } catch (const std::exception &e) {
        set_err(err, err_len, e.what());
        return C2RT_ERR_PARSE;
    }
}

void c2rt_host_scene_free(c2rt_host_scene *s)
{
    if (!s) return;
    if (s->render_thread.joinable()) s->render_thread.join();
    delete s;
}

const char *c2rt_host_scene_name(const c2rt_host_scene *s) { return s ? s->scene->name.c_str() : ""; }

const c2rt_scene_desc *c2rt_host_scene_desc(c2rt_host_scene *s)
{
    if (!s) return nullptr;
    if (!s->desc) s->desc = s->scene->flatten();
    return s->desc;
}

void c2rt_host_scene_get_settings(const c2rt_host_scene *s, c2rt_host_settings *o)
{
    if (!s || !o) return;
    const GlobalSettings &g = s->scene->settings;
    std::memset(o, 0, sizeof *o);
    o->frame_width = g.frameWidth;
    o->frame_height = g.frameHeight;
    o->fullscreen = g.fullscreen;
    o->allow_resize = g.allowResize;
    o->dynamic_aspect_ratio = g.dynamicAspectRatio;
    o->interactive = g.interactive;
    o->bucket_size = g.bucketSize;
    o->thread_count = g.threadCount;
    o->prepass_enabled = g.prepassEnabled;
    o->prepass_only = g.prepassOnly;
    o->gi_enabled = g.GIEnabled;
    o->aa_enabled = g.AAEnabled;
    o->aa_threshold = g.AAThreshold;
    o->paths_per_pixel = g.pathsPerPixel;
    o->max_trace_depth = g.maxTraceDepth;
    o->ambient[0] = g.ambientLightColor.r;
    o->ambient[1] = g.ambientLightColor.g;
    o->ambient[2] = g.ambientLightColor.b;
    o->debug_enabled = g.debugEnabled;
}

void c2rt_host_scene_get_camera(const c2rt_host_scene *s, c2rt_host_camera *o)
{
    if (!s || !o) return;
    const Camera &c = s->scene->camera;
    std::memset(o, 0, sizeof *o);
    o->frame_width = c.frameWidth;
    o->frame_height = c.frameHeight;
    o->aspect = c.aspect;
    o->pos[0] = c.pos.x; o->pos[1] = c.pos.y; o->pos[2] = c.pos.z;
    o->yaw = c.yaw; o->pitch = c.pitch; o->roll = c.roll; o->fov = c.fov;
    o->focal_plane_dist = c.focalPlaneDist;
    o->f_number = c.fNumber;
    o->disc_multiplier = c.discMultiplier;
    o->dof = c.dof;
    o->num_samples = c.numSamples;
    o->stereo_separation = c.stereoSeparation;
}

void c2rt_host_scene_set_camera(c2rt_host_scene *s, const c2rt_host_camera *i)
{
    if (!s || !i) return;
    Camera &c = s->scene->camera;
    c.frameWidth = i->frame_width;
    c.frameHeight = i->frame_height;
    c.aspect = i->aspect;
    c.pos = Vector(i->pos[0], i->pos[1], i->pos[2]);
    c.yaw = i->yaw; c.pitch = i->pitch; c.roll = i->roll; c.fov = i->fov;
    c.focalPlaneDist = i->focal_plane_dist;
    c.fNumber = i->f_number;
    c.discMultiplier = i->disc_multiplier;
    c.dof = i->dof != 0;
    c.numSamples = i->num_samples;
    c.stereoSeparation = i->stereo_separation;
}

void c2rt_host_scene_set_frame_size(c2rt_host_scene *s, uint32_t w, uint32_t h)
{
    if (!s) return;
    // RTDemo.updateToWindowSize — gui/raytracer_demo.d:126-143
    s->scene->settings.frameWidth = w;
    s->scene->settings.frameHeight = h;
    s->scene->camera.setFrameSize(w, h);
}

void c2rt_host_scene_set_aa(c2rt_host_scene *s, uint32_t aa) { if (s) s->scene->settings.AAEnabled = aa != 0; }
void c2rt_host_scene_set_dof(c2rt_host_scene *s, uint32_t dof) { if (s) s->scene->camera.dof = dof != 0; }

void c2rt_host_scene_begin_frame(c2rt_host_scene *s, c2rt_camera_frame *out)
{
    if (!s || !out) return;
    s->scene->beginFrame();
    s->scene->camera.fill(*out);
}

void c2rt_host_camera_move(c2rt_host_scene *s, double dx, double dy, double dz) { if (s) s->scene->camera.move(dx, dy, dz); }
void c2rt_host_camera_rotate(c2rt_host_scene *s, double dyaw, double droll, double dpitch) { if (s) s->scene->camera.rotate(dyaw, droll, dpitch); }

int c2rt_host_render_rt(c2rt_ctx *ctx, c2rt_host_scene *s, float *out_rgb, const volatile uint8_t *stop_flag)
{
    if (!ctx || !s || !out_rgb) return C2RT_ERR_INVALID_ARG;
    c2rt_camera_frame cam;
    c2rt_host_scene_begin_frame(s, &cam);
    Renderer r{ctx, s, out_rgb, nullptr, stop_flag};
    return r.renderRT(cam);
}

int c2rt_host_render_scene_async(c2rt_ctx *ctx, c2rt_host_scene *s, float *out_rgb, volatile uint8_t *is_rendering,
                                 const volatile uint8_t *needs_rendering)
{
    if (!ctx || !s || !out_rgb) return C2RT_ERR_INVALID_ARG;
    if (s->render_thread.joinable()) s->render_thread.join();
    // renderSceneAsync — rt/renderer.d:23-44: beginFrame on the caller's thread, then spawn
    c2rt_camera_frame cam;
    c2rt_host_scene_begin_frame(s, &cam);
    s->async_status = C2RT_OK;
    s->render_thread = std::thread([=]() {
        Renderer r{ctx, s, out_rgb, is_rendering, needs_rendering};
        s->async_status = r.renderRT(cam);
    });
    return C2RT_OK;
}

int c2rt_host_render_wait(c2rt_host_scene *s)
{
    if (!s) return C2RT_ERR_INVALID_ARG;
    if (s->render_thread.joinable()) s->render_thread.join();
    return s->async_status;
}

int c2rt_host_render_pixel(c2rt_ctx *ctx, c2rt_host_scene *s, int x, int y, c2rt_trace_result *out)
{
    if (!ctx || !s || !out) return C2RT_ERR_INVALID_ARG;
    // renderPixel — rt/renderer.d:46-57
    c2rt_camera_frame cam;
    c2rt_host_scene_begin_frame(s, &cam);
    Renderer r{ctx, s, nullptr, nullptr, nullptr};
    const int st = r.ensureUploaded();
    if (st != C2RT_OK) return st;
    const c2rt_render_opts o = opts_of(*s->scene);
    return c2rt_render_pixel(ctx, &cam, &o, x, y, out);
}

int c2rt_host_bmp_decode(const uint8_t *bytes, size_t len, uint32_t *width, uint32_t *height, float **out_rgb)
{
    if (!bytes || !width || !height || !out_rgb) return C2RT_ERR_INVALID_ARG;
    try {
        Bitmap b = loadBmpImage(bytes, len);
        float *p = (float *)std::malloc(b.pixels.size() * sizeof(float));
        if (!p) return C2RT_ERR_LIMIT;
        std::memcpy(p, b.pixels.data(), b.pixels.size() * sizeof(float));
        *width = b.width;
        *height = b.height;
        *out_rgb = p;
        return C2RT_OK;
    } catch (const SceneError &e) {
        return e.status;
    }
}

void c2rt_host_texture_gamma(float *texels, size_t n, float g) { if (texels) applyAssumedGamma(texels, n, g); }

int c2rt_host_bmp_encode(const float *rgb, uint32_t width, uint32_t height, uint8_t **out_bytes, size_t *out_len)
{
    if (!rgb || !out_bytes || !out_len) return C2RT_ERR_INVALID_ARG;
    const std::vector<uint8_t> f = saveBmp(rgb, width, height);
    uint8_t *p = (uint8_t *)std::malloc(f.size());
    if (!p) return C2RT_ERR_LIMIT;
    std::memcpy(p, f.data(), f.size());
    *out_bytes = p;
    *out_len = f.size();
    return C2RT_OK;
}

uint32_t c2rt_host_color_to_rgb32(const float rgb[3]) { return colorToRGB32(rgb); }

void c2rt_host_free(void *p) { std::free(p); }

// Transform (rt/transform.d:24-63) on the flat 30-double layout of c2rt_scene_desc::node_transform
namespace {
Transform transform_load(const double t[30])
{
    Transform x;
    std::memcpy(&x.transform, t, 9 * sizeof(double));
    std::memcpy(&x.inverseTransform, t + 9, 9 * sizeof(double));
    std::memcpy(&x.transposedInverse, t + 18, 9 * sizeof(double));
    x.offset = Vector(t[27], t[28], t[29]);
    return x;
}
void transform_store(double t[30], const Transform &x)
{
    std::memcpy(t, &x.transform, 9 * sizeof(double));
    std::memcpy(t + 9, &x.inverseTransform, 9 * sizeof(double));
    std::memcpy(t + 18, &x.transposedInverse, 9 * sizeof(double));
    t[27] = x.offset.x, t[28] = x.offset.y, t[29] = x.offset.z;
}
} // namespace
void c2rt_host_transform_reset(double t[30])
{
    Transform x;
    x.reset();
    transform_store(t, x);
}
void c2rt_host_transform_scale(double t[30], double x, double y, double z)
{
    Transform tr = transform_load(t);
    tr.scale(x, y, z);
    transform_store(t, tr);
}
void c2rt_host_transform_rotate(double t[30], double yaw, double pitch, double roll)
{
    Transform tr = transform_load(t);
    tr.rotate(yaw, pitch, roll);
    transform_store(t, tr);
}
void c2rt_host_transform_translate(double t[30], const double v[3])
{
    Transform tr = transform_load(t);
    tr.translate(Vector(v[0], v[1], v[2]));
    transform_store(t, tr);
}
void c2rt_host_transform_point(const double t[30], const double p[3], double out[3])
{
    const Transform tr = transform_load(t);
    const Vector r = mul(Vector(p[0], p[1], p[2]), tr.transform); // rt/transform.d:57-63
    out[0] = r.x + tr.offset.x, out[1] = r.y + tr.offset.y, out[2] = r.z + tr.offset.z;
}

} // extern "C"
