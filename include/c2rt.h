/*
 * c2rt.h — plain-C boundary of the MI355X render hot path for Chess2RT.
 *
 * The reference (D) has no FFI of its own; the seam this
 * library plugs into is the body of the render-thread lambda in
 * `rt/renderer.d:36-37` (`Renderer(scene, output, ..).renderRT()`) and the
 * pixel probe `renderPixel` (`rt/renderer.d:46-57`).  The D side keeps its
 * `Scene` / `Camera` / `Renderer` API; it flattens the scene into the tables
 * below once per scene load (`c2rt_upload_scene`), calls
 * `scene.beginFrame()` itself (`rt/scene.d:55-58`, `rt/camera.d:77-117`) and
 * hands the six camera vectors over per frame (`c2rt_render_frame`).
 *
 * Conventions
 *   - every function returns a `c2rt_status`; nothing throws across the ABI;
 *   - all tables are COPIED at call time (the D GC may move/free its data);
 *   - geometry is fp64, colour is fp32, exactly as in the reference
 *     (`rt/imported_types.d:10-11`, `rt/color.d:27-35`);
 *   - matrices are the 9 doubles of gfm `mat3d` in row-major order `c[i][j]`,
 *     used as ROW-vector x matrix (`rt/imported_types.d:13-20`);
 *   - the frame is `Image!Color`: W*H*3 float32, row-major, no padding
 *     (`imageio/image.d:18-54`).
 *   - one context per process and GPU; calls on a context are serialised by
 *     the caller (the reference has a single render thread).
 *
 * Environment: the product library (libc2rt.so) reads NO environment variable.
 * The measurement / test knobs of earlier rounds (C2RT_EXACT, C2RT_NO_IDN,
 * C2RT_HOST_CHUNK_MB / _FIRST_FRAC / _COPY_STREAMS / _DIRECT_STORE,
 * C2RT_CSG_FIRST_CAP, C2RT_DEBUG_CULL — none changes a pixel) exist only in the
 * diagnostics build, chess2rt_amd/libc2rt_diag.so (`make`: c2rt_api.cpp with
 * -DC2RT_DIAG=1 over the same kernel objects), where each is read once per
 * process; chess2rt_amd/csrc/c2rt_api.cpp documents them at their sites.
 */
#ifndef C2RT_H
#define C2RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2RT_ABI_VERSION 1u

/* hard limits of the device path (upload fails with C2RT_ERR_LIMIT beyond) */
#define C2RT_MAX_CSG_DEPTH 4   /* nesting levels of CsgOp under a node       */
#define C2RT_MAX_CSG_GEOMS 4096 /* geometries in a scene that has CsgOps (hit lists tag leaves in 12 bits) */
#define C2RT_MAX_CSG_HITS 8    /* hits kept per CSG child per ray (the reference
                                  grows a MyArray, util/array.d:54-63; a sane
                                  primitive yields at most 2)                 */

typedef enum c2rt_status {
    C2RT_OK = 0,
    C2RT_ERR_INVALID_ARG = 1,   /* null pointer, bad index, bad enum          */
    C2RT_ERR_NO_DEVICE = 2,     /* no gfx950 GPU / HIP runtime failure        */
    C2RT_ERR_HIP = 3,           /* a HIP call failed; see c2rt_last_error     */
    C2RT_ERR_UNSUPPORTED = 4,   /* GIEnabled, unknown entity type, ...        */
    C2RT_ERR_LIMIT = 5,         /* CSG depth, sizes beyond the device limits  */
    C2RT_ERR_NO_SCENE = 6,      /* render before upload                       */
    C2RT_ERR_CANCELLED = 7,     /* stop flag was raised between passes        */
    C2RT_ERR_IO = 8,            /* host loader: file missing / unreadable     */
    C2RT_ERR_PARSE = 9          /* host loader: invalid scene / image file    */
} c2rt_status;

/* Closed type sets of the reference. */
typedef enum c2rt_geom_type {      /* rt/geometry.d:15,73,149,357,367,377 */
    C2RT_GEOM_PLANE = 0,
    C2RT_GEOM_SPHERE = 1,
    C2RT_GEOM_CUBE = 2,
    C2RT_GEOM_CSG_UNION = 3,
    C2RT_GEOM_CSG_INTER = 4,
    C2RT_GEOM_CSG_DIFF = 5
} c2rt_geom_type;

typedef enum c2rt_shader_type {    /* rt/shader.d:54,177 */
    C2RT_SHADER_LAMBERT = 0,
    C2RT_SHADER_PHONG = 1
} c2rt_shader_type;

typedef enum c2rt_texture_type {   /* rt/texture.d:20,70,103 */
    C2RT_TEX_CHECKER = 0,
    C2RT_TEX_PROCEDURE2 = 1,
    C2RT_TEX_BITMAP = 2
} c2rt_texture_type;

typedef enum c2rt_light_type {     /* rt/light.d:52 */
    C2RT_LIGHT_POINT = 0
} c2rt_light_type;

/*
 * Flat scene: struct-of-arrays tables, indices instead of references.
 * Replaces the object graph reached from `Scene` (rt/scene.d:39-53).
 */
typedef struct c2rt_scene_desc {
    uint32_t abi_version;          /* = C2RT_ABI_VERSION */

    /* ---- geometries (Scene.geometries, rt/geometry.d) ------------------ */
    uint32_t n_geoms;
    const int32_t *geom_type;      /* [n_geoms] c2rt_geom_type */
    /* [n_geoms][4] doubles:
     *   plane : y, limit (NaN = unbounded, rt/geometry.d:18-23), -, -
     *   sphere: center.x, center.y, center.z, R          (rt/geometry.d:75-79)
     *   cube  : center.x, center.y, center.z, side       (rt/geometry.d:151-152)
     *   csg   : unused */
    const double *geom_param;
    /* [n_geoms][2] left,right geometry index for CSG (rt/geometry.d:253-257),
     * -1 otherwise.  Children must have a smaller index than their parent
     * is NOT required; cycles are rejected. */
    const int32_t *geom_child;

    /* ---- textures (Scene.textures, rt/texture.d) ----------------------- */
    uint32_t n_textures;
    const int32_t *tex_type;       /* [n_textures] c2rt_texture_type */
    /* [n_textures][18] floats:
     *   checker   : color1.rgb, color2.rgb                 (rt/texture.d:22)
     *   procedure2: colorU[0..3].rgb, colorV[0..3].rgb     (rt/texture.d:72)
     *   bitmap    : unused */
    const float *tex_color;
    /* [n_textures][6] doubles:
     *   checker   : size                                   (rt/texture.d:23)
     *   procedure2: freqU[0..3], freqV[0..3]               (rt/texture.d:73)
     *   bitmap    : unused */
    const double *tex_param;
    const float *tex_scaling;      /* [n_textures] BitmapTexture.scaling (rt/texture.d:155) */
    const uint32_t *tex_width;     /* [n_textures] bitmap width  (0 if not a bitmap) */
    const uint32_t *tex_height;    /* [n_textures] bitmap height */
    const uint64_t *tex_offset;    /* [n_textures] first texel of this bitmap in `texels` */
    /* texel pool: linear-RGB float triples, row-major, y = 0 is the TOP row,
     * already gamma-decoded (rt/bitmap.d:116-136) — i.e. Bitmap.data.pixels. */
    uint64_t n_texels;
    const float *texels;           /* [n_texels][3] */

    /* ---- shaders (Scene.shaders, rt/shader.d) -------------------------- */
    uint32_t n_shaders;
    const int32_t *shader_type;    /* [n_shaders] c2rt_shader_type */
    const float *shader_color;     /* [n_shaders][3] Shader.color (rt/shader.d:26) */
    const int32_t *shader_texture; /* [n_shaders] texture index or -1 */
    const double *shader_exponent; /* [n_shaders] Phong.exponent (rt/shader.d:180) */
    const float *shader_strength;  /* [n_shaders] Phong.strength (rt/shader.d:181) */

    /* ---- lights (Scene.lights, rt/light.d) ----------------------------- */
    uint32_t n_lights;
    const int32_t *light_type;     /* [n_lights] c2rt_light_type */
    const double *light_pos;       /* [n_lights][3] PointLight.pos */
    const float *light_color;      /* [n_lights][3] Light.lightColor */
    const float *light_power;      /* [n_lights]    Light.lightPower */

    /* ---- nodes (Scene.nodes, rt/node.d:7-10, rt/transform.d:11-14) ----- */
    uint32_t n_nodes;
    const int32_t *node_geom;      /* [n_nodes] geometry index */
    const int32_t *node_shader;    /* [n_nodes] shader index */
    const int32_t *node_bump;      /* [n_nodes] texture index or -1 (base modifyNormal is a no-op) */
    /* [n_nodes][30] doubles: transform[9], inverseTransform[9],
     * transposedInverse[9], offset[3] */
    const double *node_transform;

    /* ---- the GlobalSettings fields the path reads (rt/global_settings.d) */
    float ambient[3];              /* ambientLightColor */
    uint32_t max_trace_depth;      /* maxTraceDepth (primary rays have depth 0) */
    uint32_t gi_enabled;           /* GIEnabled: must be 0 (C2RT_ERR_UNSUPPORTED otherwise) */
} c2rt_scene_desc;

/*
 * Per-frame camera state = what `Camera.beginFrame` leaves behind
 * (rt/camera.d:47-53,77-117) plus the fields `getScreenRay` reads
 * (rt/camera.d:123-173).
 */
typedef struct c2rt_camera_frame {
    double pos[3];
    double up_left[3], up_right[3], down_left[3];
    double right_dir[3], up_dir[3], front_dir[3];
    double frame_width, frame_height;   /* Camera.frameWidth/Height as doubles */
    /* depth of field (rt/camera.d:41-44,154-173); dof==0 for parity runs */
    uint32_t dof;
    uint32_t num_samples;               /* Camera.numSamples (default 25) */
    double focal_plane_dist;
    double disc_multiplier;             /* 10 / fNumber */
    double stereo_separation;           /* 0 = mono (rt/renderer.d:305) */
} c2rt_camera_frame;

/* Sampling modes (SURVEY.md F6, section 8(d) "spp mapping"). */
typedef enum c2rt_tap_mode {
    C2RT_TAPS_1 = 1,      /* AAEnabled=false: one sample at (x, y)            */
    C2RT_TAPS_REF5 = 5,   /* AAEnabled=true: reference 5-tap table, sum / 5   */
    C2RT_TAPS_4 = 4       /* build-defined "4 spp": taps 1..4 of the table / 4 */
} c2rt_tap_mode;

typedef struct c2rt_render_opts {
    uint32_t width, height;        /* settings.frameWidth / frameHeight = output size */
    uint32_t taps;                 /* c2rt_tap_mode */
    /* Interleaved row-strip sharding (multi-GPU): this call renders strips
     * s = strip_rank, strip_rank + strip_world, ... of height strip_height
     * into a compact buffer of those strips only.  strip_world <= 1 renders
     * the whole frame. */
    uint32_t strip_height;
    uint32_t strip_rank;
    uint32_t strip_world;
    uint64_t seed;                 /* counter-based RNG seed (DOF only) */
    uint32_t count_rays;           /* 1: also count primary/shadow rays cast */
    /* 0: the normal frame.  N > 0: `prepassOnly` (rt/renderer.d:110-130) with
     * bucketSize N: every 16x16 block of every NxN bucket is filled with the one
     * sample taken at its top-left pixel (with depth of field its jitter spans
     * the clipped block, rt/renderer.d:119,277); taps are ignored (the
     * reference returns before the AA pass). */
    uint32_t prepass_bucket;
} c2rt_render_opts;

/* What `renderPixel` returns (TraceResult, rt/renderer.d:15-21). */
typedef struct c2rt_trace_result {
    float color[3];
    int32_t closest_node;          /* -1: no hit */
    int32_t leaf_geom;             /* IntersectionData.g (leaf geometry index) */
    double p[3], normal[3];
    double dist, u, v;
    double ray_orig[3], ray_dir[3];
} c2rt_trace_result;

typedef struct c2rt_ray_stats {
    uint64_t primary_rays;
    uint64_t shadow_rays;
} c2rt_ray_stats;

typedef struct c2rt_ctx c2rt_ctx;

/* ---- lifecycle ---------------------------------------------------------- */

/* device < 0: use the current HIP device.  Fails with C2RT_ERR_NO_DEVICE when
 * no GPU is visible: there is NO CPU fallback in this library. */
int c2rt_init(int device, c2rt_ctx **out);

/* ONE context over several GPUs of this process — SURVEY.md 8(b)'s
 * `c2rt_init(int device_count_or_0, ...)`: what a single-process host like the
 * reference (one render thread fanning out, rt/renderer.d:23-44,133-142) needs
 * to use a whole node.  device_count_or_0 = 0 opens every visible GPU;
 * `device_ids` (nullable) names the HIP device of each slot — entries may
 * repeat, which runs several slots on one GPU (how a 1-GPU box exercises this
 * path).  Slot 0 is the lead device.  On such a context
 *   - c2rt_upload_scene replicates the tables on every device;
 *   - c2rt_render_frame / c2rt_render_frame_rgb32 deal interleaved 8-row strips
 *     to the devices (opts->strip_world must be <= 1) and every device copies
 *     its finished strips straight into the caller's host frame over its own
 *     PCIe link (strided 2-D copies; pin the buffer with c2rt_pin_host_buffer);
 *   - c2rt_render_frame_device: `out_rgb_dev` lives on the lead device and the
 *     other devices' kernels store their strips straight into it over xGMI
 *     (peer access) — no gather buffer, no de-interleave pass;
 *   - c2rt_render_pixel and the strip / encode helpers run on the lead device.
 * C2RT_ERR_NO_DEVICE when a device id is out of range; C2RT_ERR_UNSUPPORTED
 * from c2rt_render_frame_device when a device cannot peer-map the lead.
 * EXPERIMENTAL on more than one PHYSICAL device: the GPU pool this was built
 * on shows a job one GPU, so every test runs the slots on one device (ids
 * repeated); peer access, cross-device events and the parallel copies have
 * not executed across two GPUs yet.  A failing slot leaves the lead device
 * current and the caller's stream ordered after every slot already launched. */
int c2rt_init_multi(int device_count_or_0, const int *device_ids, c2rt_ctx **out);
/* number of device slots of the context (1 for c2rt_init) */
int c2rt_device_count(const c2rt_ctx *ctx);
void c2rt_destroy(c2rt_ctx *ctx);
const char *c2rt_last_error(const c2rt_ctx *ctx);
const char *c2rt_status_string(int status);
uint32_t c2rt_abi_version(void);

/* ---- scene -------------------------------------------------------------- */

/* Validates and copies the tables into HBM.  Replaces the implicit "scene is
 * GC memory shared with the render thread" of rt/renderer.d:39-40. */
int c2rt_upload_scene(c2rt_ctx *ctx, const c2rt_scene_desc *scene);
/* Identity of the tables the context holds now: a process-wide counter value
 * assigned by each successful c2rt_upload_scene (never reused, never 0; 0 = no
 * scene).  A host-side scene object remembers the value of ITS upload and
 * re-uploads when the context has moved on to another scene. */
uint64_t c2rt_scene_generation(const c2rt_ctx *ctx);

/* ---- rendering ---------------------------------------------------------- */

/* Number of rows this rank renders under `opts` striping (== opts->height
 * when strip_world <= 1). */
uint32_t c2rt_local_rows(const c2rt_render_opts *opts);

/* Blocking frame render into caller-owned HOST memory (`Image!Color.pixels`):
 * the drop-in for `Renderer.renderRT()` (rt/renderer.d:83-192).  Writes
 * local_rows*width*3 floats.  `stop_flag` (nullable) is polled like
 * `isStopReq()` (rt/renderer.d:93-97,129,147,180): before the frame and, for a
 * frame that goes back in row chunks (a page-locked float frame), between the
 * chunks.  A frame made by ONE launch — pageable destination, or display words
 * stored straight into a page-locked frame (c2rt_render_frame_rgb32) — polls it
 * once, before the launch; C2RT_ERR_CANCELLED leaves the frame unspecified. */
int c2rt_render_frame(c2rt_ctx *ctx, const c2rt_camera_frame *cam,
                      const c2rt_render_opts *opts, float *out_rgb,
                      const volatile uint8_t *stop_flag);

/* Optional: declare a long-lived host frame buffer (the GUI's `screen`,
 * gui/raytracer_demo.d:181-182) so that c2rt_render_frame can page-lock it once
 * and stream the frame back at PCIe rate while later rows are still rendering.
 * The caller must unpin before freeing the buffer.  Without it
 * c2rt_render_frame copies into pageable memory (slower, same result). */
int c2rt_pin_host_buffer(c2rt_ctx *ctx, float *out_rgb, size_t bytes);
int c2rt_unpin_host_buffer(c2rt_ctx *ctx, float *out_rgb);

/* Same, but the output stays in HBM: `out_rgb_dev` is a device pointer
 * (e.g. a torch tensor's data_ptr) and the kernels are enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream) without a host sync.
 * Frames enqueued on one stream run in order, as everything on a HIP stream
 * does.  Frames of ONE context on DIFFERENT streams are independent of each
 * other and may overlap: the per-frame scratch a launch needs (the tile-mask
 * table, the nested-CSG retry list) exists once per stream the context has
 * rendered on (16 slots; a 17th stream recycles the least recently used one
 * after a device sync), so the call neither waits for an earlier frame nor
 * leaves an event in the queue.  The one shared resource is the ray counters:
 * frames with opts->count_rays = 1 are ordered among themselves on the device.
 * The library keeps no reference to `hip_stream` (the handle is only compared):
 * the caller may destroy it as soon as the call has returned.  The caller owns
 * the usual hazards of its buffers (two frames into one `out_rgb_dev`). */
int c2rt_render_frame_device(c2rt_ctx *ctx, const c2rt_camera_frame *cam,
                             const c2rt_render_opts *opts, float *out_rgb_dev,
                             void *hip_stream);

/* Ray counters of the last render call made with opts->count_rays = 1
 * (waits, on the host, for that frame to complete). */
int c2rt_get_ray_stats(c2rt_ctx *ctx, c2rt_ray_stats *out);

/* CsgOp child hit lists that reached C2RT_MAX_CSG_HITS during the last render call made with
 * opts->count_rays = 1 (events, over primary and shadow rays; tiles redone by the hit-stack retry launch
 * count twice).  The reference's findAllIntersections loops `while (true)`
 * (rt/geometry.d:271-290) and does not terminate when the 1e-6 step is absorbed or a hit is NaN; this
 * build stops after C2RT_MAX_CSG_HITS hits per child (a primitive yields at most 2).  0 means the cap
 * did not take part in the frame; > 0 flags build-defined results (pathological coordinates, or a
 * nested child with more than 8 boundary crossings along one ray). */
int c2rt_get_csg_truncations(c2rt_ctx *ctx, uint64_t *out);

/* How many 8x8 tiles this context has rendered TWICE since it was created (synchronises the device).
 * The frame kernels evaluate fp64 divide / sqrt / normalise with shortened, correctly rounded sequences
 * that are valid for operands within about 1e+-70 of the scene's scale (chess2rt_amd/csrc/fp64_lean.h);
 * every use tests its operands, and a tile in which any lane met a zero numerator, an infinity, a NaN or
 * an operand beyond those windows discards its result and is rendered again with the compiler's IEEE
 * expansions — the same pixels as rounds 1-2 produced, at about twice the time for that tile.  A growing
 * number therefore costs speed, never correctness: e.g. a camera placed exactly on the plane of a cube face
 * (every ray's numerator for that face is 0).  Frames rendered with opts->count_rays take the IEEE path
 * only and do not count here. */
int c2rt_get_exact_redos(c2rt_ctx *ctx, uint64_t *out);

/* Pixel probe: mirrors `renderPixel` (rt/renderer.d:46-57): one sample at
 * integer (x, y), no AA, returns the colour and the trace result. */
int c2rt_render_pixel(c2rt_ctx *ctx, const c2rt_camera_frame *cam,
                      const c2rt_render_opts *opts, int x, int y,
                      c2rt_trace_result *out);

/* Rank-0 side of the multi-GPU gather: `gathered_dev` holds `world`
 * consecutive compact strip buffers (rank-major, as ncclGather leaves them);
 * writes the de-interleaved full frame to `frame_dev`.  Both device
 * pointers; enqueued on `hip_stream`. */
int c2rt_deinterleave_strips(c2rt_ctx *ctx, const float *gathered_dev,
                             float *frame_dev, uint32_t width, uint32_t height,
                             uint32_t strip_height, uint32_t world,
                             void *hip_stream);

/* Downstream display encode, fused on device (rt/color.d:154-162,194-228,
 * gui/sdl2_gui.d:139-155): float RGB -> 0x00RRGGBB via the reference's
 * 4097-entry sRGB table (including its 12.02 quirk). */
int c2rt_encode_rgb32(c2rt_ctx *ctx, const float *frame_dev, uint32_t *out_dev,
                      uint64_t n_pixels, void *hip_stream);

/* The frame in display format: render + the encode above on the device, then
 * 4 B/pixel (instead of 12) over PCIe into the caller's host buffer — what
 * SDL2Gui.draw (gui/sdl2_gui.d:139-155) computes per pixel on the CPU from the
 * float frame.  Same blocking / stop-flag behaviour as c2rt_render_frame. */
int c2rt_render_frame_rgb32(c2rt_ctx *ctx, const c2rt_camera_frame *cam,
                            const c2rt_render_opts *opts, uint32_t *out_rgb32,
                            const volatile uint8_t *stop_flag);

/* c2rt_deinterleave_strips for packed RGB32 strips (one 32-bit word per pixel). */
int c2rt_deinterleave_strips_rgb32(c2rt_ctx *ctx, const uint32_t *gathered_dev,
                                   uint32_t *frame_dev, uint32_t width, uint32_t height,
                                   uint32_t strip_height, uint32_t world, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* C2RT_H */
