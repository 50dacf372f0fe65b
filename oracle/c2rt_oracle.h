/*
 * c2rt_oracle.h — CPU restatement of Chess2RT's render hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path (chess2rt_amd/) never links, imports or calls it.
 *
 * It follows the reference (D, /root/reference/source/rt) function by
 * function in the reference's operation order; each function cites the
 * file:line it restates.  Build: gcc -O2 -ffp-contract=off (no fast-math).
 *
 * PARITY PINNING: the reference cannot be compiled in
 * the build environment (D, no compiler) and has no tests for rt/.  The
 * only reference-held known-answer vectors on this path are the two BMP
 * decode unittests (imageio/bmp.d:446-611) — the oracle is pinned against
 * those (tests/test_oracle_golden.py) — plus the surveyor's hand-derived
 * lecture4 anchors.  Everything behind gfm:math 7.0.8 (un-vendored) is
 * "parity unpinned": restated from gfm's published algorithm.
 */
#ifndef C2RT_ORACLE_H
#define C2RT_ORACLE_H

#include "../include/c2rt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host-side, once per frame / scene (cold) --------------------------- */

/* Camera.beginFrame (rt/camera.d:77-117) + setFrameSize (rt/camera.d:231-236).
 * Fills pos, the six vectors and frame_width/height of `out`; leaves the DOF
 * fields untouched. */
void orc_camera_begin_frame(const double pos[3], double yaw, double pitch,
                            double roll, double fov, uint32_t frame_width,
                            uint32_t frame_height, c2rt_camera_frame *out);

/* Transform algebra (rt/transform.d:24-55).  `t` is the 30-double record of
 * c2rt_scene_desc.node_transform. */
void orc_transform_reset(double t[30]);
void orc_transform_scale(double t[30], double x, double y, double z);
void orc_transform_rotate(double t[30], double yaw, double pitch, double roll);
void orc_transform_translate(double t[30], const double v[3]);

/* BMP decode to Color (imageio/bmp.d:60-193, rt/color.d:60-66): returns
 * malloc'd width*height*3 floats, y = 0 is the top row; NULL on error.
 * `raw_rgb32` (nullable, width*height uint32) receives the 0xAARRGGBB words
 * the reference's loadBmp!uint would produce (what its unittests assert). */
float *orc_bmp_decode(const uint8_t *bytes, size_t len, uint32_t *width,
                      uint32_t *height, uint32_t **raw_rgb32);
/* Bitmap.decompressGamma_sRGB / decompressGamma (rt/bitmap.d:116-136),
 * dispatched as BitmapTexture.deserialize does (rt/texture.d:137-141). */
void orc_texture_gamma(float *texels, size_t n_floats, float assumed_gamma);
void orc_free(void *p);

/* ---- the hot path ------------------------------------------------------- */

/* Renderer.renderRT passes 2 and 3b (rt/renderer.d:133-142,183-186) over the
 * 48x48 serpentine bucket list (rt/renderer.d:194-213) on `n_threads`
 * pthreads (0 = all online cores).  Output layout as c2rt_render_frame
 * (striping honoured).  Returns a c2rt_status. */
int orc_render_frame(const c2rt_scene_desc *scene, const c2rt_camera_frame *cam,
                     const c2rt_render_opts *opts, float *out_rgb,
                     uint32_t n_threads, c2rt_ray_stats *stats);

/* renderPixel (rt/renderer.d:46-57). */
int orc_render_pixel(const c2rt_scene_desc *scene, const c2rt_camera_frame *cam,
                     const c2rt_render_opts *opts, int x, int y,
                     c2rt_trace_result *out);

/* ---- unit-level entry points (per-function vectors) -------- */

typedef struct orc_hit {           /* IntersectionData, rt/intersectable.d:6-33 */
    double p[3], normal[3];
    double dist, u, v;
    int32_t g;                     /* leaf geometry index, -1 = null */
    double dNdx[3], dNdy[3];
} orc_hit;

/* Geometry.intersect (rt/geometry.d): `hit->dist` is in/out. Returns 0/1. */
int orc_geom_intersect(const c2rt_scene_desc *scene, int32_t geom,
                       const double orig[3], const double dir[3], orc_hit *hit);
/* Geometry.isInside */
int orc_geom_is_inside(const c2rt_scene_desc *scene, int32_t geom, const double p[3]);
/* Node.intersect (rt/node.d:23-49) */
int orc_node_intersect(const c2rt_scene_desc *scene, int32_t node,
                       const double orig[3], const double dir[3], orc_hit *hit);
/* Texture.getTexColor (rt/texture.d:36-54,77-86,116-126) */
void orc_tex_color(const c2rt_scene_desc *scene, int32_t tex, double u, double v,
                   float out_rgb[3]);
/* Camera.getScreenRay, non-DOF (rt/camera.d:123-154) */
void orc_screen_ray(const c2rt_camera_frame *cam, double x, double y,
                    double orig[3], double dir[3]);
/* Scene.testVisibility (rt/scene.d:62-78) */
int orc_test_visibility(const c2rt_scene_desc *scene, const double from[3],
                        const double to[3]);
/* util.array.sort on IntersectionData by dist (util/array.d:95-111) */
void orc_shell_sort_hits(orc_hit *arr, size_t n);

/* Color.toRGB32 through the cached sRGB table (rt/color.d:154-162,194-228) */
uint32_t orc_color_to_rgb32(const float rgb[3]);

/* Build-defined counter-based RNG used only for depth-of-field runs (the
 * reference uses libc rand()): uniform in [0,1). */
double orc_rng_uniform(uint64_t seed, uint64_t pixel, uint32_t tap,
                       uint32_t sample, uint32_t dim);
/* (sin, cos)(2 pi u) for u in [0,1) of the lens sample (rt/camera.d:258-269),
 * libm-free so that the device produces the same bits. */
void orc_lens_sincos2pi(double u, double *sn, double *cs);

/* findAllIntersections' build-defined cap (default C2RT_MAX_CSG_HITS = 8 hits per CsgOp child, clamped to
 * 1..64) and the number of child hit lists that reached it since the last take.  Tests render with the cap
 * at 64 to show it takes no part in a frame. */
void orc_set_csg_hit_cap(unsigned cap);
unsigned long long orc_take_csg_truncations(void);

/* Algorithmic floating-point operation counts of the render path as the reference's source
 * executes it; only the library built with -DORC_COUNT_OPS
 * (oracle/libc2rt_oracle_count.so) tallies.  orc_op_counts_take copies the totals accumulated by
 * orc_render_frame / orc_render_pixel calls since the last take and clears them; returns 1 if this
 * build counts, 0 otherwise (all zeros). */
typedef struct orc_op_counts {
    uint64_t dadd, dmul, ddiv, dsqrt, dlibm; /* fp64: add/sub, mul, div, sqrt, libm calls */
    uint64_t fadd, fmul, fdiv;               /* fp32 colour arithmetic */
} orc_op_counts;
int orc_op_counts_take(orc_op_counts *out);

#ifdef __cplusplus
}
#endif
#endif
