/*
 * c2rt_device.h — device-side table layout and launch interface shared by
 * c2rt_kernels.hip (kernels) and c2rt_api.cpp (C-ABI implementation).
 *
 * The public SoA tables of c2rt_scene_desc are repacked at upload into small
 * 16-byte-aligned records: every lane of a wavefront walks the SAME node /
 * geometry / light at the same time, so the records are fetched with scalar
 * (SMEM) loads into SGPRs — one s_load per record, no VGPRs, no LDS — and only
 * texels and the frame go through the vector memory path.
 */
#ifndef C2RT_DEVICE_H
#define C2RT_DEVICE_H

#include <stdint.h>

#include "../../include/c2rt.h"

namespace c2rt {

constexpr int kMaxCsgHits = C2RT_MAX_CSG_HITS; /* per CSG child per ray */
constexpr int kCsgEntries = 2 * kMaxCsgHits;
#ifndef C2RT_TILE_W
#define C2RT_TILE_W 8
#endif
constexpr int kTileW = C2RT_TILE_W, kTileH = 64 / C2RT_TILE_W; /* one wavefront = one 8x8 pixel tile */
constexpr int kWave = 64;
constexpr int kMaxCullNodes = 32;  /* nodes beyond this are always tested */
constexpr int kHullEdges = 6;      /* a projected box is at most a hexagon */
#ifndef C2RT_MAX_CULL_LIGHTS
#define C2RT_MAX_CULL_LIGHTS 4
#endif
constexpr int kMaxCullLights = C2RT_MAX_CULL_LIGHTS; /* lights beyond this get no shadow-ray culling */
#ifndef C2RT_WAVES_PER_BLOCK
#define C2RT_WAVES_PER_BLOCK 1
#endif
constexpr int kWavesPerBlock = C2RT_WAVES_PER_BLOCK; /* horizontally adjacent tiles per workgroup */
constexpr int kBlockThreads = kWave * kWavesPerBlock;
/* LDS bytes per entry of a wavefront's CSG hit stack: dist[64] (8 B) + tag[64] (2 B) */
constexpr int kCsgLdsPerEntry = kWave * (8 + 2);
/* a CsgOp's list holds at most 8 + 8 entries and lists nest: depth x 16 entries can never overflow */
constexpr int kCsgFullCap(int levels) { return kCsgEntries * (levels > 0 ? levels : 1); }
/* first-pass capacity per nesting depth (10 / 10 / 12.5 / 12.5 KiB per wave): what non-pathological trees
 * need (two hits per primitive child = 4 entries per level) with a margin, and no more than lets THREE
 * workgroups of four waves share a CU's 160 KiB (the nested instances run three waves per SIMD, c2rt_kernels.hip:
 * occ_of); tiles that overflow are rendered again at kCsgFullCap */
constexpr int kCsgFirstCap(int levels) { return levels <= 2 ? 16 : 20; }
static_assert(3 * kWavesPerBlock * kCsgFirstCap(C2RT_MAX_CSG_DEPTH) * kCsgLdsPerEntry <= 160 * 1024, "three workgroups per CU");

enum GeomFlags : int32_t {
    kGeomBounded = 1,              /* `bound` is valid: a ray that misses it cannot hit */
    /* CsgOp shortcuts that are exact under the reference's leaf-identity walk
     * (`current.g is left`, rt/geometry.d:314-317):
     * A: no hit on the left child  => intersect() is false (Inter/Diff, and no
     *    leaf equal to `left` inside the right subtree, which would toggle inL);
     * B: no hit on the right child => false (Inter with a PRIMITIVE left child;
     *    a nested left child's hits carry a leaf != left and toggle inR). */
    kCsgShortA = 2,
    kCsgShortB = 4,
    /* every parameter and table entry (p[], q[]) of a primitive is finite: the premise of the exact early-outs that
     * reason about ordered comparisons (c2rt_trace.inc: cube_away) */
    kGeomFinite = 8,
};

struct alignas(16) DevGeom {       /* 128 B */
    int32_t type, left, right;
    int32_t flags;                 /* GeomFlags */
    double p[4];                   /* plane: y, limit | sphere: c, R | cube: c, side */
    /* Conservative bounding sphere in object space: centre, radius^2, padded
     * by 1e-6 relative so that rounding in the reject test can only keep rays,
     * never drop one the reference would hit.  Built at upload
     * (c2rt_api.cpp: cube = half diagonal, Union = both children, Inter/Diff =
     * left child, Plane = unbounded). */
    double bound[4];
    /* Wave-uniform subexpressions of the intersection tests, evaluated once at upload with the same
     * IEEE operations the reference evaluates per ray (so the bits are the same) instead of per lane
     * on the VALU:  cube: q[0..2] = center - side*0.5, q[3..5] = center + side*0.5 (the face planes
     * `center.y + side * halfSide`, side = -1 / +1, and the bounds `center.x -+ halfSide`,
     * rt/geometry.d:208-221);  sphere: q[0] = R * R (rt/geometry.d:99,129). */
    double q[6];
};

enum NodeFlags : uint32_t {
    kNodeIdentityMatrix = 1u,      /* transform == inverse == I: skip the 3x3 products (exact) */
    kNodeZeroOffset = 2u,          /* offset == 0: skip the subtraction/addition */
    /* a Plane whose inverse matrix is the identity, or diagonal with finite entries of magnitude in
     * [1e-100, 1e100] and a positive y entry: the sign of the node-space direction's y follows from
     * the world-space direction's (plane_points_away, c2rt_kernels.hip) */
    kNodeAxisPlane = 4u,
    /* a Plane under a non-identity matrix: DevNode::g.q[0..2] (unused by planes) hold the world normal
     * normalize((0,1,0) * transposedInverse) — rt/node.d:41-43 — the same for every hit on the node, evaluated at
     * upload with the reference's IEEE operations (finite results only) instead of per lane and per sample */
    kNodePlaneNormal = 8u,
};

/* What shading a hit on a node reads, flattened at upload from
 * shaders[node.shader] and textures[shader.tex]: one scalar load per distinct
 * closest node of a wave instead of a node -> shader -> texture chain of
 * per-lane loads (each link a full memory round trip). */
struct alignas(16) DevMat {        /* 80 B */
    int32_t shader_type;
    int32_t tex_type;              /* -1: untextured */
    int32_t tex;                   /* texture index (Procedure2 reads its tables from there) */
    float strength;
    float color[3];
    uint32_t pad;
    double exponent;
    uint32_t pad2[2];
    /* checker: color1.rgb, color2.rgb, size (double)  |  bitmap: width, height, scaling, -, offset (u64) */
    uint32_t texdata[8];
};

struct alignas(16) DevNode {       /* 464 B; the first 160 B are all a primitive under an identity matrix needs */
    int32_t geom, shader;
    uint32_t flags, pad;
    double off[3];
    double pad2;
    DevGeom g;                     /* copy of geoms[geom]: one scalar load instead of a dependent pair */
    double inv[9];                 /* inverseTransform */
    double m[9];                   /* transform */
    double tinv[9];                /* transposedInverse */
    double pad3;
    DevMat mat;
};

struct alignas(16) DevShader {     /* 32 B */
    int32_t type, tex;
    float color[3];
    float strength;
    double exponent;
};

struct alignas(16) DevTex {        /* 144 B */
    int32_t type;
    uint32_t width, height;
    float scaling;
    uint64_t offset;               /* first texel of the bitmap in the float4 pool */
    float color[18];
    double param[6];
};

struct alignas(16) DevLight {      /* 48 B */
    double pos[3];
    float color[3];                /* lightColor * lightPower (rt/light.d:11-14) */
    uint32_t lit;                  /* bit 0: intensity(color) != 0; bit 1: every channel +0 or within 2^+-60 (lean fp32 division) */
    uint32_t pad[2];
};

/* Kernel argument block (passed by value: lives in the kernarg segment and is
 * read with scalar loads). */
struct RenderParams {
    const DevGeom *geoms;
    const DevNode *nodes;
    const DevShader *shaders;
    const DevTex *textures;
    const DevLight *lights;
    const float *texels;           /* float4 per texel (rgb, pad): one dwordx4 gather per tap */
    uint32_t n_nodes, n_lights;
    float ambient[3];
    uint32_t max_trace_depth;

    c2rt_camera_frame cam;
    double cam_du[3], cam_dv[3];   /* up_right - up_left, down_left - up_left (rt/camera.d:140-142) */
    double cam_rw, cam_rh;         /* 1.0 / frame_width, 1.0 / frame_height, IEEE (fp64_lean.h: div_with through the rounded reciprocal) */
    uint32_t force_exact;          /* 1: every tile through exact:: (c2rt_kernels.hip, render_one) */

    uint32_t width, height;        /* frame */
    uint32_t taps;
    uint32_t prepass_bucket;       /* > 0: prepassOnly preview with this bucket size */
    uint32_t strip_height, strip_rank, strip_world;
    uint32_t local_rows;           /* rows this launch renders */
    uint32_t row_offset;           /* first local row of this launch (chunked host-output renders) */
    /* 0: `out` is this rank's compact strip buffer, indexed by LOCAL row.  1: `out` is the whole
     * frame, indexed by FRAME row — the devices of a multi-device context store their strips straight
     * into the lead device's frame over xGMI (peer-mapped): no gather buffer, no de-interleave pass. */
    uint32_t frame_rows;
    /* Per-frame screen-space culling of PRIMARY rays (host-computed, c2rt_api.cpp):
     * the pixel rectangle [x0, x1) x [y0, y1) outside of which no ray through a
     * sample of this frame can reach node n's padded bounding box.  n_cull = 0
     * disables it (depth of field, stereo, prepass). */
    uint32_t n_cull;
    int32_t cull_rect[kMaxCullNodes][4];
    /* The same, tighter: the convex hull of the 8 projected box corners (the silhouette of a box seen
     * from the eye has at most 6 vertices) as up to kHullEdges half planes a*x + b*y + c >= 0 in pixel
     * coordinates ((a, b) of unit length, pushed out by 2.5 px); a tile whose four corners all lie outside
     * ONE of them cannot contain a sample whose ray reaches the box.  A cube 250 units away under a
     * 90-degree lens: rectangle 27.6 % of the frame, hull 17 %.  Unused edges are (0, 0, 1). */
    float cull_hull[kMaxCullNodes][kHullEdges][3];
    /* SHADOW rays: all hit points of a tile lie inside the tile's view pyramid; a
     * node whose box is entirely beyond one side plane of that pyramid while the
     * light is on the inner side of the same plane cannot occlude any of the
     * tile's shadow rays.  Per light, the tile-boundary coordinates for which the
     * light is certainly on the ">= x" / "<= x" side of the vertical boundary
     * plane x (same for y), as closed integer intervals (empty when lo > hi):
     * [0..1] ">= x" lo,hi  [2..3] "<= x" lo,hi  [4..5] ">= y"  [6..7] "<= y". */
    uint32_t n_cull_lights;
    int32_t light_side[kMaxCullLights][8];
    /* every node is an "axis plane" (kNodeAxisPlane): the launcher picks the kernel instance that
     * decides plane misses from the un-normalised ray direction (plane_points_away) */
    uint32_t planes_only;
    /* every node's matrix is the identity (translations allowed): the launcher picks the kSpecIdentity instances
     * (c2rt_trace.inc) for uncounted frames with at most one light and no stereo */
    uint32_t all_identity;
    /* "Ground plane" shadow culling (c2rt_api.cpp: ground_shadow_rects).  ground_node >= 0: node
     * ground_node is a Plane under an identity matrix with zero offset, at height ground_y.  In a
     * tile whose primary rays can reach that node only, every hit point lies on the plane inside the
     * tile's footprint there, and a shadow ray towards light 0 can only meet node n if the footprint
     * meets shadow_rects[n] = {x0, x1, z0, z1}: the padded box of node n projected from the light onto
     * the plane (+-inf when that projection is not defined). */
    int32_t ground_node;
    double ground_y;
    const double *shadow_rects;    /* [kMaxCullNodes][4], device memory (scene constant) */
    uint32_t tiles_x, tiles_y;     /* tile grid over the LOCAL rows */
    uint32_t blocks_x;             /* ceil(tiles_x / kWavesPerBlock) */
    uint32_t row_group_start;      /* first group of 8 tile rows to dispatch (< ceil(tiles_y / 8)) */
    uint64_t seed;
    /* CSG hit stack (c2rt_kernels.hip, csg_intersect): entries per wave of this launch, and the list
     * of tiles whose stacks overflowed — [0] = count, then block indices; retry_mode = 1: this launch
     * renders the listed tiles (at full capacity) instead of the whole grid */
    uint32_t csg_cap;
    uint32_t retry_mode, retry_max;
    uint32_t *retry_list;
    /* per tile, 4 words: {primary mask, shadow mask of light 0, ground bits, 0}; frames with several culled lights
     * carry a second table of the same layout behind it, mask_entries entries further on: {shadow masks of lights
     * 1..3, 0} (c2rt_trace.inc: light_shadow_mask) — written by the pre-pass kernel
     * (launch_tile_masks, c2rt_trace.inc: tile_mask_entry) once per frame with n_cull != 0, read by the frame
     * kernel with one scalar load per tile.  The table covers the local rows [mask_row0, mask_row0 + mask_rows)
     * (mask_row0 a multiple of the tile height): the launches of a chunked host-output frame share one table */
    const uint32_t *tile_masks;
    uint32_t mask_row0, mask_rows;
    uint32_t mask_entries;         /* entries of the (first) table: tile_mask_entries() */
    /* diagnostics build only (make VARIANT=tilestats EXTRA_HIPFLAGS=-DC2RT_TILE_STATS=1, scripts/tile_stats.py):
     * per tile {shader-clock cycles the wave spent on it, class bits}; never read by the product build */
    uint32_t *tile_stats;
    float *out;                    /* local_rows * width * 3 floats */
    /* non-null: the frame leaves the kernel display-encoded instead (Color.toRGB32 through the reference's
     * 4097-entry sRGB table, one 32-bit word per pixel, same indexing as `out`, which is then unused) */
    uint32_t *out_rgb32;
    const uint8_t *srgb_lut;
    unsigned long long *ray_counters; /* [3] primary rays, shadow rays, CSG hit lists that reached the cap (nullable) */
    /* tiles that the production instances rendered a second time through the compiler's divide / sqrt
     * because a lane met an operand outside a lean window (c2rt_trace.inc); cumulative, never null */
    unsigned long long *redo_counter;
    /* pixel probe */
    int32_t probe_x, probe_y;
    c2rt_trace_result *probe_out;
};

static_assert(sizeof(RenderParams) <= 4096, "the kernel-argument segment holds at most 4 KiB");

/* Scene feature bits selecting the kernel instance (so that a plane-only
 * scene does not pay the registers of the CSG path). */
struct KernelVariant {
    int csg_levels;                /* 0..C2RT_MAX_CSG_DEPTH */
    bool dof_or_stereo;
};

/* implemented in c2rt_kernels.hip; return hipError_t as int */
template <int LEVELS>
int launch_render_level(const RenderParams &p, bool dof_or_stereo, void *stream);
template <> int launch_render_level<0>(const RenderParams &, bool, void *);
template <> int launch_render_level<1>(const RenderParams &, bool, void *);
template <> int launch_render_level<2>(const RenderParams &, bool, void *);
template <> int launch_render_level<3>(const RenderParams &, bool, void *);
template <> int launch_render_level<4>(const RenderParams &, bool, void *);
int launch_render(const RenderParams &p, const KernelVariant &v, void *stream);
int launch_probe(const RenderParams &p, const KernelVariant &v, void *stream);
/* entries of RenderParams::tile_masks a launch of `p` reads (4 words each, twice that with several culled lights);
 * the pre-pass that fills them */
size_t tile_mask_entries(const RenderParams &p);
int launch_tile_masks(const RenderParams &p, uint32_t *table, void *stream);
int launch_deinterleave(const float *gathered, float *frame, uint32_t width, uint32_t height,
                        uint32_t strip_height, uint32_t world, uint32_t rows_pad, uint32_t words_per_pixel, void *stream);
int launch_encode_rgb32(const float *frame, uint32_t *out, uint64_t n_pixels,
                        const uint8_t *lut_dev, void *stream);

} // namespace c2rt
#endif
