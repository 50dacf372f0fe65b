/*
 * c2rt_api.cpp — implementation of the C ABI in include/c2rt.h: context,
 * scene validation + upload (SoA tables -> scalar-loadable records in HBM),
 * frame / pixel-probe launches, strip de-interleave and display encode.
 *
 * There is no CPU fallback anywhere in this file: without a usable HIP
 * device every entry point fails with C2RT_ERR_NO_DEVICE / C2RT_ERR_HIP.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "c2rt_device.h"

using namespace c2rt;

constexpr int kMaxChunks = 16;     /* row chunks of a host-output frame */

/* Per-frame scratch a launch writes before it reads it — the tile-mask table (pre-pass kernel) and the nested-CSG
 * retry list — exists once per STREAM the context has rendered on (up to kScratchSlots of them), keyed by the
 * stream handle (compared, never dereferenced).  Frames on one stream are in order and share a slot; frames on
 * different streams touch different slots, so they need no ordering at all: no event per frame (an event record
 * is a ~2 us gap in the queue: 6 % of a 1080p lecture4 frame) and frames of one context may overlap.  A 17th
 * stream recycles the least recently used slot after a device sync (the one place a frame call can block). */
constexpr int kScratchSlots = 16;
struct FrameScratch {
    const void *key = nullptr;
    bool used = false;
    uint64_t tick = 0;
    uint32_t *tile_masks = nullptr; /* RenderParams::tile_masks: 4 words per tile of the frame's local rows, x 2 tables */
    size_t tile_mask_entries = 0;
    uint32_t *retry_list = nullptr; /* RenderParams::retry_list: [0] count, then tile (block) indices */
    size_t retry_words = 0;
};

/* Environment hooks (A/B measurement and test knobs: C2RT_EXACT, C2RT_NO_IDN, C2RT_DEBUG_CULL, C2RT_CSG_FIRST_CAP,
 * C2RT_HOST_*) exist in the DIAGNOSTICS build only — chess2rt_amd/libc2rt_diag.so, this file compiled with
 * -DC2RT_DIAG=1 over the same kernel objects (Makefile).  The product library reads no environment variable: a
 * drop-in renderer does not change kernels on a stray variable (tests/test_abi_exports.py checks that libc2rt.so
 * does not even import getenv). */
#ifndef C2RT_DIAG
#define C2RT_DIAG 0
#endif
#if C2RT_DIAG
static const char *diag_env(const char *name) { return std::getenv(name); }
#else
static constexpr const char *diag_env(const char *) { return nullptr; }
#endif

struct c2rt_ctx {
    int device = 0;
    /* c2rt_init_multi: the further device slots of this (lead) context, each a complete
     * single-device context of its own; empty for c2rt_init */
    std::vector<c2rt_ctx *> peers;
    bool peer_mapped = true;       /* (peer slot) can store into the lead device's memory */
    hipEvent_t ev_ready = nullptr, ev_done = nullptr; /* cross-device ordering of device-output frames */
    uint64_t scene_gen = 0;        /* c2rt_scene_generation */
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;              /* D2H of finished chunks */
    hipEvent_t chunk_done[kMaxChunks] = {};
    hipStream_t copy_stream2 = nullptr; /* chunks alternate between two copy streams (two SDMA engines) */
    std::vector<std::pair<float *, size_t>> pinned; /* c2rt_pin_host_buffer */

    bool has_scene = false;
    int csg_levels = 0;
    DevGeom *geoms = nullptr;
    DevNode *nodes = nullptr;
    DevShader *shaders = nullptr;
    DevTex *textures = nullptr;
    DevLight *lights = nullptr;
    float *texels = nullptr;
    uint32_t n_nodes = 0, n_lights = 0;
    float ambient[3] = {0, 0, 0};
    uint32_t max_trace_depth = 0;

    /* world-space corners of every node's padded bounding box (for the per-frame
     * screen rectangles); node_boxed[n] = 0: unbounded, never culled */
    uint32_t planes_only = 0;          /* every node is an axis plane (kNodeAxisPlane) */
    uint32_t all_identity = 0;         /* every node has kNodeIdentityMatrix */
    int32_t ground_node = -1;          /* see RenderParams::ground_node */
    double ground_y = 0;
    double *shadow_rects = nullptr;    /* [kMaxCullNodes][4] */
    std::vector<double> node_box;  /* [n_nodes][8][3] */
    std::vector<double> light_pos; /* [n_lights][3] host copy for the per-frame shadow-cull thresholds */
    std::vector<uint8_t> node_boxed;

    float *frame = nullptr;        /* staging frame for host-output renders */
    size_t frame_floats = 0;
    unsigned long long *counters = nullptr; /* [0..2]: RenderParams::ray_counters (reset per counted frame); [3]: RenderParams::redo_counter (cumulative) */
    c2rt_trace_result *probe = nullptr;
    uint8_t *srgb_lut = nullptr;   /* [4097] */
    FrameScratch scratch[kScratchSlots];
    uint64_t scratch_tick = 0;
    uint32_t *tile_stats = nullptr; /* diagnostics (c2rt_debug_set_tile_stats): caller-owned device buffer */
    bool counters_valid = false;
    /* The ray counters are the one per-frame resource shared by all streams: COUNTED frames (opts->count_rays, a
     * test / diagnostics mode) enqueued without a host sync are ordered among themselves and against the counter
     * read-back by ev_inflight, recorded on the caller's stream behind such a frame (the next counted frame waits
     * for it on the device, the blocking entry points and c2rt_get_ray_stats on the host).  The library never
     * keeps a caller's stream handle for use: the stream may be destroyed the moment the call returns.
     * has_inflight: ev_inflight has been recorded and not yet waited for by the host. */
    hipEvent_t ev_inflight = nullptr;
    bool has_inflight = false;
};

namespace {

int fail(c2rt_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return status;
}

#define HIP_TRY(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return fail(ctx, C2RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int upload(c2rt_ctx *ctx, T **dst, const std::vector<T> &src)
{
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return C2RT_OK;
}

bool is_csg(int t) { return t == C2RT_GEOM_CSG_UNION || t == C2RT_GEOM_CSG_INTER || t == C2RT_GEOM_CSG_DIFF; }

/* nesting depth of the CsgOp tree under `g` (0 for primitives); -1 on a cycle
 * or an out-of-range child */
int csg_depth(const c2rt_scene_desc *s, int32_t g, std::vector<int> &state, std::vector<int> &memo)
{
    if (g < 0 || (uint32_t)g >= s->n_geoms) return -1;
    if (state[g] == 1) return -1; /* on the current path: cycle */
    if (state[g] == 2) return memo[g];
    int d = 0;
    if (is_csg(s->geom_type[g])) {
        state[g] = 1;
        const int l = csg_depth(s, s->geom_child[2 * g + 0], state, memo);
        const int r = csg_depth(s, s->geom_child[2 * g + 1], state, memo);
        if (l < 0 || r < 0) return -1;
        d = 1 + (l > r ? l : r);
    }
    state[g] = 2;
    memo[g] = d;
    return d;
}

/* ---- conservative bounds and exact CSG shortcuts (see c2rt_device.h) ---- */
struct BoundInfo { bool done = false, bounded = false; double c[3] = {0, 0, 0}, r = 0; };

bool subtree_has_leaf(const c2rt_scene_desc *s, int32_t g, int32_t leaf)
{
    if (!is_csg(s->geom_type[g])) return g == leaf;
    return subtree_has_leaf(s, s->geom_child[2 * g], leaf) || subtree_has_leaf(s, s->geom_child[2 * g + 1], leaf);
}

BoundInfo enclose(const BoundInfo &a, const BoundInfo &b)
{
    BoundInfo o;
    o.done = true;
    if (!a.bounded || !b.bounded) return o;
    const double d = std::sqrt((a.c[0] - b.c[0]) * (a.c[0] - b.c[0]) + (a.c[1] - b.c[1]) * (a.c[1] - b.c[1]) + (a.c[2] - b.c[2]) * (a.c[2] - b.c[2]));
    o.bounded = true;
    for (int i = 0; i < 3; ++i) o.c[i] = a.c[i]; /* keep a's centre: simple and conservative */
    o.r = std::fmax(a.r, d + b.r);
    return o;
}

/* geometries must already be validated acyclic (csg_depth) */
BoundInfo bound_of(const c2rt_scene_desc *s, int32_t g, std::vector<BoundInfo> &memo, std::vector<DevGeom> &geoms)
{
    if (memo[g].done) return memo[g];
    BoundInfo b;
    b.done = true;
    const int t = s->geom_type[g];
    const double *p = s->geom_param + 4 * (size_t)g;
    if (t == C2RT_GEOM_SPHERE) {
        b.bounded = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && std::isfinite(p[3]);
        for (int i = 0; i < 3; ++i) b.c[i] = p[i];
        b.r = std::fabs(p[3]);
    } else if (t == C2RT_GEOM_CUBE) {
        b.bounded = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && std::isfinite(p[3]);
        for (int i = 0; i < 3; ++i) b.c[i] = p[i];
        b.r = std::fabs(p[3]) * 0.5 * 1.7320508075688774; /* half diagonal */
    } else if (is_csg(t)) {
        const int32_t l = s->geom_child[2 * g], r = s->geom_child[2 * g + 1];
        const BoundInfo bl = bound_of(s, l, memo, geoms), br = bound_of(s, r, memo, geoms);
        const bool shortA = t != C2RT_GEOM_CSG_UNION && !subtree_has_leaf(s, r, l);
        const bool shortB = t == C2RT_GEOM_CSG_INTER && !is_csg(s->geom_type[l]);
        if (shortA) geoms[g].flags |= kCsgShortA;
        if (shortB) geoms[g].flags |= kCsgShortB;
        if (shortA && bl.bounded) b = bl;       /* nothing on the left => false */
        else b = enclose(bl, br);               /* nothing on either side => false */
        b.done = true;
    } /* plane: unbounded */
    if (b.bounded) {
        /* pad: the reject test runs in fp64 on coordinates of this magnitude */
        const double mag = std::fabs(b.c[0]) + std::fabs(b.c[1]) + std::fabs(b.c[2]) + b.r;
        const double rp = b.r * (1 + 1e-6) + 1e-6 * mag + 1e-9;
        if (std::isfinite(rp) && rp * rp < 1e300) {
            geoms[g].flags |= kGeomBounded;
            geoms[g].bound[0] = b.c[0];
            geoms[g].bound[1] = b.c[1];
            geoms[g].bound[2] = b.c[2];
            geoms[g].bound[3] = rp * rp;
        }
    }
    memo[g] = b;
    return b;
}

/* Object-space axis-aligned box with the same contract as the bounding sphere above — a ray
 * (segment) that does not enter it cannot make Geometry.intersect return true — but tight: the
 * sphere's own box, the cube itself, the left child's box for an Inter/Diff with shortcut A
 * (no left hit => false; left hits beyond the segment put the winner beyond it too), the hull of
 * both children otherwise.  It feeds the per-frame screen rectangles and the shadow rectangles,
 * where the bounding sphere of a cube costs a factor 1.7 per axis.  Call after bound_of (flags). */
struct BoxInfo { bool done = false, bounded = false; double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; };

BoxInfo box_of(const c2rt_scene_desc *s, int32_t g, std::vector<BoxInfo> &memo, const std::vector<DevGeom> &geoms)
{
    if (memo[g].done) return memo[g];
    BoxInfo b;
    b.done = true;
    const int t = s->geom_type[g];
    const double *p = s->geom_param + 4 * (size_t)g;
    if (t == C2RT_GEOM_SPHERE || t == C2RT_GEOM_CUBE) {
        b.bounded = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && std::isfinite(p[3]);
        const double e = t == C2RT_GEOM_SPHERE ? std::fabs(p[3]) : std::fabs(p[3]) * 0.5;
        for (int i = 0; i < 3; ++i) { b.lo[i] = p[i] - e; b.hi[i] = p[i] + e; }
    } else if (is_csg(t)) {
        const BoxInfo bl = box_of(s, s->geom_child[2 * g], memo, geoms), br = box_of(s, s->geom_child[2 * g + 1], memo, geoms);
        /* Where can a HIT of this CsgOp lie?  On a leaf's surface of either subtree, in general: the union.  Inside
         * the left child's box only when the walk's `inL` really means "inside the left child": the left child is a
         * PRIMITIVE (its entries carry its own identity, so they and only they toggle inL — kCsgShortA excludes the
         * same leaf inside the right subtree) and the operator needs inL (Inter / Diff).  With a CsgOp as left child
         * no entry ever equals `left` (rt/geometry.d:314-317 compares the LEAF): inL is the parity of the left
         * list for the whole walk, and an Inter / Diff can come out "in" at an entry of the RIGHT child far outside
         * the left child's box — e.g. a shadow ray whose left hits lie beyond the light, occluded by a right-child
         * surface in front of it.  (Found by the offline sweep, seed 108921: three pixels of a 64x48 frame lost a
         * shadow to the view-pyramid culling of shadow rays, which asks where the occluder can BE.) */
        if ((geoms[g].flags & kCsgShortA) && bl.bounded && !is_csg(s->geom_type[s->geom_child[2 * g]])) {
            b = bl;
        } else if (bl.bounded && br.bounded) {
            b.bounded = true;
            for (int i = 0; i < 3; ++i) { b.lo[i] = std::min(bl.lo[i], br.lo[i]); b.hi[i] = std::max(bl.hi[i], br.hi[i]); }
        }
        b.done = true;
    } /* plane: unbounded */
    memo[g] = b;
    return b;
}

/* convertTo8bit_sRGB — rt/color.d:194-207 (note the 12.02) and the 4097-entry
 * cache built by the module constructor rt/color.d:224-228 */
void build_srgb_lut(uint8_t *lut)
{
    for (int i = 0; i < 4097; ++i) {
        float x = i / 4096.0f;
        uint8_t v;
        if (x <= 0) v = 0;
        else if (x >= 1) v = 255;
        else {
            if (x <= 0.0031308f) x = x * 12.02f;
            else x = (float)(1.055 * std::pow((double)x, 1 / 2.4) - 0.055);
            v = (uint8_t)(int)std::floor(x * 255.0f);
        }
        lut[i] = v;
    }
}

uint32_t strip_h(const c2rt_render_opts *o) { return o->strip_height ? o->strip_height : 1u; }

uint32_t local_rows_of(const c2rt_render_opts *o, uint32_t rank)
{
    if (o->strip_world <= 1) return o->height;
    const uint32_t sh = strip_h(o);
    const uint32_t n_strips = (o->height + sh - 1) / sh;
    uint32_t rows = 0;
    for (uint32_t s = rank; s < n_strips; s += o->strip_world) {
        const uint32_t y0 = s * sh;
        rows += (y0 + sh <= o->height) ? sh : (o->height - y0);
    }
    return rows;
}

int check_frame_args(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *o)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    if (!cam || !o) return fail(ctx, C2RT_ERR_INVALID_ARG, "null camera or options");
    if (!ctx->has_scene) return fail(ctx, C2RT_ERR_NO_SCENE, "no scene uploaded");
    if (o->width == 0 || o->height == 0 || o->width > (1u << 16) || o->height > (1u << 16))
        return fail(ctx, C2RT_ERR_INVALID_ARG, "bad frame size %ux%u", o->width, o->height);
    if (o->taps != C2RT_TAPS_1 && o->taps != C2RT_TAPS_REF5 && o->taps != C2RT_TAPS_4)
        return fail(ctx, C2RT_ERR_INVALID_ARG, "bad tap mode %u", o->taps);
    if (o->strip_world > 1 && o->strip_rank >= o->strip_world)
        return fail(ctx, C2RT_ERR_INVALID_ARG, "strip_rank %u >= strip_world %u", o->strip_rank, o->strip_world);
    if (cam->dof && (cam->num_samples == 0 || cam->num_samples > 4096))
        return fail(ctx, C2RT_ERR_LIMIT, "dof numSamples %u outside 1..4096", cam->num_samples);
    if (o->prepass_bucket > 65536) return fail(ctx, C2RT_ERR_UNSUPPORTED, "prepass bucket size %u > 65536", o->prepass_bucket);
    if (!(cam->frame_width > 0) || !(cam->frame_height > 0))
        return fail(ctx, C2RT_ERR_INVALID_ARG, "camera frame size must be positive");
    return C2RT_OK;
}

/* Convex hull of eight 2-D points (Andrew's monotone chain) as up to kHullEdges half planes
 * a*x + b*y + c >= 0 ((a, b) of unit length), each pushed outward by `pad`.  The central projection of a
 * box is at most a hexagon; false (nothing written) for a degenerate hull or one with more edges. */
bool hull_half_planes(const double pts[8][2], double pad, double out[kHullEdges][3])
{
    int order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    std::sort(order, order + 8, [&](int i, int j) { return pts[i][0] < pts[j][0] || (pts[i][0] == pts[j][0] && pts[i][1] < pts[j][1]); });
    auto cross = [&](int o, int a, int b) {
        return (pts[a][0] - pts[o][0]) * (pts[b][1] - pts[o][1]) - (pts[a][1] - pts[o][1]) * (pts[b][0] - pts[o][0]);
    };
    int hv[17], m = 0;
    for (int i = 0; i < 8; ++i) { /* lower chain */
        while (m >= 2 && cross(hv[m - 2], hv[m - 1], order[i]) <= 0) --m;
        hv[m++] = order[i];
    }
    for (int i = 6, t = m + 1; i >= 0; --i) { /* upper chain */
        while (m >= t && cross(hv[m - 2], hv[m - 1], order[i]) <= 0) --m;
        hv[m++] = order[i];
    }
    --m; /* the last point repeats the first; hv[0..m) is the hull, counter-clockwise */
    if (m < 3 || m > kHullEdges) return false;
    /* Two projected corners that nearly coincide (the eye almost on the line of a box edge) pass the turn
     * tests on noise-dominated cross products and would contribute an "edge" whose half plane is not a
     * supporting line of the true hull — it could cull tiles the box covers.  Such a hull is refused (the
     * caller keeps the rectangle, which has no such failure mode): every edge must be longer than 1e-6 of
     * the hull's extent. */
    double ext = 0;
    for (int i = 0; i < m; ++i)
        for (int j = i + 1; j < m; ++j)
            ext = std::fmax(ext, std::fmax(std::fabs(pts[hv[i]][0] - pts[hv[j]][0]), std::fabs(pts[hv[i]][1] - pts[hv[j]][1])));
    double tmp[kHullEdges][3];
    for (int e = 0; e < kHullEdges; ++e) { tmp[e][0] = tmp[e][1] = 0; tmp[e][2] = 1; }
    for (int e = 0; e < m; ++e) {
        const double *p0 = pts[hv[e]], *p1 = pts[hv[(e + 1) % m]];
        double a = -(p1[1] - p0[1]), b = p1[0] - p0[0]; /* interior to the left of p0 -> p1: inward normal */
        const double len = std::sqrt(a * a + b * b);
        if (!(len > 1e-6 * ext) || !std::isfinite(len)) return false;
        a /= len; b /= len;
        const double c = -(a * p0[0] + b * p0[1]) + pad;
        if (!std::isfinite(c)) return false;
        tmp[e][0] = a; tmp[e][1] = b; tmp[e][2] = c;
    }
    std::memcpy(out, tmp, sizeof tmp);
    return true;
}

void cull_rect_of(const c2rt_camera_frame *cam, const double *corners, int32_t out[4], float hull[kHullEdges][3]);
void light_side_of(const c2rt_camera_frame *cam, const double *light, int32_t out[8]);

void fill_params(const c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *o, RenderParams &p)
{
    std::memset(&p, 0, sizeof p);
    p.geoms = ctx->geoms;
    p.nodes = ctx->nodes;
    p.shaders = ctx->shaders;
    p.textures = ctx->textures;
    p.lights = ctx->lights;
    p.texels = ctx->texels;
    p.n_nodes = ctx->n_nodes;
    p.n_lights = ctx->n_lights;
    std::memcpy(p.ambient, ctx->ambient, sizeof p.ambient);
    p.max_trace_depth = ctx->max_trace_depth;
    p.cam = *cam;
    for (int i = 0; i < 3; ++i) {
        p.cam_du[i] = cam->up_right[i] - cam->up_left[i];
        p.cam_dv[i] = cam->down_left[i] - cam->up_left[i];
    }
    /* lean:: divides sample coordinates by the camera's frame size through these (c2rt_trace.inc, screen_ray);
     * a frame size that is not a sane denominator (the ABI accepts any positive double) — or, in the diagnostics
     * build, C2RT_EXACT=1 (A/B runs: the compiler's IEEE divide / sqrt everywhere, as in rounds 1-2) — sends
     * every tile down the exact:: path */
    p.cam_rw = 1.0 / cam->frame_width;
    p.cam_rh = 1.0 / cam->frame_height;
    static const bool env_exact = [] { const char *e = diag_env("C2RT_EXACT"); return e && e[0] == '1'; }();
    const auto sane = [](double v) { return v >= 0x1p-100 && v < 0x1p100; };
    p.force_exact = (env_exact || !sane(cam->frame_width) || !sane(cam->frame_height)) ? 1u : 0u;
    p.width = o->width;
    p.height = o->height;
    p.taps = o->prepass_bucket ? 1u : o->taps;
    p.prepass_bucket = o->prepass_bucket;
    p.strip_height = strip_h(o);
    p.strip_rank = o->strip_world > 1 ? o->strip_rank : 0;
    p.strip_world = o->strip_world > 1 ? o->strip_world : 1;
    p.local_rows = local_rows_of(o, p.strip_rank);
    p.tiles_x = (o->width + kTileW - 1) / kTileW;
    p.tiles_y = (p.local_rows + kTileH - 1) / kTileH;
    p.blocks_x = (p.tiles_x + kWavesPerBlock - 1) / kWavesPerBlock;
    p.seed = o->seed;
    p.tile_stats = ctx->tile_stats;
    p.row_group_start = 0;
    p.planes_only = ctx->planes_only;
    static const bool no_idn = [] { const char *e = diag_env("C2RT_NO_IDN"); return e && e[0] == '1'; }(); /* diagnostics build, A/B: the general instances */
    p.all_identity = no_idn ? 0u : ctx->all_identity;
    p.ground_node = ctx->ground_node;
    p.ground_y = ctx->ground_y;
    p.shadow_rects = ctx->shadow_rects;
    p.n_cull = 0;
    if (!cam->dof && cam->stereo_separation == 0 && !o->prepass_bucket) {
        /* up to the last bounded node; nothing bounded => no per-wave work at all */
        const uint32_t lim = ctx->n_nodes < (uint32_t)kMaxCullNodes ? ctx->n_nodes : (uint32_t)kMaxCullNodes;
        for (uint32_t n = 0; n < lim; ++n)
            if (ctx->node_boxed[n]) p.n_cull = n + 1;
        for (uint32_t n = 0; n < p.n_cull; ++n) {
            for (int e = 0; e < kHullEdges; ++e) { p.cull_hull[n][e][0] = p.cull_hull[n][e][1] = 0.0f; p.cull_hull[n][e][2] = 1.0f; }
            if (ctx->node_boxed[n]) cull_rect_of(cam, &ctx->node_box[(size_t)n * 24], p.cull_rect[n], p.cull_hull[n]);
            else { p.cull_rect[n][0] = p.cull_rect[n][1] = INT32_MIN; p.cull_rect[n][2] = p.cull_rect[n][3] = INT32_MAX; }
        }
        /* dispatch order: start at the tile rows where the boxed nodes begin (their tiles are the
         * expensive ones), wrap around to the rows above them (mostly sky) at the end */
        int32_t top = INT32_MAX;
        for (uint32_t n = 0; n < p.n_cull; ++n)
            if (ctx->node_boxed[n] && p.cull_rect[n][1] < top) top = p.cull_rect[n][1];
        if (top != INT32_MAX && top > 0 && (uint32_t)top < o->height) {
            /* frame row -> local row of this launch (strips: rows are dealt round-robin) */
            const uint32_t local = (uint32_t)top / p.strip_world;
            const uint32_t groups = (p.tiles_y + 7u) / 8u;
            const uint32_t g = local / (kTileH * 8u);
            p.row_group_start = g < groups ? g : 0;
        }
        if (p.n_cull) {
            p.n_cull_lights = ctx->n_lights < (uint32_t)kMaxCullLights ? ctx->n_lights : (uint32_t)kMaxCullLights;
            for (uint32_t l = 0; l < p.n_cull_lights; ++l) light_side_of(cam, &ctx->light_pos[3 * (size_t)l], p.light_side[l]);
        }
    }
    /* diagnostics build only (like C2RT_CSG_FIRST_CAP; frames are unchanged by construction, slower): C2RT_DEBUG_CULL bit 0:
     * no culling rectangles at all; bit 1: no ground-plane refinement of the shadow mask; bit 2: no view-pyramid
     * culling of shadow rays */
    static const int debug_cull = [] {
        const char *e = diag_env("C2RT_DEBUG_CULL");
        const int v = e ? std::atoi(e) : 0;
        if (v) std::fprintf(stderr, "libc2rt: diagnostics hook C2RT_DEBUG_CULL=%d is active (culling partly disabled; frames are unchanged, slower)\n", v);
        return v;
    }();
    if (debug_cull & 1) p.n_cull = 0;
    if (debug_cull & 2) p.ground_node = -1;
    if (debug_cull & 4) p.n_cull_lights = 0;
}

/* Screen rectangle of a node for this frame: the projection of the 8 world-space
 * box corners through the camera (a projective map, convex on the half space in
 * front of the eye), widened by 2 pixels (the AA taps reach 0.6 px, rounding is
 * ~1e-13 px).  Any corner at or behind the eye plane => the whole frame. */
void cull_rect_of(const c2rt_camera_frame *cam, const double *corners, int32_t out[4], float hull[kHullEdges][3])
{
    const int32_t kAll[4] = {INT32_MIN, INT32_MIN, INT32_MAX, INT32_MAX};
    std::memcpy(out, kAll, sizeof kAll);
    double du[3], dv[3], ul[3];
    for (int i = 0; i < 3; ++i) {
        du[i] = cam->up_right[i] - cam->up_left[i];
        dv[i] = cam->down_left[i] - cam->up_left[i];
        ul[i] = cam->up_left[i] - cam->pos[i];
    }
    /* solve a*du + b*dv + l*ul = w by Cramer's rule */
    auto det3 = [](const double *a, const double *b, const double *c) {
        return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    };
    const double det = det3(du, dv, ul);
    if (!std::isfinite(det) || det == 0) return;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    double pts[8][2];
    for (int k = 0; k < 8; ++k) {
        double w[3];
        for (int i = 0; i < 3; ++i) w[i] = corners[3 * k + i] - cam->pos[i];
        const double a = det3(w, dv, ul) / det, b = det3(du, w, ul) / det, l = det3(du, dv, w) / det;
        if (!(l > 1e-9) || !std::isfinite(a) || !std::isfinite(b)) return; /* at / behind the eye: no culling */
        const double px = a / l * cam->frame_width, py = b / l * cam->frame_height;
        if (!std::isfinite(px) || !std::isfinite(py)) return;
        pts[k][0] = px; pts[k][1] = py;
        xmin = std::fmin(xmin, px); xmax = std::fmax(xmax, px);
        ymin = std::fmin(ymin, py); ymax = std::fmax(ymax, py);
    }
    const double lim = 1e9;
    if (xmin < -lim || ymin < -lim || xmax > lim || ymax > lim) return;
    out[0] = (int32_t)std::floor(xmin) - 2;
    out[1] = (int32_t)std::floor(ymin) - 2;
    out[2] = (int32_t)std::ceil(xmax) + 3;
    out[3] = (int32_t)std::ceil(ymax) + 3;

    /* the hull of the eight projected corners, one outward-padded half plane per edge.  Only for rectangles
     * of sane size (float coefficients: |c| < 1e6 keeps the evaluation error at a tile corner below
     * 0.25 px, and the pad is 2.5 px where the rectangle's is 2). */
    if (!hull || xmin < -3e5 || ymin < -3e5 || xmax > 3e5 || ymax > 3e5) return;
    double hp[kHullEdges][3];
    if (!hull_half_planes(pts, 2.5, hp)) return;
    for (int e = 0; e < kHullEdges; ++e)
        if (std::fabs(hp[e][2]) > 1e6) return;
    for (int e = 0; e < kHullEdges; ++e)
        for (int k = 0; k < 3; ++k) hull[e][k] = (float)hp[e][k];
}

/* For one light: the integer boundary coordinates x (pixels) for which the light is
 * certainly in the closed half space a - l*x/W >= 0 (">= x" side of the vertical
 * boundary plane through the eye) resp. <= 0, and the same for y.  (a, b, l) are
 * the light's coordinates in the (du, dv, ul) basis; the half spaces are linear in
 * them, so this is valid for lights behind the eye too.  One pixel of slack. */
void light_side_of(const c2rt_camera_frame *cam, const double *light, int32_t out[8])
{
    for (int i = 0; i < 4; ++i) { out[2 * i] = 1; out[2 * i + 1] = 0; } /* empty intervals */
    double du[3], dv[3], ul[3], w[3];
    for (int i = 0; i < 3; ++i) {
        du[i] = cam->up_right[i] - cam->up_left[i];
        dv[i] = cam->down_left[i] - cam->up_left[i];
        ul[i] = cam->up_left[i] - cam->pos[i];
        w[i] = light[i] - cam->pos[i];
    }
    auto det3 = [](const double *a, const double *b, const double *c) {
        return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    };
    const double det = det3(du, dv, ul);
    if (!std::isfinite(det) || det == 0) return;
    const double a = det3(w, dv, ul) / det, b = det3(du, w, ul) / det, l = det3(du, dv, w) / det;
    if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(l)) return;
    const double lim = 1e9;
    auto axis = [&](double c, double size, int32_t *ge, int32_t *le) {
        /* phi(x) = c*size - l*x: ">= x side" <=> phi(x) >= 0, "<= x side" <=> phi(x) <= 0 */
        const double eps = 1e-12 * (std::fabs(c) + std::fabs(l));
        if (std::fabs(l) <= eps) {
            if (c > eps) { ge[0] = INT32_MIN; ge[1] = INT32_MAX; }
            if (c < -eps) { le[0] = INT32_MIN; le[1] = INT32_MAX; }
            return;
        }
        const double xs = c * size / l;
        if (!(std::fabs(xs) < lim)) return;
        if (l > 0) { /* phi decreasing: >= 0 for x <= xs */
            ge[0] = INT32_MIN; ge[1] = (int32_t)std::floor(xs) - 1;
            le[0] = (int32_t)std::ceil(xs) + 1; le[1] = INT32_MAX;
        } else {     /* phi increasing: >= 0 for x >= xs */
            ge[0] = (int32_t)std::ceil(xs) + 1; ge[1] = INT32_MAX;
            le[0] = INT32_MIN; le[1] = (int32_t)std::floor(xs) - 1;
        }
    };
    axis(a, cam->frame_width, out + 0, out + 2);
    axis(b, cam->frame_height, out + 4, out + 6);
}

KernelVariant variant_of(const c2rt_ctx *ctx, const c2rt_camera_frame *cam)
{
    KernelVariant v;
    v.csg_levels = ctx->csg_levels;
    v.dof_or_stereo = cam->dof != 0 || cam->stereo_separation != 0;
    return v;
}

/* the scratch slot of `stream` (see FrameScratch); never fails: the least recently used slot is recycled after a
 * device sync (whatever still reads its tables has finished then) */
FrameScratch &scratch_for(c2rt_ctx *ctx, hipStream_t stream)
{
    const void *key = static_cast<const void *>(stream);
    FrameScratch *pick = nullptr;
    for (FrameScratch &f : ctx->scratch)
        if (f.used && f.key == key) { pick = &f; break; }
    if (!pick)
        for (FrameScratch &f : ctx->scratch)
            if (!f.used) { pick = &f; break; }
    if (!pick) {
        pick = &ctx->scratch[0];
        for (FrameScratch &f : ctx->scratch)
            if (f.tick < pick->tick) pick = &f;
        (void)hipDeviceSynchronize();
    }
    pick->used = true;
    pick->key = key;
    pick->tick = ++ctx->scratch_tick;
    return *pick;
}

/* The tiles' culling masks for the local rows [p.row_offset, p.row_offset + p.local_rows), by the pre-pass kernel,
 * in front of the frame kernel on the same stream (one table per stream of the context: FrameScratch).  Sets p.tile_masks / mask_row0 / mask_rows; a no-op for frames without culling rectangles.
 * Returns a hipError_t. */
int prepare_tile_masks(c2rt_ctx *ctx, RenderParams &p, const KernelVariant &v, hipStream_t stream)
{
    p.tile_masks = nullptr;
    p.mask_row0 = p.row_offset;
    p.mask_rows = p.local_rows;
    if (!p.n_cull || v.dof_or_stereo || !p.local_rows) return 0;
    const size_t entries = tile_mask_entries(p);
    FrameScratch &sc = scratch_for(ctx, stream);
    if (entries > sc.tile_mask_entries) {
        if (sc.tile_masks) { (void)hipFree(sc.tile_masks); sc.tile_masks = nullptr; sc.tile_mask_entries = 0; }
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&sc.tile_masks), entries * 8 * sizeof(uint32_t));
        if (e != hipSuccess) return (int)e;
        sc.tile_mask_entries = entries;
    }
    p.tile_masks = sc.tile_masks;
    p.mask_entries = (uint32_t)entries;
    return launch_tile_masks(p, sc.tile_masks, stream);
}

/* One frame launch.  Scenes with nested CsgOps (depth >= 2) run the kernel with a reduced hit-stack
 * capacity (three waves per SIMD instead of one at depth 4) and then, on the same stream, the
 * full-capacity relaunch over the tiles that overflowed it — none, for trees whose primitives yield
 * their two hits (RenderParams::retry_list; c2rt_kernels.hip, csg_intersect).  Returns a hipError_t. */
int launch_frame(c2rt_ctx *ctx, RenderParams &p, const KernelVariant &v, hipStream_t stream)
{
    const int levels = ctx->csg_levels;
    /* test hook (diagnostics build only): C2RT_CSG_FIRST_CAP=<entries> shrinks the first pass's stack so that the
     * overflow -> retry path runs on ordinary scenes (tests/test_gpu_parity.py); never below 1, never above full */
    static const int forced_cap = [] {
        const char *e = diag_env("C2RT_CSG_FIRST_CAP");
        const int v = e ? std::atoi(e) : 0;
        if (v > 0) std::fprintf(stderr, "libc2rt: test hook C2RT_CSG_FIRST_CAP=%d is active (first-pass CSG hit stacks shrunk; frames are unchanged, nested-CSG scenes are slower)\n", v);
        return v;
    }();
    int first_cap = kCsgFirstCap(levels);
    if (forced_cap > 0 && levels >= 2) first_cap = forced_cap < kCsgFullCap(levels) ? forced_cap : kCsgFullCap(levels);
    p.csg_cap = (uint32_t)first_cap;
    if (levels == 0) p.csg_cap = 0;
    p.retry_mode = 0;
    p.redo_counter = ctx->counters + 3;
    if (!p.tile_masks) { /* (a chunked frame has prepared its table already: render_to_host) */
        const int e = prepare_tile_masks(ctx, p, v, stream);
        if (e != 0) return e;
    }
    if (levels < 2) return launch_render(p, v, stream);
    const size_t blocks = (size_t)p.blocks_x * ((p.tiles_y + 7u) / 8u * 8u);
    FrameScratch &sc = scratch_for(ctx, stream);
    if (blocks + 1 > sc.retry_words) {
        if (sc.retry_list) { (void)hipFree(sc.retry_list); sc.retry_list = nullptr; sc.retry_words = 0; }
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&sc.retry_list), (blocks + 1) * sizeof(uint32_t));
        if (e != hipSuccess) return (int)e;
        sc.retry_words = blocks + 1;
    }
    p.retry_list = sc.retry_list;
    p.retry_max = (uint32_t)blocks;
    hipError_t e = hipMemsetAsync(sc.retry_list, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    int r = launch_render(p, v, stream);
    if (r != 0 || first_cap >= kCsgFullCap(levels)) return r;
    p.retry_mode = 1;
    p.csg_cap = (uint32_t)kCsgFullCap(levels);
    r = launch_render(p, v, stream);
    p.retry_mode = 0;
    p.csg_cap = (uint32_t)first_cap;
    return r;
}

int render_device(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, float *out_dev,
                  hipStream_t stream)
{
    RenderParams p;
    fill_params(ctx, cam, opts, p);
    p.out = out_dev;
    ctx->counters_valid = false;
    if (opts->count_rays) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, 3 * sizeof(unsigned long long), stream));
        p.ray_counters = ctx->counters;
    }
    if (p.local_rows == 0) return C2RT_OK;
    const int e = launch_frame(ctx, p, variant_of(ctx, cam), stream);
    if (e != 0) return fail(ctx, C2RT_ERR_HIP, "render kernel launch: %s", hipGetErrorString((hipError_t)e));
    if (opts->count_rays) ctx->counters_valid = true;
    return C2RT_OK;
}

} // namespace

extern "C" {

uint32_t c2rt_abi_version(void) { return C2RT_ABI_VERSION; }

const char *c2rt_status_string(int status)
{
    switch (status) {
    case C2RT_OK: return "ok";
    case C2RT_ERR_INVALID_ARG: return "invalid argument";
    case C2RT_ERR_NO_DEVICE: return "no usable GPU (this library has no CPU fallback)";
    case C2RT_ERR_HIP: return "HIP runtime error";
    case C2RT_ERR_UNSUPPORTED: return "unsupported feature";
    case C2RT_ERR_LIMIT: return "device-path limit exceeded";
    case C2RT_ERR_NO_SCENE: return "no scene uploaded";
    case C2RT_ERR_CANCELLED: return "cancelled";
    case C2RT_ERR_IO: return "I/O error";
    case C2RT_ERR_PARSE: return "parse error";
    default: return "unknown status";
    }
}

const char *c2rt_last_error(const c2rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int c2rt_init(int device, c2rt_ctx **out)
{
    if (!out) return C2RT_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return C2RT_ERR_NO_DEVICE;
    if (device >= count) return C2RT_ERR_NO_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return C2RT_ERR_NO_DEVICE;
    c2rt_ctx *ctx = new c2rt_ctx();
    ctx->device = device;
    *out = ctx; /* handed out even on failure below so that c2rt_last_error works */
    HIP_TRY(ctx, hipSetDevice(device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream2, hipStreamNonBlocking));
    for (int i = 0; i < kMaxChunks; ++i) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->chunk_done[i], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_inflight, hipEventDisableTiming));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->counters), 4 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemset(ctx->counters, 0, 4 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->probe), sizeof(c2rt_trace_result)));
    uint8_t lut[4097];
    build_srgb_lut(lut);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->srgb_lut), sizeof lut));
    HIP_TRY(ctx, hipMemcpy(ctx->srgb_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    return C2RT_OK;
}

int c2rt_init_multi(int device_count_or_0, const int *device_ids, c2rt_ctx **out)
{
    if (!out) return C2RT_ERR_INVALID_ARG;
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return C2RT_ERR_NO_DEVICE;
    if (device_count_or_0 < 0 || device_count_or_0 > 64) return C2RT_ERR_INVALID_ARG;
    const int n = device_count_or_0 == 0 ? visible : device_count_or_0;
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) {
        /* device_count_or_0 == 0 means "every visible device": the caller's list (which may be empty) is not read */
        ids[i] = (device_ids && device_count_or_0 > 0) ? device_ids[i] : i;
        if (ids[i] < 0 || ids[i] >= visible) return C2RT_ERR_NO_DEVICE;
    }
    c2rt_ctx *lead = nullptr;
    int st = c2rt_init(ids[0], &lead);
    *out = lead;
    if (st != C2RT_OK) return st;
    for (int i = 1; i < n; ++i) {
        c2rt_ctx *peer = nullptr;
        st = c2rt_init(ids[i], &peer);
        if (peer) lead->peers.push_back(peer);
        if (st != C2RT_OK) return fail(lead, st, "device slot %d (HIP device %d): %s", i, ids[i], peer ? peer->err.c_str() : "init failed");
        if (ids[i] != ids[0]) {
            /* the slot's kernels store into the lead device's frame (c2rt_render_frame_device) */
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ids[i], ids[0]) != hipSuccess || !can) {
                peer->peer_mapped = false;
            } else {
                const hipError_t e = hipDeviceEnablePeerAccess(ids[0], 0); /* current device = ids[i] (c2rt_init) */
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer->peer_mapped = false;
                (void)hipGetLastError();
            }
        }
    }
    HIP_TRY(lead, hipSetDevice(lead->device));
    return C2RT_OK;
}

int c2rt_device_count(const c2rt_ctx *ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

#if defined(C2RT_TILE_STATS) && C2RT_TILE_STATS
/* Diagnostics hook, in the diagnostics build only (make VARIANT=tilestats EXTRA_HIPFLAGS=-DC2RT_TILE_STATS=1;
 * the product library does not export it and include/c2rt.h does not declare it): the frame kernel writes
 * {wave cycles, class bits} per tile (tiles_x * tiles_y pairs of uint32) to this device buffer.
 * scripts/tile_stats.py. */
void c2rt_debug_set_tile_stats(c2rt_ctx *ctx, uint32_t *dev_buffer)
{
    if (ctx) ctx->tile_stats = dev_buffer;
}
#endif

uint64_t c2rt_scene_generation(const c2rt_ctx *ctx) { return ctx && ctx->has_scene ? ctx->scene_gen : 0; }

void c2rt_destroy(c2rt_ctx *ctx)
{
    if (!ctx) return;
    for (c2rt_ctx *p : ctx->peers) c2rt_destroy(p);
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize(); /* frames may still be in flight on callers' streams */
    if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    if (ctx->ev_inflight) (void)hipEventDestroy(ctx->ev_inflight);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->copy_stream2) { (void)hipStreamSynchronize(ctx->copy_stream2); (void)hipStreamDestroy(ctx->copy_stream2); }
    for (hipEvent_t e : ctx->chunk_done)
        if (e) (void)hipEventDestroy(e);
    for (const auto &pb : ctx->pinned) (void)hipHostUnregister(pb.first);
    void *bufs[] = {ctx->geoms, ctx->nodes, ctx->shaders, ctx->textures, ctx->lights, ctx->texels,
                    ctx->frame, ctx->counters, ctx->probe, ctx->srgb_lut, ctx->shadow_rects};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (FrameScratch &f : ctx->scratch) {
        if (f.tile_masks) (void)hipFree(f.tile_masks);
        if (f.retry_list) (void)hipFree(f.retry_list);
    }
    delete ctx;
}

static int upload_one(c2rt_ctx *ctx, const c2rt_scene_desc *s);

int c2rt_upload_scene(c2rt_ctx *ctx, const c2rt_scene_desc *s)
{
    static std::atomic<uint64_t> next_gen{1};
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    int st = upload_one(ctx, s);
    /* every device slot holds its own copy of the (small) tables and textures */
    for (size_t i = 0; i < ctx->peers.size() && st == C2RT_OK; ++i) {
        st = upload_one(ctx->peers[i], s);
        if (st != C2RT_OK) {
            ctx->has_scene = false;
            return fail(ctx, st, "device slot %zu: %s", i + 1, ctx->peers[i]->err.c_str());
        }
    }
    if (st == C2RT_OK) ctx->scene_gen = next_gen.fetch_add(1);
    if (!ctx->peers.empty()) (void)hipSetDevice(ctx->device);
    return st;
}

static int upload_one(c2rt_ctx *ctx, const c2rt_scene_desc *s)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    if (!s) return fail(ctx, C2RT_ERR_INVALID_ARG, "null scene");
    if (s->abi_version != C2RT_ABI_VERSION)
        return fail(ctx, C2RT_ERR_INVALID_ARG, "scene abi_version %u != %u", s->abi_version, C2RT_ABI_VERSION);
    if (s->gi_enabled) return fail(ctx, C2RT_ERR_UNSUPPORTED, "GIEnabled scenes (path tracing) are outside the hot path");
    if ((s->n_geoms && (!s->geom_type || !s->geom_param || !s->geom_child)) ||
        (s->n_textures && (!s->tex_type || !s->tex_color || !s->tex_param || !s->tex_scaling || !s->tex_width ||
                           !s->tex_height || !s->tex_offset)) ||
        (s->n_shaders && (!s->shader_type || !s->shader_color || !s->shader_texture || !s->shader_exponent ||
                          !s->shader_strength)) ||
        (s->n_lights && (!s->light_type || !s->light_pos || !s->light_color || !s->light_power)) ||
        (s->n_nodes && (!s->node_geom || !s->node_shader || !s->node_transform)) || (s->n_texels && !s->texels))
        return fail(ctx, C2RT_ERR_INVALID_ARG, "null table with non-zero count");
    if (s->n_geoms >= (1u << 23)) return fail(ctx, C2RT_ERR_LIMIT, "too many geometries");

    ctx->has_scene = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* frames enqueued on callers' streams may still be reading the tables about to be replaced */
    HIP_TRY(ctx, hipDeviceSynchronize());

    /* geometries */
    std::vector<DevGeom> geoms(s->n_geoms);
    for (uint32_t g = 0; g < s->n_geoms; ++g) {
        const int t = s->geom_type[g];
        if (t < C2RT_GEOM_PLANE || t > C2RT_GEOM_CSG_DIFF) return fail(ctx, C2RT_ERR_UNSUPPORTED, "geometry %u: unknown type %d", g, t);
        DevGeom &d = geoms[g];
        std::memset(&d, 0, sizeof d);
        d.type = t;
        d.left = is_csg(t) ? s->geom_child[2 * g + 0] : -1;
        d.right = is_csg(t) ? s->geom_child[2 * g + 1] : -1;
        for (int i = 0; i < 4; ++i) d.p[i] = s->geom_param[4 * g + i];
        if (t == C2RT_GEOM_CUBE) {
            const double halfSide = d.p[3] * 0.5;
            for (int i = 0; i < 3; ++i) {
                d.q[i] = d.p[i] + -1 * halfSide;     /* center + side * halfSide, side = -1 (== center - halfSide) */
                d.q[3 + i] = d.p[i] + 1 * halfSide;
            }
        } else if (t == C2RT_GEOM_SPHERE) {
            d.q[0] = d.p[3] * d.p[3];
        }
        bool finite = true;
        for (int i = 0; i < 4; ++i) finite = finite && std::isfinite(d.p[i]);
        for (int i = 0; i < 6; ++i) finite = finite && std::isfinite(d.q[i]);
        if (finite && !is_csg(t)) d.flags |= kGeomFinite;
    }
    std::vector<int> state(s->n_geoms, 0), memo(s->n_geoms, 0);
    std::vector<BoundInfo> bounds(s->n_geoms);
    int levels = 0;

    /* textures: texel pool repacked to float4 */
    std::vector<DevTex> textures(s->n_textures);
    for (uint32_t t = 0; t < s->n_textures; ++t) {
        const int ty = s->tex_type[t];
        if (ty < C2RT_TEX_CHECKER || ty > C2RT_TEX_BITMAP) return fail(ctx, C2RT_ERR_UNSUPPORTED, "texture %u: unknown type %d", t, ty);
        DevTex &d = textures[t];
        std::memset(&d, 0, sizeof d);
        d.type = ty;
        d.scaling = s->tex_scaling[t];
        for (int i = 0; i < 18; ++i) d.color[i] = s->tex_color[18 * t + i];
        for (int i = 0; i < 6; ++i) d.param[i] = s->tex_param[6 * t + i];
        if (ty == C2RT_TEX_BITMAP) {
            d.width = s->tex_width[t];
            d.height = s->tex_height[t];
            d.offset = s->tex_offset[t];
            if ((uint64_t)d.width * d.height + d.offset > s->n_texels)
                return fail(ctx, C2RT_ERR_INVALID_ARG, "texture %u: texels out of the pool", t);
            if (d.width >= (1u << 24) || d.height >= (1u << 24)) return fail(ctx, C2RT_ERR_LIMIT, "texture %u too large", t);
        }
    }
    std::vector<float> texels4((size_t)s->n_texels * 4);
    for (uint64_t i = 0; i < s->n_texels; ++i) {
        texels4[4 * i + 0] = s->texels[3 * i + 0];
        texels4[4 * i + 1] = s->texels[3 * i + 1];
        texels4[4 * i + 2] = s->texels[3 * i + 2];
        texels4[4 * i + 3] = 0.0f;
    }

    /* shaders */
    std::vector<DevShader> shaders(s->n_shaders);
    for (uint32_t i = 0; i < s->n_shaders; ++i) {
        const int ty = s->shader_type[i];
        if (ty != C2RT_SHADER_LAMBERT && ty != C2RT_SHADER_PHONG) return fail(ctx, C2RT_ERR_UNSUPPORTED, "shader %u: unknown type %d", i, ty);
        DevShader &d = shaders[i];
        std::memset(&d, 0, sizeof d);
        d.type = ty;
        d.tex = s->shader_texture[i];
        if (d.tex >= (int32_t)s->n_textures) return fail(ctx, C2RT_ERR_INVALID_ARG, "shader %u: texture index %d out of range", i, d.tex);
        if (d.tex < 0) d.tex = -1;
        for (int c = 0; c < 3; ++c) d.color[c] = s->shader_color[3 * i + c];
        d.strength = s->shader_strength[i];
        d.exponent = s->shader_exponent[i];
    }

    /* lights */
    std::vector<DevLight> lights(s->n_lights);
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        if (s->light_type[i] != C2RT_LIGHT_POINT) return fail(ctx, C2RT_ERR_UNSUPPORTED, "light %u: unknown type %d", i, s->light_type[i]);
        DevLight &d = lights[i];
        std::memset(&d, 0, sizeof d);
        for (int c = 0; c < 3; ++c) d.pos[c] = s->light_pos[3 * i + c];
        /* Light.color(): lightColor * lightPower — rt/light.d:11-14 */
        for (int c = 0; c < 3; ++c) d.color[c] = s->light_color[3 * i + c] * s->light_power[i];
        /* lightColor.intensity() != 0 — rt/shader.d:88, rt/color.d:141-144 */
        const float intensity = (d.color[0] + d.color[1] + d.color[2]) / 3;
        d.lit = intensity != 0 ? 1u : 0u;
        /* bit 1: every channel is +0 or a finite float within 2^+-60 — the numerators of lean::'s fp32 division
         * by the squared distance need no test on the device (c2rt_trace.inc, shade) */
        bool chan_ok = true;
        for (int c = 0; c < 3; ++c) {
            uint32_t bits;
            std::memcpy(&bits, &d.color[c], 4);
            const float a = std::fabs(d.color[c]);
            chan_ok = chan_ok && (bits == 0u || (a >= 0x1p-60f && a < 0x1p60f));
        }
        if (chan_ok) d.lit |= 2u;
    }

    /* nodes */
    std::vector<DevNode> nodes(s->n_nodes);
    for (uint32_t n = 0; n < s->n_nodes; ++n) {
        DevNode &d = nodes[n];
        std::memset(&d, 0, sizeof d);
        d.geom = s->node_geom[n];
        d.shader = s->node_shader[n];
        if (d.shader < 0 || (uint32_t)d.shader >= s->n_shaders) return fail(ctx, C2RT_ERR_INVALID_ARG, "node %u: shader index %d out of range", n, d.shader);
        const int depth = csg_depth(s, d.geom, state, memo);
        if (depth < 0) return fail(ctx, C2RT_ERR_INVALID_ARG, "node %u: geometry index out of range or cyclic CSG", n);
        if (depth > C2RT_MAX_CSG_DEPTH) return fail(ctx, C2RT_ERR_LIMIT, "node %u: CSG nesting %d > %d", n, depth, C2RT_MAX_CSG_DEPTH);
        if (depth > levels) levels = depth;
        bound_of(s, d.geom, bounds, geoms);
        d.g = geoms[d.geom]; /* after bound_of: carries the flags and the bound */
        const double *t = s->node_transform + 30 * (size_t)n;
        std::memcpy(d.m, t, 9 * sizeof(double));
        std::memcpy(d.inv, t + 9, 9 * sizeof(double));
        std::memcpy(d.tinv, t + 18, 9 * sizeof(double));
        std::memcpy(d.off, t + 27, 3 * sizeof(double));
        static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        bool ident = true;
        for (int i = 0; i < 9; ++i) ident = ident && d.m[i] == I[i] && d.inv[i] == I[i] && d.tinv[i] == I[i];
        if (ident) d.flags |= kNodeIdentityMatrix;
        if (d.off[0] == 0 && d.off[1] == 0 && d.off[2] == 0) d.flags |= kNodeZeroOffset;
        if (d.g.type == C2RT_GEOM_PLANE && !ident) {
            /* hit_surface: normalized(mulvm((0, 1, 0), tinv)), operation for operation (this file is built
             * without contraction; sqrt and division are IEEE on both sides) */
            const double nx = 0.0, ny = 1.0, nz = 0.0;
            const double *m = d.tinv;
            const double vx = nx * m[0] + ny * m[3] + nz * m[6];
            const double vy = nx * m[1] + ny * m[4] + nz * m[7];
            const double vz = nx * m[2] + ny * m[5] + nz * m[8];
            const double sq = vx * vx + vy * vy + vz * vz;
            const double len = std::sqrt(sq);
            const double inv = 1.0 / len;
            const double wn[3] = {vx * inv, vy * inv, vz * inv};
            if (std::isfinite(wn[0]) && std::isfinite(wn[1]) && std::isfinite(wn[2])) {
                d.g.q[0] = wn[0];
                d.g.q[1] = wn[1];
                d.g.q[2] = wn[2];
                d.flags |= kNodePlaneNormal;
            }
        }
        /* shading inputs of this node in one record */
        const DevShader &sh = shaders[d.shader];
        DevMat &m = d.mat;
        m.shader_type = sh.type;
        m.tex = sh.tex;
        m.tex_type = sh.tex >= 0 ? textures[sh.tex].type : -1;
        m.strength = sh.strength;
        std::memcpy(m.color, sh.color, sizeof m.color);
        m.exponent = sh.exponent;
        if (m.tex_type == C2RT_TEX_CHECKER) {
            std::memcpy(m.texdata, textures[sh.tex].color, 6 * sizeof(float));
            std::memcpy(m.texdata + 6, &textures[sh.tex].param[0], sizeof(double));
        } else if (m.tex_type == C2RT_TEX_BITMAP) {
            const DevTex &t = textures[sh.tex];
            m.texdata[0] = t.width;
            m.texdata[1] = t.height;
            std::memcpy(m.texdata + 2, &t.scaling, sizeof(float));
            std::memcpy(m.texdata + 4, &t.offset, sizeof(uint64_t));
        }
    }

    if (levels > 0 && s->n_geoms > C2RT_MAX_CSG_GEOMS)
        return fail(ctx, C2RT_ERR_LIMIT, "%u geometries in a scene with CsgOps (limit %d)", s->n_geoms, C2RT_MAX_CSG_GEOMS);
    ctx->planes_only = s->n_nodes > 0;
    for (uint32_t n = 0; n < s->n_nodes; ++n) {
        DevNode &d = nodes[n];
        bool axis = d.g.type == C2RT_GEOM_PLANE;
        if (axis && !(d.flags & kNodeIdentityMatrix)) {
            for (int i = 0; i < 9; ++i) {
                const double a = std::fabs(d.inv[i]);
                axis = axis && (i % 4 == 0 ? (a >= 1e-100 && a <= 1e100) : d.inv[i] == 0.0);
            }
            axis = axis && d.inv[4] > 0;
        }
        if (axis) d.flags |= kNodeAxisPlane;
        else ctx->planes_only = 0;
    }
    ctx->all_identity = s->n_nodes > 0;
    for (uint32_t n = 0; n < s->n_nodes; ++n)
        if (!(nodes[n].flags & kNodeIdentityMatrix)) ctx->all_identity = 0;

    /* world-space bounding boxes of the nodes: object-space box (box_of), padded -> the 8
     * corners through Transform.point (affine: hull preserved) */
    ctx->node_box.assign((size_t)s->n_nodes * 24, 0.0);
    ctx->node_boxed.assign(s->n_nodes, 0);
    std::vector<BoxInfo> boxes(s->n_geoms);
    for (uint32_t n = 0; n < s->n_nodes; ++n) {
        const DevGeom &g = nodes[n].g;
        if (!(g.flags & kGeomBounded)) continue;
        const BoxInfo bx = box_of(s, nodes[n].geom, boxes, geoms);
        if (!bx.bounded) continue;
        /* a singular / non-finite transform (e.g. `scale 0 0 0`) sends NaN rays into the
         * geometry, whose hits follow no geometric bound: never cull such a node */
        bool sane = true;
        for (int i = 0; i < 9; ++i) sane = sane && std::isfinite(nodes[n].m[i]) && std::isfinite(nodes[n].inv[i]) && std::isfinite(nodes[n].tinv[i]);
        for (int i = 0; i < 3; ++i) sane = sane && std::isfinite(nodes[n].off[i]);
        if (!sane) continue;
        /* pads: the same relative pad as the bounding sphere (the tests run in fp64 on coordinates
         * of this magnitude); shadow rays start 1e-6 (world units) off the surface (rt/shader.d:88):
         * 4e-6 world units = 4e-6 * |M^-1|_F object units (|M^-1|_F >= 1 / smallest scale) */
        double inv_norm = 0;
        for (int i = 0; i < 9; ++i) inv_norm += nodes[n].inv[i] * nodes[n].inv[i];
        inv_norm = std::sqrt(inv_norm);
        double mag = 0, ext = 0;
        for (int i = 0; i < 3; ++i) {
            mag += std::fmax(std::fabs(bx.lo[i]), std::fabs(bx.hi[i]));
            ext = std::fmax(ext, bx.hi[i] - bx.lo[i]);
        }
        const double pad = 1e-6 * ext + 1e-6 * mag + 1e-9 + 4e-6 * (inv_norm > 1 ? inv_norm : 1.0);
        bool finite = std::isfinite(pad);
        for (int k = 0; k < 8 && finite; ++k) {
            const double q[3] = {(k & 1) ? bx.hi[0] + pad : bx.lo[0] - pad, (k & 2) ? bx.hi[1] + pad : bx.lo[1] - pad,
                                 (k & 4) ? bx.hi[2] + pad : bx.lo[2] - pad};
            double *w = &ctx->node_box[((size_t)n * 8 + k) * 3];
            for (int j = 0; j < 3; ++j) {
                w[j] = q[0] * nodes[n].m[0 + j] + q[1] * nodes[n].m[3 + j] + q[2] * nodes[n].m[6 + j] + nodes[n].off[j];
                finite = finite && std::isfinite(w[j]);
            }
        }
        ctx->node_boxed[n] = finite ? 1 : 0;
    }

    ctx->light_pos.assign(s->light_pos, s->light_pos + 3 * (size_t)s->n_lights);

    /* Ground-plane shadow culling towards light 0 (RenderParams::ground_node): the first Plane node
     * under an identity matrix with zero offset is the ground; every boxed node gets the rectangle
     * (in the plane's x, z) of its padded world box projected from the light onto the plane.
     * Let Q be a point of the box on a shadow segment from P' = P + N*1e-6 (P on the plane) to the
     * light L: L, Q and P' are collinear, so the central projection of Q from L onto the plane is
     * the point where the line L-P' meets it — within 1e-6 * (horizontal / vertical extent of the
     * segment) of P.  Hence P lies in the projected box grown by that much; the rectangle is padded
     * by 1e-5 * (1 + slope) + 1e-9 * scale, far above rounding in P.  Defined only when the light is
     * above the plane and the whole box lies strictly between plane and light (or the mirror image
     * below the plane); otherwise the rectangle is everything. */
    {
        std::vector<double> rects((size_t)kMaxCullNodes * 4);
        for (int n = 0; n < kMaxCullNodes; ++n) {
            rects[4 * n + 0] = rects[4 * n + 2] = -HUGE_VAL;
            rects[4 * n + 1] = rects[4 * n + 3] = HUGE_VAL;
        }
        ctx->ground_node = -1;
        for (uint32_t n = 0; n < s->n_nodes && n < (uint32_t)kMaxCullNodes; ++n) {
            const DevNode &d = nodes[n];
            if (d.g.type == C2RT_GEOM_PLANE && (d.flags & kNodeIdentityMatrix) && (d.flags & kNodeZeroOffset) && std::isfinite(d.g.p[0])) {
                ctx->ground_node = (int32_t)n;
                ctx->ground_y = d.g.p[0];
                break;
            }
        }
        if (ctx->ground_node >= 0 && s->n_lights > 0) {
            const double *L = s->light_pos;
            const double y0 = ctx->ground_y, h = L[1] - y0; /* light height over the plane (signed) */
            for (uint32_t n = 0; n < s->n_nodes && n < (uint32_t)kMaxCullNodes; ++n) {
                if (!ctx->node_boxed[n] || !std::isfinite(h) || h == 0) continue;
                /* axis-aligned hull of the node's (possibly sheared) world box */
                double bmin[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, bmax[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
                for (int k = 0; k < 8; ++k)
                    for (int j = 0; j < 3; ++j) {
                        const double v = ctx->node_box[((size_t)n * 8 + k) * 3 + j];
                        bmin[j] = std::min(bmin[j], v);
                        bmax[j] = std::max(bmax[j], v);
                    }
                /* heights as t = (y - y0) / h: 0 on the plane, 1 at the light's height.  Shadow segments
                 * start within 1e-6 of the plane and end at the light: what the box has beyond the plane
                 * (t < 0) is out of their reach, so it is clipped there (with slack) */
                double t_lo = (bmin[1] - y0) / h, t_hi = (bmax[1] - y0) / h;
                if (t_lo > t_hi) std::swap(t_lo, t_hi);
                const double slack = 1e-5 * (1.0 + std::fabs(y0)) / std::fabs(h);
                t_lo = std::max(t_lo, -slack);
                if (!(t_hi < 1 - 1e-9) || !(t_hi >= t_lo)) continue; /* reaches the light's height, or wholly beyond the plane: no rectangle */
                double lo[2] = {HUGE_VAL, HUGE_VAL}, hi[2] = {-HUGE_VAL, -HUGE_VAL};
                double scale = std::fabs(y0) + std::fabs(L[0]) + std::fabs(L[1]) + std::fabs(L[2]);
                bool ok = true;
                for (int k = 0; k < 8 && ok; ++k) {
                    const double wx = (k & 1) ? bmax[0] : bmin[0], wz = (k & 2) ? bmax[2] : bmin[2], t = (k & 4) ? t_hi : t_lo;
                    const double sfac = 1.0 / (1.0 - t);         /* L + (w - L) * sfac lies on the plane */
                    const double px = L[0] + (wx - L[0]) * sfac, pz = L[2] + (wz - L[2]) * sfac;
                    ok = std::isfinite(px) && std::isfinite(pz);
                    lo[0] = std::min(lo[0], px); hi[0] = std::max(hi[0], px);
                    lo[1] = std::min(lo[1], pz); hi[1] = std::max(hi[1], pz);
                    scale = std::max(scale, std::fabs(px) + std::fabs(pz));
                }
                if (!ok) continue;
                /* slope of the steepest-sideways shadow segment that can end in the rectangle */
                const double dx = std::max(std::fabs(lo[0] - L[0]), std::fabs(hi[0] - L[0]));
                const double dz = std::max(std::fabs(lo[1] - L[2]), std::fabs(hi[1] - L[2]));
                const double slope = std::sqrt(dx * dx + dz * dz) / std::fabs(h);
                const double pad = 1e-5 * (1.0 + slope) + 1e-9 * scale;
                if (!std::isfinite(pad)) continue;
                rects[4 * n + 0] = lo[0] - pad; rects[4 * n + 1] = hi[0] + pad;
                rects[4 * n + 2] = lo[1] - pad; rects[4 * n + 3] = hi[1] + pad;
            }
        } else {
            ctx->ground_node = -1;
        }
        int st0;
        if ((st0 = upload(ctx, &ctx->shadow_rects, rects)) != C2RT_OK) return st0;
    }

    int st;
    if ((st = upload(ctx, &ctx->geoms, geoms)) != C2RT_OK) return st;
    if ((st = upload(ctx, &ctx->textures, textures)) != C2RT_OK) return st;
    if ((st = upload(ctx, &ctx->texels, texels4)) != C2RT_OK) return st;
    if ((st = upload(ctx, &ctx->shaders, shaders)) != C2RT_OK) return st;
    if ((st = upload(ctx, &ctx->lights, lights)) != C2RT_OK) return st;
    if ((st = upload(ctx, &ctx->nodes, nodes)) != C2RT_OK) return st;
    ctx->n_nodes = s->n_nodes;
    ctx->n_lights = s->n_lights;
    std::memcpy(ctx->ambient, s->ambient, sizeof ctx->ambient);
    ctx->max_trace_depth = s->max_trace_depth;
    ctx->csg_levels = levels;
    ctx->has_scene = true;
    ctx->err.clear();
    return C2RT_OK;
}

/* Host-output frames (c2rt_render_frame: float RGB; c2rt_render_frame_rgb32: Color.toRGB32 words).
 * Into a buffer page-locked with c2rt_pin_host_buffer the frame is rendered in up to 8 row chunks:
 * chunk i streams back over PCIe (copy stream) while chunk i+1 renders (and is encoded), and the
 * stop flag is polled between chunks (finer than the reference's between-pass polling).  Into
 * pageable memory: one launch, one copy — chunked copies into pageable memory are slower than one
 * (measured).  ctx->frame (ensure_staging) holds the float rows OR the display words, whichever is asked for;
 * a frame the kernel stores straight into the page-locked destination has no staging buffer at all. */
/* Host-output pipeline parameters (defaults measured on MI355X, profiles/r03_variants.md); the environment
 * variables exist for that measurement, in the diagnostics build only: C2RT_HOST_CHUNK_MB, C2RT_HOST_FIRST_FRAC,
 * C2RT_HOST_COPY_STREAMS, C2RT_HOST_DIRECT_STORE. */
struct HostKnobs {
    size_t chunk_bytes = 13u << 20; /* 4K float frame (99.5 MB), ms by chunk count: 10 chunks 2.12, 9 2.04, 8 (this) 1.96, 7 1.98, 6 1.98, 5 2.02, 4 2.19 (copy alone: 1.75) */
    double first_frac = 1.0;        /* a smaller first chunk: no gain (the pipeline is copy-bound from the first copy on) */
    int copy_streams = 1;           /* two copy streams: no gain */
    /* kernel stores straight into the page-locked frame: 0 never, 1 the display-word frame (4 B/pixel: 1.30 ms
     * against 1.60 chunked; the float frame's 12-byte stores reach only 44 GB/s: 2.27 against 2.02), 2 both */
    int direct_store = 1;
};
static const HostKnobs &host_knobs()
{
    static const HostKnobs k = [] {
        HostKnobs v;
        if (const char *e = diag_env("C2RT_HOST_CHUNK_MB")) { const double mb = std::atof(e); if (mb >= 0.25 && mb <= 1024) v.chunk_bytes = (size_t)(mb * (1u << 20)); }
        if (const char *e = diag_env("C2RT_HOST_FIRST_FRAC")) { const double f = std::atof(e); if (f > 0 && f <= 1) v.first_frac = f; }
        if (const char *e = diag_env("C2RT_HOST_COPY_STREAMS")) v.copy_streams = std::atoi(e) > 1 ? 2 : 1;
        if (const char *e = diag_env("C2RT_HOST_DIRECT_STORE")) v.direct_store = std::atoi(e);
        return v;
    }();
    return k;
}

static int ensure_staging(c2rt_ctx *c, size_t bytes)
{
    const size_t floats = (bytes + sizeof(float) - 1) / sizeof(float);
    if (floats > c->frame_floats) {
        if (c->frame) { (void)hipFree(c->frame); c->frame = nullptr; c->frame_floats = 0; }
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->frame), floats * sizeof(float)));
        c->frame_floats = floats;
    }
    return C2RT_OK;
}

/* host wait for the last COUNTED stream-async frame (the ray counters are shared by all streams; everything else a
 * frame writes is per stream, FrameScratch) */
static int drain_inflight(c2rt_ctx *ctx)
{
    if (!ctx->has_inflight) return C2RT_OK;
    /* cleared whatever the wait returns: an error of the earlier frame is reported ONCE, here, and the context
     * stays usable (round-3 advisor: a failed sync used to leave has_inflight set for good) */
    ctx->has_inflight = false;
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_inflight));
    return C2RT_OK;
}

static int render_to_host(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, float *out_rgb,
                          uint32_t *out_rgb32, const volatile uint8_t *stop_flag)
{
    if (opts->count_rays)
        if (const int st = drain_inflight(ctx)) return st;
    RenderParams p;
    fill_params(ctx, cam, opts, p);
    ctx->counters_valid = false;
    if (opts->count_rays) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, 3 * sizeof(unsigned long long), ctx->stream));
        p.ray_counters = ctx->counters;
    }
    const uint32_t rows = p.local_rows;
    const size_t row_px = opts->width;
    /* the display frame leaves the render kernel encoded (RenderParams::out_rgb32): 4 B per pixel in the
     * staging buffer, no float frame, no second kernel */
    const size_t px_bytes = out_rgb ? 3 * sizeof(float) : sizeof(uint32_t);
    if (!out_rgb) {
        p.out = nullptr;
        p.srgb_lut = ctx->srgb_lut;
    }
    char *dst = out_rgb ? reinterpret_cast<char *>(out_rgb) : reinterpret_cast<char *>(out_rgb32);
    const size_t dst_row_bytes = row_px * px_bytes;
    bool is_pinned = false;
    for (const auto &pb : ctx->pinned)
        is_pinned = is_pinned || (dst >= reinterpret_cast<char *>(pb.first) &&
                                  dst + (size_t)rows * dst_row_bytes <= reinterpret_cast<char *>(pb.first) + pb.second);
    const KernelVariant variant = variant_of(ctx, cam);
    const HostKnobs &knobs = host_knobs();
    if (is_pinned && knobs.direct_store >= (out_rgb ? 2 : 1)) {
        /* the kernel stores straight into the page-locked host frame over PCIe while it renders: one launch,
         * no staging buffer; the stop flag is polled once, before the launch (include/c2rt.h) */
        void *mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, dst, 0) == hipSuccess && mapped) {
            if (out_rgb) p.out = static_cast<float *>(mapped); else p.out_rgb32 = static_cast<uint32_t *>(mapped);
            if (stop_flag && *stop_flag) return fail(ctx, C2RT_ERR_CANCELLED, "stop requested during the frame");
            const int e = launch_frame(ctx, p, variant, ctx->stream);
            if (e != 0) return fail(ctx, C2RT_ERR_HIP, "render kernel launch: %s", hipGetErrorString((hipError_t)e));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (opts->count_rays) ctx->counters_valid = true;
            return C2RT_OK;
        }
        (void)hipGetLastError();
    }
    /* staging frame on the device: 12 B per pixel of float rows, or 4 B per pixel of display words */
    if (const int st = ensure_staging(ctx, (size_t)rows * row_px * px_bytes)) return st;
    char *staging = reinterpret_cast<char *>(ctx->frame);
    if (out_rgb) p.out = ctx->frame; else p.out_rgb32 = reinterpret_cast<uint32_t *>(ctx->frame);
    /* Into a page-locked frame: row chunks, chunk i crossing PCIe on a copy stream while chunk i+1 renders.
     * The copy (99.5 MB of float frame at ~57 GB/s = 1.75 ms at 4K) is longer than the render (1.15 ms), so
     * the frame time is the first chunk's render + the copies back to back + whatever keeps them from being
     * back to back: a SMALL first chunk, then equal ones; two copy streams so that one copy's setup hides
     * behind the other's transfer.  Into pageable memory: one launch, one copy (chunked copies into pageable
     * memory are slower than one, measured).  The stop flag is polled between chunks. */
    const size_t total_bytes = (size_t)rows * dst_row_bytes;
    uint32_t want = (uint32_t)((total_bytes + knobs.chunk_bytes - 1) / knobs.chunk_bytes);
    if (want < 1) want = 1;
    if (want > (uint32_t)kMaxChunks - 1) want = kMaxChunks - 1;
    uint32_t chunk = is_pinned ? ((rows + want - 1) / want + kTileH - 1) / kTileH * kTileH : rows;
    if (chunk < 64) chunk = 64;
    uint32_t first = chunk;
    if (is_pinned && want > 1) {
        first = (uint32_t)(chunk * knobs.first_frac + kTileH - 1) / kTileH * kTileH;
        if (first < 64) first = 64;
        if (first > chunk) first = chunk;
    }
    bool cancelled = false;
    int n_chunks = 0;
    { /* one mask table for the whole frame: every chunk's launch reads its rows of it */
        const int e = prepare_tile_masks(ctx, p, variant, ctx->stream);
        if (e != 0) return fail(ctx, C2RT_ERR_HIP, "tile-mask pre-pass launch: %s", hipGetErrorString((hipError_t)e));
    }
    for (uint32_t off = 0; off < rows && n_chunks < kMaxChunks; ++n_chunks) {
        if (stop_flag && *stop_flag) { cancelled = true; break; }
        uint32_t n = n_chunks == 0 ? first : chunk;
        if (n > rows - off || n_chunks == kMaxChunks - 1) n = rows - off;
        p.row_offset = off;
        p.local_rows = n;
        p.tiles_y = (n + kTileH - 1) / kTileH;
        if (n < rows) p.row_group_start = 0; /* the rotation is relative to the whole frame's rows */
        const int e = launch_frame(ctx, p, variant, ctx->stream);
        if (e != 0) return fail(ctx, C2RT_ERR_HIP, "render kernel launch: %s", hipGetErrorString((hipError_t)e));
        hipStream_t cs = (knobs.copy_streams > 1 && (n_chunks & 1)) ? ctx->copy_stream2 : ctx->copy_stream;
        HIP_TRY(ctx, hipEventRecord(ctx->chunk_done[n_chunks], ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->chunk_done[n_chunks], 0));
        HIP_TRY(ctx, hipMemcpyAsync(dst + (size_t)off * dst_row_bytes, staging + (size_t)off * dst_row_bytes, (size_t)n * dst_row_bytes,
                                    hipMemcpyDeviceToHost, cs));
        off += n;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream2));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (cancelled) return fail(ctx, C2RT_ERR_CANCELLED, "stop requested during the frame");
    if (opts->count_rays) ctx->counters_valid = true;
    return C2RT_OK;
}


/* Interleaved strips of a multi-device context: kTileH rows (one tile row) — the finest deal the
 * kernel's tiling allows, which balances the sky / floor / object mix best. */
constexpr uint32_t kMultiStrip = kTileH;

/* Host-output frame of a MULTI-DEVICE context (c2rt_init_multi): slot d renders strips d, d+G, ...
 * into its own staging buffer and copies them straight into the caller's frame with ONE strided 2-D
 * copy (row = one strip, destination pitch = G strips) over its own PCIe link — every device's copy
 * engine works in parallel and nothing is assembled on a device.  (RGB32: each slot encodes first.) */
static int render_to_host_multi(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, float *out_rgb,
                                uint32_t *out_rgb32, const volatile uint8_t *stop_flag)
{
    if (opts->count_rays)
        if (const int st = drain_inflight(ctx)) return st;
    const uint32_t G = 1u + (uint32_t)ctx->peers.size();
    const uint32_t sh = kMultiStrip, H = opts->height, W = opts->width;
    const uint32_t n_strips = (H + sh - 1) / sh, rem = H % sh; /* rem > 0: the last strip is partial */
    char *dst = out_rgb ? reinterpret_cast<char *>(out_rgb) : reinterpret_cast<char *>(out_rgb32);
    const size_t px_bytes = out_rgb ? 3 * sizeof(float) : sizeof(uint32_t);
    const size_t strip_bytes = (size_t)sh * W * px_bytes;
    const KernelVariant variant = variant_of(ctx, cam);
    ctx->counters_valid = false;
    int st = C2RT_OK;
    bool cancelled = false;
    uint32_t launched = 0;
    for (uint32_t d = 0; d < G && st == C2RT_OK; ++d) {
        c2rt_ctx *c = d == 0 ? ctx : ctx->peers[d - 1];
        if (stop_flag && *stop_flag) { cancelled = true; break; }
        launched = d + 1;
        c2rt_render_opts o = *opts;
        o.strip_height = sh;
        o.strip_rank = d;
        o.strip_world = G;
        RenderParams p;
        fill_params(c, cam, &o, p);
        const uint32_t rows = p.local_rows;
        if (rows == 0) continue;
        if (hipSetDevice(c->device) != hipSuccess) { st = fail(ctx, C2RT_ERR_HIP, "hipSetDevice(%d)", c->device); break; }
        const size_t px = (size_t)rows * W;
        if ((st = ensure_staging(c, px * px_bytes)) != C2RT_OK) { st = fail(ctx, st, "slot %u: %s", d, c->err.c_str()); break; }
        if (out_rgb) {
            p.out = c->frame;
        } else { /* display words straight out of the render kernel (RenderParams::out_rgb32) */
            p.out = nullptr;
            p.out_rgb32 = reinterpret_cast<uint32_t *>(c->frame);
            p.srgb_lut = c->srgb_lut;
        }
        if (opts->count_rays) {
            if (hipMemsetAsync(c->counters, 0, 3 * sizeof(unsigned long long), c->stream) != hipSuccess) { st = fail(ctx, C2RT_ERR_HIP, "counter reset"); break; }
            p.ray_counters = c->counters;
        }
        const int e = launch_frame(c, p, variant, c->stream);
        if (e != 0) { st = fail(ctx, C2RT_ERR_HIP, "render kernel launch (slot %u): %s", d, hipGetErrorString((hipError_t)e)); break; }
        const char *src = reinterpret_cast<const char *>(c->frame);
        /* this slot's strips: d, d+G, ...; all full except possibly the frame's last strip */
        const uint32_t mine = (n_strips - d + G - 1) / G;
        const bool owns_partial = rem != 0 && (n_strips - 1) % G == d;
        const uint32_t full = owns_partial ? mine - 1 : mine;
        hipError_t he = hipSuccess;
        if (full)
            he = hipMemcpy2DAsync(dst + (size_t)d * strip_bytes, (size_t)G * strip_bytes, src, strip_bytes, strip_bytes, full,
                                  hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess && owns_partial)
            he = hipMemcpyAsync(dst + (size_t)(n_strips - 1) * strip_bytes, src + (size_t)full * strip_bytes, (size_t)rem * W * px_bytes,
                                hipMemcpyDeviceToHost, c->stream);
        if (he != hipSuccess) st = fail(ctx, C2RT_ERR_HIP, "strip copy (slot %u): %s", d, hipGetErrorString(he));
    }
    for (uint32_t d = 0; d < launched; ++d) {
        c2rt_ctx *c = d == 0 ? ctx : ctx->peers[d - 1];
        (void)hipSetDevice(c->device);
        const hipError_t he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess && st == C2RT_OK) st = fail(ctx, C2RT_ERR_HIP, "slot %u: %s", d, hipGetErrorString(he));
    }
    (void)hipSetDevice(ctx->device);
    if (st != C2RT_OK) return st;
    if (cancelled) return fail(ctx, C2RT_ERR_CANCELLED, "stop requested during the frame");
    if (opts->count_rays) ctx->counters_valid = true;
    return C2RT_OK;
}

/* Device-output frame of a multi-device context: every slot's kernel stores its strips straight
 * into `out_dev` on the lead device (RenderParams::frame_rows; peers reach it over xGMI through
 * peer access).  Ordered on the caller's stream: the peers start after what that stream has queued
 * so far (they overwrite the frame) and the stream continues after the last peer kernel. */
static int render_device_multi(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, float *out_dev,
                               hipStream_t stream)
{
    const uint32_t G = 1u + (uint32_t)ctx->peers.size();
    for (c2rt_ctx *c : ctx->peers)
        if (!c->peer_mapped)
            return fail(ctx, C2RT_ERR_UNSUPPORTED, "HIP device %d cannot map device %d's memory: use the host-output entry points", c->device, ctx->device);
    const KernelVariant variant = variant_of(ctx, cam);
    ctx->counters_valid = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_ready, stream));
    /* A failure half way through must not leave a peer as the current device, nor peers that were already
     * launched still storing into `out_dev` unordered against the caller's stream: whatever happens, the lead
     * device is current again and `stream` waits for the ev_done of every peer launched so far. */
    uint32_t launched = 0; /* peers whose ev_done has been recorded */
    int st = C2RT_OK;
    const auto step = [&](hipError_t e, const char *what, uint32_t d) {
        if (e != hipSuccess && st == C2RT_OK) st = fail(ctx, C2RT_ERR_HIP, "%s (slot %u): %s", what, d, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (uint32_t d = 0; d < G && st == C2RT_OK; ++d) {
        c2rt_ctx *c = d == 0 ? ctx : ctx->peers[d - 1];
        c2rt_render_opts o = *opts;
        o.strip_height = kMultiStrip;
        o.strip_rank = d;
        o.strip_world = G;
        RenderParams p;
        fill_params(c, cam, &o, p);
        p.out = out_dev;
        p.frame_rows = 1;
        if (!step(hipSetDevice(c->device), "hipSetDevice", d)) break;
        hipStream_t s = d == 0 ? stream : c->stream;
        if (d != 0 && !step(hipStreamWaitEvent(s, ctx->ev_ready, 0), "hipStreamWaitEvent", d)) break;
        if (opts->count_rays) {
            if (!step(hipMemsetAsync(c->counters, 0, 3 * sizeof(unsigned long long), s), "counter reset", d)) break;
            p.ray_counters = c->counters;
        }
        if (p.local_rows) {
            const int e = launch_frame(c, p, variant, s);
            if (e != 0) { step((hipError_t)e, "render kernel launch", d); /* fall through: order what was queued */ }
        }
        if (d != 0) {
            if (!step(hipEventRecord(c->ev_done, s), "hipEventRecord", d)) break;
            launched = d;
        }
    }
    (void)hipSetDevice(ctx->device);
    for (uint32_t d = 1; d <= launched; ++d) {
        const hipError_t e = hipStreamWaitEvent(stream, ctx->peers[d - 1]->ev_done, 0);
        if (e != hipSuccess && st == C2RT_OK) st = fail(ctx, C2RT_ERR_HIP, "hipStreamWaitEvent (slot %u): %s", d, hipGetErrorString(e));
    }
    if (st != C2RT_OK) return st;
    if (opts->count_rays) ctx->counters_valid = true;
    return C2RT_OK;
}

static int check_multi_opts(c2rt_ctx *ctx, const c2rt_render_opts *opts)
{
    if (!ctx->peers.empty() && opts->strip_world > 1)
        return fail(ctx, C2RT_ERR_INVALID_ARG, "a multi-device context shards the frame itself: strip_world must be <= 1");
    return C2RT_OK;
}

uint32_t c2rt_local_rows(const c2rt_render_opts *opts)
{
    if (!opts) return 0;
    if (opts->strip_world > 1 && opts->strip_rank >= opts->strip_world) return 0;
    return local_rows_of(opts, opts->strip_world > 1 ? opts->strip_rank : 0);
}

int c2rt_render_frame_device(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts,
                             float *out_rgb_dev, void *hip_stream)
{
    int st = check_frame_args(ctx, cam, opts);
    if (st != C2RT_OK) return st;
    if (!out_rgb_dev) return fail(ctx, C2RT_ERR_INVALID_ARG, "null output");
    if ((st = check_multi_opts(ctx, opts)) != C2RT_OK) return st;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* Frames on one stream run in order; frames of this context on OTHER streams use other scratch slots
     * (FrameScratch) and are independent of this one.  Only a counted frame shares something — the ray counters —
     * and is ordered behind the previous counted frame on the device; the call never waits on the host. */
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (opts->count_rays && ctx->has_inflight) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_inflight, 0));
    st = !ctx->peers.empty() ? render_device_multi(ctx, cam, opts, out_rgb_dev, stream)
                             : render_device(ctx, cam, opts, out_rgb_dev, stream);
    if (opts->count_rays) {
        /* also after a failed launch: whatever did get queued is ordered before the next counted frame */
        const hipError_t rec = hipEventRecord(ctx->ev_inflight, stream);
        if (rec == hipSuccess) ctx->has_inflight = true;
        else if (st == C2RT_OK) st = fail(ctx, C2RT_ERR_HIP, "hipEventRecord(ev_inflight): %s", hipGetErrorString(rec));
    }
    return st;
}

int c2rt_render_frame(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, float *out_rgb,
                      const volatile uint8_t *stop_flag)
{
    int st = check_frame_args(ctx, cam, opts);
    if (st != C2RT_OK) return st;
    if (!out_rgb) return fail(ctx, C2RT_ERR_INVALID_ARG, "null output");
    /* isStopReq() before the pass — rt/renderer.d:129 */
    if (stop_flag && *stop_flag) return fail(ctx, C2RT_ERR_CANCELLED, "stop requested before the frame");
    if ((st = check_multi_opts(ctx, opts)) != C2RT_OK) return st;
    if (!ctx->peers.empty()) return render_to_host_multi(ctx, cam, opts, out_rgb, nullptr, stop_flag);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return render_to_host(ctx, cam, opts, out_rgb, nullptr, stop_flag);
}

/* nothing of this context (any stream of any device slot) may still be copying into a buffer whose pages are
 * about to be unlocked */
static int quiesce(c2rt_ctx *ctx)
{
    ctx->has_inflight = false;
    HIP_TRY(ctx, hipDeviceSynchronize()); /* the context's own streams and whatever callers' streams hold */
    for (c2rt_ctx *c : ctx->peers) {
        HIP_TRY(ctx, hipSetDevice(c->device));
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    if (!ctx->peers.empty()) HIP_TRY(ctx, hipSetDevice(ctx->device));
    return C2RT_OK;
}

int c2rt_pin_host_buffer(c2rt_ctx *ctx, float *out_rgb, size_t bytes)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    if (!out_rgb || bytes == 0) return fail(ctx, C2RT_ERR_INVALID_ARG, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (ctx->pinned[i].first == out_rgb) {
            if (ctx->pinned[i].second == bytes) return C2RT_OK;
            /* the same address with another size (a re-allocated frame buffer): register afresh, so that
             * the recorded range is exactly what is page-locked */
            if (const int st = quiesce(ctx)) return st;
            HIP_TRY(ctx, hipHostUnregister(out_rgb));
            ctx->pinned.erase(ctx->pinned.begin() + (long)i);
            break;
        }
    /* portable: page-locked for every device slot of a multi-device context */
    HIP_TRY(ctx, hipHostRegister(out_rgb, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    ctx->pinned.emplace_back(out_rgb, bytes);
    return C2RT_OK;
}

int c2rt_unpin_host_buffer(c2rt_ctx *ctx, float *out_rgb)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (ctx->pinned[i].first == out_rgb) {
            if (const int st = quiesce(ctx)) return st;
            HIP_TRY(ctx, hipHostUnregister(out_rgb));
            ctx->pinned.erase(ctx->pinned.begin() + (long)i);
            return C2RT_OK;
        }
    return fail(ctx, C2RT_ERR_INVALID_ARG, "buffer was not pinned through this context");
}

int c2rt_get_ray_stats(c2rt_ctx *ctx, c2rt_ray_stats *out)
{
    if (!ctx || !out) return C2RT_ERR_INVALID_ARG;
    if (!ctx->counters_valid) return fail(ctx, C2RT_ERR_INVALID_ARG, "last render did not count rays");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (const int st = drain_inflight(ctx)) return st; /* a counted device-output frame may still be running */
    unsigned long long h[2];
    HIP_TRY(ctx, hipMemcpy(h, ctx->counters, sizeof h, hipMemcpyDeviceToHost));
    out->primary_rays = h[0];
    out->shadow_rays = h[1];
    for (c2rt_ctx *c : ctx->peers) { /* every slot counted its own strips */
        HIP_TRY(ctx, hipSetDevice(c->device));
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));
        HIP_TRY(ctx, hipMemcpy(h, c->counters, sizeof h, hipMemcpyDeviceToHost));
        out->primary_rays += h[0];
        out->shadow_rays += h[1];
    }
    if (!ctx->peers.empty()) HIP_TRY(ctx, hipSetDevice(ctx->device));
    return C2RT_OK;
}

int c2rt_get_csg_truncations(c2rt_ctx *ctx, uint64_t *out)
{
    if (!ctx || !out) return C2RT_ERR_INVALID_ARG;
    if (!ctx->counters_valid) return fail(ctx, C2RT_ERR_INVALID_ARG, "last render did not count rays");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (const int st = drain_inflight(ctx)) return st;
    unsigned long long h = 0;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->counters + 2, sizeof h, hipMemcpyDeviceToHost));
    *out = h;
    for (c2rt_ctx *c : ctx->peers) {
        HIP_TRY(ctx, hipSetDevice(c->device));
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));
        HIP_TRY(ctx, hipMemcpy(&h, c->counters + 2, sizeof h, hipMemcpyDeviceToHost));
        *out += h;
    }
    if (!ctx->peers.empty()) HIP_TRY(ctx, hipSetDevice(ctx->device));
    return C2RT_OK;
}

int c2rt_get_exact_redos(c2rt_ctx *ctx, uint64_t *out)
{
    if (!ctx || !out) return C2RT_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    unsigned long long h = 0;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->counters + 3, sizeof h, hipMemcpyDeviceToHost));
    *out = h;
    for (c2rt_ctx *c : ctx->peers) {
        HIP_TRY(ctx, hipSetDevice(c->device));
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipMemcpy(&h, c->counters + 3, sizeof h, hipMemcpyDeviceToHost));
        *out += h;
    }
    if (!ctx->peers.empty()) HIP_TRY(ctx, hipSetDevice(ctx->device));
    return C2RT_OK;
}

int c2rt_render_pixel(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts, int x, int y,
                      c2rt_trace_result *out)
{
    int st = check_frame_args(ctx, cam, opts);
    if (st != C2RT_OK) return st;
    if (!out) return fail(ctx, C2RT_ERR_INVALID_ARG, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, opts, p);
    p.n_cull = 0; /* the probe launch has no tile: never cull */
    p.n_cull_lights = 0;
    p.csg_cap = (uint32_t)kCsgFullCap(C2RT_MAX_CSG_DEPTH); /* the probe instance handles every depth */
    p.probe_x = x;
    p.probe_y = y;
    p.probe_out = ctx->probe;
    HIP_TRY(ctx, hipMemsetAsync(ctx->probe, 0, sizeof(c2rt_trace_result), ctx->stream));
    const int e = launch_probe(p, variant_of(ctx, cam), ctx->stream);
    if (e != 0) return fail(ctx, C2RT_ERR_HIP, "probe kernel launch: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->probe, sizeof *out, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return C2RT_OK;
}

static int deinterleave_words(c2rt_ctx *ctx, const float *gathered_dev, float *frame_dev, uint32_t width, uint32_t height,
                              uint32_t strip_height, uint32_t world, uint32_t words_per_pixel, void *hip_stream)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    if (!gathered_dev || !frame_dev || width == 0 || height == 0 || world == 0)
        return fail(ctx, C2RT_ERR_INVALID_ARG, "bad de-interleave arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    c2rt_render_opts o;
    std::memset(&o, 0, sizeof o);
    o.width = width;
    o.height = height;
    o.strip_height = strip_height;
    o.strip_world = world;
    const uint32_t sh = strip_h(&o);
    const uint32_t rows_pad = world > 1 ? local_rows_of(&o, 0) : height; /* rank 0 always owns the most rows */
    const int e = launch_deinterleave(gathered_dev, frame_dev, width, height, sh, world, rows_pad, words_per_pixel, hip_stream);
    if (e != 0) return fail(ctx, C2RT_ERR_HIP, "de-interleave launch: %s", hipGetErrorString((hipError_t)e));
    return C2RT_OK;
}

int c2rt_deinterleave_strips(c2rt_ctx *ctx, const float *gathered_dev, float *frame_dev, uint32_t width,
                             uint32_t height, uint32_t strip_height, uint32_t world, void *hip_stream)
{
    return deinterleave_words(ctx, gathered_dev, frame_dev, width, height, strip_height, world, 3, hip_stream);
}

int c2rt_deinterleave_strips_rgb32(c2rt_ctx *ctx, const uint32_t *gathered_dev, uint32_t *frame_dev, uint32_t width,
                                   uint32_t height, uint32_t strip_height, uint32_t world, void *hip_stream)
{
    return deinterleave_words(ctx, reinterpret_cast<const float *>(gathered_dev), reinterpret_cast<float *>(frame_dev), width,
                              height, strip_height, world, 1, hip_stream);
}

int c2rt_render_frame_rgb32(c2rt_ctx *ctx, const c2rt_camera_frame *cam, const c2rt_render_opts *opts,
                            uint32_t *out_rgb32, const volatile uint8_t *stop_flag)
{
    int st = check_frame_args(ctx, cam, opts);
    if (st != C2RT_OK) return st;
    if (!out_rgb32) return fail(ctx, C2RT_ERR_INVALID_ARG, "null output");
    if (stop_flag && *stop_flag) return fail(ctx, C2RT_ERR_CANCELLED, "stop requested before the frame");
    if ((st = check_multi_opts(ctx, opts)) != C2RT_OK) return st;
    if (!ctx->peers.empty()) return render_to_host_multi(ctx, cam, opts, nullptr, out_rgb32, stop_flag);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return render_to_host(ctx, cam, opts, nullptr, out_rgb32, stop_flag);
}

int c2rt_encode_rgb32(c2rt_ctx *ctx, const float *frame_dev, uint32_t *out_dev, uint64_t n_pixels, void *hip_stream)
{
    if (!ctx) return C2RT_ERR_INVALID_ARG;
    if (!frame_dev || !out_dev) return fail(ctx, C2RT_ERR_INVALID_ARG, "null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_pixels == 0) return C2RT_OK;
    const int e = launch_encode_rgb32(frame_dev, out_dev, n_pixels, ctx->srgb_lut, hip_stream);
    if (e != 0) return fail(ctx, C2RT_ERR_HIP, "encode launch: %s", hipGetErrorString((hipError_t)e));
    return C2RT_OK;
}

} /* extern "C" */
