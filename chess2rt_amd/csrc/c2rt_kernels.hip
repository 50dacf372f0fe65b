/*
 * c2rt_kernels.hip — the per-pixel ray-trace hot path of Chess2RT, written
 * for gfx950 (MI355X, CDNA4).  Not a port: the reference is a D class
 * hierarchy with virtual dispatch over a linear node list; this is a
 * wave-synchronous trace where
 *
 *   - one 64-lane wavefront owns one 8x8 pixel tile (one workgroup = one
 *     wave; the grid has >>256 workgroups, dealt round-robin to the 8 XCDs, and
 *     the block -> tile map gives each XCD every 8th tile ROW so that texel
 *     reuse stays inside one XCD's L2 while all XCDs see the same sky/floor mix);
 *   - every lane walks the SAME node / geometry / light at the same time, so
 *     the scene records are read with scalar loads into SGPRs and the type
 *     dispatch is a scalar branch, never a divergent one;
 *   - CSG hit lists live in LDS, one [entry][lane] slab per nesting level
 *     (bank = lane, conflict-free for any per-lane entry index); only
 *     (dist, tag) is kept per hit and the winning hit is re-derived, which
 *     keeps the slab at 10 KiB per wave and level;
 *   - geometry is fp64 and colour fp32 in the reference's operation order
 *     (built with -ffp-contract=off), because checker edges, shadow
 *     terminators and CSG boundaries flip on 1-ulp differences.
 *
 * What each function restates is cited as file:line of /root/reference/source.
 */
#include <hip/hip_runtime.h>

#include "c2rt_device.h"
#include "x87.h"

namespace c2rt {
namespace {

#define DEV __device__ __forceinline__

/* fp64 libm is only reached by a few lanes (sphere u,v, the Phong lobe,
 * Procedure2) but, inlined, its ~70 live registers set the whole kernel's
 * budget; as real calls the trace stays under 168 VGPRs without spills. */
__device__ __noinline__ double c2_pow(double a, double b) { return pow(a, b); }
__device__ __noinline__ double c2_atan2(double a, double b) { return atan2(a, b); }
__device__ __noinline__ double c2_asin(double a) { return asin(a); }
__device__ __noinline__ double c2_sin(double a) { return sin(a); }
__device__ __noinline__ double c2_cos(double a) { return cos(a); }
/* Register budget per kernel instance, as waves per SIMD (512 VGPRs per lane and SIMD: 128 at 4 waves,
 * 168 at 3, 256 at 2).  With no hint hipcc takes all 512 registers and runs one wave per SIMD (1.8x
 * slower).  Chosen per instance from the compiler's resource remarks (profiles/r02_resource_usage.txt)
 * so that NO instance spills VGPRs to scratch, except where a measurement says otherwise:
 *   depth 0 (no CSG), planes-only: 4 waves (111-127 VGPRs);
 *   depth 1, at most one light: 4 waves — 128 VGPRs since the cube / sphere face tables moved to the upload
 *     and the hit's lighting terms are evaluated before the shadow test (was 149 at 3 waves);
 *   depth 1, several lights (the hit stays live across the light loop) and depth-1 DOF: 3 waves (144-162);
 *   depth 2: 2 waves (176-202); depth 3 / 4: 2 waves (253-256, depth 4 spills ~45 VGPRs: measured faster
 *     than 1 wave without spills, 19.8 -> 12 ms on csg_stress). */
#ifndef C2RT_OCC_U1
#define C2RT_OCC_U1 4
#endif
#ifndef C2RT_OCC_DEEP
#define C2RT_OCC_DEEP 2
#endif
template <int LEVELS, int DOF, bool MLC>
constexpr int occ_of()
{
#ifndef C2RT_OCC_U0
#define C2RT_OCC_U0 4
#endif
    return LEVELS == 0 ? (DOF ? 4 : C2RT_OCC_U0) : (LEVELS == 1 ? ((DOF || MLC) ? 3 : C2RT_OCC_U1) : (LEVELS == 2 ? 2 : C2RT_OCC_DEEP));
}
#define C2RT_OCC_OF(L, D, M) __attribute__((amdgpu_waves_per_eu(occ_of<L, D, M>(), occ_of<L, D, M>())))
#ifndef C2RT_TILE_STATS
#define C2RT_TILE_STATS 0 /* diagnostics: per-tile wave cycles + class (RenderParams::tile_stats) */
#endif
#ifndef C2RT_XCD_SWIZZLE
#define C2RT_XCD_SWIZZLE 1
#endif

/* ------------------------------------------------------------------ */
/* small vector types (gfm vec3d / rt Color semantics)                  */
/* ------------------------------------------------------------------ */

struct D3 { double x, y, z; };
DEV D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV D3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
DEV D3 operator+(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV D3 operator-(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV D3 operator*(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV D3 operator-(D3 a) { return mk(-a.x, -a.y, -a.z); }
/* gfm dot/squaredMagnitude start from `sum = 0`; 0 + x differs from x only in
 * the sign of an exact zero, which no consumer on this path observes. */
DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV double sqmag(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV double mag(D3 a) { return sqrt(sqmag(a)); }
DEV D3 normalized(D3 a) { double inv = 1.0 / mag(a); return mk(a.x * inv, a.y * inv, a.z * inv); }
/* mul(v, M): row vector x matrix — rt/imported_types.d:13-20 */
DEV D3 mulvm(D3 v, const double *m)
{
    return mk(v.x * m[0] + v.y * m[3] + v.z * m[6],
              v.x * m[1] + v.y * m[4] + v.z * m[7],
              v.x * m[2] + v.y * m[5] + v.z * m[8]);
}

struct F3 { float r, g, b; };
DEV F3 mkf(float r, float g, float b) { F3 c; c.r = r; c.g = g; c.b = b; return c; }
DEV F3 ldf3(const float *p) { return mkf(p[0], p[1], p[2]); }
DEV F3 operator+(F3 a, F3 b) { return mkf(a.r + b.r, a.g + b.g, a.b + b.b); }
DEV F3 operator*(F3 a, F3 b) { return mkf(a.r * b.r, a.g * b.g, a.b * b.b); }
DEV F3 operator*(F3 a, float f) { return mkf(a.r * f, a.g * f, a.b * f); }
DEV F3 operator/(F3 a, float f) { return mkf(a.r / f, a.g / f, a.b / f); }

/* a double held by lane `l`, as a wave-uniform value */
DEV double read_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

/* x86 cvttsd2si / cvttss2si results for out-of-range inputs, which is what
 * the reference binary computes for cast(int) / cast(size_t). */
DEV int d2i_x86(double d)
{
    return (d > -2147483649.0 && d < 2147483648.0) ? (int)d : (int)0x80000000;
}

/* ------------------------------------------------------------------ */
/* hit record                                                           */
/* ------------------------------------------------------------------ */

struct Hit {          /* IntersectionData — rt/intersectable.d:6-33 (dNdx/dNdy are dead on this path) */
    D3 p, n;
    double dist, u, v;
    /* Sphere u,v cost an atan2 and an asin and are only read by textured
     * shaders: a sphere hit leaves (p.x-c.x, p.z-c.z, (p.y-c.y)/R) in (u, v, w)
     * with `uv_pending` set, and finish_uv() runs once for the closest hit. */
    double w;
    int g;
    bool uv_pending;
    bool axis_n;      /* n is an exact signed unit axis (plane, cube): normalize(n) == n */
};

/* Sphere.intersect's u,v — rt/geometry.d:118-120.  `PI` is an 80-bit real there: the two expressions
 * are evaluated with the x87's 64-bit significand and rounded to double on assignment; x87.h
 * reproduces that with integer arithmetic (a NaN / infinite operand takes the plain double
 * expression, whose result is the same NaN). */
DEV void finish_uv(Hit &h)
{
    if (h.uv_pending) {
        constexpr double PI = 3.14159265358979323846;
        const double angle = c2_atan2(h.v, h.u);
        const double as = c2_asin(h.w);
        h.u = fabs(angle) <= 4.0 ? x87_sphere_u(angle) : (PI + angle) / (2 * PI);
        h.v = fabs(as) <= 2.0 ? x87_sphere_v(as) : 1.0 - (PI / 2 + as) / PI;
        h.uv_pending = false;
    }
}

/* what a caller needs back from an intersect call */
enum Need {
    kBool = 0,  /* hit / no hit (and the distance) */
    kPoint = 1, /* + p and the leaf geometry */
    kFull = 2,  /* + normal and u,v */
    kRt = 3     /* kPoint, and kFull for the lanes whose run-time `full` flag is set */
};
#define C2RT_WANT_FULL(NEED, full) ((NEED) == kFull || ((NEED) == kRt && (full)))

/* what the trace needs below the shading level: table bases (SGPRs), the
 * wave's CSG slabs and the lane.  Deliberately NOT a pointer to the kernel
 * arguments (which would have to be materialised in scratch if it escaped). */
/* The kernel-argument segment (the by-value RenderParams block), as handed down from the __global__
 * entry point: helpers must not fetch it themselves (__builtin_amdgcn_kernarg_segment_ptr() is only
 * meaningful in the kernel function; an out-of-line callee would read garbage). */
typedef const RenderParams __attribute__((address_space(4))) *KArgs;

struct Ctx {
    KArgs kargs;
    const DevGeom *geoms;
    const DevNode *nodes;
    uint32_t n_nodes;
    char *lds;        /* this wave's CSG hit stack: dist[csg_cap][64] (8 B) then tag[csg_cap][64] (2 B) */
    int lane;
    int csg_cap;      /* entries of that stack (wave-uniform, RenderParams::csg_cap) */
    /* A lane's nested hit lists did not fit csg_cap entries: its results are void and the whole tile is
     * rendered again by the full-capacity launch (RenderParams::retry_list).  Written through const
     * references on purpose: everything here is inlined into the kernel and it lives in a register. */
    mutable bool overflow;
    /* Where a child's hit list reaches C2RT_MAX_CSG_HITS — the reference's findAllIntersections
     * (`while (true)`, rt/geometry.d:271-290) would have gone on, this build stops: build-defined
     * behaviour — the lane bumps this counter (c2rt_get_csg_truncations), so that nobody has to take
     * the cap's harmlessness on trust.  Null (at compile time) in the production instances: only the
     * instances launched when rays are being counted carry it (render_tile's CNT) — measured in the
     * headline kernel, which sits exactly at its 128-VGPR budget, a flag carried to the end of the tile
     * cost 3 % and an atomic on the spot 8 %. */
    unsigned long long *trunc_counter;
#if C2RT_TILE_STATS
    unsigned long long *lane_stats; /* diagnostics: {active, slots} pairs — [0..1] CSG evaluations entered, [2..3] child stepping calls */
#endif
    uint32_t block;   /* the tile this wave renders (blockIdx.x, or an entry of the retry list) */
    uint32_t primary_mask; /* bit n clear: no primary ray of this tile can reach node n (wave-uniform) */
    uint32_t shadow_mask0; /* same for the tile's shadow rays towards light 0 (further lights: shadow_cull_mask) */
    /* the ground plane (RenderParams::ground_node) is the ONLY node left in shadow_mask0: the tile's
     * shadow rays towards light 0 are decided by plane_points_away alone, without building a ray */
    bool shadow_ground_only;
    /* the ground plane is the ONLY node left in primary_mask: the tile's primary rays take the
     * straight-line ground trace (raytrace) instead of the node loop */
    bool primary_ground_only;
    double ground_y;
};

/* A ray in some object space: origin, unit direction and A = |d|^2 exactly as
 * Sphere.intersect computes it (rt/geometry.d:96) — the direction is shared by
 * every geometry under a node and by all steps of findAllIntersections, so A
 * is evaluated once per direction instead of once per sphere test. */
struct ORay {
    D3 o, d;
    double A;
};

/* ------------------------------------------------------------------ */
/* primitives                                                           */
/* ------------------------------------------------------------------ */

/* Plane.intersect — rt/geometry.d:30-59 */
template <int NEED>
DEV bool plane_intersect(const DevGeom *G, int gid, const ORay &r, Hit &h, bool full)
{
    const D3 o = r.o, d = r.d;
    const double y = G->p[0], limit = G->p[1];
    /* (bitwise forms of the reference's conditions: same truth table incl. NaN, one branch each) */
    if (((o.y > y) & (d.y > -1e-9)) | ((o.y < y) & (d.y < 1e-9))) return false;
    const double mult = (o.y - y) / -d.y;
    const D3 p = o + d * mult;
    if ((mult > h.dist) | (fabs(p.x) > limit) | (fabs(p.z) > limit)) return false;
    h.dist = mult;
    if (NEED >= kPoint) { h.p = p; h.g = gid; }
    if (C2RT_WANT_FULL(NEED, full)) { h.n = mk(0, 1, 0); h.u = p.x; h.v = p.z; h.uv_pending = false; h.axis_n = true; }
    return true;
}

/* Sphere.intersect — rt/geometry.d:92-125 */
template <int NEED>
DEV bool sphere_intersect(const DevGeom *G, int gid, const ORay &r, Hit &h, bool full)
{
    const D3 o = r.o, d = r.d;
    const D3 c = ld3(G->p);
    const double R = G->p[3];
    const D3 H = o - c;
    const double A = r.A;
    const double B = 2 * dot(H, d);
    const double C = sqmag(H) - G->q[0]; /* R * R, from the table */
    const double BB = B * B;
    const double Dscr = BB - 4 * A * C;
    /* Dscr < 0: no real root.  Or: outside the sphere and moving away — sqrt(Dscr) < B
     * by a margin far above rounding, so both roots are negative: the reference
     * returns false after a sqrt and two divisions; skip them.  (Near-degenerate
     * cases fall through to the literal evaluation.)  One branch for both. */
    if ((Dscr < 0) | ((C > 0) & (B > 0) & (Dscr < BB * (1 - 1e-8)))) return false;
    const double sq = sqrt(Dscr);
    const double A2 = 2 * A;
    /* x1 = (-B + sq) / (2A) is only read when x2 < 0 */
    double sol = (-B - sq) / A2;
    if (sol < 0) sol = (-B + sq) / A2;
    if ((sol < 0) | (sol > h.dist)) return false;
    h.dist = sol;
    if (NEED >= kPoint) {
        const D3 p = o + d * sol;
        h.p = p;
        h.g = gid;
        if (C2RT_WANT_FULL(NEED, full)) {
            h.n = normalized(p - c);
            h.u = p.x - c.x;
            h.v = p.z - c.z;
            h.w = (p.y - c.y) / R;
            h.uv_pending = true;
            h.axis_n = false;
        }
    }
    return true;
}

/* Cube.intersectCubeSide — rt/geometry.d:198-235, for the face pair whose
 * axis is `ay` after the project() permutation; (ax, az) are the other two
 * axes in permuted order.  Arithmetic is component-wise, so it is evaluated
 * in place instead of permuting (project/unproject, rt/imported_types.d:44-60).
 * `mult < 0` is decided from the operand signs (an IEEE quotient is negative
 * iff exactly one operand is and the numerator is non-zero), which skips the
 * division for every face behind the origin. */
template <int NEED, int AXIS>
DEV bool cube_sides(double oy, double dy, double ylo, double yhi, double ox, double dx, double cx, double xlo, double xhi,
                    double oz, double dz, double cz, double zlo, double zhi, D3 o, D3 d, Hit &h, bool full)
{
    if (fabs(dy) < 1e-9) return false;
    bool found = false;
    const double den = -dy;
    /* face planes and bounds come from the table (DevGeom::q): `center + side * halfSide` and
     * `center -+ halfSide` are the same for every ray */
#pragma unroll
    for (int side = -1; side <= 1; side += 2) {
        const double num = oy - (side < 0 ? ylo : yhi);
        const bool negative = ((num < 0) & (den > 0)) | ((num > 0) & (den < 0));
        if (negative) continue;          /* mult < 0 */
        const double mult = num / den;
        const double px = ox + dx * mult, pz = oz + dz * mult;
        /* the reference's four rejections (mult < 0 kept for NaN / zero corner cases) as ONE
         * predicate: one divergent branch per face instead of four */
        const bool reject = (mult < 0) | (mult > h.dist) | (px < xlo) | (px > xhi) | (pz < zlo) | (pz > zhi);
        if (reject) continue;
        h.dist = mult;
        if (NEED >= kPoint) h.p = o + d * mult;
        if (C2RT_WANT_FULL(NEED, full)) {
            /* Vector(0, side, 0) un-permuted: the face normal along AXIS */
            h.n = mk(AXIS == 0 ? (double)side : 0.0, AXIS == 1 ? (double)side : 0.0, AXIS == 2 ? (double)side : 0.0);
            h.u = px - cx;
            h.v = pz - cz;
        }
        found = true;
    }
    return found;
}

/* Cube.intersect — rt/geometry.d:172-196 */
template <int NEED>
DEV bool cube_intersect(const DevGeom *G, int gid, const ORay &r, Hit &h, bool full)
{
    const D3 o = r.o, d = r.d;
    const D3 c = ld3(G->p);
    const D3 lo = ld3(G->q), hi = ld3(G->q + 3);
    /* Y faces; X faces = project(1,0,2): (y,x,z); Z faces = project(0,2,1): (x,z,y) */
    bool found = cube_sides<NEED, 1>(o.y, d.y, lo.y, hi.y, o.x, d.x, c.x, lo.x, hi.x, o.z, d.z, c.z, lo.z, hi.z, o, d, h, full);
    found |= cube_sides<NEED, 0>(o.x, d.x, lo.x, hi.x, o.y, d.y, c.y, lo.y, hi.y, o.z, d.z, c.z, lo.z, hi.z, o, d, h, full);
    found |= cube_sides<NEED, 2>(o.z, d.z, lo.z, hi.z, o.x, d.x, c.x, lo.x, hi.x, o.y, d.y, c.y, lo.y, hi.y, o, d, h, full);
    if (found) {
        if (NEED >= kPoint) h.g = gid;
        if (C2RT_WANT_FULL(NEED, full)) {
            h.uv_pending = false;
            h.axis_n = true;
        }
    }
    return found;
}

/* Conservative reject: true when the ray cannot reach the geometry's padded
 * bounding sphere (DevGeom::bound), in which case Geometry.intersect would
 * return false after doing all of its work.  |d| = 1 up to rounding and the
 * radius is padded by 1e-6 relative, so the test only ever errs towards
 * "may hit". */
DEV bool misses_bound(const DevGeom *G, const ORay &r)
{
    const D3 H = r.o - ld3(G->bound);
    const double b = dot(H, r.d);
    const double c = sqmag(H) - G->bound[3];
    return (c > 0) & ((b > 0) | (b * b < c));
}

/* isInside — rt/geometry.d:25-28,127-130,165-170,334-337 */
template <int LEVEL>
DEV bool geom_is_inside(const Ctx &cx, int gid, D3 p)
{
    const DevGeom *G = cx.geoms + gid;
    const int type = G->type;
    if (type == C2RT_GEOM_SPHERE) {
        return sqmag(ld3(G->p) - p) < G->q[0];
    } else if (type == C2RT_GEOM_CUBE) {
        const double hs = G->p[3] * 0.5;
        return (fabs(p.x - G->p[0]) <= hs) & (fabs(p.y - G->p[1]) <= hs) & (fabs(p.z - G->p[2]) <= hs);
    } else if (type == C2RT_GEOM_PLANE) {
        return false;
    } else {
        if constexpr (LEVEL > 0) {
            const bool a = geom_is_inside<LEVEL - 1>(cx, G->left, p);
            const bool b = geom_is_inside<LEVEL - 1>(cx, G->right, p);
            return type == C2RT_GEOM_CSG_UNION ? (a | b) : (type == C2RT_GEOM_CSG_INTER ? (a & b) : (a & !b));
        } else {
            return false;
        }
    }
}

/* ------------------------------------------------------------------ */
/* CSG — rt/geometry.d:243-403                                          */
/* ------------------------------------------------------------------ */

template <int LEVEL, int NEED>
__device__ __forceinline__ bool geom_intersect(const Ctx &cx, int gid, const ORay &r, Hit &h, bool full, int base);

/* hit-list tag: leaf geometry (12 bits: C2RT_MAX_CSG_GEOMS) | child side | index of the hit in its child's list */
static_assert(kMaxCsgHits <= 8 && C2RT_MAX_CSG_GEOMS <= 4096, "16-bit hit tags");
DEV uint16_t csg_tag(int leaf, int side, int k) { return (uint16_t)(((uint32_t)leaf << 4) | ((uint32_t)side << 3) | (uint32_t)k); }

/* CsgOp.intersect (+ CsgDiff.intersect) for a CSG whose subtree has at most
 * LEVEL nesting levels.  The hit lists of findAllIntersections
 * (rt/geometry.d:271-290) are kept in this level's LDS slab as
 * (dist, 16-bit tag = leaf<<4 | side<<3 | k); they are concatenated left-then-right
 * and shell-sorted exactly as util/array.d:95-111 does (same tie behaviour),
 * walked with the in/out toggles of rt/geometry.d:303-329 (including the
 * `current.g is left` leaf-identity test), and the winning hit is then
 * re-derived by replaying its child's stepping up to k.
 *
 * The four child-stepping loops (collect left, collect right, replay left,
 * replay right) run as ONE wave-uniform loop with a SINGLE call site for the
 * level below, so every nesting level is inlined exactly once: code size is
 * linear in the depth, there are no out-of-line calls and no scratch, and the
 * child's geometry record stays a scalar load.  Whether the replayed hit needs
 * its normal / u,v is a per-lane flag (`full`), because lanes replay at
 * different steps. */
/* LDS: one hit STACK per wave instead of a fixed 16-entry slab per nesting level.  The list of a
 * CsgOp starts at the per-lane index `base`; while a child is being stepped the child's own lists
 * live above the parent's current top (base + n + k) and are dead when the child returns, so the
 * parent's next entry overwrites them.  Typical trees need 4 entries per level (two hits per
 * primitive child) where the slabs reserved 16: a depth-4 scene runs in 20 KiB per wave instead of
 * 40, i.e. two waves per SIMD instead of one.  A lane that would push beyond csg_cap raises
 * Ctx::overflow and stops collecting; its tile is redone by the full-capacity launch
 * (16 entries x depth, which cannot overflow: every level holds at most 8 + 8). */
template <int LEVEL, int NEED>
__device__ __forceinline__ bool csg_intersect(const Ctx &cx, const DevGeom *G, const ORay &ray, Hit &h, bool full, int base)
{
    static_assert(LEVEL >= 1, "CSG needs a stack");
    double *ldist = reinterpret_cast<double *>(cx.lds) + cx.lane + base * kWave;
    uint16_t *ltag = reinterpret_cast<uint16_t *>(cx.lds + cx.csg_cap * (kWave * 8)) + cx.lane + base * kWave;
    const int room = cx.csg_cap - base; /* entries this list may use */
    const int type = G->type, left = G->left, right = G->right, flags = G->flags;
    const D3 d = ray.d;
    const bool want_full = NEED == kFull || (NEED == kRt && full);

    int nL = 0, n = 0;
    int wside = -1, wk = 0; /* the winning entry: which child, which of its hits */
    Hit t;
    t.dist = 1e99;
    constexpr int kPasses = NEED == kBool ? 2 : 4; /* 0/1: collect left/right, 2/3: replay left/right */
#pragma unroll 1
    for (int pass = 0; pass < kPasses; ++pass) {
        const int side = pass & 1;
        const int child = side ? right : left; /* wave-uniform */
        const bool replay = pass >= 2;
        if (replay && wside != side) continue; /* this lane's winner came from the other child */
        const int limit = replay ? wk + 1 : kMaxCsgHits;
        ORay rr = ray;
        double cur = 0;
        int k = 0;
        while (k < limit) {
            t.dist = 1e99;
            /* the child's lists go above this list's top; a replay no longer needs this list */
            if (!geom_intersect<LEVEL - 1, kRt>(cx, child, rr, t, replay && k == wk && want_full, replay ? base : base + n + k)) break;
            t.dist += cur;
            cur = t.dist;
            rr.o = t.p + d * 1e-6;
            if (!replay) {
                if (n + k >= room) { cx.overflow = true; break; }
                ldist[(n + k) * kWave] = t.dist;
                ltag[(n + k) * kWave] = csg_tag(t.g, side, k);
            }
            ++k;
        }
        if (replay) break; /* `t` is the re-derived winner (data = current, rt/geometry.d:326) */
        if (k == kMaxCsgHits && cx.trunc_counter) atomicAdd(cx.trunc_counter, 1ull);
        if (side == 0) nL = k;
        n += k;
        /* exact shortcuts (GeomFlags): nothing can switch the operator on */
        if (k == 0 && (flags & (side ? kCsgShortB : kCsgShortA))) return false;
        if (side == 0) continue;

        /* both lists are in: sort — util/array.d:95-111 (index rewound by the inner while) */
        for (int inc = n / 2; inc;) {
            for (int i = 0; i < n; ++i) {
                const double ed = ldist[i * kWave];
                const uint16_t et = ltag[i * kWave];
                while (i >= inc && ldist[(i - inc) * kWave] > ed) {
                    ldist[i * kWave] = ldist[(i - inc) * kWave];
                    ltag[i * kWave] = ltag[(i - inc) * kWave];
                    i -= inc;
                }
                ldist[i * kWave] = ed;
                ltag[i * kWave] = et;
            }
            inc = (inc == 2) ? 1 : (int)(inc * 5.0 / 11);
        }
        bool inL = (nL & 1) != 0, inR = ((n - nL) & 1) != 0;
        int win = -1;
        for (int i = 0; i < n; ++i) {
            const uint32_t tag = ltag[i * kWave];
            const bool isL = (int)(tag >> 4) == left;
            inL ^= isL;
            inR ^= !isL;
            const bool in = type == C2RT_GEOM_CSG_UNION ? (inL | inR)
                          : (type == C2RT_GEOM_CSG_INTER ? (inL & inR) : (inL & !inR));
            if (in) { win = i; break; }
        }
        if (win < 0) return false;
        const double wdist = ldist[win * kWave];
        if (wdist > h.dist) return false;
        if (NEED == kBool) { h.dist = wdist; return true; }
        const uint32_t wtag = ltag[win * kWave];
        wside = (wtag >> 3) & 1;
        wk = wtag & 7;
    }
    h = t;

    if (want_full && type == C2RT_GEOM_CSG_DIFF) { /* CsgDiff.intersect — rt/geometry.d:382-397 */
        if (geom_is_inside<LEVEL - 1>(cx, right, h.p - d * 1e-6) != geom_is_inside<LEVEL - 1>(cx, right, h.p + d * 1e-6))
            h.n = -h.n;
    }
    return true;
}

/* The same for a CSG whose children are both primitives (the innermost level, by
 * far the most executed one): the three stepping loops are written out, with
 * compile-time NEED for the collect loops — measurably faster than the single
 * call-site form, and the duplicated primitive code is small. */
template <int NEED>
__device__ __forceinline__ bool csg_intersect_leaf(const Ctx &cx, const DevGeom *G, const ORay &ray, Hit &h, bool full, int base)
{
    constexpr int LEVEL = 1;
    double *ldist = reinterpret_cast<double *>(cx.lds) + cx.lane + base * kWave;
    uint16_t *ltag = reinterpret_cast<uint16_t *>(cx.lds + cx.csg_cap * (kWave * 8)) + cx.lane + base * kWave;
    const int room = cx.csg_cap - base;
    const int type = G->type, left = G->left, right = G->right, flags = G->flags;
    const D3 d = ray.d;
#if C2RT_TILE_STATS
    if (cx.lane_stats) {
        const unsigned long long act = __ballot(true);
        if (cx.lane == (int)__builtin_ctzll(act)) { atomicAdd(cx.lane_stats + 0, (unsigned long long)__builtin_popcountll(act)); atomicAdd(cx.lane_stats + 1, 64ull); }
    }
#endif

    int nL = 0, nR = 0;
    int n = 0;
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const int child = side ? right : left;
        ORay rr = ray;
        double cur = 0;
        int k = 0;
        while (k < kMaxCsgHits) {
            Hit t;
            t.dist = 1e99;
#if C2RT_TILE_STATS
            if (cx.lane_stats) {
                const unsigned long long act = __ballot(true);
                if (cx.lane == (int)__builtin_ctzll(act)) { atomicAdd(cx.lane_stats + 2, (unsigned long long)__builtin_popcountll(act)); atomicAdd(cx.lane_stats + 3, 64ull); }
            }
#endif
            if (!geom_intersect<0, kPoint>(cx, child, rr, t, false, 0)) break;
            t.dist += cur;
            cur = t.dist;
            rr.o = t.p + d * 1e-6;
            if (n + k >= room) { cx.overflow = true; break; }
            ldist[(n + k) * kWave] = t.dist;
            ltag[(n + k) * kWave] = csg_tag(t.g, side, k);
            ++k;
        }
        if (side == 0) nL = k; else nR = k;
        if (k == kMaxCsgHits && cx.trunc_counter) atomicAdd(cx.trunc_counter, 1ull);
        n += k;
        /* exact shortcuts (GeomFlags): nothing can switch the operator on */
        if (k == 0 && (flags & (side ? kCsgShortB : kCsgShortA))) return false;
    }

    /* sort — util/array.d:95-111 (index rewound by the inner while) */
    for (int inc = n / 2; inc;) {
        for (int i = 0; i < n; ++i) {
            const double ed = ldist[i * kWave];
            const uint16_t et = ltag[i * kWave];
            while (i >= inc && ldist[(i - inc) * kWave] > ed) {
                ldist[i * kWave] = ldist[(i - inc) * kWave];
                ltag[i * kWave] = ltag[(i - inc) * kWave];
                i -= inc;
            }
            ldist[i * kWave] = ed;
            ltag[i * kWave] = et;
        }
        inc = (inc == 2) ? 1 : (int)(inc * 5.0 / 11);
    }

    bool inL = (nL & 1) != 0, inR = (nR & 1) != 0;
    int win = -1;
    for (int i = 0; i < n; ++i) {
        const uint32_t tag = ltag[i * kWave];
        const bool isL = (int)(tag >> 4) == left;
        inL ^= isL;
        inR ^= !isL;
        const bool in = type == C2RT_GEOM_CSG_UNION ? (inL | inR)
                      : (type == C2RT_GEOM_CSG_INTER ? (inL & inR) : (inL & !inR));
        if (in) { win = i; break; }
    }
    if (win < 0) return false;
    const double wdist = ldist[win * kWave];
    if (wdist > h.dist) return false;
    if (NEED == kBool) { h.dist = wdist; return true; }

    /* re-derive the winning IntersectionData (data = current, rt/geometry.d:326) */
    const uint32_t wtag = ltag[win * kWave];
    const int wside = (wtag >> 3) & 1, wk = wtag & 7;
    const int child = wside ? right : left;
    ORay rr = ray;
    double cur = 0;
    for (int i = 0; i < wk; ++i) {
        Hit t;
        t.dist = 1e99;
        geom_intersect<0, kPoint>(cx, child, rr, t, false, 0);
        t.dist += cur;
        cur = t.dist;
        rr.o = t.p + d * 1e-6;
    }
    Hit t;
    t.dist = 1e99;
    geom_intersect<0, NEED>(cx, child, rr, t, full, 0);
    t.dist += cur;
    h = t;

    if (C2RT_WANT_FULL(NEED, full) && type == C2RT_GEOM_CSG_DIFF) { /* CsgDiff.intersect — rt/geometry.d:382-397 */
        if (geom_is_inside<LEVEL - 1>(cx, right, h.p - d * 1e-6) != geom_is_inside<LEVEL - 1>(cx, right, h.p + d * 1e-6))
            h.n = -h.n;
    }
    return true;
}

/* Geometry.intersect on a given record: `G` is wave-uniform, so this is a scalar branch. */
template <int LEVEL, int NEED>
__device__ __forceinline__ bool geom_intersect_rec(const Ctx &cx, const DevGeom *G, int gid, const ORay &r, Hit &h, bool full, int base)
{
    const int type = G->type;
    if (type == C2RT_GEOM_PLANE) return plane_intersect<NEED>(G, gid, r, h, full);
    if (type == C2RT_GEOM_SPHERE) return sphere_intersect<NEED>(G, gid, r, h, full);
    if (type == C2RT_GEOM_CUBE) return cube_intersect<NEED>(G, gid, r, h, full);
    if constexpr (LEVEL >= 2) {
        return csg_intersect<LEVEL, NEED>(cx, G, r, h, full, base);
    } else if constexpr (LEVEL == 1) {
        return csg_intersect_leaf<NEED>(cx, G, r, h, full, base);
    } else {
        return false;
    }
}

/* `base`: first free entry of this lane's hit stack (CsgOps only) */
template <int LEVEL, int NEED>
__device__ __forceinline__ bool geom_intersect(const Ctx &cx, int gid, const ORay &r, Hit &h, bool full, int base)
{
    return geom_intersect_rec<LEVEL, NEED>(cx, cx.geoms + gid, gid, r, h, full, base);
}

/* ------------------------------------------------------------------ */
/* nodes — rt/node.d:23-49, rt/transform.d:57-86                         */
/* ------------------------------------------------------------------ */

/* The world ray normalised once for every node whose matrix is the identity:
 * undoDirection(dir) == dir there, so `magnitude` and `normalize`
 * (rt/node.d:34-36) give the same bits for all of them. */
struct RayW {
    D3 o, d;
    D3 dn;      /* d * (1/|d|) */
    double len; /* |d| */
    double A;   /* |dn|^2 */
};
DEV RayW make_ray(D3 o, D3 d)
{
    RayW r;
    r.o = o;
    r.d = d;
    r.len = mag(d);
    const double inv = 1.0 / r.len;
    r.dn = mk(d.x * inv, d.y * inv, d.z * inv);
    r.A = sqmag(r.dn);
    return r;
}

/* The first node >= n that a culling mask keeps: bit k of `mask` clear means
 * node k (< kMaxCullNodes = 32) cannot be reached; nodes beyond the mask are
 * always visited.  Wave-uniform: a shift and a find-first-set on the scalar unit. */
DEV uint32_t next_node(uint32_t mask, uint32_t n)
{
    static_assert(kMaxCullNodes == 32, "one 32-bit mask");
    if (n >= 32u) return n;
    const uint32_t m = mask >> n;
    return m ? n + (uint32_t)__builtin_ctz(m) : 32u;
}

/*
 * Plane.intersect's first rejection (rt/geometry.d:33-34: origin above the plane
 * and direction not pointing down, or the mirror image), decided from the
 * UN-normalised direction `raw` of a ray that would be traced as
 * make_ray(from, normalized(raw)) against an "axis plane" node (kNodeAxisPlane:
 * a Plane whose inverse matrix is the identity, or diagonal with entries of
 * magnitude in [1e-100, 1e100] and a positive y entry b).
 * With raw.y in (1e-150, 1e150) and |raw.x|, |raw.z| < 1e150:
 *   |raw|^2 is a positive normal number, 1/|raw| is positive, finite and normal,
 *   dir.y = raw.y / |raw| > 5e-301, dir is finite and |dir| is within rounding of 1;
 *   identity matrix: the node test sees dn.y = dir.y * (1/|dir|), a positive number;
 *   diagonal matrix: it sees dd = dir . inv, dd.y = (+-0) + dir.y * b + (+-0) >= 0 (possibly
 *   underflowed to zero), |dd| in [0.5e-100, 1.1e100], and d'.y = dd.y * (1/|dd|) >= 0, not NaN;
 *   either way `d.y > -1e-9` holds, and with o'.y > y the reference returns false.
 * (Mirror image for raw.y < 0 and o'.y < y.)  o'.y is evaluated literally.  Outside
 * those bounds — zero, huge, infinite or NaN components — the answer is "don't
 * know" and the literal evaluation runs.  What this saves: both normalisations
 * (2 sqrt, 2 divisions) and the matrix products of every ray that hits nothing.
 * Used by the kernel instances for scenes made of such planes only (PO), where
 * it removes the whole shadow-ray test of every lit pixel.
 */
DEV bool plane_points_away(const DevNode *N, D3 from, D3 raw)
{
    const uint32_t fl = N->flags;
    double oy = (fl & kNodeZeroOffset) ? from.y : from.y - N->off[1];
    if (!(fl & kNodeIdentityMatrix)) { /* mulvm(o - off, inv).y, literally */
        const double ox = (fl & kNodeZeroOffset) ? from.x : from.x - N->off[0];
        const double oz = (fl & kNodeZeroOffset) ? from.z : from.z - N->off[2];
        oy = ox * N->inv[1] + oy * N->inv[4] + oz * N->inv[7];
    }
    const double y = N->g.p[0];
    const bool sane = (fabs(raw.x) < 1e150) & (fabs(raw.z) < 1e150);
    const bool up = (raw.y > 1e-150) & (raw.y < 1e150), down = (raw.y < -1e-150) & (raw.y > -1e150);
    return sane & (((oy > y) & up) | ((oy < y) & down));
}

/* Node.intersect; `best.dist` is data.dist (world units) in and out.  `last` (wave-uniform): no
 * further node will be tested against this ray, so nobody reads the updated best.dist again (it only
 * serves as the next node's limit) and its division is skipped. */
template <int LEVELS, int NEED>
DEV bool node_intersect(const Ctx &cx, const DevNode *N, const RayW &ray, Hit &best, bool last = false)
{
    const uint32_t flags = N->flags;
    ORay rc;
    rc.o = ray.o;
    double len;
    if (!(flags & kNodeZeroOffset)) rc.o = rc.o - ld3(N->off);
    if (flags & kNodeIdentityMatrix) {
        rc.d = ray.dn;
        rc.A = ray.A;
        len = ray.len;
    } else {
        rc.o = mulvm(rc.o, N->inv);
        const D3 dd = mulvm(ray.d, N->inv);
        len = mag(dd);
        const double inv = 1.0 / len;
        rc.d = mk(dd.x * inv, dd.y * inv, dd.z * inv);
        rc.A = sqmag(rc.d);
    }
    const int gid = N->geom;
    const DevGeom *G = &N->g; /* the node's own copy of its root geometry record */
    /* cubes and CSG trees are expensive to miss: bounding-sphere reject first */
    if (G->type >= C2RT_GEOM_CUBE && (G->flags & kGeomBounded) && misses_bound(G, rc)) return false;
    Hit h;
    h.dist = best.dist * len;
    if (!geom_intersect_rec<LEVELS, NEED>(cx, G, gid, rc, h, false, 0)) return false;
    if (NEED == kBool) return true;
    if (!last) best.dist = h.dist / len;
    best.g = h.g;
    best.u = h.u;
    best.v = h.v;
    best.w = h.w;
    best.uv_pending = h.uv_pending;
    if (flags & kNodeIdentityMatrix) {
        /* normalize() of an exact unit axis is the identity (sqrt(1) = 1, 1/1 = 1) */
        best.n = h.axis_n ? h.n : normalized(h.n);
        best.p = h.p;
    } else {
        best.n = normalized(mulvm(h.n, N->tinv));
        best.p = mulvm(h.p, N->m);
    }
    if (!(flags & kNodeZeroOffset)) best.p = best.p + ld3(N->off);
    return true;
}

/* ------------------------------------------------------------------ */
/* per-frame culling masks (host-computed rectangles, RenderParams)      */
/* ------------------------------------------------------------------ */

/* Frame-space pixel bounds of this wave's tile: x in [tx0, tx0 + 8), rows
 * ty0..ty1 (the strip map is monotonic in the local row). */
DEV void tile_bounds(const RenderParams &P, uint32_t b, int &tx0, int &ty0, int &ty1)
{
#if C2RT_XCD_SWIZZLE
    const uint32_t xcd = b & 7u, j = b >> 3;
    /* row groups are walked starting at P.row_group_start (where the boxed nodes begin on screen):
     * the expensive tiles are dispatched first and the launch ends on cheap ones */
    const uint32_t groups = (P.tiles_y + 7u) / 8u;
    uint32_t grp = j / P.blocks_x + P.row_group_start;
    if (grp >= groups) grp -= groups;
    const uint32_t trow = grp * 8u + xcd, bcol = j % P.blocks_x;
#else
    const uint32_t trow = b / P.blocks_x, bcol = b % P.blocks_x;
#endif
    const uint32_t tcol = bcol * kWavesPerBlock + threadIdx.x / kWave;
    tx0 = (int)(tcol * kTileW);
    const uint32_t lr_first = trow * kTileH + P.row_offset;
    uint32_t lr_last = trow * kTileH + kTileH - 1;
    if (lr_last >= P.local_rows) lr_last = P.local_rows - 1;
    lr_last += P.row_offset;
    ty0 = (int)lr_first;
    ty1 = (int)lr_last;
    if (P.strip_world > 1) {
        const uint32_t sh = P.strip_height;
        ty0 = (int)(((lr_first / sh) * P.strip_world + P.strip_rank) * sh + lr_first % sh);
        ty1 = (int)(((lr_last / sh) * P.strip_world + P.strip_rank) * sh + lr_last % sh);
    }
}

/* this lane's node rectangle (lane n stands for node n), read from the kernel-argument segment */
DEV void lane_rect(const RenderParams &P, KArgs K, int lane, bool &mine, int &r0, int &r1, int &r2, int &r3)
{
    typedef const int __attribute__((address_space(4))) *KInt;
    KInt rects = (KInt)((const char __attribute__((address_space(4))) *)K + __builtin_offsetof(RenderParams, cull_rect));
    mine = (uint32_t)lane < P.n_cull; /* lanes >= n_cull stand for "always test" */
    const int ln = mine ? lane : 0;
    r0 = rects[4 * ln + 0];
    r1 = rects[4 * ln + 1];
    r2 = rects[4 * ln + 2];
    r3 = rects[4 * ln + 3];
}

/* Nodes that may occlude this tile's shadow rays towards light l: every hit point
 * lies in the tile's view pyramid; a node whose box is entirely beyond one side
 * plane of the pyramid while the light is on the inner side of that plane is in a
 * half space none of those segments enters.  Lanes that are not active (missed,
 * left the frame) cannot vote: their nodes stay "may occlude". */
DEV uint32_t shadow_cull_mask(const RenderParams &P, KArgs K, uint32_t block, int lane, uint32_t l)
{
    if (!P.n_cull || l >= P.n_cull_lights) return 0xFFFFFFFFu;
    int tx0, ty0, ty1;
    tile_bounds(P, block, tx0, ty0, ty1);
    /* sample coordinates of this tile: x in [tx0, tx0 + 8.6), y in [ty0, ty1 + 0.6] */
    const int sx1 = tx0 + kTileW + 1, sy1 = ty1 + 1;
    bool mine;
    int r0, r1, r2, r3;
    lane_rect(P, K, lane, mine, r0, r1, r2, r3);
    const int *sd = P.light_side[l];
    const bool in_left = tx0 >= sd[0] && tx0 <= sd[1];   /* light on the ">= tx0" side of the left plane */
    const bool in_right = sx1 >= sd[2] && sx1 <= sd[3];  /* light on the "<= sx1" side of the right plane */
    const bool in_top = ty0 >= sd[4] && ty0 <= sd[5];
    const bool in_bottom = sy1 >= sd[6] && sy1 <= sd[7];
    const bool culled = mine && ((in_left && r2 <= tx0) || (in_right && r0 >= sx1) || (in_top && r3 <= ty0) || (in_bottom && r1 >= sy1));
    const unsigned long long active = __ballot(true);
    return (uint32_t)__ballot(!culled) | ~(uint32_t)active;
}

/* Scene.testVisibility — rt/scene.d:62-78 */
template <int LEVELS, bool PO>
DEV bool test_visibility(const Ctx &cx, D3 from, D3 to, uint32_t node_mask, bool ground_only)
{
    const D3 raw = to - from;
    if (!PO && ground_only) { /* wave-uniform */
        /* plane_points_away for the ground plane (identity matrix, zero offset) */
        const bool sane = (fabs(raw.x) < 1e150) & (fabs(raw.z) < 1e150);
        const bool up = (raw.y > 1e-150) & (raw.y < 1e150), down = (raw.y < -1e-150) & (raw.y > -1e150);
        if (__all(sane & (((from.y > cx.ground_y) & up) | ((from.y < cx.ground_y) & down)))) return true;
    }
    const uint32_t nn = cx.n_nodes;
    uint32_t n = next_node(node_mask, 0);
    if constexpr (PO) { /* every lane's ray provably leaves the leading planes behind: no ray needed yet */
        while (n < nn && __all(plane_points_away(cx.nodes + n, from, raw))) n = next_node(node_mask, n + 1);
        if (n >= nn) return true;
    }
    const D3 dir = normalized(raw);
    const RayW ray = make_ray(from, dir);
    Hit temp;
    temp.dist = mag(raw);
    for (; n < nn; n = next_node(node_mask, n + 1)) /* scalar loop, file order */
        if (node_intersect<LEVELS, kBool>(cx, cx.nodes + n, ray, temp)) return false;
    return true;
}

/* ------------------------------------------------------------------ */
/* textures — rt/texture.d, rt/bitmap.d                                  */
/* ------------------------------------------------------------------ */

/* Bitmap.getFilteredPixel — rt/bitmap.d:48-63 */
DEV F3 bitmap_filtered(const float4 *texels, uint32_t width, uint32_t height, float x, float y)
{
    /* isInvalidPos(cast(size_t)x, cast(size_t)y): x, y are >= 0 or NaN here */
    if (!(x < (float)width) | !(y < (float)height) | (width == 0) | (height == 0))
        return mkf(1.0f, 0.0f, 0.0f); /* NamedColors.red */
    const float fx = floorf(x), fy = floorf(y);
    const uint32_t tx = (uint32_t)fx, ty = (uint32_t)fy;
    const uint32_t txn = tx + 1 == width ? 0 : tx + 1;
    const uint32_t tyn = ty + 1 == height ? 0 : ty + 1;
    const float p = x - fx, q = y - fy;
    const float4 c00 = texels[(size_t)ty * width + tx], c10 = texels[(size_t)ty * width + txn];
    const float4 c01 = texels[(size_t)tyn * width + tx], c11 = texels[(size_t)tyn * width + txn];
    const float w00 = (1.0f - p) * (1.0f - q), w10 = p * (1.0f - q), w01 = (1.0f - p) * q, w11 = p * q;
    return mkf(c00.x, c00.y, c00.z) * w00 + mkf(c10.x, c10.y, c10.z) * w10 + mkf(c01.x, c01.y, c01.z) * w01 +
           mkf(c11.x, c11.y, c11.z) * w11;
}

/* A node's shading inputs in registers (DevMat, c2rt_device.h). */
struct Mat {
    int shader_type, tex_type, tex;
    float strength;
    F3 color;
    double exponent;
    uint32_t td[8];
};

/* One scalar record load per DISTINCT closest node of the wave (usually one or
 * two) instead of a node -> shader -> texture chain of per-lane loads. */
DEV void load_mat(const DevNode *nodes, int closest, Mat &m)
{
    /* (lanes without a hit keep an indeterminate record: they return the environment colour before anything
     * reads it — thirty zeroing moves per sample otherwise) */
    m.tex_type = -1;
    bool todo = closest >= 0;
    while (todo) {
        const int u = __builtin_amdgcn_readfirstlane(closest);
        if (closest == u) {
            const DevMat *M = &nodes[u].mat; /* wave-uniform address: scalar loads */
            m.shader_type = M->shader_type;
            m.tex_type = M->tex_type;
            m.tex = M->tex;
            m.strength = M->strength;
            m.color = ldf3(M->color);
            m.exponent = M->exponent;
#pragma unroll
            for (int i = 0; i < 8; ++i) m.td[i] = M->texdata[i];
            todo = false;
        }
    }
}

DEV F3 tex_color(const RenderParams &P, const Mat &m, double u, double v)
{
    const int type = m.tex_type;
    if (type == C2RT_TEX_CHECKER) { /* Checker.getTexColor — rt/texture.d:36-54 */
        const double size = __hiloint2double((int)m.td[7], (int)m.td[6]);
        const int x = d2i_x86(floor(u / size));
        const int y = d2i_x86(floor(v / size));
        const int white = (int)((uint32_t)x + (uint32_t)y) % 2;
        return white ? mkf(__uint_as_float(m.td[3]), __uint_as_float(m.td[4]), __uint_as_float(m.td[5]))
                     : mkf(__uint_as_float(m.td[0]), __uint_as_float(m.td[1]), __uint_as_float(m.td[2]));
    } else if (type == C2RT_TEX_PROCEDURE2) { /* Procedure2.getTexColor — rt/texture.d:77-86 */
        const DevTex *T = P.textures + m.tex;
        F3 result = mkf(0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            result = result + (ldf3(T->color + 3 * i) * (float)c2_sin(u * T->param[i]) +
                               ldf3(T->color + 9 + 3 * i) * (float)c2_sin(v * T->param[3 + i]));
        return result;
    } else { /* BitmapTexture.getTexColor — rt/texture.d:116-126 */
        const double s = (double)__uint_as_float(m.td[2]);
        u *= s;
        v *= s;
        u = u - floor(u);
        v = v - floor(v);
        const uint32_t w = m.td[0], hgt = m.td[1];
        const float tx = (float)u * (float)w;
        const float ty = (float)v * (float)hgt;
        const uint64_t offset = (uint64_t)m.td[4] | ((uint64_t)m.td[5] << 32);
        return bitmap_filtered(reinterpret_cast<const float4 *>(P.texels) + offset, w, hgt, tx, ty);
    }
}

/* ------------------------------------------------------------------ */
/* shading — rt/shader.d:67-105,197-250                                  */
/* ------------------------------------------------------------------ */

/* MLC ("multi-light"): the instance for scenes with MORE THAN ONE light — the light loop, and the
 * culling masks of lights 1.. derived per sample.  Scenes with at most one light (every scene the
 * reference ships) run the MLC = false instance, in which the one light is straight-line code:
 * nothing of the hit has to stay live for a next light, so the hit point, normal and view direction
 * die before the shadow test (below). */
template <int LEVELS, bool MLC, bool PO>
DEV F3 shade(const RenderParams &P, const Ctx &cx, const Mat &mat, D3 rd, const Hit &h, uint32_t &shadow_rays)
{
    const bool phong = mat.shader_type == C2RT_SHADER_PHONG;
    const D3 N = dot(rd, h.n) < 0 ? h.n : -h.n; /* faceforward — rt/imported_types.d:69-73 */
    const F3 diffuse = mat.tex_type >= 0 ? tex_color(P, mat, h.u, h.v) : mat.color;
    F3 lightContrib = mkf(P.ambient[0], P.ambient[1], P.ambient[2]);
    F3 specular = mkf(0, 0, 0);
    const uint32_t nl = MLC ? P.n_lights : (P.n_lights ? 1u : 0u);
    for (uint32_t l = 0; l < nl; ++l) {
        const DevLight *L = P.lights + l;
        F3 avgColor = mkf(0, 0, 0), avgSpecular = mkf(0, 0, 0);
        if (L->lit) {
            const D3 lightPos = ld3(L->pos);
            shadow_rays += 1;
            /* Everything the lit branch reads from the hit is evaluated BEFORE the visibility test (same
             * operations on the same operands, so the same bits; a shadowed sample wastes two
             * normalisations): across the test — the register peak of the kernel, a CSG walk inside
             * a node loop — only cosTheta, baseLight and cosGamma stay live instead of p, N and rd. */
            const D3 from = h.p + N * 1e-6;
            /* squaredMagnitude(p - lightPos) == squaredMagnitude(lightPos - p) bit for bit (IEEE a - b is
             * exactly -(b - a), and the squares drop the sign): one vector, one sum of squares for both
             * normalize(lightPos - p) and the 1/r^2 term */
            const D3 lv = lightPos - h.p;
            const double r2 = sqmag(lv);
            const double rinv = 1.0 / sqrt(r2);
            const D3 lightDir = mk(lv.x * rinv, lv.y * rinv, lv.z * rinv);
            const double cosTheta = dot(lightDir, N);
            const F3 baseLight = ldf3(L->color) / (float)r2;
            double cosGamma = 0;
            if (phong) {
                /* reflect(-lightDir, N) — rt/imported_types.d:62-67 */
                const D3 ml = -lightDir;
                const D3 R = normalized(ml - N * (2 * dot(ml, N)));
                cosGamma = dot(R, -rd);
            }
            if (test_visibility<LEVELS, PO>(cx, from, lightPos, l == 0 ? cx.shadow_mask0 : (MLC ? shadow_cull_mask(P, cx.kargs, cx.block, cx.lane, l) : 0xFFFFFFFFu), l == 0 && cx.shadow_ground_only)) {
                if (cosTheta > 0) avgColor = avgColor + baseLight * (float)cosTheta;
                if (phong & (cosGamma > 0))
                    avgSpecular = avgSpecular + baseLight * (float)c2_pow(cosGamma, mat.exponent) * mat.strength;
            }
        }
        /* `/ numSamples` with numSamples == 1 (rt/light.d:56-59) is exact */
        lightContrib = lightContrib + avgColor;
        specular = specular + avgSpecular;
    }
    const F3 res = diffuse * lightContrib;
    return phong ? res + specular : res;
}

/* ------------------------------------------------------------------ */
/* camera — rt/camera.d:123-173                                          */
/* ------------------------------------------------------------------ */

/* Counter-based RNG for the lens samples (build-defined: the reference draws from libc rand(),
 * util/random.d:19-28, which is not reproducible — SURVEY.md F5).  32-bit multiply-xorshift
 * finaliser ("lowbias32"); the key folds (seed, pixel, tap) once per sample loop, a draw is one
 * hash of key + golden-ratio * counter.  The CPU checker restates it statement for statement. */
DEV uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
DEV uint32_t rng_key(uint64_t seed, uint64_t pixel, uint32_t tap)
{
    uint32_t k = hash32((uint32_t)(seed >> 32) ^ 0x243f6a88u);
    k = hash32(k ^ (uint32_t)seed);
    k = hash32(k + (uint32_t)(pixel >> 32));
    k = hash32(k ^ (uint32_t)pixel);
    return hash32(k + tap);
}
/* uniform in [0, 1) with 32 random bits (rand()/RAND_MAX has 31) */
DEV double rng_uniform(uint32_t key, uint32_t sample, uint32_t dim)
{
    return (double)hash32(key + 0x9e3779b9u * (sample * 16u + dim + 1u)) * 0x1p-32;
}

struct Rng { uint32_t key, sample, dim; };
DEV double rng_next(Rng &r) { return rng_uniform(r.key, r.sample, r.dim++); }

/* (sin, cos)(2 pi u) for u in [0, 1) without libm, so that a CPU checker and the kernel produce the
 * same bits (device sincos and glibc sin/cos differ by an ulp, which flips z-fights between
 * coincident planes): exact reduction of 4u to a quadrant q and a fraction g in [0, 0.5] (mirrored
 * about the octant boundary), then the Taylor polynomials in theta = g * (pi/2) <= pi/4 — sin to
 * theta^17, cos to theta^16, truncation error < 1e-17 — evaluated by Horner in fp64 with contraction
 * off (integer and IEEE +, * only: any IEEE-754 host reproduces it). */
DEV void lens_sincos2pi(double u, double &sn, double &cs)
{
    const double t = u * 4.0;          /* exact */
    const int q = (int)t;              /* 0..3 */
    const double f = t - (double)q;    /* exact, [0, 1) */
    const bool mirror = f > 0.5;
    const double g = mirror ? 1.0 - f : f; /* exact */
    const double th = g * 0x1.921fb54442d18p+0; /* pi/2 */
    const double z = th * th;
    double ps = 0x1.952c77030ad4ap-49;             /* +1/17! */
    ps = -0x1.ae7f3e733b81fp-41 + z * ps;          /* -1/15! */
    ps = 0x1.6124613a86d09p-33 + z * ps;           /* +1/13! */
    ps = -0x1.ae64567f544e4p-26 + z * ps;          /* -1/11! */
    ps = 0x1.71de3a556c734p-19 + z * ps;           /* +1/9! */
    ps = -0x1.a01a01a01a01ap-13 + z * ps;          /* -1/7! */
    ps = 0x1.1111111111111p-7 + z * ps;            /* +1/5! */
    ps = -0x1.5555555555555p-3 + z * ps;           /* -1/3! */
    const double s = th + th * (z * ps);
    double pc = 0x1.ae7f3e733b81fp-45;             /* +1/16! */
    pc = -0x1.93974a8c07c9dp-37 + z * pc;          /* -1/14! */
    pc = 0x1.1eed8eff8d898p-29 + z * pc;           /* +1/12! */
    pc = -0x1.27e4fb7789f5cp-22 + z * pc;          /* -1/10! */
    pc = 0x1.a01a01a01a01ap-16 + z * pc;           /* +1/8! */
    pc = -0x1.6c16c16c16c17p-10 + z * pc;          /* -1/6! */
    pc = 0x1.5555555555555p-5 + z * pc;            /* +1/4! */
    pc = -0.5 + z * pc;                            /* -1/2! */
    const double c = 1.0 + z * pc;
    const double a = mirror ? c : s, b = mirror ? s : c; /* sin, cos of the in-quadrant angle */
    /* quadrant rotation: q=0 (a, b), 1 (b, -a), 2 (-a, -b), 3 (-b, a) */
    sn = (q & 1) ? b : a;
    cs = (q & 1) ? a : b;
    if (q == 2 || q == 3) sn = -sn;
    if (q == 1 || q == 2) cs = -cs;
}

template <int DOF>
DEV void screen_ray(const RenderParams &P, double x, double y, int offset, Rng &rng, D3 &orig, D3 &dir)
{
    const c2rt_camera_frame &cam = P.cam;
    const D3 pos = ld3(cam.pos), upLeft = ld3(cam.up_left);
    /* (upRight - upLeft) and (downLeft - upLeft) are per-frame constants: the host
     * does the same two IEEE subtractions once (c2rt_api.cpp fill_params) */
    const D3 target = upLeft + ld3(P.cam_du) * (x / cam.frame_width) + ld3(P.cam_dv) * (y / cam.frame_height);
    orig = pos;
    dir = target - pos; /* un-normalised: raytrace() normalises it when a node needs it */
    if constexpr (DOF) {
        const D3 raw0 = dir;
        dir = normalized(raw0);
        const D3 rightDir = ld3(cam.right_dir);
        if (DOF == 2 && offset != 0) orig = orig + rightDir * (offset > 0 ? +cam.stereo_separation : -cam.stereo_separation);
        if (!cam.dof) { dir = raw0; return; }
        const double cosTheta = dot(dir, ld3(cam.front_dir));
        const double M = cam.focal_plane_dist / cosTheta;
        const D3 T = orig + dir * M;
        /* unitDiscSample — rt/camera.d:258-269: (sin, cos)(U1 * 2 pi) * sqrt(U2) */
        const double u1 = rng_next(rng);
        const double rad = sqrt(rng_next(rng));
        double sn, cs;
        lens_sincos2pi(u1, sn, cs);
        double dx = sn * rad, dy = cs * rad;
        dx *= cam.disc_multiplier;
        dy *= cam.disc_multiplier;
        orig = pos + rightDir * dx + ld3(cam.up_dir) * dy;
        if (DOF == 2 && offset != 0) orig = orig + rightDir * (offset > 0 ? +cam.stereo_separation : -cam.stereo_separation);
        dir = T - orig; /* un-normalised */
    }
}

/* ------------------------------------------------------------------ */
/* renderer — rt/renderer.d:223-376                                      */
/* ------------------------------------------------------------------ */

struct Counters { uint32_t primary, shadow; };

/* trace + raytrace_impl — rt/renderer.d:325-376 (primary rays have depth 0) */
template <int LEVELS, bool MLC, bool PO>
DEV F3 raytrace(const RenderParams &P, const Ctx &cx, D3 o, D3 raw, Counters &cnt, c2rt_trace_result *probe)
{
    cnt.primary += 1;
    const uint32_t nn = P.n_nodes;
    uint32_t n = next_node(cx.primary_mask, 0);
    /* planes-only scenes: every lane's ray provably leaves the leading planes behind (sky tiles);
     * `raw` is the screen ray before normalisation (rt/camera.d:144-147) */
    if constexpr (PO) {
        if (!probe) {
            while (n < nn && __all(plane_points_away(P.nodes + n, o, raw))) n = next_node(cx.primary_mask, n + 1);
            if (n >= nn) return mkf(0, 0, 0); /* Environment.getEnvironment — rt/environment.d:7-10 */
        }
    }
    const D3 d = normalized(raw);
    Hit best;
    best.dist = 1e99;
    if (probe) { /* the probe reports the record even without a hit; a frame only reads it after one */
        best.p = best.n = mk(0, 0, 0);
        best.u = best.v = best.w = 0;
        best.g = -1;
    }
    best.uv_pending = false;
    best.axis_n = false;
    int closest = -1;
    Mat mat;
    if (!probe && cx.primary_ground_only) { /* wave-uniform */
        /* Ground tile (most of a frame that looks at a floor): the node loop collapses to Node.intersect +
         * Plane.intersect on ONE node known to be a Plane under the identity matrix with zero offset
         * (RenderParams::ground_node) — the same operations on the same operands as the general path
         * (node_intersect, plane_intersect), minus the loop, the flag tests, the per-lane record merge and
         * the per-lane shading-input fetch: the node is the same for every lane, so its DevMat stays in
         * scalar registers. */
        const DevNode *N = P.nodes + P.ground_node;
        const double len = mag(d);                       /* make_ray: |d|, d * (1/|d|) */
        const double inv = 1.0 / len;
        const D3 dn = mk(d.x * inv, d.y * inv, d.z * inv);
        const double y = N->g.p[0], limit = N->g.p[1];
        const bool away = ((o.y > y) & (dn.y > -1e-9)) | ((o.y < y) & (dn.y < 1e-9));
        const double mult = (o.y - y) / -dn.y;
        const D3 p = o + dn * mult;
        const bool miss = away | (mult > 1e99 * len) | (fabs(p.x) > limit) | (fabs(p.z) > limit);
        if (!miss) {
            closest = P.ground_node;
            best.p = p;
            best.n = mk(0, 1, 0);
            best.u = p.x;
            best.v = p.z;
            best.g = N->geom;
        }
        const DevMat *M = &N->mat;
        mat.shader_type = M->shader_type;
        mat.tex_type = M->tex_type;
        mat.tex = M->tex;
        mat.strength = M->strength;
        mat.color = ldf3(M->color);
        mat.exponent = M->exponent;
#pragma unroll
        for (int i = 0; i < 8; ++i) mat.td[i] = M->texdata[i];
    } else {
        const RayW ray = make_ray(o, d);
        for (uint32_t nx; n < nn; n = nx) { /* scalar loop, file order */
            nx = next_node(cx.primary_mask, n + 1);
            if (node_intersect<LEVELS, kFull>(cx, P.nodes + n, ray, best, nx >= nn && !probe)) closest = (int)n;
        }
        /* Sphere u,v are read only by textured shaders (and the probe) */
        load_mat(P.nodes, closest, mat);
        if (closest >= 0 && best.uv_pending && (probe || mat.tex_type >= 0)) finish_uv(best);
    }
    if (probe) {
        probe->closest_node = closest;
        probe->leaf_geom = closest >= 0 ? best.g : -1;
        probe->p[0] = best.p.x; probe->p[1] = best.p.y; probe->p[2] = best.p.z;
        probe->normal[0] = best.n.x; probe->normal[1] = best.n.y; probe->normal[2] = best.n.z;
        probe->dist = best.dist; probe->u = best.u; probe->v = best.v;
        probe->ray_orig[0] = o.x; probe->ray_orig[1] = o.y; probe->ray_orig[2] = o.z;
        probe->ray_dir[0] = d.x; probe->ray_dir[1] = d.y; probe->ray_dir[2] = d.z;
    }
    if (closest < 0) return mkf(0, 0, 0); /* Environment.getEnvironment — rt/environment.d:7-10 */
    return shade<LEVELS, MLC, PO>(P, cx, mat, d, best, cnt.shadow);
}

/* adjustSaturation + combineStereo — rt/color.d:10-15,77-83 */
DEV F3 desaturate(F3 c, float amount)
{
    const float mid = (c.r + c.g + c.b) / 3;
    return mkf(c.r * amount + mid * (1 - amount), c.g * amount + mid * (1 - amount), c.b * amount + mid * (1 - amount));
}
DEV F3 combine_stereo(F3 l, F3 r)
{
    l = desaturate(l, 0.25f);
    r = desaturate(r, 0.25f);
    return l * mkf(1, 0, 0) + r * mkf(0, 1, 1);
}

/* renderSample — rt/renderer.d:254-313 */
template <int LEVELS, int DOF, bool MLC, bool PO>
DEV F3 render_sample(const RenderParams &P, const Ctx &cx, double x, double y, int dx, int dy, uint64_t pixel, uint32_t tap,
                     Counters &cnt, c2rt_trace_result *probe)
{
    Rng rng = {DOF ? rng_key(P.seed, pixel, tap) : 0u, 0, 0};
    D3 o, d;
    if constexpr (!DOF) {
        screen_ray<false>(P, x, y, 0, rng, o, d);
        return raytrace<LEVELS, MLC, PO>(P, cx, o, d, cnt, probe);
    } else {
        /* renderSampleDof / renderSampleStereo / renderSampleDefault (rt/renderer.d:270-313)
         * as ONE loop around ONE trace call site (five inlined copies of the tracer made
         * this instance five times the size of the others): lens samples x eyes, in the
         * reference's order, with the random draws in its order. */
        /* DOF = 1: depth of field on a mono camera (what zaphod.sdl asks for): one eye, no offset, no
         * anaglyph merge — all of it compile-time; DOF = 2: a stereo camera, with or without depth of field */
        const bool stereo = DOF == 2 && P.cam.stereo_separation != 0;
        const bool dof = P.cam.dof != 0;
        const uint32_t ns = dof ? P.cam.num_samples : 1u;
        const int eyes = stereo ? 2 : 1;
        F3 average = mkf(0, 0, 0), sample = mkf(0, 0, 0);
        for (uint32_t i = 0; i < ns; ++i) {
            rng.sample = i;
            rng.dim = 0;
            for (int e = 0; e < eyes; ++e) {
                double sx = x, sy = y;
                if (dof) {
                    const double jx = rng_next(rng), jy = rng_next(rng);
                    sx = x + jx * dx;
                    sy = y + jy * dy;
                }
                screen_ray<DOF>(P, sx, sy, stereo ? (e == 0 ? -1 : +1) : 0, rng, o, d);
                const F3 c = raytrace<LEVELS, MLC, PO>(P, cx, o, d, cnt, e == 0 ? probe : nullptr);
                sample = e == 0 ? c : combine_stereo(sample, c);
            }
            if (!dof) return sample;
            average = average + sample;
        }
        return average / (float)ns;
    }
}

/* AA kernel — rt/renderer.d:235-242 */
__constant__ double k_aa_x[5] = {0.0, 0.3, 0.6, 0.0, 0.6};
__constant__ double k_aa_y[5] = {0.0, 0.3, 0.0, 0.6, 0.6};

/*
 * Frame kernel: Renderer.renderRT passes 2 and 3b (rt/renderer.d:133-142,
 * 183-186) fused — all taps of a pixel are accumulated in registers in the
 * reference's order and the pixel is written once (12 B of HBM traffic per
 * pixel).  One workgroup = one wavefront = one 8x8 tile.
 */
template <int LEVELS, int DOF, bool MLC, bool PO, bool CNT>
DEV void render_tile(const RenderParams &P, KArgs K, const uint32_t b)
{
    extern __shared__ __align__(16) char lds_all[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    char *lds = lds_all + (size_t)wave * P.csg_cap * kCsgLdsPerEntry;

    /* XCD-aware block -> tile: blocks b and b+8 share an XCD (round-robin
     * dispatch), so XCD x gets tile rows x, x+8, x+16, ... and walks them
     * left to right.  A block is kWavesPerBlock horizontally adjacent 8x8
     * tiles, one per wavefront. */
#if C2RT_XCD_SWIZZLE
    const uint32_t xcd = b & 7u, j = b >> 3;
    /* row groups are walked starting at P.row_group_start (where the boxed nodes begin on screen):
     * the expensive tiles are dispatched first and the launch ends on cheap ones */
    const uint32_t groups = (P.tiles_y + 7u) / 8u;
    uint32_t grp = j / P.blocks_x + P.row_group_start;
    if (grp >= groups) grp -= groups;
    const uint32_t trow = grp * 8u + xcd, bcol = j % P.blocks_x;
#else
    const uint32_t trow = b / P.blocks_x, bcol = b % P.blocks_x;
#endif
    if (trow >= P.tiles_y) return;
    const uint32_t tcol = bcol * kWavesPerBlock + wave;
#if C2RT_TILE_STATS
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
#endif

    /* Which nodes can this tile's primary rays reach, and which can occlude its
     * shadow rays towards the first light?  Lane n tests node n's rectangle and a
     * ballot makes the wave-uniform masks — before any lane leaves, so that every
     * node has its lane. */
    uint32_t pmask = 0xFFFFFFFFu, smask0 = 0xFFFFFFFFu;
    bool ground_only = false, primary_ground = false;
    if constexpr (!DOF) {
        if (P.n_cull) {
            int tx0, ty0, ty1;
            tile_bounds(P, b, tx0, ty0, ty1);
            bool mine;
            int r0, r1, r2, r3;
            lane_rect(P, K, lane, mine, r0, r1, r2, r3);
            /* the node's hull (RenderParams::cull_hull): outside one padded edge with all four tile corners.
             * Only where some node's rectangle meets the tile at all (most tiles see the ground only). */
            bool off_hull = false;
            const bool rect_meets = mine && !(r2 <= tx0 || r0 >= tx0 + kTileW || r3 <= ty0 || r1 > ty1);
            if (__ballot(rect_meets)) {
                typedef const float __attribute__((address_space(4))) *KFlt;
                KFlt hl = (KFlt)((const char __attribute__((address_space(4))) *)K + __builtin_offsetof(RenderParams, cull_hull)) +
                          (mine ? lane : 0) * (kHullEdges * 3);
                const float X0 = (float)tx0, X1 = (float)(tx0 + kTileW), Y0 = (float)ty0, Y1 = (float)(ty1 + 1);
#pragma unroll
                for (int e = 0; e < kHullEdges; ++e) {
                    const float a = hl[3 * e], bb = hl[3 * e + 1], c = hl[3 * e + 2];
                    const float ax0 = a * X0, ax1 = a * X1, by0 = bb * Y0, by1 = bb * Y1;
                    const float worst = fmaxf(fmaxf(ax0, ax1) + fmaxf(by0, by1), -3.0e38f) + c; /* the corner deepest inside */
                    off_hull |= worst < -0.25f; /* (0.25 px: float evaluation error at |coordinates| < 1e6) */
                }
            }
            pmask = (uint32_t)__ballot(!mine || !(r2 <= tx0 || r0 >= tx0 + kTileW || r3 <= ty0 || r1 > ty1 || off_hull));
            smask0 = shadow_cull_mask(P, K, b, lane, 0);
            /* Ground-plane refinement (RenderParams::ground_node): if this tile's primary rays can only
             * reach the ground plane, all its hit points lie inside the tile's footprint on that plane
             * — the convex image of the pixel rectangle (+1 px all round; the AA taps reach 0.6 px),
             * provided all four corner rays meet the plane in front of the eye — and a node whose
             * shadow rectangle misses the footprint's bounding rectangle cannot occlude any of them. */
            const int gnode = P.ground_node;
            const uint32_t nn = P.n_nodes;
            if (gnode >= 0 && nn <= 32u && (pmask & (0xFFFFFFFFu >> (32u - nn))) == (1u << gnode)) {
                primary_ground = true;
                const D3 pos = ld3(P.cam.pos), ul = ld3(P.cam.up_left), du = ld3(P.cam_du), dv = ld3(P.cam_dv);
                const double gy = P.ground_y;
                /* lane k & 3 intersects corner k's ray with the plane; lanes 0..3 are then read back */
                const int k = lane & 3;
                const double sx = (k & 1) ? (double)(tx0 + kTileW + 1) : (double)(tx0 - 1);
                const double sy = (k & 2) ? (double)(ty1 + 2) : (double)(ty0 - 1);
                const D3 dir = ul + du * (sx / P.cam.frame_width) + dv * (sy / P.cam.frame_height) - pos;
                const double t = (gy - pos.y) / dir.y;
                bool ok = (__ballot((t > 0) & (t < 1e300)) & 0xFull) == 0xFull;
                const double hx = pos.x + dir.x * t, hz = pos.z + dir.z * t;
                const double hx0 = read_lane(hx, 0), hx1 = read_lane(hx, 1), hx2 = read_lane(hx, 2), hx3 = read_lane(hx, 3);
                const double hz0 = read_lane(hz, 0), hz1 = read_lane(hz, 1), hz2 = read_lane(hz, 2), hz3 = read_lane(hz, 3);
                const double fx0 = fmin(fmin(hx0, hx1), fmin(hx2, hx3)), fx1 = fmax(fmax(hx0, hx1), fmax(hx2, hx3));
                const double fz0 = fmin(fmin(hz0, hz1), fmin(hz2, hz3)), fz1 = fmax(fmax(hz0, hz1), fmax(hz2, hz3));
                const double padx = 1e-9 * (fabs(fx0) + fabs(fx1)), padz = 1e-9 * (fabs(fz0) + fabs(fz1));
                ok = ok & (fx0 <= fx1) & (fz0 <= fz1) & (fabs(fx0) < 1e300) & (fabs(fx1) < 1e300) & (fabs(fz0) < 1e300) & (fabs(fz1) < 1e300);
                const double *sr = P.shadow_rects + 4 * (mine ? lane : 0);
                const bool apart = (sr[1] < fx0 - padx) | (sr[0] > fx1 + padx) | (sr[3] < fz0 - padz) | (sr[2] > fz1 + padz);
                smask0 &= ~(uint32_t)__ballot(ok & mine & apart);
                ground_only = (smask0 & (0xFFFFFFFFu >> (32u - nn))) == (1u << gnode);
            }
        }
    }

    /* a scene whose only node is the ground plane (zaphod.sdl, lecture4.sdl): every tile is a ground
     * tile, wherever its rays start (depth of field, stereo) */
    if (P.n_nodes == 1u && P.ground_node == 0) primary_ground = true;

    const uint32_t x = tcol * kTileW + (lane % kTileW);
    const uint32_t lr0 = trow * kTileH + (lane / kTileW); /* row within this launch */
    if (x >= P.width || lr0 >= P.local_rows) return;
    const uint32_t lr = lr0 + P.row_offset;          /* local row */

    /* local row -> frame row under interleaved strips */
    uint32_t y = lr;
    if (P.strip_world > 1) {
        const uint32_t sh = P.strip_height;
        y = ((lr / sh) * P.strip_world + P.strip_rank) * sh + lr % sh;
    }

    Ctx cx;
    cx.geoms = P.geoms;
    cx.nodes = P.nodes;
    cx.n_nodes = P.n_nodes;
    cx.kargs = K;
    cx.lds = lds;
    cx.lane = lane;
    cx.csg_cap = (int)P.csg_cap;
    cx.overflow = false;
    cx.trunc_counter = CNT ? P.ray_counters + 2 : nullptr;
#if C2RT_TILE_STATS
    cx.lane_stats = P.tile_stats ? reinterpret_cast<unsigned long long *>(P.tile_stats + 2 * (size_t)P.tiles_x * P.tiles_y) : nullptr;
#endif
    cx.block = b;
    cx.primary_mask = pmask;
    cx.shadow_mask0 = smask0;
    cx.shadow_ground_only = ground_only;
    cx.primary_ground_only = primary_ground;
    cx.ground_y = P.ground_y;
    Counters cnt = {0, 0};
    /* prepassOnly (rt/renderer.d:110-130): the pixel shows the sample of the
     * top-left pixel of its 16x16 block inside its bucket */
    uint32_t sx = x, sy = y;
    int jdx = 1, jdy = 1; /* renderPixelNoAA's dx, dy: the extent depth-of-field jitter spans */
    if (P.prepass_bucket) {
        const uint32_t bs = P.prepass_bucket;
        const uint32_t bx = x / bs * bs, by = y / bs * bs;
        sx = bx + ((x - bx) & ~15u);
        sy = by + ((y - by) & ~15u);
        /* the 16x16 block is clipped by its bucket, the bucket by the frame (rt/renderer.d:113-119, 208) */
        const uint32_t bx1 = bx + bs < P.width ? bx + bs : P.width, by1 = by + bs < P.height ? by + bs : P.height;
        jdx = (int)(sx + 16u < bx1 ? 16u : bx1 - sx);
        jdy = (int)(sy + 16u < by1 ? 16u : by1 - sy);
    }
    const uint64_t pixel = (uint64_t)sy * P.width + sx;
    const uint32_t ntaps = P.taps;

    /* renderPixelNoAA — rt/renderer.d:223-228; renderPixelAA — :233-251.
     * Tap 0 has offset (0, 0): x + 0.0 == x, and 0 + c == c for the first
     * sample, so one loop covers both passes. */
    F3 accum = mkf(0, 0, 0);
#pragma unroll 1
    for (uint32_t s = 0; s < ntaps; ++s) {
        const F3 c = render_sample<LEVELS, DOF, MLC, PO>(P, cx, (double)sx + k_aa_x[s], (double)sy + k_aa_y[s], jdx, jdy, pixel, s, cnt, nullptr);
        accum = s == 0 ? c : accum + c;
    }
    if (ntaps > 1) accum = accum / (float)ntaps; /* `accum / 5`: Color / float */
#if C2RT_TILE_STATS
    if (P.tile_stats && lane == (int)__builtin_ctzll(__ballot(true))) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - stamp0;
        const uint32_t tile = trow * P.tiles_x + tcol;
        P.tile_stats[2 * tile] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
        /* bit 0: primary rays reach the ground only, bit 1: and so do the shadow rays; bits 8..: primary mask */
        P.tile_stats[2 * tile + 1] = (primary_ground ? 1u : 0u) | (ground_only ? 2u : 0u) | (pmask << 8);
    }
#endif

    if constexpr (LEVELS >= 2) {
        /* some lane's nested hit lists outgrew the stack: nothing of this tile is kept; the
         * full-capacity launch that follows renders the tiles on this list */
        if (__ballot(cx.overflow)) {
            if (lane == (int)__builtin_ctzll(__ballot(true))) {
                const uint32_t slot = atomicAdd(P.retry_list, 1u);
                if (slot < P.retry_max) P.retry_list[1 + slot] = b;
            }
            return;
        }
    }

    float *px = P.out + ((size_t)(P.frame_rows ? y : lr) * P.width + x) * 3;
    px[0] = accum.r;
    px[1] = accum.g;
    px[2] = accum.b;

    /* CNT: the instances launched when rays are being counted (opts->count_rays).  The production
     * instances (CNT = false) never read `cnt` nor the truncation counter: the compiler drops the
     * per-lane counters and the whole bookkeeping from them. */
    if constexpr (CNT) {
        atomicAdd(P.ray_counters + 0, (unsigned long long)cnt.primary);
        atomicAdd(P.ray_counters + 1, (unsigned long long)cnt.shadow);
    }
}

/* One tile per workgroup; in retry mode (RenderParams::retry_mode: the full-capacity relaunch of
 * the nested-CSG instances) a fixed grid walks the list of tiles whose hit stacks overflowed.
 * Either way the tile code is inlined once. */
template <int LEVELS, int DOF, bool MLC, bool PO, bool CNT>
DEV void render_body(const RenderParams &P, KArgs K)
{
    if constexpr (LEVELS >= 2) {
        uint32_t i = blockIdx.x;
        do {
            uint32_t b = i;
            if (P.retry_mode) {
                const uint32_t listed = P.retry_list[0];
                if (i >= (listed < P.retry_max ? listed : P.retry_max)) break;
                b = P.retry_list[1 + i];
            }
            render_tile<LEVELS, DOF, MLC, PO, CNT>(P, K, b);
            i += gridDim.x;
        } while (P.retry_mode);
    } else {
        render_tile<LEVELS, DOF, MLC, PO, CNT>(P, K, blockIdx.x);
    }
}

template <int LEVELS, int DOF, bool MLC, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(LEVELS, DOF, MLC) render_kernel(const RenderParams P)
{
    render_body<LEVELS, DOF, MLC, false, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

/* The depth-of-field / stereo instance carries the lens sampling state on top of
 * the tracer's and has its own register budget (C2RT_OCC_DOF). */
template <int LEVELS, bool MLC, int MODE, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(LEVELS, MODE, MLC) render_kernel_dof(const RenderParams P)
{
    render_body<LEVELS, MODE, MLC, false, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

/* Scenes made of axis planes only (RenderParams::planes_only — lecture4.sdl, zaphod.sdl): the
 * instances in which a plane's miss is decided before the ray is normalised (plane_points_away). */
template <int DOF, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(0, DOF, false) render_kernel_planes(const RenderParams P)
{
    render_body<0, DOF, false, true, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr()); /* at most one light (launch_render_level); planes have no boxes, hence no culling masks */
}

/* renderPixel — rt/renderer.d:46-57: one lane, one sample, full trace result */
template <int LEVELS, int DOF>
__global__ void __launch_bounds__(kWave) probe_kernel(const RenderParams P)
{
    extern __shared__ __align__(16) char lds[];
    if (threadIdx.x != 0) return;
    Ctx cx;
    cx.geoms = P.geoms;
    cx.nodes = P.nodes;
    cx.n_nodes = P.n_nodes;
    cx.kargs = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    cx.lds = lds;
    cx.lane = 0;
    cx.csg_cap = (int)P.csg_cap;
    cx.overflow = false;
    cx.trunc_counter = nullptr;
#if C2RT_TILE_STATS
    cx.lane_stats = nullptr;
#endif
    cx.block = 0;
    cx.primary_mask = 0xFFFFFFFFu;
    cx.shadow_mask0 = 0xFFFFFFFFu;
    cx.shadow_ground_only = false;
    cx.primary_ground_only = false;
    cx.ground_y = 0;
    Counters cnt = {0, 0};
    const uint64_t pixel = (uint64_t)P.probe_y * P.width + (uint64_t)P.probe_x;
    const F3 c = render_sample<LEVELS, DOF, true, false>(P, cx, (double)P.probe_x, (double)P.probe_y, 1, 1, pixel, 0, cnt, P.probe_out); /* MLC: any number of lights (n_cull = 0: no masks) */
    P.probe_out->color[0] = c.r;
    P.probe_out->color[1] = c.g;
    P.probe_out->color[2] = c.b;
}

#if C2RT_UNIT == 5
/* Rank-major strip buffers, each `rows_pad` rows (what a gather of equal-sized
 * per-rank buffers leaves on rank 0) -> full frame.  blockIdx.y
 * = frame row; float4 copies when a row is a multiple of 16 B. */
__global__ void deinterleave_kernel(const float *__restrict__ gathered, float *__restrict__ frame,
                                    uint32_t row_floats, uint32_t strip_height, uint32_t world, uint32_t rows_pad)
{
    const uint32_t y = blockIdx.y;
    const uint32_t strip = y / strip_height;
    const uint32_t rank = strip % world;
    const uint32_t lr = (strip / world) * strip_height + y % strip_height;
    const float *src = gathered + ((size_t)rank * rows_pad + lr) * (size_t)row_floats;
    float *dst = frame + (size_t)y * row_floats;
    if ((row_floats & 3u) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_floats / 4; i += gridDim.x * blockDim.x) d4[i] = s4[i];
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_floats; i += gridDim.x * blockDim.x) dst[i] = src[i];
    }
}

/* Color.toRGB32 via convertTo8bit_sRGB_Cached — rt/color.d:154-162,209-214 */
__global__ void encode_rgb32_kernel(const float *__restrict__ frame, uint32_t *__restrict__ out, uint64_t n,
                                    const uint8_t *__restrict__ lut)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t ch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = frame[3 * i + c];
            ch[c] = !(x > 0) ? 0u : (x >= 1 ? 255u : (uint32_t)lut[(int)(x * 4096.0f)]);
        }
        out[i] = ch[2] | (ch[1] << 8) | (ch[0] << 16);
    }
}
#endif /* C2RT_UNIT == 5 */

} // namespace

/*
 * Instantiation is split over translation units so that the (slow) device
 * compiles run in parallel: the Makefile builds this file once per
 * C2RT_UNIT = 0..4 (the frame kernel for that many CSG nesting levels) and
 * once with C2RT_UNIT = 5 (probe, de-interleave, encode, dispatcher).
 */
#ifndef C2RT_UNIT
#error "compile with -DC2RT_UNIT=0..5 (see Makefile)"
#endif

#if C2RT_UNIT >= 0 && C2RT_UNIT <= C2RT_MAX_CSG_DEPTH

template <>
int launch_render_level<C2RT_UNIT>(const RenderParams &p, bool dof_or_stereo, void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
#if C2RT_XCD_SWIZZLE
    const uint32_t tiles_y_pad = (p.tiles_y + 7u) / 8u * 8u;
#else
    const uint32_t tiles_y_pad = p.tiles_y;
#endif
    /* retry mode: a fixed grid walks the overflow list (render_body) */
    const dim3 grid(p.retry_mode ? 2048u : p.blocks_x * tiles_y_pad), block(kBlockThreads);
    const size_t lds = (size_t)p.csg_cap * kCsgLdsPerEntry * kWavesPerBlock;
    const bool stereo = p.cam.stereo_separation != 0;
    const bool multi = p.n_lights > 1;
#define C2RT_LAUNCH(KERNEL, ...)                                                                        \
    do {                                                                                                \
        if (p.ray_counters) hipLaunchKernelGGL((KERNEL<__VA_ARGS__, true>), grid, block, lds, s, p);    \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__, false>), grid, block, lds, s, p);                  \
    } while (0)
#if C2RT_UNIT == 0
    if (p.planes_only && !multi) { /* (planes + several lights: the general instances below) */
        if (dof_or_stereo && stereo) C2RT_LAUNCH(render_kernel_planes, 2);
        else if (dof_or_stereo) C2RT_LAUNCH(render_kernel_planes, 1);
        else C2RT_LAUNCH(render_kernel_planes, 0);
        return (int)hipGetLastError();
    }
#endif
    if (dof_or_stereo) {
        /* stereo cameras are rare: one instance (any number of lights) */
        if (stereo) C2RT_LAUNCH(render_kernel_dof, C2RT_UNIT, true, 2);
        else if (multi) C2RT_LAUNCH(render_kernel_dof, C2RT_UNIT, true, 1);
        else C2RT_LAUNCH(render_kernel_dof, C2RT_UNIT, false, 1);
    } else if (multi) {
        C2RT_LAUNCH(render_kernel, C2RT_UNIT, 0, true);
    } else {
        C2RT_LAUNCH(render_kernel, C2RT_UNIT, 0, false);
    }
#undef C2RT_LAUNCH
    return (int)hipGetLastError();
}

#else /* C2RT_UNIT == 5 */

int launch_render(const RenderParams &p, const KernelVariant &v, void *stream)
{
    switch (v.csg_levels) {
    case 0: return launch_render_level<0>(p, v.dof_or_stereo, stream);
    case 1: return launch_render_level<1>(p, v.dof_or_stereo, stream);
    case 2: return launch_render_level<2>(p, v.dof_or_stereo, stream);
    case 3: return launch_render_level<3>(p, v.dof_or_stereo, stream);
    case 4: return launch_render_level<4>(p, v.dof_or_stereo, stream);
    default: return (int)hipErrorInvalidValue;
    }
}

/* the probe is not a hot path: one instance that handles every scene */
int launch_probe(const RenderParams &p, const KernelVariant &, void *stream)
{
    const size_t lds = (size_t)p.csg_cap * kCsgLdsPerEntry;
    hipLaunchKernelGGL((probe_kernel<C2RT_MAX_CSG_DEPTH, 2>), dim3(1), dim3(kWave), lds,
                       static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

int launch_deinterleave(const float *gathered, float *frame, uint32_t width, uint32_t height,
                        uint32_t strip_height, uint32_t world, uint32_t rows_pad, uint32_t words_per_pixel, void *stream)
{
    const uint32_t row_floats = width * words_per_pixel; /* 3: float RGB, 1: packed RGB32 */
    const uint32_t per_row = (row_floats & 3u) == 0 ? row_floats / 4 : row_floats;
    const dim3 block(256), grid((per_row + 255) / 256 > 16 ? 16 : (per_row + 255) / 256, height);
    hipLaunchKernelGGL(deinterleave_kernel, grid, block, 0, static_cast<hipStream_t>(stream), gathered, frame,
                       row_floats, strip_height, world, rows_pad);
    return (int)hipGetLastError();
}

int launch_encode_rgb32(const float *frame, uint32_t *out, uint64_t n_pixels, const uint8_t *lut_dev, void *stream)
{
    const uint64_t blocks = (n_pixels + 255) / 256;
    const dim3 block(256), grid((uint32_t)(blocks > 4096 ? 4096 : (blocks ? blocks : 1)));
    hipLaunchKernelGGL(encode_rgb32_kernel, grid, block, 0, static_cast<hipStream_t>(stream), frame, out, n_pixels,
                       lut_dev);
    return (int)hipGetLastError();
}

#endif

} // namespace c2rt
