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
 *   - CSG hit lists live in LDS, one per-lane stack per wave shared by the
 *     nesting levels ([entry][lane]: bank = lane, conflict-free for any per-lane
 *     entry index); only (dist, tag) is kept per hit and the winning hit is
 *     re-derived: 16 entries = 10 KiB per wave at depth 1 and 2, 20 entries = 12.5 KiB at depth 3 and 4 on the
 *     first pass (kCsgFirstCap, c2rt_device.h; 16 x depth on the rare full-capacity retry pass);
 *   - geometry is fp64 and colour fp32 in the reference's operation order
 *     (built with -ffp-contract=off), because checker edges, shadow
 *     terminators and CSG boundaries flip on 1-ulp differences.
 *
 * This file holds the kernel entry points and launchers.  The trace itself is
 * c2rt_trace.inc, included twice below: lean:: (divide / sqrt / normalise through
 * the shortened correctly rounded sequences of fp64_lean.h, optimistically) and
 * exact:: (the compiler's IEEE expansions); render_one() runs a tile through
 * lean:: and again through exact:: when an operand left the lean windows.
 *
 * What each function restates is cited as file:line of /root/reference/source.
 */
#include <hip/hip_runtime.h>

#include "c2rt_device.h"
#include "fp64_lean.h"
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
/* Sphere.intersect's u,v (rt/geometry.d:118-120) out of line as well: the x87 emulation (x87.h) is ~500 integer
 * instructions with ~40 live registers, reached by textured sphere hits only; inlined (twice: lean:: and exact::)
 * it was where the headline instance spilled. */
struct UV { double u, v; };
__device__ __noinline__ UV c2_sphere_uv(double dx, double dz, double w)
{
    constexpr double PI = 3.14159265358979323846;
    const double angle = atan2(dz, dx);
    const double as = asin(w);
    UV r;
    r.u = fabs(angle) <= 4.0 ? x87_sphere_u(angle) : (PI + angle) / (2 * PI);
    r.v = fabs(as) <= 2.0 ? x87_sphere_v(as) : 1.0 - (PI / 2 + as) / PI;
    return r;
}
/* Register budget per kernel instance, as waves per SIMD (512 VGPRs per lane and SIMD: 128 at 4 waves,
 * 168 at 3, 256 at 2).  With no hint hipcc takes all 512 registers and runs one wave per SIMD (1.8x
 * slower).  Chosen per instance from the compiler's resource remarks (`make resource-usage`; profiles/r04_resource_usage.md)
 * so that NO instance spills VGPRs to scratch, except where a measurement says otherwise:
 *   depth 0 (no CSG), planes-only: 4 waves (111-127 VGPRs);
 *   depth 1, at most one light: 4 waves — 128 VGPRs since the cube / sphere face tables moved to the upload
 *     and the hit's lighting terms are evaluated before the shadow test (was 149 at 3 waves);
 *   depth 1, several lights (the hit stays live across the light loop) and depth-1 DOF: 3 waves (144-162);
 *   depth 2 / 3 / 4: THREE waves (168 VGPRs) although they then spill (depth 4 multi-light: 136 VGPRs, 240 B of
 *     scratch per lane; at two waves it needs 230 and spills none): a wave of these instances issues one
 *     instruction at a time and a third of its instructions are scalar, so with two waves per SIMD the VALU idles
 *     half the time (VALU busy 0.53) — the third wave is worth more than the spills cost: csg_stress.sdl cut to
 *     depth 2 / 3 / 4: 2.33 -> 1.88, 4.60 -> 3.97, 10.21 -> 8.53 ms (scripts/depth_occupancy.sh; four waves:
 *     2.59 / 5.48 / 10.5).  The hit stacks have to fit three workgroups per CU too: kCsgFirstCap, c2rt_device.h. */
#ifndef C2RT_OCC_U1
#define C2RT_OCC_U1 4
#endif
#ifndef C2RT_OCC_DEEP
#define C2RT_OCC_DEEP 3
#endif
#ifndef C2RT_OCC_U2
#define C2RT_OCC_U2 3
#endif
#ifndef C2RT_OCC_U3
#define C2RT_OCC_U3 C2RT_OCC_DEEP
#endif
template <int LEVELS, int DOF, bool MLC>
constexpr int occ_of()
{
#ifndef C2RT_OCC_U0
#define C2RT_OCC_U0 4
#endif
    return LEVELS == 0 ? (DOF ? 4 : C2RT_OCC_U0) : (LEVELS == 1 ? ((DOF || MLC) ? 3 : C2RT_OCC_U1) : (LEVELS == 2 ? C2RT_OCC_U2 : (LEVELS == 3 ? C2RT_OCC_U3 : C2RT_OCC_DEEP)));
}
#define C2RT_OCC_OF(L, D, M) __attribute__((amdgpu_waves_per_eu(occ_of<L, D, M>(), occ_of<L, D, M>())))
#ifndef C2RT_TILE_STATS
#define C2RT_TILE_STATS 0 /* diagnostics: per-tile wave cycles + class (RenderParams::tile_stats) */
#endif
#ifndef C2RT_XCD_SWIZZLE
#define C2RT_XCD_SWIZZLE 1
#endif

namespace lean {
constexpr bool kLean = true;
#include "c2rt_trace.inc"
} // namespace lean
namespace exact {
constexpr bool kLean = false;
#include "c2rt_trace.inc"
} // namespace exact

#ifndef C2RT_LEAN
#define C2RT_LEAN 1 /* 0: the production instances run exact:: only (A/B builds) */
#endif
/* The deepest CSG nesting whose instances carry the lean:: copy.  Depth 4 does not: measured with and without it
 * at two waves per SIMD (10.39 / 10.42 ms on csg_stress) and at three (8.60 / 8.56) — no difference, twice the
 * code. */
#ifndef C2RT_LEAN_MAX_LEVELS
#define C2RT_LEAN_MAX_LEVELS 3
#endif
typedef const RenderParams __attribute__((address_space(4))) *KArgs;

/* One tile: optimistically through lean::, and again through exact:: — by the same wave, with all of its
 * lanes — if a lane reported an operand outside a lean window (c2rt_trace.inc).  The instances launched when
 * rays are being counted (CNT; tests/conftest.py renders every counted frame with BOTH instances and insists
 * on the same bits) run exact:: only, so that suite compares the two. */
/* 1: in the instances named below the cold half reads its arguments through a kernarg pointer the optimiser
 * cannot see through, so that nothing but that pointer and the tile index stays live across the lean half on its
 * behalf.  Without it values both halves use (kernel arguments, table pointers) are kept in SGPRs from the top
 * of the kernel and the lean half spills around them: render_kernel_idn<1> holds 254 v_writelane / v_readlane in
 * its lean half without, 191 with (lean:: compiled alone: 184); lecture5.sdl 4K at 1 sample per pixel 0.245 ->
 * 0.239 ms, 1080p 72 -> 71 us, 4K x5 0.999 -> 0.996 ms.  Depth-of-field, plane-only and nested-CSG instances
 * are allocated no better or worse with it (profiles/r04_variants.md, step 12) and keep the plain call. */
#ifndef C2RT_REDO_OPAQUE
#define C2RT_REDO_OPAQUE 1
#endif

template <int LEVELS, int DOF, bool MLC, int PO, bool CNT>
DEV void render_one(const RenderParams &P, KArgs K, const uint32_t b)
{
    if constexpr (CNT || !C2RT_LEAN || LEVELS > C2RT_LEAN_MAX_LEVELS) {
        exact::render_tile<LEVELS, DOF, MLC, PO, CNT>(P, (exact::KArgs)K, b);
    } else {
        if (!P.force_exact) { /* wave-uniform */
            const bool redo = lean::render_tile<LEVELS, DOF, MLC, PO, false>(P, (lean::KArgs)K, b);
            if (!__ballot(redo)) return;
            if (threadIdx.x % kWave == 0) atomicAdd(P.redo_counter, 1ull); /* c2rt_get_exact_redos */
        }
        if constexpr (C2RT_REDO_OPAQUE && LEVELS <= 1 && !DOF && PO != lean::kSpecPlanes) {
            KArgs K2 = K;
            uint32_t b2 = b;
            asm volatile("" : "+s"(K2), "+s"(b2));
            exact::render_tile<LEVELS, DOF, MLC, PO, false>(*(const RenderParams *)K2, (exact::KArgs)K2, b2);
        } else {
            exact::render_tile<LEVELS, DOF, MLC, PO, false>(P, (exact::KArgs)K, b);
        }
    }
}

/* One tile per workgroup; in retry mode (RenderParams::retry_mode: the full-capacity relaunch of
 * the nested-CSG instances) a fixed grid walks the list of tiles whose hit stacks overflowed.
 * Either way the tile code is inlined once. */
template <int LEVELS, int DOF, bool MLC, int PO, bool CNT>
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
            render_one<LEVELS, DOF, MLC, PO, CNT>(P, K, b);
            i += gridDim.x;
        } while (P.retry_mode);
    } else {
        render_one<LEVELS, DOF, MLC, PO, CNT>(P, K, blockIdx.x);
    }
}

template <int LEVELS, int DOF, bool MLC, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(LEVELS, DOF, MLC) render_kernel(const RenderParams P)
{
    render_body<LEVELS, DOF, MLC, 0, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

/* The same for scenes in which every node's matrix is the identity (RenderParams::all_identity: every scene the
 * reference ships) with at most one light and no depth of field: lean::kSpecIdentity (c2rt_trace.inc) — production
 * instances only; counted frames run the general instance, and the tests compare the two. */
template <int LEVELS>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(LEVELS, 0, false) render_kernel_idn(const RenderParams P)
{
    render_body<LEVELS, 0, false, lean::kSpecIdentity, false>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

/* The depth-of-field / stereo instance carries the lens sampling state on top of
 * the tracer's and has its own register budget (C2RT_OCC_DOF). */
template <int LEVELS, bool MLC, int MODE, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(LEVELS, MODE, MLC) render_kernel_dof(const RenderParams P)
{
    render_body<LEVELS, MODE, MLC, 0, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

/* Scenes made of axis planes only (RenderParams::planes_only — lecture4.sdl, zaphod.sdl): the
 * instances in which a plane's miss is decided before the ray is normalised (plane_points_away). */
template <int DOF, bool CNT>
__global__ void __launch_bounds__(kBlockThreads) C2RT_OCC_OF(0, DOF, false) render_kernel_planes(const RenderParams P)
{
    render_body<0, DOF, false, lean::kSpecPlanes, CNT>(P, (KArgs)__builtin_amdgcn_kernarg_segment_ptr()); /* at most one light (launch_render_level); planes have no boxes, hence no culling masks */
}
/* (no identity-matrix variant of these: single-plane scenes take the straight-line ground trace, which has no matrix
 * code to lose — measured: zaphod x4 0.601 vs 0.606 ms, DOF 4.583 vs 4.582) */

/* renderPixel — rt/renderer.d:46-57: one lane, one sample, full trace result */
template <int LEVELS, int DOF>
__global__ void __launch_bounds__(kWave) probe_kernel(const RenderParams P)
{
    using namespace exact;
    extern __shared__ __align__(16) char lds[];
    if (threadIdx.x != 0) return;
    Ctx cx;
    cx.geoms = (GeomP)P.geoms;
    cx.nodes = (NodeP)P.nodes;
    cx.n_nodes = P.n_nodes;
    cx.kargs = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    cx.lds = lds;
    cx.lane = 0;
    cx.csg_cap = (int)P.csg_cap;
    cx.overflow = false;
    oob_init(cx.bad);
    cx.trunc_counter = nullptr;
#if C2RT_TILE_STATS
    cx.lane_stats = nullptr;
#endif
    cx.block = 0;
    cx.mask_slot = 0;
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

/* The culling masks of every tile of a frame's local rows, one lane per tile (c2rt_trace.inc: tile_mask_entry,
 * tile_mask_slot).  Runs once per frame whose camera leaves culling rectangles, in front of the frame kernel's
 * launch(es), on the same stream. */
__global__ void __launch_bounds__(256) tile_masks_kernel(const RenderParams P, uint32_t *__restrict__ table, uint32_t tile_rows)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cols = P.blocks_x * kWavesPerBlock;
    const uint32_t trow = i / cols, tcol = i % cols;
    if (trow >= tile_rows) return;
    uint32_t m[8];
    exact::tile_mask_entry(P, (exact::KArgs)__builtin_amdgcn_kernarg_segment_ptr(), trow, tcol, m);
    typedef uint32_t __attribute__((ext_vector_type(4))) u4_t;
    u4_t v, w;
    v.x = m[0]; v.y = m[1]; v.z = m[2]; v.w = m[3];
    w.x = m[4]; w.y = m[5]; w.z = m[6]; w.w = m[7];
    const size_t slot = exact::tile_mask_slot(P, trow, tcol);
    reinterpret_cast<u4_t *>(table)[slot] = v;
    if (P.n_cull_lights > 1u) reinterpret_cast<u4_t *>(table)[(size_t)P.mask_entries + slot] = w; /* lights 1..3 */
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
    /* identity matrices throughout and not a counted frame: the instances specialised for that */
    const bool idn = p.all_identity && !p.ray_counters;
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
    } else if (idn) {
        hipLaunchKernelGGL((render_kernel_idn<C2RT_UNIT>), grid, block, lds, s, p);
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

size_t tile_mask_entries(const RenderParams &p)
{
    const uint32_t tile_rows = (p.mask_rows + kTileH - 1) / kTileH;
    return (size_t)((tile_rows + 7u) / 8u * 8u) * p.blocks_x * kWavesPerBlock;
}

int launch_tile_masks(const RenderParams &p, uint32_t *table, void *stream)
{
    const uint32_t tile_rows = (p.mask_rows + kTileH - 1) / kTileH;
    const uint32_t lanes = tile_rows * p.blocks_x * kWavesPerBlock;
    if (!lanes) return 0;
    hipLaunchKernelGGL(tile_masks_kernel, dim3((lanes + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), p, table, tile_rows);
    return (int)hipGetLastError();
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
