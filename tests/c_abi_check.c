/* c_abi_check.c — test-side driver: the plain-C demo scene (examples/c_abi_demo.c)
 * rendered through libc2rt.so and through the oracle, compared float for float.
 * Only tests compile this file; it is the one place the demo meets the oracle. */
#define main c_abi_demo_main
#include "../examples/c_abi_demo.c"
#undef main
#include "../oracle/c2rt_oracle.h"

int main(int argc, char **argv)
{
    const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 160, H = argc > 2 ? (uint32_t)atoi(argv[2]) : 120;
    c2rt_ctx *ctx = NULL;
    CHECK(c2rt_init(-1, &ctx));
    c2rt_scene_desc sc;
    c2rt_camera_frame cam;
    demo_scene(&sc, &cam, W, H);
    CHECK(c2rt_upload_scene(ctx, &sc));
    c2rt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.width = W; opts.height = H; opts.taps = C2RT_TAPS_REF5; opts.count_rays = 1;
    const size_t n = (size_t)W * H * 3;
    float *gpu = (float *)malloc(n * sizeof(float)), *ref = (float *)malloc(n * sizeof(float));
    CHECK(c2rt_render_frame(ctx, &cam, &opts, gpu, NULL));
    c2rt_ray_stats g, o;
    CHECK(c2rt_get_ray_stats(ctx, &g));
    if (orc_render_frame(&sc, &cam, &opts, ref, 0, &o) != 0) return 2;
    size_t differ = 0, beyond = 0;
    double lit = 0;
    for (size_t i = 0; i < n; ++i) {
        differ += memcmp(&gpu[i], &ref[i], sizeof(float)) != 0;
        beyond += !(fabsf(gpu[i] - ref[i]) <= 1e-4f);
        lit += ref[i] > 0.2f;
    }
    c2rt_trace_result tg, to;
    const int px = (int)(W * 3 / 8), py = (int)(H / 2);
    CHECK(c2rt_render_pixel(ctx, &cam, &opts, px, py, &tg));
    orc_render_pixel(&sc, &cam, &opts, px, py, &to);
    printf("differ %zu beyond_tol %zu lit %.3f rays %llu/%llu vs %llu/%llu probe node %d/%d leaf %d/%d\n", differ, beyond, lit / (double)n,
           (unsigned long long)g.primary_rays, (unsigned long long)g.shadow_rays, (unsigned long long)o.primary_rays, (unsigned long long)o.shadow_rays,
           tg.closest_node, to.closest_node, tg.leaf_geom, to.leaf_geom);
    c2rt_destroy(ctx);
    return !(beyond == 0 && g.primary_rays == o.primary_rays && g.shadow_rays == o.shadow_rays && tg.closest_node == to.closest_node &&
             tg.leaf_geom == to.leaf_geom && lit / (double)n > 0.05);
}
