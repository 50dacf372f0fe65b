/* c_multi_check.c — test-side driver of the single-process multi-device context
 * (c2rt_init_multi, SURVEY.md 8(b) `device_count_or_0`): the plain-C demo scene
 * rendered by ONE device and by a context of N slots (all on HIP device 0 when the
 * box has one GPU; one slot per GPU otherwise), through the host-output float and
 * RGB32 entry points, pinned and pageable — the frames must be the same bits, the
 * ray counts must add up, and the single-device frame must match the oracle. */
#define main c_abi_demo_main
#include "../examples/c_abi_demo.c"
#undef main
#include "../oracle/c2rt_oracle.h"

int main(int argc, char **argv)
{
    const uint32_t W = argc > 1 ? (uint32_t)atoi(argv[1]) : 203, H = argc > 2 ? (uint32_t)atoi(argv[2]) : 149;
    const int slots = argc > 3 ? atoi(argv[3]) : 2;
    const int spread = argc > 4 ? atoi(argv[4]) : 0; /* 1: slot i on HIP device i (needs that many GPUs) */
    c2rt_ctx *ctx = NULL;
    CHECK(c2rt_init(0, &ctx));
    c2rt_ctx *one = ctx;
    c2rt_scene_desc sc;
    c2rt_camera_frame cam;
    demo_scene(&sc, &cam, W, H);
    CHECK(c2rt_upload_scene(one, &sc));
    c2rt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.width = W; opts.height = H; opts.taps = C2RT_TAPS_REF5; opts.count_rays = 1;
    const size_t npx = (size_t)W * H, n = npx * 3;
    float *a = (float *)malloc(n * sizeof(float)), *b = (float *)malloc(n * sizeof(float)), *ref = (float *)malloc(n * sizeof(float));
    uint32_t *a32 = (uint32_t *)malloc(npx * 4), *b32 = (uint32_t *)malloc(npx * 4);
    CHECK(c2rt_render_frame(one, &cam, &opts, a, NULL));
    c2rt_ray_stats s1, s2, so;
    CHECK(c2rt_get_ray_stats(one, &s1));
    CHECK(c2rt_render_frame_rgb32(one, &cam, &opts, a32, NULL));
    if (orc_render_frame(&sc, &cam, &opts, ref, 0, &so) != 0) return 2;
    size_t beyond = 0;
    for (size_t i = 0; i < n; ++i) beyond += !(fabsf(a[i] - ref[i]) <= 1e-4f);

    int ids[64];
    for (int i = 0; i < slots && i < 64; ++i) ids[i] = spread ? i : 0;
    c2rt_ctx *multi = NULL;
    ctx = NULL;
    {
        int st_ = c2rt_init_multi(slots, ids, &multi);
        ctx = multi;
        if (st_ != C2RT_OK) { fprintf(stderr, "c2rt_init_multi -> %d: %s\n", st_, multi ? c2rt_last_error(multi) : ""); return 1; }
    }
    if (c2rt_device_count(multi) != slots) return 3;
    CHECK(c2rt_upload_scene(multi, &sc));
    memset(b, 0xff, n * sizeof(float));
    CHECK(c2rt_render_frame(multi, &cam, &opts, b, NULL)); /* pageable */
    CHECK(c2rt_get_ray_stats(multi, &s2));
    size_t differ = 0;
    for (size_t i = 0; i < n; ++i) differ += memcmp(&a[i], &b[i], sizeof(float)) != 0;
    memset(b, 0xff, n * sizeof(float));
    CHECK(c2rt_pin_host_buffer(multi, b, n * sizeof(float)));
    CHECK(c2rt_render_frame(multi, &cam, &opts, b, NULL)); /* pinned: the slots' copies run in parallel */
    CHECK(c2rt_unpin_host_buffer(multi, b));
    for (size_t i = 0; i < n; ++i) differ += memcmp(&a[i], &b[i], sizeof(float)) != 0;
    memset(b32, 0xff, npx * 4);
    CHECK(c2rt_render_frame_rgb32(multi, &cam, &opts, b32, NULL));
    size_t differ32 = 0;
    for (size_t i = 0; i < npx; ++i) differ32 += a32[i] != b32[i];
    /* a strip-sharded request on a multi-device context is refused, not silently misrendered */
    c2rt_render_opts bad = opts;
    bad.strip_world = 2; bad.strip_height = 8;
    const int st_bad = c2rt_render_frame(multi, &cam, &bad, b, NULL);
    printf("slots %d: differ %zu differ32 %zu beyond_tol %zu rays %llu/%llu vs %llu/%llu vs oracle %llu/%llu sharded-request status %d\n", slots,
           differ, differ32, beyond, (unsigned long long)s1.primary_rays, (unsigned long long)s1.shadow_rays,
           (unsigned long long)s2.primary_rays, (unsigned long long)s2.shadow_rays, (unsigned long long)so.primary_rays,
           (unsigned long long)so.shadow_rays, st_bad);
    c2rt_destroy(multi);
    c2rt_destroy(one);
    return !(differ == 0 && differ32 == 0 && beyond == 0 && s1.primary_rays == s2.primary_rays && s1.shadow_rays == s2.shadow_rays &&
             s1.primary_rays == so.primary_rays && s1.shadow_rays == so.shadow_rays && st_bad == C2RT_ERR_INVALID_ARG);
}
