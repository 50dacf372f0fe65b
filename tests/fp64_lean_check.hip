/*
 * fp64_lean_check.hip — device check of chess2rt_amd/csrc/fp64_lean.h (test infrastructure).
 *
 * For each routine: N operands per launch drawn on the device from a counter hash (uniform mantissas,
 * exponents uniform over the routine's window, and a share of structured mantissas: all zeros, all ones,
 * single bits, 0x5555.. / 0xAAAA.., values one ulp either side of powers of two), evaluated BOTH with the
 * lean sequence and with the compiler's own expansion of `/` and `sqrt` in the same kernel; any differing
 * bit pattern is counted and the first few operands are kept.  A slice of every launch is copied back and
 * compared with the HOST's IEEE-754 `/` and `sqrt` (x86-64 SSE2: correctly rounded).
 *
 *   fp64_lean_check [log2(operands per routine)]     default 30 (1.07e9 per routine, < 1 s on an MI355X)
 * exit code 0 = no mismatch.  Prints one JSON line.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../chess2rt_amd/csrc/fp64_lean.h"

using namespace c2rt;

#define CHECK(x)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; }   \
    } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/* a double with sign s, exponent uniform in [lo, hi) and a mantissa that is random 3 times out of 4 and
 * structured otherwise */
__host__ __device__ inline double draw(uint64_t key, int lo, int hi, bool positive)
{
    const uint64_t a = mix64(key), b = mix64(key ^ 0x1234567890abcdefull);
    const int e = lo + (int)((a >> 8) % (uint64_t)(hi - lo));
    uint64_t m = b & 0xFFFFFFFFFFFFFull;
    const unsigned kind = (unsigned)(a & 0xFF);
    if (kind < 64) {
        const unsigned bit = (unsigned)((a >> 40) % 52);
        switch (kind & 7) {
        case 0: m = 0; break;
        case 1: m = 0xFFFFFFFFFFFFFull; break;
        case 2: m = 1ull << bit; break;
        case 3: m = 0xFFFFFFFFFFFFFull ^ (1ull << bit); break;
        case 4: m = 0x5555555555555ull; break;
        case 5: m = 0xAAAAAAAAAAAAAull; break;
        case 6: m = (b & 0xFFull); break;                         /* just above a power of two */
        default: m = 0xFFFFFFFFFFFFFull - (b & 0xFFull); break;  /* just below the next one */
        }
    }
    const uint64_t sign = positive ? 0 : ((a >> 63) << 63);
    const uint64_t bits = sign | ((uint64_t)(e + 1023) << 52) | m;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

struct Report {
    unsigned long long mismatches[12];
    double first[12][4]; /* per routine: operands / results of the first mismatch seen */
};

__device__ inline uint64_t bits_of(double d) { return (uint64_t)__double_as_longlong(d); }

__device__ inline void note(Report *r, int routine, double a, double b, double got, double want)
{
    if (atomicAdd(&r->mismatches[routine], 1ull) == 0) {
        r->first[routine][0] = a;
        r->first[routine][1] = b;
        r->first[routine][2] = got;
        r->first[routine][3] = want;
    }
}

/* the compiler's expansions, kept out of line so that nothing folds them with the lean forms */
__device__ __noinline__ double ref_div(double a, double b) { return a / b; }
__device__ __noinline__ double ref_sqrt(double x) { return sqrt(x); }
__device__ __noinline__ float ref_div_f32(float a, float b) { return a / b; }

/* a float with exponent uniform in [lo, hi) and the same mix of random / structured significands */
__host__ __device__ inline float draw_f32(uint64_t key, int lo, int hi, bool positive)
{
    const uint64_t a = mix64(key ^ 0xF32F32F32ull), b = mix64(key + 0x51ull);
    const int e = lo + (int)((a >> 8) % (uint64_t)(hi - lo));
    uint32_t m = (uint32_t)b & 0x7FFFFFu;
    const unsigned kind = (unsigned)(a & 0xFF);
    if (kind < 64) {
        const unsigned bit = (unsigned)((a >> 40) % 23);
        switch (kind & 7) {
        case 0: m = 0; break;
        case 1: m = 0x7FFFFFu; break;
        case 2: m = 1u << bit; break;
        case 3: m = 0x7FFFFFu ^ (1u << bit); break;
        case 4: m = 0x555555u; break;
        case 5: m = 0x2AAAAAu; break;
        case 6: m = (uint32_t)b & 0xFu; break;
        default: m = 0x7FFFFFu - ((uint32_t)b & 0xFu); break;
        }
    }
    const uint32_t bits = (positive ? 0u : (uint32_t)((a >> 63) << 31)) | ((uint32_t)(e + 127) << 23) | m;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* sample[]: the first `n_sample` items' operands and lean results, for the host comparison */
__global__ void check_kernel(Report *rep, uint64_t seed, uint64_t per_thread, double *sample, uint64_t n_sample)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = 0; i < per_thread; ++i) {
        const uint64_t idx = i * nthreads + tid;
        const uint64_t key = seed + idx * 4;
        /* 0: division, operands anywhere in the windows */
        {
            const double a = draw(key, -900, 700, false), b = draw(key + 1, -120, 120, false);
            double got = __longlong_as_double(0x7ff8dead00000000ll);
            if (den_ok(b) && num_ok(a)) got = div_with(a, b, rcp_refined(b));
            const double want = ref_div(a, b);
            if (bits_of(got) != bits_of(want)) note(rep, 0, a, b, got, want);
            if (idx < n_sample) { sample[idx * 12 + 0] = a; sample[idx * 12 + 1] = b; sample[idx * 12 + 2] = got; }
        }
        /* 1: division as the tracer sees it: |b| in [1e-9, 2), numerators of scene scale */
        {
            const double a = draw(key + 2, -40, 40, false), b = draw(key + 3, -30, 1, false);
            double got = __longlong_as_double(0x7ff8dead00000000ll);
            if (den_ok(b) && num_ok(a)) got = div_with(a, b, rcp_refined(b));
            const double want = ref_div(a, b);
            if (bits_of(got) != bits_of(want)) note(rep, 1, a, b, got, want);
            if (idx < n_sample) { sample[idx * 12 + 3] = a; sample[idx * 12 + 4] = b; sample[idx * 12 + 5] = got; }
        }
        /* 9: division through the CORRECTLY ROUNDED reciprocal RN(1/b) (what the host computes for wave-uniform
         * denominators, and what unit_len / inv_len return) instead of the refined v_rcp_f64 */
        {
            const double a = draw(key + 3, -40, 40, false), b = draw(key + 2, -30, 30, false);
            const double y = ref_div(1.0, b);
            double got = __longlong_as_double(0x7ff8dead00000000ll);
            if (den_ok(b) && num_ok(a)) got = div_with(a, b, y);
            else got = ref_div(a, b);
            const double want = ref_div(a, b);
            if (bits_of(got) != bits_of(want)) note(rep, 9, a, b, got, want);
        }
        /* 11: fp32 — three numerators over one denominator (Color / float) */
        {
            const float f = draw_f32(key + 1, -60, 60, false);
            const float r = rcp_refined_f32(f);
            for (int c = 0; c < 3; ++c) {
                float a = draw_f32(key + 2 + c, -60, 60, false);
                if (((key >> 3) + c) % 11 == 0) a = 0.0f; /* a black channel */
                float got = __uint_as_float(0x7fc0deadu);
                if (f32_ok(f) && (f32_ok(a) || (__float_as_uint(a) == 0 && f > 0))) got = div_with_f32(a, f, r);
                else got = ref_div_f32(a, f);
                const float want = ref_div_f32(a, f);
                if (__float_as_uint(got) != __float_as_uint(want)) note(rep, 11, a, f, got, want);
            }
        }
        /* 2: sqrt, 3: the reciprocal length */
        {
            const double x = draw(key + 1, -240, 240, true);
            double len = 0, inv = 0, s = 0;
            if (sqrt_ok(x)) { s = sqrt_lean(x); inv_len(x, len, inv); }
            const double want = ref_sqrt(x);
            if (bits_of(s) != bits_of(want) || bits_of(len) != bits_of(want)) note(rep, 2, x, 0, s, want);
            const double winv = ref_div(1.0, want);
            if ((bits_of(len) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) atomicAdd(&rep->mismatches[8], 1ull); /* significands of all ones met (informational) */
            if (bits_of(inv) != bits_of(winv)) note(rep, 3, x, want, inv, winv);
            if (idx < n_sample) { sample[idx * 12 + 6] = x; sample[idx * 12 + 7] = len; sample[idx * 12 + 8] = inv; }
        }
        /* 4: squared lengths of normalised vectors (what the tracer feeds inv_len / unit_len most of the time) */
        {
            const double vx = draw(key + 2, -20, 20, false), vy = draw(key + 3, -20, 20, false), vz = draw(key, -20, 20, false);
            const double q = vx * vx + vy * vy + vz * vz;
            const double l0 = ref_sqrt(q), i0 = ref_div(1.0, l0);
            const double nx = vx * i0, ny = vy * i0, nz = vz * i0;
            const double s = nx * nx + ny * ny + nz * nz;
            double len = 0, inv = 0;
            const bool near = near_one(s);
            if (near) unit_len(s, len, inv);
            else if (sqrt_ok(s)) inv_len(s, len, inv);
            const double wl = ref_sqrt(s), wi = ref_div(1.0, wl);
            if (bits_of(len) != bits_of(wl) || bits_of(inv) != bits_of(wi)) note(rep, 4, s, near ? 1.0 : 0.0, len, wl);
            if (!near) atomicAdd(&rep->mismatches[7], 1ull); /* how often a re-normalisation is NOT within 64 ulp of 1 (informational) */
            if (idx < n_sample) { sample[idx * 12 + 9] = s; sample[idx * 12 + 10] = len; sample[idx * 12 + 11] = inv; }
        }
    }
}

/* exhaustive: every s within 64 ulp of 1 (and the first values outside, which near_one must refuse),
 * window edges of in_window */
__global__ void edges_kernel(Report *rep)
{
    const int k = (int)threadIdx.x - 128; /* -128 .. 127 */
    const double s = __longlong_as_double(0x3FF0000000000000ll + (long long)k);
    const bool near = near_one(s);
    if (near != (k >= -64 && k <= 64)) note(rep, 5, s, (double)k, near ? 1.0 : 0.0, -1.0);
    if (near) {
        double len, inv;
        unit_len(s, len, inv);
        const double wl = ref_sqrt(s), wi = ref_div(1.0, wl);
        if (bits_of(len) != bits_of(wl) || bits_of(inv) != bits_of(wi)) note(rep, 5, s, (double)k, len, wl);
        /* dividing by that length through its reciprocal (Node.intersect's dist /= len): 2^20 numerators each */
        {
            for (uint32_t i = 0; i < (1u << 20); ++i) {
                const double a = draw(0xABCDull * (uint64_t)(k + 200) + i, -40, 40, false);
                const double got = div_with(a, len, inv), want = ref_div(a, len);
                if (bits_of(got) != bits_of(want)) note(rep, 10, a, len, got, want);
            }
        }
    }
    /* fp32, exhaustive in the denominator's significand: every float in [1, 2) and in [2^20, 2^21) under 16
     * numerators each (2 x 2^23 x 16 divisions, spread over the 256 threads of this block) */
    for (uint32_t m = threadIdx.x; m < (1u << 23); m += blockDim.x)
        for (int eb = 0; eb < 2; ++eb) {
            const float f = __uint_as_float(((uint32_t)(eb ? 147 : 127) << 23) | m);
            const float r = rcp_refined_f32(f);
            for (int i = 0; i < 16; ++i) {
                const float a = i == 0 ? 0.0f : draw_f32(0x1234ull * m + i, -20, 40, false);
                const float got = div_with_f32(a, f, r), want = ref_div_f32(a, f);
                if (__float_as_uint(got) != __float_as_uint(want)) note(rep, 11, a, f, got, want);
            }
        }
    if (threadIdx.x == 0) {
        if (f32_ok(0.0f) || f32_ok(-0.0f) || f32_ok(1e-30f) || f32_ok(3e38f) || f32_ok(__uint_as_float(0x7f800000u)) ||
            f32_ok(__uint_as_float(0x7fc00000u)) || !f32_ok(1.0f) || !f32_ok(-800000.0f) || !f32_ok(ldexpf(1.0f, -60)) || f32_ok(ldexpf(1.0f, 60)))
            note(rep, 6, 0, 0, 3.0, 0.0);
        const double specials[] = {0.0, -0.0, 4.9e-324, 2.2250738585072009e-308, __longlong_as_double(0x7ff0000000000000ll),
                                   __longlong_as_double(0xfff0000000000000ll), __longlong_as_double(0x7ff8000000000000ll)};
        for (double v : specials)
            if (den_ok(v) || num_ok(v) || sqrt_ok(v)) note(rep, 6, v, 0, 1.0, 0.0);
        /* window edges: 2^lo is inside, the double below it is outside; 2^hi is outside, the double below inside */
        const double lo = ldexp(1.0, -120), hi = ldexp(1.0, 120);
        if (!den_ok(lo) || den_ok(__longlong_as_double(__double_as_longlong(lo) - 1)) || den_ok(hi) ||
            !den_ok(__longlong_as_double(__double_as_longlong(hi) - 1)) || !den_ok(-lo) || den_ok(-hi))
            note(rep, 6, lo, hi, 2.0, 0.0);
        if (sqrt_ok(-1.0) == false) { /* negative arguments pass the magnitude window: callers feed sums of squares only */ }
    }
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 30;
    const uint64_t total = 1ull << lg;
    const uint32_t blocks = 4096, threads = 256;
    const uint64_t nthreads = (uint64_t)blocks * threads;
    const uint64_t per_thread = (total + nthreads - 1) / nthreads;
    const uint64_t n_sample = 1ull << 20;
    Report *rep;
    double *sample;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    CHECK(hipMalloc(&sample, n_sample * 12 * sizeof(double)));
    CHECK(hipMemset(sample, 0, n_sample * 12 * sizeof(double)));
    /* launches of at most 2^30 operands each, so that no launch runs for long */
    const uint64_t chunk_per_thread = (1ull << 30) / nthreads;
    uint64_t done = 0, launch = 0;
    while (done < per_thread) {
        const uint64_t n = per_thread - done < chunk_per_thread ? per_thread - done : chunk_per_thread;
        hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(threads), 0, 0, rep, 0xC2C2ull + (launch << 40), n, sample, launch == 0 ? n_sample : 0);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        done += n;
        ++launch;
    }
    hipLaunchKernelGGL(edges_kernel, dim3(1), dim3(256), 0, 0, rep);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, rep, sizeof h, hipMemcpyDeviceToHost));
    std::vector<double> hs(n_sample * 12);
    CHECK(hipMemcpy(hs.data(), sample, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
    /* host IEEE comparison of the sampled slice */
    unsigned long long host_bad = 0;
    auto same = [](double a, double b) { return memcmp(&a, &b, 8) == 0; };
    for (uint64_t i = 0; i < n_sample; ++i) {
        const double *p = &hs[i * 12];
        volatile double q0 = p[0] / p[1], q1 = p[3] / p[4];
        volatile double l = std::sqrt(p[6]);
        volatile double li = 1.0 / l;
        volatile double ul = std::sqrt(p[9]);
        volatile double ui = 1.0 / ul;
        if (!same(q0, p[2]) || !same(q1, p[5]) || !same(l, p[7]) || !same(li, p[8]) || !same(ul, p[10]) || !same(ui, p[11])) ++host_bad;
    }
    const char *names[] = {"div_window", "div_tracer_range", "sqrt_lean", "inv_len", "renormalise", "unit_len_exhaustive", "window_edges", "", "", "div_by_rounded_reciprocal", "div_by_unit_len", "div_f32"};
    unsigned long long bad = host_bad;
    printf("{\"operands_per_routine\": %llu, \"host_sample\": %llu, \"host_mismatches\": %llu, \"not_near_one\": %llu",
           (unsigned long long)(per_thread * nthreads), (unsigned long long)n_sample, host_bad, h.mismatches[7]);
    printf(", \"inv_len_all_ones_significands\": %llu", h.mismatches[8]);
    for (int r = 0; r < 12; ++r) {
        if (!names[r][0]) continue;
        printf(", \"%s\": %llu", names[r], h.mismatches[r]);
        bad += h.mismatches[r];
    }
    printf("}\n");
    for (int r = 0; r < 12; ++r)
        if (names[r][0] && h.mismatches[r])
            fprintf(stderr, "%s: first mismatch: operands %a %a -> got %a, want %a\n", names[r], h.first[r][0], h.first[r][1], h.first[r][2], h.first[r][3]);
    return bad ? 1 : 0;
}
