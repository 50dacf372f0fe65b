// Issue cost (cycles per wave-instruction) of the fp64 / helper VALU instructions the render kernel is made of,
// on one SIMD with 1, 2 and 4 resident waves.  Build: hipcc --offload-arch=gfx950 -O2 fp64_issue.hip -o fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                            \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, int iters, double seed)      \
    {                                                                                                 \
        double a = seed + threadIdx.x, b = seed * 3 + 1, c = 1.5, d0, d1, d2, d3;                     \
        d0 = d1 = d2 = d3 = a;                                                                        \
        float fa = (float)a, fb = 2.0f, f0 = 1.0f, f1 = 1.0f;                                          \
        int i0 = threadIdx.x, i1 = 3, j0 = 0, j1 = 0, j2 = 0, j3 = 0;                                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int i = 0; i < iters; ++i) {                                                             \
            REP8(body)                                                                                \
        }                                                                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        if (d0 + d1 + d2 + d3 + f0 + f1 + i0 + i1 + j0 + j1 + j2 + j3 == 12345.678) out[1000] = 1;                         \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                \
    }

// four independent destinations per body -> 32 instructions per loop trip
#define OP3(ins) asm volatile(ins " %0, %4, %5, %6\n" ins " %1, %4, %5, %6\n" ins " %2, %4, %5, %6\n" ins " %3, %4, %5, %6" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(c));
#define OP2(ins) asm volatile(ins " %0, %4, %5\n" ins " %1, %4, %5\n" ins " %2, %4, %5\n" ins " %3, %4, %5" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
#define OP1(ins) asm volatile(ins " %0, %4\n" ins " %1, %4\n" ins " %2, %4\n" ins " %3, %4" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));
#define OPSCALE asm volatile("v_div_scale_f64 %0, vcc, %4, %5, %4\nv_div_scale_f64 %1, vcc, %4, %5, %4\nv_div_scale_f64 %2, vcc, %4, %5, %4\nv_div_scale_f64 %3, vcc, %4, %5, %4" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");
#define OPCMP asm volatile("v_cmp_lt_f64 vcc, %4, %5\nv_cmp_lt_f64 vcc, %5, %4\nv_cmp_gt_f64 vcc, %4, %5\nv_cmp_gt_f64 vcc, %5, %4" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");
#define OPCMP32 asm volatile("v_cmp_lt_u32 vcc, %4, %5\nv_cmp_lt_u32 vcc, %5, %4\nv_cmp_gt_u32 vcc, %4, %5\nv_cmp_gt_u32 vcc, %5, %4" \
                              : "+v"(i0) : "v"(i0), "v"(i1), "v"(i0), "v"(i1), "v"(i0) : "vcc");
#define OPCLASS asm volatile("v_cmp_class_f64 vcc, %4, %5\nv_cmp_class_f64 vcc, %4, %5\nv_cmp_class_f64 vcc, %4, %5\nv_cmp_class_f64 vcc, %4, %5" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(i1) : "vcc");
#define OPF2(ins) asm volatile(ins " %0, %2, %3\n" ins " %1, %2, %3\n" ins " %0, %2, %3\n" ins " %1, %2, %3" : "+v"(f0), "+v"(f1) : "v"(fa), "v"(fb));
#define OPI2(ins) asm volatile(ins " %0, %2, %3\n" ins " %1, %2, %3\n" ins " %0, %2, %3\n" ins " %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i0), "v"(i1));
#define OPLDEXP asm volatile("v_ldexp_f64 %0, %4, %5\nv_ldexp_f64 %1, %4, %5\nv_ldexp_f64 %2, %4, %5\nv_ldexp_f64 %3, %4, %5" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(i1));
#define OPCND asm volatile("v_cndmask_b32 %0, %2, %3, vcc\nv_cndmask_b32 %1, %2, %3, vcc\nv_cndmask_b32 %0, %2, %3, vcc\nv_cndmask_b32 %1, %2, %3, vcc" : "+v"(i0), "+v"(i1) : "v"(i0), "v"(i1) : "vcc");
#define OPCND4 asm volatile("v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3) : "v"(i0), "v"(i1) : "vcc");
#define OPCNDS asm volatile("v_cndmask_b32 %0, %4, %5, s[20:21]\nv_cndmask_b32 %1, %4, %5, s[20:21]\nv_cndmask_b32 %2, %4, %5, s[22:23]\nv_cndmask_b32 %3, %4, %5, s[22:23]" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3) : "v"(i0), "v"(i1) : "s20", "s21", "s22", "s23");
#define OPCMPCND asm volatile("v_cmp_lt_f64 vcc, %4, %5\nv_cndmask_b32 %0, %6, %7, vcc\nv_cmp_gt_f64 vcc, %4, %5\nv_cndmask_b32 %1, %6, %7, vcc" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3) : "v"(a), "v"(b), "v"(i0), "v"(i1) : "vcc");
#define OPWRL asm volatile("v_writelane_b32 %0, s20, 3\nv_writelane_b32 %1, s21, 4\nv_writelane_b32 %2, s22, 5\nv_writelane_b32 %3, s23, 6" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3) : : "s20", "s21", "s22", "s23");
#define OPFMAS asm volatile("v_fma_f64 %0, %4, s[20:21], %6\nv_fma_f64 %1, %4, s[20:21], %6\nv_fma_f64 %2, %4, s[22:23], %6\nv_fma_f64 %3, %4, s[22:23], %6" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(c) : "s20", "s21", "s22", "s23");
#define OPSAVEEXEC asm volatile("s_and_saveexec_b64 s[20:21], vcc\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[20:21], vcc\ns_mov_b64 exec, s[20:21]" : : : "s20", "s21", "scc");
#define OPMOV64 asm volatile("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));
#define OPRDL asm volatile("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %0, 4\nv_readlane_b32 s22, %0, 5\nv_readlane_b32 s23, %0, 6" : : "v"(i0) : "s20", "s21", "s22", "s23");
#define OPSNOP asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0");
#define OPSALU asm volatile("s_and_b64 s[20:21], s[20:21], exec\ns_or_b64 s[22:23], s[22:23], exec\ns_and_b64 s[20:21], s[20:21], exec\ns_or_b64 s[22:23], s[22:23], exec" : : : "s20", "s21", "s22", "s23", "scc");
// a dependent chain: the latency of back-to-back dependent fp64 FMAs
#define OPDEP asm volatile("v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(b), "v"(c));
#define OPDEPRCP asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %0, %0\nv_rcp_f64 %0, %0\nv_rcp_f64 %0, %0" : "+v"(d0));

KERNEL(k_fma, OP3("v_fma_f64"))
KERNEL(k_mul, OP2("v_mul_f64"))
KERNEL(k_add, OP2("v_add_f64"))
KERNEL(k_rcp, OP1("v_rcp_f64"))
KERNEL(k_rsq, OP1("v_rsq_f64"))
KERNEL(k_sqrt, OP1("v_sqrt_f64"))
KERNEL(k_scale, OPSCALE)
KERNEL(k_fmas, OP3("v_div_fmas_f64"))
KERNEL(k_fixup, OP3("v_div_fixup_f64"))
KERNEL(k_cmp, OPCMP)
KERNEL(k_cmp32, OPCMP32)
KERNEL(k_class, OPCLASS)
KERNEL(k_ldexp, OPLDEXP)
KERNEL(k_cnd, OPCND)
KERNEL(k_mov64, OPMOV64)
KERNEL(k_cnd4, OPCND4)
KERNEL(k_cnds, OPCNDS)
KERNEL(k_cmpcnd, OPCMPCND)
KERNEL(k_wrl, OPWRL)
KERNEL(k_fmas_s, OPFMAS)
KERNEL(k_saveexec, OPSAVEEXEC)
KERNEL(k_mov32, OPI2("v_xor_b32"))
KERNEL(k_readlane, OPRDL)
KERNEL(k_snop, OPSNOP)
KERNEL(k_salu, OPSALU)
KERNEL(k_addf32, OPF2("v_add_f32"))
KERNEL(k_addu32, OPI2("v_add_u32"))
KERNEL(k_bfe, OPF2("v_mul_f32"))
KERNEL(k_dep_fma, OPDEP)
KERNEL(k_dep_rcp, OPDEPRCP)
KERNEL(k_max64, OP2("v_max_f64"))
KERNEL(k_frexp, OP1("v_frexp_mant_f64"))
KERNEL(k_floor, OP1("v_floor_f64"))
KERNEL(k_cvt, OP1("v_rndne_f64"))

typedef void (*kern_t)(unsigned long long *, int, double);
struct Entry { const char *name; kern_t k; };

int main()
{
    Entry es[] = {{"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq},
                  {"v_sqrt_f64", k_sqrt}, {"v_div_scale_f64", k_scale}, {"v_div_fmas_f64", k_fmas}, {"v_div_fixup_f64", k_fixup},
                  {"v_cmp_f64", k_cmp}, {"v_cmp_u32", k_cmp32}, {"v_cmp_class_f64", k_class}, {"v_ldexp_f64", k_ldexp},
                  {"v_cndmask_b32", k_cnd}, {"v_mov_b64", k_mov64}, {"v_cndmask vcc indep", k_cnd4}, {"v_cndmask sgpr indep", k_cnds}, {"v_cmp_f64+v_cndmask", k_cmpcnd}, {"v_writelane_b32", k_wrl}, {"v_fma_f64 sgpr src", k_fmas_s}, {"s_and_saveexec+mov", k_saveexec}, {"v_xor_b32", k_mov32}, {"v_readlane_b32", k_readlane}, {"s_nop 0", k_snop},
                  {"s_and/or_b64", k_salu}, {"v_add_f32", k_addf32}, {"v_add_u32", k_addu32}, {"v_mul_f32", k_bfe},
                  {"v_max_f64", k_max64}, {"v_frexp_mant_f64", k_frexp}, {"v_floor_f64", k_floor}, {"v_rndne_f64", k_cvt},
                  {"dep v_fma_f64 chain", k_dep_fma}, {"dep v_rcp_f64 chain", k_dep_rcp}};
    unsigned long long *out;
    hipMalloc(&out, 8192 * 8);
    const int iters = 100000;
    // warm the clocks
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_fma, dim3(256), dim3(1024), 0, 0, out, iters / 10, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-22s %9s %9s %9s   ns per wave-instruction with 1 / 2 / 4 waves on each SIMD of one CU (x2.4 = cycles at 2.4 GHz); last: ratio to v_add_f32 at 4 waves\n", "instruction", "1 wave", "2 waves", "4 waves");
    double ref = 0;
    for (int pass = 0; pass < 2; ++pass)
    for (auto &e : es) {
        double res[3];
        int wv[3] = {1, 2, 4};
        for (int w = 0; w < 3; ++w) {
            const int threads = 64 * 4 * wv[w];
            // every CU busy (256 blocks) so that the chip sits at its loaded clock
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters / 20, 1.0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: wv waves x iters x 32 instructions in ms
            res[w] = (double)ms * 1e6 / ((double)iters * 32.0 * wv[w]);
        }
        if (pass == 0) { if (!strcmp(e.name, "v_add_f32")) ref = res[2]; continue; }
        printf("%-22s %9.3f %9.3f %9.3f   cycles@2.4: %6.2f %6.2f %6.2f   x%.2f\n", e.name, res[0], res[1], res[2], res[0] * 2.4, res[1] * 2.4, res[2] * 2.4, res[2] / ref);
    }
    return 0;
}
