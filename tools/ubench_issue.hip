// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the path kernels lean
// on, measured with s_memtime on one wave per SIMD (and with 4 waves per SIMD for throughput).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 8

#define BENCH_KERNEL(NAME, DECL, BODY, SINK)                                              \
    __global__ void NAME(unsigned long long* out, double seed, int ITER) {                            \
        DECL;                                                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
        for (int i = 0; i < ITER; ++i) { BODY }                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
        SINK;                                                                               \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                    \
    }

// 8 independent chains each
#define D8 double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double b=seed*0.5+0.25
#define SINKD if (a0+a1+a2+a3+a4+a5+a6+a7 == 12345.678) out[1]=1
#define OP8(ASM) asm volatile(ASM " %0, %0, %8, %0\n" ASM " %1, %1, %8, %1\n" ASM " %2, %2, %8, %2\n" ASM " %3, %3, %8, %3\n" \
                              ASM " %4, %4, %8, %4\n" ASM " %5, %5, %8, %5\n" ASM " %6, %6, %8, %6\n" ASM " %7, %7, %8, %7\n" \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));
#define OP8_2(ASM) asm volatile(ASM " %0, %0, %8\n" ASM " %1, %1, %8\n" ASM " %2, %2, %8\n" ASM " %3, %3, %8\n" \
                              ASM " %4, %4, %8\n" ASM " %5, %5, %8\n" ASM " %6, %6, %8\n" ASM " %7, %7, %8\n" \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));
#define OP8_1(ASM) asm volatile(ASM " %0, %0\n" ASM " %1, %1\n" ASM " %2, %2\n" ASM " %3, %3\n" \
                              ASM " %4, %4\n" ASM " %5, %5\n" ASM " %6, %6\n" ASM " %7, %7\n" \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7));

BENCH_KERNEL(k_fma_f64, D8, OP8("v_fma_f64"), SINKD)
BENCH_KERNEL(k_mul_f64, D8, OP8_2("v_mul_f64"), SINKD)
BENCH_KERNEL(k_add_f64, D8, OP8_2("v_add_f64"), SINKD)
BENCH_KERNEL(k_rcp_f64, D8, OP8_1("v_rcp_f64"), SINKD)
BENCH_KERNEL(k_rsq_f64, D8, OP8_1("v_rsq_f64"), SINKD)
BENCH_KERNEL(k_sqrt_f64, D8, OP8_1("v_sqrt_f64"), SINKD)
BENCH_KERNEL(k_rndne_f64, D8, OP8_1("v_rndne_f64"), SINKD)
BENCH_KERNEL(k_fract_f64, D8, OP8_1("v_fract_f64"), SINKD)
BENCH_KERNEL(k_frexpm_f64, D8, OP8_1("v_frexp_mant_f64"), SINKD)

// ldexp: dst f64, src f64, src i32
__global__ void k_ldexp_f64(unsigned long long* out, double seed, int ITER) {
    D8; int e = (int)seed & 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_ldexp_f64 %0, %0, %8\nv_ldexp_f64 %1, %1, %8\nv_ldexp_f64 %2, %2, %8\nv_ldexp_f64 %3, %3, %8\n"
                     "v_ldexp_f64 %4, %4, %8\nv_ldexp_f64 %5, %5, %8\nv_ldexp_f64 %6, %6, %8\nv_ldexp_f64 %7, %7, %8\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(e));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    SINKD;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// 32-bit integer ops
#define U8 uint32_t a0=(uint32_t)seed,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7; uint32_t b=a0*2654435761u+12345u
#define SINKU if (a0+a1+a2+a3+a4+a5+a6+a7 == 123456789u) out[1]=1
BENCH_KERNEL(k_mul_lo_u32, U8, OP8_2("v_mul_lo_u32"), SINKU)
BENCH_KERNEL(k_mul_hi_u32, U8, OP8_2("v_mul_hi_u32"), SINKU)
BENCH_KERNEL(k_xor_b32, U8, OP8_2("v_xor_b32"), SINKU)
#define OP8B(ASM) asm volatile(ASM " %0, %0, %8, %0 bitop3:0x96\n" ASM " %1, %1, %8, %1 bitop3:0x96\n" ASM " %2, %2, %8, %2 bitop3:0x96\n" ASM " %3, %3, %8, %3 bitop3:0x96\n" \
                              ASM " %4, %4, %8, %4 bitop3:0x96\n" ASM " %5, %5, %8, %5 bitop3:0x96\n" ASM " %6, %6, %8, %6 bitop3:0x96\n" ASM " %7, %7, %8, %7 bitop3:0x96\n" \
    : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b));
BENCH_KERNEL(k_xor3_b32, U8, OP8B("v_bitop3_b32"), SINKU)
BENCH_KERNEL(k_add_u32, U8, OP8_2("v_add_u32"), SINKU)
BENCH_KERNEL(k_mul_u32_u24, U8, OP8_2("v_mul_u32_u24"), SINKU)
BENCH_KERNEL(k_mul_hi_u32_u24, U8, OP8_2("v_mul_hi_u32_u24"), SINKU)
BENCH_KERNEL(k_fma_f32, float a0=seed;float a1=seed+1;float a2=seed+2;float a3=seed+3;float a4=seed+4;float a5=seed+5;float a6=seed+6;float a7=seed+7;float b=0.5f, OP8("v_fma_f32"), if (a0+a1+a2+a3+a4+a5+a6+a7==1234.5f) out[1]=1)

// v_mad_u64_u32 dst64, vcc, a32, b32, c64
__global__ void k_mad_u64_u32(unsigned long long* out, double seed, int ITER) {
    uint64_t a0=(uint64_t)seed,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7; uint32_t b=(uint32_t)seed*2654435761u+1u, c = b ^ 0x5555u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0+a1+a2+a3+a4+a5+a6+a7 == 1234567) out[1]=1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, double, int);
int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_ldexp_f64", k_ldexp_f64},
        {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_rndne_f64", k_rndne_f64},
        {"v_fract_f64", k_fract_f64}, {"v_frexp_mant_f64", k_frexpm_f64},
        {"v_fma_f32", k_fma_f32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
        {"v_xor_b32", k_xor_b32}, {"v_bitop3_b32", k_xor3_b32}, {"v_add_u32", k_add_u32},
    };
    printf("%-20s %10s %10s %10s   (SIMD-cycles per wave-instruction at 2.4 GHz nominal, wall-clock, all CUs busy)\n", "op", "4w/SIMD", "8w/SIMD", "ticks@1w");
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    const int ITER = 100000;
    for (auto& e : ks) {
        double res[2];
        int idx = 0;
        for (int blocks_per_cu : {1, 2}) {
            hipLaunchKernelGGL(e.k, dim3(256 * blocks_per_cu), dim3(1024), 0, 0, d, 3.0, 1000);
            hipDeviceSynchronize();
            hipEventRecord(ea);
            hipLaunchKernelGGL(e.k, dim3(256 * blocks_per_cu), dim3(1024), 0, 0, d, 3.0, ITER);
            hipEventRecord(eb);
            hipEventSynchronize(eb);
            float ms; hipEventElapsedTime(&ms, ea, eb);
            double waves_per_simd = 4.0 * blocks_per_cu;
            double instr_per_simd = waves_per_simd * (double)ITER * REP;
            res[idx++] = (ms * 1e-3) * 2.4e9 / instr_per_simd;
        }
        hipLaunchKernelGGL(e.k, dim3(256), dim3(256), 0, 0, d, 3.0, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), d, 8 * 256, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        printf("%-20s %10.3f %10.3f %10.3f\n", e.name, res[0], res[1], sum / 256 / (2000.0 * REP));
    }
    return 0;
}
