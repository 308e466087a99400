// Issue cost of packed fp32 VALU instructions for ONE wave on a SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_issue tools/microbench/pk_issue.hip && /tmp/pk_issue
// Question: a lone wave issues one v_fma_f32 per 4 cycles (half the SIMD-32's rate). Does a v_pk_fma_f32 (two fp32 FMAs per lane)
// also issue in 4? Then hand-paired arithmetic doubles what a one-wave-per-SIMD kernel gets out of an issue slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 32
#define NACC 16

template <int MODE>
__global__ void bench(unsigned long long* out, float seed) {
    float a = seed + threadIdx.x * 1e-3f, b = 0.999f;
    f2 a2 = {a, a + 1.f}, b2 = {b, b};
    float acc[NACC];
    f2 acc2[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = i + seed; acc2[i] = f2{i + seed, i - seed}; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int outer = 0; outer < 8; ++outer) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[0]) : "v"(a2), "v"(b2));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc2[i]) : "v"(a2));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc2[i]) : "v"(a2));
                if (MODE == 6) {  // alternate packed / scalar
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
                    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                }
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc2[i]) : "v"(a2), "v"(b2));  // broadcast low half of src0
                if (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[i]) : "v"(a));
                if (MODE == 9) asm volatile("v_fma_f32 %0, %1, %2, %0 " : "+v"(acc[i]) : "v"(a), "s"(b));
                if (MODE == 10) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a));
                if (MODE == 11) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "+v"(acc2[i]) : "v"(a2));
                if (MODE == 12) asm volatile("v_accvgpr_write_b32 a0, %1\n v_accvgpr_read_b32 %0, a0" : "+v"(acc[i]) : "v"(a) : "a0");
                if (MODE == 13) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
                if (MODE == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (MODE == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[0,0,1]" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)(s != 1234.5f); }
}

template <int MODE>
void run(const char* name, int nblocks, int insts_per_iter = 1) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * nblocks);
    bench<MODE><<<nblocks, 64>>>(d, 1.0f);
    bench<MODE><<<nblocks, 64>>>(d, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * nblocks);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nblocks, hipMemcpyDeviceToHost);
    double mean = 0; unsigned long long mx = 0;
    for (int i = 0; i < nblocks; ++i) { mean += h[2 * i]; if (h[2 * i] > mx) mx = h[2 * i]; }
    mean /= nblocks;
    double n = 8.0 * REP * NACC * insts_per_iter;
    printf("%-46s blocks %5d  %7.2f cycles/inst (mean), %7.2f (slowest wave)\n", name, nblocks, mean / n, mx / n);
    hipFree(d);
}

int main() {
    for (int nb : {1, 1024, 2048}) {
        run<0>("v_fma_f32 independent", nb);
        run<1>("v_fma_f32 dependent chain", nb);
        run<2>("v_pk_fma_f32 independent", nb);
        run<3>("v_pk_fma_f32 dependent chain", nb);
        run<4>("v_pk_mul_f32 independent", nb);
        run<5>("v_pk_add_f32 independent", nb);
        run<6>("alternating v_pk_fma_f32 / v_fma_f32", nb);
        run<7>("v_pk_fma_f32 op_sel_hi broadcast", nb);
        run<8>("v_mov_b32", nb);
        run<9>("v_fma_f32 with an SGPR operand", nb);
        run<10>("v_add_f32 DPP quad_perm", nb);
        run<11>("v_pk_mov_b32", nb);
        run<12>("v_accvgpr_write + v_accvgpr_read", nb, 2);
        run<13>("v_mul_f32", nb);
        run<14>("v_fmac_f32", nb);
        run<15>("v_pk_fma_f32 neg modifiers", nb);
    }
    return 0;
}
