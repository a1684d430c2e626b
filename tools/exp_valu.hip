// VALU issue-rate probe for gfx950 (development aid): cycles per wave64 instruction per SIMD
// for the instruction kinds the image kernels are made of.  8 independent chains per lane,
// 8 waves per SIMD, every CU busy; rate = total wave-instructions / (SIMDs * clock * time).
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_valu.hip -o /tmp/exp_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_ITER 4096
#define CHAINS 8
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a[CHAINS]; uint32_t u[CHAINS]; double dd[CHAINS]; uint64_t q[CHAINS];
    const uint64_t mask = __builtin_amdgcn_read_exec() ^ 0x5555555555555555ull;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); dd[i] = a[i]; q[i] = u[i]; }
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
            if (KIND == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) % CHAINS]));
            if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 7) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
            if (KIND == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 9) asm volatile("v_floor_f64 %0, %1" : "=v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 10) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(dd[i]));
            if (KIND == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(u[(i + 1) % CHAINS]) : "vcc");
            if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]) : "vcc");
            if (KIND == 13) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 14) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(u[i]), "v"(u[(i + 1) % CHAINS]) : "vcc");
            if (KIND == 15) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 16) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 18) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 19) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 20) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 21) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 22) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 23) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 24) asm volatile("v_floor_f32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 25) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 26) asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 28) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "s"(mask));
            if (KIND == 29) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 30) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 31) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 33) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 34) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 35) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 36) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 37) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]), "v"(seed));
            if (KIND == 38) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]) : "vcc");
            if (KIND == 39) asm volatile("v_cmp_lt_u32 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]) : "s20", "s21");
            if (KIND == 40) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[(i + 1) % CHAINS]), "v"(seed));
            if (KIND == 41) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[(i + 1) % CHAINS]), "v"(seed));
            if (KIND == 42) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 43) asm volatile("v_alignbit_b32 %0, %1, %1, 15" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 44) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 45) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
            if (KIND == 46) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 47) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 48) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dd[i]) : "v"(dd[(i + 1) % CHAINS]));
            if (KIND == 49) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 50) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 51) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 52) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[(i + 1) % CHAINS]));
            if (KIND == 53) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
            if (KIND == 27) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "v"(u[(i + 2) % CHAINS]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (float)u[i] + (float)dd[i] + (float)q[i];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> void run(const char* name) {
    float* out; hipMalloc(&out, 4);
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 4 * N_ITER * CHAINS;      // wave-instructions
    const double per_simd = winstr / 1024.0;
    printf("%-20s %8.3f ms  -> %6.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    run<40>("v_fma_mix_f32 hi"); run<41>("v_fma_mix_f32 lo"); run<37>("v_fmac_f32");
    run<0>("v_fma_f32"); run<1>("v_cvt_f32_ubyte1"); run<2>("v_add_u32"); run<12>("v_cndmask_b32"); run<14>("v_cmp_lt_u32");
    run<8>("v_alignbit_b32"); run<7>("v_cvt_pk_u8_f32"); run<15>("v_mov_b32_dpp wave_shr");
    run<6>("v_pk_fma_f32"); run<3>("v_add_f64"); run<13>("v_mul_f64"); run<4>("v_fma_f64"); run<9>("v_floor_f64"); run<10>("v_cvt_i32_f64");
    run<5>("v_lshl_add_u64"); run<11>("v_mad_u64_u32");
    run<16>("v_mad_u32_u24"); run<17>("v_perm_b32"); run<18>("v_dot2_u32_u16"); run<19>("v_dot4_u32_u8"); run<20>("v_pk_mov_b32");
    run<21>("v_mul_lo_u32"); run<22>("v_pk_mad_u16"); run<23>("v_sqrt_f32"); run<24>("v_floor_f32"); run<25>("v_bfe_u32");
    run<26>("v_mul_i32_i24_sdwa"); run<27>("v_add3_u32");
    run<28>("v_cndmask_b32_e64 sgpr"); run<29>("v_cndmask_b32_e32 vcc"); run<30>("v_sub_f32"); run<31>("v_mul_f32"); run<32>("v_and_b32");
    run<38>("cmp+cndmask via vcc (2 instr)"); run<39>("cmp+cndmask via sgpr pair (2 instr)");
    run<33>("v_lshrrev_b32"); run<34>("v_max_f32"); run<35>("v_mov_b32"); run<36>("v_or_b32"); run<37>("v_fmac_f32");
    run<42>("v_min3_u32"); run<43>("v_alignbit_b32 rot"); run<44>("v_lshl_add_u32"); run<45>("v_and_or_b32"); run<46>("v_add_f32");
    run<47>("v_pk_min_u16"); run<48>("v_pk_add_f32"); run<49>("v_xor_b32"); run<50>("v_sub_u32"); run<51>("v_cvt_f32_u32"); run<52>("v_cvt_u32_f32");
    run<53>("v_lshlrev_b32 1");
    return 0;
}
