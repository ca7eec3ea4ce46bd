// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// sweep kernel is made of, with 1 and 2 waves per SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) bench(long long *out, int iters, double seed)
{
    // 8 independent chains so dependent-issue latency does not dominate
    double d[8]; int i[8]; float f[8];
    for (int k = 0; k < 8; ++k) { d[k] = seed + k + threadIdx.x; i[k] = (int)seed + k + threadIdx.x; f[k] = (float)seed * 1e-3f + k; }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]), "+v"(i[4]) : );) }
        if (KIND == 1) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 3) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %4, %0\n v_mad_i32_i24 %1, %1, %4, %1\n v_mad_i32_i24 %2, %2, %4, %2\n v_mad_i32_i24 %3, %3, %4, %3" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        if (KIND == 4) { REP16(asm volatile("v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : );) }
        if (KIND == 5) { REP16(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(i[0]), "v"(i[1]), "v"(i[2]), "v"(i[3]));) }
        if (KIND == 6) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : );) }
        if (KIND == 7) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 8) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        if (KIND == 9) { REP16(asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]), "v"(i[5]), "v"(i[6]), "v"(i[7]));) }
        if (KIND == 10) { REP16(asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_scale_f64 %2, vcc, %2, %4, %2\n v_div_scale_f64 %3, vcc, %3, %4, %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]) : "vcc");) }
        if (KIND == 11) { REP16(asm volatile("v_div_fixup_f64 %0, %0, %4, %4\n v_div_fixup_f64 %1, %1, %4, %4\n v_div_fixup_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 12) { REP16(asm volatile("v_cmp_le_f64 vcc, %0, %4\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_cmp_le_f64 vcc, %1, %4\n v_addc_co_u32 %5, vcc, 0, %5, vcc" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]), "v"(i[0]) : "vcc");) }
        if (KIND == 13) { REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : );) }
        if (KIND == 14) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]) : );) }
        if (KIND == 15) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_eq_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]) : "vcc");) }
        if (KIND == 16) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]) : "s10", "s11");) }
        if (KIND == 17) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_ashrrev_i32 %1, 31, %1\n v_lshlrev_b32 %2, 3, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : );) }
        if (KIND == 18) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %0\n v_and_or_b32 %1, %1, %4, %1\n v_bfi_b32 %2, %2, %4, %2\n v_bfi_b32 %3, %3, %4, %3" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        if (KIND == 19) { REP16(asm volatile("v_cmp_eq_u32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cmp_eq_u32_e64 s[12:13], %1, %4\n v_cndmask_b32_e64 %1, %1, %4, s[12:13]" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]) : "s10", "s11", "s12", "s13");) }
        if (KIND == 20) { REP16(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        if (KIND == 21) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        if (KIND == 22) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]) : "v"(i[4]));) }
        // fp32 instructions of the dense tier 0 (round 4): is a packed fp32 instruction one issue slot or two?
        if (KIND == 23) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(f[4]));) }
        if (KIND == 24) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 25) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(d[4]));) }
        if (KIND == 26) { REP16(asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]) : "v"(i[0]), "v"(i[1]));) }
        if (KIND == 27) { REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1" : "+v"(f[0]) : );) }
        if (KIND == 28) { REP16(asm volatile("v_cmp_le_f32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cmp_le_f32_e64 s[12:13], %1, %4\n v_cndmask_b32_e64 %1, %1, %4, s[12:13]" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(f[4]) : "s10", "s11", "s12", "s13");) }
        if (KIND == 29) { REP16(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]) : "v"(i[0]), "v"(i[1]), "v"(i[2]), "v"(i[3]));) }
        if (KIND == 30) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(f[0]) : "v"(f[4]));) }
        if (KIND == 31) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[0]) : "v"(d[4]));) }
    }
    long long t1 = clock64();
    double acc = 0; int ia = 0;
    for (int k = 0; k < 8; ++k) { acc += d[k] + f[k]; ia += i[k]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 1.2345 && ia == 77) out[1] = 1;
}

template <int KIND>
void run(const char *name, long long *dout)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-20s", name);
    for (int waves = 1; waves <= 8; waves *= 2) {
        dim3 grid(256 * waves), block(256);          // one wave per SIMD per resident workgroup
        hipLaunchKernelGGL(bench<KIND>, grid, block, 0, 0, dout, 10, 1.5);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<KIND>, grid, block, 0, 0, dout, iters, 1.5);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // SIMD-nanoseconds per wave-instruction = time x 1024 SIMDs / (waves x 1024 x instr per wave)
        const double ns = ms * 1e6 / ((double)waves * iters * 64.0);
        printf("  w=%d: %.2f ns/instr/SIMD", waves, ns);
    }
    printf("\n");
}

int main()
{
    long long *dout;
    hipMalloc(&dout, 16);
    hipMemset(dout, 0, 16);
    run<0>("v_add_u32", dout);
    run<8>("v_and_b32", dout);
    run<14>("v_cndmask_b32", dout);
    run<3>("v_mad_i32_i24", dout);
    run<4>("v_bfe_i32", dout);
    run<9>("v_mov_b32_dpp", dout);
    run<13>("v_permlane16_swap", dout);
    run<1>("v_add_f64", dout);
    run<7>("v_mul_f64", dout);
    run<2>("v_fma_f64", dout);
    run<5>("v_cvt_f64_i32", dout);
    run<12>("v_cmp_le_f64+addc", dout);
    run<10>("v_div_scale_f64", dout);
    run<11>("v_div_fixup_f64", dout);
    run<6>("v_rcp_f64", dout);
    run<15>("cmp_e32+cndmask_vcc", dout);
    run<16>("cndmask_e64 sgpr", dout);
    run<19>("cmp_e64+cndmask_e64", dout);
    run<17>("lshlrev/ashrrev", dout);
    run<18>("and_or/bfi", dout);
    run<21>("v_mov_b32", dout);
    run<22>("mul_lo/hi_u32", dout);
    run<20>("ds_bpermute x4+wait", dout);
    run<23>("v_fma_f32", dout);
    run<24>("v_pk_fma_f32", dout);
    run<25>("v_pk_add_f32", dout);
    run<26>("v_cvt_f32_u32_sdwa", dout);
    run<29>("v_cvt_f32_i32", dout);
    run<28>("cmp_le_f32+cndmask", dout);
    run<27>("add_f32_dpp+nop x4 (dependent)", dout);
    run<30>("v_fma_f32 dependent", dout);
    run<31>("v_pk_fma_f32 dependent", dout);
    return 0;
}
