// valu_rate.hip -- what one gfx950 SIMD sustains on the instruction mix of the fuzzyhip stream kernels.
//
// The scoring kernels are integer-issue bound, not HBM bound, so their roofline
// needs a measured ceiling: wave64 VALU instructions per second for the chip, at the occupancy the
// kernels run at (4 waves per SIMD) and at 1 / 2 / 8 for comparison.  Each kind runs a long unrolled
// loop of inline-asm instructions (nothing for the compiler to fold) on 256 CUs x 4 SIMDs; the shader
// clock is measured in the same launch (s_memtime ticks per 100 MHz s_memrealtime tick), so the result
// is cycles per instruction per SIMD as well as instructions per second.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_rate tools/valu_rate.hip && tools/bin/valu_rate > profiles/rNN_valu_rate.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x

enum { K_ADD, K_BITOP3, K_ADD3, K_SDWA, K_STEP1, K_STEP2, K_STEP_PK, K_STEP_PK_BR, K_SALU, K_LOOP_LDS, K_ADD3_V, K_ADD_S, K_STEP_PK_V, K_CMP_OR, K_CNDMASK, K_BFE_I, K_LOOP_ADDTID, K_LOOP_ADDTID_NONOP, N_KINDS };
static const char *kNames[N_KINDS] = {
    "v_add_u32 (4 independent chains)",
    "v_bitop3_b32 (4 independent chains)",
    "v_add3_u32 (4 independent chains)",
    "v_add_u32_sdwa byte select (4 independent chains)",
    "Hyyro step, ONE dependent chain: bitop3 -> add -> bitop3",
    "Hyyro step, two interleaved chains",
    "packed-kernel step, one chain: bitop3 -> add3(sgpr) -> bitop3",
    "packed-kernel step + s_cmp + s_cbranch (not taken) after every step, as the round-1 loop (VALU counted only)",
    "s_add_u32 / s_and_b32 / s_lshl_b32 / s_cmp (4 SALU per group; counted as instructions)",
    "stream inner loop: 8 x (sdwa address + ds_read_b32), wait, 8 dependent packed steps",
    "v_add3_u32, all three sources VGPRs (4 independent chains)",
    "v_add_u32 with an SGPR as src0 (4 independent chains)",
    "packed-kernel step with the stopper in a VGPR: bitop3 -> add3(vgpr) -> bitop3, one chain",
    "2 x v_cmp_eq_u32 (SGPR-pair results) + s_or_b64 (3 instructions per group)",
    "v_cndmask_b32 with an SGPR-pair mask (4 independent chains)",
    "v_bfe_i32 v, v, 0, 1 + v_lshrrev_b32 (2 per group, dependent)",
    "stream inner loop with rows [row][lane]: 8 x (s_bfe -> m0, s_nop, ds_read_addtid_b32), wait, 8 dependent packed steps (counted as the 32 instructions of the sdwa form)",
    "the same without the s_nop between the m0 write and the read (timing only: hazard not honoured)",
};
static const int kOpsPerIter[N_KINDS] = {32, 32, 32, 32, 24, 48, 24, 24, 32, 8 * (8 + 8 * 3), 32, 32, 24, 24, 32, 16, 8 * (8 + 8 * 3), 8 * (8 + 8 * 3)};

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *clk, int iters, unsigned sval)
{
    __shared__ unsigned lds[256 * 17];                         // 17 KB: eight blocks fit one CU
    if constexpr (KIND == K_LOOP_LDS || KIND == K_LOOP_ADDTID || KIND == K_LOOP_ADDTID_NONOP) {
        for (int i = threadIdx.x; i < 256 * 17; i += 256) lds[i] = i * 2654435761u;
        __syncthreads();
    }
    unsigned a = threadIdx.x, b = threadIdx.x * 3u, c = 5u, d = 7u, e = 9u, f = 11u, g = 13u, h = 15u;
    const unsigned col = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)lds + threadIdx.x * 17u * 4u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_ADD) {
            REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (KIND == K_BITOP3) {
            REP8(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == K_ADD3) {
            REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(sval));)
        } else if (KIND == K_SDWA) {
            REP8(asm volatile("v_add_u32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                              "v_add_u32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                              "v_add_u32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                              "v_add_u32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sval));)
        } else if (KIND == K_STEP1) {
            REP8(asm volatile("v_bitop3_b32 %1, %0, %2, %3 bitop3:0x80\n v_add_u32 %4, %0, %1\n v_bitop3_b32 %0, %4, %0, %1 bitop3:0xf6\n"
                         : "+v"(a), "+v"(c) : "v"(e), "v"(f), "v"(g));)
        } else if (KIND == K_STEP2) {
            REP8(asm volatile("v_bitop3_b32 %2, %0, %4, %5 bitop3:0x80\n v_bitop3_b32 %3, %1, %4, %5 bitop3:0x80\n v_add_u32 %6, %0, %2\n v_add_u32 %7, %1, %3\n v_bitop3_b32 %0, %6, %0, %2 bitop3:0xf6\n v_bitop3_b32 %1, %7, %1, %3 bitop3:0xf6\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (KIND == K_STEP_PK) {
            REP8(asm volatile("v_bitop3_b32 %1, %0, %2, %3 bitop3:0x80\n v_add3_u32 %4, %0, %1, %5\n v_bitop3_b32 %0, %4, %0, %1 bitop3:0xf6\n"
                         : "+v"(a), "+v"(c) : "v"(e), "v"(f), "v"(g), "s"(sval));)
        } else if (KIND == K_STEP_PK_BR) {
            REP8(asm volatile("v_bitop3_b32 %1, %0, %2, %3 bitop3:0x80\n v_add3_u32 %4, %0, %1, %5\n v_bitop3_b32 %0, %4, %0, %1 bitop3:0xf6\n"
                              "s_cmp_eq_u32 %5, 77\n s_cbranch_scc1 1f\n 1:\n"
                         : "+v"(a), "+v"(c) : "v"(e), "v"(f), "v"(g), "s"(sval) : "scc");)
        } else if (KIND == K_SALU) {
            unsigned s0 = sval, s1 = sval + 1u;
            REP8(asm volatile("s_add_u32 %0, %0, %2\n s_and_b32 %1, %1, %0\n s_lshl_b32 %1, %1, 1\n s_cmp_eq_u32 %0, %1\n"
                         : "+s"(s0), "+s"(s1) : "s"(sval) : "scc");)
            a += s0 + s1;
        } else if (KIND == K_ADD3_V) {
            REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == K_ADD_S) {
            REP8(asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sval));)
        } else if (KIND == K_STEP_PK_V) {
            REP8(asm volatile("v_bitop3_b32 %1, %0, %2, %3 bitop3:0x80\n v_add3_u32 %4, %0, %1, %5\n v_bitop3_b32 %0, %4, %0, %1 bitop3:0xf6\n"
                         : "+v"(a), "+v"(c) : "v"(e), "v"(f), "v"(g), "v"(h));)
        } else if (KIND == K_CMP_OR) {
            unsigned long long m0_, m1_;
            REP8(asm volatile("v_cmp_eq_u32 %0, %3, %2\n v_cmp_eq_u32 %1, %4, %2\n s_or_b64 %0, %0, %1\n"
                         : "=&s"(m0_), "=&s"(m1_) : "v"(a), "s"(sval), "s"(sval + 1u) : "scc");)
            a += (unsigned)m0_;
        } else if (KIND == K_CNDMASK) {
            unsigned long long mk = 0x5555555555555555ull ^ sval;
            REP8(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "s"(mk));)
        } else if (KIND == K_BFE_I) {
            REP8(asm volatile("v_bfe_i32 %1, %0, 0, 1\n v_lshrrev_b32 %0, 1, %0\n" : "+v"(a), "+v"(b));)
        } else if (KIND == K_LOOP_ADDTID || KIND == K_LOOP_ADDTID_NONOP) {
            // rows stored [row][lane] (256 B per row): the row address is wave-uniform and travels in m0
            // (ds_read_addtid_b32: address = m0 + offset + 4 * lane), no vector address arithmetic
            unsigned r0_, r1, r2, r3, r4, r5, r6, r7, t8, t9;
            const unsigned w0 = ((sval & 0x3Fu) << 8) | (((sval >> 6) & 0x3Fu) << 24), w1 = (((sval >> 3) & 0x3Fu) << 8) | (((sval >> 9) & 0x3Fu) << 24);
            const unsigned w2 = (((sval >> 5) & 0x3Fu) << 8) | (((sval >> 11) & 0x3Fu) << 24), w3 = (((sval >> 7) & 0x3Fu) << 8) | (((sval >> 13) & 0x3Fu) << 24);
#define NOPQ(x) x
            if (KIND == K_LOOP_ADDTID) {
                REP8(asm volatile(
                    "s_bfe_u32 m0, %11, 0x100000\n s_nop 0\n ds_read_addtid_b32 %0\n s_bfe_u32 m0, %11, 0x100010\n s_nop 0\n ds_read_addtid_b32 %1\n"
                    "s_bfe_u32 m0, %12, 0x100000\n s_nop 0\n ds_read_addtid_b32 %2\n s_bfe_u32 m0, %12, 0x100010\n s_nop 0\n ds_read_addtid_b32 %3\n"
                    "s_bfe_u32 m0, %13, 0x100000\n s_nop 0\n ds_read_addtid_b32 %4\n s_bfe_u32 m0, %13, 0x100010\n s_nop 0\n ds_read_addtid_b32 %5\n"
                    "s_bfe_u32 m0, %14, 0x100000\n s_nop 0\n ds_read_addtid_b32 %6\n s_bfe_u32 m0, %14, 0x100010\n s_nop 0\n ds_read_addtid_b32 %7\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_bitop3_b32 %8, %10, %0, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %1, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %2, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %3, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %4, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %5, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %6, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %7, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    : "=&v"(r0_), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(t8), "=&v"(t9), "+v"(a)
                    : "s"(w0), "s"(w1), "s"(w2), "s"(w3), "v"(f), "s"(sval) : "memory", "m0", "scc");)
            } else {
                REP8(asm volatile(
                    "s_bfe_u32 m0, %11, 0x100000\n ds_read_addtid_b32 %0\n s_bfe_u32 m0, %11, 0x100010\n ds_read_addtid_b32 %1\n"
                    "s_bfe_u32 m0, %12, 0x100000\n ds_read_addtid_b32 %2\n s_bfe_u32 m0, %12, 0x100010\n ds_read_addtid_b32 %3\n"
                    "s_bfe_u32 m0, %13, 0x100000\n ds_read_addtid_b32 %4\n s_bfe_u32 m0, %13, 0x100010\n ds_read_addtid_b32 %5\n"
                    "s_bfe_u32 m0, %14, 0x100000\n ds_read_addtid_b32 %6\n s_bfe_u32 m0, %14, 0x100010\n ds_read_addtid_b32 %7\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_bitop3_b32 %8, %10, %0, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %1, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %2, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %3, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %4, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %5, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %6, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    "v_bitop3_b32 %8, %10, %7, %15 bitop3:0x80\n v_add3_u32 %9, %10, %8, %16\n v_bitop3_b32 %10, %9, %10, %8 bitop3:0xf6\n"
                    : "=&v"(r0_), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(t8), "=&v"(t9), "+v"(a)
                    : "s"(w0), "s"(w1), "s"(w2), "s"(w3), "v"(f), "s"(sval) : "memory", "m0", "scc");)
            }
        } else {
            // one "record" of the stream kernels: eight row addresses (column + symbol byte), eight LDS reads,
            // one wait, eight dependent Hyyro steps on the rows read
            unsigned o0, o1, o2, o3, o4, o5, o6, o7, r0_, r1, r2, r3, r4, r5, r6, r7;
            REP8(asm volatile(
                "v_add_u32_sdwa %8, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                "v_add_u32_sdwa %9, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                "v_add_u32_sdwa %10, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                "v_add_u32_sdwa %11, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                "v_add_u32_sdwa %12, %19, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                "v_add_u32_sdwa %13, %19, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                "v_add_u32_sdwa %14, %19, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                "v_add_u32_sdwa %15, %19, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                "ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
                "ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_bitop3_b32 %8, %16, %0, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %1, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %2, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %3, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %4, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %5, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %6, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                "v_bitop3_b32 %8, %16, %7, %20 bitop3:0x80\n v_add3_u32 %9, %16, %8, %21\n v_bitop3_b32 %16, %9, %16, %8 bitop3:0xf6\n"
                : "=&v"(r0_), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7),
                  "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7), "+v"(a)
                : "v"(col), "s"(sval & 0x3C3C3C3Cu), "s"((sval >> 1) & 0x3C3C3C3Cu), "v"(f), "s"(sval) : "memory");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1_ - r0; }
}

struct Result { double ms, cyc_per_instr, ghz, ginstr_per_s; };

template <int KIND>
static Result run(int blocks_per_cu, int n_cus)
{
    const int blocks = n_cus * blocks_per_cu, iters = (KIND == K_LOOP_LDS || KIND == K_LOOP_ADDTID || KIND == K_LOOP_ADDTID_NONOP) ? 3000 : 20000;
    unsigned *d; unsigned long long *clk;
    if (hipMalloc(&d, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&clk, (size_t)blocks * 16) != hipSuccess) exit(2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, clk, iters / 10, 0x12345678u);
    (void)hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, clk, iters, 0x12345678u);
    (void)hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) exit(3);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 2);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    Result r;
    r.ms = ms;
    r.ghz = ticks / real * 0.1;                                    // s_memrealtime ticks at 100 MHz
    const double winstr = (double)blocks * 4 * iters * kOpsPerIter[KIND];
    r.ginstr_per_s = winstr / (ms * 1e-3) / 1e9;
    // cycles one SIMD spends per wave-instruction: in-kernel cycles x SIMDs / instructions
    r.cyc_per_instr = (ticks / blocks) * (n_cus * 4.0) / winstr;
    (void)hipFree(d); (void)hipFree(clk);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return r;
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"unit\": \"wave64 VALU instructions\", \"results\": [\n", prop.gcnArchName, n_cus, n_cus * 4);
    bool first = true;
    for (int w : {1, 2, 4, 8}) {
        Result (*fns[N_KINDS])(int, int) = {run<K_ADD>, run<K_BITOP3>, run<K_ADD3>, run<K_SDWA>, run<K_STEP1>, run<K_STEP2>, run<K_STEP_PK>, run<K_STEP_PK_BR>, run<K_SALU>, run<K_LOOP_LDS>, run<K_ADD3_V>, run<K_ADD_S>, run<K_STEP_PK_V>, run<K_CMP_OR>, run<K_CNDMASK>, run<K_BFE_I>, run<K_LOOP_ADDTID>, run<K_LOOP_ADDTID_NONOP>};
        for (int kind = 0; kind < N_KINDS; ++kind) {
            const Result r = fns[kind](w, n_cus);
            printf("%s  {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f, \"cycles_per_instr_per_simd\": %.3f, \"chip_ginstr_per_s\": %.1f}",
                   first ? "" : ",\n", kNames[kind], w, r.ms, r.ghz, r.cyc_per_instr, r.ginstr_per_s);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
