// Microbenchmark: what a wave64 v_fma_f32 costs on gfx950 by the number of VGPR source operands and by which register-file
// banks (register index mod 4) they come from, and what v_exp_f32 / v_log_f32 cost beside plain instructions.
// usage: hipcc -O3 --offload-arch=gfx950 tools/ubench_bank.hip -o tools/ubench_bank && tools/ubench_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"

#define REP8(x) x x x x x x x x
// MODE 0: fma v, v, s, const (one VGPR source)            MODE 1: three VGPR sources in three different banks
// MODE 2: three VGPR sources, all in the same bank          MODE 3: two VGPR sources (different banks) + SGPR
// MODE 4: the tap mix: 12 three-source fmas + v_log + v_exp MODE 5: the same with the two transcendentals replaced by fmas
// MODE 6: v_mul / v_fmac two-source VOP2 forms (different banks)
#define INIT_REGS                                                                                                                                          \
    "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n" \
    "v_mov_b32 v16, 0.5\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.5\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n" \
    "v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n v_mov_b32 v28, 0.5\n v_mov_b32 v29, 0.5\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.5\n" \
    "s_mov_b32 s20, %2\n"                                                                                                                                   \
    "1:\n"
#define LOOP_END "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v25\n"
#define CLOBBERS "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "scc"
// The whole loop is ONE asm statement (the compiler must not place anything of its own inside it).
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float r = threadIdx.x, res = 0.f;
    if constexpr (MODE == 0)
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v8, %3, 1.0\n v_fma_f32 v9, v9, %3, 1.0\n v_fma_f32 v10, v10, %3, 1.0\n v_fma_f32 v11, v11, %3, 1.0\n"
                                    "v_fma_f32 v12, v12, %3, 1.0\n v_fma_f32 v13, v13, %3, 1.0\n v_fma_f32 v14, v14, %3, 1.0\n v_fma_f32 v15, v15, %3, 1.0\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 1) // dst/src2 vN (bank N%4), src0 bank (N+1)%4, src1 bank (N+2)%4
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, v26, v8\n v_fma_f32 v9, v18, v27, v9\n v_fma_f32 v10, v19, v24, v10\n v_fma_f32 v11, v16, v25, v11\n"
                                    "v_fma_f32 v12, v21, v30, v12\n v_fma_f32 v13, v22, v31, v13\n v_fma_f32 v14, v23, v28, v14\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 2) // all three sources in the bank of the destination
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v16, v24, v8\n v_fma_f32 v9, v17, v25, v9\n v_fma_f32 v10, v18, v26, v10\n v_fma_f32 v11, v19, v27, v11\n"
                                    "v_fma_f32 v12, v20, v28, v12\n v_fma_f32 v13, v21, v29, v13\n v_fma_f32 v14, v22, v30, v14\n v_fma_f32 v15, v23, v31, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 3)
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, %3, v8\n v_fma_f32 v9, v18, %3, v9\n v_fma_f32 v10, v19, %3, v10\n v_fma_f32 v11, v16, %3, v11\n"
                                    "v_fma_f32 v12, v21, %3, v12\n v_fma_f32 v13, v22, %3, v13\n v_fma_f32 v14, v23, %3, v14\n v_fma_f32 v15, v20, %3, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 4) // per "tap": 12 plain + log + exp = 14 instructions
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, v26, v8\n v_fma_f32 v9, v18, v27, v9\n v_fma_f32 v10, v19, v24, v10\n v_log_f32 v24, v16\n v_fma_f32 v11, v16, v25, v11\n v_fma_f32 v12, v21, v30, v12\n v_fma_f32 v13, v22, v31, v13\n"
                                    "v_fma_f32 v14, v23, v28, v14\n v_fma_f32 v15, v20, v29, v15\n v_fma_f32 v8, v17, v26, v8\n v_exp_f32 v25, v17\n v_fma_f32 v9, v18, v27, v9\n v_fma_f32 v10, v19, v28, v10\n v_fma_f32 v11, v16, v29, v11\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 5)
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, v26, v8\n v_fma_f32 v9, v18, v27, v9\n v_fma_f32 v10, v19, v24, v10\n v_fma_f32 v24, v16, v16, v16\n v_fma_f32 v11, v16, v25, v11\n v_fma_f32 v12, v21, v30, v12\n v_fma_f32 v13, v22, v31, v13\n"
                                    "v_fma_f32 v14, v23, v28, v14\n v_fma_f32 v15, v20, v29, v15\n v_fma_f32 v8, v17, v26, v8\n v_fma_f32 v25, v17, v17, v17\n v_fma_f32 v9, v18, v27, v9\n v_fma_f32 v10, v19, v28, v10\n v_fma_f32 v11, v16, v29, v11\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 7) // inline constant as the third source (no constant-bus read?)
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, v26, 1.0\n v_fma_f32 v9, v18, v27, 1.0\n v_fma_f32 v10, v19, v24, 1.0\n v_fma_f32 v11, v16, v25, 1.0\n"
                                    "v_fma_f32 v12, v21, v30, 1.0\n v_fma_f32 v13, v22, v31, 1.0\n v_fma_f32 v14, v23, v28, 1.0\n v_fma_f32 v15, v20, v29, 1.0\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 8) // 32-bit literal (VOP2 encoding)
        asm volatile(INIT_REGS REP8("v_max_f32 v8, 0x3c23d70a, v17\n v_max_f32 v9, 0x3c23d70a, v18\n v_max_f32 v10, 0x3c23d70a, v19\n v_max_f32 v11, 0x3c23d70a, v16\n"
                                    "v_max_f32 v12, 0x3c23d70a, v21\n v_max_f32 v13, 0x3c23d70a, v22\n v_max_f32 v14, 0x3c23d70a, v23\n v_max_f32 v15, 0x3c23d70a, v20\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 9) // the same instruction with a VGPR in place of the literal
        asm volatile(INIT_REGS REP8("v_max_f32 v8, v26, v17\n v_max_f32 v9, v27, v18\n v_max_f32 v10, v24, v19\n v_max_f32 v11, v25, v16\n"
                                    "v_max_f32 v12, v30, v21\n v_max_f32 v13, v31, v22\n v_max_f32 v14, v28, v23\n v_max_f32 v15, v29, v20\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 10) // v_cndmask with its lane mask in an SGPR pair (VOP3) against VCC (VOP2)
        asm volatile("s_mov_b64 s[22:23], 0x5555\n s_mov_b64 vcc, 0x5555\n" INIT_REGS REP8("v_cndmask_b32_e64 v8, v17, v26, s[22:23]\n v_cndmask_b32_e64 v9, v18, v27, s[22:23]\n v_cndmask_b32_e64 v10, v19, v24, s[22:23]\n v_cndmask_b32_e64 v11, v16, v25, s[22:23]\n"
                                    "v_cndmask_b32_e64 v12, v21, v30, s[22:23]\n v_cndmask_b32_e64 v13, v22, v31, s[22:23]\n v_cndmask_b32_e64 v14, v23, v28, s[22:23]\n v_cndmask_b32_e64 v15, v20, v29, s[22:23]\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "s22", "s23", "vcc");
    if constexpr (MODE == 11)
        asm volatile("s_mov_b64 vcc, 0x5555\n" INIT_REGS REP8("v_cndmask_b32_e32 v8, v17, v26, vcc\n v_cndmask_b32_e32 v9, v18, v27, vcc\n v_cndmask_b32_e32 v10, v19, v24, vcc\n v_cndmask_b32_e32 v11, v16, v25, vcc\n"
                                    "v_cndmask_b32_e32 v12, v21, v30, vcc\n v_cndmask_b32_e32 v13, v22, v31, vcc\n v_cndmask_b32_e32 v14, v23, v28, vcc\n v_cndmask_b32_e32 v15, v20, v29, vcc\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "vcc");
    if constexpr (MODE == 12) // v_cvt_f32_ubyteN (the quantised node's plane decode)
        asm volatile(INIT_REGS REP8("v_cvt_f32_ubyte0 v8, v17\n v_cvt_f32_ubyte1 v9, v18\n v_cvt_f32_ubyte2 v10, v19\n v_cvt_f32_ubyte3 v11, v16\n"
                                    "v_cvt_f32_ubyte0 v12, v21\n v_cvt_f32_ubyte1 v13, v22\n v_cvt_f32_ubyte2 v14, v23\n v_cvt_f32_ubyte3 v15, v20\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 13) // v_max3 / v_min3 (three sources)
        asm volatile(INIT_REGS REP8("v_max3_f32 v8, v17, v26, v8\n v_min3_f32 v9, v18, v27, v9\n v_max3_f32 v10, v19, v24, v10\n v_min3_f32 v11, v16, v25, v11\n"
                                    "v_max3_f32 v12, v21, v30, v12\n v_min3_f32 v13, v22, v31, v13\n v_max3_f32 v14, v23, v28, v14\n v_min3_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 14)
        asm volatile(INIT_REGS REP8("v_add_f32 v8, v17, v8\n v_add_f32 v9, v18, v9\n v_add_f32 v10, v19, v10\n v_add_f32 v11, v16, v11\n"
                                    "v_add_f32 v12, v21, v12\n v_add_f32 v13, v22, v13\n v_add_f32 v14, v23, v14\n v_add_f32 v15, v20, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 15)
        asm volatile(INIT_REGS REP8("v_sub_f32 v8, v17, v26\n v_sub_f32 v9, v18, v27\n v_sub_f32 v10, v19, v24\n v_sub_f32 v11, v16, v25\n"
                                    "v_sub_f32 v12, v21, v30\n v_sub_f32 v13, v22, v31\n v_sub_f32 v14, v23, v28\n v_sub_f32 v15, v20, v29\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 16)
        asm volatile(INIT_REGS REP8("v_mul_f32 v8, v17, v26\n v_mul_f32 v9, v18, v27\n v_mul_f32 v10, v19, v24\n v_mul_f32 v11, v16, v25\n"
                                    "v_mul_f32 v12, v21, v30\n v_mul_f32 v13, v22, v31\n v_mul_f32 v14, v23, v28\n v_mul_f32 v15, v20, v29\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 17) // fma with source modifiers (-|a|) and the clamp output modifier
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, -|v17|, v26, v8 clamp\n v_fma_f32 v9, -|v18|, v27, v9 clamp\n v_fma_f32 v10, -|v19|, v24, v10 clamp\n v_fma_f32 v11, -|v16|, v25, v11 clamp\n"
                                    "v_fma_f32 v12, -|v21|, v30, v12 clamp\n v_fma_f32 v13, -|v22|, v31, v13 clamp\n v_fma_f32 v14, -|v23|, v28, v14 clamp\n v_fma_f32 v15, -|v20|, v29, v15 clamp\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 18) // a - b written as fma(b, -1.0, a)
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v26, -1.0, v17\n v_fma_f32 v9, v27, -1.0, v18\n v_fma_f32 v10, v24, -1.0, v19\n v_fma_f32 v11, v25, -1.0, v16\n"
                                    "v_fma_f32 v12, v30, -1.0, v21\n v_fma_f32 v13, v31, -1.0, v22\n v_fma_f32 v14, v28, -1.0, v23\n v_fma_f32 v15, v29, -1.0, v20\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 19) // v_pk_fma_f32 (two fmas per lane and instruction)
        asm volatile(INIT_REGS REP8("v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_pk_fma_f32 v[12:13], v[20:21], v[30:31], v[12:13]\n v_pk_fma_f32 v[14:15], v[22:23], v[28:29], v[14:15]\n"
                                    "v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_pk_fma_f32 v[12:13], v[20:21], v[30:31], v[12:13]\n v_pk_fma_f32 v[14:15], v[22:23], v[28:29], v[14:15]\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 20) // v_perm_b32 (byte select from two registers by a selector register)
        asm volatile(INIT_REGS REP8("v_perm_b32 v8, v17, v26, v24\n v_perm_b32 v9, v18, v27, v24\n v_perm_b32 v10, v19, v24, v25\n v_perm_b32 v11, v16, v25, v24\n"
                                    "v_perm_b32 v12, v21, v30, v24\n v_perm_b32 v13, v22, v31, v24\n v_perm_b32 v14, v23, v28, v24\n v_perm_b32 v15, v20, v29, v24\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 21)
        asm volatile(INIT_REGS REP8("v_and_or_b32 v8, v17, v26, v24\n v_and_or_b32 v9, v18, v27, v24\n v_and_or_b32 v10, v19, v24, v25\n v_and_or_b32 v11, v16, v25, v24\n"
                                    "v_and_or_b32 v12, v21, v30, v24\n v_and_or_b32 v13, v22, v31, v24\n v_and_or_b32 v14, v23, v28, v24\n v_and_or_b32 v15, v20, v29, v24\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 22)
        asm volatile(INIT_REGS REP8("v_bfe_u32 v8, v17, 8, 8\n v_bfe_u32 v9, v18, 8, 8\n v_bfe_u32 v10, v19, 16, 8\n v_bfe_u32 v11, v16, 8, 8\n"
                                    "v_bfe_u32 v12, v21, 8, 8\n v_bfe_u32 v13, v22, 16, 8\n v_bfe_u32 v14, v23, 8, 8\n v_bfe_u32 v15, v20, 8, 8\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 23)
        asm volatile(INIT_REGS REP8("v_lshl_or_b32 v8, v17, 15, v24\n v_lshl_or_b32 v9, v18, 15, v24\n v_lshl_or_b32 v10, v19, 15, v25\n v_lshl_or_b32 v11, v16, 15, v24\n"
                                    "v_lshl_or_b32 v12, v21, 15, v24\n v_lshl_or_b32 v13, v22, 15, v24\n v_lshl_or_b32 v14, v23, 15, v24\n v_lshl_or_b32 v15, v20, 15, v24\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 24)
        asm volatile(INIT_REGS REP8("v_cvt_f32_u32 v8, v17\n v_cvt_f32_u32 v9, v18\n v_cvt_f32_u32 v10, v19\n v_cvt_f32_u32 v11, v16\n"
                                    "v_cvt_f32_u32 v12, v21\n v_cvt_f32_u32 v13, v22\n v_cvt_f32_u32 v14, v23\n v_cvt_f32_u32 v15, v20\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 25) // mixed-precision fma: an fp16 half of a register as a source
        asm volatile(INIT_REGS REP8("v_fma_mix_f32 v8, v17, v26, v8 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v9, v18, v27, v9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v10, v19, v24, v10 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v11, v16, v25, v11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                                    "v_fma_mix_f32 v12, v21, v30, v12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v13, v22, v31, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v14, v23, v28, v14 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v15, v20, v29, v15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 26)
        asm volatile(INIT_REGS REP8("v_and_b32 v8, v17, v26\n v_and_b32 v9, v18, v27\n v_or_b32 v10, v19, v24\n v_or_b32 v11, v16, v25\n"
                                    "v_and_b32 v12, v21, v30\n v_and_b32 v13, v22, v31\n v_or_b32 v14, v23, v28\n v_or_b32 v15, v20, v29\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 27) // VOP3 encoding reading VCC
        asm volatile("s_mov_b64 vcc, 0x5555\n" INIT_REGS REP8("v_cndmask_b32_e64 v8, v17, v26, vcc\n v_cndmask_b32_e64 v9, v18, v27, vcc\n v_cndmask_b32_e64 v10, v19, v24, vcc\n v_cndmask_b32_e64 v11, v16, v25, vcc\n"
                                    "v_cndmask_b32_e64 v12, v21, v30, vcc\n v_cndmask_b32_e64 v13, v22, v31, vcc\n v_cndmask_b32_e64 v14, v23, v28, vcc\n v_cndmask_b32_e64 v15, v20, v29, vcc\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "vcc");
    if constexpr (MODE == 28) // compare into VCC + select from VCC, as compilers write it (4 pairs)
        asm volatile(INIT_REGS REP8("v_cmp_lt_f32_e32 vcc, v17, v26\n v_cndmask_b32_e32 v8, v18, v27, vcc\n v_cmp_lt_f32_e32 vcc, v19, v24\n v_cndmask_b32_e32 v10, v16, v25, vcc\n"
                                    "v_cmp_lt_f32_e32 vcc, v21, v30\n v_cndmask_b32_e32 v12, v22, v31, vcc\n v_cmp_lt_f32_e32 vcc, v23, v28\n v_cndmask_b32_e32 v14, v20, v29, vcc\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "vcc");
    if constexpr (MODE == 29) // compare into an SGPR pair + select from it (VOP3 both)
        asm volatile(INIT_REGS REP8("v_cmp_lt_f32_e64 s[22:23], v17, v26\n v_cndmask_b32_e64 v8, v18, v27, s[22:23]\n v_cmp_lt_f32_e64 s[24:25], v19, v24\n v_cndmask_b32_e64 v10, v16, v25, s[24:25]\n"
                                    "v_cmp_lt_f32_e64 s[26:27], v21, v30\n v_cndmask_b32_e64 v12, v22, v31, s[26:27]\n v_cmp_lt_f32_e64 s[28:29], v23, v28\n v_cndmask_b32_e64 v14, v20, v29, s[28:29]\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
    if constexpr (MODE == 30) // the same with independent work between compare and select
        asm volatile(INIT_REGS REP8("v_cmp_lt_f32_e32 vcc, v17, v26\n v_fma_f32 v9, v18, v27, v9\n v_cndmask_b32_e32 v8, v18, v27, vcc\n v_fma_f32 v11, v16, v25, v11\n"
                                    "v_cmp_lt_f32_e32 vcc, v21, v30\n v_fma_f32 v13, v22, v31, v13\n v_cndmask_b32_e32 v12, v22, v31, vcc\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS, "vcc");
    if constexpr (MODE == 31) // v_pk_fma_f32 and v_fma_f32 alternating
        asm volatile(INIT_REGS REP8("v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_fma_f32 v12, v21, v30, v12\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_fma_f32 v13, v22, v31, v13\n v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_fma_f32 v14, v23, v28, v14\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 32) // v_cvt_f32_ubyteN and v_fma_f32 alternating
        asm volatile(INIT_REGS REP8("v_cvt_f32_ubyte0 v8, v17\n v_fma_f32 v12, v21, v30, v12\n v_cvt_f32_ubyte1 v9, v18\n v_fma_f32 v13, v22, v31, v13\n v_cvt_f32_ubyte2 v10, v19\n v_fma_f32 v14, v23, v28, v14\n v_cvt_f32_ubyte3 v11, v16\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 33) // v_max / v_min and v_fma_f32 alternating
        asm volatile(INIT_REGS REP8("v_max_f32 v8, v26, v17\n v_fma_f32 v12, v21, v30, v12\n v_min_f32 v9, v27, v18\n v_fma_f32 v13, v22, v31, v13\n v_max_f32 v10, v24, v19\n v_fma_f32 v14, v23, v28, v14\n v_min_f32 v11, v25, v16\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 34) // v_exp / v_log and v_fma_f32 alternating
        asm volatile(INIT_REGS REP8("v_exp_f32 v8, v17\n v_fma_f32 v12, v21, v30, v12\n v_log_f32 v9, v18\n v_fma_f32 v13, v22, v31, v13\n v_exp_f32 v10, v19\n v_fma_f32 v14, v23, v28, v14\n v_log_f32 v11, v16\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 35) // v_exp / v_log only
        asm volatile(INIT_REGS REP8("v_exp_f32 v8, v17\n v_log_f32 v9, v18\n v_exp_f32 v10, v19\n v_log_f32 v11, v16\n v_exp_f32 v8, v17\n v_log_f32 v9, v18\n v_exp_f32 v10, v19\n v_log_f32 v11, v16\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 36) // fma with an SGPR source and three-VGPR fma alternating
        asm volatile(INIT_REGS REP8("v_fma_f32 v8, v17, %3, v8\n v_fma_f32 v12, v21, v30, v12\n v_fma_f32 v9, v18, %3, v9\n v_fma_f32 v13, v22, v31, v13\n v_fma_f32 v10, v19, %3, v10\n v_fma_f32 v14, v23, v28, v14\n v_fma_f32 v11, v16, %3, v11\n v_fma_f32 v15, v20, v29, v15\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 37) // v_pk_fma_f32 and v_cvt_f32_ubyteN alternating
        asm volatile(INIT_REGS REP8("v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_cvt_f32_ubyte0 v8, v17\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_cvt_f32_ubyte1 v9, v18\n v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_cvt_f32_ubyte2 v10, v19\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_cvt_f32_ubyte3 v11, v16\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 38) // one v_pk_fma_f32 per two v_fma_f32 (3 + 5)
        asm volatile(INIT_REGS REP8("v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_fma_f32 v12, v21, v30, v12\n v_fma_f32 v13, v22, v31, v13\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_fma_f32 v14, v23, v28, v14\n v_fma_f32 v15, v20, v29, v15\n v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_fma_f32 v12, v21, v30, v12\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 39) // v_exp / v_log and v_pk_fma_f32 alternating
        asm volatile(INIT_REGS REP8("v_exp_f32 v8, v17\n v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_log_f32 v9, v18\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n v_exp_f32 v10, v19\n v_pk_fma_f32 v[8:9], v[16:17], v[26:27], v[8:9]\n v_log_f32 v11, v16\n v_pk_fma_f32 v[10:11], v[18:19], v[24:25], v[10:11]\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    if constexpr (MODE == 6)
        asm volatile(INIT_REGS REP8("v_fmac_f32 v8, v17, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, v19, v24\n v_fmac_f32 v11, v16, v25\n"
                                    "v_fmac_f32 v12, v21, v30\n v_fmac_f32 v13, v22, v31\n v_fmac_f32 v14, v23, v28\n v_fmac_f32 v15, v20, v29\n") LOOP_END
                     : "=&v"(res) : "v"(r), "s"(iters), "s"(s) : CLOBBERS);
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int per_iter)
{
    float* out;
    const int nb = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, nb * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256>>>(out, 100, 0.999f);
    hipEventRecord(e0);
    k<MODE><<<nb, 256>>>(out, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * per_iter * blocks_per_cu;
    printf("%-52s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, blocks_per_cu, ms, ms * 1e6 / instr_per_simd);
    hipFree(out);
}

int main(int argc, char**)
{
    const bool only_new = argc > 1; // any argument: only the compare / select forms
    for (int w : {3, 8}) {
        if (only_new) {
            run<27>("v_cndmask_b32_e64 v, v, v, vcc (VOP3 reading VCC)", w, 64);
            run<28>("v_cmp_lt_f32 vcc + v_cndmask_b32_e32 ... vcc (pairs)", w, 64);
            run<29>("v_cmp_lt_f32_e64 s[n:n+1] + v_cndmask_b32_e64 (pairs)", w, 64);
            run<30>("v_cmp vcc, fma, v_cndmask_e32 vcc, fma", w, 64);
            run<31>("v_pk_fma_f32, v_fma_f32 alternating", w, 64);
            run<32>("v_cvt_f32_ubyteN, v_fma_f32 alternating", w, 64);
            run<33>("v_max/min_f32, v_fma_f32 alternating", w, 64);
            run<34>("v_exp/log_f32, v_fma_f32 alternating", w, 64);
            run<35>("v_exp/log_f32 only", w, 64);
            run<36>("v_fma_f32 with SGPR source, v_fma_f32 alternating", w, 64);
            run<37>("v_pk_fma_f32, v_cvt_f32_ubyteN alternating", w, 64);
            run<38>("3 v_pk_fma_f32 + 5 v_fma_f32", w, 64);
            run<39>("v_exp/log_f32, v_pk_fma_f32 alternating", w, 64);
            continue;
        }
        run<0>("fma v, v, s, const (1 VGPR source)", w, 64);
        run<3>("fma v, v, s, v (2 VGPR sources, 2 banks)", w, 64);
        run<1>("fma v, v, v, v (3 VGPR sources, 3 banks)", w, 64);
        run<2>("fma v, v, v, v (3 VGPR sources, 1 bank)", w, 64);
        run<6>("fmac v, v, v (VOP2; 3 VGPR sources, 3 banks)", w, 64);
        run<7>("fma v, v, v, 1.0 (inline constant source)", w, 64);
        run<8>("v_max_f32 v, literal, v (32-bit literal, VOP2)", w, 64);
        run<9>("v_max_f32 v, v, v (VOP2)", w, 64);
        run<10>("v_cndmask_b32 v, v, v, s[n:n+1] (mask in SGPRs, VOP3)", w, 64);
        run<11>("v_cndmask_b32 v, v, v, vcc (VOP2)", w, 64);
        run<12>("v_cvt_f32_ubyteN v, v", w, 64);
        run<13>("v_max3_f32 / v_min3_f32 v, v, v, v", w, 64);
        run<14>("v_add_f32 v, v, v (VOP2)", w, 64);
        run<15>("v_sub_f32 v, v, v (VOP2)", w, 64);
        run<16>("v_mul_f32 v, v, v (VOP2)", w, 64);
        run<17>("v_fma_f32 v, -|v|, v, v clamp (modifiers)", w, 64);
        run<18>("v_fma_f32 v, v, -1.0, v (a - b as an fma)", w, 64);
        run<19>("v_pk_fma_f32 (per instruction = two fmas per lane)", w, 64);
        run<20>("v_perm_b32 v, v, v, v", w, 64);
        run<21>("v_and_or_b32 v, v, v, v", w, 64);
        run<22>("v_bfe_u32 v, v, imm, imm", w, 64);
        run<23>("v_lshl_or_b32 v, v, imm, v", w, 64);
        run<24>("v_cvt_f32_u32 v, v", w, 64);
        run<25>("v_fma_mix_f32 v, v.h, v, v (fp16 half as a source)", w, 64);
        run<26>("v_and_b32 / v_or_b32 v, v, v (VOP2)", w, 64);
        run<27>("v_cndmask_b32_e64 v, v, v, vcc (VOP3 reading VCC)", w, 64);
        run<28>("v_cmp_lt_f32 vcc + v_cndmask_b32_e32 ... vcc (pairs)", w, 64);
        run<29>("v_cmp_lt_f32_e64 s[n:n+1] + v_cndmask_b32_e64 (pairs)", w, 64);
        run<30>("v_cmp vcc, fma, v_cndmask_e32 vcc, fma", w, 64);
        run<4>("tap mix: 12 three-source fma + v_log + v_exp", w, 112);
        run<5>("tap mix with the transcendentals as fma", w, 112);
    }
    return 0;
}
