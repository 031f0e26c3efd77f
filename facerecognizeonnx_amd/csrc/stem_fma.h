// stem_fma.h — the 27-tap x 16-channel multiply-accumulate of the u8 stem convolution as inline assembly (gfx950).
// Every lane multiplies its own 27 input values by the SAME 27 x 16 weights, so the weights are fetched through the scalar cache into
// SGPRs and used as the scalar operand of v_pk_fma_f32 (two channels per instruction: 8 packed FMAs per tap instead of 16 scalar ones —
// plain fp32 VALU FMAs run at half the packed rate, and this loop is VALU-issue bound).  Left to the compiler, all 27 x 16 uniform loads
// are hoisted to the top of the kernel and ~400 SGPRs spill through v_writelane; here one block handles one image row (9 taps) with two
// 16-SGPR banks: the load of tap t+1 is issued before the FMAs of tap t (SMEM returns out of order: the only safe wait is lgkmcnt(0)).
// Operands: %0..%7 = accumulator pairs (channels 2i, 2i+1), %8..%12 = input pairs (x0,x1) (x2,x3) (x4,x5) (x6,x7) (x8,-), %13 = weight
// row pointer ([tap][16] floats, 64 B per tap).  A tap in the low / high half of its pair is broadcast to both channels by op_sel.
#pragma once
typedef float fh_v2f __attribute__((ext_vector_type(2)));
#define FH_PK_LO " op_sel_hi:[1,0,1]\n"
#define FH_PK_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
#define FH_PK_BANK(S0, S1, S2, S3, S4, S5, S6, S7, X, MOD)                                                                                  \
    "v_pk_fma_f32 %0, " S0 ", " X ", %0" MOD "v_pk_fma_f32 %1, " S1 ", " X ", %1" MOD "v_pk_fma_f32 %2, " S2 ", " X ", %2" MOD              \
    "v_pk_fma_f32 %3, " S3 ", " X ", %3" MOD "v_pk_fma_f32 %4, " S4 ", " X ", %4" MOD "v_pk_fma_f32 %5, " S5 ", " X ", %5" MOD              \
    "v_pk_fma_f32 %6, " S6 ", " X ", %6" MOD "v_pk_fma_f32 %7, " S7 ", " X ", %7" MOD
#define FH_PK_A(X, MOD) FH_PK_BANK("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", X, MOD)
#define FH_PK_B(X, MOD) FH_PK_BANK("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", X, MOD)
// ACC = fh_v2f[8], XP = fh_v2f[5], WROW = const float* (uniform), STRIDE_BYTES = bytes between taps (an integer constant expression)
#define FH_STEM_ROW_FMA(ACC, XP, WROW, STRIDE_BYTES)                                                                                        \
    asm volatile("s_load_dwordx16 s[36:51], %13, 0\n"                                                                                       \
                 "s_waitcnt lgkmcnt(0)\n"                                                                                                   \
                 "s_load_dwordx16 s[52:67], %13, %14\n" FH_PK_A("%8", FH_PK_LO) "s_waitcnt lgkmcnt(0)\n"                                    \
                 "s_load_dwordx16 s[36:51], %13, %14*2\n" FH_PK_B("%8", FH_PK_HI) "s_waitcnt lgkmcnt(0)\n"                                  \
                 "s_load_dwordx16 s[52:67], %13, %14*3\n" FH_PK_A("%9", FH_PK_LO) "s_waitcnt lgkmcnt(0)\n"                                  \
                 "s_load_dwordx16 s[36:51], %13, %14*4\n" FH_PK_B("%9", FH_PK_HI) "s_waitcnt lgkmcnt(0)\n"                                  \
                 "s_load_dwordx16 s[52:67], %13, %14*5\n" FH_PK_A("%10", FH_PK_LO) "s_waitcnt lgkmcnt(0)\n"                                 \
                 "s_load_dwordx16 s[36:51], %13, %14*6\n" FH_PK_B("%10", FH_PK_HI) "s_waitcnt lgkmcnt(0)\n"                                 \
                 "s_load_dwordx16 s[52:67], %13, %14*7\n" FH_PK_A("%11", FH_PK_LO) "s_waitcnt lgkmcnt(0)\n"                                 \
                 "s_load_dwordx16 s[36:51], %13, %14*8\n" FH_PK_B("%11", FH_PK_HI) "s_waitcnt lgkmcnt(0)\n" FH_PK_A("%12", FH_PK_LO)        \
                 : "+v"(ACC[0]), "+v"(ACC[1]), "+v"(ACC[2]), "+v"(ACC[3]), "+v"(ACC[4]), "+v"(ACC[5]), "+v"(ACC[6]), "+v"(ACC[7])           \
                 : "v"(XP[0]), "v"(XP[1]), "v"(XP[2]), "v"(XP[3]), "v"(XP[4]), "s"(WROW), "n"(STRIDE_BYTES)                                 \
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52",   \
                   "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "memory")
