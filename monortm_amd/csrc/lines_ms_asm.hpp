// lines_ms_asm.hpp - the inner loops of lines_ms_kernel (round 6): FIVE wavenumbers per lane, the lane's line records at a
// per-lane LDS address, generic molecules (KIND 0), double precision, gfx950 assembly.  Reference arithmetic:
// src/modm.f90:706-831 (LSF_LORTZ) as regrouped in lines_asm.hpp - a bracket a2 / den - pedestal formed by an FMA whose CLAMP
// is the 25 cm-1 test (modm.f90:384, :713).
//
// Why a second set of loops.  lines_kernel<double,1,1> gives a wave ONE atmospheric state and a lane one channel: 50 channels
// leave 14 of 64 lanes idle in every evaluate instruction, every record read (ds_read_b128) serves one evaluation per lane, and
// every wave pays the prologue of its state alone.  lines_ms_kernel gives a wave G states x LPS lanes, each lane WPS = 5
// channels of its state (configs[3]: 6 states x 10 lanes = 60 of 64 lanes).  A lane reads the records of ITS state (the address
// register differs between lanes, the class masks of a line are common to the wave), and a record that has been read serves
// five evaluations - so the loops below keep the records intact (d, den go to temporaries, not back into the record as in
// lines_asm.hpp) and repeat the arithmetic for the five (wavenumber, sum) operand pairs.
//
// One reciprocal for FOUR lines (one-resonance class): with P_A = den0 den1 (pair A), P_B = den2 den3 (pair B),
// r = 1 / (P_A P_B) (v_rcp_f64 + one Newton step), 1 / P_A = r P_B and 1 / P_B = r P_A: 28 vector instructions and one
// quarter-rate reciprocal per four lines and wavenumber where two pairs cost 28 + two (priced in LABNOTES round 4, built here
// because both pairs' records are resident anyway).  Products of four denominators stay far inside the double range
// (den in [1e-14, 1e4]).  Two-resonance lines go in pairs (26 instructions: LA_PAIR_K0_M1 of lines_asm.hpp, non-destructive).
//
// Registers: v[64:83] TM0 .. TM9 temporaries (O2's two-resonance pair needs ten), v[84:99] record set A (two lines), v[100:115] set B.
// Classes of a generic molecule here: ONE (one resonance; tested or not - the clamp is a no-op for an untested line) and TWO
// (two resonances, tested or not).  Hazards the assembler does not see inside an asm block (gfx940+): the result of v_rcp_f64 is
// not read by the next instruction.
#pragma once

#define MS_TM0 "v[64:65]"
#define MS_TM1 "v[66:67]"
#define MS_TM2 "v[68:69]"
#define MS_TM3 "v[70:71]"
#define MS_TM4 "v[72:73]"
#define MS_TM5 "v[74:75]"
#define MS_TM6 "v[76:77]"
#define MS_TM7 "v[78:79]"
#define MS_TM8 "v[80:81]"
#define MS_TM9 "v[82:83]"
#define MS_A_T0 "v[84:87]"
#define MS_A_U0 "v[88:91]"
#define MS_A_X0 "v[84:85]"
#define MS_A_H0 "v[86:87]"
#define MS_A_A0 "v[88:89]"
#define MS_A_P0 "v[90:91]"
#define MS_A_T1 "v[92:95]"
#define MS_A_U1 "v[96:99]"
#define MS_A_X1 "v[92:93]"
#define MS_A_H1 "v[94:95]"
#define MS_A_A1 "v[96:97]"
#define MS_A_P1 "v[98:99]"
#define MS_B_T0 "v[100:103]"
#define MS_B_U0 "v[104:107]"
#define MS_B_X0 "v[100:101]"
#define MS_B_H0 "v[102:103]"
#define MS_B_A0 "v[104:105]"
#define MS_B_P0 "v[106:107]"
#define MS_B_T1 "v[108:111]"
#define MS_B_U1 "v[112:115]"
#define MS_B_X1 "v[108:109]"
#define MS_B_H1 "v[110:111]"
#define MS_B_A1 "v[112:113]"
#define MS_B_P1 "v[114:115]"

#define MS_CLOBBERS                                                                                                            \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",   \
        "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",     \
        "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",     \
        "v114", "v115", "scc", "memory"

#define MS_I(x) x "\n\t"
#define MS_NEWTON                                                 \
    MS_I("v_fma_f64 " MS_TM0 ", -" MS_TM0 ", " MS_TM1 ", 1.0")     \
    MS_I("v_fma_f64 " MS_TM1 ", " MS_TM0 ", " MS_TM1 ", " MS_TM1)

// ---- four one-resonance lines (sets A and B) for the wavenumber / sum operands W, S: 28 instructions -------------------------
#define MS_QUAD1(W, S)                                                                    \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_A_X0)                                     \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_A_X1)                                     \
    MS_I("v_add_f64 " MS_TM4 ", %[" W "], -" MS_B_X0)                                     \
    MS_I("v_add_f64 " MS_TM5 ", %[" W "], -" MS_B_X1)                                     \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM2 ", " MS_A_H0)                        \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM3 ", " MS_A_H1)                        \
    MS_I("v_fma_f64 " MS_TM4 ", " MS_TM4 ", " MS_TM4 ", " MS_B_H0)                        \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM5 ", " MS_TM5 ", " MS_B_H1)                        \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_TM2 ", " MS_TM3)                                     \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_TM4 ", " MS_TM5)                                     \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM6 ", " MS_TM7)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("v_mul_f64 " MS_TM3 ", " MS_A_A0 ", " MS_TM3)                                    \
    MS_I("v_mul_f64 " MS_TM2 ", " MS_A_A1 ", " MS_TM2)                                    \
    MS_I("v_mul_f64 " MS_TM5 ", " MS_B_A0 ", " MS_TM5)                                    \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_B_A1 ", " MS_TM4)                                    \
    MS_NEWTON                                                                             \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM1 ", " MS_TM7)                                     \
    MS_I("v_mul_f64 " MS_TM1 ", " MS_TM1 ", " MS_TM6)                                     \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM0 ", -" MS_A_P0 " clamp")              \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM0 ", -" MS_A_P1 " clamp")              \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM5 ", " MS_TM1 ", -" MS_B_P0 " clamp")              \
    MS_I("v_fma_f64 " MS_TM4 ", " MS_TM4 ", " MS_TM1 ", -" MS_B_P1 " clamp")              \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM3)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM2)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM5)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM4)

// ---- two one-resonance lines of set A (the odd pair of a run): 14 instructions, the arithmetic of LA_PAIR_K0_M0_T1 --------------
#define MS_PAIR1Z(Z, W, S)                                                                \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_##Z##_X0)                                     \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_##Z##_X1)                                     \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM2 ", " MS_##Z##_H0)                        \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM3 ", " MS_##Z##_H1)                        \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM2 ", " MS_TM3)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("v_mul_f64 " MS_TM3 ", " MS_##Z##_A0 ", " MS_TM3)                                    \
    MS_I("v_mul_f64 " MS_TM2 ", " MS_##Z##_A1 ", " MS_TM2)                                    \
    MS_NEWTON                                                                             \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM1 ", -" MS_##Z##_P0 " clamp")              \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM1 ", -" MS_##Z##_P1 " clamp")              \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM3)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM2)
#define MS_PAIR1(W, S) MS_PAIR1Z(A, W, S)

// ---- two two-resonance lines of set Z (A or B): 24 instructions - LA_PAIR_K0_M1 (pb = pa: fast-class lines carry no Y factors) with
// the second denominator from the first: (WN + Xnu)^2 + HW^2 = (WN - Xnu)^2 + HW^2 + 4 Xnu WN, one FMA on den1 instead of an add and
// an FMA (no cancellation: every term is positive).  Brackets added in the order (+) line 0, (-) line 0, (+) line 1, (-) line 1
#define MS_PAIR2(Z, W, S)                                                                 \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_##Z##_X0)                                 \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_##Z##_X1)                                 \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM2 ", " MS_##Z##_H0)                    \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM3 ", " MS_##Z##_H1)                    \
    MS_I("v_fma_f64 " MS_TM4 ", " MS_TM8 ", %[" W "], " MS_TM2)                           \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM9 ", %[" W "], " MS_TM3)                           \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_TM2 ", " MS_TM4)                                     \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_TM3 ", " MS_TM5)                                     \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM6 ", " MS_TM7)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_##Z##_A0 ", " MS_TM7)                                \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_##Z##_A1 ", " MS_TM6)                                \
    MS_NEWTON                                                                             \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_TM7 ", " MS_TM1)                                     \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_TM6 ", " MS_TM1)                                     \
    MS_I("v_fma_f64 " MS_TM4 ", " MS_TM7 ", " MS_TM4 ", -" MS_##Z##_P0 " clamp")          \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM7 ", " MS_TM2 ", -" MS_##Z##_P0 " clamp")          \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM6 ", " MS_TM5 ", -" MS_##Z##_P1 " clamp")          \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM6 ", " MS_TM3 ", -" MS_##Z##_P1 " clamp")          \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM4)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM2)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM5)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM3)
// ... preceded once per pair (not per wavenumber) by 4 Xnu of its two lines -> TM8 / TM9
#define MS_PAIR2_PRE(Z)                                                                   \
    MS_I("v_mul_f64 " MS_TM8 ", 4.0, " MS_##Z##_X0)                                       \
    MS_I("v_mul_f64 " MS_TM9 ", 4.0, " MS_##Z##_X1)

// ---- the odd line at the end of a run (line 0 of set A), by itself: the arithmetic of uni_single<0, M2, true> (lines_device.hpp) ----
#define MS_SINGLE1(W, S)                                                                  \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_A_X0)                                     \
    MS_I("v_fma_f64 " MS_TM0 ", " MS_TM2 ", " MS_TM2 ", " MS_A_H0)                        \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("s_nop 0")                                                                       \
    MS_NEWTON                                                                             \
    MS_I("v_fma_f64 " MS_TM0 ", " MS_A_A0 ", " MS_TM1 ", -" MS_A_P0 " clamp")             \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM0)
#define MS_SINGLE2(W, S)                                                                  \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_A_X0)                                     \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], " MS_A_X0)                                      \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM2 ", " MS_TM2 ", " MS_A_H0)                        \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM3 ", " MS_TM3 ", " MS_A_H0)                        \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM2 ", " MS_TM3)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("s_nop 0")                                                                       \
    MS_NEWTON                                                                             \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_A_A0 ", " MS_TM1)                                    \
    MS_I("v_fma_f64 " MS_TM3 ", " MS_TM4 ", " MS_TM3 ", -" MS_A_P0 " clamp")              \
    MS_I("v_fma_f64 " MS_TM2 ", " MS_TM4 ", " MS_TM2 ", -" MS_A_P0 " clamp")              \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM3)                                         \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM2)
#define MS_SINGLE1_ALL MS_IFK("r0", 1, MS_SINGLE1("w0", "s0")) MS_IFK("r1", 1, MS_SINGLE1("w1", "s1")) MS_IFK("r2", 1, MS_SINGLE1("w2", "s2")) MS_IFK("r3", 1, MS_SINGLE1("w3", "s3")) MS_IFK("r4", 1, MS_SINGLE1("w4", "s4"))
#define MS_SINGLE2_ALL MS_IFQ("q0", "r0", 1, MS_SINGLE2("w0", "s0"), MS_SINGLE1("w0", "s0")) MS_IFQ("q1", "r1", 1, MS_SINGLE2("w1", "s1"), MS_SINGLE1("w1", "s1")) MS_IFQ("q2", "r2", 1, MS_SINGLE2("w2", "s2"), MS_SINGLE1("w2", "s2")) MS_IFQ("q3", "r3", 1, MS_SINGLE2("w3", "s3"), MS_SINGLE1("w3", "s3")) MS_IFQ("q4", "r4", 1, MS_SINGLE2("w4", "s4"), MS_SINGLE1("w4", "s4"))

// ---- slots out of reach.  Slot k of a wave = the k-th wavenumbers of all its lanes = LPS consecutive channels of every state.  The
// prepare stage leaves per slot a mask R_k of the lines that reach ANY channel of the slot (|WN - Xnu| <= 25 cm-1 for some state and
// channel, modm.f90:384 / :755; the negative resonance's WN + Xnu <= 25 implies it): bit 0 = the current line.  A group of lines
// none of which reaches the slot adds nothing there - every bracket clamps to zero / every lane is masked off - so the whole
// block is skipped: configs[3] (channels over 0.3-30 cm-1, lines to 55 cm-1) 17 % of the (four lines, slot) blocks.
// NB = 15 / 3 / 1: the group is four lines / a pair / one line (tested against the moving masks m15 / m3 / m1, MS_SHIFT_R).  x is scratch
// (free outside the run-length step).
#define MS_IFK(R, NB, BODY)                                                               \
    MS_I("s_and_b64 %[x], %[" R "], %[m" #NB "]")                                         \
    MS_I("s_cbranch_scc0 77f")                                                            \
    BODY                                                                                  \
    "77:\n\t"
// the masks stay where they are; the bits of the current group move: m1 = 1 << pos, m3 = 3 << pos, m15 = 15 << pos (pos = lines
// of the run walked so far) - three scalar shifts per group instead of one per mask
#define MS_SHIFT_R(N)                                                                     \
    MS_I("s_lshl_b64 %[m1], %[m1], " #N)                                                  \
    MS_I("s_lshl_b64 %[m3], %[m3], " #N)                                                  \
    MS_I("s_lshl_b64 %[m15], %[m15], " #N)
// ... and, for two-resonance lines, per slot a mask Q_k of the lines whose NEGATIVE resonance reaches a channel of the slot
// (WN + Xnu <= 25 for some state and channel): where it does not, the second bracket clamps to zero for every lane and the
// one-resonance arithmetic serves (14 instead of 24 instructions a pair).  A line at 20 cm-1 has its negative resonance within
// reach of channels below 5 cm-1 only - one slot of five on configs[3]'s channel set; about half of the (two-resonance line,
// slot) blocks go this way.
#define MS_IFQ(Q, R, NB, BODY2, BODY1)                                                    \
    MS_I("s_and_b64 %[x], %[" Q "], %[m" #NB "]")                                         \
    MS_I("s_cbranch_scc0 78f")                                                            \
    BODY2                                                                                 \
    MS_I("s_branch 77f")                                                                  \
    "78:\n\t"                                                                             \
    MS_I("s_and_b64 %[x], %[" R "], %[m" #NB "]")                                         \
    MS_I("s_cbranch_scc0 77f")                                                            \
    BODY1                                                                                 \
    "77:\n\t"
#define MS_SHIFT_Q(N)

// the five wavenumbers of the lane
#define MS_QUAD1_ALL MS_IFK("r0", 15, MS_QUAD1("w0", "s0")) MS_IFK("r1", 15, MS_QUAD1("w1", "s1")) MS_IFK("r2", 15, MS_QUAD1("w2", "s2")) MS_IFK("r3", 15, MS_QUAD1("w3", "s3")) MS_IFK("r4", 15, MS_QUAD1("w4", "s4")) MS_SHIFT_R(4) MS_SHIFT_Q(4)
#define MS_PAIR1_ALL MS_IFK("r0", 3, MS_PAIR1("w0", "s0")) MS_IFK("r1", 3, MS_PAIR1("w1", "s1")) MS_IFK("r2", 3, MS_PAIR1("w2", "s2")) MS_IFK("r3", 3, MS_PAIR1("w3", "s3")) MS_IFK("r4", 3, MS_PAIR1("w4", "s4")) MS_SHIFT_R(2) MS_SHIFT_Q(2)
#define MS_PAIR2_ALL(Z) MS_PAIR2_PRE(Z) MS_IFQ("q0", "r0", 3, MS_PAIR2(Z, "w0", "s0"), MS_PAIR1Z(Z, "w0", "s0")) MS_IFQ("q1", "r1", 3, MS_PAIR2(Z, "w1", "s1"), MS_PAIR1Z(Z, "w1", "s1")) MS_IFQ("q2", "r2", 3, MS_PAIR2(Z, "w2", "s2"), MS_PAIR1Z(Z, "w2", "s2")) MS_IFQ("q3", "r3", 3, MS_PAIR2(Z, "w3", "s3"), MS_PAIR1Z(Z, "w3", "s3")) MS_IFQ("q4", "r4", 3, MS_PAIR2(Z, "w4", "s4"), MS_PAIR1Z(Z, "w4", "s4")) MS_SHIFT_R(2) MS_SHIFT_Q(2)

// ---- LDS reads at literal byte offsets from the lane's address register ---------------------------------------------------------
#define MS_LOAD(Z, o0, o1, o2, o3)                                                        \
    MS_I("ds_read_b128 " MS_##Z##_T0 ", %[addr] offset:" #o0)                             \
    MS_I("ds_read_b128 " MS_##Z##_T1 ", %[addr] offset:" #o2)                             \
    MS_I("ds_read_b128 " MS_##Z##_U0 ", %[addr] offset:" #o1)                             \
    MS_I("ds_read_b128 " MS_##Z##_U1 ", %[addr] offset:" #o3)

// ---- run control (lines_asm.hpp, LA_RUNLEN, with the one mask that matters here) -------------------------------------------------
// x <- the pairs (even bits) that are NOT of this class from p(M) = M | M >> 1;  k <- pairs of the run (>= 1), n / M advanced
// past it, k2 <- trips of two pairs
#define MS_PM MS_I("s_lshr_b64 %[x], %[M], 1") MS_I("s_or_b64 %[x], %[x], %[M]")
#define MS_RUNLEN                                                                         \
    MS_I("s_and_b64 %[x], %[x], %[c55]")                                                  \
    MS_I("s_ff1_i32_b64 %[k], %[x]")                                                      \
    MS_I("s_lshr_b32 %[k], %[k], 1")                                                      \
    MS_I("s_lshr_b32 %[k2], %[n], 1")                                                     \
    MS_I("s_min_u32 %[k], %[k], %[k2]")                                                   \
    MS_I("s_lshl_b32 %[k2], %[k], 1")                                                     \
    MS_I("s_sub_i32 %[n], %[n], %[k2]")                                                   \
    MS_I("s_lshr_b64 %[M], %[M], %[k2]")                                                  \
    MS_I("s_lshr_b32 %[k2], %[k], 1")                                                     \
    MS_I("s_cmp_eq_u32 %[k2], 0")
#define MS_TRIP_END(L1)                                                                   \
    MS_I("v_add_u32_e32 %[addr], 0x80, %[addr]")                                          \
    MS_I("s_add_i32 %[k2], %[k2], -1")                                                    \
    MS_I("s_cmp_lg_u32 %[k2], 0")                                                         \
    MS_I("s_cbranch_scc1 " L1 "b")

// Invariant at the dispatcher (label 90) and at every class entry: the records of the pair at addr are in flight into, or present
// in, set A.  LDS returns in order.
#define MS_RUN_K0                                                                         \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_LOAD(A, 0, 16, 32, 48)                                                             \
    "90:\n\t"                                                                             \
    MS_I("s_cmp_lt_i32 %[n], 2")                                                          \
    MS_I("s_cbranch_scc1 99f")                                                            \
    MS_I("s_and_b64 %[x], %[M], 3")                                                       \
    MS_I("s_cbranch_scc1 30f")                                                            \
    /* class ONE: trips of four lines on one reciprocal, then the odd pair */             \
    MS_PM                                                                                 \
    MS_RUNLEN                                                                             \
    MS_I("s_cbranch_scc1 12f")                                                            \
    "11:\n\t"                                                                             \
    MS_LOAD(B, 64, 80, 96, 112)                                                           \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_QUAD1_ALL                                                                          \
    MS_LOAD(A, 128, 144, 160, 176)                                                        \
    MS_TRIP_END("11")                                                                     \
    "12:\n\t"                                                                             \
    MS_I("s_bitcmp1_b32 %[k], 0")                                                         \
    MS_I("s_cbranch_scc0 90b")                                                            \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_PAIR1_ALL                                                                          \
    MS_LOAD(A, 64, 80, 96, 112)                                                           \
    MS_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    MS_I("s_branch 90b")                                                                  \
    /* class TWO: pairs, alternating record sets */                                       \
    "30:\n\t"                                                                             \
    MS_PM                                                                                 \
    MS_I("s_not_b64 %[x], %[x]")                                                          \
    MS_RUNLEN                                                                             \
    MS_I("s_cbranch_scc1 32f")                                                            \
    "31:\n\t"                                                                             \
    MS_LOAD(B, 64, 80, 96, 112)                                                           \
    MS_I("s_waitcnt lgkmcnt(4)")                                                          \
    MS_PAIR2_ALL(A)                                                                       \
    MS_LOAD(A, 128, 144, 160, 176)                                                        \
    MS_I("s_waitcnt lgkmcnt(4)")                                                          \
    MS_PAIR2_ALL(B)                                                                       \
    MS_TRIP_END("31")                                                                     \
    "32:\n\t"                                                                             \
    MS_I("s_bitcmp1_b32 %[k], 0")                                                         \
    MS_I("s_cbranch_scc0 90b")                                                            \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_PAIR2_ALL(A)                                                                       \
    MS_LOAD(A, 64, 80, 96, 112)                                                           \
    MS_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    MS_I("s_branch 90b")                                                                  \
    "99:\n\t"                                                                             \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    /* the odd last line of the run (its record is line 0 of set A) */                    \
    MS_I("s_cmp_eq_u32 %[n], 1")                                                          \
    MS_I("s_cbranch_scc0 98f")                                                            \
    MS_I("s_mov_b32 %[n], 0")                                                             \
    MS_I("s_bitcmp1_b64 %[M], 0")                                                         \
    MS_I("s_cbranch_scc1 97f")                                                            \
    MS_SINGLE1_ALL                                                                        \
    MS_I("s_branch 98f")                                                                  \
    "97:\n\t"                                                                             \
    MS_SINGLE2_ALL                                                                        \
    "98:\n\t"                                                                             \
    "s_nop 0"

// ================= O2 (KIND 1): no pedestal; the limit on |WN - Xnu| sits in the record's pa slot (25, or +inf for a coupled
// line) and an ordinary line's limit on WN + Xnu is the same number - so no HotB.  The 25 cm-1 rule inside the shape function
// (modm.f90:755) and "negative resonance within reach" (:757) are EXEC masks set by v_cmpx; sv = the wave's EXEC on entry.
// Always the tested forms (an untested line passes every test).  Arithmetic of LA_PAIR_K1_M0_T1 / LA_PAIR_K1_M1_T1.
// one resonance, pair of set Z: 16 vector instructions
#define MS_PAIR1_K1(Z, W, S)                                                              \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_##Z##_X0)                                 \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_##Z##_X1)                                 \
    MS_I("v_fma_f64 " MS_TM4 ", " MS_TM2 ", " MS_TM2 ", " MS_##Z##_H0)                    \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM3 ", " MS_TM3 ", " MS_##Z##_H1)                    \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM4 ", " MS_TM5)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("v_mul_f64 " MS_TM5 ", " MS_##Z##_A0 ", " MS_TM5)                                \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_##Z##_A1 ", " MS_TM4)                                \
    MS_NEWTON                                                                             \
    MS_I("v_mul_f64 " MS_TM5 ", " MS_TM5 ", " MS_TM1)                                     \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_TM4 ", " MS_TM1)                                     \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM2 "|, " MS_##Z##_P0)                          \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM5)                                         \
    MS_I("s_mov_b64 exec, %[sv]")                                                         \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM3 "|, " MS_##Z##_P1)                          \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM4)                                         \
    MS_I("s_mov_b64 exec, %[sv]")
// two resonances, pair of set Z: d -> TM2 / TM3, d+ -> TM4 / TM5, den1 -> TM6 / TM7, e = den2 -> TM0 / TM1, P_i = den1 den2 -> TM8 / TM9;
// e += den1 for the lanes within reach of the negative resonance; then P -> TM4, r -> TM5, n_i = a2_i e_i -> TM0 / TM1.  28 instructions
#define MS_PAIR2_K1(Z, W, S)                                                              \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_##Z##_X0)                                 \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_##Z##_X1)                                 \
    MS_I("v_add_f64 " MS_TM4 ", %[" W "], " MS_##Z##_X0)                                  \
    MS_I("v_add_f64 " MS_TM5 ", %[" W "], " MS_##Z##_X1)                                  \
    MS_I("v_fma_f64 " MS_TM6 ", " MS_TM2 ", " MS_TM2 ", " MS_##Z##_H0)                    \
    MS_I("v_fma_f64 " MS_TM7 ", " MS_TM3 ", " MS_TM3 ", " MS_##Z##_H1)                    \
    MS_I("v_fma_f64 " MS_TM0 ", " MS_TM4 ", " MS_TM4 ", " MS_##Z##_H0)                    \
    MS_I("v_fma_f64 " MS_TM1 ", " MS_TM5 ", " MS_TM5 ", " MS_##Z##_H1)                    \
    MS_I("v_mul_f64 " MS_TM8 ", " MS_TM6 ", " MS_TM0)                                     \
    MS_I("v_mul_f64 " MS_TM9 ", " MS_TM7 ", " MS_TM1)                                     \
    MS_I("v_cmpx_le_f64_e64 %[cm], " MS_TM4 ", " MS_##Z##_P0)                             \
    MS_I("v_add_f64 " MS_TM0 ", " MS_TM0 ", " MS_TM6)                                     \
    MS_I("s_mov_b64 exec, %[sv]")                                                         \
    MS_I("v_cmpx_le_f64_e64 %[cm], " MS_TM5 ", " MS_##Z##_P1)                             \
    MS_I("v_add_f64 " MS_TM1 ", " MS_TM1 ", " MS_TM7)                                     \
    MS_I("s_mov_b64 exec, %[sv]")                                                         \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_TM8 ", " MS_TM9)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM5 ", " MS_TM4)                                             \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_##Z##_A0 ", " MS_TM0)                                \
    MS_I("v_mul_f64 " MS_TM1 ", " MS_##Z##_A1 ", " MS_TM1)                                \
    MS_I("v_fma_f64 " MS_TM4 ", -" MS_TM4 ", " MS_TM5 ", 1.0")                            \
    MS_I("v_fma_f64 " MS_TM5 ", " MS_TM4 ", " MS_TM5 ", " MS_TM5)                         \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM0 ", " MS_TM9)                                     \
    MS_I("v_mul_f64 " MS_TM1 ", " MS_TM1 ", " MS_TM8)                                     \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM0 ", " MS_TM5)                                     \
    MS_I("v_mul_f64 " MS_TM1 ", " MS_TM1 ", " MS_TM5)                                     \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM2 "|, " MS_##Z##_P0)                          \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM0)                                         \
    MS_I("s_mov_b64 exec, %[sv]")                                                         \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM3 "|, " MS_##Z##_P1)                          \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM1)                                         \
    MS_I("s_mov_b64 exec, %[sv]")

// ================= CO2 (KIND 2): one resonance, pedestal x (2 - d^2 / 625) (modm.f90:808-817), the 25 cm-1 rule as an EXEC mask.
// Arithmetic of LA_PAIR_K2_M0_T1: t_i = a2_i den_j r - pa_i f_i.  22 instructions
#define MS_PAIR1_K2(Z, W, S)                                                              \
    MS_I("v_add_f64 " MS_TM2 ", %[" W "], -" MS_##Z##_X0)                                 \
    MS_I("v_add_f64 " MS_TM3 ", %[" W "], -" MS_##Z##_X1)                                 \
    MS_I("v_mul_f64 " MS_TM4 ", " MS_TM2 ", " MS_TM2)                                     \
    MS_I("v_mul_f64 " MS_TM5 ", " MS_TM3 ", " MS_TM3)                                     \
    MS_I("v_fma_f64 " MS_TM6 ", " MS_TM2 ", " MS_TM2 ", " MS_##Z##_H0)                    \
    MS_I("v_fma_f64 " MS_TM7 ", " MS_TM3 ", " MS_TM3 ", " MS_##Z##_H1)                    \
    MS_I("v_mul_f64 " MS_TM0 ", " MS_TM6 ", " MS_TM7)                                     \
    MS_I("v_rcp_f64_e32 " MS_TM1 ", " MS_TM0)                                             \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_##Z##_A0 ", " MS_TM7)                                \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_##Z##_A1 ", " MS_TM6)                                \
    MS_I("v_fma_f64 " MS_TM4 ", -" MS_TM4 ", %[c625], 2.0")                               \
    MS_I("v_fma_f64 " MS_TM5 ", -" MS_TM5 ", %[c625], 2.0")                               \
    MS_NEWTON                                                                             \
    MS_I("v_mul_f64 " MS_TM7 ", " MS_TM7 ", " MS_TM1)                                     \
    MS_I("v_mul_f64 " MS_TM6 ", " MS_TM6 ", " MS_TM1)                                     \
    MS_I("v_fma_f64 " MS_TM7 ", -" MS_##Z##_P0 ", " MS_TM4 ", " MS_TM7)                   \
    MS_I("v_fma_f64 " MS_TM6 ", -" MS_##Z##_P1 ", " MS_TM5 ", " MS_TM6)                   \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM2 "|, %[c25]")                                \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM7)                                         \
    MS_I("s_mov_b64 exec, %[sv]")                                                         \
    MS_I("v_cmpx_ngt_f64_e64 %[cm], |" MS_TM3 "|, %[c25]")                                \
    MS_I("v_add_f64 %[" S "], %[" S "], " MS_TM6)                                         \
    MS_I("s_mov_b64 exec, %[sv]")

#define MS_ALL5(M, Z) MS_IFK("r0", 3, M(Z, "w0", "s0")) MS_IFK("r1", 3, M(Z, "w1", "s1")) MS_IFK("r2", 3, M(Z, "w2", "s2")) MS_IFK("r3", 3, M(Z, "w3", "s3")) MS_IFK("r4", 3, M(Z, "w4", "s4")) MS_SHIFT_R(2)

// a class whose lines go in pairs, the record sets alternating (L0 entry: the run length has been formed; L1 trip of two pairs; L2
// the odd pair)
#define MS_CLASS_PAIRS(L1, L2, PAIR)                                                      \
    MS_I("s_cbranch_scc1 " L2 "f")                                                        \
    L1 ":\n\t"                                                                            \
    MS_LOAD(B, 64, 80, 96, 112)                                                           \
    MS_I("s_waitcnt lgkmcnt(4)")                                                          \
    MS_ALL5(PAIR, A)                                                                      \
    MS_LOAD(A, 128, 144, 160, 176)                                                        \
    MS_I("s_waitcnt lgkmcnt(4)")                                                          \
    MS_ALL5(PAIR, B)                                                                      \
    MS_TRIP_END(L1)                                                                       \
    L2 ":\n\t"                                                                            \
    MS_I("s_bitcmp1_b32 %[k], 0")                                                         \
    MS_I("s_cbranch_scc0 90b")                                                            \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_ALL5(PAIR, A)                                                                      \
    MS_LOAD(A, 64, 80, 96, 112)                                                           \
    MS_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    MS_I("s_branch 90b")

// O2: classes ONE / TWO by the pair's "two resonances" bits
#define MS_RUN_K1                                                                         \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_LOAD(A, 0, 16, 32, 48)                                                             \
    MS_I("s_mov_b64 %[sv], exec")                                                         \
    "90:\n\t"                                                                             \
    MS_I("s_cmp_lt_i32 %[n], 2")                                                          \
    MS_I("s_cbranch_scc1 99f")                                                            \
    MS_I("s_and_b64 %[x], %[M], 3")                                                       \
    MS_I("s_cbranch_scc1 30f")                                                            \
    MS_PM                                                                                 \
    MS_RUNLEN                                                                             \
    MS_CLASS_PAIRS("11", "12", MS_PAIR1_K1)                                               \
    "30:\n\t"                                                                             \
    MS_PM                                                                                 \
    MS_I("s_not_b64 %[x], %[x]")                                                          \
    MS_RUNLEN                                                                             \
    MS_CLASS_PAIRS("31", "32", MS_PAIR2_K1)                                               \
    "99:\n\t"                                                                             \
    "s_waitcnt lgkmcnt(0)"
// CO2: one class
#define MS_RUN_K2                                                                         \
    MS_I("s_waitcnt lgkmcnt(0)")                                                          \
    MS_LOAD(A, 0, 16, 32, 48)                                                             \
    MS_I("s_mov_b64 %[sv], exec")                                                         \
    "90:\n\t"                                                                             \
    MS_I("s_cmp_lt_i32 %[n], 2")                                                          \
    MS_I("s_cbranch_scc1 99f")                                                            \
    MS_I("s_mov_b64 %[x], 0")                                                             \
    MS_RUNLEN                                                                             \
    MS_CLASS_PAIRS("11", "12", MS_PAIR1_K2)                                               \
    "99:\n\t"                                                                             \
    "s_waitcnt lgkmcnt(0)"

namespace {

// Generic molecule, five wavenumbers per lane.  addr: LDS byte address of the current line's HotA record OF THIS LANE'S STATE (the
// arrays of the states of a wave are laid out alike, so one wave-uniform line index serves all lanes); n: lines left in the run;
// R[k]: lines that reach slot k (see MS_IFK), Q[k]: lines whose negative resonance reaches slot k (MS_IFQ), bit 0 = current line;
// M: "two resonances" mask, bit 0 = current line (a pair takes the class of the more general of its two lines).  Walks the whole
// run, the odd last line by itself: leaves n = 0.  The record arrays must be readable two records past the run (read-ahead).
__device__ __forceinline__ void ms_run_k0(unsigned &addr, int &n, unsigned long long &M, const unsigned long long (&R)[5], const unsigned long long (&Q)[5], const double (&W)[5], double (&S)[5]) {
    unsigned long long x, m1 = 1ull, m3 = 3ull, m15 = 15ull;
    int k, k2;
    const unsigned long long c55 = 0x5555555555555555ull;
    asm volatile(MS_RUN_K0
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [addr] "+v"(addr), [n] "+s"(n),
                   [M] "+s"(M), [x] "=&s"(x), [k] "=&s"(k), [k2] "=&s"(k2), [m1] "+s"(m1), [m3] "+s"(m3), [m15] "+s"(m15)
                 : [r0] "s"(R[0]), [r1] "s"(R[1]), [r2] "s"(R[2]), [r3] "s"(R[3]), [r4] "s"(R[4]), [q0] "s"(Q[0]), [q1] "s"(Q[1]), [q2] "s"(Q[2]),
                   [q3] "s"(Q[3]), [q4] "s"(Q[4]), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [c55] "s"(c55)
                 : MS_CLOBBERS);
}

// O2 / CO2, five wavenumbers per lane (the arguments of ms_run_k0; CO2 has no mask)
__device__ __forceinline__ void ms_run_k1(unsigned &addr, int &n, unsigned long long &M, const unsigned long long (&R)[5], const double (&W)[5], double (&S)[5]) {
    unsigned long long x, sv, cm, m1 = 1ull, m3 = 3ull, m15 = 15ull;
    int k, k2;
    const unsigned long long c55 = 0x5555555555555555ull;
    asm volatile(MS_RUN_K1
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [addr] "+v"(addr), [n] "+s"(n),
                   [M] "+s"(M), [x] "=&s"(x), [k] "=&s"(k), [k2] "=&s"(k2), [sv] "=&s"(sv), [cm] "=&s"(cm), [m1] "+s"(m1), [m3] "+s"(m3), [m15] "+s"(m15)
                 : [r0] "s"(R[0]), [r1] "s"(R[1]), [r2] "s"(R[2]), [r3] "s"(R[3]), [r4] "s"(R[4]), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [c55] "s"(c55)
                 : MS_CLOBBERS);
}
__device__ __forceinline__ void ms_run_k2(unsigned &addr, int &n, const unsigned long long (&R)[5], const double (&W)[5], double (&S)[5]) {
    unsigned long long x, sv, cm, M = 0ull, m1 = 1ull, m3 = 3ull, m15 = 15ull;
    int k, k2;
    const unsigned long long c55 = 0x5555555555555555ull;
    const double c25 = 25., c625 = 1.0 / 625.;
    asm volatile(MS_RUN_K2
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [addr] "+v"(addr), [n] "+s"(n),
                   [M] "+s"(M), [x] "=&s"(x), [k] "=&s"(k), [k2] "=&s"(k2), [sv] "=&s"(sv), [cm] "=&s"(cm), [m1] "+s"(m1), [m3] "+s"(m3), [m15] "+s"(m15)
                 : [r0] "s"(R[0]), [r1] "s"(R[1]), [r2] "s"(R[2]), [r3] "s"(R[3]), [r4] "s"(R[4]), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [c55] "s"(c55), [c25] "s"(c25), [c625] "s"(c625)
                 : MS_CLOBBERS);
}

}  // namespace
