// lines_asm.hpp - the inner loops of the line sum for one wavenumber per lane in double precision, written in gfx950
// assembly (round 4).  Reference arithmetic: src/modm.f90:706-831 (LSF_LORTZ), regrouped as in lines_device.hpp.
//
// Why assembly.  The class loops of rounds 1-3 (eval_pair / eval_pair_fast, C++) carry per two lines 2-3 v_mov of wave-uniform
// LDS addresses into VGPRs and, in the tested classes, v_cmp + v_cndmask + FMA per line for the 25 cm-1 rule
// (profiles/r03_isa_census_d11: 13-15 / 21 / 31 / 36 vector instructions per pair, of which 11-26 arithmetic).  Every C++
// formulation of "one address register, records at immediate offsets, two pairs in flight in alternating register sets, the
// per-lane conditions as EXEC masks" that was tried came back from the compiler with MORE moves (the register coalescer gives
// up on the 128-bit load tuples whose halves are updated in place: 6-8 v_mov_b64 per trip).  Here the register sets are fixed:
//
//   v[64:79]    eight double temporaries
//   v[80:95]    record set A: line 0 = v[80:87] {Xnu, HW^2 | a2, pa}, line 1 = v[88:95]
//   v[96:111]   record set B
//   v[112:119]  HotB::pb of the four lines in flight (two-resonance classes)
//
// A "class" of the run loop evaluates the lines of ONE fast class that follow each other, two lines per reciprocal
//   n0/P0 + n1/P1 = (n0 P1 + n1 P0) r,  r = 1/(P0 P1)   (v_rcp_f64 + one Newton step, lines_device.hpp frcp),
// four lines per trip: while set A is evaluated the records of the next pair arrive in set B and vice versa; one v_add_u32
// per trip advances the address.  The per-lane conditions are EXEC masks set by v_cmpx:
//   25 cm-1 rule (modm.f90:384; O2: inside the shape function, :755):   SF += t   under !(|WN - Xnu| > lim)
//   negative resonance within reach (DIFF <= 0, modm.f90:713):           e += den1, pa += pb   under WN + Xnu <= lim
// i.e. one compare + one add where the 0/1 factors needed compare, select and FMA.  fma(t, 1, SF) = SF + t: the sums are those
// of the C++ loops (kept in lines_device.hpp: CO2, two wavenumbers per lane, single precision, MONORTM_NO_UNIFIED builds).
//
// Vector instructions per pair (generic molecule / O2): one resonance untested 13 / 11, tested 16 / 16; two resonances untested
// 27 / 23, tested 30 / 28 - plus a quarter of an address update.
#pragma once

#define LA_TM0 "v[64:65]"
#define LA_TM1 "v[66:67]"
#define LA_TM2 "v[68:69]"
#define LA_TM3 "v[70:71]"
#define LA_TM4 "v[72:73]"
#define LA_TM5 "v[74:75]"
#define LA_TM6 "v[76:77]"
#define LA_TM7 "v[78:79]"
// record set A
#define LA_A_T0 "v[80:83]"
#define LA_A_U0 "v[84:87]"
#define LA_A_X0 "v[80:81]"
#define LA_A_H0 "v[82:83]"
#define LA_A_A0 "v[84:85]"
#define LA_A_P0 "v[86:87]"
#define LA_A_T1 "v[88:91]"
#define LA_A_U1 "v[92:95]"
#define LA_A_X1 "v[88:89]"
#define LA_A_H1 "v[90:91]"
#define LA_A_A1 "v[92:93]"
#define LA_A_P1 "v[94:95]"
#define LA_A_B0 "v[112:113]"
#define LA_A_B1 "v[114:115]"
// record set B
#define LA_B_T0 "v[96:99]"
#define LA_B_U0 "v[100:103]"
#define LA_B_X0 "v[96:97]"
#define LA_B_H0 "v[98:99]"
#define LA_B_A0 "v[100:101]"
#define LA_B_P0 "v[102:103]"
#define LA_B_T1 "v[104:107]"
#define LA_B_U1 "v[108:111]"
#define LA_B_X1 "v[104:105]"
#define LA_B_H1 "v[106:107]"
#define LA_B_A1 "v[108:109]"
#define LA_B_P1 "v[110:111]"
#define LA_B_B0 "v[116:117]"
#define LA_B_B1 "v[118:119]"

#define LA_CLOBBERS                                                                                                            \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",   \
        "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",     \
        "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",     \
        "v114", "v115", "v116", "v117", "v118", "v119", "scc", "memory"

#define LA_I(x) x "\n\t"
// r = 1/P: seed (TM1) and one Newton step; P in TM0 is consumed
#define LA_RCP                                                    \
    LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0)                     \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0")     \
    LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1)

// ---- one resonance: d -> X (in place), den -> H (in place), P = den0 den1 -> TM0 -----------------------------------------
#define LA_HEAD1(S)                                                                       \
    LA_I("v_add_f64 " LA_##S##_X0 ", %[wn], -" LA_##S##_X0)                               \
    LA_I("v_add_f64 " LA_##S##_X1 ", %[wn], -" LA_##S##_X1)                               \
    LA_I("v_fma_f64 " LA_##S##_H0 ", " LA_##S##_X0 ", " LA_##S##_X0 ", " LA_##S##_H0)     \
    LA_I("v_fma_f64 " LA_##S##_H1 ", " LA_##S##_X1 ", " LA_##S##_X1 ", " LA_##S##_H1)     \
    LA_I("v_mul_f64 " LA_TM0 ", " LA_##S##_H0 ", " LA_##S##_H1)
// untested: num = a2_0 den1 + a2_1 den0
#define LA_NUM1(S)                                                                        \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_##S##_H0)                      \
    LA_I("v_fma_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_##S##_H1 ", " LA_##S##_A1)
// tested: n0 P1 -> A0, n1 P0 -> A1
#define LA_TERMS1(S)                                                                      \
    LA_I("v_mul_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_##S##_H1)                      \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_##S##_H0)

// generic molecule (pedestals in P0 / P1), untested: SF += num r - (pa0 + pa1)
#define LA_FIN_U_K0(S)                                                                    \
    LA_I("v_add_f64 " LA_##S##_P0 ", " LA_##S##_P0 ", " LA_##S##_P1)                      \
    LA_I("v_fma_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM1 ", -" LA_##S##_P0)         \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A0)
// O2 (no pedestal), untested: SF += num r
#define LA_FIN_U_K1(S) LA_I("v_fma_f64 %[sf], " LA_##S##_A0 ", " LA_TM1 ", %[sf]")
// tested: t_i = n_i P_j r (- pa_i), each added under its own 25 cm-1 mask; D0 / D1: the registers that hold WN - Xnu
#define LA_FIN_T_K0(S, D0, D1)                                                            \
    LA_I("v_fma_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM1 ", -" LA_##S##_P0)         \
    LA_I("v_fma_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_TM1 ", -" LA_##S##_P1)         \
    LA_I("s_mov_b64 %[sv], exec")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D0 "|, %[c25]")                                    \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A0)                                          \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D1 "|, %[c25]")                                    \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A1)                                          \
    LA_I("s_mov_b64 exec, %[sv]")
// O2: the limit on |WN - Xnu| sits in the record's pa slot (25, or +inf for a coupled line)
#define LA_FIN_T_K1(S, D0, D1)                                                            \
    LA_I("v_mul_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM1)                           \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_TM1)                           \
    LA_I("s_mov_b64 %[sv], exec")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D0 "|, " LA_##S##_P0)                              \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A0)                                          \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D1 "|, " LA_##S##_P1)                              \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A1)                                          \
    LA_I("s_mov_b64 exec, %[sv]")

#define LA_PAIR_K0_M0_T0(S) LA_HEAD1(S) LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0) LA_NUM1(S) \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0") LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1) LA_FIN_U_K0(S)
#define LA_PAIR_K1_M0_T0(S) LA_HEAD1(S) LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0) LA_NUM1(S) \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0") LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1) LA_FIN_U_K1(S)
#define LA_PAIR_K0_M0_T1(S) LA_HEAD1(S) LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0) LA_TERMS1(S) \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0") LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1) \
    LA_FIN_T_K0(S, LA_##S##_X0, LA_##S##_X1)
#define LA_PAIR_K1_M0_T1(S) LA_HEAD1(S) LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0) LA_TERMS1(S) \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0") LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1) \
    LA_FIN_T_K1(S, LA_##S##_X0, LA_##S##_X1)

// ---- two resonances: d -> TM2 / TM3, d+ = WN + Xnu -> X, den1 -> TM4 / TM5, e = den2 -> H, P_i = den1 den2 -> TM6 / TM7 ----
#define LA_HEAD2(S)                                                                       \
    LA_I("v_add_f64 " LA_TM2 ", %[wn], -" LA_##S##_X0)                                    \
    LA_I("v_add_f64 " LA_TM3 ", %[wn], -" LA_##S##_X1)                                    \
    LA_I("v_add_f64 " LA_##S##_X0 ", %[wn], " LA_##S##_X0)                                \
    LA_I("v_add_f64 " LA_##S##_X1 ", %[wn], " LA_##S##_X1)                                \
    LA_I("v_fma_f64 " LA_TM4 ", " LA_TM2 ", " LA_TM2 ", " LA_##S##_H0)                    \
    LA_I("v_fma_f64 " LA_TM5 ", " LA_TM3 ", " LA_TM3 ", " LA_##S##_H1)                    \
    LA_I("v_fma_f64 " LA_##S##_H0 ", " LA_##S##_X0 ", " LA_##S##_X0 ", " LA_##S##_H0)     \
    LA_I("v_fma_f64 " LA_##S##_H1 ", " LA_##S##_X1 ", " LA_##S##_X1 ", " LA_##S##_H1)     \
    LA_I("v_mul_f64 " LA_TM6 ", " LA_TM4 ", " LA_##S##_H0)                                \
    LA_I("v_mul_f64 " LA_TM7 ", " LA_TM5 ", " LA_##S##_H1)                                \
    LA_I("v_mul_f64 " LA_TM0 ", " LA_TM6 ", " LA_TM7)                                     \
    LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0)
// the lanes within reach of the negative resonance: generic molecules (limit 25, second pedestal in B) / O2 (limit in B)
#define LA_M2_K0(S)                                                                       \
    LA_I("s_mov_b64 %[sv], exec")                                                         \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X0 ", %[c25]")                              \
    LA_I("v_add_f64 " LA_##S##_H0 ", " LA_##S##_H0 ", " LA_TM4)                           \
    LA_I("v_add_f64 " LA_##S##_P0 ", " LA_##S##_P0 ", " LA_##S##_B0)                      \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X1 ", %[c25]")                              \
    LA_I("v_add_f64 " LA_##S##_H1 ", " LA_##S##_H1 ", " LA_TM5)                           \
    LA_I("v_add_f64 " LA_##S##_P1 ", " LA_##S##_P1 ", " LA_##S##_B1)                      \
    LA_I("s_mov_b64 exec, %[sv]")
#define LA_M2_K1(S)                                                                       \
    LA_I("s_mov_b64 %[sv], exec")                                                         \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X0 ", " LA_##S##_B0)                        \
    LA_I("v_add_f64 " LA_##S##_H0 ", " LA_##S##_H0 ", " LA_TM4)                           \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X1 ", " LA_##S##_B1)                        \
    LA_I("v_add_f64 " LA_##S##_H1 ", " LA_##S##_H1 ", " LA_TM5)                           \
    LA_I("s_mov_b64 exec, %[sv]")
// n_i = a2_i (den2_i + [m2] den1_i) -> A, then the Newton step of the reciprocal
#define LA_N2(S)                                                                          \
    LA_I("v_mul_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_##S##_H0)                      \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_##S##_H1)                      \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0")                            \
    LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1)
#define LA_NUM2(S)                                                                        \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_TM6)                           \
    LA_I("v_fma_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM7 ", " LA_##S##_A1)
#define LA_TERMS2(S)                                                                      \
    LA_I("v_mul_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM7)                           \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_TM6)

#define LA_PAIR_K0_M1_T0(S) LA_HEAD2(S) LA_M2_K0(S) LA_N2(S) LA_NUM2(S) LA_FIN_U_K0(S)
#define LA_PAIR_K1_M1_T0(S) LA_HEAD2(S) LA_M2_K1(S) LA_N2(S) LA_NUM2(S) LA_FIN_U_K1(S)
#define LA_PAIR_K0_M1_T1(S) LA_HEAD2(S) LA_M2_K0(S) LA_N2(S) LA_TERMS2(S) LA_FIN_T_K0(S, LA_TM2, LA_TM3)
// (O2: LA_M2_K1 leaves the test limits in P0 / P1 untouched)
#define LA_PAIR_K1_M1_T1(S) LA_HEAD2(S) LA_M2_K1(S) LA_N2(S) LA_TERMS2(S) LA_FIN_T_K1(S, LA_TM2, LA_TM3)

// ---- LDS reads of one pair into set S at byte offset OFF (a literal) from %[addr]; pb of the pair for the two-resonance
// classes from the immediate operands OB0 / OB1 ------------------------------------------------------------------------------
#define LA_LOAD(S, o0, o1, o2, o3)                                                        \
    LA_I("ds_read_b128 " LA_##S##_T0 ", %[addr] offset:" #o0)                             \
    LA_I("ds_read_b128 " LA_##S##_U0 ", %[addr] offset:" #o1)                             \
    LA_I("ds_read_b128 " LA_##S##_T1 ", %[addr] offset:" #o2)                             \
    LA_I("ds_read_b128 " LA_##S##_U1 ", %[addr] offset:" #o3)
#define LA_LOADB(S, OB0, OB1)                                                             \
    LA_I("ds_read_b64 " LA_##S##_B0 ", %[addr] offset:%[" #OB0 "]")                       \
    LA_I("ds_read_b64 " LA_##S##_B1 ", %[addr] offset:%[" #OB1 "]")

// leave for LBL unless the pair at the bits MASK of T / M is of this class.  s_and_b64 sets SCC = (result != 0).
#define LA_CLS_CHECK(MASK, BRANCH_T, BRANCH_M, LBL)                                       \
    LA_I("s_and_b64 %[tmp], %[T], " #MASK)                                                \
    LA_I(BRANCH_T " " LBL)                                                                \
    LA_I("s_and_b64 %[tmp], %[M], " #MASK)                                                \
    LA_I(BRANCH_M " " LBL)

// One class.  Entered from the dispatcher (label 90) with the precondition: the pair at addr (bits 0, 1 of T / M) is of this
// class and its records are in flight into, or present in, set A.  While the NEXT pair is of the class too: four lines per
// trip - the next pair's records are read into set B while A is evaluated, the pair after that into A while B is.  NB = LDS
// reads per pair (4, or 6 with pb): `s_waitcnt lgkmcnt(NB)` after the reads of the following pair have been issued = the
// current pair has arrived (LDS returns in order).  Then the current pair by itself if it is of the class (it is unless a
// trip ran), with the read-ahead of its successor, and back to the dispatcher.
// L0 / L1 / L2: the labels of this class (literals): entry, trip, single pair.
#define LA_CLASS(L0, L1, L2, PAIR, LOADB_A0, LOADB_B, LOADB_A2, NB, BR_T, BR_M)           \
    L0 ":\n\t"                                                                            \
    LOADB_A0                                                                              \
    L1 ":\n\t"                                                                            \
    LA_I("s_cmp_lt_i32 %[n], 4")                                                          \
    LA_I("s_cbranch_scc1 " L2 "f")                                                        \
    LA_CLS_CHECK(12, BR_T, BR_M, L2 "f")                                                  \
    LA_LOAD(B, 64, 80, 96, 112)                                                           \
    LOADB_B                                                                               \
    LA_I("s_waitcnt lgkmcnt(" #NB ")")                                                    \
    PAIR(A)                                                                               \
    LA_LOAD(A, 128, 144, 160, 176)                                                        \
    LOADB_A2                                                                              \
    LA_I("s_waitcnt lgkmcnt(" #NB ")")                                                    \
    PAIR(B)                                                                               \
    LA_I("v_add_u32_e32 %[addr], 0x80, %[addr]")                                          \
    LA_I("s_sub_i32 %[n], %[n], 4")                                                       \
    LA_I("s_lshr_b64 %[T], %[T], 4")                                                      \
    LA_I("s_lshr_b64 %[M], %[M], 4")                                                      \
    LA_I("s_branch " L1 "b")                                                              \
    L2 ":\n\t"                                                                            \
    LA_I("s_cmp_lt_i32 %[n], 2")                                                          \
    LA_I("s_cbranch_scc1 99f")                                                            \
    LA_CLS_CHECK(3, BR_T, BR_M, "90b")                                                    \
    LA_I("s_waitcnt lgkmcnt(0)")                                                          \
    PAIR(A)                                                                               \
    LA_LOAD(A, 64, 80, 96, 112)                                                           \
    LA_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    LA_I("s_sub_i32 %[n], %[n], 2")                                                       \
    LA_I("s_lshr_b64 %[T], %[T], 2")                                                      \
    LA_I("s_lshr_b64 %[M], %[M], 2")                                                      \
    LA_I("s_branch 90b")

#define LA_PB_A0 LA_LOADB(A, ob0, ob1)
#define LA_PB_B LA_LOADB(B, ob2, ob3)
#define LA_PB_A2 LA_LOADB(A, ob4, ob5)
// The run: dispatcher + the four classes.  SCC = 1 after s_and <=> the bit is set (tested / two resonances).
#define LA_RUN(K)                                                                                              \
    LA_I("s_waitcnt lgkmcnt(0)")                                                                               \
    LA_LOAD(A, 0, 16, 32, 48)                                                                                  \
    "90:\n\t"                                                                                                  \
    LA_I("s_cmp_lt_i32 %[n], 2")                                                                               \
    LA_I("s_cbranch_scc1 99f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[M], 3")                                                                          \
    LA_I("s_cbranch_scc1 91f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 20f")                                                                                 \
    LA_CLASS("10", "11", "12", LA_PAIR_##K##_M0_T0, "", "", "", 4, "s_cbranch_scc1", "s_cbranch_scc1")               \
    LA_CLASS("20", "21", "22", LA_PAIR_##K##_M0_T1, "", "", "", 4, "s_cbranch_scc0", "s_cbranch_scc1")               \
    "91:\n\t"                                                                                                  \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 40f")                                                                                 \
    LA_CLASS("30", "31", "32", LA_PAIR_##K##_M1_T0, LA_PB_A0, LA_PB_B, LA_PB_A2, 6, "s_cbranch_scc1", "s_cbranch_scc0") \
    LA_CLASS("40", "41", "42", LA_PAIR_##K##_M1_T1, LA_PB_A0, LA_PB_B, LA_PB_A2, 6, "s_cbranch_scc0", "s_cbranch_scc0") \
    "99:\n\t"                                                                                                  \
    "s_waitcnt lgkmcnt(0)"

namespace {

// KIND 0 generic molecule, 1 O2.  BOFF: byte distance from a line's HotA record to its HotB::pb (one LDS object).
// addr: LDS byte address of the current line's HotA record; n: lines left in the run; T / M: "tested" / "two resonances"
// masks, bit 0 = current line.  Evaluates the lines of the run two at a time - a pair takes the class of the more general of
// its two lines - and leaves n = 0 or 1 with addr, T, M advanced to the odd last line.
template <int KIND, unsigned BOFF>
__device__ __forceinline__ void asm_run(unsigned &addr, int &n, unsigned long long &T, unsigned long long &M, double WN, double &SF) {
    unsigned long long sv, cm, tmp;
    const double c25 = 25.;
#define LA_OPERANDS                                                                                                            \
    : [sf] "+v"(SF), [addr] "+v"(addr), [n] "+s"(n), [T] "+s"(T), [M] "+s"(M), [sv] "=&s"(sv), [cm] "=&s"(cm), [tmp] "=&s"(tmp) \
    : [wn] "v"(WN), [c25] "s"(c25), [ob0] "i"(BOFF), [ob1] "i"(BOFF + 32u), [ob2] "i"(BOFF + 64u), [ob3] "i"(BOFF + 96u),      \
      [ob4] "i"(BOFF + 128u), [ob5] "i"(BOFF + 160u)                                                                           \
    : LA_CLOBBERS
    if constexpr (KIND == 0) asm volatile(LA_RUN(K0) LA_OPERANDS);
    else asm volatile(LA_RUN(K1) LA_OPERANDS);
#undef LA_OPERANDS
}

}  // namespace
