// lines_asm.hpp - the inner loops of the line sum for one wavenumber per lane in double precision, written in gfx950
// assembly (round 4).  Reference arithmetic: src/modm.f90:706-831 (LSF_LORTZ), regrouped as in lines_device.hpp.
//
// Why assembly.  The class loops of rounds 1-3 (eval_pair / eval_pair_fast, C++) carry per two lines 2-3 v_mov of wave-uniform
// LDS addresses into VGPRs and, in the tested classes, v_cmp + v_cndmask + FMA per line for the 25 cm-1 rule
// (profiles/r03_isa_census_d11: 13-15 / 21 / 31 / 36 vector instructions per pair, of which 11-26 arithmetic).  Every C++
// formulation of "one address register, records at immediate offsets, two pairs in flight in alternating register sets, the
// per-lane conditions as EXEC masks" that was tried came back from the compiler with MORE moves (the register coalescer gives
// up on the 128-bit load tuples whose halves are updated in place: 6-8 v_mov_b64 per trip).  Here the registers are fixed -
// 40 of them (the first version pinned 56 and pushed the prepare stage of the kernel into scratch: 89 scratch accesses per
// wave, VALU busy 0.82 -> 0.75):
//
//   v[64:67]    TM0, TM1: product of the denominators / its reciprocal - and, before they are formed, HotB::pb of the two
//               lines of a two-resonance pair
//   v[68:79]    TM2 .. TM7: temporaries of the two-resonance classes; v[72:79] = second halves {a2, pa} of record set B in the
//               one-resonance classes (which use no temporaries beyond TM0, TM1)
//   v[80:95]    record set A: line 0 = v[80:83] {Xnu, HW^2} + v[84:87] {a2, pa}, line 1 = v[88:95]
//   v[96:103]   first halves {Xnu, HW^2} of record set B
//
// A "class" of the run loop evaluates the lines of ONE fast class that follow each other, two lines per reciprocal
//   n0/P0 + n1/P1 = (n0 P1 + n1 P0) r,  r = 1/(P0 P1)   (v_rcp_f64 + one Newton step, lines_device.hpp frcp),
// four lines per trip, one v_add_u32 per trip for the address.  One-resonance classes: while set A is evaluated the records
// of the next pair arrive in set B and vice versa.  Two-resonance classes: the first halves alternate between A and B, the
// second halves and pb - needed a dozen instructions into a pair - are read into the same registers as soon as the pair
// before has used them.  The per-lane conditions of O2 and CO2 are EXEC masks set by v_cmpx:
//   25 cm-1 rule (modm.f90:384; O2: inside the shape function, :755):   SF += t   under !(|WN - Xnu| > lim)
//   negative resonance within reach (DIFF <= 0, modm.f90:713):           e += den1   under WN + Xnu <= lim
// i.e. one compare + one add where the 0/1 factors needed compare, select and FMA.  fma(t, 1, SF) = SF + t: the sums are those
// of the C++ loops (kept in lines_device.hpp: two wavenumbers per lane, single precision, MONORTM_NO_UNIFIED builds).
// Generic molecules need no mask at all: a bracket a2 / den - pedestal is >= 0 exactly where its rule admits it, so the CLAMP
// modifier of the FMA that forms it is the test (LA_FIN_T_K0, LA_PAIR_K0_M1) - and one two-resonance loop serves tested and
// untested lines alike.
//
// Vector instructions per pair (generic molecule / O2): one resonance untested 13 / 11, tested 14 / 16; two resonances 26 (tested
// or not) / untested 23, tested 28 - plus a quarter of an address update; per trip of four lines 36 / 38 / 67 instructions of
// every kind for the three classes of a generic molecule (LDS reads, waits and the three scalar instructions of the trip count
// included).  Hazards the assembler does not see inside an asm block
// (gfx940+): the result of a transcendental (v_rcp_f64) is not read by the next instruction; v_cmpx writes EXEC for ordinary
// VALU instructions only (no DPP / lane access follows).
#pragma once

#define LA_TM0 "v[64:65]"
#define LA_TM1 "v[66:67]"
#define LA_TM2 "v[68:69]"
#define LA_TM3 "v[70:71]"
#define LA_TM4 "v[72:73]"
#define LA_TM5 "v[74:75]"
#define LA_TM6 "v[76:77]"
#define LA_TM7 "v[78:79]"
// HotB::pb of the pair (two-resonance classes): lives in TM0 / TM1 until the product of the denominators is formed
#define LA_PB0 "v[64:65]"
#define LA_PB1 "v[66:67]"
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
// record set B: first halves of its own, second halves in TM4 .. TM7 (one-resonance classes only)
#define LA_B_T0 "v[96:99]"
#define LA_B_X0 "v[96:97]"
#define LA_B_H0 "v[98:99]"
#define LA_B_T1 "v[100:103]"
#define LA_B_X1 "v[100:101]"
#define LA_B_H1 "v[102:103]"
#define LA_B_U0 "v[72:75]"
#define LA_B_A0 "v[72:73]"
#define LA_B_P0 "v[74:75]"
#define LA_B_U1 "v[76:79]"
#define LA_B_A1 "v[76:77]"
#define LA_B_P1 "v[78:79]"

#define LA_CLOBBERS                                                                                                            \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",   \
        "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",     \
        "v99", "v100", "v101", "v102", "v103", "scc", "memory"

#define LA_I(x) x "\n\t"
#define LA_RCP_SEED LA_I("v_rcp_f64_e32 " LA_TM1 ", " LA_TM0)
#define LA_NEWTON                                                 \
    LA_I("v_fma_f64 " LA_TM0 ", -" LA_TM0 ", " LA_TM1 ", 1.0")     \
    LA_I("v_fma_f64 " LA_TM1 ", " LA_TM0 ", " LA_TM1 ", " LA_TM1)

// ================= one resonance: set S holds the pair =========================================================================
// d -> X (in place), den -> H (in place), P = den0 den1 -> TM0
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

// ---- the end of a pair; U = the set that holds {a2, pa} (S itself for one resonance, always A for two) ------------------------
// generic molecule (pedestals in P0 / P1), untested: SF += num r - (pa0 + pa1)
#define LA_FIN_U_K0(U)                                                                    \
    LA_I("v_add_f64 " LA_##U##_P0 ", " LA_##U##_P0 ", " LA_##U##_P1)                      \
    LA_I("v_fma_f64 " LA_##U##_A0 ", " LA_##U##_A0 ", " LA_TM1 ", -" LA_##U##_P0)         \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##U##_A0)
// O2 (no pedestal), untested: SF += num r
#define LA_FIN_U_K1(U) LA_I("v_fma_f64 %[sf], " LA_##U##_A0 ", " LA_TM1 ", %[sf]")
// tested, generic molecule: t_i = n_i P_j r - pa_i = a2 / den - a2 / (625 + HW^2) is >= 0 exactly where |WN - Xnu| <= 25, and the rule
// of modm.f90:384 drops the line where it is negative - so the test is the CLAMP modifier of the FMA that forms the term (VOP3
// clamp: result to [0, 1]; the upper bound is out of reach, a2 / den <= S~ / (pi HW) is below 1e-12 for any line).  No compare,
// no EXEC mask.  At |WN - Xnu| = 25 to the last bit the term is a rounding residue of ~1e-16 of the line's peak either way.
// D0 / D1 are unused.
#define LA_FIN_T_K0(U, D0, D1)                                                            \
    LA_I("v_fma_f64 " LA_##U##_A0 ", " LA_##U##_A0 ", " LA_TM1 ", -" LA_##U##_P0 " clamp") \
    LA_I("v_fma_f64 " LA_##U##_A1 ", " LA_##U##_A1 ", " LA_TM1 ", -" LA_##U##_P1 " clamp") \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##U##_A0)                                          \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##U##_A1)
// O2: the limit on |WN - Xnu| sits in the record's pa slot (25, or +inf for a coupled line)
#define LA_FIN_T_K1(U, D0, D1)                                                            \
    LA_I("v_mul_f64 " LA_##U##_A0 ", " LA_##U##_A0 ", " LA_TM1)                           \
    LA_I("v_mul_f64 " LA_##U##_A1 ", " LA_##U##_A1 ", " LA_TM1)                           \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D0 "|, " LA_##U##_P0)                              \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##U##_A0)                                          \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" D1 "|, " LA_##U##_P1)                              \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##U##_A1)                                          \
    LA_I("s_mov_b64 exec, %[sv]")

#define LA_PAIR_K0_M0_T0(S) LA_HEAD1(S) LA_RCP_SEED LA_NUM1(S) LA_NEWTON LA_FIN_U_K0(S)
#define LA_PAIR_K1_M0_T0(S) LA_HEAD1(S) LA_RCP_SEED LA_NUM1(S) LA_NEWTON LA_FIN_U_K1(S)
#define LA_PAIR_K0_M0_T1(S) LA_HEAD1(S) LA_RCP_SEED LA_TERMS1(S) LA_NEWTON LA_FIN_T_K0(S, LA_##S##_X0, LA_##S##_X1)
#define LA_PAIR_K1_M0_T1(S) LA_HEAD1(S) LA_RCP_SEED LA_TERMS1(S) LA_NEWTON LA_FIN_T_K1(S, LA_##S##_X0, LA_##S##_X1)

// ---- CO2 (KIND 2): one resonance, pedestal x (2 - d^2 / 625) (modm.f90:808-817): t_i = a2_i den_j r - pa_i f_i,
// f_i = 2 - d_i^2 / 625.  d^2 -> TM2 / TM3 (free in the one-resonance classes), then f in place.  20 / 22 instructions per pair.
#define LA_HEAD1_K2(S)                                                                    \
    LA_I("v_add_f64 " LA_##S##_X0 ", %[wn], -" LA_##S##_X0)                               \
    LA_I("v_add_f64 " LA_##S##_X1 ", %[wn], -" LA_##S##_X1)                               \
    LA_I("v_mul_f64 " LA_TM2 ", " LA_##S##_X0 ", " LA_##S##_X0)                           \
    LA_I("v_mul_f64 " LA_TM3 ", " LA_##S##_X1 ", " LA_##S##_X1)                           \
    LA_I("v_fma_f64 " LA_##S##_H0 ", " LA_##S##_X0 ", " LA_##S##_X0 ", " LA_##S##_H0)     \
    LA_I("v_fma_f64 " LA_##S##_H1 ", " LA_##S##_X1 ", " LA_##S##_X1 ", " LA_##S##_H1)     \
    LA_I("v_mul_f64 " LA_TM0 ", " LA_##S##_H0 ", " LA_##S##_H1)                           \
    LA_RCP_SEED                                                                           \
    LA_TERMS1(S)                                                                          \
    LA_I("v_fma_f64 " LA_TM2 ", -" LA_TM2 ", %[c625], 2.0")                               \
    LA_I("v_fma_f64 " LA_TM3 ", -" LA_TM3 ", %[c625], 2.0")                               \
    LA_NEWTON                                                                             \
    LA_I("v_mul_f64 " LA_##S##_A0 ", " LA_##S##_A0 ", " LA_TM1)                           \
    LA_I("v_mul_f64 " LA_##S##_A1 ", " LA_##S##_A1 ", " LA_TM1)                           \
    LA_I("v_fma_f64 " LA_##S##_A0 ", -" LA_##S##_P0 ", " LA_TM2 ", " LA_##S##_A0)         \
    LA_I("v_fma_f64 " LA_##S##_A1 ", -" LA_##S##_P1 ", " LA_TM3 ", " LA_##S##_A1)
#define LA_PAIR_K2_M0_T0(S) LA_HEAD1_K2(S)                                                \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A0)                                          \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A1)
#define LA_PAIR_K2_M0_T1(S) LA_HEAD1_K2(S)                                                \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" LA_##S##_X0 "|, %[c25]")                           \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A0)                                          \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_ngt_f64_e64 %[cm], |" LA_##S##_X1 "|, %[c25]")                           \
    LA_I("v_add_f64 %[sf], %[sf], " LA_##S##_A1)                                          \
    LA_I("s_mov_b64 exec, %[sv]")

// ================= two resonances: set S holds {Xnu, HW^2}, set A {a2, pa}, PB0 / PB1 the second pedestals / limits ============
// d -> TM2 / TM3, d+ = WN + Xnu -> X, den1 -> TM4 / TM5, e = den2 -> H, P_i = den1 den2 -> TM6 / TM7
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
    LA_I("v_mul_f64 " LA_TM7 ", " LA_TM5 ", " LA_##S##_H1)
// the lanes within reach of the negative resonance: generic molecules (limit 25, second pedestal in PB) / O2 (limit in PB);
// then - PB is dead - the product of the four denominators and the seed of its reciprocal
#define LA_M2_K1(S)                                                                       \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X0 ", " LA_PB0)                             \
    LA_I("v_add_f64 " LA_##S##_H0 ", " LA_##S##_H0 ", " LA_TM4)                           \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_cmpx_le_f64_e64 %[cm], " LA_##S##_X1 ", " LA_PB1)                             \
    LA_I("v_add_f64 " LA_##S##_H1 ", " LA_##S##_H1 ", " LA_TM5)                           \
    LA_I("s_mov_b64 exec, %[sv]")                                                         \
    LA_I("v_mul_f64 " LA_TM0 ", " LA_TM6 ", " LA_TM7)                                     \
    LA_RCP_SEED
// n_i = a2_i (den2_i + [m2] den1_i) -> A0 / A1 of set A, then the Newton step of the reciprocal
#define LA_N2(S)                                                                          \
    LA_I("v_mul_f64 " LA_A_A0 ", " LA_A_A0 ", " LA_##S##_H0)                              \
    LA_I("v_mul_f64 " LA_A_A1 ", " LA_A_A1 ", " LA_##S##_H1)                              \
    LA_NEWTON
#define LA_NUM2                                                                           \
    LA_I("v_mul_f64 " LA_A_A1 ", " LA_A_A1 ", " LA_TM6)                                   \
    LA_I("v_fma_f64 " LA_A_A0 ", " LA_A_A0 ", " LA_TM7 ", " LA_A_A1)
#define LA_TERMS2                                                                         \
    LA_I("v_mul_f64 " LA_A_A0 ", " LA_A_A0 ", " LA_TM7)                                   \
    LA_I("v_mul_f64 " LA_A_A1 ", " LA_A_A1 ", " LA_TM6)

// The reads a two-resonance pair depends on are always the oldest outstanding ones, in the order {first halves (2)}, {second
// halves (2)}, {pb (2)}, followed by the two first-half reads of the pair after it (LA_CLASS2; at the single pair of a class
// everything has arrived): lgkmcnt(6) = the first halves are here, lgkmcnt(2) = second halves and pb too.
#define LA_WAIT_T LA_I("s_waitcnt lgkmcnt(6)")
#define LA_WAIT_U LA_I("s_waitcnt lgkmcnt(2)")
// generic molecule, two resonances - tested or not, negative resonance within reach of a lane or not: with ONE reciprocal
// r = 1 / (P0 P1), P_i = den1_i den2_i, the two brackets of a line are
//   a2 / den1 - pa = (a2 P_j r) den2 - pa,     a2 / den2 - pb = (a2 P_j r) den1 - pb,     pb = pa (fast-class lines carry no Y factors)
// and each is >= 0 exactly where its rule admits it (|WN - Xnu| <= 25: modm.f90:384; WN + Xnu <= 25: :713) - the clamp of the
// FMA is both tests.  26 vector instructions per pair, no scalar ones, no read of HotB.
#define LA_PAIR_K0_M1(S)                                                                  \
    LA_I("s_waitcnt lgkmcnt(4)")                                                          \
    LA_HEAD2(S)                                                                           \
    LA_I("v_mul_f64 " LA_TM0 ", " LA_TM6 ", " LA_TM7)                                     \
    LA_RCP_SEED                                                                           \
    LA_I("s_waitcnt lgkmcnt(2)")                                                          \
    LA_I("v_mul_f64 " LA_A_A0 ", " LA_A_A0 ", " LA_TM7)                                   \
    LA_I("v_mul_f64 " LA_A_A1 ", " LA_A_A1 ", " LA_TM6)                                   \
    LA_NEWTON                                                                             \
    LA_I("v_mul_f64 " LA_A_A0 ", " LA_A_A0 ", " LA_TM1)                                   \
    LA_I("v_mul_f64 " LA_A_A1 ", " LA_A_A1 ", " LA_TM1)                                   \
    LA_I("v_fma_f64 " LA_TM2 ", " LA_A_A0 ", " LA_##S##_H0 ", -" LA_A_P0 " clamp")        \
    LA_I("v_fma_f64 " LA_TM4 ", " LA_A_A0 ", " LA_TM4 ", -" LA_A_P0 " clamp")             \
    LA_I("v_fma_f64 " LA_TM3 ", " LA_A_A1 ", " LA_##S##_H1 ", -" LA_A_P1 " clamp")        \
    LA_I("v_fma_f64 " LA_TM5 ", " LA_A_A1 ", " LA_TM5 ", -" LA_A_P1 " clamp")             \
    LA_I("v_add_f64 %[sf], %[sf], " LA_TM2)                                               \
    LA_I("v_add_f64 %[sf], %[sf], " LA_TM4)                                               \
    LA_I("v_add_f64 %[sf], %[sf], " LA_TM3)                                               \
    LA_I("v_add_f64 %[sf], %[sf], " LA_TM5)
#define LA_PAIR_K1_M1_T0(S) LA_WAIT_T LA_HEAD2(S) LA_WAIT_U LA_M2_K1(S) LA_N2(S) LA_NUM2 LA_FIN_U_K1(A)
// (O2: LA_M2_K1 leaves the test limits in P0 / P1 of set A untouched)
#define LA_PAIR_K1_M1_T1(S) LA_WAIT_T LA_HEAD2(S) LA_WAIT_U LA_M2_K1(S) LA_N2(S) LA_TERMS2 LA_FIN_T_K1(A, LA_TM2, LA_TM3)

// ================= LDS reads at literal byte offsets from %[addr] ================================================================
// a whole pair into set S: {Xnu, HW^2} and {a2, pa} of both lines
// (first halves first: a two-resonance class may start on them while the second halves are still on their way)
#define LA_LOAD(S, o0, o1, o2, o3)                                                        \
    LA_I("ds_read_b128 " LA_##S##_T0 ", %[addr] offset:" #o0)                             \
    LA_I("ds_read_b128 " LA_##S##_T1 ", %[addr] offset:" #o2)                             \
    LA_I("ds_read_b128 " LA_##S##_U0 ", %[addr] offset:" #o1)                             \
    LA_I("ds_read_b128 " LA_##S##_U1 ", %[addr] offset:" #o3)
// first halves of a pair into set S / second halves into set A
#define LA_LOAD_T(S, o0, o2)                                                              \
    LA_I("ds_read_b128 " LA_##S##_T0 ", %[addr] offset:" #o0)                             \
    LA_I("ds_read_b128 " LA_##S##_T1 ", %[addr] offset:" #o2)
#define LA_LOAD_U(o1, o3)                                                                 \
    LA_I("ds_read_b128 " LA_A_U0 ", %[addr] offset:" #o1)                                 \
    LA_I("ds_read_b128 " LA_A_U1 ", %[addr] offset:" #o3)
// HotB::pb of a pair (immediate operands: the distance between the two LDS arrays is a template parameter)
#define LA_LOAD_PB(OB0, OB1)                                                              \
    LA_I("ds_read_b64 " LA_PB0 ", %[addr] offset:%[" #OB0 "]")                            \
    LA_I("ds_read_b64 " LA_PB1 ", %[addr] offset:%[" #OB1 "]")

// ---- run control -------------------------------------------------------------------------------------------------------------
// Per wave every instruction costs about the same wall time whatever unit executes it (four waves per SIMD take turns: ~11
// cycles per instruction in these loops, tools/loop_rate.hip), so the loop control is part of the bill.  The first version
// re-tested the class of the NEXT pair in every trip (s_cmp / 2 s_and / 3 s_cbranch + s_sub and two 64-bit shifts to advance:
// 10 scalar instructions per four lines).  Now a class entry finds the length of its run ONCE - the pair-level masks
// p(Z) = Z | Z >> 1 on the even bits, s_ff1 on the bits that break the class - advances n, T, M by the whole run, and the trip
// is counted down: decrement, compare, branch.
//   x   <- the pairs (even bits) that are NOT of this class, from tmp = p(T) and x = p(M)
#define LA_PT LA_I("s_lshr_b64 %[tmp], %[T], 1") LA_I("s_or_b64 %[tmp], %[tmp], %[T]")
#define LA_PM LA_I("s_lshr_b64 %[x], %[M], 1") LA_I("s_or_b64 %[x], %[x], %[M]")
#define LA_V_T0_M0 LA_PT LA_PM LA_I("s_or_b64 %[x], %[x], %[tmp]")
#define LA_V_T1_M0 LA_PT LA_PM LA_I("s_orn2_b64 %[x], %[x], %[tmp]")
#define LA_V_T0_M1 LA_PT LA_PM LA_I("s_orn2_b64 %[x], %[tmp], %[x]")
#define LA_V_T1_M1 LA_PT LA_PM LA_I("s_nand_b64 %[x], %[tmp], %[x]")
// generic molecules: two resonances, tested or not (one loop serves both: LA_PAIR_K0_M1)
#define LA_V_M1 LA_PM LA_I("s_not_b64 %[x], %[x]")
// CO2: no second mask
#define LA_V_T0 LA_PT LA_I("s_mov_b64 %[x], %[tmp]")
#define LA_V_T1 LA_PT LA_I("s_not_b64 %[x], %[tmp]")
// k <- pairs of the run (>= 1: the dispatcher saw the current pair), n / T / M advanced past it, k2 <- trips of two pairs.
// (s_ff1 = -1 without a set bit: as an unsigned number it loses against n / 2.  A shift by 64 is a shift by 0, but then n = 0.)
#define LA_RUNLEN                                                                         \
    LA_I("s_and_b64 %[x], %[x], %[c55]")                                                  \
    LA_I("s_ff1_i32_b64 %[k], %[x]")                                                      \
    LA_I("s_lshr_b32 %[k], %[k], 1")                                                      \
    LA_I("s_lshr_b32 %[k2], %[n], 1")                                                     \
    LA_I("s_min_u32 %[k], %[k], %[k2]")                                                   \
    LA_I("s_lshl_b32 %[k2], %[k], 1")                                                     \
    LA_I("s_sub_i32 %[n], %[n], %[k2]")                                                   \
    LA_I("s_lshr_b64 %[T], %[T], %[k2]")                                                  \
    LA_I("s_lshr_b64 %[M], %[M], %[k2]")                                                  \
    LA_I("s_lshr_b32 %[k2], %[k], 1")                                                     \
    LA_I("s_cmp_eq_u32 %[k2], 0")
#define LA_TRIP_END(L1)                                                                   \
    LA_I("v_add_u32_e32 %[addr], 0x80, %[addr]")                                          \
    LA_I("s_add_i32 %[k2], %[k2], -1")                                                    \
    LA_I("s_cmp_lg_u32 %[k2], 0")                                                         \
    LA_I("s_cbranch_scc1 " L1 "b")

// Invariant at the dispatcher (label 90) and at every class entry: the records of the pair at addr are in flight into, or
// present in, set A.  LDS returns in order, so `s_waitcnt lgkmcnt(k)` = everything but the last k reads has arrived.
//
// One-resonance class.  Trips of four lines - the next pair is read into set B while A is evaluated, the pair after that into A
// while B is - then the odd pair of the run by itself, with the read-ahead of its successor, and back to the dispatcher.
// L0 / L1 / L2: entry, trip, odd pair.  VMASK: the LA_V_* of the class.
#define LA_CLASS1(L0, L1, L2, PAIR, VMASK)                                                \
    L0 ":\n\t"                                                                            \
    VMASK                                                                                 \
    LA_RUNLEN                                                                             \
    LA_I("s_cbranch_scc1 " L2 "f")                                                        \
    L1 ":\n\t"                                                                            \
    LA_LOAD(B, 64, 80, 96, 112)                                                           \
    LA_I("s_waitcnt lgkmcnt(4)")                                                          \
    PAIR(A)                                                                               \
    LA_LOAD(A, 128, 144, 160, 176)                                                        \
    LA_I("s_waitcnt lgkmcnt(4)")                                                          \
    PAIR(B)                                                                               \
    LA_TRIP_END(L1)                                                                       \
    L2 ":\n\t"                                                                            \
    LA_I("s_bitcmp1_b32 %[k], 0")                                                         \
    LA_I("s_cbranch_scc0 90b")                                                            \
    LA_I("s_waitcnt lgkmcnt(0)")                                                          \
    PAIR(A)                                                                               \
    LA_LOAD(A, 64, 80, 96, 112)                                                           \
    LA_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    LA_I("s_branch 90b")

// Two-resonance class.  pb of the current pair is read at the entry.  Per trip: the first halves of the next pair into set B,
// the current pair from (A, A, PB); its second halves and pb are then re-read for the next pair, the first halves of the pair
// after that go to set A, and the next pair is evaluated from (B, A, PB); second halves and pb for the pair after.
#define LA_CLASS2(L0, L1, L2, PAIR, VMASK)                                                \
    L0 ":\n\t"                                                                            \
    LA_LOAD_PB(ob0, ob1)                                                                  \
    VMASK                                                                                 \
    LA_RUNLEN                                                                             \
    LA_I("s_cbranch_scc1 " L2 "f")                                                        \
    L1 ":\n\t"                                                                            \
    LA_LOAD_T(B, 64, 96)                                                                  \
    PAIR(A)                                                                               \
    LA_LOAD_U(80, 112)                                                                    \
    LA_LOAD_PB(ob2, ob3)                                                                  \
    LA_LOAD_T(A, 128, 160)                                                                \
    PAIR(B)                                                                               \
    LA_LOAD_U(144, 176)                                                                   \
    LA_LOAD_PB(ob4, ob5)                                                                  \
    LA_TRIP_END(L1)                                                                       \
    L2 ":\n\t"                                                                            \
    LA_I("s_bitcmp1_b32 %[k], 0")                                                         \
    LA_I("s_cbranch_scc0 90b")                                                            \
    LA_I("s_waitcnt lgkmcnt(0)")                                                          \
    PAIR(A)                                                                               \
    LA_LOAD(A, 64, 80, 96, 112)                                                           \
    LA_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    LA_I("s_branch 90b")

// The same without HotB (generic molecules: pb = pa, and the clamps need no limits): the waits sit in LA_PAIR_K0_M1 - the reads
// a pair depends on are the oldest outstanding ones, {first halves (2)}, {second halves (2)}, followed by the two first-half
// reads of the pair after it: lgkmcnt(4) = first halves here, lgkmcnt(2) = second halves too.
#define LA_CLASS2N(L0, L1, L2, PAIR, VMASK)                                               \
    L0 ":\n\t"                                                                            \
    VMASK                                                                                 \
    LA_RUNLEN                                                                             \
    LA_I("s_cbranch_scc1 " L2 "f")                                                        \
    L1 ":\n\t"                                                                            \
    LA_LOAD_T(B, 64, 96)                                                                  \
    PAIR(A)                                                                               \
    LA_LOAD_U(80, 112)                                                                    \
    LA_LOAD_T(A, 128, 160)                                                                \
    PAIR(B)                                                                               \
    LA_LOAD_U(144, 176)                                                                   \
    LA_TRIP_END(L1)                                                                       \
    L2 ":\n\t"                                                                            \
    LA_I("s_bitcmp1_b32 %[k], 0")                                                         \
    LA_I("s_cbranch_scc0 90b")                                                            \
    LA_I("s_waitcnt lgkmcnt(0)")                                                          \
    PAIR(A)                                                                               \
    LA_LOAD(A, 64, 80, 96, 112)                                                           \
    LA_I("v_add_u32_e32 %[addr], 64, %[addr]")                                            \
    LA_I("s_branch 90b")

// Generic molecules: three classes - one resonance untested / tested (clamp), two resonances (clamps, tested or not).
#define LA_RUN_K0                                                                                              \
    LA_I("s_waitcnt lgkmcnt(0)")                                                                               \
    LA_LOAD(A, 0, 16, 32, 48)                                                                                  \
    "90:\n\t"                                                                                                  \
    LA_I("s_cmp_lt_i32 %[n], 2")                                                                               \
    LA_I("s_cbranch_scc1 99f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[M], 3")                                                                          \
    LA_I("s_cbranch_scc1 30f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 20f")                                                                                 \
    LA_CLASS1("10", "11", "12", LA_PAIR_K0_M0_T0, LA_V_T0_M0)                                                  \
    LA_CLASS1("20", "21", "22", LA_PAIR_K0_M0_T1, LA_V_T1_M0)                                                  \
    LA_CLASS2N("30", "31", "32", LA_PAIR_K0_M1, LA_V_M1)                                                       \
    "99:\n\t"                                                                                                  \
    "s_waitcnt lgkmcnt(0)"

// The run: dispatcher + the four classes (O2).  SCC = 1 after s_and <=> a bit of the current pair is set (tested / two resonances).
// sv = the wave's EXEC on entry: every masked add restores it.
#define LA_RUN(K)                                                                                              \
    LA_I("s_waitcnt lgkmcnt(0)")                                                                               \
    LA_LOAD(A, 0, 16, 32, 48)                                                                                  \
    LA_I("s_mov_b64 %[sv], exec")                                                                              \
    "90:\n\t"                                                                                                  \
    LA_I("s_cmp_lt_i32 %[n], 2")                                                                               \
    LA_I("s_cbranch_scc1 99f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[M], 3")                                                                          \
    LA_I("s_cbranch_scc1 91f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 20f")                                                                                 \
    LA_CLASS1("10", "11", "12", LA_PAIR_##K##_M0_T0, LA_V_T0_M0)                                               \
    LA_CLASS1("20", "21", "22", LA_PAIR_##K##_M0_T1, LA_V_T1_M0)                                               \
    "91:\n\t"                                                                                                  \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 40f")                                                                                 \
    LA_CLASS2("30", "31", "32", LA_PAIR_##K##_M1_T0, LA_V_T0_M1)                                               \
    LA_CLASS2("40", "41", "42", LA_PAIR_##K##_M1_T1, LA_V_T1_M1)                                               \
    "99:\n\t"                                                                                                  \
    "s_waitcnt lgkmcnt(0)"

// CO2 has no negative resonance (its "two resonances" mask is empty): the two one-resonance classes alone
#define LA_RUN1(K)                                                                                             \
    LA_I("s_waitcnt lgkmcnt(0)")                                                                               \
    LA_LOAD(A, 0, 16, 32, 48)                                                                                  \
    LA_I("s_mov_b64 %[sv], exec")                                                                              \
    "90:\n\t"                                                                                                  \
    LA_I("s_cmp_lt_i32 %[n], 2")                                                                               \
    LA_I("s_cbranch_scc1 99f")                                                                                 \
    LA_I("s_and_b64 %[tmp], %[T], 3")                                                                          \
    LA_I("s_cbranch_scc1 20f")                                                                                 \
    LA_CLASS1("10", "11", "12", LA_PAIR_##K##_M0_T0, LA_V_T0)                                                  \
    LA_CLASS1("20", "21", "22", LA_PAIR_##K##_M0_T1, LA_V_T1)                                                  \
    "99:\n\t"                                                                                                  \
    "s_waitcnt lgkmcnt(0)"

namespace {

// KIND 0 generic molecule, 1 O2, 2 CO2.  BOFF: byte distance from a line's HotA record to its HotB::pb (one LDS object).
// addr: LDS byte address of the current line's HotA record; n: lines left in the run; T / M: "tested" / "two resonances"
// masks, bit 0 = current line.  Evaluates the lines of the run two at a time - a pair takes the class of the more general of
// its two lines - and leaves n = 0 or 1 with addr, T, M advanced to the odd last line.
template <int KIND, unsigned BOFF>
__device__ __forceinline__ void asm_run(unsigned &addr, int &n, unsigned long long &T, unsigned long long &M, double WN, double &SF) {
    unsigned long long sv, cm, tmp, x;
    int k, k2;
    const double c25 = 25.;
    const unsigned long long c55 = 0x5555555555555555ull;
#define LA_OPERANDS                                                                                                            \
    : [sf] "+v"(SF), [addr] "+v"(addr), [n] "+s"(n), [T] "+s"(T), [M] "+s"(M), [sv] "=&s"(sv), [cm] "=&s"(cm), [tmp] "=&s"(tmp), \
      [x] "=&s"(x), [k] "=&s"(k), [k2] "=&s"(k2)                                                                               \
    : [wn] "v"(WN), [c25] "s"(c25), [c55] "s"(c55), [ob0] "i"(BOFF), [ob1] "i"(BOFF + 32u), [ob2] "i"(BOFF + 64u),             \
      [ob3] "i"(BOFF + 96u), [ob4] "i"(BOFF + 128u), [ob5] "i"(BOFF + 160u)                                                    \
    : LA_CLOBBERS
    if constexpr (KIND == 0) asm volatile(LA_RUN_K0 LA_OPERANDS);
    else if constexpr (KIND == 1) asm volatile(LA_RUN(K1) LA_OPERANDS);
    else {
        const double c625 = 1.0 / 625.;
        asm volatile(LA_RUN1(K2)
                     : [sf] "+v"(SF), [addr] "+v"(addr), [n] "+s"(n), [T] "+s"(T), [M] "+s"(M), [sv] "=&s"(sv), [cm] "=&s"(cm), [tmp] "=&s"(tmp),
                       [x] "=&s"(x), [k] "=&s"(k), [k2] "=&s"(k2)
                     : [wn] "v"(WN), [c25] "s"(c25), [c625] "s"(c625), [c55] "s"(c55)
                     : LA_CLOBBERS);
    }
#undef LA_OPERANDS
}

}  // namespace
