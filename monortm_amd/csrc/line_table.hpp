// Host-side TAPE3 reader and line-table builder (plain C++17, no HIP).
//
// Replaces GET_LNFL / RDLNFL / PRLNHD (reference src/lnfl_mod.f90:22-331) and hoists the parts of
// LINES that depend only on the line list (reference src/modm.f90:324-372: the record walk that
// pairs a line with its coupling records, and S0_adj).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace monortm {

constexpr int kMaxMol = 39;     // MXMOL, reference src/lblparams.f90:28
constexpr int kSlots = 250;     // NLINEREC, reference src/struct_types.f90:27
constexpr int kBlockWords = 9750;

// meta word of one table entry
//   bits 0-5 molecule (1..39) | 6-9 isotopologue (0 = outside 1..9) | 10-11 coupling code
//   (0 none, 1: XG=-1, 2: XG=-3, 3: XG=-5) | 12 self-coupling set present | 13 O2/N2 air->foreign
//   width fix applies | 14 O2 shift fix applies (lnfl_mod.f90:98-113) | 15-31 index of the first
//   coupling set in `lc`
inline uint32_t pack_meta(int mol, int iso, int code, int self, int wfix, int sfix, uint32_t lcidx) {
    return uint32_t(mol) | (uint32_t(iso) << 6) | (uint32_t(code) << 10) | (uint32_t(self) << 12) |
           (uint32_t(wfix) << 13) | (uint32_t(sfix) << 14) | (lcidx << 15);
}
constexpr uint32_t kMaxLcSets = 1u << 17;

struct LineTable {
    // one entry per record the reference's LINES loop *treats as a line*, grouped by molecule in the
    // order the loop visits them (file order); 44 bytes per entry in total.
    std::vector<double> vnu;     // XNU0
    std::vector<double> s0adj;   // S0 * nu0 * (1 - exp(-RADCT nu0 / T0))            modm.f90:372
    std::vector<float> alfa;     // TAPE3 ALFA  (air/foreign HWHM, before the O2/N2 fix lnfl_mod.f90:98-113)
    std::vector<float> hwhm;     // TAPE3 HWHM  (self HWHM)
    std::vector<float> epp;      // lower-state energy
    std::vector<float> tmpalf;   // temperature exponent
    std::vector<float> pshift;   // pressure shift
    std::vector<float> sdep;     // speed dependence
    std::vector<uint32_t> meta;
    std::vector<double> lc;      // 8 values per coupling set: A(1..4) then B(1..4)      modm.f90:331-338
    // species-by-species broadening (IBRD), 7 flags + 7x(hw,tmp,shift) per entry, only filled for mol <= 7
    std::vector<int32_t> brd_flg;
    std::vector<float> brd_dat;
    int mol_start[kMaxMol + 2] = {0};   // entries of molecule m: [mol_start[m], mol_start[m+1])
    long long n_physical[kMaxMol + 1] = {0};  // records with IFLG >= 0 per molecule (index 0: total)
    bool sorted[kMaxMol + 1] = {false};       // entries of the molecule ascending in vnu
    double max_abs_shift = 0.0;   // max over entries of 2 |delt_eff| + max_j |species shift_j|: |Xnu - XNU0| <= this x RHORAT
    bool any_brd = false;
    size_t size() const { return vnu.size(); }
};

// Returns 0 or a MONORTM_E* code; `err` receives the message.  real_kind: 8 = the reference's "dbl" build, 4 = "sgl" (only a
// coupling record that is the FIRST record of a block depends on it: line_table.cpp, mol0_owner)
int load_tape3(const std::string &path, double v1, double v2, LineTable &out, std::string &err, int real_kind = 8);

}  // namespace monortm
