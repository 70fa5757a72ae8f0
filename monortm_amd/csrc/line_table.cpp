// Host-side TAPE3 reader and line-table builder.  See line_table.hpp.
#include "line_table.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/monortm_hip.h"

namespace monortm {
namespace {

struct Rec {  // one 156-byte TAPE3 slot, reference src/struct_types.f90:33-43
    double vnu;
    float sp, alfa, epp, hwhm, tmpalf, pshift, sdep;
    int32_t mol, iflg;
    int32_t brd_flg[7];
    float brd_dat[21];
};

class RecordFile {  // Fortran sequential unformatted, 4-byte markers (build/makefile.common:198)
   public:
    explicit RecordFile(FILE *f) : f_(f) {}
    // 1 = record read, 0 = clean EOF, -1 = framing error
    int next(std::vector<unsigned char> &buf) {
        int32_t m1 = 0, m2 = 0;
        if (std::fread(&m1, 4, 1, f_) != 1) return 0;
        if (m1 < 0) return -1;
        buf.resize(size_t(m1));
        if (m1 && std::fread(buf.data(), 1, size_t(m1), f_) != size_t(m1)) return -1;
        if (std::fread(&m2, 4, 1, f_) != 1 || m2 != m1) return -1;
        return 1;
    }

   private:
    FILE *f_;
};

template <class T>
T at(const std::vector<unsigned char> &b, size_t byte_off, size_t i) {
    T v;
    std::memcpy(&v, b.data() + byte_off + i * sizeof(T), sizeof(T));
    return v;
}

}  // namespace

int load_tape3(const std::string &path, double v1, double v2, LineTable &out, std::string &err, int real_kind) {
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) {
        err = "ERROR OPENING HITRAN FILE: " + path;
        return MONORTM_EIO;
    }
    RecordFile rf(fp);
    std::vector<unsigned char> buf;
    auto fail = [&](const char *msg) {
        err = std::string("TAPE3: ") + msg;
        std::fclose(fp);
        return MONORTM_EFORMAT;
    };
    if (rf.next(buf) != 1 || buf.size() < 1664) return fail("header record missing or short");
    // HLINID(7) char 8 == '^' announces a second header record; HLINID(10) char 8 must be 'I'
    // (reference src/lnfl_mod.f90:258-262, :297-302)
    const bool second_header = buf[6 * 8 + 7] == '^';
    if (buf[9 * 8 + 7] != 'I') return fail("PRLNHD - NO ISOTOPE INFO ON LINFIL");
    if (second_header && rf.next(buf) != 1) return fail("second header record missing");

    // ---- pass 1: keep the blocks the reference keeps, regroup records per molecule ----------------
    std::vector<std::vector<Rec>> per_mol(kMaxMol + 1);
    const double vlo_adj = std::max(0.0, v1 - 25.0);  // lnfl_mod.f90:160
    int mo_prev = 0;
    // What the reference finds when a coupling record is slot 1 of a block (its owner line was the last record of the block
    // before): GET_LNFL reads bufr%mol(ik-1) / bufr%iflg(ik-1) with ik = 1 (lnfl_mod.f90:50,53,55), i.e. the array elements
    // BEFORE mol(1) / iflg(1) of TYPE(LINE_DATA) - which are epp(250) and pshift(250) (struct_types.f90:45-58: components in
    // declaration order, every one 250 default-kind words; measured with LOC() under amdflang for both flag sets).  RDLNFL
    // overwrites bufr(1:NREC) only (lnfl_mod.f90:177-200), so those two hold the values of the most recent KEPT block that
    // had 250 records - zero before any.  The owner becomes MOD(<bits of epp(250) as a default INTEGER>, 100): the bits of the
    // REAL*8 widening in the "dbl" build, of the REAL*4 value in the "sgl" build.  Reproduced literally when it names a
    // molecule 1..39; otherwise the reference writes outside its arrays (nblm(0) aliases bufr%speed_dep(250), iso(0,1) aliases
    // nblm(39), ...) and the file is refused.  LNFL itself keeps a line and its coupling records in one block.
    float bufr_epp250 = 0.f, bufr_pshift250 = 0.f;
    auto mol0_owner = [&]() -> long long {
        if (real_kind == 4) { int32_t b; std::memcpy(&b, &bufr_epp250, 4); return (long long)(b % 100); }
        const double d = (double)bufr_epp250;
        int64_t b; std::memcpy(&b, &d, 8);
        return (long long)(b % 100);
    };
    for (;;) {
        int rc = rf.next(buf);
        if (rc == 0) break;  // EOF on a panel header ends the read (lnfl_mod.f90:161)
        if (rc < 0 || buf.size() < 24) return fail("bad panel header");
        const double vmax = at<double>(buf, 8, 0);
        const int32_t nrec = at<int32_t>(buf, 16, 0);
        if (rf.next(buf) != 1) return fail("line block missing after panel header");
        if (vmax < vlo_adj) continue;  // whole block below the window (lnfl_mod.f90:162-165)
        if (buf.size() < size_t(4 * kBlockWords) || nrec < 0 || nrec > kSlots) return fail("bad line block");
        double last_vnu = 0.0;
        int32_t prev_mol = 0, prev_iflg = 0;
        if (nrec >= kSlots) {  // RDLNFL has filled bufr(1:250) before GET_LNFL walks the block
            bufr_epp250 = at<float>(buf, 4000, kSlots - 1);
            bufr_pshift250 = at<float>(buf, 8000, kSlots - 1);
        }
        for (int ik = 0; ik < nrec; ik++) {
            Rec r;
            r.vnu = at<double>(buf, 0, ik);
            r.sp = at<float>(buf, 2000, ik);
            r.alfa = at<float>(buf, 3000, ik);
            r.epp = at<float>(buf, 4000, ik);
            r.mol = at<int32_t>(buf, 5000, ik);
            r.hwhm = at<float>(buf, 6000, ik);
            r.tmpalf = at<float>(buf, 7000, ik);
            r.pshift = at<float>(buf, 8000, ik);
            r.iflg = at<int32_t>(buf, 9000, ik);
            for (int j = 0; j < 7; j++) r.brd_flg[j] = at<int32_t>(buf, 10000, size_t(ik) * 7 + j);
            for (int j = 0; j < 21; j++) r.brd_dat[j] = at<float>(buf, 17000, size_t(ik) * 21 + j);
            r.sdep = at<float>(buf, 38000, ik);
            // owner molecule of the record (lnfl_mod.f90:46-64)
            int mo;
            bool slot1 = false;
            if (r.iflg >= 0 && r.iflg <= 100) mo = r.mol % 100;
            else if (r.iflg >= -3 && r.iflg <= -1) {
                if (ik > 0) mo = prev_mol % 100;
                else { mo = (int)mol0_owner(); slot1 = true; }
            } else if (r.iflg == -5) {
                // ik = 1: bufr%iflg(0) aliases pshift(250) - as an INTEGER it is >= 0 exactly when the sign bit is clear
                const bool prev_is_line = (ik > 0) ? prev_iflg >= 0 : !std::signbit(bufr_pshift250);
                if (prev_is_line) {
                    if (ik > 0) mo = prev_mol % 100;
                    else { mo = (int)mol0_owner(); slot1 = true; }
                    mo_prev = mo;
                } else mo = mo_prev;
            } else {
                char m[96];
                std::snprintf(m, sizeof m, "LC flag not recognized: %d. Must be 1, 3 or 5.", r.iflg);
                return fail(m);
            }
            if (slot1 && (mo < 1 || mo > kMaxMol))
                return fail("a coupling record is the first record of a block and the reference's owner rule for it (bits of the block's "
                            "250th lower-state energy, lnfl_mod.f90:50-58) names no molecule 1..39: the reference corrupts its tables on this file");
            if (mo < 1 || mo > kMaxMol) return fail("molecule number outside 1..39");
            per_mol[mo].push_back(r);
            prev_mol = r.mol;
            prev_iflg = r.iflg;
            last_vnu = r.vnu;
        }
        if (nrec > 0 && last_vnu > v2 + 25.0) break;  // lnfl_mod.f90:116
    }
    std::fclose(fp);

    // ---- pass 2: the LINES record walk (modm.f90:324-354, :434), hoisted --------------------------
    const double PLANCK = 6.62606876E-27, BOLTZ = 1.3806503E-16, CLIGHT = 2.99792458E+10;  // PhysConstants.f90:27-29
    const double RADCT = PLANCK * CLIGHT / BOLTZ, T0 = 296.0;                                 // modm.f90:874-875
    out = LineTable();
    auto xg_of = [](const Rec &r) { return r.iflg >= 0 ? -r.iflg : r.iflg; };  // lnfl_mod.f90:75-79
    auto is_lc = [](int xg) { return xg == -1 || xg == -3 || xg == -5; };
    auto push_set = [&](const Rec &c) {
        float xmol;
        std::memcpy(&xmol, &c.mol, 4);  // RMOL: the MOL word re-read as REAL*4 (lnfl_mod.f90:80-82)
        // A(1) = XNU0 is REAL*8 on file, the other seven fields are REAL*4 (struct_types.f90:33-40)
        const double s[8] = {c.vnu, c.alfa, xmol, c.tmpalf, c.sp, c.epp, c.hwhm, c.pshift};
        out.lc.insert(out.lc.end(), s, s + 8);
    };
    for (int mo = 1; mo <= kMaxMol; mo++) {
        out.mol_start[mo] = int(out.size());
        const auto &L = per_mol[mo];
        const int n = int(L.size());
        bool asc = true;
        double prevv = -1e300;
        int J = 0;  // 1-based like the reference
        while (J < n) {
            J = J + 1;
            int JJ = J;
            const Rec &r = L[J - 1];
            const int xg = xg_of(r);
            int code = 0, self = 0;
            uint32_t lcidx = 0;
            if (is_lc(xg)) {
                // A coupling set that would lie past the end of the molecule's list: the reference reads its module arrays
                // beyond NBLM(I), which GET_LNFL never wrote and which still hold their initial zeros -> an all-zero set
                // (AIP = BIP = 0), not "no coupling" (the XG code still selects the coupled branch of the shape function)
                const Rec zero{};
                auto rec_at = [&](int jj) -> const Rec & { return (jj <= n) ? L[jj - 1] : zero; };
                JJ = J + 1;
                code = (xg == -1) ? 1 : (xg == -3 ? 2 : 3);
                lcidx = uint32_t(out.lc.size() / 8);
                push_set(rec_at(JJ));
                const int xg_prev = (J >= 2) ? xg_of(L[J - 2]) : 0;  // XG(I,0) is an out-of-bounds read in the reference
                if (xg == -5 && xg_prev == -5) {
                    JJ = JJ + 1;
                    self = 1;
                    push_set(rec_at(JJ));
                }
                if (lcidx + 2 >= kMaxLcSets) {
                    err = "TAPE3: too many line-coupling records";
                    return MONORTM_EUNSUPPORTED;
                }
            }
            int iso = (r.mol % 1000) / 100;  // lnfl_mod.f90:67
            if (iso < 1 || iso > 9) iso = 0;
            const bool phys = r.iflg >= 0;
            const int wfix = phys && (mo == 7 || mo == 22);
            const int sfix = phys && mo == 7 && r.brd_flg[6] > 0;
            out.vnu.push_back(r.vnu);
            out.s0adj.push_back(double(r.sp) * (r.vnu * (1.0 - std::exp(-(RADCT * r.vnu / T0)))));
            out.alfa.push_back(r.alfa);
            out.hwhm.push_back(r.hwhm);
            out.epp.push_back(r.epp);
            out.tmpalf.push_back(r.tmpalf);
            out.pshift.push_back(r.pshift);
            out.sdep.push_back(r.sdep);
            out.meta.push_back(pack_meta(mo, iso, code, self, wfix, sfix, lcidx));
            for (int j = 0; j < 7; j++) {
                const int f = (mo <= 7) ? r.brd_flg[j] : 0;
                out.brd_flg.push_back(f);
                if (f) out.any_brd = true;
            }
            for (int j = 0; j < 21; j++) out.brd_dat.push_back(mo <= 7 ? r.brd_dat[j] : 0.f);
            {   // bound on |Xnu - XNU0| / RHORAT of this entry (modm.f90:375-380): the shift the kernel applies is
                // delt_eff * RHORAT + sum_j rho_j flg_j (shift_j - delt_eff) with sum_j rho_j <= RHORAT, and delt_eff is
                // the O2 self-shift-corrected value of lnfl_mod.f90:107-112 (up to 1.53 x the raw field)
                double delt_eff = double(r.pshift), mx = 0.0;
                if (sfix) delt_eff = (delt_eff - 0.21 * double(r.brd_dat[3 * 6 + 2])) / (1.0 - 0.21);
                if (mo <= 7)
                    for (int j = 0; j < 7; j++) mx = std::max(mx, std::fabs(double(r.brd_dat[3 * j + 2])));
                out.max_abs_shift = std::max(out.max_abs_shift, 2.0 * std::fabs(delt_eff) + mx);
            }
            if (r.vnu < prevv) asc = false;
            prevv = r.vnu;
            J = JJ;
        }
        out.sorted[mo] = asc;
        // every IFLG >= 0 record of the kept blocks is a physical line, walked as one or not
        for (const Rec &r : L) out.n_physical[mo] += (r.iflg >= 0);
        out.n_physical[0] += out.n_physical[mo];
    }
    out.mol_start[kMaxMol + 1] = int(out.size());
    return MONORTM_OK;
}

}  // namespace monortm
