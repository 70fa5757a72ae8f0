// Host check of far_kernel's workgroup placement and interval numbering (monortm_amd/csrc/device_common.hpp): every (interval,
// molecule) of a level is the item of exactly one (XCD, slot), for any distribution of the lines over the molecules; the levels'
// offsets tile the interval numbers.  Built and run by tests/test_far_layout.py (no GPU: host code only).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../monortm_amd/csrc/device_common.hpp"

using namespace monortm_dev;

int main() {
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    int checked = 0;
    for (int trial = 0; trial < 4000; trial++) {
        const int nmol = 1 + (int)(rnd() % MXMOL);
        int mol_start[MXMOL + 2] = {0};
        mol_start[0] = 0;
        mol_start[1] = (int)(rnd() % 3);   // (lines of "molecule 0" never exist; a non-zero base must not matter)
        for (int m = 0; m < MXMOL; m++) {
            int cnt = 0;
            if (m < nmol) {
                const unsigned k = rnd() % 10;
                cnt = (k < 3) ? 0 : ((k < 6) ? (int)(rnd() % 50) : (int)(rnd() % 60000));
            } else cnt = (int)(rnd() % 100);   // molecules beyond nmol may own lines of the table too
            mol_start[m + 2] = mol_start[m + 1] + cnt;
        }
        const int ntile = 1 + (int)(rnd() % 90), levels = 1 + (int)(rnd() % FAR_MAXLEV);
        int expect_off = 0;
        for (int l = 0; l < levels; l++) {
            if (far_level_offset(ntile, l) != expect_off) { std::printf("offset of level %d: %d != %d\n", l, far_level_offset(ntile, l), expect_off); return 1; }
            const int nint = far_level_count(ntile, l);
            if (nint != (ntile + (1 << l) - 1) / (1 << l) || nint < 1) { std::printf("count of level %d\n", l); return 1; }
            expect_off += nint;
            // the table the host passes and the walk a workgroup makes through it (far_kernel.hip)
            FarPlace place{};
            int most = 0;
            for (int k = 0; k < 8; k++) {
                int items = 0;
                for (int q = 0; q < nmol; q++) {
                    int xlo, nx;
                    far_xcd_share(mol_start, nmol, q, &xlo, &nx);
                    if (nx < 0 || xlo < 0 || xlo + nx > 8) { std::printf("share of molecule %d: [%d, %d)\n", q, xlo, xlo + nx); return 1; }
                    if ((mol_start[q + 2] > mol_start[q + 1]) != (nx > 0)) { std::printf("molecule %d: lines %d, XCDs %d\n", q, mol_start[q + 2] - mol_start[q + 1], nx); return 1; }
                    place.xlo[q] = (unsigned char)xlo;
                    place.nx[q] = (unsigned char)nx;
                    place.cnt[q][k] = (unsigned short)far_xcd_items(nint, k, xlo, nx);
                    items += place.cnt[q][k];
                }
                most = items > most ? items : most;
            }
            std::vector<int> seen((size_t)nint * nmol, 0);
            for (int x = 0; x < 8 * most; x++) {
                const int k = x & 7;
                int slot = x >> 3, m = -1, j = 0;
                for (int q = 0; q < nmol; q++) {
                    const int cnt = place.cnt[q][k];
                    if (slot < cnt) { m = q; j = (k - (int)place.xlo[q]) + slot * (int)place.nx[q]; break; }
                    slot -= cnt;
                }
                if (m < 0) continue;
                if (j < 0 || j >= nint) { std::printf("interval %d of %d\n", j, nint); return 1; }
                seen[(size_t)j * nmol + m]++;
            }
            for (int j = 0; j < nint; j++)
                for (int q = 0; q < nmol; q++) {
                    const int want = mol_start[q + 2] > mol_start[q + 1] ? 1 : 0;
                    if (seen[(size_t)j * nmol + q] != want) {
                        std::printf("trial %d level %d: (interval %d, molecule %d) served %d times, expected %d\n", trial, l, j, q, seen[(size_t)j * nmol + q], want);
                        return 1;
                    }
                    checked++;
                }
        }
    }
    std::printf("ok %d (interval, molecule) pairs\n", checked);
    return 0;
}
