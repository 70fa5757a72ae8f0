/* call_abi.c - the C ABI of include/monortm_hip.h used from plain C (no Python, no Fortran, no torch):
 * one MODM + CALCTMR/RTM call for a small isothermal-layer profile, brightness temperatures printed.
 *
 *   gcc -std=c99 -I include examples/call_abi.c -L monortm_amd/lib -lmonortm_hip -Wl,-rpath,$PWD/monortm_amd/lib -lm -o call_abi
 *   ./call_abi TAPE3
 *
 * Inputs follow the reference's MODM / RTM arguments (src/modm.f90:21-25, src/RTMmono.f90:13-14); arrays are C-ordered
 * with the wavenumber axis fastest. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "monortm_hip.h"

#define NWN 6
#define NLAY 4
#define NMOL 7

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s TAPE3\n", argv[0]);
        return 2;
    }
    const double wn[NWN] = {0.7417, 0.7939, 1.0474, 1.7, 3.0, 5.0}; /* cm-1, ascending */
    double P[NLAY], T[NLAY], TZ[NLAY + 1], CLW[NLAY] = {0}, WKL[NLAY][NMOL], WBRODL[NLAY];
    const double vmr[NMOL] = {0.0, 4.0e-4, 3.0e-7, 3.2e-7, 1.5e-7, 1.7e-6, 0.209};
    for (int k = 0; k <= NLAY; k++) TZ[k] = 288.0 - 6.5 * 2.0 * k; /* levels every 2 km */
    for (int k = 0; k < NLAY; k++) {
        const double z = 2.0 * k + 1.0, p = 1013.0 * exp(-z / 7.5), dp = 1013.0 * (exp(-(z - 1.0) / 7.5) - exp(-(z + 1.0) / 7.5));
        const double air = 2.1e25 * dp / 1013.0; /* molecules / cm2 of the layer */
        P[k] = p;
        T[k] = 0.5 * (TZ[k] + TZ[k + 1]);
        for (int m = 0; m < NMOL; m++) WKL[k][m] = vmr[m] * air;
        WKL[k][0] = 0.01 * exp(-z / 2.0) * air; /* water vapour */
        WBRODL[k] = 0.781 * air;
    }
    const double cntnm[7] = {1, 1, 1, 1, 1, 1, 1};
    const int nlay[1] = {NLAY}, irt[1] = {3}; /* downwelling */
    double O[NLAY][NWN], OBM[NLAY][NMOL][NWN], OC[NLAY][MONORTM_NCONT][NWN], OCLW[NLAY][NWN];
    double tmpsfc[1] = {288.0}, emiss[NWN], reflc[NWN], RUP[NWN], RDN[NWN], TRTOT[NWN], RAD[NWN], TB[NWN], TMR[NWN];
    for (int i = 0; i < NWN; i++) { emiss[i] = 1.0; reflc[i] = 0.0; }

    void *ctx = NULL;
    int rc = monortm_hip_init(argv[1], wn[0], wn[NWN - 1], 1, 8, -1, &ctx);
    if (rc) { fprintf(stderr, "init failed (%d): %s\n", rc, monortm_hip_last_error(NULL)); return 1; }
    rc = monortm_hip_modm(ctx, 1, NWN, wn, 0.0, nlay, NLAY, NMOL, P, T, CLW, WKL, WBRODL, cntnm, 1.0, 1.0, 0.0, 0, 0, O, OBM, OC, OCLW);
    if (!rc) rc = monortm_hip_rtm(ctx, 1, NWN, wn, nlay, NLAY, irt, 1, T, TZ, O, tmpsfc, emiss, reflc, RUP, RDN, TRTOT, RAD, TB, TMR);
    if (rc) { fprintf(stderr, "call failed (%d): %s\n", rc, monortm_hip_last_error(ctx)); return 1; }
    printf("lines loaded: %lld\n", monortm_hip_line_count(ctx, 0));
    for (int i = 0; i < NWN; i++) {
        double od = 0;
        for (int k = 0; k < NLAY; k++) od += O[k][i];
        printf("%9.4f cm-1  TB %10.5f K  TMR %10.5f K  total OD %.9e\n", wn[i], TB[i], TMR[i], od);
    }
    monortm_hip_finalize(ctx);
    return 0;
}
