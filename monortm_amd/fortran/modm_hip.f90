! Drop-in replacement of the reference's MODULE ModmMod (reference src/modm.f90:5-274): same module name, the
! same single PUBLIC procedure MODM with the same argument list, so PROGRAM MONORTM's call
! (reference src/monortm.f90:557-561) compiles unchanged.  All arithmetic (line sum, MT_CKD continuum,
! cloud liquid, totals) runs on the MI355X through the C ABI entry monortm_hip_modm.
MODULE ModmMod
  USE, INTRINSIC :: ISO_C_BINDING
  USE monortm_hip_c
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: MODM

CONTAINS

  SUBROUTINE MODM(IPR, ICP, NWN, WN, dvset, NLAY, P, T, CLW, &
                  O, O_BY_MOL, OC, O_CLW, ODXSEC, &
                  NMOL, WKL, WBRODL, &
                  SCLCPL, SCLHW, Y0RES, HFILE, cntnmScaleFac, ixsect, IBRD)
    USE CntnmFactors, ONLY: CntnmFactors_t
    USE RTMmono, ONLY: NWNMX
    USE lblparams, ONLY: MXLAY, MXMOL, MX_XS
    USE xsec_hip, ONLY: xsec_tables_to_device
    INTEGER, INTENT(IN) :: IPR
    INTEGER ICP, NWN, NLAY, NMOL, ixsect, IBRD
    REAL O(:, :), OC(:, :, :), O_BY_MOL(:, :, :), O_CLW(:, :), odxsec(:, :), CLW(MXLAY), P(MXLAY), T(MXLAY)
    REAL*8 WN(NWNMX)
    REAL WKL(MXMOL, MXLAY), WBRODL(MXLAY)
    REAL dvset, SCLCPL, SCLHW, Y0RES
    CHARACTER HFILE*80
    TYPE(CntnmFactors_t) :: cntnmScaleFac

    INTEGER, PARAMETER :: index_cont(5) = (/1, 2, 3, 7, 22/)      ! reference src/modm.f90:166
    ! compact staging copies in the caller's own REAL kind (the caller's arrays are dimensioned MXLAY / MXMOL / NWNMX)
    ! (kept between calls: a driver that loops over profiles of one shape allocates them once)
    REAL(hreal), ALLOCATABLE, SAVE :: p8(:), t8(:), c8(:), w8(:, :), b8(:)
    REAL(hreal), ALLOCATABLE, SAVE :: o8(:, :), om8(:, :, :), oc8(:, :, :), ol8(:, :)
    REAL(C_DOUBLE) :: fac(7)
    CHARACTER(KIND=C_CHAR) :: cpath(81)
    INTEGER(C_INT) :: rc, nl(1)
    INTEGER :: i, n
    ! Cross-section molecules: MODM's hidden inputs for IXSECT = 1, exactly the COMMON blocks MONORTM_XSEC_SUB reads
    ! (reference src/monortm_sub.F90:1603-1610): the request and the layer amounts (/PATHX/, filled by the driver,
    ! src/monortm.f90:492-530) and the spectral regions / file names XSREAD found on FSCDXS (/XSECTR/, /XSECTF/)
    INTEGER :: IXMAX, IXMOLS, IXINDX(MX_XS)
    REAL :: XAMNT(MX_XS, MXLAY)
    COMMON /PATHX/ IXMAX, IXMOLS, IXINDX, XAMNT
    REAL(hreal), ALLOCATABLE, SAVE :: xa8(:, :), ox8(:, :)

    ! first call: load the line file for [wn(1)-25, wn(nwn)+25] (reference src/modm.f90:187-190)
    ! (a context that RTM / CALCTMR created before the first MODM call holds no line table: replace it)
    IF (C_ASSOCIATED(hip_ctx)) THEN
       IF (monortm_hip_has_lines(hip_ctx) == 0) THEN
          CALL monortm_hip_finalize(hip_ctx)
          hip_ctx = C_NULL_PTR
       END IF
    END IF
    IF (.NOT. C_ASSOCIATED(hip_ctx)) THEN
       n = LEN_TRIM(HFILE)
       DO i = 1, n
          cpath(i) = HFILE(i:i)
       END DO
       cpath(n + 1) = C_NULL_CHAR
       rc = monortm_hip_init(cpath, WN(1), WN(NWN), INT(ICP, C_INT), hip_real_kind, -1_C_INT, hip_ctx)
       IF (rc /= 0) CALL hip_fail('GET_LNFL (monortm_hip_init)', rc)
       CALL log_tape3_header(IPR, HFILE)     ! the line-file summary GET_LNFL leaves on the log unit (PRLNHD)
    END IF

    IF (ALLOCATED(om8)) THEN
       IF (ANY(SHAPE(om8) /= (/NWN, NMOL, NLAY/))) DEALLOCATE (p8, t8, c8, w8, b8, o8, om8, oc8, ol8)
    END IF
    IF (.NOT. ALLOCATED(om8)) THEN
       ALLOCATE (p8(NLAY), t8(NLAY), c8(NLAY), w8(NMOL, NLAY), b8(NLAY))
       ALLOCATE (o8(NWN, NLAY), om8(NWN, NMOL, NLAY), oc8(NWN, 5, NLAY), ol8(NWN, NLAY))
    END IF
    p8 = P(1:NLAY)
    t8 = T(1:NLAY)
    c8 = CLW(1:NLAY)
    w8 = WKL(1:NMOL, 1:NLAY)
    b8 = WBRODL(1:NLAY)
    fac = (/cntnmScaleFac%xself, cntnmScaleFac%xfrgn, cntnmScaleFac%xco2c, cntnmScaleFac%xo3cn, &
            cntnmScaleFac%xo2cn, cntnmScaleFac%xn2cn, cntnmScaleFac%xrayl/)
    nl(1) = INT(NLAY, C_INT)

    IF (ixsect == 1) THEN
       ! the reference re-reads the xs files in every MONORTM_XSEC_SUB call (src/monortm_sub.F90:1659-1673); so does this
       CALL xsec_tables_to_device(hip_ctx)
       IF (ALLOCATED(xa8)) THEN
          IF (ANY(SHAPE(xa8) /= (/IXMOLS, NLAY/)) .OR. ANY(SHAPE(ox8) /= (/NWN, NLAY/))) DEALLOCATE (xa8, ox8)
       END IF
       IF (.NOT. ALLOCATED(xa8)) ALLOCATE (xa8(IXMOLS, NLAY), ox8(NWN, NLAY))
       xa8 = XAMNT(1:IXMOLS, 1:NLAY)
       rc = monortm_hip_modm_xs(hip_ctx, 1_C_INT, INT(NWN, C_INT), WN, REAL(dvset, C_DOUBLE), nl, INT(NLAY, C_INT), &
            INT(NMOL, C_INT), p8, t8, c8, w8, b8, fac, REAL(SCLCPL, C_DOUBLE), REAL(SCLHW, C_DOUBLE), &
            REAL(Y0RES, C_DOUBLE), INT(IBRD, C_INT), 1_C_INT, xa8, ox8, o8, om8, oc8, ol8)
    ELSE
       rc = monortm_hip_modm(hip_ctx, 1_C_INT, INT(NWN, C_INT), WN, REAL(dvset, C_DOUBLE), nl, INT(NLAY, C_INT), &
            INT(NMOL, C_INT), p8, t8, c8, w8, b8, fac, REAL(SCLCPL, C_DOUBLE), REAL(SCLHW, C_DOUBLE), &
            REAL(Y0RES, C_DOUBLE), INT(IBRD, C_INT), INT(ixsect, C_INT), o8, om8, oc8, ol8)
    END IF
    IF (rc /= 0) CALL hip_fail('MODM', rc)

    ! scatter the compact results into the caller's strided arrays; the reference zeroes
    ! oc(1:nwn,1:mxmol,1:nlay) and odxsec(1:nwn,1:nlay) itself (src/modm.f90:192-195)
    oc(1:NWN, 1:MXMOL, 1:NLAY) = 0.
    odxsec(1:NWN, 1:NLAY) = 0.
    IF (ixsect == 1) odxsec(1:NWN, 1:NLAY) = ox8   ! (indexed as the caller dimensioned it: see INTEGRATION.md on src/monortm_sub.F90:1611)
    o(1:NWN, 1:NLAY) = o8
    O_BY_MOL(1:NWN, 1:NMOL, 1:NLAY) = om8
    O_CLW(1:NWN, 1:NLAY) = ol8
    DO i = 1, 5
       oc(1:NWN, index_cont(i), 1:NLAY) = oc8(:, i, :)
    END DO
  END SUBROUTINE MODM

  ! The "LINE FILE INFORMATION" block that the reference writes to unit IPR (MONORTM.LOG) when it opens TAPE3: PRLNHD,
  ! src/lnfl_mod.f90:273-289, formats 900-920 / 960-965.  The shim reads the header record(s) itself (the line table proper is
  ! parsed by the library); a unit that is not open for writing is left alone.
  SUBROUTINE log_tape3_header(IPR, HFILE)
    INTEGER, INTENT(IN) :: IPR
    CHARACTER(LEN=*), INTENT(IN) :: HFILE
    CHARACTER(LEN=8) :: HLINID(10), BMOLID(64), HID1(2)
    INTEGER(KIND=4) :: MOLCNT(64), MCNTLC(64), MCNTNL(64), LINMOL, LINCNT, ILINLC, ILINNL, IREC, IRECTL
    INTEGER(KIND=4) :: N_NEGEPP(64), N_RESETEPP(64)
    REAL(KIND=4) :: SUMSTR(64), FLINLO, FLINHI, XSPACE(4096)
    INTEGER :: lu, ios, I
    LOGICAL :: isopen
    CHARACTER(LEN=16) :: act
    INQUIRE (UNIT=IPR, OPENED=isopen, ACTION=act)
    IF (.NOT. isopen) RETURN
    IF (INDEX(act, 'WRITE') == 0) RETURN
    OPEN (NEWUNIT=lu, FILE=TRIM(HFILE), FORM='UNFORMATTED', STATUS='OLD', ACTION='READ', IOSTAT=ios)
    IF (ios /= 0) RETURN
    READ (lu, IOSTAT=ios) HLINID, BMOLID, MOLCNT, MCNTLC, MCNTNL, SUMSTR, LINMOL, FLINLO, FLINHI, LINCNT, ILINLC, ILINNL, IREC, &
       IRECTL, HID1
    IF (ios /= 0) THEN
       CLOSE (lu)
       RETURN
    END IF
    LINMOL = MAX(0, MIN(LINMOL, 64))
    WRITE (IPR, 900)
    WRITE (IPR, 905) HLINID, HID1
    IF (HLINID(7) (8:8) == '^') THEN     ! negative lower-state energies were found by LNFL: a second header record
       READ (lu, IOSTAT=ios) N_NEGEPP, N_RESETEPP, XSPACE
       IF (ios /= 0) THEN
          N_NEGEPP = 0
          N_RESETEPP = 0
       END IF
       WRITE (IPR, 960)
       WRITE (IPR, 965) (BMOLID(I), MOLCNT(I), MCNTLC(I), MCNTNL(I), N_NEGEPP(I), N_RESETEPP(I), SUMSTR(I), I=1, LINMOL)
    ELSE
       WRITE (IPR, 910)
       WRITE (IPR, 915) (BMOLID(I), MOLCNT(I), MCNTLC(I), MCNTNL(I), SUMSTR(I), I=1, LINMOL)
    END IF
    WRITE (IPR, 920) FLINLO, FLINHI, LINCNT
    CLOSE (lu)
900 FORMAT('0'/'0', 20X, '   LINE FILE INFORMATION ')
905 FORMAT('0', 10A8, 2X, 2(1X, A8, 1X))
910 FORMAT('0', /, 23X, 'COUPLED', 4X, 'NLTE', 3X, 'SUM LBLRTM ', /, 7X, 'MOL', 5X, 'LINES', 4X, 'LINES', 4X, 'LINES', 4X, 'STRENGTHS', /)
915 FORMAT(' ', 4X, A6, ' = ', I6, 3X, I6, 3X, I6, 2X, 1PE12.4, 0P)
920 FORMAT(/, '0 LOWEST LINE = ', F10.3, 5X, '  HIGHEST LINE = ', F10.3, 5X, ' TOTAL NUMBER OF LINES =', I8)
960 FORMAT('0', /, 23X, 'COUPLED', 4X, 'NLTE', 3X, 'NEGATIVE', 3X, 'RESET', 4X, 'SUM LBLRTM', /, 7X, 'MOL', 5X, 'LINES', 4X, &
           'LINES', 4X, 'LINES', 6X, 'EPP', 6X, 'EPP', 6X, 'STRENGTHS', /)
965 FORMAT(' ', 4X, A6, ' = ', I6, 3X, I6, 3X, I6, 3X, I6, 3X, i6, 3X, 1PE12.4)
  END SUBROUTINE log_tape3_header

END MODULE ModmMod
