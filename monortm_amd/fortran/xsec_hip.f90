! xsec_hip: the tables of the cross-section molecules from the reference's COMMON blocks to the GPU context.
! Reads the xs files of every spectral region XSREAD kept (as MONORTM_XSEC_SUB does in every call, reference
! src/monortm_sub.F90:1640-1673: header in format 910, values list directed, pressure in torr when the third source word says
! so) and hands the parsed tables to monortm_hip_xsec_tables.  A region none of whose files is needed (no wavenumber within
! 1 cm-1, :1645-1653) is passed all the same: the kernel applies that test itself.  Used by the drop-in MODM and by the own driver.
MODULE xsec_hip
  USE, INTRINSIC :: ISO_C_BINDING
  USE monortm_hip_c
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: xsec_tables_to_device

CONTAINS

  SUBROUTINE xsec_tables_to_device(ctx)
    USE lblparams, ONLY: MXLAY, MX_XS
    TYPE(C_PTR), INTENT(IN) :: ctx
    INTEGER :: IXMAX, IXMOLS, IXINDX(MX_XS)
    REAL :: XAMNT(MX_XS, MXLAY)
    COMMON /PATHX/ IXMAX, IXMOLS, IXINDX, XAMNT
    CHARACTER*10 :: XSFILE, XSNAME, ALIAS
    COMMON /XSECTF/ XSFILE(6, 5, MX_XS), XSNAME(MX_XS), ALIAS(4, MX_XS)
    REAL*8 :: V1FX, V2FX          ! (IMPLICIT REAL*8 (V) in the reference, src/monortm_sub.F90:1574)
    REAL :: DVFX, WXM, XSMASS, XDOPLR
    INTEGER :: NTEMPF, NSPECR, IXFORM, NUMXS, IXSBIN
    COMMON /XSECTR/ V1FX(5, MX_XS), V2FX(5, MX_XS), DVFX(5, MX_XS), WXM(MX_XS), NTEMPF(5, MX_XS), NSPECR(MX_XS), &
         IXFORM(5, MX_XS), XSMASS(MX_XS), XDOPLR(5, MX_XS), NUMXS, IXSBIN
    REAL(C_DOUBLE), ALLOCATABLE :: reg(:, :), tmp(:, :), prs(:, :), pool(:), buf(:)
    INTEGER(C_LONG_LONG), ALLOCATABLE :: offs(:, :)
    INTEGER :: nreg, ir, ixm, isr, itp, npts, nlast, iu, ios, k
    INTEGER(C_LONG_LONG) :: pos, cap
    INTEGER(C_INT) :: rc
    REAL(C_DOUBLE) :: amolv1, amolv2, tx, pres, smax
    CHARACTER*10 :: amol, source(3)
    ! what the last upload was made from: the reference re-reads the xs files in every MONORTM_XSEC_SUB call, i.e. once per
    ! profile (src/monortm_sub.F90:1659-1673); here the tables stay on the device while the context and the /XSECTR/, /XSECTF/
    ! entries that name them are unchanged (MONORTM_XS_REREAD=1 restores the re-read, e.g. for files rewritten during a run)
    TYPE(C_PTR), SAVE :: last_ctx = C_NULL_PTR
    INTEGER, SAVE :: last_ixmols = -1
    INTEGER, SAVE :: last_nspecr(MX_XS), last_ntempf(5, MX_XS)
    REAL*8, SAVE :: last_v1(5, MX_XS), last_v2(5, MX_XS)
    REAL, SAVE :: last_dop(5, MX_XS)
    CHARACTER*10, SAVE :: last_files(6, 5, MX_XS)
    CHARACTER(LEN=8) :: envv
    INTEGER :: envl, envs
    LOGICAL :: same
    CALL GET_ENVIRONMENT_VARIABLE('MONORTM_XS_REREAD', envv, envl, envs)
    same = C_ASSOCIATED(ctx, last_ctx) .AND. IXMOLS == last_ixmols .AND. .NOT. (envs == 0 .AND. envl > 0 .AND. envv(1:1) /= '0')
    IF (same) THEN
       DO ixm = 1, IXMOLS
          IF (NSPECR(ixm) /= last_nspecr(ixm)) same = .FALSE.
          IF (.NOT. same) EXIT
          DO isr = 1, NSPECR(ixm)
             IF (NTEMPF(isr, ixm) /= last_ntempf(isr, ixm) .OR. V1FX(isr, ixm) /= last_v1(isr, ixm) .OR. &
                 V2FX(isr, ixm) /= last_v2(isr, ixm) .OR. XDOPLR(isr, ixm) /= last_dop(isr, ixm)) same = .FALSE.
             DO itp = 1, MIN(NTEMPF(isr, ixm), 6)
                IF (XSFILE(itp, isr, ixm) /= last_files(itp, isr, ixm)) same = .FALSE.
             END DO
          END DO
       END DO
    END IF
    nreg = 0
    DO ixm = 1, IXMOLS
       nreg = nreg + NSPECR(ixm)
    END DO
    ! (a new context may sit at the address of one that was finalised: it must hold the regions as well)
    IF (same .AND. nreg > 0 .AND. monortm_hip_xsec_regions(ctx) == nreg) RETURN
    ALLOCATE (reg(8, MAX(nreg, 1)), tmp(6, MAX(nreg, 1)), prs(6, MAX(nreg, 1)), offs(6, MAX(nreg, 1)))
    reg = 0; tmp = 0; prs = 0; offs = 0
    cap = 1000000
    ALLOCATE (pool(cap))
    pos = 0
    ir = 0
    iu = 97
    DO ixm = 1, IXMOLS
       DO isr = 1, NSPECR(ixm)
          ir = ir + 1
          ! Two passes over the region's files.  The reference reads every file into its own column of XSDAT and then takes V1,
          ! V2 and the NUMBER OF POINTS of the grid from the LAST file it read (src/monortm_sub.F90:1663-1666,:1709): every
          ! spectrum is therefore packed at a stride of the last file's point count, zero-filled where a file is shorter (the
          ! reference's work array holds zeros - or stale values - there) and cut where it is longer, as monortm_amd/xsec.py
          ! flatten() does.  (Round 4 packed each spectrum with its own count: a shorter earlier file made the kernel read the
          ! tail of that spectrum from the next file's values.)
          nlast = 0
          DO itp = 1, NTEMPF(isr, ixm)
             OPEN (iu, FILE=TRIM(XSFILE(itp, isr, ixm)), FORM='FORMATTED', STATUS='OLD', IOSTAT=ios)
             IF (ios /= 0) THEN
                WRITE (*, '(3a)') ' monortm_hip: cross-section file ', TRIM(XSFILE(itp, isr, ixm)), ' cannot be opened'
                ERROR STOP 1
             END IF
             READ (iu, '(A10,2F10.4,I10,3G10.3,3A10)') amol, amolv1, amolv2, npts, tx, pres, smax, source
             CLOSE (iu)
             nlast = npts
          END DO
          DO itp = 1, NTEMPF(isr, ixm)
             OPEN (iu, FILE=TRIM(XSFILE(itp, isr, ixm)), FORM='FORMATTED', STATUS='OLD', IOSTAT=ios)
             IF (ios /= 0) THEN
                WRITE (*, '(3a)') ' monortm_hip: cross-section file ', TRIM(XSFILE(itp, isr, ixm)), ' cannot be opened'
                ERROR STOP 1
             END IF
             READ (iu, '(A10,2F10.4,I10,3G10.3,3A10)') amol, amolv1, amolv2, npts, tx, pres, smax, source
             IF (pos + MAX(npts, nlast) > cap) THEN
                ALLOCATE (buf(pos))
                buf = pool(1:pos)
                DEALLOCATE (pool)
                cap = 2*(pos + MAX(npts, nlast))
                ALLOCATE (pool(cap))
                pool(1:pos) = buf
                DEALLOCATE (buf)
             END IF
             READ (iu, *) (pool(pos + k), k=1, npts)   ! (a longer file is read whole: its tail is overwritten by the next spectrum)
             CLOSE (iu)
             IF (npts < nlast) pool(pos + npts + 1:pos + nlast) = 0
             tmp(itp, ir) = tx
             IF (source(3) == '      TORR') THEN
                prs(itp, ir) = pres*(1013.0_C_DOUBLE/760)        ! PTORMB (src/monortm_sub.F90:1626)
             ELSE
                prs(itp, ir) = pres
             END IF
             offs(itp, ir) = pos
             pos = pos + nlast
             ! (the reference takes V1, V2 and the number of points of the GRID from the LAST file it read, :1663-1666,:1709)
             reg(7, ir) = amolv1
             reg(8, ir) = amolv2
             reg(4, ir) = npts
          END DO
          ! the FSCDXS bounds decide whether the region is processed at all (:1645)
          reg(2, ir) = V1FX(isr, ixm)
          reg(3, ir) = V2FX(isr, ixm)
          reg(1, ir) = ixm - 1
          reg(5, ir) = NTEMPF(isr, ixm)
          reg(6, ir) = XDOPLR(isr, ixm)
       END DO
    END DO
    rc = monortm_hip_xsec_tables(ctx, INT(IXMOLS, C_INT), INT(nreg, C_INT), reg, tmp, prs, offs, pool, pos)
    IF (rc /= 0) CALL hip_fail('MONORTM_XSEC_SUB (monortm_hip_xsec_tables)', rc)
    last_ctx = ctx
    last_ixmols = IXMOLS
    last_nspecr = NSPECR
    last_ntempf = NTEMPF
    last_v1 = V1FX
    last_v2 = V2FX
    last_dop = XDOPLR
    last_files = XSFILE
  END SUBROUTINE xsec_tables_to_device

END MODULE xsec_hip
