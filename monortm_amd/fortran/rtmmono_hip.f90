! Drop-in replacement of the reference's MODULE RTMmono (reference src/RTMmono.f90): same module name,
! same PUBLIC names (RTM, calctmr, NWNMX) and argument lists, so PROGRAM MONORTM's calls
! (reference src/monortm.f90:567, :573-574) compile unchanged.  The arithmetic runs on the MI355X through
! the C ABI entry monortm_hip_rtm (include/monortm_hip.h).
MODULE RTMmono
  USE, INTRINSIC :: ISO_C_BINDING
  USE monortm_hip_c
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: RTM, calctmr
  INTEGER, PARAMETER, PUBLIC :: NWNMX = 80000          ! reference src/RTMmono.f90:10

  ! compact staging copies in the caller's REAL kind, kept between calls (RTM and CALCTMR of one profile share them)
  REAL(hreal), ALLOCATABLE, SAVE :: o8(:, :), t8(:), tz8(:), em8(:), rf8(:), r(:, :)
  REAL(hreal), ALLOCATABLE, TARGET, SAVE :: tm8(:)

CONTAINS

  SUBROUTINE stage(nw, nl)
    INTEGER, INTENT(IN) :: nw, nl
    IF (ALLOCATED(o8)) THEN
       IF (ANY(SHAPE(o8) /= (/nw, nl/))) DEALLOCATE (o8, t8, tz8, em8, rf8, r, tm8)
    END IF
    IF (.NOT. ALLOCATED(o8)) ALLOCATE (o8(nw, nl), t8(nl), tz8(0:nl), em8(nw), rf8(nw), r(nw, 5), tm8(nw))
  END SUBROUTINE stage

  SUBROUTINE RTM(IOUT, IRT, NWN, WN, NLAY, T, TZ, O, TMPSFC, RUP, TRTOT, RDN, REFLC, EMISS, RAD, TB, IDU)
    USE lblparams, ONLY: MXLAY
    INTEGER NWN, NLAY, IRT, IOUT, IDU
    REAL*8 WN(NWNMX)
    REAL TMPSFC
    REAL O(:, :)
    REAL T(MXLAY), TZ(0:MXLAY)
    REAL, DIMENSION(:) :: RAD, EMISS, REFLC, RUP, TRTOT, TB, RDN
    REAL(hreal) :: ts8(1)
    INTEGER(C_INT) :: rc, nl(1), ir(1)

    IF (IDU .NE. 1) STOP 'ERROR IN IDU. OPTION NOT SUPPORTED YET'      ! reference RTMmono.f90:173
    CALL hip_require_ctx()
    CALL stage(NWN, NLAY)
    o8 = O(1:NWN, 1:NLAY)
    t8 = T(1:NLAY)
    tz8 = TZ(0:NLAY)
    em8 = EMISS(1:NWN)
    rf8 = REFLC(1:NWN)
    ts8(1) = TMPSFC
    nl(1) = INT(NLAY, C_INT)
    ir(1) = INT(IRT, C_INT)
    IF (IRT .EQ. 3 .OR. IRT .EQ. 2) THEN                                  ! reference RTMmono.f90:113-120
       PRINT *, 'NB: for Downwelling or Limb Calculations the Boundary is ', &
            'Internally Set to the Cosmic Value: 2.75K'
    END IF
    r = 0
    rc = monortm_hip_rtm(hip_ctx, 1_C_INT, INT(NWN, C_INT), WN, nl, INT(NLAY, C_INT), ir, INT(IOUT, C_INT), t8, tz8, o8, &
         ts8, em8, rf8, r(:, 1), r(:, 2), r(:, 3), r(:, 4), r(:, 5), C_NULL_PTR)
    IF (rc /= 0) CALL hip_fail('RTM', rc)
    RUP(1:NWN) = r(:, 1)
    RDN(1:NWN) = r(:, 2)
    TRTOT(1:NWN) = r(:, 3)
    RAD(1:NWN) = r(:, 4)
    IF (IOUT .EQ. 1) TB(1:NWN) = r(:, 5)
    TMPSFC = ts8(1)          ! the reference overwrites TMPSFC for IRT = 2,3 (RTMmono.f90:122)
  END SUBROUTINE RTM

  SUBROUTINE calctmr(nlayrs, nwn, wn, T, tz, O, tmr)
    USE lblparams, ONLY: MXLAY
    REAL*8 wn(NWNMX)
    REAL t(MXLAY), tz(0:MXLAY), o(:, :)
    INTEGER nlayrs, nwn
    REAL tmr(:)
    REAL(hreal) :: ts8(1)
    INTEGER(C_INT) :: rc, nl(1), ir(1)

    CALL hip_require_ctx()
    CALL stage(nwn, nlayrs)
    o8 = o(1:nwn, 1:nlayrs)
    t8 = t(1:nlayrs)
    tz8 = tz(0:nlayrs)
    em8 = 1
    rf8 = 0
    ts8(1) = 2.75
    nl(1) = INT(nlayrs, C_INT)
    ir(1) = 3_C_INT            ! the mean radiating temperature is a downwelling quantity (RTMmono.f90:254-255)
    rc = monortm_hip_rtm(hip_ctx, 1_C_INT, INT(nwn, C_INT), wn, nl, INT(nlayrs, C_INT), ir, 0_C_INT, t8, tz8, o8, ts8, &
         em8, rf8, r(:, 1), r(:, 2), r(:, 3), r(:, 4), r(:, 5), C_LOC(tm8))
    IF (rc /= 0) CALL hip_fail('CALCTMR', rc)
    tmr(1:nwn) = tm8
  END SUBROUTINE calctmr

END MODULE RTMmono
