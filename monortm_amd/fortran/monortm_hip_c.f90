! ISO_C_BINDING interfaces of the C ABI (include/monortm_hip.h) and the per-process context.
!
! The reference keeps its line list in module memory and loads it on the first MODM call
! (reference src/modm.f90:161-163,187-190); the shim keeps one opaque context the same way.
MODULE monortm_hip_c
  USE, INTRINSIC :: ISO_C_BINDING
  IMPLICIT NONE
  PUBLIC

  TYPE(C_PTR), SAVE :: hip_ctx = C_NULL_PTR

  ! The caller's default REAL: 8 bytes in the reference's "dbl" build (-fdefault-real-8, build/makefile.common:195-198),
  ! 4 bytes in its "sgl" build.  The C ABI takes the arrays in that kind (monortm_real, real_kind fixed at init).
  INTEGER, PARAMETER :: hreal = KIND(1.0)
  INTEGER(C_INT), PARAMETER :: hip_real_kind = INT(STORAGE_SIZE(1.0) / 8, C_INT)

  INTERFACE
     INTEGER(C_INT) FUNCTION monortm_hip_init(tape3_path, v1, v2, icp, real_kind, device, ctx) &
          BIND(C, NAME='monortm_hip_init')
       IMPORT :: C_INT, C_DOUBLE, C_CHAR, C_PTR
       CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: tape3_path
       REAL(C_DOUBLE), VALUE :: v1, v2
       INTEGER(C_INT), VALUE :: icp, real_kind, device
       TYPE(C_PTR), INTENT(OUT) :: ctx
     END FUNCTION monortm_hip_init

     ! the same on ngpu devices (ngpu <= 0: all visible): the host-buffer calls shard the batch of profiles over them
     INTEGER(C_INT) FUNCTION monortm_hip_init_multi(tape3_path, v1, v2, icp, real_kind, ngpu, ctx) &
          BIND(C, NAME='monortm_hip_init_multi')
       IMPORT :: C_INT, C_DOUBLE, C_CHAR, C_PTR
       CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: tape3_path
       REAL(C_DOUBLE), VALUE :: v1, v2
       INTEGER(C_INT), VALUE :: icp, real_kind, ngpu
       TYPE(C_PTR), INTENT(OUT) :: ctx
     END FUNCTION monortm_hip_init_multi

     INTEGER(C_INT) FUNCTION monortm_hip_device_count(ctx) BIND(C, NAME='monortm_hip_device_count')
       IMPORT :: C_INT, C_PTR
       TYPE(C_PTR), VALUE :: ctx
     END FUNCTION monortm_hip_device_count

     SUBROUTINE monortm_hip_finalize(ctx) BIND(C, NAME='monortm_hip_finalize')
       IMPORT :: C_PTR
       TYPE(C_PTR), VALUE :: ctx
     END SUBROUTINE monortm_hip_finalize

     INTEGER(C_INT) FUNCTION monortm_hip_has_lines(ctx) BIND(C, NAME='monortm_hip_has_lines')
       IMPORT :: C_INT, C_PTR
       TYPE(C_PTR), VALUE :: ctx
     END FUNCTION monortm_hip_has_lines

     INTEGER(C_INT) FUNCTION monortm_hip_xsec_regions(ctx) BIND(C, NAME='monortm_hip_xsec_regions')
       IMPORT :: C_INT, C_PTR
       TYPE(C_PTR), VALUE :: ctx
     END FUNCTION monortm_hip_xsec_regions

     TYPE(C_PTR) FUNCTION monortm_hip_last_error(ctx) BIND(C, NAME='monortm_hip_last_error')
       IMPORT :: C_PTR
       TYPE(C_PTR), VALUE :: ctx
     END FUNCTION monortm_hip_last_error

     INTEGER(C_INT) FUNCTION monortm_hip_modm(ctx, nprof, nwn, wn, dvset, nlay, nlay_max, nmol, P, T, CLW, WKL, &
          WBRODL, cntnm_fac, sclcpl, sclhw, y0res, ibrd, ixsect, O, O_BY_MOL, OC, O_CLW) &
          BIND(C, NAME='monortm_hip_modm')
       IMPORT :: C_INT, C_DOUBLE, C_PTR, hreal
       TYPE(C_PTR), VALUE :: ctx
       INTEGER(C_INT), VALUE :: nprof, nwn, nlay_max, nmol, ibrd, ixsect
       REAL(C_DOUBLE), VALUE :: dvset, sclcpl, sclhw, y0res
       INTEGER(C_INT), INTENT(IN) :: nlay(*)
       REAL(C_DOUBLE), INTENT(IN) :: wn(*), cntnm_fac(7)
       REAL(hreal), INTENT(IN) :: P(*), T(*), CLW(*), WKL(*), WBRODL(*)
       REAL(hreal), INTENT(OUT) :: O(*), O_BY_MOL(*), OC(*), O_CLW(*)
     END FUNCTION monortm_hip_modm

     ! cross-section molecules (IXSECT = 1): the parsed tables, then MODM with XAMNT / ODXSEC as arguments
     ! one process per GPU: the single gather of a profile-sharded job over RCCL (include/monortm_hip.h)
     INTEGER(C_INT) FUNCTION monortm_hip_comm_unique_id(id128) BIND(C, NAME='monortm_hip_comm_unique_id')
       IMPORT :: C_INT, C_CHAR
       CHARACTER(KIND=C_CHAR), INTENT(OUT) :: id128(128)
     END FUNCTION monortm_hip_comm_unique_id
     INTEGER(C_INT) FUNCTION monortm_hip_comm_init(ctx, world, rank, id128) BIND(C, NAME='monortm_hip_comm_init')
       IMPORT :: C_INT, C_PTR, C_CHAR
       TYPE(C_PTR), VALUE :: ctx
       INTEGER(C_INT), VALUE :: world, rank
       CHARACTER(KIND=C_CHAR), INTENT(IN) :: id128(128)
     END FUNCTION monortm_hip_comm_init
     INTEGER(C_INT) FUNCTION monortm_hip_gather_dev(ctx, send, bytes, recv, root, stream) BIND(C, NAME='monortm_hip_gather_dev')
       IMPORT :: C_INT, C_PTR, C_SIZE_T
       TYPE(C_PTR), VALUE :: ctx, send, recv, stream
       INTEGER(C_SIZE_T), VALUE :: bytes
       INTEGER(C_INT), VALUE :: root
     END FUNCTION monortm_hip_gather_dev
     INTEGER(C_INT) FUNCTION monortm_hip_xsec_tables(ctx, nxs, nreg, reg, temps, pres_mb, offs, pool, npool) &
          BIND(C, NAME='monortm_hip_xsec_tables')
       IMPORT :: C_INT, C_DOUBLE, C_PTR, C_LONG_LONG
       TYPE(C_PTR), VALUE :: ctx
       INTEGER(C_INT), VALUE :: nxs, nreg
       REAL(C_DOUBLE), INTENT(IN) :: reg(*), temps(*), pres_mb(*), pool(*)
       INTEGER(C_LONG_LONG), INTENT(IN) :: offs(*)
       INTEGER(C_LONG_LONG), VALUE :: npool
     END FUNCTION monortm_hip_xsec_tables

     INTEGER(C_INT) FUNCTION monortm_hip_modm_xs(ctx, nprof, nwn, wn, dvset, nlay, nlay_max, nmol, P, T, CLW, WKL, &
          WBRODL, cntnm_fac, sclcpl, sclhw, y0res, ibrd, ixsect, XAMNT, ODXSEC, O, O_BY_MOL, OC, O_CLW) &
          BIND(C, NAME='monortm_hip_modm_xs')
       IMPORT :: C_INT, C_DOUBLE, C_PTR, hreal
       TYPE(C_PTR), VALUE :: ctx
       INTEGER(C_INT), VALUE :: nprof, nwn, nlay_max, nmol, ibrd, ixsect
       REAL(C_DOUBLE), VALUE :: dvset, sclcpl, sclhw, y0res
       INTEGER(C_INT), INTENT(IN) :: nlay(*)
       REAL(C_DOUBLE), INTENT(IN) :: wn(*), cntnm_fac(7)
       REAL(hreal), INTENT(IN) :: P(*), T(*), CLW(*), WKL(*), WBRODL(*), XAMNT(*)
       REAL(hreal), INTENT(OUT) :: ODXSEC(*), O(*), O_BY_MOL(*), OC(*), O_CLW(*)
     END FUNCTION monortm_hip_modm_xs

     INTEGER(C_INT) FUNCTION monortm_hip_rtm(ctx, nprof, nwn, wn, nlay, nlay_max, irt, iout, T, TZ, O, tmpsfc, &
          emiss, reflc, RUP, RDN, TRTOT, RAD, TB, TMR) BIND(C, NAME='monortm_hip_rtm')
       IMPORT :: C_INT, C_DOUBLE, C_PTR, hreal
       TYPE(C_PTR), VALUE :: ctx
       INTEGER(C_INT), VALUE :: nprof, nwn, nlay_max, iout
       INTEGER(C_INT), INTENT(IN) :: nlay(*), irt(*)
       REAL(C_DOUBLE), INTENT(IN) :: wn(*)
       REAL(hreal), INTENT(IN) :: T(*), TZ(*), O(*), emiss(*), reflc(*)
       REAL(hreal), INTENT(INOUT) :: tmpsfc(*)
       REAL(hreal), INTENT(OUT) :: RUP(*), RDN(*), TRTOT(*), RAD(*), TB(*)
       TYPE(C_PTR), VALUE :: TMR      ! monortm_real* or NULL
     END FUNCTION monortm_hip_rtm
  END INTERFACE

CONTAINS

  ! The reference has no status codes: every failure is a STOP with console text
  ! (e.g. src/lnfl_mod.f90:131-132, src/tips_2003.f90:277, src/modm.f90:1062).
  SUBROUTINE hip_fail(where, rc)
    CHARACTER(LEN=*), INTENT(IN) :: where
    INTEGER(C_INT), INTENT(IN) :: rc
    TYPE(C_PTR) :: p
    CHARACTER(KIND=C_CHAR), POINTER :: s(:)
    INTEGER :: n
    p = monortm_hip_last_error(hip_ctx)
    WRITE (*, '(a,a,a,i3)') ' monortm_hip: ', where, ' failed, status', rc
    IF (C_ASSOCIATED(p)) THEN
       CALL C_F_POINTER(p, s, [512])
       n = 0
       DO WHILE (n < 512)
          IF (s(n + 1) == C_NULL_CHAR) EXIT
          n = n + 1
       END DO
       IF (n > 0) WRITE (*, '(1x,512a1)') s(1:n)
    END IF
    WRITE (*, '(a)') ' monortm_hip error'
    ERROR STOP 1   ! non-zero exit status: a script (or test) driving the program sees the failure
  END SUBROUTINE hip_fail

  ! make sure a context exists (RTM / CALCTMR may be called without a preceding MODM)
  SUBROUTINE hip_require_ctx()
    INTEGER(C_INT) :: rc
    CHARACTER(KIND=C_CHAR) :: empty(1)
    IF (C_ASSOCIATED(hip_ctx)) RETURN
    empty(1) = C_NULL_CHAR
    rc = monortm_hip_init(empty, 0.0_C_DOUBLE, 0.0_C_DOUBLE, 1_C_INT, hip_real_kind, -1_C_INT, hip_ctx)
    IF (rc /= 0) CALL hip_fail('monortm_hip_init', rc)
  END SUBROUTINE hip_require_ctx

END MODULE monortm_hip_c
