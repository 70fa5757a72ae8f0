! Continuum scale-factor record passed to MODM (reference src/CntnmFactors.f90:17-19 defines the
! seven default-REAL components; :141-186 the ICNTNM presets).  In a real drop-in build the
! reference's own CntnmFactors.f90 is kept; this stand-alone version only exists so that the shim
! and its test harness can be built without the reference tree.
MODULE CntnmFactors
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: CntnmFactors_t, applyCntnmCombo

  TYPE CntnmFactors_t
     REAL :: xself, xfrgn, xco2c, xo3cn, xo2cn, xn2cn, xrayl
  END TYPE CntnmFactors_t

CONTAINS

  ! ICNTNM presets of record 1.2: 0 none, 1 all, 2 no self, 3 no foreign, 4 neither, 5 no Rayleigh,
  ! 6 user supplied (factors left untouched)
  SUBROUTINE applyCntnmCombo(ICNTNM, f)
    INTEGER, INTENT(IN) :: ICNTNM
    TYPE(CntnmFactors_t), INTENT(INOUT) :: f
    IF (ICNTNM == 6) RETURN
    IF (ICNTNM < 0 .OR. ICNTNM > 6) THEN
       PRINT *, 'err:[CntnmFactors::applyCntnmCombo] Invalid ICNTNM:', ICNTNM
       STOP
    END IF
    f = CntnmFactors_t(1., 1., 1., 1., 1., 1., 1.)
    IF (ICNTNM == 0) f = CntnmFactors_t(0., 0., 0., 0., 0., 0., 0.)
    IF (ICNTNM == 2 .OR. ICNTNM == 4) f%xself = 0.
    IF (ICNTNM == 3 .OR. ICNTNM == 4) f%xfrgn = 0.
    IF (ICNTNM == 5) f%xrayl = 0.
  END SUBROUTINE applyCntnmCombo

END MODULE CntnmFactors
