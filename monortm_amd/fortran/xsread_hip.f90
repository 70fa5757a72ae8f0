! XSREAD for builds WITHOUT the reference tree (the dump harness and the own driver linked against the drop-in modules).
! Own implementation of what the reference's XSREAD (reference src/monortm_sub.F90:1246-1421, BLOCK DATA BXSECT :1424-1483)
! leaves in COMMON /PATHX/, /XSECTR/, /XSECTF/ for MODM's IXSECT = 1 path: it reads the requested molecule names from unit
! ipf (record 2.2.1), maps them to the cross-section species through their aliases, and scans the master file FSCDXS for the
! spectral regions of those species that overlap [XV1, XV2].  In a real drop-in build the reference's own monortm_sub.F90
! (driver, I/O, XSREAD) is kept and this file is not linked.
!
! FSCDXS: two header lines; per (species, region) one record (A10,2F10.4,F10.8,I5,5X,I5,A1,4X,6A10) = name, V1, V2, DV,
! number of temperatures, format code, format letter, file names in ascending temperature; '*' comment, '%' end.
SUBROUTINE XSREAD(ipf, XV1, XV2)
  USE lblparams, ONLY: MX_XS, MXLAY
  IMPLICIT NONE
  INTEGER, INTENT(IN) :: ipf
  REAL*8, INTENT(IN) :: XV1, XV2
  INTEGER :: IXMAX, IXMOLS, IXINDX(MX_XS)
  REAL :: XAMNT(MX_XS, MXLAY)
  COMMON /PATHX/ IXMAX, IXMOLS, IXINDX, XAMNT
  CHARACTER*10 :: XSFILE, XSNAME, ALIAS
  COMMON /XSECTF/ XSFILE(6, 5, MX_XS), XSNAME(MX_XS), ALIAS(4, MX_XS)
  REAL*8 :: V1FX, V2FX          ! (IMPLICIT REAL*8 (V) in the reference, src/monortm_sub.F90:1249,:1574)
  REAL :: DVFX, WXM, XSMASS, XDOPLR
  INTEGER :: NTEMPF, NSPECR, IXFORM, NUMXS, IXSBIN
  COMMON /XSECTR/ V1FX(5, MX_XS), V2FX(5, MX_XS), DVFX(5, MX_XS), WXM(MX_XS), NTEMPF(5, MX_XS), NSPECR(MX_XS), &
       IXFORM(5, MX_XS), XSMASS(MX_XS), XDOPLR(5, MX_XS), NUMXS, IXSBIN
  LOGICAL, SAVE :: tables_set = .FALSE.
  CHARACTER*120 :: rec
  CHARACTER*10 :: xname, files(6)
  CHARACTER*1 :: cfrm
  REAL*8 :: v1x, v2x
  REAL :: dvx
  INTEGER :: i, j, k, ntemp, ifrm, ios, iu, n
  LOGICAL :: found(MX_XS), hit

  IF (.NOT. tables_set) THEN
     CALL set_species()
     tables_set = .TRUE.
  END IF
  IXMAX = MX_XS
  XSNAME = ' '
  IF (IXMOLS > 7) THEN
     READ (ipf, '(7A10)') (XSNAME(i), i=1, 7)
     READ (ipf, '(8A10)') (XSNAME(i), i=8, IXMOLS)
  ELSE
     READ (ipf, '(7A10)') (XSNAME(i), i=1, IXMOLS)
  END IF
  DO i = 1, IXMOLS
     XSNAME(i) = ADJUSTL(XSNAME(i))
     IXINDX(i) = 0
     DO j = 1, MX_XS
        IF (ANY(XSNAME(i) == ALIAS(:, j))) THEN
           IXINDX(i) = j
           EXIT
        END IF
     END DO
     IF (IXINDX(i) == 0) THEN
        WRITE (*, '(3a)') '  THE NAME: ', XSNAME(i), ' IS NOT ONE OF THE CROSS SECTION MOLECULES. CHECK THE SPELLING.'
        STOP 'STOPPED IN XSREAD'
     END IF
  END DO
  found = .FALSE.
  iu = 8
  OPEN (iu, FILE='FSCDXS', STATUS='OLD', FORM='FORMATTED', IOSTAT=ios)
  IF (ios /= 0) STOP 'FSCDXS does not exist - XSREAD'
  READ (iu, '(A)') rec
  READ (iu, '(A)') rec
  NUMXS = IXMOLS
  DO
     READ (iu, '(A120)', IOSTAT=ios) rec
     IF (ios /= 0) EXIT
     IF (rec(1:1) == '*') CYCLE
     IF (rec(1:1) == '%') EXIT
     files = ' '
     READ (rec, '(A10,2F10.4,F10.8,I5,5X,I5,A1,4X,6A10)') xname, v1x, v2x, dvx, ntemp, ifrm, cfrm, (files(k), k=1, MIN(ntemp, 6))
     xname = ADJUSTL(xname)
     DO i = 1, IXMOLS
        hit = ANY(xname == ALIAS(:, IXINDX(i)))
        IF (hit) THEN
           found(i) = .TRUE.
           IF (v2x > XV1 .AND. v1x < XV2) THEN
              ! (the reference ADDS to NSPECR on every call, src/monortm_sub.F90:1365 - a second XSREAD of a run doubles
              ! the regions; reproduced as is)
              NSPECR(i) = NSPECR(i) + 1
              ! the reference tests .GT. 6 (src/monortm_sub.F90:1369) although V1FX / V2FX / NTEMPF / XDOPLR / IXFORM hold FIVE
              ! regions per molecule: a sixth one would be written into the next molecule's slots.  Stop where the arrays end.
              IF (NSPECR(i) > 5) STOP ' XSREAD - NSPECR .GT. 5 (the region tables of COMMON /XSECTR/ hold five per molecule)'
              n = NSPECR(i)
              IXFORM(n, i) = 91
              IF (ifrm == 86) IXFORM(n, i) = ifrm
              IF (cfrm /= 'N') IXFORM(n, i) = IXFORM(n, i) + 100
              IF (cfrm == 'F') IXFORM(n, i) = -IXFORM(n, i)
              NTEMPF(n, i) = ntemp
              V1FX(n, i) = v1x
              V2FX(n, i) = v2x
              ! 3.58115E-07 = SQRT(2 LOG(2) AVOGAD BOLTZ / CLIGHT**2); Doppler width at 296 K at the centre of the region
              XDOPLR(n, i) = 3.58115E-07*(0.5*(v1x + v2x))*SQRT(296.0/XSMASS(IXINDX(i)))
              DO k = 1, ntemp
                 XSFILE(k, n, i) = files(k)
              END DO
           END IF
        END IF
     END DO
  END DO
  CLOSE (iu)
  DO i = 1, IXMOLS
     IF (.NOT. found(i)) THEN
        WRITE (*, '(3a)') '******* MOLECULE SELECTED -', XSNAME(i), '- IS NOT FOUND ON FILE FSCDXS *******'
        STOP ' IXFLAG - XSREAD '
     END IF
  END DO

CONTAINS

  SUBROUTINE set_species()   ! names, aliases and molecular masses of the cross-section species (15 of the 38 slots are used)
    INTEGER :: m
    ALIAS = ' ZZZZZZZZ '
    XSMASS = 0.0
    V1FX = 0.0; V2FX = 0.0; DVFX = 0.0; WXM = 0.0
    NTEMPF = 0; NSPECR = 0; IXFORM = 0; NUMXS = 0
    CALL sp(1, 'CLONO2', 'CLNO3', ' ', ' ', 97.46)
    CALL sp(2, 'HNO4', ' ', ' ', ' ', 79.01)
    CALL sp(3, 'CHCL2F', 'CFC21', 'CFC21', 'F21', 102.92)
    CALL sp(4, 'CCL4', ' ', ' ', ' ', 153.82)
    CALL sp(5, 'CCL3F', 'CFCL3', 'CFC11', 'F11', 137.37)
    CALL sp(6, 'CCL2F2', 'CF2CL2', 'CFC12', 'F12', 120.91)
    CALL sp(7, 'C2CL2F4', 'C2F4CL2', 'CFC114', 'F114', 170.92)
    CALL sp(8, 'C2CL3F3', 'C2F3CL3', 'CFC113', 'F113', 187.38)
    CALL sp(9, 'N2O5', ' ', ' ', ' ', 108.01)
    CALL sp(10, 'HNO3', ' ', ' ', ' ', 63.01)
    CALL sp(11, 'CF4', ' ', 'CFC14', 'F14', 88.00)
    CALL sp(12, 'CHCLF2', 'CHF2CL', 'CFC22', 'F22', 86.47)
    CALL sp(13, 'CCLF3', ' ', 'CFC13', 'F13', 104.46)
    CALL sp(14, 'C2CLF5', ' ', 'CFC115', 'F115', 154.47)
    CALL sp(15, 'NO2', ' ', ' ', ' ', 45.99)
    m = 0
  END SUBROUTINE set_species

  SUBROUTINE sp(m, a1, a2, a3, a4, mass)
    INTEGER, INTENT(IN) :: m
    CHARACTER(*), INTENT(IN) :: a1, a2, a3, a4
    REAL, INTENT(IN) :: mass
    ALIAS(1, m) = a1
    IF (LEN_TRIM(a2) > 0) ALIAS(2, m) = a2
    IF (LEN_TRIM(a3) > 0) ALIAS(3, m) = a3
    IF (LEN_TRIM(a4) > 0) ALIAS(4, m) = a4
    XSMASS(m) = mass
  END SUBROUTINE sp

END SUBROUTINE XSREAD
