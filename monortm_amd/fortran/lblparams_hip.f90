! Array-dimension parameters the MODM / RTM boundary is declared with.  Same public names and
! values as the reference's module (reference src/lblparams.f90:28-35); only the ones the
! boundary needs are defined.  In a real drop-in build the reference's own lblparams.f90 is kept.
MODULE lblparams
  IMPLICIT NONE
  INTEGER, PARAMETER :: MXMOL = 39
  INTEGER, PARAMETER :: MXFSC = 600, MXLAY = MXFSC + 3
  INTEGER, PARAMETER :: N_ABSRB = 5050
  INTEGER, PARAMETER :: MX_XS = 38
  PUBLIC
END MODULE lblparams
