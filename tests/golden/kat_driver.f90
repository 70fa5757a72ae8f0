! TEST INFRASTRUCTURE (own code): known-answer driver for the small functions on the MODM / RTM path.
! Linked by tests/golden/make_kat.py against the reference's modules compiled from a SCRATCH copy of src/modm.f90 whose
! PRIVATE statement is removed (SURVEY.md 8(c)(iii): W4, SD_Humlicek, SDVOIGT are module-private in the reference).
! Reads kat_in.bin (stream of REAL*8: count n, then n x 4 arguments per function), writes kat_out.bin.
!   1 W4(x,y)                          modm.f90:1100      -> re, im
!   2 SD_Humlicek(x1,y1,x2,y2)         modm.f90:1150      -> re, im
!   3 SDVOIGT(deltnu,alphal,alphad,sdep) modm.f90:965     -> value
!   4 RADFN(vi,xkt)                    lblrtm_sub.f90:36  -> value
!   5 AtoB(aa,bb,A,B,119), B = table   tips_2003.f90:4610 -> bb
!   6 ODCLW_TKC(wn,temp,clw)           CloudOptProp.f90:29 -> value
!   7 TIPS_2003(39,T,scor), args (T,mol,iso)  tips_2003.f90:2 -> scor(mol,iso) (scor zeroed before the call)
program kat_driver
  use ModmMod
  use CloudOptProp, only: ODCLW_TKC
  implicit none
  integer, parameter :: dp = 8   ! REAL*8 whatever the default-kind flags say (-fdefault-real-8 makes kind(1.0d0) 16)
  real(dp), external :: RADFN
  real(dp) :: cnt, a(4), tab(600), grid(600), bb, vi
  real :: r1, r2, r3, r4, scor(42,9), tlast
  integer :: nmol39
  complex :: z
  integer :: n, i, f, iu, ou
  common /LAMCHN/ r1, r2, r3, r4     ! RADFN declares it (unused there)
  iu = 31; ou = 32
  open (iu, file='kat_in.bin', access='stream', form='unformatted', status='old')
  open (ou, file='kat_out.bin', access='stream', form='unformatted', status='replace')
  do i = 1, 600
     grid(i) = 60.0_dp + 25.0_dp*(i - 1)        ! TIPS temperature grid (tips_2003.f90:312-336)
  end do
  tlast = -1.0
  nmol39 = 39
  do f = 1, 7
     read (iu) cnt
     n = int(cnt)
     if (f == 5) then
        tab = 0
        read (iu) tab(1:119)
     end if
     do i = 1, n
        read (iu) a
        select case (f)
        case (1)
           z = W4(real(a(1)), real(a(2)))
           write (ou) real(real(z), dp), real(aimag(z), dp)
        case (2)
           z = SD_Humlicek(real(a(1)), real(a(2)), real(a(3)), real(a(4)))
           write (ou) real(real(z), dp), real(aimag(z), dp)
        case (3)
           write (ou) real(SDVOIGT(real(a(1)), real(a(2)), real(a(3)), real(a(4))), dp), 0.0_dp
        case (4)
           vi = a(1)
           write (ou) real(RADFN(vi, real(a(2))), dp), 0.0_dp
        case (5)
           call AtoB(real(a(1)), r1, grid, tab, 119)
           bb = r1
           write (ou) bb, 0.0_dp
        case (6)
           vi = a(1)
           write (ou) real(ODCLW_TKC(vi, real(a(2)), real(a(3))), dp), 0.0_dp
        case (7)
           if (real(a(1)) /= tlast) then
              scor = 0.0
              tlast = real(a(1))
              call TIPS_2003(nmol39, tlast, scor)
           end if
           write (ou) real(scor(int(a(2)), int(a(3))), dp), 0.0_dp
        end select
     end do
  end do
  close (iu); close (ou)
end program kat_driver
