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
!   8 HALFWHM_D(mol,iso,xnu,T)         modm.f90:442       -> value (masses: the reference's own COMMON /ISVECT/)
!   9 bb_fn(v,fbeta)                   RTMmono.f90:223    -> value (PRIVATE there too: scratch copy as for modm.f90)
! from here on 12 arguments per row:
!  10 INTENS(T,S0s,Es,RADCT,T0,Xnus,STILD,XIPSF)          modm.f90:860 -> STILD
!  11 HALFWHM_C(AF,AS,RT,XTILD,RHORAT,MOL,rho_molec,...)  modm.f90:833 -> value (rho_molec(MOL) = arg 7, no species data)
!  12 LSF_LORTZ(XF,RP,RP2,AIP,BIP,HWHM,WN,Xnu,SLS,MOL)    modm.f90:706 -> SLS
!  13 LSF_SDVOIGT(XF,RP,RP2,AIP,BIP,HWHM,WN,Xnu,SLS,AD,MOL,SDEP) modm.f90:567 -> SLS
program kat_driver
  use ModmMod
  use CloudOptProp, only: ODCLW_TKC
  use RTMmono, only: bb_fn
  implicit none
  integer, parameter :: dp = 8   ! REAL*8 whatever the default-kind flags say (-fdefault-real-8 makes kind(1.0d0) 16)
  real(dp), external :: RADFN
  real(dp) :: cnt, a(4), w(12), tab(600), grid(600), bb, vi, vj
  real :: stild, sls, asv, rho(7), zh(7), zt(7)
  integer*4 :: zf(7)
  integer :: molk
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
  do f = 1, 13
     read (iu) cnt
     n = int(cnt)
     if (f == 5) then
        tab = 0
        read (iu) tab(1:119)
     end if
     do i = 1, n
        if (f >= 10) then
           read (iu) w
        else
           read (iu) a
        end if
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
        case (8)
           vi = a(3)
           write (ou) real(HALFWHM_D(int(a(1)), int(a(2)), vi, real(a(4))), dp), 0.0_dp
        case (9)
           vi = a(1)
           write (ou) real(bb_fn(vi, real(a(2))), dp), 0.0_dp
        case (10)
           vi = w(6)
           call INTENS(real(w(1)), real(w(2)), real(w(3)), real(w(4)), real(w(5)), vi, stild, real(w(7)))
           write (ou) real(stild, dp), 0.0_dp
        case (11)
           molk = int(w(6))
           rho = 0.0; zf = 0; zh = 0.0; zt = 0.0
           if (molk >= 1 .and. molk <= 7) rho(molk) = real(w(7))
           asv = real(w(2))
           write (ou) real(HALFWHM_C(real(w(1)), asv, real(w(3)), real(w(4)), real(w(5)), molk, rho, zf, zh, zt), dp), 0.0_dp
        case (12)
           vi = w(7); vj = w(8)
           call LSF_LORTZ(real(w(1)), real(w(2)), real(w(3)), real(w(4)), real(w(5)), real(w(6)), vi, vj, sls, int(w(9)))
           write (ou) real(sls, dp), 0.0_dp
        case (13)
           vi = w(7); vj = w(8)
           call LSF_SDVOIGT(real(w(1)), real(w(2)), real(w(3)), real(w(4)), real(w(5)), real(w(6)), vi, vj, sls, real(w(9)), &
                            int(w(10)), real(w(11)))
           write (ou) real(sls, dp), 0.0_dp
        end select
     end do
  end do
  close (iu); close (ou)
end program kat_driver
