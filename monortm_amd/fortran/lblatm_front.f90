! lblatm_front - own layering front end for IATM = 1 decks (SURVEY.md section 8 row f2): from records 3.1 - 3.3B of MONORTM.IN
! to the layer quantities the hot path needs (mean pressure and temperature, column amounts per molecule, broadening
! gas column, level temperatures), for the built-in model atmospheres (MODEL 1-6) and slant / vertical paths given by
! H1, H2, ANGLE (ITYPE 2, case 2A) or H1, ANGLE to space (ITYPE 3, case 3A), with automatic layering (IBMAX = 0) or
! boundary altitudes on record 3.3B.
!
! What it reproduces of the reference (src/lblatm.f90; all REAL*8, the "dbl" build): the profile set-up of MDLATM
! (:2918-3041) on the AFGL tables (atm_models_data), the LOWTRAN6 refractivity (:1118-1134), the path reduction of
! FSCGEO / FNDHMN (:4140-4450, :4678-4803), the boundary selection of AUTLAY / HALFWD (:5582-5773, :5830-5880), the merge
! of boundaries and model levels of AMERGE (:5111-5287), the refracted ray trace with its sub-layer quadrature and the
! exponential / linear density interpolation of RFPATH / ALAYER (:4917-5110, :5289-5580), and the packing into output
! layers of FPACK (:5882-6042).  Written from those algorithms in an own structure: one derived type per stage instead of
! COMMON blocks, no printed report (TAPE6 / TAPE7 are not produced).
!
! MODEL = 0: a user-supplied profile on altitude levels (records 3.4 - 3.6.n) is read as NSMDL / RDUNIT read it (:3044-3160,
! :3222-3398), with the unit keys of records 3.5 (pressure mb / atm / torr, temperature K / C, molecules ppmv, number
! density, mass mixing ratio, mass density, partial pressure, and for water dew point or relative humidity: CONVRT / WATVAP
! :3884-4138) and the "default to model atmosphere 1-6" keys (4-point Lagrange interpolation of DEFALT, :3489-3686).
!
! Not covered (the reference's own LBLATM linked to the drop-in modules serves those decks, INTEGRATION.md): profiles or
! boundaries on pressure levels (IMMAX < 0, IBMAX < 0: CMPALT hydrostatics), horizontal paths (ITYPE 1), the RANGE / BETA
! path cases 2B-2D and 3B, cross sections.
module lblatm_front
  use atm_models_data
  implicit none
  private
  public :: atm_request, atm_layers, read_atm_request, build_atm_layers
  integer, parameter :: dp = selected_real_kind(15)
  integer, parameter :: MXMOLF = 39, MXBND = 600, MXPTH = 6600

  ! constants: src/PhysConstants.f90:19-39 and BLOCK DATA ATMCON (src/lblatm.f90:1725-1806)
  real(dp), parameter :: PI = 3.1415926535898_dp, AVOGAD = 6.02214199E+23_dp, GASCON = 8.314472E+07_dp, ALOSMT = 2.6867775E+19_dp
  real(dp), parameter :: CLIGHT = 2.99792458E+10_dp
  real(dp), parameter :: DELTAS = 5.0_dp, PZERO = 1013.25_dp, TZERO = 273.15_dp, ALZERO = 0.04_dp, AVMWT = 36.0_dp
  real(dp), parameter :: AIRMWT = 28.964_dp                                          ! src/PlanetEarth.f90:19-20
  real(dp), parameter :: AMWT(39) = (/ 18.015_dp, 44.010_dp, 47.998_dp, 44.01_dp, 28.011_dp, 16.043_dp, 31.999_dp, 30.01_dp, &
       64.06_dp, 46.01_dp, 17.03_dp, 63.01_dp, 17.00_dp, 20.01_dp, 36.46_dp, 80.92_dp, 127.91_dp, 51.45_dp, 60.08_dp, 30.03_dp, &
       52.46_dp, 28.014_dp, 27.03_dp, 50.49_dp, 34.01_dp, 26.03_dp, 30.07_dp, 34.00_dp, 66.01_dp, 146.05_dp, 34.08_dp, 46.03_dp, &
       33.00_dp, 15.99_dp, 98.0_dp, 30.00_dp, 97.0_dp, 28.05_dp, 32.04_dp /)          ! BLOCK DATA ATMCON, molecular weights

  type atm_request                       ! records 3.1, 3.2, 3.3A / 3.3B
     integer :: model = 0, itype = 0, ibmax = 0, n_zero = 0, noprnt = 0, nmol = 0, ipunch = 0, munits = 0
     real(dp) :: re = 0, hspace = 0, xvbar = 0
     real(dp) :: h1 = 0, h2 = 0, angle = 0, range = 0, beta = 0, hobs = 0
     integer :: len = 0
     real(dp) :: avtrat = 0, tdiff1 = 0, tdiff2 = 0, altd1 = 0, altd2 = 0
     real(dp) :: zbnd(MXBND) = 0
     ! IBMAX < 0 on record 3.1: boundaries (record 3.3B), H1 and H2 are pressures (mb); IMMAX < 0 on record 3.4: the user
     ! profile is given on pressure levels and its altitudes follow from the hydrostatic equation (CMPALT)
     integer :: ibmax_b = 0, immax_b = 0
     real(dp) :: pbnd(MXBND) = 0, ref_lat = 45
     ! MODEL = 0: the user profile of records 3.4 - 3.6.n, already converted to mb, K and number densities (cm-3)
     integer :: immax = 0
     character(len=24) :: hmod = ' '
     real(dp), allocatable :: zm(:), pm(:), tm(:), denm(:, :)
  end type atm_request

  type atm_layers                        ! what COMMON /PATHD/ + /MANE/ hand to the driver (src/monortm.f90:229-230)
     integer :: nlay = 0, nmol = 0
     real(dp) :: angle = 0, h1 = 0, h2 = 0
     real(dp), allocatable :: pbar(:), tbar(:), wbrodl(:), secnta(:)   ! (nlay)
     real(dp), allocatable :: amount(:, :)                             ! (nmol, nlay)
     real(dp), allocatable :: altz(:), pz(:), tz(:)                    ! (0:nlay)
     integer, allocatable :: ipath(:)
     character(len=24) :: hmod = ' '
  end type atm_layers

  type profile                           ! the atmosphere on its own levels (COMMON /CMN/, /DEAMT/)
     integer :: n = 0
     real(dp), allocatable :: z(:), p(:), t(:), rfndx(:), den(:, :)
     real(dp) :: zmax = 0, re = 0
     character(len=24) :: hmod = ' '
  end type profile

contains

  subroutine fail(msg)
    character(len=*), intent(in) :: msg
    write (*, '(a)') ' lblatm_front: '//msg
    stop 1
  end subroutine fail

  ! EXPINT (src/monortm_sub.F90:1213-1223): exponential interpolation, linear if an end value is zero
  pure function expint(x1, x2, a) result(x)
    real(dp), intent(in) :: x1, x2, a
    real(dp) :: x
    if (x1 == 0.0_dp .or. x2 == 0.0_dp) then
       x = x1 + (x2 - x1)*a
    else
       x = x1*(x2/x1)**a
    end if
  end function expint

  ! ------------------------------------------------------------------ input
  subroutine read_atm_request(u, rq)
    integer, intent(in) :: u
    type(atm_request), intent(out) :: rq
    integer :: ios, ifxtyp, ibmax_b, ib
    real(dp) :: dumrd
    character(len=10) :: sref_lat
    read (u, '(7I5,I2,1X,I2,4F10.3,A10)', iostat=ios) rq%model, rq%itype, ibmax_b, rq%n_zero, rq%noprnt, rq%nmol, rq%ipunch, &
         ifxtyp, rq%munits, rq%re, rq%hspace, rq%xvbar, dumrd, sref_lat                         ! record 3.1
    if (ios /= 0) call fail('error reading record 3.1')
    if (dumrd /= 0) call fail('a value has been read for co2mx (record 3.1): option replaced, see the instructions')
    if (rq%model < 0 .or. rq%model > 6) call fail('MODEL must be 0 .. 6')
    if (rq%itype /= 2 .and. rq%itype /= 3) call fail('ITYPE must be 2 or 3 (slant path)')
    if (sref_lat == ' ') then                                  ! src/lblatm.f90:583-588
       rq%ref_lat = 45.0_dp
    else
       read (sref_lat, '(F10.3)', iostat=ios) rq%ref_lat
       if (ios /= 0) call fail('error reading REF_LAT on record 3.1')
    end if
    if (ifxtyp /= 0 .or. rq%munits /= 0) call fail('IFXTYP / MUNITS options are not built into this front end')
    rq%ibmax_b = ibmax_b
    rq%ibmax = abs(ibmax_b)
    if (rq%ibmax > MXBND) call fail('IBMAX exceeds the boundary dimension')
    if (rq%nmol == 0) rq%nmol = 7
    if (rq%model > 0 .and. rq%nmol > 28) call fail('NMOL > 28: no built-in profile beyond molecule 28')
    if (rq%nmol > MXMOLF) call fail('NMOL exceeds 39')
    read (u, '(5F10.4,I5,5X,F10.4)', iostat=ios) rq%h1, rq%h2, rq%angle, rq%range, rq%beta, rq%len, rq%hobs   ! record 3.2
    if (ios /= 0) call fail('error reading record 3.2')
    if (rq%range > 0 .or. rq%beta > 0) call fail('path cases with RANGE or BETA (2B-2D) are not built into this front end')
    if (rq%ibmax == 0) then
       read (u, '(5F10.3)', iostat=ios) rq%avtrat, rq%tdiff1, rq%tdiff2, rq%altd1, rq%altd2     ! record 3.3A
       if (ios /= 0) call fail('error reading record 3.3A')
       if (rq%avtrat == 0) rq%avtrat = 1.5_dp
       if (rq%tdiff1 == 0) rq%tdiff1 = 5.0_dp
       if (rq%tdiff2 == 0) rq%tdiff2 = 8.0_dp
       if (rq%altd2 <= 0 .or. rq%altd2 <= rq%altd1) then
          rq%altd1 = 0
          rq%altd2 = 100
       end if
       if (rq%avtrat <= 1 .or. rq%tdiff1 <= 0 .or. rq%tdiff2 <= 0) call fail('AVTRAT, TDIFF1 or TDIFF2 out of range')
    else if (rq%ibmax_b < 0) then
       read (u, '(8F10.3)', iostat=ios) rq%pbnd(1:rq%ibmax)                                    ! record 3.3B, pressures
       if (ios /= 0) call fail('error reading record 3.3B')
       do ib = 2, rq%ibmax
          if (rq%pbnd(ib) >= rq%pbnd(ib - 1)) call fail('BOUNDARY PRESSURES ARE POSITIVE OR NOT IN DESCENDING ORDER')
       end do
    else
       read (u, '(8F10.3)', iostat=ios) rq%zbnd(1:rq%ibmax)                                    ! record 3.3B
       if (ios /= 0) call fail('error reading record 3.3B')
       do ib = 2, rq%ibmax
          if (rq%zbnd(ib) <= rq%zbnd(ib - 1)) call fail('boundary altitudes not in ascending order')
       end do
    end if
    if (rq%model == 0) call read_user_profile(u, rq)
  end subroutine read_atm_request

  ! ------------------------------------------------------------------ MODEL = 0: NSMDL / RDUNIT / DEFALT / CONVRT / WATVAP
  integer function jou(c)                                 ! unit key -> code (JOU, src/lblatm.f90:3402-3436)
    character(len=1), intent(in) :: c
    character(len=1), parameter :: keys(17) = (/ '1', '2', '3', '4', '5', '6', ' ', 'A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'J' /)
    integer, parameter :: codes(17) = (/ 1, 2, 3, 4, 5, 6, 10, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19 /)
    integer :: i
    jou = 0
    do i = 1, 17
       if (keys(i) == c) jou = codes(i)
    end do
    if (c == 'K') jou = 20
    if (jou == 0) call fail('JOU: invalid unit key "'//c//'" on record 3.5')
  end function jou

  subroutine read_user_profile(u, rq)
    integer, intent(in) :: u
    type(atm_request), intent(inout) :: rq
    integer :: ios, immax_b, im, k, nmol, junitp, junitt, junit(MXMOLF)
    character(len=1) :: jcharp, jchart, jlong, jchar(MXMOLF)
    real(dp) :: wmol(MXMOLF), re
    nmol = rq%nmol
    read (u, '(I5,3A8)', iostat=ios) immax_b, rq%hmod                                           ! record 3.4
    if (ios /= 0) call fail('error reading record 3.4')
    rq%immax_b = immax_b
    rq%immax = abs(immax_b)
    if (rq%immax < 2 .or. rq%immax > 6000) call fail('IMMAX out of range')
    allocate (rq%zm(rq%immax), rq%pm(rq%immax), rq%tm(rq%immax), rq%denm(MXMOLF, rq%immax))
    rq%denm = 0
    do im = 1, rq%immax
       jchar = ' '
       read (u, '(3E10.3,5X,2A1,1X,A1,1X,39A1)', iostat=ios) rq%zm(im), rq%pm(im), rq%tm(im), jcharp, jchart, jlong, &
            jchar(1:MXMOLF)                                                                     ! record 3.5
       if (ios /= 0) call fail('error reading record 3.5')
       junitp = jou(jcharp)
       junitt = jou(jchart)
       do k = 1, nmol
          junit(k) = jou(jchar(k))
       end do
       wmol = 0
       if (jlong == 'L') then
          read (u, '(8E15.8)', iostat=ios) wmol(1:nmol)                                        ! record 3.6
       else if (jlong == ' ') then
          read (u, '(8E10.3)', iostat=ios) wmol(1:nmol)
       else
          call fail('INVALID VALUE FOR JLONG ON RECORD 3.5')
       end if
       if (ios /= 0) call fail('error reading record 3.6')
       ! CHECK: pressure and temperature units
       if (junitp == 11) rq%pm(im) = rq%pm(im)*1013.25_dp
       if (junitp == 12) rq%pm(im) = rq%pm(im)*1013.25_dp/760.0_dp
       if (junitp > 12) call fail('CHECK(P): invalid pressure unit')
       if (junitt == 11) rq%tm(im) = rq%tm(im) + 273.15_dp
       if (junitt > 11) call fail('CHECK(T): invalid temperature unit')
       if (immax_b < 0) then
          if (junitp <= 6) call fail('a pressure-level profile (IMMAX < 0) must state its pressures')
          call default_values_p(rq%pm(im), rq%tm(im), nmol, junitt, junit, wmol)
       else
          call default_values(rq%zm(im), rq%pm(im), rq%tm(im), nmol, junitp, junitt, junit, wmol)
       end if
       call convert_units(rq%pm(im), rq%tm(im), nmol, junit, wmol, rq%denm(:, im))
    end do
    if (immax_b < 0) then                      ! altitudes of the levels from the hydrostatic equation, upwards from ZM(1)
       re = rq%re
       if (re == 0) re = 6371.23_dp            ! (MODEL = 0; src/lblatm.f90:643-647)
       call cmpalt(rq%immax, rq%pm, rq%tm, rq%denm(1, :), rq%zm(1), rq%ref_lat, re, rq%zm)
    end if
    do im = 2, rq%immax
       if (rq%zm(im) <= rq%zm(im - 1)) call fail('INPUT ALTITUDES NOT IN ASCENDING ORDER')
    end do
  end subroutine read_user_profile

  ! CMPALT (src/lblatm.f90:7896-8017): altitudes of pressure levels from the hydrostatic equation with the water vapour
  ! mixing ratio and temperature linear in ln p between levels, gravity at the reference latitude falling off with
  ! altitude, and the compressibility factor of moist air (Ciddor 1996).
  subroutine cmpalt(n, pm, tm, denw, ref_z, ref_lat, re, zmdl)
    integer, intent(in) :: n
    real(dp), intent(in) :: pm(n), tm(n), denw(n), ref_z, ref_lat, re
    real(dp), intent(inout) :: zmdl(n)
    real(dp), parameter :: BOLTZ = 1.3806503E-16_dp, XMASS_H2O = 18.015_dp*1.E-3_dp, XMASS_DRY = AIRMWT*1.E-3_dp
    real(dp), parameter :: CA0 = 1.58123E-6_dp, CA1 = -2.9331E-8_dp, CA2 = 1.1043E-10_dp, CB0 = 5.707E-6_dp, CB1 = -2.051E-8_dp, &
         CC0 = 1.9898E-4_dp, CC1 = -2.376E-6_dp, CD = 1.83E-11_dp, CE = -0.0765E-8_dp
    real(dp) :: h2o_mixrat(n), comp_factor(n), ztemp(n)
    real(dp) :: g0, xmass_ratio, dt, total_air, dry_air, chim, gave, y, chi0, dchi, t0, c1, c2, c3, a, b, alpha, xint_tot, zref
    integer :: i, j
    g0 = 9.80665_dp - 0.02586_dp*cos(2.0_dp*PI*ref_lat/180.0_dp)           ! gravConst, src/PlanetEarth.f90:81
    xmass_ratio = XMASS_H2O/XMASS_DRY
    do j = 1, n
       dt = tm(j) - 273.15_dp
       total_air = pm(j)*1.0E+3_dp/(BOLTZ*tm(j))
       dry_air = total_air - denw(j)
       h2o_mixrat(j) = denw(j)/dry_air
       chim = xmass_ratio*h2o_mixrat(j)
       comp_factor(j) = 1.0_dp - (pm(j)*100/tm(j))*(CA0 + CA1*dt + CA2*dt**2 + (CB0 + CB1*dt)*chim + (CC0 + CC1*dt)*chim**2) + &
            (CD + CE*chim**2)*(pm(j)*100.0_dp/tm(j))**2
    end do
    zref = ref_z
    ztemp(1) = zref*1000.0_dp
    zmdl(1) = zref
    do i = 1, n - 1
       gave = g0*(re/(re + ztemp(i)/1000.0_dp))**2
       y = log(pm(i + 1)/pm(i))
       if (y /= 0) then
          chi0 = h2o_mixrat(i)
          dchi = (h2o_mixrat(i + 1) - h2o_mixrat(i))/y
          t0 = tm(i)
          dt = (tm(i + 1) - tm(i))/y
          c1 = t0 + t0*chi0
          c2 = t0*dchi + dt*chi0 + dt
          c3 = dt*dchi
          b = 1 + xmass_ratio*chi0
          a = xmass_ratio*dchi
          alpha = a/b
          if (abs(alpha*y) >= 0.01_dp) call fail('CMPALT: LAYER TOO THICK')
          xint_tot = c1*y + 0.5_dp*(c2 - c1*alpha)*y**2 + 0.3333_dp*(c3 - c2*alpha + c1*alpha**2)*y**3
          xint_tot = -xint_tot*(GASCON*1.0E-7_dp)/(XMASS_DRY*gave*b)
          ztemp(i + 1) = ztemp(i) + xint_tot*comp_factor(i)
          zmdl(i + 1) = ztemp(i + 1)/1000.0_dp
       else
          ztemp(i + 1) = zmdl(i)*1000.0_dp
          zmdl(i + 1) = zmdl(i)
       end if
    end do
  end subroutine cmpalt

  ! A pressure -> the altitude on the profile's levels (src/lblatm.f90:891-1086): between two levels a blend of the
  ! interpolation in ln p (weight A) and the hydrostatic altitude above the lower level (weight 1 - A), A = (fraction of the
  ! interval in ln p)**3, so that the result joins the levels' own altitudes at both ends
  real(dp) function pressure_to_altitude(n, zm, pm, tm, denw, p, ref_lat, re) result(z)
    integer, intent(in) :: n
    real(dp), intent(in) :: zm(n), pm(n), tm(n), denw(n), p, ref_lat, re
    real(dp) :: ptmp(2), ttmp(2), wvtmp(2), ztmp(2), hip, zint, tip, wvip, ratp, a
    integer :: lip
    do lip = 2, n
       if (p > pm(lip)) exit
    end do
    if (lip > n) lip = n
    if (p == pm(lip - 1)) then
       z = zm(lip - 1)
    else if (p == pm(lip)) then
       z = zm(lip)
    else
       hip = (zm(lip) - zm(lip - 1))/log(pm(lip)/pm(lip - 1))
       zint = zm(lip - 1) + hip*log(p/pm(lip - 1))
       ptmp(1) = pm(lip - 1)
       ztmp(1) = zm(lip - 1)
       ttmp(1) = tm(lip - 1)
       wvtmp(1) = denw(lip - 1)
       ptmp(2) = p
       tip = (tm(lip) - tm(lip - 1))/log(pm(lip)/pm(lip - 1))
       ttmp(2) = tm(lip - 1) + tip*log(p/pm(lip - 1))
       wvip = (denw(lip) - denw(lip - 1))/log(pm(lip)/pm(lip - 1))
       wvtmp(2) = denw(lip - 1) + wvip*log(p/pm(lip - 1))
       ztmp(2) = 0
       call cmpalt(2, ptmp, ttmp, wvtmp, ztmp(1), ref_lat, re, ztmp)
       ratp = log(p/pm(lip - 1))/log(pm(lip)/pm(lip - 1))
       a = ratp**3
       z = a*zint + (1 - a)*ztmp(2)
    end if
  end function pressure_to_altitude

  ! DEFALT: keys 1-6 on pressure, temperature or a molecule ask for the value of that model atmosphere at altitude Z
  ! (4-point Lagrange in altitude; ln p for the pressure); molecules > 7 only have the U.S. standard trace profiles
  subroutine default_values(z, p, t, nmol, junitp, junitt, junit, wmol)
    real(dp), intent(in) :: z
    real(dp), intent(inout) :: p, t, wmol(MXMOLF)
    integer, intent(in) :: nmol, junitp, junitt
    integer, intent(inout) :: junit(MXMOLF)
    integer :: im, i0, i1, i2, i3, iupper, k, matm
    real(dp) :: z0, z1, z2, z3, den1, den2, den3, den4, a1, a2, a3, a4, x1, x2, x3, x4
    logical :: needed
    needed = junitp <= 6 .or. junitt <= 6 .or. any(junit(1:nmol) <= 6)
    if (.not. needed) return
    iupper = 0
    i2 = NLEV_MDL
    do im = 2, NLEV_MDL
       if (alt_mdl(im) >= z) then
          i2 = im
          exit
       end if
    end do
    i1 = i2 - 1
    i0 = i2 - 2
    i3 = i2 + 1
    if (i0 < 1) then                      ! lower end point
       i0 = i1
       i1 = i2
       i2 = i3
       i3 = i3 + 1
    else if (i3 > NLEV_MDL) then          ! upper end point
       if (z > alt_mdl(NLEV_MDL)) call fail('DEFAULT Z: altitude above 120 km with a model-atmosphere default')
       i3 = i2
       i2 = i1
       i1 = i0
       i0 = i1 - 1
    end if
    z1 = alt_mdl(i1); z2 = alt_mdl(i2); z0 = alt_mdl(i0); z3 = alt_mdl(i3)
    den1 = (z0 - z1)*(z0 - z2)*(z0 - z3)
    den2 = (z1 - z2)*(z1 - z3)*(z1 - z0)
    den3 = (z2 - z3)*(z2 - z0)*(z2 - z1)
    den4 = (z3 - z0)*(z3 - z1)*(z3 - z2)
    a1 = ((z - z1)*(z - z2)*(z - z3))/den1
    a2 = ((z - z2)*(z - z3)*(z - z0))/den2
    a3 = ((z - z3)*(z - z0)*(z - z1))/den3
    a4 = ((z - z0)*(z - z1)*(z - z2))/den4
    if (junitp <= 6) then
       matm = junitp
       x1 = log(pmdl(i0, matm)); x2 = log(pmdl(i1, matm)); x3 = log(pmdl(i2, matm)); x4 = log(pmdl(i3, matm))
       p = exp(a1*x1 + a2*x2 + a3*x3 + a4*x4)
    end if
    if (junitt <= 6) then
       matm = junitt
       t = a1*tmdl(i0, matm) + a2*tmdl(i1, matm) + a3*tmdl(i2, matm) + a4*tmdl(i3, matm)
    end if
    do k = 1, nmol
       if (junit(k) > 6) cycle
       if (k <= 7) then
          matm = junit(k)
          x1 = amol(i0, k, matm); x2 = amol(i1, k, matm); x3 = amol(i2, k, matm); x4 = amol(i3, k, matm)
       else
          if (k > 28) call fail('no default profile beyond molecule 28')
          x1 = trac(i0, k - 7); x2 = trac(i1, k - 7); x3 = trac(i2, k - 7); x4 = trac(i3, k - 7)
       end if
       wmol(k) = a1*x1 + a2*x2 + a3*x3 + a4*x4
       junit(k) = 10
    end do
  end subroutine default_values

  ! DEFALT_P (src/lblatm.f90:3688-3870): the same for a profile on pressure levels - for each model atmosphere that a key
  ! names, four-point Lagrange weights in ln p on THAT model's pressure grid
  subroutine default_values_p(p, t, nmol, junitt, junit, wmol)
    real(dp), intent(in) :: p
    real(dp), intent(inout) :: t, wmol(MXMOLF)
    integer, intent(in) :: nmol, junitt
    integer, intent(inout) :: junit(MXMOLF)
    integer :: jm, lvl, i0, i1, i2, i3, k, kd(MXMOLF)
    real(dp) :: xlp, p0, p1, p2, p3, den1, den2, den3, den4, a1, a2, a3, a4, x1, x2, x3, x4
    if (.not. (junitt <= 6 .or. any(junit(1:nmol) <= 6))) return
    xlp = log(p)
    kd(1:nmol) = junit(1:nmol)             ! the keys as read: a molecule is served by the model it names
    do jm = 1, 6
       if (.not. (junitt == jm .or. any(kd(1:nmol) == jm))) cycle
       i2 = NLEV_MDL
       do lvl = 2, NLEV_MDL
          if (p >= pmdl(lvl, jm)) then
             i2 = lvl
             exit
          end if
       end do
       i1 = i2 - 1
       i0 = i2 - 2
       i3 = i2 + 1
       if (i0 < 1) then                    ! lower end point
          i0 = i1
          i1 = i2
          i2 = i3
          i3 = i3 + 1
       else if (i3 > NLEV_MDL) then        ! upper end point
          if (p <= pmdl(NLEV_MDL, jm)) call fail('DEFAULT P: pressure above the top of a model atmosphere')
          i3 = i2
          i2 = i1
          i1 = i0
          i0 = i1 - 1
       end if
       p0 = log(pmdl(i0, jm)); p1 = log(pmdl(i1, jm)); p2 = log(pmdl(i2, jm)); p3 = log(pmdl(i3, jm))
       den1 = (p0 - p1)*(p0 - p2)*(p0 - p3)
       den2 = (p1 - p2)*(p1 - p3)*(p1 - p0)
       den3 = (p2 - p3)*(p2 - p0)*(p2 - p1)
       den4 = (p3 - p0)*(p3 - p1)*(p3 - p2)
       a1 = ((xlp - p1)*(xlp - p2)*(xlp - p3))/den1
       a2 = ((xlp - p2)*(xlp - p3)*(xlp - p0))/den2
       a3 = ((xlp - p3)*(xlp - p0)*(xlp - p1))/den3
       a4 = ((xlp - p0)*(xlp - p1)*(xlp - p2))/den4
       if (junitt == jm) t = a1*tmdl(i0, jm) + a2*tmdl(i1, jm) + a3*tmdl(i2, jm) + a4*tmdl(i3, jm)
       do k = 1, nmol
          if (kd(k) /= jm) cycle
          if (k <= 7) then
             x1 = amol(i0, k, jm); x2 = amol(i1, k, jm); x3 = amol(i2, k, jm); x4 = amol(i3, k, jm)
          else
             if (k > 28) call fail('no default profile beyond molecule 28')
             x1 = trac(i0, k - 7); x2 = trac(i1, k - 7); x3 = trac(i2, k - 7); x4 = trac(i3, k - 7)
          end if
          wmol(k) = a1*x1 + a2*x2 + a3*x3 + a4*x4
          junit(k) = 10
       end do
    end do
  end subroutine default_values_p

  ! CONVRT + WATVAP: the level's molecular amounts in their input units -> number densities (cm-3); water first, the
  ! others relative to DRY air
  subroutine convert_units(p, t, nmol, junit, wmol, den)
    real(dp), intent(in) :: p, t
    integer, intent(in) :: nmol, junit(MXMOLF)
    real(dp), intent(inout) :: wmol(MXMOLF)
    real(dp), intent(out) :: den(MXMOLF)
    real(dp), parameter :: C1 = 18.9766_dp, C2 = -14.9595_dp, C3 = -2.4388_dp
    real(dp) :: rhoair, a, b, r, dryair, atd, w
    integer :: k
    den = 0
    rhoair = ALOSMT*(p/PZERO)*(TZERO/t)
    a = TZERO/t
    b = AVOGAD/AMWT(1)
    r = AIRMWT/AMWT(1)
    select case (junit(1))
    case (10)                                   ! volume mixing ratio (ppmv) w.r.t. dry air
       w = wmol(1)*1.0E-06_dp
       den(1) = (w/(1.0_dp + w))*rhoair
    case (11); den(1) = wmol(1)                 ! number density
    case (12)                                   ! mass mixing ratio (g/kg)
       w = wmol(1)*r*1.0E-3_dp
       den(1) = (w/(1.0_dp + w))*rhoair
    case (13); den(1) = b*wmol(1)*1.0E-6_dp     ! mass density (g m-3)
    case (14); den(1) = ALOSMT*(wmol(1)/PZERO)*(TZERO/t)   ! partial pressure (mb)
    case (15)                                   ! dew point (K)
       atd = TZERO/wmol(1)
       den(1) = densat(atd)*wmol(1)/t
    case (16)                                   ! dew point (C)
       atd = TZERO/(TZERO + wmol(1))
       den(1) = densat(atd)*(TZERO + wmol(1))/t
    case (17); den(1) = densat(a)*(wmol(1)/100.0_dp)       ! relative humidity (percent)
    case default; call fail('WATVAP: invalid unit key for water vapour')
    end select
    dryair = rhoair - den(1)
    do k = 2, nmol
       b = AVOGAD/AMWT(k)
       r = AIRMWT/AMWT(k)
       select case (junit(k))
       case (:10); den(k) = wmol(k)*dryair*1.0E-6_dp
       case (11); den(k) = wmol(k)
       case (12); den(k) = r*wmol(k)*1.0E-3_dp*dryair
       case (13); den(k) = b*wmol(k)*1.0E-6_dp
       case (14); den(k) = ALOSMT*(wmol(k)/PZERO)*(TZERO/t)
       case default; call fail('CONVRT: invalid unit key for a molecule')
       end select
    end do
  contains
    real(dp) function densat(atemp)             ! saturation water vapour density over water (LOWTRAN), molecules cm-3
      real(dp), intent(in) :: atemp
      densat = atemp*(AVOGAD/AMWT(1))*exp(C1 + C2*atemp + C3*atemp**2)*1.0E-6_dp
    end function densat
  end subroutine convert_units

  ! ------------------------------------------------------------------ model atmosphere (MDLATM) + refractivity
  subroutine load_model(rq, xvbar, pr)
    type(atm_request), intent(in) :: rq
    real(dp), intent(in) :: xvbar
    type(profile), intent(out) :: pr
    integer :: i, k, ispace, nlev
    real(dp) :: dryair, pph2o, hspace
    hspace = rq%hspace
    if (hspace == 0) hspace = 100
    ispace = 1
    nlev = merge(rq%immax, NLEV_MDL, rq%model == 0)
    allocate (pr%z(nlev), pr%p(nlev), pr%t(nlev), pr%rfndx(nlev), pr%den(MXMOLF, nlev))
    pr%den = 0
    pr%rfndx = 0
    do i = 1, nlev
       if (rq%model == 0) then
          pr%z(i) = rq%zm(i)
          pr%p(i) = rq%pm(i)
          pr%t(i) = rq%tm(i)
          pr%den(:, i) = rq%denm(:, i)
       else
          pr%z(i) = alt_mdl(i)
          pr%p(i) = pmdl(i, rq%model)
          pr%t(i) = tmdl(i, rq%model)
          pr%den(1, i) = amol(i, 1, rq%model)*amol(i, 8, rq%model)*1.0E-6_dp     ! water first, dry air = total - water
          dryair = amol(i, 8, rq%model) - pr%den(1, i)
          do k = 1, min(7, rq%nmol)
             pr%den(k, i) = amol(i, k, rq%model)*1.0E-6_dp*dryair
          end do
          do k = 8, min(28, rq%nmol)
             pr%den(k, i) = trac(i, k - 7)*1.0E-6_dp*dryair
          end do
       end if
       if (hspace + 0.001_dp > pr%z(i)) ispace = i
    end do
    pr%hmod = rq%hmod
    if (rq%model > 0) pr%hmod = atmnam(rq%model)
    pr%n = ispace
    pr%zmax = pr%z(pr%n)
    pr%re = rq%re
    if (pr%re == 0) then
       pr%re = 6371.23_dp
       if (rq%model == 1) pr%re = 6378.39_dp
       if (rq%model == 4 .or. rq%model == 5) pr%re = 6356.91_dp
    end if
    do i = 1, pr%n                                                             ! LOWTRAN6 form, 1 - index
       pph2o = pr%den(1, i)*PZERO*pr%t(i)/(TZERO*ALOSMT)
       pr%rfndx(i) = ((83.42_dp + (185.08_dp/(1.0_dp - (xvbar/1.14E+5_dp)**2)) + (4.11_dp/(1.0_dp - (xvbar/6.24E+4_dp)**2)))* &
            (pr%p(i)*288.15_dp)/(1013.25_dp*pr%t(i)) - (43.49_dp - (xvbar/1.7E+4_dp)**2)*(pph2o/1013.25_dp))*1.0E-06_dp
    end do
  end subroutine load_model

  ! scale height and ground value of the refractivity between two levels (SCALHT), index of refraction (ANDEX), radius of
  ! curvature of a horizontal ray (RADREF)
  subroutine scalht(z1, z2, rf1in, rf2in, sh, gamma)
    real(dp), intent(in) :: z1, z2, rf1in, rf2in
    real(dp), intent(out) :: sh, gamma
    real(dp) :: rf1, rf2, ratio
    rf1 = rf1in + 1.0E-20_dp
    rf2 = rf2in + 1.0E-20_dp
    ratio = rf1/rf2
    if (abs(ratio - 1.0_dp) < 1.0E-05_dp) then
       sh = 0
       gamma = rf1in
    else
       sh = (z2 - z1)/log(ratio)
       gamma = rf1*(rf2/rf1)**(-z1/(z2 - z1))
    end if
  end subroutine scalht

  subroutine findsh(pr, h, sh, gamma)
    type(profile), intent(in) :: pr
    real(dp), intent(in) :: h
    real(dp), intent(out) :: sh, gamma
    integer :: im, i2
    i2 = pr%n
    do im = 2, pr%n
       if (pr%z(im) >= h) then
          i2 = im
          exit
       end if
    end do
    call scalht(pr%z(i2 - 1), pr%z(i2), pr%rfndx(i2 - 1), pr%rfndx(i2), sh, gamma)
  end subroutine findsh

  pure function andex(h, sh, gamma) result(v)
    real(dp), intent(in) :: h, sh, gamma
    real(dp) :: v
    if (sh == 0) then
       v = 1.0_dp + gamma
    else
       v = 1.0_dp + gamma*exp(-h/sh)
    end if
  end function andex

  pure function radref(h, sh, gamma) result(v)
    real(dp), intent(in) :: h, sh, gamma
    real(dp) :: v
    if (sh == 0) then
       v = 1.0E36_dp
    else
       v = sh*(1.0_dp + exp(h/sh)/gamma)
    end if
  end function radref

  ! ------------------------------------------------------------------ path geometry (FSCGEO cases 2A / 3A, FNDHMN)
  subroutine fndhmn(pr, h1, angle, h2, len, hmin, phi)
    type(profile), intent(in) :: pr
    real(dp), intent(in) :: h1, angle
    real(dp), intent(inout) :: h2
    integer, intent(inout) :: len
    real(dp), intent(out) :: hmin, phi
    real(dp), parameter :: DH = 0.2_dp, ETA = 5.0E-7_dp
    real(dp) :: deg, sh, gamma, cpath, ch2, cmin, ht1, htp, ct1, ctp, deriv
    integer :: n
    deg = 180.0_dp/PI
    call findsh(pr, h1, sh, gamma)
    cpath = (pr%re + h1)*andex(h1, sh, gamma)*sin(angle/deg)
    call findsh(pr, h2, sh, gamma)
    ch2 = (pr%re + h2)*andex(h2, sh, gamma)
    if (abs(cpath/ch2) > 1.0_dp) call fail('H2 IS LESS THAN THE TANGENT HEIGHT FOR THIS PATH')
    if (angle <= 90.0_dp) then
       len = 0
       hmin = h1
    else
       if (h1 <= h2) len = 1
       if (len /= 1) then
          len = 0
          hmin = h2
       else                                   ! long path through a tangent height: Newton iteration on index*(RE+H) = CPATH
          call findsh(pr, 0.0_dp, sh, gamma)
          cmin = pr%re*andex(0.0_dp, sh, gamma)
          if (cpath < cmin) then               ! tangent path intersects the earth
             h2 = 0
             hmin = 0
             len = 0
             ch2 = cmin
          else
             ht1 = h1*sin(angle/deg) + (sin(angle/deg) - 1.0_dp)*pr%re
             n = 0
             do
                n = n + 1
                call findsh(pr, ht1, sh, gamma)
                ct1 = (pr%re + ht1)*andex(ht1, sh, gamma)
                if (abs((cpath - ct1)/cpath) < ETA) exit
                if (n > 15) call fail('FNDHMN: tangent height iteration did not converge')
                htp = ht1 - DH
                call findsh(pr, htp, sh, gamma)
                ctp = (pr%re + htp)*andex(htp, sh, gamma)
                deriv = (ct1 - ctp)/DH
                ht1 = ht1 + (cpath - ct1)/deriv
             end do
             hmin = ht1
          end if
       end if
    end if
    phi = asin(cpath/ch2)*deg
    if (angle <= 90.0_dp .or. len == 1) phi = 180.0_dp - phi
  end subroutine fndhmn

  subroutine reduce_path(rq, pr, h1, h2, angle, len, hmin, phi)
    type(atm_request), intent(in) :: rq
    type(profile), intent(in) :: pr
    real(dp), intent(inout) :: h1, h2, angle
    integer, intent(inout) :: len
    real(dp), intent(out) :: hmin, phi
    real(dp) :: h2st, htan, h1c, dum
    integer :: ldum
    if (rq%itype == 3 .and. h2 /= 0) then      ! case 3B: H1, tangent height (read as H2), space (src/lblatm.f90:4198-4209)
       htan = h2
       h2 = pr%zmax
       if (h1 < htan) call fail('FSCGEO case 3B: H1 below the tangent height: error in input data')
       h1c = h1
       call fndhmn(pr, htan, 90.0_dp, h1c, len, dum, angle)    ! zenith angle at H1 of the ray that is horizontal at HMIN
       call fndhmn(pr, htan, 90.0_dp, h2, len, hmin, phi)
       if (hmin < h1) len = 1
       ldum = len
    else if (rq%itype == 3) then               ! case 3A: H1, space, ANGLE
       h2 = pr%zmax
       call fndhmn(pr, h1, angle, h2, len, hmin, phi)
    else                                       ! case 2A: H1, H2, ANGLE
       if (h1 >= h2 .and. angle <= 90.0_dp) call fail('FSCGEO case 2A: H1 >= H2 with ANGLE <= 90: error in input data')
       if (h1 == 0 .and. angle > 90.0_dp) call fail('FSCGEO: slant path intersects the earth')
       h2st = h2
       call fndhmn(pr, h1, angle, h2, len, hmin, phi)
       if (h2 /= h2st) call fail('FSCGEO: slant path intersects the earth')
    end if
    len = 0
    if (hmin < min(h1, h2)) len = 1
    if (hmin >= pr%zmax) call fail('FSCGEO: the entire path lies above the top of the profile')
    if (h1 > pr%zmax .or. h2 > pr%zmax) then    ! REDUCE: end points above the profile come down to its top, along the ray
       block
         real(dp) :: sh, gamma, cpath, czmax, angmax, deg
         deg = 180.0_dp/PI
         call findsh(pr, h1, sh, gamma)
         cpath = andex(h1, sh, gamma)*(pr%re + h1)*sin(angle/deg)
         call findsh(pr, pr%zmax, sh, gamma)
         czmax = andex(pr%zmax, sh, gamma)*(pr%re + pr%zmax)
         angmax = 180.0_dp - asin(cpath/czmax)*deg
         if (h1 > pr%zmax) then
            h1 = pr%zmax
            angle = angmax
         end if
         if (h2 > pr%zmax) then
            h2 = pr%zmax
            phi = angmax
         end if
       end block
    end if
  end subroutine reduce_path

  ! ------------------------------------------------------------------ layer boundaries (AUTLAY, HALFWD)
  subroutine halfwd(pr, z, xvbar, p, t, alornz, adopp, avoigt)
    type(profile), intent(in) :: pr
    real(dp), intent(in) :: z, xvbar
    real(dp), intent(out) :: p, t, alornz, adopp, avoigt
    integer :: im, i2
    real(dp) :: fac, adcon
    adcon = sqrt(2.0_dp*log(2.0_dp)*GASCON/CLIGHT**2)
    im = pr%n
    do i2 = 2, pr%n
       if (pr%z(i2) >= z) then
          im = i2
          exit
       end if
    end do
    fac = (z - pr%z(im - 1))/(pr%z(im) - pr%z(im - 1))
    p = expint(pr%p(im - 1), pr%p(im), fac)
    t = pr%t(im - 1) + (pr%t(im) - pr%t(im - 1))*fac
    alornz = ALZERO*(p/PZERO)*sqrt(296.0_dp/t)
    adopp = adcon*xvbar*sqrt(t/AVMWT)
    avoigt = 0.5_dp*(alornz + sqrt(alornz**2 + 4.0_dp*adopp**2))
  end subroutine halfwd

  subroutine autlay(rq, pr, hmin_in, hmax, xvbar, zbnd, ibmax)
    type(atm_request), intent(in) :: rq
    type(profile), intent(in) :: pr
    real(dp), intent(in) :: hmin_in, hmax, xvbar
    real(dp), intent(out) :: zbnd(MXBND)
    integer, intent(out) :: ibmax
    real(dp) :: pbnd(MXBND), tbnd(MXBND), avoigt(MXBND), avtm(pr%n + 1)
    real(dp) :: hmin, htop, p, t, al, ad, tmin, tmax, zbndti, x, alogx, y, alogy, fac, tdiff
    integer :: im, ihmin, ib, ind, ipass
    hmin = max(hmin_in, pr%z(1))
    ihmin = pr%n
    do im = 2, pr%n
       ihmin = im
       if (pr%z(im) > hmin) exit
    end do
    htop = min(hmax, pr%zmax)
    im = ihmin - 1
    call halfwd(pr, pr%z(im), xvbar, p, t, al, ad, avtm(im))
    ib = 1
    zbnd(ib) = hmin
    im = ihmin
    call halfwd(pr, zbnd(ib), xvbar, pbnd(ib), tbnd(ib), al, ad, avoigt(ib))
    outer: do
       ib = ib + 1
       if (ib > MXBND) call fail('AUTLAY: the number of generated boundaries exceeds the dimension')
       tmin = tbnd(ib - 1)
       tmax = tbnd(ib - 1)
       ind = 0
       inner: do
          ipass = 0
          zbnd(ib) = pr%z(im)
          zbndti = pr%z(im)
          if (zbnd(ib) >= htop) zbnd(ib) = htop
          call halfwd(pr, zbnd(ib), xvbar, pbnd(ib), tbnd(ib), al, ad, avoigt(ib))
          avtm(im) = avoigt(ib)
          if (.not. (avoigt(ib - 1)/avoigt(ib) < rq%avtrat)) then        ! Voigt-width ratio test failed at this level
             ipass = 1
             avoigt(ib) = avoigt(ib - 1)/rq%avtrat
             x = avtm(im)/avtm(im - 1)
             alogx = 1.0_dp - x
             if (abs(alogx) < 0.001_dp) then
                zbnd(ib) = (pr%z(im) + pr%z(im - 1))/2.0_dp
             else
                alogx = log(x)
                y = avoigt(ib)/avtm(im - 1)
                alogy = 1.0_dp - y
                if (abs(alogy) > 0.001_dp) alogy = log(y)
                zbnd(ib) = pr%z(im - 1) + (pr%z(im) - pr%z(im - 1))*alogy/alogx
             end if
          end if
          fac = (zbnd(ib - 1) - rq%altd1)/(rq%altd2 - rq%altd1)            ! temperature difference test
          tdiff = expint(rq%tdiff1, rq%tdiff2, fac)
          if (pr%t(im) > tmax) then
             ind = 1
             tmax = pr%t(im)
          end if
          if (pr%t(im) < tmin) then
             ind = 2
             tmin = pr%t(im)
          end if
          if (.not. (tmax - tmin <= tdiff)) then
             if (ind == 1) tbnd(ib) = tmin + tdiff
             if (ind == 2) tbnd(ib) = tmax - tdiff
             ipass = 2
             if (abs(pr%t(im) - pr%t(im - 1)) < 0.0001_dp) then
                zbndti = (pr%z(im) + pr%z(im - 1))/2.0_dp
             else
                zbndti = pr%z(im - 1) + (pr%z(im) - pr%z(im - 1))*(tbnd(ib) - pr%t(im - 1))/(pr%t(im) - pr%t(im - 1))
             end if
          end if
          if (zbndti < zbnd(ib)) zbnd(ib) = zbndti
          if (zbnd(ib) >= htop) then
             zbnd(ib) = htop
             if (zbnd(ib) - zbnd(ib - 1) <= 0.1_dp) then
                ib = ib - 1
                zbnd(ib) = htop
             end if
             exit outer
          end if
          if (ipass /= 0) exit inner
          im = im + 1                                                       ! both tests pass: try the next model level
       end do inner
       zbnd(ib) = 0.1_dp*real(int(10.0_dp*zbnd(ib)), dp)                    ! ZROUND: down to the nearest tenth of a km
       call halfwd(pr, zbnd(ib), xvbar, pbnd(ib), tbnd(ib), al, ad, avoigt(ib))
    end do outer
    ibmax = ib
  end subroutine autlay

  ! ------------------------------------------------------------------ ray trace through the merged levels
  ! path levels: the output boundaries between HMIN and HMAX merged with the model levels (AMERGE)
  subroutine amerge(pr, nmol, zbnd, ibmax, h1, h2, hmin, len, zout, ioutmx, zp, pp, tp, rfp, denp, ipmax, iphmid)
    type(profile), intent(inout) :: pr
    integer, intent(in) :: nmol, ibmax, len
    real(dp), intent(inout) :: zbnd(MXBND), h1
    real(dp), intent(in) :: h2, hmin
    real(dp), intent(out) :: zout(MXBND + 3), zp(MXPTH), pp(MXPTH), tp(MXPTH), rfp(MXPTH), denp(MXMOLF, MXPTH)
    integer, intent(out) :: ioutmx, ipmax, iphmid
    real(dp), parameter :: TOL = 5.0E-4_dp
    real(dp) :: zh(3), hmid, hmax, a
    integer :: ihmax, i1, iout, ib, ih, im, ip, jm, k
    hmid = min(h1, h2)
    hmax = max(h1, h2)
    ihmax = 2
    zh(1) = hmin
    if (len == 0) then
       zh(2) = hmax
    else
       zh(2) = hmid
       if (abs(h1 - h2) < TOL) h1 = h2
       if (h1 /= h2) then
          ihmax = 3
          zh(3) = hmax
       end if
    end if
    zout(1) = zh(1)
    i1 = ibmax
    do ib = 1, ibmax
       if (abs(zbnd(ib) - zh(1)) < TOL) zh(1) = zbnd(ib)
       if (zbnd(ib) > zh(1)) then
          i1 = ib
          exit
       end if
    end do
    iout = 1
    ib = i1
    ih = 2
    do
       iout = iout + 1
       if (ib <= ibmax) then
          if (abs(zbnd(ib) - zh(ih)) < TOL) zh(ih) = zbnd(ib)
          if (zbnd(ib) < zh(ih)) then
             zout(iout) = zbnd(ib)
             ib = ib + 1
             cycle
          end if
          if (zbnd(ib) == zh(ih)) ib = ib + 1
       end if
       zout(iout) = zh(ih)
       ih = ih + 1
       if (ih > ihmax) exit
    end do
    ioutmx = iout
    im = 0
    do k = 1, pr%n
       if (pr%z(k) >= hmin) then
          im = k
          exit
       end if
    end do
    if (im == 0) call fail('AMERGE: the profile does not extend up to HMIN')
    iphmid = 0
    ip = 0
    iout = 1
    do
       ip = ip + 1
       if (ip > MXPTH) call fail('AMERGE: too many merged levels')
       if (im <= pr%n) then
          if (abs(zout(iout) - pr%z(im)) < TOL) pr%z(im) = zout(iout)
       end if
       if (im <= pr%n .and. .not. (zout(iout) < pr%z(min(im, pr%n)))) then   ! take the model level
          if (zout(iout) == pr%z(im)) iout = iout + 1
          zp(ip) = pr%z(im)
          pp(ip) = pr%p(im)
          tp(ip) = pr%t(im)
          rfp(ip) = pr%rfndx(im)
          denp(1:nmol, ip) = pr%den(1:nmol, im)
          im = im + 1
          if (abs(zp(ip) - hmid) < TOL) hmid = zp(ip)
          if (zp(ip) == hmid) iphmid = ip
          if (abs(zp(ip) - zout(ioutmx)) < TOL) zout(ioutmx) = zp(ip)
          if (zp(ip) == zout(ioutmx)) exit
       else                                                                   ! insert the output boundary, interpolated
          zp(ip) = zout(iout)
          jm = max(im, 2)
          a = (zout(iout) - pr%z(jm - 1))/(pr%z(jm) - pr%z(jm - 1))
          pp(ip) = expint(pr%p(jm - 1), pr%p(jm), a)
          tp(ip) = pr%t(jm - 1) + (pr%t(jm) - pr%t(jm - 1))*a
          rfp(ip) = expint(pr%rfndx(jm - 1), pr%rfndx(jm), a)
          do k = 1, nmol
             denp(k, ip) = expint(pr%den(k, jm - 1), pr%den(k, jm), a)
          end do
          if (abs(zp(ip) - hmid) < TOL) zp(ip) = hmid
          if (zp(ip) == hmid) iphmid = ip
          iout = iout + 1
          if (abs(zp(ip) - zout(ioutmx)) < TOL) zp(ip) = zout(ioutmx)
          if (zp(ip) == zout(ioutmx)) exit
       end if
    end do
    ipmax = ip
  end subroutine amerge

  ! one path layer J: refracted ray from ZP(J) to ZP(J+1) in steps of at most DELTAS along the ray, three-point quadrature
  ! with unequally spaced points; density-weighted pressure / temperature sums and column amounts with exponential (or
  ! linear) interpolation of the densities inside the layer (ALAYER)
  subroutine alayer(re, gcair, nmol, z1, z2, pa_in, pb_in, ta, tb, dena_in, denb_in, sinai, cosai, cpath, sh, gamma, &
                    s, bend, ppsum, tpsum, rhopsm, amtp)
    real(dp), intent(in) :: re, gcair, z1, z2, pa_in, pb_in, ta, tb, cpath, sh, gamma
    integer, intent(in) :: nmol
    real(dp), intent(in) :: dena_in(MXMOLF), denb_in(MXMOLF)
    real(dp), intent(inout) :: sinai, cosai
    real(dp), intent(out) :: s, bend, ppsum, tpsum, rhopsm, amtp(MXMOLF)
    real(dp), parameter :: EPSILN = 1.0E-5_dp
    real(dp) :: h1, r1, dhmin, sinai1, cosai1, y1, y3, x1, x2, x3, ratio1, ratio2, ratio3, dsdx1, dsdx2, dsdx3, dbndx1, dbndx2, &
         dbndx3, pa, pb, rhoa, rhob, dz, hp, hrho, hden(MXMOLF), dena(MXMOLF), denb(MXMOLF), dh, h2, h3, r2, r3, sinai2, &
         sinai3, cosai2, cosai3, dx, w1, w2, w3, d31, d32, d21, ds, dbend, dsdz
    integer :: k
    h1 = z1
    r1 = re + h1
    dhmin = DELTAS**2/(2.0_dp*r1)
    sinai1 = sinai
    cosai1 = cosai
    y1 = 0
    if ((1.0_dp - sinai) < EPSILN) y1 = cosai1**2/2.0_dp + cosai1**4/8.0_dp + cosai1**6*3.0_dp/48.0_dp
    y3 = 0
    x1 = -r1*cosai1
    ratio1 = r1/radref(h1, sh, gamma)
    dsdx1 = 1.0_dp/(1.0_dp - ratio1*sinai1**2)
    dbndx1 = dsdx1*sinai1*ratio1/r1
    s = 0
    bend = 0
    ppsum = 0
    tpsum = 0
    rhopsm = 0
    amtp = 0
    pa = pa_in
    pb = pb_in
    if (pb == pa) call fail('LBLATM: PRESSURES IN ADJOINING LAYERS MUST DIFFER')
    rhoa = pa/(gcair*ta)
    rhob = pb/(gcair*tb)
    dz = z2 - z1
    hp = -dz/log(pb/pa)
    if (abs(rhob/rhoa - 1.0_dp) >= EPSILN) then
       hrho = -dz/log(rhob/rhoa)
    else
       hrho = 1.0E30_dp
    end if
    do k = 1, nmol
       dena(k) = dena_in(k)
       denb(k) = denb_in(k)
       if (dena(k) == 0 .or. denb(k) == 0) then
          hden(k) = 0
       else if (abs(1.0_dp - dena(k)/denb(k)) <= EPSILN) then
          hden(k) = 0
       else
          hden(k) = -dz/log(denb(k)/dena(k))
       end if
    end do
    do
       dh = -DELTAS*cosai1
       dh = max(dh, dhmin)
       h3 = h1 + dh
       if (h3 > z2) h3 = z2
       dh = h3 - h1
       r3 = re + h3
       h2 = h1 + dh/2.0_dp
       r2 = re + h2
       sinai2 = cpath/(andex(h2, sh, gamma)*r2)
       sinai3 = cpath/(andex(h3, sh, gamma)*r3)
       ratio2 = r2/radref(h2, sh, gamma)
       ratio3 = r3/radref(h3, sh, gamma)
       if ((1.0_dp - sinai2) <= EPSILN) then        ! near a tangent height: COSAI from the series in Y
          y3 = y1 + (sinai1*(1.0_dp - ratio1)/r1 + 4.0_dp*sinai2*(1.0_dp - ratio2)/r2 + sinai3*(1.0_dp - ratio3)/r3)*dh/6.0_dp
          cosai3 = -sqrt(2.0_dp*y3 - y3**2)
          x3 = -r3*cosai3
          dx = x3 - x1
          w1 = 0.5_dp*dx
          w2 = 0
          w3 = 0.5_dp*dx
       else
          cosai2 = -sqrt(1.0_dp - sinai2**2)
          cosai3 = -sqrt(1.0_dp - sinai3**2)
          x2 = -r2*cosai2
          x3 = -r3*cosai3
          d31 = x3 - x1
          d32 = x3 - x2
          d21 = x2 - x1
          if (d32 == 0 .or. d21 == 0) then
             w1 = 0.5_dp*d31
             w2 = 0
             w3 = 0.5_dp*d31
          else
             w1 = (2.0_dp - d32/d21)*d31/6.0_dp
             w2 = d31**3/(d32*d21*6.0_dp)
             w3 = (2.0_dp - d21/d32)*d31/6.0_dp
          end if
       end if
       dsdx2 = 1.0_dp/(1.0_dp - ratio2*sinai2**2)
       dsdx3 = 1.0_dp/(1.0_dp - ratio3*sinai3**2)
       dbndx2 = dsdx2*sinai2*ratio2/r2
       dbndx3 = dsdx3*sinai3*ratio3/r3
       ds = w1*dsdx1 + w2*dsdx2 + w3*dsdx3
       dbend = w1*dbndx1 + w2*dbndx2 + w3*dbndx3
       ! (the reference's 2013 "elevation bug fix" re-weights in R when R / RADREF >= 1 at a layer end: super-refraction,
       !  which the built-in model atmospheres never reach - not carried over)
       s = s + ds
       bend = bend + dbend
       dsdz = ds/dh
       pb = pa*exp(-dh/hp)
       rhob = rhoa*exp(-dh/hrho)
       if ((dh/hrho) >= EPSILN) then
          ppsum = ppsum + dsdz*(hp/(1.0_dp + hp/hrho))*(pa*rhoa - pb*rhob)
          tpsum = tpsum + dsdz*hp*(pa - pb)/gcair
          rhopsm = rhopsm + dsdz*hrho*(rhoa - rhob)
       else
          ppsum = ppsum + 0.5_dp*ds*(pa*rhoa + pb*rhob)
          tpsum = tpsum + 0.5_dp*ds*(pa + pb)/gcair
          rhopsm = rhopsm + 0.5_dp*ds*(rhoa + rhob)
       end if
       do k = 1, nmol
          if (hden(k) == 0) then
             denb(k) = dena_in(k) + (denb_in(k) - dena_in(k))*(h3 - z1)/dz
             amtp(k) = amtp(k) + 0.5_dp*(dena(k) + denb(k))*ds*1.0E5_dp
          else if (abs(dh/hden(k)) < EPSILN) then
             denb(k) = dena_in(k) + (denb_in(k) - dena_in(k))*(h3 - z1)/dz
             amtp(k) = amtp(k) + 0.5_dp*(dena(k) + denb(k))*ds*1.0E5_dp
          else
             denb(k) = dena_in(k)*exp(-(h3 - z1)/hden(k))
             amtp(k) = amtp(k) + dsdz*hden(k)*(dena(k) - denb(k))*1.0E5_dp
          end if
       end do
       pa = pb
       rhoa = rhob
       dena(1:nmol) = denb(1:nmol)
       if (h3 < z2) then
          h1 = h3
          r1 = r3
          sinai1 = sinai3
          ratio1 = ratio3
          y1 = y3
          cosai1 = cosai3
          x1 = x3
          dsdx1 = dsdx3
          dbndx1 = dbndx3
       else
          sinai = sinai3
          cosai = cosai3
          exit
       end if
    end do
  end subroutine alayer

  ! ------------------------------------------------------------------ driver: records 3.x -> layers
  subroutine build_atm_layers(rq, iemit, out)
    type(atm_request), intent(in) :: rq
    integer, intent(in) :: iemit
    type(atm_layers), intent(out) :: out
    type(profile) :: pr
    real(dp) :: xvbar, h1, h2, angle, hmin, phi, hmax, deg, gcair, zbnd(MXBND), zout(MXBND + 3)
    real(dp), allocatable :: zp(:), pp(:), tp(:), rfp(:), denp(:, :), sp(:), ppsum(:), tpsum(:), rhopsm(:), amtp(:, :)
    real(dp) :: ha, anglea, sh, gamma, cpath, sinai, cosai, ds, dbend, hmid, fac, amttot(MXMOLF), amtcum(MXMOLF), sumamt
    real(dp), allocatable :: pbar(:), tbar(:), rhosum(:), sout(:), amount(:, :), pz(:), tz(:)
    integer :: len, ibmax, ioutmx, ipmax, iphmid, iorder, j, k, l, iout, lmax, iskip(MXMOLF), iskpt, nmol_max, nmol, ib
    integer, allocatable :: ipath(:)
    nmol = rq%nmol
    deg = 180.0_dp/PI
    gcair = 1.0E-3_dp*GASCON/AVOGAD
    xvbar = rq%xvbar                 ! the reference takes (V1+V2)/2 of COMMON /ADRIVE/, which monoRTM never sets: 0 unless given
    if (xvbar <= 0) xvbar = 0
    call load_model(rq, xvbar, pr)
    h1 = rq%h1
    h2 = rq%h2
    angle = rq%angle
    len = rq%len
    zbnd = rq%zbnd
    ibmax = rq%ibmax
    if (rq%ibmax_b < 0) then             ! boundaries, H1 and H2 given as pressures (src/lblatm.f90:891-1086)
       do ib = 1, ibmax
          zbnd(ib) = pressure_to_altitude(pr%n, pr%z, pr%p, pr%t, pr%den(1, :), rq%pbnd(ib), rq%ref_lat, pr%re)
       end do
       h1 = pressure_to_altitude(pr%n, pr%z, pr%p, pr%t, pr%den(1, :), h1, rq%ref_lat, pr%re)
       if (h1 < 0) call fail('COMPUTED ALTITUDE VALUE OF H1 IS NEGATIVE')
       if (rq%itype == 2) then           ! (ITYPE = 3: H2 is the top of the atmosphere whatever was read)
          h2 = pressure_to_altitude(pr%n, pr%z, pr%p, pr%t, pr%den(1, :), h2, rq%ref_lat, pr%re)
          if (h2 < 0) call fail('COMPUTED ALTITUDE VALUE OF H2 IS NEGATIVE')
       end if
    end if
    if (ibmax >= 1) then
       if (zbnd(1) < pr%z(1)) then
          if (abs(zbnd(1) - pr%z(1)) <= 0.0001_dp) then
             zbnd(1) = pr%z(1)
          else
             call fail('BOUNDARIES OUTSIDE OF ATMOS')
          end if
       end if
    end if
    call reduce_path(rq, pr, h1, h2, angle, len, hmin, phi)
    if (ibmax == 0) then
       hmax = max(h1, h2)
       call autlay(rq, pr, hmin, hmax, xvbar, zbnd, ibmax)
    end if

    allocate (zp(MXPTH), pp(MXPTH), tp(MXPTH), rfp(MXPTH), denp(MXMOLF, MXPTH))
    call amerge(pr, nmol, zbnd, ibmax, h1, h2, hmin, len, zout, ioutmx, zp, pp, tp, rfp, denp, ipmax, iphmid)
    allocate (sp(ipmax), ppsum(ipmax), tpsum(ipmax), rhopsm(ipmax), amtp(MXMOLF, ipmax))
    if (h1 <= h2) then                 ! RFPATH: trace from the lowest point upwards
       iorder = 1
       ha = h1
       anglea = angle
    else
       iorder = -1
       ha = h2
       anglea = phi
    end if
    if (len == 0) then
       call findsh(pr, ha, sh, gamma)
       cpath = (pr%re + ha)*andex(ha, sh, gamma)*sin(anglea/deg)
       if (anglea <= 45.0_dp) then
          sinai = sin(anglea/deg)
          cosai = -cos(anglea/deg)
       else
          sinai = cos((90.0_dp - anglea)/deg)
          cosai = -sin((90.0_dp - anglea)/deg)
       end if
    else
       call findsh(pr, hmin, sh, gamma)
       cpath = (pr%re + hmin)*andex(hmin, sh, gamma)
       sinai = 1
       cosai = 0
    end if
    do j = 1, ipmax - 1
       call scalht(zp(j), zp(j + 1), rfp(j), rfp(j + 1), sh, gamma)
       call alayer(pr%re, gcair, nmol, zp(j), zp(j + 1), pp(j), pp(j + 1), tp(j), tp(j + 1), denp(:, j), denp(:, j + 1), &
                   sinai, cosai, cpath, sh, gamma, ds, dbend, ppsum(j), tpsum(j), rhopsm(j), amtp(:, j))
       sp(j) = ds
    end do

    ! totals along the path (for the 0.1 % zeroing rule of FPACK)
    hmid = min(h1, h2)
    amttot = 0
    do j = 1, ipmax - 1
       fac = 1
       if (len == 1 .and. zp(j + 1) <= hmid) fac = 2
       amttot(1:nmol) = amttot(1:nmol) + fac*amtp(1:nmol, j)
    end do

    ! FPACK: condense the path layers into the output layers ZOUT
    allocate (pbar(ioutmx), tbar(ioutmx), rhosum(ioutmx), sout(ioutmx), amount(nmol, ioutmx), pz(0:ioutmx), tz(0:ioutmx), &
              ipath(ioutmx))
    pbar = 0; tbar = 0; rhosum = 0; sout = 0; amount = 0; pz = 0; tz = 0; ipath = 0
    iout = 1
    pz(0) = pp(1)
    tz(0) = tp(1)
    do j = 1, ipmax - 1
       pbar(iout) = pbar(iout) + ppsum(j)
       tbar(iout) = tbar(iout) + tpsum(j)
       rhosum(iout) = rhosum(iout) + rhopsm(j)
       sout(iout) = sout(iout) + sp(j)
       amount(1:nmol, iout) = amount(1:nmol, iout) + amtp(1:nmol, j)
       if (zp(j + 1) == zout(iout + 1)) then
          pz(iout) = pp(j + 1)
          tz(iout) = tp(j + 1)
          iout = iout + 1
       end if
    end do
    if (iout /= ioutmx) call fail('FPACK: output layers do not match the merged levels')
    amtcum = 0
    iskip = 0
    do k = 1, nmol
       if (amttot(k) == 0) iskip(k) = 1
    end do
    lmax = ioutmx - 1
    allocate (out%wbrodl(lmax), out%secnta(lmax), out%altz(0:lmax))
    layers: do l = 1, ioutmx - 1
       pbar(l) = pbar(l)/rhosum(l)
       tbar(l) = tbar(l)/rhosum(l)
       rhosum(l) = rhosum(l)*1.0E+5_dp
       sumamt = 0
       do k = 1, nmol
          sumamt = sumamt + amount(k, l)
       end do
       out%wbrodl(l) = rhosum(l) - sumamt
       out%secnta(l) = sout(l)/(zout(l + 1) - zout(l))
       if (l == 1) out%altz(0) = zout(1)
       out%altz(l) = zout(l + 1)
       if (len == 1) then
          if (zout(l) < hmid) ipath(l) = 2
          if (zout(l) >= hmid .and. h1 > h2) ipath(l) = 1
          if (zout(l) >= hmid .and. h1 < h2) ipath(l) = 3
       else
          if (h1 < h2) ipath(l) = 3
          if (h1 > h2) ipath(l) = 1
       end if
       iskpt = 0
       nmol_max = nmol
       if (iskip(7) == 1) nmol_max = nmol - 1
       fac = 1
       if (ipath(l) == 2) fac = 2
       do k = 1, nmol
          if (rq%n_zero == 2) then
             if (iskip(k) /= 1) then
                if (k == 7 .or. (iemit == 1 .and. ipath(l) /= 3)) then
                   amtcum(k) = amtcum(k) + fac*amount(k, l)
                   cycle
                end if
                if (((amttot(k) - amtcum(k))/amttot(k)) > 0.001_dp) then
                   amtcum(k) = amtcum(k) + fac*amount(k, l)
                   cycle
                end if
             end if
             iskip(k) = 1
             amount(k, l) = 0
             iskpt = iskpt + 1
             if (iskpt >= nmol_max) exit layers
          else
             amtcum(k) = amtcum(k) + fac*amount(k, l)
          end if
       end do
       lmax = l
    end do layers

    ! what ATMPTH leaves for the caller (src/lblatm.f90:1300-1500): amounts below one molecule per cm2 are zeroed when the
    ! layers are also punched to TAPE7 (IPUNCH >= 1), so that no reader takes them for mixing ratios
    if (rq%ipunch >= 1) then
       do l = 1, lmax
          do k = 1, nmol
             if (amount(k, l) < 1.0_dp) amount(k, l) = 0
          end do
       end do
    end if
    out%nlay = lmax
    out%nmol = nmol
    out%angle = angle
    out%h1 = h1
    out%h2 = h2
    out%hmod = pr%hmod
    allocate (out%pbar(lmax), out%tbar(lmax), out%amount(nmol, lmax), out%pz(0:lmax), out%tz(0:lmax), out%ipath(lmax))
    out%pbar = pbar(1:lmax)
    out%tbar = tbar(1:lmax)
    out%amount = amount(:, 1:lmax)
    out%pz = pz(0:lmax)
    out%tz = tz(0:lmax)
    out%ipath = ipath(1:lmax)
    out%wbrodl = out%wbrodl(1:lmax)
    out%secnta = out%secnta(1:lmax)
    out%altz = out%altz(0:lmax)
    ib = ibmax  ! (kept for symmetry with the reference's IBMAXOUT; unused)
  end subroutine build_atm_layers

end module lblatm_front
