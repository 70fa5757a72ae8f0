! monortm_hip - stand-alone driver for layer input (IATM = 0) on the MI355X path.
!
! Reads the reference's own input files and writes its output file, so that a monoRTM user can run
! the hot path without any reference source:
!   MONORTM.IN        records 1.1-1.4   (formats: reference src/monortm_sub.F90:138-308, :402-408;
!                                        doc/monortm_instructions:27-271)
!   MONORTM_PROF.IN   records 2.1-2.1.3 (reference src/monortm.f90:380-408, :599-612), any number of
!                                        concatenated profiles, mixing ratios converted to column
!                                        amounts as in src/monortm.f90:423-483
!   TAPE3             binary line file  (parsed by the C++ side)
!   MONORTM.OUT       (reference src/monortm_sub.F90:617-675, formats 11/21 at :780-782)
!
! MI355X-first structure: ALL profiles are read first and handed to the GPU as ONE batch through the C ABI
! (monortm_hip_modm / monortm_hip_rtm with nprof > 1) instead of one MODM call per profile; the results
! are then written profile by profile.  Boundary emissivity / reflectivity files (in/EMISSION, in/REFLECTION) and profile
! scaling (NMOL_SCAL) are handled as the reference handles them.  IATM = 1 decks go through the own layering front end
! lblatm_front.f90 (built-in model atmospheres, slant paths H1 / H2 / ANGLE, automatic or given boundary altitudes); user
! supplied profiles (MODEL = 0) and cross sections are not part of this driver (use the reference's driver with the
! drop-in modules for those, INTEGRATION.md).
module monortm_driver_io
  use, intrinsic :: iso_c_binding
  use lblatm_front, only: atm_request, atm_layers, read_atm_request, build_atm_layers
  implicit none
  integer, parameter :: dp = c_double
  integer, parameter :: MXMOL = 39, NCONT = 5
  integer, parameter :: index_cont(NCONT) = (/1, 2, 3, 7, 22/)

  type run_config
     integer :: ihirac = 0, icntnm = 0, iemit = 0, iplot = 0, iatm = 0, iod = 0, ixsect = 0, ispd = 0, ibrd = 0
     real(dp) :: fac(7) = 1.0_dp
     real(dp) :: v1 = 0, v2 = 0, dvset = 0
     integer :: nwn = 0
     real(dp), allocatable :: wn(:)
     real(dp) :: tbound = 0, bndemi(3) = 0, bndrfl(3) = 0
     ! boundary emissivity / reflectivity tables of in/EMISSION, in/REFLECTION (BNDEMI(1) < 0 / BNDRFL(1) < 0;
     ! reference src/monortm_sub.F90:1-29, :317-335): V1, V2, DV, count, values
     real(dp) :: v1emis = 0, v2emis = 0, dvemis = 0, v1rflt = 0, v2rflt = 0, dvrflt = 0
     integer :: nlimem = 0, nlimrf = 0
     real(dp), allocatable :: zemis(:), zrflt(:)
     ! profile scaling, record 1.3.a / 1.3.b (reference src/monortm_sub.F90:208-216, :937-1046)
     integer :: nmol_scal = 0
     character(len=1) :: hmol_scal(64) = ' '
     real(dp) :: xmol_scal(64) = 0
     type(atm_request) :: atm                  ! records 3.1 - 3.3B (IATM = 1)
  end type run_config

  type profile_set
     integer :: nprof = 0, nlay_max = 0, nmol = 0
     integer, allocatable :: nlay(:), irt(:)
     real(dp), allocatable :: angle(:)
     real(dp), allocatable :: p(:, :), t(:, :), clw(:, :), wbrodl(:, :)     ! (nlay_max, nprof)
     real(dp), allocatable :: tz(:, :)                                      ! (0:nlay_max, nprof)
     real(dp), allocatable :: wkl(:, :, :)                                  ! (nmol, nlay_max, nprof)
     ! cross-section molecules (IXSECT = 1, records 2.2.x): amounts (nxs, nlay_max, nprof)
     integer :: nxs = 0
     real(dp), allocatable :: xamnt(:, :, :)
  end type profile_set

contains

  subroutine die(msg)
    character(len=*), intent(in) :: msg
    write (*, '(a)') ' monortm_hip: '//msg
    stop 1
  end subroutine die

  ! ---------------------------------------------------------------- MONORTM.IN
  subroutine read_monortm_in(fname, cfg)
    character(len=*), intent(in) :: fname
    type(run_config), intent(out) :: cfg
    character(len=120) :: line
    integer :: u, ios, ilnflg, nmol_scal, i
    real(dp) :: sample, alfal0, avmass, dptmin, dptfac, dvout, xvmid, tst

    open (newunit=u, file=fname, status='old', action='read', iostat=ios)
    if (ios /= 0) call die('cannot open '//trim(fname))
    do                                                   ! record 1.1: the line starting with '$'
       read (u, '(a)', iostat=ios) line
       if (ios /= 0) call die('EOF on '//trim(fname)//' before record 1.1')
       if (line(1:1) == '%') call die('end-of-data mark before any run deck in '//trim(fname))
       if (line(1:1) == '$') exit
    end do
    read (u, '(4X,I1,9X,I1,9X,I1,14X,I1,9X,I1,14X,I1,4X,I1,16X,I4,I4)', iostat=ios) cfg%ihirac, cfg%icntnm, &
         cfg%iemit, cfg%iplot, cfg%iatm, cfg%iod, cfg%ixsect, cfg%ispd, cfg%ibrd          ! record 1.2
    if (ios /= 0) call die('error reading record 1.2')
    if (cfg%ispd == 1) call die('The ISPD=1 option is no longer valid: build the appropriate TAPE3')
    if (cfg%iemit == 3) call die('derivatives (IEMIT=3) are not handled by monoRTM')
    select case (cfg%icntnm)                             ! continuum presets, record 1.2 / 1.2a
    case (0); cfg%fac = 0
    case (1); cfg%fac = 1
    case (2); cfg%fac = 1; cfg%fac(1) = 0
    case (3); cfg%fac = 1; cfg%fac(2) = 0
    case (4); cfg%fac = 1; cfg%fac(1:2) = 0
    case (5); cfg%fac = 1; cfg%fac(7) = 0
    case (6); read (u, *, iostat=ios) cfg%fac
       if (ios /= 0) call die('error reading record 1.2a (continuum scale factors)')
    case default; call die('invalid ICNTNM')
    end select
    if (cfg%iemit == 2) read (u, '(a)') line              ! record 1.2.1 (unused by monoRTM)
    read (u, '(8E10.3,4X,I1,5x,e10.3,i5)', iostat=ios) cfg%v1, cfg%v2, sample, cfg%dvset, alfal0, avmass, dptmin, &
         dptfac, ilnflg, dvout, nmol_scal                                                  ! record 1.3
    if (ios /= 0) call die('error reading record 1.3')
    if (ilnflg > 0) call die('ILNFLG MUST BE 0 FOR MONORTM')
    if (nmol_scal > 0) then                               ! records 1.3.a / 1.3.b
       if (nmol_scal > 38) call die(' nmol_scal .gt. 38 ')
       cfg%nmol_scal = nmol_scal
       read (u, '(64a1)', iostat=ios) cfg%hmol_scal(1:nmol_scal)
       if (ios /= 0) call die('error reading record 1.3.a (HMOL_SCAL)')
       read (u, '(7e15.7,/,(8e15.7,/))', iostat=ios) cfg%xmol_scal(1:nmol_scal)
       if (ios /= 0) call die('error reading record 1.3.b (XMOL_SCAL)')
    end if
    if (cfg%v1 < 0 .or. cfg%v2 < 0) then                  ! records 1.3.1 / 1.3.2: explicit wavenumbers
       read (u, '(I8)', iostat=ios) cfg%nwn
       if (ios /= 0 .or. cfg%nwn < 1) call die('error reading record 1.3.1')
       if (cfg%nwn > 80000) call die('NUMBER OF WAVENUMBERS EXCEEDS LIMIT (NWNMX = 80000)')
       allocate (cfg%wn(cfg%nwn))
       do i = 1, cfg%nwn
          read (u, '(E19.7)', iostat=ios) cfg%wn(i)
          if (ios /= 0) call die('error reading record 1.3.2')
       end do
       cfg%dvset = 0
    else if (cfg%dvset /= 0) then
       if (cfg%dvset < 0) call die('MONORTM REQUIRES POSITIVE DVSET')
       cfg%nwn = nint(((cfg%v2 - cfg%v1)/cfg%dvset) + 1.0_dp)
       if (cfg%nwn > 80000) call die('NUMBER OF WAVENUMBERS EXCEEDS LIMIT (NWNMX = 80000)')
       allocate (cfg%wn(cfg%nwn))
       do i = 1, cfg%nwn
          cfg%wn(i) = cfg%v1 + (i - 1)*cfg%dvset
       end do
    else
       if (cfg%v1 /= cfg%v2) call die('AMBIGUITY IN THE WAVENUMBER: V1 /= V2 with DVSET = 0')
       cfg%nwn = 1
       allocate (cfg%wn(1))
       cfg%wn(1) = cfg%v1
    end if
    read (u, '(8E10.3)', iostat=ios) cfg%tbound, cfg%bndemi, cfg%bndrfl                    ! record 1.4
    if (ios /= 0) call die('error reading record 1.4')
    xvmid = (cfg%v1 + cfg%v2)/2
    if (cfg%bndemi(1) < 0) then                           ! record 1.4 continued: tabulated emissivities
       call read_table('in/EMISSION', cfg%v1emis, cfg%v2emis, cfg%dvemis, cfg%nlimem, cfg%zemis, 'EMISSION')
    else
       tst = cfg%bndemi(1) + cfg%bndemi(2)*xvmid + cfg%bndemi(3)*xvmid*xvmid
       if (tst < 0 .or. tst > 1) call die('BNDEMI OUTSIDE PHYSICAL RANGE')
    end if
    if (cfg%bndrfl(1) < 0) then
       call read_table('in/REFLECTION', cfg%v1rflt, cfg%v2rflt, cfg%dvrflt, cfg%nlimrf, cfg%zrflt, 'REFLECTION')
    else
       tst = cfg%bndrfl(1) + cfg%bndrfl(2)*xvmid + cfg%bndrfl(3)*xvmid*xvmid
       if (tst < 0 .or. tst > 1) call die('BNDRFL OUTSIDE PHYSICAL RANGE')
    end if
    if (cfg%iatm == 1) call read_atm_request(u, cfg%atm)      ! records 3.1 - 3.3B follow record 1.4
    close (u)
  end subroutine read_monortm_in

  ! READEM / READRF (reference src/monortm_sub.F90:1-29): header 3E10.3,5X,I5 then one E15.7 value per line
  subroutine read_table(fname, v1t, v2t, dvt, n, z, what)
    character(len=*), intent(in) :: fname, what
    real(dp), intent(out) :: v1t, v2t, dvt
    integer, intent(out) :: n
    real(dp), allocatable, intent(out) :: z(:)
    integer :: u, ios, i
    open (newunit=u, file=fname, status='old', action='read', iostat=ios)
    if (ios /= 0) call die('EXIT; ERROR OPENING '//what//' FILE')
    read (u, '(3E10.3,5X,I5)', iostat=ios) v1t, v2t, dvt, n
    if (ios /= 0 .or. n < 1 .or. n > 4040) call die('INCONSISTENT DATA OR ERROR OPENING IN READ'//what(1:2))
    allocate (z(n + 1))
    z = 0
    do i = 1, n
       read (u, '(E15.7)', iostat=ios) z(i)
       if (ios /= 0) call die('INCONSISTENT DATA OR ERROR OPENING IN READ'//what(1:2))
    end do
    close (u)
  end subroutine read_table

  ! EMISFN / REFLFN (reference src/monortm_sub.F90:426-491) with LINTCO (:493-501): tabulated values when A < 0 - element
  ! NELMNT = INT((VI-V1)/DV) is taken as the value AT V1 + DV*NELMNT, exactly as the reference indexes it - else the
  ! constant or the quadratic in the wavenumber
  function boundary_fn(vi, abc, v1t, v2t, dvt, n, z, what) result(val)
    real(dp), intent(in) :: vi, abc(3), v1t, v2t, dvt
    integer, intent(in) :: n
    real(dp), allocatable, intent(in) :: z(:)
    character(len=*), intent(in) :: what
    real(dp) :: val, v1a, v1b, zdel, zcept
    integer :: nel
    if (abc(1) < 0) then
       nel = int((vi - v1t)/dvt)
       if (nel <= 0 .or. nel >= n) then
          write (*, *) 'Frequency range of calculation exceeded ', what, ' input.'
          write (*, *) ' VI = ', vi, ' V1 = ', v1t, ' V2 = ', v2t
          call die('ERROR IN '//what)
       end if
       v1a = v1t + dvt*nel
       v1b = v1t + dvt*(nel + 1)
       zdel = (z(nel + 1) - z(nel))/(v1b - v1a)
       zcept = z(nel) - zdel*v1a
       val = zdel*vi + zcept
    else if (abc(2) == 0 .and. abc(3) == 0) then
       val = abc(1)
    else
       val = abc(1) + abc(2)*vi + abc(3)*vi*vi
    end if
  end function boundary_fn

  ! profil_scal_sub (reference src/monortm_sub.F90:937-1046) for one profile.  Like the reference, the derived scale
  ! factor REPLACES xmol_scal(m) in the run configuration (there: COMMON /profil_scal/), so that with the 'C', 'M', 'P', 'D'
  ! options a second profile starts from the first profile's factor - reproduced literally.
  subroutine scale_profile(cfg, nmol, nlay, wkl, wbrodl)
    type(run_config), intent(inout) :: cfg
    integer, intent(in) :: nmol, nlay
    real(dp), intent(inout) :: wkl(:, :)
    real(dp), intent(in) :: wbrodl(:)
    real(dp) :: wmt(64), wsum_brod, wsum_drair, x
    integer :: m, l
    wmt = 0
    do m = 1, nmol
       do l = 1, nlay
          wmt(m) = wmt(m) + wkl(m, l)
       end do
    end do
    wsum_brod = 0
    do l = 1, nlay
       wsum_brod = wsum_brod + wbrodl(l)
    end do
    wsum_drair = merge(0.0_dp, wsum_brod, nmol >= 22)
    do m = 2, nmol
       wsum_drair = wsum_drair + wmt(m)
    end do
    do m = 1, cfg%nmol_scal
       if (m > size(wkl, 1)) call die('NMOL_SCAL exceeds the number of molecules of the profile')
       x = cfg%xmol_scal(m)
       select case (cfg%hmol_scal(m))
       case (' '); cfg%xmol_scal(m) = 1
       case ('0'); cfg%xmol_scal(m) = 0
       case ('1'); cfg%xmol_scal(m) = x
       case ('C', 'c'); cfg%xmol_scal(m) = x/wmt(m)
       case ('M', 'm')
          if (.not. wsum_drair > 0) call die('mixing ratio failure: wsum_drair = 0.')
          cfg%xmol_scal(m) = x/(wmt(m)/wsum_drair)
       case ('P', 'p')
          if (m /= 1) call die(' (hmol_scal(m).eq."P" .and. m.ne.1) ')
          cfg%xmol_scal(1) = (x/2.99150e-23_dp)/wmt(1)
       case ('D', 'd'); cfg%xmol_scal(m) = (x*2.68678e16_dp)/wmt(m)
       end select
       wmt(m) = 0
       do l = 1, nlay
          wkl(m, l) = wkl(m, l)*cfg%xmol_scal(m)
          wmt(m) = wmt(m) + wkl(m, l)
       end do
    end do
  end subroutine scale_profile

  ! ---------------------------------------------------------------- IATM = 1: layers from the own LBLATM front end
  ! One profile, as the reference produces per run deck (src/monortm_sub.F90:352-360: CLW is not returned by LBLATM and is
  ! set to zero; P, T, WKL, WBRODL, TZ of COMMON /PATHD/ are LBLATM's PBAR, TBAR, AMOUNT, WN2L, TZ; ANGLE of COMMON /MANE/)
  subroutine profiles_from_lblatm(cfg, ps)
    type(run_config), intent(in) :: cfg
    type(profile_set), intent(out) :: ps
    type(atm_layers) :: lay
    integer :: n
    call build_atm_layers(cfg%atm, cfg%iemit, lay)
    n = lay%nlay
    if (n > 200) call die('NLAYRS MUST BE LESS THAN 200')
    ps%nprof = 1
    ps%nlay_max = n
    ps%nmol = lay%nmol
    allocate (ps%nlay(1), ps%irt(1), ps%angle(1), ps%p(n, 1), ps%t(n, 1), ps%clw(n, 1), ps%wbrodl(n, 1), ps%tz(0:n, 1), &
              ps%wkl(lay%nmol, n, 1))
    ps%nlay(1) = n
    ps%angle(1) = lay%angle
    if (lay%angle > 90) ps%irt(1) = 1
    if (lay%angle < 90) ps%irt(1) = 3
    if (lay%angle == 90) ps%irt(1) = 2
    ps%p(:, 1) = lay%pbar
    ps%t(:, 1) = lay%tbar
    ps%clw(:, 1) = 0
    ps%wbrodl(:, 1) = lay%wbrodl
    ps%tz(0:n, 1) = lay%tz(0:n)
    ps%wkl(:, :, 1) = lay%amount
  end subroutine profiles_from_lblatm

  ! ---------------------------------------------------------------- MONORTM_PROF.IN
  subroutine read_profiles(fname, ps, ixsect, xv1, xv2)
    use lblparams, only: MX_XS, MXLAY
    character(len=*), intent(in) :: fname
    type(profile_set), intent(out) :: ps
    integer, intent(in) :: ixsect
    real(dp), intent(in) :: xv1, xv2                    ! smallest / largest wavenumber of the run (XSREAD keeps the regions inside)
    integer :: IXMAX, IXMOLS, IXINDX(MX_XS)
    real(dp) :: XAMNT(MX_XS, MXLAY)
    common /PATHX/ IXMAX, IXMOLS, IXINDX, XAMNT
    integer :: ixmols_in, ixsbin, ifrmx, nlayxs, ixmol
    real(dp) :: secntx, xa(MX_XS), wbrodx
    character(len=120) :: line
    logical :: xs_read
    character(len=8) :: hmod(2)
    integer :: u, ios, pass, ip, il, k, iform, nlayrs, nmol, len_, ipath, m
    real(dp) :: secnt0, h1, h2, angle, secnt, altz0, pz0, tz0, altz, pz, tzl, clw, pl, tl
    real(dp) :: wk(MXMOL), wbrod, wdnsty, wmxrat, wdrair

    open (newunit=u, file=fname, status='old', action='read', iostat=ios)
    if (ios /= 0) call die('cannot open '//trim(fname))
    xs_read = .false.
    do pass = 1, 2                                       ! pass 1 sizes the batch, pass 2 fills it
       rewind (u)
       ip = 0
       do
          read (u, '(1X,I1,I3,I5,F10.6,2A8,4X,F8.2,4X,F8.2,5X,F8.3,5X,I2)', iostat=ios) iform, nlayrs, nmol, secnt0, hmod, &
               h1, h2, angle, len_                                                         ! record 2.1
          if (ios /= 0) exit
          if (nmol == 0) nmol = 7
          if (nlayrs < 1 .or. nmol < 7 .or. nmol > MXMOL) call die('bad NLAYRS / NMOL in record 2.1')
          ip = ip + 1
          if (pass == 1) then
             ps%nlay_max = max(ps%nlay_max, nlayrs)
             if (ip == 1) ps%nmol = nmol
             if (nmol /= ps%nmol) call die('profiles with different NMOL in one MONORTM_PROF.IN are not batched')
          else
             ps%nlay(ip) = nlayrs
             ps%angle(ip) = angle
             if (angle > 90) ps%irt(ip) = 1              ! space-based observer (reference monortm.f90:383-385)
             if (angle < 90) ps%irt(ip) = 3
             if (angle == 90) ps%irt(ip) = 2
          end if
          do il = 1, nlayrs                              ! records 2.1.1-2.1.3
             if (il == 1) then
                if (iform == 0) then
                   read (u, '(3f10.4,3x,i2,1x,2(f7.2,f8.3,f7.2),f7.3)', iostat=ios) pl, tl, secnt, ipath, altz0, pz0, tz0, &
                        altz, pz, tzl, clw
                else
                   read (u, '(e15.7,2f10.4,3x,i2,1x,2(f7.2,f8.3,f7.2),f7.3)', iostat=ios) pl, tl, secnt, ipath, altz0, pz0, &
                        tz0, altz, pz, tzl, clw
                end if
             else
                if (iform == 0) then
                   read (u, '(3f10.4,3x,i2,1x,22x,1(f7.2,f8.3,f7.2),f7.3)', iostat=ios) pl, tl, secnt, ipath, altz, pz, tzl, clw
                else
                   read (u, '(e15.7,2f10.4,3x,i2,1x,22x,1(f7.2,f8.3,f7.2),f7.3)', iostat=ios) pl, tl, secnt, ipath, altz, pz, &
                        tzl, clw
                end if
             end if
             if (ios /= 0) call die('error reading a layer record of '//trim(fname))
             wk = 0
             read (u, '(8E15.7)', iostat=ios) wk(1:7), wbrod
             if (ios /= 0) call die('error reading layer amounts')
             if (nmol > 7) read (u, '(8E15.7)', iostat=ios) wk(8:nmol)
             if (ios /= 0) call die('error reading layer amounts (molecules 8..NMOL)')
             if (pass == 1) cycle
             ! amounts below 1 are mixing ratios w.r.t. dry air: convert with the dry-air column
             ! (reference src/monortm.f90:423-483)
             wdnsty = wbrod
             wmxrat = 0
             do m = 2, nmol
                if (wk(m) > 1) then
                   wdnsty = wdnsty + wk(m)
                else
                   wmxrat = wmxrat + wk(m)
                end if
             end do
             if (wbrod < 1 .and. wbrod /= 0) call die('WBROAD must be a column density')
             if (wdnsty == 0 .and. wmxrat /= 0) call die('WMXRAT AND/OR WDNSTY NOT PROPERLY SPECIFIED IN PATH')
             if (wmxrat >= 1) call die('WMXRAT EXCEEDS 1.0')
             wdrair = wdnsty/(1 - wmxrat)
             if (wk(1) <= 1 .and. wk(1) /= 0 .and. wdrair == 0) call die('WMXRAT NOT PROPERLY SPECIFIED IN PATH')
             do m = 1, nmol
                if (wk(m) < 1) wk(m) = wk(m)*wdrair
             end do
             ps%p(il, ip) = pl
             ps%t(il, ip) = tl
             ps%clw(il, ip) = clw
             ps%wbrodl(il, ip) = wbrod
             ps%wkl(1:nmol, il, ip) = wk(1:nmol)
             if (il == 1) ps%tz(0, ip) = tz0
             ps%tz(il, ip) = tzl
          end do
          if (ixsect >= 1) then
             ! records 2.2 - 2.2.5 (reference src/monortm.f90:492-530): number of cross-section molecules, their names, a header
             ! and per layer a layer record + the amounts (8E15.7: seven amounts and the broadening gas, then the rest)
             read (u, '(I5,5X,I5)', iostat=ios) ixmols_in, ixsbin
             if (ios /= 0 .or. ixmols_in < 1 .or. ixmols_in > MX_XS) call die('bad record 2.2 (IXMOLS) in '//trim(fname))
             if (pass == 2 .and. .not. xs_read) then
                IXMOLS = ixmols_in
                call XSREAD(u, xv1, xv2)                 ! reads the names (record 2.2.1) and FSCDXS
                xs_read = .true.
             else                                        ! (the reference calls XSREAD for every profile, which DOUBLES its
                read (u, '(A)', iostat=ios) line         !  region count, src/monortm_sub.F90:1365: the names are skipped here)
                if (ixmols_in > 7) read (u, '(A)', iostat=ios) line
             end if
             read (u, '(1X,I1,I3,I5,F10.2)', iostat=ios) ifrmx, nlayxs, ixmol, secntx
             if (ios /= 0) call die('bad record 2.2.3 in '//trim(fname))
             if (ixmol == 0) call die(' PATH - IXMOL 0 ')
             if (ixmol /= ixmols_in) call die(' PATH - IXMOL .NE. IXMOLS ')
             if (nlayrs /= nlayxs) call die(' PATH - NLAYRS .NE. NLAYXS ')
             if (pass == 1) then
                if (ip == 1) ps%nxs = ixmols_in
                if (ixmols_in /= ps%nxs) call die('profiles with different cross-section molecules are not batched')
             end if
             do il = 1, nlayxs
                read (u, '(A)', iostat=ios) line         ! layer record (format 910 / 915): the values repeat record 2.1.1
                xa = 0
                read (u, '(8E15.7)', iostat=ios) xa(1:7), wbrodx    ! (XAMNT(M,L),M=1,7),WBRODX
                if (ios /= 0) call die('error reading the cross-section amounts')
                if (ixmol > 7) read (u, '(8E15.7)', iostat=ios) xa(8:ixmol)
                if (ios /= 0) call die('error reading the cross-section amounts (molecules 8..)')
                if (pass == 2) ps%xamnt(1:ixmol, il, ip) = xa(1:ixmol)
             end do
          end if
       end do
       if (pass == 1) then
          ps%nprof = ip
          if (ip == 0) call die('NO PROFILE FOUND IN '//trim(fname))
          k = ps%nlay_max
          allocate (ps%nlay(ip), ps%irt(ip), ps%angle(ip))
          allocate (ps%p(k, ip), ps%t(k, ip), ps%clw(k, ip), ps%wbrodl(k, ip), ps%tz(0:k, ip), ps%wkl(ps%nmol, k, ip))
          ps%p = 0; ps%t = 0; ps%clw = 0; ps%wbrodl = 0; ps%tz = 0; ps%wkl = 0; ps%irt = 3
          if (ps%nxs > 0) then
             allocate (ps%xamnt(ps%nxs, k, ip))
             ps%xamnt = 0
          end if
       end if
    end do
    close (u)
  end subroutine read_profiles

end module monortm_driver_io

program monortm_hip
  use, intrinsic :: iso_c_binding
  use monortm_hip_c
  use monortm_driver_io
  use xsec_hip, only: xsec_tables_to_device
  use netcdf3_writer, only: write_monortm_nc
  implicit none
  character(len=8), parameter :: hmolc(MXMOL) = (/ &
       '  H2O   ', '  CO2   ', '   O3   ', '  N2O   ', '   CO   ', '  CH4   ', '   O2   ', '   NO   ', &
       '  SO2   ', '  NO2   ', '  NH3   ', ' HNO3   ', '   OH   ', '   HF   ', '  HCL   ', '  HBR   ', &
       '   HI   ', '  CLO   ', '  OCS   ', ' H2CO   ', ' HOCL   ', '   N2   ', '  HCN   ', ' CH3CL  ', &
       ' H2O2   ', ' C2H2   ', ' C2H6   ', '  PH3   ', ' COF2   ', '  SF6   ', '  H2S   ', ' HCOOH  ', &
       '  HO2   ', '   O+   ', ' ClONO2 ', '   NO+  ', '  HOBr  ', ' C2H4   ', ' CH3OH  '/)
  real(dp), parameter :: CLIGHT = 2.99792458E+10_dp
  type(run_config) :: cfg
  type(profile_set) :: ps
  real(dp), allocatable, target :: tmr(:, :)
  real(dp), allocatable :: o(:, :, :), obm(:, :, :, :), oc(:, :, :, :), oclw(:, :, :), odx(:, :, :)
  real(dp), allocatable :: rup(:, :), rdn(:, :), trtot(:, :), rad(:, :), tb(:, :), emiss(:, :), reflc(:, :), tmpsfc(:)
  real(dp), allocatable :: otot_by_mol(:)
  real(dp) :: wk_tot(MXMOL), otot, freq, wvcolmn, clwcolmn, xvi
  integer(c_int), allocatable :: nlay_c(:), irt_c(:)
  integer(c_int) :: rc
  character(kind=c_char) :: cpath(6)
  character(len=12) :: wnunits
  character(len=8) :: cmol(MXMOL)
  integer :: id_mol(MXMOL), kount, ip, iw, j, ik, im, u, nwn, lm, np, nm, slot
  logical :: giga
  character(len=16) :: envv
  integer :: envl, ngpu, ios

  call read_monortm_in('MONORTM.IN', cfg)
  if (cfg%iatm /= 0 .and. cfg%iatm /= 1) call die('IATM must be 0 (layer input) or 1 (LBLATM front end)')
  if (cfg%ixsect /= 0 .and. cfg%iatm == 1) call die('IXSECT=1 with IATM=1 (cross-section profiles through LBLATM) is not built: '// &
       'give the layer amounts in MONORTM_PROF.IN (IATM=0)')
  if (cfg%iatm == 1) then
     call profiles_from_lblatm(cfg, ps)
  else
     call read_profiles('MONORTM_PROF.IN', ps, cfg%ixsect, minval(cfg%wn(1:cfg%nwn)), maxval(cfg%wn(1:cfg%nwn)))
  end if
  nwn = cfg%nwn; lm = ps%nlay_max; np = ps%nprof; nm = ps%nmol
  ! MONORTM_LAYERS_ONLY=1: write the layer quantities the hot path would receive (full precision, the content of the
  ! reference's TAPE7 / MONORTM_PROF.IN) to LAYERS.OUT and stop before anything touches the GPU - used to check the
  ! input side (IATM = 1 front end, profile scaling) on machines without one
  call get_environment_variable('MONORTM_LAYERS_ONLY', envv, envl)
  if (envl > 0) then                                    ! (before any profile scaling: what the reference punches to TAPE7)
     open (newunit=u, file='LAYERS.OUT', status='replace', action='write')
     do ip = 1, np
        write (u, '(3i6,f12.5)') ip, ps%nlay(ip), nm, ps%angle(ip)
        do j = 1, ps%nlay(ip)
           write (u, '(i5,1p,4e24.15)') j, ps%p(j, ip), ps%t(j, ip), ps%tz(j - 1, ip), ps%tz(j, ip)
           write (u, '(1p,8e24.15)') ps%wkl(1:nm, j, ip), ps%wbrodl(j, ip)
        end do
     end do
     close (u)
     write (*, '(a)') ' monortm_hip: LAYERS.OUT written (MONORTM_LAYERS_ONLY)'
     stop
  end if
  write (*, '(a,i6,a,i4,a,i6,a)') ' monortm_hip:', np, ' profile(s), up to', lm, ' layers,', nwn, ' wavenumbers'

  cpath = (/'T', 'A', 'P', 'E', '3', c_null_char/)
  ! MONORTM_NGPU=n: shard the profiles over n GPUs of this node (0 = all visible) - the reference's independent-profile
  ! loop (src/monortm.f90:357) is the one parallel axis; unset or 1: the current device
  call get_environment_variable('MONORTM_NGPU', envv, envl)
  ngpu = 1
  if (envl > 0) read (envv, *, iostat=ios) ngpu
  if (envl > 0 .and. ios /= 0) call die('MONORTM_NGPU must be an integer')
  if (ngpu == 1) then
     rc = monortm_hip_init(cpath, cfg%wn(1), cfg%wn(nwn), 1_c_int, hip_real_kind, -1_c_int, hip_ctx)
  else
     rc = monortm_hip_init_multi(cpath, cfg%wn(1), cfg%wn(nwn), 1_c_int, hip_real_kind, int(ngpu, c_int), hip_ctx)
  end if
  if (rc /= 0) call hip_fail('monortm_hip_init', rc)
  if (ngpu /= 1) write (*, '(a,i3,a)') ' monortm_hip: profiles sharded over', monortm_hip_device_count(hip_ctx), ' device context(s)'

  allocate (o(nwn, lm, np), obm(nwn, nm, lm, np), oc(nwn, NCONT, lm, np), oclw(nwn, lm, np))
  allocate (rup(nwn, np), rdn(nwn, np), trtot(nwn, np), rad(nwn, np), tb(nwn, np), tmr(nwn, np))
  allocate (emiss(nwn, np), reflc(nwn, np), tmpsfc(np), nlay_c(np), irt_c(np))
  nlay_c = int(ps%nlay, c_int)
  irt_c = int(ps%irt, c_int)
  tmpsfc = cfg%tbound
  tb = 0
  do iw = 1, nwn                                        ! EMISS_REFLEC: EMISFN / REFLFN per wavenumber
     xvi = cfg%wn(iw)
     reflc(iw, :) = boundary_fn(xvi, cfg%bndrfl, cfg%v1rflt, cfg%v2rflt, cfg%dvrflt, cfg%nlimrf, cfg%zrflt, 'REFLFN')
     emiss(iw, :) = boundary_fn(xvi, cfg%bndemi, cfg%v1emis, cfg%v2emis, cfg%dvemis, cfg%nlimem, cfg%zemis, 'EMISFN')
  end do
  if (cfg%nmol_scal > 0) then                           ! profile scaling, in profile order (src/monortm.f90:539)
     do ip = 1, np
        call scale_profile(cfg, nm, ps%nlay(ip), ps%wkl(:, :, ip), ps%wbrodl(:, ip))
     end do
  end if

  ! one batched pass of the hot path over all profiles
  allocate (odx(nwn, lm, np))
  odx = 0
  if (cfg%ixsect >= 1 .and. ps%nxs > 0) then
     call xsec_tables_to_device(hip_ctx)                ! the regions XSREAD kept -> the context (every device of it)
     rc = monortm_hip_modm_xs(hip_ctx, int(np, c_int), int(nwn, c_int), cfg%wn, cfg%dvset, nlay_c, int(lm, c_int), &
          int(nm, c_int), ps%p, ps%t, ps%clw, ps%wkl, ps%wbrodl, cfg%fac, 1.0_dp, 1.0_dp, 0.0_dp, int(cfg%ibrd, c_int), &
          1_c_int, ps%xamnt, odx, o, obm, oc, oclw)
  else
     rc = monortm_hip_modm(hip_ctx, int(np, c_int), int(nwn, c_int), cfg%wn, cfg%dvset, nlay_c, int(lm, c_int), int(nm, c_int), &
          ps%p, ps%t, ps%clw, ps%wkl, ps%wbrodl, cfg%fac, 1.0_dp, 1.0_dp, 0.0_dp, int(cfg%ibrd, c_int), 0_c_int, &
          o, obm, oc, oclw)
  end if
  if (rc /= 0) call hip_fail('MODM', rc)
  rc = monortm_hip_rtm(hip_ctx, int(np, c_int), int(nwn, c_int), cfg%wn, nlay_c, int(lm, c_int), irt_c, int(cfg%iplot, c_int), &
       ps%t, ps%tz, o, tmpsfc, emiss, reflc, rup, rdn, trtot, rad, tb, c_loc(tmr))
  if (rc /= 0) call hip_fail('RTM', rc)

  ! ---------------------------------------------------------------- MONORTM.OUT
  ! molecule columns: those with a non-zero total column in the FIRST profile (reference STOREOUT :600-612);
  ! with fewer than 22 molecules the broadening gas is reported in the N2 slot
  wk_tot = 0
  do im = 1, nm
     wk_tot(im) = sum(ps%wkl(im, 1:ps%nlay(1), 1))
  end do
  if (nm < 22) wk_tot(22) = sum(ps%wbrodl(1:ps%nlay(1), 1))
  kount = 0
  do im = 1, MXMOL
     if (wk_tot(im) > 0) then
        kount = kount + 1
        id_mol(kount) = im
        cmol(kount) = hmolc(im)
     end if
  end do
  allocate (otot_by_mol(kount))
  giga = cfg%wn(1) < 100
  wnunits = 'FREQ(cm-1)'
  if (giga) wnunits = 'FREQ(GHz)'
  open (newunit=u, file='MONORTM.OUT', status='replace', action='write')
  do ip = 1, np
     write (u, '(a)') 'MONORTM RESULTS:'
     write (u, '(a)') '----------------'
     write (u, '(a5,I8,101x,a42)') 'NWN :', nwn, 'Molecular Optical Depths -->'
     write (u, '(a5,a10,2a11,a22,a8,2a8,3a8,a9,36a12)') 'PROF ', wnunits, 'BT(K) ', 'TMR(K)', '  RAD(W/cm2_ster_cm-1)', &
          'TRANS', 'PWV', 'CLW', 'TBOUND', 'EMIS', 'REFL', 'ANGLE', 'TOTAL_OD', cmol(1:kount), 'XSEC_OD'
     wvcolmn = sum(ps%wkl(1, 1:ps%nlay(ip), ip))*2.99150e-23_dp        ! INTEGR, reference monortm_sub.F90:831-845
     clwcolmn = sum(ps%clw(1:ps%nlay(ip), ip))
     do iw = 1, nwn
        freq = cfg%wn(iw)
        if (giga) freq = cfg%wn(iw)*CLIGHT/1.E9_dp
        otot = 0
        otot_by_mol = 0
        do j = 1, ps%nlay(ip)
           otot = otot + o(iw, j, ip)
           do ik = 1, kount
              im = id_mol(ik)
              if (im <= nm) otot_by_mol(ik) = otot_by_mol(ik) + obm(iw, im, j, ip)
              do slot = 1, NCONT
                 if (index_cont(slot) == im) otot_by_mol(ik) = otot_by_mol(ik) + oc(iw, slot, j, ip)
              end do
           end do
        end do
        write (u, '(i5,f10.3,2f11.5,1p,E21.9,0p,f9.5,2f8.4,3f8.2,f9.3,1p,36E12.4)') ip, freq, tb(iw, ip), tmr(iw, ip), &
             rad(iw, ip), trtot(iw, ip), wvcolmn, clwcolmn, tmpsfc(ip), emiss(iw, ip), reflc(iw, ip), ps%angle(ip), otot, &
             otot_by_mol(1:kount), sum(odx(iw, 1:ps%nlay(ip), ip))          ! ODXTOT (reference monortm_sub.F90:651)
     end do
  end do
  close (u)
  if (cfg%iod == 1) call write_layer_od()
  ! MONORTM_NETCDF=1: the per-profile netCDF files of the reference's USENETCDF build (src/monortm_sub.F90:698-778)
  call get_environment_variable('MONORTM_NETCDF', envv, envl)
  if (envl > 0 .and. envv(1:1) == '1') call write_netcdf()
  call monortm_hip_finalize(hip_ctx)
  write (*, '(a)') ' monortm_hip: MONORTM.OUT written'

contains

  subroutine write_netcdf()                              ! MONORTM.NNNNN.nc, one per profile (reference :698-778)
    real(dp), allocatable :: fr(:), ot(:), obym(:, :), odxt(:), obl(:, :, :)
    character(len=16) :: ncname
    integer :: jp, jw, jl, jk, jm, js, nl
    do jp = 1, np
       nl = ps%nlay(jp)
       allocate (fr(nwn), ot(nwn), obym(kount, nwn), odxt(nwn), obl(nwn, kount, nl))
       do jw = 1, nwn
          fr(jw) = cfg%wn(jw)
          if (giga) fr(jw) = cfg%wn(jw)*CLIGHT/1.E9_dp
          ot(jw) = 0
          obym(:, jw) = 0
          do jl = 1, nl                                   ! (the sums of the MONORTM.OUT line, same order)
             ot(jw) = ot(jw) + o(jw, jl, jp)
             do jk = 1, kount
                jm = id_mol(jk)
                if (jm <= nm) obym(jk, jw) = obym(jk, jw) + obm(jw, jm, jl, jp)
                do js = 1, NCONT
                   if (index_cont(js) == jm) obym(jk, jw) = obym(jk, jw) + oc(jw, js, jl, jp)
                end do
             end do
          end do
          odxt(jw) = sum(odx(jw, 1:nl, jp))
       end do
       ! O_BY_MOL_LAYER(1:NWN, 1:kount, 1:nlay) = (O_BY_MOL + OC)(:, 1:kount, :): molecule SLOTS 1..kount (:704, :770)
       obl = 0
       do jl = 1, nl
          do jk = 1, kount
             if (jk <= nm) obl(:, jk, jl) = obm(:, jk, jl, jp)
             do js = 1, NCONT
                if (index_cont(js) == jk) obl(:, jk, jl) = obl(:, jk, jl) + oc(:, js, jl, jp)
             end do
          end do
       end do
       write (ncname, '(a,i5.5,a)') 'MONORTM.', jp, '.nc'
       call write_monortm_nc(trim(ncname), nwn, kount, nl, wnunits, fr, tb(:, jp), rad(:, jp), trtot(:, jp), &
            sum(ps%wkl(1, 1:nl, jp))*2.99150e-23_dp, sum(ps%clw(1:nl, jp)), tmpsfc(jp), emiss(:, jp), reflc(:, jp), ps%angle(jp), &
            tmr(:, jp), ot, obym, odxt, cmol(1:kount), o(:, 1:nl, jp), obl)
       deallocate (fr, ot, obym, odxt, obl)
    end do
    write (*, '(a,i0,a)') ' monortm_hip: ', np, ' netCDF file(s) MONORTM.NNNNN.nc written'
  end subroutine write_netcdf

  subroutine write_layer_od()                            ! IOD = 1: ODmono_prfNNNN_layNNNN (reference :677-694)
    character(len=22) :: fileod
    integer :: v, jp, jl, jw
    real(dp) :: f
    do jp = 1, np
       do jl = 1, ps%nlay(jp)
          write (fileod, '(a10,i4.4,a4,i4.4)') 'ODmono_prf', jp, '_lay', jl
          open (newunit=v, file=fileod, status='replace', action='write')
          write (v, '(a5,I8)') 'NWN :', nwn
          write (v, '(2a10)') wnunits, ' LAYER_OD'
          do jw = 1, nwn
             f = cfg%wn(jw)
             if (giga) f = cfg%wn(jw)*CLIGHT/1.E9_dp
             write (v, '(f10.3,e12.4)') f, o(jw, jl, jp)
          end do
          close (v)
       end do
    end do
  end subroutine write_layer_od

end program monortm_hip
