! Example caller of the Fortran boundary (and the dump harness of the parity tests); no reference code.
!
! Dump harness for the MODM / CALCTMR / RTM boundary (SURVEY.md section 8(b)).
! It calls the three public entry points exactly the way the reference driver does
! (reference: src/monortm.f90:557-574) and writes every output array at full
! precision, because MONORTM.OUT only prints 5 significant digits for optical depths
! (reference: src/monortm_sub.F90:780-782).
!
! The same source is linked twice:
!   * against the reference's own compiled modules  -> oracle/_ref/harness_ref_{dbl,sgl}
!   * against monortm_amd/fortran shim modules      -> monortm_amd/lib/harness_hip_{dbl,sgl}
! which is the source-level drop-in test for the boundary.
!
! usage: harness <case.bin> <TAPE3> <out.bin> [repeat]
!
! case.bin (little-endian stream):
!   int32 magic(=1297241155 'CTRM'), int32 nprof
!   per profile:
!     int32 nwn, nlay, nmol, irt, iout, icp, ibrd, ixsect
!     real64 dvset, sclcpl, sclhw, y0res, tmpsfc, cntnm(7)
!     real64 wn(nwn), p(nlay), t(nlay), clw(nlay), wbrodl(nlay), tz(0:nlay)
!     real64 wkl(nmol,nlay), emiss(nwn), reflc(nwn)
!     [ixsect = 1:] int32 nxs, nxs x character*10 names, real64 xamnt(nxs,nlay)   (cross-section molecules; FSCDXS and the
!                   xs files are expected in the working directory; XSREAD is called for the FIRST such profile only - the
!                   reference's XSREAD adds to NSPECR on every call, src/monortm_sub.F90:1365)
! out.bin: per profile
!     int32 nwn, nlay, nmol
!     real64 o(nwn,nlay), o_by_mol(nwn,nmol,nlay), oc(nwn,5,nlay) [molecules 1,2,3,7,22], o_clw(nwn,nlay)
!     real64 rup(nwn), rdn(nwn), trtot(nwn), rad(nwn), tb(nwn), tmr(nwn), tmpsfc_out
!     [ixsect = 1:] int32 -7777, real64 odxsec(nwn,nlay)
program harness
  use ModmMod, only: MODM
  use RTMmono, only: RTM, calctmr, NWNMX
  use CntnmFactors, only: CntnmFactors_t
  use lblparams, only: MXLAY, MXMOL, MX_XS
  implicit none
  integer, parameter :: ipts = 5050
  ! reference: src/monortm.f90:267-268,275 (the driver pre-sets icflg = -999)
  integer :: icflg, iuf, nptabsc
  real(8) :: v1absc, v2absc, dvabsc
  real :: delT_pert, dqh2oC(ipts), dTh2oC(ipts), dUh2o
  common /CDERIV/ icflg, iuf, v1absc, v2absc, dvabsc, nptabsc, delT_pert, dqh2oC, dTh2oC, dUh2o

  ! cross-section molecules: the hidden inputs of MODM for IXSECT = 1 (src/monortm.f90:233, src/monortm_sub.F90:1268)
  integer :: IXMAX, IXMOLS, IXINDX(MX_XS)
  real :: XAMNT(MX_XS, MXLAY)
  common /PATHX/ IXMAX, IXMOLS, IXINDX, XAMNT
  integer(4) :: nxs4, mark4
  character(len=10) :: xsn(MX_XS)
  logical :: xs_read
  real(8) :: xv1, xv2
  character(len=512) :: fcase, ftape, fout, arg
  character(len=80)  :: hfile
  integer(4) :: magic, nprof4, hdr(8)
  integer :: nprof, ip, nwn, nlay, nmol, irt, iout, icp, ibrd, ixsect, idu, ipr, rep, nrep
  integer :: iu, ou, i, k, m
  real(8) :: sc(12)
  real(8), allocatable :: buf(:)
  real(8) :: wn(NWNMX)
  real :: dvset, sclcpl, sclhw, y0res, tmpsfc, tmpsfc_in
  real :: p(MXLAY), t(MXLAY), clw(MXLAY), wbrodl(MXLAY), tz(0:MXLAY), wkl(MXMOL, MXLAY)
  real, allocatable :: o(:,:), o_by_mol(:,:,:), oc(:,:,:), o_clw(:,:), odxsec(:,:)
  real, allocatable :: tmr(:), rad(:), emiss(:), reflc(:), rup(:), trtot(:), rdn(:), tb(:)
  type(CntnmFactors_t) :: fac
  integer(8) :: c0, c1, c2, c3, crate
  real(8) :: tsec, evals, tfirst, tstage(3)
  character(len=16) :: envv
  integer :: envl

  if (command_argument_count() < 3) then
     print *, 'usage: harness case.bin TAPE3 out.bin [repeat]'
     stop 2
  end if
  call get_command_argument(1, fcase)
  call get_command_argument(2, ftape)
  call get_command_argument(3, fout)
  nrep = 1
  if (command_argument_count() >= 4) then
     call get_command_argument(4, arg)
     read (arg, *) nrep
  end if
  hfile = ftape(1:80)

  icflg = -999
  iuf = 0
  v1absc = 0; v2absc = 0; dvabsc = 0; nptabsc = 0; delT_pert = 0

  ipr = 66
  open (ipr, file='HARNESS.LOG', status='unknown')
  iu = 21
  ou = 22
  open (iu, file=trim(fcase), access='stream', form='unformatted', status='old')
  open (ou, file=trim(fout), access='stream', form='unformatted', status='replace')
  read (iu) magic, nprof4
  if (magic /= 1297241155) stop 'harness: bad magic'
  nprof = nprof4
  idu = 1
  tsec = 0
  evals = 0
  tfirst = 0
  tstage = 0
  ! HARNESS_RTM_FIRST=1: CALCTMR and RTM are called once BEFORE the first MODM (legal for the reference: they need no
  ! line file); a drop-in must then still load the line file in its first MODM call (src/modm.f90:187-190)
  call get_environment_variable('HARNESS_RTM_FIRST', envv, envl)

  xs_read = .false.
  do ip = 1, nprof
     read (iu) hdr
     nwn = hdr(1); nlay = hdr(2); nmol = hdr(3); irt = hdr(4)
     iout = hdr(5); icp = hdr(6); ibrd = hdr(7); ixsect = hdr(8)
     read (iu) sc
     dvset = real(sc(1)); sclcpl = real(sc(2)); sclhw = real(sc(3)); y0res = real(sc(4))
     tmpsfc_in = real(sc(5))
     fac%xself = real(sc(6)); fac%xfrgn = real(sc(7)); fac%xco2c = real(sc(8))
     fac%xo3cn = real(sc(9)); fac%xo2cn = real(sc(10)); fac%xn2cn = real(sc(11))
     fac%xrayl = real(sc(12))
     if (nwn > NWNMX .or. nlay > MXLAY .or. nmol > MXMOL) stop 'harness: case too large'
     wn = 0
     read (iu) wn(1:nwn)
     allocate (buf(max(nwn, nlay + 1, nmol*nlay)))
     p = 0; t = 0; clw = 0; wbrodl = 0; tz = 0; wkl = 0
     read (iu) buf(1:nlay); p(1:nlay) = real(buf(1:nlay))
     read (iu) buf(1:nlay); t(1:nlay) = real(buf(1:nlay))
     read (iu) buf(1:nlay); clw(1:nlay) = real(buf(1:nlay))
     read (iu) buf(1:nlay); wbrodl(1:nlay) = real(buf(1:nlay))
     read (iu) buf(1:nlay + 1); tz(0:nlay) = real(buf(1:nlay + 1))
     read (iu) buf(1:nmol*nlay)
     do k = 1, nlay
        do m = 1, nmol
           wkl(m, k) = real(buf((k - 1)*nmol + m))
        end do
     end do
     ! same shapes the reference driver allocates (src/monortm.f90:352-355), trimmed in the
     ! layer dimension to keep the 10000-wavenumber case within memory
     allocate (o(nwn, nlay), o_clw(nwn, nlay))
     ! MONORTM_XSEC_SUB declares its output ODXSEC(NWNMX,MXLAY) (src/monortm_sub.F90:1611) and writes odxsec(iwn,il) with that
     ! leading dimension, while MODM hands it an assumed-shape array: a caller that allocates (nwn, .) - as the reference's own
     ! driver does, src/monortm.f90:352 - makes it write out of bounds for every layer but the first.  With the first extent
     ! NWNMX the indexing is the intended one.
     if (ixsect == 1) then
        allocate (odxsec(NWNMX, nlay))
     else
        allocate (odxsec(nwn, nlay))
     end if
     allocate (o_by_mol(nwn, MXMOL, nlay), oc(nwn, MXMOL, nlay))
     allocate (tmr(nwn), rad(nwn), emiss(nwn), reflc(nwn), rup(nwn), trtot(nwn), rdn(nwn), tb(nwn))
     read (iu) buf(1:nwn); emiss = real(buf(1:nwn))
     read (iu) buf(1:nwn); reflc = real(buf(1:nwn))
     if (ixsect == 1) then
        read (iu) nxs4
        IXMOLS = nxs4
        read (iu) xsn(1:IXMOLS)
        deallocate (buf)
        allocate (buf(IXMOLS*nlay))
        read (iu) buf(1:IXMOLS*nlay)
        XAMNT = 0
        do k = 1, nlay
           do m = 1, IXMOLS
              XAMNT(m, k) = real(buf((k - 1)*IXMOLS + m))
           end do
        end do
        if (.not. xs_read) then   ! the names go through a scratch file: XSREAD reads them from unit ipf (record 2.2.1)
           open (23, file='XSNAMES.TMP', status='replace', form='formatted')
           write (23, '(7A10)') xsn(1:min(7, IXMOLS))
           if (IXMOLS > 7) write (23, '(8A10)') xsn(8:IXMOLS)
           close (23)
           open (23, file='XSNAMES.TMP', status='old', form='formatted')
           xv1 = minval(wn(1:nwn)); xv2 = maxval(wn(1:nwn))
           call XSREAD(23, xv1, xv2)
           close (23)
           xs_read = .true.
        end if
     end if

     if (ip == 1 .and. envl > 0) then
        o = 1.0e-3
        tmpsfc = tmpsfc_in
        call calctmr(nlay, nwn, wn, t, tz, o, tmr)
        call RTM(iout, irt, nwn, wn, nlay, t, tz, o, &
                 tmpsfc, rup, trtot, rdn, reflc, emiss, rad, tb, idu)
     end if
     do rep = 1, nrep
        tmpsfc = tmpsfc_in
        call system_clock(c0, crate)
        call MODM(ipr, icp, nwn, wn, dvset, nlay, p, t, clw, &
                  o, o_by_mol, oc, o_clw, odxsec, &
                  nmol, wkl, wbrodl, &
                  sclcpl, sclhw, y0res, hfile, fac, ixsect, ibrd)
        call system_clock(c2)
        call calctmr(nlay, nwn, wn, t, tz, o, tmr)
        call system_clock(c3)
        call RTM(iout, irt, nwn, wn, nlay, t, tz, o, &
                 tmpsfc, rup, trtot, rdn, reflc, emiss, rad, tb, idu)
        call system_clock(c1)
        tsec = tsec + real(c1 - c0, 8)/real(crate, 8)
        if (ip == 1 .and. rep == 1) then   ! the first MODM call loads the line file (and, for the GPU path, starts the device)
           tfirst = real(c1 - c0, 8)/real(crate, 8)
        else
           tstage(1) = tstage(1) + real(c2 - c0, 8)/real(crate, 8)
           tstage(2) = tstage(2) + real(c3 - c2, 8)/real(crate, 8)
           tstage(3) = tstage(3) + real(c1 - c3, 8)/real(crate, 8)
        end if
     end do

     hdr(1) = nwn; hdr(2) = nlay; hdr(3) = nmol
     write (ou) hdr(1:3)
     write (ou) real(o(1:nwn, 1:nlay), 8)
     write (ou) real(o_by_mol(1:nwn, 1:nmol, 1:nlay), 8)
     ! the five continuum slots MODM fills (index_cont, reference src/modm.f90:166); slot 22 (N2)
     ! exists even when nmol < 22
     write (ou) real(oc(1:nwn, (/1, 2, 3, 7, 22/), 1:nlay), 8)
     write (ou) real(o_clw(1:nwn, 1:nlay), 8)
     write (ou) real(rup, 8), real(rdn, 8), real(trtot, 8), real(rad, 8), real(tb, 8), real(tmr, 8)
     write (ou) real(tmpsfc, 8)
     if (ixsect == 1) then
        mark4 = -7777
        write (ou) mark4
        write (ou) real(odxsec(1:nwn, 1:nlay), 8)
     end if
     deallocate (buf, o, o_clw, odxsec, o_by_mol, oc, tmr, rad, emiss, reflc, rup, trtot, rdn, tb)
  end do
  close (iu)
  close (ou)
  close (ipr)
  write (*, '(a,f12.6,a,i6,a,i4)') 'HARNESS_SECONDS ', tsec, ' nprof ', nprof, ' repeat ', nrep
  write (*, '(a,f12.6,a,3f12.6)') 'HARNESS_FIRST_CALL ', tfirst, ' STAGES_AFTER_FIRST(MODM,CALCTMR,RTM) ', tstage
end program harness
