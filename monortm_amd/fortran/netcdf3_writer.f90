! netCDF-3 "classic" (CDF-1) writer for the per-profile output file MONORTM.NNNNN.nc of the reference's STOREOUT
! (src/monortm_sub.F90:698-778, compiled there only with -DUSENETCDF against libnetcdff).  libnetcdff is not a dependency here:
! the classic format is a fixed header (dimension list, variable list with types / sizes / offsets) followed by the
! variables' values, all big-endian, so it is written directly with stream I/O.
! Same dimensions (FREQUENCY, MOLECULE, LAYERS, STRING_LENGTH), the same seventeen variables in the same order with the same
! types - the dbl build's REAL is NF90_DOUBLE, LAYER_OPTICAL_DEPTH_BY_MOLECULE is NF90_FLOAT in every build - and the "units"
! attribute of FREQUENCY.  A netCDF library (scipy.io.netcdf_file in tests/test_netcdf_output.py) reads it back.
module netcdf3_writer
  implicit none
  private
  public :: write_monortm_nc
  integer, parameter :: dp = selected_real_kind(15, 307), sp = selected_real_kind(6, 37), i4 = selected_int_kind(9)  ! (whatever -fdefault-real-8 does)
  integer(i4), parameter :: NC_CHAR = 2, NC_FLOAT = 5, NC_DOUBLE = 6, NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12

contains

  integer function pad4(n)
    integer, intent(in) :: n
    pad4 = 4*((n + 3)/4)
  end function pad4

  subroutine put_name(u, s)
    integer, intent(in) :: u
    character(len=*), intent(in) :: s
    integer :: k
    write (u) int(len(s), i4), s
    do k = len(s) + 1, pad4(len(s))
       write (u) achar(0)
    end do
  end subroutine put_name

  ! fname: MONORTM.NNNNN.nc.  Arrays in the driver's layout: o(nwn, nlay), obm_layer(nwn, kount, nlay) = the first `kount`
  ! molecule slots of O_BY_MOL + OC (the reference writes O_BY_MOL_LAYER(1:NWN, 1:kount, 1:nlay), :770 - slots 1..kount, not
  ! the molecules of the MOLECULE variable: reproduced), otot_by_mol(kount, nwn).
  subroutine write_monortm_nc(fname, nwn, kount, nlay, wnunits, freq, tb, rad, trtot, pwv, clw, sfct, emis, refl, angle, tmr, &
                              otot, otot_by_mol, odxtot, cmol, o, obm_layer)
    character(len=*), intent(in) :: fname, wnunits
    integer, intent(in) :: nwn, kount, nlay
    real(dp), intent(in) :: freq(nwn), tb(nwn), rad(nwn), trtot(nwn), pwv, clw, sfct, emis(nwn), refl(nwn), angle, tmr(nwn)
    real(dp), intent(in) :: otot(nwn), otot_by_mol(kount, nwn), odxtot(nwn), o(nwn, nlay), obm_layer(nwn, kount, nlay)
    character(len=8), intent(in) :: cmol(kount)
    integer, parameter :: NV = 17
    character(len=31) :: vname(NV)
    integer :: vnd(NV), vdim(3, NV), u, k, iw, ik, j
    integer(i4) :: vtype(NV), vsize(NV), begin(NV), hdr
    integer :: dlen(4)
    character(len=11) :: units11
    ! dimension ids (0-based, file order): 0 FREQUENCY, 1 MOLECULE, 2 LAYERS, 3 STRING_LENGTH.  The Fortran API lists the
    ! fastest dimension first; the file lists the slowest first.
    dlen = (/nwn, kount, nlay, 8/)
    vname = (/ character(len=31) :: 'FREQUENCY', 'BT', 'RAD', 'TRANS', 'PWV', 'CLW', 'SFCT', 'EMIS', 'REFL', 'ANGLE', 'TMR', 'TOTAL_OD', &
              'TOTAL_OD_BY_MOLECULE', 'XSEC_OD', 'MOLECULE', 'LAYER_OPTICAL_DEPTH', 'LAYER_OPTICAL_DEPTH_BY_MOLECULE' /)
    vnd = 1
    vdim = 0
    vtype = NC_DOUBLE
    vnd(13) = 2; vdim(1:2, 13) = (/0, 1/)          ! (FREQUENCY, MOLECULE): Fortran dimids (mol, wn)
    vnd(15) = 2; vdim(1:2, 15) = (/1, 3/); vtype(15) = NC_CHAR
    vnd(16) = 2; vdim(1:2, 16) = (/2, 0/)          ! (LAYERS, FREQUENCY)
    vnd(17) = 3; vdim(1:3, 17) = (/2, 1, 0/); vtype(17) = NC_FLOAT
    do k = 1, NV
       vsize(k) = 1
       do j = 1, vnd(k)
          vsize(k) = vsize(k)*dlen(vdim(j, k) + 1)
       end do
       if (vtype(k) == NC_DOUBLE) vsize(k) = vsize(k)*8
       if (vtype(k) == NC_FLOAT) vsize(k) = vsize(k)*4
       vsize(k) = pad4(int(vsize(k)))
    end do
    ! header size: magic + numrecs, dimension list, (absent) global attributes, variable list
    hdr = 8 + 8
    do k = 1, 4
       hdr = hdr + 4 + pad4(len_trim(dimname(k))) + 4
    end do
    hdr = hdr + 8 + 8
    do k = 1, NV
       hdr = hdr + 4 + pad4(len_trim(vname(k))) + 4 + 4*vnd(k)
       if (k == 1) then
          hdr = hdr + 8 + (4 + pad4(5) + 4 + 4 + pad4(11))      ! one attribute: "units", NC_CHAR x 11
       else
          hdr = hdr + 8
       end if
       hdr = hdr + 4 + 4 + 4
    end do
    begin(1) = hdr
    do k = 2, NV
       begin(k) = begin(k - 1) + vsize(k - 1)
    end do

    open (newunit=u, file=fname, access='stream', form='unformatted', status='replace', action='write', convert='BIG_ENDIAN')
    write (u) 'CDF', achar(1), 0_i4
    write (u) NC_DIMENSION, 4_i4
    do k = 1, 4
       call put_name(u, trim(dimname(k)))
       write (u) int(dlen(k), i4)
    end do
    write (u) 0_i4, 0_i4                       ! no global attributes
    write (u) NC_VARIABLE, int(NV, i4)
    units11 = wnunits
    do k = 1, NV
       call put_name(u, trim(vname(k)))
       write (u) int(vnd(k), i4)
       do j = 1, vnd(k)
          write (u) int(vdim(j, k), i4)
       end do
       if (k == 1) then
          write (u) NC_ATTRIBUTE, 1_i4
          call put_name(u, 'units')
          write (u) NC_CHAR, 11_i4, units11, achar(0)
       else
          write (u) 0_i4, 0_i4
       end if
       write (u) vtype(k), vsize(k), begin(k)
    end do
    write (u) freq, tb, rad, trtot
    write (u) (pwv, iw=1, nwn), (clw, iw=1, nwn), (sfct, iw=1, nwn)
    write (u) emis, refl
    write (u) (angle, iw=1, nwn)
    write (u) tmr, otot
    write (u) ((otot_by_mol(ik, iw), ik=1, kount), iw=1, nwn)
    write (u) odxtot
    write (u) (cmol(ik), ik=1, kount)
    do k = 8*kount + 1, pad4(8*kount)
       write (u) achar(0)
    end do
    write (u) ((o(iw, j), iw=1, nwn), j=1, nlay)
    write (u) (((real(obm_layer(iw, ik, j), sp), iw=1, nwn), ik=1, kount), j=1, nlay)
    close (u)
  contains
    function dimname(k) result(s)
      integer, intent(in) :: k
      character(len=13) :: s
      select case (k)
      case (1); s = 'FREQUENCY'
      case (2); s = 'MOLECULE'
      case (3); s = 'LAYERS'
      case default; s = 'STRING_LENGTH'
      end select
    end function dimname
  end subroutine write_monortm_nc

end module netcdf3_writer
