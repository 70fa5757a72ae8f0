! TEST INFRASTRUCTURE: writes one MONORTM.NNNNN.nc with monortm_amd/fortran/netcdf3_writer.f90 from closed-form arrays
! (tests/test_netcdf_output.py reads it back with a netCDF library and checks every value).
program nc_writer_check
  use netcdf3_writer, only: write_monortm_nc
  implicit none
  integer, parameter :: dp = selected_real_kind(15, 307)
  integer, parameter :: nwn = 5, kount = 3, nlay = 4
  real(dp) :: freq(nwn), tb(nwn), rad(nwn), tr(nwn), em(nwn), rf(nwn), tmr(nwn), ot(nwn), obym(kount, nwn), odx(nwn)
  real(dp) :: o(nwn, nlay), obl(nwn, kount, nlay)
  character(len=8) :: cmol(kount)
  integer :: i, k, j
  do i = 1, nwn
     freq(i) = 22.0_dp + i; tb(i) = 250.0_dp + 0.5_dp*i; rad(i) = 1.0e-7_dp*i; tr(i) = 0.9_dp - 0.01_dp*i
     em(i) = 0.6_dp; rf(i) = 0.4_dp; tmr(i) = 270.0_dp + i; ot(i) = 0.1_dp*i; odx(i) = 1.0e-3_dp*i
     do k = 1, kount
        obym(k, i) = k + 0.01_dp*i
        do j = 1, nlay
           obl(i, k, j) = i + 10.0_dp*k + 100.0_dp*j
        end do
     end do
     do j = 1, nlay
        o(i, j) = i + 0.25_dp*j
     end do
  end do
  cmol = (/'  H2O   ', '  CO2   ', '   O2   '/)
  call write_monortm_nc('MONORTM.00001.nc', nwn, kount, nlay, 'FREQ(GHz)', freq, tb, rad, tr, 1.25_dp, 0.03_dp, 2.75_dp, em, rf, &
                        180.0_dp, tmr, ot, obym, odx, cmol, o, obl)
end program nc_writer_check
