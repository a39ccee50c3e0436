!  oh_mock_driver -- BASELINE.json config #1: a synthetic MAPL-like state pushed
!  through predict_OH_with_XGB the way OH_GridCompMod's Run1 does
!  (reference OH_GridComp/OH_GridCompMod.F90:1557-1574), without MAPL/ESMF.
!
!  usage: oh_mock_driver <state.bin> <model file> <out.bin> [compat|fused] [ncalls]
!
!  state.bin (stream, little endian, written by tests/ and bench.py):
!     int32 im, jm, km, dynamic_k_range(0/1); real32 tropp_min, ohscale
!     real32 pl(im,jm,km)  tropp(im,jm)
!     27 fields in the order of OH_GridCompMod.F90:313-339, (im,jm) for LAT,
!     GMISTRATO3, ALBUV, SZA and (im,jm,km) for the others; PL in Pa
!  out.bin: int32 rc, k1, k2; real32 OH_ML(im,jm,km); real64 seconds per call (mean); int32 resident boosters
!  out.bin.times (text): seconds of every call, one per line - the first holds the one-time model load (:242-271)
program oh_mock_driver
   use, intrinsic :: iso_c_binding
   use oh_xgb_predict
   implicit none

   character(len=1024) :: state_file, model_file, out_file, mode, arg
   integer(c_int32_t) :: im, jm, km, dyn
   real(c_float) :: tropp_min, ohscale
   real, allocatable, target :: pl(:,:,:), tropp(:,:), f2(:,:,:), f3(:,:,:,:)
   real, allocatable, target :: OH_ML(:,:,:)
   type(OH_BOOST_INPUT_DATA) :: bb
   integer :: rc, k1, k2, u, n2, n3, ncalls, it, rc2
   integer(8) :: t0, t1, ta, tb, rate
   real(8) :: seconds
   real(8), allocatable :: per_call(:)
   logical, parameter :: two_d(27) = [ .true., .false., .false., .false., .false., .false., .false., .false., .false., &
                                       .false., .false., .false., .false., .false., .false., .false., .false., .false., &
                                       .false., .false., .false., .true., .true., .false., .false., .false., .true. ]
   integer :: f, slot2(27), slot3(27)

   if (command_argument_count() < 3) then
      print *, 'usage: oh_mock_driver <state.bin> <model or %m2 pattern> <out.bin> [compat|fused] [ncalls] [reference|by_name]'
      stop 2
   end if
   call get_command_argument(1, state_file)
   call get_command_argument(2, model_file)
   call get_command_argument(3, out_file)
   mode = 'compat'
   if (command_argument_count() >= 4) call get_command_argument(4, mode)
   ncalls = 1
   if (command_argument_count() >= 5) then
      call get_command_argument(5, arg)
      read(arg, *) ncalls
   end if
   !  call number `it` is dated 2024-<it>-01: a month-templated model name (OH_instance_OH.rc:20)
   !  rolls over between calls, and the policy says what that does
   if (command_argument_count() >= 6) then
      call get_command_argument(6, arg)
      if (trim(arg) == 'by_name') call oh_xgb_set_model_policy(OH_XGB_POLICY_BY_NAME)
   end if

   open(newunit=u, file=trim(state_file), access='stream', form='unformatted', status='old', action='read')
   read(u) im, jm, km, dyn, tropp_min, ohscale
   allocate(pl(im,jm,km), tropp(im,jm), OH_ML(im,jm,km))
   read(u) pl
   read(u) tropp
   n2 = count(two_d)
   n3 = 27 - n2
   allocate(f2(im,jm,n2), f3(im,jm,km,n3))
   n2 = 0
   n3 = 0
   do f = 1, 27
      if (two_d(f)) then
         n2 = n2 + 1
         slot2(f) = n2
         read(u) f2(:,:,n2)
      else
         n3 = n3 + 1
         slot3(f) = n3
         read(u) f3(:,:,:,n3)
      end if
   end do
   close(u)

   bb%LAT        => f2(:,:,slot2(1))
   bb%PL         => f3(:,:,:,slot3(2))
   bb%T          => f3(:,:,:,slot3(3))
   bb%NO2        => f3(:,:,:,slot3(4))
   bb%O3         => f3(:,:,:,slot3(5))
   bb%CH4        => f3(:,:,:,slot3(6))
   bb%CO         => f3(:,:,:,slot3(7))
   bb%ISOP       => f3(:,:,:,slot3(8))
   bb%ACET       => f3(:,:,:,slot3(9))
   bb%C2H6       => f3(:,:,:,slot3(10))
   bb%C3H8       => f3(:,:,:,slot3(11))
   bb%PRPE       => f3(:,:,:,slot3(12))
   bb%ALK4       => f3(:,:,:,slot3(13))
   bb%MP         => f3(:,:,:,slot3(14))
   bb%H2O2       => f3(:,:,:,slot3(15))
   bb%TAUCLWDN   => f3(:,:,:,slot3(16))
   bb%TAUCLIDN   => f3(:,:,:,slot3(17))
   bb%TAUCLIUP   => f3(:,:,:,slot3(18))
   bb%TAUCLWUP   => f3(:,:,:,slot3(19))
   bb%CLOUD      => f3(:,:,:,slot3(20))
   bb%QV         => f3(:,:,:,slot3(21))
   bb%GMISTRATO3 => f2(:,:,slot2(22))
   bb%ALBUV      => f2(:,:,slot2(23))
   bb%AODUP      => f3(:,:,:,slot3(24))
   bb%AODDN      => f3(:,:,:,slot3(25))
   bb%CH2O       => f3(:,:,:,slot3(26))
   bb%SZA        => f2(:,:,slot2(27))

   allocate(per_call(max(ncalls, 1)))
   per_call = 0.0d0
   call system_clock(t0, rate)
   do it = 1, ncalls
      call system_clock(ta)
      OH_ML(:,:,:) = 0.0                       ! OH_GridCompMod.F90:1559
      if (trim(mode) == 'fused') then
         call predict_OH_with_XGB_fused(oh_xgb_fill_template(model_file, 20240001 + 100*it, 0), im, jm, km, &
                                        dyn /= 0, tropp_min, pl, tropp, bb, ohscale, OH_ML, rc)
      else
         call predict_OH_with_XGB(oh_xgb_fill_template(model_file, 20240001 + 100*it, 0), im, jm, km, &
                                  dyn /= 0, tropp_min, pl, tropp, bb, OH_ML, rc)
         if (rc == OH_XGB_SUCCESS) OH_ML(:,:,:) = OH_ML(:,:,:) * ohscale      ! :1569
      end if
      call system_clock(tb)
      per_call(it) = real(tb - ta, 8) / real(rate, 8)
      if (rc /= OH_XGB_SUCCESS) exit
   end do
   call system_clock(t1)
   seconds = real(t1 - t0, 8) / real(rate, 8) / real(max(ncalls, 1), 8)

   k1 = 0
   k2 = 0
   if (rc == OH_XGB_SUCCESS) then
      call oh_xgb_k_slab(im, jm, km, dyn /= 0, tropp_min, pl, tropp, k1, k2, rc2)
   else
      print '(a)', 'oh_mock_driver: '//oh_xgb_error_text()
   end if

   open(newunit=u, file=trim(out_file), access='stream', form='unformatted', status='replace', action='write')
   write(u) int(rc, c_int32_t), int(k1, c_int32_t), int(k2, c_int32_t)
   write(u) OH_ML
   write(u) seconds
   write(u) int(oh_xgb_resident_models(), c_int32_t)
   close(u)
   open(newunit=u, file=trim(out_file)//'.times', status='replace', action='write')
   do it = 1, ncalls
      write(u, '(es16.8)') per_call(it)
   end do
   close(u)
   if (rc /= OH_XGB_SUCCESS) stop 1
end program oh_mock_driver
