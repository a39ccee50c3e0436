!  oh_run1_driver -- reads an OH Run1 state (written by tests/helpers.py), calls oh_run1_boost the way
!  OH_GridCompMod's Run1 reaches CALL_BOOST, writes OH / OH_boost / NDWET.  No MAPL, no ESMF.
program oh_run1_driver
   use, intrinsic :: iso_c_binding
   use oh_xgb_predict, only: OH_XGB_SUCCESS, oh_xgb_error_text
   use oh_run1
   implicit none
   character(len=1024) :: state_file, model_file, out_file
   integer :: u, rc, k1, k2
   integer(c_int32_t) :: im, jm, km, dyn
   real(c_float) :: tropp_min, ohscale, avogad, runiv, epsilon
   type(OH_RUN1_STATE) :: st
   real, allocatable, target :: OH(:,:,:), OH_boost(:,:,:), NDWET(:,:,:)

   if (command_argument_count() < 3) then
      print *, 'usage: oh_run1_driver <state.bin> <model> <out.bin>'
      stop 2
   end if
   call get_command_argument(1, state_file)
   call get_command_argument(2, model_file)
   call get_command_argument(3, out_file)

   open(newunit=u, file=trim(state_file), access='stream', form='unformatted', status='old', action='read')
   read(u) im, jm, km, dyn, tropp_min, ohscale, avogad, runiv, epsilon
   call edge(st%PLE_MOD); call vol(st%T_MOD); call vol(st%Q_MOD); call plane(st%TROPP_MOD)
   call edge(st%PLE_BST); call edge(st%ZLE_BST); call vol(st%TAUCLW); call vol(st%TAUCLI)
   call vol(st%BCscacoef); call vol(st%OCscacoef); call vol(st%BRscacoef); call vol(st%DUscacoef)
   call vol(st%SUscacoef); call vol(st%SSscacoef); call vol(st%NIscacoef)
   call plane(st%GMITO3); call plane(st%GMITTO3); call plane(st%latarr)
   call vol(st%T_BST); call vol(st%NO2); call vol(st%O3); call vol(st%CH4); call vol(st%CO); call vol(st%ISOP)
   call vol(st%ACET); call vol(st%C2H6); call vol(st%C3H8); call vol(st%PRPE); call vol(st%ALK4); call vol(st%MP)
   call vol(st%H2O2); call vol(st%CLOUD); call vol(st%QV); call plane(st%ALBUV); call vol(st%CH2O)
   call plane(st%sza_noon); call vol(st%default_OH)
   close(u)

   allocate(OH(im,jm,km), OH_boost(im,jm,km), NDWET(im,jm,km))
   call oh_run1_boost(trim(model_file), int(im), int(jm), int(km), dyn /= 0, tropp_min, ohscale, avogad, runiv, &
                      epsilon, st, OH, OH_boost, NDWET, k1, k2, rc)
   if (rc /= OH_XGB_SUCCESS) print '(a)', 'oh_run1_driver: '//oh_run1_error_text()//' '//oh_xgb_error_text()

   open(newunit=u, file=trim(out_file), access='stream', form='unformatted', status='replace', action='write')
   write(u) int(rc, c_int32_t), int(k1, c_int32_t), int(k2, c_int32_t)
   write(u) OH
   write(u) OH_boost
   write(u) NDWET
   close(u)
   if (rc /= OH_XGB_SUCCESS) stop 1

contains

   subroutine vol(p)
      real, pointer, intent(out) :: p(:,:,:)
      allocate(p(im,jm,km))
      read(u) p
   end subroutine

   subroutine edge(p)
      real, pointer, intent(out) :: p(:,:,:)
      allocate(p(im,jm,0:km))
      read(u) p
   end subroutine

   subroutine plane(p)
      real, pointer, intent(out) :: p(:,:)
      allocate(p(im,jm))
      read(u) p
   end subroutine
end program oh_run1_driver
