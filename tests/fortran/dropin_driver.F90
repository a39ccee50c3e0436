!  dropin_driver -- proves the drop-in claim at the reference's own boundary: it
!  USEs the reference's unmodified binding module xgb_fortran_api (compiled in
!  place from /root/reference/Shared/xgb_fortran_api.F90 into oracle/_ref/, never
!  copied into this repository) and makes the calls predict_OH_with_XGB makes
!  (reference OH_GridComp/OH_GridCompMod.F90:251,256,261,264,347,356,377), linked
!  against libohxgb.so (or, for the CPU plumbing check, the oracle library).
!
!  usage: dropin_driver <rows.bin> <model file> <pred.bin>
!    rows.bin: int64 nrow, int64 ncol, real32 rows(ncol,nrow)
!    pred.bin: int64 len, real32 pred(len)
program dropin_driver
   use iso_c_binding
   use xgb_fortran_api
   implicit none
   character(len=1024) :: rows_file, model_file, pred_file
   integer(c_int64_t) :: nrow, ncol, plen, zero_len
   real(c_float), allocatable :: rows(:,:), small(:,:)
   real(c_float), pointer :: pred(:)
   real(c_float), parameter :: miss = -999.0
   type(c_ptr) :: dm, bst, cpred
   integer(c_int) :: rc
   integer :: u

   call get_command_argument(1, rows_file)
   call get_command_argument(2, model_file)
   call get_command_argument(3, pred_file)
   open(newunit=u, file=trim(rows_file), access='stream', form='unformatted', status='old', action='read')
   read(u) nrow, ncol
   allocate(rows(ncol, nrow))
   read(u) rows
   close(u)

   allocate(small(ncol, 1))
   small = 0.0
   rc = XGDMatrixCreateFromMat_f(small, 1_c_int64_t, ncol, miss, dm);  call must(rc, 'XGDMatrixCreateFromMat_f (dummy)')
   zero_len = 0
   rc = XGBoosterCreate_f(dm, zero_len, bst);                           call must(rc, 'XGBoosterCreate_f')
   rc = XGBoosterLoadModel_f(bst, model_file);                          call must(rc, 'XGBoosterLoadModel_f')
   rc = XGDMatrixFree_f(dm);                                            call must(rc, 'XGDMatrixFree_f (dummy)')

   rc = XGDMatrixCreateFromMat_f(rows, nrow, ncol, miss, dm);           call must(rc, 'XGDMatrixCreateFromMat_f')
   rc = XGBoosterPredict_f(bst, dm, 0_c_int, 0_c_int, 0_c_int, plen, cpred); call must(rc, 'XGBoosterPredict_f')
   if (plen /= nrow) then
      print *, 'Wrong value returned for xx_pred_len', plen, nrow
      stop 1
   end if
   call c_f_pointer(cpred, pred, [plen])
   open(newunit=u, file=trim(pred_file), access='stream', form='unformatted', status='replace', action='write')
   write(u) plen
   write(u) pred
   close(u)
   rc = XGDMatrixFree_f(dm);                                            call must(rc, 'XGDMatrixFree_f')
   rc = XGBoosterFree_f(bst);                                           call must(rc, 'XGBoosterFree_f')
contains
   subroutine must(code, what)
      integer(c_int), intent(in) :: code
      character(len=*), intent(in) :: what
      if (code /= 0) then
         print *, 'Failed in ', what
         stop 1
      end if
   end subroutine
end program dropin_driver
