!  oh_xgb_predict -- Fortran host side of the OH predictor.
!
!  Mirrors the reference's predict_OH_with_XGB (OH_GridComp/OH_GridCompMod.F90:123-398)
!  and its input type OH_BOOST_INPUT_DATA (:82-114): same names, same argument
!  meaning, same checks, so that OH_GridCompMod's CALL_BOOST block (:1557-1574)
!  can call it unchanged.  MAPL is not needed here: where the reference uses
!  _ASSERT / RETURN_ this module sets rc and an error text instead
!  (rc = 0 success, as ESMF_SUCCESS).
!
!  Two routes to the same OH_ML:
!    predict_OH_with_XGB        the reference's own five-call sequence over the
!                               XGBoost C symbols (served by the HIP kernels)
!    predict_OH_with_XGB_fused  ONE call: gather, PL/100, tree walk and 10**pred
!                               run in a single gfx950 kernel
module oh_xgb_predict
   use, intrinsic :: iso_c_binding
   use ohx_bindings
   implicit none
   private

   public :: OH_BOOST_INPUT_DATA, predict_OH_with_XGB, predict_OH_with_XGB_fused
   public :: oh_xgb_k_slab, oh_xgb_reset, oh_xgb_error_text
   public :: oh_xgb_set_model_policy, oh_xgb_fill_template, oh_xgb_resident_models, oh_xgb_booster

   !  Which booster a call uses when the file name changes between calls.  The reference keeps the
   !  booster of the FIRST call for the life of the process and ignores later names, although the
   !  name is month-templated (first_time, :209,269; OH_instance_OH.rc:20).
   integer, parameter, public :: OH_XGB_POLICY_REFERENCE = 0   ! as the reference: load once
   integer, parameter, public :: OH_XGB_POLICY_BY_NAME   = 1   ! one resident booster per file name

   integer, parameter, public :: OH_XGB_SUCCESS = 0
   integer, parameter, public :: OH_XGB_FAILURE = 1

   !  Field names and order are the reference's (OH_GridCompMod.F90:82-114)
   type OH_BOOST_INPUT_DATA
      real, pointer, dimension(:,:  ) :: LAT        => null()
      real, pointer, dimension(:,:,:) :: PL         => null()
      real, pointer, dimension(:,:,:) :: T          => null()
      real, pointer, dimension(:,:,:) :: NO2        => null()
      real, pointer, dimension(:,:,:) :: O3         => null()
      real, pointer, dimension(:,:,:) :: CH4        => null()
      real, pointer, dimension(:,:,:) :: CO         => null()
      real, pointer, dimension(:,:,:) :: ISOP       => null()
      real, pointer, dimension(:,:,:) :: ACET       => null()
      real, pointer, dimension(:,:,:) :: C2H6       => null()
      real, pointer, dimension(:,:,:) :: C3H8       => null()
      real, pointer, dimension(:,:,:) :: PRPE       => null()
      real, pointer, dimension(:,:,:) :: ALK4       => null()
      real, pointer, dimension(:,:,:) :: MP         => null()
      real, pointer, dimension(:,:,:) :: H2O2       => null()
      real, pointer, dimension(:,:,:) :: TAUCLWDN   => null()
      real, pointer, dimension(:,:,:) :: TAUCLIDN   => null()
      real, pointer, dimension(:,:,:) :: TAUCLIUP   => null()
      real, pointer, dimension(:,:,:) :: TAUCLWUP   => null()
      real, pointer, dimension(:,:,:) :: CLOUD      => null()
      real, pointer, dimension(:,:,:) :: QV         => null()
      real, pointer, dimension(:,:  ) :: GMISTRATO3 => null()
      real, pointer, dimension(:,:  ) :: ALBUV      => null()
      real, pointer, dimension(:,:,:) :: AODUP      => null()
      real, pointer, dimension(:,:,:) :: AODDN      => null()
      real, pointer, dimension(:,:,:) :: CH2O       => null()
      real, pointer, dimension(:,:  ) :: SZA        => null()
   end type OH_BOOST_INPUT_DATA

   integer(c_int64_t), parameter :: xx_param_count = 27   ! OH_GridCompMod.F90:228
   real(c_float), parameter      :: xx_miss = -999.0      ! :213

   !  One booster per process, created and loaded on the first call (:182,209,242-271)
   type(c_ptr), save :: xx_bst = c_null_ptr
   logical, save     :: first_time = .true.
   character(len=512), save :: last_error = ''

   !  OH_XGB_POLICY_BY_NAME: the twelve monthly boosters stay resident in HBM (46 MB each for the
   !  100-tree model; 288 GB per GPU), a month roll-over is a table look-up
   integer, parameter :: max_resident = 16
   integer, save :: model_policy = OH_XGB_POLICY_REFERENCE
   integer, save :: n_resident = 0
   character(len=1024), save :: resident_name(max_resident) = ''
   type(c_ptr), save :: resident_bst(max_resident) = c_null_ptr

contains

   function oh_xgb_error_text() result(msg)
      character(len=:), allocatable :: msg
      msg = trim(last_error)
   end function

   !  Forget the process-wide booster(s) (the reference never frees it, :389-392)
   subroutine oh_xgb_reset()
      integer(c_int) :: rc
      integer :: q
      logical :: listed
      listed = .false.
      do q = 1, n_resident
         if (c_associated(resident_bst(q), xx_bst)) listed = .true.
         if (c_associated(resident_bst(q))) rc = XGBoosterFree(resident_bst(q))
         resident_bst(q) = c_null_ptr
         resident_name(q) = ''
      end do
      n_resident = 0
      if (c_associated(xx_bst) .and. .not. listed) rc = XGBoosterFree(xx_bst)
      xx_bst = c_null_ptr
      first_time = .true.
   end subroutine

   subroutine oh_xgb_set_model_policy(policy)
      integer, intent(in) :: policy
      model_policy = policy
   end subroutine

   integer function oh_xgb_resident_models()
      oh_xgb_resident_models = n_resident
   end function

   !  The GrADS-style tokens MAPL's fill_grads_template expands in XGBoostFile
   !  (OH_GridCompMod.F90:1187, OH_instance_OH.rc:20): %y4 %m2 %d2 %h2 %n2
   function oh_xgb_fill_template(pattern, nymd, nhms) result(name)
      character(len=*), intent(in) :: pattern
      integer, intent(in) :: nymd, nhms
      character(len=:), allocatable :: name
      character(len=4) :: buf
      integer :: p
      name = trim(pattern)
      do
         p = index(name, '%')
         if (p == 0 .or. p + 2 > len(name)) exit
         select case (name(p+1:p+2))
         case ('y4'); write(buf, '(i4.4)') nymd / 10000
            name = name(:p-1)//buf(1:4)//name(p+3:)
         case ('m2'); write(buf, '(i2.2)') mod(nymd, 10000) / 100
            name = name(:p-1)//buf(1:2)//name(p+3:)
         case ('d2'); write(buf, '(i2.2)') mod(nymd, 100)
            name = name(:p-1)//buf(1:2)//name(p+3:)
         case ('h2'); write(buf, '(i2.2)') nhms / 10000
            name = name(:p-1)//buf(1:2)//name(p+3:)
         case ('n2'); write(buf, '(i2.2)') mod(nhms, 10000) / 100
            name = name(:p-1)//buf(1:2)//name(p+3:)
         case default
            exit          ! an unknown token is left as it is
         end select
      end do
   end function

   !  Select (loading it on first sight) the booster for this call
   subroutine select_booster(xgb_fname, rc)
      character(len=*), intent(in) :: xgb_fname
      integer, intent(out) :: rc
      integer :: q
      rc = OH_XGB_SUCCESS
      if (model_policy == OH_XGB_POLICY_REFERENCE) then
         if (first_time) call one_time_setup(xgb_fname, rc)
         return
      end if
      do q = 1, n_resident
         if (trim(resident_name(q)) == trim(xgb_fname)) then
            xx_bst = resident_bst(q)
            return
         end if
      end do
      if (n_resident == max_resident) then
         last_error = 'more resident boosters than oh_xgb_predict keeps (16)'
         rc = OH_XGB_FAILURE
         return
      end if
      call one_time_setup(xgb_fname, rc)
      if (rc /= OH_XGB_SUCCESS) return
      n_resident = n_resident + 1
      resident_name(n_resident) = trim(xgb_fname)
      resident_bst(n_resident) = xx_bst
   end subroutine

   subroutine fail(what, rc)
      character(len=*), intent(in) :: what
      integer, intent(out) :: rc
      last_error = trim(what)//' :: '//ohx_last_error()
      rc = OH_XGB_FAILURE
   end subroutine

   !  ONE_TIME_SETUP (:242-271)
   subroutine one_time_setup(xgb_fname, rc)
      character(len=*), intent(in) :: xgb_fname
      integer, intent(out) :: rc
      real(c_float), allocatable :: xx_carr_small(:,:)
      type(c_ptr) :: xx_dmtrx
      integer(c_int64_t) :: xx_dmtrx_len
      integer(c_int) :: crc

      rc = OH_XGB_SUCCESS
      allocate(xx_carr_small(xx_param_count, 1))
      xx_carr_small(:,:) = 0.0
      crc = XGDMatrixCreateFromMat(xx_carr_small, 1_c_int64_t, xx_param_count, xx_miss, xx_dmtrx)
      if (crc /= 0) then; call fail('Failed in XGDMatrixCreateFromMat_f', rc); return; end if
      xx_dmtrx_len = 0          ! as in the reference: a handle by value, length 0 (:255-256)
      crc = XGBoosterCreate(xx_dmtrx, xx_dmtrx_len, xx_bst)
      if (crc /= 0) then; call fail('Failed in XGBoosterCreate_f', rc); return; end if
      crc = XGBoosterLoadModel(xx_bst, ohx_c_string(xgb_fname))
      if (crc /= 0) then; call fail('Failed in XGBoosterLoadModel_f', rc); return; end if
      crc = XGDMatrixFree(xx_dmtrx)
      if (crc /= 0) then; call fail('Failed in XGDMatrixFree_f', rc); return; end if
      deallocate(xx_carr_small)
      first_time = .false.
   end subroutine

   !  The booster a call on this file name uses, loaded if need be (for callers of the OHX* entry points)
   subroutine oh_xgb_booster(xgb_fname, bst, rc)
      character(len=*), intent(in) :: xgb_fname
      type(c_ptr), intent(out) :: bst
      integer, intent(out) :: rc
      rc = OH_XGB_SUCCESS
      if (first_time .or. model_policy /= OH_XGB_POLICY_REFERENCE) call select_booster(xgb_fname, rc)
      bst = xx_bst
   end subroutine

   !  The slab of levels that needs a prediction (:275-301)
   subroutine oh_xgb_k_slab(icount, jcount, kcount, dynamic_k_range, tropp_min, pl, tropp, k1, k2, rc)
      integer, intent(in)  :: icount, jcount, kcount
      logical, intent(in)  :: dynamic_k_range
      real, intent(in)     :: tropp_min
      real, intent(in)     :: pl(:,:,:)
      real, intent(in)     :: tropp(:,:)
      integer, intent(out) :: k1, k2, rc
      integer :: i, j, k, ksubcount

      rc = OH_XGB_SUCCESS
      ksubcount = 0
      if (dynamic_k_range) then
         do j = 1, jcount
            do i = 1, icount
               k = count(pl(i,j,:) > tropp(i,j))
               if (k > ksubcount) ksubcount = k
            end do
         end do
      else
         k = count(tropp <= tropp_min)
         if (k /= 0) then
            last_error = 'OH Prediction: Minimum tropopause pressure is not low enough!'
            rc = OH_XGB_FAILURE
            k1 = kcount + 1
            k2 = kcount
            return
         end if
         do j = 1, jcount
            do i = 1, icount
               k = count(pl(i,j,:) > tropp_min)
               if (k > ksubcount) ksubcount = k
            end do
         end do
      end if
      k1 = kcount - ksubcount + 1
      k2 = kcount
   end subroutine

   !  Same interface as the reference's subroutine (:123-124)
   subroutine predict_OH_with_XGB(xgb_fname, icount, jcount, kcount, dynamic_k_range, tropp_min, pl, tropp, bb, OH_ML, rc)
      character(len=*),          intent(in)    :: xgb_fname
      integer,                   intent(in)    :: icount, jcount, kcount
      logical,                   intent(in)    :: dynamic_k_range
      real,                      intent(in)    :: tropp_min      ! Pa
      real,                      intent(in)    :: pl(:,:,:)      ! Pa
      real,                      intent(in)    :: tropp(:,:)     ! Pa
      type(OH_BOOST_INPUT_DATA), intent(in)    :: bb
      real,                      intent(inout) :: OH_ML(:,:,:)   ! mol/mol, top-down
      integer,                   intent(out)   :: rc

      integer(c_int64_t) :: xx_prediction_count, xx_pred_len
      type(c_ptr) :: xx_dmtrx, xx_cpred
      real(c_float), allocatable :: xx_carr(:,:)
      real(c_float), pointer :: xx_pred(:)
      integer(c_int) :: xx_option_mask, xx_ntree_limit, xx_training, crc
      integer :: i, j, k, k1, k2, ksubcount
      integer(c_int64_t) :: m

      xx_option_mask = 0      ! :231
      xx_ntree_limit = 0      ! :232
      xx_training    = 0      ! :235

      if (first_time .or. model_policy /= OH_XGB_POLICY_REFERENCE) then
         call select_booster(xgb_fname, rc)
         if (rc /= OH_XGB_SUCCESS) return
      end if

      call oh_xgb_k_slab(icount, jcount, kcount, dynamic_k_range, tropp_min, pl, tropp, k1, k2, rc)
      if (rc /= OH_XGB_SUCCESS) return
      ksubcount = k2 - k1 + 1

      xx_prediction_count = int(icount, c_int64_t) * jcount * ksubcount
      allocate(xx_carr(xx_param_count, max(xx_prediction_count, 1_c_int64_t)))

      m = 1
      do k = k1, k2
      do j = 1, jcount
      do i = 1, icount
         xx_carr( 1,m) = bb%LAT(       i,j  )
         xx_carr( 2,m) = bb%PL(        i,j,k) / 100.0    ! Pa -> hPa
         xx_carr( 3,m) = bb%T(         i,j,k)
         xx_carr( 4,m) = bb%NO2(       i,j,k)
         xx_carr( 5,m) = bb%O3(        i,j,k)
         xx_carr( 6,m) = bb%CH4(       i,j,k)
         xx_carr( 7,m) = bb%CO(        i,j,k)
         xx_carr( 8,m) = bb%ISOP(      i,j,k)
         xx_carr( 9,m) = bb%ACET(      i,j,k)
         xx_carr(10,m) = bb%C2H6(      i,j,k)
         xx_carr(11,m) = bb%C3H8(      i,j,k)
         xx_carr(12,m) = bb%PRPE(      i,j,k)
         xx_carr(13,m) = bb%ALK4(      i,j,k)
         xx_carr(14,m) = bb%MP(        i,j,k)
         xx_carr(15,m) = bb%H2O2(      i,j,k)
         xx_carr(16,m) = bb%TAUCLWDN(  i,j,k)
         xx_carr(17,m) = bb%TAUCLIDN(  i,j,k)
         xx_carr(18,m) = bb%TAUCLIUP(  i,j,k)
         xx_carr(19,m) = bb%TAUCLWUP(  i,j,k)
         xx_carr(20,m) = bb%CLOUD(     i,j,k)
         xx_carr(21,m) = bb%QV(        i,j,k)
         xx_carr(22,m) = bb%GMISTRATO3(i,j  )
         xx_carr(23,m) = bb%ALBUV(     i,j  )
         xx_carr(24,m) = bb%AODUP(     i,j,k)
         xx_carr(25,m) = bb%AODDN(     i,j,k)
         xx_carr(26,m) = bb%CH2O(      i,j,k)
         xx_carr(27,m) = bb%SZA(       i,j  )
         m = m + 1
      end do
      end do
      end do

      crc = XGDMatrixCreateFromMat(xx_carr, xx_prediction_count, xx_param_count, xx_miss, xx_dmtrx)
      if (crc /= 0) then; call fail('Failed in XGDMatrixCreateFromMat_f', rc); return; end if
      ! not in the reference: tell the library which grid the rows were gathered from (speed only)
      crc = OHXDMatrixSetGrid(xx_dmtrx, int(icount, c_int), int(jcount, c_int), 0_c_int64_t)
      if (crc /= 0) then; call fail('Failed in OHXDMatrixSetGrid', rc); return; end if

      crc = XGBoosterPredict(xx_bst, xx_dmtrx, xx_option_mask, xx_ntree_limit, xx_training, xx_pred_len, xx_cpred)
      if (crc /= 0) then; call fail('Failed in XGBoosterPredict_f', rc); return; end if
      if (xx_pred_len /= xx_prediction_count) then
         last_error = 'Wrong value returned for xx_pred_len'
         rc = OH_XGB_FAILURE
         return
      end if

      call c_f_pointer(xx_cpred, xx_pred, [xx_pred_len])

      m = 1
      do k = k1, k2
      do j = 1, jcount
      do i = 1, icount
         OH_ML(i,j,k) = 10.0 ** (xx_pred(m))    ! mol/mol
         m = m + 1
      end do
      end do
      end do

      crc = XGDMatrixFree(xx_dmtrx)
      if (crc /= 0) then; call fail('Failed in XGDMatrixFree_f', rc); return; end if
      nullify(xx_pred)
      deallocate(xx_carr)
      rc = OH_XGB_SUCCESS
   end subroutine predict_OH_with_XGB

   !  C address of a field; the fused kernel reads n1*n2(*n3) floats in place, so the array must be
   !  contiguous (MAPL pointers are) and of exactly that shape.
   function addr3(a, n1, n2, n3, ok) result(p)
      real, pointer, intent(in) :: a(:,:,:)
      integer, intent(in) :: n1, n2, n3
      logical, intent(inout) :: ok
      type(c_ptr) :: p
      p = c_null_ptr
      if (.not. associated(a)) then
         ok = .false.
      else if (.not. is_contiguous(a) .or. size(a,1) /= n1 .or. size(a,2) /= n2 .or. size(a,3) /= n3) then
         ok = .false.
      else
         p = c_loc(a(lbound(a,1), lbound(a,2), lbound(a,3)))
      end if
   end function

   function addr2(a, n1, n2, ok) result(p)
      real, pointer, intent(in) :: a(:,:)
      integer, intent(in) :: n1, n2
      logical, intent(inout) :: ok
      type(c_ptr) :: p
      p = c_null_ptr
      if (.not. associated(a)) then
         ok = .false.
      else if (.not. is_contiguous(a) .or. size(a,1) /= n1 .or. size(a,2) /= n2) then
         ok = .false.
      else
         p = c_loc(a(lbound(a,1), lbound(a,2)))
      end if
   end function

   !  Same arguments plus OHscale: OH_ML(i,j,k1:k2) = 10**pred * ohscale in one
   !  kernel (the reference applies OHscale right after the call, :1569).
   subroutine predict_OH_with_XGB_fused(xgb_fname, icount, jcount, kcount, dynamic_k_range, tropp_min, pl, tropp, &
                                        bb, ohscale, OH_ML, rc)
      character(len=*),          intent(in)    :: xgb_fname
      integer,                   intent(in)    :: icount, jcount, kcount
      logical,                   intent(in)    :: dynamic_k_range
      real,                      intent(in)    :: tropp_min
      real,                      intent(in)    :: pl(:,:,:)
      real,                      intent(in)    :: tropp(:,:)
      type(OH_BOOST_INPUT_DATA), intent(in)    :: bb
      real,                      intent(in)    :: ohscale
      real, target, contiguous,  intent(inout) :: OH_ML(:,:,:)
      integer,                   intent(out)   :: rc

      type(c_ptr) :: fields(27)
      integer(c_int32_t) :: is2d(27)
      integer :: k1, k2
      integer(c_int) :: crc
      logical :: ok

      if (first_time .or. model_policy /= OH_XGB_POLICY_REFERENCE) then
         call select_booster(xgb_fname, rc)
         if (rc /= OH_XGB_SUCCESS) return
      end if
      call oh_xgb_k_slab(icount, jcount, kcount, dynamic_k_range, tropp_min, pl, tropp, k1, k2, rc)
      if (rc /= OH_XGB_SUCCESS) return

      ok = all(shape(OH_ML) == [icount, jcount, kcount])
      is2d(:) = 0
      fields( 1) = addr2(bb%LAT, icount, jcount, ok);        is2d( 1) = 1
      fields( 2) = addr3(bb%PL, icount, jcount, kcount, ok)
      fields( 3) = addr3(bb%T, icount, jcount, kcount, ok)
      fields( 4) = addr3(bb%NO2, icount, jcount, kcount, ok)
      fields( 5) = addr3(bb%O3, icount, jcount, kcount, ok)
      fields( 6) = addr3(bb%CH4, icount, jcount, kcount, ok)
      fields( 7) = addr3(bb%CO, icount, jcount, kcount, ok)
      fields( 8) = addr3(bb%ISOP, icount, jcount, kcount, ok)
      fields( 9) = addr3(bb%ACET, icount, jcount, kcount, ok)
      fields(10) = addr3(bb%C2H6, icount, jcount, kcount, ok)
      fields(11) = addr3(bb%C3H8, icount, jcount, kcount, ok)
      fields(12) = addr3(bb%PRPE, icount, jcount, kcount, ok)
      fields(13) = addr3(bb%ALK4, icount, jcount, kcount, ok)
      fields(14) = addr3(bb%MP, icount, jcount, kcount, ok)
      fields(15) = addr3(bb%H2O2, icount, jcount, kcount, ok)
      fields(16) = addr3(bb%TAUCLWDN, icount, jcount, kcount, ok)
      fields(17) = addr3(bb%TAUCLIDN, icount, jcount, kcount, ok)
      fields(18) = addr3(bb%TAUCLIUP, icount, jcount, kcount, ok)
      fields(19) = addr3(bb%TAUCLWUP, icount, jcount, kcount, ok)
      fields(20) = addr3(bb%CLOUD, icount, jcount, kcount, ok)
      fields(21) = addr3(bb%QV, icount, jcount, kcount, ok)
      fields(22) = addr2(bb%GMISTRATO3, icount, jcount, ok); is2d(22) = 1
      fields(23) = addr2(bb%ALBUV, icount, jcount, ok);      is2d(23) = 1
      fields(24) = addr3(bb%AODUP, icount, jcount, kcount, ok)
      fields(25) = addr3(bb%AODDN, icount, jcount, kcount, ok)
      fields(26) = addr3(bb%CH2O, icount, jcount, kcount, ok)
      fields(27) = addr2(bb%SZA, icount, jcount, ok);        is2d(27) = 1
      if (.not. ok) then
         last_error = 'predict_OH_with_XGB_fused: every bb field must be associated, contiguous and (icount,jcount[,kcount])'
         rc = OH_XGB_FAILURE
         return
      end if

      ! feature 2 (0-based 1) is PL: the kernel divides it by 100 (:314)
      crc = OHXBoosterPredictFields(xx_bst, fields, is2d, 27_c_int, 1_c_int, int(icount, c_int), int(jcount, c_int), &
                                    int(kcount, c_int), int(k1, c_int), int(k2, c_int), xx_miss, 1_c_int, &
                                    real(ohscale, c_float), c_loc(OH_ML(1,1,1)), c_null_ptr)
      if (crc /= 0) then; call fail('Failed in OHXBoosterPredictFields', rc); return; end if
      rc = OH_XGB_SUCCESS
   end subroutine predict_OH_with_XGB_fused

end module oh_xgb_predict
