!  ohx_bindings -- ISO_C_BINDING view of libohxgb.so (include/ohxgb.h).
!
!  This is the thin shim BASELINE.json's north star asks for: Fortran host code
!  reaches the HIP kernels through plain C symbols, no CUDA-compat layer.  The
!  first group are the XGBoost C-API symbols that QuickChem's own
!  Shared/xgb_fortran_api.F90 binds (that module links against libohxgb.so
!  unchanged -- see INTEGRATION.md); they are declared again here, under their C
!  names, so that this tree builds without the reference.  The second group is
!  the additive fused entry point.
module ohx_bindings
   use, intrinsic :: iso_c_binding
   implicit none
   private

   public :: XGDMatrixCreateFromMat, XGDMatrixFree, XGDMatrixNumRow, XGDMatrixNumCol
   public :: XGBoosterCreate, XGBoosterFree, XGBoosterLoadModel, XGBoosterSaveModel
   public :: XGBoosterPredict, XGBoosterSetParam, OHXBoosterPredictFields, OHXDMatrixSetGrid
   public :: OHXCommGetUniqueId, OHXCommInitRank, OHXCommFree, OHXCommInfo, OHXShardRows, OHXAllGatherOH, OHX_UNIQUE_ID_BYTES
   public :: ohx_last_error, ohx_c_string

   integer, parameter :: OHX_UNIQUE_ID_BYTES = 128

   interface
      ! ---- symbols bound by the reference (Shared/xgb_fortran_api.F90:19-119) ----
      function XGDMatrixCreateFromMat(data, nrow, ncol, missing, out) bind(C, name="XGDMatrixCreateFromMat") result(rc)
         import :: c_int, c_float, c_int64_t, c_ptr
         real(c_float), intent(in)      :: data(*)
         integer(c_int64_t), value      :: nrow, ncol
         real(c_float), value           :: missing
         type(c_ptr), intent(out)       :: out
         integer(c_int)                 :: rc
      end function

      function XGDMatrixFree(handle) bind(C, name="XGDMatrixFree") result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int)     :: rc
      end function

      function XGDMatrixNumRow(handle, out) bind(C, name="XGDMatrixNumRow") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value              :: handle
         integer(c_int64_t), intent(out) :: out
         integer(c_int)                  :: rc
      end function

      function XGDMatrixNumCol(handle, out) bind(C, name="XGDMatrixNumCol") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value              :: handle
         integer(c_int64_t), intent(out) :: out
         integer(c_int)                  :: rc
      end function

      ! dmats: the reference hands over ONE handle by value with len = 0
      ! (OH_GridCompMod.F90:255-256); the library never reads it then.
      function XGBoosterCreate(dmats, len, out) bind(C, name="XGBoosterCreate") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value        :: dmats
         integer(c_int64_t), value :: len
         type(c_ptr), intent(out)  :: out
         integer(c_int)            :: rc
      end function

      function XGBoosterFree(handle) bind(C, name="XGBoosterFree") result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int)     :: rc
      end function

      function XGBoosterLoadModel(handle, fname) bind(C, name="XGBoosterLoadModel") result(rc)
         import :: c_int, c_ptr, c_char
         type(c_ptr), value                 :: handle
         character(kind=c_char), intent(in) :: fname(*)
         integer(c_int)                     :: rc
      end function

      function XGBoosterSaveModel(handle, fname) bind(C, name="XGBoosterSaveModel") result(rc)
         import :: c_int, c_ptr, c_char
         type(c_ptr), value                 :: handle
         character(kind=c_char), intent(in) :: fname(*)
         integer(c_int)                     :: rc
      end function

      function XGBoosterPredict(handle, dmat, option_mask, ntree_limit, training, length, prediction) &
            bind(C, name="XGBoosterPredict") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value              :: handle, dmat
         integer(c_int), value           :: option_mask, ntree_limit, training
         integer(c_int64_t), intent(out) :: length
         type(c_ptr), intent(out)        :: prediction
         integer(c_int)                  :: rc
      end function

      ! ---- beyond the reference's bindings ----
      function XGBoosterSetParam(handle, name, val) bind(C, name="XGBoosterSetParam") result(rc)
         import :: c_int, c_ptr, c_char
         type(c_ptr), value                 :: handle
         character(kind=c_char), intent(in) :: name(*), val(*)
         integer(c_int)                     :: rc
      end function

      function XGBGetLastError_c() bind(C, name="XGBGetLastError") result(msg)
         import :: c_ptr
         type(c_ptr) :: msg
      end function

      ! The whole RUN section of predict_OH_with_XGB in one kernel (ohxgb.h).
      function OHXBoosterPredictFields(handle, fields, is2d, nfield, pl_feature, im, jm, km, k1, k2, missing, &
                                       apply_pow10, ohscale, oh_ml, margin) &
            bind(C, name="OHXBoosterPredictFields") result(rc)
         import :: c_int, c_ptr, c_float, c_int32_t
         type(c_ptr), value             :: handle
         type(c_ptr), intent(in)        :: fields(*)
         integer(c_int32_t), intent(in) :: is2d(*)
         integer(c_int), value          :: nfield, pl_feature, im, jm, km, k1, k2
         real(c_float), value           :: missing
         integer(c_int), value          :: apply_pow10
         real(c_float), value           :: ohscale
         type(c_ptr), value             :: oh_ml, margin
         integer(c_int)                 :: rc
      end function

      ! Optional hint: the DMatrix rows are rows row0.. of the (im,jm,*) gather (ohxgb.h); speed only.
      function OHXDMatrixSetGrid(handle, im, jm, row0) bind(C, name="OHXDMatrixSetGrid") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value        :: handle
         integer(c_int), value     :: im, jm
         integer(c_int64_t), value :: row0
         integer(c_int)            :: rc
      end function

      ! ---- part 4 of ohxgb.h: the OH field reassembled on every GPU of a node, for a host that has MPI but no
      !      torch.distributed.  Rank 0 gets the id, MPI_Bcast carries its OHX_UNIQUE_ID_BYTES bytes, every rank
      !      (hipSetDevice done) inits; d_shard / d_full are DEVICE addresses, stream a hipStream_t (c_null_ptr = default)
      function OHXCommGetUniqueId(id) bind(C, name="OHXCommGetUniqueId") result(rc)
         import :: c_int, c_char
         character(kind=c_char), intent(out) :: id(*)
         integer(c_int)                      :: rc
      end function

      function OHXCommInitRank(id, nranks, rank, comm) bind(C, name="OHXCommInitRank") result(rc)
         import :: c_int, c_char, c_ptr
         character(kind=c_char), intent(in) :: id(*)
         integer(c_int), value              :: nranks, rank
         type(c_ptr), intent(out)           :: comm
         integer(c_int)                     :: rc
      end function

      function OHXCommFree(comm) bind(C, name="OHXCommFree") result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: comm
         integer(c_int)     :: rc
      end function

      function OHXCommInfo(rccl_version) bind(C, name="OHXCommInfo") result(rc)
         import :: c_int
         integer(c_int), intent(out) :: rccl_version
         integer(c_int)              :: rc
      end function

      function OHXShardRows(nrows_total, nranks, rank, row0, nrows) bind(C, name="OHXShardRows") result(rc)
         import :: c_int, c_int64_t
         integer(c_int64_t), value       :: nrows_total
         integer(c_int), value           :: nranks, rank
         integer(c_int64_t), intent(out) :: row0, nrows
         integer(c_int)                  :: rc
      end function

      function OHXAllGatherOH(comm, d_shard, nrows_local, nrows_total, d_full, stream) &
            bind(C, name="OHXAllGatherOH") result(rc)
         import :: c_int, c_ptr, c_int64_t
         type(c_ptr), value        :: comm, d_shard, d_full, stream
         integer(c_int64_t), value :: nrows_local, nrows_total
         integer(c_int)            :: rc
      end function

      function c_strlen(s) bind(C, name="strlen") result(n)
         import :: c_ptr, c_size_t
         type(c_ptr), value :: s
         integer(c_size_t)  :: n
      end function
   end interface

contains

   !  NUL-terminated copy of a trimmed Fortran string
   function ohx_c_string(s) result(cs)
      character(len=*), intent(in) :: s
      character(kind=c_char, len=:), allocatable :: cs
      cs = trim(s)//c_null_char
   end function

   !  The library's per-thread error text (the reference never asks for it; it only
   !  asserts rc == 0, OH_GridCompMod.F90:252-265,353-378).
   function ohx_last_error() result(msg)
      character(len=:), allocatable :: msg
      type(c_ptr) :: p
      character(kind=c_char), pointer :: chars(:)
      integer :: n, i
      p = XGBGetLastError_c()
      if (.not. c_associated(p)) then
         msg = ''
         return
      end if
      n = int(c_strlen(p))
      call c_f_pointer(p, chars, [n])
      allocate(character(len=n) :: msg)
      do i = 1, n
         msg(i:i) = chars(i)
      end do
   end function

end module ohx_bindings
