!  mapl_lite/MAPL.F90 -- a mock of the small part of MAPL that QuickChem's grid components touch, under MAPL's OWN
!  module name, procedure names and keyword arguments (see ESMF.F90 in this directory for why).
!
!  Restated, and only as far as the reference relies on it:
!     MAPL_AddImportSpec / MAPL_AddExportSpec / MAPL_AddInternalSpec (SHORT_NAME=, LONG_NAME=, UNITS=, DIMS=,
!        VLOCATION=, RESTART=, REFRESH_INTERVAL=, AVERAGING_INTERVAL=, UNGRIDDED_DIMS=, ADD2EXPORT=, FRIENDLYTO=,
!        CHILD_ID=, RC=)                               OH_GridCompMod.F90:611-783, QuickChem_GridCompMod.F90:185
!     MAPL_GetPointer(state, ptr, name, RC=): an export nobody asked for gives a null pointer, not an error
!                                                      OH_GridCompMod.F90:1553,1571-1572,1598-1728
!     MAPL_GridCompSetEntryPoint: run methods registered in order become phase 1, 2, ...         :601-605
!     MAPL_AddChild runs the child's SetServices at once and returns its id       QuickChem_GridCompMod.F90:531
!     MAPL_GenericInitialize: storage for the INTERNAL state, the run alarm from <COMP>_DT and
!        <COMP>_REFERENCE_TIME of the universal config, backed off by one heartbeat "since the clock is advanced
!        AFTER the run method", then the children                  OH_GridCompMod.F90:897; NOTES.wiki:57-59
!     MAPL_Get(meta, INTERNAL_ESMF_STATE=, LATS=, LONS=, RUNALARM=, GCS=, GIM=, GEX=, RC=), MAPL_GetResource,
!     MAPL_GridGet, MAPL_GetObjectFromGC, MAPL_PackTime, MAPL_AM_I_ROOT, MAPL_MaxMin, the constants
!  Everything else MAPL does (ExtData, restarts, HISTORY, couplers, timers, the MPI layout) is not here.
!  MAPL_VRFY / MAPL_ASRT / MAPL_RTRN serve this directory's MAPL_Generic.h.  Names beginning with mapll_ are the
!  mock's own (the cap uses them); nothing in the grid components does.
module MAPL
   use, intrinsic :: iso_fortran_env, only: int64, real64
   use ESMF
   implicit none
   private

   !  MAPL's constants, as MAPL defines them (PhysicalConstants / MathConstants)
   real(real64), parameter, public :: MAPL_PI_R8 = 3.14159265358979323846d0
   real, parameter, public :: MAPL_PI = MAPL_PI_R8
   real, parameter, public :: MAPL_DEGREES_TO_RADIANS = MAPL_PI / 180.0
   real, parameter, public :: MAPL_RADIANS_TO_DEGREES = 180.0 / MAPL_PI
   real, parameter, public :: MAPL_AVOGAD = 6.023e26      ! molec / kmol
   real, parameter, public :: MAPL_RUNIV  = 8314.47       ! J / (kmol K)
   real, parameter, public :: MAPL_H2OMW  = 18.015, MAPL_AIRMW = 28.965
   real, parameter, public :: MAPL_EPSILON = MAPL_H2OMW / MAPL_AIRMW

   integer, parameter, public :: MAPL_DimsHorzOnly = ESMFL_DIMS_HORZ_ONLY, MAPL_DimsHorzVert = ESMFL_DIMS_HORZ_VERT
   integer, parameter, public :: MAPL_VLocationNone = ESMFL_VLOC_NONE, MAPL_VLocationCenter = ESMFL_VLOC_CENTER, &
                                 MAPL_VLocationEdge = ESMFL_VLOC_EDGE
   integer, parameter, public :: MAPL_RestartOptional = 0, MAPL_RestartSkip = 1, MAPL_RestartRequired = 2

   !  MAPL's generic state of a component
   type, public :: MAPL_MetaComp
      type(ESMF_State) :: internal
      type(ESMF_Alarm) :: runalarm
      type(ESMF_Grid) :: grid
      type(ESMF_Config) :: cf
      character(len=ESMF_MAXSTR) :: name = ''
      type(ESMF_GridComp), pointer :: gcs(:) => null()
      type(ESMF_State), pointer :: gim(:) => null(), gex(:) => null()
      integer :: nchildren = 0
      integer :: n_run_phases = 0, n_init_phases = 0
      !  AddExportSpec(SHORT_NAME=..., CHILD_ID=...): exports of this component that ARE a child's field
      character(len=ESMF_MAXSTR), allocatable :: child_export_name(:)
      integer, allocatable :: child_export_id(:)
   end type

   public :: MAPL_GetObjectFromGC, MAPL_Get, MAPL_GetResource, MAPL_GridGet
   public :: MAPL_AddImportSpec, MAPL_AddExportSpec, MAPL_AddInternalSpec
   public :: MAPL_GridCompSetEntryPoint, MAPL_GenericSetServices, MAPL_GenericInitialize, MAPL_AddChild
   public :: MAPL_GetPointer, MAPL_PackTime, MAPL_AM_I_ROOT, MAPL_MaxMin
   public :: MAPL_VRFY, MAPL_ASRT, MAPL_RTRN
   public :: mapll_child_export_field

   interface MAPL_GetPointer
      module procedure get_pointer_2d, get_pointer_3d, get_pointer_4d
   end interface

   interface MAPL_GetResource
      module procedure get_resource_int, get_resource_real, get_resource_string, get_resource_logical
   end interface

contains

   subroutine set_rc(rc, value)
      integer, intent(out), optional :: rc
      integer, intent(in) :: value
      if (present(rc)) rc = value
   end subroutine

   ! =================================================================== what MAPL_Generic.h expands to

   !  VERIFY_(A): a non-zero status ends the caller, with the status as its RC
   logical function MAPL_VRFY(A, iam, line, rc)
      integer, intent(in) :: A
      character(len=*), intent(in) :: iam
      integer, intent(in) :: line
      integer, intent(out), optional :: rc
      MAPL_VRFY = A /= ESMF_SUCCESS
      if (MAPL_VRFY) then
         if (present(rc)) rc = A
         print '(a,i0,a,i0)', trim(iam)//' ', line, ' status=', A
      end if
   end function

   !  _ASSERT(cond, message)
   logical function MAPL_ASRT(cond, message, iam, line, rc)
      logical, intent(in) :: cond
      character(len=*), intent(in) :: message, iam
      integer, intent(in) :: line
      integer, intent(out), optional :: rc
      MAPL_ASRT = .not. cond
      if (MAPL_ASRT) then
         if (present(rc)) rc = ESMF_FAILURE
         print '(a,i0,a)', trim(iam)//' ', line, ' '//trim(message)
      end if
   end function

   !  RETURN_(A)
   logical function MAPL_RTRN(A, iam, line, rc)
      integer, intent(in) :: A
      character(len=*), intent(in) :: iam
      integer, intent(in) :: line
      integer, intent(out), optional :: rc
      MAPL_RTRN = .true.
      if (present(rc)) rc = A
      if (A /= ESMF_SUCCESS) print '(a,i0,a,i0)', trim(iam)//' ', line, ' returns ', A
   end function

   ! =================================================================== generic state

   !  MAPL keeps its generic state inside the ESMF component and makes it at first use
   function meta_of(gc) result(meta)
      type(ESMF_GridComp), intent(in) :: gc
      type(MAPL_MetaComp), pointer :: meta
      meta => null()
      if (.not. associated(gc%p)) return
      if (.not. associated(gc%p%generic_state)) then
         allocate(meta)
         meta%internal = ESMF_StateCreate(name=trim(gc%p%name)//'_Internals')
         meta%grid = gc%p%grid
         meta%cf = gc%p%config
         meta%name = gc%p%name
         allocate(meta%gcs(8), meta%gim(8), meta%gex(8))
         allocate(meta%child_export_name(0), meta%child_export_id(0))
         gc%p%generic_state => meta
         return
      end if
      select type (p => gc%p%generic_state)
      type is (MAPL_MetaComp)
         meta => p
      end select
   end function

   subroutine MAPL_GetObjectFromGC(GC, MAPLOBJ, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(MAPL_MetaComp), pointer :: MAPLOBJ
      integer, intent(out), optional :: RC
      MAPLOBJ => meta_of(GC)
      call set_rc(RC, merge(ESMF_SUCCESS, ESMF_FAILURE, associated(MAPLOBJ)))
   end subroutine

   !  gcs / gim / gex name the children in the order they were added
   subroutine MAPL_Get(STATE, INTERNAL_ESMF_STATE, LATS, LONS, RUNALARM, GCS, GIM, GEX, IM, JM, LM, CF, RC)
      type(MAPL_MetaComp), intent(inout) :: STATE
      type(ESMF_State), intent(out), optional :: INTERNAL_ESMF_STATE
      real, pointer, optional :: LATS(:,:), LONS(:,:)
      type(ESMF_Alarm), intent(out), optional :: RUNALARM
      type(ESMF_GridComp), pointer, optional :: GCS(:)
      type(ESMF_State), pointer, optional :: GIM(:), GEX(:)
      integer, intent(out), optional :: IM, JM, LM
      type(ESMF_Config), intent(out), optional :: CF
      integer, intent(out), optional :: RC
      if (present(INTERNAL_ESMF_STATE)) INTERNAL_ESMF_STATE = STATE%internal
      if (present(LATS)) LATS => STATE%grid%p%LATS
      if (present(LONS)) LONS => STATE%grid%p%LONS
      if (present(RUNALARM)) RUNALARM = STATE%runalarm
      if (present(GCS)) GCS => STATE%gcs(1:STATE%nchildren)
      if (present(GIM)) GIM => STATE%gim(1:STATE%nchildren)
      if (present(GEX)) GEX => STATE%gex(1:STATE%nchildren)
      if (present(IM)) IM = STATE%grid%p%im
      if (present(JM)) JM = STATE%grid%p%jm
      if (present(LM)) LM = STATE%grid%p%km
      if (present(CF)) CF = STATE%cf
      call set_rc(RC, ESMF_SUCCESS)
   end subroutine

   !  a key of the universal config (AGCM.rc); MAPL also looks for "<COMP>_<label>", which nothing here needs
   subroutine get_resource_int(STATE, VAL, LABEL, DEFAULT, RC)
      type(MAPL_MetaComp), intent(inout) :: STATE
      integer, intent(out) :: VAL
      character(len=*), intent(in) :: LABEL
      integer, intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      if (.not. associated(STATE%cf%p)) then
         VAL = 0
         if (present(DEFAULT)) VAL = DEFAULT
         call set_rc(RC, merge(ESMF_SUCCESS, ESMF_FAILURE, present(DEFAULT)))
         return
      end if
      call ESMF_ConfigGetAttribute(STATE%cf, VAL, label=LABEL, default=DEFAULT, rc=RC)
   end subroutine

   subroutine get_resource_real(STATE, VAL, LABEL, DEFAULT, RC)
      type(MAPL_MetaComp), intent(inout) :: STATE
      real, intent(out) :: VAL
      character(len=*), intent(in) :: LABEL
      real, intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      if (.not. associated(STATE%cf%p)) then
         VAL = 0.0
         if (present(DEFAULT)) VAL = DEFAULT
         call set_rc(RC, merge(ESMF_SUCCESS, ESMF_FAILURE, present(DEFAULT)))
         return
      end if
      call ESMF_ConfigGetAttribute(STATE%cf, VAL, label=LABEL, default=DEFAULT, rc=RC)
   end subroutine

   subroutine get_resource_string(STATE, VAL, LABEL, DEFAULT, RC)
      type(MAPL_MetaComp), intent(inout) :: STATE
      character(len=*), intent(out) :: VAL
      character(len=*), intent(in) :: LABEL
      character(len=*), intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      if (.not. associated(STATE%cf%p)) then
         VAL = ''
         if (present(DEFAULT)) VAL = DEFAULT
         call set_rc(RC, merge(ESMF_SUCCESS, ESMF_FAILURE, present(DEFAULT)))
         return
      end if
      call ESMF_ConfigGetAttribute(STATE%cf, VAL, label=LABEL, default=DEFAULT, rc=RC)
   end subroutine

   subroutine get_resource_logical(STATE, VAL, LABEL, DEFAULT, RC)
      type(MAPL_MetaComp), intent(inout) :: STATE
      logical, intent(out) :: VAL
      character(len=*), intent(in) :: LABEL
      logical, intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      if (.not. associated(STATE%cf%p)) then
         VAL = .false.
         if (present(DEFAULT)) VAL = DEFAULT
         call set_rc(RC, merge(ESMF_SUCCESS, ESMF_FAILURE, present(DEFAULT)))
         return
      end if
      call ESMF_ConfigGetAttribute(STATE%cf, VAL, label=LABEL, default=DEFAULT, rc=RC)
   end subroutine

   !  one rank: the local block is the globe
   subroutine MAPL_GridGet(GRID, globalCellCountPerDim, localCellCountPerDim, RC)
      type(ESMF_Grid), intent(in) :: GRID
      integer, intent(out), optional :: globalCellCountPerDim(3), localCellCountPerDim(3)
      integer, intent(out), optional :: RC
      call set_rc(RC, ESMF_FAILURE)
      if (.not. associated(GRID%p)) return
      if (present(globalCellCountPerDim)) globalCellCountPerDim = [GRID%p%im, GRID%p%jm, GRID%p%km]
      if (present(localCellCountPerDim)) localCellCountPerDim = [GRID%p%im, GRID%p%jm, GRID%p%km]
      call set_rc(RC, ESMF_SUCCESS)
   end subroutine

   ! =================================================================== specs

   function spec_field(SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, RESTART, REFRESH_INTERVAL, AVERAGING_INTERVAL, &
                       UNGRIDDED_DIMS, ADD2EXPORT) result(f)
      character(len=*), intent(in) :: SHORT_NAME
      character(len=*), intent(in), optional :: LONG_NAME, UNITS
      integer, intent(in), optional :: DIMS, VLOCATION, RESTART, REFRESH_INTERVAL, AVERAGING_INTERVAL
      integer, intent(in), optional :: UNGRIDDED_DIMS(:)
      logical, intent(in), optional :: ADD2EXPORT
      type(esmfl_field) :: f
      f%name = SHORT_NAME
      if (present(LONG_NAME)) f%long_name = LONG_NAME
      if (present(UNITS)) f%units = UNITS
      if (present(DIMS)) f%dims = DIMS
      f%vloc = merge(MAPL_VLocationNone, MAPL_VLocationCenter, f%dims == MAPL_DimsHorzOnly)
      if (present(VLOCATION)) f%vloc = VLOCATION
      if (present(RESTART)) f%restart = RESTART
      if (present(REFRESH_INTERVAL)) f%refresh_interval = REFRESH_INTERVAL
      if (present(AVERAGING_INTERVAL)) f%averaging_interval = AVERAGING_INTERVAL
      if (present(UNGRIDDED_DIMS)) then
         if (size(UNGRIDDED_DIMS) > 0) f%ungridded = UNGRIDDED_DIMS(1)
      end if
      if (present(ADD2EXPORT)) f%add2export = ADD2EXPORT
   end function

   subroutine MAPL_AddImportSpec(GC, SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, RESTART, REFRESH_INTERVAL, &
                                 AVERAGING_INTERVAL, UNGRIDDED_DIMS, FRIENDLYTO, DEFAULT, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      character(len=*), intent(in) :: SHORT_NAME
      character(len=*), intent(in), optional :: LONG_NAME, UNITS, FRIENDLYTO
      integer, intent(in), optional :: DIMS, VLOCATION, RESTART, REFRESH_INTERVAL, AVERAGING_INTERVAL
      integer, intent(in), optional :: UNGRIDDED_DIMS(:)
      real, intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      integer :: status
      call esmfl_state_add(GC%p%importState, spec_field(SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, RESTART, &
                           REFRESH_INTERVAL, AVERAGING_INTERVAL, UNGRIDDED_DIMS), status)
      call set_rc(RC, status)
   end subroutine

   !  with CHILD_ID the parent re-exports a child's field of that name (QuickChem_GridCompMod.F90:185)
   subroutine MAPL_AddExportSpec(GC, SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, UNGRIDDED_DIMS, CHILD_ID, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      character(len=*), intent(in) :: SHORT_NAME
      character(len=*), intent(in), optional :: LONG_NAME, UNITS
      integer, intent(in), optional :: DIMS, VLOCATION
      integer, intent(in), optional :: UNGRIDDED_DIMS(:)
      integer, intent(in), optional :: CHILD_ID
      integer, intent(out), optional :: RC
      type(MAPL_MetaComp), pointer :: meta
      integer :: status
      if (present(CHILD_ID)) then
         meta => meta_of(GC)
         call set_rc(RC, ESMF_FAILURE)
         if (CHILD_ID < 1 .or. CHILD_ID > meta%nchildren) return
         meta%child_export_name = [character(len=ESMF_MAXSTR) :: meta%child_export_name, SHORT_NAME]
         meta%child_export_id = [meta%child_export_id, CHILD_ID]
         call set_rc(RC, ESMF_SUCCESS)
         return
      end if
      call esmfl_state_add(GC%p%exportState, spec_field(SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, &
                           UNGRIDDED_DIMS=UNGRIDDED_DIMS), status)
      call set_rc(RC, status)
   end subroutine

   subroutine MAPL_AddInternalSpec(GC, SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, RESTART, UNGRIDDED_DIMS, &
                                   FRIENDLYTO, ADD2EXPORT, DEFAULT, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      character(len=*), intent(in) :: SHORT_NAME
      character(len=*), intent(in), optional :: LONG_NAME, UNITS, FRIENDLYTO
      integer, intent(in), optional :: DIMS, VLOCATION, RESTART
      integer, intent(in), optional :: UNGRIDDED_DIMS(:)
      logical, intent(in), optional :: ADD2EXPORT
      real, intent(in), optional :: DEFAULT
      integer, intent(out), optional :: RC
      type(MAPL_MetaComp), pointer :: meta
      integer :: status
      meta => meta_of(GC)
      call esmfl_state_add(meta%internal, spec_field(SHORT_NAME, LONG_NAME, UNITS, DIMS, VLOCATION, RESTART, &
                           UNGRIDDED_DIMS=UNGRIDDED_DIMS, ADD2EXPORT=ADD2EXPORT), status)
      call set_rc(RC, status)
   end subroutine

   !  A name the state does not declare is an error; a declared field without storage (an export nobody asked
   !  for) gives a null pointer and RC = 0.  Pointers keep the field's bounds: edge fields are (im,jm,0:km).
   subroutine get_pointer_2d(STATE, PTR, NAME, ALLOC, RC)
      type(ESMF_State), intent(inout) :: STATE
      real, pointer :: PTR(:,:)
      character(len=*), intent(in) :: NAME
      logical, intent(in), optional :: ALLOC
      integer, intent(out), optional :: RC
      integer :: k
      PTR => null()
      k = esmfl_state_index(STATE, NAME)
      call set_rc(RC, ESMF_FAILURE)
      if (k == 0) return
      if (STATE%p%f(k)%dims /= MAPL_DimsHorzOnly) return
      PTR => STATE%p%f(k)%p2
      call set_rc(RC, ESMF_SUCCESS)
   end subroutine

   subroutine get_pointer_3d(STATE, PTR, NAME, ALLOC, RC)
      type(ESMF_State), intent(inout) :: STATE
      real, pointer :: PTR(:,:,:)
      character(len=*), intent(in) :: NAME
      logical, intent(in), optional :: ALLOC
      integer, intent(out), optional :: RC
      integer :: k
      PTR => null()
      k = esmfl_state_index(STATE, NAME)
      call set_rc(RC, ESMF_FAILURE)
      if (k == 0) return
      if (STATE%p%f(k)%dims /= MAPL_DimsHorzVert .or. STATE%p%f(k)%ungridded > 0) return
      PTR => STATE%p%f(k)%p3
      call set_rc(RC, ESMF_SUCCESS)
   end subroutine

   subroutine get_pointer_4d(STATE, PTR, NAME, ALLOC, RC)
      type(ESMF_State), intent(inout) :: STATE
      real, pointer :: PTR(:,:,:,:)
      character(len=*), intent(in) :: NAME
      logical, intent(in), optional :: ALLOC
      integer, intent(out), optional :: RC
      integer :: k
      PTR => null()
      k = esmfl_state_index(STATE, NAME)
      call set_rc(RC, ESMF_FAILURE)
      if (k == 0) return
      if (STATE%p%f(k)%ungridded <= 0) return
      PTR => STATE%p%f(k)%p4
      call set_rc(RC, ESMF_SUCCESS)
   end subroutine

   ! =================================================================== services

   !  run methods registered in order become phase 1, 2, ...
   subroutine MAPL_GridCompSetEntryPoint(GC, registeredMethod, usersRoutine, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_Method_Flag), intent(in) :: registeredMethod
      procedure(esmfl_method) :: usersRoutine
      integer, intent(out), optional :: RC
      type(MAPL_MetaComp), pointer :: meta
      integer :: phase
      meta => meta_of(GC)
      if (registeredMethod%v == ESMF_METHOD_RUN%v) then
         meta%n_run_phases = meta%n_run_phases + 1
         phase = meta%n_run_phases
      else if (registeredMethod%v == ESMF_METHOD_INITIALIZE%v) then
         meta%n_init_phases = meta%n_init_phases + 1
         phase = meta%n_init_phases
      else
         phase = 1
      end if
      call ESMF_GridCompSetEntryPoint(GC, registeredMethod, usersRoutine, phase=phase, rc=RC)
   end subroutine

   !  a component that registered no Initialize of its own gets the generic one
   subroutine MAPL_GenericSetServices(GC, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      integer, intent(out), optional :: RC
      type(MAPL_MetaComp), pointer :: meta
      meta => meta_of(GC)
      call set_rc(RC, ESMF_SUCCESS)
      if (meta%n_init_phases == 0) call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_INITIALIZE, MAPL_GenericInitialize, RC)
   end subroutine

   !  creates the child on the parent's grid and config and runs ITS SetServices; returns its id
   integer function MAPL_AddChild(GC, NAME, SS, RC) result(id)
      type(ESMF_GridComp), intent(inout) :: GC
      character(len=*), intent(in) :: NAME
      procedure(esmfl_set_services) :: SS
      integer, intent(out), optional :: RC
      type(MAPL_MetaComp), pointer :: meta
      type(ESMF_GridComp), pointer :: gcs(:)
      type(ESMF_State), pointer :: gim(:), gex(:)
      integer :: n
      meta => meta_of(GC)
      n = meta%nchildren
      if (n == size(meta%gcs)) then
         allocate(gcs(2 * n), gim(2 * n), gex(2 * n))
         gcs(1:n) = meta%gcs(1:n); gim(1:n) = meta%gim(1:n); gex(1:n) = meta%gex(1:n)
         deallocate(meta%gcs, meta%gim, meta%gex)
         meta%gcs => gcs; meta%gim => gim; meta%gex => gex
      end if
      id = n + 1
      meta%nchildren = id
      meta%gcs(id) = ESMF_GridCompCreate(name=NAME, config=GC%p%config, grid=GC%p%grid)
      meta%gim(id) = meta%gcs(id)%p%importState
      meta%gex(id) = meta%gcs(id)%p%exportState
      call ESMF_GridCompSetServices(meta%gcs(id), SS, rc=RC)
   end function

   !  what a consumer of the parent's export gets: the child's export field of that name, or - for a field the
   !  child declared ADD2EXPORT from its INTERNAL state - that internal field
   subroutine mapll_child_export_field(GC, SHORT_NAME, PTR, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      character(len=*), intent(in) :: SHORT_NAME
      real, pointer :: PTR(:,:,:)
      integer, intent(out) :: RC
      type(MAPL_MetaComp), pointer :: meta, child
      integer :: q
      PTR => null()
      RC = ESMF_FAILURE
      meta => meta_of(GC)
      do q = 1, size(meta%child_export_id)
         if (trim(meta%child_export_name(q)) /= trim(SHORT_NAME)) cycle
         child => meta_of(meta%gcs(meta%child_export_id(q)))
         if (esmfl_state_index(child%internal, SHORT_NAME) > 0) then
            call MAPL_GetPointer(child%internal, PTR, SHORT_NAME, RC=RC)
         else
            call MAPL_GetPointer(meta%gex(meta%child_export_id(q)), PTR, SHORT_NAME, RC=RC)
         end if
         return
      end do
   end subroutine

   !  As far as OH relies on it: storage for the INTERNAL state; the run alarm from <NAME>_DT /
   !  <NAME>_REFERENCE_TIME of the universal config (RUN_DT when absent: rings at every step), backed off by one
   !  heartbeat - with the default reference time it rings during the LAST step of every interval, with
   !  <NAME>_REFERENCE_TIME = the heartbeat during the FIRST (NOTES.wiki:57-59); then the children's Initialize.
   recursive subroutine MAPL_GenericInitialize(GC, IMPORT, EXPORT, CLOCK, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: IMPORT, EXPORT
      type(ESMF_Clock), intent(inout) :: CLOCK
      integer, optional, intent(out) :: RC
      type(MAPL_MetaComp), pointer :: meta
      type(ESMF_Time) :: now, ring
      type(ESMF_TimeInterval) :: step, every
      integer :: i, dt, heartbeat, ref_hms, yy, mm, dd, status
      meta => meta_of(GC)
      call set_rc(RC, ESMF_SUCCESS)
      do i = 1, meta%internal%p%n
         call esmfl_state_allocate(meta%internal, meta%internal%p%f(i)%name, meta%grid, status)
         if (status /= ESMF_SUCCESS) then
            call set_rc(RC, status)
            return
         end if
      end do
      call ESMF_ClockGet(CLOCK, currTime=now, timeStep=step)
      call ESMF_TimeIntervalGet(step, S=heartbeat)
      call MAPL_GetResource(meta, dt, LABEL=trim(meta%name)//'_DT:', DEFAULT=heartbeat)
      call MAPL_GetResource(meta, ref_hms, LABEL=trim(meta%name)//'_REFERENCE_TIME:', DEFAULT=0)
      call ESMF_TimeGet(now, YY=yy, MM=mm, DD=dd)
      call ESMF_TimeSet(ring, YY=yy, MM=mm, DD=dd, H=ref_hms / 10000, M=mod(ref_hms, 10000) / 100, S=mod(ref_hms, 100))
      ring%s = ring%s - heartbeat
      call ESMF_TimeIntervalSet(every, S=dt)
      meta%runalarm = ESMF_AlarmCreate(CLOCK, ringTime=ring, ringInterval=every, name=trim(meta%name)//'_RunAlarm')
      do i = 1, meta%nchildren
         call ESMF_GridCompInitialize(meta%gcs(i), importState=meta%gim(i), exportState=meta%gex(i), clock=CLOCK, rc=status)
         if (status /= ESMF_SUCCESS) then
            call set_rc(RC, status)
            return
         end if
      end do
   end subroutine

   ! =================================================================== small things

   subroutine MAPL_PackTime(packed, a, b, c)
      integer, intent(out) :: packed
      integer, intent(in) :: a, b, c
      packed = a * 10000 + b * 100 + c
   end subroutine

   logical function MAPL_AM_I_ROOT()
      MAPL_AM_I_ROOT = .true.          ! one rank in the mock
   end function

   !  a global max/min print (a reduction over ranks in MAPL; OH_GridCompMod.F90:1550)
   subroutine MAPL_MaxMin(label, a)
      character(len=*), intent(in) :: label
      real, intent(in), contiguous :: a(:,:,:)
      real :: hi, lo
      integer :: i, j, k
      if (.not. MAPL_AM_I_ROOT()) return
      !  one pass the compiler vectorises (maxval + minval through the descriptor were 0.4 ms on a rank's 83 k gridcells -
      !  half of the product shell's tick, and nothing either child computes)
      hi = -huge(hi); lo = huge(lo)
      do k = 1, size(a, 3)
         do j = 1, size(a, 2)
            do i = 1, size(a, 1)
               hi = max(hi, a(i,j,k)); lo = min(lo, a(i,j,k))
            end do
         end do
      end do
      print '(a,2es16.7)', trim(label)//' max, min = ', hi, lo
   end subroutine

end module MAPL
