#include "MAPL_Generic.h"
!  OH_StandaloneCap -- a parent grid component that runs OH instances and nothing else.
!
!  Inside GEOS the OH child hangs under QuickChem's own parent, QuickChem_GridCompMod.F90, which this repository does
!  not carry (it is the reference's file; tests compile it in place where /root/reference is present, oracle/Makefile
!  `ref`).  This cap is for everywhere else: a host that wants the OH child alone - the product's own tests and
!  drivers on a machine without the reference tree, a benchmark, another model - gets a parent written against the same
!  MAPL calls, so that `make -C quickchem_amd/fortran` always yields a GridComp that runs (ADVICE r4).
!
!  It is NOT a QuickChem parent: it knows one species, keeps no instance table, answers no IS_QC_INSTANCE_RUNNING.
!  What it shares with one is what any MAPL parent of OH must do (QuickChem_GridCompMod.F90:126-128,185,339,419,457-474):
!    * the instance names come from QuickChem_GridComp.rc, ACTIVE_INSTANCES_OH: first, then PASSIVE_INSTANCES_OH:
!    * every name becomes a child whose SetServices is OH's (MAPL_AddChild)
!    * the parent's export OH is the first child's
!    * run phase 1 runs every child's phase 1; run phase 2 the phase 2 of the children that registered one - the
!      computational instances (a name without 'data', Shared/QuickChem_Generic.F90:50-77)
module OH_StandaloneCap
   use ESMF
   use MAPL
   use OH_GridCompMod, only: OH_SetServices => SetServices
   implicit none
   private
   public :: SetServices

contains

   subroutine SetServices(GC, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      integer, optional, intent(out) :: RC
      type(ESMF_Config) :: cfg
      character(len=ESMF_MAXSTR) :: name
      character(len=32) :: label(2)
      integer :: which, n, q, id, first
      __Iam__('OH_StandaloneCap::SetServices')

      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_INITIALIZE, Initialize, __RC__)
      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_RUN, RunChildren1, __RC__)
      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_RUN, RunChildren2, __RC__)

      label = [character(len=32) :: 'ACTIVE_INSTANCES_OH:', 'PASSIVE_INSTANCES_OH:']
      cfg = ESMF_ConfigCreate(__RC__)
      call ESMF_ConfigLoadFile(cfg, 'QuickChem_GridComp.rc', __RC__)
      first = 0
      do which = 1, 2
         n = ESMF_ConfigGetLen(cfg, label=trim(label(which)), rc=STATUS)
         if (STATUS /= ESMF_SUCCESS) n = 0                      ! a list that is not there is an empty list
         if (n == 0) cycle
         call ESMF_ConfigFindLabel(cfg, trim(label(which)), __RC__)
         do q = 1, n
            call ESMF_ConfigGetAttribute(cfg, name, __RC__)
            id = MAPL_AddChild(GC, NAME=trim(name), SS=OH_SetServices, __RC__)
            if (first == 0) first = id
         end do
      end do
      call ESMF_ConfigDestroy(cfg, __RC__)
      if (first > 0) then
         call MAPL_AddExportSpec(GC, SHORT_NAME='OH', CHILD_ID=first, __RC__)
      end if
      call MAPL_GenericSetServices(GC, __RC__)
      RETURN_(ESMF_SUCCESS)
   end subroutine SetServices

   subroutine Initialize(GC, IMPORT, EXPORT, CLOCK, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: IMPORT, EXPORT
      type(ESMF_Clock), intent(inout) :: CLOCK
      integer, optional, intent(out) :: RC
      __Iam__('OH_StandaloneCap::Initialize')
      call MAPL_GenericInitialize(GC, IMPORT, EXPORT, CLOCK, __RC__)       ! the children's Initialize with it
      RETURN_(ESMF_SUCCESS)
   end subroutine Initialize

   subroutine RunChildren1(GC, IMPORT, EXPORT, CLOCK, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: IMPORT, EXPORT
      type(ESMF_Clock), intent(inout) :: CLOCK
      integer, optional, intent(out) :: RC
      __Iam__('OH_StandaloneCap::RunChildren1')
      call run_children(GC, CLOCK, 1, __RC__)
      RETURN_(ESMF_SUCCESS)
   end subroutine RunChildren1

   subroutine RunChildren2(GC, IMPORT, EXPORT, CLOCK, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: IMPORT, EXPORT
      type(ESMF_Clock), intent(inout) :: CLOCK
      integer, optional, intent(out) :: RC
      __Iam__('OH_StandaloneCap::RunChildren2')
      call run_children(GC, CLOCK, 2, __RC__)
      RETURN_(ESMF_SUCCESS)
   end subroutine RunChildren2

   subroutine run_children(GC, CLOCK, phase, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_Clock), intent(inout) :: CLOCK
      integer, intent(in) :: phase
      integer, optional, intent(out) :: RC
      type(MAPL_MetaComp), pointer :: meta
      type(ESMF_GridComp), pointer :: gcs(:)
      type(ESMF_State), pointer :: gim(:), gex(:)
      character(len=ESMF_MAXSTR) :: name
      integer :: c
      __Iam__('OH_StandaloneCap::run_children')
      call MAPL_GetObjectFromGC(GC, meta, __RC__)
      call MAPL_Get(meta, GCS=gcs, GIM=gim, GEX=gex, __RC__)
      do c = 1, size(gcs)
         if (phase == 2) then                      ! a data-driven instance has no second phase
            call ESMF_GridCompGet(gcs(c), NAME=name, __RC__)
            if (index(name, 'data') > 0) cycle
         end if
         call ESMF_GridCompRun(gcs(c), importState=gim(c), exportState=gex(c), clock=CLOCK, phase=phase, __RC__)
      end do
      RETURN_(ESMF_SUCCESS)
   end subroutine run_children

end module OH_StandaloneCap
