#include "MAPL_Generic.h"
!  QuickChem_GridCompMod -- the parent grid component: reads the instance lists, creates one OH child per
!  instance, re-exports the first child's OH, and runs the children's phases.
!
!  Surface and behaviour of the reference's QuickChem_GridCompMod.F90, written against MAPL and ESMF by their own
!  names (inside GEOS: the real libraries; here: the mock in mapl_lite/):
!     SetServices      entry points Initialize, Run1, Run2 (:124-128); private state as ESMF user state
!                      'QuickChem_State' (:132); instances of OH from QuickChem_GridComp.rc (:161, getInstances_
!                      :432-482: ACTIVE_INSTANCES_OH first, PASSIVE_INSTANCES_OH after them); one child per instance
!                      through MAPL_AddChild with OH's SetServices (:509,531); export 'OH' of the FIRST instance (:185)
!     Initialize       MAPL_GenericInitialize, which initialises the children (:260)
!     Run1             every child, phase 1 (:338-340)
!     Run2             phase 2 of the children whose name does not contain 'data' (:416-421)
!     IS_QC_INSTANCE_RUNNING   is the name in either list? (:544-605)
!  No arithmetic happens here.
module QuickChem_GridCompMod
   use ESMF
   use MAPL
   use OH_GridCompMod, only: OH_setServices => SetServices
   implicit none
   private

   public :: SetServices
   public :: IS_QC_INSTANCE_RUNNING

   character(len=*), parameter :: QUICKCHEM_RESOURCE_FILE = 'QuickChem_GridComp.rc'

   type :: Instance
      integer :: id = -1
      logical :: is_active = .false.
      character(len=:), allocatable :: name
   end type Instance

   type Constituent
      type(Instance), allocatable :: instances(:)
      integer :: n_active = 0
   end type Constituent

   type QuickChem_State
      private
      type(Constituent) :: OH
   end type QuickChem_State

   type wrap_
      type(QuickChem_State), pointer :: PTR => null()
   end type wrap_

contains

   subroutine SetServices(GC, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      integer, intent(out) :: RC

      character(len=ESMF_MAXSTR) :: COMP_NAME
      type(ESMF_Config) :: myCF, cf
      type(QuickChem_State), pointer :: self
      type(wrap_) :: wrap
      integer :: i

      __Iam__('SetServices')

      call ESMF_GridCompGet(GC, name=COMP_NAME, config=cf, __RC__)
      Iam = trim(COMP_NAME)//'::'//'SetServices'

      allocate(self, __STAT__)
      wrap%ptr => self
      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_INITIALIZE, Initialize, __RC__)
      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_RUN, Run1, __RC__)
      call MAPL_GridCompSetEntryPoint(GC, ESMF_METHOD_RUN, Run2, __RC__)
      call ESMF_UserCompSetInternalState(GC, 'QuickChem_State', wrap, STATUS)
      VERIFY_(STATUS)

      myCF = ESMF_ConfigCreate(__RC__)
      call ESMF_ConfigLoadFile(myCF, QUICKCHEM_RESOURCE_FILE, __RC__)
      call getInstances_('OH', myCF, species=self%OH, __RC__)
      call ESMF_ConfigDestroy(myCF, __RC__)

      !  children are created in list order: active instances first
      do i = 1, size(self%OH%instances)
         self%OH%instances(i)%id = MAPL_AddChild(GC, name=self%OH%instances(i)%name, SS=OH_setServices, __RC__)
      end do
      !  "Allow children of Chemistry to connect to these fields"
      if (size(self%OH%instances) > 0) then
         call MAPL_AddExportSpec(GC, SHORT_NAME='OH', CHILD_ID=self%OH%instances(1)%id, __RC__)
      end if
      call MAPL_GenericSetServices(GC, __RC__)
      RETURN_(ESMF_SUCCESS)
   end subroutine SetServices

   subroutine Initialize(GC, import, export, clock, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: import, export
      type(ESMF_Clock), intent(inout) :: clock
      integer, optional, intent(out) :: RC
      character(len=ESMF_MAXSTR) :: COMP_NAME

      __Iam__('Initialize')

      call ESMF_GridCompGet(GC, name=COMP_NAME, __RC__)
      Iam = trim(COMP_NAME)//'::'//'Initialize'
      if (MAPL_AM_I_ROOT()) print *, trim(Iam)//': Starting...'
      call MAPL_GenericInitialize(GC, import, export, clock, __RC__)     ! initialises the children
      RETURN_(ESMF_SUCCESS)
   end subroutine Initialize

   subroutine Run1(GC, import, export, clock, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: import, export
      type(ESMF_Clock), intent(inout) :: clock
      integer, optional, intent(out) :: RC
      character(len=ESMF_MAXSTR) :: COMP_NAME
      type(MAPL_MetaComp), pointer :: meta
      type(ESMF_GridComp), pointer :: gcs(:)
      type(ESMF_State), pointer :: gim(:), gex(:)
      integer :: i

      __Iam__('Run1')

      call ESMF_GridCompGet(GC, NAME=COMP_NAME, __RC__)
      if (index(Iam, '::') == 0) Iam = trim(COMP_NAME)//'::'//Iam
      call MAPL_GetObjectFromGC(GC, meta, __RC__)
      call MAPL_Get(meta, gcs=gcs, gim=gim, gex=gex, __RC__)
      do i = 1, size(gcs)
         call ESMF_GridCompRun(gcs(i), importState=gim(i), exportState=gex(i), phase=1, clock=clock, __RC__)
      end do
      RETURN_(ESMF_SUCCESS)
   end subroutine Run1

   subroutine Run2(GC, import, export, clock, RC)
      type(ESMF_GridComp), intent(inout) :: GC
      type(ESMF_State), intent(inout) :: import, export
      type(ESMF_Clock), intent(inout) :: clock
      integer, optional, intent(out) :: RC
      character(len=ESMF_MAXSTR) :: COMP_NAME, child_name
      type(MAPL_MetaComp), pointer :: meta
      type(ESMF_GridComp), pointer :: gcs(:)
      type(ESMF_State), pointer :: gim(:), gex(:)
      integer :: i

      __Iam__('Run2')

      call ESMF_GridCompGet(GC, NAME=COMP_NAME, __RC__)
      if (index(Iam, '::') == 0) Iam = trim(COMP_NAME)//'::'//Iam
      call MAPL_GetObjectFromGC(GC, meta, __RC__)
      call MAPL_Get(meta, gcs=gcs, gim=gim, gex=gex, __RC__)
      do i = 1, size(gcs)
         call ESMF_GridCompGet(gcs(i), NAME=child_name, __RC__)
         if (index(child_name, 'data') == 0) then      ! only a computational instance has a phase 2
            call ESMF_GridCompRun(gcs(i), importState=gim(i), exportState=gex(i), phase=2, clock=clock, __RC__)
         end if
      end do
      RETURN_(ESMF_SUCCESS)
   end subroutine Run2

   !  the instance names of a species: the active list first, then the passive one
   subroutine getInstances_(species_name, myCF, species, rc)
      character(len=*), intent(in) :: species_name
      type(ESMF_Config), intent(inout) :: myCF
      type(Constituent), intent(inout) :: species
      integer, intent(out) :: rc
      character(len=*), parameter :: list(2) = [character(len=18) :: 'ACTIVE_INSTANCES_', 'PASSIVE_INSTANCES_']
      character(len=ESMF_MAXSTR) :: inst_name
      integer :: i, q, n(2), first

      __Iam__('QuickChem::getInstances_')

      do q = 1, 2
         n(q) = ESMF_ConfigGetLen(myCF, label=trim(list(q))//trim(species_name)//':', __RC__)
      end do
      allocate(species%instances(n(1) + n(2)), __STAT__)
      species%n_active = n(1)
      first = 0
      do q = 1, 2
         call ESMF_ConfigFindLabel(myCF, trim(list(q))//trim(species_name)//':', __RC__)
         do i = first + 1, first + n(q)
            call ESMF_ConfigGetAttribute(myCF, inst_name, __RC__)
            species%instances(i)%name = trim(inst_name)
            species%instances(i)%is_active = q == 1
         end do
         first = first + n(q)
      end do
      RETURN_(ESMF_SUCCESS)
   end subroutine getInstances_

   subroutine IS_QC_INSTANCE_RUNNING(species_name, instance_name, running, RC)
      character(len=*), intent(in) :: species_name, instance_name
      logical, intent(out) :: running
      integer, optional :: RC
      type(ESMF_Config) :: myCF
      type(Constituent) :: species
      integer :: i

      __Iam__('QuickChem::IS_QC_INSTANCE_RUNNING')

      running = .false.
      myCF = ESMF_ConfigCreate(__RC__)
      call ESMF_ConfigLoadFile(myCF, QUICKCHEM_RESOURCE_FILE, __RC__)
      call getInstances_(species_name, myCF, species, __RC__)
      call ESMF_ConfigDestroy(myCF, __RC__)
      do i = 1, size(species%instances)
         if (trim(species%instances(i)%name) == trim(instance_name)) running = .true.
      end do
      RETURN_(ESMF_SUCCESS)
   end subroutine IS_QC_INSTANCE_RUNNING

end module QuickChem_GridCompMod
