!  QuickChem_GridCompMod -- the parent grid component: reads the instance lists, creates one OH child per
!  instance, re-exports the first child's OH, and runs the children's phases.
!
!  Surface and behaviour of the reference's QuickChem_GridCompMod.F90:
!     SetServices      entry points Initialize, Run1, Run2 (:124-126); instances of OH from
!                      QuickChem_GridComp.rc (:161, getInstances_ :432-482: ACTIVE_INSTANCES_OH first,
!                      PASSIVE_INSTANCES_OH after them); one child per instance through MAPL_AddChild with
!                      OH's SetServices (:509,531); export 'OH' of the FIRST instance (:185)
!     Run1             every child, phase 1 (:338-340)
!     Run2             phase 2 of the children whose name does not contain 'data' (:416-421)
!     IS_QC_INSTANCE_RUNNING   is the name in either list? (:544-605)
!  written against mapl_lite.  No arithmetic happens here.
module QuickChem_GridCompMod
   use mapl_lite
   use OH_GridCompMod, only: OH_setServices => SetServices
   implicit none
   private

   public :: SetServices
   public :: IS_QC_INSTANCE_RUNNING

   character(len=*), parameter :: QUICKCHEM_RESOURCE_FILE = 'QuickChem_GridComp.rc'

   type :: Instance
      integer :: id = -1
      logical :: is_active = .false.
      character(len=ML_MAXSTR) :: name = ''
   end type Instance

   type Constituent
      type(Instance), allocatable :: instances(:)
      integer :: n_active = 0
   end type Constituent

   type QuickChem_State
      type(Constituent) :: OH
   end type QuickChem_State

contains

   subroutine SetServices(GC, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      integer, intent(out) :: RC
      type(QuickChem_State), pointer :: self
      type(ml_config) :: myCF
      integer :: i

      allocate(self)
      call ml_set_entry_point(GC, ML_METHOD_INITIALIZE, Initialize, RC)
      call ml_set_entry_point(GC, ML_METHOD_RUN, Run1, RC)
      call ml_set_entry_point(GC, ML_METHOD_RUN, Run2, RC)
      GC%private_state => self
      GC%private_key = 'QuickChem_State'

      call myCF%load(trim(GC%rc_dir)//'/'//QUICKCHEM_RESOURCE_FILE, RC)
      if (RC /= ML_SUCCESS) then
         if (ml_am_i_root()) print *, 'QuickChem: cannot read '//QUICKCHEM_RESOURCE_FILE
         return
      end if
      call getInstances_('OH', myCF, self%OH, RC)
      if (RC /= ML_SUCCESS) return

      !  children are created in list order: active instances first
      do i = 1, size(self%OH%instances)
         self%OH%instances(i)%id = ml_add_child(GC, trim(self%OH%instances(i)%name), OH_setServices, RC)
         if (RC /= ML_SUCCESS) return
      end do
      !  "Allow children of Chemistry to connect to these fields"
      if (size(self%OH%instances) > 0) call ml_add_child_export(GC, 'OH', self%OH%instances(1)%id, RC)
   end subroutine SetServices

   subroutine Initialize(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      if (ml_am_i_root()) print *, trim(GC%name)//'::Initialize: Starting...'
      call ml_generic_initialize(GC, import, export, clock, RC)      ! initialises the children
   end subroutine Initialize

   subroutine Run1(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      integer :: i
      RC = ML_SUCCESS
      do i = 1, GC%nchildren
         call ml_gridcomp_run(GC%children(i)%gc, clock, 1, RC)
         if (RC /= ML_SUCCESS) return
      end do
   end subroutine Run1

   subroutine Run2(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      integer :: i
      RC = ML_SUCCESS
      do i = 1, GC%nchildren
         if (index(GC%children(i)%gc%name, 'data') == 0) then      ! only a computational instance has a phase 2
            call ml_gridcomp_run(GC%children(i)%gc, clock, 2, RC)
            if (RC /= ML_SUCCESS) return
         end if
      end do
   end subroutine Run2

   subroutine getInstances_(species_name, myCF, species, rc)
      character(len=*), intent(in) :: species_name
      type(ml_config), intent(inout) :: myCF
      type(Constituent), intent(inout) :: species
      integer, intent(out) :: rc
      integer :: i, n_active, n_passive
      n_active = myCF%get_len('ACTIVE_INSTANCES_'//trim(species_name)//':', rc)
      if (rc /= ML_SUCCESS) return
      n_passive = myCF%get_len('PASSIVE_INSTANCES_'//trim(species_name)//':', rc)
      if (rc /= ML_SUCCESS) return
      allocate(species%instances(n_active + n_passive))
      call myCF%find_label('ACTIVE_INSTANCES_'//trim(species_name)//':', rc)
      do i = 1, n_active
         call myCF%next_token(species%instances(i)%name, rc)
         species%instances(i)%is_active = .true.
      end do
      species%n_active = n_active
      call myCF%find_label('PASSIVE_INSTANCES_'//trim(species_name)//':', rc)
      do i = n_active + 1, n_active + n_passive
         call myCF%next_token(species%instances(i)%name, rc)
         species%instances(i)%is_active = .false.
      end do
      rc = ML_SUCCESS
   end subroutine getInstances_

   subroutine IS_QC_INSTANCE_RUNNING(species_name, instance_name, running, RC, rc_dir)
      character(len=*), intent(in) :: species_name, instance_name
      logical, intent(out) :: running
      integer, intent(out) :: RC
      character(len=*), intent(in), optional :: rc_dir
      type(ml_config) :: myCF
      type(Constituent) :: species
      integer :: i
      running = .false.
      if (present(rc_dir)) then
         call myCF%load(trim(rc_dir)//'/'//QUICKCHEM_RESOURCE_FILE, RC)
      else
         call myCF%load(QUICKCHEM_RESOURCE_FILE, RC)
      end if
      if (RC /= ML_SUCCESS) return
      call getInstances_(species_name, myCF, species, RC)
      if (RC /= ML_SUCCESS) return
      do i = 1, size(species%instances)
         if (trim(species%instances(i)%name) == trim(instance_name)) running = .true.
      end do
   end subroutine IS_QC_INSTANCE_RUNNING

end module QuickChem_GridCompMod
