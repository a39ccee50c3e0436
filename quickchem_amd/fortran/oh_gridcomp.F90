!  OH_GridCompMod -- QuickChem's OH grid component with its prediction on the MI355X.
!
!  The surface is the reference's (OH_GridComp/OH_GridCompMod.F90): one public SetServices that reads
!  OH_instance_<NAME>.rc and registers Initialize, Run (phase 1) and - for a computational instance - Run2
!  (:475-802); Initialize sizes self%OH_ML and keeps km and the chemistry time step (:810-949); Run dispatches
!  to Run_data or Run1 (:957-1010); Run1 is gated by the run alarm and by compute_once_per_day, picks every
!  "ONLINE" input according to OH_data_source and the spin-up state, and ends in the INTERNAL field OH in
!  molec/cm3 (:1017-1741); Run2 silences the alarm (:1749-1824); Run_data copies oh_OH (:1831-1891).
!  State names are those of OH_StateSpecs.rc (imports :18-33, exports :41-75, internal :84) and of the
!  conditional import block (:693-783).
!
!  What is different is where the arithmetic happens.  Everything between the imports and INTERNAL OH - the
!  engineered features, the k-slab, the gather, the tree walk, 10**, OHscale, the tropopause mask and the
!  unit conversion (:1247-1257, 1444-1488, 1557-1595) - is ONE call into libohxgb.so (oh_run1_boost ->
!  OHXBoosterRun1, HIP kernels); on a tick that skips Boost the mask and the conversion alone are one call
!  (oh_post_process -> OHXOHPostProcess).  This module moves pointers and keeps the reference's decisions.
!
!  It is written against mapl_lite (mapl_lite.F90), a mock of the few MAPL/ESMF services used here; against
!  real MAPL the same statements read ESMF_GridComp / MAPL_GetPointer / ESMF_Config (INTEGRATION.md).
!
!  Deviations from the reference, on purpose:
!    * self%OH_ML is zero-filled when it is allocated.  The reference leaves it uninitialised (:893): a run
!      that starts at nhms > 0 with compute_once_per_day skips Boost and reads whatever was there (:1582).
!    * "XGBoost_model_policy:" in OH_instance_OH.rc (optional, default "reference"): "by_name" keeps one
!      resident booster per expanded XGBoostFile name, so the month in the %m2 template takes effect; the
!      reference loads the first file once and ignores the name from then on (:209,269).
module OH_GridCompMod
   use mapl_lite
   use oh_xgb_predict, only: oh_xgb_fill_template, oh_xgb_set_model_policy, oh_xgb_error_text, &
                             OH_XGB_SUCCESS, OH_XGB_POLICY_REFERENCE, OH_XGB_POLICY_BY_NAME
   use oh_run1
   implicit none
   private

   public :: SetServices
   public :: OH_GridComp, oh_gridcomp_state, oh_last_run          ! for drivers and tests: what the last tick did

   !  QC_Environment (QC_Environment/QC_EnvironmentMod.F90:13-25) is four numbers: kept inline
   integer, parameter :: instanceComputational = 1, instanceData = 2

   integer, parameter :: DS_UNDEFINED = 0, PRECOMPUTED = 1, ONLINE_INST = 2, ONLINE_AVG24 = 3

   type OH_GridComp
      integer :: nbins = 1, km = 0, instance = 0, klid = 1
      real    :: CDT = 0.0
      character(len=ML_MAXPATH) :: XGBoostFilePattern = ''
      integer :: OH_data_source = DS_UNDEFINED
      logical :: spinup_24hr_imports = .false.
      logical :: use_inst_values = .false.
      logical :: compute_once_per_day = .false.
      real    :: OHscale = 1.0
      integer :: n_wavelengths_profile = 0
      real    :: wavelength_for_scacoef = 0.0
      integer :: wavelength_index = 0
      real, allocatable :: OH_ML(:,:,:)
      !  bookkeeping of the last Run1, for the drivers' logs (not in the reference)
      logical :: last_ran = .false., last_called_boost = .false.
      character(len=ML_MAXPATH) :: last_model_file = ''
      integer :: last_k1 = 0, last_k2 = 0
   end type OH_GridComp

   character(len=*), parameter :: private_key = 'OH_GridComp'
   character(len=512), save :: last_error = ''

   !  OH_StateSpecs.rc, category IMPORT (:18-33): what ACG would generate into OH_Import___.h
   character(len=*), parameter :: import_2d(3) = [character(len=10) :: 'oh_GMITO3', 'oh_GMITTO3', 'oh_ALBUV']
   character(len=*), parameter :: import_3d(12) = [character(len=8) :: 'oh_NO2', 'oh_O3', 'oh_ISOP', 'oh_ACET', &
      'oh_C2H6', 'oh_C3H8', 'oh_PRPE', 'oh_ALK4', 'oh_MP', 'oh_H2O2', 'oh_CH2O', 'oh_OH']
   !  category EXPORT (:41-75)
   character(len=*), parameter :: export_2d(4) = [character(len=15) :: 'DIAG_SZA', 'DIAG_LAT', 'DIAG_GMISTRATO3', &
      'DIAG_ALBUV']
   character(len=*), parameter :: export_3d(28) = [character(len=13) :: 'DIAG_T_avg24', 'DIAG_TAUCLWDN', &
      'DIAG_TAUCLIDN', 'DIAG_TAUCLIUP', 'DIAG_TAUCLWUP', 'DIAG_AODUP', 'DIAG_AODDN', 'DIAG_T_in_OH', 'DIAG_PL', &
      'DIAG_T', 'DIAG_CO', 'DIAG_CH4', 'DIAG_CLOUD', 'DIAG_QV', 'DIAG_OH_M2G', 'DIAG_SC_BC', 'DIAG_SC_OC', &
      'DIAG_SC_BR', 'DIAG_SC_DU', 'DIAG_SC_SU', 'DIAG_SC_SS', 'DIAG_SC_NI', 'DIAG_AOD', 'DIAG_NDWET', 'DIAG_C2H6', &
      'DIAG_ISOP', 'OH_boost', '']
   character(len=*), parameter :: aerosol(7) = [character(len=2) :: 'BC', 'OC', 'BR', 'DU', 'SU', 'SS', 'NI']

contains

   logical function data_driven_name(COMP_NAME)          ! Shared/QuickChem_Generic.F90:50-77
      character(len=*), intent(in) :: COMP_NAME
      data_driven_name = index(COMP_NAME, 'data') > 0
   end function

   function oh_gridcomp_state(gc) result(self)
      type(ml_gridcomp), intent(in) :: gc
      type(OH_GridComp), pointer :: self
      self => null()
      if (.not. associated(gc%private_state)) return
      select type (p => gc%private_state)
      type is (OH_GridComp)
         self => p
      end select
   end function

   subroutine oh_last_run(gc, ran, called_boost, model_file, k1, k2)
      type(ml_gridcomp), intent(in) :: gc
      logical, intent(out) :: ran, called_boost
      character(len=*), intent(out) :: model_file
      integer, intent(out) :: k1, k2
      type(OH_GridComp), pointer :: self
      self => oh_gridcomp_state(gc)
      ran = self%last_ran
      called_boost = self%last_called_boost
      model_file = self%last_model_file
      k1 = self%last_k1
      k2 = self%last_k2
   end subroutine

   subroutine complain(where, what)
      character(len=*), intent(in) :: where, what
      last_error = trim(where)//': '//trim(what)
      if (ml_am_i_root()) print '(a)', 'OH_GridCompMod::'//trim(last_error)
   end subroutine

   !  OH_instance_<NAME>.rc, else OH_instance_OH.rc (:532-538, 884-890)
   subroutine load_instance_config(gc, cfg, rc)
      type(ml_gridcomp), intent(in) :: gc
      type(ml_config), intent(inout) :: cfg
      integer, intent(out) :: rc
      call cfg%load(trim(gc%rc_dir)//'/OH_instance_'//trim(gc%name)//'.rc', rc)
      if (rc /= ML_SUCCESS) then
         if (ml_am_i_root()) print *, 'OH_instance_'//trim(gc%name)//'.rc does not exist! Loading OH_instance_OH.rc instead'
         call cfg%load(trim(gc%rc_dir)//'/OH_instance_OH.rc', rc)
      end if
   end subroutine

   ! =================================================================== SetServices

   subroutine SetServices(GC, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      integer, intent(out) :: RC
      type(OH_GridComp), pointer :: self
      type(ml_config) :: cfg, g2g_cfg
      character(len=ML_MAXSTR) :: token
      real, allocatable :: wavelengths_profile(:)
      logical :: data_driven
      integer :: i, n4

      allocate(self)
      call load_instance_config(GC, cfg, RC)
      if (RC /= ML_SUCCESS) then
         call complain('SetServices', 'no OH_instance_'//trim(GC%name)//'.rc and no OH_instance_OH.rc in '//trim(GC%rc_dir))
         return
      end if
      call cfg%get_int(self%nbins, 'nbins:', RC)                       ! QC_Environment%load_from_config
      if (RC /= ML_SUCCESS) return
      call cfg%get_string(self%XGBoostFilePattern, 'XGBoostFile:', RC)
      if (RC /= ML_SUCCESS) then
         call complain('SetServices', 'XGBoostFile: is missing')
         return
      end if
      call cfg%get_string(token, 'OH_data_source:', RC)
      if (RC /= ML_SUCCESS) return
      self%OH_data_source = DS_UNDEFINED
      if (trim(token) == 'PRECOMPUTED')  self%OH_data_source = PRECOMPUTED
      if (trim(token) == 'ONLINE_INST')  self%OH_data_source = ONLINE_INST
      if (trim(token) == 'ONLINE_AVG24') self%OH_data_source = ONLINE_AVG24
      if (self%OH_data_source == DS_UNDEFINED) then
         call complain('SetServices', 'Invalid OH_data_source: '//trim(token))
         RC = 99                                                        ! VERIFY_(99), :556
         return
      end if
      if (ml_am_i_root()) print *, 'OH_data source for ONLINE fields: '//trim(token)
      call cfg%get_logical(self%compute_once_per_day, 'compute_once_per_day:', RC)
      if (RC /= ML_SUCCESS) return
      call cfg%get_logical(self%spinup_24hr_imports, 'spinup_24hr_imports:', RC)
      if (RC /= ML_SUCCESS) return
      call cfg%get_real(self%wavelength_for_scacoef, 'wavelength_for_scacoef:', RC)
      if (RC /= ML_SUCCESS) return
      call cfg%get_real(self%OHscale, 'OHscale:', RC)
      if (RC /= ML_SUCCESS) return
      call cfg%get_string(token, 'XGBoost_model_policy:', RC, default='reference')
      if (trim(token) == 'by_name') then
         call oh_xgb_set_model_policy(OH_XGB_POLICY_BY_NAME)
      else if (trim(token) == 'reference') then
         call oh_xgb_set_model_policy(OH_XGB_POLICY_REFERENCE)
      else
         call complain('SetServices', 'Invalid XGBoost_model_policy: '//trim(token))
         RC = 99
         return
      end if

      !  the wavelength of the GOCART2G scattering coefficients (:572-592)
      call g2g_cfg%load(trim(GC%rc_dir)//'/GOCART2G_GridComp.rc', RC)
      if (RC /= ML_SUCCESS) then
         call complain('SetServices', 'GOCART2G_GridComp.rc not found')
         return
      end if
      self%n_wavelengths_profile = g2g_cfg%get_len('wavelengths_for_profile_aop_in_nm:', RC)
      if (RC /= ML_SUCCESS) return
      allocate(wavelengths_profile(self%n_wavelengths_profile))
      call g2g_cfg%get_reals(wavelengths_profile, 'wavelengths_for_profile_aop_in_nm:', RC)
      if (RC /= ML_SUCCESS) return
      self%wavelength_index = 0
      do i = 1, self%n_wavelengths_profile
         if (wavelengths_profile(i) == self%wavelength_for_scacoef) self%wavelength_index = i
      end do
      if (self%wavelength_index <= 0) then
         call complain('SetServices', 'Did not find OH wavelength_for_scacoef in GOCART2G wavelengths_for_profile_aop_in_nm')
         RC = ML_FAILURE
         return
      end if

      data_driven = data_driven_name(GC%name)
      call ml_set_entry_point(GC, ML_METHOD_INITIALIZE, Initialize, RC)
      call ml_set_entry_point(GC, ML_METHOD_RUN, Run, RC)
      if (.not. data_driven) call ml_set_entry_point(GC, ML_METHOD_RUN, Run2, RC)

      if (data_driven) then                                             ! :611-634
         call GC%internal%add_spec('OH', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='???', long_name='Hydroxyl Radical')
         call GC%import%add_spec('oh_OH', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
      else
         call GC%internal%add_spec('OH', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='molec/cm3', long_name='Hydroxyl Radical')
         do i = 1, size(import_2d)
            call GC%import%add_spec(trim(import_2d(i)), ML_DIMS_HORZ_ONLY, ML_VLOC_NONE, units='???')
         end do
         do i = 1, size(import_3d)
            call GC%import%add_spec(trim(import_3d(i)), ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='???')
         end do
         !  always imported: units conversion of the predicted OH, and the tropopause (:693-698)
         call GC%import%add_spec('TROPP', ML_DIMS_HORZ_ONLY, ML_VLOC_NONE, units='Pa', restart_skip=.true.)
         call GC%import%add_spec('T', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='K', restart_skip=.true.)
         call GC%import%add_spec('Q', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='kg kg-1', restart_skip=.true.)
         call GC%import%add_spec('PLE', ML_DIMS_HORZ_VERT, ML_VLOC_EDGE, units='Pa', restart_skip=.true.)
         n4 = self%n_wavelengths_profile
         if (self%OH_data_source == ONLINE_INST .or. &
             (self%OH_data_source == ONLINE_AVG24 .and. self%spinup_24hr_imports)) then       ! IMPORT_INST, :707-733
            call GC%import%add_spec('CH4', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('CO', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('FCLD', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('ZLE', ML_DIMS_HORZ_VERT, ML_VLOC_EDGE, units='m', restart_skip=.true.)
            do i = 1, size(aerosol)
               call GC%import%add_spec(aerosol(i)//'SCACOEF', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='m-1', &
                                       ungridded=n4, restart_skip=.true.)
            end do
            call GC%import%add_spec('TAUCLW', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='1')
            call GC%import%add_spec('TAUCLI', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, units='1')
         end if
         if (self%OH_data_source == ONLINE_AVG24) then                                         ! IMPORT_24, :736-758
            call add24('CH4_avg24', ML_VLOC_CENTER, 0)
            call add24('CO_avg24', ML_VLOC_CENTER, 0)
            call add24('T_avg24', ML_VLOC_CENTER, 0)
            call add24('FCLD_avg24', ML_VLOC_CENTER, 0)
            call add24('Q_avg24', ML_VLOC_CENTER, 0)
            call add24('TAUCLW_avg24', ML_VLOC_CENTER, 0)
            call add24('TAUCLI_avg24', ML_VLOC_CENTER, 0)
            call add24('PLE_avg24', ML_VLOC_EDGE, 0)
            call add24('ZLE_avg24', ML_VLOC_EDGE, 0)
            do i = 1, size(aerosol)
               call add24(aerosol(i)//'SCACOEF_avg24', ML_VLOC_CENTER, n4)
            end do
         end if
         if (self%OH_data_source == PRECOMPUTED) then                                          ! IMPORT_PRECOMPUTED, :761-783
            call GC%import%add_spec('oh_CH4', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_CO', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_T', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_FCLD', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_Q', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_TAUCLW', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_TAUCLI', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            call GC%import%add_spec('oh_PLE', ML_DIMS_HORZ_VERT, ML_VLOC_EDGE, restart_skip=.true.)
            call GC%import%add_spec('oh_ZLE', ML_DIMS_HORZ_VERT, ML_VLOC_EDGE, restart_skip=.true.)
            do i = 1, size(aerosol)                      ! the archived scattering coefficients are 3-D
               call GC%import%add_spec('oh_'//aerosol(i)//'SCACOEF', ML_DIMS_HORZ_VERT, ML_VLOC_CENTER, restart_skip=.true.)
            end do
         end if
         do i = 1, size(export_2d)
            call GC%export%add_spec(trim(export_2d(i)), ML_DIMS_HORZ_ONLY, ML_VLOC_NONE, units='???')
         end do
         do i = 1, size(export_3d)
            if (len_trim(export_3d(i)) > 0) call GC%export%add_spec(trim(export_3d(i)), ML_DIMS_HORZ_VERT, ML_VLOC_CENTER)
         end do
         call GC%export%add_spec('DIAG_ZLE', ML_DIMS_HORZ_VERT, ML_VLOC_EDGE, units='???')
      end if

      GC%private_state => self                                          ! ESMF_UserCompSetInternalState, :793
      GC%private_key = private_key
      RC = ML_SUCCESS

   contains
      subroutine add24(name, vloc, ungridded)                           ! saved in the import restart, daily mean
         character(len=*), intent(in) :: name
         integer, intent(in) :: vloc, ungridded
         call GC%import%add_spec(name, ML_DIMS_HORZ_VERT, vloc, ungridded=ungridded, averaging_interval=86400)
      end subroutine
   end subroutine SetServices

   ! =================================================================== Initialize

   subroutine Initialize(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      type(OH_GridComp), pointer :: self
      integer :: HDT, trc
      real :: CDT

      self => oh_gridcomp_state(GC)
      if (.not. associated(self)) then
         call complain('Initialize', 'no private state under key '//private_key)
         RC = ML_FAILURE
         return
      end if
      self%km = GC%grid%km
      HDT = clock%dt
      CDT = real(HDT)
      if (associated(GC%config)) then
         call GC%config%get_int(HDT, 'RUN_DT:', trc, default=clock%dt)
         call GC%config%get_real(CDT, 'QUICKCHEM_DT:', trc, default=real(HDT))
      end if
      self%CDT = CDT
      if (allocated(self%OH_ML)) deallocate(self%OH_ML)
      allocate(self%OH_ML(GC%grid%im, GC%grid%jm, GC%grid%km))
      self%OH_ML = 0.0                    ! the reference does not (:893); see the header
      call ml_generic_initialize(GC, import, export, clock, RC)
      if (RC /= ML_SUCCESS) return
      self%instance = merge(instanceData, instanceComputational, data_driven_name(GC%name))
   end subroutine Initialize

   ! =================================================================== Run

   subroutine Run(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      if (data_driven_name(GC%name)) then
         call Run_data(GC, import, export, GC%internal, RC)
      else
         call Run1(GC, import, export, clock, RC)
      end if
   end subroutine Run

   !  the import a "3 options" input comes from (:1326-1436)
   function source_name(self, base) result(name)
      type(OH_GridComp), intent(in) :: self
      character(len=*), intent(in) :: base
      character(len=:), allocatable :: name
      select case (self%OH_data_source)
      case (PRECOMPUTED)
         name = 'oh_'//base
      case (ONLINE_INST)
         name = base
      case default
         name = base
         if (.not. self%use_inst_values) name = base//'_avg24'
      end select
   end function

   subroutine Run1(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC

      type(OH_GridComp), pointer :: self
      type(OH_RUN1_STATE) :: st
      integer :: nymd, nhms, iyr, imm, idd, ihr, imn, isc, im, jm, km, a, JDAY, k1, k2
      logical :: need_to_call_BOOST, dynamic_k_range
      real :: tropp_min
      character(len=ML_MAXPATH) :: XGBoostFilename
      real, pointer :: ptr2d(:,:), ptr3d(:,:,:), ptr4d(:,:,:,:), OH(:,:,:), default_OH(:,:,:)
      real, pointer :: sca3(:,:,:)
      real, allocatable, target :: latarr(:,:), sza_noon(:,:), NDWET_MOD(:,:,:), OH_boost(:,:,:)
      real, allocatable, target :: sca_slice(:,:,:,:)
      type(OH_RUN1_DIAG) :: dg
      real, allocatable, target :: d_pl(:,:,:), d_wdn(:,:,:), d_idn(:,:,:), d_iup(:,:,:), d_wup(:,:,:), d_aup(:,:,:), &
                                   d_adn(:,:,:), d_aod(:,:,:), d_so3(:,:)

      self => oh_gridcomp_state(GC)
      self%last_ran = .false.
      self%last_called_boost = .false.
      call clock%get(iyr, imm, idd, ihr, imn, isc)
      call ml_pack_time(nymd, iyr, imm, idd)
      call ml_pack_time(nhms, ihr, imn, isc)

      RC = ML_SUCCESS
      if (.not. GC%runalarm%is_ringing()) return                        ! :1180-1185
      self%last_ran = .true.

      XGBoostFilename = oh_xgb_fill_template(trim(self%XGBoostFilePattern), nymd, nhms)   ! :1187
      need_to_call_BOOST = .not. (self%compute_once_per_day .and. nhms > 0)               ! :1189-1193

      call import%get_pointer(ptr3d, 'oh_ISOP', RC)                     ! a 3-D field that is always imported (:1199)
      if (RC /= ML_SUCCESS .or. .not. associated(ptr3d)) then
         call complain('Run1', 'import oh_ISOP has no storage')
         RC = ML_FAILURE
         return
      end if
      im = size(ptr3d, 1); jm = size(ptr3d, 2); km = size(ptr3d, 3)

      !  the model's own state: number density and tropopause (:1240-1257)
      call need3(import, st%T_MOD, 'T');      if (RC /= ML_SUCCESS) return
      call need3(import, st%Q_MOD, 'Q');      if (RC /= ML_SUCCESS) return
      call need3(import, st%PLE_MOD, 'PLE');  if (RC /= ML_SUCCESS) return
      call need2(import, st%TROPP_MOD, 'TROPP'); if (RC /= ML_SUCCESS) return
      if (lbound(st%PLE_MOD, 3) /= 0) then
         call complain('Run1', 'Error. Expecting PLE starting index 0')
         RC = ML_FAILURE
         return
      end if
      call need3(import, default_OH, 'oh_OH'); if (RC /= ML_SUCCESS) return               ! :1548
      st%default_OH => default_OH
      call ml_maxmin('OH: OH From M2G ', default_OH)
      call export%get_pointer(ptr3d, 'DIAG_OH_M2G', RC)
      if (associated(ptr3d)) ptr3d(:,:,:) = default_OH(:,:,:)
      call GC%internal%get_pointer(OH, 'OH', RC)
      if (RC /= ML_SUCCESS .or. .not. associated(OH)) then
         call complain('Run1', 'INTERNAL OH has no storage')
         RC = ML_FAILURE
         return
      end if
      allocate(NDWET_MOD(im, jm, km))

      if (.not. need_to_call_BOOST) then
         !  Boost is not due: the mask and the conversion on the OH_ML of the last Boost (:1579-1595)
         call oh_post_process(im, jm, km, MAPL_AVOGAD, MAPL_RUNIV, MAPL_EPSILON, st%PLE_MOD, st%T_MOD, st%Q_MOD, &
                              st%TROPP_MOD, default_OH, self%OH_ML, OH, NDWET_MOD, RC)
         if (RC /= OH_XGB_SUCCESS) then
            call complain('Run1', oh_run1_error_text())
            return
         end if
         call export%get_pointer(ptr3d, 'DIAG_NDWET', RC)
         if (associated(ptr3d)) ptr3d(:,:,:) = NDWET_MOD(:,:,:)
         RC = ML_SUCCESS
         return
      end if

      ! ---- PREP_FOR_BOOST (:1259-1544): which import feeds which input
      self%use_inst_values = .false.
      if (self%OH_data_source == ONLINE_AVG24) then                     ! :1309-1317
         call need3(import, ptr3d, 'T_avg24'); if (RC /= ML_SUCCESS) return
         if (ptr3d(1,1,1) == 0.0) self%use_inst_values = .true.
      end if
      if (ml_am_i_root()) then
         if (self%use_inst_values) print *, 'OH is in the SPINUP period for 24-hour averages'
         if (.not. self%use_inst_values) print *, 'OH is *NOT* in the SPINUP period for 24-hour averages'
      end if
      call need3(import, st%T_BST, source_name(self, 'T'));       if (RC /= ML_SUCCESS) return
      call need3(import, st%QV, source_name(self, 'Q'));          if (RC /= ML_SUCCESS) return
      call need3(import, st%PLE_BST, source_name(self, 'PLE'));   if (RC /= ML_SUCCESS) return
      call need3(import, st%ZLE_BST, source_name(self, 'ZLE'));   if (RC /= ML_SUCCESS) return
      call need3(import, st%TAUCLW, source_name(self, 'TAUCLW')); if (RC /= ML_SUCCESS) return
      call need3(import, st%TAUCLI, source_name(self, 'TAUCLI')); if (RC /= ML_SUCCESS) return
      if (lbound(st%ZLE_BST, 3) /= 0) then
         call complain('Run1', 'Error. Expecting ZLE starting index 0')
         RC = ML_FAILURE
         return
      end if
      !  scattering coefficients: archived ones are 3-D, online ones carry the wavelength as 4th dimension (:1389-1436)
      if (self%OH_data_source /= PRECOMPUTED) allocate(sca_slice(im, jm, km, size(aerosol)))
      do a = 1, size(aerosol)
         if (self%OH_data_source == PRECOMPUTED) then
            call need3(import, sca3, 'oh_'//aerosol(a)//'SCACOEF'); if (RC /= ML_SUCCESS) return
         else
            call import%get_pointer(ptr4d, source_name(self, aerosol(a)//'SCACOEF'), RC)
            if (RC /= ML_SUCCESS .or. .not. associated(ptr4d)) then
               call complain('Run1', 'import '//source_name(self, aerosol(a)//'SCACOEF')//' has no storage')
               RC = ML_FAILURE
               return
            end if
            sca_slice(:,:,:,a) = ptr4d(:,:,:,self%wavelength_index)
            sca3 => sca_slice(:,:,:,a)
         end if
         select case (a)
         case (1); st%BCscacoef => sca3
         case (2); st%OCscacoef => sca3
         case (3); st%BRscacoef => sca3
         case (4); st%DUscacoef => sca3
         case (5); st%SUscacoef => sca3
         case (6); st%SSscacoef => sca3
         case (7); st%NIscacoef => sca3
         end select
      end do
      call need2(import, st%GMITO3, 'oh_GMITO3');   if (RC /= ML_SUCCESS) return
      call need2(import, st%GMITTO3, 'oh_GMITTO3'); if (RC /= ML_SUCCESS) return
      call need3(import, st%NO2, 'oh_NO2');   if (RC /= ML_SUCCESS) return
      call need3(import, st%O3, 'oh_O3');     if (RC /= ML_SUCCESS) return
      call need3(import, st%CH4, source_name(self, 'CH4')); if (RC /= ML_SUCCESS) return
      call need3(import, st%CO, source_name(self, 'CO'));   if (RC /= ML_SUCCESS) return
      call need3(import, st%ISOP, 'oh_ISOP'); if (RC /= ML_SUCCESS) return
      call need3(import, st%ACET, 'oh_ACET'); if (RC /= ML_SUCCESS) return
      call need3(import, st%C2H6, 'oh_C2H6'); if (RC /= ML_SUCCESS) return
      call need3(import, st%C3H8, 'oh_C3H8'); if (RC /= ML_SUCCESS) return
      call need3(import, st%PRPE, 'oh_PRPE'); if (RC /= ML_SUCCESS) return
      call need3(import, st%ALK4, 'oh_ALK4'); if (RC /= ML_SUCCESS) return
      call need3(import, st%MP, 'oh_MP');     if (RC /= ML_SUCCESS) return
      call need3(import, st%H2O2, 'oh_H2O2'); if (RC /= ML_SUCCESS) return
      call need3(import, st%CLOUD, source_name(self, 'FCLD')); if (RC /= ML_SUCCESS) return
      call need2(import, st%ALBUV, 'oh_ALBUV'); if (RC /= ML_SUCCESS) return
      call need3(import, st%CH2O, 'oh_CH2O'); if (RC /= ML_SUCCESS) return

      !  latitudes in degrees and the local-noon zenith angle (:1444, 1481-1482)
      allocate(latarr(im, jm), sza_noon(im, jm))
      JDAY = oh_julian_day(nymd)
      call oh_solar_geometry(JDAY, GC%grid%LATS, GC%grid%LONS, MAPL_DEGREES_TO_RADIANS, MAPL_RADIANS_TO_DEGREES, &
                             latarr, sza_noon, RC)
      if (RC /= OH_XGB_SUCCESS) then
         call complain('Run1', oh_run1_error_text())
         return
      end if
      st%latarr => latarr
      st%sza_noon => sza_noon

      !  the engineered features are dumped only into DIAG exports somebody asked for (:1607-1640)
      call want3('DIAG_PL', d_pl, dg%PL_BST);            call want3('DIAG_TAUCLWDN', d_wdn, dg%tauclwDN)
      call want3('DIAG_TAUCLIDN', d_idn, dg%taucliDN);   call want3('DIAG_TAUCLIUP', d_iup, dg%taucliUP)
      call want3('DIAG_TAUCLWUP', d_wup, dg%tauclwUP);   call want3('DIAG_AODUP', d_aup, dg%aodUP)
      call want3('DIAG_AODDN', d_adn, dg%aodDN);         call want3('DIAG_AOD', d_aod, dg%aod)
      if (export%is_allocated('DIAG_GMISTRATO3')) then
         allocate(d_so3(im, jm))
         dg%stratO3 => d_so3
      end if

      ! ---- CALL_BOOST and what follows it (:1557-1595), one device-resident pass
      dynamic_k_range = .not. self%compute_once_per_day                 ! :1561
      tropp_min = 40.0 * 100                                            ! :1563
      allocate(OH_boost(im, jm, km))
      call oh_run1_boost(trim(XGBoostFilename), im, jm, km, dynamic_k_range, tropp_min, self%OHscale, &
                         MAPL_AVOGAD, MAPL_RUNIV, MAPL_EPSILON, st, OH, OH_boost, NDWET_MOD, k1, k2, RC, diag=dg)
      if (RC /= OH_XGB_SUCCESS) then
         call complain('Run1', oh_run1_error_text()//' '//oh_xgb_error_text())
         return
      end if
      self%OH_ML(:,:,:) = OH_boost(:,:,:)                                ! persists to the ticks that skip Boost
      self%last_called_boost = .true.
      self%last_model_file = XGBoostFilename
      self%last_k1 = k1
      self%last_k2 = k2
      call export%get_pointer(ptr3d, 'OH_boost', RC)
      if (associated(ptr3d)) ptr3d(:,:,:) = self%OH_ML(:,:,:)
      call export%get_pointer(ptr3d, 'DIAG_NDWET', RC)
      if (associated(ptr3d)) ptr3d(:,:,:) = NDWET_MOD(:,:,:)

      ! ---- AFTER_BOOST: diagnostics that are only meaningful when Boost was called (:1601-1728)
      call put2('DIAG_LAT', latarr);            call put2('DIAG_SZA', sza_noon)
      call put2('DIAG_ALBUV', st%ALBUV)
      if (associated(dg%stratO3)) call put2('DIAG_GMISTRATO3', dg%stratO3)
      if (associated(dg%PL_BST)) call put3('DIAG_PL', dg%PL_BST)
      if (associated(dg%tauclwDN)) call put3('DIAG_TAUCLWDN', dg%tauclwDN)
      if (associated(dg%taucliDN)) call put3('DIAG_TAUCLIDN', dg%taucliDN)
      if (associated(dg%taucliUP)) call put3('DIAG_TAUCLIUP', dg%taucliUP)
      if (associated(dg%tauclwUP)) call put3('DIAG_TAUCLWUP', dg%tauclwUP)
      if (associated(dg%aodUP)) call put3('DIAG_AODUP', dg%aodUP)
      if (associated(dg%aodDN)) call put3('DIAG_AODDN', dg%aodDN)
      if (associated(dg%aod)) call put3('DIAG_AOD', dg%aod)
      call put3('DIAG_T', st%T_BST);     call put3('DIAG_CH4', st%CH4);   call put3('DIAG_CO', st%CO)
      call put3('DIAG_CLOUD', st%CLOUD); call put3('DIAG_QV', st%QV)
      call put3('DIAG_C2H6', st%C2H6);   call put3('DIAG_ISOP', st%ISOP)
      call put3('DIAG_SC_BC', st%BCscacoef); call put3('DIAG_SC_OC', st%OCscacoef); call put3('DIAG_SC_BR', st%BRscacoef)
      call put3('DIAG_SC_DU', st%DUscacoef); call put3('DIAG_SC_SU', st%SUscacoef); call put3('DIAG_SC_SS', st%SSscacoef)
      call put3('DIAG_SC_NI', st%NIscacoef)
      call export%get_pointer(ptr3d, 'DIAG_ZLE', RC)
      if (associated(ptr3d)) ptr3d(:,:,:) = st%ZLE_BST(:,:,:)
      RC = ML_SUCCESS

   contains

      subroutine need3(state, p, name)
         type(ml_state), intent(in) :: state
         real, pointer, intent(out) :: p(:,:,:)
         character(len=*), intent(in) :: name
         call state%get_pointer(p, name, RC)
         if (RC == ML_SUCCESS .and. .not. associated(p)) RC = ML_FAILURE
         if (RC /= ML_SUCCESS) call complain('Run1', 'import '//name//' is not in the state or has no storage')
      end subroutine

      subroutine need2(state, p, name)
         type(ml_state), intent(in) :: state
         real, pointer, intent(out) :: p(:,:)
         character(len=*), intent(in) :: name
         call state%get_pointer(p, name, RC)
         if (RC == ML_SUCCESS .and. .not. associated(p)) RC = ML_FAILURE
         if (RC /= ML_SUCCESS) call complain('Run1', 'import '//name//' is not in the state or has no storage')
      end subroutine

      subroutine want3(name, buf, slot)
         character(len=*), intent(in) :: name
         real, allocatable, target, intent(inout) :: buf(:,:,:)
         real, pointer, intent(out) :: slot(:,:,:)
         slot => null()
         if (.not. export%is_allocated(name)) return
         allocate(buf(im, jm, km))
         slot => buf
      end subroutine

      subroutine put3(name, src)
         character(len=*), intent(in) :: name
         real, intent(in) :: src(:,:,:)
         real, pointer :: p(:,:,:)
         integer :: prc
         call export%get_pointer(p, name, prc)
         if (associated(p)) p(:,:,:) = src(:,:,:)
      end subroutine

      subroutine put2(name, src)
         character(len=*), intent(in) :: name
         real, intent(in) :: src(:,:)
         real, pointer :: p(:,:)
         integer :: prc
         call export%get_pointer(p, name, prc)
         if (associated(p)) p(:,:) = src(:,:)
      end subroutine
   end subroutine Run1

   ! =================================================================== Run2

   subroutine Run2(GC, import, export, clock, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: RC
      RC = ML_SUCCESS
      if (.not. GC%runalarm%is_ringing()) return
      call GC%runalarm%ringer_off()                                     ! :1820
   end subroutine Run2

   ! =================================================================== Run_data

   subroutine Run_data(GC, import, export, internal, RC)
      type(ml_gridcomp), intent(inout), target :: GC
      type(ml_state), intent(inout) :: import, export, internal
      integer, intent(out) :: RC
      type(OH_GridComp), pointer :: self
      real, pointer :: ptr3d_intern(:,:,:), ptr3d_import(:,:,:)
      self => oh_gridcomp_state(GC)
      call internal%get_pointer(ptr3d_intern, 'OH', RC)
      if (RC /= ML_SUCCESS) return
      if (self%nbins /= 1) then
         print *, 'expecting only 1 OH bin'
         RC = 123                                                        ! :1871-1874
         return
      end if
      call import%get_pointer(ptr3d_import, 'oh_OH', RC)
      if (RC /= ML_SUCCESS) return
      if (.not. (associated(ptr3d_intern) .and. associated(ptr3d_import))) then
         RC = ML_FAILURE
         return
      end if
      ptr3d_intern(:,:,:) = ptr3d_import
   end subroutine Run_data

end module OH_GridCompMod
