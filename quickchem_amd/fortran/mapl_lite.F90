!  mapl_lite -- the small part of MAPL/ESMF that QuickChem's OH GridComp touches, as a mock.
!
!  MAPL and ESMF cannot be built here (SURVEY.md §8c), and GEOS itself is out of tree.  The OH GridComp shell
!  (oh_gridcomp.F90, quickchem_gridcomp.F90) is written against THIS module so that BASELINE.json config #1 -
!  "a synthetic MAPL state through OH_GridComp Run" - runs without either: named-field states whose pointers
!  come back unassociated for exports nobody asked for, an ESMF_Config-style resource-file reader, a clock
!  with MAPL's run alarm, and grid components with SetServices / Initialize / Run phases and children.
!  It restates MAPL behaviour only as far as the reference relies on it:
!     MAPL_GetPointer on an unallocated export gives a null pointer, not an error
!                                                  (OH_GridComp/OH_GridCompMod.F90:1553,1571-1572,1598-1728)
!     edge fields are indexed 0:km                 (:1246,1450 "_ASSERT(lbound(PLE_MOD,3)==0")
!     the run alarm of a component rings DT apart, backed off by one heartbeat unless
!     <COMP>_REFERENCE_TIME says otherwise, and stays silent once rung off   (:1171-1185,1820; NOTES.wiki:57-59)
!     ESMF_ConfigGetAttribute / GetLen / FindLabel on "label: value value # comment" lines   (:532-589)
!  Everything else MAPL does (ExtData, restarts, HISTORY, couplers, the MPI layout) is not here.
module mapl_lite
   use, intrinsic :: iso_fortran_env, only: int64, real64
   implicit none
   private

   integer, parameter, public :: ML_SUCCESS = 0, ML_FAILURE = 1
   integer, parameter, public :: ML_MAXSTR = 256, ML_MAXPATH = 1024

   !  MAPL's constants, as MAPL defines them (MAPL Constants: PhysicalConstants / MathConstants)
   real(real64), parameter, public :: MAPL_PI_R8 = 3.14159265358979323846d0
   real, parameter, public :: MAPL_PI = MAPL_PI_R8
   real, parameter, public :: MAPL_DEGREES_TO_RADIANS = MAPL_PI / 180.0
   real, parameter, public :: MAPL_RADIANS_TO_DEGREES = 180.0 / MAPL_PI
   real, parameter, public :: MAPL_AVOGAD = 6.023e26      ! molec / kmol
   real, parameter, public :: MAPL_RUNIV  = 8314.47       ! J / (kmol K)
   real, parameter, public :: MAPL_H2OMW  = 18.015, MAPL_AIRMW = 28.965
   real, parameter, public :: MAPL_EPSILON = MAPL_H2OMW / MAPL_AIRMW

   integer, parameter, public :: ML_DIMS_HORZ_ONLY = 2, ML_DIMS_HORZ_VERT = 3
   integer, parameter, public :: ML_VLOC_NONE = 0, ML_VLOC_CENTER = 1, ML_VLOC_EDGE = 2
   integer, parameter, public :: ML_METHOD_INITIALIZE = 1, ML_METHOD_RUN = 2

   ! ------------------------------------------------------------------ resource files
   type, public :: ml_config
      character(len=ML_MAXPATH), allocatable :: line(:)
      integer :: nline = 0
      integer :: cur = 0, pos = 0          ! where FindLabel left the cursor
   contains
      procedure :: load        => config_load
      procedure :: set_text    => config_set_text
      procedure :: find_label  => config_find_label
      procedure :: next_token  => config_next_token
      procedure :: get_len     => config_get_len
      procedure :: get_string  => config_get_string
      procedure :: get_int     => config_get_int
      procedure :: get_real    => config_get_real
      procedure :: get_logical => config_get_logical
      procedure :: get_reals   => config_get_reals
   end type

   ! ------------------------------------------------------------------ states
   type, public :: ml_field
      character(len=ML_MAXSTR) :: name = ''
      character(len=ML_MAXSTR) :: units = '', long_name = ''
      integer :: dims = ML_DIMS_HORZ_VERT
      integer :: vloc = ML_VLOC_CENTER
      integer :: ungridded = 0             ! size of a 4th dimension, 0 = none
      logical :: restart_skip = .false.
      integer :: averaging_interval = 0    ! seconds; 86400 for the *_avg24 imports
      real, pointer :: p2(:,:) => null(), p3(:,:,:) => null(), p4(:,:,:,:) => null()
   end type

   type, public :: ml_state
      character(len=ML_MAXSTR) :: name = ''
      type(ml_field), allocatable :: f(:)
      integer :: n = 0
   contains
      procedure :: add_spec   => state_add_spec
      procedure :: index_of   => state_index_of
      procedure :: has        => state_has
      procedure :: allocate_field => state_allocate_field
      procedure :: is_allocated   => state_is_allocated
      procedure :: state_get_pointer_2d, state_get_pointer_3d, state_get_pointer_4d
      generic :: get_pointer => state_get_pointer_2d, state_get_pointer_3d, state_get_pointer_4d
   end type

   ! ------------------------------------------------------------------ clock and alarm
   type, public :: ml_clock
      integer(int64) :: now = 0            ! seconds since 0001-01-01 00:00:00 (proleptic Gregorian)
      integer :: dt = 450                  ! the heartbeat (RUN_DT)
   contains
      procedure :: set      => clock_set
      procedure :: get      => clock_get
      procedure :: advance  => clock_advance
   end type

   type, public :: ml_alarm
      integer(int64) :: first_ring = 0     ! a ring time; the others are first_ring + k * interval
      integer :: interval = 0              ! seconds; 0 = rings at every step
      logical :: ringing = .true.
   contains
      procedure :: is_ringing => alarm_is_ringing
      procedure :: ringer_off => alarm_ringer_off
      procedure :: update     => alarm_update
   end type

   ! ------------------------------------------------------------------ grid and grid components
   type, public :: ml_grid
      integer :: im = 0, jm = 0, km = 0               ! local = global in the mock (one rank)
      real, pointer :: LATS(:,:) => null(), LONS(:,:) => null()     ! radians
   end type

   type, public :: ml_gridcomp_ptr
      type(ml_gridcomp), pointer :: gc => null()
   end type

   type, public :: ml_gridcomp
      character(len=ML_MAXSTR) :: name = ''
      character(len=ML_MAXPATH) :: rc_dir = '.'          ! where the component looks for its resource files
      type(ml_config), pointer :: config => null()       ! the "universal" config (AGCM.rc)
      type(ml_grid) :: grid
      type(ml_state) :: import, export, internal
      type(ml_alarm) :: runalarm
      class(*), pointer :: private_state => null()       ! ESMF_UserCompSetInternalState
      character(len=ML_MAXSTR) :: private_key = ''
      type(ml_entry_points), pointer :: entry => null()  ! MAPL_GridCompSetEntryPoint
      type(ml_gridcomp_ptr), allocatable :: children(:)
      integer :: nchildren = 0
      !  AddExportSpec(SHORT_NAME=..., CHILD_ID=...): exports of this component that ARE a child's field
      character(len=ML_MAXSTR), allocatable :: child_export_name(:)
      integer, allocatable :: child_export_id(:)
   end type

   abstract interface
      subroutine ml_method(gc, import, export, clock, rc)
         import :: ml_gridcomp, ml_state, ml_clock
         type(ml_gridcomp), intent(inout), target :: gc
         type(ml_state), intent(inout) :: import, export
         type(ml_clock), intent(inout) :: clock
         integer, intent(out) :: rc
      end subroutine
      subroutine ml_set_services(gc, rc)
         import :: ml_gridcomp
         type(ml_gridcomp), intent(inout), target :: gc
         integer, intent(out) :: rc
      end subroutine
   end interface

   type :: ml_method_ptr
      procedure(ml_method), pointer, nopass :: p => null()
   end type

   type :: ml_entry_points
      type(ml_method_ptr) :: initialize
      type(ml_method_ptr) :: run_phase(4)
      integer :: n_run_phases = 0
   end type

   public :: ml_gridcomp_create, ml_add_child, ml_set_entry_point, ml_generic_initialize, ml_gridcomp_initialize
   public :: ml_gridcomp_run, ml_add_child_export, ml_child_export_field, ml_pack_time, ml_am_i_root, ml_maxmin
   public :: ml_method, ml_set_services, ml_time_to_seconds, ml_advance

contains

   ! =================================================================== config

   subroutine config_load(cfg, file, rc)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(in) :: file
      integer, intent(out) :: rc
      integer :: u, ios, n
      character(len=ML_MAXPATH) :: buf
      rc = ML_FAILURE
      open(newunit=u, file=trim(file), status='old', action='read', iostat=ios)
      if (ios /= 0) return
      n = 0
      do
         read(u, '(a)', iostat=ios) buf
         if (ios /= 0) exit
         n = n + 1
      end do
      rewind(u)
      if (allocated(cfg%line)) deallocate(cfg%line)
      allocate(cfg%line(max(n, 1)))
      cfg%nline = n
      do n = 1, cfg%nline
         read(u, '(a)') cfg%line(n)
      end do
      close(u)
      cfg%cur = 0
      cfg%pos = 0
      rc = ML_SUCCESS
   end subroutine

   !  the same from memory, lines separated by new_line('a')
   subroutine config_set_text(cfg, text)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(in) :: text
      integer :: n, a, b
      n = 1
      do a = 1, len(text)
         if (text(a:a) == new_line('a')) n = n + 1
      end do
      if (allocated(cfg%line)) deallocate(cfg%line)
      allocate(cfg%line(n))
      cfg%nline = n
      n = 0
      a = 1
      do b = 1, len(text) + 1
         if (b > len(text)) then
            n = n + 1
            cfg%line(n) = text(a:len(text))
         else if (text(b:b) == new_line('a')) then
            n = n + 1
            cfg%line(n) = text(a:b-1)
            a = b + 1
         end if
      end do
      cfg%cur = 0
      cfg%pos = 0
   end subroutine

   !  where the meaningful part of a line ends ('#' starts a comment)
   integer function content_end(s)
      character(len=*), intent(in) :: s
      integer :: h
      h = index(s, '#')
      if (h == 0) then
         content_end = len_trim(s)
      else
         content_end = len_trim(s(1:h-1))
      end if
   end function

   !  ESMF_ConfigFindLabel: the label starts a line (blanks may precede it); the cursor is left behind it
   subroutine config_find_label(cfg, label, rc)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      integer :: i, a, n
      rc = ML_FAILURE
      n = len_trim(label)
      do i = 1, cfg%nline
         a = verify(cfg%line(i), ' '//achar(9))
         if (a == 0) cycle
         if (a + n - 1 > len(cfg%line(i))) cycle
         if (cfg%line(i)(a:a+n-1) == label(1:n)) then
            cfg%cur = i
            cfg%pos = a + n
            rc = ML_SUCCESS
            return
         end if
      end do
   end subroutine

   !  next blank-separated token on the cursor's line; rc /= 0 when the line (before any comment) is used up
   subroutine config_next_token(cfg, token, rc)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(out) :: token
      integer, intent(out) :: rc
      integer :: e, a, b
      rc = ML_FAILURE
      token = ''
      if (cfg%cur < 1 .or. cfg%cur > cfg%nline) return
      e = content_end(cfg%line(cfg%cur))
      a = cfg%pos
      do while (a <= e)
         if (cfg%line(cfg%cur)(a:a) /= ' ' .and. cfg%line(cfg%cur)(a:a) /= achar(9)) exit
         a = a + 1
      end do
      if (a > e) return
      b = a
      do while (b <= e)
         if (cfg%line(cfg%cur)(b:b) == ' ' .or. cfg%line(cfg%cur)(b:b) == achar(9)) exit
         b = b + 1
      end do
      token = cfg%line(cfg%cur)(a:b-1)
      cfg%pos = b
      rc = ML_SUCCESS
   end subroutine

   !  ESMF_ConfigGetLen: how many values follow the label (-1 with rc /= 0 when the label is absent)
   integer function config_get_len(cfg, label, rc) result(n)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      character(len=ML_MAXPATH) :: tok
      integer :: trc
      n = -1
      call cfg%find_label(label, rc)
      if (rc /= ML_SUCCESS) return
      n = 0
      do
         call cfg%next_token(tok, trc)
         if (trc /= ML_SUCCESS) exit
         n = n + 1
      end do
   end function

   subroutine config_get_string(cfg, value, label, rc, default)
      class(ml_config), intent(inout) :: cfg
      character(len=*), intent(out) :: value
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      character(len=*), intent(in), optional :: default
      call cfg%find_label(label, rc)
      if (rc == ML_SUCCESS) call cfg%next_token(value, rc)
      if (rc /= ML_SUCCESS .and. present(default)) then
         value = default
         rc = ML_SUCCESS
      end if
   end subroutine

   subroutine config_get_int(cfg, value, label, rc, default)
      class(ml_config), intent(inout) :: cfg
      integer, intent(out) :: value
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      integer, intent(in), optional :: default
      character(len=ML_MAXSTR) :: tok
      integer :: ios
      value = 0
      call cfg%get_string(tok, label, rc)
      if (rc == ML_SUCCESS) then
         read(tok, *, iostat=ios) value
         if (ios /= 0) rc = ML_FAILURE
      end if
      if (rc /= ML_SUCCESS .and. present(default)) then
         value = default
         rc = ML_SUCCESS
      end if
   end subroutine

   subroutine config_get_real(cfg, value, label, rc, default)
      class(ml_config), intent(inout) :: cfg
      real, intent(out) :: value
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      real, intent(in), optional :: default
      character(len=ML_MAXSTR) :: tok
      integer :: ios
      value = 0.0
      call cfg%get_string(tok, label, rc)
      if (rc == ML_SUCCESS) then
         read(tok, *, iostat=ios) value
         if (ios /= 0) rc = ML_FAILURE
      end if
      if (rc /= ML_SUCCESS .and. present(default)) then
         value = default
         rc = ML_SUCCESS
      end if
   end subroutine

   subroutine config_get_logical(cfg, value, label, rc, default)
      class(ml_config), intent(inout) :: cfg
      logical, intent(out) :: value
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      logical, intent(in), optional :: default
      character(len=ML_MAXSTR) :: tok
      value = .false.
      call cfg%get_string(tok, label, rc)
      if (rc == ML_SUCCESS) then
         select case (trim(lowercase(tok)))
         case ('t', 'true', '.true.', '.t.', 'yes', 'y', 'on')
            value = .true.
         case ('f', 'false', '.false.', '.f.', 'no', 'n', 'off')
            value = .false.
         case default
            rc = ML_FAILURE
         end select
      end if
      if (rc /= ML_SUCCESS .and. present(default)) then
         value = default
         rc = ML_SUCCESS
      end if
   end subroutine

   subroutine config_get_reals(cfg, values, label, rc)
      class(ml_config), intent(inout) :: cfg
      real, intent(out) :: values(:)
      character(len=*), intent(in) :: label
      integer, intent(out) :: rc
      character(len=ML_MAXSTR) :: tok
      integer :: i, ios
      values = 0.0
      call cfg%find_label(label, rc)
      if (rc /= ML_SUCCESS) return
      do i = 1, size(values)
         call cfg%next_token(tok, rc)
         if (rc /= ML_SUCCESS) return
         read(tok, *, iostat=ios) values(i)
         if (ios /= 0) then
            rc = ML_FAILURE
            return
         end if
      end do
   end subroutine

   function lowercase(s) result(t)
      character(len=*), intent(in) :: s
      character(len=len(s)) :: t
      integer :: i, c
      t = s
      do i = 1, len(s)
         c = iachar(s(i:i))
         if (c >= iachar('A') .and. c <= iachar('Z')) t(i:i) = achar(c + 32)
      end do
   end function

   ! =================================================================== states

   subroutine state_add_spec(st, short_name, dims, vloc, units, long_name, ungridded, restart_skip, averaging_interval, rc)
      class(ml_state), intent(inout) :: st
      character(len=*), intent(in) :: short_name
      integer, intent(in) :: dims, vloc
      character(len=*), intent(in), optional :: units, long_name
      integer, intent(in), optional :: ungridded, averaging_interval
      logical, intent(in), optional :: restart_skip
      integer, intent(out), optional :: rc
      type(ml_field), allocatable :: grown(:)
      if (present(rc)) rc = ML_SUCCESS
      if (st%index_of(short_name) > 0) then        ! MAPL refuses a second spec of the same name
         if (present(rc)) rc = ML_FAILURE
         return
      end if
      if (.not. allocated(st%f)) allocate(st%f(16))
      if (st%n == size(st%f)) then
         allocate(grown(2 * size(st%f)))
         grown(1:st%n) = st%f(1:st%n)
         call move_alloc(grown, st%f)
      end if
      st%n = st%n + 1
      st%f(st%n)%name = short_name
      st%f(st%n)%dims = dims
      st%f(st%n)%vloc = vloc
      if (present(units)) st%f(st%n)%units = units
      if (present(long_name)) st%f(st%n)%long_name = long_name
      if (present(ungridded)) st%f(st%n)%ungridded = ungridded
      if (present(restart_skip)) st%f(st%n)%restart_skip = restart_skip
      if (present(averaging_interval)) st%f(st%n)%averaging_interval = averaging_interval
   end subroutine

   integer function state_index_of(st, name) result(k)
      class(ml_state), intent(in) :: st
      character(len=*), intent(in) :: name
      integer :: i
      k = 0
      do i = 1, st%n
         if (trim(st%f(i)%name) == trim(name)) then
            k = i
            return
         end if
      end do
   end function

   logical function state_has(st, name)
      class(ml_state), intent(in) :: st
      character(len=*), intent(in) :: name
      state_has = st%index_of(name) > 0
   end function

   logical function state_is_allocated(st, name)
      class(ml_state), intent(in) :: st
      character(len=*), intent(in) :: name
      integer :: k
      state_is_allocated = .false.
      k = st%index_of(name)
      if (k == 0) return
      state_is_allocated = associated(st%f(k)%p2) .or. associated(st%f(k)%p3) .or. associated(st%f(k)%p4)
   end function

   !  Gives a declared field its storage (zero-filled): what the rest of GEOS does for a component's imports,
   !  HISTORY for the exports somebody asked for, and MAPL_GenericInitialize for the INTERNAL state.
   subroutine state_allocate_field(st, name, grid, rc)
      class(ml_state), intent(inout) :: st
      character(len=*), intent(in) :: name
      type(ml_grid), intent(in) :: grid
      integer, intent(out) :: rc
      integer :: k, k0
      rc = ML_FAILURE
      k = st%index_of(name)
      if (k == 0) return
      rc = ML_SUCCESS
      if (st%is_allocated(name)) return
      if (st%f(k)%dims == ML_DIMS_HORZ_ONLY) then
         allocate(st%f(k)%p2(grid%im, grid%jm))
         st%f(k)%p2 = 0.0
      else
         k0 = merge(0, 1, st%f(k)%vloc == ML_VLOC_EDGE)
         if (st%f(k)%ungridded > 0) then
            allocate(st%f(k)%p4(grid%im, grid%jm, k0:grid%km, st%f(k)%ungridded))
            st%f(k)%p4 = 0.0
         else
            allocate(st%f(k)%p3(grid%im, grid%jm, k0:grid%km))
            st%f(k)%p3 = 0.0
         end if
      end if
   end subroutine

   !  MAPL_GetPointer: a name the state does not declare is an error; a declared field without storage
   !  (an export nobody asked for) gives a null pointer and rc = 0
   subroutine state_get_pointer_2d(st, ptr, name, rc)
      class(ml_state), intent(in) :: st
      real, pointer, intent(out) :: ptr(:,:)
      character(len=*), intent(in) :: name
      integer, intent(out) :: rc
      integer :: k
      ptr => null()
      k = st%index_of(name)
      rc = merge(ML_SUCCESS, ML_FAILURE, k > 0)
      if (k == 0) return
      if (st%f(k)%dims /= ML_DIMS_HORZ_ONLY) then
         rc = ML_FAILURE
         return
      end if
      ptr => st%f(k)%p2
   end subroutine

   subroutine state_get_pointer_3d(st, ptr, name, rc)
      class(ml_state), intent(in) :: st
      real, pointer, intent(out) :: ptr(:,:,:)
      character(len=*), intent(in) :: name
      integer, intent(out) :: rc
      integer :: k
      ptr => null()
      k = st%index_of(name)
      rc = merge(ML_SUCCESS, ML_FAILURE, k > 0)
      if (k == 0) return
      if (st%f(k)%dims /= ML_DIMS_HORZ_VERT .or. st%f(k)%ungridded > 0) then
         rc = ML_FAILURE
         return
      end if
      ptr => st%f(k)%p3          ! keeps its bounds: edge fields are (im,jm,0:km)
   end subroutine

   subroutine state_get_pointer_4d(st, ptr, name, rc)
      class(ml_state), intent(in) :: st
      real, pointer, intent(out) :: ptr(:,:,:,:)
      character(len=*), intent(in) :: name
      integer, intent(out) :: rc
      integer :: k
      ptr => null()
      k = st%index_of(name)
      rc = merge(ML_SUCCESS, ML_FAILURE, k > 0)
      if (k == 0) return
      if (st%f(k)%ungridded <= 0) then
         rc = ML_FAILURE
         return
      end if
      ptr => st%f(k)%p4
   end subroutine

   ! =================================================================== time

   !  days since 0001-01-01 of a proleptic Gregorian date
   integer(int64) function days_from_civil(y, m, d) result(days)
      integer, intent(in) :: y, m, d
      integer(int64) :: yy, era, yoe, doy, doe, mm
      yy = y
      if (m <= 2) yy = yy - 1
      era = yy / 400
      if (yy < 0) era = (yy - 399) / 400
      yoe = yy - era * 400
      mm = m
      doy = (153 * (mm + merge(-3_int64, 9_int64, mm > 2)) + 2) / 5 + d - 1
      doe = yoe * 365 + yoe / 4 - yoe / 100 + doy
      days = era * 146097 + doe - 306            ! 0001-01-01 is day 0
   end function

   subroutine civil_from_days(days, y, m, d)
      integer(int64), intent(in) :: days
      integer, intent(out) :: y, m, d
      integer(int64) :: z, era, doe, yoe, doy, mp
      z = days + 306
      era = z / 146097
      if (z < 0) era = (z - 146096) / 146097
      doe = z - era * 146097
      yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365
      doy = doe - (365 * yoe + yoe / 4 - yoe / 100)
      mp = (5 * doy + 2) / 153
      d = int(doy - (153 * mp + 2) / 5 + 1)
      m = int(mp + merge(3_int64, -9_int64, mp < 10))
      y = int(yoe + era * 400)
      if (m <= 2) y = y + 1
   end subroutine

   integer(int64) function ml_time_to_seconds(yy, mm, dd, h, m, s) result(t)
      integer, intent(in) :: yy, mm, dd, h, m, s
      t = days_from_civil(yy, mm, dd) * 86400_int64 + int(h, int64) * 3600 + int(m, int64) * 60 + s
   end function

   subroutine clock_set(clock, yy, mm, dd, h, m, s, dt)
      class(ml_clock), intent(inout) :: clock
      integer, intent(in) :: yy, mm, dd, h, m, s, dt
      clock%now = ml_time_to_seconds(yy, mm, dd, h, m, s)
      clock%dt = dt
   end subroutine

   subroutine clock_get(clock, yy, mm, dd, h, m, s)
      class(ml_clock), intent(in) :: clock
      integer, intent(out) :: yy, mm, dd, h, m, s
      integer(int64) :: days, rest
      days = clock%now / 86400
      rest = clock%now - days * 86400
      call civil_from_days(days, yy, mm, dd)
      h = int(rest / 3600)
      m = int(mod(rest, 3600_int64) / 60)
      s = int(mod(rest, 60_int64))
   end subroutine

   subroutine clock_advance(clock)
      class(ml_clock), intent(inout) :: clock
      clock%now = clock%now + clock%dt
   end subroutine

   !  MAPL_PackTime
   subroutine ml_pack_time(packed, a, b, c)
      integer, intent(out) :: packed
      integer, intent(in) :: a, b, c
      packed = a * 10000 + b * 100 + c
   end subroutine

   !  ESMF non-sticky alarm as MAPL uses it: after the clock has moved from t_old to t_new the alarm rings
   !  iff one of its ring times lies in (t_old, t_new]; it rings for that step only, and not at all once
   !  someone has turned the ringer off.
   subroutine alarm_update(alarm, t_old, t_new)
      class(ml_alarm), intent(inout) :: alarm
      integer(int64), intent(in) :: t_old, t_new
      integer(int64) :: k
      if (alarm%interval <= 0) then
         alarm%ringing = .true.
         return
      end if
      !  largest ring time <= t_new
      k = floor_div(t_new - alarm%first_ring, int(alarm%interval, int64))
      alarm%ringing = alarm%first_ring + k * alarm%interval > t_old
   end subroutine

   integer(int64) function floor_div(a, b) result(q)
      integer(int64), intent(in) :: a, b
      q = a / b
      if (mod(a, b) /= 0 .and. ((a < 0) .neqv. (b < 0))) q = q - 1
   end function

   logical function alarm_is_ringing(alarm)
      class(ml_alarm), intent(in) :: alarm
      alarm_is_ringing = alarm%ringing
   end function

   subroutine alarm_ringer_off(alarm)
      class(ml_alarm), intent(inout) :: alarm
      alarm%ringing = .false.
   end subroutine

   ! =================================================================== grid components

   logical function ml_am_i_root()
      ml_am_i_root = .true.          ! one rank in the mock
   end function

   !  MAPL_MaxMin: a global max/min print (a reduction over ranks in MAPL; OH_GridCompMod.F90:1550)
   subroutine ml_maxmin(label, a)
      character(len=*), intent(in) :: label
      real, intent(in) :: a(:,:,:)
      if (ml_am_i_root()) print '(a,2es16.7)', trim(label)//' max, min = ', maxval(a), minval(a)
   end subroutine

   function ml_gridcomp_create(name, config, grid, rc_dir) result(gc)
      character(len=*), intent(in) :: name
      type(ml_config), pointer, intent(in) :: config
      type(ml_grid), intent(in) :: grid
      character(len=*), intent(in), optional :: rc_dir
      type(ml_gridcomp), pointer :: gc
      allocate(gc)
      gc%name = name
      gc%config => config
      gc%grid = grid
      if (present(rc_dir)) gc%rc_dir = rc_dir
      gc%import%name = trim(name)//'_Imports'
      gc%export%name = trim(name)//'_Exports'
      gc%internal%name = trim(name)//'_Internals'
      allocate(gc%entry)
      allocate(gc%children(8))
      allocate(gc%child_export_name(0), gc%child_export_id(0))
   end function

   !  MAPL_AddChild: creates the child on the parent's grid and config and runs ITS SetServices; returns its id
   integer function ml_add_child(gc, name, SS, rc) result(id)
      type(ml_gridcomp), intent(inout) :: gc
      character(len=*), intent(in) :: name
      procedure(ml_set_services) :: SS
      integer, intent(out) :: rc
      type(ml_gridcomp_ptr), allocatable :: grown(:)
      id = -1
      if (gc%nchildren == size(gc%children)) then
         allocate(grown(2 * size(gc%children)))
         grown(1:gc%nchildren) = gc%children(1:gc%nchildren)
         call move_alloc(grown, gc%children)
      end if
      gc%nchildren = gc%nchildren + 1
      id = gc%nchildren
      gc%children(id)%gc => ml_gridcomp_create(name, gc%config, gc%grid, gc%rc_dir)
      call SS(gc%children(id)%gc, rc)
   end function

   !  MAPL_GridCompSetEntryPoint: run methods registered in order become phase 1, 2, ...
   subroutine ml_set_entry_point(gc, method, proc, rc)
      type(ml_gridcomp), intent(inout) :: gc
      integer, intent(in) :: method
      procedure(ml_method) :: proc
      integer, intent(out) :: rc
      rc = ML_SUCCESS
      if (method == ML_METHOD_INITIALIZE) then
         gc%entry%initialize%p => proc
      else if (method == ML_METHOD_RUN .and. gc%entry%n_run_phases < size(gc%entry%run_phase)) then
         gc%entry%n_run_phases = gc%entry%n_run_phases + 1
         gc%entry%run_phase(gc%entry%n_run_phases)%p => proc
      else
         rc = ML_FAILURE
      end if
   end subroutine

   !  MAPL_AddExportSpec(GC, SHORT_NAME=..., CHILD_ID=...): the parent re-exports a child's field
   subroutine ml_add_child_export(gc, short_name, child_id, rc)
      type(ml_gridcomp), intent(inout) :: gc
      character(len=*), intent(in) :: short_name
      integer, intent(in) :: child_id
      integer, intent(out) :: rc
      rc = ML_FAILURE
      if (child_id < 1 .or. child_id > gc%nchildren) return
      gc%child_export_name = [character(len=ML_MAXSTR) :: gc%child_export_name, short_name]
      gc%child_export_id = [gc%child_export_id, child_id]
      rc = ML_SUCCESS
   end subroutine

   !  what a consumer of the parent's export gets: the child's export field of that name, or - for a field the
   !  child declared add2export from its INTERNAL state - that internal field
   subroutine ml_child_export_field(gc, short_name, ptr, rc)
      type(ml_gridcomp), intent(in) :: gc
      character(len=*), intent(in) :: short_name
      real, pointer, intent(out) :: ptr(:,:,:)
      integer, intent(out) :: rc
      integer :: q
      ptr => null()
      rc = ML_FAILURE
      do q = 1, size(gc%child_export_id)
         if (trim(gc%child_export_name(q)) /= trim(short_name)) cycle
         associate (child => gc%children(gc%child_export_id(q))%gc)
            if (child%internal%has(short_name)) then
               call child%internal%get_pointer(ptr, short_name, rc)
            else
               call child%export%get_pointer(ptr, short_name, rc)
            end if
         end associate
         return
      end do
   end subroutine

   !  MAPL_GenericInitialize, as far as OH relies on it: storage for the INTERNAL state, the run alarm from
   !  <NAME>_DT / <NAME>_REFERENCE_TIME of the universal config (RUN_DT when absent: rings at every step),
   !  then the children's Initialize.  The alarm is backed off by one heartbeat, as MAPL does "since we
   !  advance the clock AFTER the run method": with the default reference time it rings during the LAST
   !  step of every interval, with <NAME>_REFERENCE_TIME = the heartbeat during the FIRST (NOTES.wiki:57-59).
   subroutine ml_generic_initialize(gc, import, export, clock, rc)
      type(ml_gridcomp), intent(inout), target :: gc
      type(ml_state), intent(inout) :: import, export
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: rc
      integer :: i, dt, ref_hms, yy, mm, dd, h, m, s, trc
      integer(int64) :: ref
      rc = ML_SUCCESS
      do i = 1, gc%internal%n
         call gc%internal%allocate_field(gc%internal%f(i)%name, gc%grid, rc)
         if (rc /= ML_SUCCESS) return
      end do
      dt = clock%dt
      ref_hms = 0
      if (associated(gc%config)) then
         call gc%config%get_int(dt, trim(gc%name)//'_DT:', trc, default=clock%dt)
         call gc%config%get_int(ref_hms, trim(gc%name)//'_REFERENCE_TIME:', trc, default=0)
      end if
      call clock%get(yy, mm, dd, h, m, s)
      ref = ml_time_to_seconds(yy, mm, dd, ref_hms / 10000, mod(ref_hms, 10000) / 100, mod(ref_hms, 100))
      gc%runalarm%interval = dt
      gc%runalarm%first_ring = ref - clock%dt
      !  "if (ringTime == currTime) call ESMF_AlarmRingerOn": the state at the very first step
      call gc%runalarm%update(clock%now - clock%dt, clock%now)
      do i = 1, gc%nchildren
         call ml_gridcomp_initialize(gc%children(i)%gc, clock, rc)
         if (rc /= ML_SUCCESS) return
      end do
   end subroutine

   !  ESMF_GridCompInitialize: the component's own Initialize if it registered one, else the generic one
   recursive subroutine ml_gridcomp_initialize(gc, clock, rc)
      type(ml_gridcomp), intent(inout), target :: gc
      type(ml_clock), intent(inout) :: clock
      integer, intent(out) :: rc
      if (associated(gc%entry%initialize%p)) then
         call gc%entry%initialize%p(gc, gc%import, gc%export, clock, rc)
      else
         call ml_generic_initialize(gc, gc%import, gc%export, clock, rc)
      end if
   end subroutine

   !  ESMF_GridCompRun(phase=...)
   recursive subroutine ml_gridcomp_run(gc, clock, phase, rc)
      type(ml_gridcomp), intent(inout), target :: gc
      type(ml_clock), intent(inout) :: clock
      integer, intent(in) :: phase
      integer, intent(out) :: rc
      rc = ML_FAILURE
      if (phase < 1 .or. phase > gc%entry%n_run_phases) return
      call gc%entry%run_phase(phase)%p(gc, gc%import, gc%export, clock, rc)
   end subroutine

   !  ESMF_ClockAdvance as the cap calls it AFTER the run methods: moves the clock and re-evaluates the run alarm
   !  of the component and of everything below it
   recursive subroutine ml_advance(gc, clock, t_old)
      type(ml_gridcomp), intent(inout) :: gc
      type(ml_clock), intent(inout) :: clock
      integer(int64), intent(in), optional :: t_old
      integer(int64) :: before
      integer :: i
      if (present(t_old)) then
         before = t_old
      else
         before = clock%now
         call clock%advance()
      end if
      call gc%runalarm%update(before, clock%now)
      do i = 1, gc%nchildren
         call ml_advance(gc%children(i)%gc, clock, before)
      end do
   end subroutine

end module mapl_lite
