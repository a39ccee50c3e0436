!  mapl_lite/ESMF.F90 -- a mock of the small part of ESMF that QuickChem's grid components touch, under ESMF's OWN
!  module name, procedure names and keyword arguments.
!
!  ESMF cannot be built here (SURVEY.md §8c), and GEOS itself is out of tree.  The OH GridComp shell
!  (../oh_gridcomp.F90; and the reference's own QuickChem_GridCompMod.F90, compiled in place by oracle/Makefile) says `use ESMF` / `use MAPL` and calls ESMF_ConfigGetAttribute,
!  ESMF_GridCompGet, ESMF_UserCompSetInternalState, ESMF_ClockGet, ESMF_AlarmIsRinging ... exactly as the
!  reference does (OH_GridComp/OH_GridCompMod.F90:516-605, 793, 855-897, 1147-1185, 1820): inside GEOS the same
!  two source files compile against the real libraries, here against this directory (mapl_lite/), which a GEOS
!  build simply does not compile.
!
!  Only what the reference relies on is restated:
!     ESMF_Config     "label: value value # comment" lines; FindLabel / GetAttribute (scalar, next token, vector) /
!                     GetLen                                                    (OH_GridCompMod.F90:532-589)
!     ESMF_Clock / ESMF_Time / ESMF_Alarm   a heartbeat clock; non-sticky alarms that ring for the step in which a
!                     ring time falls, and not after ESMF_AlarmRingerOff           (:1161-1185, 1820)
!     ESMF_State      named fields with storage or without (an export nobody asked for)
!     ESMF_GridComp   name, config, grid, states, entry points by method and phase, one named user state
!  Every object is a handle (a type holding one pointer), as in ESMF: copies of a handle name the same object.
!  Names beginning with esmfl_ are the mock's own (the cap and module MAPL use them); nothing in the grid
!  components does.
module ESMF
   use, intrinsic :: iso_fortran_env, only: int64, real64
   use, intrinsic :: iso_c_binding
   implicit none
   private

   integer, parameter, public :: ESMF_SUCCESS = 0, ESMF_FAILURE = -1
   integer, parameter, public :: ESMF_MAXSTR = 128, ESMF_MAXPATHLEN = 1024
   integer, parameter, public :: ESMF_KIND_R4 = selected_real_kind(6), ESMF_KIND_R8 = selected_real_kind(12)
   integer, parameter, public :: ESMF_KIND_I4 = selected_int_kind(9), ESMF_KIND_I8 = selected_int_kind(18)

   type, public :: ESMF_Method_Flag
      integer :: v = 0
   end type
   type(ESMF_Method_Flag), parameter, public :: ESMF_METHOD_INITIALIZE = ESMF_Method_Flag(1), &
      ESMF_METHOD_RUN = ESMF_Method_Flag(2), ESMF_METHOD_FINALIZE = ESMF_Method_Flag(3)

   ! ------------------------------------------------------------------ resource files
   type, public :: esmfl_config_impl
      character(len=ESMF_MAXPATHLEN), allocatable :: line(:)
      integer :: nline = 0
      integer :: cur = 0, pos = 0          ! where FindLabel left the cursor
   end type
   type, public :: ESMF_Config
      type(esmfl_config_impl), pointer :: p => null()
   end type

   ! ------------------------------------------------------------------ time
   type, public :: ESMF_Time
      integer(int64) :: s = 0              ! seconds since 0001-01-01 00:00:00 (proleptic Gregorian)
   end type
   type, public :: ESMF_TimeInterval
      integer(int64) :: s = 0
   end type
   type, public :: esmfl_alarm_impl
      integer(int64) :: first_ring = 0     ! a ring time; the others are first_ring + k * interval
      integer(int64) :: interval = 0       ! seconds; 0 = rings at every step
      logical :: ringing = .true.
   end type
   type, public :: ESMF_Alarm
      type(esmfl_alarm_impl), pointer :: p => null()
   end type
   type, public :: esmfl_clock_impl
      integer(int64) :: now = 0
      integer(int64) :: dt = 450           ! the heartbeat
      type(ESMF_Alarm), allocatable :: alarms(:)
      integer :: nalarms = 0
   end type
   type, public :: ESMF_Clock
      type(esmfl_clock_impl), pointer :: p => null()
   end type

   ! ------------------------------------------------------------------ grid, states
   type, public :: esmfl_grid_impl
      integer :: im = 0, jm = 0, km = 0               ! local = global in the mock (one rank)
      real, pointer :: LATS(:,:) => null(), LONS(:,:) => null()     ! radians
   end type
   type, public :: ESMF_Grid
      type(esmfl_grid_impl), pointer :: p => null()
   end type

   integer, parameter, public :: ESMFL_DIMS_HORZ_ONLY = 2, ESMFL_DIMS_HORZ_VERT = 3
   integer, parameter, public :: ESMFL_VLOC_NONE = 0, ESMFL_VLOC_CENTER = 1, ESMFL_VLOC_EDGE = 2

   type, public :: esmfl_field
      character(len=ESMF_MAXSTR) :: name = ''
      character(len=ESMF_MAXSTR) :: units = '', long_name = ''
      integer :: dims = ESMFL_DIMS_HORZ_VERT
      integer :: vloc = ESMFL_VLOC_CENTER
      integer :: ungridded = 0             ! size of a 4th dimension, 0 = none
      integer :: restart = 0               ! MAPL_Restart* code of the spec
      integer :: refresh_interval = 0, averaging_interval = 0    ! seconds; 86400 for the *_avg24 imports
      logical :: add2export = .false.
      real, pointer :: p2(:,:) => null(), p3(:,:,:) => null(), p4(:,:,:,:) => null()
   end type
   type, public :: esmfl_state_impl
      character(len=ESMF_MAXSTR) :: name = ''
      type(esmfl_field), allocatable :: f(:)
      integer :: n = 0
   end type
   type, public :: ESMF_State
      type(esmfl_state_impl), pointer :: p => null()
   end type
   !  a handle on one field of a state: the reference's OH Initialize declares two and never uses them
   !  (OH_GridCompMod.F90:839; its ESMF_StateGet / ESMF_AttributeSet calls are comments, :912-919)
   type, public :: ESMF_Field
      type(esmfl_field), pointer :: p => null()
   end type

   ! ------------------------------------------------------------------ grid components
   type, public :: ESMF_GridComp
      type(esmfl_gridcomp_impl), pointer :: p => null()
   end type

   abstract interface
      !  what ESMF_GridCompSetEntryPoint registers (RC optional, as the components declare it: OH_GridCompMod.F90:823,967)
      subroutine esmfl_method(gc, importState, exportState, clock, rc)
         import :: ESMF_GridComp, ESMF_State, ESMF_Clock
         type(ESMF_GridComp), intent(inout) :: gc
         type(ESMF_State), intent(inout) :: importState, exportState
         type(ESMF_Clock), intent(inout) :: clock
         integer, optional, intent(out) :: rc
      end subroutine
      subroutine esmfl_set_services(gc, rc)
         import :: ESMF_GridComp
         type(ESMF_GridComp), intent(inout) :: gc
         integer, intent(out) :: rc
      end subroutine
   end interface

   type :: method_ptr
      procedure(esmfl_method), pointer, nopass :: p => null()
   end type

   integer, parameter :: kMaxPhases = 8
   type, public :: esmfl_gridcomp_impl
      character(len=ESMF_MAXSTR) :: name = ''
      type(ESMF_Config) :: config                       ! the "universal" config (AGCM.rc)
      type(ESMF_Grid) :: grid
      type(ESMF_State) :: importState, exportState
      type(method_ptr) :: initialize(kMaxPhases), run(kMaxPhases)
      !  ESMF_UserCompSetInternalState: the bytes of the caller's wrap object under its name
      character(len=ESMF_MAXSTR) :: user_state_name = ''
      integer(c_int8_t), allocatable :: user_state(:)
      class(*), pointer :: generic_state => null()       ! where MAPL keeps its MAPL_MetaComp
   end type

   character(len=ESMF_MAXPATHLEN), save :: run_dir = ''     ! where ESMF_ConfigLoadFile looks for relative names

   public :: esmfl_method, esmfl_set_services
   public :: ESMF_ConfigCreate, ESMF_ConfigDestroy, ESMF_ConfigLoadFile, ESMF_ConfigFindLabel, ESMF_ConfigGetLen
   public :: ESMF_ConfigGetAttribute
   public :: ESMF_TimeSet, ESMF_TimeGet, ESMF_TimeIntervalSet, ESMF_TimeIntervalGet
   public :: ESMF_ClockCreate, ESMF_ClockGet, ESMF_ClockAdvance
   public :: ESMF_AlarmCreate, ESMF_AlarmIsRinging, ESMF_AlarmRingerOff, ESMF_AlarmRingerOn
   public :: ESMF_GridCompCreate, ESMF_GridCompGet, ESMF_GridCompSetServices, ESMF_GridCompSetEntryPoint
   public :: ESMF_GridCompInitialize, ESMF_GridCompRun
   public :: ESMF_UserCompSetInternalState, ESMF_UserCompGetInternalState
   public :: ESMF_StateCreate
   public :: esmfl_set_run_dir, esmfl_grid_create, esmfl_state_add, esmfl_state_index, esmfl_state_allocate
   public :: esmfl_field_has_storage

   interface ESMF_ConfigGetAttribute
      module procedure config_get_string, config_get_int, config_get_real, config_get_logical, config_get_reals
   end interface

contains

   subroutine set_rc(rc, value)
      integer, intent(out), optional :: rc
      integer, intent(in) :: value
      if (present(rc)) rc = value
   end subroutine

   !  the cap's run directory: relative resource-file names are looked up there (GEOS runs in it)
   subroutine esmfl_set_run_dir(dir)
      character(len=*), intent(in) :: dir
      run_dir = dir
   end subroutine

   ! =================================================================== config

   function ESMF_ConfigCreate(rc) result(config)
      integer, intent(out), optional :: rc
      type(ESMF_Config) :: config
      allocate(config%p)
      allocate(config%p%line(1))
      call set_rc(rc, ESMF_SUCCESS)
   end function

   subroutine ESMF_ConfigDestroy(config, rc)
      type(ESMF_Config), intent(inout) :: config
      integer, intent(out), optional :: rc
      if (associated(config%p)) deallocate(config%p)
      config%p => null()
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_ConfigLoadFile(config, filename, rc)
      type(ESMF_Config), intent(inout) :: config
      character(len=*), intent(in) :: filename
      integer, intent(out), optional :: rc
      integer :: u, ios, n
      character(len=ESMF_MAXPATHLEN) :: buf, path
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(config%p)) return
      path = filename
      if (len_trim(run_dir) > 0 .and. filename(1:1) /= '/') path = trim(run_dir)//'/'//filename
      open(newunit=u, file=trim(path), status='old', action='read', iostat=ios)
      if (ios /= 0) return
      n = 0
      do
         read(u, '(a)', iostat=ios) buf
         if (ios /= 0) exit
         n = n + 1
      end do
      rewind(u)
      if (allocated(config%p%line)) deallocate(config%p%line)
      allocate(config%p%line(max(n, 1)))
      config%p%nline = n
      do n = 1, config%p%nline
         read(u, '(a)') config%p%line(n)
      end do
      close(u)
      config%p%cur = 0
      config%p%pos = 0
      call set_rc(rc, ESMF_SUCCESS)
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

   !  the label starts a line (blanks may precede it); the cursor is left behind it
   subroutine ESMF_ConfigFindLabel(config, label, isPresent, rc)
      type(ESMF_Config), intent(inout) :: config
      character(len=*), intent(in) :: label
      logical, intent(out), optional :: isPresent
      integer, intent(out), optional :: rc
      integer :: i, a, n
      if (present(isPresent)) isPresent = .false.
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(config%p)) return
      if (present(isPresent)) call set_rc(rc, ESMF_SUCCESS)        ! with isPresent an absent label is not an error
      n = len_trim(label)
      associate (c => config%p)
         do i = 1, c%nline
            a = verify(c%line(i), ' '//achar(9))
            if (a == 0) cycle
            if (a + n - 1 > len(c%line(i))) cycle
            if (c%line(i)(a:a+n-1) == label(1:n)) then
               c%cur = i
               c%pos = a + n
               if (present(isPresent)) isPresent = .true.
               call set_rc(rc, ESMF_SUCCESS)
               return
            end if
         end do
      end associate
   end subroutine

   !  next blank-separated token on the cursor's line; ok = .false. when the line (before any comment) is used up
   subroutine next_token(c, token, ok)
      type(esmfl_config_impl), intent(inout) :: c
      character(len=*), intent(out) :: token
      logical, intent(out) :: ok
      integer :: e, a, b
      ok = .false.
      token = ''
      if (c%cur < 1 .or. c%cur > c%nline) return
      e = content_end(c%line(c%cur))
      a = c%pos
      do while (a <= e)
         if (c%line(c%cur)(a:a) /= ' ' .and. c%line(c%cur)(a:a) /= achar(9)) exit
         a = a + 1
      end do
      if (a > e) return
      b = a
      do while (b <= e)
         if (c%line(c%cur)(b:b) == ' ' .or. c%line(c%cur)(b:b) == achar(9)) exit
         b = b + 1
      end do
      token = c%line(c%cur)(a:b-1)
      c%pos = b
      ok = .true.
   end subroutine

   !  how many values follow the label
   integer function ESMF_ConfigGetLen(config, label, rc) result(n)
      type(ESMF_Config), intent(inout) :: config
      character(len=*), intent(in), optional :: label
      integer, intent(out), optional :: rc
      character(len=ESMF_MAXPATHLEN) :: tok
      logical :: ok
      integer :: status
      n = -1
      status = ESMF_SUCCESS
      if (present(label)) call ESMF_ConfigFindLabel(config, label, rc=status)
      call set_rc(rc, status)
      if (status /= ESMF_SUCCESS) return
      n = 0
      do
         call next_token(config%p, tok, ok)
         if (.not. ok) exit
         n = n + 1
      end do
   end function

   !  With label: the first value behind it.  Without: the next token on the line FindLabel stopped at.
   subroutine config_get_string(config, value, label, default, rc)
      type(ESMF_Config), intent(inout) :: config
      character(len=*), intent(out) :: value
      character(len=*), intent(in), optional :: label, default
      integer, intent(out), optional :: rc
      integer :: status
      logical :: ok
      status = ESMF_SUCCESS
      value = ''
      if (present(label)) call ESMF_ConfigFindLabel(config, label, rc=status)
      if (status == ESMF_SUCCESS) then
         call next_token(config%p, value, ok)
         if (.not. ok) status = ESMF_FAILURE
      end if
      if (status /= ESMF_SUCCESS .and. present(default)) then
         value = default
         status = ESMF_SUCCESS
      end if
      call set_rc(rc, status)
   end subroutine

   subroutine config_get_int(config, value, label, default, rc)
      type(ESMF_Config), intent(inout) :: config
      integer, intent(out) :: value
      character(len=*), intent(in), optional :: label
      integer, intent(in), optional :: default
      integer, intent(out), optional :: rc
      character(len=ESMF_MAXSTR) :: tok
      integer :: status, ios
      value = 0
      call config_get_string(config, tok, label=label, rc=status)
      if (status == ESMF_SUCCESS) then
         read(tok, *, iostat=ios) value
         if (ios /= 0) status = ESMF_FAILURE
      end if
      if (status /= ESMF_SUCCESS .and. present(default)) then
         value = default
         status = ESMF_SUCCESS
      end if
      call set_rc(rc, status)
   end subroutine

   subroutine config_get_real(config, value, label, default, rc)
      type(ESMF_Config), intent(inout) :: config
      real, intent(out) :: value
      character(len=*), intent(in), optional :: label
      real, intent(in), optional :: default
      integer, intent(out), optional :: rc
      character(len=ESMF_MAXSTR) :: tok
      integer :: status, ios
      value = 0.0
      call config_get_string(config, tok, label=label, rc=status)
      if (status == ESMF_SUCCESS) then
         read(tok, *, iostat=ios) value
         if (ios /= 0) status = ESMF_FAILURE
      end if
      if (status /= ESMF_SUCCESS .and. present(default)) then
         value = default
         status = ESMF_SUCCESS
      end if
      call set_rc(rc, status)
   end subroutine

   subroutine config_get_logical(config, value, label, default, rc)
      type(ESMF_Config), intent(inout) :: config
      logical, intent(out) :: value
      character(len=*), intent(in), optional :: label
      logical, intent(in), optional :: default
      integer, intent(out), optional :: rc
      character(len=ESMF_MAXSTR) :: tok
      integer :: status
      value = .false.
      call config_get_string(config, tok, label=label, rc=status)
      if (status == ESMF_SUCCESS) then
         select case (trim(lowercase(tok)))
         case ('t', 'true', '.true.', '.t.', 'yes', 'y', 'on')
            value = .true.
         case ('f', 'false', '.false.', '.f.', 'no', 'n', 'off')
            value = .false.
         case default
            status = ESMF_FAILURE
         end select
      end if
      if (status /= ESMF_SUCCESS .and. present(default)) then
         value = default
         status = ESMF_SUCCESS
      end if
      call set_rc(rc, status)
   end subroutine

   subroutine config_get_reals(config, valueList, count, label, default, rc)
      type(ESMF_Config), intent(inout) :: config
      real, intent(out) :: valueList(:)
      integer, intent(in), optional :: count
      character(len=*), intent(in), optional :: label
      real, intent(in), optional :: default
      integer, intent(out), optional :: rc
      character(len=ESMF_MAXSTR) :: tok
      integer :: i, ios, n, status
      logical :: ok
      valueList = 0.0
      if (present(default)) valueList = default
      n = size(valueList)
      if (present(count)) n = min(n, count)
      status = ESMF_SUCCESS
      if (present(label)) call ESMF_ConfigFindLabel(config, label, rc=status)
      do i = 1, n
         if (status /= ESMF_SUCCESS) exit
         call next_token(config%p, tok, ok)
         if (.not. ok) then
            status = ESMF_FAILURE
            exit
         end if
         read(tok, *, iostat=ios) valueList(i)
         if (ios /= 0) status = ESMF_FAILURE
      end do
      if (status /= ESMF_SUCCESS .and. present(default)) status = ESMF_SUCCESS
      call set_rc(rc, status)
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

   subroutine ESMF_TimeSet(time, YY, MM, DD, H, M, S, rc)
      type(ESMF_Time), intent(inout) :: time
      integer, intent(in), optional :: YY, MM, DD, H, M, S
      integer, intent(out), optional :: rc
      integer :: y, mo, d
      y = 1; mo = 1; d = 1
      if (present(YY)) y = YY
      if (present(MM)) mo = MM
      if (present(DD)) d = DD
      time%s = days_from_civil(y, mo, d) * 86400_int64
      if (present(H)) time%s = time%s + int(H, int64) * 3600
      if (present(M)) time%s = time%s + int(M, int64) * 60
      if (present(S)) time%s = time%s + S
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_TimeGet(time, YY, MM, DD, H, M, S, rc)
      type(ESMF_Time), intent(in) :: time
      integer, intent(out), optional :: YY, MM, DD, H, M, S
      integer, intent(out), optional :: rc
      integer(int64) :: days, rest
      integer :: y, mo, d
      days = time%s / 86400
      rest = time%s - days * 86400
      call civil_from_days(days, y, mo, d)
      if (present(YY)) YY = y
      if (present(MM)) MM = mo
      if (present(DD)) DD = d
      if (present(H)) H = int(rest / 3600)
      if (present(M)) M = int(mod(rest, 3600_int64) / 60)
      if (present(S)) S = int(mod(rest, 60_int64))
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_TimeIntervalSet(timeinterval, S, rc)
      type(ESMF_TimeInterval), intent(inout) :: timeinterval
      integer, intent(in), optional :: S
      integer, intent(out), optional :: rc
      timeinterval%s = 0
      if (present(S)) timeinterval%s = S
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_TimeIntervalGet(timeinterval, S, rc)
      type(ESMF_TimeInterval), intent(in) :: timeinterval
      integer, intent(out), optional :: S
      integer, intent(out), optional :: rc
      if (present(S)) S = int(timeinterval%s)
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   function ESMF_ClockCreate(timeStep, startTime, rc) result(clock)
      type(ESMF_TimeInterval), intent(in) :: timeStep
      type(ESMF_Time), intent(in) :: startTime
      integer, intent(out), optional :: rc
      type(ESMF_Clock) :: clock
      allocate(clock%p)
      clock%p%now = startTime%s
      clock%p%dt = timeStep%s
      allocate(clock%p%alarms(8))
      call set_rc(rc, ESMF_SUCCESS)
   end function

   subroutine ESMF_ClockGet(clock, currTime, timeStep, rc)
      type(ESMF_Clock), intent(in) :: clock
      type(ESMF_Time), intent(out), optional :: currTime
      type(ESMF_TimeInterval), intent(out), optional :: timeStep
      integer, intent(out), optional :: rc
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(clock%p)) return
      if (present(currTime)) currTime%s = clock%p%now
      if (present(timeStep)) timeStep%s = clock%p%dt
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   !  After the clock has moved from t_old to t_new a non-sticky alarm rings iff one of its ring times lies in
   !  (t_old, t_new]; it rings for that step only, and not at all once someone has turned the ringer off.
   subroutine alarm_update(a, t_old, t_new)
      type(esmfl_alarm_impl), intent(inout) :: a
      integer(int64), intent(in) :: t_old, t_new
      integer(int64) :: k
      if (a%interval <= 0) then
         a%ringing = .true.
         return
      end if
      k = floor_div(t_new - a%first_ring, a%interval)           ! largest ring time <= t_new
      a%ringing = a%first_ring + k * a%interval > t_old
   end subroutine

   integer(int64) function floor_div(a, b) result(q)
      integer(int64), intent(in) :: a, b
      q = a / b
      if (mod(a, b) /= 0 .and. ((a < 0) .neqv. (b < 0))) q = q - 1
   end function

   !  the cap calls it AFTER the run methods; every alarm of the clock is re-evaluated
   subroutine ESMF_ClockAdvance(clock, rc)
      type(ESMF_Clock), intent(inout) :: clock
      integer, intent(out), optional :: rc
      integer(int64) :: before
      integer :: i
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(clock%p)) return
      before = clock%p%now
      clock%p%now = clock%p%now + clock%p%dt
      do i = 1, clock%p%nalarms
         call alarm_update(clock%p%alarms(i)%p, before, clock%p%now)
      end do
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   !  ringInterval absent: rings at every step.  The state at creation is that of a clock that has just stepped
   !  onto its current time ("if (ringTime == currTime) call ESMF_AlarmRingerOn" in MAPL's use of it).
   function ESMF_AlarmCreate(clock, ringTime, ringInterval, sticky, name, rc) result(alarm)
      type(ESMF_Clock), intent(inout) :: clock
      type(ESMF_Time), intent(in), optional :: ringTime
      type(ESMF_TimeInterval), intent(in), optional :: ringInterval
      logical, intent(in), optional :: sticky
      character(len=*), intent(in), optional :: name
      integer, intent(out), optional :: rc
      type(ESMF_Alarm) :: alarm
      type(ESMF_Alarm), allocatable :: grown(:)
      allocate(alarm%p)
      if (present(ringTime)) alarm%p%first_ring = ringTime%s
      if (present(ringInterval)) alarm%p%interval = ringInterval%s
      call alarm_update(alarm%p, clock%p%now - clock%p%dt, clock%p%now)
      if (clock%p%nalarms == size(clock%p%alarms)) then
         allocate(grown(2 * size(clock%p%alarms)))
         grown(1:clock%p%nalarms) = clock%p%alarms(1:clock%p%nalarms)
         call move_alloc(grown, clock%p%alarms)
      end if
      clock%p%nalarms = clock%p%nalarms + 1
      clock%p%alarms(clock%p%nalarms) = alarm
      call set_rc(rc, ESMF_SUCCESS)
   end function

   logical function ESMF_AlarmIsRinging(alarm, rc) result(ringing)
      type(ESMF_Alarm), intent(in) :: alarm
      integer, intent(out), optional :: rc
      ringing = .false.
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(alarm%p)) return
      ringing = alarm%p%ringing
      call set_rc(rc, ESMF_SUCCESS)
   end function

   subroutine ESMF_AlarmRingerOff(alarm, rc)
      type(ESMF_Alarm), intent(inout) :: alarm
      integer, intent(out), optional :: rc
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(alarm%p)) return
      alarm%p%ringing = .false.
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_AlarmRingerOn(alarm, rc)
      type(ESMF_Alarm), intent(inout) :: alarm
      integer, intent(out), optional :: rc
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(alarm%p)) return
      alarm%p%ringing = .true.
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   ! =================================================================== grid and states

   function esmfl_grid_create(im, jm, km, LATS, LONS) result(grid)
      integer, intent(in) :: im, jm, km
      real, intent(in) :: LATS(:,:), LONS(:,:)
      type(ESMF_Grid) :: grid
      allocate(grid%p)
      grid%p%im = im; grid%p%jm = jm; grid%p%km = km
      allocate(grid%p%LATS(im, jm), grid%p%LONS(im, jm))
      grid%p%LATS = LATS
      grid%p%LONS = LONS
   end function

   function ESMF_StateCreate(name, rc) result(state)
      character(len=*), intent(in), optional :: name
      integer, intent(out), optional :: rc
      type(ESMF_State) :: state
      allocate(state%p)
      if (present(name)) state%p%name = name
      allocate(state%p%f(16))
      call set_rc(rc, ESMF_SUCCESS)
   end function

   integer function esmfl_state_index(state, name) result(k)
      type(ESMF_State), intent(in) :: state
      character(len=*), intent(in) :: name
      integer :: i
      k = 0
      if (.not. associated(state%p)) return
      do i = 1, state%p%n
         if (trim(state%p%f(i)%name) == trim(name)) then
            k = i
            return
         end if
      end do
   end function

   !  a spec becomes a field without storage; a second spec of the same name is refused, as MAPL does
   subroutine esmfl_state_add(state, field, rc)
      type(ESMF_State), intent(inout) :: state
      type(esmfl_field), intent(in) :: field
      integer, intent(out) :: rc
      type(esmfl_field), allocatable :: grown(:)
      rc = ESMF_FAILURE
      if (.not. associated(state%p)) return
      if (esmfl_state_index(state, field%name) > 0) return
      associate (st => state%p)
         if (st%n == size(st%f)) then
            allocate(grown(2 * size(st%f)))
            grown(1:st%n) = st%f(1:st%n)
            call move_alloc(grown, st%f)
         end if
         st%n = st%n + 1
         st%f(st%n) = field
      end associate
      rc = ESMF_SUCCESS
   end subroutine

   logical function esmfl_field_has_storage(state, name) result(yes)
      type(ESMF_State), intent(in) :: state
      character(len=*), intent(in) :: name
      integer :: k
      yes = .false.
      k = esmfl_state_index(state, name)
      if (k == 0) return
      yes = associated(state%p%f(k)%p2) .or. associated(state%p%f(k)%p3) .or. associated(state%p%f(k)%p4)
   end function

   !  Gives a declared field its storage (zero-filled): what the rest of GEOS does for a component's imports,
   !  HISTORY for the exports somebody asked for, and MAPL_GenericInitialize for the INTERNAL state.
   !  Edge fields are indexed 0:km (OH_GridCompMod.F90:1246, 1450).
   subroutine esmfl_state_allocate(state, name, grid, rc)
      type(ESMF_State), intent(inout) :: state
      character(len=*), intent(in) :: name
      type(ESMF_Grid), intent(in) :: grid
      integer, intent(out) :: rc
      integer :: k, k0
      rc = ESMF_FAILURE
      k = esmfl_state_index(state, name)
      if (k == 0 .or. .not. associated(grid%p)) return
      rc = ESMF_SUCCESS
      if (esmfl_field_has_storage(state, name)) return
      associate (f => state%p%f(k), g => grid%p)
         if (f%dims == ESMFL_DIMS_HORZ_ONLY) then
            allocate(f%p2(g%im, g%jm))
            f%p2 = 0.0
         else
            k0 = merge(0, 1, f%vloc == ESMFL_VLOC_EDGE)
            if (f%ungridded > 0) then
               allocate(f%p4(g%im, g%jm, k0:g%km, f%ungridded))
               f%p4 = 0.0
            else
               allocate(f%p3(g%im, g%jm, k0:g%km))
               f%p3 = 0.0
            end if
         end if
      end associate
   end subroutine

   ! =================================================================== grid components

   function ESMF_GridCompCreate(name, config, grid, rc) result(gc)
      character(len=*), intent(in) :: name
      type(ESMF_Config), intent(in), optional :: config
      type(ESMF_Grid), intent(in), optional :: grid
      integer, intent(out), optional :: rc
      type(ESMF_GridComp) :: gc
      allocate(gc%p)
      gc%p%name = name
      if (present(config)) gc%p%config = config
      if (present(grid)) gc%p%grid = grid
      gc%p%importState = ESMF_StateCreate(name=trim(name)//'_Imports')
      gc%p%exportState = ESMF_StateCreate(name=trim(name)//'_Exports')
      call set_rc(rc, ESMF_SUCCESS)
   end function

   subroutine ESMF_GridCompGet(gridcomp, name, config, grid, importState, exportState, rc)
      type(ESMF_GridComp), intent(in) :: gridcomp
      character(len=*), intent(out), optional :: name
      type(ESMF_Config), intent(out), optional :: config
      type(ESMF_Grid), intent(out), optional :: grid
      type(ESMF_State), intent(out), optional :: importState, exportState
      integer, intent(out), optional :: rc
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(gridcomp%p)) return
      if (present(name)) name = gridcomp%p%name
      if (present(config)) config = gridcomp%p%config
      if (present(grid)) grid = gridcomp%p%grid
      if (present(importState)) importState = gridcomp%p%importState
      if (present(exportState)) exportState = gridcomp%p%exportState
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   subroutine ESMF_GridCompSetServices(gridcomp, userRoutine, userRc, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      procedure(esmfl_set_services) :: userRoutine
      integer, intent(out), optional :: userRc
      integer, intent(out), optional :: rc
      integer :: status
      call userRoutine(gridcomp, status)
      if (present(userRc)) then
         userRc = status
         call set_rc(rc, ESMF_SUCCESS)
      else
         call set_rc(rc, status)
      end if
   end subroutine

   subroutine ESMF_GridCompSetEntryPoint(gridcomp, methodflag, userRoutine, phase, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      type(ESMF_Method_Flag), intent(in) :: methodflag
      procedure(esmfl_method) :: userRoutine
      integer, intent(in), optional :: phase
      integer, intent(out), optional :: rc
      integer :: ph
      ph = 1
      if (present(phase)) ph = phase
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(gridcomp%p) .or. ph < 1 .or. ph > kMaxPhases) return
      if (methodflag%v == ESMF_METHOD_INITIALIZE%v) then
         gridcomp%p%initialize(ph)%p => userRoutine
      else if (methodflag%v == ESMF_METHOD_RUN%v) then
         gridcomp%p%run(ph)%p => userRoutine
      else
         return                                          ! no Finalize in the mock (nor in the reference, :606)
      end if
      call set_rc(rc, ESMF_SUCCESS)
   end subroutine

   !  the component's states are the default for importState / exportState, as in a GEOS cap
   recursive subroutine ESMF_GridCompInitialize(gridcomp, importState, exportState, clock, phase, userRc, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      type(ESMF_State), intent(inout), optional :: importState, exportState
      type(ESMF_Clock), intent(inout) :: clock
      integer, intent(in), optional :: phase
      integer, intent(out), optional :: userRc, rc
      call invoke(gridcomp, ESMF_METHOD_INITIALIZE, importState, exportState, clock, phase, userRc, rc)
   end subroutine

   recursive subroutine ESMF_GridCompRun(gridcomp, importState, exportState, clock, phase, userRc, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      type(ESMF_State), intent(inout), optional :: importState, exportState
      type(ESMF_Clock), intent(inout) :: clock
      integer, intent(in), optional :: phase
      integer, intent(out), optional :: userRc, rc
      call invoke(gridcomp, ESMF_METHOD_RUN, importState, exportState, clock, phase, userRc, rc)
   end subroutine

   recursive subroutine invoke(gridcomp, method, importState, exportState, clock, phase, userRc, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      type(ESMF_Method_Flag), intent(in) :: method
      type(ESMF_State), intent(inout), optional :: importState, exportState
      type(ESMF_Clock), intent(inout) :: clock
      integer, intent(in), optional :: phase
      integer, intent(out), optional :: userRc, rc
      procedure(esmfl_method), pointer :: proc
      type(ESMF_State) :: imp, ex
      integer :: ph, status
      ph = 1
      if (present(phase)) ph = phase
      if (present(userRc)) userRc = ESMF_SUCCESS
      call set_rc(rc, ESMF_FAILURE)
      if (.not. associated(gridcomp%p) .or. ph < 1 .or. ph > kMaxPhases) return
      if (method%v == ESMF_METHOD_INITIALIZE%v) then
         proc => gridcomp%p%initialize(ph)%p
      else
         proc => gridcomp%p%run(ph)%p
      end if
      if (.not. associated(proc)) return                 ! nothing registered for this method and phase
      imp = gridcomp%p%importState
      ex = gridcomp%p%exportState
      if (present(importState)) imp = importState
      if (present(exportState)) ex = exportState
      call proc(gridcomp, imp, ex, clock, status)
      if (present(userRc)) then
         userRc = status
         call set_rc(rc, ESMF_SUCCESS)
      else
         call set_rc(rc, status)
      end if
   end subroutine

   !  ESMF keeps the caller's `wrap` object (a derived type holding one pointer, OH_GridCompMod.F90:116-118) under
   !  a name and hands the same bytes back.  The pointer inside keeps naming its target: nothing is deep-copied.
   subroutine ESMF_UserCompSetInternalState(gridcomp, name, wrap, rc)
      type(ESMF_GridComp), intent(inout) :: gridcomp
      character(len=*), intent(in) :: name
      class(*), intent(in), target :: wrap
      integer, intent(out) :: rc
      integer :: n
      rc = ESMF_FAILURE
      if (.not. associated(gridcomp%p)) return
      n = storage_size(wrap) / 8
      if (allocated(gridcomp%p%user_state)) deallocate(gridcomp%p%user_state)
      allocate(gridcomp%p%user_state(n))
      call save_bytes(gridcomp%p%user_state, wrap)
      gridcomp%p%user_state_name = name
      rc = ESMF_SUCCESS
   end subroutine

   subroutine ESMF_UserCompGetInternalState(gridcomp, name, wrap, rc)
      type(ESMF_GridComp), intent(in) :: gridcomp
      character(len=*), intent(in) :: name
      class(*), intent(inout), target :: wrap
      integer, intent(out) :: rc
      integer :: n
      rc = ESMF_FAILURE
      if (.not. associated(gridcomp%p)) return
      if (.not. allocated(gridcomp%p%user_state) .or. trim(gridcomp%p%user_state_name) /= trim(name)) return
      n = storage_size(wrap) / 8
      if (n /= size(gridcomp%p%user_state)) return
      call restore_bytes(wrap, gridcomp%p%user_state)
      rc = ESMF_SUCCESS
   end subroutine

   subroutine save_bytes(store, obj)
      integer(c_int8_t), intent(out) :: store(:)
      type(*), intent(in), target :: obj
      integer(c_int8_t), pointer :: b(:)
      call c_f_pointer(c_loc(obj), b, [size(store)])
      store(:) = b(:)
   end subroutine

   subroutine restore_bytes(obj, store)
      type(*), intent(inout), target :: obj
      integer(c_int8_t), intent(in) :: store(:)
      integer(c_int8_t), pointer :: b(:)
      call c_f_pointer(c_loc(obj), b, [size(store)])
      b(:) = store(:)
   end subroutine

end module ESMF
