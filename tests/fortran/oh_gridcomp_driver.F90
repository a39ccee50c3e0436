!  oh_gridcomp_driver -- a mock GEOS cap around QuickChem: BASELINE.json config #1, "a synthetic MAPL state
!  through OH_GridComp Run".  SetServices of the parent (which creates the OH instances), Initialize, then
!  `nticks` heartbeats of Run phase 1, Run phase 2 and clock advance, through the ESMF / MAPL calls a cap makes, served by
!  the mock in quickchem_amd/fortran/mapl_lite/ (no MAPL, no ESMF here).  The parent, QuickChem_GridCompMod, is the
!  reference's own unmodified file, compiled in place by oracle/Makefile (target `ref`); the child is the product's -
!  (built with -DOHX_STANDALONE_CAP the parent is the product's own quickchem_amd/fortran/oh_standalone_cap.F90: the drivers
!  quickchem_amd/lib/oh_gridcomp_driver_hip and oracle/lib/oh_gridcomp_driver_oracle, which need no reference tree)
!  or, built with -DOHX_REFERENCE_CHILD, the reference's own OH_GridComp/OH_GridCompMod.F90, also compiled in place (the
!  drivers oracle/_ref/refchild/oh_refchild_driver_*: the reference's lines as the checker of the product's).  Being the mock's cap it also plays "the rest of GEOS": storage for
!  the imports, HISTORY's wish list of exports, the model moving between heartbeats (esmfl_ / mapll_ calls).
!
!  usage: oh_gridcomp_driver <run dir> <state.bin> <out.bin> <nticks>
!
!  <run dir> holds the resource files the components read: AGCM.rc (RUN_DT, QUICKCHEM_DT, OH_DT,
!  OH_REFERENCE_TIME, and for this driver BEG_DATE: yyyymmdd hhmmss, OH_EXPORTS: names HISTORY would ask for,
!  AVG24_READY_TICK: the heartbeat before which the *_avg24 imports become valid, -1 = never; SPEC_DUMP: a file that
!  receives what every instance's SetServices registered),
!  QuickChem_GridComp.rc, OH_instance_<name>.rc, GOCART2G_GridComp.rc.
!
!  state.bin (stream, little endian, written by tests/):
!     int32 im, jm, km, n4, nrec;  real32 LATS(im,jm), LONS(im,jm)   [radians]
!     nrec records:  character(32) name; int32 kind (2 = (im,jm), 3 = (im,jm,km), 4 = (im,jm,0:km),
!                    5 = (im,jm,km,n4));  real32 data
!     every import of every instance that has a record of its name is filled from it; the others stay zero
!  Between heartbeats the "model" moves a little, deterministically: T *= 1.0005, TROPP *= 1.002, so that a
!  tick that skips Boost still sees a new tropopause mask and number density.
!
!  out.bin: int32 nticks, ninst, nexports;  per tick: int32 tick, nymd, nhms; per computational instance:
!     int32 ran, called_boost, k1, k2; character(256) model file; real32 INTERNAL OH(im,jm,km);
!     int32 parent_export_is_first_childs_OH; the requested exports in OH_EXPORTS order;
!  per data instance: real32 INTERNAL OH(im,jm,km).
program oh_gridcomp_driver
   use, intrinsic :: iso_c_binding
   use ESMF
   use MAPL
#ifdef OHX_STANDALONE_CAP
   use OH_StandaloneCap, only: QuickChem_SetServices => SetServices      ! the product's own minimal parent
#else
   use QuickChem_GridCompMod, only: QuickChem_SetServices => SetServices, IS_QC_INSTANCE_RUNNING
#endif
#ifndef OHX_REFERENCE_CHILD
   use OH_GridCompMod, only: oh_last_run
#endif
   implicit none

   character(len=ESMF_MAXPATHLEN) :: rundir, state_file, out_file, arg, model_file, spec_file
   character(len=32) :: recname
   character(len=ESMF_MAXSTR) :: tok, child_name
   character(len=ESMF_MAXSTR), allocatable :: want(:)
   type(ESMF_Config) :: agcm
   type(ESMF_GridComp) :: root
   type(ESMF_GridComp), pointer :: gcs(:)
   type(ESMF_State), pointer :: gim(:), gex(:)
   type(ESMF_State) :: internal
   type(MAPL_MetaComp), pointer :: meta, cmeta
   type(ESMF_Grid) :: grid
   type(ESMF_Clock) :: clock
   type(ESMF_Time) :: start, now
   type(ESMF_TimeInterval) :: heartbeat
   integer(c_int32_t) :: im, jm, km, n4, nrec, kind
   integer :: rc, u, uo, us, nticks, tick, i, c, q, beg_date, beg_time, dt, nymd, nhms, yy, mm, dd, h, m, s, nwant, trc
   integer :: avg24_tick, k1, k2, ninst
   integer(kind=8) :: clock0, clock1, clock_rate
   character(len=16) :: timing
   character(len=32) :: meet_at
   real(kind=8) :: meet_sod
   logical :: ran, boosted, running
   real, pointer :: p2(:,:), p3(:,:,:), p4(:,:,:,:), oh(:,:,:), parent_oh(:,:,:)
   real, allocatable :: buf(:), lats(:,:), lons(:,:)

   if (command_argument_count() < 4) then
      print *, 'usage: oh_gridcomp_driver <run dir> <state.bin> <out.bin> <nticks>'
      stop 2
   end if
   call get_command_argument(1, rundir)
   call get_command_argument(2, state_file)
   call get_command_argument(3, out_file)
   call get_command_argument(4, arg)
   read(arg, *) nticks

   call esmfl_set_run_dir(trim(rundir))                 ! GEOS runs in its run directory: resource files are found there
   agcm = ESMF_ConfigCreate(rc=rc)
   call ESMF_ConfigLoadFile(agcm, 'AGCM.rc', rc=rc)
   if (rc /= ESMF_SUCCESS) call die('cannot read AGCM.rc in '//trim(rundir))
   call ESMF_ConfigGetAttribute(agcm, dt, label='RUN_DT:', default=450, rc=rc)
   call ESMF_ConfigFindLabel(agcm, 'BEG_DATE:', rc=rc)
   if (rc /= ESMF_SUCCESS) call die('AGCM.rc: BEG_DATE: yyyymmdd hhmmss is missing')
   call ESMF_ConfigGetAttribute(agcm, beg_date, rc=rc)
   call ESMF_ConfigGetAttribute(agcm, beg_time, rc=rc)
   call ESMF_ConfigGetAttribute(agcm, avg24_tick, label='AVG24_READY_TICK:', default=-1, rc=rc)
   nwant = ESMF_ConfigGetLen(agcm, label='OH_EXPORTS:', rc=rc)
   if (rc /= ESMF_SUCCESS) nwant = 0
   allocate(want(max(nwant, 0)))
   if (nwant > 0) then
      call ESMF_ConfigFindLabel(agcm, 'OH_EXPORTS:', rc=rc)
      do i = 1, nwant
         call ESMF_ConfigGetAttribute(agcm, want(i), rc=rc)
      end do
   end if

   open(newunit=u, file=trim(state_file), access='stream', form='unformatted', status='old', action='read')
   read(u) im, jm, km, n4, nrec
   allocate(lats(im, jm), lons(im, jm))
   read(u) lats
   read(u) lons
   grid = esmfl_grid_create(int(im), int(jm), int(km), lats, lons)

   call ESMF_TimeSet(start, YY=beg_date / 10000, MM=mod(beg_date, 10000) / 100, DD=mod(beg_date, 100), &
                     H=beg_time / 10000, M=mod(beg_time, 10000) / 100, S=mod(beg_time, 100))
   call ESMF_TimeIntervalSet(heartbeat, S=dt)
   clock = ESMF_ClockCreate(timeStep=heartbeat, startTime=start, rc=rc)

   root = ESMF_GridCompCreate(name='QUICKCHEM', config=agcm, grid=grid, rc=rc)
   call set_services_of(root, QuickChem_SetServices, rc)
   if (rc /= ESMF_SUCCESS) call die('QuickChem SetServices failed')
   call MAPL_GetObjectFromGC(root, meta, rc)
   call MAPL_Get(meta, gcs=gcs, gim=gim, gex=gex, rc=rc)
   call ESMF_GridCompGet(gcs(1), name=child_name)
#ifndef OHX_STANDALONE_CAP
   call IS_QC_INSTANCE_RUNNING('OH', trim(child_name), running, rc)
   if (rc /= ESMF_SUCCESS .or. .not. running) call die('IS_QC_INSTANCE_RUNNING does not know the first OH instance')
#endif

   !  AGCM.rc `SPEC_DUMP: <file>`: what every instance's SetServices registered, one line per field -
   !  instance|state|short name|dims|vlocation|restart|refresh|averaging|ungridded|add2export|units|long name
   call ESMF_ConfigGetAttribute(agcm, spec_file, label='SPEC_DUMP:', default='', rc=rc)
   if (len_trim(spec_file) > 0) then
      open(newunit=us, file=trim(spec_file), status='replace', action='write')
      do c = 1, size(gcs)
         call ESMF_GridCompGet(gcs(c), name=child_name)
         call MAPL_GetObjectFromGC(gcs(c), cmeta, rc)
         call MAPL_Get(cmeta, INTERNAL_ESMF_STATE=internal, rc=rc)
         call dump_state(us, trim(child_name), 'IMPORT', gim(c))
         call dump_state(us, trim(child_name), 'EXPORT', gex(c))
         call dump_state(us, trim(child_name), 'INTERNAL', internal)
      end do
      close(us)
   end if

   !  the rest of GEOS: storage for every import, filled from the state file by name; HISTORY: the exports asked for
   do c = 1, size(gcs)
      do i = 1, gim(c)%p%n
         call esmfl_state_allocate(gim(c), gim(c)%p%f(i)%name, grid, rc)
      end do
      if (.not. is_data(c)) then
         do i = 1, nwant
            call esmfl_state_allocate(gex(c), trim(want(i)), grid, rc)
            if (rc /= ESMF_SUCCESS) call die('OH_EXPORTS names an export OH does not have: '//trim(want(i)))
         end do
      end if
   end do
   do q = 1, nrec
      read(u) recname, kind
      select case (kind)
      case (2); allocate(buf(im * jm))
      case (3); allocate(buf(im * jm * km))
      case (4); allocate(buf(im * jm * (km + 1)))
      case (5); allocate(buf(im * jm * km * n4))
      case default; call die('state file: unknown record kind')
      end select
      read(u) buf
      do c = 1, size(gcs)
         if (esmfl_state_index(gim(c), trim(recname)) == 0) cycle
         select case (kind)
         case (2)
            call MAPL_GetPointer(gim(c), p2, trim(recname), rc=rc)
            if (rc == ESMF_SUCCESS) p2 = reshape(buf, shape(p2))
         case (3, 4)
            call MAPL_GetPointer(gim(c), p3, trim(recname), rc=rc)
            if (rc == ESMF_SUCCESS) p3 = reshape(buf, shape(p3))
         case (5)
            call MAPL_GetPointer(gim(c), p4, trim(recname), rc=rc)
            if (rc == ESMF_SUCCESS) p4 = reshape(buf, shape(p4))
         end select
         if (rc /= ESMF_SUCCESS) call die('state file: record '//trim(recname)//' does not fit the import of that name')
      end do
      deallocate(buf)
   end do
   close(u)

   call ESMF_GridCompInitialize(root, clock=clock, rc=rc)
   if (rc /= ESMF_SUCCESS) call die('Initialize failed')

   ninst = size(gcs)
   open(newunit=uo, file=trim(out_file), access='stream', form='unformatted', status='replace', action='write')
   write(uo) int(nticks, c_int32_t), int(ninst, c_int32_t), int(nwant, c_int32_t)
   !  OHX_DRIVER_TIMING in the environment: the wall time of every tick's two run phases on stdout, "TICK_US <tick> <us> <nhms>"
   call get_environment_variable('OHX_DRIVER_TIMING', timing)
   !  OHX_DRIVER_MEET_AT=<seconds of the day, UTC>: after tick 0 (which loads the model) wait for that time - several
   !  drivers started together then tick together, as the ranks of a model do
   call get_environment_variable('OHX_DRIVER_MEET_AT', meet_at)
   meet_sod = -1.0d0
   if (len_trim(meet_at) > 0) read(meet_at, *) meet_sod
   do tick = 0, nticks - 1
      if (tick > 0) call model_moves()
      if (tick == 1 .and. meet_sod >= 0.0d0) call wait_until(meet_sod)
      if (tick == avg24_tick) call daily_means_arrive()
      call system_clock(clock0, clock_rate)
      call ESMF_GridCompRun(root, clock=clock, phase=1, rc=rc)
      if (rc /= ESMF_SUCCESS) call die('Run phase 1 failed')
      call ESMF_GridCompRun(root, clock=clock, phase=2, rc=rc)
      if (rc /= ESMF_SUCCESS) call die('Run phase 2 failed')
      call system_clock(clock1)
      call ESMF_ClockGet(clock, currTime=now)
      call ESMF_TimeGet(now, YY=yy, MM=mm, DD=dd, H=h, M=m, S=s)
      call MAPL_PackTime(nymd, yy, mm, dd)
      call MAPL_PackTime(nhms, h, m, s)
      if (len_trim(timing) > 0) print '(a,i0,1x,f0.1,1x,i0)', 'TICK_US ', tick, real(clock1 - clock0, 8) * 1.0d6 / real(clock_rate, 8), nhms
      write(uo) int(tick, c_int32_t), int(nymd, c_int32_t), int(nhms, c_int32_t)
      do c = 1, size(gcs)
         call MAPL_GetObjectFromGC(gcs(c), cmeta, rc)
         call MAPL_Get(cmeta, INTERNAL_ESMF_STATE=internal, rc=rc)
         if (is_data(c)) then                   ! INTERNAL OH of a data instance: (im,jm,km,nbins), nbins = 1
            call MAPL_GetPointer(internal, p4, 'OH', rc=rc)
            write(uo) p4(:,:,:,1)
            cycle
         end if
         call MAPL_GetPointer(internal, oh, 'OH', rc=rc)
#ifdef OHX_REFERENCE_CHILD
         !  the reference's child says nothing about its last tick: -1 = not known (the test works it out)
         write(uo) -1_c_int32_t, -1_c_int32_t, -1_c_int32_t, -1_c_int32_t
         model_file = ''
#else
         call oh_last_run(gcs(c), ran, boosted, model_file, k1, k2)
         write(uo) merge(1_c_int32_t, 0_c_int32_t, ran), merge(1_c_int32_t, 0_c_int32_t, boosted), &
                   int(k1, c_int32_t), int(k2, c_int32_t)
#endif
         write(uo) model_file(1:256)
         write(uo) oh
         !  what GEOS_ChemGridComp's other children would connect to: the parent's export OH
         call mapll_child_export_field(root, 'OH', parent_oh, trc)
         write(uo) merge(1_c_int32_t, 0_c_int32_t, trc == ESMF_SUCCESS .and. c == 1 .and. associated(parent_oh, oh))
         do i = 1, nwant
            q = esmfl_state_index(gex(c), trim(want(i)))
            if (gex(c)%p%f(q)%dims == MAPL_DimsHorzOnly) then
               write(uo) gex(c)%p%f(q)%p2
            else
               write(uo) gex(c)%p%f(q)%p3
            end if
         end do
      end do
      call ESMF_ClockAdvance(clock, rc=rc)             ! AFTER the run methods; re-evaluates every run alarm
   end do
   close(uo)

contains

   subroutine wait_until(sod)
      real(kind=8), intent(in) :: sod
      integer :: v(8)
      do
         call date_and_time(values=v)
         if (real(v(5) * 3600 + v(6) * 60 + v(7), 8) + real(v(8), 8) * 1.0d-3 - real(v(4), 8) * 60.0d0 >= sod) exit
      end do
   end subroutine

   !  The parent's SetServices is the reference's own routine, whose RC is OPTIONAL (QuickChem_GridCompMod.F90:78-83);
   !  GEOS hands such routines on as `external` procedures (as the parent itself does with its children's, :516-531)
   subroutine set_services_of(gc, ss, rc)
      type(ESMF_GridComp), intent(inout) :: gc
      external :: ss
      integer, intent(out) :: rc
      call ESMF_GridCompSetServices(gc, ss, rc=rc)
   end subroutine

   subroutine dump_state(unit, inst, which, state)
      integer, intent(in) :: unit
      character(len=*), intent(in) :: inst, which
      type(ESMF_State), intent(in) :: state
      integer :: n
      if (.not. associated(state%p)) return
      do n = 1, state%p%n
         associate (f => state%p%f(n))
            write(unit, '(a,"|",a,"|",a,"|",i0,"|",i0,"|",i0,"|",i0,"|",i0,"|",i0,"|",l1,"|",a,"|",a)') inst, which, &
               trim(f%name), f%dims, f%vloc, f%restart, f%refresh_interval, f%averaging_interval, f%ungridded, &
               f%add2export, trim(f%units), trim(f%long_name)
         end associate
      end do
   end subroutine

   subroutine die(msg)
      character(len=*), intent(in) :: msg
      print '(a)', 'oh_gridcomp_driver: '//msg
      stop 1
   end subroutine

   logical function is_data(cc)
      integer, intent(in) :: cc
      character(len=ESMF_MAXSTR) :: nm
      call ESMF_GridCompGet(gcs(cc), name=nm)
      is_data = index(nm, 'data') > 0
   end function

   !  the model state of every computational instance drifts between heartbeats
   subroutine model_moves()
      integer :: cc, r
      real, pointer :: t(:,:,:), tp(:,:)
      do cc = 1, size(gcs)
         if (is_data(cc)) cycle
         call MAPL_GetPointer(gim(cc), t, 'T', rc=r)
         if (r == ESMF_SUCCESS .and. associated(t)) t = t * 1.0005
         call MAPL_GetPointer(gim(cc), tp, 'TROPP', rc=r)
         if (r == ESMF_SUCCESS .and. associated(tp)) tp = tp * 1.002
      end do
   end subroutine

   !  the couplers deliver the first complete daily means: X_avg24 = X * 0.99 for every import that has one
   subroutine daily_means_arrive()
      integer :: cc, ii, r, n
      character(len=ESMF_MAXSTR) :: base, name
      real, pointer :: a3(:,:,:), b3(:,:,:), a4(:,:,:,:), b4(:,:,:,:)
      do cc = 1, size(gcs)
         do ii = 1, gim(cc)%p%n
            name = gim(cc)%p%f(ii)%name
            n = len_trim(name)
            if (n <= 6) cycle
            if (name(n-5:n) /= '_avg24') cycle
            base = name(1:n-6)
            if (esmfl_state_index(gim(cc), trim(base)) == 0) cycle
            if (gim(cc)%p%f(ii)%ungridded > 0) then
               call MAPL_GetPointer(gim(cc), a4, trim(name), rc=r)
               call MAPL_GetPointer(gim(cc), b4, trim(base), rc=r)
               a4 = b4 * 0.99
            else
               call MAPL_GetPointer(gim(cc), a3, trim(name), rc=r)
               call MAPL_GetPointer(gim(cc), b3, trim(base), rc=r)
               a3 = b3 * 0.99
            end if
         end do
      end do
   end subroutine

end program oh_gridcomp_driver
