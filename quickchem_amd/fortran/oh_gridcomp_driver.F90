!  oh_gridcomp_driver -- a mock GEOS cap around QuickChem: BASELINE.json config #1, "a synthetic MAPL state
!  through OH_GridComp Run".  SetServices of the parent (which creates the OH instances), Initialize, then
!  `nticks` heartbeats of Run phase 1, Run phase 2 and clock advance, on mapl_lite.  No MAPL, no ESMF.
!
!  usage: oh_gridcomp_driver <run dir> <state.bin> <out.bin> <nticks>
!
!  <run dir> holds the resource files the components read: AGCM.rc (RUN_DT, QUICKCHEM_DT, OH_DT,
!  OH_REFERENCE_TIME, and for this driver BEG_DATE: yyyymmdd hhmmss, OH_EXPORTS: names HISTORY would ask for,
!  AVG24_READY_TICK: the heartbeat before which the *_avg24 imports become valid, -1 = never),
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
   use mapl_lite
   use QuickChem_GridCompMod, only: QuickChem_SetServices => SetServices, IS_QC_INSTANCE_RUNNING
   use OH_GridCompMod, only: oh_last_run
   implicit none

   character(len=ML_MAXPATH) :: rundir, state_file, out_file, arg, model_file
   character(len=32) :: recname
   character(len=ML_MAXSTR) :: tok
   character(len=ML_MAXSTR), allocatable :: want(:)
   type(ml_config), pointer :: agcm
   type(ml_gridcomp), pointer :: root, child
   type(ml_grid) :: grid
   type(ml_clock) :: clock
   integer(c_int32_t) :: im, jm, km, n4, nrec, kind
   integer :: rc, u, uo, nticks, tick, i, c, q, beg_date, beg_time, dt, nymd, nhms, yy, mm, dd, h, m, s, nwant, trc
   integer :: avg24_tick, k1, k2, ninst
   logical :: ran, boosted, running
   real, pointer :: p2(:,:), p3(:,:,:), p4(:,:,:,:), oh(:,:,:), parent_oh(:,:,:)
   real, allocatable :: buf(:)

   if (command_argument_count() < 4) then
      print *, 'usage: oh_gridcomp_driver <run dir> <state.bin> <out.bin> <nticks>'
      stop 2
   end if
   call get_command_argument(1, rundir)
   call get_command_argument(2, state_file)
   call get_command_argument(3, out_file)
   call get_command_argument(4, arg)
   read(arg, *) nticks

   allocate(agcm)
   call agcm%load(trim(rundir)//'/AGCM.rc', rc)
   if (rc /= ML_SUCCESS) call die('cannot read AGCM.rc in '//trim(rundir))
   call agcm%get_int(dt, 'RUN_DT:', rc, default=450)
   call agcm%find_label('BEG_DATE:', rc)
   if (rc /= ML_SUCCESS) call die('AGCM.rc: BEG_DATE: yyyymmdd hhmmss is missing')
   call agcm%next_token(tok, rc); read(tok, *) beg_date
   call agcm%next_token(tok, rc); read(tok, *) beg_time
   call agcm%get_int(avg24_tick, 'AVG24_READY_TICK:', rc, default=-1)
   nwant = agcm%get_len('OH_EXPORTS:', rc)
   if (rc /= ML_SUCCESS) nwant = 0
   allocate(want(max(nwant, 0)))
   if (nwant > 0) then
      call agcm%find_label('OH_EXPORTS:', rc)
      do i = 1, nwant
         call agcm%next_token(want(i), rc)
      end do
   end if

   open(newunit=u, file=trim(state_file), access='stream', form='unformatted', status='old', action='read')
   read(u) im, jm, km, n4, nrec
   grid%im = im; grid%jm = jm; grid%km = km
   allocate(grid%LATS(im, jm), grid%LONS(im, jm))
   read(u) grid%LATS
   read(u) grid%LONS

   call clock%set(beg_date / 10000, mod(beg_date, 10000) / 100, mod(beg_date, 100), &
                  beg_time / 10000, mod(beg_time, 10000) / 100, mod(beg_time, 100), dt)

   root => ml_gridcomp_create('QUICKCHEM', agcm, grid, trim(rundir))
   call QuickChem_SetServices(root, rc)
   if (rc /= ML_SUCCESS) call die('QuickChem SetServices failed')
   call IS_QC_INSTANCE_RUNNING('OH', root%children(1)%gc%name, running, rc, rc_dir=trim(rundir))
   if (rc /= ML_SUCCESS .or. .not. running) call die('IS_QC_INSTANCE_RUNNING does not know the first OH instance')

   !  the rest of GEOS: storage for every import, filled from the state file by name; HISTORY: the exports asked for
   do c = 1, root%nchildren
      child => root%children(c)%gc
      do i = 1, child%import%n
         call child%import%allocate_field(child%import%f(i)%name, grid, rc)
      end do
      if (index(child%name, 'data') == 0) then
         do i = 1, nwant
            call child%export%allocate_field(trim(want(i)), grid, rc)
            if (rc /= ML_SUCCESS) call die('OH_EXPORTS names an export OH does not have: '//trim(want(i)))
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
      do c = 1, root%nchildren
         child => root%children(c)%gc
         if (.not. child%import%has(trim(recname))) cycle
         select case (kind)
         case (2)
            call child%import%get_pointer(p2, trim(recname), rc)
            if (rc == ML_SUCCESS) p2 = reshape(buf, shape(p2))
         case (3, 4)
            call child%import%get_pointer(p3, trim(recname), rc)
            if (rc == ML_SUCCESS) p3 = reshape(buf, shape(p3))
         case (5)
            call child%import%get_pointer(p4, trim(recname), rc)
            if (rc == ML_SUCCESS) p4 = reshape(buf, shape(p4))
         end select
         if (rc /= ML_SUCCESS) call die('state file: record '//trim(recname)//' does not fit the import of that name')
      end do
      deallocate(buf)
   end do
   close(u)

   call ml_gridcomp_initialize(root, clock, rc)
   if (rc /= ML_SUCCESS) call die('Initialize failed')

   ninst = root%nchildren
   open(newunit=uo, file=trim(out_file), access='stream', form='unformatted', status='replace', action='write')
   write(uo) int(nticks, c_int32_t), int(ninst, c_int32_t), int(nwant, c_int32_t)
   do tick = 0, nticks - 1
      if (tick > 0) call model_moves()
      if (tick == avg24_tick) call daily_means_arrive()
      call ml_gridcomp_run(root, clock, 1, rc)
      if (rc /= ML_SUCCESS) call die('Run phase 1 failed')
      call ml_gridcomp_run(root, clock, 2, rc)
      if (rc /= ML_SUCCESS) call die('Run phase 2 failed')
      call clock%get(yy, mm, dd, h, m, s)
      call ml_pack_time(nymd, yy, mm, dd)
      call ml_pack_time(nhms, h, m, s)
      write(uo) int(tick, c_int32_t), int(nymd, c_int32_t), int(nhms, c_int32_t)
      do c = 1, root%nchildren
         child => root%children(c)%gc
         call child%internal%get_pointer(oh, 'OH', rc)
         if (index(child%name, 'data') > 0) then
            write(uo) oh
            cycle
         end if
         call oh_last_run(child, ran, boosted, model_file, k1, k2)
         write(uo) merge(1_c_int32_t, 0_c_int32_t, ran), merge(1_c_int32_t, 0_c_int32_t, boosted), &
                   int(k1, c_int32_t), int(k2, c_int32_t)
         write(uo) model_file(1:256)
         write(uo) oh
         !  what GEOS_ChemGridComp's other children would connect to: the parent's export OH
         call ml_child_export_field(root, 'OH', parent_oh, trc)
         write(uo) merge(1_c_int32_t, 0_c_int32_t, trc == ML_SUCCESS .and. c == 1 .and. associated(parent_oh, oh))
         do i = 1, nwant
            q = child%export%index_of(trim(want(i)))
            if (child%export%f(q)%dims == ML_DIMS_HORZ_ONLY) then
               write(uo) child%export%f(q)%p2
            else
               write(uo) child%export%f(q)%p3
            end if
         end do
      end do
      call ml_advance(root, clock)
   end do
   close(uo)

contains

   subroutine die(msg)
      character(len=*), intent(in) :: msg
      print '(a)', 'oh_gridcomp_driver: '//msg
      stop 1
   end subroutine

   !  the model state of every computational instance drifts between heartbeats
   subroutine model_moves()
      integer :: cc, r
      real, pointer :: t(:,:,:), tp(:,:)
      do cc = 1, root%nchildren
         if (index(root%children(cc)%gc%name, 'data') > 0) cycle
         call root%children(cc)%gc%import%get_pointer(t, 'T', r)
         if (r == ML_SUCCESS .and. associated(t)) t = t * 1.0005
         call root%children(cc)%gc%import%get_pointer(tp, 'TROPP', r)
         if (r == ML_SUCCESS .and. associated(tp)) tp = tp * 1.002
      end do
   end subroutine

   !  the couplers deliver the first complete daily means: X_avg24 = X * 0.99 for every import that has one
   subroutine daily_means_arrive()
      integer :: cc, ii, r, n
      character(len=ML_MAXSTR) :: base
      real, pointer :: a3(:,:,:), b3(:,:,:), a4(:,:,:,:), b4(:,:,:,:)
      do cc = 1, root%nchildren
         associate (imp => root%children(cc)%gc%import)
            do ii = 1, imp%n
               n = len_trim(imp%f(ii)%name)
               if (n <= 6) cycle
               if (imp%f(ii)%name(n-5:n) /= '_avg24') cycle
               base = imp%f(ii)%name(1:n-6)
               if (.not. imp%has(trim(base))) cycle
               if (imp%f(ii)%ungridded > 0) then
                  call imp%get_pointer(a4, trim(imp%f(ii)%name), r)
                  call imp%get_pointer(b4, trim(base), r)
                  a4 = b4 * 0.99
               else
                  call imp%get_pointer(a3, trim(imp%f(ii)%name), r)
                  call imp%get_pointer(b3, trim(base), r)
                  a3 = b3 * 0.99
               end if
            end do
         end associate
      end do
   end subroutine

end program oh_gridcomp_driver
