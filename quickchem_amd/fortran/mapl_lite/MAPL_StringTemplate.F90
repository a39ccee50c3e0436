!  mapl_lite/MAPL_StringTemplate.F90 -- MAPL's module of that name, as far as QuickChem uses it: the reference's OH
!  child says `use MAPL_StringTemplate` and expands the month-templated model file name with
!     call fill_grads_template(XGBoostFilename, self%XGBoostFilePattern, nymd=nymd, nhms=nhms, __RC__)
!  (OH_GridComp/OH_GridCompMod.F90:19,1187; the pattern is OH_instance_OH.rc:20, "...%m2...").  Part of the mock
!  (see ESMF.F90 in this directory): a GEOS build has the real module and does not compile this file.
!  Restated: the GrADS tokens a file name can hold - %y4 %y2 %m2 %m1 %d2 %d1 %h2 %h1 %n2 and %s (experiment id);
!  the optional arguments MAPL has and QuickChem does not pass (time=, preserve=) are not here.
module MAPL_StringTemplate
   use ESMF
   implicit none
   private
   public :: fill_grads_template
contains
   subroutine fill_grads_template(output_string, template, unusable, experiment_id, nymd, nhms, rc)
      character(len=*), intent(out) :: output_string
      character(len=*), intent(in) :: template
      logical, intent(in), optional :: unusable          ! MAPL's KeywordEnforcer slot: what follows goes by keyword
      character(len=*), intent(in), optional :: experiment_id
      integer, intent(in), optional :: nymd, nhms
      integer, intent(out), optional :: rc
      character(len=:), allocatable :: name
      character(len=8) :: buf
      integer :: p, from, ymd, hms, width
      ymd = 0
      hms = 0
      if (present(nymd)) ymd = nymd
      if (present(nhms)) hms = nhms
      name = trim(template)
      from = 1
      do
         if (from > len(name)) exit
         p = index(name(from:), '%')
         if (p == 0) exit
         p = p + from - 1
         if (p + 1 > len(name)) exit
         width = 3
         buf = ''
         if (name(p+1:p+1) == 's') then
            width = 2
            if (present(experiment_id)) then
               name = name(:p-1)//trim(experiment_id)//name(p+2:)
               from = p + len_trim(experiment_id)
            else
               from = p + 2
            end if
            cycle
         end if
         if (p + 2 > len(name)) exit
         select case (name(p+1:p+2))
         case ('y4'); write(buf, '(i4.4)') ymd / 10000
         case ('y2'); write(buf, '(i2.2)') mod(ymd / 10000, 100)
         case ('m2'); write(buf, '(i2.2)') mod(ymd, 10000) / 100
         case ('m1'); write(buf, '(i0)') mod(ymd, 10000) / 100
         case ('d2'); write(buf, '(i2.2)') mod(ymd, 100)
         case ('d1'); write(buf, '(i0)') mod(ymd, 100)
         case ('h2'); write(buf, '(i2.2)') hms / 10000
         case ('h1'); write(buf, '(i0)') hms / 10000
         case ('n2'); write(buf, '(i2.2)') mod(hms, 10000) / 100
         case default
            from = p + 1                 ! not a token: the % stays
            cycle
         end select
         name = name(:p-1)//trim(buf)//name(p+width:)
         from = p + len_trim(buf)
      end do
      output_string = name
      if (present(rc)) rc = ESMF_SUCCESS
   end subroutine
end module MAPL_StringTemplate
