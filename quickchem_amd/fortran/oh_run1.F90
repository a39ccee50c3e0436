!  oh_run1 -- Fortran host side of OHXBoosterRun1 (include/ohxgb.h part 3): the arithmetic of
!  OH Run1 from the imports to the INTERNAL field OH in one device-resident pass
!  (OH_GridComp/OH_GridCompMod.F90:1240-1257, 1444-1478, 1488, 1557-1595), plus the solar geometry
!  of :401-466, 1444, 1481-1482.  Variable names are the reference's.  MAPL is not needed: its
!  constants are arguments, and where the reference uses _ASSERT this module sets rc and an error
!  text (oh_xgb_error_text of module oh_xgb_predict).
module oh_run1
   use, intrinsic :: iso_c_binding
   use ohx_bindings, only: ohx_last_error, XGBoosterSetParam
   use oh_xgb_predict, only: oh_xgb_booster, OH_XGB_SUCCESS, OH_XGB_FAILURE
   implicit none
   private

   public :: OH_RUN1_STATE, OH_RUN1_DIAG, oh_run1_boost, oh_post_process, oh_solar_geometry, oh_julian_day
   public :: oh_solar_geometry_host, oh_post_process_host
   public :: oh_run1_error_text, oh_run1_register_host_arrays

   !  struct OHXRun1Args, member for member
   type, bind(C) :: OHXRun1Args
      integer(c_int32_t) :: im, jm, km
      integer(c_int32_t) :: dynamic_k_range
      real(c_float)      :: tropp_min, ohscale, missing
      real(c_float)      :: avogad, runiv, epsilon
      type(c_ptr) :: ple_mod, t_mod, q_mod, tropp_mod
      type(c_ptr) :: ple_bst, zle_bst, tauclw, taucli
      type(c_ptr) :: scacoef(7)
      type(c_ptr) :: gmito3, gmitto3
      type(c_ptr) :: lat_deg, t_bst, no2, o3, ch4, co, isop, acet, c2h6, c3h8, prpe, alk4, mp, h2o2
      type(c_ptr) :: cloud, qv, albuv, ch2o, sza
      type(c_ptr) :: default_oh
      type(c_ptr) :: oh, oh_boost, ndwet
      type(c_ptr) :: k1, k2
      type(c_ptr) :: diag_pl_bst, diag_tauclwdn, diag_tauclidn, diag_taucliup, diag_tauclwup
      type(c_ptr) :: diag_aodup, diag_aoddn, diag_aod, diag_strato3
   end type

   !  What Run1 has in hand when it reaches CALL_BOOST (:1557), by the reference's names.
   !  Edge fields are (im,jm,0:km); scacoef are BC OC BR DU SU SS NI at the chosen wavelength (:1451-1458).
   type OH_RUN1_STATE
      real, pointer, dimension(:,:,:) :: PLE_MOD => null(), T_MOD => null(), Q_MOD => null()
      real, pointer, dimension(:,:)   :: TROPP_MOD => null()
      real, pointer, dimension(:,:,:) :: PLE_BST => null(), ZLE_BST => null(), TAUCLW => null(), TAUCLI => null()
      real, pointer, dimension(:,:,:) :: BCscacoef => null(), OCscacoef => null(), BRscacoef => null(), &
                                         DUscacoef => null(), SUscacoef => null(), SSscacoef => null(), &
                                         NIscacoef => null()
      real, pointer, dimension(:,:)   :: GMITO3 => null(), GMITTO3 => null(), latarr => null()
      real, pointer, dimension(:,:,:) :: T_BST => null(), NO2 => null(), O3 => null(), CH4 => null(), CO => null(), &
                                         ISOP => null(), ACET => null(), C2H6 => null(), C3H8 => null(), &
                                         PRPE => null(), ALK4 => null(), MP => null(), H2O2 => null(), &
                                         CLOUD => null(), QV => null(), CH2O => null()
      real, pointer, dimension(:,:)   :: ALBUV => null(), sza_noon => null()
      real, pointer, dimension(:,:,:) :: default_OH => null()
   end type

   !  The engineered features, for the DIAG_* exports of :1607-1640: associate the ones that are wanted
   type OH_RUN1_DIAG
      real, pointer, dimension(:,:,:) :: PL_BST => null(), tauclwDN => null(), taucliDN => null(), taucliUP => null(), &
                                         tauclwUP => null(), aodUP => null(), aodDN => null(), aod => null()
      real, pointer, dimension(:,:)   :: stratO3 => null()
   end type

   interface
      function OHXOHPostProcess(im, jm, km, avogad, runiv, epsilon, ple_mod, t_mod, q_mod, tropp_mod, default_oh, &
                                oh_ml, oh, ndwet) bind(C, name="OHXOHPostProcess") result(rc)
         import :: c_int, c_float, c_ptr
         integer(c_int), value :: im, jm, km
         real(c_float), value  :: avogad, runiv, epsilon
         type(c_ptr), value    :: ple_mod, t_mod, q_mod, tropp_mod, default_oh, oh_ml, oh, ndwet
         integer(c_int)        :: rc
      end function
      function OHXBoosterRun1(handle, args) bind(C, name="OHXBoosterRun1") result(rc)
         import :: c_ptr, c_int, OHXRun1Args
         type(c_ptr), value            :: handle
         type(OHXRun1Args), intent(in) :: args
         integer(c_int)                :: rc
      end function
      function OHXJulianDay(nymd, jday) bind(C, name="OHXJulianDay") result(rc)
         import :: c_int
         integer(c_int), value       :: nymd
         integer(c_int), intent(out) :: jday
         integer(c_int)              :: rc
      end function
      function OHXSolarGeometry(jday, lats, lons, im, jm, deg2rad, rad2deg, lat_deg, sza_noon) &
            bind(C, name="OHXSolarGeometry") result(rc)
         import :: c_int, c_float
         integer(c_int), value      :: jday, im, jm
         real(c_float), intent(in)  :: lats(*), lons(*)
         real(c_float), value       :: deg2rad, rad2deg
         real(c_float), intent(out) :: lat_deg(*), sza_noon(*)
         integer(c_int)             :: rc
      end function
   end interface

   character(len=512), save :: last_error = ''

contains

   function oh_run1_error_text() result(msg)
      character(len=:), allocatable :: msg
      msg = trim(last_error)
   end function

   integer function oh_julian_day(nymd)                       ! JulianDay (:1905-1936)
      integer, intent(in) :: nymd
      integer(c_int) :: jd, crc
      crc = OHXJulianDay(int(nymd, c_int), jd)
      oh_julian_day = jd
   end function

   !  latarr = LATS*MAPL_RADIANS_TO_DEGREES (:1444); sza_noon = computeSolarZenithAngle_LocalNoon (:1482)
   subroutine oh_solar_geometry(jday, LATS, LONS, degrees_to_radians, radians_to_degrees, latarr, sza_noon, rc)
      integer, intent(in) :: jday
      real, intent(in), contiguous  :: LATS(:,:), LONS(:,:)
      real, intent(in)  :: degrees_to_radians, radians_to_degrees
      real, intent(out), contiguous :: latarr(:,:), sza_noon(:,:)
      integer, intent(out) :: rc
      integer(c_int) :: crc
      rc = OH_XGB_SUCCESS
      crc = OHXSolarGeometry(int(jday, c_int), LATS, LONS, int(size(LATS,1), c_int), int(size(LATS,2), c_int), &
                             degrees_to_radians, radians_to_degrees, latarr, sza_noon)
      if (crc /= 0) then
         last_error = 'Failed in OHXSolarGeometry :: '//ohx_last_error()
         rc = OH_XGB_FAILURE
      end if
   end subroutine

   !  The same two fields with the HOST's libm, as the reference computes them (:427-462, 1444): the zenith angle feeds
   !  tree splits, so a run that must give the OH of a CPU run of the same executable takes its sines and cosines from
   !  the same library.  A 2-D field once per Boost tick: microseconds.  The OH shell's default (solar_geometry: host).
   subroutine oh_solar_geometry_host(jday, LATS, LONS, degrees_to_radians, radians_to_degrees, latarr, sza_noon)
      integer, intent(in) :: jday
      real, intent(in)  :: LATS(:,:), LONS(:,:)
      real, intent(in)  :: degrees_to_radians, radians_to_degrees
      real, intent(out) :: latarr(:,:), sza_noon(:,:)
      real :: sindec, cosdec, sinlat, coslat, mylon, tau, loct, cosz
      integer :: i, j
      sindec = 0.3978 * sin(0.9863 * (jday - 80.0) * degrees_to_radians)
      cosdec = cos(asin(sindec))
      do j = 1, size(LATS, 2)
         do i = 1, size(LATS, 1)
            latarr(i,j) = LATS(i,j) * radians_to_degrees
            sinlat = sin(LATS(i,j))
            coslat = cos(asin(sinlat))
            mylon = LONS(i,j) * radians_to_degrees                 ! local noon: the hour at which the sun is overhead here
            if (mylon > 180.0) mylon = mylon - 360.0
            if (mylon < -180.0) mylon = mylon + 360.0
            tau = 12.0 + (mylon / (-180.0)) * 12.0
            loct = ((tau * 15.0) - 180.0) * degrees_to_radians + LONS(i,j)
            cosz = cosdec * coslat * cos(loct) + sindec * sinlat
            cosz = max(-1.0, min(1.0, cosz))
            sza_noon(i,j) = acos(cosz) * radians_to_degrees
         end do
      end do
   end subroutine

   !  Address of a field for the C side, or NULL with an error text: the library reads n1*n2*n3 floats from
   !  it, so a strided section or an array of another shape must never get through (ADVICE r1).
   type(c_ptr) function loc3(a, n1, n2, n3, name)
      real, pointer, intent(in) :: a(:,:,:)
      integer, intent(in) :: n1, n2, n3
      character(len=*), intent(in) :: name
      loc3 = c_null_ptr
      if (.not. associated(a)) then
         if (len_trim(last_error) == 0) last_error = 'oh_run1: field '//name//' is not associated'
         return
      end if
      if (.not. is_contiguous(a) .or. size(a,1) /= n1 .or. size(a,2) /= n2 .or. size(a,3) /= n3) then
         if (len_trim(last_error) == 0) last_error = 'oh_run1: field '//name//' is not a contiguous array of the expected shape'
         return
      end if
      loc3 = c_loc(a(lbound(a,1), lbound(a,2), lbound(a,3)))
   end function

   type(c_ptr) function loc2(a, n1, n2, name)
      real, pointer, intent(in) :: a(:,:)
      integer, intent(in) :: n1, n2
      character(len=*), intent(in) :: name
      loc2 = c_null_ptr
      if (.not. associated(a)) then
         if (len_trim(last_error) == 0) last_error = 'oh_run1: field '//name//' is not associated'
         return
      end if
      if (.not. is_contiguous(a) .or. size(a,1) /= n1 .or. size(a,2) /= n2) then
         if (len_trim(last_error) == 0) last_error = 'oh_run1: field '//name//' is not a contiguous array of the expected shape'
         return
      end if
      loc2 = c_loc(a(lbound(a,1), lbound(a,2)))
   end function

   !  optional outputs: NULL when not wanted, checked like the inputs when they are
   type(c_ptr) function opt3(a, n1, n2, n3, name)
      real, pointer, intent(in) :: a(:,:,:)
      integer, intent(in) :: n1, n2, n3
      character(len=*), intent(in) :: name
      opt3 = c_null_ptr
      if (associated(a)) opt3 = loc3(a, n1, n2, n3, name)
   end function

   !  CALL_BOOST and what surrounds it (:1444-1478, 1488, 1557-1595) in one call.
   !  OH is the INTERNAL field (molec/cm3), OH_boost the export OH_ML*OHscale, NDWET the diagnostic.
   !  want_boost / want_ndwet = .false.: that array is neither filled nor moved (it keeps what it held) - a rank whose
   !  tick needs neither (Boost on every tick, nobody asking for the OH_boost or DIAG_NDWET exports) gets a third of the
   !  results' bytes back over PCIe.
   subroutine oh_run1_boost(XGBoostFilename, im, jm, km, dynamic_k_range, tropp_min, OHscale, &
                            avogad, runiv, epsilon, st, OH, OH_boost, NDWET, k1, k2, rc, diag, want_boost, want_ndwet)
      character(len=*), intent(in) :: XGBoostFilename
      integer, intent(in)  :: im, jm, km
      logical, intent(in)  :: dynamic_k_range
      real, intent(in)     :: tropp_min, OHscale, avogad, runiv, epsilon
      type(OH_RUN1_STATE), intent(in) :: st
      real, intent(out), target, contiguous :: OH(:,:,:)
      real, intent(inout), target, contiguous :: OH_boost(:,:,:), NDWET(:,:,:)
      integer, intent(out) :: k1, k2, rc
      type(OH_RUN1_DIAG), intent(in), optional :: diag
      logical, intent(in), optional :: want_boost, want_ndwet
      type(OHXRun1Args) :: a
      type(c_ptr) :: bst
      integer(c_int32_t), target :: ck1, ck2
      integer(c_int) :: crc

      k1 = 0
      k2 = 0
      call oh_xgb_booster(XGBoostFilename, bst, rc)
      if (rc /= OH_XGB_SUCCESS) then
         last_error = 'oh_run1_boost: the booster could not be loaded'
         return
      end if
      if (any(shape(OH) /= [im, jm, km]) .or. any(shape(OH_boost) /= [im, jm, km]) .or. any(shape(NDWET) /= [im, jm, km])) then
         last_error = 'oh_run1_boost: OH, OH_boost and NDWET must be (im,jm,km)'
         rc = OH_XGB_FAILURE
         return
      end if
      last_error = ''
      a%im = im; a%jm = jm; a%km = km
      a%dynamic_k_range = merge(1, 0, dynamic_k_range)
      a%tropp_min = tropp_min; a%ohscale = OHscale; a%missing = -999.0     ! :213
      a%avogad = avogad; a%runiv = runiv; a%epsilon = epsilon
      a%ple_mod = loc3(st%PLE_MOD, im, jm, km + 1, 'PLE_MOD'); a%t_mod = loc3(st%T_MOD, im, jm, km, 'T_MOD')
      a%q_mod = loc3(st%Q_MOD, im, jm, km, 'Q_MOD'); a%tropp_mod = loc2(st%TROPP_MOD, im, jm, 'TROPP_MOD')
      a%ple_bst = loc3(st%PLE_BST, im, jm, km + 1, 'PLE_BST'); a%zle_bst = loc3(st%ZLE_BST, im, jm, km + 1, 'ZLE_BST')
      a%tauclw = loc3(st%TAUCLW, im, jm, km, 'TAUCLW'); a%taucli = loc3(st%TAUCLI, im, jm, km, 'TAUCLI')
      a%scacoef(1) = loc3(st%BCscacoef, im, jm, km, 'BCscacoef'); a%scacoef(2) = loc3(st%OCscacoef, im, jm, km, 'OCscacoef')
      a%scacoef(3) = loc3(st%BRscacoef, im, jm, km, 'BRscacoef'); a%scacoef(4) = loc3(st%DUscacoef, im, jm, km, 'DUscacoef')
      a%scacoef(5) = loc3(st%SUscacoef, im, jm, km, 'SUscacoef'); a%scacoef(6) = loc3(st%SSscacoef, im, jm, km, 'SSscacoef')
      a%scacoef(7) = loc3(st%NIscacoef, im, jm, km, 'NIscacoef')
      a%gmito3 = loc2(st%GMITO3, im, jm, 'GMITO3'); a%gmitto3 = loc2(st%GMITTO3, im, jm, 'GMITTO3')
      a%lat_deg = loc2(st%latarr, im, jm, 'latarr')
      a%t_bst = loc3(st%T_BST, im, jm, km, 'T_BST'); a%no2 = loc3(st%NO2, im, jm, km, 'NO2'); a%o3 = loc3(st%O3, im, jm, km, 'O3')
      a%ch4 = loc3(st%CH4, im, jm, km, 'CH4'); a%co = loc3(st%CO, im, jm, km, 'CO'); a%isop = loc3(st%ISOP, im, jm, km, 'ISOP')
      a%acet = loc3(st%ACET, im, jm, km, 'ACET'); a%c2h6 = loc3(st%C2H6, im, jm, km, 'C2H6')
      a%c3h8 = loc3(st%C3H8, im, jm, km, 'C3H8'); a%prpe = loc3(st%PRPE, im, jm, km, 'PRPE')
      a%alk4 = loc3(st%ALK4, im, jm, km, 'ALK4'); a%mp = loc3(st%MP, im, jm, km, 'MP'); a%h2o2 = loc3(st%H2O2, im, jm, km, 'H2O2')
      a%cloud = loc3(st%CLOUD, im, jm, km, 'CLOUD'); a%qv = loc3(st%QV, im, jm, km, 'QV')
      a%albuv = loc2(st%ALBUV, im, jm, 'ALBUV'); a%ch2o = loc3(st%CH2O, im, jm, km, 'CH2O')
      a%sza = loc2(st%sza_noon, im, jm, 'sza_noon'); a%default_oh = loc3(st%default_OH, im, jm, km, 'default_OH')
      if (len_trim(last_error) /= 0) then
         rc = OH_XGB_FAILURE
         return
      end if
      a%oh = c_loc(OH(1,1,1)); a%oh_boost = c_loc(OH_boost(1,1,1)); a%ndwet = c_loc(NDWET(1,1,1))
      if (present(want_boost)) then
         if (.not. want_boost) a%oh_boost = c_null_ptr
      end if
      if (present(want_ndwet)) then
         if (.not. want_ndwet) a%ndwet = c_null_ptr
      end if
      a%k1 = c_loc(ck1); a%k2 = c_loc(ck2)
      a%diag_pl_bst = c_null_ptr; a%diag_tauclwdn = c_null_ptr; a%diag_tauclidn = c_null_ptr
      a%diag_taucliup = c_null_ptr; a%diag_tauclwup = c_null_ptr; a%diag_aodup = c_null_ptr
      a%diag_aoddn = c_null_ptr; a%diag_aod = c_null_ptr; a%diag_strato3 = c_null_ptr
      if (present(diag)) then
         a%diag_pl_bst = opt3(diag%PL_BST, im, jm, km, 'diag PL_BST'); a%diag_tauclwdn = opt3(diag%tauclwDN, im, jm, km, 'diag tauclwDN')
         a%diag_tauclidn = opt3(diag%taucliDN, im, jm, km, 'diag taucliDN'); a%diag_taucliup = opt3(diag%taucliUP, im, jm, km, 'diag taucliUP')
         a%diag_tauclwup = opt3(diag%tauclwUP, im, jm, km, 'diag tauclwUP'); a%diag_aodup = opt3(diag%aodUP, im, jm, km, 'diag aodUP')
         a%diag_aoddn = opt3(diag%aodDN, im, jm, km, 'diag aodDN'); a%diag_aod = opt3(diag%aod, im, jm, km, 'diag aod')
         if (associated(diag%stratO3)) a%diag_strato3 = loc2(diag%stratO3, im, jm, 'diag stratO3')
         if (len_trim(last_error) /= 0) then
            rc = OH_XGB_FAILURE
            return
         end if
      end if
      crc = OHXBoosterRun1(bst, a)
      if (crc /= 0) then
         last_error = 'Failed in OHXBoosterRun1 :: '//ohx_last_error()
         rc = OH_XGB_FAILURE
         return
      end if
      k1 = ck1
      k2 = ck2
   end subroutine

   !  The tropopause mask and the unit conversion alone (:1247-1257, 1579-1595), for a tick that does not call
   !  Boost (compute_once_per_day, :1189-1193): OH_ML is the persisted self%OH_ML, already scaled.
   !  ohx_register_host (include/ohxgb.h): process-wide, no booster needed.  The caller promises that every array it hands
   !  to the host forms from now on stays allocated until the end of the run.
   subroutine oh_run1_register_host_arrays(on)
      logical, intent(in) :: on
      integer(c_int) :: rc
      rc = XGBoosterSetParam(c_null_ptr, 'ohx_register_host'//c_null_char, merge('1', '0', on)//c_null_char)
   end subroutine

   subroutine oh_post_process(im, jm, km, avogad, runiv, epsilon, PLE_MOD, T_MOD, Q_MOD, TROPP_MOD, default_OH, OH_ML, &
                              OH, NDWET, rc)
      integer, intent(in) :: im, jm, km
      real, intent(in)    :: avogad, runiv, epsilon
      real, pointer, intent(in) :: PLE_MOD(:,:,:), T_MOD(:,:,:), Q_MOD(:,:,:), TROPP_MOD(:,:), default_OH(:,:,:)
      real, intent(in), target, contiguous  :: OH_ML(:,:,:)
      real, intent(out), target, contiguous :: OH(:,:,:), NDWET(:,:,:)
      integer, intent(out) :: rc
      type(c_ptr) :: p_ple, p_t, p_q, p_tropp, p_def
      integer(c_int) :: crc
      rc = OH_XGB_SUCCESS
      last_error = ''
      p_ple = loc3(PLE_MOD, im, jm, km + 1, 'PLE_MOD'); p_t = loc3(T_MOD, im, jm, km, 'T_MOD')
      p_q = loc3(Q_MOD, im, jm, km, 'Q_MOD'); p_tropp = loc2(TROPP_MOD, im, jm, 'TROPP_MOD')
      p_def = loc3(default_OH, im, jm, km, 'default_OH')
      if (len_trim(last_error) == 0 .and. (any(shape(OH_ML) /= [im, jm, km]) .or. any(shape(OH) /= [im, jm, km]) .or. &
                                           any(shape(NDWET) /= [im, jm, km]))) &
         last_error = 'oh_post_process: OH_ML, OH and NDWET must be (im,jm,km)'
      if (len_trim(last_error) /= 0) then
         rc = OH_XGB_FAILURE
         return
      end if
      crc = OHXOHPostProcess(int(im, c_int), int(jm, c_int), int(km, c_int), avogad, runiv, epsilon, p_ple, p_t, p_q, &
                             p_tropp, p_def, c_loc(OH_ML(1,1,1)), c_loc(OH(1,1,1)), c_loc(NDWET(1,1,1)))
      if (crc /= 0) then
         last_error = 'Failed in OHXOHPostProcess :: '//ohx_last_error()
         rc = OH_XGB_FAILURE
      end if
   end subroutine

   !  The same tick on the rank's own core - the tick OH_instance_OH.rc:38 (compute_once_per_day: T) makes 23 times out
   !  of 24: OH_ML is the last Boost's, nothing on it needs a tree.  One pass, each gridcell read once and written once,
   !  with the reference's expressions in the reference's order of operations (default REAL throughout):
   !    PL = (PLE(k-1) + PLE(k)) * 0.5, TV = T * (1 + Q/eps) / (1 + Q), NDWET = (AVOGAD*PL) / (RUNIV*TV)     :1247-1257
   !    OH = OH_ML where PL > TROPP, default_OH elsewhere                                                      :1579-1587
   !    OH = (OH * NDWET) * 1.0e-6                                                                             :1595
   !  The reference makes four whole-array passes with three temporaries allocated per tick (:1240-1242); the results
   !  are the same bits (tests/test_reference_child.py, every skip tick of the two model days).
   subroutine oh_post_process_host(im, jm, km, avogad, runiv, epsilon, PLE_MOD, T_MOD, Q_MOD, TROPP_MOD, default_OH, &
                                   OH_ML, OH, NDWET)
      integer, intent(in) :: im, jm, km
      real, intent(in)  :: avogad, runiv, epsilon
      real, intent(in)  :: PLE_MOD(im, jm, 0:km), T_MOD(im, jm, km), Q_MOD(im, jm, km), TROPP_MOD(im, jm)
      real, intent(in)  :: default_OH(im, jm, km), OH_ML(im, jm, km)
      real, intent(out) :: OH(im, jm, km), NDWET(im, jm, km)
      real :: pl, tv, nd
      integer :: i, j, k
      do k = 1, km
         do j = 1, jm
            do i = 1, im
               pl = (PLE_MOD(i,j,k-1) + PLE_MOD(i,j,k)) * 0.5
               tv = T_MOD(i,j,k) * (1.0 + Q_MOD(i,j,k) / epsilon) / (1.0 + Q_MOD(i,j,k))
               nd = (avogad * pl) / (runiv * tv)
               NDWET(i,j,k) = nd
               OH(i,j,k) = (merge(OH_ML(i,j,k), default_OH(i,j,k), pl > TROPP_MOD(i,j)) * nd) * 1.0e-6
            end do
         end do
      end do
   end subroutine

end module oh_run1
