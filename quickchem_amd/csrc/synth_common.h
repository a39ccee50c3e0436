// Synthetic OH-predictor inputs: one definition shared by the host (g++) and the
// device (hipcc) so that any shard of the C48..C720 batches is reproducible
// bit-for-bit anywhere without communication.
//
// The 27 features and their ORDER are the reference's xx_carr(1:27,m) gather
// (/root/reference OH_GridComp/OH_GridCompMod.F90:313-339); the row index runs
// i fastest, then j, then k (OH_GridCompMod.F90:309-311,341).
//
// Everything here is integer hashing plus IEEE-754 single add/sub/mul/compare,
// no fused multiply-add (both compilers get -ffp-contract=off and the bodies
// carry the clang pragma), no libm: host and device agree to the last bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define OHX_HD __host__ __device__ __forceinline__
#else
#define OHX_HD static inline
#endif

#define OHX_NFEAT 27
#define OHX_SYNTH_SEED 20241108u

enum {
  OHX_F_LAT = 0, OHX_F_PL, OHX_F_T, OHX_F_NO2, OHX_F_O3, OHX_F_CH4, OHX_F_CO,
  OHX_F_ISOP, OHX_F_ACET, OHX_F_C2H6, OHX_F_C3H8, OHX_F_PRPE, OHX_F_ALK4,
  OHX_F_MP, OHX_F_H2O2, OHX_F_TAUCLWDN, OHX_F_TAUCLIDN, OHX_F_TAUCLIUP,
  OHX_F_TAUCLWUP, OHX_F_CLOUD, OHX_F_QV, OHX_F_GMISTRATO3, OHX_F_ALBUV,
  OHX_F_AODUP, OHX_F_AODDN, OHX_F_CH2O, OHX_F_SZA
};

OHX_HD uint32_t ohx_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

OHX_HD uint32_t ohx_hash4(uint32_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t h = seed;
  h = ohx_fmix32(h ^ (a * 0x9E3779B1u));
  h = ohx_fmix32(h ^ (b * 0x85EBCA77u));
  h = ohx_fmix32(h ^ (c * 0xC2B2AE3Du));
  h = ohx_fmix32(h ^ (d * 0x27D4EB2Fu));
  return h;
}

// 24 random bits -> [0,1), exact in float
OHX_HD float ohx_u01(uint32_t h) { return (float)(h >> 8) * 5.9604644775390625e-08f; }

OHX_HD float ohx_clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

// piecewise-linear 2**x for x >= 0 (continuous, monotone, exact in float)
OHX_HD float ohx_pexp2(float x) {
#pragma clang fp contract(off)
  int e = (int)x;
  if (e > 30) e = 30;
  float fr = x - (float)e;
  return (1.0f + fr) * (float)(1u << e);
}

// bilinear interpolation of a hashed lattice with spacing 2**lg cells
OHX_HD float ohx_smooth(uint32_t seed, uint32_t chan, int i, int j, int lg) {
#pragma clang fp contract(off)
  const int g = 1 << lg;
  const float inv = 1.0f / (float)g;
  const uint32_t gx = (uint32_t)(i >> lg), gy = (uint32_t)(j >> lg);
  const float fx = (float)(i & (g - 1)) * inv, fy = (float)(j & (g - 1)) * inv;
  const float v00 = ohx_u01(ohx_hash4(seed, 0x51000000u + chan * 8u + (uint32_t)lg, gx, gy, 0u));
  const float v10 = ohx_u01(ohx_hash4(seed, 0x51000000u + chan * 8u + (uint32_t)lg, gx + 1u, gy, 0u));
  const float v01 = ohx_u01(ohx_hash4(seed, 0x51000000u + chan * 8u + (uint32_t)lg, gx, gy + 1u, 0u));
  const float v11 = ohx_u01(ohx_hash4(seed, 0x51000000u + chan * 8u + (uint32_t)lg, gx + 1u, gy + 1u, 0u));
  const float a = v00 + (v10 - v00) * fx;
  const float b = v01 + (v11 - v01) * fx;
  return a + (b - a) * fy;
}

OHX_HD float ohx_noise(uint32_t seed, uint32_t f, int i, int j, int k) {
  return ohx_u01(ohx_hash4(seed, 0x77000000u + f, (uint32_t)i, (uint32_t)j, (uint32_t)k));
}

// one optical-depth layer: sparse (mostly zero) non-negative value
OHX_HD float ohx_cloud_layer(uint32_t seed, uint32_t kind, int i, int j, int kk, int km, float cover) {
#pragma clang fp contract(off)
  // clouds only in the lower 70% of the column
  if (kk * 10 < km * 3) return 0.0f;
  const uint32_t h = ohx_hash4(seed, 0x33000000u + kind, (uint32_t)i, (uint32_t)j, (uint32_t)kk);
  const float pick = (float)(h & 0xFFu) * (1.0f / 256.0f);
  if (!(pick < cover)) return 0.0f;
  return 6.0f * ohx_u01(h);
}

OHX_HD float ohx_aod_layer(uint32_t seed, int i, int j, int kk, int km, float load) {
#pragma clang fp contract(off)
  const float s = ((float)kk + 0.5f) / (float)km;
  const uint32_t h = ohx_hash4(seed, 0x34000000u, (uint32_t)i, (uint32_t)j, (uint32_t)kk);
  return 0.02f * load * (s * s) * (0.25f + ohx_u01(h));
}

// mix of vertical profile p, coarse field, fine field and white noise, all in [0,1]
OHX_HD float ohx_mix(uint32_t seed, uint32_t f, int i, int j, int k, float p,
                     float wp, float ws, float wm, float wn) {
#pragma clang fp contract(off)
  float v = wp * p;
  v = v + ws * ohx_smooth(seed, f, i, j, 5);
  v = v + wm * ohx_smooth(seed, f, i, j, 2);
  v = v + wn * ohx_noise(seed, f, i, j, k);
  return ohx_clamp01(v);
}

// Feature f (0-based, reference order) of gridcell (i,j,k), 0-based, k = 0 is the
// model top.  PL is returned in hPa, i.e. already divided by 100 as at
// OH_GridCompMod.F90:314; ohx_synth_pl_pa() gives the Pa field the caller holds.
OHX_HD float ohx_synth_feature(uint32_t seed, int f, int i, int j, int k, int im, int jm, int km) {
#pragma clang fp contract(off)
  (void)im; (void)jm;
  const float s = ((float)k + 0.5f) / (float)km;  // 0 = top, 1 = surface
  switch (f) {
    case OHX_F_LAT:
      return -90.0f + 180.0f * ohx_smooth(seed, 0u, i, j, 5);
    case OHX_F_PL: {
      const float ps = 500.0f + 540.0f * ohx_smooth(seed, 1u, i, j, 5);
      return 0.01f + (ps - 0.01f) * (s * s);
    }
    case OHX_F_T:
      return 180.0f + 140.0f * ohx_mix(seed, 2u, i, j, k, s, 0.70f, 0.20f, 0.05f, 0.05f);
    case OHX_F_NO2:
      return 1.0e-12f * ohx_pexp2(10.0f * ohx_mix(seed, 3u, i, j, k, s, 0.40f, 0.30f, 0.15f, 0.15f));
    case OHX_F_O3:
      return 1.0e-8f * ohx_pexp2(10.0f * ohx_mix(seed, 4u, i, j, k, 1.0f - s, 0.60f, 0.20f, 0.10f, 0.10f));
    case OHX_F_CH4:
      return 1.6e-6f + 0.4e-6f * ohx_mix(seed, 5u, i, j, k, s, 0.30f, 0.40f, 0.20f, 0.10f);
    case OHX_F_CO:
      return 2.0e-8f * ohx_pexp2(4.0f * ohx_mix(seed, 6u, i, j, k, s, 0.35f, 0.35f, 0.15f, 0.15f));
    case OHX_F_ISOP:
      return 1.0e-14f * ohx_pexp2(14.0f * ohx_mix(seed, 7u, i, j, k, s * s, 0.50f, 0.25f, 0.10f, 0.15f));
    case OHX_F_ACET:
      return 1.0e-11f * ohx_pexp2(9.0f * ohx_mix(seed, 8u, i, j, k, s, 0.35f, 0.35f, 0.15f, 0.15f));
    case OHX_F_C2H6:
      return 1.0e-11f * ohx_pexp2(8.0f * ohx_mix(seed, 9u, i, j, k, s, 0.30f, 0.40f, 0.15f, 0.15f));
    case OHX_F_C3H8:
      return 1.0e-12f * ohx_pexp2(10.0f * ohx_mix(seed, 10u, i, j, k, s, 0.30f, 0.40f, 0.15f, 0.15f));
    case OHX_F_PRPE:
      return 1.0e-13f * ohx_pexp2(11.0f * ohx_mix(seed, 11u, i, j, k, s * s, 0.40f, 0.30f, 0.15f, 0.15f));
    case OHX_F_ALK4:
      return 1.0e-12f * ohx_pexp2(10.0f * ohx_mix(seed, 12u, i, j, k, s, 0.35f, 0.35f, 0.15f, 0.15f));
    case OHX_F_MP:
      return 1.0e-11f * ohx_pexp2(8.0f * ohx_mix(seed, 13u, i, j, k, s, 0.30f, 0.35f, 0.15f, 0.20f));
    case OHX_F_H2O2:
      return 1.0e-11f * ohx_pexp2(9.0f * ohx_mix(seed, 14u, i, j, k, s, 0.35f, 0.30f, 0.15f, 0.20f));
    case OHX_F_TAUCLWDN:
    case OHX_F_TAUCLIDN:
    case OHX_F_TAUCLIUP:
    case OHX_F_TAUCLWUP: {
      // TAUCL?DN(k) = SUM(layer(k:km)), TAUCL?UP(k) = SUM(layer(1:k))
      // (OH_GridCompMod.F90:1470-1473), summed here in ascending layer order.
      const uint32_t kind = (f == OHX_F_TAUCLWDN || f == OHX_F_TAUCLWUP) ? 0u : 1u;
      const float cover = 0.6f * ohx_smooth(seed, 15u + kind, i, j, 4);
      const bool up = (f == OHX_F_TAUCLIUP || f == OHX_F_TAUCLWUP);
      const int k0 = up ? 0 : k, k1 = up ? k : km - 1;
      float acc = 0.0f;
      for (int kk = k0; kk <= k1; ++kk) acc = acc + ohx_cloud_layer(seed, kind, i, j, kk, km, cover);
      return acc;
    }
    case OHX_F_CLOUD: {
      const float v = ohx_mix(seed, 19u, i, j, k, s, 0.10f, 0.40f, 0.25f, 0.25f);
      return ohx_clamp01(1.8f * v - 0.6f);
    }
    case OHX_F_QV:
      return 1.0e-7f * ohx_pexp2(18.0f * ohx_mix(seed, 20u, i, j, k, s, 0.75f, 0.15f, 0.05f, 0.05f));
    case OHX_F_GMISTRATO3:
      return 200.0f + 250.0f * ohx_clamp01(0.7f * ohx_smooth(seed, 21u, i, j, 5) + 0.3f * ohx_smooth(seed, 21u, i, j, 2));
    case OHX_F_ALBUV:
      return 0.02f + 0.88f * ohx_clamp01(0.5f * ohx_smooth(seed, 22u, i, j, 5) + 0.5f * ohx_smooth(seed, 22u, i, j, 2));
    case OHX_F_AODUP:
    case OHX_F_AODDN: {
      // aodUP(k) = SUM(aod(1:k)), aodDN(k) = SUM(aod(k:km)) (OH_GridCompMod.F90:1475-1476)
      const float load = 0.2f + ohx_smooth(seed, 23u, i, j, 4);
      const bool up = (f == OHX_F_AODUP);
      const int k0 = up ? 0 : k, k1 = up ? k : km - 1;
      float acc = 0.0f;
      for (int kk = k0; kk <= k1; ++kk) acc = acc + ohx_aod_layer(seed, i, j, kk, km, load);
      return acc;
    }
    case OHX_F_CH2O:
      return 1.0e-12f * ohx_pexp2(11.0f * ohx_mix(seed, 25u, i, j, k, s, 0.45f, 0.30f, 0.10f, 0.15f));
    case OHX_F_SZA: {
      const float l = 2.0f * ohx_smooth(seed, 0u, i, j, 5) - 1.0f;
      return 113.0f * (l < 0.0f ? -l : l);
    }
    default:
      return 0.0f;
  }
}

OHX_HD bool ohx_feature_is_2d(int f) {
  return f == OHX_F_LAT || f == OHX_F_GMISTRATO3 || f == OHX_F_ALBUV || f == OHX_F_SZA;
}

// mid-level pressure in Pa, the field predict_OH_with_XGB receives as bb%PL
OHX_HD float ohx_synth_pl_pa(uint32_t seed, int i, int j, int k, int im, int jm, int km) {
#pragma clang fp contract(off)
  return ohx_synth_feature(seed, OHX_F_PL, i, j, k, im, jm, km) * 100.0f;
}

// tropopause pressure in Pa, 2-D, 90..320 hPa
OHX_HD float ohx_synth_tropp_pa(uint32_t seed, int i, int j) {
#pragma clang fp contract(off)
  return 9000.0f + 23000.0f * ohx_smooth(seed, 40u, i, j, 5);
}
