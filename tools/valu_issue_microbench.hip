// valu_issue_microbench — what a wave64 vector-ALU instruction costs a gfx950 SIMD to ISSUE, by how many
// waves share the SIMD (1 / 2 / 4 / 8) and by instruction kind: the ones the tree walk's step is made of
// (v_cmp_*_f32 to an SGPR pair, v_cndmask_b32 on an SGPR pair, v_addc_co_u32 with a mask as carry-in,
// v_add_lshl_u32) next to v_add_u32 and v_fma_f32 as yardsticks, plus "walk": the 14-instruction step of
// predict_rows_ring_kernel as a dependent chain, four independent chains per lane as the kernel has.
//
// DESIGN.md §5 priced the ring kernel's VALU load at 4 cycles per wave64 instruction (83 % busy);
// MI355X_MICROARCH.md says a CDNA4 SIMD is SIMD-32: 2 cycles when other waves fill the gaps, 4 for a wave
// alone.  This measures which it is for THESE instructions, and - run under rocprofv3 --pmc - calibrates
// what SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES read for a known instruction stream
// (tools/pmc.sh prints the same counters for the real kernel).
//
// Standalone: hipcc --offload-arch=gfx950 -O3 valu_issue_microbench.hip -o valu_issue_microbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

enum Kind { K_ADD = 0, K_FMA, K_CMP, K_CNDMASK, K_ADDC, K_ADDLSHL, K_WALK, K_CMP32, K_CNDMASK32, K_ADDC32, K_WALK32, K_COUNT };
static const char* kKindName[K_COUNT] = {"v_add_u32", "v_fma_f32", "v_cmp_lt_f32(e64)", "v_cndmask_b32(e64)",
                                         "v_addc_co_u32", "v_add_lshl_u32", "walk-step x4 chains",
                                         "v_cmp_lt_f32_e32 vcc", "v_cndmask_b32_e32 vcc", "v_addc_co_u32_e32 vcc",
                                         "walk-step, vcc forms"};
// vector instructions per loop trip, per kind
static const int kPerTrip[K_COUNT] = {32, 32, 32, 32, 32, 32, 56, 32, 32, 32, 56};

// One block = 256 threads = one wave per SIMD.  `iters` trips of a straight-line body of independent (or, for
// K_WALK, chain-dependent) instructions; s_memtime around the loop, per wave.
template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(int iters, uint32_t seed, uint32_t* __restrict__ sink,
                                                   unsigned long long* __restrict__ cycles) {
  extern __shared__ uint32_t lds_pad[];   // sized by the host to set how many blocks a CU takes
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;
  uint32_t a4 = a0 ^ 0x55u, a5 = a1 ^ 0x33u, a6 = a2 ^ 0x0Fu, a7 = a3 ^ 0xF0u;
  float f0 = (float)a0, f1 = (float)a1 * 0.5f, f2 = (float)a2 * 0.25f, f3 = (float)a3 * 0.125f;
  const uint32_t b = seed | 1u;
  const float fb = (float)(seed & 0xFFu) + 0.5f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memrealtime %1\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (KIND == K_ADD) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b));
    } else if (KIND == K_FMA) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        asm volatile(
            "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
            : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
            : "v"(fb));
    } else if (KIND == K_CMP) {
      // 32 compares into four SGPR pairs in turn (the walk keeps its decisions in SGPR pairs)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        asm volatile(
            "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n"
            "v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n"
            : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
            : "v"(fb)
            : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (KIND == K_CNDMASK) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
            "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
            "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n"
            "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b)
            : "s20", "s21", "s22", "s23");
    } else if (KIND == K_ADDC) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_addc_co_u32_e64 %0, s[24:25], %0, %8, s[20:21]\n v_addc_co_u32_e64 %1, s[26:27], %1, %8, s[22:23]\n"
            "v_addc_co_u32_e64 %2, s[24:25], %2, %8, s[20:21]\n v_addc_co_u32_e64 %3, s[26:27], %3, %8, s[22:23]\n"
            "v_addc_co_u32_e64 %4, s[24:25], %4, %8, s[20:21]\n v_addc_co_u32_e64 %5, s[26:27], %5, %8, s[22:23]\n"
            "v_addc_co_u32_e64 %6, s[24:25], %6, %8, s[20:21]\n v_addc_co_u32_e64 %7, s[26:27], %7, %8, s[22:23]\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b)
            : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (KIND == K_ADDLSHL) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_add_lshl_u32 %0, %0, %8, 4\n v_add_lshl_u32 %1, %1, %8, 4\n v_add_lshl_u32 %2, %2, %8, 4\n"
            "v_add_lshl_u32 %3, %3, %8, 4\n v_add_lshl_u32 %4, %4, %8, 4\n v_add_lshl_u32 %5, %5, %8, 4\n"
            "v_add_lshl_u32 %6, %6, %8, 4\n v_add_lshl_u32 %7, %7, %8, 4\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b));
    } else if (KIND == K_CMP32) {
      // the 4-byte encodings (VOPC / VOP2 with the mask in VCC) of the same three instructions: what the step could be
      // written in, chain by chain (r5: an 8-byte VOP3 costs a SIMD 4.3-4.5 cycles, a 4-byte v_add_u32 2.8)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        asm volatile(
            "v_cmp_lt_f32_e32 vcc, %0, %4\n v_cmp_lt_f32_e32 vcc, %1, %4\n"
            "v_cmp_lt_f32_e32 vcc, %2, %4\n v_cmp_lt_f32_e32 vcc, %3, %4\n"
            : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
            : "v"(fb)
            : "vcc");
    } else if (KIND == K_CNDMASK32) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n"
            "v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
            "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n"
            "v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b)
            : "vcc");
    } else if (KIND == K_ADDC32) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile(
            "v_addc_co_u32_e32 %0, vcc, %0, %8, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n"
            "v_addc_co_u32_e32 %2, vcc, %2, %8, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %8, vcc\n"
            "v_addc_co_u32_e32 %4, vcc, %4, %8, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %8, vcc\n"
            "v_addc_co_u32_e32 %6, vcc, %6, %8, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %8, vcc\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b)
            : "vcc");
    } else if (KIND == K_WALK32) {
      // the same 14 instructions per chain with every mask in VCC: block A (compare at the node, the child's threshold
      // and shift selected, the group index doubled with the decision carried in), the two VOP3 that have no short
      // form (v_bfe_u32, v_lshl_add_u32), block B (compare at the child, second doubling, leaf test, leaf kept),
      // v_add_lshl_u32; chain after chain - within a block the mask lives in VCC, so blocks do not interleave; with the
      // two wait states gfx940+ wants between a VALU writing VCC and a VALU reading it (s_nop: not vector instructions)
      uint32_t t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_, t8_, t9_, t10_, t11_, t12_, t13_, t14_, t15_;
#define W32(A, F, T0, T1, T2, T3)                                                         \
  "v_cmp_nlt_f32_e32 vcc, %" F ", %25\n"                                                  \
  "v_lshrrev_b32_e32 %" T3 ", 18, %" A "\n"                                               \
  "s_nop 0\n"                                                                             \
  "v_cndmask_b32_e32 %" T1 ", %" A ", %24, vcc\n"                                          \
  "v_cndmask_b32_e32 %" T0 ", 0, %24, vcc\n"                                              \
  "v_addc_co_u32_e32 %" T3 ", vcc, %" T3 ", %" T3 ", vcc\n"                                \
  "v_bfe_u32 %" T2 ", %" A ", %" T0 ", 5\n"                                               \
  "v_lshl_add_u32 %" T0 ", %" T2 ", 8, %24\n"                                             \
  "v_cmp_nlt_f32_e32 vcc, %" F ", %25\n"                                                  \
  "s_nop 1\n"                                                                             \
  "v_addc_co_u32_e32 %" T3 ", vcc, %" T3 ", %" T3 ", vcc\n"                                \
  "v_cmp_eq_u32_e32 vcc, 31, %" T2 "\n"                                                   \
  "s_nop 1\n"                                                                             \
  "v_cndmask_b32_e32 %" F ", %" F ", %25, vcc\n"                                           \
  "v_and_b32_e32 %" T1 ", 0x1f00, %" T1 "\n"                                              \
  "v_add_lshl_u32 %" A ", %" T3 ", %24, 4\n"                                              \
  "v_xor_b32_e32 %" A ", %" A ", %" T1 "\n"
      asm volatile(W32("0", "4", "8", "9", "10", "11") W32("1", "5", "12", "13", "14", "15") W32("2", "6", "16", "17", "18", "19")
                       W32("3", "7", "20", "21", "22", "23")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "=&v"(t0_), "=&v"(t1_),
                     "=&v"(t2_), "=&v"(t3_), "=&v"(t4_), "=&v"(t5_), "=&v"(t6_), "=&v"(t7_), "=&v"(t8_), "=&v"(t9_),
                     "=&v"(t10_), "=&v"(t11_), "=&v"(t12_), "=&v"(t13_), "=&v"(t14_), "=&v"(t15_)
                   : "v"(b), "v"(fb)
                   : "vcc");
#undef W32
      a4 ^= t2_ ^ t6_ ^ t10_ ^ t14_;
    } else {
      // the ring kernel's two-level step without its LDS read and its gather, per chain 14 vector instructions:
      // bfe (feature row), 3 compares, 2 mask selects of child data, leaf test + keep (cmp, cndmask), the record
      // number as two add-with-carry, the address (add_lshl), and three instructions' worth of bookkeeping; within a
      // chain every instruction depends on the one before it, the four chains are independent and interleaved
      // instruction by instruction, as hipcc schedules the kernel's.
      uint32_t t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_, t8_, t9_, t10_, t11_, t12_, t13_, t14_, t15_;
#define ROW4(FMT) FMT(0) FMT(1) FMT(2) FMT(3)
#define A(c) A##c
#define A0 "0"
#define A1 "1"
#define A2 "2"
#define A3 "3"
#define F(c) F##c
#define F0 "4"
#define F1 "5"
#define F2 "6"
#define F3 "7"
#define T0(c) T0##c
#define T00 "8"
#define T01 "12"
#define T02 "16"
#define T03 "20"
#define T1(c) T1##c
#define T10 "9"
#define T11 "13"
#define T12 "17"
#define T13 "21"
#define T2(c) T2##c
#define T20 "10"
#define T21 "14"
#define T22 "18"
#define T23 "22"
#define T3(c) T3##c
#define T30 "11"
#define T31 "15"
#define T32 "19"
#define T33 "23"
#define SA(c) SA##c
#define SA0 "s[20:21]"
#define SA1 "s[24:25]"
#define SA2 "s[28:29]"
#define SA3 "s[40:41]"
#define SB(c) SB##c
#define SB0 "s[22:23]"
#define SB1 "s[26:27]"
#define SB2 "s[30:31]"
#define SB3 "s[42:43]"
#define R01(c) "v_bfe_u32 %" T0(c) ", %" A(c) ", 8, 5\n"
#define R02(c) "v_cmp_lt_f32_e64 " SA(c) ", %" F(c) ", %25\n"
#define R03(c) "v_cndmask_b32_e64 %" T1(c) ", %" A(c) ", %24, " SA(c) "\n"
#define R04(c) "v_cmp_lt_f32_e64 " SB(c) ", %" F(c) ", %25\n"
#define R05(c) "v_cndmask_b32_e64 %" T2(c) ", %" T1(c) ", %" T0(c) ", " SB(c) "\n"
#define R06(c) "v_cmp_eq_u32_e64 " SB(c) ", %" T2(c) ", %24\n"
#define R07(c) "v_cndmask_b32_e64 %" F(c) ", %" F(c) ", %25, " SB(c) "\n"
#define R08(c) "v_lshrrev_b32 %" T3(c) ", 18, %" A(c) "\n"
#define R09(c) "v_addc_co_u32_e64 %" T3(c) ", " SB(c) ", %" T3(c) ", %" T3(c) ", " SA(c) "\n"
#define R10(c) "v_addc_co_u32_e64 %" T3(c) ", " SB(c) ", %" T3(c) ", %" T3(c) ", " SA(c) "\n"
#define R11(c) "v_and_b32 %" T0(c) ", 0x1f00, %" A(c) "\n"
#define R12(c) "v_add_lshl_u32 %" A(c) ", %" T3(c) ", %24, 4\n"
#define R13(c) "v_xor_b32 %" A(c) ", %" A(c) ", %" T0(c) "\n"
#define R14(c) "v_or_b32 %" A(c) ", %" A(c) ", %" T2(c) "\n"
      asm volatile(ROW4(R01) ROW4(R02) ROW4(R03) ROW4(R04) ROW4(R05) ROW4(R06) ROW4(R07) ROW4(R08) ROW4(R09) ROW4(R10)
                       ROW4(R11) ROW4(R12) ROW4(R13) ROW4(R14)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "=&v"(t0_), "=&v"(t1_),
                     "=&v"(t2_), "=&v"(t3_), "=&v"(t4_), "=&v"(t5_), "=&v"(t6_), "=&v"(t7_), "=&v"(t8_), "=&v"(t9_),
                     "=&v"(t10_), "=&v"(t11_), "=&v"(t12_), "=&v"(t13_), "=&v"(t14_), "=&v"(t15_)
                   : "v"(b), "v"(fb)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s40", "s41", "s42",
                     "s43");
      a4 ^= t2_ ^ t6_ ^ t10_ ^ t14_;
    }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  const uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3);
  if (acc == 0x12345678u) sink[0] = acc + lds_pad[0];   // keep everything alive
  if ((threadIdx.x & 63u) == 0) {
    const size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    cycles[2 * w] = t1 - t0;          // s_memtime: shader-clock ticks
    cycles[2 * w + 1] = r1 - r0;      // s_memrealtime: the constant 100 MHz reference clock
  }
}

typedef void (*kernel_t)(int, uint32_t, uint32_t*, unsigned long long*);
static kernel_t kKernels[K_COUNT] = {valu_kernel<K_ADD>,     valu_kernel<K_FMA>,  valu_kernel<K_CMP>, valu_kernel<K_CNDMASK>,
                                     valu_kernel<K_ADDC>,    valu_kernel<K_ADDLSHL>, valu_kernel<K_WALK>, valu_kernel<K_CMP32>,
                                     valu_kernel<K_CNDMASK32>, valu_kernel<K_ADDC32>, valu_kernel<K_WALK32>};

int main(int argc, char** argv) {
  int only = -1, only_w = -1, iters = 20000;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--kind") && i + 1 < argc) only = atoi(argv[++i]);
    if (!strcmp(argv[i], "--waves") && i + 1 < argc) only_w = atoi(argv[++i]);
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
  }
  int cus = 0, clock_khz = 0, lds_max = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  CHECK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, 0));
  printf("# gfx950 VALU issue microbenchmark: %d CUs, nominal clock %.0f MHz, %d trips per wave\n", cus, clock_khz / 1000.0,
         iters);
  printf("# cyc/instr/SIMD = (s_memtime ticks of the median wave) x waves on the SIMD / instructions the SIMD issued;\n");
  printf("# wall = the same from the launch's event time at the nominal clock.  One block = one wave per SIMD.\n");
  printf("# shader MHz = s_memtime ticks per microsecond of s_memrealtime (100 MHz) over the median wave's span;\n");
  printf("# span = that span / the launch's event time (1.0 = every wave ran from the launch's start to its end)\n");
  printf("%-22s %10s %12s %16s %16s %12s %10s\n", "instruction", "waves/SIMD", "wall ms", "cyc/instr (wall)", "cyc/instr (tick)",
         "shader MHz", "span");
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 4));
  const int wlist[] = {1, 2, 4, 8};
  for (int k = 0; k < K_COUNT; ++k) {
    if (only >= 0 && k != only) continue;
    for (int wi = 0; wi < 4; ++wi) {
      const int w = wlist[wi];
      if (only_w >= 0 && w != only_w) continue;
      // w blocks per CU and no more: each takes 1/w of the CU's 160 KB of LDS (less a little for alignment) - the
      // attribute reports the 64 KB a block gets by default, which would let two "one per CU" blocks share a CU
      (void)lds_max;
      const int lds = (160 * 1024 / w) - 2048;
      CHECK(hipFuncSetAttribute((const void*)kKernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      const int blocks = cus * w;
      unsigned long long* cyc;
      CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4 * 2));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(kKernels[k], dim3(blocks), dim3(256), lds, 0, iters / 8, 17u, sink, cyc);   // warm-up
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kKernels[k], dim3(blocks), dim3(256), lds, 0, iters, 17u, sink, cyc);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> both(blocks * 4 * 2), h(blocks * 4), real(blocks * 4);
      CHECK(hipMemcpy(both.data(), cyc, both.size() * 8, hipMemcpyDeviceToHost));
      for (size_t q = 0; q < h.size(); ++q) h[q] = both[2 * q], real[q] = both[2 * q + 1];
      std::sort(h.begin(), h.end());
      std::sort(real.begin(), real.end());
      const double ticks = (double)h[h.size() / 2];
      const double shader_mhz = ticks / ((double)real[real.size() / 2] / 100.0);     // ticks per microsecond of the wave's span
      const double span_frac = ((double)real[real.size() / 2] / 100.0) / (ms * 1e3);  // the median wave's span over the launch's
      const double instr_per_simd = (double)w * iters * kPerTrip[k];
      const double cyc_wall = ms * 1e-3 * clock_khz * 1e3 / instr_per_simd;
      const double cyc_tick = ticks / instr_per_simd;   // the median wave's span over what its SIMD issued meanwhile
      printf("%-22s %10d %12.3f %16.2f %16.2f %12.0f %10.2f\n", kKindName[k], w, ms, cyc_wall, cyc_tick, shader_mhz, span_frac);
      CHECK(hipFree(cyc));
    }
  }
  CHECK(hipFree(sink));
  return 0;
}
