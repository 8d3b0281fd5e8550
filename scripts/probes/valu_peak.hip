// valu_peak.hip — what the chip's vector ALUs sustain per instruction class (the ceiling bench.py's `roofline` divides by).
//
// Every kernel is one long loop over a block of 64 instructions of ONE class, issued by W waves per SIMD on every SIMD of the device
// (blocks of 256 threads = one wave per SIMD of a CU, n_cu x W blocks), no memory traffic, no dependencies between neighbouring
// instructions (8 accumulators).  Measured with HIP events; the shader clock during the run comes from s_memtime against the constant
// 100 MHz s_memrealtime.  Output: one JSON object with, per class, wave-instructions per second over the whole device and the implied
// SIMD cycles per wave instruction at the measured clock.
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak scripts/probes/valu_peak.hip && /tmp/valu_peak [waves_per_simd ...] > profiles/r04_valu_peak.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one block of 64 instructions; the accumulators a0..a7 and the operands b, c stay in VGPRs
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP64(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I)

enum Cls { ADD_U32, PK_ADD_U16, PK_MAX_I16, PK_SUB_U16_CLAMP, CNDMASK, ALIGNBIT, DPP_ROW_SHR, DPP_WAVE_SHR, BITOP, CMP, LSHL_B64, ADD_CO_PAIR, SWAP, MIX_SALU, BFE, FFBL,
           AND_B32, XOR_B32, MAX_I32, MIN_U32, SUB_U32, LSHL_B32, LSHR_B32, MOV_B32, ADD3, MAX3, CNDMASK_SGPR, CMP_CNDMASK, CMP_SGPR, MAD_U24, PERM, BFI, ADD_E64, ADD_DPP, READLANE, SALU_ONLY, MAX_I32_DPP, MIN3, MED3, OR3, LSHL_ADD,
           OR_B32, LSHL_ADD_U64, MOV_B64, WRITELANE, ASHR_I32, ADD_SDWA, MIN_I32, READFIRSTLANE, MBCNT, NOT_B32, SUBREV, BFREV, MUL_LO, BITOP3, LSHL_OR, LSHR_B64, CMP_EQ_E32, MAX_U32, N_CLS };
static const char* NAME[N_CLS] = {"v_add_u32", "v_pk_add_u16", "v_pk_max_i16", "v_pk_sub_u16 clamp", "v_cndmask_b32", "v_alignbit_b32", "v_mov_b32 dpp row_shr:1",
                                  "v_mov_b32 dpp wave_shr:1", "v_and_or_b32", "v_cmp_gt_i32 (vcc)", "v_lshlrev_b64", "v_add_co_u32 + v_addc_co_u32", "v_swap_b32",
                                  "v_add_u32 + s_add_u32 interleaved (counts the vector half)", "v_bfe_u32", "v_ffbl_b32",
                                  "v_and_b32", "v_xor_b32", "v_max_i32", "v_min_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_add3_u32", "v_max3_i32",
                                  "v_cndmask_b32 (mask in an SGPR pair)", "v_cmp_gt_i32 vcc + v_cndmask_b32 vcc (pairs)", "v_cmp_gt_i32 into an SGPR pair", "v_mad_u32_u24", "v_perm_b32",
                                  "v_bfi_b32", "v_add_u32_e64", "v_add_u32 dpp row_shr:1", "v_readlane_b32 (scalar result)", "s_add_u32 alone (scalar instructions per second)",
                                  "v_max_i32 dpp row_shr:1", "v_min3_i32", "v_med3_i32", "v_or3_b32", "v_lshl_add_u32",
                                  "v_or_b32", "v_lshl_add_u64", "v_mov_b64", "v_writelane_b32", "v_ashrrev_i32", "v_add_u32_sdwa", "v_min_i32", "v_readfirstlane_b32", "v_mbcnt_lo + v_mbcnt_hi", "v_not_b32",
                                  "v_subrev_u32", "v_bfrev_b32", "v_mul_lo_u32", "v_bitop3_b32", "v_lshl_or_b32", "v_lshrrev_b64", "v_cmp_eq_u32 (vcc)", "v_max_u32"};
static const int PER_SLOT[N_CLS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1,  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,  1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1};      // vector instructions per slot of the block

template <int C>
__global__ __launch_bounds__(256) void probe(int iters, unsigned* out, unsigned long long* clk)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x | 1u, c = 3;
  unsigned long long w0 = 0x12345678ull + threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
  unsigned s0 = 1;
  unsigned long long msk = 0x5555555555555555ull ^ blockIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (C == ADD_U32) {
#define I(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == PK_ADD_U16) {
#define I(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == PK_MAX_I16) {
#define I(n) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == PK_SUB_U16_CLAMP) {
#define I(n) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == CNDMASK) {
#define I(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(b) : );
      REP64(I)
#undef I
    } else if constexpr (C == ALIGNBIT) {
#define I(n) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == DPP_ROW_SHR) {
#define I(n) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == DPP_WAVE_SHR) {
#define I(n) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##n) : "v"(b));
      REP64(I)
#undef I
    } else if constexpr (C == BITOP) {
#define I(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
      REP64(I)
#undef I
    } else if constexpr (C == CMP) {
#define I(n) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a##n), "v"(b) : "vcc");
      REP64(I)
#undef I
    } else if constexpr (C == LSHL_B64) {
#define I(n) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w0)); 
      REP64(I)
#undef I
    } else if constexpr (C == ADD_CO_PAIR) {
#define I(n) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a##n), "+v"(c) : "v"(b) : "vcc");
      REP64(I)
#undef I
    } else if constexpr (C == SWAP) {
#define I(n) asm volatile("v_swap_b32 %0, %1" : "+v"(a##n), "+v"(c));
      REP64(I)
#undef I
    } else if constexpr (C == MIX_SALU) {
#define I(n) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(a##n), "+s"(s0) : "v"(b) : "scc");
      REP64(I)
#undef I
    } else if constexpr (C == BFE) {
#define I(n) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a##n));
      REP64(I)
#undef I
    } else if constexpr (C == FFBL) {
#define I(n) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a##n));
      REP64(I)
#undef I
    }
#define SIMPLE(CLS, TXT) else if constexpr (C == CLS) { REP64(I_##CLS) }
#define I_AND_B32(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_XOR_B32(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_MAX_I32(n) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_MIN_U32(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_SUB_U32(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_LSHL_B32(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##n));
#define I_LSHR_B32(n) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a##n));
#define I_MOV_B32(n) asm volatile("v_mov_b32 %0, %1" : "+v"(a##n) : "v"(b));
#define I_ADD3(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_MAX3(n) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_CNDMASK_SGPR(n) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(msk));
#define I_CMP_CNDMASK(n) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a##n) : "v"(b), "v"(c) : "vcc");
#define I_CMP_SGPR(n) asm volatile("v_cmp_gt_i32 %0, %1, %2" : "=s"(msk) : "v"(a##n), "v"(b));
#define I_MAD_U24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_PERM(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_BFI(n) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_ADD_E64(n) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_ADD_DPP(n) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##n) : "v"(b));
#define I_READLANE(n) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a##n));
#define I_SALU_ONLY(n) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
#define I_MAX_I32_DPP(n) asm volatile("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##n) : "v"(b));
#define I_MIN3(n) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_MED3(n) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_OR3(n) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define I_LSHL_ADD(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a##n) : "v"(b));
#define I_OR_B32(n) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_LSHL_ADD_U64(n) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w0) : "v"(w1));
#define I_MOV_B64(n) asm volatile("v_mov_b64 %0, %1" : "+v"(w0) : "v"(w1));
#define I_WRITELANE(n) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a##n) : "s"(s0));
#define I_ASHR_I32(n) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a##n));
#define I_ADD_SDWA(n) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(a##n) : "v"(b));
#define I_MIN_I32(n) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_READFIRSTLANE(n) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a##n));
#define I_MBCNT(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define I_NOT_B32(n) asm volatile("v_not_b32 %0, %0" : "+v"(a##n));
#define I_SUBREV(n) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_BFREV(n) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a##n));
#define I_MUL_LO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define I_BITOP3(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##n) : "v"(b), "v"(c));
#define I_LSHL_OR(n) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a##n) : "v"(b));
#define I_LSHR_B64(n) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w0));
#define I_CMP_EQ_E32(n) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a##n), "v"(b) : "vcc");
#define I_MAX_U32(n) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
    SIMPLE(AND_B32, 0) SIMPLE(XOR_B32, 0) SIMPLE(MAX_I32, 0) SIMPLE(MIN_U32, 0) SIMPLE(SUB_U32, 0) SIMPLE(LSHL_B32, 0) SIMPLE(LSHR_B32, 0) SIMPLE(MOV_B32, 0)
    SIMPLE(ADD3, 0) SIMPLE(MAX3, 0) SIMPLE(CNDMASK_SGPR, 0) SIMPLE(CMP_CNDMASK, 0) SIMPLE(CMP_SGPR, 0) SIMPLE(MAD_U24, 0) SIMPLE(PERM, 0) SIMPLE(BFI, 0)
    SIMPLE(ADD_E64, 0) SIMPLE(ADD_DPP, 0) SIMPLE(READLANE, 0) SIMPLE(SALU_ONLY, 0) SIMPLE(MAX_I32_DPP, 0) SIMPLE(MIN3, 0) SIMPLE(MED3, 0) SIMPLE(OR3, 0) SIMPLE(LSHL_ADD, 0)
    SIMPLE(OR_B32, 0) SIMPLE(LSHL_ADD_U64, 0) SIMPLE(MOV_B64, 0) SIMPLE(WRITELANE, 0) SIMPLE(ASHR_I32, 0) SIMPLE(ADD_SDWA, 0) SIMPLE(MIN_I32, 0) SIMPLE(READFIRSTLANE, 0) SIMPLE(MBCNT, 0)
    SIMPLE(NOT_B32, 0) SIMPLE(SUBREV, 0) SIMPLE(BFREV, 0) SIMPLE(MUL_LO, 0) SIMPLE(BITOP3, 0) SIMPLE(LSHL_OR, 0) SIMPLE(LSHR_B64, 0) SIMPLE(CMP_EQ_E32, 0) SIMPLE(MAX_U32, 0)
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ c ^ s0 ^ (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^ (unsigned)w3 ^ (unsigned)msk;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int C>
static void run(int n_cu, int wps, int iters, unsigned* d_out, unsigned long long* d_clk, std::string& js)
{
  const int grid = n_cu * wps;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<C>, dim3(grid), dim3(256), 0, 0, iters / 8, d_out, d_clk);      // warm-up
  CHECK(hipDeviceSynchronize());
  double best = 1e30; unsigned long long clk[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<C>, dim3(grid), dim3(256), 0, 0, iters, d_out, d_clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) { best = ms; CHECK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost)); }
  }
  const double waves = (double)grid * 4.0;
  const double insts = waves * (double)iters * 64.0 * PER_SLOT[C];
  const double rate = insts / (best * 1e-3);
  const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0;       // s_memrealtime ticks at 100 MHz
  const double simds = (double)n_cu * 4.0;
  const double cyc_per_inst = mhz > 0 ? simds * mhz * 1e6 / rate : 0.0;
  char buf[512];
  snprintf(buf, sizeof(buf), "%s{\"class\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_insts_per_s\": %.4g, \"shader_clock_mhz\": %.0f, \"simd_cycles_per_wave_inst\": %.3f}",
           js.empty() ? "" : ",\n  ", NAME[C], wps, best, rate, mhz, cyc_per_inst);
  js += buf;
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int C>
static void run_all(int n_cu, const std::vector<int>& wps, int iters, unsigned* d_out, unsigned long long* d_clk, std::string& js)
{
  for (int w : wps) run<C>(n_cu, w, iters, d_out, d_clk, js);
  if constexpr (C + 1 < N_CLS) run_all<C + 1>(n_cu, wps, iters, d_out, d_clk, js);
}

int main(int argc, char** argv)
{
  std::vector<int> wps;
  for (int i = 1; i < argc; ++i) wps.push_back(atoi(argv[i]));
  if (wps.empty()) wps = {1, 4, 8};
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned* d_out; unsigned long long* d_clk;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * sizeof(unsigned)));
  CHECK(hipMalloc(&d_clk, 16));
  std::string js;
  run_all<0>(n_cu, wps, 4000, d_out, d_clk, js);
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_rate_khz_reported\": %d,\n \"what\": \"wave64 vector instructions per second over the whole device, one class at a time, W waves per SIMD, 64-instruction blocks x 4000 iterations\",\n \"classes\": [\n  %s\n ]}\n",
         prop.name, prop.gcnArchName, n_cu, n_cu * 4, prop.clockRate, js.c_str());
  return 0;
}
