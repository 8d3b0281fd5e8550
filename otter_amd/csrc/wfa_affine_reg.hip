// wfa_affine_reg.hip — the register-resident tiers of the gap-affine wavefront aligner (gfx950).
//
// Replaces wfa::WFAlignerGapAffine(4,6,2, Alignment, MemoryMed)::alignEnd2End / alignEndsFree + getAlignmentCigar() (reference:
// src/assemble.cpp:50; call sites src/analignments.cpp:25,31,37,268-280) for score-bounded alignments whose diamond of reachable cells fits a
// window of CAP = NW * S2 * 128 diagonals; penalties (2,4,1) after gcd reduction.  Everything else of the chain — the score-bound pass, the
// HBM-row tiers behind these, the generic kernel — is in wfa_affine.hip; the shared backtrace and the provenance layouts in
// wfa_affine_common.hpp.
//
//   * Window index x = k - kbase; lane l of pair-slot g owns the two ADJACENT diagonals x = 128 g + 2 l (even, E) and + 1 (odd, O).  Every
//     quantity of the pair travels as ONE 32-bit word {lo16: E, hi16: O} of signed 16-bit offsets (null = -32768, creeping up by one per
//     gap extension: still negative after 8 K scores).  Per slot six words stay in VGPRs for the whole alignment: M[s-4] and M[s-2] of the
//     current score parity, the same two of the other parity (a score only reads M rows of its own parity; the two sets trade places
//     after every score), I[s-1] and D[s-1].  The slot loop is unrolled at compile time: every register index is static, which slots a
//     score touches is a wave-uniform branch per slot.
//   * The recurrence runs on PACKED 16-bit operations (v_pk_max_i16 / v_pk_add_u16 / v_pk_sub_i16 clamp / v_pk_min_u16), two cells per
//     instruction:  X_I = max(M[s-4], I[s-1]) and X_D = max(M[s-4], D[s-1]) are taken on the pair's own words; I[s] = X_I of the diagonal
//     to the left + 1 and D[s] = X_D of the diagonal to the right are ONE DPP wave shift + ONE 16-bit funnel shift (v_alignbit) each (lane 0
//     / lane 63 take the neighbouring slot's value through the DPP `old` operand); M[s] = max(M[s-2] + 1, I[s], D[s]).
//   * Provenance (shifted layout, wfa_affine_common.hpp): "extension >= open" is the sign of a saturating packed subtraction of the cell's
//     OWN words, the origin of M falls out of min(M[s] - candidate, 1); two bytes per lane and slot visit in one 16-bit store.  Because a
//     cell's byte carries the choice its NEIGHBOUR makes, a score touches the slots that cover [lo - 1, hi + 1].
//   * Cells outside the score's range [lo, hi] but inside a touched slot are computed like any other: every value a cell ever holds is the
//     offset of a real alignment prefix of at most that score, so such a cell can only matter if it lies on an alignment of score <= U —
//     and then it is inside the diamond by its definition.
//   * Sequences: 2 bits per base in LDS, a probe covers 32 bases (two unaligned 64-bit LDS reads + v_alignbit).  A cell whose match run
//     outlives the probe gets a second probe in the same slot visit; what is still running after 64 bases (one slot visit in 100) goes
//     {x, h} to a per-wave LDS queue, is finished in 64-lane batches at the end of the score, and comes back through a 16-bit patch table
//     in the pair layout: one packed max per slot folds it into M[s].
//   * NW == 1: one wave per alignment, four alignments per block, no barrier at all.  NW > 1: NW waves share ONE alignment; pair-slot g
//     belongs to wave g % NW (cyclic, so every wave holds a share of the live diamond at every score), the X_I / X_D words a neighbouring
//     slot needs cross through double-buffered LDS export tables written at the END of a score for the next one — ONE barrier per score,
//     which also publishes the waves' termination candidates.
#include "wfa_affine_common.hpp"
#include "wfa_affine_reg.hpp"
#include <cstdlib>

using namespace otg_affine;

namespace {

typedef short otg_short2 __attribute__((ext_vector_type(2)));
typedef unsigned short otg_ushort2 __attribute__((ext_vector_type(2)));

// compile-time loop: the body sees its index as a constant expression, so register arrays are only ever indexed statically
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ otg_short2 as_s2(uint32_t a) { otg_short2 x; __builtin_memcpy(&x, &a, 4); return x; }
__device__ __forceinline__ otg_ushort2 as_u2(uint32_t a) { otg_ushort2 x; __builtin_memcpy(&x, &a, 4); return x; }
__device__ __forceinline__ uint32_t from_s2(otg_short2 x) { uint32_t o; __builtin_memcpy(&o, &x, 4); return o; }
__device__ __forceinline__ uint32_t from_u2(otg_ushort2 x) { uint32_t o; __builtin_memcpy(&o, &x, 4); return o; }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { return from_s2(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) { return from_u2(as_u2(a) + as_u2(b)); }
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return from_u2(as_u2(a) - as_u2(b)); }
__device__ __forceinline__ uint32_t pk_sub_sat_i16(uint32_t a, uint32_t b) { return from_s2(__builtin_elementwise_sub_sat(as_s2(a), as_s2(b))); }
// min(a - b, 1) per half = "a != b" as 0 / 1 — written as instructions: hipcc recognises the idiom and turns it into two 16-bit compares, two
// selects and a pack
__device__ __forceinline__ uint32_t pk_ne01_u16(uint32_t a, uint32_t b)
{
  uint32_t d, r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(d));
  return r;
}
__device__ __forceinline__ void opaque_v(int& x) { asm volatile("" : "+v"(x)); }
// lane masks of comparisons (v_cmp into an SGPR pair) and selects on a lane mask — see the slot visit
__device__ __forceinline__ unsigned long long mk_eq(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 32); }
__device__ __forceinline__ unsigned long long mk_gt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 38); }
__device__ __forceinline__ unsigned long long mk_ule(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 37); }
// (r04, measured and not kept: the selects on vcc — "s_mov_b64 vcc, mask; v_cndmask_b32_e32", a 2.2-cycle vector instruction + a scalar move instead of the
//  4.2-cycle e64 form on an SGPR pair: affine stage of a 2 500-region slice of configs[1] 105.2 -> 107.1 ms; the scalar unit is the scarcer resource here)
__device__ __forceinline__ int sel(unsigned long long m, int a, int b) { int r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m)); return r; }     // lane in m ? a : b
__device__ __forceinline__ int sel0(unsigned long long m, int a) { int r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m)); return r; }                   // lane in m ? a : 0
__device__ __forceinline__ bool lane_in(unsigned long long m, int lane) { return ((m >> lane) & 1ull) != 0ull; }
__device__ __forceinline__ void opaque_v(uint32_t& x) { asm volatile("" : "+v"(x)); }
// In-place update of a loop-carried wavefront word.  Written as a plain assignment, the conditional slot visit leaves a phi per word at its join,
// and the allocator resolves part of them with copies at the loop header (dozens of v_mov per score) and spills; tied to its register, a word
// never moves.
__device__ __forceinline__ void vset(uint32_t& dst, uint32_t src) { asm("v_mov_b32 %0, %1" : "+v"(dst) : "v"(src)); }
__device__ __forceinline__ void vswap(uint32_t& a, uint32_t& b) { asm("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }   // (hipcc refuses asm operands that are lambda captures)
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }   // {lo16: lo, hi16: hi}
__device__ __forceinline__ uint64_t uniform64(uint64_t x)
{
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}

// Scheduling fence: the two probes of a lane are independent, and left alone the scheduler interleaves them with each other and with the
// recurrence of the slot — twice the live temporaries, which in the 12- and 16-slot bodies turn into spills of the wavefront words themselves.
#ifndef OTG_SCHED_FENCE
#define OTG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

template <int NW, int S2, int SEQB, int WPEU, int QCAP = 512, int LB = 2>
__global__ __launch_bounds__(NW == 1 ? 256 : NW * 64, WPEU) void wfa_affine_reg_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ seg, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, const int32_t* __restrict__ bound, unsigned long long* __restrict__ visited)
{
  constexpr int xs = 2, oes = 4, es = 1;
  constexpr int CAP = NW * S2 * 128;
  constexpr int ALN = NW == 1 ? 4 : 1;              // alignments per block
  constexpr int WAVES = NW == 1 ? 4 : NW;           // waves per block
  constexpr int GS = NW * S2;                        // pair-slots of the window
  constexpr int FAILV = -2147483647 - 1;
  constexpr int NOCAND = 0x7fffffff;
  constexpr uint32_t NN = 0x80008000u;
  constexpr uint32_t ONE2 = 0x00010001u;
  constexpr int NUL16 = -32768;
  __shared__ uint32_t s_seq[ALN][SEQB / 4];
  __shared__ uint32_t s_patch[ALN][CAP / 2];
  __shared__ uint32_t s_queue[WAVES][QCAP];
  // (NW > 1) what the waves hand each other at the end of a score, double-buffered by score parity: [0, GS + 2) entry g + 1 = X_I of slot g's lane 63,
  // [GS + 2, 2 GS + 4) entry g + 1 = X_D of slot g's lane 0, then one termination candidate per wave
  constexpr int XROW = 2 * (GS + 2) + WAVES;
  __shared__ uint32_t s_x[2 * XROW + 64];       // (+ 64: a row is read back by all 64 lanes)
  __shared__ int s_misc[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int al = NW == 1 ? wv : 0;                  // which alignment of the block this wave works on
  const int ww = NW == 1 ? 0 : wv;                  // its rank among the waves of that alignment
  uint32_t* SQ = &s_seq[al][0];
  volatile lds_u32* PT = (volatile lds_u32*)&s_patch[al][0];
  volatile lds_u32* QU = (volatile lds_u32*)&s_queue[wv][0];
  volatile lds_u16* PT16 = (volatile lds_u16*)&s_patch[al][0];
  volatile lds_u32* XT = (volatile lds_u32*)&s_x[0];
  volatile __attribute__((address_space(3))) int* MISC = (volatile __attribute__((address_space(3))) int*)&s_misc[0];
  uint8_t* my = ws.base + (size_t)(blockIdx.x * ALN + al) * ws.stride;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  const uint32_t seg0 = seg[0], n_todo = seg[1] - seg[0];

  for (;;) {
    uint32_t tk;
    if (NW > 1) {
      if (wv == 0) MISC[0] = (int)otg_wave_atomic_add(ticket, 1u);
      __syncthreads();
      tk = (uint32_t)MISC[0];
    } else tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    // The descriptor is wave-uniform and must live in SGPRs: everything the score loop branches on derives from it.  Whether hipcc turns
    // these loads into scalar loads depends on the size of the instantiation (its clobber walk over the persistent loop gives up on the
    // larger bodies, and the whole score loop then runs on vector compares and exec masks), so every field is pinned explicitly.
    const uint32_t ti = (uint32_t)__builtin_amdgcn_readfirstlane((int)todo[seg0 + tk]);
    otg_align_task t = tasks[ti];
    t.pattern_off = uniform64(t.pattern_off); t.text_off = uniform64(t.text_off);
    t.pattern_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.pattern_len); t.text_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.text_len);
    t.pattern_begin_free = __builtin_amdgcn_readfirstlane(t.pattern_begin_free); t.pattern_end_free = __builtin_amdgcn_readfirstlane(t.pattern_end_free);
    t.text_begin_free = __builtin_amdgcn_readfirstlane(t.text_begin_free); t.text_end_free = __builtin_amdgcn_readfirstlane(t.text_end_free);
    t.endsfree = __builtin_amdgcn_readfirstlane(t.endsfree);
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = t.pattern_end_free, tef = t.text_end_free;
    const int kend = tl - pl;
    const int U = __builtin_amdgcn_readfirstlane(bound[ti]);
    const int elo = kend - (ef ? tef : 0), ehi = kend + (ef ? pef : 0);
    int lo0 = 0, hi0 = 0, kbase = 0;
    bool fail = U < 0 || U >= 0x40000000 || pl >= 32766 || tl >= 32766;
    const int offT = (pl + 15) / 16 + 3;
    if ((offT + (tl + 15) / 16 + 3) * 4 > SEQB) fail = true;
    if (!fail) {
      int need = 0;
      if (!affine_window(t, U, &kbase, &need, &lo0, &hi0) || need >= CAP) fail = true;
    }
    if (!fail) {
      for (int q = (NW == 1 ? lane : (int)threadIdx.x); q < CAP / 2; q += (NW == 1 ? 64 : NW * 64)) PT[q] = NN;
      if (NW > 1) for (int q = (int)threadIdx.x; q < 2 * XROW; q += NW * 64) XT[q] = NN;
      bool bad = false;
      auto pack = [&](const uint8_t* S, int len, int woff) {
        for (int q = (NW == 1 ? lane : (int)threadIdx.x); q < (len + 15) / 16; q += (NW == 1 ? 64 : NW * 64)) {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int b0 = 16 * q + 8 * j;
            const uint64_t x = b0 < len + 8 ? otg_load8(S + (b0 < len ? b0 : len)) : 0ull;
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) {
              const uint32_t c = (uint32_t)(x >> (8 * t2)) & 0xffu;
              const uint32_t code = (c >> 1) & 3u;
              if (b0 + t2 < len && c != ((0x47544341u >> (8 * code)) & 0xffu)) bad = true;
              w |= code << (2 * (8 * j + t2));
            }
          }
          SQ[woff + q] = w;
        }
      };
      pack(P, pl, 0);
      pack(T, tl, offT);
      // slack words a probe may read past the packed ends
      if ((NW == 1 ? lane : (int)threadIdx.x) < 3) { const int q3 = NW == 1 ? lane : (int)threadIdx.x; SQ[(pl + 15) / 16 + q3] = 0; SQ[offT + (tl + 15) / 16 + q3] = 0; }
      if (NW > 1) { if (threadIdx.x == 0) MISC[2] = 0; __syncthreads(); if (bad) MISC[2] = 1; __syncthreads(); fail = MISC[2] != 0; }
      else fail = __ballot(bad) != 0ull;
    }
    if (NW > 1) __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    auto ld32b = [&](int woff, int pos) -> uint64_t {
      const int w = woff + (pos >> 4);
      const uint32_t sh = (uint32_t)(pos & 15) * 2u;
      const uint32_t d0 = SQ[w], d1 = SQ[w + 1], d2 = SQ[w + 2];
      return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
    };
    // A probe in two halves: the three words of each sequence around a position are requested first — for BOTH cells of a lane before anything is
    // combined, so a slot visit waits for LDS once, not four times in a row — and shifted into place after a scheduling fence.
    struct Ld3 { uint32_t d0, d1, d2; };
    auto ld3 = [&](int woff, int pos) -> Ld3 { const int w = woff + (pos >> 4); return Ld3{SQ[w], SQ[w + 1], SQ[w + 2]}; };
    auto cat32 = [&](const Ld3& a, int pos) -> uint64_t {
      const uint32_t sh = (uint32_t)(pos & 15) * 2u;
      return (uint64_t)__builtin_amdgcn_alignbit(a.d1, a.d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(a.d2, a.d1, sh) << 32);
    };
    auto probe_of = [&](const Ld3& p, int v, const Ld3& t, int h) -> int {
      const uint64_t x = cat32(p, v) ^ cat32(t, h);
      uint32_t flo, fhi;      // v_ffbl_b32: index of the lowest set bit, 0xffffffff for zero — which is what the min below wants
      asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"((uint32_t)x));
      asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"((uint32_t)(x >> 32)));
      return (int)(umin3(flo, fhi | 32u, 64u) >> 1);
    };
    // equal leading bases of pattern[v ..] and text[h ..], looking 32 bases ahead (32 = all equal); not limited by the sequence ends
    auto probe32 = [&](int v, int h) -> int {
      const uint64_t x = ld32b(0, v) ^ ld32b(offT, h);
      uint32_t flo, fhi;      // v_ffbl_b32: index of the lowest set bit, 0xffffffff for zero — which is what the min below wants
      asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"((uint32_t)x));
      asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"((uint32_t)(x >> 32)));
      const uint32_t f = umin3(flo, fhi | 32u, 64u);
      return (int)(f >> 1);
    };
    // equal leading bases looking at most 32 * nb bases ahead (and at most rem); a rolled loop: the register arrays of the sweep stay live
    // across the drain, so this must not turn into nb independent probes in flight
    auto match_n = [&](int v, int h, int rem, int nb) -> int {
      int m = 0;
#pragma nounroll
      for (int i = 0; i < nb; ++i) {
        if (m >= rem) break;
        const uint64_t x = ld32b(0, v + m) ^ ld32b(offT, h + m);
        if (x) { m += (int)(__builtin_ctzll(x) >> 1); break; }
        m += 32;
      }
      return m < rem ? m : rem;
    };
    auto wave_match = [&](int v, int h, int rem) -> int {
      int total = 0;
      while (total < rem) {
        const int off = total + lane * 32;
        uint64_t x = ~0ull;
        if (off < rem) x = ld32b(0, v + off) ^ ld32b(offT, h + off);
        const int m = x ? (int)(__builtin_ctzll(x) >> 1) : 32;
        const unsigned long long stop = __ballot(m < 32);
        if (stop) { const int f = (int)__builtin_ctzll(stop); total += f * 32 + __builtin_amdgcn_readlane(m, f); break; }
        total += 2048;
      }
      return total < rem ? total : rem;
    };

    uint32_t M4[2][S2], M2[2][S2];     // [parity set][pair-slot] = {lo16: even diagonal, hi16: odd diagonal}; set 0 = the parity of the current score: M[s-4], M[s-2]
    uint32_t WI[S2], WD[S2];           // I[s-1], D[s-1]
    static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
      M4[0][i] = NN; M2[0][i] = NN; M4[1][i] = NN; M2[1][i] = NN; WI[i] = NN; WD[i] = NN;
      opaque_v(M4[0][i]); opaque_v(M2[0][i]); opaque_v(M4[1][i]); opaque_v(M2[1][i]); });     // vector registers from the start (the swaps at the loop top are tied to them)
    uint32_t slab_top = 0;                       // (a slab is a few hundred megabytes at most: 32-bit bookkeeping on the scalar unit)
    const uint32_t slab_cap = ws.slab_bytes > 0xfff00000ull ? 0xfff00000u : (uint32_t)ws.slab_bytes;
    int s_end = -1, k_end = 0;
    int r1lo = 1, r1hi = 0, r2lo = 1, r2hi = 0, r3lo = 1, r3hi = 0, r4lo = 1, r4hi = 0, idlo = 1, idhi = 0;
    uint32_t xlv = NN, xrv = NN;                 // (NW > 1) lane g: X_I of slot g - 1 / X_D of slot g + 1 as exported at the end of the previous score
    const int xe = kend - kbase;                 // window index of the end diagonal (end-to-end termination)

    int lane2 = 2 * lane, kb = __builtin_amdgcn_readfirstlane(kbase);
    const int nul16v = NUL16;
#ifdef OTG_REG_TIMING   // where a wave's cycles go, per section of a score (s_memtime; sums over all waves land behind the visited-cell counter)
    unsigned long long tm_pre = 0, tm_sweep = 0, tm_drain = 0, tm_exp = 0, tm_bar = 0, tm_n = 0, tm_vis = 0, tm_last = __builtin_amdgcn_s_memtime();
#define OTG_TM(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tm_last; tm_last = now_; }
#else
#define OTG_TM(acc)
#endif
    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      // opaque to the optimiser: per-slot expressions built on these are recomputed where they are used instead of being hoisted out of
      // the score loop into S2 live registers each (loop-invariant code motion knows nothing about register pressure)
      asm volatile("" : "+v"(lane2));
      asm volatile("" : "+s"(kb));
      int lo, hi;
      if (s == 0) { lo = lo0; hi = hi0; }
      else {
        lo = 1 << 30; hi = -(1 << 30);
        if (r2hi >= r2lo) { lo = imin(lo, r2lo); hi = imax(hi, r2hi); }
        if (r4hi >= r4lo) { lo = imin(lo, r4lo - 1); hi = imax(hi, r4hi + 1); }
        if (idhi >= idlo) { lo = imin(lo, idlo - 1); hi = imax(hi, idhi + 1); }
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
        if (hi >= lo) {
          const int room = U - s;
          lo = imax(lo, elo - room); hi = imin(hi, ehi + room);
          if (room < 0 || hi < lo) { fail = true; break; }
        }
      }
      lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);     // wave-uniform by construction: say so, the whole score loop stays scalar
      r4lo = r3lo; r4hi = r3hi; r3lo = r2lo; r3hi = r2hi; r2lo = r1lo; r2hi = r1hi;
      int cand = NOCAND;
      int j0 = 1, j1 = 0;                        // touched pair-slots of this score (none when the score is unreachable)
      if (hi < lo) {   // unreachable score: nothing is written, the parity sets still trade places
        r1lo = 1; r1hi = 0; idlo = 1; idhi = 0;
        rowtab[s] = -1;
        if (s > 2 * (oes + es * (pl + tl)) + 8) { fail = true; break; }
      } else {
      r1lo = lo; r1hi = hi;
      const int xlo = lo - kbase, xhi = hi - kbase;
      if (xlo < 2 || xhi + 3 >= CAP) { fail = true; break; }
      j0 = (xlo - 1) >> 7; j1 = (xhi + 1) >> 7;  // one diagonal of margin on both sides: a cell's provenance byte carries its neighbours' gap choices
      const int width = (j1 - j0 + 1) * 128;
      if (slab_top + (uint32_t)width > slab_cap) { fail = true; break; }
      uint8_t* brow = slab + slab_top - 128 * j0;                  // provenance byte of window index x: brow[x]
      rowtab[s] = (int64_t)slab_top - (int64_t)(kbase + 128 * j0);  // wave-uniform store (same value from every lane and every wave)
      slab_top += (uint32_t)width;
      int qn = 0;
      // ---- drain: queued cells {x | h << 16} are extended to the end of their match run in 64-lane batches; final offsets go to the patch table
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int e = 0; e < qn; ++e) {
              const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)QU[e]);      // same address in every lane: keep it (and all that follows from it) scalar
              const int x = (int)(ent & 0xffffu), h = (int)(ent >> 16), kk = kbase + x, v = h - kk;
              const int m = wave_match(v, h, imin(pl - v, tl - h));
              const int hf = h + m, vf = v + m;
              PT16[x] = (uint16_t)hf;
              if (ef ? ((hf >= tl && pl - vf <= pef) || (vf >= pl && tl - hf <= tef)) : (x == xe && hf >= tl)) cand = imin(cand, kk);
            }
            qn = 0;
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int x = 0, kk = 0, h = 0, v = 0;
            bool more = false, fin = false;
            if (act) {
              const uint32_t ent = QU[q0 + lane];
              x = (int)(ent & 0xffffu); h = (int)(ent >> 16); kk = kbase + x; v = h - kk;
              const int rem = imin(pl - v, tl - h);
              int m, full;
              if (pass == 0) { m = match_n(v, h, rem, 2); full = 64; }
              else { m = match_n(v, h, rem, 8); full = 256; }
              v += m; h += m;
              more = (m == full) && v < pl && h < tl;
              if (!more) {
                PT16[x] = (uint16_t)h;
                fin = ef ? ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)) : (x == xe && h >= tl);
              }
            }
            const unsigned long long fm = __ballot(fin);
            if (fm) {                                            // lowest diagonal among this batch's finishing cells
              int kc = fin ? kk : NOCAND;
              kc = -otg_wave_max_i32(-kc);
              cand = imin(cand, kc);
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              QU[wq + rank] = (uint32_t)x | ((uint32_t)h << 16);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      int pushed = 0, qfull = 0;
      auto push2 = [&](bool moreE, bool moreO, unsigned long long mE, unsigned long long mO, int xE, int hE, int hO) {     // mE / mO = the lane masks of moreE / moreO
        if (mE | mO) {
          if (qn + 128 > QCAP) { qfull = 1; return; }          // more unfinished match runs in one score than the queue holds: the next tier takes the alignment
          const int nE = __builtin_popcountll(mE);
          if (moreE) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mE >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mE, 0u));
            QU[qn + rank] = (uint32_t)xE | ((uint32_t)hE << 16);
          }
          if (moreO) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mO >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mO, 0u));
            QU[qn + nE + rank] = (uint32_t)(xE + 1) | ((uint32_t)hO << 16);
          }
          qn += nE + __builtin_popcountll(mO);
          pushed = 1;
        }
      };
      // end condition of a fully extended cell (ends-free form; the end-to-end form is checked on the one end diagonal)
      auto fin_ef = [&](int h, int v) -> bool { return h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)); };

      OTG_TM(tm_pre);
      if (s == 0) {
        // score 0: offset max(k, 0) on every start diagonal of the range, nothing elsewhere, no I / D wavefronts.  Written as "M[-2]" = that offset
        // minus one, so that the sweep below makes M[0] out of it like any other score (mismatch term M[s-2] + 1, then the probes): an ends-free
        // alignment can have thousands of start diagonals, and pushed through the queue one by one — as this branch once did — more than 384 of
        // them sent the alignment to the HBM-row tier behind the register tiers (0.1 % of the alignments, 5 % of the time of a small batch).
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (gi < j0 || gi > j1) return;
          const int xE = 128 * gi + lane2, kE = kb + xE, kO = kE + 1;
          const int hE = kE > 0 ? kE : 0, vE = hE - kE, hO = kO > 0 ? kO : 0, vO = hO - kO;
          const bool validE = kE >= lo && kE <= hi && hE <= tl && vE <= pl, validO = kO >= lo && kO <= hi && hO <= tl && vO <= pl;
          M2[0][i] = pack16(validE ? hE - 1 : NUL16, validO ? hO - 1 : NUL16);
        });
      }
      {
        // ---- the sweep over the touched pair-slots, ascending
        uint32_t carryI = NN;                     // X_I of lane 63 of the slot left of the current one, from before that slot's update (NW == 1)
        uint32_t XDc = NN;                        // X_D of the next slot, computed one visit ahead (NW == 1)
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (NW == 1 && gi + 1 == j0)             // the slot left of the first touched one: its lane 63 is the left neighbour of the sweep
            carryI = (uint32_t)__builtin_amdgcn_readlane((int)pk_max_i16(M4[0][i], WI[i]), 63);
          if (gi < j0 || gi > j1) return;
          opaque_v(lane2);                         // nothing derived from the lane's window index is computed ahead of its slot visit (one 64-bit store address per slot otherwise)
          const uint32_t m4 = M4[0][i], m2 = M2[0][i], wi = WI[i], wd = WD[i];
          // what the pair offers its neighbours, and which of the two it is (sign clear = the extension wins or ties)
          const uint32_t XI = pk_max_i16(m4, wi), fI = pk_sub_sat_i16(wi, m4);
          const uint32_t fD = pk_sub_sat_i16(wd, m4);
          uint32_t XD, lcar, rcar;
          if (NW == 1) {
            XD = gi == j0 ? pk_max_i16(m4, wd) : XDc;
            lcar = carryI; rcar = NN;
            if constexpr (i + 1 < S2) { XDc = pk_max_i16(M4[0][i + 1], WD[i + 1]); rcar = (uint32_t)__builtin_amdgcn_readlane((int)XDc, 0); }   // the next slot's lane 0, still old
          } else {
            XD = pk_max_i16(m4, wd);
            lcar = (uint32_t)__builtin_amdgcn_readlane((int)xlv, gi);                       // slot gi - 1
            rcar = (uint32_t)__builtin_amdgcn_readlane((int)xrv, gi);                       // slot gi + 1
          }
          const uint32_t XIl = (uint32_t)__builtin_amdgcn_update_dpp((int)lcar, (int)XI, 0x138, 0xf, 0xf, false);   // lane l <- lane l-1, lane 0 <- the left slot
          const uint32_t XDr = (uint32_t)__builtin_amdgcn_update_dpp((int)rcar, (int)XD, 0x130, 0xf, 0xf, false);   // lane l <- lane l+1, lane 63 <- the right slot
          if (NW == 1) carryI = (uint32_t)__builtin_amdgcn_readlane((int)XI, 63);
          // I[s] = {X_I of the left lane's odd diagonal, X_I of the own even diagonal} + 1;  D[s] = {X_D of the own odd one, X_D of the right lane's even one}
          const uint32_t Inew = pk_add_u16(__builtin_amdgcn_alignbit(XI, XIl, 16), ONE2);
          const uint32_t Dnew = __builtin_amdgcn_alignbit(XDr, XD, 16);
          const uint32_t mis = pk_add_u16(m2, ONE2);
          const uint32_t Mx = pk_max_i16(pk_max_i16(mis, Inew), Dnew);
          // provenance: origin of M (mismatch wins ties over deletion over insertion) + the two "extension >= open" bits of the pair's own words
          const uint32_t neM = pk_ne01_u16(Mx, mis), neD = pk_ne01_u16(Mx, Dnew);
          uint32_t bw = neM + (neM & neD);
          bw = ((fI >> 13) & 0x00040004u) | bw;
          bw = ((fD >> 12) & 0x00080008u) | bw;
          bw ^= 0x000C000Cu;
          const uint16_t b2 = (uint16_t)__builtin_amdgcn_perm(bw, bw, 0x0c0c0200u);
          const int xE = 128 * gi + lane2, kE = kb + xE;
          __builtin_memcpy(brow + 128 * gi + (uint32_t)lane2, &b2, 2);     // uniform base + unsigned lane offset: no 64-bit address arithmetic per lane
          WI[i] = Inew; WD[i] = Dnew;
          int hE = (int)(int16_t)(Mx & 0xffffu), hO = (int)Mx >> 16;
          const int vE = hE - kE, vO = hO - kE - 1;
          // From here on every per-lane condition is a 64-bit LANE MASK in scalar registers: comparisons deliver one (v_cmp into an SGPR pair,
          // mk_*), the logic runs on the scalar unit, "does any lane ..." is a scalar test, and selects take the mask as their condition
          // (v_cndmask, sel).  Written with bools, every wave-level test came out as a select and a compare on the vector unit in front of its branch.
          const unsigned long long vmE = mk_ule((uint32_t)hE, (uint32_t)tl) & mk_ule((uint32_t)vE, (uint32_t)pl);
          const unsigned long long vmO = mk_ule((uint32_t)hO, (uint32_t)tl) & mk_ule((uint32_t)vO, (uint32_t)pl);
          // Probes read wherever the offsets point: an invalid cell (null or past an end) yields an LDS address outside the pair — possibly outside
          // the block's allocation, where reads return zero — and its result is dropped by the `valid` selects below; a valid cell at an end of
          // a sequence gets m = 0 from the remaining lengths.  No clamps, no gating of the probe itself.
          const int rvE = pl - vE, rhE = tl - hE, rvO = pl - vO, rhO = tl - hO;
          // LB = how many of the lane's two probes have their LDS requests in flight together: 2 = both (one wait per slot visit; twelve more live
          // registers), 1 = one probe at a time (two waits), 0 = pattern and text one after the other (four)
          int mE, mO;
          if constexpr (LB == 2) {
            const Ld3 pE = ld3(0, vE), tE = ld3(offT, hE), pO = ld3(0, vO), tO = ld3(offT, hO);
            OTG_SCHED_FENCE();
            mE = imin(probe_of(pE, vE, tE, hE), imin(rvE, rhE));
            mO = imin(probe_of(pO, vO, tO, hO), imin(rvO, rhO));
            OTG_SCHED_FENCE();
          } else if constexpr (LB == 1) {
            OTG_SCHED_FENCE();
            { const Ld3 pE = ld3(0, vE), tE = ld3(offT, hE); OTG_SCHED_FENCE(); mE = imin(probe_of(pE, vE, tE, hE), imin(rvE, rhE)); }
            OTG_SCHED_FENCE();
            { const Ld3 pO = ld3(0, vO), tO = ld3(offT, hO); OTG_SCHED_FENCE(); mO = imin(probe_of(pO, vO, tO, hO), imin(rvO, rhO)); }
            OTG_SCHED_FENCE();
          } else {
            OTG_SCHED_FENCE();
            mE = imin(probe32(vE, hE), imin(rvE, rhE));
            OTG_SCHED_FENCE();
            mO = imin(probe32(vO, hO), imin(rvO, rhO));
            OTG_SCHED_FENCE();
          }
          hE += mE; hO += mO;
          // more = the run is still going after 32 bases (valid cell, full probe, more than 32 bases left of both sequences)
          unsigned long long mmE = vmE & mk_eq(mE, 32) & mk_gt(rvE, 32) & mk_gt(rhE, 32), mmO = vmO & mk_eq(mO, 32) & mk_gt(rvO, 32) & mk_gt(rhO, 32);
          // a second probe where a run outlives the first (one in 120 cells at ONT divergence, i.e. most slot visits have one): the queue, its
          // drain and the fold-back of the patch table — a fixed cost per score — are then left to runs beyond 64 bases (one slot visit in 100)
          if ((mmE | mmO) != 0ull) {
            // one probe sequence for both cells of the lane: it extends the even cell if that one needs it, else the odd one (a lane where both do —
            // one in 15 000 — leaves the odd cell to the queue)
            const unsigned long long selO = mmO & ~mmE;
            const int h2 = sel(selO, hO, hE);
            const int v2 = h2 - sel(selO, kE + 1, kE);
            const int rv2 = pl - v2, rh2 = tl - h2;
            int m2nd;
            if constexpr (LB > 0) { const Ld3 p2 = ld3(0, v2), t2 = ld3(offT, h2); OTG_SCHED_FENCE(); m2nd = imin(probe_of(p2, v2, t2, h2), imin(rv2, rh2)); }
            else m2nd = imin(probe32(v2, h2), imin(rv2, rh2));
            const unsigned long long m2m = mk_eq(m2nd, 32) & mk_gt(rv2, 32) & mk_gt(rh2, 32);
            hO += sel0(selO, m2nd); hE += sel0(mmE, m2nd);
            mmO &= mmE | m2m; mmE &= m2m;             // (the odd cell of a lane whose even cell took the probe stays as it was)
          }
          M4[0][i] = m2;
          M2[0][i] = pack16(sel(vmE, hE, nul16v), sel(vmO, hO, nul16v));
          if (ef) {
            // a (valid) cell can end the alignment only where it has reached the end of a sequence: h == tl, or v == pl i.e. h == pl + k
            const int hp = pl + kE;
            if ((mk_eq(hE, tl) | mk_eq(hE, hp) | mk_eq(hO, tl) | mk_eq(hO, hp + 1)) != 0ull) {
              const bool fE = lane_in(vmE & ~mmE, lane) && fin_ef(hE, hE - kE), fO = lane_in(vmO & ~mmO, lane) && fin_ef(hO, hO - kE - 1);
              const unsigned long long fm = __ballot(fE || fO);
              if (fm) { int kc = fE ? kE : (fO ? kE + 1 : NOCAND); kc = -otg_wave_max_i32(-kc); cand = imin(cand, kc); }
            }
          } else if (xe >= 128 * gi && xe < 128 * gi + 128) {
            const bool odd = (xe & 1) != 0;
            const int hx = __builtin_amdgcn_readlane(sel(odd ? (vmO & ~mmO) : (vmE & ~mmE), odd ? hO : hE, -1), (xe & 127) >> 1);
            if (hx >= tl) cand = kend;
          }
          if ((mmE | mmO) != 0ull) push2(lane_in(mmE, lane), lane_in(mmO, lane), mmE, mmO, xE, hE, hO);
        });
      }
      if (s == 0) static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; M4[0][i] = NN; opaque_v(M4[0][i]); });     // (what stood in for M[-2])
      OTG_TM(tm_sweep);
#ifdef OTG_REG_TIMING
      tm_n += 1; tm_vis += (unsigned long long)(NW == 1 ? (j1 - j0 + 1) : ((j1 - ww + NW) / NW - (j0 - ww + NW - 1) / NW));
#endif
      if (qfull) cand = FAILV;
      else if (pushed) {
        drain();
        // fold the final offsets of the queued cells into M[s] (partial offset <= final offset: one packed max; the table is in the pair layout)
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (gi < j0 || gi > j1) return;
          const int pi = 64 * gi + (lane2 >> 1);   // (from the opaque lane index: an address per slot is not kept live across the score loop)
          const uint32_t pw = PT[pi];
          if (__ballot(pw != NN)) {
            M2[0][i] = pk_max_i16(M2[0][i], pw);
            PT[pi] = NN;
          }
        });
      }
      idlo = s == 0 ? 1 : lo; idhi = s == 0 ? 0 : hi;
      OTG_TM(tm_drain);
      }
      // ---- close the score: (NW > 1) publish what the neighbouring slots need for score s + 1 — X_I / X_D from the M[s-3] of the OTHER parity
      // set, which is the current one of the next score, and the I / D words just written — and this wave's candidate; one barrier; then every
      // wave sees every candidate
      cand = __builtin_amdgcn_readfirstlane(cand);
      if (NW > 1) {
        // Every slot is exported, touched or not (a slot's X words change with the score parity even while it rests), by its two edge lanes under ONE
        // execution mask; after the barrier one row read per table serves all slot visits of the next score and one more the candidates.  (Written slot
        // by slot with a lane test and a range test each, and read back one word per slot visit, this section took a fifth of the four-wave tier's time.)
        const int np = ((s + 1) & 1) * XROW;
        if (lane == 0 || lane == 63) {
          volatile lds_u32* dst = XT + np + (lane == 63 ? 0 : GS + 2) + ww + 1;
          static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
            dst[i * NW] = pk_max_i16(M4[1][i], lane == 63 ? WI[i] : WD[i]); });
          if (lane == 0) XT[np + 2 * (GS + 2) + ww] = (uint32_t)cand;
        }
        OTG_TM(tm_exp);
        __syncthreads();
        xlv = XT[np + lane];                      // entry g = slot g - 1
        xrv = XT[np + GS + 4 + lane];             // entry GS + 2 + g + 2 = slot g + 1
        const int cv = (int)XT[np + 2 * (GS + 2) + (lane & (WAVES - 1))];
        int gc = NOCAND;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) { const int c2 = __builtin_amdgcn_readlane(cv, w2); gc = c2 < gc ? c2 : gc; }
        cand = __builtin_amdgcn_readfirstlane(gc);
        OTG_TM(tm_bar);
      }
      if (cand == FAILV) fail = true;
      else if (cand != NOCAND) { s_end = s; k_end = cand; }
      // the other parity is next.  Unconditional, also in the last pass: with the two exits above in front of them the same sixteen swaps came
      // wrapped in a copy of the whole state out of its registers and back (80 instructions per score)
      static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
        vswap(M4[0][i], M4[1][i]);
        vswap(M2[0][i], M2[1][i]); });
      if (s_end >= 0) break;
    }

#ifdef OTG_REG_TIMING
    if (visited && lane == 0) {
      atomicAdd(visited + 2, tm_pre); atomicAdd(visited + 3, tm_sweep); atomicAdd(visited + 4, tm_drain); atomicAdd(visited + 5, tm_exp);
      atomicAdd(visited + 6, tm_bar); atomicAdd(visited + 7, tm_n); atomicAdd(visited + 8, tm_vis);
    }
#endif
    if (NW > 1 && wv != 0) continue;           // wave 0 reports / unpacks; the others wait at the next ticket barrier
    if (fail || s_end < 0) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;                                   // wave-uniform store
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    if (!backtrace_unpack<true>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g,
                                (volatile lds_u32*)&s_queue[wv][0], EqPacked{(volatile lds_u32*)&s_seq[al][0], offT})) continue;
    if (cells) cells[ti] = affine_cells(t, xs, oes, s_end);
    if (visited && lane == 0) atomicAdd(visited, (unsigned long long)slab_top);
  }
}

} // namespace

void otg_affine_reg_geometry(int tier, int shape, int* aln_per_block, int* blocks_per_cu)
{
  int a = 4, b = 4;
  switch (tier * 10 + shape) {
    case 0: a = 4; b = 4; break;        // <1,8>   16 waves per CU
    case 1: a = 1; b = 10; break;       // <2,4>   20
    case 10: a = 4; b = 3; break;       // <1,12>  12 (no spills)
    case 11: a = 4; b = 4; break;       // <1,12>  16 (32 spilled registers)
    case 12: a = 1; b = 8; break;       // <2,6>   16
    case 20: a = 1; b = 8; break;       // <2,8>   16
    case 21: a = 4; b = 3; break;       // <1,16>  12 (32 spilled registers)
    case 22: a = 4; b = 2; break;       // <1,16>  8 (no spills)
    case 30: a = 1; b = 4; break;       // <4,8>   16
    case 31: a = 1; b = 3; break;       // <8,4>   24
    case 40: a = 1; b = 2; break;       // <8,8>   16
    // (r04, measured and not kept: <2,16> for the 4096 window — half the exchange, an even split of ~10 live slots — at 2 waves per SIMD without spills
    //  or 3 with: affine stage of a 2 500-region slice of the 1-10 kb shard 364 ms with <4,8>, 436 / 392 ms with <2,16>: occupancy weighs more)

    default: break;
  }
  *aln_per_block = a; *blocks_per_cu = b;
}

// One tier launch: `shape` selects among the instantiations of a window (0 = the default of the chain).
int otg_launch_affine_reg_tier(otg_ctx* ctx, int tier, int shape, uint32_t blocks, const uint8_t* d_arena, const otg_align_task* d_tasks,
                               const uint32_t* d_sorted, const uint32_t* d_seg, int g, int32_t* d_scores, const uint64_t* d_cig_off,
                               uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells, uint32_t* ticket, uint32_t* n_overflow,
                               uint32_t* overflow_list, const AffWs& ws, const int32_t* d_bound, unsigned long long* visited)
{
#define OTG_REG_LAUNCH(NWV, S2V, SEQV, WPEUV, ...)                                                                                              \
  hipLaunchKernelGGL((wfa_affine_reg_kernel<NWV, S2V, SEQV, WPEUV, ##__VA_ARGS__>), dim3(blocks), dim3(NWV == 1 ? 256 : NWV * 64), 0, ctx->stream, d_arena, \
                     d_tasks, d_sorted, d_seg, g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, ticket, n_overflow, overflow_list, ws, d_bound, visited)
  switch (tier * 10 + shape) {
    case 0: OTG_REG_LAUNCH(1, 8, 4096, 4); break;
    case 30: OTG_REG_LAUNCH(4, 8, 8192, 4); break;
#ifndef OTG_REG_PROBE       // (a compile-time probe of the two main bodies: -DOTG_REG_PROBE)
    case 1: OTG_REG_LAUNCH(2, 4, 4096, 5, 256); break;
    case 10: OTG_REG_LAUNCH(1, 12, 4608, 3); break;
    case 11: OTG_REG_LAUNCH(1, 12, 4608, 4, 512, 0); break;
    case 12: OTG_REG_LAUNCH(2, 6, 4608, 4); break;
    case 20: OTG_REG_LAUNCH(2, 8, 6144, 4); break;
    case 21: OTG_REG_LAUNCH(1, 16, 6144, 3, 512, 1); break;
    case 22: OTG_REG_LAUNCH(1, 16, 6144, 2); break;
    case 31: OTG_REG_LAUNCH(8, 4, 8192, 6, 256); break;
    case 40: OTG_REG_LAUNCH(8, 8, 12288, 4); break;

#endif
    default: return otg_fail(ctx, OTG_ERR_ARG, "no register tier %d shape %d", tier, shape);
  }
#undef OTG_REG_LAUNCH
  HIP_TRY(ctx, hipGetLastError());
  return OTG_OK;
}
