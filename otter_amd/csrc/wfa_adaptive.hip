// wfa_adaptive.hip — the wavefront aligners under WFA2-lib's adaptive wavefront reduction, wf_heuristic_wfadaptive(min_wavefront_length,
// max_distance_threshold, steps_between_cutoffs) (gfx950).
//
// Why it exists: the reference constructs WFAlignerEdit(Score, MemoryMed) and WFAlignerGapAffine(4,6,2, Alignment, MemoryMed)
// (src/assemble.cpp:49-50) and never calls setHeuristic*, so every distance (src/analignments.cpp:70-71,88-97) and every op string
// (src/analignments.cpp:25,31,37,268-280) inherits WFA2-lib's DEFAULT heuristic — which that un-vendored library build may set to
// wfadaptive(10, 50, 1).  Exact mode stays the default of this library; this file is the
// documented switch: otg_set_heuristic / otg_params.heuristic (the adapter's setHeuristicWFadaptive / setHeuristicNone).
//
// Semantics (WFA2-lib wavefront_heuristic.c, wavefront_heuristic_cufoff -> wfadaptive; the test-side CPU restatement holds the same rule): after the M wavefront of a score has been extended and the
// end test has failed, every `steps` scores, if the wavefront spans at least `min_wf_len` diagonals: left(k) = what is left to align
// from the offset of diagonal k (end-to-end max(plen - v, tlen - h); ends-free the smaller of the two free-end variants); diagonals
// whose left(k) exceeds the smallest by more than `max_dist` are dropped from both ends of the wavefront, never past the end
// diagonal(s); the I / D wavefronts of that score are cut to the same range.  The cut reads the offset of EVERY live diagonal, so none
// of the exact chain's devices applies here (score bound, diamond, bit-parallel tiers, reversed sequences): these kernels run the
// plain wavefront recurrence, score by score, on the diagonals the cut leaves.
//
// What the cut buys: a wavefront no longer grows by two diagonals per score.  On the tandem-repeat workloads it stays 50-500 diagonals
// wide (shifts by a repeat unit extend as far as the main diagonal, so they survive the cut; scripts/heuristic_widths.py), against
// thousands in exact mode.  Mapping: ONE wave64 per alignment, lanes = diagonals, chunks of 64 diagonals swept in ascending order;
// the wavefronts live in LDS in a MODULAR window (slot = k mod CAP: the live range drifts across the diagonals as the alignment
// proceeds, a fixed window would have to span the whole drift), 16-bit offsets; tiers by window size, the last one keeps int32 rows in
// HBM and takes any length and any penalties.  The smallest `left` is gathered while the diagonals finish their extension (per-lane
// minimum + one wave reduction per score); the cut itself is one ballot over the 32 diagonals at either end.
//
// The tiers as they stand (r04):
//   edit        wfa_edit_adaptive_lds_kernel<1024> (packed pair in LDS, lane masks, two probes per sweep; a wavefront wider than the window — the
//               start of an ends-free pair, a wide stretch on the way — runs on a global row and moves back in) -> <4096> -> the packed 16 384
//               window with eight waves per pair (wfa_edit_adaptive_mw_kernel) -> byte probes 2 048 / 16 384 -> int32 row in HBM;
//   gap-affine  wfa_affine_adaptive_lds_kernel<256> (eleven i16 rows per wave, null discipline per row) -> the 1 024 window with four waves per
//               alignment and the 4 096 one with eight (wfa_affine_adaptive_mw_kernel: chunks dealt to the waves, one LDS barrier per score) ->
//               byte probes 1 024 -> int32 rings in HBM.
#include "wfa_affine_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>

using namespace otg_affine;

namespace {

constexpr int BIG = 1 << 30;

struct Heur { int min_wf_len, max_dist, steps; };

__device__ __forceinline__ int wave_min_i32(int v) { return -otg_wave_max_i32(-v); }      // |v| <= 2^30 here
__device__ __forceinline__ int dpp_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }   // lane i <- lane i-1

// lane masks of comparisons (v_cmp into an SGPR pair; the logic on them runs on the scalar unit) and a select on a lane mask.  Written with bools,
// hipcc 7.2 turns "does any lane ..." and "or into the flag" into a select, an and and a compare on the vector unit each (see wfa_affine_reg.hip)
__device__ __forceinline__ unsigned long long mk_eq(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 32); }
__device__ __forceinline__ unsigned long long mk_sle(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 41); }
__device__ __forceinline__ unsigned long long mk_ule(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 37); }
__device__ __forceinline__ int sel(unsigned long long m, int a, int b) { int r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m)); return r; }     // lane in m ? a : b

// what is left to align from offset h of diagonal k (the distance the cut compares)
__device__ __forceinline__ int left_to_align(int h, int k, int pl, int tl, bool ef, int pef, int tef)
{
  if (h < 0) return BIG;
  const int lv = pl - (h - k), lh = tl - h;
  if (!ef) return imax(lv, lh);
  return imin(imax(lh, lv - pef), imax(lv, lh - tef));
}

// The cut.  [lo, hi] = range of the extended M wavefront, mind = smallest left_to_align over it, off(k) = offset of diagonal k (called
// for lo <= k <= hi only).  Wave-uniform control flow; returns the trimmed range in lo / hi.
template <class Off>
__device__ __forceinline__ void wfadaptive_cut(const Heur& H, int& steps_wait, int mind, int pl, int tl, bool ef, int pef, int tef,
                                               int& lo, int& hi, int lane, Off off)
{
  --steps_wait;
  if (steps_wait > 0) return;
  if (hi - lo + 1 < H.min_wf_len) return;
  const int kend = tl - pl;
  const int min_k = ef ? kend - tef : kend, max_k = ef ? kend + pef : kend;
  const int top_limit = imin(min_k - 1, hi);
  int nlo = lo;
  if (top_limit > lo) {
    nlo = top_limit;
    for (int c = lo; c < top_limit; c += 64) {
      const int k = c + lane;
      bool ok = false;
      if (k < top_limit) ok = left_to_align(off(k), k, pl, tl, ef, pef, tef) - mind <= H.max_dist;
      const unsigned long long b = __ballot(ok);
      if (b) { nlo = c + (int)__builtin_ctzll(b); break; }
    }
  }
  const int bottom_limit = imax(max_k + 1, nlo);
  int nhi = hi;
  if (hi > bottom_limit) {
    nhi = bottom_limit;
    for (int c = hi; c > bottom_limit; c -= 64) {
      const int k = c - 63 + lane;                      // this chunk covers [c - 63, c]
      bool ok = false;
      if (k > bottom_limit) ok = left_to_align(off(k), k, pl, tl, ef, pef, tef) - mind <= H.max_dist;
      const unsigned long long b = __ballot(ok);
      if (b) { nhi = c - (int)__builtin_clzll(b); break; }
    }
  }
  lo = nlo; hi = nhi;
  steps_wait = H.steps;
}

// The same cut with both ends looked at in ONE step (the fast tiers): lanes 0-31 hold the 32 lowest diagonals of the range, lanes 32-63 the 32
// highest; one offset read, one distance, one ballot.  A scan that would have to look further than 32 diagonals from an end — nothing in range
// among them and the limit not reached — falls back to the general form above (same result by construction: both implement "first diagonal
// from the end within the threshold, but not past the limit").
template <class Off>
__device__ __forceinline__ void wfadaptive_cut32(const Heur& H, int& steps_wait, int mind, int pl, int tl, bool ef, int pef, int tef,
                                                 int& lo, int& hi, int lane, Off off)
{
  if (steps_wait - 1 > 0 || hi - lo + 1 < H.min_wf_len) { wfadaptive_cut(H, steps_wait, mind, pl, tl, ef, pef, tef, lo, hi, lane, off); return; }
  const int kend = tl - pl;
  const int min_k = ef ? kend - tef : kend, max_k = ef ? kend + pef : kend;
  const int top_limit = imin(min_k - 1, hi);
  const bool low = lane < 32;
  const int k = low ? lo + lane : hi - 63 + lane;
  bool ok = false;
  if (k >= lo && k <= hi) ok = left_to_align(off(k), k, pl, tl, ef, pef, tef) - mind <= H.max_dist;
  const unsigned long long b = __ballot(ok);
  // low end: the first diagonal in [lo, top_limit) within the threshold, else top_limit
  int nlo = lo;
  bool fallback = false;
  if (top_limit > lo) {
    const int n = top_limit - lo;                                  // candidates lo .. top_limit - 1
    const uint32_t cand = (uint32_t)b & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
    if (cand) nlo = lo + (int)__builtin_ctz(cand);
    else if (n <= 32) nlo = top_limit;
    else fallback = true;
  }
  int nhi = hi;
  if (!fallback) {
    const int bottom_limit = imax(max_k + 1, nlo);
    if (hi > bottom_limit) {
      const int n = hi - bottom_limit;                             // candidates bottom_limit + 1 .. hi = the top n bits
      const uint32_t hb = (uint32_t)(b >> 32);
      const uint32_t cand = hb & (n >= 32 ? 0xffffffffu : ~((1u << (32 - n)) - 1u));
      if (cand) nhi = hi - (int)__builtin_clz(cand);
      else if (n <= 32) nhi = bottom_limit;
      else fallback = true;
    }
  }
  if (fallback) { wfadaptive_cut(H, steps_wait, mind, pl, tl, ef, pef, tef, lo, hi, lane, off); return; }
  lo = nlo; hi = nhi;
  steps_wait = H.steps;
}

// ---------------------------------------------------------------------------------------------------
// Edit distance, score only.  Replaces WFAlignerEdit(Score, MemoryMed)::alignEnd2End / alignEndsFree + getAlignmentScore() under the
// heuristic.  The wavefront is updated IN PLACE (ascending sweep: the left neighbour of lane 0 is carried in a scalar, the right
// neighbour of lane 63 is still the old value); the extend step probes 8 bytes in the sweep and leaves longer runs to a
// ballot-compacted LDS queue drained 16 -> 64 -> wave-cooperative 512 bytes per pass.
template <int CAP>
struct EditLds {
  volatile lds_u16* wf;
  __device__ __forceinline__ bool fits(int pl, int tl) const { return pl < 65535 && tl < 65535; }
  __device__ __forceinline__ int cap() const { return CAP; }
  __device__ __forceinline__ int rd(int k) const { const int x = wf[k & (CAP - 1)]; return x == 0xFFFF ? OTG_NULL_OFF : x; }
  __device__ __forceinline__ void wr(int k, int h) const { wf[k & (CAP - 1)] = (uint16_t)(h < 0 ? 0xFFFF : h); }
  __device__ __forceinline__ void sync() const {}
};
struct EditGlobal {
  volatile int32_t* wf; int kb; int gcap;
  __device__ __forceinline__ bool fits(int pl, int tl) const { return pl + tl + 3 <= gcap; }
  __device__ __forceinline__ int cap() const { return gcap; }
  __device__ __forceinline__ int rd(int k) const { return wf[k + kb]; }
  __device__ __forceinline__ void wr(int k, int h) const { wf[k + kb] = h < 0 ? OTG_NULL_OFF : h; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
};

template <int CAP, int QCAP, int WPB>
__global__ __launch_bounds__(WPB * 64) void wfa_edit_adaptive_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    Heur H, int32_t* gws, int gcap)
{
  constexpr bool GLOBAL_WF = CAP == 0;
  constexpr int LCAP = GLOBAL_WF ? 2 : CAP;
  __shared__ uint16_t s_wf[WPB][LCAP];
  __shared__ uint32_t s_q[WPB][QCAP];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  volatile lds_u32* queue = (volatile lds_u32*)&s_q[wib][0];
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;

  for (;;) {
    const uint32_t tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int res_s = 0;
    uint64_t res_W = 0;
    auto body = [&](auto st) -> int {
      int lo = ef ? -t.pattern_begin_free : 0, hi = ef ? t.text_begin_free : 0;
      if (lo < -pl) lo = -pl;
      if (hi > tl) hi = tl;
      int plo = lo, phi = hi;                       // range of the previous score's wavefront (after its cut)
      int s = 0, steps_wait = 0;
      uint64_t W = 0;
      bool done = false, overflow = !st.fits(pl, tl);
      while (!overflow) {
        if (hi - lo + 1 > st.cap()) { overflow = true; break; }
        W += (uint64_t)(hi - lo + 1);
        int carry = OTG_NULL_OFF;
        int dmin = BIG;                             // per lane: smallest left_to_align among the diagonals this lane finished
        bool fin_l = false;
        int qn = 0;
        auto finished = [&](int h, int k) {         // diagonal k is fully extended at offset h
          dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
          if (ef) { const int v = h - k; fin_l = fin_l || (h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef); }
        };
        auto drain = [&]() {
          st.sync();
          int pass = 0;
          while (qn > 0) {
            if (qn <= 4 && pass > 0) {
              for (int q = 0; q < qn; ++q) {
                const int k = lo + __builtin_amdgcn_readfirstlane((int)queue[q]);
                int h = __builtin_amdgcn_readfirstlane(st.rd(k));
                const int v = h - k;
                const int m = otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
                h += m;
                st.wr(k, h);                      // the same value from every lane
                finished(h, k);
              }
              qn = 0;
              st.sync();
              break;
            }
            int wq = 0;
            for (int q0 = 0; q0 < qn; q0 += 64) {
              const bool act = q0 + lane < qn;
              int k = 0, h = 0, v = 0;
              bool more = false;
              if (act) {
                k = lo + (int)queue[q0 + lane];
                h = st.rd(k);
                v = h - k;
                const int rem = imin(pl - v, tl - h);
                int m, full;
                if (pass == 0) {
                  const uint64_t xl = otg_load8(P + v) ^ otg_load8(T + h), xh = otg_load8(P + v + 8) ^ otg_load8(T + h + 8);
                  m = xl ? (__builtin_ctzll(xl) >> 3) : (xh ? 8 + (__builtin_ctzll(xh) >> 3) : 16);
                  m = imin(m, rem); full = 16;
                } else { m = otg_match64(P, T, v, h, rem); full = 64; }
                v += m; h += m;
                more = (m == full) && v < pl && h < tl;
                st.wr(k, h);
                if (!more) finished(h, k);
              }
              const unsigned long long mm = __ballot(more);
              if (more) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                queue[wq + rank] = (uint32_t)(k - lo);
              }
              wq += __builtin_popcountll(mm);
            }
            qn = wq; ++pass;
            st.sync();
          }
        };
        // ---- sweep
        for (int c = lo; c <= hi; c += 64) {
          const int k = c + lane;
          const bool in = k <= hi;
          int mx;
          if (s == 0) {
            mx = k > 0 ? k : 0;
          } else {
            int o = OTG_NULL_OFF, r = OTG_NULL_OFF;
            if (k >= plo && k <= phi) o = st.rd(k);
            if (k + 1 >= plo && k + 1 <= phi) r = st.rd(k + 1);
            int l = dpp_shr1(o);
            if (lane == 0) l = carry;
            carry = __builtin_amdgcn_readlane(o, 63);
            const int a = l + 1, b = o + 1;
            mx = a > b ? a : b;
            mx = r > mx ? r : mx;
          }
          int h = mx, v = mx - k;
          const bool valid = in && mx >= 0 && h <= tl && v <= pl;
          bool more = false;
          if (valid && v < pl && h < tl) {
            const uint64_t x = otg_load8(P + v) ^ otg_load8(T + h);
            int m = x ? (__builtin_ctzll(x) >> 3) : 8;
            m = imin(m, imin(pl - v, tl - h));
            v += m; h += m;
            more = (m == 8) && v < pl && h < tl;
          }
          if (in) st.wr(k, valid ? h : OTG_NULL_OFF);
          if (valid && !more) finished(h, k);
          const unsigned long long mm = __ballot(more);
          if (more) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            queue[qn + rank] = (uint32_t)(k - lo);
          }
          qn += __builtin_popcountll(mm);
          if (qn + 64 > QCAP) drain();
        }
        drain();
        // ---- end test on the fully extended wavefront
        bool any_done;
        if (ef) any_done = __ballot(fin_l) != 0ull;
        else {
          any_done = false;
          if (kend >= lo && kend <= hi) { const int x = __builtin_amdgcn_readfirstlane(st.rd(kend)); any_done = x >= tl; }
        }
        if (any_done) { done = true; break; }
        // ---- the cut
        const int mind = wave_min_i32(dmin);
        wfadaptive_cut(H, steps_wait, mind, pl, tl, ef, pef, tef, lo, hi, lane, [&](int k) { return st.rd(k); });
        plo = lo; phi = hi;
        lo = lo - 1 < -pl ? -pl : lo - 1;
        hi = hi + 1 > tl ? tl : hi + 1;
        ++s;
        if (s > pl + tl + 2) break;             // cannot happen: the end diagonal is never cut and gains at least one base per score
      }
      res_s = s; res_W = W;
      return done ? 0 : (overflow ? 1 : 2);
    };
    int status;
    if constexpr (GLOBAL_WF) status = body(EditGlobal{(volatile int32_t*)(gws + (size_t)(blockIdx.x * WPB + wib) * (size_t)gcap), pl + 1, gcap});
    else status = body(EditLds<CAP>{(volatile lds_u16*)&s_wf[wib][0]});
    // wave-uniform tail: every lane stores the same value to the same address
    if (status == 0) {
      scores[ti] = res_s;
      if (cells) cells[ti] = res_W;
    } else if (status == 1 && overflow_list) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;
    } else {
      scores[ti] = -1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The packed pair of PackedPair below written by a whole block (the multi-wave tiers): thread `tid` of `nt` packs words tid, tid + nt, ...; returns
// whether THIS thread met a byte outside ACGT.
__device__ __forceinline__ bool pack_pair_block(volatile lds_u32* SQ, const uint8_t* P, int pl, const uint8_t* T, int tl, int offT, int tid, int nt)
{
  bool bad = false;
  auto pack = [&](const uint8_t* S, int len, int woff) {
    for (int q = tid; q < (len + 15) / 16 + 3; q += nt) {
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int b0 = 16 * q + 8 * j;
        const uint64_t x = b0 < len ? otg_load8(S + b0) : 0ull;        // (the arena has 64 bytes of slack behind its last sequence)
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
  return __ballot(bad) != 0ull;
}

// Both sequences of a pair packed to 2 bits per base in LDS (word q = bases 16 q .. 16 q + 15, code (byte >> 1) & 3; pattern at word 0,
// text at word offT): a probe compares 32 bases with six LDS reads instead of two HBM round trips — under the cut a wave advances one
// score at a time and every score ends in a probe, so the probe latency IS the kernel's speed.  A pair with a byte outside ACGT, or too
// long for the wave's share of LDS, is not packed (`ok` false) and runs on the byte probes of the generic kernels.
struct PackedPair {
  volatile lds_u32* SQ; int offT; bool ok;
  // seqw = words of LDS this wave owns for the pair (dynamic shared memory, sized by the launcher from the batch's longest read)
  __device__ __forceinline__ void init(const uint8_t* P, int pl, const uint8_t* T, int tl, int lane, int seqw)
  {
    offT = (pl + 15) / 16 + 3;
    ok = offT + (tl + 15) / 16 + 3 <= seqw;
    if (!ok) return;
    bool bad = false;
    auto pack = [&](const uint8_t* S, int len, int woff) {
      for (int q = lane; q < (len + 15) / 16 + 3; q += 64) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int b0 = 16 * q + 8 * j;
          const uint64_t x = b0 < len ? otg_load8(S + b0) : 0ull;        // (the arena has 64 bytes of slack behind its last sequence)
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
    ok = __ballot(bad) == 0ull;
  }
  __device__ __forceinline__ uint64_t ld32(int woff, int pos) const
  {
    const int w = woff + (pos >> 4);
    const uint32_t sh = (uint32_t)(pos & 15) * 2u;
    const uint32_t d0 = SQ[w], d1 = SQ[w + 1], d2 = SQ[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
  }
  // equal bases from pattern position v / text position h on, at most 32 and at most rem (0 <= v, h; rem >= 0)
  __device__ __forceinline__ int match32(int v, int h, int rem) const
  {
    const uint64_t xx = ld32(0, v) ^ ld32(offT, h);
    const int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
    return imin(m, rem);
  }
};

// The fast edit tier: window of CAP diagonals (signed 16-bit offsets, null = -32768) + the packed pair in LDS.  Invariant that removes every
// range test from the sweep: a slot of the window is non-null only while its diagonal is inside the live range — the window is null-filled per
// pair, and the diagonals a cut drops are nulled right there — so the recurrence reads its neighbours unconditionally.
//
// The sweep is written BRANCH-FREE (r04, second version): on gfx950 a wave64 vector instruction costs 2.2 or 4.2 SIMD cycles by class and the one
// scalar unit of a CU serves four SIMDs (profiles/r04_valu_peak.json), and the first version spent as many scalar as vector instructions on
// execution-mask bookkeeping around its `if`s.  Now every lane of a chunk computes, probes and stores unconditionally — an invalid or out-of-range
// lane probes wherever its offset points (LDS reads outside the allocation return zero, inside it harmless words) and stores NULL, which is what a
// slot outside the live range holds anyway — and the only branches left are wave-uniform: "did any lane's run outlive its probe" and the loop.
// Score 0 runs through the same sweep: the start diagonals are seeded with their offset MINUS ONE, as if by a score -1.
// Where the fast edit tier keeps its wavefront: the modular LDS window, or — while a pair's wavefront is wider than the window — a row in HBM / L2.
// On long reads three ends-free pairs in ten START wider than 1 024 diagonals (the free begin seeds thousands of diagonals) and are cut below that
// within one or two scores (scripts/heuristic_widths.py: 1.5 wide scores of 394 on the 1-10 kb shard); sent to the 4 096-diagonal tier for that, they
// ran their whole alignment at its 2.5 waves per SIMD.  Now the same sweep runs those first scores on the global row and moves into the window as
// soon as the range fits.
template <int CAP>
struct EdWfLds {
  volatile lds_i16* wf;
  __device__ __forceinline__ int ld(int k) const { return wf[k & (CAP - 1)]; }
  __device__ __forceinline__ void st(int k, int v) const { wf[k & (CAP - 1)] = (int16_t)v; }
  __device__ __forceinline__ void sync() const {}
};
struct EdWfGlobal {
  volatile int16_t* g;      // g[k] for -pl - 2 <= k <= tl + 66 (the pointer is already biased)
  __device__ __forceinline__ int ld(int k) const { return g[k]; }
  __device__ __forceinline__ void st(int k, int v) const { g[k] = (int16_t)v; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
};

template <int CAP, int QCAP, int WPB>
__global__ __launch_bounds__(WPB * 64) void wfa_edit_adaptive_lds_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list, Heur H, int seqw,
    int16_t* __restrict__ gscratch, int gcap)        // gscratch: gcap offsets per wave for the wide start of a pair (null: such pairs go to the next tier)
{
  __shared__ __attribute__((aligned(16))) int16_t s_wf[WPB][CAP];
  __shared__ uint16_t s_q[WPB][QCAP];
  extern __shared__ uint32_t s_dyn[];                         // [WPB][seqw]: the packed pairs
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  volatile lds_i16* wf = (volatile lds_i16*)&s_wf[wib][0];
  volatile lds_u32* wf32 = (volatile lds_u32*)&s_wf[wib][0];
  volatile lds_u16* queue = (volatile lds_u16*)&s_q[wib][0];
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  constexpr int MASK = CAP - 1;
  constexpr int NUL = -32768;

  for (;;) {
    const uint32_t tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int lo = ef ? -t.pattern_begin_free : 0, hi = ef ? t.text_begin_free : 0;
    if (lo < -pl) lo = -pl;
    if (hi > tl) hi = tl;
    // (the last chunk of a sweep stores 64 lanes whatever the range: the slots behind hi must not alias the live range)
    bool wide = hi - lo + 68 > CAP;             // the pair starts wider than the window: its first scores run on the global row
    bool overflow = pl > 32766 || tl > 32766 || (wide && (!gscratch || pl + tl + 72 > gcap));
    PackedPair pk{(volatile lds_u32*)(s_dyn + (size_t)wib * seqw), 0, false};
    if (!overflow) { pk.init(P, pl, T, tl, lane, seqw); overflow = !pk.ok; }
    int s = 0, steps_wait = 0;
    uint64_t W = 0;
    bool done = false;
    volatile int16_t* grow = gscratch ? gscratch + (size_t)(blockIdx.x * WPB + wib) * (size_t)gcap + (pl + 2) : nullptr;      // grow[k], k >= -pl - 2
    int gnlo = 0, gnhi = -1;                    // the span of the global row that has been written (nulls or offsets) for this pair
    if (!overflow) {
      for (int q = lane; q < CAP / 2; q += 64) wf32[q] = 0x80008000u;
      if (!wide) { for (int c = lo; c <= hi; c += 64) { const int k = c + lane; if (k <= hi) wf[k & MASK] = (int16_t)((k > 0 ? k : 0) - 1); } }      // "score -1"
      else {
        gnlo = lo - 2; gnhi = hi + 66;
        for (int c = gnlo; c <= gnhi; c += 64) { const int k = c + lane; if (k <= gnhi) grow[k] = (int16_t)((k >= lo && k <= hi) ? (k > 0 ? k : 0) - 1 : NUL); }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
    }
    const int offT4 = pk.offT * 4;
    // one score on the wavefront kept in `wfs`: 0 = go on, 1 = done, 2 = the range outgrew `cap_here`, 3 = no alignment within pl + tl + 2
    auto one_score = [&](auto wfs, int cap_here) -> int {
      if (hi - lo + 68 > cap_here) return 2;
      W += (uint64_t)(hi - lo + 1);
      int carry = NUL;
      int dmin = BIG;
      // "some cell ends the alignment" is kept as a LANE MASK in scalar registers (the conditions' compare results or-ed on the scalar unit): as a
      // bool per lane the compiler rebuilt it with two selects, an and and a compare per chunk
      unsigned long long finm = 0ull;
      int qn = 0;
      auto finished = [&](int h, int k) -> bool {          // (the caller collects the lanes' answers: it may be called under a lane condition)
        dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
        if (ef) { const int v = h - k; return (h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef); }
        return k == kend && h >= tl;
      };
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          wfs.sync();              // (global row: a queued cell is extended by another lane than the one that stored it)
          if (qn <= 4 && pass > 0) {
            for (int q = 0; q < qn; ++q) {
              const int k = lo + __builtin_amdgcn_readfirstlane((int)queue[q]);
              int h = __builtin_amdgcn_readfirstlane(wfs.ld(k));
              const int v = h - k;
              h += otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
              wfs.st(k, h);                            // the same value from every lane
              if (finished(h, k)) finm = ~0ull;
            }
            qn = 0;
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int k = 0, h = 0, v = 0;
            bool more = false, fin = false;
            if (act) {
              k = lo + (int)queue[q0 + lane];
              h = wfs.ld(k);
              v = h - k;
              const int m = pk.match32(v, h, imin(pl - v, tl - h));
              v += m; h += m;
              more = (m == 32) && v < pl && h < tl;
              wfs.st(k, h);
              if (!more) fin = finished(h, k);
            }
            finm |= __ballot(fin);
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint16_t)(k - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      // ---- sweep (branch-free per lane)
      const volatile lds_u32* SQ = pk.SQ;
      for (int c = lo; c <= hi; c += 64) {
        const int k = c + lane;
        const int o = wfs.ld(k), r = wfs.ld(k + 1);            // null outside the live range: no range tests
        const int l = __builtin_amdgcn_update_dpp(carry, o, 0x138, 0xf, 0xf, false);      // lane i <- lane i-1, lane 0 <- the previous chunk's lane 63
        carry = __builtin_amdgcn_readlane(o, 63);
        const int mx = imax(imax(l + 1, o + 1), r);
        const int v = mx - k;
        const int t1 = tl - mx, t2 = pl - v;                              // what is left of the text / the pattern
        const unsigned long long vm = mk_ule((uint32_t)mx, (uint32_t)tl) & mk_ule((uint32_t)v, (uint32_t)pl) & mk_sle(k, hi);      // valid cells
        // the probe, wherever the offsets point: equal leading bases of pattern[pv ..] and text[ph ..], at most 32
        auto probe = [&](int pv, int ph) -> int {
          const int wp = (pv >> 2) & ~3, wt = offT4 + ((ph >> 2) & ~3);    // byte addresses of the first word of each 32-base window
          const uint32_t sp = (uint32_t)(pv & 15) * 2u, st = (uint32_t)(ph & 15) * 2u;
          const volatile lds_u32* pp = (const volatile lds_u32*)((const volatile __attribute__((address_space(3))) char*)SQ + wp);
          const volatile lds_u32* pt = (const volatile lds_u32*)((const volatile __attribute__((address_space(3))) char*)SQ + wt);
          const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], q0 = pt[0], q1 = pt[1], q2 = pt[2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, sp) ^ __builtin_amdgcn_alignbit(q1, q0, st);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, sp) ^ __builtin_amdgcn_alignbit(q2, q1, st);
          uint32_t flo, fhi;      // v_ffbl_b32: index of the lowest set bit, 0xffffffff for zero — which is what the min below wants
          asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"(xl));
          asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"(xh));
          const uint32_t a = flo < (fhi | 32u) ? flo : (fhi | 32u);
          return (int)((a < 64u ? a : 64u) >> 1);
        };
        const int pm = probe(v, mx);
        int m = imin(imin(pm, t1), t2);                                  // the run, limited by the sequence ends
        unsigned long long mm = vm & mk_eq(imin(imin(pm, t1 - 1), t2 - 1), 32);        // more: a full probe with more than 32 bases left of both sequences
        // A run that outlives its probe (one cell in 120 at ONT divergence, i.e. four chunks in ten) gets a second probe right here, by the whole
        // wave: the queue, its drain and their loops — a fixed cost of ~40 vector and as many scalar instructions per score — are then left to runs
        // beyond 64 bases (one score in a hundred)
        if (mm) {
          const int pm2 = probe(v + 32, mx + 32);
          const int m2 = imin(imin(pm2, t1 - 32), t2 - 32);
          m = sel(mm, m2 + 32, m);
          mm &= mk_eq(imin(imin(pm2, t1 - 33), t2 - 33), 32);
        }
        const int h2 = mx + m;
        wfs.st(k, sel(vm, h2, NUL));                        // every lane stores (see the head of the kernel)
        const unsigned long long hm = vm & ~mm;                          // cells that are final here
        const int lh = t1 - m, lv = t2 - m;
        int d;
        if (!ef) { d = imax(lh, lv); finm |= hm & mk_eq(k, kend) & mk_sle(lh, 0); }      // the end diagonal has reached the end of the text
        else {
          d = imin(imax(lh, lv - pef), imax(lv, lh - tef));
          finm |= hm & ((mk_sle(lh, 0) & mk_sle(lv, pef)) | (mk_sle(lv, 0) & mk_sle(lh, tef)));
        }
        dmin = imin(dmin, sel(hm, d, BIG));
        if (mm) {
          if (sel(mm, 1, 0)) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            queue[qn + rank] = (uint16_t)(k - lo);
          }
          qn += __builtin_popcountll(mm);
          if (qn + 64 > QCAP) drain();
        }
      }
      if (qn) drain();
      wfs.sync();
      // ---- end test on the fully extended wavefront
      if (finm != 0ull) return 1;
      // ---- the cut; what it drops is nulled (the invariant above)
      const int olo = lo, ohi = hi;
      const int mind = wave_min_i32(dmin);
      wfadaptive_cut32(H, steps_wait, mind, pl, tl, ef, pef, tef, lo, hi, lane, [&](int k) { const int x = wfs.ld(k); return x < 0 ? OTG_NULL_OFF : x; });
      for (int c = olo; c < lo; c += 64) if (c + lane < lo) wfs.st(c + lane, NUL);
      for (int c = hi + 1; c <= ohi; c += 64) if (c + lane <= ohi) wfs.st(c + lane, NUL);
      wfs.sync();
      lo = lo - 1 < -pl ? -pl : lo - 1;
      hi = hi + 1 > tl ? tl : hi + 1;
      ++s;
      return s > pl + tl + 2 ? 3 : 0;
    };
    // two loops, so that the window's loop is as tight as when it was the only one: the scores on the global row (a pair that starts wide, or whose
    // wavefront outgrew the window on the way: 2 % of the pairs of the 1-10 kb shard, for ~80 of their ~2 400 scores — restarted in the 4 096 tier they
    // cost an eighth of that shard's step), then the scores in the window; a range that outgrows the window is spilled to the row, one that has
    // shrunk well below the window (128 diagonals of hysteresis) moves back in.
    const bool can_spill = gscratch != nullptr && pl + tl + 72 <= gcap;
    int rc = overflow ? 4 : 0;              // 4: not a pair for this tier (too long, does not pack, starts wide without a row to start on)
    for (;;) {
      while (rc == 0 && wide) {
        rc = one_score(EdWfGlobal{grow}, 32767 + 68);
        if (rc != 0) break;
        if (hi - lo + 68 + 128 <= CAP) {         // move in: the window is all null
          for (int c = lo; c <= hi; c += 64) { const int k = c + lane; if (k <= hi) wf[k & MASK] = grow[k]; }
          wide = false;
        } else {                                // the slots a score may read or store beyond what has been written so far are nulled first
          if (lo - 2 < gnlo) { for (int c = lo - 2; c < gnlo; c += 64) if (c + lane < gnlo) grow[c + lane] = (int16_t)NUL; gnlo = lo - 2; }
          if (hi + 66 > gnhi) { for (int c = gnhi + 1; c <= hi + 66; c += 64) if (c + lane <= hi + 66) grow[c + lane] = (int16_t)NUL; gnhi = hi + 66; }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
      }
      while (rc == 0) rc = one_score(EdWfLds<CAP>{wf}, CAP);
      if (rc != 2 || wide || !can_spill) break;
      // spill: the window holds the wavefront (nulls outside its range; hi - lo + 1 < CAP, so no two diagonals share a slot); the row gets it with
      // nulls around it, the window goes back to all null
      gnlo = lo - 2; gnhi = hi + 66;
      for (int c = gnlo; c <= gnhi; c += 64) { const int k = c + lane; if (k <= gnhi) grow[k] = (int16_t)((k >= lo && k <= hi) ? wf[k & MASK] : NUL); }
      for (int q = lane; q < CAP / 2; q += 64) wf32[q] = 0x80008000u;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      wide = true; rc = 0;
    }
    if (rc == 1) done = true;
    if (rc == 2 || rc == 4) overflow = true;
    // wave-uniform tail: every lane stores the same value to the same address
    if (done) {
      scores[ti] = s;
      if (cells) cells[ti] = W;
    } else if (overflow && overflow_list) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;
    } else {
      scores[ti] = -1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The widest packed edit tier: NW waves on ONE pair (16 384 diagonals: the ends-free pairs of long reads whose wavefront outgrows 4 096 — they went
// through the byte-probe tiers before).  Same plan as the wide gap-affine tiers further down: chunks of a score dealt to the
// waves, ranges / cut / step counter replicated in every wave's scalar registers, ONE LDS-only barrier per score.  The in-place sweep of the
// one-wave tier (left neighbour carried, right neighbour not yet overwritten) does not survive waves working side by side, so the wavefront
// alternates between TWO rows — score s reads row s & 1 and writes the other — with the null discipline per row: what the write row still holds
// of the wavefront two scores back is nulled outside the new range when the row is taken over (wave 0), what the cut drops right after the
// cut (every wave; a late wave's cut reads nulls exactly where it would have cut).
template <int CAP, int QCAP, int NW>
__global__ __launch_bounds__(NW * 64) void wfa_edit_adaptive_mw_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list, Heur H, int seqw)
{
  constexpr int MASK = CAP - 1, NUL = -32768, PAD = 4, NT = NW * 64;
  __shared__ __attribute__((aligned(16))) int16_t s_wf[2][CAP];
  __shared__ uint16_t s_q[NW][QCAP];
  __shared__ int s_x[2][NW][2];
  __shared__ uint32_t s_tk;
  __shared__ int s_bad;
  extern __shared__ uint32_t s_dyn[];                         // [seqw]: the packed pair of the block's alignment
  using lds_char = __attribute__((address_space(3))) char;
  const int tid = threadIdx.x, lane = tid & 63;
  auto U = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  const int ww = U(tid >> 6);
  volatile lds_char* ROWS = (volatile lds_char*)&s_wf[0][0];
  volatile lds_u32* ROWS32 = (volatile lds_u32*)&s_wf[0][0];
  volatile lds_u16* queue = (volatile lds_u16*)&s_q[ww][0];
  volatile lds_u32* SQ = (volatile lds_u32*)s_dyn;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  auto rd = [&](int row, int k) -> int { return *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)); };
  auto wr = [&](int row, int k, int v) { *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)) = (int16_t)v; };
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto null_wave = [&](int row, int a, int b) { for (int c = a + lane; c <= b; c += 64) wr(row, c, NUL); };
  // both ends of a row's nulling in one pass (lanes 0-31 the low interval, 32-63 the high one; wider intervals: the loops)
  auto null_halves = [&](int row, int a0, int b0, int a1, int b1) {
    const int n0 = b0 - a0 + 1, n1 = b1 - a1 + 1;
    if (imax(n0, n1) <= 0) return;
    if (imax(n0, n1) > 32) { null_wave(row, a0, b0); null_wave(row, a1, b1); return; }
    const bool low = lane < 32;
    const int a = low ? a0 : a1, n = low ? n0 : n1, jj = lane & 31;
    if (jj < n) wr(row, a + jj, NUL);
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) { s_tk = atomicAdd(ticket, 1u); s_bad = 0; }
    __syncthreads();
    const uint32_t tk = (uint32_t)U((int)s_tk);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int lo = ef ? -t.pattern_begin_free : 0, hi = ef ? t.text_begin_free : 0;
    if (lo < -pl) lo = -pl;
    if (hi > tl) hi = tl;
    const int offT = (pl + 15) / 16 + 3;
    bool overflow = pl > 32766 || tl > 32766 || hi - lo + PAD > CAP || offT + (tl + 15) / 16 + 3 > seqw;
    if (!overflow) {
      if (pack_pair_block(SQ, P, pl, T, tl, offT, tid, NT) && lane == 0) s_bad = 1;
      for (int q = tid; q < CAP; q += NT) ROWS32[q] = 0x80008000u;                  // both rows
    }
    __syncthreads();
    if (!overflow) {
      overflow = U(s_bad) != 0;
      if (!overflow) for (int k = lo + tid; k <= hi; k += NT) wr(0, k, (k > 0 ? k : 0) - 1);       // "score -1" in the row score 0 reads
    }
    __syncthreads();
    const PackedPair pk{SQ, offT, true};
    const int offT4 = offT * 4;
    const volatile lds_char* SQB = (const volatile lds_char*)SQ;
    int s = 0, steps_wait = 0;
    uint64_t W = 0;
    bool done = false;
    int r2lo = 1, r2hi = 0;                // what the write row still holds (the wavefront two scores back; null elsewhere)
    int r1lo = lo, r1hi = hi;              // what the read row holds
    while (!overflow) {
      if (hi - lo + PAD > CAP) { overflow = true; break; }
      W += (uint64_t)(hi - lo + 1);
      const int rr = s & 1, rw = rr ^ 1;
      if (ww == 0) null_halves(rw, r2lo, imin(r2hi, lo - 1), imax(r2lo, hi + 1), r2hi);
      int dmin = BIG;
      unsigned long long finm = 0ull;
      int qn = 0;
      auto finished = [&](int h, int k) -> bool {
        dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
        if (ef) { const int v = h - k; return (h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef); }
        return k == kend && h >= tl;
      };
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int q = 0; q < qn; ++q) {
              const int k = lo + U((int)queue[q]);
              int h = U(rd(rw, k));
              const int v = h - k;
              h += otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
              wr(rw, k, h);                           // the same value from every lane
              if (finished(h, k)) finm = ~0ull;
            }
            qn = 0;
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int k = 0, h = 0, v = 0;
            bool more = false, fin = false;
            if (act) {
              k = lo + (int)queue[q0 + lane];
              h = rd(rw, k);
              v = h - k;
              const int m = pk.match32(v, h, imin(pl - v, tl - h));
              v += m; h += m;
              more = (m == 32) && v < pl && h < tl;
              wr(rw, k, h);
              if (!more) fin = finished(h, k);
            }
            finm |= __ballot(fin);
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint16_t)(k - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      // ---- sweep of this wave's chunks (branch-free per lane)
      const int br = rr * (CAP * 2), bw = rw * (CAP * 2);
      for (int c = lo + 64 * ww; c <= hi; c += 64 * NW) {
        const int k = c + lane;
        const int a0 = (k & MASK) << 1, am = ((k - 1) & MASK) << 1, ap = ((k + 1) & MASK) << 1;
        const int o = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + br + a0);
        const int l = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + br + am);
        const int r = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + br + ap);
        const int mx = imax(imax(l, o) + 1, r);
        const int v = mx - k;
        const int t1 = tl - mx, t2 = pl - v;
        const bool inr = k <= hi;
        const unsigned long long inrm = __builtin_amdgcn_ballot_w64(inr);
        const unsigned long long vm = inrm & mk_ule((uint32_t)mx, (uint32_t)tl) & mk_ule((uint32_t)v, (uint32_t)pl);
        auto probe = [&](int pv, int ph) -> int {
          const int wp = (pv >> 2) & ~3, wt = offT4 + ((ph >> 2) & ~3);
          const uint32_t sp = (uint32_t)(pv & 15) * 2u, st = (uint32_t)(ph & 15) * 2u;
          const volatile lds_u32* pp = (const volatile lds_u32*)(SQB + wp);
          const volatile lds_u32* pt = (const volatile lds_u32*)(SQB + wt);
          const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], q0 = pt[0], q1 = pt[1], q2 = pt[2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, sp) ^ __builtin_amdgcn_alignbit(q1, q0, st);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, sp) ^ __builtin_amdgcn_alignbit(q2, q1, st);
          uint32_t flo, fhi;
          asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"(xl));
          asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"(xh));
          const uint32_t a = flo < (fhi | 32u) ? flo : (fhi | 32u);
          return (int)((a < 64u ? a : 64u) >> 1);
        };
        const int pm = probe(v, mx);
        int m = imin(imin(pm, t1), t2);
        unsigned long long mm = vm & mk_eq(imin(imin(pm, t1 - 1), t2 - 1), 32);
        if (mm) {
          const int pm2 = probe(v + 32, mx + 32);
          const int m2 = imin(imin(pm2, t1 - 32), t2 - 32);
          m = sel(mm, m2 + 32, m);
          mm &= mk_eq(imin(imin(pm2, t1 - 33), t2 - 33), 32);
        }
        const int h2 = mx + m;
        if (inr) *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bw + a0) = (int16_t)sel(vm, h2, NUL);      // lanes behind the range do not store
        const unsigned long long hm = vm & ~mm;
        const int lh = t1 - m, lv = t2 - m;
        int d;
        if (!ef) { d = imax(lh, lv); finm |= hm & mk_eq(k, kend) & mk_sle(lh, 0); }
        else {
          d = imin(imax(lh, lv - pef), imax(lv, lh - tef));
          finm |= hm & ((mk_sle(lh, 0) & mk_sle(lv, pef)) | (mk_sle(lv, 0) & mk_sle(lh, tef)));
        }
        dmin = imin(dmin, sel(hm, d, BIG));
        if (mm) {
          if (sel(mm, 1, 0)) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            queue[qn + rank] = (uint16_t)(k - lo);
          }
          qn += __builtin_popcountll(mm);
          if (qn + 64 > QCAP) drain();
        }
      }
      if (qn) drain();
      // ---- the one barrier of the score, then every wave reduces the NW (minimum, end flag) pairs
      {
        const int wd = wave_min_i32(dmin);
        if (lane == 0) { s_x[s & 1][ww][0] = wd; s_x[s & 1][ww][1] = finm != 0ull ? 1 : 0; }
      }
      lds_barrier();
      int mind = BIG, anyfin = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) { mind = imin(mind, U(s_x[s & 1][w2][0])); anyfin |= U(s_x[s & 1][w2][1]); }
      if (anyfin) { done = true; break; }
      // ---- the cut on the row just written (every wave: same inputs, same result); what it drops is nulled by every wave
      int clo = lo, chi = hi;
      wfadaptive_cut32(H, steps_wait, mind, pl, tl, ef, pef, tef, clo, chi, lane, [&](int k) { const int x = rd(rw, k); return x < 0 ? OTG_NULL_OFF : x; });
      clo = U(clo); chi = U(chi);
      null_halves(rw, lo, clo - 1, chi + 1, hi);
      r2lo = r1lo; r2hi = r1hi; r1lo = clo; r1hi = chi;
      lo = clo - 1 < -pl ? -pl : clo - 1;
      hi = chi + 1 > tl ? tl : chi + 1;
      ++s;
      if (s > pl + tl + 2) break;
    }
    if (ww == 0) {        // wave-uniform tail: every lane of wave 0 stores the same value to the same address
      if (done) {
        scores[ti] = s;
        if (cells) cells[ti] = W;
      } else if (overflow && overflow_list) {
        const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
        overflow_list[q] = ti;
      } else {
        scores[ti] = -1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Gap-affine, full op string.  Replaces WFAlignerGapAffine(x, o, e, Alignment, MemoryMed)::alignEnd2End / alignEndsFree +
// getAlignmentCigar() under the heuristic.  Recurrence, provenance bytes, row table and backtrace as in the exact chain's generic kernel
// (wfa_affine.hip), scores walked in units of g = gcd(x, o + e, e); every read of a history
// row is predicated on that row's OWN range (each score's wavefronts were cut to their own range), which is also what makes the
// modular window safe: two diagonals share a slot only if they are CAP apart, and no row is wider than CAP.
//   LDS policy (penalties (4,6,2) -> (2,4,1) only): M ring of 5 rows, I and D rings of 2, signed 16-bit offsets (I / D offsets run
//   past the text end like the reference's do — up to pl + tl — so the pair must satisfy pl + tl < 32767);
//   global policy: int32 rings in HBM / L2, any penalties, any lengths.
template <int CAP>
struct AffLds {
  volatile lds_i16* m; volatile lds_i16* i; volatile lds_i16* d;      // [rm][CAP], [ri][CAP], [ri][CAP]
  __device__ __forceinline__ bool fits(int pl, int tl, int xs, int oes, int es) const { return pl + tl < 32767 && xs == 2 && oes == 4 && es == 1; }
  __device__ __forceinline__ int cap() const { return CAP; }
  static __device__ __forceinline__ int ld(volatile lds_i16* row, int k) { const int x = row[k & (CAP - 1)]; return x < 0 ? OTG_NULL_OFF : x; }
  static __device__ __forceinline__ void stv(volatile lds_i16* row, int k, int h) { row[k & (CAP - 1)] = (int16_t)(h < 0 ? -32768 : h); }
  __device__ __forceinline__ int rdM(int r, int k) const { return ld(m + r * CAP, k); }
  __device__ __forceinline__ int rdI(int r, int k) const { return ld(i + r * CAP, k); }
  __device__ __forceinline__ int rdD(int r, int k) const { return ld(d + r * CAP, k); }
  __device__ __forceinline__ void wrM(int r, int k, int h) const { stv(m + r * CAP, k, h); }
  __device__ __forceinline__ void wrI(int r, int k, int h) const { stv(i + r * CAP, k, h); }
  __device__ __forceinline__ void wrD(int r, int k, int h) const { stv(d + r * CAP, k, h); }
  __device__ __forceinline__ void sync() const {}
};
struct AffGlobal {
  volatile int32_t* m; volatile int32_t* i; volatile int32_t* d; int capa; int kb;
  __device__ __forceinline__ bool fits(int pl, int tl, int, int, int) const { return pl + tl + 3 <= capa; }
  __device__ __forceinline__ int cap() const { return capa; }
  __device__ __forceinline__ int rdM(int r, int k) const { return m[(size_t)r * capa + k + kb]; }
  __device__ __forceinline__ int rdI(int r, int k) const { return i[(size_t)r * capa + k + kb]; }
  __device__ __forceinline__ int rdD(int r, int k) const { return d[(size_t)r * capa + k + kb]; }
  __device__ __forceinline__ void wrM(int r, int k, int h) const { m[(size_t)r * capa + k + kb] = h < 0 ? OTG_NULL_OFF : h; }
  __device__ __forceinline__ void wrI(int r, int k, int h) const { i[(size_t)r * capa + k + kb] = h < 0 ? OTG_NULL_OFF : h; }
  __device__ __forceinline__ void wrD(int r, int k, int h) const { d[(size_t)r * capa + k + kb] = h < 0 ? OTG_NULL_OFF : h; }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
};

// CAP > 0: LDS policy with a window of CAP diagonals; CAP == 0: global policy.  RMAX = ring depth the LDS range tables hold.
template <int CAP, int QCAP, int WPB, int RMAX>
__global__ __launch_bounds__(WPB * 64) void wfa_affine_adaptive_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int xs, int oes, int es, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, Heur H, int seqw)
{
  constexpr bool GLOBAL_WF = CAP == 0;
  constexpr int LCAP = GLOBAL_WF ? 2 : CAP;
  constexpr int QWORDS = QCAP < 512 ? 512 : QCAP;            // the backtrace stages its window (2 KB) in the queue
  __shared__ __attribute__((aligned(16))) int16_t s_rows[WPB][9][LCAP];      // M 0..4, I 5..6, D 7..8
  __shared__ __attribute__((aligned(16))) uint32_t s_q[WPB][QWORDS];
  __shared__ int s_rng[WPB][6][RMAX];                         // mlo, mhi, ilo, ihi, dlo, dhi per ring row
  extern __shared__ uint32_t s_dyn[];                         // [WPB][seqw]: the pair packed to 2 bits per base (seqw == 0: byte probes from HBM only)
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  uint8_t* my = ws.base + (size_t)(blockIdx.x * WPB + wib) * ws.stride;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  volatile __attribute__((address_space(3))) int* mlo = (volatile __attribute__((address_space(3))) int*)&s_rng[wib][0][0];
  volatile __attribute__((address_space(3))) int* mhi = mlo + RMAX;
  volatile __attribute__((address_space(3))) int* ilo = mlo + 2 * RMAX;
  volatile __attribute__((address_space(3))) int* ihi = mlo + 3 * RMAX;
  volatile __attribute__((address_space(3))) int* dlo = mlo + 4 * RMAX;
  volatile __attribute__((address_space(3))) int* dhi = mlo + 5 * RMAX;
  auto U = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  volatile lds_u32* queue = (volatile lds_u32*)&s_q[wib][0];
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  // the LDS policy only takes the penalties (2,4,1): ring depths 5 and 2 at compile time (a remainder by a run-time divisor is ~40 instructions)
  const int rm = GLOBAL_WF ? ws.rm : 5, ri = GLOBAL_WF ? ws.ri : 2;

  for (;;) {
    const uint32_t tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int s_end = -1, k_end = 0;
    uint64_t W = 0;
    bool fail = false;
    PackedPair pk{(volatile lds_u32*)(s_dyn + (size_t)wib * seqw), 0, false};
    if (seqw > 0 && pl + tl < 32767) pk.init(P, pl, T, tl, lane, seqw);

    auto forward = [&](auto st) {
      fail = !st.fits(pl, tl, xs, oes, es) || ws.rm > RMAX || ws.ri > RMAX;
      size_t slab_top = 0;
      int steps_wait = 0;
      for (int s = 0; !fail; ++s) {
        if (s >= ws.nrows) { fail = true; break; }
        const int sm = s % rm, si = s % ri;
        int lo, hi;
        int qx = -1, qo = -1, qe = -1;
        int mxlo = 1, mxhi = 0, molo = 1, mohi = 0, ielo = 1, iehi = 0, delo = 1, dehi = 0;
        if (s == 0) {
          lo = ef ? imax(-t.pattern_begin_free, -pl) : 0;
          hi = ef ? imin(t.text_begin_free, tl) : 0;
        } else {
          lo = 1 << 30; hi = -(1 << 30);
          if (s - xs >= 0) { qx = (s - xs) % rm; mxlo = U(mlo[qx]); mxhi = U(mhi[qx]); }
          if (s - oes >= 0) { qo = (s - oes) % rm; molo = U(mlo[qo]); mohi = U(mhi[qo]); }
          if (s - es >= 0) { qe = (s - es) % ri; ielo = U(ilo[qe]); iehi = U(ihi[qe]); delo = U(dlo[qe]); dehi = U(dhi[qe]); }
          if (mxhi >= mxlo) { lo = imin(lo, mxlo); hi = imax(hi, mxhi); }
          if (mohi >= molo) { lo = imin(lo, molo - 1); hi = imax(hi, mohi + 1); }
          if (iehi >= ielo) { lo = imin(lo, ielo + 1); hi = imax(hi, iehi + 1); }
          if (dehi >= delo) { lo = imin(lo, delo - 1); hi = imax(hi, dehi - 1); }
          if (lo < -pl) lo = -pl;
          if (hi > tl) hi = tl;
        }
        if (hi < lo) {       // null wavefront: this score is not reachable (the reference skips the heuristic too)
          mlo[sm] = 1; mhi[sm] = 0; ilo[si] = 1; ihi[si] = 0; dlo[si] = 1; dhi[si] = 0;
          rowtab[s] = -1;
          if (s > 2 * (oes + es * (pl + tl)) + 8) fail = true;
          continue;
        }
        const int width = hi - lo + 1;
        if (width > st.cap() || slab_top + (size_t)width > ws.slab_bytes) { fail = true; break; }
        uint8_t* btrow = slab + slab_top - lo;     // btrow[k]
        rowtab[s] = (int64_t)slab_top - lo;      // wave-uniform store
        slab_top += (size_t)width;
        W += 3ull * (uint64_t)width;
        int dmin = BIG, kfin = BIG;
        int qn = 0;
        auto finished = [&](int h, int k) {
          dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
          if (ef) { const int v = h - k; if ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)) kfin = imin(kfin, k); }
        };
        auto drain = [&]() {
          st.sync();
          int pass = 0;
          while (qn > 0) {
            if (qn <= 4 && pass > 0) {
              for (int e = 0; e < qn; ++e) {
                const int kk = lo + U((int)queue[e]);
                int h = U(st.rdM(sm, kk));
                const int v = h - kk;
                const int m = otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
                h += m;
                st.wrM(sm, kk, h);                // the same value from every lane
                finished(h, kk);
              }
              qn = 0;
              st.sync();
              break;
            }
            int wq = 0;
            for (int q0 = 0; q0 < qn; q0 += 64) {
              const bool act = q0 + lane < qn;
              int kk = 0, h = 0, v = 0;
              bool more = false;
              if (act) {
                kk = lo + (int)queue[q0 + lane];
                h = st.rdM(sm, kk);
                v = h - kk;
                const int rem = imin(pl - v, tl - h);
                int m, full;
                if (pk.ok) { m = pk.match32(v, h, rem); full = 32; }
                else if (pass == 0) {
                  const uint64_t xl = otg_load8(P + v) ^ otg_load8(T + h), xh = otg_load8(P + v + 8) ^ otg_load8(T + h + 8);
                  m = xl ? (__builtin_ctzll(xl) >> 3) : (xh ? 8 + (__builtin_ctzll(xh) >> 3) : 16);
                  m = imin(m, rem); full = 16;
                } else { m = otg_match64(P, T, v, h, rem); full = 64; }
                v += m; h += m;
                more = (m == full) && v < pl && h < tl;
                st.wrM(sm, kk, h);
                if (!more) finished(h, kk);
              }
              const unsigned long long mm = __ballot(more);
              if (more) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                queue[wq + rank] = (uint32_t)(kk - lo);
              }
              wq += __builtin_popcountll(mm);
            }
            qn = wq; ++pass;
            st.sync();
          }
        };
        for (int c = lo; c <= hi; c += 64) {
          const int k = c + lane;
          const bool in = k <= hi;
          int mx, ins = OTG_NULL_OFF, del = OTG_NULL_OFF;
          uint32_t bits = 0;
          if (s == 0) {
            mx = k > 0 ? k : 0;
          } else {
            int io = OTG_NULL_OFF, dop = OTG_NULL_OFF, ix = OTG_NULL_OFF, dx = OTG_NULL_OFF, mm = OTG_NULL_OFF;
            if (in) {
              if (k - 1 >= molo && k - 1 <= mohi) io = st.rdM(qo, k - 1);
              if (k + 1 >= molo && k + 1 <= mohi) dop = st.rdM(qo, k + 1);
              if (k - 1 >= ielo && k - 1 <= iehi) ix = st.rdI(qe, k - 1);
              if (k + 1 >= delo && k + 1 <= dehi) dx = st.rdD(qe, k + 1);
              if (k >= mxlo && k <= mxhi) mm = st.rdM(qx, k);
            }
            if (ix >= io) { ins = ix; bits |= 4u; } else ins = io;
            ins += 1;
            if (dx >= dop) { del = dx; bits |= 8u; } else del = dop;
            const int mis = mm + 1;
            mx = imax(del, imax(mis, ins));
            uint32_t org = 0;
            if (mx == ins) org = 2;
            if (mx == del) org = 1;
            if (mx == mis) org = 0;
            bits |= org;
            if (ins < 0) ins = OTG_NULL_OFF;
            if (del < 0) del = OTG_NULL_OFF;
          }
          int h = mx, v = mx - k;
          const bool valid = in && mx >= 0 && h <= tl && v <= pl;
          bool more = false;
          if (valid && v < pl && h < tl) {
            int m, full;
            if (pk.ok) { m = pk.match32(v, h, imin(pl - v, tl - h)); full = 32; }
            else {
              const uint64_t xx = otg_load8(P + v) ^ otg_load8(T + h);
              m = xx ? (__builtin_ctzll(xx) >> 3) : 8;
              m = imin(m, imin(pl - v, tl - h)); full = 8;
            }
            v += m; h += m;
            more = (m == full) && v < pl && h < tl;
          }
          if (in) {
            st.wrM(sm, k, valid ? h : OTG_NULL_OFF);
            if (s > 0) { st.wrI(si, k, ins); st.wrD(si, k, del); }
            btrow[k] = (uint8_t)bits;
          }
          if (valid && !more) finished(h, k);
          const unsigned long long mq = __ballot(more);
          if (more) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
            queue[qn + rank] = (uint32_t)(k - lo);
          }
          qn += __builtin_popcountll(mq);
          if (qn + 64 > QCAP) drain();
        }
        drain();
        // termination on the fully extended wavefront: end-to-end the end diagonal; ends-free the lowest diagonal that qualifies
        bool done = false;
        if (!ef) {
          if (kend >= lo && kend <= hi) { const int x = U(st.rdM(sm, kend)); if (x >= tl) { done = true; k_end = kend; } }
        } else {
          const int kf = wave_min_i32(kfin);
          if (kf != BIG) { done = true; k_end = kf; }
        }
        if (done) { s_end = s; break; }
        // the cut: M[s], and the score's I / D wavefronts to the same range
        int clo = lo, chi = hi;
        const int mind = wave_min_i32(dmin);
        wfadaptive_cut(H, steps_wait, mind, pl, tl, ef, pef, tef, clo, chi, lane, [&](int k) { return st.rdM(sm, k); });
        mlo[sm] = clo; mhi[sm] = chi;
        if (s == 0) { ilo[si] = 1; ihi[si] = 0; dlo[si] = 1; dhi[si] = 0; }       // no I / D wavefront at score 0
        else { ilo[si] = clo; ihi[si] = chi; dlo[si] = clo; dhi[si] = chi; }
        st.sync();
      }
    };
    if constexpr (GLOBAL_WF) {
      int32_t* ringM = (int32_t*)my;
      int32_t* ringI = ringM + (size_t)ws.rm * ws.capa;
      int32_t* ringD = ringI + (size_t)ws.ri * ws.capa;
      forward(AffGlobal{ringM, ringI, ringD, ws.capa, pl + 1});
    } else {
      volatile lds_i16* rows = (volatile lds_i16*)&s_rows[wib][0][0];
      forward(AffLds<CAP>{rows, rows + 5 * CAP, rows + 7 * CAP});
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // row table and provenance rows: written by lane 0 / every lane, read by all

    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }
    if (!backtrace_unpack<false>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g,
                                 (volatile lds_u32*)&s_q[wib][0], EqBytes{P, T})) continue;
    if (cells) cells[ti] = W;
  }
}

// ---------------------------------------------------------------------------------------------------
// The fast gap-affine tier (penalties (4,6,2) -> (2,4,1), pairs with pl + tl < 32766 that pack): the same recurrence and provenance as the kernel
// above, its sweep written BRANCH-FREE like the fast edit tier's.  Eleven rows of CAP signed 16-bit offsets per wave in a modular window: the M
// ring (5), the I and D rings (2 each), a row that is always null (what a score reads in place of wavefronts that do not exist yet) and a seed
// row (the start diagonals' offsets minus one, read as "M[s-2]" by score 0, so that score 0 is a sweep like any other).  The null discipline of the
// edit tier holds PER ROW — a slot is non-null only while its diagonal lies in the range its row currently stands for — so the five operands of a
// cell are read unconditionally: when a ring row is taken over by a new score, what its previous score left outside the new range is nulled, and
// so is what a cut drops.  Every lane of a chunk computes, probes and stores (NULL where the lane lies behind the range); a provenance row is
// padded to whole chunks for the same reason.
template <int CAP, int QCAP, int WPB, bool MASKED>
__global__ __launch_bounds__(WPB * 64) void wfa_affine_adaptive_lds_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, Heur H, int seqw)
{
  constexpr int xs = 2, oes = 4, es = 1, RM = 5, RI = 2;
  constexpr int ROW_I = RM, ROW_D = RM + RI, ROW_NULL = RM + 2 * RI, ROW_SEED = ROW_NULL + 1, NROWS = ROW_SEED + 1;
  constexpr int MASK = CAP - 1, NUL = -32768;
  constexpr int PAD = MASKED ? 4 : 68;        // diagonals of the window a range must leave free (unmasked: the last chunk's 64 lanes all store)
  static_assert(NROWS * CAP * 2 >= 2048, "the backtrace stages its 2 KB window in the rows (free once the forward pass is over)");
  __shared__ __attribute__((aligned(16))) int16_t s_rows[WPB][NROWS][CAP];
  __shared__ __attribute__((aligned(16))) uint32_t s_q[WPB][QCAP];
  extern __shared__ uint32_t s_dyn[];                         // [WPB][seqw]: the packed pairs
  using lds_char = __attribute__((address_space(3))) char;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  uint8_t* my = ws.base + (size_t)(blockIdx.x * WPB + wib) * ws.stride;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  volatile lds_char* ROWS = (volatile lds_char*)&s_rows[wib][0][0];
  volatile lds_u32* ROWS32 = (volatile lds_u32*)&s_rows[wib][0][0];
  auto U = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  volatile lds_u32* queue = (volatile lds_u32*)&s_q[wib][0];
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  auto rd = [&](int row, int k) -> int { return *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)); };
  auto wr = [&](int row, int k, int v) { *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)) = (int16_t)v; };
  // nulls the diagonals [a, b] of a row (a no-op for an empty interval)
  auto null_range = [&](int row, int a, int b) { for (int c = a; c <= b; c += 64) if (c + lane <= b) wr(row, c + lane, NUL); };
  // The nulling of a score in ONE pass each.  Ranges move by a few diagonals per score, so what a take-over or a cut has to null is four (two)
  // intervals a few diagonals wide; as six loops per event they were half of the kernel's scalar instructions (and the scalar unit, shared by the
  // four SIMDs of a CU, was its busiest resource).  null_quarters: lanes 16 q .. 16 q + 15 take interval q — intervals 0 and 1 of row rA, 2 and 3
  // of rows rB AND rC; null_halves: lanes 0-31 / 32-63 take the two intervals of three rows.  Wider intervals fall back to the loops.
  const int lq = lane >> 4, lj = lane & 15;
  auto null_quarters = [&](int rA, int a0, int b0, int a1, int b1, int rB, int rC, int a2, int b2, int a3, int b3) {
    const int n0 = b0 - a0 + 1, n1 = b1 - a1 + 1, n2 = b2 - a2 + 1, n3 = b3 - a3 + 1;
    const int nmax = imax(imax(n0, n1), imax(n2, n3));
    if (nmax <= 0) return;
    if (nmax > 16) {
      null_range(rA, a0, b0); null_range(rA, a1, b1);
      null_range(rB, a2, b2); null_range(rB, a3, b3); null_range(rC, a2, b2); null_range(rC, a3, b3);
      return;
    }
    const int a = lq < 2 ? (lq == 0 ? a0 : a1) : (lq == 2 ? a2 : a3);
    const int n = lq < 2 ? (lq == 0 ? n0 : n1) : (lq == 2 ? n2 : n3);
    if (lj < n) {
      const int ad = ((a + lj) & MASK) << 1;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + (lq < 2 ? rA : rB) * (CAP * 2) + ad) = (int16_t)NUL;
      if (lq >= 2) *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + rC * (CAP * 2) + ad) = (int16_t)NUL;
    }
  };
  auto null_halves = [&](int r0, int r1, int r2, int a0, int b0, int a1, int b1) {
    const int n0 = b0 - a0 + 1, n1 = b1 - a1 + 1;
    if (imax(n0, n1) <= 0) return;
    if (imax(n0, n1) > 32) {
      null_range(r0, a0, b0); null_range(r0, a1, b1); null_range(r1, a0, b0); null_range(r1, a1, b1); null_range(r2, a0, b0); null_range(r2, a1, b1);
      return;
    }
    const bool low = lane < 32;
    const int a = low ? a0 : a1, n = low ? n0 : n1, jj = lane & 31;
    if (jj < n) {
      const int ad = ((a + jj) & MASK) << 1;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r0 * (CAP * 2) + ad) = (int16_t)NUL;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r1 * (CAP * 2) + ad) = (int16_t)NUL;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r2 * (CAP * 2) + ad) = (int16_t)NUL;
    }
  };

  for (;;) {
    const uint32_t tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int s_end = -1, k_end = 0;
    uint64_t W = 0;
    bool fail = pl + tl >= 32766;
    PackedPair pk{(volatile lds_u32*)(s_dyn + (size_t)wib * seqw), 0, false};
    if (!fail) { pk.init(P, pl, T, tl, lane, seqw); fail = !pk.ok; }
    const int offT4 = pk.offT * 4;
    const volatile lds_char* SQB = (const volatile lds_char*)pk.SQ;
    size_t slab_top = 0;
    int steps_wait = 0;
    if (!fail) {
      for (int q = lane; q < NROWS * CAP / 2; q += 64) ROWS32[q] = 0x80008000u;
      const int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
      if (hi0 - lo0 + PAD > CAP) fail = true;
      else for (int c = lo0; c <= hi0; c += 64) { const int k = c + lane; if (k <= hi0) wr(ROW_SEED, k, (k > 0 ? k : 0) - 1); }
    }
    // the ranges the ring rows stand for, in scalar registers: M[s-1] .. M[s-5] (the last one = what this score's own row still holds) and the
    // I / D wavefronts of s-1 and s-2; (1, 0) = null
    int r1lo = 1, r1hi = 0, r2lo = 1, r2hi = 0, r3lo = 1, r3hi = 0, r4lo = 1, r4hi = 0, r5lo = 1, r5hi = 0, i1lo = 1, i1hi = 0, i2lo = 1, i2hi = 0;
    int sm = RM - 1, si = 1;
    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      sm = sm + 1 == RM ? 0 : sm + 1; si ^= 1;                   // s % 5, s & 1
      // the rows this score reads (the null row where the wavefront does not exist) and the range it covers
      int lo, hi;
      int qx = ROW_NULL, qo = ROW_NULL, qi = ROW_NULL, qd = ROW_NULL;
      if (s == 0) {
        lo = ef ? imax(-t.pattern_begin_free, -pl) : 0;
        hi = ef ? imin(t.text_begin_free, tl) : 0;
        qx = ROW_SEED;
      } else {
        lo = 1 << 30; hi = -(1 << 30);
        if (r2hi >= r2lo) { lo = imin(lo, r2lo); hi = imax(hi, r2hi); qx = sm >= xs ? sm - xs : sm - xs + RM; }
        if (r4hi >= r4lo) { lo = imin(lo, r4lo - 1); hi = imax(hi, r4hi + 1); qo = sm >= oes ? sm - oes : sm - oes + RM; }
        if (i1hi >= i1lo) { lo = imin(lo, i1lo - 1); hi = imax(hi, i1hi + 1); qi = ROW_I + (si ^ 1); qd = ROW_D + (si ^ 1); }      // (the I and D wavefronts of a score share their range)
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
      }
      // the ring rows of this score are taken over: what their previous scores left is nulled (outside the new range; everything when the score is unreachable)
      const int omlo = r5lo, omhi = r5hi, oilo = i2lo, oihi = i2hi;
      r5lo = r4lo; r5hi = r4hi; r4lo = r3lo; r4hi = r3hi; r3lo = r2lo; r3hi = r2hi; r2lo = r1lo; r2hi = r1hi; i2lo = i1lo; i2hi = i1hi;
      if (hi < lo) {       // null wavefront: this score is not reachable (the reference skips the heuristic too)
        null_quarters(sm, omlo, omhi, 1, 0, ROW_I + si, ROW_D + si, oilo, oihi, 1, 0);
        r1lo = 1; r1hi = 0; i1lo = 1; i1hi = 0;
        rowtab[s] = -1;
        if (s > 2 * (oes + es * (pl + tl)) + 8) fail = true;
        continue;
      }
      const int width = hi - lo + 1, padded = ((width + 63) >> 6) << 6;
      if (width + PAD > CAP || slab_top + (size_t)padded > ws.slab_bytes) { fail = true; break; }
      null_quarters(sm, omlo, imin(omhi, lo - 1), imax(omlo, hi + 1), omhi, ROW_I + si, ROW_D + si, oilo, imin(oihi, lo - 1), imax(oilo, hi + 1), oihi);
      uint8_t* btbase = slab + slab_top;         // provenance byte of diagonal k: btbase[k - lo]; the row is padded to whole chunks
      rowtab[s] = (int64_t)slab_top - lo;        // wave-uniform store
      slab_top += (size_t)padded;
      W += 3ull * (uint64_t)width;
      int dmin = BIG, kfin = BIG;
      int qn = 0;
      auto finished = [&](int h, int k) {
        dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
        if (ef) { const int v = h - k; if ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)) kfin = imin(kfin, k); }
        else if (k == kend && h >= tl) kfin = k;
      };
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int e = 0; e < qn; ++e) {
              const int kk = lo + U((int)queue[e]);
              int h = U(rd(sm, kk));
              const int v = h - kk;
              h += otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
              wr(sm, kk, h);                         // the same value from every lane
              finished(h, kk);
            }
            qn = 0;
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int kk = 0, h = 0, v = 0;
            bool more = false;
            if (act) {
              kk = lo + (int)queue[q0 + lane];
              h = rd(sm, kk);
              v = h - kk;
              const int m = pk.match32(v, h, imin(pl - v, tl - h));
              v += m; h += m;
              more = (m == 32) && v < pl && h < tl;
              wr(sm, kk, h);
              if (!more) finished(h, kk);
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint32_t)(kk - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      // ---- sweep (branch-free per lane)
      const int bx = qx * (CAP * 2), bo = qo * (CAP * 2), bi = qi * (CAP * 2), bd = qd * (CAP * 2);
      const int bm = sm * (CAP * 2), bI = (ROW_I + si) * (CAP * 2), bD = (ROW_D + si) * (CAP * 2);
      for (int c = lo; c <= hi; c += 64) {
        const int k = c + lane;
        const int a0 = (k & MASK) << 1, am = ((k - 1) & MASK) << 1, ap = ((k + 1) & MASK) << 1;
        const int io = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bo + am), dop = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bo + ap);
        const int ix = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bi + am), dx = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bd + ap);
        const int mm = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bx + a0);
        const bool ext_i = ix >= io, ext_d = dx >= dop;
        const int insv = imax(ix, io) + 1, delv = imax(dx, dop), mis = mm + 1;
        const int mx = imax(imax(delv, mis), insv);
        const uint32_t org = mx == mis ? 0u : (mx == delv ? 1u : 2u);      // mismatch wins ties over deletion over insertion
        const uint32_t bits = org | (ext_i ? 4u : 0u) | (ext_d ? 8u : 0u);
        // per-lane conditions as lane masks on the scalar unit from here on (see mk_eq)
        const bool inr = k <= hi;
        const unsigned long long inrm = __builtin_amdgcn_ballot_w64(inr);
        const int v = mx - k;
        const int t1 = tl - mx, t2 = pl - v;
        const unsigned long long vm = inrm & mk_ule((uint32_t)mx, (uint32_t)tl) & mk_ule((uint32_t)v, (uint32_t)pl);      // valid cells
        auto probe = [&](int pv, int ph) -> int {      // equal leading bases of pattern[pv ..] and text[ph ..], at most 32
          const int wp = (pv >> 2) & ~3, wt = offT4 + ((ph >> 2) & ~3);
          const uint32_t sp = (uint32_t)(pv & 15) * 2u, st = (uint32_t)(ph & 15) * 2u;
          const volatile lds_u32* pp = (const volatile lds_u32*)(SQB + wp);
          const volatile lds_u32* pt = (const volatile lds_u32*)(SQB + wt);
          const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], q0 = pt[0], q1 = pt[1], q2 = pt[2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, sp) ^ __builtin_amdgcn_alignbit(q1, q0, st);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, sp) ^ __builtin_amdgcn_alignbit(q2, q1, st);
          uint32_t flo, fhi;
          asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"(xl));
          asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"(xh));
          const uint32_t a = flo < (fhi | 32u) ? flo : (fhi | 32u);
          return (int)((a < 64u ? a : 64u) >> 1);
        };
        const int pm = probe(v, mx);
        int m = imin(imin(pm, t1), t2);
        unsigned long long mq = vm & mk_eq(imin(imin(pm, t1 - 1), t2 - 1), 32);            // more: a full probe with more than 32 bases left of both sequences
        if (mq) {             // a second probe by the whole wave where a run outlives the first (as in the edit tier): the queue is for runs beyond 64 bases
          const int pm2 = probe(v + 32, mx + 32);
          const int m2 = imin(imin(pm2, t1 - 32), t2 - 32);
          m = sel(mq, m2 + 32, m);
          mq &= mk_eq(imin(imin(pm2, t1 - 33), t2 - 33), 32);
        }
        const int h2 = mx + m;
        if (!MASKED || inr) {       // MASKED: lanes behind the range do not store, the window is usable up to CAP - 4 diagonals (one execution-mask region per chunk)
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bm + a0) = (int16_t)sel(vm, h2, NUL);
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bI + a0) = (int16_t)sel(inrm & mk_sle(0, insv), insv, NUL);
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bD + a0) = (int16_t)sel(inrm & mk_sle(0, delv), delv, NUL);
        }
        btbase[(uint32_t)(k - lo)] = (uint8_t)bits;                      // every lane stores: the row is padded (uniform base + unsigned lane offset)
        const unsigned long long hm = vm & ~mq;                          // cells that are final here
        const int lh = t1 - m, lv = t2 - m;
        int d;
        unsigned long long fm;
        if (!ef) { d = imax(lh, lv); fm = hm & mk_eq(k, kend) & mk_sle(lh, 0); }
        else { d = imin(imax(lh, lv - pef), imax(lv, lh - tef)); fm = hm & ((mk_sle(lh, 0) & mk_sle(lv, pef)) | (mk_sle(lv, 0) & mk_sle(lh, tef))); }
        if (fm) kfin = imin(kfin, sel(fm, k, BIG));
        dmin = imin(dmin, sel(hm, d, BIG));
        if (mq) {
          if (sel(mq, 1, 0)) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
            queue[qn + rank] = (uint32_t)(k - lo);
          }
          qn += __builtin_popcountll(mq);
          if (qn + 64 > QCAP) drain();
        }
      }
      if (qn) drain();
      // termination on the fully extended wavefront: end-to-end the end diagonal; ends-free the lowest diagonal that qualifies
      if (__ballot(kfin != BIG) != 0ull) { k_end = wave_min_i32(kfin); s_end = s; break; }
      // the cut: M[s], and the score's I / D wavefronts to the same range; what it drops is nulled
      int clo = lo, chi = hi;
      const int mind = wave_min_i32(dmin);
      wfadaptive_cut32(H, steps_wait, mind, pl, tl, ef, pef, tef, clo, chi, lane, [&](int k) { const int x = rd(sm, k); return x < 0 ? OTG_NULL_OFF : x; });
      null_halves(sm, ROW_I + si, ROW_D + si, lo, clo - 1, chi + 1, hi);
      r1lo = clo; r1hi = chi;
      if (s == 0) { i1lo = 1; i1hi = 0; }       // no I / D wavefront at score 0 (its sweep stored nulls: both came from the null row)
      else { i1lo = clo; i1hi = chi; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // row table and provenance rows are read back by all lanes

    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }
    if (!backtrace_unpack<false>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g,
                                 (volatile lds_u32*)&s_rows[wib][0][0], EqPacked{pk.SQ, pk.offT})) continue;
    if (cells) cells[ti] = W;
  }
}

// ---------------------------------------------------------------------------------------------------
// The wide gap-affine tiers: NW waves on ONE alignment.  What reaches them are the alignments whose wavefront outgrows the 256-diagonal window —
// 5-15 chunks of 64 diagonals per score — and a wave on its own spends a score waiting for LDS, chunk after chunk, with less than one wave per
// SIMD resident (the provenance slab of an alignment, not LDS, bounds how many are in flight).  Here the chunks of a score are dealt to the waves
// of the block (chunk j to wave j mod NW); everything else of a score is REPLICATED: every wave keeps the ranges, the row indices, the slab
// cursor and the cut's step counter in its own scalar registers and computes the same values from the same inputs, so a score needs one barrier:
//
//   take-over nulling (wave 0; disjoint from what the sweep stores)  ->  sweep of the own chunks, own queue, own drain  ->  own minimum of
//   "left to align" and own end candidate to LDS  ->  BARRIER  ->  every wave reduces the NW pairs, every wave computes the cut from the M row
//   and nulls what it drops.
//
// The cut is safe to compute while another wave already nulls: a wave nulls only diagonals the cut dropped, i.e. diagonals that were NOT within
// the threshold, and a null offset reads as "not within the threshold" — a late wave finds the same first diagonal from either end.  The rows a
// wave that runs ahead writes in the next score (M[s+1], the I / D rows of the other parity) are not the rows a late wave still reads (M[s]).
// The exchange words alternate with the parity of the score.  The backtrace is wave 0's; the others wait at the next ticket.
template <int CAP, int QCAP, int NW>
__global__ __launch_bounds__(NW * 64) void wfa_affine_adaptive_mw_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, Heur H, int seqw)
{
  constexpr int xs = 2, oes = 4, es = 1, RM = 5, RI = 2;
  constexpr int ROW_I = RM, ROW_D = RM + RI, ROW_NULL = RM + 2 * RI, ROW_SEED = ROW_NULL + 1, NROWS = ROW_SEED + 1;
  constexpr int MASK = CAP - 1, NUL = -32768, PAD = 4, NT = NW * 64;
  __shared__ __attribute__((aligned(16))) int16_t s_rows[NROWS][CAP];
  __shared__ __attribute__((aligned(16))) uint32_t s_q[NW][QCAP];
  __shared__ int s_x[2][NW][2];
  __shared__ uint32_t s_tk;
  __shared__ int s_bad;
  extern __shared__ uint32_t s_dyn[];                         // [seqw]: the packed pair of the block's alignment
  using lds_char = __attribute__((address_space(3))) char;
  const int tid = threadIdx.x, lane = tid & 63;
  auto U = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  const int ww = U(tid >> 6);
  uint8_t* my = ws.base + (size_t)blockIdx.x * ws.stride;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  volatile lds_char* ROWS = (volatile lds_char*)&s_rows[0][0];
  volatile lds_u32* ROWS32 = (volatile lds_u32*)&s_rows[0][0];
  volatile lds_u32* queue = (volatile lds_u32*)&s_q[ww][0];
  volatile lds_u32* SQ = (volatile lds_u32*)s_dyn;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  auto rd = [&](int row, int k) -> int { return *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)); };
  auto wr = [&](int row, int k, int v) { *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + row * (CAP * 2) + ((k & MASK) << 1)) = (int16_t)v; };
  // the barrier of a score orders LDS only: the provenance bytes and the row table (global memory) are read back after the forward pass, behind
  // a full __syncthreads(); waiting for their acknowledgement at every score costs more than the score's sweep
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto null_wave = [&](int row, int a, int b) { for (int c = a + lane; c <= b; c += 64) wr(row, c, NUL); };          // (this wave nulls all of it)
  // one-pass nulling as in the one-wave tier: quarters of a wave for the four intervals of a take-over (wave 0's job: the rows taken over are next
  // read behind this score's barrier), halves for the two intervals of a cut (every wave's job, see above)
  const int lq = lane >> 4, lj = lane & 15;
  auto null_quarters = [&](int rA, int a0, int b0, int a1, int b1, int rB, int rC, int a2, int b2, int a3, int b3) {
    const int n0 = b0 - a0 + 1, n1 = b1 - a1 + 1, n2 = b2 - a2 + 1, n3 = b3 - a3 + 1;
    const int nmax = imax(imax(n0, n1), imax(n2, n3));
    if (nmax <= 0) return;
    if (nmax > 16) {
      null_wave(rA, a0, b0); null_wave(rA, a1, b1);
      null_wave(rB, a2, b2); null_wave(rB, a3, b3); null_wave(rC, a2, b2); null_wave(rC, a3, b3);
      return;
    }
    const int a = lq < 2 ? (lq == 0 ? a0 : a1) : (lq == 2 ? a2 : a3);
    const int n = lq < 2 ? (lq == 0 ? n0 : n1) : (lq == 2 ? n2 : n3);
    if (lj < n) {
      const int ad = ((a + lj) & MASK) << 1;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + (lq < 2 ? rA : rB) * (CAP * 2) + ad) = (int16_t)NUL;
      if (lq >= 2) *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + rC * (CAP * 2) + ad) = (int16_t)NUL;
    }
  };
  auto null_halves = [&](int r0, int r1, int r2, int a0, int b0, int a1, int b1) {
    const int n0 = b0 - a0 + 1, n1 = b1 - a1 + 1;
    if (imax(n0, n1) <= 0) return;
    if (imax(n0, n1) > 32) {
      null_wave(r0, a0, b0); null_wave(r0, a1, b1); null_wave(r1, a0, b0); null_wave(r1, a1, b1); null_wave(r2, a0, b0); null_wave(r2, a1, b1);
      return;
    }
    const bool low = lane < 32;
    const int a = low ? a0 : a1, n = low ? n0 : n1, jj = lane & 31;
    if (jj < n) {
      const int ad = ((a + jj) & MASK) << 1;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r0 * (CAP * 2) + ad) = (int16_t)NUL;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r1 * (CAP * 2) + ad) = (int16_t)NUL;
      *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + r2 * (CAP * 2) + ad) = (int16_t)NUL;
    }
  };

  for (;;) {
    __syncthreads();                      // the previous alignment is over for every wave (wave 0 staged its backtrace in the rows)
    if (tid == 0) { s_tk = atomicAdd(ticket, 1u); s_bad = 0; }
    __syncthreads();
    const uint32_t tk = (uint32_t)U((int)s_tk);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    const int kend = tl - pl;
    int s_end = -1, k_end = 0;
    uint64_t W = 0;
    const int offT = (pl + 15) / 16 + 3;
    bool fail = pl + tl >= 32766 || offT + (tl + 15) / 16 + 3 > seqw;
    const int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
    if (hi0 - lo0 + PAD > CAP) fail = true;
    if (!fail) {
      if (pack_pair_block(SQ, P, pl, T, tl, offT, tid, NT) && lane == 0) s_bad = 1;
      for (int q = tid; q < NROWS * CAP / 2; q += NT) ROWS32[q] = 0x80008000u;
    }
    __syncthreads();
    if (!fail) {
      fail = U(s_bad) != 0;
      if (!fail) for (int k = lo0 + tid; k <= hi0; k += NT) wr(ROW_SEED, k, (k > 0 ? k : 0) - 1);
    }
    __syncthreads();
    const PackedPair pk{SQ, offT, true};
    const int offT4 = offT * 4;
    const volatile lds_char* SQB = (const volatile lds_char*)SQ;
    size_t slab_top = 0;
    int steps_wait = 0;
    int r1lo = 1, r1hi = 0, r2lo = 1, r2hi = 0, r3lo = 1, r3hi = 0, r4lo = 1, r4hi = 0, r5lo = 1, r5hi = 0, i1lo = 1, i1hi = 0, i2lo = 1, i2hi = 0;
    int sm = RM - 1, si = 1;
    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      sm = sm + 1 == RM ? 0 : sm + 1; si ^= 1;
      int lo, hi;
      int qx = ROW_NULL, qo = ROW_NULL, qi = ROW_NULL, qd = ROW_NULL;
      if (s == 0) { lo = lo0; hi = hi0; qx = ROW_SEED; }
      else {
        lo = 1 << 30; hi = -(1 << 30);
        if (r2hi >= r2lo) { lo = imin(lo, r2lo); hi = imax(hi, r2hi); qx = sm >= xs ? sm - xs : sm - xs + RM; }
        if (r4hi >= r4lo) { lo = imin(lo, r4lo - 1); hi = imax(hi, r4hi + 1); qo = sm >= oes ? sm - oes : sm - oes + RM; }
        if (i1hi >= i1lo) { lo = imin(lo, i1lo - 1); hi = imax(hi, i1hi + 1); qi = ROW_I + (si ^ 1); qd = ROW_D + (si ^ 1); }
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
      }
      lo = U(lo); hi = U(hi);
      const int omlo = r5lo, omhi = r5hi, oilo = i2lo, oihi = i2hi;
      r5lo = r4lo; r5hi = r4hi; r4lo = r3lo; r4hi = r3hi; r3lo = r2lo; r3hi = r2hi; r2lo = r1lo; r2hi = r1hi; i2lo = i1lo; i2hi = i1hi;
      if (hi < lo) {       // unreachable score
        if (ww == 0) null_quarters(sm, omlo, omhi, 1, 0, ROW_I + si, ROW_D + si, oilo, oihi, 1, 0);
        r1lo = 1; r1hi = 0; i1lo = 1; i1hi = 0;
        if (tid == 0) rowtab[s] = -1;
        if (s > 2 * (oes + es * (pl + tl)) + 8) fail = true;
        lds_barrier();
        continue;
      }
      const int width = hi - lo + 1, padded = ((width + 63) >> 6) << 6;
      if (width + PAD > CAP || slab_top + (size_t)padded > ws.slab_bytes) { fail = true; break; }
      if (ww == 0) null_quarters(sm, omlo, imin(omhi, lo - 1), imax(omlo, hi + 1), omhi, ROW_I + si, ROW_D + si, oilo, imin(oihi, lo - 1), imax(oilo, hi + 1), oihi);
      uint8_t* btbase = slab + slab_top;         // provenance byte of diagonal k: btbase[k - lo]
      if (tid == 0) rowtab[s] = (int64_t)slab_top - lo;
      slab_top += (size_t)padded;
      W += 3ull * (uint64_t)width;
      int dmin = BIG, kfin = BIG;
      int qn = 0;
      auto finished = [&](int h, int k) {
        dmin = imin(dmin, left_to_align(h, k, pl, tl, ef, pef, tef));
        if (ef) { const int v = h - k; if ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)) kfin = imin(kfin, k); }
        else if (k == kend && h >= tl) kfin = k;
      };
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int e = 0; e < qn; ++e) {
              const int kk = lo + U((int)queue[e]);
              int h = U(rd(sm, kk));
              const int v = h - kk;
              h += otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
              wr(sm, kk, h);
              finished(h, kk);
            }
            qn = 0;
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int kk = 0, h = 0, v = 0;
            bool more = false;
            if (act) {
              kk = lo + (int)queue[q0 + lane];
              h = rd(sm, kk);
              v = h - kk;
              const int m = pk.match32(v, h, imin(pl - v, tl - h));
              v += m; h += m;
              more = (m == 32) && v < pl && h < tl;
              wr(sm, kk, h);
              if (!more) finished(h, kk);
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint32_t)(kk - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      // ---- sweep of this wave's chunks (branch-free per lane, as in the one-wave tier)
      const int bx = qx * (CAP * 2), bo = qo * (CAP * 2), bi = qi * (CAP * 2), bd = qd * (CAP * 2);
      const int bm = sm * (CAP * 2), bI = (ROW_I + si) * (CAP * 2), bD = (ROW_D + si) * (CAP * 2);
      for (int c = lo + 64 * ww; c <= hi; c += 64 * NW) {
        const int k = c + lane;
        const int a0 = (k & MASK) << 1, am = ((k - 1) & MASK) << 1, ap = ((k + 1) & MASK) << 1;
        const int io = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bo + am), dop = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bo + ap);
        const int ix = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bi + am), dx = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bd + ap);
        const int mm = *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bx + a0);
        const bool ext_i = ix >= io, ext_d = dx >= dop;
        const int insv = imax(ix, io) + 1, delv = imax(dx, dop), mis = mm + 1;
        const int mx = imax(imax(delv, mis), insv);
        const uint32_t org = mx == mis ? 0u : (mx == delv ? 1u : 2u);      // mismatch wins ties over deletion over insertion
        const uint32_t bits = org | (ext_i ? 4u : 0u) | (ext_d ? 8u : 0u);
        // per-lane conditions as lane masks on the scalar unit from here on (see mk_eq)
        const bool inr = k <= hi;
        const unsigned long long inrm = __builtin_amdgcn_ballot_w64(inr);
        const int v = mx - k;
        const int t1 = tl - mx, t2 = pl - v;
        const unsigned long long vm = inrm & mk_ule((uint32_t)mx, (uint32_t)tl) & mk_ule((uint32_t)v, (uint32_t)pl);      // valid cells
        auto probe = [&](int pv, int ph) -> int {      // equal leading bases of pattern[pv ..] and text[ph ..], at most 32
          const int wp = (pv >> 2) & ~3, wt = offT4 + ((ph >> 2) & ~3);
          const uint32_t sp = (uint32_t)(pv & 15) * 2u, st = (uint32_t)(ph & 15) * 2u;
          const volatile lds_u32* pp = (const volatile lds_u32*)(SQB + wp);
          const volatile lds_u32* pt = (const volatile lds_u32*)(SQB + wt);
          const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], q0 = pt[0], q1 = pt[1], q2 = pt[2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, sp) ^ __builtin_amdgcn_alignbit(q1, q0, st);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, sp) ^ __builtin_amdgcn_alignbit(q2, q1, st);
          uint32_t flo, fhi;
          asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"(xl));
          asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"(xh));
          const uint32_t a = flo < (fhi | 32u) ? flo : (fhi | 32u);
          return (int)((a < 64u ? a : 64u) >> 1);
        };
        const int pm = probe(v, mx);
        int m = imin(imin(pm, t1), t2);
        unsigned long long mq = vm & mk_eq(imin(imin(pm, t1 - 1), t2 - 1), 32);            // more: a full probe with more than 32 bases left of both sequences
        if (mq) {             // a second probe by the whole wave where a run outlives the first (as in the edit tier): the queue is for runs beyond 64 bases
          const int pm2 = probe(v + 32, mx + 32);
          const int m2 = imin(imin(pm2, t1 - 32), t2 - 32);
          m = sel(mq, m2 + 32, m);
          mq &= mk_eq(imin(imin(pm2, t1 - 33), t2 - 33), 32);
        }
        const int h2 = mx + m;
        if (inr) {            // lanes behind the range do not store: the window is usable up to CAP - 4 diagonals
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bm + a0) = (int16_t)sel(vm, h2, NUL);
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bI + a0) = (int16_t)sel(mk_sle(0, insv), insv, NUL);
          *(volatile __attribute__((address_space(3))) int16_t*)(ROWS + bD + a0) = (int16_t)sel(mk_sle(0, delv), delv, NUL);
        }
        btbase[(uint32_t)(k - lo)] = (uint8_t)bits;                      // every lane stores: the row is padded to whole chunks
        const unsigned long long hm = vm & ~mq;                          // cells that are final here
        const int lh = t1 - m, lv = t2 - m;
        int d;
        unsigned long long fm;
        if (!ef) { d = imax(lh, lv); fm = hm & mk_eq(k, kend) & mk_sle(lh, 0); }
        else { d = imin(imax(lh, lv - pef), imax(lv, lh - tef)); fm = hm & ((mk_sle(lh, 0) & mk_sle(lv, pef)) | (mk_sle(lv, 0) & mk_sle(lh, tef))); }
        if (fm) kfin = imin(kfin, sel(fm, k, BIG));
        dmin = imin(dmin, sel(hm, d, BIG));
        if (mq) {
          if (sel(mq, 1, 0)) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
            queue[qn + rank] = (uint32_t)(k - lo);
          }
          qn += __builtin_popcountll(mq);
          if (qn + 64 > QCAP) drain();
        }
      }
      if (qn) drain();
      // ---- the one barrier of the score: every wave's chunks are final, the block's minimum and end candidate follow from the NW pairs
      {
        const int wd = wave_min_i32(dmin), wk = wave_min_i32(kfin);
        if (lane == 0) { s_x[s & 1][ww][0] = wd; s_x[s & 1][ww][1] = wk; }
      }
      lds_barrier();
      int mind = BIG, kf = BIG;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) { mind = imin(mind, U(s_x[s & 1][w2][0])); kf = imin(kf, U(s_x[s & 1][w2][1])); }
      if (kf != BIG) { k_end = kf; s_end = s; break; }
      // the cut: computed by every wave (same inputs, same result), what it drops nulled by every wave
      int clo = lo, chi = hi;
      wfadaptive_cut32(H, steps_wait, mind, pl, tl, ef, pef, tef, clo, chi, lane, [&](int k) { const int x = rd(sm, k); return x < 0 ? OTG_NULL_OFF : x; });
      clo = U(clo); chi = U(chi);
      null_halves(sm, ROW_I + si, ROW_D + si, lo, clo - 1, chi + 1, hi);
      r1lo = clo; r1hi = chi;
      if (s == 0) { i1lo = 1; i1hi = 0; }
      else { i1lo = clo; i1hi = chi; }
    }
    __syncthreads();                      // provenance rows and the row table: written by all waves / thread 0, read back by wave 0
    if (ww != 0) continue;
    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }
    if (!backtrace_unpack<false>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g,
                                 (volatile lds_u32*)&s_rows[0][0], EqPacked{SQ, offT})) continue;
    if (cells) cells[ti] = W;
  }
}

int gcd3(int a, int b, int c)
{
  auto g2 = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
  return g2(g2(a, b), c);
}

} // namespace

// ---------------------------------------------------------------------------------------------------
// Launch chains.  Same contracts as otg_launch_edit_todo / otg_launch_affine_todo (which hand over to these when the context's heuristic is
// wfadaptive).  Counters: SLOT_COUNTERS words 108..129 (free of the exact chains' words).
int otg_launch_edit_adaptive_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                                  const uint32_t* d_n_todo, uint32_t n_tasks, int32_t* d_scores, uint64_t* d_cells,
                                  float* kernel_ms, uint64_t* launches)
{
  if (n_tasks == 0) return OTG_OK;
  if (ctx->pool[SLOT_COUNTERS].cap < OTG_COUNTER_WORDS * sizeof(uint32_t)) ctx->affine_visited = nullptr;      // (it points into this slot; the exact chain sets it up again)
  uint32_t* cnt = (uint32_t*)otg_slot(ctx, SLOT_COUNTERS, OTG_COUNTER_WORDS * sizeof(uint32_t));
  uint32_t* lists = (uint32_t*)otg_slot(ctx, SLOT_TODO, 5 * (size_t)n_tasks * sizeof(uint32_t));
  if (!cnt || !lists) return OTG_ERR_HIP;
  uint32_t* c = cnt + 108;                    // c[0..3] tickets of the first four tiers, c[4..7] lengths of their overflow lists, c[8..9] the last tier's
  uint32_t* c16 = cnt + 128;                  // ticket and overflow length of the packed 16384-diagonal tier
  HIP_TRY(ctx, hipMemsetAsync(c, 0, 10 * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c16, 0, 2 * sizeof(uint32_t), ctx->stream));
  const Heur H{ctx->heur_min_wf_len, ctx->heur_max_dist, ctx->heur_steps < 1 ? 1 : ctx->heur_steps};
  const uint32_t ncu = (uint32_t)ctx->n_cu;
  // test switch, bit t = tier t runs: 1 packed 1024, 2 packed 4096, 4 bytes 2048, 8 bytes 16384, 16 packed 16384 (eight waves per pair, behind the packed
  // 4096 one: what outgrew that went through the byte-probe tiers before); the HBM tier always runs
  static const int only = getenv("OTG_ADAPTIVE_EDIT_TIERS") ? atoi(getenv("OTG_ADAPTIVE_EDIT_TIERS")) : 31;
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const uint32_t* in = d_todo; const uint32_t* in_n = d_n_todo; uint32_t in_imm = n_tasks;
  uint32_t* l0 = lists; uint32_t* l1 = lists + n_tasks; uint32_t* l2 = lists + 2 * (size_t)n_tasks;
  uint32_t* l3 = lists + 3 * (size_t)n_tasks;
  uint32_t* l4 = lists + 4 * (size_t)n_tasks;
  // words of LDS per wave for the packed pair: two reads of the batch's longest length (at most 32 KB: 2 x 32766 bases is what 16-bit offsets hold anyway)
  const int seqw = (int)std::min<size_t>(2 * (((size_t)ctx->max_seq_len + 15) / 16 + 3) + 2, 8192);
  // SLOT_WF_WS serves the first tier (a global row of 16-bit offsets per wave, for pairs that START wider than the window) and the last one (int32
  // rows): asked for once, for the larger of the two, so that the slot is not re-allocated between two kernels of the chain
  constexpr int WPB0 = 4, WPBL = 4;
  const size_t dyn0 = (size_t)WPB0 * seqw * 4;
  const uint32_t per_cu0 = std::max<uint32_t>(1, std::min<uint32_t>(8, (uint32_t)((160 * 1024) / (WPB0 * 3072 + dyn0))));
  const uint32_t grid0 = std::min<uint32_t>(ncu * per_cu0, (n_tasks + WPB0 - 1) / WPB0);
  const int gcap0 = (int)std::min<size_t>((2 * (size_t)ctx->max_seq_len + 128) & ~(size_t)1, 65664);      // (the tier takes pairs of two sequences below 32 767 bases: pl + tl + 72 never needs more)
  const int gcapL = (int)(2 * (size_t)ctx->max_seq_len + 4);
  static const bool wide_start = getenv("OTG_ADAPTIVE_NO_WIDE_START") == nullptr;
  const size_t need0 = (only & 1) && wide_start ? (size_t)grid0 * WPB0 * (size_t)gcap0 * sizeof(int16_t) : 0;
  const size_t needL = (size_t)ncu * WPBL * (size_t)gcapL * sizeof(int32_t);
  uint8_t* wsp = (uint8_t*)otg_slot(ctx, SLOT_WF_WS, std::max(need0, needL));
  if (!wsp) return OTG_ERR_HIP;
  if (only & 1) {       // fast tier: window of 1024 diagonals (3 KB of LDS per wave) + the packed pair
    hipLaunchKernelGGL((wfa_edit_adaptive_lds_kernel<1024, 512, WPB0>), dim3(grid0), dim3(WPB0 * 64), dyn0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c + 0, c + 4, l0, H, seqw, need0 ? (int16_t*)wsp : (int16_t*)nullptr, gcap0);
    in = l0; in_n = c + 4; in_imm = 0;
  }
  if (only & 2) {       // 4096 diagonals, one wave per pair: 10 KB per wave + the packed pair.  (Measured and not kept: this window on the multi-wave kernel below —
                        // the reassignment pass sends it 437 000 pairs on the 1-10 kb shard, work enough for one wave each: 4 waves per pair 388 ms, 2 waves 363, one 322)
    constexpr int WPB = 2;
    const size_t dyn = (size_t)WPB * seqw * 4;
    const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(8, (uint32_t)((160 * 1024) / (WPB * 10240 + dyn))));
    const uint32_t grid = std::min<uint32_t>(ncu * per_cu, (n_tasks + WPB - 1) / WPB);
    hipLaunchKernelGGL((wfa_edit_adaptive_lds_kernel<4096, 1024, WPB>), dim3(grid), dim3(WPB * 64), dyn, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c + 1, c + 5, l1, H, seqw, (int16_t*)nullptr, 0);
    in = l1; in_n = c + 5; in_imm = 0;
  }
  if (only & 16) {                       // 16384 diagonals, eight waves per pair: two rows of 32 KB, two blocks per CU
    constexpr int NW = 8;
    const size_t dyn = (size_t)seqw * 4;
    const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(2, (uint32_t)((160 * 1024) / (2 * 16384 * 2 + NW * 256 * 2 + 256 + dyn))));
    const uint32_t grid = std::min<uint32_t>(ncu * per_cu, n_tasks);
    hipLaunchKernelGGL((wfa_edit_adaptive_mw_kernel<16384, 256, NW>), dim3(grid), dim3(NW * 64), dyn, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c16 + 0, c16 + 1, l4, H, seqw);
    in = l4; in_n = c16 + 1; in_imm = 0;
  }
  if (only & 4) {       // byte probes (pairs with bytes outside ACGT, pairs too long to pack), 2048 diagonals
    constexpr int WPB = 2;
    const uint32_t grid = std::min<uint32_t>(ncu * 6, (n_tasks + WPB - 1) / WPB);
    hipLaunchKernelGGL((wfa_edit_adaptive_kernel<2048, 2048, WPB>), dim3(grid), dim3(WPB * 64), 0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c + 2, c + 6, l2, H, (int32_t*)nullptr, 0);
    in = l2; in_n = c + 6; in_imm = 0;
  }
  if (only & 8) {       // 16384 diagonals: 40 KB per wave
    constexpr int WPB = 1;
    const uint32_t grid = std::min<uint32_t>(ncu * 3, n_tasks);
    hipLaunchKernelGGL((wfa_edit_adaptive_kernel<16384, 2048, WPB>), dim3(grid), dim3(WPB * 64), 0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c + 3, c + 7, l3, H, (int32_t*)nullptr, 0);
    in = l3; in_n = c + 7; in_imm = 0;
  }
  {                     // int32 wavefront in HBM, sized for the longest pair of the batch
    hipLaunchKernelGGL((wfa_edit_adaptive_kernel<0, 2048, WPBL>), dim3(ncu), dim3(WPBL * 64), 0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       d_scores, d_cells, c + 8, c + 9, (uint32_t*)nullptr, H, (int32_t*)wsp, gcapL);
  }
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  if (getenv("OTG_DEBUG")) {
    hipError_t er = hipStreamSynchronize(ctx->stream);
    uint32_t h[10], h16[2];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h16, c16, sizeof(h16), hipMemcpyDeviceToHost);
    fprintf(stderr, "[otg] edit, wfadaptive(%d,%d,%d): %s; the packed 1024-diagonal tier passes on %u pairs, the packed 4096 one %u, the packed 16384 one %u, the byte-probe 2048 one %u, the 16384 one %u\n",
            H.min_wf_len, H.max_dist, H.steps, hipGetErrorString(er), h[4], h[5], h16[1], h[6], h[7]);
  }
  if (kernel_ms) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *kernel_ms += ms;
    if (launches) *launches += 1;
  }
  return OTG_OK;
}

int otg_launch_affine_adaptive_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                                    const uint32_t* d_n_todo, uint32_t n_tasks, int x, int o, int e, int32_t* d_scores,
                                    const uint64_t* d_cig_off, uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells,
                                    float* kernel_ms, uint64_t* launches)
{
  if (n_tasks == 0) return OTG_OK;
  if (x <= 0 || e <= 0 || o < 0) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties must satisfy x>0, o>=0, e>0");
  const int g = gcd3(x, o + e, e);
  const int xs = x / g, oes = (o + e) / g, es = e / g;
  if (std::max(xs, oes) + 1 > 64 || es + 1 > 64) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties too large after gcd reduction");
  if (ctx->pool[SLOT_COUNTERS].cap < OTG_COUNTER_WORDS * sizeof(uint32_t)) ctx->affine_visited = nullptr;
  uint32_t* cnt = (uint32_t*)otg_slot(ctx, SLOT_COUNTERS, OTG_COUNTER_WORDS * sizeof(uint32_t));
  uint32_t* lists = (uint32_t*)otg_slot(ctx, SLOT_TODO, 2 * (size_t)n_tasks * sizeof(uint32_t));
  if (!cnt || !lists) return OTG_ERR_HIP;
  uint32_t* c = cnt + 120;                    // c[0..2] tickets, c[3..4] lengths of the overflow lists, c[6..7] the byte tier's pair, c[-2..-1] the 4096 window's
  HIP_TRY(ctx, hipMemsetAsync(c - 2, 0, 10 * sizeof(uint32_t), ctx->stream));
  const Heur H{ctx->heur_min_wf_len, ctx->heur_max_dist, ctx->heur_steps < 1 ? 1 : ctx->heur_steps};
  const uint32_t ncu = (uint32_t)ctx->n_cu;
  const size_t maxlen = ((size_t)ctx->max_seq_len + 4095) & ~(size_t)4095;
  // bit t = tier t runs: 1 packed 256, 2 packed 1024, 4 bytes 1024, 8 packed 4096; 16 = the 1024 window as ONE wave per alignment (the tier as first
  // built) instead of four; the int32 tier always runs
  static const int only = getenv("OTG_ADAPTIVE_AFFINE_TIERS") ? atoi(getenv("OTG_ADAPTIVE_AFFINE_TIERS")) : 15;

  AffWs ws;
  ws.capa = (int)(2 * maxlen + 16) & ~1;
  ws.rm = std::max(xs, oes) + 1;
  ws.ri = es + 1;
  ws.nrows = (int)(2 * (size_t)oes + (size_t)es * 2 * maxlen + 16);
  ws.rev_cap = 4 * maxlen + 64;
  ws.dbg = 0; ws.visited = nullptr;
  // LDS tiers: no rings in HBM, only row table + reversed op list + provenance slab
  auto lds_ws = [&](size_t slab) {
    AffWs w = ws;
    w.off_rowtab = 0;
    w.off_rev = ((size_t)w.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
    w.off_slab = (w.off_rev + w.rev_cap + 255) & ~(size_t)255;
    w.slab_bytes = slab & ~(size_t)255;
    w.stride = w.off_slab + w.slab_bytes;
    return w;
  };
  constexpr int WPB0 = 4, WPB1 = 1, WPB2 = 4;
  const int seqw = (int)std::min<size_t>(2 * (((size_t)ctx->max_seq_len + 15) / 16 + 3) + 2, 8192);      // LDS words per wave for the packed pair
  // provenance: a row per score, as wide as the wavefront.  Mean width under the cut ~100 diagonals, scores ~0.4 per base
  // (gcd units): 40 x maxlen bytes hold the typical alignment of the first tier four times over.
  AffWs w0 = lds_ws(std::max<size_t>((size_t)160 * maxlen, (size_t)1 << 19));
  // (1024 window: 640 x maxlen = 5 MB for 5 kb reads — a wavefront of 400 diagonals over 2 000 scores takes 0.8 MB — so that LDS, not the budget, bounds the
  // alignments in flight: 1 280 instead of 640 with the 2 560 x maxlen of the one-wave tier, gap-affine stage 211 -> 186 ms; what outgrows it has the 4096 tier behind it)
  AffWs w1 = lds_ws(std::max<size_t>((size_t)640 * maxlen, (size_t)1 << 21));
  AffWs w3 = lds_ws(std::max<size_t>((size_t)8192 * maxlen, (size_t)1 << 25));       // the 4096-diagonal window: a block per CU (90 KB of rows)
  // blocks per CU by LDS: rows 9 x CAP x 2 B + queue + range tables + the packed pair
  const uint32_t pc0 = std::max<uint32_t>(1, std::min<uint32_t>(6, (uint32_t)((160 * 1024) / (WPB0 * (11 * 256 * 2 + 1024 + 64 + (size_t)seqw * 4)))));
  const uint32_t pc1 = std::max<uint32_t>(1, std::min<uint32_t>(8, (uint32_t)((160 * 1024) / (11 * 1024 * 2 + 4096 + 128 + (size_t)seqw * 4))));
  constexpr int NW1 = 4, NW3 = 8, QMW = 256;
  uint32_t grid0 = std::min<uint32_t>(ncu * pc0, (n_tasks + WPB0 - 1) / WPB0), grid1 = std::min<uint32_t>(ncu * pc1, n_tasks), grid2 = 8;
  uint32_t grid3 = std::min<uint32_t>(ncu, n_tasks);
  AffWs w2 = ws;
  {
    const size_t ring_bytes = (size_t)(ws.rm + 2 * ws.ri) * ws.capa * sizeof(int32_t);
    w2.off_rowtab = (ring_bytes + 255) & ~(size_t)255;
    w2.off_rev = (w2.off_rowtab + (size_t)ws.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
    w2.off_slab = (w2.off_rev + ws.rev_cap + 255) & ~(size_t)255;
  }
  {
    std::lock_guard<std::mutex> alloc_lock(otg_device_mutex(ctx->device));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::min<size_t>((size_t)(total_b * 0.15), (size_t)((free_b + ctx->pool[SLOT_WF_WS].cap) * 0.8));
    while (grid1 > 8 && w1.stride * grid1 * WPB1 > budget / 2) grid1 /= 2;
    while (grid0 > 8 && w0.stride * grid0 * WPB0 > budget / 2) grid0 /= 2;
    while (grid3 > 8 && w3.stride * grid3 > budget / 2) grid3 /= 2;
    // the generic tier: the worst case of the longest pair of the batch (every diagonal at every score), as far as the budget goes
    size_t slab2 = std::min<size_t>((size_t)2 * maxlen * (size_t)ws.nrows, budget / (grid2 * WPB2));
    if (slab2 > w2.off_slab + 256) slab2 -= w2.off_slab + 256;
    w2.slab_bytes = slab2 & ~(size_t)255; w2.stride = w2.off_slab + w2.slab_bytes;
    const size_t need = std::max(std::max(std::max(w0.stride * grid0 * WPB0, w1.stride * grid1 * WPB1), w2.stride * (size_t)grid2 * WPB2), w3.stride * grid3);
    uint8_t* wsp = (uint8_t*)otg_slot(ctx, SLOT_WF_WS, need);
    if (!wsp) return OTG_ERR_HIP;
    w0.base = w1.base = w2.base = w3.base = wsp;
  }
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const uint32_t* in = d_todo; const uint32_t* in_n = d_n_todo; uint32_t in_imm = n_tasks;
  uint32_t* l0 = lists; uint32_t* l1 = lists + n_tasks;
  const bool std_pen = xs == 2 && oes == 4 && es == 1;
  if ((only & 1) && std_pen) {      // fast tier, 256 diagonals
    hipLaunchKernelGGL((wfa_affine_adaptive_lds_kernel<256, 256, WPB0, true>), dim3(grid0), dim3(WPB0 * 64), (size_t)WPB0 * seqw * 4, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c + 0, c + 3, l0, w0, H, seqw);
    in = l0; in_n = c + 3; in_imm = 0;
  }
  if ((only & 2) && std_pen) {      // 1024 diagonals, four waves per alignment (bit 16: one)
    if (only & 16)
      hipLaunchKernelGGL((wfa_affine_adaptive_lds_kernel<1024, 1024, WPB1, false>), dim3(grid1), dim3(WPB1 * 64), (size_t)WPB1 * seqw * 4, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                         g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c + 1, c + 4, l1, w1, H, seqw);
    else
      hipLaunchKernelGGL((wfa_affine_adaptive_mw_kernel<1024, QMW, NW1>), dim3(grid1), dim3(NW1 * 64), (size_t)seqw * 4, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                         g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c + 1, c + 4, l1, w1, H, seqw);
    in = l1; in_n = c + 4; in_imm = 0;
  }
  if ((only & 8) && std_pen) {      // 4096 diagonals, eight waves per alignment, one block per CU
    uint32_t* lout = in == l0 ? l1 : l0;
    hipLaunchKernelGGL((wfa_affine_adaptive_mw_kernel<4096, QMW, NW3>), dim3(grid3), dim3(NW3 * 64), (size_t)seqw * 4, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c - 2, c - 1, lout, w3, H, seqw);
    in = lout; in_n = c - 1; in_imm = 0;
  }
  if ((only & 4) && std_pen) {      // byte probes, 1024 diagonals: pairs with bytes outside ACGT or too long to pack
    hipLaunchKernelGGL((wfa_affine_adaptive_kernel<1024, 1024, WPB1, 8>), dim3(grid1), dim3(WPB1 * 64), 0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                       xs, oes, es, g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c + 6, c + 7, in == l0 ? l1 : l0, w1, H, 0);
    in = in == l0 ? l1 : l0; in_n = c + 7; in_imm = 0;
  }
  hipLaunchKernelGGL((wfa_affine_adaptive_kernel<0, 2048, WPB2, 64>), dim3(grid2), dim3(WPB2 * 64), 0, ctx->stream, d_arena, d_tasks, in, in_n, in_imm,
                     xs, oes, es, g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, c + 2, c + 5, (uint32_t*)nullptr, w2, H, 0);
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  if (getenv("OTG_DEBUG")) {
    hipError_t er = hipStreamSynchronize(ctx->stream);
    uint32_t h[10];
    (void)hipMemcpy(h, c - 2, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[otg] affine, wfadaptive(%d,%d,%d): %s; the 256-diagonal window passes on %u alignments, the 1024 one %u, the 4096 one %u, the byte probes %u; %u / %u / %u alignments in flight, %.2f / %.2f / %.2f MB each\n",
            H.min_wf_len, H.max_dist, H.steps, hipGetErrorString(er), h[5], h[6], h[1], h[9], grid0 * WPB0, grid1 * WPB1, grid3, (double)w0.stride / 1e6, (double)w1.stride / 1e6, (double)w3.stride / 1e6);
  }
  if (kernel_ms) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *kernel_ms += ms;
    if (launches) *launches += 1;
  }
  return OTG_OK;
}
