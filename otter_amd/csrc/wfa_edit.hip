// wfa_edit.hip — batched unit-cost wavefront aligner, score only (gfx950).
//
// Replaces wfa::WFAlignerEdit(Score, MemoryMed)::alignEnd2End / alignEndsFree + getAlignmentScore()
// (reference: src/assemble.cpp:49; call sites src/analignments.cpp:70-71,88-97).
//
// Mapping: ONE wave64 per alignment.  The wavefront of furthest-reaching offsets lives in LDS
// (one int32 per diagonal, updated IN PLACE: chunks of 64 diagonals are swept in ascending order, the
// left neighbour of lane 0 is carried in an SGPR-uniform register, the right neighbour of lane 63 is
// still the old value).  Lanes = diagonals; the extend step compares 8 sequence bytes per iteration
// (unaligned 64-bit loads, xor + ctz) and the wave leaves the extend loop on a ballot.
// Work distribution: persistent waves pull tickets from one device counter (tasks differ ~100x in cost).
// Three capacity tiers share this code: LDS 2048 diagonals/wave (20 waves/CU), LDS 16384/wave
// (2 waves/CU), and a global-memory wavefront for anything wider; a tier that runs out of diagonals
// appends the task to the next tier's todo list on the device (no host round trip).
//
// Recurrence: M[s][k] = max(M[s-1][k-1]+1, M[s-1][k]+1, M[s-1][k+1]), nulled
// when h>tlen or v>plen; k = h - v, offset = h.  End2end stops when M[s][tlen-plen] == tlen; ends-free
// when any diagonal reaches a permitted boundary (score only: which diagonal does not matter).
#include "otg_common.hpp"
#include <algorithm>
#include <cstdlib>

namespace {

using lds_i32 = __attribute__((address_space(3))) int32_t;

__device__ __forceinline__ uint64_t load8(const uint8_t* p)
{
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

template <int CAP, int WPB, bool GLOBAL_WF>
__global__ __launch_bounds__(WPB * 64) void wfa_edit_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    int32_t* gws, int gcap)
{
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int cap = GLOBAL_WF ? gcap : CAP;
  // volatile keeps program order of the in-place, cross-lane wavefront updates (a wave's DS / VMEM ops
  // execute in issue order); the LDS pointer keeps its address space so these stay ds_read/ds_write.
  volatile int32_t* gwf = GLOBAL_WF ? (volatile int32_t*)(gws + (size_t)(blockIdx.x * WPB + wib) * (size_t)gcap) : nullptr;
  volatile lds_i32* lwf = (volatile lds_i32*)smem + wib * CAP;
  auto wf_rd = [&](int j) -> int { if constexpr (GLOBAL_WF) return gwf[j]; else return lwf[j]; };
  auto wf_wr = [&](int j, int v) { if constexpr (GLOBAL_WF) gwf[j] = v; else lwf[j] = v; };
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
    const int pef = t.pattern_end_free, tef = t.text_end_free;
    int lo = ef ? -t.pattern_begin_free : 0, hi = ef ? t.text_begin_free : 0;
    if (lo < -pl) lo = -pl;
    if (hi > tl) hi = tl;
    const int kend = tl - pl;
    const int kbase = lo - ((cap - (hi - lo + 1)) >> 1);
    int lo_prev = lo, hi_prev = hi;
    int s = 0;
    uint64_t W = 0;
    bool done = false, overflow = false;
    if (hi - lo + 1 > cap) overflow = true;

    while (!overflow) {
      W += (uint64_t)(hi - lo + 1);
      int carry = OTG_NULL_OFF;
      bool any_done = false;
      for (int c = lo; c <= hi; c += 64) {
        const int k = c + lane;
        const int j = k - kbase;
        const bool in = k <= hi;
        int mx;
        if (s == 0) {
          mx = k > 0 ? k : 0;
        } else {
          int o = (k >= lo_prev && k <= hi_prev) ? wf_rd(j) : OTG_NULL_OFF;
          int r = (k + 1 >= lo_prev && k + 1 <= hi_prev) ? wf_rd(j + 1) : OTG_NULL_OFF;
          int l = __shfl_up(o, 1);
          if (lane == 0) l = carry;
          carry = __shfl(o, 63);
          int a = l + 1, b = o + 1;
          mx = a > b ? a : b;
          mx = r > mx ? r : mx;
        }
        int h = mx, v = mx - k;
        bool valid = in && mx >= 0 && h <= tl && v <= pl;
        bool act = valid;
        for (;;) {
          const bool go = act && v < pl && h < tl;
          if (!__any(go)) break;
          if (go) {
            const uint64_t x = load8(P + v) ^ load8(T + h);
            int m = x ? (__builtin_ctzll(x) >> 3) : 8;
            int rem = pl - v < tl - h ? pl - v : tl - h;
            m = m < rem ? m : rem;
            v += m; h += m;
            act = (m == 8);
          } else act = false;
        }
        if (in) wf_wr(j, valid ? h : OTG_NULL_OFF);
        bool fin;
        if (ef) fin = valid && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
        else fin = valid && k == kend && h >= tl;
        any_done |= __any(fin) != 0;
      }
      if (any_done) { done = true; break; }
      lo_prev = lo; hi_prev = hi;
      lo = lo - 1 < -pl ? -pl : lo - 1;
      hi = hi + 1 > tl ? tl : hi + 1;
      ++s;
      if (lo - kbase < 0 || hi - kbase + 1 >= cap) overflow = true;
      if (s > pl + tl + 2) break; /* cannot happen: edit distance <= max(pl,tl) */
    }
    // wave-uniform tail: every lane stores the same value to the same address (one write after coalescing)
    if (done) {
      scores[ti] = s;
      if (cells) cells[ti] = (t._pad & 1) ? otg_edit_wfa_cells(s, pl, tl, ef ? (int)t.pattern_end_free : 0, ef ? (int)t.text_end_free : 0) : W;   // mirrored task: cells of the un-reversed form
    } else if (overflow && overflow_list) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;
    } else {
      scores[ti] = -1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// v2 kernel (tiers 1 and 2): 16-bit offsets in LDS (requires sequence lengths < 65535), and the extend step
// split in two so that no lane waits for the slowest diagonal of its chunk:
//   sweep : per 64-diagonal chunk compute the new offsets (left neighbour by DPP wave_shr, right neighbour
//           from LDS), probe 8 bytes, store; diagonals whose probe matched fully are appended to a
//           wave-compacted queue in LDS (ballot + mbcnt);
//   drain : the queue is re-read 64 entries at a time, 16 bytes are compared per iteration, finished
//           diagonals drop out, the rest are re-compacted in place.
// Work per wavefront is then ~Σ(iterations per diagonal) instead of Σ_chunks max(iterations in chunk).
using lds_u16 = __attribute__((address_space(3))) uint16_t;

__device__ __forceinline__ int dpp_wave_shr1(int x)
{
  // lane i receives lane i-1 (lane 0 keeps its own value); GFX9 DPP wave_shr:1
  return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}

struct u128 { uint64_t lo, hi; };
__device__ __forceinline__ u128 load16(const uint8_t* p)
{
  u128 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// bit-parallel tier whose band holds an estimated distance `need` (same threshold function the tier itself applies)
__device__ __forceinline__ int otg_route_tier(const otg_align_task& t, int need, uint32_t tier_mask)
{
  const int pl = (int)t.pattern_len, tl = (int)t.text_len;
  const bool ef = t.endsfree != 0;
  int tier = OTG_MYERS_TIERS;           // (none fits: the wide wavefront tier)
  const bool swap = !ef && pl < tl;                       // the bit-parallel kernel puts the longer sequence in the rows
  const int dd = swap ? tl - pl : pl - tl;
  const int fb = ef ? (int)t.pattern_begin_free : 0, fe2 = ef ? (int)t.pattern_end_free : 0;
  if (dd >= 0) {
    const int rows[OTG_MYERS_TIERS] = {456, 904, 1352, 1936, 2896, 4000, 8128, 16192};    // (GL-1)*64*BPL + GL of the tiers (myers_edit.hip)
#pragma unroll
    for (int q = OTG_MYERS_TIERS - 1; q >= 0; --q) if (((tier_mask >> q) & 1u) && need <= otg_myers_threshold(rows[q], dd, fb < dd ? fb : dd, fe2 < dd ? fe2 : dd)) tier = q;
  }
  return tier;
}

template <int CAP, int WPB>
__global__ __launch_bounds__(WPB * 64) void wfa_edit_kernel_v2(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    float cap_coeff, uint32_t* __restrict__ route_cnt, uint32_t* __restrict__ route_lists, uint32_t route_stride, float route_margin, uint32_t tier_mask)
{
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  volatile lds_u16* wf = (volatile lds_u16*)smem + (size_t)wib * 2 * CAP;
  volatile lds_u16* queue = wf + CAP;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  constexpr int NUL = 0xFFFF;

  for (;;) {
    const uint32_t tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = t.pattern_end_free, tef = t.text_end_free;
    int lo = ef ? -t.pattern_begin_free : 0, hi = ef ? t.text_begin_free : 0;
    if (lo < -pl) lo = -pl;
    if (hi > tl) hi = tl;
    const int kend = tl - pl;
    const int kbase = lo - ((CAP - (hi - lo + 1)) >> 1);
    int lo_prev = lo, hi_prev = hi;
    int s = 0;
    uint64_t W = 0;
    bool done = false;
    bool overflow = (hi - lo + 1 > CAP) || pl >= 65535 || tl >= 65535;
    // score cap: beyond ~cap_coeff*sqrt(len) the O(n) bit-parallel tier is cheaper than O(s^2) wavefronts
    int cap = 0x7fffffff;
    if (cap_coeff > 0.0f) {
      cap = (int)(cap_coeff * sqrtf((float)(pl > tl ? pl : tl)));
      if (cap < 48) cap = 48;
      const int dlen = pl > tl ? pl - tl : tl - pl;
      if (!ef && dlen > cap) {
        // cannot finish here; with routing, a short run still measures the local divergence for the tier choice
        if (route_cnt && !overflow) cap = 24; else overflow = true;
      }
    }

    while (!overflow) {
      W += (uint64_t)(hi - lo + 1);
      int carry = OTG_NULL_OFF;
      bool any_done = false;
      int qn = 0;
      // ---- sweep
      for (int c = lo; c <= hi; c += 64) {
        const int k = c + lane;
        const int j = k - kbase;
        const bool in = k <= hi;
        int mx;
        if (s == 0) {
          mx = k > 0 ? k : 0;
        } else {
          int o = OTG_NULL_OFF, r = OTG_NULL_OFF;
          if (k >= lo_prev && k <= hi_prev) { const int x = wf[j]; o = x == NUL ? OTG_NULL_OFF : x; }
          if (k + 1 >= lo_prev && k + 1 <= hi_prev) { const int x = wf[j + 1]; r = x == NUL ? OTG_NULL_OFF : x; }
          int l = dpp_wave_shr1(o);
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
          const uint64_t x = load8(P + v) ^ load8(T + h);
          int m = x ? (__builtin_ctzll(x) >> 3) : 8;
          const int rem = pl - v < tl - h ? pl - v : tl - h;
          m = m < rem ? m : rem;
          v += m; h += m;
          more = (m == 8) && v < pl && h < tl;
        }
        if (in) wf[j] = (uint16_t)(valid ? h : NUL);
        const unsigned long long mm = __ballot(more);
        if (more) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
          queue[qn + rank] = (uint16_t)j;
        }
        qn += __builtin_popcountll(mm);
        bool fin;
        if (ef) fin = valid && !more && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
        else fin = false;
        any_done |= __ballot(fin) != 0ull;
      }
      // ---- drain: 16 bytes in the first pass, then 64 bytes per pass; when <= 4 diagonals are left the whole
      // wave extends them one at a time, 512 bytes per iteration (HiFi / TR reads: exact runs of hundreds of bases)
      int pass = 0;
      while (qn > 0) {
        if (qn <= 4 && pass > 0) {
          for (int q = 0; q < qn; ++q) {
            const int j = queue[q];
            int h = wf[j];
            int v = h - (j + kbase);
            const int rem = pl - v < tl - h ? pl - v : tl - h;
            const int m = otg_wave_match(P, T, v, h, rem, lane);
            h += m; v += m;
            wf[j] = (uint16_t)h;
            if (ef && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef))) any_done = true;
          }
          qn = 0;
          break;
        }
        int wq = 0;
        for (int q0 = 0; q0 < qn; q0 += 64) {
          const bool act = q0 + lane < qn;
          int j = 0, h = 0, v = 0;
          bool more = false;
          if (act) {
            j = queue[q0 + lane];
            h = wf[j];
            v = h - (j + kbase);
            const int rem = pl - v < tl - h ? pl - v : tl - h;
            int m, full;
            if (pass == 0) {
              const u128 a = load16(P + v), b = load16(T + h);
              const uint64_t xl = a.lo ^ b.lo, xh = a.hi ^ b.hi;
              m = xl ? (__builtin_ctzll(xl) >> 3) : (xh ? 8 + (__builtin_ctzll(xh) >> 3) : 16);
              m = m < rem ? m : rem; full = 16;
            } else { m = otg_match64(P, T, v, h, rem); full = 64; }
            v += m; h += m;
            more = (m == full) && v < pl && h < tl;
            wf[j] = (uint16_t)h;
          }
          const unsigned long long mm = __ballot(more);
          if (more) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            queue[wq + rank] = (uint16_t)j;
          }
          wq += __builtin_popcountll(mm);
          if (ef) {
            const bool fin = act && !more && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
            any_done |= __ballot(fin) != 0ull;
          }
        }
        qn = wq; ++pass;
      }
      if (!ef && kend >= lo && kend <= hi) {
        const int x = wf[kend - kbase];
        any_done = (x != NUL && x >= tl);
      }
      if (any_done) { done = true; break; }
      lo_prev = lo; hi_prev = hi;
      lo = lo - 1 < -pl ? -pl : lo - 1;
      hi = hi + 1 > tl ? tl : hi + 1;
      ++s;
      if (lo - kbase < 0 || hi - kbase + 1 >= CAP || s > cap) overflow = true;
      if (route_cnt && s == 16 && cap > 32) {
        // early exit: if 16 edits covered so little that the projected distance is far beyond the cap, this pair is
        // not a near-identical one; hand it to the bit-parallel tiers now (the projection also routes it)
        int am = 0;
        for (int c = lo_prev; c <= hi_prev; c += 64) {
          const int k = c + lane;
          if (k <= hi_prev) { const int x = wf[k - kbase]; if (x != NUL) { const int a = 2 * x - k; am = a > am ? a : am; } }
        }
        am = otg_wave_max_i32(am);
        if ((long long)s * (pl + tl) > 3ll * cap * (am > 0 ? am : 1)) overflow = true;
      }
      if (s > pl + tl + 2) break;
    }
    if (done) {
      scores[ti] = s;
      if (cells) cells[ti] = (t._pad & 1) ? otg_edit_wfa_cells(s, pl, tl, ef ? (int)t.pattern_end_free : 0, ef ? (int)t.text_end_free : 0) : W;   // mirrored task: cells of the un-reversed form
    } else if (overflow && route_cnt) {
      // Route to the bit-parallel tier whose band fits the ESTIMATED distance: score so far scaled by the share of
      // both sequences the furthest wavefront point has covered, plus the length difference.  Only a scheduling
      // hint: a tier that turns out too narrow passes the pair on, every tier is exact.
      int amax = 0;
      if (s > 0) {
        for (int c = lo_prev; c <= hi_prev; c += 64) {
          const int k = c + lane;
          if (k <= hi_prev) { const int x = wf[k - kbase]; if (x != NUL) { const int a = 2 * x - k; amax = a > amax ? a : amax; } }
        }
        amax = otg_wave_max_i32(amax);
      }
      const int dlen = pl > tl ? pl - tl : tl - pl;
      const int minlen = pl < tl ? pl : tl;
      float est = (s > 0 && amax > 0) ? (float)s * (float)(pl + tl) / (float)amax : 0.25f * (float)minlen;
      if (est > (float)(pl + tl)) est = (float)(pl + tl);
      // no safety margin: trying a tier that fails costs half of going one tier up straight away, so the median
      // estimate is the cheapest choice (OTG_EDIT_ROUTE_MARGIN overrides, in percent)
      const int need = (int)(route_margin * (est + (ef ? 0.0f : (float)dlen)));
      // threshold of tier t for this pair's length difference and free ends (same function the tier itself uses)
      const int tier = otg_route_tier(t, need, tier_mask);
      const uint32_t q = otg_wave_atomic_add(route_cnt + tier, 1u);
      route_lists[(size_t)tier * route_stride + q] = ti;
    } else if (overflow && overflow_list) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;
    } else {
      scores[ti] = -1;
    }
  }
}

// ---- tier routing by sampling (one pair per LANE).  The bit-parallel tiers only need to know roughly how far apart two reads
// are; measuring that with wavefronts costs a chain of dependent sequence probes per score.  Instead: edit distance of the first 64
// bases of the pattern against a prefix of the text, and of the last 64 against a suffix (one 64-bit block of Myers' recurrence per
// lane, registers only, text end free), scaled to the whole pair.  A scheduling hint exactly like the wavefront estimate it replaces:
// every tier is exact and passes a pair on when its band turns out too narrow.  Pairs the estimate cannot judge (free ends, short
// sequences) and pairs that look near-identical (the wavefront pass finishes those by itself) go to the wavefront pass.
__device__ __forceinline__ uint32_t otg_ld_u32_unaligned(const uint8_t* p)
{
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3u);
  const uint32_t lo = q[0];
  if (sh == 0) return lo;
  return __builtin_amdgcn_alignbyte(q[1], lo, sh);
}

template <bool REV>
__device__ __forceinline__ void otg_sample64(const uint8_t* P, int pl, const uint8_t* T, int tl, int ncols, int* dmin, int* jmin)
{
  // rows: P[0..63] (REV: P[pl-1-i]); columns: T[0..ncols) (REV: T[tl-1-j]); D[0][j] = j, D[i][0] = i; result min_j D[64][j]
  uint64_t eq0 = 0, eq1 = 0, eq2 = 0, eq3 = 0;
#pragma unroll 4
  for (int w = 0; w < 16; ++w) {
    uint32_t v = otg_ld_u32_unaligned(REV ? P + pl - 4 - 4 * w : P + 4 * w);
    if (REV) v = __builtin_bswap32(v);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t c = ((v >> (8 * b)) >> 1) & 3u;
      const uint64_t m = 1ull << (4 * w + b);
      eq0 |= c == 0 ? m : 0ull; eq1 |= c == 1 ? m : 0ull; eq2 |= c == 2 ? m : 0ull; eq3 |= c == 3 ? m : 0ull;
    }
  }
  uint64_t Pv = ~0ull, Mv = 0ull;
  int score = 64, best = 64, bj = 0;
  for (int j0 = 0; j0 < ncols; j0 += 4) {
    uint32_t v = otg_ld_u32_unaligned(REV ? T + tl - 4 - j0 : T + j0);
    if (REV) v = __builtin_bswap32(v);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t c = ((v >> (8 * b)) >> 1) & 3u;
      const uint64_t Eq = c == 0 ? eq0 : c == 1 ? eq1 : c == 2 ? eq2 : eq3;
      const uint64_t Xv = Eq | Mv;
      const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
      uint64_t Ph = Mv | ~(Xh | Pv);
      uint64_t Mh = Pv & Xh;
      score += (int)(Ph >> 63) - (int)(Mh >> 63);
      Ph = (Ph << 1) | 1ull;                      // D[0][j] = j: the top boundary grows by one per column
      Mh = Mh << 1;
      Pv = Mh | ~(Xv | Ph);
      Mv = Ph & Xv;
      if (score < best) { best = score; bj = j0 + b + 1; }
    }
  }
  *dmin = best; *jmin = bj;
}

// route_cnt[0..OTG_MYERS_TIERS] / route_lists as in wfa_edit_kernel_v2; one more list (count wf_cnt) = input of the wavefront pass
__global__ __launch_bounds__(256) void edit_route_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks, const uint32_t* __restrict__ todo,
    const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm, float cap_coeff, uint32_t* __restrict__ route_cnt,
    uint32_t* __restrict__ route_lists, uint32_t route_stride, float route_margin, uint32_t* __restrict__ wf_cnt, uint32_t* __restrict__ wf_list, uint32_t tier_mask)
{
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_todo; base += gridDim.x * blockDim.x) {
    const uint32_t tk = base + threadIdx.x;
    const bool act = tk < n_todo;
    uint32_t ti = 0;
    int tier = -1;                      // -1: not active; 0..OTG_MYERS_TIERS: list of that tier; OTG_MYERS_TIERS + 1: wavefront pass
    if (act) {
      ti = todo ? todo[tk] : tk;
      const otg_align_task t = tasks[ti];
      const int pl = (int)t.pattern_len, tl = (int)t.text_len;
      tier = OTG_MYERS_TIERS + 1;
      // an end that is free on either sequence cannot anchor a sample: use the other end alone, or leave the pair to the wavefront pass
      const bool ef = t.endsfree != 0;
      const bool fwd_ok = !ef || (t.pattern_begin_free == 0 && t.text_begin_free == 0);
      const bool rev_ok = !ef || (t.pattern_end_free == 0 && t.text_end_free == 0);
      if ((fwd_ok || rev_ok) && pl >= 160 && tl >= 160 && pl < 65535 && tl < 65535) {
        const uint8_t* P = arena + t.pattern_off;
        const uint8_t* T = arena + t.text_off;
        int d1 = 0, j1 = -64, d2 = 0, j2 = -64;
        if (fwd_ok) otg_sample64<false>(P, pl, T, tl, 96, &d1, &j1);
        if (rev_ok) otg_sample64<true>(P, pl, T, tl, 96, &d2, &j2);
        const int dlen = ef ? 0 : (pl > tl ? pl - tl : tl - pl);
        // the sampled rate (edits per row + column) times the rows + columns the alignment will cover: everything for a global
        // alignment; with free ends on one sequence only, the other one twice (it is covered whole, the free-ended one by as much)
        int extent = pl + tl;
        if (ef) {
          const bool pfree = t.pattern_begin_free > 0 || t.pattern_end_free > 0, tfree = t.text_begin_free > 0 || t.text_end_free > 0;
          if (pfree && !tfree) extent = 2 * tl < extent ? 2 * tl : extent;
          else if (tfree && !pfree) extent = 2 * pl < extent ? 2 * pl : extent;
        }
        float est = (float)(d1 + d2) * (float)extent / (float)(128 + j1 + j2);
        if (est > (float)(pl + tl)) est = (float)(pl + tl);
        int cap = (int)(cap_coeff * sqrtf((float)(pl > tl ? pl : tl)));
        if (cap < 48) cap = 48;
        const int need = (int)(route_margin * (est + (float)dlen));
        // near-identical pairs (projected distance within the wavefront pass's score cap) are finished there
        if ((int)est + dlen > cap) tier = otg_route_tier(t, need, tier_mask);
      }
    }
#pragma unroll
    for (int q = 0; q < OTG_MYERS_TIERS + 2; ++q) {
      const unsigned long long m = __ballot(tier == q);
      if (m) {
        uint32_t* c = q == OTG_MYERS_TIERS + 1 ? wf_cnt : route_cnt + q;
        uint32_t b0 = 0;
        if (lane == (int)__builtin_ctzll(m)) b0 = atomicAdd(c, (uint32_t)__builtin_popcountll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, (int)__builtin_ctzll(m));
        if (tier == q) {
          uint32_t* l = q == OTG_MYERS_TIERS + 1 ? wf_list : route_lists + (size_t)q * route_stride;
          l[b0 + (uint32_t)__builtin_popcountll(m & lt)] = ti;
        }
      }
    }
  }
}

// ---- counting sort of a todo list by the number of columns of each pair (longest first, 32-column buckets).  The
// bit-parallel tiers put 2-8 pairs into one wave, and a wave runs as long as its longest pair: neighbours in a sorted
// list have nearly equal lengths, and starting with the long pairs shortens the tail of the persistent grid.
constexpr int SORT_BUCKETS = 512;
__device__ __forceinline__ int sort_bucket(const otg_align_task& t)
{
  const uint32_t n = t.pattern_len < t.text_len ? t.pattern_len : t.text_len;
  const int b = (int)(n >> 5);
  return SORT_BUCKETS - 1 - (b < SORT_BUCKETS ? b : SORT_BUCKETS - 1);          // descending length
}
__global__ __launch_bounds__(256) void K_sort_hist(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, const otg_align_task* __restrict__ tasks,
                                                   uint32_t* __restrict__ hist)
{
  __shared__ uint32_t h[SORT_BUCKETS];
  for (int b = (int)threadIdx.x; b < SORT_BUCKETS; b += 256) h[b] = 0;
  __syncthreads();
  const uint32_t n = *n_ptr;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&h[sort_bucket(tasks[list[i]])], 1u);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < SORT_BUCKETS; b += 256) if (h[b]) atomicAdd(&hist[b], h[b]);
}
__global__ __launch_bounds__(SORT_BUCKETS) void K_sort_scan(uint32_t* __restrict__ hist)
{
  __shared__ uint32_t s[SORT_BUCKETS];
  const int i = (int)threadIdx.x;
  s[i] = hist[i];
  __syncthreads();
  for (int off = 1; off < SORT_BUCKETS; off <<= 1) {
    const uint32_t v = i >= off ? s[i - off] : 0u;
    __syncthreads();
    s[i] += v;
    __syncthreads();
  }
  hist[i] = s[i] - hist[i];                   // exclusive prefix = first output position of the bucket
}
// every block owns a contiguous slice: local histogram in LDS, one global reservation per non-empty bucket, local scatter
__global__ __launch_bounds__(256) void K_sort_scatter(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, const otg_align_task* __restrict__ tasks,
                                                      uint32_t* __restrict__ pos, uint32_t* __restrict__ out)
{
  __shared__ uint32_t cnt[SORT_BUCKETS], basep[SORT_BUCKETS];
  const uint32_t n = *n_ptr;
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (int b = (int)threadIdx.x; b < SORT_BUCKETS; b += 256) cnt[b] = 0;
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&cnt[sort_bucket(tasks[list[i]])], 1u);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < SORT_BUCKETS; b += 256) { basep[b] = cnt[b] ? atomicAdd(&pos[b], cnt[b]) : 0u; cnt[b] = 0; }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t ti = list[i];
    const int b = sort_bucket(tasks[ti]);
    out[basep[b] + atomicAdd(&cnt[b], 1u)] = ti;
  }
}

} // namespace

// Enqueue the tier chain: wavefront tier 1 (LDS, score-capped) -> bit-parallel tiers 0..4 -> wavefront tier 2 -> global.  Requires: d_arena padded with >= 8 readable bytes after the last
// sequence byte.  Uses SLOT_COUNTERS (64 u32), SLOT_TODO ((OTG_MYERS_TIERS + 4) * n_tasks u32), SLOT_WF_WS (last tier only).
int otg_launch_edit(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, uint32_t n_tasks,
                    int32_t* d_scores, uint64_t* d_cells, float* kernel_ms, uint64_t* launches)
{
  return otg_launch_edit_todo(ctx, d_arena, d_tasks, nullptr, nullptr, n_tasks, d_scores, d_cells, kernel_ms, launches);
}

// Same, for a compacted todo list of task slots whose length lives on the device (d_n_todo); n_tasks is
// then the number of task SLOTS (upper bound, sizes the overflow lists and the grid).
int otg_launch_edit_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                         const uint32_t* d_n_todo, uint32_t n_tasks, int32_t* d_scores, uint64_t* d_cells,
                         float* kernel_ms, uint64_t* launches)

{
  if (n_tasks == 0) return OTG_OK;
  if (ctx->heur_strategy == OTG_HEURISTIC_WFADAPTIVE)
    return otg_launch_edit_adaptive_todo(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, kernel_ms, launches);
  uint32_t* cnt = (uint32_t*)otg_slot(ctx, SLOT_COUNTERS, 64 * sizeof(uint32_t));
  constexpr int NT = OTG_MYERS_TIERS;
  uint32_t* todo = (uint32_t*)otg_slot(ctx, SLOT_TODO, (size_t)(NT + 4) * n_tasks * sizeof(uint32_t));
  if (!cnt || !todo) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(cnt, 0, 8 * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(cnt + 16, 0, 16 * sizeof(uint32_t), ctx->stream));
  // lists[t] (t = 0..NT-1) feed the bit-parallel tiers, lists[NT] the wide wavefront tier, listG its overflow.  The
  // first kernel routes every pair it cannot finish to the tier matching its estimated distance; a tier that is too
  // narrow appends the pair to the next list.  cnt[32 + t] = length of lists[t].
  uint32_t* const lists = todo;
  uint32_t* listG = todo + (size_t)(NT + 1) * n_tasks;
  uint32_t* const rc = cnt + 32;
  HIP_TRY(ctx, hipMemsetAsync(rc, 0, 16 * sizeof(uint32_t), ctx->stream));      // rc[0 .. NT], then (cnt + 44 ..) the counters of the sampling router
  // The two three-block tiers pay where their lists are long (a pair pays for the band of its tier: <3,8> and <3,16> take what would run on tiers
  // 1.45 x as costly — edit stage 723 -> 664 ms on the 1-10 kb shard, the reassignment pass of a batch with clipped reads likewise) and cost a
  // launch each — sort + a kernel that lasts as long as its longest pair, ~2 ms — where they are short (a batch of 1 000 regions, the reassignment
  // pass of fully spanning reads).  The batches of a job are alike, so each kind of pass decides from what its previous pass saw: an optional tier
  // that ran stays while it got 20 000 pairs, one that did not run comes in when the tier above it got 40 000.  No history: by the task slots.
  static const int tiers_env = getenv("OTG_EDIT_TIERS") ? atoi(getenv("OTG_EDIT_TIERS")) : 0;      // test switch: the tiers that run, as a bit mask (tier 0 and the last always do)
  const int kind = ctx->edit_pass_kind & 1;
  uint32_t tier_mask = (n_tasks >= 2000000u && kind == 0) ? 0xFFu : 0xEBu;
  if (ctx->edit_hist && ctx->edit_hist_mask[kind] && hipEventQuery(ctx->edit_hist_ev[kind]) == hipSuccess) {
    const uint32_t* h = ctx->edit_hist + 16 * kind;
    const uint32_t was = ctx->edit_hist_mask[kind];
    tier_mask = 0xEBu;
    if ((was >> 2) & 1u ? h[2] >= 20000u : h[3] >= 40000u) tier_mask |= 1u << 2;
    if ((was >> 4) & 1u ? h[4] >= 20000u : h[5] >= 40000u) tier_mask |= 1u << 4;
  }
  (void)hipGetLastError();
  if (tiers_env) tier_mask = ((uint32_t)tiers_env & 0xFFu) | 0x81u;
  static const bool no_myers = getenv("OTG_NO_MYERS") != nullptr;
  static const bool no_route = getenv("OTG_NO_EDIT_ROUTE") != nullptr;
  // a tier that turns out too narrow costs about half of going one tier up straight away, so the cheapest choice sits a little below
  // the median estimate (measured at config 1: 0.85-0.92 flat optimum; OTG_EDIT_ROUTE_MARGIN overrides, in percent)
  static const float route_margin = getenv("OTG_EDIT_ROUTE_MARGIN") ? (float)atoi(getenv("OTG_EDIT_ROUTE_MARGIN")) / 100.0f : 0.88f;
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  {
    // with the bit-parallel tiers behind it this pass only has to hold wavefronts of ~2 x cap diagonals: 1024 diagonals
    // (4 KB per wave) leave room for 32 waves per CU, which is what hides the probe latency of this short pass
    constexpr int WPB = 4;
    uint32_t want = (n_tasks + WPB - 1) / WPB;
    if (no_myers) {
      constexpr int CAP = 2048;                                // 2 x 2048 x u16 = 8 KB per wave -> 20 waves / CU
      uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * 5, want);
      hipLaunchKernelGGL((wfa_edit_kernel_v2<CAP, WPB>), dim3(grid), dim3(WPB * 64), (size_t)CAP * 2 * WPB * sizeof(uint16_t), ctx->stream, d_arena, d_tasks,
                         d_todo, d_n_todo, n_tasks, d_scores, d_cells, cnt + 0, rc + NT, lists + (size_t)NT * n_tasks, 0.0f,
                         (uint32_t*)nullptr, lists, n_tasks, route_margin, tier_mask);
    } else {
      constexpr int CAP = 1024;
      uint32_t grid = std::min<uint32_t>((uint32_t)ctx->n_cu * 8, want);
      const uint32_t* in = d_todo; const uint32_t* in_n = d_n_todo; uint32_t in_imm = n_tasks;
      static const bool no_sample = getenv("OTG_NO_EDIT_SAMPLE") != nullptr;
      if (!no_route && !no_sample) {
        // tier choice from two 64-base samples per pair (one pair per lane); only what it leaves goes through the wavefront pass
        uint32_t* wf_list = todo + (size_t)(NT + 3) * n_tasks;
        const uint32_t rg = std::min<uint32_t>((n_tasks + 255) / 256, (uint32_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(edit_route_kernel, dim3(rg), dim3(256), 0, ctx->stream, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, 1.0f, rc, lists,
                           n_tasks, route_margin, cnt + 44, wf_list, tier_mask);
        in = wf_list; in_n = cnt + 44; in_imm = 0;
      }
      hipLaunchKernelGGL((wfa_edit_kernel_v2<CAP, WPB>), dim3(grid), dim3(WPB * 64), (size_t)CAP * 2 * WPB * sizeof(uint16_t), ctx->stream, d_arena, d_tasks,
                         in, in_n, in_imm, d_scores, d_cells, cnt + 0, rc + 0, lists, 1.0f,
                         no_route ? (uint32_t*)nullptr : rc, lists, n_tasks, route_margin, tier_mask);
    }
  }
  uint32_t dbg_routed[16] = {0};
  if (getenv("OTG_DEBUG")) { (void)hipStreamSynchronize(ctx->stream); (void)hipMemcpy(dbg_routed, rc, sizeof(dbg_routed), hipMemcpyDeviceToHost); }      // what the routers sent where, before any tier passed pairs on
  if (!no_myers) {
    uint32_t* const tick[NT] = {cnt + 2, cnt + 4, cnt + 6, cnt + 20, cnt + 22, cnt + 24, cnt + 26, cnt + 28};
    static const bool no_sort = getenv("OTG_NO_EDIT_SORT") != nullptr;
    uint32_t* sorted = todo + (size_t)(NT + 2) * n_tasks;
    uint32_t* hist = (uint32_t*)otg_slot(ctx, SLOT_ROWTAB, NT * SORT_BUCKETS * sizeof(uint32_t));
    if (!hist) return OTG_ERR_HIP;
    if (!no_sort) HIP_TRY(ctx, hipMemsetAsync(hist, 0, NT * SORT_BUCKETS * sizeof(uint32_t), ctx->stream));
    for (int tier = 0; tier < NT; ++tier) {
      if (!((tier_mask >> tier) & 1u)) continue;
      int next = tier + 1;                       // where a pair goes whose band turns out too narrow: the next tier that runs
      while (next < NT && !((tier_mask >> next) & 1u)) ++next;
      const uint32_t* in = lists + (size_t)tier * n_tasks;
      if (!no_sort && tier < NT - 2) {               // tiers that share a wave between pairs, and the whole-wave tier for its tail
        uint32_t* h = hist + tier * SORT_BUCKETS;
        const uint32_t sg = std::min<uint32_t>((n_tasks + 2047) / 2048, (uint32_t)ctx->n_cu * 2);
        hipLaunchKernelGGL(K_sort_hist, dim3(sg), dim3(256), 0, ctx->stream, in, (const uint32_t*)(rc + tier), d_tasks, h);
        hipLaunchKernelGGL(K_sort_scan, dim3(1), dim3(SORT_BUCKETS), 0, ctx->stream, h);
        hipLaunchKernelGGL(K_sort_scatter, dim3(sg), dim3(256), 0, ctx->stream, in, (const uint32_t*)(rc + tier), d_tasks, h, sorted);
        in = sorted;
      }
      const int rc_ = otg_launch_myers(ctx, tier, d_arena, d_tasks, in, rc + tier, n_tasks, d_scores, d_cells,
                                       tick[tier], rc + next, lists + (size_t)next * n_tasks);
      if (rc_) return rc_;
    }
  }
  const uint32_t* cur = lists + (size_t)NT * n_tasks; const uint32_t* cur_n = rc + NT;
  {
    constexpr int CAP = 8192, WPB = 1;                         // 32 KB per wave -> 5 waves / CU, scores up to ~4000
    const size_t lds = (size_t)CAP * 2 * WPB * sizeof(uint16_t);
    uint32_t grid = (uint32_t)ctx->n_cu * 5;
    hipLaunchKernelGGL((wfa_edit_kernel_v2<CAP, WPB>), dim3(grid), dim3(WPB * 64), lds, ctx->stream, d_arena, d_tasks,
                       cur, cur_n, 0u, d_scores, d_cells, cnt + 16, cnt + 17, listG, 0.0f, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 1.0f, 0u);
  }
  {
    // last tier: global-memory wavefront sized for the longest possible pair; only reached by huge inputs
    constexpr int WPB = 4;
    uint32_t grid = (uint32_t)ctx->n_cu;
    int gcap = (int)(2 * (size_t)ctx->max_seq_len + 4);
    int32_t* ws = (int32_t*)otg_slot(ctx, SLOT_WF_WS, (size_t)grid * WPB * (size_t)gcap * sizeof(int32_t));
    if (!ws) return OTG_ERR_HIP;
    hipLaunchKernelGGL((wfa_edit_kernel<0, WPB, true>), dim3(grid), dim3(WPB * 64), 0, ctx->stream, d_arena, d_tasks,
                       (const uint32_t*)listG, (const uint32_t*)(cnt + 17), 0u, d_scores, d_cells, cnt + 18, cnt + 19,
                       (uint32_t*)nullptr, ws, gcap);
  }
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));      // after the LAST tier of the chain
  if (!no_myers) {                    // what the tiers saw, for the next pass of this kind
    if (!ctx->edit_hist) {
      HIP_TRY(ctx, hipHostMalloc((void**)&ctx->edit_hist, 32 * sizeof(uint32_t), hipHostMallocDefault));
      for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->edit_hist_ev[i], hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->edit_hist + 16 * kind, rc, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->edit_hist_ev[kind], ctx->stream));
    ctx->edit_hist_mask[kind] = tier_mask;
  }
  HIP_TRY(ctx, hipGetLastError());
  if (getenv("OTG_DEBUG")) {
    hipError_t er = hipStreamSynchronize(ctx->stream);
    uint32_t h[48];
    (void)hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[otg] edit (tiers %02x): %s; wavefront pass input %u; inputs of the bit-parallel tiers 0..7: %u %u %u %u %u %u %u %u, wide wavefront tier %u, its overflow %u\n",
            tier_mask, hipGetErrorString(er), h[44], h[32], h[33], h[34], h[35], h[36], h[37], h[38], h[39], h[40], h[17]);
    fprintf(stderr, "[otg] edit:   of which routed there directly: %u %u %u %u %u %u %u %u (the rest came up from the tier below: its band was too narrow)\n",
            dbg_routed[0], dbg_routed[1], dbg_routed[2], dbg_routed[3], dbg_routed[4], dbg_routed[5], dbg_routed[6], dbg_routed[7]);
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
