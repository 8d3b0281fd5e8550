// wfa_affine.hip — batched gap-affine wavefront aligner with full op string (gfx950): the launch chain and the tiers around the
// register-resident ones (wfa_affine_reg.hip).
//
// Replaces wfa::WFAlignerGapAffine(x,o,e, Alignment, MemoryMed)::alignEnd2End / alignEndsFree +
// getAlignmentCigar() (reference: src/assemble.cpp:50; call sites src/analignments.cpp:25,31,37,268-280).
//
// The chain, for the default penalties (4,6,2) -> (2,4,1) after gcd reduction:
//   1. score-bound pass (wfa_affine_bound1_kernel): a banded, score-only run whose result U is an upper bound of the optimum;
//   2. one counting sort on (tier, bound): which register tier takes an alignment follows from U and its shape alone;
//   3. the register tiers (wfa_affine_reg.hip: windows of 1024 ... 8192 diagonals) — exact, restricted to the diamond of cells that can
//      lie on an alignment of score <= U;
//   4. the HBM-row tiers (wfa_affine_kernel_v3: 16-bit M rows in HBM / L2, I and D in LDS, byte compares) for what the register tiers
//      cannot take or give up: bytes outside ACGT, no bound, windows beyond 8192 diagonals, a full match-run queue;
//   5. the generic kernel (any penalties, any lengths; int32 rings in HBM).
// Every tier writes one provenance byte per kept (score, diagonal) cell into a per-alignment slab and shares the backtrace
// (wfa_affine_common.hpp).  An alignment a tier cannot finish is queued on the device for the next one.
#include "wfa_affine_common.hpp"
#include "wfa_affine_reg.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

using namespace otg_affine;

namespace {

// The generic kernel.  One wave64 per alignment, persistent waves + device ticket.  Per wave, in HBM/L2:
//   * rings of wavefront rows (int32 offsets, index = k + plen + 1): M keeps max(x,o+e)/g + 1 rows,
//     I and D keep e/g + 1 rows (scores are walked in units of g = gcd(x, o+e, e): (4,6,2) -> 2,4,1);
//   * one provenance byte per (score, diagonal), rows bump-allocated in a per-wave slab and addressed through a per-wave row table;
//   * the reversed op list of the backtrace.
// After the forward pass the same wave walks the provenance back (uniform scalar walk), then unpacks the
// ops forward, re-deriving match runs 64 bytes at a time with a ballot (pcigar_unpack_affine semantics:
// matches are extended only in the M state; a gap close is a marker, not an op), and writes the op string
// (M X I D, free end gaps explicit).
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void wfa_affine_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int xs, int oes, int es, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws)
{
  constexpr int QCAP = 2048;
  __shared__ int s_lo[WPB][3][64];
  __shared__ int s_hi[WPB][3][64];
  __shared__ uint32_t s_queue[WPB][QCAP];        // 32-bit entries: wavefronts wider than 65 535 diagonals (pairs beyond 32 kb) end up here
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  uint8_t* my = ws.base + (size_t)(blockIdx.x * WPB + wib) * ws.stride;
  int32_t* ringM = (int32_t*)my;
  int32_t* ringI = ringM + (size_t)ws.rm * ws.capa;
  int32_t* ringD = ringI + (size_t)ws.ri * ws.capa;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  int* mlo = s_lo[wib][0]; int* mhi = s_hi[wib][0];
  int* ilo = s_lo[wib][1]; int* ihi = s_hi[wib][1];
  int* dlo = s_lo[wib][2]; int* dhi = s_hi[wib][2];
  volatile lds_u32* queue = (volatile lds_u32*)&s_queue[wib][0];
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
    const int kb = pl + 1; // ring index of diagonal k is k + kb
    const int kend = tl - pl;
    bool fail = pl + tl + 3 > ws.capa;
    size_t slab_top = 0;
    uint64_t W = 0;
    int s_end = -1, k_end = 0;

    // ---------------- forward pass
    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      const int sm = s % ws.rm, si = s % ws.ri;
      int lo, hi;
      const int32_t *Mx = nullptr, *Mo = nullptr, *Ie = nullptr, *De = nullptr;
      int mxlo = 1, mxhi = 0, molo = 1, mohi = 0, ielo = 1, iehi = 0, delo = 1, dehi = 0;
      if (s == 0) {
        lo = ef ? imax(-t.pattern_begin_free, -pl) : 0;
        hi = ef ? imin(t.text_begin_free, tl) : 0;
      } else {
        lo = 1 << 30; hi = -(1 << 30);
        if (s - xs >= 0) { int q = (s - xs) % ws.rm; mxlo = mlo[q]; mxhi = mhi[q]; Mx = ringM + (size_t)q * ws.capa; }
        if (s - oes >= 0) { int q = (s - oes) % ws.rm; molo = mlo[q]; mohi = mhi[q]; Mo = ringM + (size_t)q * ws.capa; }
        if (s - es >= 0) { int q = (s - es) % ws.ri; ielo = ilo[q]; iehi = ihi[q]; delo = dlo[q]; dehi = dhi[q];
                           Ie = ringI + (size_t)q * ws.capa; De = ringD + (size_t)q * ws.capa; }
        if (mxhi >= mxlo) { lo = imin(lo, mxlo); hi = imax(hi, mxhi); }
        if (mohi >= molo) { lo = imin(lo, molo - 1); hi = imax(hi, mohi + 1); }
        if (iehi >= ielo) { lo = imin(lo, ielo + 1); hi = imax(hi, iehi + 1); }
        if (dehi >= delo) { lo = imin(lo, delo - 1); hi = imax(hi, dehi - 1); }
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
      }
      if (hi < lo) { // null wavefront: this score is not reachable
        mlo[sm] = 1; mhi[sm] = 0; ilo[si] = 1; ihi[si] = 0; dlo[si] = 1; dhi[si] = 0; rowtab[s] = -1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (s > g * 0 + 2 * (oes + es * (pl + tl)) + 8) { fail = true; }
        continue;
      }
      const int width = hi - lo + 1;
      if (slab_top + (size_t)width > ws.slab_bytes) { fail = true; break; }
      uint8_t* btrow = slab + slab_top - lo;   // btrow[k]
      rowtab[s] = (int64_t)slab_top - lo; mlo[sm] = lo; mhi[sm] = hi;
      if (s == 0) { ilo[si] = 1; ihi[si] = 0; dlo[si] = 1; dhi[si] = 0; }   // no I/D wavefront at score 0
      else { ilo[si] = lo; ihi[si] = hi; dlo[si] = lo; dhi[si] = hi; }
      slab_top += (size_t)width;
      W += 3ull * (uint64_t)width;
      int32_t* Mc = ringM + (size_t)sm * ws.capa;
      int32_t* Ic = ringI + (size_t)si * ws.capa;
      int32_t* Dc = ringD + (size_t)si * ws.capa;
      bool done = false;
      int qn = 0;
      // drain: diagonals whose 8-byte probe matched fully are extended 16 bytes per iteration from a
      // wave-compacted LDS queue (entries = k - lo), so no lane waits for the slowest diagonal of its chunk
      auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        while (qn > 0) {
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int kk = 0, h = 0, v = 0;
            bool more = false;
            if (act) {
              kk = lo + (int)queue[q0 + lane];
              h = Mc[kk + kb];
              v = h - kk;
              uint64_t a0, a1, b0, b1;
              __builtin_memcpy(&a0, P + v, 8); __builtin_memcpy(&a1, P + v + 8, 8);
              __builtin_memcpy(&b0, T + h, 8); __builtin_memcpy(&b1, T + h + 8, 8);
              const uint64_t xl = a0 ^ b0, xh = a1 ^ b1;
              int m = xl ? (__builtin_ctzll(xl) >> 3) : (xh ? 8 + (__builtin_ctzll(xh) >> 3) : 16);
              const int rem = imin(pl - v, tl - h);
              m = imin(m, rem);
              v += m; h += m;
              more = (m == 16) && v < pl && h < tl;
              Mc[kk + kb] = h;
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint32_t)(kk - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
      };
      for (int c = lo; c <= hi; c += 64) {
        const int k = c + lane;
        const int j = k + kb;
        const bool in = k <= hi;
        int mx, ins = OTG_NULL_OFF, del = OTG_NULL_OFF;
        uint32_t bits = 0;
        if (s == 0) {
          mx = k > 0 ? k : 0;
        } else {
          int io = (Mo && k - 1 >= molo && k - 1 <= mohi) ? Mo[j - 1] : OTG_NULL_OFF;
          int dop = (Mo && k + 1 >= molo && k + 1 <= mohi) ? Mo[j + 1] : OTG_NULL_OFF;
          int ix = (Ie && k - 1 >= ielo && k - 1 <= iehi) ? Ie[j - 1] : OTG_NULL_OFF;
          int dx = (De && k + 1 >= delo && k + 1 <= dehi) ? De[j + 1] : OTG_NULL_OFF;
          int mm = (Mx && k >= mxlo && k <= mxhi) ? Mx[j] : OTG_NULL_OFF;
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
          uint64_t a, b;
          __builtin_memcpy(&a, P + v, 8);
          __builtin_memcpy(&b, T + h, 8);
          const uint64_t xx = a ^ b;
          int m = xx ? (__builtin_ctzll(xx) >> 3) : 8;
          const int rem = imin(pl - v, tl - h);
          m = imin(m, rem);
          v += m; h += m;
          more = (m == 8) && v < pl && h < tl;
        }
        if (in) {
          Mc[j] = valid ? h : OTG_NULL_OFF;
          Ic[j] = ins; Dc[j] = del;
          btrow[k] = (uint8_t)bits;
        }
        const unsigned long long mq = __ballot(more);
        if (more) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
          queue[qn + rank] = (uint32_t)(k - lo);
        }
        qn += __builtin_popcountll(mq);
        if (qn + 64 > QCAP) drain();
      }
      drain();
      // termination (the wavefront is fully extended now)
      if (!ef) {
        if (kend >= lo && kend <= hi && Mc[kend + kb] >= tl) { done = true; s_end = s; k_end = kend; }
      } else {
        for (int c = lo; c <= hi && !done; c += 64) {      // lowest diagonal first (WFA2 scans k ascending)
          const int k = c + lane;
          bool fin = false;
          if (k <= hi) {
            const int h = Mc[k + kb];
            const int v = h - k;
            fin = h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
          }
          const unsigned long long fm = __ballot(fin);
          if (fm) { done = true; s_end = s; k_end = c + (int)__builtin_ctzll(fm); }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (done) break;
    }

    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }

    if (!backtrace_unpack<false>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_queue[wib][0], EqBytes{P, T})) continue;
    if (cells) cells[ti] = W;
  }
}

// ---------------------------------------------------------------------------------------------------
// Score bound pass.  A banded, score-only run of the same recurrence, entirely in registers.  Any alignment it finds is a valid
// alignment of the pair, so its score U is an UPPER bound of the optimum (equal to it whenever the optimal path stays inside the
// band, which is the normal case for reads of one allele).  The exact tiers use U only to skip cells that cannot lie on an alignment
// of score <= U, so a loose U costs time, never correctness.  U = INT_MAX when the band cannot hold the start diagonals.
__device__ __forceinline__ int dpp_shl1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false); }   // lane i <- lane i+1
__device__ __forceinline__ int dpp_shr1b(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }  // lane i <- lane i-1

// Sliding variant of the bound pass: ONE diagonal per lane (64-diagonal band) that follows the diagonal with the
// furthest anti-diagonal progress (checked every second score, one diagonal per move; the state moves with a DPP
// shift).  Four times cheaper per score than the static 256-diagonal band and not limited by the distance
// between start and end diagonals; the bound is as valid (any alignment found is an alignment), just looser
// when the optimal path strays more than ~20 diagonals from the leader.
template <int XS, int OES>
__global__ __launch_bounds__(256) void wfa_affine_bound1_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ bound, uint32_t* __restrict__ ticket)
{
  static_assert(XS == 2 && OES == 4, "ring roles below are written out for (2,4,1)");
  constexpr int NULLV = -(1 << 29);
  // per wave: both sequences packed to 2 bits per base (as in the LDS tiers of the exact pass): a probe covers 32 bases
  // and comes from LDS; pairs that do not fit or contain a byte outside ACGT use the byte probes from HBM
  constexpr int SEQW = 800;                               // words per wave: pl + tl up to ~12.6 kb
  __shared__ uint32_t s_pk[4][SEQW];
  using lds_u32b = __attribute__((address_space(3))) uint32_t;
  const int lane = threadIdx.x & 63;
  volatile lds_u32b* SQ = (volatile lds_u32b*)&s_pk[threadIdx.x >> 6][0];
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
    int pbf = ef ? imin(t.pattern_begin_free, pl) : 0, tbf = ef ? imin(t.text_begin_free, tl) : 0;
    int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0;
    // A start range too wide for the band (free leading gaps) is handled on the REVERSED sequences: gap-affine
    // scores are symmetric under reversing both strings, free begins become free ends, which the band follows.
    const bool rev = pbf + tbf + 1 > 40 && imin(pef, pl) + imin(tef, tl) + 1 <= 40;
    if (rev) { int x = pbf; pbf = imin(pef, pl); pef = x; x = tbf; tbf = imin(tef, tl); tef = x; }
    const int lo0 = -pbf, hi0 = tbf;
    // 8 bytes of the (possibly reversed) pattern / text starting at position pos in [0, len]
    auto ld8s = [&](const uint8_t* S, int len, int pos) -> uint64_t {
      if (!rev) return otg_load8(S + pos);
      const int a = len - 8 - pos;                        // reversed byte i = S[len-1-pos-i]
      const uint64_t x = otg_load8(S + (a > 0 ? a : 0)) << (8 * imin(a < 0 ? -a : 0, 7));
      return __builtin_bswap64(x);
    };
    int result = 0x7fffffff;
    if (hi0 - lo0 + 1 <= 40 && pl > 0 && tl > 0 && pl < 32766 && tl < 32766) {
      // pack (in the orientation the pass runs in): word q = bases 16q .. 16q+15, code (byte >> 1) & 3
      const int offT = (pl + 15) / 16 + 3;
      bool packed = offT + (tl + 15) / 16 + 3 <= SEQW;
      if (packed) {
        bool bad = false;
        auto pack = [&](const uint8_t* S, int len, int woff) {
          for (int q = lane; q < (len + 15) / 16; q += 64) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int b0 = 16 * q + 8 * j;
              const uint64_t x = ld8s(S, len, b0 < len ? b0 : len);
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
        packed = __ballot(bad) == 0ull;
      }
      auto ld32b = [&](int woff, int pos) -> uint64_t {
        const int w = woff + (pos >> 4);
        const uint32_t sh = (uint32_t)(pos & 15) * 2u;
        const uint32_t d0 = SQ[w], d1 = SQ[w + 1], d2 = SQ[w + 2];
        return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
      };
      int bk0 = ((lo0 + hi0) >> 1) - 32;                 // diagonal of lane 0
      int M1 = NULLV, M2 = NULLV, M3 = NULLV, M4 = NULLV, I = NULLV, D = NULLV, cur;
      { const int k = bk0 + lane, h = k > 0 ? k : 0, v = h - k;
        cur = (k >= lo0 && k <= hi0 && h <= tl && v <= pl) ? h : NULLV; }
      auto extend_and_test = [&]() -> bool {
        const int k = bk0 + lane;
        bool more = cur >= 0;
        if (packed) {
          do {
            const int h = cur, v = h - k;
            const int vc = imin(imax(v, 0), pl), hc = imin(imax(h, 0), tl);
            const uint64_t xx = ld32b(0, vc) ^ ld32b(offT, hc);
            int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
            m = imin(m, imin(pl - v, tl - h));
            if (more) cur = h + m;
            more = more && m == 32 && v + 32 < pl && h + 32 < tl;
          } while (__ballot(more));
        } else {
          do {
            const int h = cur, v = h - k;
            const int vc = imin(imax(v, 0), pl), hc = imin(imax(h, 0), tl);
            const uint64_t xx = ld8s(P, pl, vc) ^ ld8s(T, tl, hc);
            int m = xx ? (int)(__builtin_ctzll(xx) >> 3) : 8;
            m = imin(m, imin(pl - v, tl - h));
            if (more) cur = h + m;
            more = more && m == 8 && v + 8 < pl && h + 8 < tl;
          } while (__ballot(more));
        }
        const int h = cur, v = h - k;
        const bool fin = h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
        return __ballot(fin) != 0;
      };
      if (extend_and_test()) result = 0;
      const int smax = pl + tl + 64;
      for (int s = 1; result == 0x7fffffff && s <= smax; ++s) {
        M4 = M3; M3 = M2; M2 = M1; M1 = cur;             // M1 = row s-1 ... M4 = row s-4
        if ((s & 1) == 0) {
          // follow the leader: keep the diagonal with the furthest anti-diagonal inside lanes [24, 40)
          const int prog = M1 >= 0 ? 2 * M1 - (bk0 + lane) : NULLV;
          const int best = otg_wave_max_i32(prog);
          const unsigned long long at = __ballot(prog == best && best > NULLV);
          if (at) {
            const int bl = (int)__builtin_ctzll(at);
            if (bl >= 40) {                               // band moves up: lane i takes over lane i+1
              M1 = dpp_shl1(M1); M2 = dpp_shl1(M2); M3 = dpp_shl1(M3); M4 = dpp_shl1(M4); I = dpp_shl1(I); D = dpp_shl1(D);
              if (lane == 63) { M1 = M2 = M3 = M4 = I = D = NULLV; }
              ++bk0;
            } else if (bl < 24) {
              M1 = dpp_shr1b(M1); M2 = dpp_shr1b(M2); M3 = dpp_shr1b(M3); M4 = dpp_shr1b(M4); I = dpp_shr1b(I); D = dpp_shr1b(D);
              if (lane == 0) { M1 = M2 = M3 = M4 = I = D = NULLV; }
              --bk0;
            }
          }
        }
        const int k = bk0 + lane;
        int mo_l = dpp_shr1b(M4), i_l = dpp_shr1b(I), mo_r = dpp_shl1(M4), d_r = dpp_shl1(D);
        if (lane == 0) { mo_l = NULLV; i_l = NULLV; }
        if (lane == 63) { mo_r = NULLV; d_r = NULLV; }
        int ins = imax(i_l, mo_l) + 1;
        int del = imax(d_r, mo_r);
        const int mis = M2 + 1;
        if (ins < 0 || ins > tl || ins - k > pl) ins = NULLV;
        if (del < 0 || del > tl || del - k > pl) del = NULLV;
        int mx = imax(del, imax(mis, ins));
        if (mx < 0 || mx > tl || mx - k > pl) mx = NULLV;
        I = ins; D = del; cur = mx;
        if (extend_and_test()) result = s;
      }
    }
    bound[ti] = result;      // wave-uniform value, same store from every lane
  }
}

// ---------------------------------------------------------------------------------------------------
// v3 forward kernel (tier 1; gap-extension step 1 after gcd reduction, sequences < 32767):
//   * I and D wavefronts live in LDS as SIGNED 16-bit offsets (null = any negative value) and are updated
//     IN PLACE (I[s][k] needs I[s-1][k-1]: left neighbour by DPP wave_shr + carry; D[s][k] needs D[s-1][k+1]:
//     the right neighbour is still old in an ascending sweep);
//   * the M ring (max(x,o+e)/g + 1 rows) stays in HBM/L2 as signed 16-bit rows; rows and the LDS arrays are
//     null-filled once per task and ranges only grow, so there are NO per-lane range predicates: whatever lies
//     outside a row's range reads as null; an unreachable score maps to a permanently null row;
//   * operands are software-pipelined: M-ring words 4 chunks ahead (raw dwords, sign-extended on use), LDS
//     words and the 8-byte sequence probe 1 chunk ahead, so every wait in the sweep is a counted one;
//   * per 64-diagonal chunk the HBM traffic drops from ~2.1 KB (five int32 row reads, three row writes) to
//     ~0.7 KB (two dword row reads, one 16-bit row write, 64 provenance bytes).
// Same provenance bytes, row table, backtrace and unpack as the generic kernel (its last tier).

__device__ __forceinline__ int dpp_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }

template <int CAP, int QCAP, int NW>
__global__ __launch_bounds__(NW * 64) void wfa_affine_kernel_v3(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int xs, int oes, int es, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, const int32_t* __restrict__ bound)
{
  __shared__ __attribute__((aligned(16))) int16_t s_I[CAP];
  __shared__ __attribute__((aligned(16))) int16_t s_D[CAP];
  __shared__ uint16_t s_q[NW][QCAP];
  __shared__ int s_mlo[64];
  __shared__ int s_mhi[64];
  __shared__ int s_misc[16];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // NW waves cooperate on ONE alignment: wave wv sweeps a contiguous 1/NW of each wavefront
  volatile lds_i16* LI = (volatile lds_i16*)&s_I[0];
  volatile lds_i16* LD = (volatile lds_i16*)&s_D[0];
  volatile lds_u16* queue = (volatile lds_u16*)&s_q[wv][0];
  volatile __attribute__((address_space(3))) int* misc = (volatile __attribute__((address_space(3))) int*)&s_misc[0];
  uint8_t* my = ws.base + (size_t)blockIdx.x * ws.stride;
  int16_t* ringM = (int16_t*)my;                          // rm rows + one permanently null row
  int16_t* nullrow = ringM + (size_t)ws.rm * ws.capa;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  constexpr int NUL16 = -32768;

  for (;;) {
    if (wv == 0) misc[0] = (int)otg_wave_atomic_add(ticket, 1u);
    __syncthreads();
    const uint32_t tk = (uint32_t)misc[0];
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = t.pattern_end_free, tef = t.text_end_free;
    const int kb = pl + 4;                        // ring index of diagonal k (4 entries of slack on the left)
    const int kend = tl - pl;
    // Score bound U (units of g; INT_MAX = none).  A cell (s, k) of any component can only lie on an alignment of
    // total score <= U if it can still reach an end diagonal in [elo, ehi] with the remaining budget, and changing
    // the diagonal by one costs at least one gap extension (es == 1 here): dist(k, [elo, ehi]) <= U - s.  Cells
    // outside are skipped (read as null).  This cannot change the result: every cell on the optimal path, and
    // every predecessor that attains (or ties) the maximum of such a cell, lies itself on an alignment of score
    // <= the optimum <= U and is therefore kept with its true value; dropped cells only lower values that were
    // not the maximum.  So offsets, provenance bits on the path, the end diagonal and the score are unchanged.
    const int U = bound ? bound[ti] : 0x7fffffff;
    const bool bounded = U < 0x40000000;
    const int elo = kend - (ef ? tef : 0), ehi = kend + (ef ? pef : 0);
    bool fail = (pl + tl + 12 > ws.capa) || pl >= 32766 || tl >= 32766 || es != 1;
    size_t slab_top = 0;
    int s_end = -1, k_end = 0;
    int idlo = 1, idhi = 0;                       // range of the I/D wavefronts of the previous score (null)
    int kbase = 0;
    if (!fail) {
      // null-fill: LDS I/D arrays, the ring rows this task can touch, the null row
      volatile lds_u32* li32 = (volatile lds_u32*)&s_I[0];
      volatile lds_u32* ld32 = (volatile lds_u32*)&s_D[0];
      for (int q = (int)threadIdx.x; q < CAP / 2; q += NW * 64) { li32[q] = 0x80008000u; ld32[q] = 0x80008000u; }
      const int nfill = (pl + tl + 12 + 1) / 2;   // dwords per row
      uint32_t* r32 = (uint32_t*)ringM;
      const int row_dw = ws.capa / 2;
      for (int r = 0; r <= ws.rm; ++r) for (int q = (int)threadIdx.x; q < nfill; q += NW * 64) r32[(size_t)r * row_dw + q] = 0x80008000u;
    }
    __syncthreads();

    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      const int sm = s % ws.rm;
      int lo, hi;
      const int16_t *MxP = nullrow, *MoP = nullrow;
      int mxlo = 1, mxhi = 0, molo = 1, mohi = 0;
      if (s == 0) {
        lo = ef ? imax(-t.pattern_begin_free, -pl) : 0;
        hi = ef ? imin(t.text_begin_free, tl) : 0;
        if (bounded) {
          // the LDS window is centred on the diamond [ (lo0+elo-U)/2, (hi0+ehi+U)/2 ] that all kept cells live in
          const int dlo = imax(lo - U, (lo + elo - U) >> 1), dhi = imin(hi + U, (hi + ehi + U + 1) >> 1);
          lo = imax(lo, elo - U); hi = imin(hi, ehi + U);
          if (hi < lo) { fail = true; break; }
          kbase = ((dlo + dhi) >> 1) - (CAP >> 1);
          if (lo - kbase < 2 || hi - kbase + 132 >= CAP) kbase = lo - ((CAP - (hi - lo + 1)) >> 1);
        } else kbase = lo - ((CAP - (hi - lo + 1)) >> 1);
        if (hi - lo + 140 > CAP) { fail = true; break; }
      } else {
        lo = 1 << 30; hi = -(1 << 30);
        if (s - xs >= 0) { const int q = (s - xs) % ws.rm; mxlo = s_mlo[q]; mxhi = s_mhi[q]; if (mxhi >= mxlo) MxP = ringM + (size_t)q * ws.capa; }
        if (s - oes >= 0) { const int q = (s - oes) % ws.rm; molo = s_mlo[q]; mohi = s_mhi[q]; if (mohi >= molo) MoP = ringM + (size_t)q * ws.capa; }
        if (mxhi >= mxlo) { lo = imin(lo, mxlo); hi = imax(hi, mxhi); }
        if (mohi >= molo) { lo = imin(lo, molo - 1); hi = imax(hi, mohi + 1); }
        if (idhi >= idlo) { lo = imin(lo, idlo - 1); hi = imax(hi, idhi + 1); }
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
        if (bounded && hi >= lo) {
          const int room = U - s;
          lo = imax(lo, elo - room); hi = imin(hi, ehi + room);
          if (room < 0 || hi < lo) { fail = true; break; }     // cannot happen with a valid bound: next tier decides
        }
      }
      if (hi < lo) {   // null wavefront (I/D of the previous score are null too, see header)
        if (wv == 0) { s_mlo[sm] = 1; s_mhi[sm] = 0; rowtab[s] = -1; }
        idlo = 1; idhi = 0;
        __syncthreads();
        if (s > 2 * (oes + es * (pl + tl)) + 8) fail = true;
        continue;
      }
      if (lo - kbase < 2 || hi - kbase + 132 >= CAP) { fail = true; break; }     // LDS window exhausted -> next tier
      const int width = hi - lo + 1;
      if (slab_top + (size_t)width > ws.slab_bytes) { fail = true; break; }
      uint8_t* btrow = slab + slab_top - lo;
      // this wave's share of the wavefront: chunks [c0, c1)
      const int nch = (width + 63) >> 6;
      const int c0 = lo + 64 * ((nch * wv) / NW), c1 = lo + 64 * ((nch * (wv + 1)) / NW);
      // values at the share boundaries that a neighbouring wave overwrites during its own sweep
      // (with a score bound the previous wavefronts can be WIDER than this one, so the entries just outside
      // [lo, hi] are read like any other; without one they are null-filled and never written)
      const int bI = (int)LI[c0 - 1 - kbase];                                  // I[s-1][c0-1]
      const int bD = (int)LD[c1 - kbase];                                      // D[s-1][c1]
      __syncthreads();
      if (wv == 0) { rowtab[s] = (int64_t)slab_top - lo; s_mlo[sm] = lo; s_mhi[sm] = hi; misc[1] = 0; }
      slab_top += (size_t)width;
      int16_t* Mc = ringM + (size_t)sm * ws.capa;
      bool done = false;
      int qn = 0;
      // drain: queued diagonals (probe matched all 8 bytes).  Pass 1 looks 16 bytes ahead; survivors are in long
      // match runs: 64 bytes per pass, and once <= 4 diagonals remain the whole wave extends them one at a time
      // (512 bytes per iteration) — TR reads have exact runs of hundreds of bases.
      auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int e = 0; e < qn; ++e) {
              const int kk = lo + (int)queue[e];
              const int h = Mc[kk + kb];
              const int v = h - kk;
              const int m = otg_wave_match(P, T, v, h, imin(pl - v, tl - h), lane);
              Mc[kk + kb] = (int16_t)(h + m);
            }
            qn = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            break;
          }
          int wq = 0;
          for (int q0 = 0; q0 < qn; q0 += 64) {
            const bool act = q0 + lane < qn;
            int kk = 0, h = 0, v = 0;
            bool more = false;
            if (act) {
              kk = lo + (int)queue[q0 + lane];
              h = Mc[kk + kb];
              v = h - kk;
              const int rem = imin(pl - v, tl - h);
              int m, full;
              if (pass == 0) {
                const uint64_t xl = otg_load8(P + v) ^ otg_load8(T + h), xh = otg_load8(P + v + 8) ^ otg_load8(T + h + 8);
                m = xl ? (__builtin_ctzll(xl) >> 3) : (xh ? 8 + (__builtin_ctzll(xh) >> 3) : 16);
                m = imin(m, rem); full = 16;
              } else { m = otg_match64(P, T, v, h, rem); full = 64; }
              v += m; h += m;
              more = (m == full) && v < pl && h < tl;
              Mc[kk + kb] = (int16_t)h;
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint16_t)(kk - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
      };
      // raw 32-bit words (two adjacent 16-bit offsets) are queued untouched: any conversion right after the
      // load would force a wait for it.  Index k + kb - 1 .. : w0 = {Mo[k-1] | Mo[k]} is not used; we load
      // {Mo[k], Mo[k+1]} and take Mo[k-1] from the left lane (DPP) like the reference recurrence needs.
      auto load_m = [&](int c, uint32_t& w0, uint32_t& w1) {
        const int jc = imin(c + lane + kb, ws.capa - 2);
        __builtin_memcpy(&w0, MoP + jc, 4);      // Mo[k], Mo[k+1]
        __builtin_memcpy(&w1, MxP + jc, 4);      // Mx[k] (low half)
      };
      constexpr int PF = 4;
      uint32_t q_w0[PF], q_w1[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) load_m(c0 + 64 * u, q_w0[u], q_w1[u]);
      // LDS operands one chunk ahead (the current chunk only overwrites its own 64 entries)
      int n_iold = LI[c0 + lane - kbase], n_dx = LD[c0 + lane - kbase + 1];
      int carryI = bI;
      int carryMo = (int)MoP[c0 - 1 + kb];
      // software pipeline: the 8-byte sequence probe of chunk c is issued in iteration c and consumed in
      // iteration c+1, so its latency overlaps the LDS/compute work of the next chunk
      bool p_pending = false, p_in = false, p_valid = false, p_probe = false;
      int p_k = 0, p_h = 0, p_v = 0;
      uint64_t p_a = 0, p_b = 0;
      auto finish = [&]() {
        int h = p_h, v = p_v;
        bool more = false;
        if (p_probe) {
          const uint64_t xx = p_a ^ p_b;
          int m = xx ? (__builtin_ctzll(xx) >> 3) : 8;
          const int rem = imin(pl - v, tl - h);
          m = imin(m, rem);
          v += m; h += m;
          more = (m == 8) && v < pl && h < tl;
        }
        if (p_in) Mc[p_k + kb] = (int16_t)(p_valid ? h : NUL16);
        const unsigned long long mq = __ballot(more);
        if (more) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
          queue[qn + rank] = (uint16_t)(p_k - lo);
        }
        qn += __builtin_popcountll(mq);
      };
      for (int c = c0; c < c1; c += 64) {
        const int k = c + lane;
        const int jl = k - kbase;
        const bool in = k <= hi;
        const uint32_t w0 = q_w0[0], w1 = q_w1[0];
#pragma unroll
        for (int u = 0; u + 1 < PF; ++u) { q_w0[u] = q_w0[u + 1]; q_w1[u] = q_w1[u + 1]; }
        load_m(c + 64 * PF, q_w0[PF - 1], q_w1[PF - 1]);      // clamped address: always legal
        const int iold = n_iold;                              // I[s-1][k]
        int dx = n_dx;                                        // D[s-1][k+1]
        if (lane == 63 && c + 64 >= c1) dx = bD;              // first diagonal of the next wave's share
        n_iold = LI[jl + 64]; n_dx = LD[jl + 65];
        const int mo = (int)(int16_t)(w0 & 0xFFFFu);          // M[s-o-e][k]
        const int dop = (int)w0 >> 16;                        // M[s-o-e][k+1]
        const int mm = (int)(int16_t)(w1 & 0xFFFFu);          // M[s-x][k]
        int ix = dpp_shr1(iold);                              // I[s-1][k-1]
        if (lane == 0) ix = carryI;
        carryI = __builtin_amdgcn_readlane(iold, 63);
        int io = dpp_shr1(mo);                                // M[s-o-e][k-1]
        if (lane == 0) io = carryMo;
        carryMo = __builtin_amdgcn_readlane(mo, 63);
        uint32_t bits = 0;
        int ins, del;
        if (ix >= io) { ins = ix; bits |= 4u; } else ins = io;
        ins += 1;
        if (dx >= dop) { del = dx; bits |= 8u; } else del = dop;
        const int mis = mm + 1;
        int mx = imax(del, imax(mis, ins));
        uint32_t org = 0;
        if (mx == ins) org = 2;
        if (mx == del) org = 1;
        if (mx == mis) org = 0;
        bits |= org;
        if (s == 0) { mx = k > 0 ? k : 0; ins = NUL16; del = NUL16; bits = 0; }   // selects, not a branch
        const int h = mx, v = mx - k;
        const bool valid = in && mx >= 0 && h <= tl && v <= pl;
        const bool probe = valid && v < pl && h < tl;
        // retire the previous chunk first: its probe was issued one iteration ago, only this iteration's
        // prefetch loads are younger, so the wait is a counted vmcnt and not a drain
        if (p_pending) finish();
        uint64_t a, b;
        { const int vc = imin(imax(v, 0), pl), hc = imin(imax(h, 0), tl);      // clamped: always inside arena + slack
          a = otg_load8(P + vc); b = otg_load8(T + hc); }
        if (in) {
          LI[jl] = (int16_t)(ins < 0 ? NUL16 : ins);
          LD[jl] = (int16_t)(del < 0 ? NUL16 : del);
          btrow[k] = (uint8_t)bits;
        }
        p_pending = true; p_in = in; p_valid = valid; p_probe = probe; p_k = k; p_h = h; p_v = v; p_a = a; p_b = b;
        if (qn + 128 > QCAP) { finish(); p_pending = false; drain(); }
      }
      if (p_pending) finish();
      drain();
      idlo = s == 0 ? 1 : lo; idhi = s == 0 ? 0 : hi;
      // termination: the first diagonal (ascending) whose fully extended offset satisfies the end condition
      {
        int cand = 0x7fffffff;
        if (!ef) {
          if (kend >= c0 && kend < c1 && kend <= hi) { const int x = Mc[kend + kb]; if (x >= tl) cand = kend; }
        } else {
          for (int c = c0; c < c1 && cand == 0x7fffffff; c += 64) {
            const int k = c + lane;
            bool fin = false;
            if (k <= hi) {
              const int h = Mc[k + kb];
              const int v = h - k;
              fin = h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
            }
            const unsigned long long fm = __ballot(fin);
            if (fm) cand = c + (int)__builtin_ctzll(fm);
          }
        }
        misc[4 + wv] = cand;
      }
      __syncthreads();
      {
        int best = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int x = misc[4 + w]; best = x < best ? x : best; }
        if (best != 0x7fffffff) { done = true; s_end = s; k_end = best; }
      }
      if (done) break;
    }
    __syncthreads();

    if (wv != 0) continue;            // wave 0 reports / unpacks; the others wait at the next ticket barrier
    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }
    if (!backtrace_unpack<false>(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_I[0], EqBytes{P, T})) continue;
    if (cells) cells[ti] = affine_cells(t, xs, oes, s_end);
    if (ws.visited && lane == 0) atomicAdd(ws.visited, (unsigned long long)slab_top);
  }
}

int gcd3(int a, int b, int c)
{
  auto g2 = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
  return g2(g2(a, b), c);
}

// ---- counting sort of the alignment list by (tier, score bound), largest bound first inside a tier: the exact pass of an alignment costs
// ~ bound^2, and the persistent tier kernels hand alignments out in list order, so the longest run first and the tail of each kernel is
// short ones.  Which tier takes an alignment follows from its score bound and shape alone — the same window arithmetic as the kernels
// (affine_window) — so the one sort hands every tier its own list.
constexpr int ASORT_BUCKETS = 512;
__device__ __forceinline__ int asort_bucket(int U)
{
  const int b = (U < 0 || U >= 0x40000000) ? ASORT_BUCKETS - 1 : (U >> 3);
  return ASORT_BUCKETS - 1 - (b < ASORT_BUCKETS ? b : ASORT_BUCKETS - 1);
}
constexpr int TSORT_BUCKETS = (OTG_REG_TIERS + 1) * ASORT_BUCKETS;       // last tier = everything else (HBM-row tiers, generic kernel)
__device__ __forceinline__ int reg_tier(const otg_align_task& t, int U, int mask)
{
  const int pl = (int)t.pattern_len, tl = (int)t.text_len;
  if (U < 0 || U >= 0x40000000 || pl >= 32766 || tl >= 32766) return OTG_REG_TIERS;
  int kbase, need, lo0, hi0;
  if (!affine_window(t, U, &kbase, &need, &lo0, &hi0)) return OTG_REG_TIERS;
  const int seqb = ((pl + 15) / 16 + 3 + (tl + 15) / 16 + 3) * 4;
  if ((mask & 1) && need < OTG_REG_CAP[0] && seqb <= OTG_REG_SEQB[0]) return 0;
  if ((mask & 2) && need < OTG_REG_CAP[1] && seqb <= OTG_REG_SEQB[1]) return 1;
  if ((mask & 4) && need < OTG_REG_CAP[2] && seqb <= OTG_REG_SEQB[2]) return 2;
  // the multi-wave tiers take what lies beyond the smaller windows (those stay with the one-wave tiers, which are faster on narrow rows)
  if ((mask & 8) && need < OTG_REG_CAP[3] && seqb <= OTG_REG_SEQB[3]) return 3;
  if ((mask & 16) && need < OTG_REG_CAP[4] && seqb <= OTG_REG_SEQB[4]) return 4;
  return OTG_REG_TIERS;
}
__device__ __forceinline__ int tsort_bucket(const otg_align_task& t, int U, int mask) { return reg_tier(t, U, mask) * ASORT_BUCKETS + asort_bucket(U); }
__global__ __launch_bounds__(256) void K_tsort_hist(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, uint32_t n_imm,
                                                    const otg_align_task* __restrict__ tasks, const int32_t* __restrict__ bound, uint32_t* __restrict__ hist, int mask)
{
  __shared__ uint32_t h[TSORT_BUCKETS];
  for (int b = (int)threadIdx.x; b < TSORT_BUCKETS; b += 256) h[b] = 0;
  __syncthreads();
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t ti = list ? list[i] : i; atomicAdd(&h[tsort_bucket(tasks[ti], bound[ti], mask)], 1u); }
  __syncthreads();
  for (int b = (int)threadIdx.x; b < TSORT_BUCKETS; b += 256) if (h[b]) atomicAdd(&hist[b], h[b]);
}
// exclusive scan of the bucket counts (one block) + the tier segment bounds seg[0 .. OTG_REG_TIERS + 1]
__global__ __launch_bounds__(1024) void K_tsort_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ seg)
{
  __shared__ uint32_t part[1024];
  constexpr int PER = (TSORT_BUCKETS + 1023) / 1024;
  const int t = (int)threadIdx.x;
  uint32_t v[PER], s = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { const int b = t * PER + j; v[j] = b < TSORT_BUCKETS ? hist[b] : 0u; s += v[j]; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t x = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  uint32_t acc = part[t] - s;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = t * PER + j;
    if (b < TSORT_BUCKETS) { hist[b] = acc; if (b % ASORT_BUCKETS == 0) seg[b / ASORT_BUCKETS] = acc; }
    acc += v[j];
  }
  if (t == 1023) seg[OTG_REG_TIERS + 1] = part[1023];
}
__global__ __launch_bounds__(256) void K_tsort_scatter(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, uint32_t n_imm,
                                                       const otg_align_task* __restrict__ tasks, const int32_t* __restrict__ bound,
                                                       uint32_t* __restrict__ pos, uint32_t* __restrict__ out, int mask)
{
  __shared__ uint32_t cnt[TSORT_BUCKETS], basep[TSORT_BUCKETS];
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (int b = (int)threadIdx.x; b < TSORT_BUCKETS; b += 256) cnt[b] = 0;
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) { const uint32_t ti = list ? list[i] : i; atomicAdd(&cnt[tsort_bucket(tasks[ti], bound[ti], mask)], 1u); }
  __syncthreads();
  for (int b = (int)threadIdx.x; b < TSORT_BUCKETS; b += 256) { basep[b] = cnt[b] ? atomicAdd(&pos[b], cnt[b]) : 0u; cnt[b] = 0; }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t ti = list ? list[i] : i;
    const int b = tsort_bucket(tasks[ti], bound[ti], mask);
    out[basep[b] + atomicAdd(&cnt[b], 1u)] = ti;
  }
}
// the last segment (alignments no register tier takes) opens the list the LDS / HBM tiers work on; the register tiers append what they give up
__global__ void K_seg_copy(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ seg, uint32_t* __restrict__ out, uint32_t* __restrict__ n_out)
{
  const uint32_t a = seg[0], n = seg[1] - seg[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = sorted[a + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n;
}

} // namespace

int otg_launch_affine(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, uint32_t n_tasks,
                      int x, int o, int e, int32_t* d_scores, const uint64_t* d_cig_off, uint32_t* d_cig_len,
                      uint8_t* d_cig_arena, uint64_t* d_cells)
{
  return otg_launch_affine_todo(ctx, d_arena, d_tasks, nullptr, nullptr, n_tasks, x, o, e, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells);
}

int otg_launch_affine_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                           const uint32_t* d_n_todo, uint32_t n_tasks, int x, int o, int e, int32_t* d_scores,
                           const uint64_t* d_cig_off, uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells,
                           float* kernel_ms, uint64_t* launches)
{
  if (n_tasks == 0) return OTG_OK;
  if (ctx->heur_strategy == OTG_HEURISTIC_WFADAPTIVE)
    return otg_launch_affine_adaptive_todo(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, x, o, e, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells,
                                           kernel_ms, launches);
  if (x <= 0 || e <= 0 || o < 0) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties must satisfy x>0, o>=0, e>0");
  const int g = gcd3(x, o + e, e);
  const int xs = x / g, oes = (o + e) / g, es = e / g;
  if (std::max(xs, oes) + 1 > 64 || es + 1 > 64) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties too large after gcd reduction");
  const bool fresh_cnt = ctx->pool[SLOT_COUNTERS].cap < OTG_COUNTER_WORDS * sizeof(uint32_t);
  uint32_t* cnt = (uint32_t*)otg_slot(ctx, SLOT_COUNTERS, OTG_COUNTER_WORDS * sizeof(uint32_t));
  uint32_t* todo = (uint32_t*)otg_slot(ctx, SLOT_TODO, 4 * (size_t)n_tasks * sizeof(uint32_t));
  if (!cnt || !todo) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(cnt + 8, 0, 8 * sizeof(uint32_t), ctx->stream));        // tickets / overflow counters of the tiers behind the register tiers
  HIP_TRY(ctx, hipMemsetAsync(cnt + 64, 0, 16 * sizeof(uint32_t), ctx->stream));      // register tiers: segment bounds, overflow count, tickets
  // visited-cell counter of the exact tiers (accumulates over the launches of a run; otg_assemble_run zeroes it)
  if (fresh_cnt || !ctx->affine_visited) { ctx->affine_visited = (unsigned long long*)(cnt + 96); HIP_TRY(ctx, hipMemsetAsync(cnt + 96, 0, 8, ctx->stream)); }

  // workspace sizes must not follow the batch: every change of size is a hipFree + hipMalloc of gigabytes, and batches of one job differ in
  // their longest read and their task count — the longest read is rounded up to 4 kb steps and the grids are sized for a full device
  const size_t maxlen = ((size_t)ctx->max_seq_len + 4095) & ~(size_t)4095;
  AffWs ws;
  ws.capa = (int)(2 * maxlen + 16) & ~1;
  ws.rm = std::max(xs, oes) + 1;
  ws.ri = es + 1;
  ws.nrows = (int)(2 * (size_t)oes + (size_t)es * 2 * maxlen + 16);
  ws.rev_cap = 4 * maxlen + 64;
  ws.dbg = getenv("OTG_DEBUG") != nullptr ? 1 : 0;
  ws.visited = ctx->affine_visited;
  size_t ring_bytes = (size_t)(ws.rm + 2 * ws.ri) * ws.capa * sizeof(int32_t);
  ws.off_rowtab = (ring_bytes + 255) & ~(size_t)255;
  ws.off_rev = (ws.off_rowtab + (size_t)ws.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
  ws.off_slab = (ws.off_rev + ws.rev_cap + 255) & ~(size_t)255;

  static const bool no_v3 = getenv("OTG_NO_AFFINE_V3") != nullptr;            // generic kernel only (test switch)
  static const bool no_bound = getenv("OTG_NO_AFFINE_BOUND") != nullptr;      // no score bound: the HBM-row tiers without pruning (test switch)
  // register tiers that run (bit mask, test switch OTG_AFFINE_REG: 1 / 2 / 4 = the one-wave tiers of 1024 / 1536 / 2048 diagonals, 8 = the four-wave
  // tier of 4096, 16 = the eight-wave tier of 8192 — that one only when the batch can need it: its slabs are the largest)
  static const int reg_mask_env = getenv("OTG_AFFINE_REG") ? atoi(getenv("OTG_AFFINE_REG")) : 31;
  // measurement switch: which instantiation a window runs on, one digit per tier (wfa_affine_reg.hip; 0 = default)
  static const int shape_env = getenv("OTG_REG_SHAPE") ? atoi(getenv("OTG_REG_SHAPE")) : 0;
  const int shape[OTG_REG_TIERS] = {(shape_env / 10000) % 10, (shape_env / 1000) % 10, (shape_env / 100) % 10, (shape_env / 10) % 10, shape_env % 10};
  const bool bounded = !no_v3 && es == 1 && xs == 2 && oes == 4 && !no_bound;
  int reg_mask = !bounded ? 0 : (maxlen <= 4096 ? (reg_mask_env & ~16) : reg_mask_env) & 31;

  // ---- workspaces.  Contexts that share a device size theirs from what is free: one at a time (otg_device_mutex).
  constexpr int NWA = 4;                     // waves cooperating on one alignment in the HBM-row tiers
  constexpr int WPB = 4;                     // alignments (waves) per block of the generic kernel
  AffWs wsA = ws, wsB = ws, wsC = ws, wr[OTG_REG_TIERS];
  uint32_t wavesA = 0, wavesB = 0, gridC = 2, blocks[OTG_REG_TIERS] = {0, 0, 0, 0, 0};
  const uint32_t ncu = (uint32_t)ctx->n_cu;
  {
    std::lock_guard<std::mutex> alloc_lock(otg_device_mutex(ctx->device));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    // a fixed share of the device's memory (not of what happens to be free: several contexts share a device in the dispatcher, and a budget
    // that follows the other contexts' allocations would resize this workspace batch after batch)
    const size_t budget = std::min<size_t>((size_t)(total_b * 0.2), (size_t)((free_b + ctx->pool[SLOT_WF_WS].cap) * 0.8));
    // Provenance slabs of the HBM-row tiers: what a diamond of the tier's window can hold, not more.  Workspaces are kept small on purpose:
    // the first launch on a fresh allocation pays for every gigabyte (2.98 s for 52 GB measured, scripts/cold_probe.py), and the
    // dispatcher starts fresh contexts per job.
    auto fit = [&](uint32_t& nwaves, size_t& slab) {
      while (nwaves > 1 && (ws.off_slab + slab) * (size_t)nwaves > budget) {
        if (slab > ((size_t)4 << 20)) slab /= 2; else nwaves = (nwaves + 1) / 2;
      }
    };
    // a diamond of bound U holds ~U^2 / 2 cells and U stays below ~0.45 x length at ONT divergence: 0.2 x maxlen^2 is twice that; an alignment
    // that needs more moves on to the next tier
    auto diamond_slab = [&](size_t window) { return std::min<size_t>((size_t)(0.2 * (double)maxlen * (double)maxlen), window * window * 5 / 8) + (1 << 16); };
    // tier A: HBM rows, LDS window of 4096 diagonals; tier B: 12288 diagonals (the longest reads of a 1-10 kb job without a usable bound)
    wavesA = ncu * 2; size_t slabA = diamond_slab(4096);
    fit(wavesA, slabA);
    wsA.slab_bytes = slabA & ~(size_t)255; wsA.stride = wsA.off_slab + wsA.slab_bytes;
    wavesB = maxlen > 8192 ? ncu : ncu / 2; size_t slabB = diamond_slab(12288);
    fit(wavesB, slabB);
    wsB.slab_bytes = slabB & ~(size_t)255; wsB.stride = wsB.off_slab + wsB.slab_bytes;
    // tier C: generic kernel (global int32 rings), a few waves with the largest useful slabs
    size_t slabC = std::min<size_t>((size_t)2 * maxlen * (size_t)(ws.nrows), budget / (gridC * WPB));
    static const size_t slab_cap_env = getenv("OTG_AFFINE_LAST_SLAB_MB") ? (size_t)atoi(getenv("OTG_AFFINE_LAST_SLAB_MB")) << 20 : 0;      // test switch: a last-resort tier that runs out
    if (slab_cap_env) slabC = std::min(slabC, slab_cap_env);
    if (slabC > ws.off_slab + 256) slabC -= ws.off_slab + 256;
    wsC.slab_bytes = slabC & ~(size_t)255; wsC.stride = wsC.off_slab + wsC.slab_bytes;
    const size_t need = std::max(std::max(wsA.stride * wavesA, wsB.stride * wavesB), wsC.stride * (size_t)gridC * WPB);
    uint8_t* wsp = (uint8_t*)otg_slot(ctx, SLOT_WF_WS, need);
    if (!wsp) return OTG_ERR_HIP;
    wsA.base = wsB.base = wsC.base = wsp;

    if (reg_mask) {
      // the register tiers' workspaces lie side by side (the tiers run back to back on the stream; nothing else touches them): row table sized
      // by the window (a score is a row), slab for the largest diamond the window admits
      HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));                 // after the allocation above
      const size_t budget_r = std::min<size_t>((size_t)(total_b * 0.25), (size_t)((free_b + ctx->pool[SLOT_REVOPS].cap) * 0.8));
      uint32_t al[OTG_REG_TIERS];
      for (int t = 0; t < OTG_REG_TIERS; ++t) {
        AffWs w = ws;
        const size_t cap = (size_t)OTG_REG_CAP[t];
        w.nrows = OTG_REG_CAP[t] + 64;
        w.off_rowtab = 0;
        w.off_rev = ((size_t)w.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
        w.off_slab = (w.off_rev + ws.rev_cap + 255) & ~(size_t)255;
        w.slab_bytes = (cap * cap * 7 / 8 + (1 << 16)) & ~(size_t)255;
        if (t == 4) w.slab_bytes = std::min<size_t>(w.slab_bytes, ((size_t)(0.2 * (double)maxlen * (double)maxlen) + (1 << 20)) & ~(size_t)255);
        w.stride = w.off_slab + w.slab_bytes;
        wr[t] = w;
        int apb = 1, bpc = 1;
        otg_affine_reg_geometry(t, shape[t], &apb, &bpc);
        blocks[t] = (reg_mask >> t) & 1 ? ncu * (uint32_t)bpc : 0u;
        al[t] = blocks[t] * (uint32_t)apb;
      }
      auto total = [&]() { size_t s = 256; for (int t = 0; t < OTG_REG_TIERS; ++t) s += wr[t].stride * al[t]; return s; };
      // beyond the budget: fewer alignments in flight in the two widest tiers first (their slabs are the largest), then whole tiers, widest
      // first — their alignments then run in the HBM-row tiers
      while (total() > budget_r) {
        int t = -1;
        if (blocks[4] > ncu / 4 && (wr[4].stride * al[4] >= wr[3].stride * al[3] || blocks[3] <= ncu / 2)) t = 4;
        else if (blocks[3] > ncu / 2) t = 3;
        if (t >= 0) { blocks[t] /= 2; al[t] = blocks[t]; continue; }
        for (t = OTG_REG_TIERS - 1; t >= 0 && !blocks[t]; --t) {}
        if (t < 0) break;
        blocks[t] = 0; al[t] = 0; reg_mask &= ~(1 << t);
      }
      uint8_t* wsr = reg_mask ? (uint8_t*)otg_slot(ctx, SLOT_REVOPS, total()) : nullptr;
      if (reg_mask && !wsr) {                       // the device could not give that much after all: the chain works without the register tiers
        (void)hipGetLastError();
        reg_mask = 0;
      }
      if (ws.dbg) fprintf(stderr, "[otg] affine: register tiers keep %u / %u / %u / %u / %u alignments in flight, workspaces %.1f GB of a budget of %.1f GB, mask %d\n",
                          al[0], al[1], al[2], al[3], al[4], (double)total() / 1e9, (double)budget_r / 1e9, reg_mask);
      uint8_t* p = wsr;
      for (int t = 0; t < OTG_REG_TIERS; ++t) { wr[t].base = p; if (p) p += wr[t].stride * al[t]; }
    }
  }

  uint32_t* listA = todo;                               // what tier A gives up
  uint32_t* listB = todo + n_tasks;                     // what tier B gives up
  uint32_t* sorted = todo + 2 * (size_t)n_tasks;        // the counting sort's output: one segment per register tier + the rest
  uint32_t* ovf_r = todo + 3 * (size_t)n_tasks;         // the rest + what the register tiers give up: the input of tier A
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const uint32_t* cur = d_todo; const uint32_t* cur_n = d_n_todo; uint32_t cur_imm = n_tasks;
  if (!no_v3 && es == 1) {
    int32_t* d_bound = nullptr;
    if (bounded) {
      d_bound = (int32_t*)otg_slot(ctx, SLOT_BT_POOL, (size_t)n_tasks * sizeof(int32_t));
      if (!d_bound) return OTG_ERR_HIP;
      const uint32_t gridU = std::min<uint32_t>(ncu * 8, (n_tasks + 3) / 4);
      hipLaunchKernelGGL((wfa_affine_bound1_kernel<2, 4>), dim3(gridU), dim3(256), 0, ctx->stream, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_bound, cnt + 14);
    }
    const uint32_t* inA = d_todo; const uint32_t* inA_n = d_n_todo; uint32_t inA_imm = n_tasks;
    if (d_bound && reg_mask) {
      uint32_t* hist = (uint32_t*)otg_slot(ctx, SLOT_ROWTAB, TSORT_BUCKETS * sizeof(uint32_t));
      uint32_t* seg = cnt + 64;               // seg[0 .. OTG_REG_TIERS + 1]
      uint32_t* n_ovf = cnt + 71;
      if (!hist) return OTG_ERR_HIP;
      HIP_TRY(ctx, hipMemsetAsync(hist, 0, TSORT_BUCKETS * sizeof(uint32_t), ctx->stream));
      const uint32_t sg = std::min<uint32_t>((n_tasks + 2047) / 2048, ncu * 2);
      hipLaunchKernelGGL(K_tsort_hist, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, d_tasks, (const int32_t*)d_bound, hist, reg_mask);
      hipLaunchKernelGGL(K_tsort_scan, dim3(1), dim3(1024), 0, ctx->stream, hist, seg);
      hipLaunchKernelGGL(K_tsort_scatter, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, d_tasks, (const int32_t*)d_bound, hist, sorted, reg_mask);
      hipLaunchKernelGGL(K_seg_copy, dim3(std::min<uint32_t>((n_tasks + 255) / 256, 1024u)), dim3(256), 0, ctx->stream, (const uint32_t*)sorted,
                         (const uint32_t*)(seg + OTG_REG_TIERS), ovf_r, n_ovf);
      // Small batches: the tiers next to each other on side streams (disjoint lists, workspaces and tickets; what they give up goes to one list
      // through an atomic counter) — each tier alone would leave most of the device idle and still last as long as its longest alignment.
      // Large batches: one after the other (side by side the tiers' blocks share CUs and the chain takes 12 % longer, measured).
      static const int conc_env = getenv("OTG_AFFINE_CONCURRENT") ? atoi(getenv("OTG_AFFINE_CONCURRENT")) : -1;
      const bool concurrent = conc_env >= 0 ? conc_env != 0 : n_tasks <= 50000u;      // measured: 6 250 alignments 44 -> 30 ms, 25 000: 67 -> 59 ms, 100 000: 194 -> 201 ms
      hipStream_t main_stream = ctx->stream;
      if (concurrent) {
        if (!ctx->ev_fork) {
          HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
          for (int t = 0; t < OTG_REG_TIERS; ++t) {
            HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->tier_stream[t], hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[t], hipEventDisableTiming));
          }
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, main_stream));
      }
      // Which stream a tier runs on when they run side by side.  HIP gives a process 4 hardware queues; with one stream per tier two pairs of tiers
      // shared a queue and ran one after the other (r03's timeline of a 1 000-region step: <1,12> started when <2,8> ended, <1,8> when <4,8>
      // ended), and a tier whose list is EMPTY still lasts as long as the persistent blocks of the others keep its own blocks from being
      // dispatched (<8,8>: 36 ms on a batch that gave it nothing) and holds up whatever shares its stream.  So on small batches the host reads the
      // seven segment bounds back (one small copy behind the bound pass and the sort, which it would otherwise only queue behind), launches the
      // tiers that have work, and deals them to THREE streams — main + two side streams — longest expected run first.
      int side_of[OTG_REG_TIERS] = {-2, -2, -2, -2, -2};         // -2: not launched, -1: the main stream, 0 / 1: a side stream
      int order[OTG_REG_TIERS] = {4, 3, 2, 1, 0};
      if (concurrent) {
        uint32_t hseg[OTG_REG_TIERS + 2];
        HIP_TRY(ctx, hipMemcpyAsync(hseg, seg, sizeof(hseg), hipMemcpyDeviceToHost, main_stream));
        HIP_TRY(ctx, hipStreamSynchronize(main_stream));
        static const double weight[OTG_REG_TIERS] = {1.0, 2.2, 4.0, 16.0, 64.0};      // per alignment, ~ window^2
        double cost[OTG_REG_TIERS], load[3] = {0.0, 0.0, 0.0};
        for (int t = 0; t < OTG_REG_TIERS; ++t) cost[t] = blocks[t] ? weight[t] * (double)(hseg[t + 1] - hseg[t]) : 0.0;
        std::sort(order, order + OTG_REG_TIERS, [&](int a, int b) { return cost[a] > cost[b]; });
        for (int q = 0; q < OTG_REG_TIERS; ++q) {
          const int t = order[q];
          if (cost[t] <= 0.0) continue;
          int best = 0;
          for (int j = 1; j < 3; ++j) if (load[j] < load[best]) best = j;
          load[best] += cost[t];
          side_of[t] = best - 1;
        }
      } else {
        for (int t = 0; t < OTG_REG_TIERS; ++t) side_of[t] = blocks[t] ? -1 : -2;
      }
      bool side_used[2] = {false, false};
      for (int q = 0; q < OTG_REG_TIERS; ++q) {
        const int t = order[q];
        if (side_of[t] == -2) continue;
        if (side_of[t] >= 0) {
          ctx->stream = ctx->tier_stream[side_of[t]];
          if (!side_used[side_of[t]]) { HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0)); side_used[side_of[t]] = true; }
        }
        const int rc = otg_launch_affine_reg_tier(ctx, t, shape[t], blocks[t], d_arena, d_tasks, sorted, seg + t, g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells,
                                                  cnt + 72 + t, n_ovf, ovf_r, wr[t], d_bound, ctx->affine_visited);
        ctx->stream = main_stream;
        if (rc) return rc;
      }
      for (int sd = 0; sd < 2; ++sd) if (side_used[sd]) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_join[sd], ctx->tier_stream[sd]));
        HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join[sd], 0));
      }
      inA = ovf_r; inA_n = n_ovf; inA_imm = 0;
    }
    hipLaunchKernelGGL((wfa_affine_kernel_v3<4096, 512, NWA>), dim3(wavesA), dim3(NWA * 64), 0, ctx->stream, d_arena, d_tasks,
                       inA, inA_n, inA_imm, xs, oes, es, g, d_scores, d_cig_off, d_cig_len,
                       d_cig_arena, d_cells, cnt + 8, cnt + 9, listA, wsA, (const int32_t*)d_bound);
    hipLaunchKernelGGL((wfa_affine_kernel_v3<12288, 512, NWA>), dim3(wavesB), dim3(NWA * 64), 0, ctx->stream, d_arena, d_tasks,
                       (const uint32_t*)listA, (const uint32_t*)(cnt + 9), 0u, xs, oes, es, g, d_scores, d_cig_off, d_cig_len,
                       d_cig_arena, d_cells, cnt + 10, cnt + 11, listB, wsB, (const int32_t*)d_bound);
    cur = listB; cur_n = cnt + 11; cur_imm = 0;
  }
  hipLaunchKernelGGL((wfa_affine_kernel<WPB>), dim3(gridC), dim3(WPB * 64), 0, ctx->stream, d_arena, d_tasks,
                     cur, cur_n, cur_imm, xs, oes, es, g, d_scores, d_cig_off, d_cig_len,
                     d_cig_arena, d_cells, cnt + 12, cnt + 13, (uint32_t*)nullptr, wsC);
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));      // after the LAST tier of the chain
  HIP_TRY(ctx, hipGetLastError());
  if (ws.dbg) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t h[16], h5[8];
    HIP_TRY(ctx, hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(h5, cnt + 64, sizeof(h5), hipMemcpyDeviceToHost));
    fprintf(stderr, "[otg] affine: register tiers take %u / %u / %u / %u / %u alignments, %u go to the HBM-row tiers (of which given up by a register tier: %u); tier A gives up %u, tier B %u\n",
            h5[1] - h5[0], h5[2] - h5[1], h5[3] - h5[2], h5[4] - h5[3], h5[5] - h5[4], h5[7], h5[7] - (h5[6] - h5[5]), h[9], h[11]);
    const int32_t* dbg_bound = (const int32_t*)ctx->pool[SLOT_BT_POOL].p;
    if (dbg_bound && bounded) {
      // How loose is the bound?  The exact tiers visit the diamond of the bound U (~U^2 / 2 cells); the diamond of the final score s would do.
      // Histogram of U / s and the share of diamond cells that lie outside the exact score's diamond (sum U^2 against sum s^2).
      std::vector<int32_t> hb(n_tasks), hs(n_tasks);
      HIP_TRY(ctx, hipMemcpy(hb.data(), dbg_bound, (size_t)n_tasks * 4, hipMemcpyDeviceToHost));
      HIP_TRY(ctx, hipMemcpy(hs.data(), d_scores, (size_t)n_tasks * 4, hipMemcpyDeviceToHost));
      uint32_t n_live = 0;
      std::vector<uint32_t> ids;
      if (d_todo) {
        if (d_n_todo) HIP_TRY(ctx, hipMemcpy(&n_live, d_n_todo, 4, hipMemcpyDeviceToHost)); else n_live = n_tasks;
        ids.resize(n_live);
        if (n_live) HIP_TRY(ctx, hipMemcpy(ids.data(), d_todo, (size_t)n_live * 4, hipMemcpyDeviceToHost));
      } else { n_live = n_tasks; ids.resize(n_live); for (uint32_t i = 0; i < n_live; ++i) ids[i] = i; }
      double su2 = 0, ss2 = 0; uint64_t hist[6] = {0, 0, 0, 0, 0, 0}, n_ok = 0;
      for (uint32_t q = 0; q < n_live; ++q) {
        const int32_t U = hb[ids[q]], sc = hs[ids[q]];
        if (U < 0 || U >= 0x40000000 || sc <= 0) continue;
        const double u = (double)U * g, r = u / (double)sc;       // the bound is in units of g, the score in penalty units
        su2 += u * u; ss2 += (double)sc * sc; ++n_ok;
        ++hist[r <= 1.0 ? 0 : r <= 1.02 ? 1 : r <= 1.05 ? 2 : r <= 1.1 ? 3 : r <= 1.2 ? 4 : 5];
      }
      if (n_ok) fprintf(stderr, "[otg] affine: bound / final score over %llu alignments: == 1: %.1f %%, <= 1.02: %.1f %%, <= 1.05: %.1f %%, <= 1.1: %.1f %%, <= 1.2: %.1f %%, above: %.1f %%; diamond cells outside the exact score's diamond: %.1f %%\n",
                        (unsigned long long)n_ok, 100.0 * hist[0] / n_ok, 100.0 * hist[1] / n_ok, 100.0 * hist[2] / n_ok, 100.0 * hist[3] / n_ok, 100.0 * hist[4] / n_ok, 100.0 * hist[5] / n_ok,
                        100.0 * (su2 - ss2) / su2);
    }
    if (h5[7] && dbg_bound && bounded && reg_mask) {        // who left a register tier (the first few): lengths, free ends, bound
      const uint32_t nshow = std::min<uint32_t>(h5[7], 24u);
      std::vector<uint32_t> ids(nshow);
      HIP_TRY(ctx, hipMemcpy(ids.data(), todo + 3 * (size_t)n_tasks, nshow * sizeof(uint32_t), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < nshow; ++i) {
        otg_align_task t; int32_t U = 0, sc = 0;
        HIP_TRY(ctx, hipMemcpy(&t, d_tasks + ids[i], sizeof(t), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(&U, dbg_bound + ids[i], sizeof(U), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(&sc, d_scores + ids[i], sizeof(sc), hipMemcpyDeviceToHost));
        fprintf(stderr, "[otg] affine:   left its register tier: task %u, pattern %u, text %u, ends-free %d (pattern begin / end %d / %d, text begin / end %d / %d), bound %d, score %d\n",
                ids[i], t.pattern_len, t.text_len, t.endsfree, t.pattern_begin_free, t.pattern_end_free, t.text_begin_free, t.text_end_free, U, sc);
      }
    }
#ifdef OTG_REG_TIMING
    unsigned long long tm[7];
    HIP_TRY(ctx, hipMemcpy(tm, cnt + 100, sizeof(tm), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemsetAsync(cnt + 100, 0, sizeof(tm), ctx->stream));
    if (tm[5]) fprintf(stderr, "[otg] register tiers, clock ticks per wave and score: preamble %.0f sweep %.0f (%.2f slot visits) drain+fold %.0f exports %.0f barrier %.0f; %llu wave-scores\n",
                       (double)tm[0] / tm[5], (double)tm[1] / tm[5], (double)tm[6] / tm[5], (double)tm[2] / tm[5], (double)tm[3] / tm[5], (double)tm[4] / tm[5], tm[5]);
#endif
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
