// wfa_affine.hip — batched gap-affine wavefront aligner with full op string (gfx950).
//
// Replaces wfa::WFAlignerGapAffine(x,o,e, Alignment, MemoryMed)::alignEnd2End / alignEndsFree +
// getAlignmentCigar() (reference: src/assemble.cpp:50; call sites src/analignments.cpp:25,31,37,268-280).
//
// One wave64 per alignment, persistent waves + device ticket.  Per wave, in HBM/L2:
//   * rings of wavefront rows (int32 offsets, index = k + plen + 1): M keeps max(x,o+e)/g + 1 rows,
//     I and D keep e/g + 1 rows (scores are walked in units of g = gcd(x, o+e, e): (4,6,2) -> 2,4,1);
//   * one provenance byte per (score, diagonal): bits0-1 M origin (0 mismatch, 1 deletion, 2 insertion),
//     bit2 I came from extension, bit3 D came from extension — the piggy-back rule of WFA2
//     (ext >= open; M provenance tested in the order ins, del, mism so
//     mismatch wins ties over deletion over insertion); rows are bump-allocated in a per-wave slab and
//     addressed through a per-wave row table;
//   * the reversed op list of the backtrace.
// After the forward pass the same wave walks the provenance back (uniform scalar walk), then unpacks the
// ops forward, re-deriving match runs 64 bytes at a time with a ballot (pcigar_unpack_affine semantics:
// matches are extended only in the M state; a gap close is a marker, not an op), and writes the op string
// (M X I D, free end gaps explicit).  A task whose provenance does not fit the tier-1 slab is queued on the
// device for tier 2 (few waves, large slabs).
#include "otg_common.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

__device__ unsigned long long otg_dbg_v4_cells[2];     // OTG_DEBUG: cells the LDS tiers visited / alignments they finished

struct AffWs {
  uint8_t* base;        // per-wave workspaces, contiguous
  size_t stride;        // bytes per wave
  size_t off_rowtab, off_rev, off_slab;
  size_t slab_bytes;
  int capa;             // diagonals per ring row
  int rm, ri;           // ring depths
  int nrows;            // row-table entries
  size_t rev_cap;
  int dbg;              // OTG_DEBUG: count visited cells
  unsigned long long* visited;   // device counter of visited (score, diagonal) cells, all exact tiers (nullable)
};

using lds_i16 = __attribute__((address_space(3))) int16_t;
using lds_u16 = __attribute__((address_space(3))) uint16_t;
using lds_u32 = __attribute__((address_space(3))) uint32_t;
using lds_u8 = __attribute__((address_space(3))) uint8_t;

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// Walks the provenance back from (s_end, k_end) and unpacks the op string (shared by every forward kernel).
// Uniform control flow: every lane follows the same path and stores the same bytes.
//
// The walk is a chain of dependent reads (row table entry -> provenance byte -> next row), one HBM round trip per step when read in
// place.  Instead the wave stages a WINDOW of the provenance in LDS: lane l fetches 32 bytes of row s0 - l around the current diagonal
// (64 rows in flight at once: two round trips per window), and the walk then reads LDS until it leaves the window — the score drops by
// 1, 2 or 4 per step and the diagonal moves by at most one, so a window lasts 16-64 steps.  `win` = 2 KB of LDS owned by this wave
// (the forward kernels hand over state they no longer need).  `eq(v, h)` compares pattern base v with text base h (from LDS where the
// kernel keeps the sequences there).
constexpr int BT_ROWS = 64, BT_COLS = 32;
template <class Eq>
__device__ bool backtrace_unpack(const uint8_t* P, int pl, const uint8_t* T, int tl, int s_end, int k_end, int xs, int oes, int es,
                                 const int64_t* rowtab, const uint8_t* slab, uint8_t* rev, size_t rev_cap, uint8_t* out, int lane,
                                 int32_t* score_out, uint32_t* len_out, int g, volatile lds_u32* win, Eq eq)
{
  uint32_t nrev = 0;
  int k0;
  {
    int s = s_end, k = k_end, comp = 0;
    int ws0 = -1, wk0 = 0;                       // the staged window: rows ws0 .. ws0 - 63, columns wk0 .. wk0 + 31
    volatile lds_u8* win8 = (volatile lds_u8*)win;
    while (s > 0 || comp != 0) {
      if (ws0 < 0 || ws0 - s >= BT_ROWS || s > ws0 || k < wk0 || k >= wk0 + BT_COLS) {
        ws0 = s; wk0 = k - BT_COLS / 2;
        const int r = ws0 - lane;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r >= 0) {
          const int64_t rb = rowtab[r];
          if (rb != -1) __builtin_memcpy(w, slab + rb + wk0, 32);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) win[lane * 8 + j] = w[j];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
      const uint32_t bits = win8[(ws0 - s) * BT_COLS + (k - wk0)];
      uint8_t op;
      if (comp == 0) {
        const uint32_t org = bits & 3u;
        if (org == 0) { op = 'X'; s -= xs; }
        else if (org == 1) { op = 'c'; comp = 2; }
        else { op = 'c'; comp = 1; }
      } else if (comp == 1) {
        op = 'I';
        if (bits & 4u) s -= es; else { s -= oes; comp = 0; }
        k -= 1;
      } else {
        op = 'D';
        if (bits & 8u) s -= es; else { s -= oes; comp = 0; }
        k += 1;
      }
      if (nrev >= rev_cap || s < 0) { *score_out = -2; *len_out = 0; return false; }
      rev[nrev] = op;
      ++nrev;
    }
    k0 = k;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  uint32_t pos = 0;
  int h = k0 > 0 ? k0 : 0, v = k0 < 0 ? -k0 : 0;
  for (int q = lane; q < h; q += 64) out[q] = 'I';
  pos += h;
  for (int q = lane; q < v; q += 64) out[pos + q] = 'D';
  pos += v;
  auto emit_matches = [&]() {
    for (;;) {
      const int rem = imin(pl - v, tl - h);
      if (rem <= 0) break;
      const int n = rem < 64 ? rem : 64;
      const bool same = lane < n && eq(v + lane, h + lane);
      const unsigned long long ne = ~__ballot(same);
      const int m = ne ? (int)__builtin_ctzll(ne) : 64;
      if (lane < m) out[pos + lane] = 'M';
      v += m; h += m; pos += m;
      if (m < 64) break;
    }
  };
  // the reversed op list is read back 64 ops at a time (one load per chunk instead of one dependent load per op)
  int state = 0;
  for (int q0 = (int)nrev - 1; q0 >= 0; q0 -= 64) {
    const int qi = q0 - lane;
    const int myop = qi >= 0 ? (int)rev[qi] : 0;
    const int nin = q0 + 1 < 64 ? q0 + 1 : 64;
    for (int j = 0; j < nin; ++j) {
      if (state == 0) emit_matches();
      const int op = __builtin_amdgcn_readlane(myop, j);
      if (op == 'I') { out[pos] = 'I'; ++pos; ++h; state = 1; }
      else if (op == 'D') { out[pos] = 'D'; ++pos; ++v; state = 2; }
      else if (op == 'c') { state = 0; }
      else { out[pos] = 'X'; ++pos; ++v; ++h; }
    }
  }
  emit_matches();
  { const int n = tl - h; for (int q = lane; q < n; q += 64) out[pos + q] = 'I'; if (n > 0) { pos += n; h = tl; } }
  { const int n = pl - v; for (int q = lane; q < n; q += 64) out[pos + q] = 'D'; if (n > 0) { pos += n; v = pl; } }
  *score_out = s_end * g;
  *len_out = pos;
  return true;
}
// base comparison straight from the byte sequences in HBM / L2 (kernels that do not keep the pair in LDS)
struct EqBytes {
  const uint8_t* P; const uint8_t* T;
  __device__ __forceinline__ bool operator()(int v, int h) const { return P[v] == T[h]; }
};
// base comparison on the 2-bit packed pair in LDS (word q holds bases 16q .. 16q+15; pattern at word 0, text at word offT)
struct EqPacked {
  volatile lds_u32* SQ; int offT;
  __device__ __forceinline__ bool operator()(int v, int h) const
  {
    const uint32_t a = (SQ[v >> 4] >> (2 * (v & 15))) & 3u, b = (SQ[offT + (h >> 4)] >> (2 * (h & 15))) & 3u;
    return a == b;
  }
};

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
  __shared__ uint16_t s_queue[WPB][QCAP];
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
  using lds_u16 = __attribute__((address_space(3))) uint16_t;
  volatile lds_u16* queue = (volatile lds_u16*)&s_queue[wib][0];
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
    bool fail = (pl + tl + 3 > ws.capa) || (pl + tl + 3 > 65535);   // queue entries are 16-bit diagonal indices
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
              queue[wq + rank] = (uint16_t)(kk - lo);
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
          queue[qn + rank] = (uint16_t)(k - lo);
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

    if (!backtrace_unpack(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_queue[wib][0], EqBytes{P, T})) continue;
    if (cells) cells[ti] = W;
  }
}

// W_p = 3 * sum of the wavefront widths the un-bounded aligner evaluates.  The ranges follow from the
// lengths, the free ends and the penalties alone: score 0 spans [lo0, hi0]; the next reachable score is
// f = min(x, o+e) (same span, widened by one on both sides when it is a gap open); from then on the previous score's
// I/D wavefronts widen the range by one diagonal per side and score (gap extension 1), clipped to [-pl, tl].
__device__ uint64_t affine_cells(const otg_align_task& t, int xs, int oes, int s_end)
{
  const int pl = (int)t.pattern_len, tl = (int)t.text_len;
  const bool ef = t.endsfree != 0;
  const int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
  uint64_t W = (uint64_t)(hi0 - lo0 + 1);
  const int sf = imin(xs, oes);
  if (s_end >= sf) {
    const int lof = oes <= xs ? imax(lo0 - 1, -pl) : lo0, hif = oes <= xs ? imin(hi0 + 1, tl) : hi0;
    const long long n = s_end - sf;
    auto ramp = [](long long base, long long room, long long n) -> long long {    // sum_{d=0..n} min(base + d, base + room)
      return (n + 1) * base + (n <= room ? n * (n + 1) / 2 : room * (room + 1) / 2 + (n - room) * room);
    };
    W += (uint64_t)(ramp(hif, tl - hif, n) + ramp(-lof, pl + lof, n) + (n + 1));
  }
  return 3ull * W;
}

// ---------------------------------------------------------------------------------------------------
// Score bound pass.  A banded (64*DPL diagonals, static band around the start and end diagonals), score-only run
// of the same recurrence, entirely in registers: lane l owns DPL adjacent diagonals, the M ring (OES rows), I and
// D are VGPR arrays, neighbours come from the lane itself or one DPP shift.  Any alignment it finds is a valid
// alignment of the pair, so its score U is an UPPER bound of the optimum (equal to it whenever the optimal path
// stays inside the band, which is the normal case for reads of one allele).  The exact kernel below uses U only
// to skip cells that cannot lie on an alignment of score <= U (see there), so a loose U costs time, never
// correctness.  U = INT_MAX when the band cannot hold the start and end diagonals.
__device__ __forceinline__ int dpp_shl1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false); }   // lane i <- lane i+1
__device__ __forceinline__ int dpp_shr1b(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }  // lane i <- lane i-1

template <int DPL, int XS, int OES>
__global__ __launch_bounds__(256) void wfa_affine_bound_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ bound, uint32_t* __restrict__ ticket)
{
  constexpr int NULLV = -(1 << 29);
  constexpr int R = XS > OES ? XS : OES;
  constexpr int BAND = 64 * DPL;
  const int lane = threadIdx.x & 63;
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
    const int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
    const int need_lo = imin(lo0, kend - tef), need_hi = imax(hi0, kend + pef);
    int result = 0x7fffffff;
    if (need_hi - need_lo + 1 + 64 <= BAND && pl > 0 && tl > 0 && pl < 32766 && tl < 32766) {
      const int blo = ((need_lo + need_hi) >> 1) - BAND / 2;
      const int k0 = blo + lane * DPL;
      int M[R][DPL], I[DPL], D[DPL], cur[DPL];
#pragma unroll
      for (int j = 0; j < DPL; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) M[r][j] = NULLV;
        I[j] = NULLV; D[j] = NULLV;
        const int k = k0 + j;
        const int h = k > 0 ? k : 0, v = h - k;
        cur[j] = (k >= lo0 && k <= hi0 && h <= tl && v <= pl) ? h : NULLV;
      }
      // extends cur[] along matches and reports whether some diagonal satisfies the end condition
      auto extend_and_test = [&]() -> bool {
        uint64_t a[DPL], b[DPL];
        bool more[DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
          const int h = cur[j], v = h - (k0 + j);
          const int vc = imin(imax(v, 0), pl), hc = imin(imax(h, 0), tl);
          a[j] = otg_load8(P + vc); b[j] = otg_load8(T + hc);
        }
        bool any = false;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
          const int h = cur[j], v = h - (k0 + j);
          const bool act = h >= 0 && v < pl && h < tl;
          const uint64_t xx = a[j] ^ b[j];
          int m = xx ? (int)(__builtin_ctzll(xx) >> 3) : 8;
          m = imin(m, imin(pl - v, tl - h));
          if (act) cur[j] = h + m;
          more[j] = act && m == 8 && v + 8 < pl && h + 8 < tl;
          any = any || more[j];
        }
        while (__ballot(any)) {      // another 8 bytes for the diagonals still inside a match run (predicated, no divergence)
          any = false;
#pragma unroll
          for (int j = 0; j < DPL; ++j) {
            const int h = cur[j], v = h - (k0 + j);
            const int vc = imin(imax(v, 0), pl), hc = imin(imax(h, 0), tl);
            a[j] = otg_load8(P + vc); b[j] = otg_load8(T + hc);
          }
#pragma unroll
          for (int j = 0; j < DPL; ++j) {
            const int h = cur[j], v = h - (k0 + j);
            const uint64_t xx = a[j] ^ b[j];
            int m = xx ? (int)(__builtin_ctzll(xx) >> 3) : 8;
            m = imin(m, imin(pl - v, tl - h));
            if (more[j]) cur[j] = h + m;
            more[j] = more[j] && m == 8 && v + 8 < pl && h + 8 < tl;
            any = any || more[j];
          }
        }
        bool fin = false;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
          const int h = cur[j], v = h - (k0 + j);
          fin = fin || (h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)));
        }
        return __ballot(fin) != 0;
      };
      if (extend_and_test()) result = 0;
      const int smax = 2 * (OES + pl + tl) + 8;
      for (int s = 1; result == 0x7fffffff && s <= smax; ++s) {
        // ring shift: M[0] = row s-1
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
#pragma unroll
          for (int r = R - 1; r > 0; --r) M[r][j] = M[r - 1][j];
          M[0][j] = cur[j];
        }
        int mo_l = dpp_shr1b(M[OES - 1][DPL - 1]), i_l = dpp_shr1b(I[DPL - 1]);
        int mo_r = dpp_shl1(M[OES - 1][0]), d_r = dpp_shl1(D[0]);
        if (lane == 0) { mo_l = NULLV; i_l = NULLV; }
        if (lane == 63) { mo_r = NULLV; d_r = NULLV; }
        int nI[DPL], nD[DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
          const int k = k0 + j;
          const int ml = j > 0 ? M[OES - 1][j - 1] : mo_l, il = j > 0 ? I[j - 1] : i_l;
          const int mr = j < DPL - 1 ? M[OES - 1][j + 1] : mo_r, dr = j < DPL - 1 ? D[j + 1] : d_r;
          int ins = imax(il, ml) + 1;
          int del = imax(dr, mr);
          const int mis = M[XS - 1][j] + 1;
          if (ins < 0 || ins > tl || ins - k > pl) ins = NULLV;
          if (del < 0 || del > tl || del - k > pl) del = NULLV;
          int mx = imax(del, imax(mis, ins));
          if (mx < 0 || mx > tl || mx - k > pl) mx = NULLV;
          nI[j] = ins; nD[j] = del; cur[j] = mx;
        }
#pragma unroll
        for (int j = 0; j < DPL; ++j) { I[j] = nI[j]; D[j] = nD[j]; }
        if (extend_and_test()) result = s;
      }
    }
    bound[ti] = result;      // wave-uniform value, same store from every lane
  }
}

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
    if (!backtrace_unpack(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_I[0], EqBytes{P, T})) continue;
    if (cells) cells[ti] = affine_cells(t, xs, oes, s_end);
    if (ws.visited && lane == 0) atomicAdd(ws.visited, (unsigned long long)slab_top);
  }
}

// ---------------------------------------------------------------------------------------------------
// v4 forward kernel: score-bounded alignments whose whole diamond of kept cells fits an LDS window of CAP
// diagonals.  Penalties (2,4,1) after gcd reduction (the default 4/6/2).  ALL wavefronts live in LDS as signed
// 16-bit offsets: I and D updated in place as in v3; M as four rows — a score only reads M rows of its own
// parity (s-2, s-4), and M[s] overwrites M[s-4] in place during the ascending sweep (left neighbour carried in
// a register, right neighbour still old), so two rows per parity suffice.  Six arrays x CAP x 2 bytes:
// 12 KB (CAP 1024) or 24 KB (CAP 2048) per alignment -> 12 / 6 alignments resident per CU, and the only HBM
// traffic left in the sweep is the 8-byte sequence probe and the provenance byte.  With a bound the ranges
// first grow and then shrink by one diagonal per side and score; whatever an array holds outside the current
// range is either null (never written) or a value of an older, wider wavefront that no later score reads
// (the readers' ranges have shrunk past it), see DESIGN.md §4.
template <int CAP, int QCAP, int NW, int SEQB, int WPEU>
__global__ __launch_bounds__(NW * 64, WPEU) void wfa_affine_kernel_v4(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm, int g,
    int32_t* __restrict__ scores, const uint64_t* __restrict__ cig_off, uint32_t* __restrict__ cig_len,
    uint8_t* __restrict__ cig_arena, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    AffWs ws, const int32_t* __restrict__ bound)
{
  constexpr int xs = 2, oes = 4, es = 1;
  __shared__ __attribute__((aligned(16))) int16_t s_I[CAP];
  __shared__ __attribute__((aligned(16))) int16_t s_D[CAP];
  __shared__ __attribute__((aligned(16))) int16_t s_M[4][CAP];
  __shared__ uint16_t s_q[NW][QCAP];
  __shared__ int s_misc[16];
  // Both sequences are packed to 2 bits per base into LDS once per alignment (pattern at word 0, text at word
  // offT): a probe is two unaligned 64-bit LDS reads covering 32 bases, so match runs almost never outlive the
  // probe and the sweep does not touch HBM except for the provenance byte.  Sequences with a byte outside ACGT
  // cannot be packed; those alignments go to the byte-compare tier.
  __shared__ uint32_t s_seq[SEQB / 4];
  volatile lds_u32* SQ = (volatile lds_u32*)&s_seq[0];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: keeps the share bounds and the sweep loop scalar
  volatile lds_i16* LI = (volatile lds_i16*)&s_I[0];
  volatile lds_i16* LD = (volatile lds_i16*)&s_D[0];
  volatile lds_u16* queue = (volatile lds_u16*)&s_q[wv][0];
  volatile __attribute__((address_space(3))) int* misc = (volatile __attribute__((address_space(3))) int*)&s_misc[0];
  uint8_t* my = ws.base + (size_t)blockIdx.x * ws.stride;
  int64_t* rowtab = (int64_t*)(my + ws.off_rowtab);
  uint8_t* rev = my + ws.off_rev;
  uint8_t* slab = my + ws.off_slab;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  constexpr int NUL16 = -32768;
  auto sync = [&]() { if (NW > 1) __syncthreads(); };

  for (;;) {
    uint32_t tk;
    if (NW > 1) {
      if (wv == 0) misc[0] = (int)otg_wave_atomic_add(ticket, 1u);
      __syncthreads();
      tk = (uint32_t)misc[0];
    } else tk = otg_wave_atomic_add(ticket, 1u);
    if (tk >= n_todo) break;
    const uint32_t ti = todo ? todo[tk] : tk;
    const otg_align_task t = tasks[ti];
    const uint8_t* P = arena + t.pattern_off;
    const uint8_t* T = arena + t.text_off;
    const int pl = (int)t.pattern_len, tl = (int)t.text_len;
    const bool ef = t.endsfree != 0;
    const int pef = t.pattern_end_free, tef = t.text_end_free;
    const int kend = tl - pl;
    const int U = bound[ti];
    const int elo = kend - (ef ? tef : 0), ehi = kend + (ef ? pef : 0);
    int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
    bool fail = U >= 0x40000000 || pl >= 32766 || tl >= 32766;
    const int offT = (pl + 15) / 16 + 3;                 // in words; three words of slack: a probe reads three words from its own
    if ((offT + (tl + 15) / 16 + 3) * 4 > SEQB) fail = true;
    int kbase = 0;
    if (!fail) {
      lo0 = imax(lo0, elo - U); hi0 = imin(hi0, ehi + U);
      // all kept cells lie in [wlo, whi]: the range of score s is within [lo0 - s, hi0 + s] and [elo - (U-s), ehi + (U-s)]
      const int wlo = imax((lo0 + elo - U) >> 1, -pl) - 1, whi = imin((hi0 + ehi + U + 1) >> 1, tl) + 1;
      kbase = wlo - 2;
      if (hi0 < lo0 || whi - kbase + 4 >= CAP) fail = true;
    }
    size_t slab_top = 0;
    int s_end = -1, k_end = 0;
    // ranges of the last four scores (r1 = s-1 ... r4 = s-4) and of the previous I/D wavefronts; null = lo > hi
    int r1lo = 1, r1hi = 0, r2lo = 1, r2hi = 0, r3lo = 1, r3hi = 0, r4lo = 1, r4hi = 0, idlo = 1, idhi = 0;
    if (!fail) {
      volatile lds_u32* f32 = (volatile lds_u32*)&s_I[0];     // s_I, s_D, s_M are contiguous? not guaranteed: fill each
      volatile lds_u32* d32 = (volatile lds_u32*)&s_D[0];
      volatile lds_u32* m32 = (volatile lds_u32*)&s_M[0][0];
      for (int q = (int)threadIdx.x; q < CAP / 2; q += NW * 64) { f32[q] = 0x80008000u; d32[q] = 0x80008000u; }
      for (int q = (int)threadIdx.x; q < 2 * CAP; q += NW * 64) m32[q] = 0x80008000u;
      // pack: word q holds bases 16q .. 16q+15, base b in bits 2(b&15)..+1, code = (byte >> 1) & 3 (A C T G -> 0 1 2 3)
      bool bad = false;
      auto pack = [&](const uint8_t* S, int len, int woff) {
        for (int q = (int)threadIdx.x; q < (len + 15) / 16; q += NW * 64) {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int b0 = 16 * q + 8 * j;
            uint64_t x = b0 < len + 8 ? otg_load8(S + (b0 < len ? b0 : len)) : 0ull;     // stays within 8 bytes past the end
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
      if (NW > 1) { if (threadIdx.x == 0) misc[2] = 0; __syncthreads(); if (bad) misc[2] = 1; __syncthreads(); fail = misc[2] != 0; }
      else fail = __ballot(bad) != 0ull;
    }
    sync();
    // 32 bases (64 bits) of the packed pattern / text starting at base position pos
    auto ld32b = [&](int woff, int pos) -> uint64_t {
      const int w = woff + (pos >> 4);
      const uint32_t sh = (uint32_t)(pos & 15) * 2u;
      const uint32_t d0 = SQ[w], d1 = SQ[w + 1], d2 = SQ[w + 2];
      return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
    };
    // equal leading bases looking at most 32*nb bases ahead (and at most rem)
    auto match_n = [&](int v, int h, int rem, int nb) -> int {
      int m = 32 * nb;
      for (int i = nb - 1; i >= 0; --i) {
        const uint64_t x = (32 * i < rem) ? (ld32b(0, v + 32 * i) ^ ld32b(offT, h + 32 * i)) : ~0ull;
        if (x) m = 32 * i + (int)(__builtin_ctzll(x) >> 1);
      }
      return m < rem ? m : rem;
    };
    // wave-cooperative extension of ONE diagonal (arguments wave-uniform), 2048 bases per iteration
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

    for (int s = 0; !fail; ++s) {
      if (s >= ws.nrows) { fail = true; break; }
      int lo, hi;
      if (s == 0) { lo = lo0; hi = hi0; }
      else {
        lo = 1 << 30; hi = -(1 << 30);
        if (r2hi >= r2lo) { lo = imin(lo, r2lo); hi = imax(hi, r2hi); }            // M[s-x]
        if (r4hi >= r4lo) { lo = imin(lo, r4lo - 1); hi = imax(hi, r4hi + 1); }    // M[s-o-e]
        if (idhi >= idlo) { lo = imin(lo, idlo - 1); hi = imax(hi, idhi + 1); }    // I/D[s-e]
        if (lo < -pl) lo = -pl;
        if (hi > tl) hi = tl;
        if (hi >= lo) {
          const int room = U - s;
          lo = imax(lo, elo - room); hi = imin(hi, ehi + room);
          if (room < 0 || hi < lo) { fail = true; break; }
        }
      }
      r4lo = r3lo; r4hi = r3hi; r3lo = r2lo; r3hi = r2hi; r2lo = r1lo; r2hi = r1hi;
      if (hi < lo) {   // unreachable score (only before the first gap-open score); nothing is written
        r1lo = 1; r1hi = 0; idlo = 1; idhi = 0;
        if (wv == 0) rowtab[s] = -1;
        if (s > 2 * (oes + es * (pl + tl)) + 8) fail = true;
        continue;
      }
      r1lo = lo; r1hi = hi;
      if (lo - kbase < 2 || hi - kbase + 3 >= CAP) { fail = true; break; }
      const int width = hi - lo + 1;
      if (slab_top + (size_t)width > ws.slab_bytes) { fail = true; break; }
      uint8_t* btrow = slab + slab_top - lo;
      if (wv == 0) rowtab[s] = (int64_t)slab_top - lo;
      slab_top += (size_t)width;
      const int par = (s & 1) * 2, slot = (s >> 1) & 1;
      volatile lds_i16* Mn = (volatile lds_i16*)&s_M[par + slot][0];          // M[s-4] on entry, M[s] on exit
      volatile lds_i16* Mm = (volatile lds_i16*)&s_M[par + (slot ^ 1)][0];    // M[s-2]
      const int nch = (width + 63) >> 6;
      const int c0 = lo + 64 * ((nch * wv) / NW), c1 = lo + 64 * ((nch * (wv + 1)) / NW);
      // values at the share boundaries that a neighbouring wave overwrites during its own sweep
      int bI = NUL16, bD = NUL16, bMl = NUL16, bMr = NUL16;
      {
        const int jl0 = c0 - 1 - kbase, jr0 = imin(c1 - kbase, CAP - 1);
        bI = (int)LI[jl0]; bMl = (int)Mn[jl0]; bD = (int)LD[jr0]; bMr = (int)Mn[jr0];
      }
      sync();
      bool done = false;
      int qn = 0;
      auto drain = [&]() {
        int pass = 0;
        while (qn > 0) {
          if (qn <= 4 && pass > 0) {
            for (int e = 0; e < qn; ++e) {
              const int kk = lo + (int)queue[e];
              const int h = Mn[kk - kbase];
              const int v = h - kk;
              const int m = wave_match(v, h, imin(pl - v, tl - h));
              Mn[kk - kbase] = (int16_t)(h + m);
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
              h = Mn[kk - kbase];
              v = h - kk;
              const int rem = imin(pl - v, tl - h);
              int m, full;
              if (pass == 0) { m = match_n(v, h, rem, 2); full = 64; }
              else { m = match_n(v, h, rem, 8); full = 256; }
              v += m; h += m;
              more = (m == full) && v < pl && h < tl;
              Mn[kk - kbase] = (int16_t)h;
            }
            const unsigned long long mm = __ballot(more);
            if (more) {
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
              queue[wq + rank] = (uint16_t)(kk - lo);
            }
            wq += __builtin_popcountll(mm);
          }
          qn = wq; ++pass;
        }
      };
      if (s == 0) {
        // score 0: offset max(k, 0) on every start diagonal, no I/D wavefronts; the drain extends them from scratch
        for (int c = c0; c < c1; c += 64) {
          const int k = c + lane;
          const int jl = k - kbase;
          const bool in = k <= hi;
          const int h = k > 0 ? k : 0, v = h - k;
          const bool valid = in && h <= tl && v <= pl;
          const bool more = valid && v < pl && h < tl;
          if (in) { LI[jl] = (int16_t)NUL16; LD[jl] = (int16_t)NUL16; Mn[jl] = (int16_t)(valid ? h : NUL16); btrow[k] = 0; }
          const unsigned long long mq = __ballot(more);
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
          queue[more ? qn + rank : QCAP - 1] = (uint16_t)(k - lo);
          qn += __builtin_popcountll(mq);
          if (qn + 128 > QCAP) drain();
        }
        drain();
      } else {
        // LDS operands one chunk ahead (clamped index: lanes past the window never store)
        int jn = imin(c0 + lane - kbase, CAP - 2);
        int n_iold = LI[jn], n_dx = LD[jn + 1], n_mo = Mn[jn], n_mor = Mn[jn + 1], n_mm = Mm[jn];
        int carryI = bI, carryMo = bMl;
        bool p_pending = false, p_in = false, p_valid = false, p_probe = false;
        int p_k = 0, p_h = 0, p_v = 0;
        uint64_t p_a = 0, p_b = 0;
        // retire a chunk: branch-free, lanes that must not store aim at the unused last slot of the row / queue
        auto finish = [&]() {
          const uint64_t xx = p_a ^ p_b;
          int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
          m = imin(m, imin(pl - p_v, tl - p_h));
          m = p_probe ? m : 0;
          int h = p_h + m, v = p_v + m;
          bool more = p_probe && m == 32 && v < pl && h < tl;
          if (__ballot(more)) {      // a second probe where a run outlives the first (4 chunks in 10): the queue and its drain — a fixed cost per wave and score — are left to runs beyond 64 bases
            const uint64_t x2 = ld32b(0, more ? v : 0) ^ ld32b(offT, more ? h : 0);
            int m2 = x2 ? (int)(__builtin_ctzll(x2) >> 1) : 32;
            m2 = imin(m2, imin(pl - v, tl - h));
            m2 = more ? m2 : 0;
            h += m2; v += m2;
            more = more && m2 == 32 && v < pl && h < tl;
          }
          Mn[p_in ? p_k - kbase : CAP - 1] = (int16_t)(p_valid ? h : NUL16);
          const unsigned long long mq = __ballot(more);
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
          queue[more ? qn + rank : QCAP - 1] = (uint16_t)(p_k - lo);
          qn += __builtin_popcountll(mq);
        };
        for (int c = c0; c < c1; c += 64) {
          const int k = c + lane;
          const int jl = k - kbase;
          const bool in = k <= hi;
          const int iold = n_iold;                              // I[s-1][k]
          int dx = n_dx;                                        // D[s-1][k+1]
          const int mo = n_mo;                                  // M[s-o-e][k]
          int dop = n_mor;                                      // M[s-o-e][k+1]
          const int mm = n_mm;                                  // M[s-x][k]
          if (lane == 63 && c + 64 >= c1) { dx = bD; dop = bMr; }   // first diagonal of the next wave's share
          jn = imin(jl + 64, CAP - 2);
          n_iold = LI[jn]; n_dx = LD[jn + 1]; n_mo = Mn[jn]; n_mor = Mn[jn + 1]; n_mm = Mm[jn];
          int ix = dpp_shr1(iold);                              // I[s-1][k-1]
          if (lane == 0) ix = carryI;
          carryI = __builtin_amdgcn_readlane(iold, 63);
          int io = dpp_shr1(mo);                                // M[s-o-e][k-1]
          if (lane == 0) io = carryMo;
          carryMo = __builtin_amdgcn_readlane(mo, 63);
          uint32_t bits = 0;
          int ins, del;
          if (ix >= io) { ins = ix; bits |= 4u; } else ins = io;
          ins += 1;                                             // a null (-32768) creeps up by one per score: stays negative for < 32768 scores
          if (dx >= dop) { del = dx; bits |= 8u; } else del = dop;
          const int mis = mm + 1;
          const int mx = imax(del, imax(mis, ins));
          uint32_t org = 0;
          if (mx == ins) org = 2;
          if (mx == del) org = 1;
          if (mx == mis) org = 0;
          bits |= org;
          const int h = mx, v = mx - k;
          const bool valid = in && (uint32_t)h <= (uint32_t)tl && (uint32_t)v <= (uint32_t)pl;
          const bool probe = valid && v < pl && h < tl;
          if (p_pending) finish();                              // retire the previous chunk (its probe was issued one iteration ago)
          const uint64_t a = ld32b(0, valid ? v : 0), b = ld32b(offT, valid ? h : 0);
          if (in) {
            LI[jl] = (int16_t)ins;
            LD[jl] = (int16_t)del;
            btrow[k] = (uint8_t)bits;
          }
          p_pending = true; p_in = in; p_valid = valid; p_probe = probe; p_k = k; p_h = h; p_v = v; p_a = a; p_b = b;
          if (qn + 128 > QCAP) { finish(); p_pending = false; drain(); }
        }
        if (p_pending) finish();
        drain();
      }
      idlo = s == 0 ? 1 : lo; idhi = s == 0 ? 0 : hi;
      // termination: the first diagonal (ascending) whose fully extended offset satisfies the end condition
      int cand = 0x7fffffff;
      if (!ef) {
        if (kend >= c0 && kend < c1 && kend <= hi) { const int x = Mn[kend - kbase]; if (x >= tl) cand = kend; }
      } else {
        for (int c = c0; c < c1 && cand == 0x7fffffff; c += 64) {
          const int k = c + lane;
          bool fin = false;
          if (k <= hi) {
            const int h = Mn[k - kbase];
            const int v = h - k;
            fin = h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef));
          }
          const unsigned long long fm = __ballot(fin);
          if (fm) cand = c + (int)__builtin_ctzll(fm);
        }
      }
      if (NW > 1) {
        misc[4 + wv] = cand;
        __syncthreads();
        int best = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int x = misc[4 + w]; best = x < best ? x : best; }
        cand = best;
      }
      if (cand != 0x7fffffff) { done = true; s_end = s; k_end = cand; }
      if (done) break;
    }
    sync();

    if (wv != 0) continue;            // wave 0 reports / unpacks; the others wait at the next ticket barrier
    if (fail || s_end < 0) {
      if (overflow_list) { const uint32_t q = otg_wave_atomic_add(n_overflow, 1u); overflow_list[q] = ti; }
      else { scores[ti] = -1; cig_len[ti] = 0; }
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // provenance bytes of the other waves (same CU, through L2)
    if (ws.dbg & 2) { scores[ti] = s_end * g; cig_len[ti] = 0; continue; }
    if (!backtrace_unpack(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_M[0][0], EqPacked{SQ, offT})) continue;
    if (cells) cells[ti] = affine_cells(t, xs, oes, s_end);
    if ((ws.dbg & 1) && threadIdx.x == 0) { atomicAdd(&otg_dbg_v4_cells[0], (unsigned long long)slab_top); atomicAdd(&otg_dbg_v4_cells[1], 1ull); }
    if (ws.visited && threadIdx.x == 0) atomicAdd(ws.visited, (unsigned long long)slab_top);
  }
}

// ---------------------------------------------------------------------------------------------------
// v5 forward kernel: REGISTER-RESIDENT wavefronts, one wave per alignment, no barriers (penalties (2,4,1) after gcd
// reduction, score-bounded alignments whose diamond fits a window of CAP = 128 * S2 diagonals).
//   * Window index x = k - kbase; lane l of pair-slot i owns the two ADJACENT diagonals x = 128 i + 2 l (+1).  Per diagonal three
//     32-bit words live in VGPRs for the whole alignment: two M words — one per score parity, {lo16: M[s-4], hi16: M[s-2]}, a score
//     only reads M rows of its own parity — and one {lo16: I[s-1], hi16: D[s-1]} word.  The slot loop is fully unrolled, so every
//     register index is static; which pair-slots a score touches is a wave-uniform branch per slot.  After a score the two parity
//     arrays trade places (v_swap per register).
//   * Of the four neighbours a pair of cells needs, two sit in the lane itself (the even cell's right neighbour is the lane's odd
//     cell and vice versa); the other two come with ONE DPP wave shift each of a packed export word (lane 0 / 63 take the value of
//     the adjacent pair-slot through readlane).  M[s] replaces M[s-4] by a 16-bit rotate of the word, I and D are updated in place
//     (ascending sweep: the left neighbour's old value travels in an SGPR, the right neighbour's is still old).
//   * Cells outside the score's range [lo, hi] but inside an active pair-slot are computed like any other: every value a cell ever
//     holds is the offset of a real alignment prefix of at most that score, so such cells can only matter if they lie on an alignment
//     of score <= U — and then they are inside the diamond by its definition.  (Same argument as for the pruned cells of v3/v4.)
//   * Sequences: 2 bits per base in LDS (as v4), a probe covers 32 bases.  A cell whose match run outlives the probe (1.2 % at ONT
//     divergence) is pushed {x, h} to a per-wave LDS queue; the queue is drained in full 64-lane batches at the end of the score and
//     the final offsets come back through a 16-bit patch table that one packed max per cell folds into the M words.
//   * Provenance: one byte per cell of the touched pair-slots (row = 128 * (j1 - j0 + 1) bytes), the lane's two bytes in one
//     16-bit store.  Same row table, backtrace and unpack as every other tier.
typedef short otg_short2 __attribute__((ext_vector_type(2)));
// compile-time loop: the body sees its index as a constant expression, so register arrays are only ever indexed statically
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
  otg_short2 x, y;
  __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  const otg_short2 r = __builtin_elementwise_max(x, y);
  uint32_t o; __builtin_memcpy(&o, &r, 4); return o;
}
__device__ __forceinline__ int lo16s(uint32_t w) { return (int)(int16_t)(w & 0xffffu); }
__device__ __forceinline__ int hi16s(uint32_t w) { return (int)w >> 16; }
__device__ __forceinline__ uint64_t v5_uniform64(uint64_t x)
{
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }   // {lo16: lo, hi16: hi}

// NW > 1: NW waves share ONE alignment whose window is NW x S2 pair-slots wide.  Pair-slot g belongs to wave g % NW (cyclic, so the touched
// slots of a score spread evenly over the waves); every neighbour of a slot then lives in another wave, and the two export words per slot
// travel through a double-buffered LDS table that the owner writes at the END of a score for the next one.  One barrier per score: it
// closes the score (exports and the waves' termination candidates are published before it, read after it).  NW == 1 is the kernel above
// (four independent one-wave alignments per block; neighbours through registers).
template <int NW, int S2, int SEQB, int WPEU>
__global__ __launch_bounds__(NW == 1 ? 256 : NW * 64, WPEU) void wfa_affine_kernel_v5(
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
  constexpr int QCAP = 512;
  constexpr uint32_t NN = 0x80008000u;
  constexpr int NUL16 = -32768;
  __shared__ uint32_t s_seq[ALN][SEQB / 4];
  __shared__ uint32_t s_patch[ALN][CAP / 2];
  __shared__ uint32_t s_queue[WAVES][QCAP];
  __shared__ uint32_t s_xl[2][GS + 2], s_xr[2][GS + 2];     // [score parity][slot + 1]: export words of the slots (left-going / right-going)
  __shared__ int s_cand[2][WAVES];
  __shared__ int s_misc[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int al = NW == 1 ? wv : 0;                  // which alignment of the block this wave works on
  const int ww = NW == 1 ? 0 : wv;                  // its rank among the waves of that alignment
  uint32_t* SQ = &s_seq[al][0];
  volatile lds_u32* PT = (volatile lds_u32*)&s_patch[al][0];
  volatile lds_u32* QU = (volatile lds_u32*)&s_queue[wv][0];
  volatile lds_u16* PT16 = (volatile lds_u16*)&s_patch[al][0];
  volatile lds_u32* XL = (volatile lds_u32*)&s_xl[0][0];
  volatile lds_u32* XR = (volatile lds_u32*)&s_xr[0][0];
  volatile __attribute__((address_space(3))) int* CA = (volatile __attribute__((address_space(3))) int*)&s_cand[0][0];
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
    // 12- and 16-slot bodies, and the whole score loop then runs on vector compares and exec masks), so every field is pinned explicitly.
    const uint32_t ti = (uint32_t)__builtin_amdgcn_readfirstlane((int)todo[seg0 + tk]);
    otg_align_task t = tasks[ti];
    t.pattern_off = v5_uniform64(t.pattern_off); t.text_off = v5_uniform64(t.text_off);
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
    int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
    bool fail = U >= 0x40000000 || pl >= 32766 || tl >= 32766;
    const int offT = (pl + 15) / 16 + 3;
    if ((offT + (tl + 15) / 16 + 3) * 4 > SEQB) fail = true;
    int kbase = 0;
    if (!fail) {
      lo0 = imax(lo0, elo - U); hi0 = imin(hi0, ehi + U);
      const int wlo = imax((lo0 + elo - U) >> 1, -pl) - 1, whi = imin((hi0 + ehi + U + 1) >> 1, tl) + 1;
      kbase = wlo - 2;
      if (hi0 < lo0 || whi - kbase + 4 >= CAP) fail = true;
    }
    uint32_t MC[2][S2 + 1][2];          // [parity slot][pair-slot][even / odd diagonal]: {lo16: M[s-4], hi16: M[s-2]}; [0] = the parity of the current score
    uint32_t ID[S2 + 1][2];           // {lo16: I[s-1], hi16: D[s-1]}
    static_for<0, S2 + 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
      MC[0][i][0] = NN; MC[0][i][1] = NN; MC[1][i][0] = NN; MC[1][i][1] = NN; ID[i][0] = NN; ID[i][1] = NN; });
    if (!fail) {
      for (int q = (NW == 1 ? lane : (int)threadIdx.x); q < CAP / 2; q += (NW == 1 ? 64 : NW * 64)) PT[q] = NN;
      if (NW > 1) for (int q = (int)threadIdx.x; q < 2 * (GS + 2); q += NW * 64) { XL[q] = NN; XR[q] = NN; }
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

    size_t slab_top = 0;
    int s_end = -1, k_end = 0;
    int r1lo = 1, r1hi = 0, r2lo = 1, r2hi = 0, r3lo = 1, r3hi = 0, r4lo = 1, r4hi = 0, idlo = 1, idhi = 0;
    const int xe = kend - kbase;                 // window index of the end diagonal (end-to-end termination)

    int lane2 = 2 * lane, kb = __builtin_amdgcn_readfirstlane(kbase);
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
      int cand = 0x7fffffff;
      int j0 = 1, j1 = 0;                        // touched pair-slots of this score (none when the score is unreachable)
      if (hi < lo) {   // unreachable score: nothing is written, the parity arrays still trade places
        r1lo = 1; r1hi = 0; idlo = 1; idhi = 0;
        rowtab[s] = -1;
        if (s > 2 * (oes + es * (pl + tl)) + 8) { fail = true; break; }
      } else {
      r1lo = lo; r1hi = hi;
      const int xlo = lo - kbase, xhi = hi - kbase;
      if (xlo < 2 || xhi + 3 >= CAP) { fail = true; break; }
      j0 = xlo >> 7; j1 = xhi >> 7;
      const int width = (j1 - j0 + 1) * 128;
      if (slab_top + (size_t)width > ws.slab_bytes) { fail = true; break; }
      uint8_t* brow = slab + slab_top - 128 * j0;                  // provenance byte of window index x: brow[x]
      rowtab[s] = (int64_t)slab_top - (int64_t)(kbase + 128 * j0);  // wave-uniform store (same value from every lane and every wave)
      slab_top += (size_t)width;
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
              int kc = fin ? kk : 0x7fffffff;
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
      bool pushed = false, qfull = false;
      auto push2 = [&](bool moreE, bool moreO, int xE, int hE, int hO) {
        const unsigned long long mE = __ballot(moreE), mO = __ballot(moreO);
        if (mE | mO) {
          if (qn + 128 > QCAP) { qfull = true; return; }          // more unfinished match runs in one score than the queue holds: the next tier takes the alignment
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
          pushed = true;
        }
      };
      // end condition of a fully extended cell (ends-free form; the end-to-end form is checked on the one end diagonal)
      auto fin_ef = [&](int h, int v) -> bool { return h >= 0 && ((h >= tl && pl - v <= pef) || (v >= pl && tl - h <= tef)); };

      if (s == 0) {
        // score 0: offset max(k, 0) on every start diagonal, no I / D wavefronts; everything is extended through the queue
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (gi < j0 || gi > j1) return;
          const int xE = 128 * gi + lane2, kE = kb + xE, kO = kE + 1;
          const int hE = kE > 0 ? kE : 0, vE = hE - kE, hO = kO > 0 ? kO : 0, vO = hO - kO;
          const bool validE = kE >= lo && kE <= hi && hE <= tl && vE <= pl, validO = kO >= lo && kO <= hi && hO <= tl && vO <= pl;
          MC[0][i][0] = pack16(NUL16, validE ? hE : NUL16);
          MC[0][i][1] = pack16(NUL16, validO ? hO : NUL16);
          const bool moreE = validE && vE < pl && hE < tl, moreO = validO && vO < pl && hO < tl;
          if (ef) {
            const bool fE = validE && !moreE && fin_ef(hE, vE), fO = validO && !moreO && fin_ef(hO, vO);
            const unsigned long long fm = __ballot(fE || fO);
            if (fm) { int kc = fE ? kE : (fO ? kO : 0x7fffffff); kc = -otg_wave_max_i32(-kc); cand = imin(cand, kc); }
          } else if (xe >= 128 * gi && xe < 128 * gi + 128) {
            const int hx = __builtin_amdgcn_readlane((xe & 1) ? (validO && !moreO ? hO : -1) : (validE && !moreE ? hE : -1), (xe & 127) >> 1);
            if (hx >= tl) cand = kend;
          }
          push2(moreE, moreO, xE, hE, hO);
        });
      } else {
        // ---- the sweep over the touched pair-slots, ascending
        uint32_t carryL = NN;                     // {M[s-4], I[s-1]} of the diagonal left of the current pair-slot (NW == 1: carried in a register)
        const int xp = (s & 1) * (GS + 2);        // this score's half of the export tables
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (NW == 1 && gi + 1 == j0) {           // the slot left of the first touched one: its lane 63 is the left neighbour of the sweep
            // (packed on the scalar side: a vector pack here is hoisted above the branch and then paid by every untouched slot)
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)MC[0][i][1], 63), b = (uint32_t)__builtin_amdgcn_readlane((int)ID[i][1], 63);
            carryL = (a & 0xffffu) | (b << 16);
          }
          if (gi < j0 || gi > j1) return;
          const uint32_t mE = MC[0][i][0], mO = MC[0][i][1], dE = ID[i][0], dO = ID[i][1];
          const uint32_t Lx = pack16(lo16s(mO), lo16s(dO));                               // what the lane to the right needs: {M[s-4][odd], I[s-1][odd]}
          const uint32_t Rx = pack16(lo16s(mE), hi16s(dE));                               // what the lane to the left needs: {M[s-4][even], D[s-1][even]}
          uint32_t rcar = NN, lcar = carryL;
          if (NW == 1) {
            if (i + 1 < S2) {                      // the next pair-slot's lane 0, still old
              const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)MC[0][i + 1][0], 0), b = (uint32_t)__builtin_amdgcn_readlane((int)ID[i + 1][0], 0);
              rcar = (a & 0xffffu) | (b & 0xffff0000u);
            }
          } else {
            lcar = (uint32_t)__builtin_amdgcn_readfirstlane((int)XL[xp + gi]);             // export of slot gi - 1 (entry g + 1 holds slot g)
            rcar = (uint32_t)__builtin_amdgcn_readfirstlane((int)XR[xp + gi + 2]);         // export of slot gi + 1
          }
          const uint32_t lnb = (uint32_t)__builtin_amdgcn_update_dpp((int)lcar, (int)Lx, 0x138, 0xf, 0xf, false);   // lane l <- lane l-1, lane 0 <- the left slot
          const uint32_t rnb = (uint32_t)__builtin_amdgcn_update_dpp((int)rcar, (int)Rx, 0x130, 0xf, 0xf, false);   // lane l <- lane l+1, lane 63 <- the right slot
          if (NW == 1) carryL = (uint32_t)__builtin_amdgcn_readlane((int)Lx, 63);
          const int xE = 128 * gi + lane2, kE = kb + xE;
          int insE, delE, mxE, insO, delO, mxO;
          uint32_t bitsE, bitsO;
          {   // even diagonal: left neighbour from lane l-1, right neighbour is the lane's own odd diagonal
            const int io = lo16s(lnb), ix = hi16s(lnb), dop = lo16s(mO), dx = hi16s(dO), mm = hi16s(mE);
            bitsE = 0;
            if (ix >= io) { insE = ix; bitsE |= 4u; } else insE = io;
            insE += 1;
            if (dx >= dop) { delE = dx; bitsE |= 8u; } else delE = dop;
            const int mis = mm + 1;
            mxE = imax(delE, imax(mis, insE));
            uint32_t org = 0;
            if (mxE == insE) org = 2;
            if (mxE == delE) org = 1;
            if (mxE == mis) org = 0;
            bitsE |= org;
          }
          {   // odd diagonal: left neighbour is the lane's own even diagonal (old values), right neighbour from lane l+1
            const int io = lo16s(mE), ix = lo16s(dE), dop = lo16s(rnb), dx = hi16s(rnb), mm = hi16s(mO);
            bitsO = 0;
            if (ix >= io) { insO = ix; bitsO |= 4u; } else insO = io;
            insO += 1;
            if (dx >= dop) { delO = dx; bitsO |= 8u; } else delO = dop;
            const int mis = mm + 1;
            mxO = imax(delO, imax(mis, insO));
            uint32_t org = 0;
            if (mxO == insO) org = 2;
            if (mxO == delO) org = 1;
            if (mxO == mis) org = 0;
            bitsO |= org;
          }
          ID[i][0] = pack16(insE, delE);
          ID[i][1] = pack16(insO, delO);
          int hE = mxE, hO = mxO;
          const int vE = mxE - kE, vO = mxO - kE - 1;
          const bool validE = (uint32_t)hE <= (uint32_t)tl && (uint32_t)vE <= (uint32_t)pl;
          const bool validO = (uint32_t)hO <= (uint32_t)tl && (uint32_t)vO <= (uint32_t)pl;
          const bool probeE = validE && vE < pl && hE < tl, probeO = validO && vO < pl && hO < tl;
          const uint64_t aE = ld32b(0, validE ? vE : 0), bE = ld32b(offT, validE ? hE : 0);
          const uint64_t aO = ld32b(0, validO ? vO : 0), bO = ld32b(offT, validO ? hO : 0);
          const uint16_t b2 = (uint16_t)(bitsE | (bitsO << 8));
          __builtin_memcpy(brow + xE, &b2, 2);
          bool moreE, moreO;
          {
            const uint64_t xx = aE ^ bE;
            int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
            m = imin(m, imin(pl - vE, tl - hE));
            m = probeE ? m : 0;
            hE += m;
            moreE = probeE && m == 32 && vE + m < pl && hE < tl;
          }
          {
            const uint64_t xx = aO ^ bO;
            int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
            m = imin(m, imin(pl - vO, tl - hO));
            m = probeO ? m : 0;
            hO += m;
            moreO = probeO && m == 32 && vO + m < pl && hO < tl;
          }
          // a second probe where a run outlives the first (one in 120 cells at ONT divergence, i.e. most slot visits have one): the queue, its
          // drain and the fold-back of the patch table — a fixed cost per score — are then left to runs beyond 64 bases (one slot visit in 100)
          if (__ballot(moreE || moreO)) {
            // one probe sequence for both cells of the lane: it extends the even cell if that one needs it, else the odd one (a lane where both do —
            // one in 15 000 — leaves the odd cell to the queue)
            const bool any = moreE || moreO, selO = !moreE && moreO;
            int h2 = selO ? hO : hE;
            const int v2 = h2 - (selO ? kE + 1 : kE);
            const uint64_t xx = ld32b(0, any ? v2 : 0) ^ ld32b(offT, any ? h2 : 0);
            int m = xx ? (int)(__builtin_ctzll(xx) >> 1) : 32;
            m = imin(m, imin(pl - v2, tl - h2));
            m = any ? m : 0;
            h2 += m;
            const bool more2 = any && m == 32 && v2 + m < pl && h2 < tl;
            if (selO) { hO = h2; moreO = more2; }
            else if (moreE) { hE = h2; moreE = more2; }
          }
          const int sE = validE ? hE : NUL16, sO = validO ? hO : NUL16;
          MC[0][i][0] = (mE >> 16) | ((uint32_t)sE << 16);
          MC[0][i][1] = (mO >> 16) | ((uint32_t)sO << 16);
          if (ef) {
            const bool fE = validE && !moreE && fin_ef(hE, hE - kE), fO = validO && !moreO && fin_ef(hO, hO - kE - 1);
            const unsigned long long fm = __ballot(fE || fO);
            if (fm) { int kc = fE ? kE : (fO ? kE + 1 : 0x7fffffff); kc = -otg_wave_max_i32(-kc); cand = imin(cand, kc); }
          } else if (xe >= 128 * gi && xe < 128 * gi + 128) {
            const int hx = __builtin_amdgcn_readlane((xe & 1) ? (validO && !moreO ? hO : -1) : (validE && !moreE ? hE : -1), (xe & 127) >> 1);
            if (hx >= tl) cand = kend;
          }
          push2(moreE, moreO, xE, hE, hO);
        });
      }
      if (qfull) cand = FAILV;
      else if (pushed) {
        drain();
        // fold the final offsets of the queued cells into the M words (partial offset <= final offset: a packed max)
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = NW == 1 ? i : i * NW + ww;
          if (gi < j0 || gi > j1) return;
          const uint32_t pw = PT[64 * gi + lane];
          if (__ballot(pw != NN)) {
            MC[0][i][0] = pk_max_i16(MC[0][i][0], pack16(NUL16, lo16s(pw)));
            MC[0][i][1] = pk_max_i16(MC[0][i][1], pack16(NUL16, hi16s(pw)));
            PT[64 * gi + lane] = NN;
          }
        });
      }
      idlo = s == 0 ? 1 : lo; idhi = s == 0 ? 0 : hi;
      }
      // ---- close the score: (NW > 1) publish the export words the neighbours need for score s + 1 — from the arrays of the OTHER parity,
      // which are the current ones of the next score, and the I / D words just written — and this wave's candidate; one barrier; then every
      // wave sees every candidate
      cand = __builtin_amdgcn_readfirstlane(cand);
      if (NW > 1) {
        const int np = ((s + 1) & 1) * (GS + 2);
        static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
          const int gi = i * NW + ww;
          if (gi + 2 < j0 || gi > j1 + 2) return;          // the next score's range moves by at most one diagonal
          const uint32_t Lx = pack16(lo16s(MC[1][i][1]), lo16s(ID[i][1]));
          const uint32_t Rx = pack16(lo16s(MC[1][i][0]), hi16s(ID[i][0]));
          if (lane == 63) XL[np + gi + 1] = Lx;
          if (lane == 0) XR[np + gi + 1] = Rx;
        });
        if (lane == 0) CA[(s & 1) * WAVES + ww] = cand;
        __syncthreads();
        int gc = 0x7fffffff;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) { const int c2 = CA[(s & 1) * WAVES + w2]; gc = c2 < gc ? c2 : gc; }
        cand = __builtin_amdgcn_readfirstlane(gc);
      }
      if (cand == FAILV) { fail = true; break; }
      if (cand != 0x7fffffff) { s_end = s; k_end = cand; break; }
      // the other parity is next
      static_for<0, S2>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value;
        { const uint32_t t_ = MC[0][i][0]; MC[0][i][0] = MC[1][i][0]; MC[1][i][0] = t_; }
        { const uint32_t t_ = MC[0][i][1]; MC[0][i][1] = MC[1][i][1]; MC[1][i][1] = t_; } });
    }

    if (NW > 1 && wv != 0) continue;           // wave 0 reports / unpacks; the others wait at the next ticket barrier
    if (fail || s_end < 0) {
      const uint32_t q = otg_wave_atomic_add(n_overflow, 1u);
      overflow_list[q] = ti;                                   // wave-uniform store
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    if (!backtrace_unpack(P, pl, T, tl, s_end, k_end, xs, oes, es, rowtab, slab, rev, ws.rev_cap, cig_arena + cig_off[ti], lane, &scores[ti], &cig_len[ti], g, (volatile lds_u32*)&s_queue[wv][0], EqPacked{(volatile lds_u32*)&s_seq[al][0], offT})) continue;
    if (cells) cells[ti] = affine_cells(t, xs, oes, s_end);
    if (visited && lane == 0) atomicAdd(visited, (unsigned long long)slab_top);
  }
}

int gcd3(int a, int b, int c)
{
  auto g2 = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
  return g2(g2(a, b), c);
}

// ---- counting sort of the alignment list by score bound, largest first: the exact pass of an alignment costs ~ bound^2, and the
// persistent tier kernels hand alignments out in list order, so the longest run first and the tail of each kernel is short ones
constexpr int ASORT_BUCKETS = 512;
__device__ __forceinline__ int asort_bucket(int U)
{
  const int b = (U < 0 || U >= 0x40000000) ? ASORT_BUCKETS - 1 : (U >> 3);
  return ASORT_BUCKETS - 1 - (b < ASORT_BUCKETS ? b : ASORT_BUCKETS - 1);
}
__global__ __launch_bounds__(256) void K_asort_hist(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, uint32_t n_imm,
                                                    const int32_t* __restrict__ bound, uint32_t* __restrict__ hist)
{
  __shared__ uint32_t h[ASORT_BUCKETS];
  for (int b = (int)threadIdx.x; b < ASORT_BUCKETS; b += 256) h[b] = 0;
  __syncthreads();
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&h[asort_bucket(bound[list ? list[i] : i])], 1u);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < ASORT_BUCKETS; b += 256) if (h[b]) atomicAdd(&hist[b], h[b]);
}
__global__ __launch_bounds__(ASORT_BUCKETS) void K_asort_scan(uint32_t* __restrict__ hist)
{
  __shared__ uint32_t sc[ASORT_BUCKETS];
  const int i = (int)threadIdx.x;
  sc[i] = hist[i];
  __syncthreads();
  for (int off = 1; off < ASORT_BUCKETS; off <<= 1) {
    const uint32_t v = i >= off ? sc[i - off] : 0u;
    __syncthreads();
    sc[i] += v;
    __syncthreads();
  }
  hist[i] = sc[i] - hist[i];
}
__global__ __launch_bounds__(256) void K_asort_scatter(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr, uint32_t n_imm,
                                                       const int32_t* __restrict__ bound, uint32_t* __restrict__ pos, uint32_t* __restrict__ out)
{
  __shared__ uint32_t cnt[ASORT_BUCKETS], basep[ASORT_BUCKETS];
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (int b = (int)threadIdx.x; b < ASORT_BUCKETS; b += 256) cnt[b] = 0;
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&cnt[asort_bucket(bound[list ? list[i] : i])], 1u);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < ASORT_BUCKETS; b += 256) { basep[b] = cnt[b] ? atomicAdd(&pos[b], cnt[b]) : 0u; cnt[b] = 0; }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t ti = list ? list[i] : i;
    const int b = asort_bucket(bound[ti]);
    out[basep[b] + atomicAdd(&cnt[b], 1u)] = ti;
  }
}


// ---- register-resident tiers (v5): which tier takes an alignment follows from its score bound and shape alone — the same window
// arithmetic as the kernel — so one counting sort on (tier, bound) hands every tier its own list, longest alignments first
#ifndef OTG_V5_DEFAULT_MASK
#define OTG_V5_DEFAULT_MASK 31     /* measured: every register tier beats the LDS tier of its window (one-wave 1024 / 1536 / 2048 at 4 / 4 / 3 waves per SIMD, four-wave 4096, eight-wave 8192) */
#endif
#ifndef OTG_V5_DEFAULT_SHAPE
#define OTG_V5_DEFAULT_SHAPE 0
#endif
constexpr int V5_TIERS = 5;                                         // pair-slots 8 / 12 / 16 (one wave each): windows of 1024 / 1536 / 2048 diagonals; 4 waves x 8: 4096; 8 waves x 8: 8192
constexpr int TSORT_BUCKETS = (V5_TIERS + 1) * ASORT_BUCKETS;       // last tier = everything else (LDS / HBM tiers)
__device__ __forceinline__ int v5_tier(const otg_align_task& t, int U, int mask)
{
  const int pl = (int)t.pattern_len, tl = (int)t.text_len;
  if (U < 0 || U >= 0x40000000 || pl >= 32766 || tl >= 32766) return V5_TIERS;
  const bool ef = t.endsfree != 0;
  const int kend = tl - pl;
  const int elo = kend - (ef ? t.text_end_free : 0), ehi = kend + (ef ? t.pattern_end_free : 0);
  int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
  lo0 = imax(lo0, elo - U); hi0 = imin(hi0, ehi + U);
  if (hi0 < lo0) return V5_TIERS;
  const int wlo = imax((lo0 + elo - U) >> 1, -pl) - 1, whi = imin((hi0 + ehi + U + 1) >> 1, tl) + 1;
  const int need = whi - (wlo - 2) + 4;                             // the kernel wants need < CAP
  const int seqb = ((pl + 15) / 16 + 3 + (tl + 15) / 16 + 3) * 4;
  if ((mask & 1) && need < 1024 && seqb <= 4096) return 0;
  if ((mask & 2) && need < 1536 && seqb <= 4608) return 1;
  if ((mask & 4) && need < 2048 && seqb <= 6144) return 2;
  // the multi-wave tiers take what lies beyond the smaller windows (those stay with the one-wave / LDS tiers, which are faster on narrow rows)
  if ((mask & 8) && need < 4096 && seqb <= 8192 && (need >= 2048 || seqb > 3072)) return 3;
  if ((mask & 16) && need < 8192 && seqb <= 12288 && (need >= 4096 || seqb > 8192)) return 4;
  return V5_TIERS;
}
__device__ __forceinline__ int tsort_bucket(const otg_align_task& t, int U, int mask) { return v5_tier(t, U, mask) * ASORT_BUCKETS + asort_bucket(U); }
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
// exclusive scan of the bucket counts (one block) + the tier segment bounds seg[0 .. V5_TIERS + 1]
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
  if (t == 1023) seg[V5_TIERS + 1] = part[1023];
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
  if (x <= 0 || e <= 0 || o < 0) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties must satisfy x>0, o>=0, e>0");
  const int g = gcd3(x, o + e, e);
  const int xs = x / g, oes = (o + e) / g, es = e / g;
  if (std::max(xs, oes) + 1 > 64 || es + 1 > 64) return otg_fail(ctx, OTG_ERR_ARG, "affine penalties too large after gcd reduction");
  const bool fresh_cnt = ctx->pool[SLOT_COUNTERS].cap < 128 * sizeof(uint32_t);
  uint32_t* cnt = (uint32_t*)otg_slot(ctx, SLOT_COUNTERS, 128 * sizeof(uint32_t));
  uint32_t* todo = (uint32_t*)otg_slot(ctx, SLOT_TODO, 8 * (size_t)n_tasks * sizeof(uint32_t));
  if (!cnt || !todo) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(cnt + 8, 0, 8 * sizeof(uint32_t), ctx->stream));   // tickets / overflow counters of the tiers
  HIP_TRY(ctx, hipMemsetAsync(cnt + 24, 0, 8 * sizeof(uint32_t), ctx->stream));
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
  ws.dbg = (getenv("OTG_DEBUG") != nullptr ? 1 : 0) | (getenv("OTG_DBG_NO_BT") != nullptr ? 2 : 0);
  ws.visited = ctx->affine_visited;
  if (ws.dbg & 1) { const unsigned long long z[2] = {0, 0}; HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(otg_dbg_v4_cells), z, sizeof(z))); }
  size_t ring_bytes = (size_t)(ws.rm + 2 * ws.ri) * ws.capa * sizeof(int32_t);
  ws.off_rowtab = (ring_bytes + 255) & ~(size_t)255;
  ws.off_rev = (ws.off_rowtab + (size_t)ws.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
  ws.off_slab = (ws.off_rev + ws.rev_cap + 255) & ~(size_t)255;

  size_t free_b = 0, total_b = 0;
  HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  // a fixed share of the device's memory (not of what happens to be free: two contexts share a device in the dispatcher, and a budget that
  // follows the other context's allocations would resize this workspace batch after batch)
  const size_t budget = std::min<size_t>((size_t)(total_b * 0.2), (size_t)((free_b + ctx->pool[SLOT_WF_WS].cap) * 0.8));
  // Provenance slabs of the HBM-row tiers: what a diamond of the tier's window can hold, not more (these tiers take what the LDS tiers
  // cannot: windows beyond 4096 diagonals, bytes outside ACGT, alignments without a bound).  Workspaces are kept small on purpose: the
  // first launch on a fresh allocation pays for every gigabyte (2.98 s for 52 GB measured, scripts/cold_probe.py), and the dispatcher
  // starts fresh contexts per job.
  auto fit = [&](uint32_t& nwaves, size_t& slab) {
    while (nwaves > 1 && (ws.off_slab + slab) * (size_t)nwaves > budget) {
      if (slab > ((size_t)4 << 20)) slab /= 2; else nwaves = (nwaves + 1) / 2;
    }
  };
  // a diamond of bound U holds ~U^2 / 2 cells and U stays below ~0.45 x length at ONT divergence: 0.2 x maxlen^2 is twice that; an alignment
  // that needs more moves on to the next tier
  auto diamond_slab = [&](size_t window) { return std::min<size_t>((size_t)(0.2 * (double)maxlen * (double)maxlen), window * window * 5 / 8) + (1 << 16); };
  // tier A: v3, LDS window 4096 diagonals
  constexpr int NWA = 4;                     // waves cooperating on one alignment
  AffWs wsA = ws; uint32_t wavesA = (uint32_t)ctx->n_cu * 2; size_t slabA = diamond_slab(4096);   // blocks (one alignment each)
  fit(wavesA, slabA);
  wsA.slab_bytes = slabA & ~(size_t)255; wsA.stride = wsA.off_slab + wsA.slab_bytes;
  // tier B: v3, LDS window 12288 diagonals
  // (takes the longest reads of a 1-10 kb job, so it gets a full grid when the batch has reads beyond 8 kb)
  AffWs wsB = ws; uint32_t wavesB = maxlen > 8192 ? (uint32_t)ctx->n_cu * 3 : (uint32_t)ctx->n_cu / 2; size_t slabB = diamond_slab(12288);
  fit(wavesB, slabB);
  wsB.slab_bytes = slabB & ~(size_t)255; wsB.stride = wsB.off_slab + wsB.slab_bytes;
  // tier C: generic kernel (global int32 rings), a few waves with the largest useful slabs
  constexpr int WPB = 4;
  AffWs wsC = ws; uint32_t gridC = 2;
  size_t slabC = std::min<size_t>((size_t)2 * maxlen * (size_t)(ws.nrows), budget / (gridC * WPB));
  if (slabC > ws.off_slab + 256) slabC -= ws.off_slab + 256;
  wsC.slab_bytes = slabC & ~(size_t)255; wsC.stride = wsC.off_slab + wsC.slab_bytes;
  // LDS-resident tiers (v4): no rings, provenance slab for a diamond of at most CAP diagonals
  auto lds_ws = [&](int cap, uint32_t& blocks) {
    AffWs w = ws;
    w.off_rowtab = 0;
    w.off_rev = ((size_t)ws.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
    w.off_slab = (w.off_rev + ws.rev_cap + 255) & ~(size_t)255;
    size_t slab = (size_t)cap * (size_t)cap * 5 / 8 + (1 << 16);
    while (blocks > 1 && (w.off_slab + slab) * (size_t)blocks > budget) blocks = (blocks + 1) / 2;
    w.slab_bytes = slab & ~(size_t)255; w.stride = w.off_slab + w.slab_bytes;
    return w;
  };
  uint32_t blocksS = (uint32_t)ctx->n_cu * 10, blocksM = (uint32_t)ctx->n_cu * 5;   // resident blocks per CU (LDS / VGPR limits)
  uint32_t blocksX = (uint32_t)ctx->n_cu * 7;
  uint32_t blocksL = (uint32_t)ctx->n_cu * 2;
  AffWs wsS = lds_ws(1024, blocksS), wsM = lds_ws(2048, blocksM), wsX = lds_ws(1472, blocksX), wsL = lds_ws(4096, blocksL);
  const size_t need = std::max(std::max(std::max(wsA.stride * wavesA, wsB.stride * wavesB), wsC.stride * (size_t)gridC * WPB),
                               std::max(std::max(wsS.stride * blocksS, wsM.stride * blocksM), std::max(wsX.stride * blocksX, wsL.stride * blocksL)));
  uint8_t* wsp = (uint8_t*)otg_slot(ctx, SLOT_WF_WS, need);
  if (!wsp) return OTG_ERR_HIP;
  wsA.base = wsB.base = wsC.base = wsS.base = wsM.base = wsX.base = wsL.base = wsp;
  uint32_t* listA = todo;                  // overflow of tier A
  uint32_t* listB = todo + n_tasks;        // overflow of tier B
  uint32_t* listS = todo + 2 * (size_t)n_tasks;   // overflow of the LDS tier with 1024 diagonals
  uint32_t* listM = todo + 3 * (size_t)n_tasks;   // ... 2048 diagonals
  uint32_t* listX = todo + 4 * (size_t)n_tasks;   // ... 1472 diagonals
  uint32_t* listL = todo + 5 * (size_t)n_tasks;   // ... 4096 diagonals
  static const bool no_v3 = getenv("OTG_NO_AFFINE_V3") != nullptr;
  if (kernel_ms) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const uint32_t* cur = d_todo; const uint32_t* cur_n = d_n_todo; uint32_t cur_imm = n_tasks;
  const int32_t* d_bound_dbg = nullptr;
  if (!no_v3 && es == 1) {
    // score bound pass (register-resident banded run) for the default (4,6,2) -> (2,4,1) penalties
    static const bool no_bound = getenv("OTG_NO_AFFINE_BOUND") != nullptr;
    int32_t* d_bound = nullptr;
    if (!no_bound && xs == 2 && oes == 4) {
      d_bound = (int32_t*)otg_slot(ctx, SLOT_BT_POOL, (size_t)n_tasks * sizeof(int32_t));
      if (!d_bound) return OTG_ERR_HIP;
      const uint32_t want = (n_tasks + 3) / 4;
      const uint32_t gridU = std::min<uint32_t>((uint32_t)ctx->n_cu * 8, want);
      static const bool wide_band = getenv("OTG_AFFINE_BOUND_STATIC") != nullptr;
      if (wide_band)
        hipLaunchKernelGGL((wfa_affine_bound_kernel<4, 2, 4>), dim3(gridU), dim3(256), 0, ctx->stream, d_arena, d_tasks,
                           d_todo, d_n_todo, n_tasks, d_bound, cnt + 14);
      else
        hipLaunchKernelGGL((wfa_affine_bound1_kernel<2, 4>), dim3(gridU), dim3(256), 0, ctx->stream, d_arena, d_tasks,
                           d_todo, d_n_todo, n_tasks, d_bound, cnt + 14);
      d_bound_dbg = d_bound;
    }
    const uint32_t* inA = d_todo; const uint32_t* inA_n = d_n_todo; uint32_t inA_imm = n_tasks;
    static const bool no_v4 = getenv("OTG_NO_AFFINE_V4") != nullptr;
    if (d_bound && !no_v4) {
      static const int nws = getenv("OTG_V4_NWS") ? atoi(getenv("OTG_V4_NWS")) : 2;
      static const int nwm = getenv("OTG_V4_NWM") ? atoi(getenv("OTG_V4_NWM")) : 4;
      // alignments in decreasing order of their bound (work ~ bound^2): short tails in every tier kernel
      const uint32_t* inS = d_todo; const uint32_t* inS_n = d_n_todo; uint32_t inS_imm = n_tasks;
      static const bool no_asort = getenv("OTG_NO_AFFINE_SORT") != nullptr;
      // register-resident tiers (bit mask: 1 / 2 / 4 = the one-wave tiers of 1024 / 1536 / 2048 diagonals, 8 = the four-wave tier of 4096)
      static const int v5_mask_env = getenv("OTG_AFFINE_V5") ? atoi(getenv("OTG_AFFINE_V5")) : OTG_V5_DEFAULT_MASK;
      // the 8192 tier only when the batch can need it (reads beyond 4 kb): its slabs are the largest.  The sort sees the mask of the tiers that run.
      const int v5_mask = maxlen <= 4096 ? (v5_mask_env & ~16) : v5_mask_env;
      const bool no_v5 = v5_mask == 0;
      if (!no_v5) {
        // one counting sort on (tier, bound) gives each tier its list; what no tier takes (or a tier gives up) is the input of the LDS / HBM tiers below
        uint32_t* hist = (uint32_t*)otg_slot(ctx, SLOT_ROWTAB, TSORT_BUCKETS * sizeof(uint32_t));
        uint32_t* sorted = todo + 6 * (size_t)n_tasks;
        uint32_t* ovf5 = todo + 7 * (size_t)n_tasks;
        uint32_t* seg = cnt + 64;               // seg[0 .. V5_TIERS + 1]
        uint32_t* n_ovf5 = cnt + 71;
        if (!hist) return OTG_ERR_HIP;
        HIP_TRY(ctx, hipMemsetAsync(hist, 0, TSORT_BUCKETS * sizeof(uint32_t), ctx->stream));
        const uint32_t sg = std::min<uint32_t>((n_tasks + 2047) / 2048, (uint32_t)ctx->n_cu * 2);
        hipLaunchKernelGGL(K_tsort_hist, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, d_tasks, (const int32_t*)d_bound, hist, v5_mask);
        hipLaunchKernelGGL(K_tsort_scan, dim3(1), dim3(1024), 0, ctx->stream, hist, seg);
        hipLaunchKernelGGL(K_tsort_scatter, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, d_tasks, (const int32_t*)d_bound, hist, sorted, v5_mask);
        hipLaunchKernelGGL(K_seg_copy, dim3(std::min<uint32_t>((n_tasks + 255) / 256, 1024u)), dim3(256), 0, ctx->stream, (const uint32_t*)sorted,
                           (const uint32_t*)(seg + V5_TIERS), ovf5, n_ovf5);
        // workspaces of the tiers side by side; row table sized by the window (a score is a row); `units` = alignments in flight
        auto v5_ws = [&](int cap, uint32_t units) {
          AffWs w = ws;
          w.nrows = cap + 64;
          w.off_rowtab = 0;
          w.off_rev = ((size_t)w.nrows * sizeof(int64_t) + 255) & ~(size_t)255;
          w.off_slab = (w.off_rev + ws.rev_cap + 255) & ~(size_t)255;
          w.slab_bytes = ((size_t)cap * (size_t)cap * 7 / 8 + (1 << 16)) & ~(size_t)255;
          w.stride = w.off_slab + w.slab_bytes;
          (void)units;
          return w;
        };
        // tier shapes: waves per alignment x pair-slots per wave (experiment switch OTG_V5_SHAPE = four digits, one per tier, 0 = the first shape)
        static const int shape = getenv("OTG_V5_SHAPE") ? atoi(getenv("OTG_V5_SHAPE")) : OTG_V5_DEFAULT_SHAPE;
        const int sh0 = (shape / 1000) % 10, sh1 = (shape / 100) % 10, sh2 = (shape / 10) % 10, sh3 = shape % 10;
        // alignments in flight per tier (blocks x alignments per block) size the workspaces
        const uint32_t ncu = (uint32_t)ctx->n_cu;
        const uint32_t bl0 = !(v5_mask & 1) ? 0 : (sh0 == 0 ? ncu * 4 : ncu * 8), al0 = sh0 == 0 ? bl0 * 4 : bl0;
        const uint32_t bl1 = !(v5_mask & 2) ? 0 : (sh1 == 0 ? ncu * 4 : ncu * 8), al1 = sh1 == 0 ? bl1 * 4 : bl1;
        const uint32_t bl2 = !(v5_mask & 4) ? 0 : (sh2 == 0 ? ncu * 3 : (sh2 == 1 ? ncu * 8 : ncu * 4)), al2 = sh2 == 0 ? bl2 * 4 : bl2;
        uint32_t bl3 = !(v5_mask & 8) ? 0 : (sh3 == 0 ? ncu * 4 : (sh3 == 2 ? ncu * 6 : ncu * 2)), al3 = bl3;
        uint32_t bl4 = !(v5_mask & 16) ? 0 : ncu * 2, al4 = bl4;
        AffWs w0 = v5_ws(1024, al0), w1 = v5_ws(1536, al1), w2 = v5_ws(2048, al2), w3 = v5_ws(4096, al3), w4 = v5_ws(8192, al4);
        w4.slab_bytes = std::min<size_t>(w4.slab_bytes, ((size_t)(0.2 * (double)maxlen * (double)maxlen) + (1 << 20)) & ~(size_t)255); w4.stride = w4.off_slab + w4.slab_bytes;
        // the same share of the device as the tiers behind: beyond it the two widest tiers keep fewer alignments in flight (reads beyond ~16 kb only)
        auto total5 = [&]() { return w0.stride * al0 + w1.stride * al1 + w2.stride * al2 + w3.stride * al3 + w4.stride * al4 + 256; };
        const size_t budget5 = std::min<size_t>((size_t)(total_b * 0.25), (size_t)((free_b + ctx->pool[SLOT_REVOPS].cap) * 0.8));     // five tiers side by side: a quarter of the device
        while (total5() > budget5 && (bl4 > ncu / 4 || bl3 > ncu / 2)) {
          if (bl4 > ncu / 4 && (w4.stride * al4 >= w3.stride * al3 || bl3 <= ncu / 2)) { bl4 /= 2; al4 = bl4; } else { bl3 /= 2; al3 = bl3; }
        }
        const size_t need5 = total5();
        if (ws.dbg & 1) fprintf(stderr, "[otg] affine: register tiers keep %u / %u / %u / %u / %u alignments in flight, workspaces %.1f GB of a budget of %.1f GB\n", al0, al1, al2, al3, al4, (double)need5 / 1e9, (double)budget5 / 1e9);
        uint8_t* ws5 = (uint8_t*)otg_slot(ctx, SLOT_REVOPS, need5);
        if (!ws5) return OTG_ERR_HIP;
        w0.base = ws5; w1.base = w0.base + w0.stride * al0; w2.base = w1.base + w1.stride * al1; w3.base = w2.base + w2.stride * al2; w4.base = w3.base + w3.stride * al3;
        unsigned long long* vis = ctx->affine_visited;
#define OTG_V5_LAUNCH(NWV, S2V, SEQV, WPEUV, BLOCKS, SEGI, TICK, WS)                                                                      \
        hipLaunchKernelGGL((wfa_affine_kernel_v5<NWV, S2V, SEQV, WPEUV>), dim3(BLOCKS), dim3(NWV == 1 ? 256 : NWV * 64), 0, ctx->stream, d_arena, d_tasks, \
                           (const uint32_t*)sorted, (const uint32_t*)(seg + SEGI), g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, cnt + TICK, n_ovf5, ovf5, \
                           WS, (const int32_t*)d_bound, vis)
        if (bl0) { if (sh0 == 0) OTG_V5_LAUNCH(1, 8, 4096, 4, bl0, 0, 72, w0); else OTG_V5_LAUNCH(2, 4, 4096, 4, bl0, 0, 72, w0); }
        if (bl1) { if (sh1 == 0) OTG_V5_LAUNCH(1, 12, 4608, 4, bl1, 1, 73, w1); else OTG_V5_LAUNCH(2, 6, 4608, 4, bl1, 1, 73, w1); }
        if (bl2) { if (sh2 == 0) OTG_V5_LAUNCH(1, 16, 6144, 3, bl2, 2, 74, w2); else if (sh2 == 1) OTG_V5_LAUNCH(2, 8, 6144, 4, bl2, 2, 74, w2); else OTG_V5_LAUNCH(4, 4, 6144, 4, bl2, 2, 74, w2); }
        if (bl3) { if (sh3 == 0) OTG_V5_LAUNCH(4, 8, 8192, 4, bl3, 3, 75, w3); else if (sh3 == 2) OTG_V5_LAUNCH(2, 16, 8192, 3, bl3, 3, 75, w3); else OTG_V5_LAUNCH(8, 4, 8192, 4, bl3, 3, 75, w3); }
        if (bl4) OTG_V5_LAUNCH(8, 8, 12288, 4, bl4, 4, 76, w4);
#undef OTG_V5_LAUNCH
        inS = ovf5; inS_n = n_ovf5; inS_imm = 0;
      } else if (!no_asort) {
        uint32_t* hist = (uint32_t*)otg_slot(ctx, SLOT_ROWTAB, ASORT_BUCKETS * sizeof(uint32_t));
        uint32_t* sortedU = todo + 6 * (size_t)n_tasks;
        if (!hist) return OTG_ERR_HIP;
        HIP_TRY(ctx, hipMemsetAsync(hist, 0, ASORT_BUCKETS * sizeof(uint32_t), ctx->stream));
        const uint32_t sg = std::min<uint32_t>((n_tasks + 2047) / 2048, (uint32_t)ctx->n_cu * 2);
        hipLaunchKernelGGL(K_asort_hist, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, (const int32_t*)d_bound, hist);
        hipLaunchKernelGGL(K_asort_scan, dim3(1), dim3(ASORT_BUCKETS), 0, ctx->stream, hist);
        hipLaunchKernelGGL(K_asort_scatter, dim3(sg), dim3(256), 0, ctx->stream, d_todo, d_n_todo, n_tasks, (const int32_t*)d_bound, hist, sortedU);
        inS = sortedU;
      }
#define OTG_V4_LAUNCH(CAPV, NWV, SEQV, WPEUV, BLOCKS, TODO, NTODO, IMM, TICK, OVF, LIST, WS)                                 \
      hipLaunchKernelGGL((wfa_affine_kernel_v4<CAPV, 256, NWV, SEQV, WPEUV>), dim3(BLOCKS), dim3(NWV * 64), 0, ctx->stream, d_arena, d_tasks, \
                         TODO, NTODO, IMM, g, d_scores, d_cig_off, d_cig_len, d_cig_arena, d_cells, TICK, OVF, LIST, WS,       \
                         (const int32_t*)d_bound)
      static const bool no_mid = getenv("OTG_V4_NO_MID") != nullptr;
      if (nws == 1) OTG_V4_LAUNCH(1024, 1, 2304, 3, blocksS, inS, inS_n, inS_imm, cnt + 24, cnt + 25, listS, wsS);
      else OTG_V4_LAUNCH(1024, 2, 2304, 5, blocksS, inS, inS_n, inS_imm, cnt + 24, cnt + 25, listS, wsS);
      const uint32_t* inM = listS; const uint32_t* inM_n = cnt + 25;
      if (!no_mid) {
        OTG_V4_LAUNCH(1472, 4, 2688, 7, blocksX, (const uint32_t*)listS, (const uint32_t*)(cnt + 25), 0u, cnt + 28, cnt + 29, listX, wsX);
        inM = listX; inM_n = cnt + 29;
      }
      if (nwm == 4) OTG_V4_LAUNCH(2048, 4, 3072, 5, blocksM, inM, inM_n, 0u, cnt + 26, cnt + 27, listM, wsM);
      else OTG_V4_LAUNCH(2048, 2, 3072, 3, blocksM, inM, inM_n, 0u, cnt + 26, cnt + 27, listM, wsM);
      const uint32_t* outM = listM; const uint32_t* outM_n = cnt + 27;
      // diamonds of up to 4096 diagonals: eight waves per alignment, two alignments per CU
      static const bool no_l = getenv("OTG_V4_NO_L") != nullptr;
      if (!no_l) {
        OTG_V4_LAUNCH(4096, 8, 6144, 2, blocksL, outM, outM_n, 0u, cnt + 30, cnt + 31, listL, wsL);
        outM = listL; outM_n = cnt + 31;
      }
#undef OTG_V4_LAUNCH
      inA = outM; inA_n = outM_n; inA_imm = 0;
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
  if (getenv("OTG_DEBUG")) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t h[32];
    HIP_TRY(ctx, hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[otg] affine: LDS tiers overflow %u / %u, tier A overflow %u, tier B overflow %u\n", h[25], h[27], h[9], h[11]);
    { uint32_t h5[8]; HIP_TRY(ctx, hipMemcpy(h5, cnt + 64, sizeof(h5), hipMemcpyDeviceToHost));
      fprintf(stderr, "[otg] affine: register tiers take %u / %u / %u / %u / %u alignments, %u go to the LDS tiers (of which given up by a register tier: %u)\n",
              h5[1] - h5[0], h5[2] - h5[1], h5[3] - h5[2], h5[4] - h5[3], h5[5] - h5[4], h5[7], h5[7] - (h5[6] - h5[5])); }
    {
      unsigned long long vc[2] = {0, 0};
      HIP_TRY(ctx, hipMemcpyFromSymbol(vc, HIP_SYMBOL(otg_dbg_v4_cells), sizeof(vc)));
      fprintf(stderr, "[otg] affine: LDS tiers finished %llu alignments over %llu visited cells (%.0f per alignment)\n", vc[1], vc[0], vc[1] ? (double)vc[0] / (double)vc[1] : 0.0);
    }
    if (d_bound_dbg) {
      std::vector<int32_t> hb(n_tasks), hs(n_tasks);
      HIP_TRY(ctx, hipMemcpy(hb.data(), d_bound_dbg, (size_t)n_tasks * 4, hipMemcpyDeviceToHost));
      HIP_TRY(ctx, hipMemcpy(hs.data(), d_scores, (size_t)n_tasks * 4, hipMemcpyDeviceToHost));
      std::vector<otg_align_task> ht(n_tasks);
      HIP_TRY(ctx, hipMemcpy(ht.data(), d_tasks, (size_t)n_tasks * sizeof(otg_align_task), hipMemcpyDeviceToHost));
      uint64_t n_inf = 0, n_eq = 0, n_fin = 0, n_inf_band_ok = 0; double su = 0, ss = 0;
      std::vector<uint32_t> idx;
      if (d_todo) {
        uint32_t nt = 0;
        HIP_TRY(ctx, hipMemcpy(&nt, d_n_todo, 4, hipMemcpyDeviceToHost));
        idx.resize(nt);
        HIP_TRY(ctx, hipMemcpy(idx.data(), d_todo, (size_t)nt * 4, hipMemcpyDeviceToHost));
      } else { idx.resize(n_tasks); for (uint32_t i = 0; i < n_tasks; ++i) idx[i] = i; }
      uint64_t hist[8] = {0};   // excess U - s: 0, 1-2, 3-5, 6-10, 11-20, 21-50, 51-100, >100
      for (uint32_t i : idx) {
        if (hs[i] < 0) continue;
        if (hb[i] < 0x40000000) { const int ex = hb[i] - hs[i] / g; ++hist[ex <= 0 ? 0 : ex <= 2 ? 1 : ex <= 5 ? 2 : ex <= 10 ? 3 : ex <= 20 ? 4 : ex <= 50 ? 5 : ex <= 100 ? 6 : 7]; }
        if (hb[i] >= 0x40000000) {
          ++n_inf;
          const otg_align_task& t = ht[i];
          const int pl = (int)t.pattern_len, tl = (int)t.text_len; const bool ef = t.endsfree != 0;
          const int pef = ef ? t.pattern_end_free : 0, tef = ef ? t.text_end_free : 0, kend = tl - pl;
          const int lo0 = ef ? std::max(-(int)t.pattern_begin_free, -pl) : 0, hi0 = ef ? std::min((int)t.text_begin_free, tl) : 0;
          const int nl = std::min(lo0, kend - tef), nh = std::max(hi0, kend + pef);
          if (nh - nl + 1 + 64 <= 256) { if (n_inf_band_ok < 5) fprintf(stderr, "[otg]   unbounded although band ok: pl %d tl %d ef %d pbf %d pef %d tbf %d tef %d score %d\n", pl, tl, (int)ef, (int)t.pattern_begin_free, pef, (int)t.text_begin_free, tef, hs[i]); ++n_inf_band_ok; }
          continue;
        }
        ++n_fin; su += hb[i]; ss += hs[i] / g; if (hb[i] == hs[i] / g) ++n_eq;
      }
      fprintf(stderr, "[otg] affine bound: %llu unbounded, %llu bounded (%llu tight), %llu unbounded with band ok, mean U %.1f mean score %.1f (units of g)\n",
              (unsigned long long)n_inf, (unsigned long long)n_fin, (unsigned long long)n_eq, (unsigned long long)n_inf_band_ok, n_fin ? su / n_fin : 0.0, n_fin ? ss / n_fin : 0.0);
      fprintf(stderr, "[otg] affine bound excess histogram (0, 1-2, 3-5, 6-10, 11-20, 21-50, 51-100, >100): %llu %llu %llu %llu %llu %llu %llu %llu\n",
              (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3],
              (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
    }
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
