// wfa_affine_common.hpp — device code shared by the gap-affine wavefront kernels (wfa_affine.hip: the generic and the HBM-row tiers, the
// score-bound pass, the launch chain; wfa_affine_reg.hip: the register-resident tiers).  Replaces wfa::WFAlignerGapAffine(x, o, e,
// Alignment, MemoryMed)::alignEnd2End / alignEndsFree + getAlignmentCigar() (reference: src/assemble.cpp:50; call sites
// src/analignments.cpp:25,31,37,268-280).
//
// Provenance: one byte per (score, diagonal) cell — bits 0-1 the origin of M (0 mismatch, 1 deletion, 2 insertion), bit 2 "I came from an
// extension", bit 3 "D came from an extension" — the piggy-back rule of WFA2 (ext >= open; M provenance
// tested in the order ins, del, mism, so mismatch wins ties over deletion over insertion).  Rows are bump-allocated in a per-alignment
// slab and addressed through a row table (row s: byte of diagonal k at slab + rowtab[s] + k).  Two layouts of bits 2-3:
//   plain    — the byte of cell (s, k) tells where I[s][k] and D[s][k] came from;
//   shifted  — the byte of cell (s, k) tells what the cell offers its neighbours: bit 2 = I[s-e][k] >= M[s-o-e][k] (the choice made by
//              I[s][k+1]), bit 3 = D[s-e][k] >= M[s-o-e][k] (the choice made by D[s][k-1]).  The register tiers decide both with packed
//              16-bit operations on the cell's own words before anything is shifted to the neighbours.
#pragma once
#include "otg_common.hpp"
#include <type_traits>

namespace otg_affine {

struct AffWs {
  uint8_t* base;        // per-alignment workspaces, contiguous
  size_t stride;        // bytes per alignment in flight
  size_t off_rowtab, off_rev, off_slab;
  size_t slab_bytes;
  int capa;             // diagonals per ring row
  int rm, ri;           // ring depths
  int nrows;            // row-table entries
  size_t rev_cap;
  int dbg;              // OTG_DEBUG switches
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
// kernel keeps the sequences there).  SHIFTED: the provenance layout of the register tiers (see the head of this file).
constexpr int BT_ROWS = 64, BT_COLS = 32;
template <bool SHIFTED, class Eq>
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
    constexpr int MARGIN = SHIFTED ? 1 : 0;      // the shifted layout also reads the bytes of the two neighbouring diagonals
    while (s > 0 || comp != 0) {
      if (ws0 < 0 || ws0 - s >= BT_ROWS || s > ws0 || k - MARGIN < wk0 || k + MARGIN >= wk0 + BT_COLS) {
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
      const int at = (ws0 - s) * BT_COLS + (k - wk0);
      uint8_t op;
      if (comp == 0) {
        const uint32_t org = (uint32_t)win8[at] & 3u;
        if (org == 0) { op = 'X'; s -= xs; }
        else if (org == 1) { op = 'c'; comp = 2; }
        else { op = 'c'; comp = 1; }
      } else if (comp == 1) {
        op = 'I';
        const uint32_t bits = win8[SHIFTED ? at - 1 : at];
        if (bits & 4u) s -= es; else { s -= oes; comp = 0; }
        k -= 1;
      } else {
        op = 'D';
        const uint32_t bits = win8[SHIFTED ? at + 1 : at];
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

// W_p = 3 * sum of the wavefront widths the un-bounded aligner evaluates.  The ranges follow from the
// lengths, the free ends and the penalties alone: score 0 spans [lo0, hi0]; the next reachable score is
// f = min(x, o+e) (same span, widened by one on both sides when it is a gap open); from then on the previous score's
// I/D wavefronts widen the range by one diagonal per side and score (gap extension 1), clipped to [-pl, tl].
__device__ inline uint64_t affine_cells(const otg_align_task& t, int xs, int oes, int s_end)
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

// Window arithmetic shared by the tier selection (counting sort) and the register-tier kernels: the diamond of cells that can lie on an
// alignment of score <= U spans the diagonals [wlo, whi]; `need` diagonals of window are wanted (the kernels test need < CAP).
__device__ __forceinline__ bool affine_window(const otg_align_task& t, int U, int* kbase, int* need, int* lo0_out, int* hi0_out)
{
  const int pl = (int)t.pattern_len, tl = (int)t.text_len;
  const bool ef = t.endsfree != 0;
  const int kend = tl - pl;
  const int elo = kend - (ef ? t.text_end_free : 0), ehi = kend + (ef ? t.pattern_end_free : 0);
  int lo0 = ef ? imax(-t.pattern_begin_free, -pl) : 0, hi0 = ef ? imin(t.text_begin_free, tl) : 0;
  lo0 = imax(lo0, elo - U); hi0 = imin(hi0, ehi + U);
  const int wlo = imax((lo0 + elo - U) >> 1, -pl) - 1, whi = imin((hi0 + ehi + U + 1) >> 1, tl) + 1;
  *kbase = wlo - 2;
  *need = whi - (wlo - 2) + 4;
  *lo0_out = lo0; *hi0_out = hi0;
  return hi0 >= lo0;
}

} // namespace otg_affine
