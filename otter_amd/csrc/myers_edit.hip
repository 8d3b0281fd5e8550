// myers_edit.hip — banded bit-parallel edit distance (Myers 1999 / Hyyrö 2003 block formulation), gfx950.
//
// Second engine behind otg_edit_distance_batch / the pipeline's distance stages.  Unit-cost edit distance has a
// unique optimum, so any exact algorithm returns what WFAlignerEdit::getAlignmentScore() returns
// (reference call sites src/analignments.cpp:70-71,88-97).  The wavefront kernel (wfa_edit.hip) costs O(s^2) and
// is unbeatable for HiFi-like pairs (s ~ 0..50); for ONT-error reads (s = 0.1..0.4 L) this kernel costs
// O(n) wave-steps regardless of s.
//
// Mapping: a GROUP of GL lanes (16, 32 or 64) per pair, 64/GL pairs per wave.  The DP matrix (pattern = rows,
// text = columns) is cut into 64-row blocks held as vertical-delta bit-vectors (Pv, Mv: 64-bit).  Lane l of a
// group owns "superblocks" l, l+GL, ... (BPL consecutive blocks each) and at time step t processes column
// j = t - B of its superblock B: the anti-diagonal skew turns the block-to-block carry (hout -> hin) into a
// one-lane rotate per step (DPP row_ror for 16-lane groups, ds_bpermute for 32, wave_ror for 64).  Only the
// Ukkonen band -KL <= i - j <= KU (KL = (K - d)/2, KU = (K + d)/2 for a global alignment, d = m - n >= 0) is evaluated; a superblock enters the band initialised as in
// Edlib (Pv = ~0, score = score_above + rows) and the block at the top of the band takes hin = +1.  With
// KL + KU <= (GL-1)*64*BPL + GL (i.e. K up to about that many rows) a lane has left its superblock before the next one (B + GL) enters the band, so
// narrow bands (within-allele pairs) run four to a wave and only wide ones need the whole wave.
// The computed score is exact iff it is <= K; otherwise the task is appended to the overflow list and handled
// by the next tier (larger group / more blocks per lane, finally the wavefront kernel).
// Pattern match masks (A, C, G, T + at most one further byte value occurring in the pattern, e.g. N) are built
// once per pair into an L2-resident scratch and fetched when a lane moves to its next superblock; richer
// alphabets, text-side free ends and patterns > 16384 bytes go to the wavefront kernel.
#include "otg_common.hpp"
#include <cstdlib>

namespace {

constexpr int MAXBLK = 256;           // 64-row blocks per pattern (m <= 16384)
using u64 = unsigned long long;

__device__ __forceinline__ u64 load8(const uint8_t* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }

// lane i <- lane i-1 within its group of GL lanes (wrap-around)
template <int GL>
__device__ __forceinline__ int group_ror1(int x, int lane)
{
  if (GL == 64) return __builtin_amdgcn_update_dpp(x, x, 0x13C, 0xf, 0xf, false);   // wave_ror:1
  if (GL == 16) return __builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false);   // row_ror:1
  const int src = (lane & ~(GL - 1)) | ((lane - 1) & (GL - 1));
  return __builtin_amdgcn_ds_bpermute(src << 2, x);
}

// W_p of SURVEY.md §8d for a finished alignment with score s (what the wavefront aligner would have evaluated):
// per score t the diagonals [max(-pbf - t, -m), min(tbf + t, n)]; the lanes of the group stride over t.
template <int GL>
__device__ u64 wfa_cells(int s, int m, int n, int pbf, int tbf, int gl)
{
  u64 w = 0;
  for (int t = gl; t <= s; t += GL) {
    const int lo = -pbf - t < -m ? -m : -pbf - t;
    const int hi = tbf + t > n ? n : tbf + t;
    w += (u64)(hi - lo + 1);
  }
  for (int off = GL / 2; off > 0; off >>= 1) w += __shfl_xor(w, off, GL);
  return w;
}

template <int BPL, int GL, int WPB>
__global__ __launch_bounds__(WPB * 64) void myers_edit_kernel(
    const uint8_t* __restrict__ arena, const otg_align_task* __restrict__ tasks,
    const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo_ptr, uint32_t n_todo_imm,
    int32_t* __restrict__ scores, uint64_t* __restrict__ cells,
    uint32_t* __restrict__ ticket, uint32_t* __restrict__ n_overflow, uint32_t* __restrict__ overflow_list,
    u64* __restrict__ peq_ws, int maxblk)
{
  constexpr int G = 64 / GL;            // pairs per wave
  constexpr int SB = 64 * BPL;
  constexpr int NT = WPB * 64;
  // Match-mask table of the lane's CURRENT superblock: row (sym * BPL + q), column = thread; a lane only ever reads
  // its own column (same-wave program order, no barrier) and the row stride is a multiple of the bank count, so the
  // per-step fetch `row(symbol of this column)` is one conflict-free ds_read_b64 instead of a select tree.
  // sym: 0..3 = A C T G (code (byte >> 1) & 3), 4 = the one extra byte value of the pattern, 5 = anything else (zero).
  __shared__ u64 s_eq[6 * BPL][NT];
  // byte -> row index (sym * BPL) per lane group; text bytes are translated when a lane loads its next 8 columns
  __shared__ uint8_t s_lut[WPB * G][256];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int gl = lane & (GL - 1);       // lane within its group
  const int grp = lane / GL;
  u64* peq = peq_ws + ((size_t)(blockIdx.x * WPB + wib) * G + grp) * (size_t)maxblk * 5;
  const uint32_t n_todo = n_todo_ptr ? *n_todo_ptr : n_todo_imm;
  using lds_u64 = __attribute__((address_space(3))) u64;
  using lds_u8 = __attribute__((address_space(3))) uint8_t;
  volatile lds_u64* EQ = (volatile lds_u64*)&s_eq[0][0] + threadIdx.x;          // this thread's column; row r at EQ[r * NT]
  volatile lds_u8* LUT = (volatile lds_u8*)&s_lut[wib * G + grp][0];
  {
    // static part of the tables: zero rows, ACGT entries of the byte map (the extra symbol is patched per pair)
#pragma unroll
    for (int q = 0; q < BPL; ++q) EQ[(5 * BPL + q) * NT] = 0ull;
    for (int c = gl; c < 256; c += GL) {
      const uint32_t code = ((uint32_t)c >> 1) & 3u;
      LUT[c] = (uint8_t)(((uint32_t)c == ((0x47544341u >> (8 * code)) & 0xffu) ? code : 5u) * BPL);
    }
  }
  int lut_other = -1;                    // byte currently mapped to row 4 in this group's map

  for (;;) {
    const uint32_t tk0 = otg_wave_atomic_add(ticket, (uint32_t)G);
    if (tk0 >= n_todo) break;
    const uint32_t tk = tk0 + (uint32_t)grp;
    const bool has_task = tk < n_todo;
    const uint32_t ti = has_task ? (todo ? todo[tk] : tk) : (todo ? todo[tk0] : tk0);
    const otg_align_task tsk = tasks[ti];
    const bool ef = tsk.endsfree != 0;
    const uint8_t* P = arena + tsk.pattern_off;
    const uint8_t* T = arena + tsk.text_off;
    int m = (int)tsk.pattern_len, n = (int)tsk.text_len;
    int pbf = ef ? tsk.pattern_begin_free : 0, pef = ef ? tsk.pattern_end_free : 0;
    bool unsupported = ef && (tsk.text_begin_free != 0 || tsk.text_end_free != 0);
    if (!ef && m < n) { const uint8_t* q = P; P = T; T = q; const int x = m; m = n; n = x; }   // edit distance is symmetric
    if (m < n) unsupported = true;
    if (pbf > m) pbf = m;
    if (pef > m) pef = m;
    const int d = m - n;
    const int nblk = (m + 63) >> 6;
    const int nsb = (m + SB - 1) / SB;
    // Ukkonen band for the largest threshold K this lane schedule (R rows) can certify, see otg_myers_band
    // R: a lane must have left superblock B (last step SB*B + SB-1 + KL + B) before B + GL enters the band
    // (first step SB*(B+GL) - KU + B + GL)  <=>  KL + KU <= SB*(GL-1) + GL
    constexpr int R = (GL - 1) * SB + GL;
    const int K = otg_myers_threshold(R, d, pbf, pef);
    int KL, KU;
    otg_myers_band(K, d, pbf, pef, &KL, &KU);
    if (nblk > maxblk || K < d - pbf - pef || K < 1 || KL + KU > R || n < 1 || pbf > d || pef > d) unsupported = true;
    if (!has_task) unsupported = true;

    // ---- pattern match masks per 64-row block: A, C, G, T, X (one further byte value), built by the group
    int other = -1;
    bool bad_alpha = false;
    if (!unsupported) {
      for (int b = gl; b < nblk; b += GL) {
        u64 ea = 0, ec = 0, eg = 0, et = 0;
        const int base = b << 6;
        for (int r = 0; r < 64; ++r) {
          const int i = base + r;
          if (i >= m) break;
          const uint8_t ch = P[i];
          const u64 bit = 1ull << r;
          if (ch == 'A') ea |= bit; else if (ch == 'C') ec |= bit; else if (ch == 'G') eg |= bit; else if (ch == 'T') et |= bit;
          else { if (other < 0) other = ch; else if (other != ch) bad_alpha = true; }
        }
        peq[b * 5 + 0] = ea; peq[b * 5 + 1] = ec; peq[b * 5 + 2] = eg; peq[b * 5 + 3] = et;
      }
    }
    // agree on the single extra symbol across the lanes of the group
    {
      const u64 gmask = (GL == 64) ? ~0ull : (((1ull << GL) - 1ull) << (grp * GL));
      const u64 has = __ballot(other >= 0) & gmask;
      int x = -1;
      const int src = has ? (int)__builtin_ctzll(has) : lane;
      const int xo = __shfl(other, src);
      if (has) x = xo;
      const u64 bad = __ballot(bad_alpha || (other >= 0 && other != x)) & gmask;
      if (bad) unsupported = true;
      other = x;
      if (!unsupported) {
        for (int b = gl; b < nblk; b += GL) {
          u64 ex = 0;
          if (x >= 0) {
            const int base = b << 6;
            for (int r = 0; r < 64; ++r) { const int i = base + r; if (i >= m) break; if (P[i] == (uint8_t)x) ex |= 1ull << r; }
          }
          peq[b * 5 + 4] = ex;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (other != lut_other) {            // group-uniform
      if (gl == 0) {
        if (lut_other >= 0) LUT[lut_other] = (uint8_t)(5 * BPL);
        if (other >= 0) LUT[other] = (uint8_t)(4 * BPL);
      }
      lut_other = other;
    }

    // ---- skewed sweep
    int B = gl;                         // current superblock of this lane
    if (!unsupported) while (B < nsb && SB * B + SB - 1 + KL < 0) B += GL;      // (KL < 0: the band starts below diagonal 0, the superblocks above it never enter it)
    bool inited = false;
    u64 Pv[BPL], Mv[BPL];
#pragma unroll
    for (int q = 0; q < BPL; ++q) { Pv[q] = ~0ull; Mv[q] = 0; }
    int score = 0, hout = 0;
    int best = 0x3fffffff;
    const int i_lo = m - pef;           // the answer is min over rows i in [i_lo, m] of D[i][n]
    if (i_lo <= 0) best = n;            // D[0][n] = n
    const int t_end = unsupported ? -1 : n - 1 + nsb - 1;
    // per-superblock time window (recomputed only when the lane moves to its next superblock)
    int t_start, t_stop, t_hin_stop, t_last; bool exact_init;
    auto setup = [&]() {
      int jlo = SB * B - KU; if (jlo < 0) jlo = 0;
      int jhi = SB * B + SB - 1 + KL; if (jhi > n - 1) jhi = n - 1;
      exact_init = (jlo == 0);
      if (!unsupported && B < nsb && jlo <= jhi) {
        t_start = jlo + B; t_stop = jhi + B;
        int jh = SB * B - 1 + KL; if (jh > jhi) jh = jhi;
        t_hin_stop = B > 0 ? jh + B : -1;              // block above still inside the band
        t_last = (jhi == n - 1) ? n - 1 + B : -1;
      } else { t_start = 0x7fffffff; t_stop = 0x7ffffffe; t_hin_stop = -1; t_last = -1; }
    };
    setup();
    // text bytes: one unaligned 8-byte load per 8 steps and lane (prefetched 4 steps ahead); byte (t & 7) of c8
    // is column (t & ~7) - B + (t & 7) = t - B
    auto load_group = [&](int tg) -> u64 {              // tg = first step of the group; returns 8 row indices
      int a = tg - B;
      if (a > n - 1) a = n - 1;
      u64 x;
      if (a >= 0) x = load8(T + a);
      else { const int sh = -a; x = sh < 8 ? (load8(T) << (8 * sh)) : 0ull; }
      uint32_t r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = LUT[(uint32_t)(x >> (8 * i)) & 0xffu];
      const uint32_t lo = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24), hi = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
      return (u64)lo | ((u64)hi << 32);
    };
    u64 c8 = load_group(0), c8n = 0;
    // the wave runs until its longest pair is done
    int t_end_w = t_end;
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(t_end_w, off); t_end_w = o > t_end_w ? o : t_end_w; }
    t_end_w = __builtin_amdgcn_readfirstlane(t_end_w);
    // one column per step; unrolled by the 8 columns of a translated text group so that every byte position is static
    auto step = [&](const int t, const int ph) {
      // values of lane-1 (within the group) after its previous step
      const int up_score = group_ror1<GL>(score, lane);
      const int up_hout = group_ror1<GL>(hout, lane);
      bool switched = false;
      if (t > t_stop) {                 // this superblock left the band: move to the next one owned by the lane
        B += GL; inited = false; setup();
        c8 = load_group(t - ph);
        if (ph >= 4) c8n = load_group(t - ph + 8);
        switched = true;
      }
      if (ph == 4) c8n = load_group(t + 4);
      else if (ph == 0 && t > 0 && !switched) c8 = c8n;     // (after a switch c8 is already this group's text; c8n still belongs to the old superblock)
      if (t >= t_start && t <= t_stop) {
        if (!inited) {
#pragma unroll
          for (int q = 0; q < BPL; ++q) {
            const int b = B * BPL + q;
            const bool have = b < nblk;
#pragma unroll
            for (int y = 0; y < 5; ++y) EQ[(y * BPL + q) * NT] = have ? peq[b * 5 + (y == 2 ? 3 : y == 3 ? 2 : y)] : 0ull;   // rows A C T G X <- scratch A C G T X
            Mv[q] = 0;
            if (exact_init) {
              // true first column: D[i][0] = max(0, i - pbf)  ->  vertical delta +1 for rows i > pbf
              const int r0 = b << 6;                       // row i = r0 + bit + 1
              const int z = pbf - r0;                      // bits [0, z) are 0
              Pv[q] = z <= 0 ? ~0ull : (z >= 64 ? 0ull : (~0ull << z));
            } else Pv[q] = ~0ull;
          }
          if (exact_init) { const int rows = SB * (B + 1); score = rows > pbf ? rows - pbf : 0; }
          else score = (up_score - up_hout) + SB;
          inited = true;
        }
        const uint32_t row = (uint32_t)(c8 >> (8 * ph)) & 0xffu;               // sym * BPL of this column's text byte
        int hin = t <= t_hin_stop ? up_hout : 1;
#pragma unroll
        for (int q = 0; q < BPL; ++q) {
          u64 Eq = EQ[(row + q) * NT];
          const u64 pv = Pv[q], mv = Mv[q];
          const u64 hneg = hin < 0 ? 1ull : 0ull;
          const u64 Xv = Eq | mv;
          Eq |= hneg;
          const u64 Xh = (((Eq & pv) + pv) ^ pv) | Eq;
          u64 Ph = mv | ~(Xh | pv);
          u64 Mh = pv & Xh;
          const int ho = (int)(Ph >> 63) - (int)(Mh >> 63);
          Ph = (Ph << 1) | (hin > 0 ? 1ull : 0ull);
          Mh = (Mh << 1) | hneg;
          Pv[q] = Mh | ~(Xv | Ph);
          Mv[q] = Ph & Xv;
          hin = ho;
        }
        hout = hin;
        score += hout;
        if (t == t_last) {
          // last column: collect D[i][n] for the rows of this superblock that may end the alignment
          const int row_top = SB * B;                     // rows row_top+1 .. row_top+SB
          if (row_top + SB >= i_lo && row_top < m) {
            int sc = score;
#pragma unroll
            for (int q = BPL - 1; q >= 0; --q) {
              const u64 pv = Pv[q], mv = Mv[q];
#pragma unroll 1
              for (int r = 63; r >= 0; --r) {
                const int i = row_top + 64 * q + r + 1;
                if (i <= m && i >= i_lo && i >= 1 && sc < best) best = sc;
                sc -= (int)((pv >> r) & 1ull) - (int)((mv >> r) & 1ull);
              }
            }
          }
        }
      }
    };
    for (int t8 = 0; t8 <= t_end_w; t8 += 8) {
      step(t8 + 0, 0); step(t8 + 1, 1); step(t8 + 2, 2); step(t8 + 3, 3);
      step(t8 + 4, 4); step(t8 + 5, 5); step(t8 + 6, 6); step(t8 + 7, 7);
    }
    for (int off = GL / 2; off > 0; off >>= 1) { const int o = __shfl_xor(best, off, GL); best = o < best ? o : best; }
    const bool ok = !unsupported && best <= K;
    u64 w = 0;
    // a mirrored task (_pad bit 0: both sequences reversed by the pipeline) reports the cells of the un-reversed alignment: its free
    // prefixes are this task's free suffixes
    const bool mirrored = (tsk._pad & 1) != 0;
    if (cells) w = wfa_cells<GL>(ok ? best : -1, (int)tsk.pattern_len, (int)tsk.text_len, ef ? (mirrored ? tsk.pattern_end_free : tsk.pattern_begin_free) : 0,
                                 ef ? (mirrored ? tsk.text_end_free : tsk.text_begin_free) : 0, gl);
    if (has_task && gl == 0) {
      if (ok) { scores[ti] = best; if (cells) cells[ti] = w; }
      else if (overflow_list) { const uint32_t q = atomicAdd(n_overflow, 1u); overflow_list[q] = ti; }
      else scores[ti] = -1;
    }
  }
}

template <int BPL, int GL>
int launch_one(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
               const uint32_t* d_n_todo, uint32_t n_tasks, int32_t* d_scores, uint64_t* d_cells,
               uint32_t* ticket, uint32_t* n_overflow, uint32_t* overflow_list)
{
  constexpr int WPB = 4, G = 64 / GL;
  int maxblk = (int)((ctx->max_seq_len + 63) / 64) + 1;
  if (maxblk > MAXBLK) maxblk = MAXBLK;
  uint32_t want = (n_tasks + WPB * G - 1) / (WPB * G);
  // persistent blocks: exactly as many as are resident at once (VGPR / LDS limits differ per instantiation)
  static int per_cu = 0;
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, myers_edit_kernel<BPL, GL, WPB>, WPB * 64, 0) != hipSuccess || nb < 1) nb = 4;
    per_cu = nb > 8 ? 8 : nb;
  }
  const uint32_t grid_max = (uint32_t)ctx->n_cu * 8;      // sizes the scratch (upper bound of per_cu)
  const uint32_t grid_res = (uint32_t)ctx->n_cu * (uint32_t)per_cu;
  uint32_t grid = grid_res < want ? grid_res : want;
  if (grid == 0) return OTG_OK;
  u64* ws = (u64*)otg_slot(ctx, SLOT_AUX8, (size_t)grid_max * WPB * 8 * (size_t)MAXBLK * 5 * sizeof(u64));   // up to 8 pairs per wave
  if (!ws) return OTG_ERR_HIP;
  hipLaunchKernelGGL((myers_edit_kernel<BPL, GL, WPB>), dim3(grid), dim3(WPB * 64), 0, ctx->stream, d_arena, d_tasks, d_todo, d_n_todo, n_tasks,
                     d_scores, d_cells, ticket, n_overflow, overflow_list, ws, maxblk);
  return OTG_OK;
}

} // namespace

// One tier of the bit-parallel engine: tasks from (d_todo, d_n_todo) (or all n_tasks when d_todo is null);
// tasks it cannot finish exactly are appended to overflow_list / n_overflow.
// tier (OTG_MYERS_TIERS of them): 0 = 8-lane groups x 1 block per lane (8 pairs / wave); then 8 / 16 / 32 / 64-lane groups x 2 blocks per lane
//       (two blocks share the per-column overhead of a lane: ~19 % fewer instructions per row than one block per lane), with a three-block step
//       between the 8- and 16-lane and between the 16- and 32-lane ones (<3,8> = 1352 rows at 0.7 x the cost of <2,16>, <3,16> = 2896 rows at
//       0.7 x the cost of <2,32>: a pair pays for the band of its tier, not for its own); last = whole wave x 4 blocks per lane.
int otg_launch_myers(otg_ctx* ctx, int tier, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                     const uint32_t* d_n_todo, uint32_t n_tasks, int32_t* d_scores, uint64_t* d_cells,
                     uint32_t* ticket, uint32_t* n_overflow, uint32_t* overflow_list)
{
  int rc;
  switch (tier) {
    case 0: rc = launch_one<1, 8>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 1: rc = launch_one<2, 8>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 2: rc = launch_one<3, 8>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 3: rc = launch_one<2, 16>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 4: rc = launch_one<3, 16>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 5: rc = launch_one<2, 32>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    case 6: rc = launch_one<2, 64>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
    default: rc = launch_one<4, 64>(ctx, d_arena, d_tasks, d_todo, d_n_todo, n_tasks, d_scores, d_cells, ticket, n_overflow, overflow_list); break;
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipGetLastError());
  return OTG_OK;
}
