// otg_common.hpp — shared host-side plumbing of libotter_gpu.so (HIP runtime only, no torch).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/otter_gpu.h"

#define OTG_NULL_OFF (-(1 << 30))

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct otg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  std::string err;
  int exp_variant = 1;
  int n_cu = 256;
  uint32_t max_seq_len = 65536;   // longest sequence of the current batch (sizes the tier-3 edit workspace)
  // grow-only device scratch, keyed by purpose
  std::vector<DevBuf> pool;
  // resident batch of the L3 pipeline
  struct Pipeline* pipe = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // side streams of the gap-affine chain: on small batches the register tiers run next to each other (wfa_affine.hip); created on first use
  hipStream_t tier_stream[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // Which of the two optional bit-parallel edit tiers run (wfa_edit.hip): decided per KIND of pass (0 = distance matrix / operator-level call,
  // 1 = reassignment; set around the call) from the tier loads the previous pass of that kind left behind (a job's batches are alike), copied to
  // pinned host memory behind the chain — no host synchronisation; without history, from the number of task slots.
  int edit_pass_kind = 0;
  uint32_t* edit_hist = nullptr;                  // pinned: [kind][16] = tier input counts of the last pass
  hipEvent_t edit_hist_ev[2] = {nullptr, nullptr};
  uint32_t edit_hist_mask[2] = {0u, 0u};          // the mask that pass ran with (0: no history)
  double last_kernel_ms = 0.0;                    // HIP-event time of the kernels of the latest operator-level call that reports one (otg_last_kernel_ms)
  unsigned long long* affine_visited = nullptr;   // device counter: (score, diagonal) cells the exact gap-affine tiers visited (wfa_affine.hip)
  // aligner heuristic of the L1 calls and of the running pipeline (otg_set_heuristic / otg_params.heuristic; wfa_adaptive.hip)
  int heur_strategy = OTG_HEURISTIC_NONE, heur_min_wf_len = 10, heur_max_dist = 50, heur_steps = 1;
};

extern thread_local std::string g_otg_err;

int otg_fail(otg_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess)                                                                      \
      return otg_fail(ctx, OTG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                                      \
  } while (0)

// Grow-only device allocation slot; returns nullptr on failure (error recorded).
void* otg_slot(otg_ctx* ctx, int slot, size_t bytes);

// words of SLOT_COUNTERS every aligner chain asks for (one size: the slot is never re-allocated between two chains of a batch)
constexpr size_t OTG_COUNTER_WORDS = 160;
enum {
  SLOT_ARENA = 0, SLOT_TASKS, SLOT_SCORES, SLOT_CELLS, SLOT_COUNTERS, SLOT_WF_WS, SLOT_CIG_OFF, SLOT_CIG_LEN,
  SLOT_CIG_ARENA, SLOT_BT_POOL, SLOT_ROWTAB, SLOT_REVOPS, SLOT_TASKSTATE, SLOT_TODO, SLOT_AUX0, SLOT_AUX1,
  SLOT_AUX2, SLOT_AUX3, SLOT_AUX4, SLOT_AUX5, SLOT_AUX6, SLOT_AUX7, SLOT_AUX8, SLOT_AUX9,
  SLOT_P0, SLOT_P1, SLOT_P2, SLOT_P3, SLOT_P4, SLOT_P5, SLOT_P6, SLOT_P7, SLOT_P8, SLOT_P9,
  SLOT_P10, SLOT_P11, SLOT_P12, SLOT_P13, SLOT_P14, SLOT_P15, SLOT_P16, SLOT_P17, SLOT_P18, SLOT_P19,
  SLOT_P20, SLOT_P21, SLOT_P22, SLOT_P23, SLOT_P24, SLOT_P25, SLOT_P26, SLOT_P27, SLOT_P28, SLOT_P29,
  SLOT_COUNT
};

void otg_pipeline_free(otg_ctx* ctx);

// Contexts that share a device (the dispatcher runs 2-4 per GPU) size their multi-gigabyte workspaces from hipMemGetInfo; two of them doing so
// at the same moment would both claim the same free bytes.  Every "measure free memory, then allocate" section holds this lock.
std::mutex& otg_device_mutex(int device);

#if defined(__HIPCC__)
// One atomic add per WAVE, issued with exec forced to lane 0 and no divergent branch in the HIP source.
// (A source-level `if (lane == 0) atomicAdd(..)` + readfirstlane at the head of a persistent-wave loop was
// structurized by hipcc 7.2 so that lanes 1..63 re-entered the loop with lane 0 masked: an endless loop.)
// Call only from wave-uniform control flow.
__device__ __forceinline__ uint32_t otg_wave_atomic_add(uint32_t* ctr, uint32_t val)
{
  uint32_t r;
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_atomic_add %0, %2, %3, off sc0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %1"
      : "=&v"(r), "=&s"(saved)
      : "v"(ctr), "v"(val)
      : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}

__device__ __forceinline__ int otg_imax(int a, int b) { return a > b ? a : b; }
// Maximum over the 64 lanes (DPP row shifts + row broadcasts), returned wave-uniform.
__device__ __forceinline__ int otg_wave_max_i32(int v)
{
  constexpr int NEG = -2147483647 - 1;
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of each row = row max
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
  v = otg_imax(v, __builtin_amdgcn_update_dpp(NEG, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

// Band of the bit-parallel edit tiers (myers_edit.hip) for a cost threshold K.  Rows i (pattern), columns j (text),
// d = m - n >= 0; the alignment starts on a diagonal e0 in [0, pbf] and ends on e1 in [d - pef, d].  A path of cost
// <= K that visits diagonal e pays at least |e - e0| + |e - e1| indels and needs |e0 - e1| <= K, hence
//   e <= KU = min((K + d + pbf) / 2, K + pbf)      and      e >= -KL,
//   KL = min((K - d + pef) / 2, K) while the alignment may start on diagonal 0 (K >= d - pef), and beyond that — a start on diagonal 0 would
//   cost more than K, so e0 >= d - pef - K and the lowest diagonal any path can visit is that one — KL = K - (d - pef) < 0: the band's
//   lower edge lies ABOVE diagonal 0.  (A read that covers only the end of its pattern, d = pbf = 1500, K = 300: 450 diagonals instead of
//   1650 — the reassignment pass of a batch with clipped reads ran on tiers four times as wide as its alignments needed.)
__device__ __forceinline__ void otg_myers_band(int K, int d, int pbf, int pef, int* KL, int* KU)
{
  const int u1 = (K + d + pbf + 1) / 2, u2 = K + pbf;
  *KU = u1 < u2 ? u1 : u2;
  int l = (K - d + pef + 1) / 2;
  if (l > K) l = K;
  if (K - d + pef < 0) l = K - d + pef;
  *KL = l;
}
// Largest threshold K whose band fits R rows of lane schedule (KL + KU <= R, monotone in K).
// W_p of SURVEY.md §8d in closed form: wavefront cells an edit-distance WFA evaluates up to score s when the wavefront of score t spans
// the diagonals [max(-pbf - t, -m), min(tbf + t, n)] (pbf / tbf = free pattern / text prefix, 0 for a global alignment).
__device__ __forceinline__ unsigned long long otg_edit_wfa_cells(int s, int m, int n, int pbf, int tbf)
{
  if (s < 0) return 0ull;
  if (pbf > m) pbf = m;
  if (tbf > n) tbf = n;
  const long long kh = s < n - tbf ? s : n - tbf, kl = s < m - pbf ? s : m - pbf;
  const long long sum_hi = (kh + 1) * tbf + kh * (kh + 1) / 2 + ((long long)s - kh) * n;
  const long long sum_lo = (kl + 1) * pbf + kl * (kl + 1) / 2 + ((long long)s - kl) * m;
  return (unsigned long long)(sum_hi + sum_lo + (long long)s + 1);
}

__device__ __forceinline__ int otg_myers_threshold(int R, int d, int pbf, int pef)
{
  int lo = 0, hi = R;            // invariant: band(lo) fits (K = 0: KL + KU <= pbf + small), band(hi + 1) unknown
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    int kl, ku;
    otg_myers_band(mid, d, pbf, pef, &kl, &ku);
    if (kl + ku <= R) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---- match-run extension helpers shared by the wavefront kernels -------------------------------------
__device__ __forceinline__ uint64_t otg_load8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// Lane-private: number of equal leading bytes of P+v.. and T+h.., looking at most 64 bytes ahead and at most `rem`.
// Eight independent 8-byte loads per operand are issued back to back (one latency, not eight).
__device__ __forceinline__ int otg_match64(const uint8_t* P, const uint8_t* T, int v, int h, int rem)
{
  uint64_t x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (8 * i < rem) ? (otg_load8(P + v + 8 * i) ^ otg_load8(T + h + 8 * i)) : ~0ull;
  int m = 64;
#pragma unroll
  for (int i = 7; i >= 0; --i) if (x[i]) m = 8 * i + (__builtin_ctzll(x[i]) >> 3);
  return m < rem ? m : rem;
}

// Wave-cooperative (all arguments wave-uniform): extends ONE diagonal, 64 lanes x 8 bytes = 512 bytes per iteration.
__device__ __forceinline__ int otg_wave_match(const uint8_t* P, const uint8_t* T, int v, int h, int rem, int lane)
{
  int total = 0;
  while (total < rem) {
    const int off = total + lane * 8;
    const bool in = off < rem;
    uint64_t x = ~0ull;
    if (in) x = otg_load8(P + v + off) ^ otg_load8(T + h + off);
    const int m = x ? (int)(__builtin_ctzll(x) >> 3) : 8;
    const unsigned long long stop = __ballot(m < 8);          // lanes beyond `rem` report a mismatch at byte 0
    if (stop) {
      const int first = (int)__builtin_ctzll(stop);
      total += first * 8 + __builtin_amdgcn_readlane(m, first);
      break;
    }
    total += 512;
  }
  return total < rem ? total : rem;
}
#endif

// ---- device-resident launch helpers implemented in the .hip files -------------------------------------
// All take device pointers; they enqueue on ctx->stream and do not synchronise unless stated.

// wfa_edit.hip
int otg_launch_edit(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, uint32_t n_tasks,
                    int32_t* d_scores, uint64_t* d_cells, float* kernel_ms, uint64_t* launches);

// bit-parallel edit tiers (myers_edit.hip), narrowest first: <blocks per lane, lanes per pair> = <1,8> <2,8> <3,8> <2,16> <3,16> <2,32> <2,64> <4,64>
constexpr int OTG_MYERS_TIERS = 8;
int otg_launch_myers(otg_ctx* ctx, int tier, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                     const uint32_t* d_n_todo, uint32_t n_tasks, int32_t* d_scores, uint64_t* d_cells,
                     uint32_t* ticket, uint32_t* n_overflow, uint32_t* overflow_list);
int otg_launch_edit_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                         const uint32_t* d_n_todo, uint32_t n_task_slots, int32_t* d_scores, uint64_t* d_cells,
                         float* kernel_ms, uint64_t* launches);
int otg_launch_affine_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                           const uint32_t* d_n_todo, uint32_t n_task_slots, int x, int o, int e, int32_t* d_scores,
                           const uint64_t* d_cig_off, uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells,
                           float* kernel_ms = nullptr, uint64_t* launches = nullptr);

// wfa_affine.hip — forward + backtrace + unpack; CIGARs land in d_cig_arena at d_cig_off (precomputed
// exclusive prefix of pattern_len+text_len per task), lengths in d_cig_len.  Synchronises internally
// (multi-round when the backtrace pool is smaller than the batch needs).
int otg_launch_affine(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, uint32_t n_tasks,
                      int x, int o, int e, int32_t* d_scores, const uint64_t* d_cig_off, uint32_t* d_cig_len,
                      uint8_t* d_cig_arena, uint64_t* d_cells);

// wfa_adaptive.hip — the same two chains under wfadaptive(min_wf_len, max_dist, steps); otg_launch_edit_todo / otg_launch_affine_todo hand
// over to them when ctx->heur_strategy says so
int otg_launch_edit_adaptive_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                                  const uint32_t* d_n_todo, uint32_t n_task_slots, int32_t* d_scores, uint64_t* d_cells,
                                  float* kernel_ms, uint64_t* launches);
int otg_launch_affine_adaptive_todo(otg_ctx* ctx, const uint8_t* d_arena, const otg_align_task* d_tasks, const uint32_t* d_todo,
                                    const uint32_t* d_n_todo, uint32_t n_task_slots, int x, int o, int e, int32_t* d_scores,
                                    const uint64_t* d_cig_off, uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells,
                                    float* kernel_ms, uint64_t* launches);

// cluster.hip
int otg_launch_cluster(otg_ctx* ctx, const otg_params* P, const double* d_dist, const uint64_t* d_dist_off,
                       const uint32_t* d_read_len, const uint64_t* d_len_off, const uint32_t* d_n_valid,
                       uint32_t n_regions, int32_t* d_labels, int32_t* d_ic, int32_t* d_fc, double* d_bounds,
                       int32_t* d_err);

// poa.hip
int otg_launch_poa(otg_ctx* ctx, const uint8_t* d_seq_arena, const uint8_t* d_cig_arena,
                   const otg_poa_member* d_members, uint32_t n_members, const otg_poa_graph* d_graphs,
                   const otg_poa_graph* h_graphs, uint32_t n_graphs, uint32_t* d_out_len,
                   std::vector<uint64_t>& node_off);

// cluster.hip (genotype_kernel)
int otg_launch_genotype(otg_ctx* ctx, const otg_params* P, const uint8_t* d_arena, const uint64_t* d_seq_off,
                        const uint32_t* d_seq_len, const uint32_t* d_first, const uint32_t* d_n, uint32_t n_regions,
                        const uint64_t* d_pair_off, uint64_t n_pairs_total, uint64_t n_alleles_total,
                        int32_t* d_gt, int32_t* d_gtl, int32_t* d_gtk, double* d_hsd, int32_t* d_ngt, int32_t* d_reps,
                        int32_t* d_err);
