// pipeline.hip — L3 region-batch pipeline: the five calls inside the region loop of assemble_process
// (reference: src/assemble.cpp:71-150) for a whole batch of regions, device-resident between stages.
//
//   [local_realignment]  src/analignments.cpp:11-60     K_realign_prepare -> affine WFA -> K_realign_apply
//   partition_valid_reads src/assemble.cpp:27-37,91-122 K_region_prepare
//   fill_dist_matrix     src/analignments.cpp:62-124    K_pair_tasks -> edit WFA -> K_dist_epilogue
//   otter_hclust         src/otterclust.cpp:118-320     cluster kernel -> K_scatter_labels
//   invalid_reassignment src/analignments.cpp:126-177   K_reassign_tasks -> edit WFA -> K_dist_epilogue -> K_reassign_apply
//   rapid_consensus      src/analignments.cpp:192-298   K_consensus_prepare -> affine WFA -> POA -> K_finalize/K_gather
//
// Task lists are generated on the device; slots are dense and host-sized from upper bounds known at
// submit time (pairs: n(n-1)/2 per region; reassignment: n^2 per region; op strings: one slot per read),
// the aligners consume compacted todo lists whose length stays on the device.  The host only reads back a
// handful of scalars (POA layout totals, output sizes).
#include "otg_common.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <cstdlib>

enum { AL_NONE = 0, AL_COPY = 1, AL_POA = 2 };

struct AlleleSlot {
  uint64_t src_off;      // AL_COPY: offset in arena
  uint32_t src_len;
  int32_t kind;
  int32_t scov, acov, tcov;
  float se;
  int32_t ps, hp;
};

struct Pipeline {
  otg_params P;
  uint32_t n_reads = 0, n_regions = 0;
  uint64_t arena_bytes = 0;
  uint64_t rev_base = 0;         // the second half of the device arena: every read reversed in place (same offset + rev_base)
  std::vector<otg_read> h_reads;
  std::vector<otg_region> h_regions;
  std::vector<uint64_t> h_dist_off, h_re_off, h_cig_off;
  uint64_t n_pair_slots = 0, n_re_slots = 0, cig_bytes = 0;
  std::vector<DevBuf> buf;
  otg_run_stats stats;
  uint32_t out_alleles = 0;
  uint64_t out_seq_bytes = 0;
  bool ran = false;
};

enum {
  B_ARENA = 0, B_READS, B_REGIONS, B_READ_REGION, B_DIST_OFF, B_RE_OFF, B_CIG_OFF, B_FIRST_READ64,
  B_STATUS, B_NVALID, B_IGNHAPS, B_VALID, B_VPOS, B_VLEN,
  B_TASKS, B_TODO, B_SCORES, B_DEN, B_DIST, B_CELLS, B_CNT,
  B_CLLAB, B_IC, B_FC, B_BOUNDS, B_CLERR, B_LABELS,
  B_REDIST, B_RTASKS, B_RKIND, B_CIGLEN, B_CIG,
  B_ALLELES, B_GRAPHS, B_MEMBERS, B_POALEN,
  B_ALLEN, B_ALOFF, B_ALIDX, B_OUTSEQ, B_OUTAL, B_REGRES, B_STATS, B_SCAN_TMP, B_TOTALS,
  B_COUNT
};

static void* pbuf(otg_ctx* ctx, Pipeline* pl, int i, size_t bytes)
{
  if (bytes == 0) bytes = 16;
  DevBuf& b = pl->buf[i];
  if (b.cap >= bytes) return b.p;
  if (b.p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + (bytes >> 4) + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { otg_fail(ctx, OTG_ERR_HIP, "hipMalloc(%zu) for pipeline buffer %d failed: %s", want, i, hipGetErrorString(e)); b.p = nullptr; return nullptr; }
  b.cap = want;
  return b.p;
}

void otg_pipeline_free(otg_ctx* ctx)
{
  if (!ctx || !ctx->pipe) return;
  for (auto& b : ctx->pipe->buf) if (b.p) (void)hipFree(b.p);
  delete ctx->pipe;
  ctx->pipe = nullptr;
}

namespace {

__device__ __forceinline__ size_t didx(int N, int r, int c) { return (size_t)((((long long)(2 * N - 3 - r)) * r) >> 1) + c - 1; }

__device__ bool seq_equal(const uint8_t* a, const uint8_t* b, uint32_t n)
{
  for (uint32_t i = 0; i < n; ++i) if (a[i] != b[i]) return false;
  return true;
}

__device__ __forceinline__ void make_task(otg_align_task& t, uint64_t po, uint32_t pl, uint64_t to, uint32_t tl,
                                          int ef, int pbf, int pef, int tbf, int tef)
{
  t.pattern_off = po; t.text_off = to; t.pattern_len = pl; t.text_len = tl;
  t.pattern_begin_free = pbf; t.pattern_end_free = pef; t.text_begin_free = tbf; t.text_end_free = tef;
  t.endsfree = ef; t._pad = 0;
}

// align_anreads (src/analignments.cpp:62-101) as a task; returns 0: distance already known (*known),
// 1: task written (denominator in *den).
__device__ int anreads_task(const uint8_t* arena, const otg_read& x, const otg_read& y, otg_align_task& t, uint32_t* den, double* known, uint64_t rev_base = 0)
{
  const bool xs = x.spanning_l && x.spanning_r, ys = y.spanning_l && y.spanning_r;
  if (x.seq_len == y.seq_len && seq_equal(arena + x.seq_off, arena + y.seq_off, x.seq_len)) { *known = 0.0; return 0; }
  if ((xs && ys) || (ys && x.seq_len >= y.seq_len)) {
    const bool x_is_smallest = x.seq_len < y.seq_len;
    if (x_is_smallest) { make_task(t, y.seq_off, y.seq_len, x.seq_off, x.seq_len, 0, 0, 0, 0, 0); *den = y.seq_len; }
    else { make_task(t, x.seq_off, x.seq_len, y.seq_off, y.seq_len, 0, 0, 0, 0, 0); *den = x.seq_len; }
    return 1;
  }
  if (ys) {
    const int length_diff = (int)y.seq_len - (int)x.seq_len;   // > 0 here (:93-98)
    int pbf, pef;
    if (x.spanning_l) { pbf = 0; pef = length_diff; }
    else if (x.spanning_r) { pbf = length_diff; pef = 0; }
    else { pbf = length_diff / 2; pef = length_diff / 2; }
    if (x.spanning_r && !x.spanning_l && rev_base) {
      // a free BEGIN lets the alignment start on any of length_diff + 1 diagonals (wide band); the same distance read from the
      // other end has a free END and a band of ~1.5 x the distance: align the reversed copies instead
      make_task(t, rev_base + y.seq_off, y.seq_len, rev_base + x.seq_off, x.seq_len, 1, 0, length_diff, 0, 0);
      t._pad = 1;             // mirrored: the wavefront-cell statistics are those of the un-reversed alignment
    } else
    make_task(t, y.seq_off, y.seq_len, x.seq_off, x.seq_len, 1, pbf, pef, 0, 0);
    *den = x.seq_len;
    return 1;
  }
  *known = -1.0;
  return 0;
}

// ---- local_realignment, part 1 (src/analignments.cpp:15-32)
__global__ void K_realign_prepare(const otg_read* __restrict__ reads, const otg_region* __restrict__ regions,
                                  const uint32_t* __restrict__ read_region, uint32_t n_reads, int flank,
                                  otg_align_task* __restrict__ tasks, uint8_t* __restrict__ kind,
                                  uint32_t* __restrict__ todo, uint32_t* __restrict__ n_todo)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_reads) return;
  if (read_region[i] == 0xffffffffu) { kind[i] = 0; return; }
  const otg_read rd = reads[i];
  const otg_region rg = regions[read_region[i]];
  uint8_t k = 0;
  const bool spanning = rd.spanning_l && rd.spanning_r;
  if (!spanning && (rd.spanning_l || rd.spanning_r) && rg.flank_l_len && rg.flank_r_len) {
    const bool left = rd.spanning_r && rd.ccoord_first >= flank;
    const bool right = rd.spanning_l && (int)rd.seq_len - rd.ccoord_second >= flank;
    if (left && rd.ccoord_first > 0) {
      make_task(tasks[i], rd.seq_off, (uint32_t)rd.ccoord_first, rg.flank_l_off, rg.flank_l_len, 0, 0, 0, 0, 0);
      k = 1;
    } else if (!left && right && (int)rd.seq_len - rd.ccoord_second > 0) {
      make_task(tasks[i], rd.seq_off + (uint32_t)rd.ccoord_second, rd.seq_len - (uint32_t)rd.ccoord_second, rg.flank_r_off, rg.flank_r_len, 0, 0, 0, 0, 0);
      k = 2;
    }
  }
  kind[i] = k;
  if (k) todo[atomicAdd(n_todo, 1u)] = i;
}

// ---- local_realignment, part 2 (:34-57): running +1/-1 score over the pattern-consuming ops
__global__ void K_realign_apply(otg_read* __restrict__ reads, uint32_t n_reads, const uint8_t* __restrict__ kind,
                                const uint8_t* __restrict__ cig, const uint64_t* __restrict__ cig_off,
                                const uint32_t* __restrict__ cig_len, int flank, double min_sim)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_reads || !kind[i]) return;
  const uint8_t* c = cig + cig_off[i];
  const uint32_t n = cig_len[i];
  int prev = 0, j = 0, best = 0, best_i = 0, last_nonpos = 0, start_i = 0;
  bool have0 = false;
  for (uint32_t q = 0; q < n; ++q) {
    const uint8_t op = c[q];
    if (op == 'I') continue;
    int sc = 0;
    if (op == 'M') sc = (j == 0) ? 1 : prev + 1;
    else if (j > 0 && prev > 0) sc = prev - 1;
    if (!have0) { best = sc; best_i = 0; have0 = true; last_nonpos = 0; start_i = 0; }
    if (sc <= 0) last_nonpos = j;
    if (sc > best) { best = sc; best_i = j; start_i = last_nonpos; }
    prev = sc; ++j;
  }
  if (!have0) return;
  // start_i: walk back from max_sum_i while scores > 0 (stops at index 0 regardless of its score)
  if (((double)best / (double)flank) >= min_sim) {
    otg_read rd = reads[i];
    if (kind[i] == 1) { rd.seq_off += (uint32_t)best_i; rd.seq_len -= (uint32_t)best_i; }
    else { rd.seq_len = (uint32_t)(rd.ccoord_second + start_i); }
    rd.spanning_l = 1; rd.spanning_r = 1;
    reads[i] = rd;
  }
}

// ---- partition_valid_reads + region status (src/assemble.cpp:27-37,71,91-122)
__global__ void K_region_prepare(const otg_read* __restrict__ reads, const otg_region* __restrict__ regions, uint32_t n_regions,
                                 int max_cov, int ignore_haps, int32_t* __restrict__ status, uint32_t* __restrict__ n_valid,
                                 int32_t* __restrict__ ign, uint32_t* __restrict__ valid, int32_t* __restrict__ vpos,
                                 uint32_t* __restrict__ vlen)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_regions) return;
  const otg_region rg = regions[r];
  const uint32_t f = rg.first_read, n = rg.n_reads;
  int st = OTG_REGION_OK;
  uint32_t nv = 0;
  int lig = ignore_haps;
  for (uint32_t i = 0; i < n; ++i) vpos[f + i] = -1;
  if (n == 0) st = OTG_REGION_EMPTY;
  else if ((int)n > max_cov) st = OTG_REGION_SKIP_MAXCOV;
  else {
    uint32_t span = 0;
    for (uint32_t i = 0; i < n; ++i) span += (reads[f + i].spanning_l && reads[f + i].spanning_r);
    if (span == 0) st = OTG_REGION_NO_SPANNING;
    else {
      for (int pass = 0; pass < 2; ++pass) {
        nv = 0;
        for (uint32_t i = 0; i < n; ++i) {
          const otg_read& rd = reads[f + i];
          const bool sp = rd.spanning_l && rd.spanning_r;
          const bool ok = sp && (lig || (rd.ps >= 0 && rd.hp >= 0));
          if (ok) valid[f + nv++] = f + i;
        }
        if (nv < 2 && !lig) lig = 1; else break;
      }
      if (nv == 0) st = OTG_REGION_NO_SPANNING;
      else for (uint32_t v = 0; v < nv; ++v) { vpos[valid[f + v]] = (int32_t)v; vlen[f + v] = reads[valid[f + v]].seq_len; }
    }
  }
  status[r] = st; n_valid[r] = (st == OTG_REGION_OK) ? nv : 0; ign[r] = lig;
}

// ---- fill_dist_matrix (src/analignments.cpp:103-124): one block per region
__global__ void K_pair_tasks(const uint8_t* __restrict__ arena, const otg_read* __restrict__ reads,
                             const otg_region* __restrict__ regions, uint32_t n_regions, const uint32_t* __restrict__ n_valid,
                             const int32_t* __restrict__ ign, const uint32_t* __restrict__ valid,
                             const uint64_t* __restrict__ dist_off, int max_alleles,
                             otg_align_task* __restrict__ tasks, uint32_t* __restrict__ den, double* __restrict__ dist,
                             uint32_t* __restrict__ todo, uint32_t* __restrict__ n_todo)
{
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
    const int n = (int)n_valid[r];
    if (n < 2) continue;
    const uint32_t f = regions[r].first_read;
    const uint64_t base = dist_off[r];
    const size_t np = (size_t)n * (n - 1) / 2;
    if (max_alleles == 1) { for (size_t q = threadIdx.x; q < np; q += blockDim.x) dist[base + q] = 1.0; continue; }   // DistMatrix default (src/andistmat.cpp:10)
    for (int i = 0; i < n - 1; ++i) {
      const otg_read x = reads[valid[f + i]];
      for (int j = i + 1 + (int)threadIdx.x; j < n; j += (int)blockDim.x) {
        const otg_read y = reads[valid[f + j]];
        const uint64_t slot = base + didx(n, i, j);
        if (!ign[r]) {
          const bool bx = x.ps >= 0 && x.hp >= 0, by = y.ps >= 0 && y.hp >= 0;
          dist[slot] = (bx && by && x.ps == y.ps && x.hp == y.hp) ? 0.0 : 1.0;      // get_dist_anreads :108-113
        } else {
          double known = 0;
          uint32_t d = 1;
          if (anreads_task(arena, x, y, tasks[slot], &d, &known)) { den[slot] = d; todo[atomicAdd(n_todo, 1u)] = (uint32_t)slot; }
          else dist[slot] = known;
        }
      }
    }
  }
}

__global__ void K_dist_epilogue(const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo,
                                const int32_t* __restrict__ scores, const uint32_t* __restrict__ den, double* __restrict__ dist,
                                uint32_t* __restrict__ fail_flag)
{
  const uint32_t n = *n_todo;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t slot = todo[t];
    if (scores[slot] < 0) *fail_flag = 1u;
    dist[slot] = scores[slot] / (double)den[slot];                                     // src/analignments.cpp:72,97
  }
}

// per-stage workload statistics: Σ cells, Σ sequence bytes over the todo list -> acc[0], acc[1]; acc[2] += n
__global__ void K_stats(const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo,
                        const otg_align_task* __restrict__ tasks, const uint64_t* __restrict__ cells, unsigned long long* acc,
                        const int32_t* __restrict__ scores, uint32_t* __restrict__ fail_flag)
{
  __shared__ unsigned long long sc[256], sb[256];
  const uint32_t n = *n_todo;
  unsigned long long c = 0, b = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t slot = todo[t];
    if (scores[slot] < 0) { *fail_flag = 1u; continue; }
    c += cells[slot];
    b += (unsigned long long)tasks[slot].pattern_len + tasks[slot].text_len;
  }
  sc[threadIdx.x] = c; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { sc[threadIdx.x] += sc[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; } __syncthreads(); }
  if (threadIdx.x == 0) { atomicAdd(&acc[0], sc[0]); atomicAdd(&acc[1], sb[0]); if (blockIdx.x == 0) atomicAdd(&acc[2], (unsigned long long)n); }
}

// A gap-affine alignment the device could not hold (score < 0: the last-resort tier ran out of provenance storage) takes its REGION out of
// the results — status OTG_REGION_ALIGN_CAPACITY, no records — and nothing else: task slot = read index, so the region follows from the read.
__global__ void K_mark_failed_affine(const uint32_t* __restrict__ todo, const uint32_t* __restrict__ n_todo, const int32_t* __restrict__ scores,
                                     const uint32_t* __restrict__ read_region, int32_t* __restrict__ status)
{
  const uint32_t n = *n_todo;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t slot = todo[t];
    if (scores[slot] < 0) { const uint32_t r = read_region[slot]; if (r != 0xffffffffu) status[r] = OTG_REGION_ALIGN_CAPACITY; }
  }
}

__global__ void K_scatter_labels(const uint32_t* __restrict__ read_region, const otg_region* __restrict__ regions,
                                 const int32_t* __restrict__ vpos, const int32_t* __restrict__ cl_labels, uint32_t n_reads,
                                 int32_t* __restrict__ labels)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_reads) return;
  if (read_region[i] == 0xffffffffu) { labels[i] = -1; return; }
  const int v = vpos[i];
  labels[i] = v >= 0 ? cl_labels[regions[read_region[i]].first_read + v] : -1;        // src/assemble.cpp:130-133
}

// ---- invalid_reassignment, task generation (src/analignments.cpp:129-157): slot(i,j) = re_off[r] + i*n + j
// reversed copy (at + rev_base) of every read of a region that has reads to reassign: a non-spanning read that spans only the
// right side is compared from the other end (see anreads_task)
__global__ void K_reverse_reads(uint8_t* __restrict__ arena, const otg_read* __restrict__ reads, const otg_region* __restrict__ regions,
                                const uint32_t* __restrict__ read_region, uint32_t n_reads, const int32_t* __restrict__ status,
                                const uint32_t* __restrict__ n_valid, uint64_t rev_base)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t nw = gridDim.x * (blockDim.x >> 6);
  for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n_reads; i += nw) {
    const uint32_t r = read_region[i];
    if (r == 0xffffffffu || status[r] != OTG_REGION_OK || n_valid[r] >= regions[r].n_reads) continue;
    const otg_read rd = reads[i];
    const uint8_t* src = arena + rd.seq_off;
    uint8_t* dst = arena + rev_base + rd.seq_off;
    for (uint32_t j = lane; j < rd.seq_len; j += 64) dst[j] = src[rd.seq_len - 1 - j];
  }
}

__global__ void K_reassign_tasks(const uint8_t* __restrict__ arena, const otg_read* __restrict__ reads,
                                 const otg_region* __restrict__ regions, uint32_t n_regions, const int32_t* __restrict__ status,
                                 const uint32_t* __restrict__ n_valid, const int32_t* __restrict__ fc, const int32_t* __restrict__ labels,
                                 const uint64_t* __restrict__ re_off, otg_align_task* __restrict__ tasks,
                                 uint32_t* __restrict__ den, double* __restrict__ dist,
                                 uint32_t* __restrict__ todo, uint32_t* __restrict__ n_todo, uint64_t rev_base)
{
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
    if (status[r] != OTG_REGION_OK || fc[r] <= 0) continue;                            // fc == 0: `-a 0` defect, reference UB
    const otg_region rg = regions[r];
    const int n = (int)rg.n_reads;
    if ((int)n_valid[r] >= n) continue;                                               // no invalid reads (src/assemble.cpp:135)
    const uint32_t f = rg.first_read;
    for (int i = 0; i < n; ++i) {
      if (labels[f + i] >= 0) continue;
      const otg_read x = reads[f + i];
      for (int j = (int)threadIdx.x; j < n; j += (int)blockDim.x) {
        if (j == i) continue;
        const otg_read y = reads[f + j];
        if (!(y.spanning_l && y.spanning_r)) continue;
        // targets are reads that are labelled now OR may become labelled earlier in this pass (unlabelled spanning reads, --haps)
        if (labels[f + j] < 0 && j > i) continue;
        const uint64_t slot = re_off[r] + (uint64_t)i * n + j;
        double known = 0;
        uint32_t d = 1;
        if (anreads_task(arena, x, y, tasks[slot], &d, &known, rev_base)) { den[slot] = d; todo[atomicAdd(n_todo, 1u)] = (uint32_t)slot; }
        else dist[slot] = known;
      }
    }
  }
}

// ---- invalid_reassignment, decision (:134-175): sequential in read order, one thread per region
__global__ void K_reassign_apply(const otg_read* __restrict__ reads, const otg_region* __restrict__ regions, uint32_t n_regions,
                                 const int32_t* __restrict__ status, const uint32_t* __restrict__ n_valid,
                                 const int32_t* __restrict__ fc, const uint64_t* __restrict__ re_off,
                                 const double* __restrict__ dist, double min_sim, double max_error,
                                 int32_t* __restrict__ labels, int32_t* __restrict__ status_out)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_regions || status[r] != OTG_REGION_OK) return;
  const otg_region rg = regions[r];
  const int n = (int)rg.n_reads;
  if ((int)n_valid[r] >= n) return;
  const uint32_t f = rg.first_read;
  const int total = fc[r];
  if (total <= 0) return;
  for (int i = 0; i < n; ++i) {
    if (labels[f + i] >= 0) continue;
    // max similarity per allele, first pass finds the best label, later passes the rest (no per-thread arrays)
    int max_label = 0; double max_val = 0.0; int same = 0;
    for (int l = 0; l < total; ++l) {
      double ms = 0.0;
      for (int j = 0; j < n; ++j) {
        if (j == i || labels[f + j] != l) continue;
        const otg_read& y = reads[f + j];
        if (!(y.spanning_l && y.spanning_r)) continue;
        const double d = dist[re_off[r] + (uint64_t)i * n + j];
        if (d < 0) { status_out[r] = OTG_ERR_FATAL; return; }
        const double sim = 1 - d;
        if (sim > ms) ms = sim;
      }
      if (l == 0 || ms > max_val) { max_val = ms; max_label = l; }
    }
    double min_diff = 1.0;
    for (int l = 0; l < total; ++l) {
      double ms = 0.0;
      for (int j = 0; j < n; ++j) {
        if (j == i || labels[f + j] != l) continue;
        const otg_read& y = reads[f + j];
        if (!(y.spanning_l && y.spanning_r)) continue;
        const double sim = 1 - dist[re_off[r] + (uint64_t)i * n + j];
        if (sim > ms) ms = sim;
      }
      if (ms == max_val) ++same;
      if (l != max_label) { const double diff = max_val - ms; if (diff < min_diff) min_diff = diff; }
    }
    if (same == 1 && max_val >= min_sim && min_diff >= max_error) labels[f + i] = max_label;
  }
}

// ---- rapid_consensus bookkeeping (src/analignments.cpp:198-283): one thread per region
__global__ void K_consensus_prepare(const otg_read* __restrict__ reads, const otg_region* __restrict__ regions, uint32_t n_regions,
                                    int32_t* __restrict__ status, const uint32_t* __restrict__ n_valid, const int32_t* __restrict__ ign,
                                    const uint32_t* __restrict__ valid, const int32_t* __restrict__ labels, const int32_t* __restrict__ fc,
                                    const uint64_t* __restrict__ dist_off, const double* __restrict__ dist,
                                    const uint64_t* __restrict__ cig_off,
                                    AlleleSlot* __restrict__ alleles, otg_poa_graph* __restrict__ graphs,
                                    otg_poa_member* __restrict__ members, otg_align_task* __restrict__ tasks,
                                    uint32_t* __restrict__ todo, uint32_t* __restrict__ n_todo, uint32_t* __restrict__ scratch)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_regions) return;
  const otg_region rg = regions[r];
  const uint32_t f = rg.first_read;
  const int n = (int)rg.n_reads;
  for (int i = 0; i < n; ++i) { alleles[f + i].kind = AL_NONE; graphs[f + i].backbone_len = 0; graphs[f + i].n_members = 0; graphs[f + i].first_member = f; graphs[f + i].backbone_off = 0; graphs[f + i].c = 1.0f; graphs[f + i].t = 0.3f; }
  if (status[r] != OTG_REGION_OK) return;
  const int nv = (int)n_valid[r];
  const int total = fc[r];
  const double* D = dist + dist_off[r];
  uint32_t* liv = scratch + f;            // valid indices (vi) of the current label
  uint32_t mpos = f;                      // next free member slot of this region
  for (int label = 0; label < total; ++label) {
    int nl = 0;
    for (int vi = 0; vi < nv; ++vi) if (labels[valid[f + vi]] == label) liv[nl++] = (uint32_t)vi;
    if (nl == 0) { status[r] = OTG_ERR_FATAL; return; }                               // :210-213 exit(1)
    // DistMatrix::get_medoid (src/andistmat.cpp:36-50)
    uint32_t rep_vi = liv[0];
    double min_sum = 100000000.0;
    for (int a = 0; a < nl; ++a) {
      double s = 0.0;
      for (int b = 0; b < nl; ++b) if (liv[a] != liv[b]) { const int i = (int)liv[a], j = (int)liv[b]; s += i < j ? D[didx(nv, i, j)] : D[didx(nv, j, i)]; }
      if (s < min_sum) { rep_vi = liv[a]; min_sum = s; }
    }
    const uint32_t rep = valid[f + rep_vi];
    int n_all = 0;
    for (int i = 0; i < n; ++i) if (f + i != rep && labels[f + i] == label) ++n_all;
    AlleleSlot A;
    A.tcov = n; A.acov = n_all + 1; A.scov = nl;
    if (nl == 1) A.se = 0;
    else if (nl == 2) { const int i = (int)liv[0], j = (int)liv[1]; A.se = (float)(i < j ? D[didx(nv, i, j)] : D[didx(nv, j, i)]); }
    else {                                                                            // compute_se :179-190
      double u = 0.0, q = 0.0; int cnt = 0;
      for (int a = 0; a < nl; ++a) if (liv[a] != rep_vi) { const int i = (int)liv[a], j = (int)rep_vi; u += i < j ? D[didx(nv, i, j)] : D[didx(nv, j, i)]; ++cnt; }
      u /= cnt;
      for (int a = 0; a < nl; ++a) if (liv[a] != rep_vi) { const int i = (int)liv[a], j = (int)rep_vi; const double v = (i < j ? D[didx(nv, i, j)] : D[didx(nv, j, i)]); q += (v - u) * (v - u); }
      A.se = (float)(sqrt(q / (cnt - 1)) / sqrt((double)cnt));
    }
    int ps = -1, hp = -1; bool conflicting = false;
    if (!ign[r]) {
      for (int a = 0; a < nl; ++a) {
        const otg_read& rd = reads[valid[f + liv[a]]];
        if (ps < 0) ps = rd.ps; else if (ps != rd.ps) conflicting = true;
        if (hp < 0) hp = rd.hp; else if (hp != rd.hp) conflicting = true;
      }
    }
    if (conflicting) { status[r] = OTG_REGION_HAP_CONFLICT; return; }                 // :249-254 exit(1)
    const otg_read rep_read = reads[rep];
    A.ps = ign[r] ? -1 : rep_read.ps; A.hp = ign[r] ? -1 : rep_read.hp;
    A.src_off = 0; A.src_len = 0;
    if (n_all + 1 <= 2) {
      const otg_read& fr = reads[valid[f + liv[0]]];
      A.kind = AL_COPY; A.src_off = fr.seq_off; A.src_len = fr.seq_len;                // :259
    } else {
      A.kind = AL_POA;
      otg_poa_graph G;
      G.backbone_off = rep_read.seq_off; G.backbone_len = rep_read.seq_len; G.first_member = mpos; G.n_members = (uint32_t)n_all;
      float c = (float)((n_all + 1) * 0.4);                                           // :285-287
      if (n_all + 1 < 4) c = 1.0f;
      G.c = c; G.t = 0.3f; G._pad = 0;
      graphs[f + label] = G;
      uint64_t prev_cig_read = (uint64_t)-1;
      for (int i = 0; i < n; ++i) {
        if (f + i == rep || labels[f + i] != label) continue;
        const otg_read rd = reads[f + i];
        const int length_diff = (int)rep_read.seq_len - (int)rd.seq_len;
        const bool sp = rd.spanning_l && rd.spanning_r;
        int ef = 0, pbf = 0, pef = 0, tbf = 0, tef = 0; bool do_align = true;
        if (sp || length_diff < 0) {                                                  // :267-273
          if (length_diff >= 0) { }
          else if (rd.spanning_l) { ef = 1; tef = -length_diff; }
          else if (rd.spanning_r) { ef = 1; tbf = -length_diff; }
          else do_align = false;       // reference performs no alignment and re-reads the aligner's previous CIGAR
        } else {                                                                      // :274-279
          ef = 1;
          if (rd.spanning_l) pef = length_diff;
          else if (rd.spanning_r) pbf = length_diff;
          else { pbf = length_diff / 2; pef = length_diff / 2; }
        }
        otg_poa_member M;
        M.seq_off = rd.seq_off; M.seq_len = rd.seq_len; M.spanning_l = rd.spanning_l; M.spanning_r = rd.spanning_r;
        for (int q = 0; q < 6; ++q) M._pad[q] = 0;
        if (do_align) {
          make_task(tasks[f + i], rep_read.seq_off, rep_read.seq_len, rd.seq_off, rd.seq_len, ef, pbf, pef, tbf, tef);
          todo[atomicAdd(n_todo, 1u)] = f + i;
          prev_cig_read = f + i;
        }
        // cigar_len is filled by K_member_cigars after the aligner ran; remember which read's op string to use
        M.cigar_off = prev_cig_read == (uint64_t)-1 ? (uint64_t)-1 : cig_off[prev_cig_read];
        M.cigar_len = prev_cig_read == (uint64_t)-1 ? 0u : (uint32_t)prev_cig_read;   // temporarily the read index
        members[mpos++] = M;
      }
    }
    alleles[f + label] = A;
  }
}

__global__ void K_member_cigars(otg_poa_member* __restrict__ members, const otg_poa_graph* __restrict__ graphs, uint32_t n_slots,
                                const uint32_t* __restrict__ cig_len)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_slots) return;
  const otg_poa_graph G = graphs[g];
  for (uint32_t m = 0; m < G.n_members; ++m) {
    otg_poa_member& M = members[G.first_member + m];
    if (M.cigar_off == (uint64_t)-1) { M.cigar_off = 0; M.cigar_len = 0; }
    else M.cigar_len = cig_len[M.cigar_len];
  }
}

// ---- allele sequence lengths (":291-292 empty -> N")
__global__ void K_allele_len(const AlleleSlot* __restrict__ alleles, const int32_t* __restrict__ status,
                             const uint32_t* __restrict__ read_region, const uint32_t* __restrict__ poa_len, const int32_t* __restrict__ poa_status,
                             uint32_t n_slots, uint32_t* __restrict__ al_len, uint32_t* __restrict__ al_flag, int32_t* __restrict__ status_out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  uint32_t len = 0, flag = 0;
  const uint32_t r = read_region[i];
  if (r != 0xffffffffu && status[r] == OTG_REGION_OK && alleles[i].kind != AL_NONE) {
    flag = 1;
    if (alleles[i].kind == AL_COPY) len = alleles[i].src_len;
    else {
      if (poa_status[i]) status_out[r] = OTG_ERR_FATAL;
      len = poa_len[i] ? poa_len[i] : 1;
    }
  }
  al_len[i] = len; al_flag[i] = flag;
}

// single-block exclusive scan (u32 -> u64), total to *total
__global__ __launch_bounds__(1024) void K_scan(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n, uint64_t* __restrict__ total)
{
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t chunk = (n + 1023) / 1024;
  const uint32_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
  uint64_t s = 0;
  for (uint32_t i = a; i < b; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) { uint64_t acc = 0; for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = acc; acc += v; } *total = acc; }
  __syncthreads();
  uint64_t acc = part[t];
  for (uint32_t i = a; i < b; ++i) { out[i] = acc; acc += in[i]; }
}

__global__ void K_gather(const uint8_t* __restrict__ arena, const AlleleSlot* __restrict__ alleles, const uint32_t* __restrict__ al_len,
                         const uint32_t* __restrict__ al_flag, const uint64_t* __restrict__ al_off, const uint64_t* __restrict__ al_idx,
                         const uint32_t* __restrict__ read_region, const otg_region* __restrict__ regions, const int32_t* __restrict__ ic,
                         const uint8_t* __restrict__ poa_out, const uint64_t* __restrict__ poa_node_off, const uint32_t* __restrict__ poa_start,
                         const uint32_t* __restrict__ poa_len, uint32_t n_slots, uint8_t* __restrict__ out_seq, otg_allele* __restrict__ out_al)
{
  for (uint32_t i = blockIdx.x; i < n_slots; i += gridDim.x) {
    if (!al_flag[i]) continue;
    const AlleleSlot A = alleles[i];
    const uint32_t len = al_len[i];
    uint8_t* dst = out_seq + al_off[i];
    if (A.kind == AL_COPY) { const uint8_t* src = arena + A.src_off; for (uint32_t q = threadIdx.x; q < len; q += blockDim.x) dst[q] = src[q]; }
    else if (poa_len[i] == 0) { if (threadIdx.x == 0) dst[0] = 'N'; }
    else { const uint8_t* src = poa_out + poa_node_off[i] + poa_start[i]; for (uint32_t q = threadIdx.x; q < len; q += blockDim.x) dst[q] = src[q]; }
    if (threadIdx.x == 0) {
      const uint32_t r = read_region[i];
      otg_allele o;
      o.seq_off = al_off[i]; o.seq_len = len; o.scov = A.scov; o.acov = A.acov; o.tcov = A.tcov; o.se = A.se; o.ic = ic[r];
      o.ps = A.ps; o.hp = A.hp; o.region = r; o.label = (int32_t)(i - regions[r].first_read);
      out_al[al_idx[i]] = o;
    }
  }
}

__global__ void K_region_results(const otg_region* __restrict__ regions, uint32_t n_regions, const int32_t* __restrict__ status,
                                 const int32_t* __restrict__ ic, const int32_t* __restrict__ fc, const uint32_t* __restrict__ n_valid,
                                 const uint64_t* __restrict__ al_idx, const uint32_t* __restrict__ al_flag,
                                 const int32_t* __restrict__ clerr, otg_region_result* __restrict__ out)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_regions) return;
  otg_region_result o;
  o.status = status[r]; o.ic = 0; o.fc = 0; o.n_valid = (int32_t)n_valid[r]; o.first_allele = 0; o.n_alleles = 0;
  if (o.status == OTG_REGION_OK && clerr[r]) o.status = OTG_ERR_FATAL;   // otter_find_clustering_dist exit(1) paths, src/otterclust.cpp:39-109
  if (regions[r].n_reads) o.first_allele = (uint32_t)al_idx[regions[r].first_read];
  if (o.status == OTG_REGION_OK) {
    o.ic = ic[r]; o.fc = fc[r];
    uint32_t na = 0;
    for (uint32_t i = 0; i < regions[r].n_reads; ++i) na += al_flag[regions[r].first_read + i];
    o.n_alleles = na;
  }
  out[r] = o;
}

__global__ void K_fill_f64(double* p, size_t n, double v)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static void dbg(otg_ctx* ctx, const char* what)
{
  if (!getenv("OTG_DEBUG")) return;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  hipError_t e2 = hipGetLastError();
  fprintf(stderr, "[otg] %s: %s / %s\n", what, hipGetErrorString(e), hipGetErrorString(e2));
}

struct Timer {
  otg_ctx* ctx; std::chrono::steady_clock::time_point t0;
  explicit Timer(otg_ctx* c) : ctx(c) { (void)hipStreamSynchronize(c->stream); t0 = std::chrono::steady_clock::now(); }
  double ms() { (void)hipStreamSynchronize(ctx->stream); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

} // namespace

extern "C" {

int otg_assemble_submit(otg_ctx* ctx, const otg_params* params, const uint8_t* seq_arena, uint64_t arena_bytes,
                        const otg_read* reads, uint32_t n_reads, const otg_region* regions, uint32_t n_regions)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_assemble_submit: no context (no HIP device?)");
  if (!params || (n_reads && (!reads || !seq_arena)) || (n_regions && !regions)) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_submit: NULL argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->pipe) { ctx->pipe = new Pipeline(); ctx->pipe->buf.resize(B_COUNT); }
  Pipeline* pl = ctx->pipe;
  pl->P = *params; pl->n_reads = n_reads; pl->n_regions = n_regions; pl->arena_bytes = arena_bytes; pl->ran = false;
  pl->h_reads.assign(reads, reads + n_reads);
  pl->h_regions.assign(regions, regions + n_regions);
  std::vector<uint32_t> read_region(n_reads, 0xffffffffu);   // reads outside every region are ignored
  std::vector<uint8_t> covered(n_reads, 0);
  pl->h_dist_off.assign(n_regions + 1, 0); pl->h_re_off.assign(n_regions + 1, 0); pl->h_cig_off.assign((size_t)n_reads + 1, 0);
  uint32_t maxlen = 1;
  std::vector<uint32_t> region_maxlen(n_regions, 1);
  for (uint32_t r = 0; r < n_regions; ++r) {
    const otg_region& g = regions[r];
    if ((uint64_t)g.first_read + g.n_reads > n_reads) return otg_fail(ctx, OTG_ERR_ARG, "region %u: read range out of bounds", r);
    if (g.flank_l_off + g.flank_l_len > arena_bytes || g.flank_r_off + g.flank_r_len > arena_bytes) return otg_fail(ctx, OTG_ERR_ARG, "region %u: flank outside the arena", r);
    uint64_t n = g.n_reads;
    uint32_t ml = std::max(g.flank_l_len, g.flank_r_len);
    for (uint32_t i = 0; i < g.n_reads; ++i) {
      const otg_read& q = reads[g.first_read + i];
      if (covered[g.first_read + i]) return otg_fail(ctx, OTG_ERR_ARG, "read %u belongs to two regions", g.first_read + i);
      covered[g.first_read + i] = 1; read_region[g.first_read + i] = r;
      if (q.seq_off + q.seq_len > arena_bytes) return otg_fail(ctx, OTG_ERR_ARG, "read %u: sequence outside the arena", g.first_read + i);
      ml = std::max(ml, q.seq_len);
    }
    region_maxlen[r] = ml; maxlen = std::max(maxlen, ml);
    const bool big = (int)g.n_reads > params->max_cov;
    pl->h_dist_off[r + 1] = pl->h_dist_off[r] + (big ? 0 : n * (n ? n - 1 : 0) / 2);
    pl->h_re_off[r + 1] = pl->h_re_off[r] + (big ? 0 : n * n);
  }
  for (uint32_t i = 0; i < n_reads; ++i)
    pl->h_cig_off[i + 1] = pl->h_cig_off[i] + (covered[i] ? (((uint64_t)region_maxlen[read_region[i]] + reads[i].seq_len + 15) & ~15ull) : 0);
  pl->n_pair_slots = pl->h_dist_off[n_regions]; pl->n_re_slots = pl->h_re_off[n_regions]; pl->cig_bytes = pl->h_cig_off[n_reads];
  if (pl->n_pair_slots >= 0xffffffffull || pl->n_re_slots >= 0xffffffffull)
    return otg_fail(ctx, OTG_ERR_CAPACITY, "batch too large: split it (pair slots %llu, reassignment slots %llu)", (unsigned long long)pl->n_pair_slots, (unsigned long long)pl->n_re_slots);
  ctx->max_seq_len = maxlen;
  std::vector<uint64_t> first64(n_regions);
  for (uint32_t r = 0; r < n_regions; ++r) first64[r] = regions[r].first_read;
  pl->rev_base = (arena_bytes + 64 + 255) & ~(uint64_t)255;
  uint8_t* d_arena = (uint8_t*)pbuf(ctx, pl, B_ARENA, pl->rev_base + arena_bytes + 64);
  void* d_reads = pbuf(ctx, pl, B_READS, (size_t)n_reads * sizeof(otg_read));
  void* d_regions = pbuf(ctx, pl, B_REGIONS, (size_t)n_regions * sizeof(otg_region));
  void* d_rr = pbuf(ctx, pl, B_READ_REGION, (size_t)n_reads * 4);
  void* d_do = pbuf(ctx, pl, B_DIST_OFF, (size_t)(n_regions + 1) * 8);
  void* d_ro = pbuf(ctx, pl, B_RE_OFF, (size_t)(n_regions + 1) * 8);
  void* d_co = pbuf(ctx, pl, B_CIG_OFF, (size_t)(n_reads + 1) * 8);
  void* d_f64 = pbuf(ctx, pl, B_FIRST_READ64, (size_t)n_regions * 8);
  if (!d_arena || !d_reads || !d_regions || !d_rr || !d_do || !d_ro || !d_co || !d_f64) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(d_arena + arena_bytes, 0, 64, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_arena + pl->rev_base + arena_bytes, 0, 64, ctx->stream));
  if (arena_bytes) HIP_TRY(ctx, hipMemcpyAsync(d_arena, seq_arena, arena_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_reads) HIP_TRY(ctx, hipMemcpyAsync(d_rr, read_region.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_regions) HIP_TRY(ctx, hipMemcpyAsync(d_regions, regions, (size_t)n_regions * sizeof(otg_region), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_do, pl->h_dist_off.data(), (size_t)(n_regions + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_ro, pl->h_re_off.data(), (size_t)(n_regions + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_co, pl->h_cig_off.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (n_regions) HIP_TRY(ctx, hipMemcpyAsync(d_f64, first64.data(), (size_t)n_regions * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // working buffers (allocated here so that otg_assemble_run does no allocation for the alignment stages)
  const size_t ntask = std::max<size_t>(std::max<size_t>(pl->n_pair_slots, pl->n_re_slots), n_reads) + 1;
  if (!pbuf(ctx, pl, B_TASKS, ntask * sizeof(otg_align_task)) || !pbuf(ctx, pl, B_TODO, ntask * 4) || !pbuf(ctx, pl, B_SCORES, ntask * 4) ||
      !pbuf(ctx, pl, B_DEN, ntask * 4) || !pbuf(ctx, pl, B_CELLS, ntask * 8) || !pbuf(ctx, pl, B_DIST, (pl->n_pair_slots + 1) * 8) ||
      !pbuf(ctx, pl, B_REDIST, (pl->n_re_slots + 1) * 8) || !pbuf(ctx, pl, B_CNT, 64 * 8) || !pbuf(ctx, pl, B_STATS, 64 * 8) ||
      !pbuf(ctx, pl, B_STATUS, (size_t)n_regions * 4 + 4) || !pbuf(ctx, pl, B_NVALID, (size_t)n_regions * 4 + 4) ||
      !pbuf(ctx, pl, B_IGNHAPS, (size_t)n_regions * 4 + 4) || !pbuf(ctx, pl, B_VALID, (size_t)n_reads * 4 + 4) ||
      !pbuf(ctx, pl, B_VPOS, (size_t)n_reads * 4 + 4) || !pbuf(ctx, pl, B_VLEN, (size_t)n_reads * 4 + 4) ||
      !pbuf(ctx, pl, B_CLLAB, (size_t)n_reads * 4 + 4) || !pbuf(ctx, pl, B_IC, (size_t)n_regions * 4 + 4) || !pbuf(ctx, pl, B_FC, (size_t)n_regions * 4 + 4) ||
      !pbuf(ctx, pl, B_BOUNDS, (size_t)n_regions * 24 + 8) || !pbuf(ctx, pl, B_CLERR, (size_t)n_regions * 4 + 4) || !pbuf(ctx, pl, B_LABELS, (size_t)n_reads * 4 + 4) ||
      !pbuf(ctx, pl, B_RKIND, (size_t)n_reads + 4) || !pbuf(ctx, pl, B_CIGLEN, (size_t)n_reads * 4 + 4) || !pbuf(ctx, pl, B_CIG, pl->cig_bytes + 64) ||
      !pbuf(ctx, pl, B_ALLELES, (size_t)(n_reads + 1) * sizeof(AlleleSlot)) || !pbuf(ctx, pl, B_GRAPHS, (size_t)(n_reads + 1) * sizeof(otg_poa_graph)) ||
      !pbuf(ctx, pl, B_MEMBERS, (size_t)(n_reads + 1) * sizeof(otg_poa_member)) || !pbuf(ctx, pl, B_POALEN, (size_t)n_reads * 4 + 4) ||
      !pbuf(ctx, pl, B_ALLEN, (size_t)n_reads * 4 + 4) || !pbuf(ctx, pl, B_ALOFF, (size_t)n_reads * 8 + 8) || !pbuf(ctx, pl, B_ALIDX, (size_t)n_reads * 8 + 8) ||
      !pbuf(ctx, pl, B_SCAN_TMP, (size_t)n_reads * 4 + 4) || !pbuf(ctx, pl, B_TOTALS, 64) || !pbuf(ctx, pl, B_REGRES, (size_t)(n_regions + 1) * sizeof(otg_region_result)) ||
      !otg_slot(ctx, SLOT_AUX9, (pl->n_pair_slots + 1) * 8))
    return OTG_ERR_HIP;
  return OTG_OK;
}

static int assemble_run_impl(otg_ctx* ctx, bool realign_only);

int otg_assemble_run(otg_ctx* ctx) { return assemble_run_impl(ctx, false); }

// `otter assemble --reads-only -r`: only local_realignment runs (src/assemble.cpp:72-89); the trimmed reads are read back with
// otg_assemble_collect_reads
int otg_assemble_realign(otg_ctx* ctx) { return assemble_run_impl(ctx, true); }

int otg_assemble_collect_reads(otg_ctx* ctx, otg_read* reads_out, uint32_t n_reads)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_assemble_collect_reads: no context");
  Pipeline* pl = ctx->pipe;
  if (!pl) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_collect_reads: nothing submitted");
  if (n_reads != pl->n_reads || (n_reads && !reads_out)) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_collect_reads: expected room for %u reads", pl->n_reads);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_reads) HIP_TRY(ctx, hipMemcpyAsync(reads_out, pl->buf[B_READS].p, (size_t)n_reads * sizeof(otg_read), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OTG_OK;
}

static int assemble_run_body(otg_ctx* ctx, bool realign_only);

// the pipeline's aligners run under the heuristic of the submitted otg_params, whatever otg_set_heuristic left on the context for the L1 calls
static int assemble_run_impl(otg_ctx* ctx, bool realign_only)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_assemble_run: no context");
  Pipeline* pl = ctx->pipe;
  if (!pl) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_run: nothing submitted");
  const int sv[4] = {ctx->heur_strategy, ctx->heur_min_wf_len, ctx->heur_max_dist, ctx->heur_steps};
  const otg_params& P = pl->P;
  if (P.heuristic != OTG_HEURISTIC_NONE && P.heuristic != OTG_HEURISTIC_WFADAPTIVE) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_run: unknown heuristic %d in otg_params", P.heuristic);
  ctx->heur_strategy = P.heuristic;
  if (P.heuristic == OTG_HEURISTIC_WFADAPTIVE) {
    if (P.heur_min_wavefront_length < 0 || P.heur_max_distance_threshold < 0) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_run: negative heuristic parameter in otg_params");
    ctx->heur_min_wf_len = P.heur_min_wavefront_length; ctx->heur_max_dist = P.heur_max_distance_threshold;
    ctx->heur_steps = P.heur_steps_between_cutoffs < 1 ? 1 : P.heur_steps_between_cutoffs;
  }
  const int rc = assemble_run_body(ctx, realign_only);
  ctx->heur_strategy = sv[0]; ctx->heur_min_wf_len = sv[1]; ctx->heur_max_dist = sv[2]; ctx->heur_steps = sv[3];
  return rc;
}

static int assemble_run_body(otg_ctx* ctx, bool realign_only)
{
  Pipeline* pl = ctx->pipe;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const otg_params& P = pl->P;
  const uint32_t NR = pl->n_reads, NG = pl->n_regions;
  hipStream_t st = ctx->stream;
  auto B = [&](int i) { return pl->buf[i].p; };
  uint8_t* d_arena = (uint8_t*)B(B_ARENA);
  otg_read* d_reads = (otg_read*)B(B_READS);
  const otg_region* d_regions = (const otg_region*)B(B_REGIONS);
  const uint32_t* d_rr = (const uint32_t*)B(B_READ_REGION);
  otg_align_task* d_tasks = (otg_align_task*)B(B_TASKS);
  uint32_t* d_todo = (uint32_t*)B(B_TODO);
  int32_t* d_scores = (int32_t*)B(B_SCORES);
  uint32_t* d_den = (uint32_t*)B(B_DEN);
  uint64_t* d_cells = (uint64_t*)B(B_CELLS);
  double* d_dist = (double*)B(B_DIST);
  double* d_redist = (double*)B(B_REDIST);
  uint32_t* d_cnt = (uint32_t*)B(B_CNT);
  unsigned long long* d_stats = (unsigned long long*)B(B_STATS);
  int32_t* d_status = (int32_t*)B(B_STATUS);
  uint32_t* d_nvalid = (uint32_t*)B(B_NVALID);
  int32_t* d_ign = (int32_t*)B(B_IGNHAPS);
  uint32_t* d_valid = (uint32_t*)B(B_VALID);
  int32_t* d_vpos = (int32_t*)B(B_VPOS);
  uint32_t* d_vlen = (uint32_t*)B(B_VLEN);
  int32_t* d_labels = (int32_t*)B(B_LABELS);
  uint64_t* d_cig_off = (uint64_t*)B(B_CIG_OFF);
  uint32_t* d_cig_len = (uint32_t*)B(B_CIGLEN);
  uint8_t* d_cig = (uint8_t*)B(B_CIG);
  memset(&pl->stats, 0, sizeof(pl->stats));
  pl->ran = false;               // a run that fails below must not leave the previous run's results collectable
  pl->out_alleles = 0; pl->out_seq_bytes = 0;
  if (NG == 0 || NR == 0) { pl->ran = true; return OTG_OK; }
  Timer total(ctx);
  // the resident read descriptors are restored from the submitted ones (realignment mutates them)
  HIP_TRY(ctx, hipMemcpyAsync(d_reads, pl->h_reads.data(), (size_t)NR * sizeof(otg_read), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 64 * 8, st));
  HIP_TRY(ctx, hipMemsetAsync(d_stats, 0, 64 * 8, st));
  if (ctx->affine_visited) HIP_TRY(ctx, hipMemsetAsync(ctx->affine_visited, 0, 8, st));
  const int TB = 256;
  const uint32_t gr_reads = (NR + TB - 1) / TB, gr_regions = (NG + TB - 1) / TB;
  const uint32_t gr_blocks = std::min<uint32_t>(NG, (uint32_t)ctx->n_cu * 16);
  int rc;
  // ------------------------------------------------------------------ [local_realignment]
  if (P.realign) {
    Timer t(ctx);
    hipLaunchKernelGGL(K_realign_prepare, dim3(gr_reads), dim3(TB), 0, st, d_reads, d_regions, d_rr, NR, P.flank, d_tasks, (uint8_t*)B(B_RKIND), d_todo, d_cnt + 16);
    HIP_TRY(ctx, hipMemsetAsync(d_cig_len, 0, (size_t)NR * 4, st));
    { float kms = 0; uint64_t kl = 0;
      rc = otg_launch_affine_todo(ctx, d_arena, d_tasks, d_todo, d_cnt + 16, NR, P.mismatch, P.gap_open, P.gap_ext, d_scores, d_cig_off, d_cig_len, d_cig, d_cells, &kms, &kl);
      if (rc) return rc;
      pl->stats.ms_affine_kernel += kms; pl->stats.affine_kernel_launches += kl; }
    hipLaunchKernelGGL(K_stats, dim3(64), dim3(256), 0, st, d_todo, d_cnt + 16, d_tasks, d_cells, d_stats + 4, d_scores, d_cnt + 42);
    hipLaunchKernelGGL(K_realign_apply, dim3(gr_reads), dim3(TB), 0, st, d_reads, NR, (const uint8_t*)B(B_RKIND), d_cig, d_cig_off, d_cig_len, P.flank, P.min_sim);
    pl->stats.ms_realign = t.ms();
  }
  dbg(ctx, "realign done");
  if (realign_only) {
    HIP_TRY(ctx, hipStreamSynchronize(st));
    pl->stats.ms_total = total.ms();
    uint32_t hf = 0;
    HIP_TRY(ctx, hipMemcpy(&hf, d_cnt + 42, sizeof(hf), hipMemcpyDeviceToHost));
    if (hf) return otg_fail(ctx, OTG_ERR_CAPACITY, "a gap-affine alignment exhausted its backtrace storage on the device");
    return OTG_OK;             // pl->ran stays false: there are no allele results to collect
  }
  // ------------------------------------------------------------------ partition + fill_dist_matrix
  {
    Timer t(ctx);
    hipLaunchKernelGGL(K_region_prepare, dim3(gr_regions), dim3(TB), 0, st, d_reads, d_regions, NG, P.max_cov, P.ignore_haps, d_status, d_nvalid, d_ign, d_valid, d_vpos, d_vlen);
    hipLaunchKernelGGL(K_pair_tasks, dim3(gr_blocks), dim3(64), 0, st, d_arena, d_reads, d_regions, NG, d_nvalid, d_ign, d_valid, (const uint64_t*)B(B_DIST_OFF), P.max_alleles, d_tasks, d_den, d_dist, d_todo, d_cnt + 20);
    float kms = 0; uint64_t kl = 0;
    dbg(ctx, "pair tasks done");
    rc = otg_launch_edit_todo(ctx, d_arena, d_tasks, d_todo, d_cnt + 20, (uint32_t)pl->n_pair_slots, d_scores, d_cells, &kms, &kl);
    if (rc) return rc;
    pl->stats.ms_edit_kernel += kms; pl->stats.edit_kernel_launches += kl;
    hipLaunchKernelGGL(K_dist_epilogue, dim3(1024), dim3(256), 0, st, d_todo, d_cnt + 20, d_scores, d_den, d_dist, d_cnt + 40);
    hipLaunchKernelGGL(K_stats, dim3(64), dim3(256), 0, st, d_todo, d_cnt + 20, d_tasks, d_cells, d_stats + 0, d_scores, d_cnt + 40);
    pl->stats.ms_edit = t.ms();
  }
  dbg(ctx, "edit done");
  // ------------------------------------------------------------------ otter_hclust
  {
    Timer t(ctx);
    rc = otg_launch_cluster(ctx, &P, d_dist, (const uint64_t*)B(B_DIST_OFF), d_vlen, (const uint64_t*)B(B_FIRST_READ64), d_nvalid, NG,
                            (int32_t*)B(B_CLLAB), (int32_t*)B(B_IC), (int32_t*)B(B_FC), (double*)B(B_BOUNDS), (int32_t*)B(B_CLERR));
    if (rc) return rc;
    hipLaunchKernelGGL(K_scatter_labels, dim3(gr_reads), dim3(TB), 0, st, d_rr, d_regions, d_vpos, (const int32_t*)B(B_CLLAB), NR, d_labels);
    pl->stats.ms_cluster = t.ms();
  }
  dbg(ctx, "cluster done");
  // ------------------------------------------------------------------ invalid_reassignment
  {
    Timer t(ctx);
    static const bool no_rev = getenv("OTG_NO_REASSIGN_REV") != nullptr;
    // (under the adaptive heuristic an alignment and its mirror image are different computations: no reversed copies there)
    const uint64_t rev_base = (no_rev || ctx->heur_strategy != OTG_HEURISTIC_NONE) ? 0 : pl->rev_base;
    if (rev_base) hipLaunchKernelGGL(K_reverse_reads, dim3(std::min<uint32_t>((NR + 3) / 4, (uint32_t)ctx->n_cu * 32)), dim3(256), 0, st, d_arena, d_reads, d_regions, d_rr, NR,
                                     d_status, d_nvalid, rev_base);
    hipLaunchKernelGGL(K_reassign_tasks, dim3(gr_blocks), dim3(64), 0, st, d_arena, d_reads, d_regions, NG, d_status, d_nvalid, (const int32_t*)B(B_FC), d_labels,
                       (const uint64_t*)B(B_RE_OFF), d_tasks, d_den, d_redist, d_todo, d_cnt + 24, rev_base);
    float kms = 0; uint64_t kl = 0;
    ctx->edit_pass_kind = 1;
    rc = otg_launch_edit_todo(ctx, d_arena, d_tasks, d_todo, d_cnt + 24, (uint32_t)pl->n_re_slots, d_scores, d_cells, &kms, &kl);
    ctx->edit_pass_kind = 0;
    if (rc) return rc;
    pl->stats.ms_edit_kernel += kms; pl->stats.edit_kernel_launches += kl;
    hipLaunchKernelGGL(K_dist_epilogue, dim3(1024), dim3(256), 0, st, d_todo, d_cnt + 24, d_scores, d_den, d_redist, d_cnt + 40);
    hipLaunchKernelGGL(K_stats, dim3(64), dim3(256), 0, st, d_todo, d_cnt + 24, d_tasks, d_cells, d_stats + 0, d_scores, d_cnt + 40);
    hipLaunchKernelGGL(K_reassign_apply, dim3((NG + 63) / 64), dim3(64), 0, st, d_reads, d_regions, NG, d_status, d_nvalid, (const int32_t*)B(B_FC),
                       (const uint64_t*)B(B_RE_OFF), d_redist, P.min_sim, P.max_error, d_labels, d_status);
    pl->stats.ms_reassign = t.ms();
  }
  dbg(ctx, "reassign done");
  // ------------------------------------------------------------------ rapid_consensus
  std::vector<uint64_t> node_off;
  {
    Timer t(ctx);
    HIP_TRY(ctx, hipMemsetAsync(d_cig_len, 0, (size_t)NR * 4, st));
    HIP_TRY(ctx, hipMemsetAsync(B(B_ALLELES), 0, (size_t)NR * sizeof(AlleleSlot), st));     // slots of reads outside every region stay empty
    HIP_TRY(ctx, hipMemsetAsync(B(B_GRAPHS), 0, (size_t)NR * sizeof(otg_poa_graph), st));
    HIP_TRY(ctx, hipMemsetAsync(B(B_MEMBERS), 0, (size_t)(NR + 1) * sizeof(otg_poa_member), st));   // unused member slots: empty op strings
    hipLaunchKernelGGL(K_consensus_prepare, dim3((NG + 63) / 64), dim3(64), 0, st, d_reads, d_regions, NG, d_status, d_nvalid, d_ign, d_valid, d_labels,
                       (const int32_t*)B(B_FC), (const uint64_t*)B(B_DIST_OFF), d_dist, d_cig_off, (AlleleSlot*)B(B_ALLELES),
                       (otg_poa_graph*)B(B_GRAPHS), (otg_poa_member*)B(B_MEMBERS), d_tasks, d_todo, d_cnt + 28, (uint32_t*)B(B_SCAN_TMP));
    { float kms = 0; uint64_t kl = 0;
      rc = otg_launch_affine_todo(ctx, d_arena, d_tasks, d_todo, d_cnt + 28, NR, P.mismatch, P.gap_open, P.gap_ext, d_scores, d_cig_off, d_cig_len, d_cig, d_cells, &kms, &kl);
      if (rc) return rc;
      pl->stats.ms_affine_kernel += kms; pl->stats.affine_kernel_launches += kl; }
    hipLaunchKernelGGL(K_stats, dim3(64), dim3(256), 0, st, d_todo, d_cnt + 28, d_tasks, d_cells, d_stats + 4, d_scores, d_cnt + 41);
    hipLaunchKernelGGL(K_mark_failed_affine, dim3(64), dim3(256), 0, st, d_todo, d_cnt + 28, d_scores, d_rr, d_status);      // such a region drops out; the batch goes on
    hipLaunchKernelGGL(K_member_cigars, dim3(gr_reads), dim3(TB), 0, st, (otg_poa_member*)B(B_MEMBERS), (const otg_poa_graph*)B(B_GRAPHS), NR, d_cig_len);
    pl->stats.ms_affine = t.ms();
    dbg(ctx, "affine done");
    Timer t2(ctx);
    std::vector<otg_poa_graph> h_graphs(NR);
    HIP_TRY(ctx, hipMemcpyAsync(h_graphs.data(), B(B_GRAPHS), (size_t)NR * sizeof(otg_poa_graph), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    rc = otg_launch_poa(ctx, d_arena, d_cig, (const otg_poa_member*)B(B_MEMBERS), NR, (const otg_poa_graph*)B(B_GRAPHS), h_graphs.data(), NR,
                        (uint32_t*)B(B_POALEN), node_off);
    if (rc) return rc;
    pl->stats.ms_poa = t2.ms();
  }
  dbg(ctx, "consensus done");
  // ------------------------------------------------------------------ allele records, compacted in region order
  {
    uint32_t* d_allen = (uint32_t*)B(B_ALLEN);
    uint32_t* d_alflag = (uint32_t*)B(B_SCAN_TMP);
    uint64_t* d_aloff = (uint64_t*)B(B_ALOFF);
    uint64_t* d_alidx = (uint64_t*)B(B_ALIDX);
    uint64_t* d_tot = (uint64_t*)B(B_TOTALS);
    hipLaunchKernelGGL(K_allele_len, dim3(gr_reads), dim3(TB), 0, st, (const AlleleSlot*)B(B_ALLELES), d_status, d_rr, (const uint32_t*)B(B_POALEN),
                       (const int32_t*)ctx->pool[SLOT_P28].p, NR, d_allen, d_alflag, d_status);
    hipLaunchKernelGGL(K_scan, dim3(1), dim3(1024), 0, st, d_allen, d_aloff, NR, d_tot + 0);
    hipLaunchKernelGGL(K_scan, dim3(1), dim3(1024), 0, st, d_alflag, d_alidx, NR, d_tot + 1);
    uint64_t h_tot[2];
    HIP_TRY(ctx, hipMemcpyAsync(h_tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    pl->out_seq_bytes = h_tot[0]; pl->out_alleles = (uint32_t)h_tot[1];
    uint8_t* d_outseq = (uint8_t*)pbuf(ctx, pl, B_OUTSEQ, pl->out_seq_bytes + 16);
    otg_allele* d_outal = (otg_allele*)pbuf(ctx, pl, B_OUTAL, (size_t)(pl->out_alleles + 1) * sizeof(otg_allele));
    if (!d_outseq || !d_outal) return OTG_ERR_HIP;
    uint64_t* d_nodeoff = (uint64_t*)ctx->pool[SLOT_P0].p;
    hipLaunchKernelGGL(K_gather, dim3(std::min<uint32_t>(NR, (uint32_t)ctx->n_cu * 16)), dim3(128), 0, st, d_arena, (const AlleleSlot*)B(B_ALLELES), d_allen, d_alflag,
                       d_aloff, d_alidx, d_rr, d_regions, (const int32_t*)B(B_IC), (const uint8_t*)ctx->pool[SLOT_P17].p, d_nodeoff,
                       (const uint32_t*)ctx->pool[SLOT_P29].p, (const uint32_t*)B(B_POALEN), NR, d_outseq, d_outal);
    hipLaunchKernelGGL(K_region_results, dim3(gr_regions), dim3(TB), 0, st, d_regions, NG, d_status, (const int32_t*)B(B_IC), (const int32_t*)B(B_FC), d_nvalid,
                       d_alidx, d_alflag, (const int32_t*)B(B_CLERR), (otg_region_result*)B(B_REGRES));
  }
  dbg(ctx, "gather done");
  HIP_TRY(ctx, hipGetLastError());
  pl->stats.ms_total = total.ms();
  {
    uint32_t hf[3];
    HIP_TRY(ctx, hipMemcpy(hf, d_cnt + 40, sizeof(hf), hipMemcpyDeviceToHost));
    if (hf[0]) return otg_fail(ctx, OTG_ERR_FATAL, "an edit-distance alignment did not complete on the device");
    // hf[1]: a consensus alignment outgrew the last-resort tier's workspace: its region carries OTG_REGION_ALIGN_CAPACITY, everything else is delivered
    if (hf[2]) return otg_fail(ctx, OTG_ERR_CAPACITY, "a flank re-alignment exhausted its backtrace storage on the device");
  }
  // statistics
  unsigned long long hs[8];
  HIP_TRY(ctx, hipMemcpy(hs, d_stats, sizeof(hs), hipMemcpyDeviceToHost));
  pl->stats.edit_cells = hs[0]; pl->stats.edit_seq_bytes = hs[1]; pl->stats.edit_tasks = hs[2];
  pl->stats.affine_cells = hs[4]; pl->stats.affine_seq_bytes = hs[5]; pl->stats.affine_tasks = hs[6];
  pl->stats.n_regions = NG;
  if (ctx->affine_visited) { unsigned long long v = 0; HIP_TRY(ctx, hipMemcpy(&v, ctx->affine_visited, 8, hipMemcpyDeviceToHost)); pl->stats.affine_visited_cells = v; }
  pl->stats.allele_bytes = pl->out_seq_bytes + 40ull * pl->out_alleles;
  pl->stats.algorithmic_bytes = pl->stats.edit_seq_bytes + 4 * pl->stats.edit_cells + pl->stats.affine_seq_bytes + 4 * pl->stats.affine_cells +
                                (pl->stats.affine_cells + 1) / 2 + pl->stats.allele_bytes;
  {
    std::vector<otg_region_result> rr(NG);
    HIP_TRY(ctx, hipMemcpy(rr.data(), B(B_REGRES), (size_t)NG * sizeof(otg_region_result), hipMemcpyDeviceToHost));
    uint64_t ok = 0;
    for (auto& r : rr) { if (r.status < 0) return otg_fail(ctx, OTG_ERR_FATAL, "a region hit a condition on which the reference exit(1)s (status %d)", r.status); ok += (r.n_alleles > 0); }
    pl->stats.n_regions_ok = ok;
  }
  pl->ran = true;
  return OTG_OK;
}

int otg_assemble_result_sizes(otg_ctx* ctx, uint32_t* n_alleles, uint64_t* seq_bytes)
{
  if (!ctx || !ctx->pipe || !ctx->pipe->ran) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_result_sizes: no completed run");
  if (n_alleles) *n_alleles = ctx->pipe->out_alleles;
  if (seq_bytes) *seq_bytes = ctx->pipe->out_seq_bytes;
  return OTG_OK;
}

int otg_assemble_collect(otg_ctx* ctx, otg_region_result* region_out, otg_allele* alleles_out, uint32_t allele_capacity,
                         uint8_t* seq_out, uint64_t seq_capacity, int32_t* labels_out)
{
  if (!ctx || !ctx->pipe || !ctx->pipe->ran) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_collect: no completed run");
  Pipeline* pl = ctx->pipe;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (allele_capacity < pl->out_alleles || seq_capacity < pl->out_seq_bytes) return otg_fail(ctx, OTG_ERR_CAPACITY, "otg_assemble_collect: output buffers too small");
  if (pl->n_regions == 0 || pl->n_reads == 0) {
    for (uint32_t r = 0; r < pl->n_regions; ++r) { if (region_out) { memset(&region_out[r], 0, sizeof(otg_region_result)); region_out[r].status = OTG_REGION_EMPTY; } }
    return OTG_OK;
  }
  if (region_out) HIP_TRY(ctx, hipMemcpyAsync(region_out, pl->buf[B_REGRES].p, (size_t)pl->n_regions * sizeof(otg_region_result), hipMemcpyDeviceToHost, ctx->stream));
  if (alleles_out && pl->out_alleles) HIP_TRY(ctx, hipMemcpyAsync(alleles_out, pl->buf[B_OUTAL].p, (size_t)pl->out_alleles * sizeof(otg_allele), hipMemcpyDeviceToHost, ctx->stream));
  if (seq_out && pl->out_seq_bytes) HIP_TRY(ctx, hipMemcpyAsync(seq_out, pl->buf[B_OUTSEQ].p, pl->out_seq_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (labels_out) HIP_TRY(ctx, hipMemcpyAsync(labels_out, pl->buf[B_LABELS].p, (size_t)pl->n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OTG_OK;
}

int otg_assemble_device_results(otg_ctx* ctx, const otg_region_result** d_regions, const otg_allele** d_alleles, const uint8_t** d_seqs)
{
  if (!ctx || !ctx->pipe || !ctx->pipe->ran) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_device_results: no completed run");
  Pipeline* pl = ctx->pipe;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const bool empty = pl->n_regions == 0 || pl->n_reads == 0;
  if (d_regions) *d_regions = empty ? nullptr : (const otg_region_result*)pl->buf[B_REGRES].p;
  if (d_alleles) *d_alleles = (empty || !pl->out_alleles) ? nullptr : (const otg_allele*)pl->buf[B_OUTAL].p;
  if (d_seqs) *d_seqs = (empty || !pl->out_seq_bytes) ? nullptr : (const uint8_t*)pl->buf[B_OUTSEQ].p;
  return OTG_OK;
}

int otg_assemble_stats(otg_ctx* ctx, otg_run_stats* out)
{
  if (!ctx || !ctx->pipe || !out) return otg_fail(ctx, OTG_ERR_ARG, "otg_assemble_stats: no run");
  *out = ctx->pipe->stats;
  return OTG_OK;
}

} // extern "C"
