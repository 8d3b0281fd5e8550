// otg_api.hip — context, device scratch and the L1/L2 C-ABI entry points (host side).
#include "otg_common.hpp"
#include <cmath>
#include <cstdarg>
#include <algorithm>

thread_local std::string g_otg_err;

int otg_fail(otg_ctx* ctx, int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_otg_err = buf;
  if (ctx) ctx->err = buf;
  return code;
}

std::mutex& otg_device_mutex(int device)
{
  static std::mutex m[64];
  return m[(unsigned)device % 64u];
}

void* otg_slot(otg_ctx* ctx, int slot, size_t bytes)
{
  if (bytes == 0) bytes = 16;
  DevBuf& b = ctx->pool[slot];
  if (b.cap >= bytes) return b.p;
  if (b.p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  // 25 % headroom (at most 1 GB): a slightly larger batch must not cost a hipFree + hipMalloc; the multi-gigabyte kernel workspaces are
  // sized independently of the batch by their callers and need none
  size_t want = bytes + std::min<size_t>(bytes >> 2, (size_t)1 << 30) + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess && want > bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&b.p, want); }
  if (e != hipSuccess) {
    otg_fail(ctx, OTG_ERR_HIP, "hipMalloc(%zu bytes, slot %d) failed: %s", want, slot, hipGetErrorString(e));
    b.p = nullptr;
    return nullptr;
  }
  b.cap = want;
  return b.p;
}

// ---- glibc exp() restated on the host, used ONLY to detect which build of exp() the host libm runs
// (the device KDE mirrors that build bit for bit; see cluster.hip / DESIGN.md §5). -------------------
#include "exp_table.inc"
static double host_exp_variant(double x, bool use_fma);

extern "C" {

void otg_params_default(otg_params* p)
{
  memset(p, 0, sizeof(*p));
  p->max_alleles = 2; p->ignore_haps = 1; p->max_cov = 200; p->flank = 100; p->bandwidth_length = 500;
  p->min_cov_fraction2_l = 500; p->mismatch = 4; p->gap_open = 6; p->gap_ext = 2; p->realign = 0;
  p->bandwidth_short = 0.01; p->bandwidth_long = 0.015; p->max_error = 0.01; p->min_cov_fraction = 0.2;
  p->min_cov_fraction2_f = 0.1; p->min_sim = 0.9; p->gt_max_error = 0.025; p->gt_max_cosdis = 0.025;
  p->heuristic = OTG_HEURISTIC_NONE; p->heur_min_wavefront_length = 10; p->heur_max_distance_threshold = 50; p->heur_steps_between_cutoffs = 1;
}

int otg_set_heuristic(otg_ctx* ctx, int strategy, int min_wavefront_length, int max_distance_threshold, int steps_between_cutoffs)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_set_heuristic: no context");
  if (strategy != OTG_HEURISTIC_NONE && strategy != OTG_HEURISTIC_WFADAPTIVE) return otg_fail(ctx, OTG_ERR_ARG, "otg_set_heuristic: unknown strategy %d", strategy);
  if (strategy == OTG_HEURISTIC_WFADAPTIVE && (min_wavefront_length < 0 || max_distance_threshold < 0))
    return otg_fail(ctx, OTG_ERR_ARG, "otg_set_heuristic: negative wavefront length or distance threshold");
  ctx->heur_strategy = strategy;
  if (strategy == OTG_HEURISTIC_WFADAPTIVE) {
    ctx->heur_min_wf_len = min_wavefront_length; ctx->heur_max_dist = max_distance_threshold;
    ctx->heur_steps = steps_between_cutoffs < 1 ? 1 : steps_between_cutoffs;
  }
  return OTG_OK;
}

int otg_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* otg_last_error(otg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_otg_err.c_str(); }

int otg_create(int device, otg_ctx** out)
{
  if (!out) return otg_fail(nullptr, OTG_ERR_ARG, "otg_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "no HIP device available (%s); libotter_gpu has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return otg_fail(nullptr, OTG_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
  otg_ctx* ctx = new otg_ctx();
  ctx->device = device;
  ctx->pool.resize(SLOT_COUNT);
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return otg_fail(nullptr, OTG_ERR_HIP, "HIP device %d could not be initialised", device);
  }
  ctx->n_cu = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
  // probe the host libm: does exp() round like glibc's FMA build or its non-FMA build?
  {
    int agree_fma = 0, agree_nofma = 0;
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 4096; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
      double z = u * 39.0, x = -(z * z / 2);
      double ref = std::exp(x);
      agree_fma += (memcmp(&ref, (const void*)&(const double&)(host_exp_variant(x, true)), 8) == 0);
      agree_nofma += (memcmp(&ref, (const void*)&(const double&)(host_exp_variant(x, false)), 8) == 0);
    }
    ctx->exp_variant = (agree_nofma > agree_fma) ? 0 : 1;
  }
  *out = ctx;
  return OTG_OK;
}

void otg_destroy(otg_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  otg_pipeline_free(ctx);
  for (auto& b : ctx->pool) if (b.p) (void)hipFree(b.p);
  for (int i = 0; i < 5; ++i) { if (ctx->tier_stream[i]) (void)hipStreamDestroy(ctx->tier_stream[i]); if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->edit_hist) (void)hipHostFree(ctx->edit_hist);
  for (int i = 0; i < 2; ++i) if (ctx->edit_hist_ev[i]) (void)hipEventDestroy(ctx->edit_hist_ev[i]);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int otg_trim(otg_ctx* ctx)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_trim: no context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the aligners' per-launch workspaces (provenance slabs, row tables, op lists): nothing in them outlives a call
  for (int slot : {SLOT_WF_WS, SLOT_REVOPS, SLOT_BT_POOL}) {
    DevBuf& b = ctx->pool[slot];
    if (b.p) { HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  }
  return OTG_OK;
}

int otg_exp_variant(otg_ctx* ctx) { return ctx ? ctx->exp_variant : -1; }

static uint32_t max_len_of(const otg_align_task* tasks, uint32_t n)
{
  uint32_t m = 1;
  for (uint32_t i = 0; i < n; ++i) {
    if (tasks[i].pattern_len > m) m = tasks[i].pattern_len;
    if (tasks[i].text_len > m) m = tasks[i].text_len;
  }
  return m;
}

static int check_tasks(otg_ctx* ctx, const otg_align_task* tasks, uint32_t n, uint64_t arena_bytes)
{
  for (uint32_t i = 0; i < n; ++i) {
    const otg_align_task& t = tasks[i];
    if (t.pattern_off + t.pattern_len > arena_bytes || t.text_off + t.text_len > arena_bytes)
      return otg_fail(ctx, OTG_ERR_ARG, "task %u: sequence range outside the arena", i);
    if (t.endsfree && (t.pattern_begin_free < 0 || t.pattern_end_free < 0 || t.text_begin_free < 0 || t.text_end_free < 0))
      return otg_fail(ctx, OTG_ERR_ARG, "task %u: negative free-end length", i);
  }
  return OTG_OK;
}

int otg_edit_distance_batch(otg_ctx* ctx, const uint8_t* seq_arena, uint64_t arena_bytes,
                            const otg_align_task* tasks, uint32_t n_tasks, int32_t* scores_out, uint64_t* cells_out)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_edit_distance_batch: no context (no HIP device?)");
  if (n_tasks == 0) return OTG_OK;
  if (!seq_arena || !tasks || !scores_out) return otg_fail(ctx, OTG_ERR_ARG, "otg_edit_distance_batch: NULL argument");
  int rc = check_tasks(ctx, tasks, n_tasks, arena_bytes);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t* d_arena = (uint8_t*)otg_slot(ctx, SLOT_ARENA, arena_bytes + 64);
  otg_align_task* d_tasks = (otg_align_task*)otg_slot(ctx, SLOT_TASKS, (size_t)n_tasks * sizeof(otg_align_task));
  int32_t* d_scores = (int32_t*)otg_slot(ctx, SLOT_SCORES, (size_t)n_tasks * sizeof(int32_t));
  uint64_t* d_cells = (uint64_t*)otg_slot(ctx, SLOT_CELLS, (size_t)n_tasks * sizeof(uint64_t));
  if (!d_arena || !d_tasks || !d_scores || !d_cells) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(d_arena + arena_bytes, 0, 64, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_arena, seq_arena, arena_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_tasks, tasks, (size_t)n_tasks * sizeof(otg_align_task), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_scores, 0xff, (size_t)n_tasks * sizeof(int32_t), ctx->stream));
  ctx->max_seq_len = max_len_of(tasks, n_tasks);
  rc = otg_launch_edit(ctx, d_arena, d_tasks, n_tasks, d_scores, d_cells, nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(scores_out, d_scores, (size_t)n_tasks * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (cells_out) HIP_TRY(ctx, hipMemcpyAsync(cells_out, d_cells, (size_t)n_tasks * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n_tasks; ++i)
    if (scores_out[i] < 0) return otg_fail(ctx, OTG_ERR_FATAL, "edit task %u did not terminate", i);
  return OTG_OK;
}

int otg_affine_align_batch(otg_ctx* ctx, const uint8_t* seq_arena, uint64_t arena_bytes,
                           const otg_align_task* tasks, uint32_t n_tasks,
                           int32_t mismatch, int32_t gap_open, int32_t gap_ext,
                           int32_t* scores_out, uint64_t* cigar_off_out, uint32_t* cigar_len_out,
                           uint8_t* cigar_arena, uint64_t cigar_capacity, uint64_t* cigar_bytes_used,
                           uint64_t* cells_out)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_affine_align_batch: no context (no HIP device?)");
  if (cigar_bytes_used) *cigar_bytes_used = 0;
  if (n_tasks == 0) return OTG_OK;
  if (!seq_arena || !tasks || !scores_out || !cigar_off_out || !cigar_len_out || !cigar_arena)
    return otg_fail(ctx, OTG_ERR_ARG, "otg_affine_align_batch: NULL argument");
  int rc = check_tasks(ctx, tasks, n_tasks, arena_bytes);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // device CIGAR slots: task i may emit at most pattern_len + text_len ops
  std::vector<uint64_t> slot(n_tasks + 1);
  slot[0] = 0;
  for (uint32_t i = 0; i < n_tasks; ++i) slot[i + 1] = slot[i] + (((uint64_t)tasks[i].pattern_len + tasks[i].text_len + 15) & ~15ull);
  uint8_t* d_arena = (uint8_t*)otg_slot(ctx, SLOT_ARENA, arena_bytes + 64);
  otg_align_task* d_tasks = (otg_align_task*)otg_slot(ctx, SLOT_TASKS, (size_t)n_tasks * sizeof(otg_align_task));
  int32_t* d_scores = (int32_t*)otg_slot(ctx, SLOT_SCORES, (size_t)n_tasks * sizeof(int32_t));
  uint64_t* d_cells = (uint64_t*)otg_slot(ctx, SLOT_CELLS, (size_t)n_tasks * sizeof(uint64_t));
  uint64_t* d_off = (uint64_t*)otg_slot(ctx, SLOT_CIG_OFF, (size_t)(n_tasks + 1) * sizeof(uint64_t));
  uint32_t* d_len = (uint32_t*)otg_slot(ctx, SLOT_CIG_LEN, (size_t)n_tasks * sizeof(uint32_t));
  uint8_t* d_cig = (uint8_t*)otg_slot(ctx, SLOT_CIG_ARENA, slot[n_tasks] + 64);
  if (!d_arena || !d_tasks || !d_scores || !d_cells || !d_off || !d_len || !d_cig) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemsetAsync(d_arena + arena_bytes, 0, 64, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_arena, seq_arena, arena_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_tasks, tasks, (size_t)n_tasks * sizeof(otg_align_task), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_off, slot.data(), (size_t)(n_tasks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_scores, 0xff, (size_t)n_tasks * sizeof(int32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_len, 0, (size_t)n_tasks * sizeof(uint32_t), ctx->stream));
  ctx->max_seq_len = max_len_of(tasks, n_tasks);
  rc = otg_launch_affine(ctx, d_arena, d_tasks, n_tasks, mismatch, gap_open, gap_ext, d_scores, d_off, d_len, d_cig, d_cells);
  if (rc) return rc;
  std::vector<uint8_t> h_cig(slot[n_tasks] + 1);
  HIP_TRY(ctx, hipMemcpyAsync(scores_out, d_scores, (size_t)n_tasks * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(cigar_len_out, d_len, (size_t)n_tasks * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (cells_out) HIP_TRY(ctx, hipMemcpyAsync(cells_out, d_cells, (size_t)n_tasks * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_cig.data(), d_cig, slot[n_tasks], hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t pos = 0;
  rc = OTG_OK;
  for (uint32_t i = 0; i < n_tasks; ++i) {
    if (scores_out[i] < 0)
      return otg_fail(ctx, scores_out[i] == -1 ? OTG_ERR_CAPACITY : OTG_ERR_FATAL,
                      "affine task %u failed on the device (code %d: -1 = backtrace storage exhausted)", i, scores_out[i]);
    cigar_off_out[i] = pos;
    if (pos + cigar_len_out[i] <= cigar_capacity) memcpy(cigar_arena + pos, h_cig.data() + slot[i], cigar_len_out[i]);
    else rc = OTG_ERR_CAPACITY;
    pos += cigar_len_out[i];
  }
  if (cigar_bytes_used) *cigar_bytes_used = pos;
  if (rc) return otg_fail(ctx, rc, "cigar_capacity %llu too small, %llu needed", (unsigned long long)cigar_capacity, (unsigned long long)pos);
  return OTG_OK;
}

int otg_cluster_batch(otg_ctx* ctx, const otg_params* params,
                      const double* dist, const uint64_t* dist_off,
                      const uint32_t* read_len, const uint64_t* len_off,
                      const uint32_t* n_valid, uint32_t n_regions,
                      int32_t* labels_out, int32_t* ic_out, int32_t* fc_out, double* bounds_out)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_cluster_batch: no context (no HIP device?)");
  if (n_regions == 0) return OTG_OK;
  if (!params || !dist_off || !read_len || !len_off || !n_valid || !labels_out || !ic_out || !fc_out)
    return otg_fail(ctx, OTG_ERR_ARG, "otg_cluster_batch: NULL argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t n_dist = 0, n_len = 0;
  for (uint32_t r = 0; r < n_regions; ++r) {
    uint64_t n = n_valid[r];
    n_dist = std::max<uint64_t>(n_dist, dist_off[r] + n * (n ? n - 1 : 0) / 2);
    n_len = std::max<uint64_t>(n_len, len_off[r] + n);
  }
  if (n_dist && !dist) return otg_fail(ctx, OTG_ERR_ARG, "otg_cluster_batch: dist is NULL");
  double* d_dist = (double*)otg_slot(ctx, SLOT_AUX0, (n_dist + 1) * sizeof(double));
  uint64_t* d_doff = (uint64_t*)otg_slot(ctx, SLOT_AUX1, (size_t)n_regions * sizeof(uint64_t));
  uint32_t* d_len = (uint32_t*)otg_slot(ctx, SLOT_AUX2, (n_len + 1) * sizeof(uint32_t));
  uint64_t* d_loff = (uint64_t*)otg_slot(ctx, SLOT_AUX3, (size_t)n_regions * sizeof(uint64_t));
  uint32_t* d_nv = (uint32_t*)otg_slot(ctx, SLOT_AUX4, (size_t)n_regions * sizeof(uint32_t));
  int32_t* d_lab = (int32_t*)otg_slot(ctx, SLOT_AUX5, (n_len + 1) * sizeof(int32_t));
  int32_t* d_ic = (int32_t*)otg_slot(ctx, SLOT_AUX6, (size_t)n_regions * 3 * sizeof(int32_t));
  double* d_bounds = (double*)otg_slot(ctx, SLOT_AUX7, (size_t)n_regions * 3 * sizeof(double));
  double* d_work = (double*)otg_slot(ctx, SLOT_AUX9, (n_dist + 1) * sizeof(double));
  if (!d_dist || !d_doff || !d_len || !d_loff || !d_nv || !d_lab || !d_ic || !d_bounds || !d_work) return OTG_ERR_HIP;
  int32_t* d_fc = d_ic + n_regions;
  int32_t* d_err = d_fc + n_regions;
  if (n_dist) HIP_TRY(ctx, hipMemcpyAsync(d_dist, dist, n_dist * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_doff, dist_off, (size_t)n_regions * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_len, read_len, n_len * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_loff, len_off, (size_t)n_regions * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_nv, n_valid, (size_t)n_regions * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_lab, 0xff, (n_len + 1) * sizeof(int32_t), ctx->stream));
  int rc = otg_launch_cluster(ctx, params, d_dist, d_doff, d_len, d_loff, d_nv, n_regions, d_lab, d_ic, d_fc, d_bounds, d_err);
  if (rc) return rc;
  std::vector<int32_t> h_err(n_regions);
  HIP_TRY(ctx, hipMemcpyAsync(labels_out, d_lab, n_len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ic_out, d_ic, (size_t)n_regions * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(fc_out, d_fc, (size_t)n_regions * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_err.data(), d_err, (size_t)n_regions * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (bounds_out) HIP_TRY(ctx, hipMemcpyAsync(bounds_out, d_bounds, (size_t)n_regions * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t r = 0; r < n_regions; ++r)
    if (h_err[r])
      return otg_fail(ctx, h_err[r] == 10 ? OTG_ERR_CAPACITY : OTG_ERR_FATAL,
                      "region %u: clustering failed with code %d (1-4: the reference exit(1)s here, src/otterclust.cpp:39-109; "
                      "5: std::sort emulation depth; 10: more than 256 valid reads)", r, h_err[r]);
  return OTG_OK;
}

int otg_poa_consensus_batch(otg_ctx* ctx, const uint8_t* seq_arena, uint64_t arena_bytes,
                            const uint8_t* cigar_arena, uint64_t cigar_bytes,
                            const otg_poa_member* members, uint32_t n_members,
                            const otg_poa_graph* graphs, uint32_t n_graphs,
                            uint64_t* out_off, uint32_t* out_len,
                            uint8_t* out_arena, uint64_t out_capacity, uint64_t* out_bytes_used)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_poa_consensus_batch: no context (no HIP device?)");
  if (out_bytes_used) *out_bytes_used = 0;
  if (n_graphs == 0) return OTG_OK;
  if (!seq_arena || !graphs || !out_off || !out_len || !out_arena || (n_members && (!members || !cigar_arena)))
    return otg_fail(ctx, OTG_ERR_ARG, "otg_poa_consensus_batch: NULL argument");
  for (uint32_t g = 0; g < n_graphs; ++g) {
    if (graphs[g].backbone_off + graphs[g].backbone_len > arena_bytes || (uint64_t)graphs[g].first_member + graphs[g].n_members > n_members)
      return otg_fail(ctx, OTG_ERR_ARG, "graph %u: backbone or member range out of bounds", g);
  }
  for (uint32_t m = 0; m < n_members; ++m)
    if (members[m].seq_off + members[m].seq_len > arena_bytes || members[m].cigar_off + members[m].cigar_len > cigar_bytes)
      return otg_fail(ctx, OTG_ERR_ARG, "member %u: sequence or op string out of bounds", m);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t* d_arena = (uint8_t*)otg_slot(ctx, SLOT_ARENA, arena_bytes + 64);
  uint8_t* d_cig = (uint8_t*)otg_slot(ctx, SLOT_CIG_ARENA, cigar_bytes + 64);
  otg_poa_member* d_mem = (otg_poa_member*)otg_slot(ctx, SLOT_AUX0, (size_t)(n_members + 1) * sizeof(otg_poa_member));
  otg_poa_graph* d_gr = (otg_poa_graph*)otg_slot(ctx, SLOT_AUX1, (size_t)n_graphs * sizeof(otg_poa_graph));
  uint32_t* d_len = (uint32_t*)otg_slot(ctx, SLOT_AUX2, (size_t)n_graphs * sizeof(uint32_t));
  if (!d_arena || !d_cig || !d_mem || !d_gr || !d_len) return OTG_ERR_HIP;
  HIP_TRY(ctx, hipMemcpyAsync(d_arena, seq_arena, arena_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (cigar_bytes) HIP_TRY(ctx, hipMemcpyAsync(d_cig, cigar_arena, cigar_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_members) HIP_TRY(ctx, hipMemcpyAsync(d_mem, members, (size_t)n_members * sizeof(otg_poa_member), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_gr, graphs, (size_t)n_graphs * sizeof(otg_poa_graph), hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint64_t> node_off;
  int rc = otg_launch_poa(ctx, d_arena, d_cig, d_mem, n_members, d_gr, graphs, n_graphs, d_len, node_off);
  if (rc) return rc;
  std::vector<uint32_t> h_start(n_graphs);
  std::vector<int32_t> h_status(n_graphs);
  std::vector<uint8_t> h_out(node_off[n_graphs] + 1);
  HIP_TRY(ctx, hipMemcpyAsync(out_len, d_len, (size_t)n_graphs * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_start.data(), ctx->pool[SLOT_P29].p, (size_t)n_graphs * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_status.data(), ctx->pool[SLOT_P28].p, (size_t)n_graphs * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_out.data(), ctx->pool[SLOT_P17].p, node_off[n_graphs], hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t pos = 0;
  rc = OTG_OK;
  for (uint32_t g = 0; g < n_graphs; ++g) {
    if (h_status[g]) return otg_fail(ctx, OTG_ERR_FATAL, "POA graph %u failed on the device (status %d)", g, h_status[g]);
    out_off[g] = pos;
    if (pos + out_len[g] <= out_capacity) memcpy(out_arena + pos, h_out.data() + node_off[g] + h_start[g], out_len[g]);
    else rc = OTG_ERR_CAPACITY;
    pos += out_len[g];
  }
  if (out_bytes_used) *out_bytes_used = pos;
  if (rc) return otg_fail(ctx, rc, "out_capacity %llu too small, %llu needed", (unsigned long long)out_capacity, (unsigned long long)pos);
  return OTG_OK;
}

int otg_genotype_cluster_batch(otg_ctx* ctx, const otg_params* params, const uint8_t* seq_arena, uint64_t arena_bytes,
                               const uint64_t* seq_off, const uint32_t* seq_len,
                               const uint32_t* first_allele, const uint32_t* n_alleles, uint32_t n_regions,
                               int32_t* gt_out, int32_t* gt_l_out, int32_t* gt_k_out, double* hsd_out,
                               int32_t* n_gt_out, int32_t* reps_out)
{
  if (!ctx) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_genotype_cluster_batch: no context (no HIP device?)");
  if (n_regions == 0) return OTG_OK;
  if (!params || !seq_arena || !seq_off || !seq_len || !first_allele || !n_alleles || !gt_out || !gt_l_out || !gt_k_out || !hsd_out || !n_gt_out || !reps_out)
    return otg_fail(ctx, OTG_ERR_ARG, "otg_genotype_cluster_batch: NULL argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t na = 0;
  std::vector<uint64_t> pair_off(n_regions + 1, 0);
  for (uint32_t r = 0; r < n_regions; ++r) {
    na = std::max<uint64_t>(na, (uint64_t)first_allele[r] + n_alleles[r]);
    uint64_t A = n_alleles[r];
    pair_off[r + 1] = pair_off[r] + A * (A ? A - 1 : 0) / 2;
  }
  for (uint64_t i = 0; i < na; ++i) if (seq_off[i] + seq_len[i] > arena_bytes) return otg_fail(ctx, OTG_ERR_ARG, "allele %llu: sequence outside the arena", (unsigned long long)i);
  uint8_t* d_arena = (uint8_t*)otg_slot(ctx, SLOT_ARENA, arena_bytes + 64);
  uint64_t* d_off = (uint64_t*)otg_slot(ctx, SLOT_AUX0, (na + 1) * 8);
  uint32_t* d_len = (uint32_t*)otg_slot(ctx, SLOT_AUX1, (na + 1) * 4);
  uint32_t* d_first = (uint32_t*)otg_slot(ctx, SLOT_AUX2, (size_t)n_regions * 4);
  uint32_t* d_n = (uint32_t*)otg_slot(ctx, SLOT_AUX3, (size_t)n_regions * 4);
  uint64_t* d_poff = (uint64_t*)otg_slot(ctx, SLOT_AUX4, (size_t)(n_regions + 1) * 8);
  int32_t* d_gt = (int32_t*)otg_slot(ctx, SLOT_AUX5, (na + 1) * 4 * 4);
  double* d_hsd = (double*)otg_slot(ctx, SLOT_AUX6, (na + 1) * 8);
  int32_t* d_ngt = (int32_t*)otg_slot(ctx, SLOT_AUX7, (size_t)n_regions * 2 * 4);
  if (!d_arena || !d_off || !d_len || !d_first || !d_n || !d_poff || !d_gt || !d_hsd || !d_ngt) return OTG_ERR_HIP;
  int32_t *d_gtl = d_gt + (na + 1), *d_gtk = d_gtl + (na + 1), *d_reps = d_gtk + (na + 1), *d_err = d_ngt + n_regions;
  HIP_TRY(ctx, hipMemcpyAsync(d_arena, seq_arena, arena_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_off, seq_off, na * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_len, seq_len, na * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_first, first_allele, (size_t)n_regions * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_n, n_alleles, (size_t)n_regions * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_poff, pair_off.data(), (size_t)(n_regions + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_gt, 0xff, (na + 1) * 4 * 4, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = otg_launch_genotype(ctx, params, d_arena, d_off, d_len, d_first, d_n, n_regions, d_poff, pair_off[n_regions], na,
                               d_gt, d_gtl, d_gtk, d_hsd, d_ngt, d_reps, d_err);
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  std::vector<int32_t> h_err(n_regions);
  HIP_TRY(ctx, hipMemcpyAsync(gt_out, d_gt, na * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(gt_l_out, d_gtl, na * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(gt_k_out, d_gtk, na * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(reps_out, d_reps, na * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(hsd_out, d_hsd, na * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(n_gt_out, d_ngt, (size_t)n_regions * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h_err.data(), d_err, (size_t)n_regions * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  { float ms = 0; HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1)); ctx->last_kernel_ms = ms; }
  for (uint32_t r = 0; r < n_regions; ++r) if (h_err[r]) return otg_fail(ctx, OTG_ERR_CAPACITY, "region %u: more than 256 alleles", r);
  return OTG_OK;
}

int otg_last_kernel_ms(otg_ctx* ctx, double* ms)
{
  if (!ctx || !ms) return otg_fail(ctx, OTG_ERR_ARG, "otg_last_kernel_ms: NULL argument");
  *ms = ctx->last_kernel_ms;
  return OTG_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------
static inline uint64_t asu64(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double asf64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

static double host_exp_variant(double x, bool use_fma)
{
  // glibc 2.28+ exp (sysdeps/ieee754/dbl-64/e_exp.c), N = 128; use_fma mirrors the x86-64 ifunc'd FMA build
  const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double Shift = 0x1.8p52;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  auto F = [use_fma](double a, double b, double c) { return use_fma ? std::fma(a, b, c) : a * b + c; };
  uint32_t abstop = (uint32_t)(asu64(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9 >= 0x408 - 0x3c9) {
    if (abstop - 0x3c9 >= 0x80000000u) return 1.0 + x;
    if (abstop >= 0x409) {
      if (asu64(x) == asu64(-INFINITY)) return 0.0;
      if (abstop >= 0x7ff) return 1.0 + x;
      return (asu64(x) >> 63) ? 0.0 : INFINITY;
    }
    abstop = 0;
  }
  double z = InvLn2N * x;
  double kd = z + Shift;
  uint64_t ki = asu64(kd);
  kd -= Shift;
  double r = F(kd, NegLn2loN, F(kd, NegLn2hiN, x));
  uint64_t idx = 2 * (ki % 128), top = ki << 45;
  double tail = asf64(OTG_EXP_TAB[idx]);
  uint64_t sbits = OTG_EXP_TAB[idx + 1] + top;
  double r2 = r * r;
  double tmp = use_fma ? F(r2 * r2, F(r, C5, C4), F(r2, F(r, C3, C2), tail + r))
                       : tail + r + r2 * (C2 + r * C3) + r2 * r2 * (C4 + r * C5);
  if (abstop == 0) {
    double scale, y;
    if ((ki & 0x80000000) == 0) { sbits -= 1009ull << 52; scale = asf64(sbits); y = 0x1p1009 * (scale + scale * tmp); return y; }
    sbits += 1022ull << 52; scale = asf64(sbits);
    double st = scale * tmp;
    y = scale + st;
    if (y < 1.0) { double hi, lo; lo = scale - y + st; hi = 1.0 + y; lo = 1.0 - hi + y + lo; y = (hi + lo) - 1.0; if (y == 0.0) y = 0.0; }
    return 0x1p-1022 * y;
  }
  double scale = asf64(sbits);
  return F(scale, tmp, scale);
}
