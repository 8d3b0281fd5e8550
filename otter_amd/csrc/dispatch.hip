// dispatch.hip — the per-region dispatcher of `otter assemble` as host code of the library.
//
// Reference: assemble() / assemble_process() (src/assemble.cpp:39-179) with BS::thread_pool::parallelize_loop
// (src/BS_thread_pool.hpp:175-200): the BED list is cut into contiguous blocks, one per worker thread; every worker opens its own BAM /
// FASTA handle, walks its block region by region (ingest -> the five hot-path calls -> emit under a mutex).
//
// Here a worker is a GPU.  The BED list is cut into one contiguous shard per device (the same static split), and every shard is cut into
// bounded BATCHES of regions that flow through three stages running concurrently on host threads:
//     ingest (otg_ingest_regions on T host threads, + reference flanks with -r)          -> queue (depth 2)
//     hot path (otg_assemble_submit / run / collect; two contexts per device, so the upload of batch k+1 overlaps the kernels of batch k)
//     emit (otg_emit_alleles / otg_emit_reads) + the caller's write callback, strictly in BED order
// Memory is bounded by the batch size, not by the BED file; output order is the BED order whatever the number of devices or batches
// (the reference's order with -t 1; with -t > 1 the reference prints in completion order).
#include "otg_common.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace {

// The two large buffers of a batch — read sequences and read table — are plain uninitialised memory: a std::vector would zero-fill its
// ~200 KB per region on the ingest thread before the readers overwrite it (measured: a fresh batch of 1 000 regions was ingested in 66 ms,
// a recycled one in 32), while pages touched first by the readers themselves are mapped by 16 threads side by side.
template <class T> struct HostBuf {
  T* p = nullptr; size_t n = 0;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { free(p); }
  T* data() { return p; }
  size_t size() const { return n; }
  // at least `want` elements; the first `keep` stay what they were, the rest is undefined
  void resize(size_t want, size_t keep = 0) {
    if (want <= n) return;
    T* q = (T*)malloc(want * sizeof(T));
    if (!q) throw std::bad_alloc();
    if (keep && p) memcpy(q, p, std::min(keep, n) * sizeof(T));
    free(p);
    p = q; n = want;
  }
};

struct Batch {
  uint32_t index = 0;                   // batch number inside its shard
  uint32_t first = 0, n = 0;            // BED range [first, first + n)
  HostBuf<uint8_t> arena;
  HostBuf<otg_read> reads;
  std::vector<otg_region> regions;
  std::vector<otg_read_meta> meta;      // --reads-only
  std::vector<char> names;
  uint64_t arena_used = 0;
  uint32_t n_reads = 0;
  std::string text;                     // emitted records
};

template <class T>
class BoundedQueue {
 public:
  explicit BoundedQueue(size_t cap) : cap_(cap) {}
  bool push(T v) {
    std::unique_lock<std::mutex> lk(m_);
    cv_space_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
    if (closed_) return false;
    q_.push_back(std::move(v));
    cv_item_.notify_one();
    return true;
  }
  bool pop(T& out) {
    std::unique_lock<std::mutex> lk(m_);
    cv_item_.wait(lk, [&] { return !q_.empty() || done_ || closed_; });
    if (closed_ || q_.empty()) return false;
    out = std::move(q_.front());
    q_.pop_front();
    cv_space_.notify_one();
    return true;
  }
  void finish() { std::lock_guard<std::mutex> lk(m_); done_ = true; cv_item_.notify_all(); }           // no more items will come
  void abort() { std::lock_guard<std::mutex> lk(m_); closed_ = true; cv_item_.notify_all(); cv_space_.notify_all(); }
 private:
  std::mutex m_;
  std::condition_variable cv_item_, cv_space_;
  std::deque<T> q_;
  size_t cap_;
  bool done_ = false, closed_ = false;
};

using BatchPtr = std::unique_ptr<Batch>;
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
// OTG_DISPATCH_TRACE=1: one stderr line per stage of every batch (milliseconds since the job started) — the host-side timeline of a job
const bool g_trace = getenv("OTG_DISPATCH_TRACE") != nullptr;
Clock::time_point g_trace_t0;
void trace(const char* what, uint32_t batch, uint32_t n, Clock::time_point from)
{
  if (!g_trace) return;
  fprintf(stderr, "[otg trace] %-8s batch %3u (%5u regions) %9.2f -> %9.2f ms\n", what, batch, n, std::chrono::duration<double, std::milli>(from - g_trace_t0).count(), ms_since(g_trace_t0));
}

struct Job {
  const otg_assemble_job* j = nullptr;
  std::vector<otg_bed> beds;
  std::vector<char> chr_arena;
  otg_bam* bam = nullptr;
  otg_fasta* fasta = nullptr;
  std::atomic<int> rc{OTG_OK};
  std::mutex err_m;
  std::string err;
  // statistics (summed over threads)
  std::mutex st_m;
  otg_job_stats st{};
  void fail(int code, const std::string& what) {
    int expected = OTG_OK;
    if (rc.compare_exchange_strong(expected, code)) { std::lock_guard<std::mutex> lk(err_m); err = what; }
  }
};

// Contexts survive the job: a fresh context pays for its first launches (page mapping of newly allocated workspaces, ~0.2-1 s), so a process
// that runs several jobs — or bench.py's repeated file-to-text leg — keeps them in a pool; otg_assemble_files_release() empties it.
std::mutex g_pool_m;
std::vector<std::pair<int, otg_ctx*>> g_pool;
otg_ctx* pool_acquire(int device)
{
  {
    std::lock_guard<std::mutex> lk(g_pool_m);
    for (size_t i = 0; i < g_pool.size(); ++i) if (g_pool[i].first == device) { otg_ctx* c = g_pool[i].second; g_pool.erase(g_pool.begin() + (long)i); return c; }
  }
  otg_ctx* c = nullptr;
  return otg_create(device, &c) == OTG_OK ? c : nullptr;
}
void pool_release(int device, otg_ctx* c) { std::lock_guard<std::mutex> lk(g_pool_m); g_pool.emplace_back(device, c); }

int dispatch_contexts()
{
  const char* e = getenv("OTG_DISPATCH_CONTEXTS");
  const int v = e ? atoi(e) : 2;
  return v < 1 ? 1 : (v > 4 ? 4 : v);
}
// End of a job: the pool keeps what ONE shard per device needs for the next job and destroys the rest — contexts hold multi-gigabyte aligner
// workspaces, and a caller that creates its own otg_ctx afterwards is budgeted against what is left of the device.
void pool_trim(const std::vector<int>& devs)
{
  std::lock_guard<std::mutex> lk(g_pool_m);
  const int keep = dispatch_contexts();
  std::map<int, int> seen;
  std::vector<std::pair<int, otg_ctx*>> kept;
  for (auto& p : g_pool) {
    if (++seen[p.first] <= keep) kept.push_back(p);
    else otg_destroy(p.second);
  }
  (void)devs;
  g_pool.swap(kept);
}

std::string last_err() { const char* e = otg_last_error(nullptr); return e ? std::string(e) : std::string(); }

// ---- stage 1: one batch of regions from the BAM (and the FASTA flanks with -r), buffers grown on OTG_ERR_CAPACITY
int ingest_batch(Job& J, Batch& b, int threads)
{
  const otg_assemble_job& j = *J.j;
  otg_ingest_opts o = j.ingest;
  o.threads = threads;
  b.regions.assign(b.n, otg_region{});
  size_t cap_reads = std::max<size_t>(b.reads.size(), (size_t)b.n * 48 + 256), cap_arena = std::max<size_t>(b.arena.size(), (size_t)b.n * 48 * 4096 + 4096);
  size_t cap_names = std::max<size_t>(b.names.size(), j.reads_only ? (size_t)b.n * 48 * 48 : 0);
  for (int attempt = 0; attempt < 4; ++attempt) {
    b.reads.resize(cap_reads); b.arena.resize(cap_arena);
    if (j.reads_only) { b.meta.resize(cap_reads); b.names.resize(cap_names); }
    uint64_t used = 0, nused = 0; uint32_t nr = 0;
    const int rc = j.reads_only
        ? otg_ingest_regions_named(J.bam, J.beds.data() + b.first, J.chr_arena.data(), b.n, &o, b.arena.data(), b.arena.size(), &used, b.reads.data(),
                                   (uint32_t)b.reads.size(), &nr, b.regions.data(), b.meta.data(), b.names.data(), b.names.size(), &nused)
        : otg_ingest_regions(J.bam, J.beds.data() + b.first, J.chr_arena.data(), b.n, &o, b.arena.data(), b.arena.size(), &used, b.reads.data(),
                             (uint32_t)b.reads.size(), &nr, b.regions.data());
    if (rc == OTG_ERR_CAPACITY) {      // the counters hold the needed totals
      cap_reads = (size_t)nr + 256; cap_arena = (size_t)used + 4096 + (J.fasta ? (size_t)b.n * 2 * ((size_t)j.params.flank + 8) : 0); cap_names = (size_t)nused + 256;
      continue;
    }
    if (rc != OTG_OK) return rc;
    b.arena_used = used; b.n_reads = nr;
    if (J.fasta) {
      const size_t need = used + (size_t)b.n * 2 * ((size_t)j.params.flank + 8) + 128;
      if (b.arena.size() < need) b.arena.resize(need, (size_t)used);
      uint64_t u2 = used;
      const int rf = otg_fasta_region_flanks(J.fasta, J.beds.data() + b.first, J.chr_arena.data(), b.n, o.offset_l, o.offset_r, j.params.flank, b.arena.data(), b.arena.size(),
                                             &u2, b.regions.data());
      if (rf != OTG_OK) return rf;
      b.arena_used = u2;
    }
    return OTG_OK;
  }
  return OTG_ERR_CAPACITY;
}

// ---- stage 2 + 3 of one batch on one context: hot path, then the record text
int run_batch(Job& J, otg_ctx* ctx, Batch& b, std::vector<otg_region_result>& rr, std::vector<otg_allele>& al, std::vector<uint8_t>& seqs, double* ms_gpu, double* ms_emit)
{
  const otg_assemble_job& j = *J.j;
  otg_params P = j.params;
  P.realign = J.fasta ? 1 : 0;
  const char* rg = j.read_group ? j.read_group : "";
  auto t0 = Clock::now();
  uint64_t need = 0;
  if (j.reads_only) {
    // --reads-only: the reads of each region; with -r they are printed after local_realignment trimmed them (src/assemble.cpp:72-89)
    if (J.fasta && b.n_reads) {
      int rc = otg_assemble_submit(ctx, &P, b.arena.data(), b.arena_used, b.reads.data(), b.n_reads, b.regions.data(), b.n);
      if (rc == OTG_OK) rc = otg_assemble_realign(ctx);
      if (rc == OTG_OK) rc = otg_assemble_collect_reads(ctx, b.reads.data(), b.n_reads);
      if (rc != OTG_OK) return rc;
    }
    *ms_gpu += ms_since(t0);
    t0 = Clock::now();
    int rc = otg_emit_reads(J.beds.data() + b.first, J.chr_arena.data(), b.n, b.regions.data(), b.reads.data(), b.arena.data(), b.meta.data(), b.names.data(), rg, j.is_fasta,
                            P.max_cov, nullptr, 0, &need);
    if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) return rc;
    b.text.resize(need);
    rc = otg_emit_reads(J.beds.data() + b.first, J.chr_arena.data(), b.n, b.regions.data(), b.reads.data(), b.arena.data(), b.meta.data(), b.names.data(), rg, j.is_fasta,
                        P.max_cov, b.text.empty() ? nullptr : &b.text[0], b.text.size(), &need);
    *ms_emit += ms_since(t0);
    return rc;
  }
  int rc = otg_assemble_submit(ctx, &P, b.arena.data(), b.arena_used, b.reads.data(), b.n_reads, b.regions.data(), b.n);
  trace("submit", b.index, b.n, t0);
  auto t1 = Clock::now();
  if (rc == OTG_OK) rc = otg_assemble_run(ctx);
  trace("run", b.index, b.n, t1);
  t1 = Clock::now();
  uint32_t na = 0; uint64_t sb = 0;
  if (rc == OTG_OK) rc = otg_assemble_result_sizes(ctx, &na, &sb);
  if (rc != OTG_OK) return rc;
  rr.resize(b.n); al.resize((size_t)na + 1); seqs.resize((size_t)sb + 64);
  rc = otg_assemble_collect(ctx, rr.data(), al.data(), (uint32_t)al.size(), seqs.data(), seqs.size(), nullptr);
  if (rc != OTG_OK) return rc;
  trace("collect", b.index, b.n, t1);
  *ms_gpu += ms_since(t0);
  t0 = Clock::now();
  rc = otg_emit_alleles(J.beds.data() + b.first, J.chr_arena.data(), b.n, rr.data(), al.data(), seqs.data(), rg, j.is_fasta, nullptr, 0, &need);
  if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) return rc;
  b.text.resize(need);
  rc = otg_emit_alleles(J.beds.data() + b.first, J.chr_arena.data(), b.n, rr.data(), al.data(), seqs.data(), rg, j.is_fasta, b.text.empty() ? nullptr : &b.text[0], b.text.size(), &need);
  *ms_emit += ms_since(t0);
  trace("emit", b.index, b.n, t0);
  {
    std::lock_guard<std::mutex> lk(J.st_m);
    J.st.n_alleles += na;
    for (uint32_t r = 0; r < b.n; ++r) { if (rr[r].n_alleles) ++J.st.n_regions_ok; if (rr[r].status == OTG_REGION_SKIP_MAXCOV) ++J.st.n_regions_skipped; }
  }
  return rc;
}

// ---- one device's shard [a, b): ingest thread -> two hot-path threads -> ordered text
struct ShardOut {
  std::mutex m;
  std::condition_variable cv;
  std::map<uint32_t, std::string> ready;    // batch index -> text
  uint32_t n_batches = 0;
  uint32_t next = 0;                        // the batch the writer takes next
  size_t cap = 3;                           // finished batches a shard may hold back (the writer drains the shards one after the other)
};

// Batch objects cycle between the ingest thread and the hot-path threads of a shard: their vectors keep the capacity of the batches they
// have carried (no 400 MB zero-fill per batch; at most 2 x contexts + 2 objects exist per shard).
struct BatchPool {
  std::mutex m;
  std::vector<std::unique_ptr<Batch>> free_;
  void clear() { std::lock_guard<std::mutex> lk(m); free_.clear(); }
  std::unique_ptr<Batch> get() {
    std::lock_guard<std::mutex> lk(m);
    if (free_.empty()) return std::unique_ptr<Batch>(new Batch());
    std::unique_ptr<Batch> b = std::move(free_.back());
    free_.pop_back();
    return b;
  }
  void put(std::unique_ptr<Batch> b) { b->text.clear(); std::lock_guard<std::mutex> lk(m); free_.push_back(std::move(b)); }
};

// How a shard [a, b) is cut into batches.  A size given by the caller is kept as it is.  Left to the library (batch_regions == 0): the device
// idles while the FIRST batch is being read, so the shard starts small (256, 512, 1024 regions); then full batches of 2048 — a batch costs
// ~35 ms + 64 ms per 1 000 regions, its longest alignment and its largest graph only overlap with the NEXT batch's body —; and the last
// stretch goes to the two contexts of the device as two equal halves, so that they run out of work together (ending on ever smaller batches
// instead — thirds of what is left — cost 5 % of a 10 000-region job: six small batches, each with the fixed cost of a batch).
std::vector<std::pair<uint32_t, uint32_t>> batch_plan(uint32_t a, uint32_t b, uint32_t requested)
{
  std::vector<std::pair<uint32_t, uint32_t>> plan;
  if (a >= b) return plan;
  if (requested) { for (uint32_t f = a; f < b; f += requested) plan.emplace_back(f, std::min(requested, b - f)); return plan; }
  const uint32_t per = 2048u;      // (r04, measured again on 10 000 loci: 1024 -> 10 400 regions/s, 2048 -> 11 500, 3072 -> 11 500, 4096 -> 11 300;
                                   //  on 60 000 loci: 2048 -> 12 270, 4096 -> 12 200, 8192 -> 12 020 — larger batches lower the hot path's busy time, not the wall)
  uint32_t f = a, rem = b - a;
  auto take = [&](uint32_t n) { plan.emplace_back(f, n); f += n; rem -= n; };
  for (uint32_t r = per / 8; r < per; r *= 2) if (rem > 4 * r) take(r);
  while (rem) {
    if (rem > per + per / 4) take(per);
    else if (rem > per / 2) { const uint32_t h = (rem + 1) / 2; take(h); take(rem); }
    else take(rem);
  }
  return plan;
}

// Like the contexts, batch objects outlive the job (their buffers are a few hundred megabytes each: unmapping them at the end of a job and
// mapping them again at the start of the next costs tens of milliseconds per object); otg_assemble_files_release() frees them.
BatchPool g_batches;

void shard_worker(Job& J, int device, uint32_t a, uint32_t bnd, int ingest_threads, ShardOut& out)
{
  if (a >= bnd) return;                 // more devices than regions: nothing to do here (and no context to create)
  const std::vector<std::pair<uint32_t, uint32_t>> plan = batch_plan(a, bnd, J.j->batch_regions);
  // hot-path threads (one context each) per device: a batch of a few hundred regions cannot fill the device — its stages wait for their
  // longest alignment / graph — so several batches are in flight; OTG_DISPATCH_CONTEXTS overrides (1..4)
  const int n_gpu_threads = dispatch_contexts();
  BoundedQueue<BatchPtr> q_in((size_t)n_gpu_threads);
  BatchPool& recycled = g_batches;
  { std::lock_guard<std::mutex> lk(out.m); out.cap = (size_t)n_gpu_threads + 1; }
  double ms_ingest = 0, ms_gpu[4] = {0, 0, 0, 0}, ms_emit[4] = {0, 0, 0, 0};
  std::thread ingest([&] {
    try {
      for (uint32_t idx = 0; idx < (uint32_t)plan.size() && J.rc.load() == OTG_OK; ++idx) {
        BatchPtr b = recycled.get();
        b->index = idx; b->first = plan[idx].first; b->n = plan[idx].second;
        const auto t0 = Clock::now();
        const int rc = ingest_batch(J, *b, ingest_threads);
        ms_ingest += ms_since(t0);
        trace("ingest", idx, b->n, t0);
        if (rc != OTG_OK) { J.fail(rc, "ingest: " + last_err()); break; }
        { std::lock_guard<std::mutex> lk(J.st_m); J.st.n_reads += b->n_reads; J.st.input_bytes += b->arena_used; }
        if (!q_in.push(std::move(b))) break;
      }
    } catch (const std::exception& e) { J.fail(OTG_ERR_ARG, std::string("ingest: ") + e.what()); }
    q_in.finish();
  });
  auto gpu_thread = [&](int slot) {
    otg_ctx* ctx = nullptr;
    const bool need_gpu = !J.j->reads_only || J.fasta;
    if (need_gpu && !(ctx = pool_acquire(device))) { J.fail(OTG_ERR_NO_DEVICE, "otg_create: " + last_err()); q_in.abort(); return; }
    std::vector<otg_region_result> rr; std::vector<otg_allele> al; std::vector<uint8_t> seqs;
    try {
      BatchPtr b;
      while (J.rc.load() == OTG_OK && q_in.pop(b)) {
        const int rc = run_batch(J, ctx, *b, rr, al, seqs, &ms_gpu[slot], &ms_emit[slot]);
        if (rc != OTG_OK) { J.fail(rc, "hot path: " + (ctx && otg_last_error(ctx) ? std::string(otg_last_error(ctx)) : last_err())); q_in.abort(); break; }
        {
          // Back-pressure: the writer drains the shards strictly in order, so a shard it has not reached yet may hold back `cap` finished
          // batches and no more (host memory stays bounded by the batch size, not by the shard).  The batch the writer wants next always
          // gets in — the threads of a shard finish out of order, and that batch may be the last one to arrive.
          std::unique_lock<std::mutex> lk(out.m);
          while (!(out.ready.size() < out.cap || b->index == out.next || J.rc.load() != OTG_OK)) out.cv.wait_for(lk, std::chrono::milliseconds(50));
          out.ready.emplace(b->index, std::move(b->text));
        }
        out.cv.notify_all();
        recycled.put(std::move(b));
      }
    } catch (const std::exception& e) { J.fail(OTG_ERR_ARG, std::string("hot path: ") + e.what()); }
    if (J.rc.load() != OTG_OK) q_in.abort();               // whatever stopped the job: release the ingest thread
    if (ctx) pool_release(device, ctx);
  };
  std::vector<std::thread> gts;
  for (int t = 0; t < n_gpu_threads; ++t) gts.emplace_back(gpu_thread, t);
  ingest.join();
  for (auto& t : gts) t.join();
  out.cv.notify_all();
  std::lock_guard<std::mutex> lk(J.st_m);
  J.st.ms_ingest += ms_ingest;
  for (int t = 0; t < n_gpu_threads; ++t) { J.st.ms_hot_path += ms_gpu[t]; J.st.ms_emit += ms_emit[t]; }
}

} // namespace

extern "C" {

int otg_assemble_files(const otg_assemble_job* job, otg_write_fn write, void* user, otg_job_stats* stats)
{
  if (!job || !write || !job->bam_path || !job->bed_path) return otg_fail(nullptr, OTG_ERR_ARG, "otg_assemble_files: NULL job, writer, BAM or BED path");
  if (job->n_devices < 0 || (job->n_devices > 0 && !job->devices)) return otg_fail(nullptr, OTG_ERR_ARG, "otg_assemble_files: bad device list");
  const auto t_all = Clock::now();
  g_trace_t0 = t_all;
  Job J;
  J.j = job;
  // BED file (size protocol: first call reports the needed sizes)
  {
    uint32_t n = 0, skipped = 0; uint64_t cu = 0;
    int rc = otg_parse_bed_file(job->bed_path, nullptr, 0, &n, nullptr, 0, &cu, &skipped);
    if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) return rc;
    J.beds.resize((size_t)n + 1); J.chr_arena.resize((size_t)cu + 16);
    rc = otg_parse_bed_file(job->bed_path, J.beds.data(), (uint32_t)J.beds.size(), &n, J.chr_arena.data(), J.chr_arena.size(), &cu, &skipped);
    if (rc != OTG_OK) return rc;
    J.beds.resize(n);
    J.st.n_regions = n;
  }
  int rc = otg_bam_open(job->bam_path, &J.bam);
  if (rc != OTG_OK) return rc;
  if (job->fasta_path && job->fasta_path[0]) {
    rc = otg_fasta_open(job->fasta_path, &J.fasta);
    if (rc != OTG_OK) { otg_bam_close(J.bam); return rc; }
  }
  auto cleanup = [&] { if (J.fasta) otg_fasta_close(J.fasta); otg_bam_close(J.bam); };
  // SAM header (src/assemble.cpp:167-177); FASTA output has none
  if (!job->is_fasta) {
    const uint32_t nt = otg_bam_n_targets(J.bam);
    std::string names; std::vector<uint64_t> off(nt), len(nt); std::vector<uint32_t> nl(nt);
    for (uint32_t i = 0; i < nt; ++i) { uint64_t l = 0; const char* nm = otg_bam_target(J.bam, i, &l); off[i] = names.size(); nl[i] = (uint32_t)strlen(nm); len[i] = l; names += nm; }
    uint64_t need = 0;
    rc = otg_emit_sam_header(names.data(), off.data(), nl.data(), len.data(), nt, job->read_group ? job->read_group : "", job->ingest.offset_l, job->ingest.offset_r, nullptr, 0, &need);
    if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) { cleanup(); return rc; }
    std::string hdr(need, '\0');
    rc = otg_emit_sam_header(names.data(), off.data(), nl.data(), len.data(), nt, job->read_group ? job->read_group : "", job->ingest.offset_l, job->ingest.offset_r, need ? &hdr[0] : nullptr, need, &need);
    if (rc != OTG_OK) { cleanup(); return rc; }
    if (write(user, hdr.data(), hdr.size()) != 0) { cleanup(); return otg_fail(nullptr, OTG_ERR_ARG, "otg_assemble_files: the writer failed"); }
    J.st.output_bytes += hdr.size();
  }
  // static contiguous split over the devices (BS::thread_pool::parallelize_loop: block = total / workers, the last takes the remainder)
  std::vector<int> devs;
  if (job->n_devices > 0) devs.assign(job->devices, job->devices + job->n_devices); else devs.push_back(0);
  const uint32_t R = (uint32_t)J.beds.size(), W = (uint32_t)devs.size();
  const uint32_t block = R / W;
  const int threads_total = job->ingest.threads > 0 ? job->ingest.threads : 1;
  const int threads_per = std::max(1, threads_total / (int)W);
  std::vector<std::unique_ptr<ShardOut>> outs;
  std::vector<std::thread> workers;
  std::vector<std::pair<uint32_t, uint32_t>> bounds;
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t a = block == 0 ? std::min(w, R) : w * block;
    const uint32_t b = block == 0 ? std::min(w + 1, R) : (w == W - 1 ? R : a + block);
    bounds.emplace_back(a, b);
    outs.emplace_back(new ShardOut());
    outs.back()->n_batches = (uint32_t)batch_plan(a, b, job->batch_regions).size();
  }
  for (uint32_t w = 0; w < W; ++w) workers.emplace_back(shard_worker, std::ref(J), devs[w], bounds[w].first, bounds[w].second, threads_per, std::ref(*outs[w]));
  // the writer: shards in order, batches in order
  for (uint32_t w = 0; w < W; ++w) {
    ShardOut& o = *outs[w];
    for (uint32_t k = 0; k < o.n_batches; ++k) {
      std::string text;
      {
        std::unique_lock<std::mutex> lk(o.m);
        o.cv.wait_for(lk, std::chrono::milliseconds(50), [&] { return o.ready.count(k) || J.rc.load() != OTG_OK; });
        while (!o.ready.count(k) && J.rc.load() == OTG_OK) o.cv.wait_for(lk, std::chrono::milliseconds(50));
        if (!o.ready.count(k)) break;
        text = std::move(o.ready[k]);
        o.ready.erase(k);
        o.next = k + 1;
      }
      o.cv.notify_all();                    // room for the shard's hot-path threads
      const auto tw = Clock::now();
      if (!text.empty() && write(user, text.data(), text.size()) != 0) { J.fail(OTG_ERR_ARG, "the writer failed"); break; }
      trace("write", k, (uint32_t)(text.size() >> 10), tw);
      J.st.output_bytes += text.size();
    }
    if (J.rc.load() != OTG_OK) break;
  }
  for (auto& t : workers) t.join();
  cleanup();
  pool_trim(devs);
  trace("job", 0, R, t_all);
  J.st.ms_total = ms_since(t_all);
  J.st.n_devices = W;
  if (stats) *stats = J.st;
  if (J.rc.load() != OTG_OK) return otg_fail(nullptr, J.rc.load(), "otg_assemble_files: %s", J.err.c_str());
  return OTG_OK;
}

// ---- `otter genotype` from files to text in one call: genotype() / genotype_process() (src/genotype.cpp:69-192).  Regions in bounded
// batches: allele ingest (host threads) -> anallele_cluster on the device -> VCF lines (or, without a reference, the two-length table),
// text in BED order.  The reference's worker loop does the same region by region on a thread pool and prints under a mutex.
int otg_genotype_files(const otg_genotype_job* job, otg_write_fn write, void* user, otg_job_stats* stats)
{
  if (!job || !write || !job->bam_path || !job->bed_path) return otg_fail(nullptr, OTG_ERR_ARG, "otg_genotype_files: NULL job, writer, BAM or BED path");
  const auto t_all = Clock::now();
  otg_job_stats st{};
  std::vector<otg_bed> beds; std::vector<char> chr_arena;
  {
    uint32_t n = 0, skipped = 0; uint64_t cu = 0;
    int rc = otg_parse_bed_file(job->bed_path, nullptr, 0, &n, nullptr, 0, &cu, &skipped);
    if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) return rc;
    beds.resize((size_t)n + 1); chr_arena.resize((size_t)cu + 16);
    rc = otg_parse_bed_file(job->bed_path, beds.data(), (uint32_t)beds.size(), &n, chr_arena.data(), chr_arena.size(), &cu, &skipped);
    if (rc != OTG_OK) return rc;
    beds.resize(n);
    st.n_regions = n;
  }
  otg_bam* bam = nullptr; otg_fasta* fasta = nullptr; otg_ctx* ctx = nullptr;
  int rc = otg_bam_open(job->bam_path, &bam);
  if (rc != OTG_OK) return rc;
  auto cleanup = [&] { if (ctx) pool_release(job->device, ctx); if (fasta) otg_fasta_close(fasta); otg_bam_close(bam); };
  uint32_t n_samples = 0; int32_t ol = 0, orr = 0;
  rc = otg_bam_sample_index(bam, &n_samples, &ol, &orr);                      // SampleIndex::init (src/anbamdb.cpp:42-63)
  if (rc != OTG_OK) { cleanup(); return rc; }
  const bool with_ref = job->fasta_path && job->fasta_path[0];
  if (with_ref) {
    rc = otg_fasta_open(job->fasta_path, &fasta);
    if (rc != OTG_OK) { cleanup(); return rc; }
    if (!(ctx = pool_acquire(job->device))) { cleanup(); return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_genotype_files: %s", last_err().c_str()); }
    uint64_t need = 0;
    rc = otg_emit_vcf_header(bam, nullptr, 0, &need);                        // output_vcf_header (src/genotype.cpp:16-40)
    if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) { cleanup(); return rc; }
    std::string hdr(need, '\0');
    rc = otg_emit_vcf_header(bam, need ? &hdr[0] : nullptr, need, &need);
    if (rc != OTG_OK) { cleanup(); return rc; }
    if (write(user, hdr.data(), hdr.size()) != 0) { cleanup(); return otg_fail(nullptr, OTG_ERR_ARG, "otg_genotype_files: the writer failed"); }
    st.output_bytes += hdr.size();
  }
  const uint32_t per = job->batch_regions ? job->batch_regions : 1024u;
  const int threads = job->threads > 0 ? job->threads : 1;
  // Two batches in flight: while one is clustered on the GPU and formatted, the next is ingested (the reference does all three one region at a
  // time, src/genotype.cpp:80-157; SURVEY finding 9: this command is BAM-I/O-bound, so ingest is what must never wait).  The VCF text of a batch is
  // formatted by `threads` host threads, each a contiguous slice of its regions, concatenated in order.
  struct GtBatch {
    std::vector<uint8_t> arena; std::vector<otg_allele> alleles; std::vector<uint32_t> first;
    uint32_t f = 0, n = 0, na = 0; uint64_t used = 0; int rc = OTG_OK; double ms = 0; std::string err;
  };
  GtBatch bufs[2];
  auto ingest_into = [&](GtBatch& B, uint32_t f) {
    B.f = f; B.n = std::min<uint32_t>(per, (uint32_t)beds.size() - f);
    B.first.assign((size_t)B.n + 1, 0);
    size_t cap_al = std::max<size_t>(B.alleles.size(), (size_t)B.n * 128 + 64), cap_ar = std::max<size_t>(B.arena.size(), (size_t)B.n * 128 * 4096 + 4096);
    const auto t0 = Clock::now();
    for (int attempt = 0; attempt < 4; ++attempt) {
      B.alleles.resize(cap_al); B.arena.resize(cap_ar);
      B.na = 0; B.used = 0;
      B.rc = otg_ingest_alleles(bam, beds.data() + f, chr_arena.data(), B.n, threads, fasta, B.arena.data(), B.arena.size(), &B.used, B.alleles.data(), (uint32_t)B.alleles.size(), &B.na, B.first.data());
      if (B.rc != OTG_ERR_CAPACITY) break;
      cap_al = (size_t)B.na + 64; cap_ar = (size_t)B.used + 4096;
    }
    if (B.rc != OTG_OK) B.err = last_err();
    B.ms = ms_since(t0);
  };
  std::vector<uint64_t> seq_off; std::vector<uint32_t> seq_len, n_al;
  std::vector<int32_t> gt, gtl, gtk, ngt, reps; std::vector<double> hsd;
  std::string text;
  std::vector<std::string> parts;
  const uint32_t n_beds = (uint32_t)beds.size();
  if (n_beds) ingest_into(bufs[0], 0);
  for (uint32_t f = 0, idx = 0; f < n_beds; f += per, ++idx) {
    GtBatch& B = bufs[idx & 1];
    if (B.rc != OTG_OK) { rc = B.rc; const std::string e = B.err; cleanup(); return otg_fail(nullptr, rc, "otg_genotype_files: %s", e.c_str()); }
    std::thread next;
    const bool overlap = with_ref && f + per < n_beds;            // (the two-length table without -r reads the BAM handle's sample list while it formats: kept in sequence)
    if (overlap) next = std::thread([&, f, idx] { ingest_into(bufs[(idx + 1) & 1], f + per); });
    auto join_next = [&] { if (next.joinable()) next.join(); };       // before every return: the ingest thread writes into bufs[] and reads the handles cleanup() closes
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{next};
    const uint32_t n = B.n, na = B.na; const uint64_t used = B.used;
    std::vector<uint32_t>& first = B.first; std::vector<otg_allele>& alleles = B.alleles; std::vector<uint8_t>& arena = B.arena;
    st.ms_ingest += B.ms; st.n_reads += na; st.input_bytes += used;
    uint64_t need = 0;
    auto t0 = Clock::now();
    if (with_ref) {
      seq_off.resize(na); seq_len.resize(na); n_al.resize(n);
      for (uint32_t i = 0; i < na; ++i) { seq_off[i] = alleles[i].seq_off; seq_len[i] = alleles[i].seq_len; }
      for (uint32_t r = 0; r < n; ++r) n_al[r] = first[r + 1] - first[r];
      gt.resize((size_t)na + 1); gtl.resize((size_t)na + 1); gtk.resize((size_t)na + 1); reps.resize((size_t)na + 1); hsd.resize((size_t)na + 1); ngt.resize((size_t)n + 1);
      if (arena.size() < used + 64) arena.resize(used + 64);
      if (na) {
        rc = otg_genotype_cluster_batch(ctx, &job->params, arena.data(), used + 64, seq_off.data(), seq_len.data(), first.data(), n_al.data(), n,
                                        gt.data(), gtl.data(), gtk.data(), hsd.data(), ngt.data(), reps.data());      // anallele_cluster (src/genotype.cpp:138)
        if (rc != OTG_OK) { const std::string e = otg_last_error(ctx) ? otg_last_error(ctx) : ""; join_next(); cleanup(); return otg_fail(nullptr, rc, "otg_genotype_files: %s", e.c_str()); }
      } else std::fill(ngt.begin(), ngt.end(), 0);
      st.ms_hot_path += ms_since(t0);
      t0 = Clock::now();
      // VCF lines: slices of the batch's regions on host threads (output_vcf_line, src/genotype.cpp:43-78, is a pure function of its region)
      const uint32_t nslice = (uint32_t)std::max(1, std::min<int>(threads, (int)((n + 31) / 32)));
      parts.assign(nslice, std::string());
      std::vector<int> prc(nslice, OTG_OK);
      std::vector<std::string> perr(nslice);
      auto emit_slice = [&](uint32_t sidx) {
        const uint32_t a = (uint32_t)((uint64_t)n * sidx / nslice), e = (uint32_t)((uint64_t)n * (sidx + 1) / nslice);
        std::string& out = parts[sidx];
        uint64_t bytes = 0;
        for (uint32_t r = a; r < e; ++r) bytes += 512 + 80ull * (n_samples + 1);
        for (uint32_t i = first[a]; i < first[e]; ++i) bytes += alleles[i].seq_len + 8;          // every allele sequence could be an ALT
        out.resize(bytes);
        uint64_t len = 0;
        int r2 = otg_emit_vcf_lines(beds.data() + f + a, chr_arena.data(), e - a, first.data() + a, alleles.data(), arena.data(), n_samples, gt.data(), hsd.data(), ngt.data() + a,
                                    reps.data(), ol, orr, bytes ? &out[0] : nullptr, bytes, &len);
        if (r2 == OTG_ERR_CAPACITY) {         // (the estimate above is an upper bound; kept as a safety net)
          out.resize(len);
          r2 = otg_emit_vcf_lines(beds.data() + f + a, chr_arena.data(), e - a, first.data() + a, alleles.data(), arena.data(), n_samples, gt.data(), hsd.data(), ngt.data() + a,
                                  reps.data(), ol, orr, len ? &out[0] : nullptr, len, &len);
        }
        if (r2 != OTG_OK) perr[sidx] = last_err();
        prc[sidx] = r2;
        out.resize(r2 == OTG_OK ? len : 0);
      };
      if (nslice == 1) emit_slice(0);
      else {
        std::vector<std::thread> th;
        for (uint32_t sidx = 0; sidx < nslice; ++sidx) th.emplace_back(emit_slice, sidx);
        for (auto& t : th) t.join();
      }
      text.clear();
      for (uint32_t sidx = 0; sidx < nslice; ++sidx) {
        if (prc[sidx] != OTG_OK) { rc = prc[sidx]; const std::string e = perr[sidx]; join_next(); cleanup(); return otg_fail(nullptr, rc, "otg_genotype_files: %s", e.c_str()); }
        text += parts[sidx];
      }
      rc = OTG_OK;
    } else {
      rc = otg_emit_genotype_lengths(bam, beds.data() + f, chr_arena.data(), n, first.data(), alleles.data(), n_samples, nullptr, 0, &need);   // src/genotype.cpp:112-121
      if (rc != OTG_OK && rc != OTG_ERR_CAPACITY) { cleanup(); return rc; }
      text.resize(need);
      rc = otg_emit_genotype_lengths(bam, beds.data() + f, chr_arena.data(), n, first.data(), alleles.data(), n_samples, need ? &text[0] : nullptr, need, &need);
    }
    if (rc != OTG_OK) { join_next(); cleanup(); return rc; }
    st.ms_emit += ms_since(t0);
    for (uint32_t r = 0; r < n; ++r) { if (first[r + 1] > first[r]) ++st.n_regions_ok; }
    st.n_alleles += na;
    if (!text.empty() && write(user, text.data(), text.size()) != 0) { join_next(); cleanup(); return otg_fail(nullptr, OTG_ERR_ARG, "otg_genotype_files: the writer failed"); }
    st.output_bytes += text.size();
    if (next.joinable()) next.join();
    if (!overlap && f + per < n_beds) ingest_into(bufs[(idx + 1) & 1], f + per);
  }
  cleanup();
  st.ms_total = ms_since(t_all); st.n_devices = 1;
  if (stats) *stats = st;
  return OTG_OK;
}

int otg_assemble_batch_plan(uint32_t n_regions, uint32_t batch_regions, uint32_t* sizes, uint32_t capacity, uint32_t* n_batches)
{
  if (!n_batches || (capacity && !sizes)) return otg_fail(nullptr, OTG_ERR_ARG, "otg_assemble_batch_plan: NULL argument");
  const std::vector<std::pair<uint32_t, uint32_t>> plan = batch_plan(0, n_regions, batch_regions);
  *n_batches = (uint32_t)plan.size();
  if (plan.size() > capacity) return OTG_ERR_CAPACITY;
  for (size_t i = 0; i < plan.size(); ++i) sizes[i] = plan[i].second;
  return OTG_OK;
}

void otg_assemble_files_release(void)
{
  std::lock_guard<std::mutex> lk(g_pool_m);
  for (auto& p : g_pool) otg_destroy(p.second);
  g_pool.clear();
  g_batches.clear();
}

} // extern "C"
