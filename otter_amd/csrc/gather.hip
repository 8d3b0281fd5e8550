// gather.hip — end-of-run gather of the per-region allele records to rank 0 over RCCL (xGMI), for hosts that run one process per GPU.
//
// north_star: "regions shard embarrassingly across the 8 GPUs of one node with a simple static BED split and an RCCL-over-xGMI gather of
// per-region allele records at the end".  The reference is one process whose worker threads print under a mutex (src/assemble.cpp:143-149);
// rank order here = BED order of the static split (BS::thread_pool::parallelize_loop, src/BS_thread_pool.hpp:183-198), so rank 0 ends up
// with the records of the whole job in BED order.  Two collectives on the library's stream, straight from the device-resident result
// buffers of the last otg_assemble_run (no host round trip on the sending ranks):
//   otg_gather_sizes    ncclAllGather of three counters per rank (regions, allele records, sequence bytes);
//   otg_gather_records  one group of ncclSend / ncclRecv (a gather with per-rank sizes), then ONE device-to-host copy on rank 0 and the
//                       rebasing of region / allele / sequence indices to job-wide ones.
// librccl is resolved at run time (dlopen) the first time a communicator is asked for: libotter_gpu.so has no link-time dependency on
// it, and a process that already carries PyTorch's RCCL uses that copy.
#include "otg_common.hpp"
#include <dlfcn.h>
#include <mutex>

namespace {

typedef int ncclResult;
typedef void* ncclCommT;
struct ncclId { char internal[128]; };
enum { NCCL_UINT8 = 1, NCCL_UINT64 = 5 };      // ncclDataType_t (rccl.h)

struct Rccl {
  void* h = nullptr;
  ncclResult (*GetUniqueId)(ncclId*) = nullptr;
  ncclResult (*CommInitRank)(ncclCommT*, int, ncclId, int) = nullptr;
  ncclResult (*CommDestroy)(ncclCommT) = nullptr;
  ncclResult (*AllGather)(const void*, void*, size_t, int, ncclCommT, hipStream_t) = nullptr;
  ncclResult (*Send)(const void*, size_t, int, int, ncclCommT, hipStream_t) = nullptr;
  ncclResult (*Recv)(void*, size_t, int, int, ncclCommT, hipStream_t) = nullptr;
  ncclResult (*GroupStart)() = nullptr;
  ncclResult (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult) = nullptr;
  std::string err;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load()
{
  std::call_once(g_rccl_once, [] {
    Rccl& R = g_rccl;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (R.h) break; }
    if (!R.h) { R.err = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p && R.err.empty()) R.err = std::string("librccl.so lacks ") + n; return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  });
  return g_rccl.h && g_rccl.err.empty();
}

#define RCCL_TRY(ctx, call)                                                                                                                   \
  do { const ncclResult r__ = (call); if (r__ != 0) return otg_fail(ctx, OTG_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); } while (0)

} // namespace

struct otg_comm {
  ncclCommT comm = nullptr;
  int device = 0, rank = 0, world = 1;
  uint64_t* d_counts = nullptr;          // 3 own + 3 * world gathered
  DevBuf stage[3];                       // rank 0: regions / alleles / sequences of the whole job
};

extern "C" {

int otg_comm_unique_id(uint8_t* id_out)
{
  if (!id_out) return otg_fail(nullptr, OTG_ERR_ARG, "otg_comm_unique_id: NULL");
  if (!rccl_load()) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "%s", g_rccl.err.c_str());
  ncclId id;
  RCCL_TRY(nullptr, g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, OTG_COMM_ID_BYTES);
  return OTG_OK;
}

int otg_comm_create(int device, int rank, int world, const uint8_t* id, otg_comm** out)
{
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return otg_fail(nullptr, OTG_ERR_ARG, "otg_comm_create: bad argument");
  *out = nullptr;
  if (!rccl_load()) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "%s", g_rccl.err.c_str());
  if (hipSetDevice(device) != hipSuccess) return otg_fail(nullptr, OTG_ERR_NO_DEVICE, "otg_comm_create: device %d cannot be selected", device);
  otg_comm* c = new otg_comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclId nid;
  memcpy(nid.internal, id, OTG_COMM_ID_BYTES);
  const ncclResult r = g_rccl.CommInitRank(&c->comm, world, nid, rank);
  if (r != 0) { delete c; return otg_fail(nullptr, OTG_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r)); }
  if (hipMalloc((void**)&c->d_counts, (size_t)(3 + 3 * world) * sizeof(uint64_t)) != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; return otg_fail(nullptr, OTG_ERR_HIP, "otg_comm_create: hipMalloc failed"); }
  *out = c;
  return OTG_OK;
}

void otg_comm_destroy(otg_comm* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& b : c->stage) if (b.p) (void)hipFree(b.p);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->comm) g_rccl.CommDestroy(c->comm);
  delete c;
}

int otg_gather_sizes(otg_ctx* ctx, otg_comm* c, uint64_t* counts_out)
{
  if (!ctx || !c || !counts_out) return otg_fail(ctx, OTG_ERR_ARG, "otg_gather_sizes: NULL argument");
  if (ctx->device != c->device) return otg_fail(ctx, OTG_ERR_ARG, "otg_gather_sizes: the context and the communicator sit on different devices");
  uint32_t na = 0; uint64_t sb = 0; otg_run_stats st;
  int rc = otg_assemble_result_sizes(ctx, &na, &sb);
  if (rc == OTG_OK) rc = otg_assemble_stats(ctx, &st);
  if (rc != OTG_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t mine[3] = {st.n_regions, na, sb};
  HIP_TRY(ctx, hipMemcpyAsync(c->d_counts, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
  RCCL_TRY(ctx, g_rccl.AllGather(c->d_counts, c->d_counts + 3, 3, NCCL_UINT64, c->comm, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(counts_out, c->d_counts + 3, (size_t)3 * c->world * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OTG_OK;
}

int otg_gather_records(otg_ctx* ctx, otg_comm* c, const uint64_t* counts, otg_region_result* regions_out, otg_allele* alleles_out, uint8_t* seqs_out)
{
  if (!ctx || !c || !counts) return otg_fail(ctx, OTG_ERR_ARG, "otg_gather_records: NULL argument");
  const bool root = c->rank == 0;
  if (root && (!regions_out || !alleles_out || !seqs_out)) return otg_fail(ctx, OTG_ERR_ARG, "otg_gather_records: rank 0 needs the three output buffers");
  const otg_region_result* d_reg = nullptr; const otg_allele* d_al = nullptr; const uint8_t* d_seq = nullptr;
  int rc = otg_assemble_device_results(ctx, &d_reg, &d_al, &d_seq);
  if (rc != OTG_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t unit[3] = {sizeof(otg_region_result), sizeof(otg_allele), 1};
  const void* src[3] = {d_reg, d_al, d_seq};
  uint64_t total[3] = {0, 0, 0};
  for (int r = 0; r < c->world; ++r) for (int k = 0; k < 3; ++k) total[k] += counts[3 * r + k];
  if (root) {
    for (int k = 0; k < 3; ++k) {
      const size_t need = (size_t)total[k] * unit[k] + 16;
      if (c->stage[k].cap < need) {
        if (c->stage[k].p) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(c->stage[k].p); c->stage[k].p = nullptr; c->stage[k].cap = 0; }
        HIP_TRY(ctx, hipMalloc(&c->stage[k].p, need));
        c->stage[k].cap = need;
      }
    }
  }
  // a gather with per-rank sizes: one group of point-to-point transfers (rank 0 has seven direct xGMI links on an 8-GPU node: the peers send at once)
  RCCL_TRY(ctx, g_rccl.GroupStart());
  if (root) {
    uint64_t off[3] = {counts[0], counts[1], counts[2]};          // rank 0's own part comes first
    for (int r = 1; r < c->world; ++r)
      for (int k = 0; k < 3; ++k) {
        const size_t bytes = (size_t)counts[3 * r + k] * unit[k];
        if (bytes) RCCL_TRY(ctx, g_rccl.Recv((uint8_t*)c->stage[k].p + (size_t)off[k] * unit[k], bytes, NCCL_UINT8, r, c->comm, ctx->stream));
        off[k] += counts[3 * r + k];
      }
  } else {
    for (int k = 0; k < 3; ++k) {
      const size_t bytes = (size_t)counts[3 * c->rank + k] * unit[k];
      if (bytes) RCCL_TRY(ctx, g_rccl.Send(src[k], bytes, NCCL_UINT8, 0, c->comm, ctx->stream));
    }
  }
  RCCL_TRY(ctx, g_rccl.GroupEnd());
  if (root) {
    for (int k = 0; k < 3; ++k) {
      const size_t bytes = (size_t)counts[k] * unit[k];
      if (bytes) HIP_TRY(ctx, hipMemcpyAsync(c->stage[k].p, src[k], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    void* dst[3] = {regions_out, alleles_out, seqs_out};
    for (int k = 0; k < 3; ++k) if (total[k]) HIP_TRY(ctx, hipMemcpyAsync(dst[k], c->stage[k].p, (size_t)total[k] * unit[k], hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (root) {
    // job-wide indices: a rank's region results point into its own allele array, its alleles into its own regions and sequence arena
    uint64_t rbase = 0, abase = 0, sbase = 0;
    for (int r = 0; r < c->world; ++r) {
      for (uint64_t i = 0; i < counts[3 * r]; ++i) regions_out[rbase + i].first_allele += (uint32_t)abase;
      for (uint64_t i = 0; i < counts[3 * r + 1]; ++i) { alleles_out[abase + i].region += (uint32_t)rbase; alleles_out[abase + i].seq_off += sbase; }
      rbase += counts[3 * r]; abase += counts[3 * r + 1]; sbase += counts[3 * r + 2];
    }
  }
  return OTG_OK;
}

} // extern "C"
