// poa.hip — batched "pseudo-POA" consensus (gfx950).
//
// Replaces PPOA (reference: src/anppoa.hpp:64-380) as driven by rapid_consensus
// (src/analignments.cpp:261-292): backbone graph, per-read op-string threading with alt-node reuse,
// weight damping (adjust_weights :243-252) and heaviest path ending in an ending node (:254-380).
//
// Graph image in HBM (per graph, sizes from a counting pre-pass over the op strings):
//   nodes: base byte, is_end flag, in-degree, extra-edge list head/tail, heaviest weight, predecessor;
//   backbone edge i -> i+1 is implicit: weight 1 + bb_cnt[i] (integer count, exact in FP32);
//   extra edges (everything else) in creation order with a per-source singly linked list, which is the
//   reference's per-node insertion order.
// The reference finds the heaviest path by scanning, for every node, all edges of the graph (O(N*E)) and
// copying path vectors; here it is one Kahn sweep that PUSHES along out-edges and keeps, per node, the best
// (weight, source) with the reference's tie-breaks restated: candidates are compared with strict '>' in
// incoming-scan order = ascending source id (one edge per (source,sink) pair, so the source id alone orders
// ties), first candidate always accepted; the end node is the lowest id among the heaviest ending nodes.
// Damping is applied on the fly with the reference's FP32 operations (-ffp-contract=off).
#include "otg_common.hpp"
#include <vector>
#include <cstdlib>
#include <algorithm>

namespace {

// node / edge records: everything the heaviest-path sweep and the list walks need about a node (or an edge) comes with ONE load
struct NodeG { float hw; int32_t pred; uint32_t indeg; int32_t head; };          // best weight, its source (-1 = none yet), in-degree, first extra out-edge;
                                                                                  // while the graph is being built `pred` holds the LAST extra out-edge (list tail)
struct EdgeG { uint32_t sink; float w; int32_t next; uint32_t base; };            // base = base of an alt-node sink (0 for backbone sinks); second generation: bits 8.. = the edge's source
struct NodeL { float hw; int16_t pred; uint16_t indeg; int16_t head; uint16_t pad; };
struct EdgeL { uint16_t sink; uint16_t w; int16_t next; uint16_t base; };

struct PoaDev {
  const uint8_t* seq_arena;
  const uint8_t* cig_arena;
  const otg_poa_member* members;
  const otg_poa_graph* graphs;
  uint32_t n_graphs;
  // per-graph layout
  const uint64_t* node_off;   // [n_graphs+1]  capacities (differences) and the place of graph g's consensus in out_arena
  const uint64_t* edge_off;   // [n_graphs+1]  capacities
  // where graph g's image lives in the work arrays below: the graphs of one launch (a bounded share of the batch, see otg_launch_poa) are
  // packed from offset 0, the next launch reuses the arrays
  const uint64_t* wnode_off; const uint64_t* wedge_off; const uint64_t* wstart_off; const uint64_t* wanch_off;
  // node arrays
  NodeG* nodes; uint8_t* node_base; uint8_t* is_end; uint32_t* bb_cnt; uint32_t* queue;
  // edge array
  EdgeG* edges;
  uint32_t* start_list;
  // outputs
  const uint64_t* out_off; uint32_t* out_len; uint8_t* out_arena; uint32_t* out_start; int32_t* status;
  const uint32_t* order;      // launch lists: graphs for the LDS kernel, then graphs for the global-memory kernel (longest first each)
  unsigned long long* prof;   // OTG_POA_PROFILE: wall-clock ticks per phase, summed over graphs (null otherwise)
  unsigned long long* prof_graph;   // ... and six words per graph: threading ticks, wide / narrow / serial chunks, ticks in the narrow + serial code, members | backbone << 32
  uint32_t* fb_list;          // graphs left to the global-memory kernel by the LDS kernel (outgrew the optimistic capacities)
  uint32_t* fb_count;
  // second generation of the global-memory path (v2 != 0): per edge its successor in the edge list of its ANCHOR (the backbone node its
  // subtree of alt nodes hangs off; -1 = subtrees of start nodes) — its source shares a word with the sink base —, per anchor (index
  // anchor + 1, base wanch_off[g]) that list's head / tail / length; an alt node keeps its one in-edge (+ 1) in the in-degree field
  int v2;
  int32_t* anext; int32_t* ahead; int32_t* atail; uint32_t* acnt;
};
struct PoaAux { int32_t* anext; int32_t* ahead; int32_t* atail; uint32_t* acnt; };

// capacity bounds per member: alt = ops that can create a node (X, I); mrun = 'M' ops not preceded by an 'M' — with the X / I ops the
// only ones that can create an edge outside the backbone (insert_edge from a non-'M' predecessor, src/anppoa.hpp:96-110)
__global__ void poa_count_kernel(const uint8_t* __restrict__ cig_arena, const otg_poa_member* __restrict__ members,
                                 uint32_t n_members, uint32_t* __restrict__ n_alt, uint32_t* __restrict__ n_mrun)
{
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_members) return;
  const uint8_t* c = cig_arena + members[m].cigar_off;
  const uint32_t n = members[m].cigar_len;
  uint32_t alt = 0, mrun = 0;
  uint8_t prev = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t op = c[i];
    alt += (op == 'X' || op == 'I');
    mrun += (op == 'M' && prev != 'M');
    prev = op;
  }
  n_alt[m] = alt; n_mrun[m] = mrun;
}

// Mapping: one WAVE per graph.  Everything that is sequential in the reference stays sequential but runs
// wave-uniform (all lanes execute the same instruction on the same address: one memory request, no divergence);
// the bulk of the threading is data-parallel: op strings are read 64 ops at a time, positions come from ballot
// prefix counts, and an 'M' that follows an 'M' only bumps the implicit backbone edge ref-1 -> ref (distinct
// addresses per lane), so only the ops around mismatches and gaps (15-35 % for ONT reads) take the serial path.
// One wave owns a graph, so plain read-modify-writes need no atomics.
//
// The serial part is a chain of dependent loads, so the data layout is built to keep the chain short: a node record
// {weight, source, in-degree, first edge} and an edge record {sink, weight, next, sink base} are one 16-byte load each; the
// Kahn sweep carries the record it has just written in registers (on a backbone run the next node IS that record) and
// remembers its latest push instead of reading the queue back, so a backbone step needs one round of independent loads
// (successor record + backbone count) where separate arrays needed four dependent ones; an op that follows an alt node
// compares the base stored in the edge instead of loading the sink node.
//
// Small graphs live in LDS (LDS = true: 16-bit ids, 18 bytes per node + 8 per edge).  The LDS capacities are optimistic
// (half of the worst-case alt-node / edge bounds); a graph that outgrows them is queued for the global-memory instantiation
// (LDS = false), which has the worst-case capacities and is also the path for large graphs — otter's allele graphs
// (~6200 nodes for a 3 kb allele) among them.
#define OTG_LDS __attribute__((address_space(3)))
template <bool LDS> struct PoaStore;
template <> struct PoaStore<false> {
  NodeG* nodes; EdgeG* edges;
  uint8_t *nbase, *isend; uint32_t *bbc, *queue, *starts;
  uint32_t node_cap, edge_cap; bool reduced;
  __device__ __forceinline__ NodeG ldN(uint32_t i) const { return nodes[i]; }
  __device__ __forceinline__ void stN(uint32_t i, const NodeG& n) const { nodes[i] = n; }
  __device__ __forceinline__ void stList(uint32_t i, int head, int tail) const { nodes[i].head = head; nodes[i].pred = tail; }
  __device__ __forceinline__ void stTail(uint32_t i, int tail) const { nodes[i].pred = tail; }
  __device__ __forceinline__ void stPred(uint32_t i, int pr) const { nodes[i].pred = pr; }
  __device__ __forceinline__ void incIndeg(uint32_t i) const { nodes[i].indeg = nodes[i].indeg + 1u; }
  __device__ __forceinline__ void incIndegAtomic(uint32_t i) const { atomicAdd(&nodes[i].indeg, 1u); }
  __device__ __forceinline__ float ldHw(uint32_t i) const { return nodes[i].hw; }
  __device__ __forceinline__ void stHw(uint32_t i, float w) const { nodes[i].hw = w; }
  __device__ __forceinline__ void stHwPred(uint32_t i, float w, int pr) const { nodes[i].hw = w; nodes[i].pred = pr; }
  __device__ __forceinline__ int ldPred(uint32_t i) const { return nodes[i].pred; }
  __device__ __forceinline__ EdgeG ldE(int e) const { return edges[e]; }
  __device__ __forceinline__ void stE(int e, const EdgeG& x) const { edges[e] = x; }
  __device__ __forceinline__ void stEw(int e, float w) const { edges[e].w = w; }
  __device__ __forceinline__ void stEnext(int e, int nx) const { edges[e].next = nx; }
  __device__ __forceinline__ uint32_t ldEbase(int e) const { return edges[e].base; }
  __device__ __forceinline__ void stIndeg(uint32_t i, uint32_t v) const { nodes[i].indeg = v; }
};
template <> struct PoaStore<true> {
  OTG_LDS NodeL* nodes; OTG_LDS EdgeL* edges;
  OTG_LDS uint8_t *nbase, *isend; OTG_LDS uint16_t *bbc, *queue, *starts;
  uint32_t node_cap, edge_cap; bool reduced;
  __device__ __forceinline__ NodeG ldN(uint32_t i) const { NodeG n; n.hw = nodes[i].hw; n.pred = nodes[i].pred; n.indeg = nodes[i].indeg; n.head = nodes[i].head; return n; }
  __device__ __forceinline__ void stN(uint32_t i, const NodeG& n) const { nodes[i].hw = n.hw; nodes[i].pred = (int16_t)n.pred; nodes[i].indeg = (uint16_t)n.indeg; nodes[i].head = (int16_t)n.head; }
  __device__ __forceinline__ void stList(uint32_t i, int head, int tail) const { nodes[i].head = (int16_t)head; nodes[i].pred = (int16_t)tail; }
  __device__ __forceinline__ void stTail(uint32_t i, int tail) const { nodes[i].pred = (int16_t)tail; }
  __device__ __forceinline__ void stPred(uint32_t i, int pr) const { nodes[i].pred = (int16_t)pr; }
  __device__ __forceinline__ void incIndeg(uint32_t i) const { nodes[i].indeg = (uint16_t)(nodes[i].indeg + 1u); }
  __device__ __forceinline__ void incIndegAtomic(uint32_t i) const { incIndeg(i); }     // not used: the LDS pass is wave-uniform
  __device__ __forceinline__ float ldHw(uint32_t i) const { return nodes[i].hw; }
  __device__ __forceinline__ void stHw(uint32_t i, float w) const { nodes[i].hw = w; }
  __device__ __forceinline__ void stHwPred(uint32_t i, float w, int pr) const { nodes[i].hw = w; nodes[i].pred = (int16_t)pr; }
  __device__ __forceinline__ int ldPred(uint32_t i) const { return nodes[i].pred; }
  __device__ __forceinline__ EdgeG ldE(int e) const { EdgeG x; x.sink = edges[e].sink; x.w = (float)edges[e].w; x.next = edges[e].next; x.base = edges[e].base; return x; }
  __device__ __forceinline__ void stE(int e, const EdgeG& x) const { edges[e].sink = (uint16_t)x.sink; edges[e].w = (uint16_t)x.w; edges[e].next = (int16_t)x.next; edges[e].base = (uint16_t)x.base; }
  __device__ __forceinline__ void stEw(int e, float w) const { edges[e].w = (uint16_t)w; }
  __device__ __forceinline__ void stEnext(int e, int nx) const { edges[e].next = (int16_t)nx; }
  __device__ __forceinline__ uint32_t ldEbase(int e) const { return edges[e].base; }
  __device__ __forceinline__ void stIndeg(uint32_t i, uint32_t v) const { nodes[i].indeg = (uint16_t)v; }
};

template <bool LDS> __device__ __forceinline__ void poa_phase_fence()
{
  if (LDS) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  else __threadfence();
}

// returns false when the graph has to be redone with larger capacities (LDS only)
using lds_u32 = __attribute__((address_space(3))) uint32_t;
using lds_i32 = __attribute__((address_space(3))) int32_t;
using lds_f32 = __attribute__((address_space(3))) float;
constexpr int POA_WK = 8;     // ops per lane in a wide chunk of the threading (64 * POA_WK ops, at most 64 runs)
struct PoaScratch {          // per-wave LDS of the second-generation path (3.3 KB)
  volatile lds_u32* ops;     // [64 * POA_WK] the chunk's ops: op | target base << 8
  volatile lds_u32* stage;   // [64]  edge ids of the block being swept
  volatile lds_f32* hw;      // [128] window of backbone weights ...
  volatile lds_i32* pred;    // [128] ... and their sources (circular: node v at v & 127)
};

// exclusive prefix sum over the 64 lanes; *total = the sum (wave-uniform).  DPP row shifts + row broadcasts: no LDS round trips.
__device__ __forceinline__ uint32_t poa_wave_excl_sum(uint32_t v, int lane, uint32_t* total)
{
  (void)lane;
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive within each row of 16
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  *total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  return (uint32_t)x - v;
}
// LDS written by some lanes, read by others of the same wave: LDS requests of a wave are served in order, so only the compiler has to keep
// the order (the accesses are volatile); no wait on the outstanding global loads and stores, which a fence would add
__device__ __forceinline__ void poa_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ float poa_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Second generation (global-memory instantiation, `v2`): the two serial phases of the first are reorganised around one structural fact — an
// alt node is created by exactly one predecessor (alt_step) and only backbone nodes ever get a second in-edge (insert_edge's sink is the
// current backbone position), so the alt nodes form TREES hanging off the backbone node an op string left the backbone at (their ANCHOR),
// and every edge of such a tree ends inside it or on a backbone node further right.
//   * Threading.  A maximal run of non-'M' ops plus the 'M' that closes it only touches the subtree of its anchor, and along one op string
//     every backbone node is left at most once: the runs of a member are independent.  A chunk of the op string is cut after its last
//     'M'; ONE LANE PER RUN walks the existing subtree (its dependent loads overlap with those of 63 other runs), counts the nodes and
//     edges it has to create from its first miss on, a wave prefix sum hands out node ids in op order — the reference's creation order,
//     which its tie-breaks see — and the lanes then write their chains.  Anything irregular (the ops before a member's first 'M', a
//     window of 64 ops without an 'M', positions beyond the backbone) takes the serial code of the first generation.
//   * Heaviest path.  Anchors in backbone order are a topological order (a subtree is entered from its anchor and left to the right), and
//     the edges of one subtree, kept in creation order in a per-anchor list, have every node's in-edge before its out-edges.  So the sweep
//     walks the backbone once: a block of consecutive anchors with <= 64 subtree edges is gathered into registers (one lane per anchor
//     follows its list; one lane per edge loads the record and finds the slot of its source's in-edge), then a wave-uniform loop relaxes
//     them with every operand in registers or in a 128-node LDS window of backbone (weight, source) pairs — no load depends on a load.
//     Same FP32 operations in the same order per path as the reference's push (weights add up along a path, never across), same
//     tie-breaks; in-degrees and the work queue are not needed.
// Graphs that break the tree property (op strings running past the backbone: the reference indexes out of range there) keep the Kahn sweep.
template <bool LDS>
__device__ __forceinline__ bool poa_graph_body(const PoaDev& P, const uint32_t g, const PoaStore<LDS>& S, const uint32_t out_cap,
                                               const PoaAux& X, const PoaScratch& L, const bool v2_in)
{
  const int lane = threadIdx.x & 63;
  const otg_poa_graph G = P.graphs[g];
  const uint32_t node_cap = S.node_cap, edge_cap = S.edge_cap;
  auto nbase = S.nbase; auto isend = S.isend; auto bbc = S.bbc; auto queue = S.queue; auto starts = S.starts;
  const int B = (int)G.backbone_len;
  const bool V2 = !LDS && v2_in && B >= 2;
  constexpr int WK = POA_WK;
  // V2: a stretch of plain 'M' ops raises the counts of a RANGE of backbone edges by one; the lane-parallel threading only marks the two
  // ends of each stretch (+1 / -1, atomics without a return value) and one prefix sum after the last member turns the marks into counts
  int* bdiff = (int*)(void*)&queue[0];
  uint32_t n_nodes = (uint32_t)B, n_edges = 0, n_start = B >= 2 ? 1u : 0u;
  int status = 0;
  bool tree_ok = true;                 // V2: every alt node has one in-edge and every edge list belongs to one anchor
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long t0 = P.prof ? wall_clock64() : 0ull;
  unsigned long long pf_wide = 0, pf_wide_ops = 0, pf_narrow = 0, pf_serial = 0, pf_follow = 0, pf_sM = 0, pf_sX = 0, pf_sI = 0, pf_sD = 0, pf_snom = 0, pf_sirr = 0, pf_shead = 0;     // OTG_POA_PROFILE: chunks by kind
  unsigned long long tq_serial = 0, tq_layout = 0, tq_mid = 0, tq_a = 0, tq_turn = 0, tq_c = 0, tq_mark = 0;      // ... and where a wide chunk's time goes (10 ns ticks)
  auto tick = [&](unsigned long long& acc) { if (P.prof) { const unsigned long long now = wall_clock64(); acc += now - tq_mark; tq_mark = now; } };

  // ---- PPOA::init (src/anppoa.hpp:64-84)
  {
    const uint8_t* bb = P.seq_arena + G.backbone_off;
    for (uint32_t i = (uint32_t)lane; i < node_cap; i += 64) {
      const bool isb = (int)i < B && B >= 2;
      nbase[i] = isb ? bb[i] : (uint8_t)0;
      isend[i] = (isb && i >= 1 && B - (int)i <= 10) ? 1 : 0;
      NodeG n; n.hw = 0.0f; n.pred = -1; n.indeg = (isb && i >= 1) ? 1u : 0u; n.head = -1;
      S.stN(i, n);              // pred = -1: empty extra-edge list (tail) for now
      bbc[i] = 0;
      if (V2) queue[i] = 0;     // V2: the work queue of the Kahn sweep doubles as the difference array of the backbone counts (bdiff below)
    }
    if (V2) for (int i = lane; i <= B; i += 64) { X.ahead[i] = -1; X.atail[i] = -1; X.acnt[i] = 0u; }
    if (B >= 2) starts[0] = 0;
    poa_phase_fence<LDS>();
  }

  // Extra out-edge list (head, tail) of a node.  Two ways around the dependent load: the lists of the node created last are
  // still in registers (cn_*: a fresh node has none, and an op string that leaves the known graph keeps creating nodes), and the
  // lists of the backbone node an op starts from are read for all 64 ops of a chunk at once (pf_* below).
  uint32_t cn_id = 0xffffffffu; int cn_head = -1, cn_tail = -1;
  int cur_anc = 0;                     // V2, serial code: anchor of the subtree `prev` is in
  auto node_lists = [&](uint32_t x, int& head, int& tail) {
    if (x == cn_id) { head = cn_head; tail = cn_tail; }
    else { const NodeG n = S.ldN(x); head = n.head; tail = n.pred; }
  };
  auto new_node = [&](uint8_t base) -> uint32_t {
    if (n_nodes >= node_cap) { status = 1; return n_nodes - 1; }
    nbase[n_nodes] = base;
    cn_id = n_nodes; cn_head = -1; cn_tail = -1;
    return n_nodes++;
  };
  auto append_edge = [&](uint32_t src, int src_head, int src_tail, uint32_t sink, uint32_t sink_base) {
    if (n_edges >= edge_cap) { status = 2; return; }
    const int e = (int)n_edges++;
    EdgeG x; x.sink = sink; x.w = 1.0f; x.next = -1; x.base = V2 ? (sink_base | (src << 8)) : sink_base;
    S.stE(e, x);
    if (src_tail >= 0) { S.stEnext(src_tail, e); S.stTail(src, e); } else { src_head = e; S.stList(src, e, e); }
    if (src == cn_id) { cn_head = src_head; cn_tail = e; }
    // the sink's in-degree is counted in one pass over the edge array before the sweep (no read-modify-write on this path)
    if (V2) {
      X.anext[e] = -1;
      if ((int)sink >= B) S.stIndeg(sink, (uint32_t)(e + 1));      // the one in-edge of an alt node
      if (cur_anc < -1 || cur_anc >= B || ((int)src < B && (int)src != cur_anc)) tree_ok = false;
      else {
        const int t = X.atail[cur_anc + 1];
        if (t >= 0) X.anext[t] = e; else X.ahead[cur_anc + 1] = e;
        X.atail[cur_anc + 1] = e;
        X.acnt[cur_anc + 1] = X.acnt[cur_anc + 1] + 1u;
      }
    }
  };
  auto insert_edge = [&](uint32_t src, int src_head, int src_tail, uint32_t sink) {     // src/anppoa.hpp:96-110
    if ((int)src < B - 1 && sink == src + 1) { bbc[src] = bbc[src] + 1; return; }
    for (int e = src_head; e >= 0;) {
      const EdgeG x = S.ldE(e);
      if (x.sink == sink) { S.stEw(e, x.w + 1.0f); return; }
      e = x.next;
    }
    if ((int)sink >= B) tree_ok = false;      // a second way into an alt node (only op strings that run past the backbone get here)
    append_edge(src, src_head, src_tail, sink, (int)sink >= B ? (uint32_t)nbase[sink] : 0u);
  };
  int as_edge = -1;                    // the edge the latest alt_step followed (-1: it made a node)
  auto alt_step = [&](uint32_t prev, int prev_head, int prev_tail, uint8_t tc) -> uint32_t {   // :162-186 / :206-233
    for (int e = prev_head; e >= 0;) {
      const EdgeG x = S.ldE(e);
      if ((int)x.sink >= B && (x.base & 0xffu) == (uint32_t)tc) { S.stEw(e, x.w + 1.0f); as_edge = e; return x.sink; }
      e = x.next;
    }
    as_edge = -1;
    const uint32_t nn = new_node(tc);
    if (!status) append_edge(prev, prev_head, prev_tail, nn, tc);     // a fresh node: no edge to it can exist yet
    return nn;
  };

  // the member whose op string is being threaded
  const uint8_t* seq = nullptr; const uint8_t* cig = nullptr;
  int clen = 0, slen = 0;
  bool spl = false, spr = false;
  auto set_member = [&](const uint32_t mi) {
    const otg_poa_member M = P.members[G.first_member + mi];
    seq = P.seq_arena + M.seq_off; cig = P.cig_arena + M.cigar_off;
    clen = (int)M.cigar_len; slen = (int)M.seq_len;
    spl = M.spanning_l != 0; spr = M.spanning_r != 0;
  };
  // one lane per run (lead: this lane has one; p0 = its first op in L.ops, ref0 = the backbone position there, nv = ops in the chunk)
  auto par_runs = [&](const bool lead, const int p0, const int ref0, const int nv) {
      uint32_t n_newn = 0, n_newe = 0, pv = 0, last_ex = 0;
      int ph = -1, pt = -1, ltl = -1, r = 0, r_miss = 0, jmiss = -1, r_close = -1, anc = 0;
      bool missed = false, close_new = false, closed = false;
      if (lead) {
        // phase A: follow the run through the subtree that exists (weights of the edges it reuses go up), stop creating at the first miss
        anc = ref0 - 1; pv = (uint32_t)anc; r = ref0;
        { const NodeG na = S.ldN(pv); ph = na.head; pt = na.pred; }
        for (int j = p0; j < nv; ++j) {
          const uint32_t o = L.ops[j];
          const int op = (int)(o & 0xffu);
          const uint32_t tc = (o >> 8) & 0xffu;
          if (op == 'M') { closed = true; r_close = r; break; }
          if (op == 'D') { r += 1; if (!missed && B - r <= 10 && spr) isend[pv] = 1; continue; }
          if (!missed) {
            bool found = false;
            for (int e = ph; e >= 0;) {
              const EdgeG x = S.ldE(e);
              if ((int)x.sink >= B && (x.base & 0xffu) == tc) { S.stEw(e, x.w + 1.0f); pv = x.sink; found = true; break; }
              e = x.next;
            }
            if (found) { const NodeG nn = S.ldN(pv); ph = nn.head; pt = nn.pred; }
            else { missed = true; jmiss = j; r_miss = r; last_ex = pv; ltl = pt; }
          }
          if (missed) ++n_newn;
          if (op == 'X') r += 1;
          if (!missed && B - r <= 10 && spr) isend[pv] = 1;
        }
        if (closed) {
          if (!missed) {
            if ((int)pv < B - 1 && r_close == (int)pv + 1) bbc[pv] = bbc[pv] + 1;
            else {
              bool found = false;
              for (int e = ph; e >= 0;) {
                const EdgeG x = S.ldE(e);
                if ((int)x.sink == r_close) { S.stEw(e, x.w + 1.0f); found = true; break; }
                e = x.next;
              }
              if (!found) { close_new = true; last_ex = pv; ltl = pt; }
            }
          }
          if (B - (r_close + 1) <= 10 && spr) isend[r_close] = 1;
        }
        n_newe = n_newn + ((closed && (missed || close_new)) ? 1u : 0u);
      }
      // phase B: ids in op order
      tick(tq_a);
      tick(tq_turn);
      uint32_t tot_n = 0, tot_e = 0;
      const uint32_t nb = n_nodes + poa_wave_excl_sum(n_newn, lane, &tot_n);
      const uint32_t eb = n_edges + poa_wave_excl_sum(n_newe, lane, &tot_e);
      if (n_nodes + tot_n > node_cap) status = 1;
      else if (n_edges + tot_e > edge_cap) status = 2;
      if (!status) {
        // phase C: the new chain of each run
        if (lead && n_newe) {
          uint32_t src = last_ex, k = 0;
          if (missed) {
            int r2 = r_miss;
            for (int j = jmiss; j < nv; ++j) {
              const uint32_t o = L.ops[j];
              const int op = (int)(o & 0xffu);
              const uint32_t tc = (o >> 8) & 0xffu;
              if (op == 'M') break;
              if (op == 'D') { r2 += 1; if (B - r2 <= 10 && spr) isend[src] = 1; continue; }
              const uint32_t id = nb + k;
              const int ein = (int)(eb + k);
              const bool has_out = k + 1u < n_newe;
              nbase[id] = (uint8_t)tc;
              EdgeG x; x.sink = id; x.w = 1.0f; x.next = -1; x.base = tc | (src << 8);
              S.stE(ein, x);
              X.anext[ein] = has_out ? ein + 1 : -1;
              NodeG nn; nn.hw = 0.0f; nn.pred = has_out ? ein + 1 : -1; nn.indeg = (uint32_t)(ein + 1); nn.head = has_out ? ein + 1 : -1;
              S.stN(id, nn);
              src = id; ++k;
              if (op == 'X') r2 += 1;
              if (B - r2 <= 10 && spr) isend[id] = 1;
            }
          }
          if (closed && (missed || close_new)) {
            const int ec = (int)(eb + k);
            EdgeG x; x.sink = (uint32_t)r_close; x.w = 1.0f; x.next = -1; x.base = src << 8;
            S.stE(ec, x);
            X.anext[ec] = -1;
          }
          if (ltl >= 0) { S.stEnext(ltl, (int)eb); S.stTail(last_ex, (int)eb); } else S.stList(last_ex, (int)eb, (int)eb);
          const int t = X.atail[anc + 1];
          if (t >= 0) X.anext[t] = (int)eb; else X.ahead[anc + 1] = (int)eb;
          X.atail[anc + 1] = (int)(eb + n_newe - 1u);
          X.acnt[anc + 1] = X.acnt[anc + 1] + n_newe;
        }
        n_nodes += tot_n; n_edges += tot_e;
      }
      cn_id = 0xffffffffu;
  };
  int pre_ci = -1; int pre_cc[WK];      // V2: ops read ahead for the chunk that starts at pre_ci
#pragma unroll
  for (int k = 0; k < WK; ++k) pre_cc[k] = 0;
  // ---- one wide chunk: WK ops per lane from op ci0 on (up to 64 * WK ops, cut after their last 'M' so that the next chunk starts right after an
  // 'M' and every run lies inside one chunk; at most 64 runs).  The round trips of a chunk (ops, target bases, backbone counts, the runs'
  // subtrees) are what the threading waits for, so few, full chunks.  Returns false — nothing done — when the window holds something out
  // of the ordinary: the narrow code below takes it.
  auto wide_chunk = [&](const int ci0, const int ref0, const int tgt0, int& nv_out, int& totr_out, int& tott_out, int& lastop_out) -> bool {
    const int rem = clen - ci0;
    const int rem4 = rem < 64 * WK ? rem : 64 * WK;
    const int o0 = WK * lane;
    int cc[WK];
    if (P.prof) tq_mark = wall_clock64();
    if (pre_ci == ci0) {
#pragma unroll
      for (int k = 0; k < WK; ++k) cc[k] = o0 + k < rem4 ? pre_cc[k] : 0;
    } else {
#pragma unroll
      for (int k = 0; k < WK; ++k) cc[k] = o0 + k < rem4 ? (int)cig[ci0 + o0 + k] : 0;
    }
    int nv = rem4;
    if (rem > 64 * WK) {
      int lm = -1;
#pragma unroll
      for (int k = 0; k < WK; ++k) if (cc[k] == 'M') lm = o0 + k;
      lm = otg_wave_max_i32(lm);
      if (lm < 0) return false;
      nv = lm + 1;
    }
    uint32_t totr = 0, tott = 0, totl = 0;
    int refk[WK], tgtk[WK]; bool simk[WK], leadk[WK];
    uint32_t li = 0;
    // Two passes at most: a chunk with more than 64 runs (one lane per run) is cut in front of its 65th run — a run starts right after an
    // 'M', so the shorter chunk still ends on one — and laid out again.  (At ONT divergence a 512-op chunk holds ~63 runs.)
    for (int attempt = 0; attempt < 2; ++attempt) {
      uint32_t nref = 0, ntgt = 0;
#pragma unroll
      for (int k = 0; k < WK; ++k) {
        const bool v = o0 + k < nv;
        const int c4 = cc[k];
        nref += (v && (c4 == 'M' || c4 == 'X' || c4 == 'D')) ? 1u : 0u;
        ntgt += (v && (c4 == 'M' || c4 == 'X' || c4 == 'I')) ? 1u : 0u;
      }
      int ra = ref0 + (int)poa_wave_excl_sum(nref, lane, &totr), ta = tgt0 + (int)poa_wave_excl_sum(ntgt, lane, &tott);
      int pcur = __shfl_up(cc[WK - 1], 1);
      if (lane == 0) pcur = 'M';
      bool irregular = false;
      uint32_t nl = 0;
#pragma unroll
      for (int k = 0; k < WK; ++k) {
        const bool v = o0 + k < nv;
        const int c4 = cc[k];
        const bool m = c4 == 'M', x = c4 == 'X', d = c4 == 'D', in = c4 == 'I';
        refk[k] = ra; tgtk[k] = ta;
        simk[k] = v && m && pcur == 'M' && ra < B;
        leadk[k] = v && !m && pcur == 'M';
        if (v && (!(m || x || d || in) || (m && ra >= B) || (!m && ra > B) || ((x || in) && ta >= slen))) irregular = true;
        if (v) { ra += (m || x || d) ? 1 : 0; ta += (m || x || in) ? 1 : 0; pcur = c4; }
        nl += leadk[k] ? 1u : 0u;
      }
      li = poa_wave_excl_sum(nl, lane, &totl);
      if (__ballot(irregular) || (totl > 64u && attempt == 1)) return false;
      if (totl <= 64u) break;
      int cut = -(1 << 30);                              // minus the position of the run that would be the 65th
      uint32_t rank = li;
#pragma unroll
      for (int k = 0; k < WK; ++k) if (leadk[k]) { if (rank == 64u) cut = -(o0 + k); ++rank; }
      nv = -otg_wave_max_i32(cut);
    }
    nv = __builtin_amdgcn_readfirstlane(nv);
    int lsel = 0;                                          // the chunk's last op: the op in front of the next chunk
#pragma unroll
    for (int k = 0; k < WK; ++k) if (k == ((nv - 1) & (WK - 1))) lsel = cc[k];
    const int lastop_n = __builtin_amdgcn_readlane(lsel, (nv - 1) / WK);
    tick(tq_layout);
    {
      // the ops of the chunk after this one are asked for now: first-touch reads of the op string cost far more than a cache hit, and
      // this one then overlaps with the rest of this chunk
      pre_ci = ci0 + nv;
#pragma unroll
      for (int k = 0; k < WK; ++k) pre_cc[k] = pre_ci + o0 + k < clen ? (int)cig[pre_ci + o0 + k] : 0;
    }
    // every load of the chunk head first (backbone counts of the plain 'M' ops, target bases), then the stores: one round trip
    int tbv[WK];
#pragma unroll
    for (int k = 0; k < WK; ++k) tbv[k] = (o0 + k < nv && (cc[k] == 'X' || cc[k] == 'I')) ? (int)seq[tgtk[k]] : 0;
    {
      const unsigned long long s3 = __ballot(simk[WK - 1]), s0 = __ballot(simk[0]);
      const bool prev_s = lane > 0 && ((s3 >> (lane - 1)) & 1ull), next_s = lane < 63 && ((s0 >> (lane + 1)) & 1ull);
#pragma unroll
      for (int k = 0; k < WK; ++k) {
        if (simk[k]) {
          const bool ps = k > 0 ? simk[k > 0 ? k - 1 : 0] : prev_s, ns = k < WK - 1 ? simk[k < WK - 1 ? k + 1 : WK - 1] : next_s;
          if (!ps) atomicAdd(&bdiff[refk[k] - 1], 1);
          if (!ns) atomicAdd(&bdiff[refk[k]], -1);
          if (B - (refk[k] + 1) <= 10 && spr && (uint32_t)refk[k] < node_cap) isend[refk[k]] = 1;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < WK; ++k) {
      if (o0 + k < nv) L.ops[o0 + k] = (uint32_t)cc[k] | ((uint32_t)tbv[k] << 8);
      if (leadk[k]) { L.stage[li] = (uint32_t)(o0 + k) | ((uint32_t)refk[k] << 10); ++li; }
    }
    poa_lds_order();
    const bool lead = lane < (int)totl;
    const uint32_t ld = lead ? L.stage[lane] : 0u;
    tick(tq_mid);
    par_runs(lead, (int)(ld & 0x3ffu), (int)(ld >> 10), nv);
    tick(tq_c);
    pf_wide += 1; pf_wide_ops += (unsigned long long)nv;
    nv_out = nv; totr_out = (int)totr; tott_out = (int)tott; lastop_out = lastop_n;
    return true;
  };
  // ---- insert_alignment for every member, in order (src/anppoa.hpp:112-241)
  for (uint32_t mi = 0; mi < G.n_members && !status; ++mi) {
    set_member(mi);
    pre_ci = -1;
    int prev = 0, ref_i = 0, tgt = 0, ci = 0;
    bool first = true;
    cur_anc = 0;
    if (!spl) {
      first = false;
      bool stop = false;
      while (ci < clen && !stop) {     // leading gap ops of a read that does not span the left flank
        const int i = ci + lane;
        const uint8_t c = i < clen ? cig[i] : 0;
        const unsigned long long notdi = ~__ballot(c == 'D' || c == 'I');
        const int upto = notdi ? (int)__builtin_ctzll(notdi) : 64;
        const unsigned long long below = upto >= 64 ? ~0ull : ((1ull << upto) - 1ull);
        const int nD = __builtin_popcountll(__ballot(c == 'D') & below), nI = __builtin_popcountll(__ballot(c == 'I') & below);
        ref_i += nD; tgt += nI;
        if (nD) { prev = ref_i; cur_anc = prev; }
        ci += upto; stop = upto < 64;
      }
    }
    int lastop = 0;                    // op just before position ci in the main phase (0 = none)
    while (ci < clen && !status) {
      const int i = ci + lane;
      const int rem = clen - ci;
      if (V2 && lastop == 'M' && !first && rem > 64) {
        int nv = 0, tr = 0, tt = 0, lo = 0;
        if (wide_chunk(ci, ref_i, tgt, nv, tr, tt, lo)) {
          lastop = lo;
          ref_i += tr; tgt += tt;
          if (lastop == 'M') { prev = ref_i - 1; first = false; cur_anc = prev; }
          ci += nv;
          continue;
        }
      }
      const unsigned long long ts0 = P.prof ? wall_clock64() : 0ull;
      const int c = lane < rem ? (int)cig[i] : 0;
      // V2: the chunk ends after the last 'M' of the 64-op window (the whole rest of the op string when that is shorter), so that the
      // next chunk starts right after an 'M' and every run of non-'M' ops lies inside one chunk
      int nvalid = rem < 64 ? rem : 64;
      bool no_m = false;               // a window of 64 ops without an 'M' in the middle of the op string: serial code
      if (V2 && rem > 64) {
        const unsigned long long mm = __ballot(c == 'M');
        if (!mm) no_m = true;
        else if (lastop == 0) nvalid = (int)__builtin_ctzll(mm) + 1;     // a member's first chunk runs on the serial code: only as far as its first 'M'
        else nvalid = 64 - (int)__builtin_clzll(mm);
      }
      const bool valid = lane < nvalid;
      int pc = __shfl_up(c, 1);
      if (lane == 0) pc = lastop;
      const bool isM = c == 'M', isX = c == 'X', isD = c == 'D', isI = c == 'I';
      const unsigned long long mMXD = __ballot(valid && (isM || isX || isD)), mMXI = __ballot(valid && (isM || isX || isI));
      const int ref_at = ref_i + __builtin_popcountll(mMXD & lt), tgt_at = tgt + __builtin_popcountll(mMXI & lt);
      // 'M' after 'M': prev == ref-1, not the first op -> insert_edge(ref-1, ref) is the implicit backbone edge
      const bool simple = valid && isM && pc == 'M' && ref_at < B;
      const unsigned long long simm = __ballot(simple);
      if (simple) {
        if (V2) {
          const bool ps = lane > 0 && ((simm >> (lane - 1)) & 1ull), ns = lane < 63 && ((simm >> (lane + 1)) & 1ull);
          if (!ps) atomicAdd(&bdiff[ref_at - 1], 1);
          if (!ns) atomicAdd(&bdiff[ref_at], -1);
        } else bbc[ref_at - 1] = bbc[ref_at - 1] + 1;
        if (B - (ref_at + 1) <= 10 && spr && (uint32_t)ref_at < node_cap) isend[ref_at] = 1;
      }
      // the target base of every op travels with it (one coalesced read instead of a dependent load per serial op), and so do
      // the edge lists of the backbone node it starts from when the op before it is an 'M' (prev = ref-1).  Reading them ahead is
      // safe: along one op string every node is left at most once, and an op that leaves ref-1 earlier in this chunk (an 'I'
      // at the same ref) makes the op before this one a non-'M'.
      const int tb = (valid && (isX || isI) && tgt_at < slen) ? (int)seq[tgt_at] : 0;
      // one lane per run when the chunk starts right after an 'M' and holds nothing out of the ordinary
      bool par = V2 && lastop == 'M' && !first && !no_m;
      if (par && __ballot(valid && (!(isM || isX || isD || isI) || (isM && ref_at >= B) || (!isM && ref_at > B) || ((isX || isI) && tgt_at >= slen)))) par = false;
      if (par) {
        L.ops[lane] = valid ? ((uint32_t)c | ((uint32_t)tb << 8)) : 0u;
        poa_lds_order();
        par_runs(valid && !isM && pc == 'M', lane, ref_at, nvalid);
        pf_narrow += 1;
      } else {
      pf_serial += 1;
      if (no_m) pf_snom += 1; else if (lastop == 0 || first) pf_shead += 1; else pf_sirr += 1;
      int pf_head = -1, pf_tail = -1;
      if (valid && !simple && pc == 'M' && ref_at >= 1 && ref_at <= B) { const NodeG n = S.ldN((uint32_t)(ref_at - 1)); pf_head = n.head; pf_tail = n.pred; }
      unsigned long long todo = __ballot(valid && !simple);
      // A stretch of 'X' / 'I' ops that retraces what an earlier member put down in one go (a long insertion shared by the reads of an allele:
      // every step a dependent load of the node's list head and of that edge, ~2 us each, 64 of them per window).  A chain made in one go has
      // consecutive node and edge ids, node u + j leaving through edge e + j + 1 at the head of its list.  So once the op of lane l has
      // followed edge e into node u, the lanes behind it test that guess for their own step in parallel — two coalesced loads — and the
      // longest prefix of steps whose test holds is taken at once: same edges, same weight increments, same end marks as one by one.
      auto follow_chain = [&](const int l) {
        if (LDS || as_edge < 0 || l >= 63) return;
        const unsigned long long xi = __ballot(valid && (isX || isI)) >> (l + 1);
        const int run = (int)__builtin_ctzll(~xi);           // ops of the stretch behind lane l
        if (run < 4) return;
        const int k = lane - l;                              // this lane's step, 1 .. run
        const uint32_t nu = (uint32_t)prev + (uint32_t)(k - 1), ek = (uint32_t)as_edge + (uint32_t)k;
        bool good = false; float w = 0.0f; uint32_t snk = 0;
        if (k >= 1 && k <= run && nu < n_nodes && ek < n_edges && (int)nu >= B) {
          const NodeG N = S.ldN(nu); const EdgeG E = S.ldE((int)ek);
          good = N.head == (int)ek && E.sink == nu + 1u && (E.base & 0xffu) == (uint32_t)tb;
          w = E.w; snk = E.sink;
        }
        const unsigned long long gm = __ballot(good) >> (l + 1);
        const int m = (int)__builtin_ctzll(~gm);             // steps that go as guessed
        if (m == 0) return;
        if (k >= 1 && k <= m) {
          S.stEw((int)ek, w + 1.0f);
          if (B - (ref_at + (isX ? 1 : 0)) <= 10 && spr && snk < node_cap) isend[snk] = 1;
        }
        prev += m;
        todo &= ~(((1ull << m) - 1ull) << (l + 1));
        cn_id = 0xffffffffu;
        pf_follow += (unsigned long long)m;
      };
      // The other long stretch: 'X' / 'I' ops that leave from the node made last, which has no successors yet — every one of them makes a
      // node (an insertion at the head of a read starts at a start node of its own: ~200 ops per member at 2.7 us each where alleles differ
      // by a few repeat units).  Made one by one, ids and edge ids are consecutive and each edge is the only one of its source: so the lanes
      // of the stretch write their node, their edge and their part of the anchor's list side by side, as the one-lane-per-run code does
      // for the chain of a run.  Returns false when the stretch is left to the loop (too short, no room: the loop reports that).
      auto create_chain = [&](const int l) -> bool {
        if (LDS || first || (uint32_t)prev != cn_id || cn_head >= 0) return false;
        const unsigned long long xi = __ballot(valid && (isX || isI)) >> l;
        const int run = ~xi ? (int)__builtin_ctzll(~xi) : 64;     // ops of the stretch from lane l on
        if (run < 4 || n_nodes + (uint32_t)run > node_cap || n_edges + (uint32_t)run > edge_cap) return false;
        const bool anc_ok = !(cur_anc < -1 || cur_anc >= B || (prev < B && prev != cur_anc));
        const int k = lane - l;
        if (k >= 0 && k < run) {
          const uint32_t id = n_nodes + (uint32_t)k, src = k == 0 ? (uint32_t)prev : id - 1u;
          const int ein = (int)(n_edges + (uint32_t)k);
          const bool has_out = k + 1 < run;
          nbase[id] = (uint8_t)tb;
          EdgeG x; x.sink = id; x.w = 1.0f; x.next = -1; x.base = V2 ? ((uint32_t)tb | (src << 8)) : (uint32_t)tb;
          S.stE(ein, x);
          NodeG nn; nn.hw = 0.0f; nn.pred = has_out ? ein + 1 : -1; nn.indeg = V2 ? (uint32_t)(ein + 1) : 0u; nn.head = has_out ? ein + 1 : -1;
          S.stN(id, nn);
          if (k == 0) S.stList((uint32_t)prev, ein, ein);
          if (V2) X.anext[ein] = (has_out && anc_ok) ? ein + 1 : -1;
          if (B - (ref_at + (isX ? 1 : 0)) <= 10 && spr) isend[id] = 1;
        }
        if (V2) {
          if (!anc_ok) tree_ok = false;
          else {
            const int e0 = (int)n_edges, t = X.atail[cur_anc + 1];
            if (t >= 0) X.anext[t] = e0; else X.ahead[cur_anc + 1] = e0;
            X.atail[cur_anc + 1] = e0 + run - 1;
            X.acnt[cur_anc + 1] = X.acnt[cur_anc + 1] + (uint32_t)run;
          }
        }
        n_nodes += (uint32_t)run; n_edges += (uint32_t)run;
        prev = (int)n_nodes - 1;
        cn_id = (uint32_t)prev; cn_head = -1; cn_tail = -1;
        if (run > 1) todo &= ~(((1ull << (run - 1)) - 1ull) << (l + 1));
        pf_follow += (unsigned long long)run;
        return true;
      };
      while (todo && !status) {
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        bool stepped = false;
        if (P.prof) { const int o_ = __builtin_amdgcn_readlane(c, l); pf_sM += o_ == 'M'; pf_sX += o_ == 'X'; pf_sI += o_ == 'I'; pf_sD += o_ == 'D'; }
        const int op = __builtin_amdgcn_readlane(c, l), pop = __builtin_amdgcn_readlane(pc, l);
        const int r = __builtin_amdgcn_readlane(ref_at, l);
        const uint8_t tc = (uint8_t)__builtin_amdgcn_readlane(tb, l);
        bool have_lists = false;
        int ph = -1, pt = -1;
        if (pop == 'M') {
          prev = r - 1; first = false; cur_anc = prev;
          if (r >= 1 && r <= B) { ph = __builtin_amdgcn_readlane(pf_head, l); pt = __builtin_amdgcn_readlane(pf_tail, l); have_lists = true; }
        }
        int ref_after = r;
        if (op == 'M') {
          if (first || prev == r) first = false;
          else if ((uint32_t)prev >= n_nodes || (uint32_t)r >= n_nodes) { status = 3; break; }    // the reference indexes its vectors out of range here
          else {
            if (!((int)prev < B - 1 && r == prev + 1) && !have_lists) node_lists((uint32_t)prev, ph, pt);
            insert_edge((uint32_t)prev, ph, pt, (uint32_t)r);
          }
          prev = r; ref_after = r + 1; cur_anc = prev;
        } else if (op == 'X') {
          if (first) {
            bool need_new = true;
            for (uint32_t q = 0; q < n_start; ++q) if (nbase[starts[q]] == tc) { need_new = false; break; }
            if (need_new) { prev = (int)new_node(tc); starts[n_start++] = (uint32_t)prev; cur_anc = -1; }
            first = false;
          } else if ((uint32_t)prev >= n_nodes) { status = 3; break; }
          else if (create_chain(l)) continue;
          else {
            if (!have_lists) node_lists((uint32_t)prev, ph, pt);
            prev = (int)alt_step((uint32_t)prev, ph, pt, tc);
            stepped = true;
          }
          ref_after = r + 1;
        } else if (op == 'D') {
          ref_after = r + 1;
          if (first) { prev = ref_after; cur_anc = prev; }
        } else if (op == 'I') {
          if (first) { prev = (int)new_node(tc); starts[n_start++] = (uint32_t)prev; first = false; cur_anc = -1; }
          else if ((uint32_t)prev >= n_nodes) { status = 3; break; }
          else if (create_chain(l)) continue;
          else {
            if (!have_lists) node_lists((uint32_t)prev, ph, pt);
            prev = (int)alt_step((uint32_t)prev, ph, pt, tc);
            stepped = true;
          }
        }
        if (B - ref_after <= 10 && spr && (uint32_t)prev < node_cap) isend[prev] = 1;   // ids not yet created are remembered too
        if (stepped && !status) follow_chain(l);
      }
      }
      lastop = __builtin_amdgcn_readlane(c, nvalid - 1);
      if (P.prof) tq_serial += wall_clock64() - ts0;
      ref_i += __builtin_popcountll(mMXD); tgt += __builtin_popcountll(mMXI);
      if (lastop == 'M') { prev = ref_i - 1; first = false; cur_anc = prev; }
      ci += nvalid;
    }
  }
  poa_phase_fence<LDS>();
  if (V2 && !status) {                  // the marks of the plain 'M' stretches become backbone counts
    int carry = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
      const int i = b0 + lane;
      const int d = i < B ? bdiff[i] : 0;
      uint32_t tot = 0;
      const int run = carry + (int)poa_wave_excl_sum((uint32_t)d, lane, &tot) + d;
      if (i < B && run) bbc[i] = bbc[i] + (uint32_t)run;
      carry += (int)tot;
    }
    poa_phase_fence<LDS>();
  }
  if (LDS && S.reduced && (status == 1 || status == 2)) return false;     // outgrew the optimistic LDS capacities
  const unsigned long long t1 = P.prof ? wall_clock64() : 0ull;

  const float c_ = G.c, t_ = G.t;
  auto damp = [&](float w) -> float {                      // adjust_weights :243-252
    const float t_applied = t_ * w;
    const float final_weight = c_ > t_applied ? c_ : t_applied;
    return w - final_weight;
  };
  uint32_t qh = 0, qt = 0;
  if (!status && V2 && tree_ok) {
    // ---- heaviest path, second generation: one pass along the backbone, subtree edges from registers (see the head of this function)
    for (uint32_t i = (uint32_t)lane; i < n_nodes; i += 64) {      // an alt node's source is the source of its one in-edge
      int pr = -1;
      if ((int)i >= B) { const int ie = (int)S.ldN(i).indeg - 1; if (ie >= 0) pr = (int)(S.ldEbase(ie) >> 8); }
      NodeG n; n.hw = 0.0f; n.pred = pr; n.indeg = 0u; n.head = -1;
      S.stN(i, n);
    }
    L.hw[lane] = 0.0f; L.hw[lane + 64] = 0.0f; L.pred[lane] = -1; L.pred[lane + 64] = -1;
    poa_phase_fence<LDS>();
    __builtin_amdgcn_wave_barrier();
    int wb = 0;                                             // the LDS window holds backbone nodes [wb, wb + 128)
    auto relax = [&](uint32_t v, float cand, int u) {       // wave-uniform; same acceptance rule as relax_into below
      if ((int)v < wb + 128) {
        const int ix = (int)(v & 127u);
        const float h = L.hw[ix]; const int pr = L.pred[ix];
        if (pr < 0 || cand > h || (cand == h && u < pr)) { L.hw[ix] = cand; L.pred[ix] = u; }
      } else {                                              // a landing beyond the window (a deletion of more than 64 bases)
        const NodeG nv = S.ldN(v);
        if (nv.pred < 0 || cand > nv.hw || (cand == nv.hw && u < nv.pred)) S.stHwPred(v, cand, u);
      }
    };
    auto window_to = [&](int u) {                            // slide so that u lies in the lower half
      while (u >= wb + 64) {
        const int i0 = wb + lane;
        if (i0 < B) S.stHwPred((uint32_t)i0, L.hw[i0 & 127], L.pred[i0 & 127]);
        wb += 64;
        const int i1 = wb + 64 + lane;
        float h = 0.0f; int pr = -1;
        if (i1 < B) { const NodeG t = S.ldN((uint32_t)i1); h = t.hw; pr = t.pred; }
        L.hw[i1 & 127] = h; L.pred[i1 & 127] = pr;
        poa_lds_order();
      }
    };
    // one block: anchors ai0 .. ai0 + nA - 1 (index = node + 1; 0 = subtrees of start nodes), T <= 64 edges staged in L.stage in list order,
    // lane a < nA knows its anchor's slice [off, off + cnt); bb_step = also take the backbone edge of each anchor
    auto sweep_block = [&](int ai0, int nA, int T, uint32_t off, uint32_t cnt, bool bb_step) {
      uint32_t sink = 0, src = 0; float w = 0.0f, hsg = 0.0f, val = 0.0f; int sp = -1;
      if (lane < T) { const int e = (int)L.stage[lane]; const EdgeG x = S.ldE(e); sink = x.sink; w = x.w; src = x.base >> 8; if ((int)src >= B) sp = -2; }
      for (int k = 0; k < T; ++k) {                          // where in the block does my source get its weight
        const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)sink, k);
        if (sp == -2 && src == sk && k < lane) sp = k;
      }
      if (lane < T && sp == -2) hsg = S.ldHw(src);           // a start node (weight 0) or a node of an earlier block of the same anchor
      float bbw = 0.0f;
      if (lane < nA) { const int u = ai0 + lane - 1; if (u >= 0 && u < B - 1) bbw = (float)bbc[u]; }
      for (int a = 0; a < nA; ++a) {
        const int u = ai0 + a - 1;
        float hw_u = 0.0f;
        if (u >= 0) {
          window_to(u);
          hw_u = L.hw[u & 127];
          if (bb_step && u < B - 1) relax((uint32_t)(u + 1), hw_u + damp(1.0f + poa_readlane_f(bbw, a)), u);
        }
        const int o = __builtin_amdgcn_readlane((int)off, a), cn = __builtin_amdgcn_readlane((int)cnt, a);
        for (int j = o; j < o + cn; ++j) {
          const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)sink, j), sid = (uint32_t)__builtin_amdgcn_readlane((int)src, j);
          const int spk = __builtin_amdgcn_readlane(sp, j);
          const float wk = poa_readlane_f(w, j);
          const float hs = spk == -1 ? hw_u : (spk >= 0 ? poa_readlane_f(val, spk) : poa_readlane_f(hsg, j));
          const float cand = hs + damp(wk);
          if (lane == j) val = cand;
          if ((int)sk < B) relax(sk, cand, (int)sid);
        }
      }
      if (lane < T && (int)sink >= B) S.stHw(sink, val);
    };
    int ai = 0;
    while (ai <= B) {
      const int aidx = ai + lane;
      const uint32_t cnt = aidx <= B ? X.acnt[aidx] : 0u;
      uint32_t tot = 0;
      const uint32_t off = poa_wave_excl_sum(cnt, lane, &tot);
      const unsigned long long fits = __ballot(aidx <= B && off + cnt <= 64u);
      const int nA = (~fits) ? (int)__builtin_ctzll(~fits) : 64;
      if (nA == 0) {
        // one anchor with more than 64 subtree edges: its list in pieces of 64 (weights of nodes of earlier pieces come back from memory)
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 0);
        int e = X.ahead[ai];
        bool first_piece = true;
        for (uint32_t done = 0; done < c0;) {
          const uint32_t T = c0 - done < 64u ? c0 - done : 64u;
          if (lane == 0) for (uint32_t k = 0; k < T; ++k) { L.stage[k] = (uint32_t)e; e = X.anext[e]; }
          e = __builtin_amdgcn_readfirstlane(e);
          poa_lds_order();
          sweep_block(ai, 1, (int)T, 0u, lane == 0 ? T : 0u, first_piece);
          __threadfence();
          first_piece = false; done += T;
        }
        ai += 1;
        continue;
      }
      const int T = __builtin_amdgcn_readlane((int)(off + cnt), nA - 1);
      if (lane < nA) { int e = X.ahead[aidx]; for (uint32_t k = 0; k < cnt; ++k) { L.stage[off + k] = (uint32_t)e; e = X.anext[e]; } }
      poa_lds_order();
      sweep_block(ai, nA, T, off, cnt, true);
      ai += nA;
    }
    // what is left of the window
    for (int h2 = 0; h2 < 2; ++h2) { const int i0 = wb + 64 * h2 + lane; if (i0 < B) S.stHwPred((uint32_t)i0, L.hw[i0 & 127], L.pred[i0 & 127]); }
  } else
  if (!status) {
    // ---- heaviest path: Kahn sweep pushing (weight, source) along out-edges (wave-uniform).  The result does not depend on the
    // order in which ready nodes are taken (max weight, ties to the lowest source id), so the queue is only a work list.
    for (uint32_t i = (uint32_t)lane; i < n_nodes; i += 64) { S.stPred(i, -1); if (V2 && (int)i >= B) S.stIndeg(i, 0u); }      // the list tails (and the in-edge notes of the second generation) have done their job
    poa_phase_fence<LDS>();
    // in-degrees of the extra edges (the backbone in-edge was counted at init)
    if (LDS) { for (uint32_t e = 0; e < n_edges; ++e) S.incIndeg(S.ldE((int)e).sink); }
    else { for (uint32_t e = (uint32_t)lane; e < n_edges; e += 64) S.incIndegAtomic(S.ldE((int)e).sink); }
    poa_phase_fence<LDS>();
    for (uint32_t b0 = 0; b0 < n_nodes; b0 += 64) {
      const uint32_t i = b0 + (uint32_t)lane;
      const bool z = i < n_nodes && S.ldN(i).indeg == 0;
      const unsigned long long zm = __ballot(z);
      if (z) queue[qt + __builtin_popcountll(zm & lt)] = i;
      qt += (uint32_t)__builtin_popcountll(zm);
    }
    poa_phase_fence<LDS>();
    uint32_t pushed_val = 0, pushed_at = 0xffffffffu;      // the latest push, so that a chain never reads the queue back
    NodeG rv; rv.hw = 0.0f; rv.pred = -1; rv.indeg = 0; rv.head = -1;
    uint32_t rv_id = 0xffffffffu;                          // rv = the record last written (node rv_id)
    auto relax_into = [&](uint32_t u, float hw_u, uint32_t v, float w) {
      NodeG nv = v == rv_id ? rv : S.ldN(v);
      const float cand = hw_u + w;
      if (nv.pred < 0) { nv.hw = cand; nv.pred = (int32_t)u; }
      else if (cand > nv.hw || (cand == nv.hw && (int32_t)u < nv.pred)) { nv.hw = cand; nv.pred = (int32_t)u; }
      nv.indeg = nv.indeg - 1u;
      S.stN(v, nv);
      rv = nv; rv_id = v;
      if (nv.indeg == 0) { queue[qt] = v; pushed_val = v; pushed_at = qt; ++qt; }
    };
    while (qh < qt) {
      const uint32_t u = qh == pushed_at ? pushed_val : (uint32_t)queue[qh];
      ++qh;
      const NodeG nu = u == rv_id ? rv : S.ldN(u);
      // NOTE: pred < 0 doubles as "no weight yet" (hdef of the reference); start nodes keep hw = 0 and are never relaxed into
      if ((int)u < B - 1) relax_into(u, nu.hw, u + 1, damp(1.0f + (float)bbc[u]));
      for (int e = nu.head; e >= 0;) {
        const EdgeG x = S.ldE(e);
        relax_into(u, nu.hw, x.sink, damp(x.w));
        e = x.next;
      }
    }
    if (qt != n_nodes) status = 4;    // cycle: the reference would never return
  }
  // ---- pick the end node (:346-367: first strictly heaviest in ascending id) and emit the path (:373-378)
  const unsigned long long t2 = P.prof ? wall_clock64() : 0ull;
  uint32_t len = 0, startpos = 0;
  if (!status && n_nodes > 0) {
    poa_phase_fence<LDS>();
    float bw = 0.0f; uint32_t bi = 0xffffffffu;
    for (uint32_t i = (uint32_t)lane; i < n_nodes; i += 64) {
      if (isend[i]) { const float w = S.ldHw(i); if (bi == 0xffffffffu || w > bw) { bw = w; bi = i; } }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ow = __shfl_xor(bw, off); const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
      if (oi != 0xffffffffu && (bi == 0xffffffffu || ow > bw || (ow == bw && oi < bi))) { bw = ow; bi = oi; }
    }
    const uint32_t h_node = bi == 0xffffffffu ? 0u : bi;
    uint8_t* out = P.out_arena + P.out_off[g];
    const uint32_t cap = out_cap;
    int32_t cur = (int32_t)h_node;
    if (LDS) {
      // walk the predecessor chain in LDS (bases parked in the now idle queue), then one coalesced copy to HBM
      uint32_t k = 0;
      while (cur >= 0 && k < cap && k < node_cap) {
        const uint8_t b = nbase[cur];
        if (b) queue[k++] = b;
        cur = S.ldPred((uint32_t)cur);
      }
      poa_phase_fence<LDS>();
      for (uint32_t j = (uint32_t)lane; j < k; j += 64) out[cap - 1 - j] = (uint8_t)queue[j];
      startpos = cap - k; len = k;
    } else {
      // the path mostly runs down the backbone: (source, base) of 64 consecutive backbone nodes come with one round of loads and are
      // followed through lanes; off the backbone (an alt node) it is one node per round trip
      uint32_t pos = cap;
      while (cur >= 0 && pos > 0) {
        if (cur < B) {
          const int base0 = cur >= 63 ? cur - 63 : 0;
          const int my = base0 + lane;
          int mypred = -1, myb = 0;
          if (my <= cur) { mypred = S.ldPred((uint32_t)my); myb = (int)nbase[my]; }
          while (cur >= base0 && cur < B && pos > 0) {
            const int l = cur - base0;
            const int b = __builtin_amdgcn_readlane(myb, l);
            if (b) out[--pos] = (uint8_t)b;
            cur = __builtin_amdgcn_readlane(mypred, l);
          }
        } else {
          const uint8_t b = nbase[cur];
          if (b) out[--pos] = b;
          cur = S.ldPred((uint32_t)cur);
        }
      }
      startpos = pos; len = cap - pos;
    }
  }
  if (P.prof && lane == 0) {
    const unsigned long long t3 = wall_clock64();
    atomicAdd(P.prof + 0, t1 - t0); atomicAdd(P.prof + 1, t2 - t1); atomicAdd(P.prof + 2, t3 - t2); atomicAdd(P.prof + 3, 1ull);
    atomicAdd(P.prof + 4, (unsigned long long)n_nodes); atomicAdd(P.prof + 5, (unsigned long long)n_edges); atomicAdd(P.prof + 6, (unsigned long long)B);
    if (LDS) atomicAdd(P.prof + 7, 1ull);
    atomicAdd(P.prof + 8, pf_wide); atomicAdd(P.prof + 9, pf_wide_ops); atomicAdd(P.prof + 10, pf_narrow); atomicAdd(P.prof + 11, pf_serial);
    atomicMax(P.prof + 12, t1 - t0); atomicMax(P.prof + 13, t3 - t0);
    if (P.prof_graph) { unsigned long long* q = P.prof_graph + 9 * (size_t)g; q[0] = t1 - t0; q[1] = pf_wide; q[2] = pf_narrow; q[3] = pf_serial; q[4] = tq_serial | (pf_follow << 40); q[5] = (unsigned long long)G.n_members | ((unsigned long long)B << 32); q[6] = pf_sM | (pf_sX << 32); q[7] = pf_sI | (pf_sD << 32); q[8] = pf_snom | (pf_shead << 20) | (pf_sirr << 40); }
    atomicAdd(P.prof + 27, tq_layout); atomicAdd(P.prof + 28, tq_mid); atomicAdd(P.prof + 29, tq_a); atomicAdd(P.prof + 30, tq_turn); atomicAdd(P.prof + 31, tq_c); atomicAdd(P.prof + 20, tq_serial);
  }
  P.out_len[g] = len; P.out_start[g] = startpos; P.status[g] = status;
  return true;
}

// LDS instantiation: one single-wave block per graph (the dispatcher refills the slot as soon as its graph is done); a graph
// that does not fit, or outgrows its optimistic capacities, goes onto the list of the global-memory kernel below.
constexpr uint32_t POA_LDS_NODE_B = 18, POA_LDS_EDGE_B = 8;      // NodeL 12 + bbc 2 + queue 2 + base 1 + end 1; EdgeL 8
__global__ __launch_bounds__(64) void poa_graph_lds_kernel(PoaDev P, const uint32_t* __restrict__ list, uint32_t lds_bytes)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t s_graph[];
  const int lane = threadIdx.x & 63;
  const uint32_t g = list[blockIdx.x];
  if (g >= P.n_graphs) return;
  const otg_poa_graph G = P.graphs[g];
  const uint32_t ncap_true = (uint32_t)(P.node_off[g + 1] - P.node_off[g]), ecap_true = (uint32_t)(P.edge_off[g + 1] - P.edge_off[g]);
  const uint32_t B = G.backbone_len, nm = G.n_members;
  // optimistic need: the backbone + half of the worst-case alt nodes, half of the worst-case edges
  const uint32_t start_b = (2u * (nm + 2u) + 7u) & ~7u;
  bool fits = lds_bytes > start_b + 64u && ncap_true < 32768u && ecap_true < 32768u && nm < 32768u;
  uint32_t N = 0, E = 0;
  bool reduced = false;
  if (fits) {
    const uint32_t avail = lds_bytes - start_b - 32u;
    const float estN = (float)B + 0.5f * (float)(ncap_true - B) + 4.0f, estE = 0.5f * (float)ecap_true + 4.0f;
    const float f = (float)avail / ((float)POA_LDS_NODE_B * estN + (float)POA_LDS_EDGE_B * estE);
    N = (uint32_t)(estN * f) & ~3u; E = (uint32_t)(estE * f) & ~1u;
    if (N >= ncap_true) { N = ncap_true; E = ((avail - POA_LDS_NODE_B * N) / POA_LDS_EDGE_B) & ~1u; }     // node arrays at their bound: edges take the rest
    if (E >= ecap_true) { E = ecap_true; const uint32_t n2 = ((avail - POA_LDS_EDGE_B * E) / POA_LDS_NODE_B) & ~3u; N = n2 < ncap_true ? n2 : ncap_true; }
    reduced = N < ncap_true || E < ecap_true;
    fits = N >= B + 2u && f >= 1.0f;
  }
  if (!fits) {
    if (lane == 0) { const uint32_t k = atomicAdd(P.fb_count, 1u); P.fb_list[k] = g; }
    return;
  }
  PoaStore<true> S;
  uint32_t o = 0;
  S.nodes = (OTG_LDS NodeL*)(s_graph + o); o += 12u * N;
  S.edges = (OTG_LDS EdgeL*)(s_graph + o); o += 8u * E;
  S.bbc = (OTG_LDS uint16_t*)(s_graph + o); o += 2u * N;
  S.queue = (OTG_LDS uint16_t*)(s_graph + o); o += 2u * N;
  S.nbase = (OTG_LDS uint8_t*)(s_graph + o); o += N;
  S.isend = (OTG_LDS uint8_t*)(s_graph + o); o += N;
  o = (o + 3u) & ~3u;
  S.starts = (OTG_LDS uint16_t*)(s_graph + o);
  S.node_cap = N; S.edge_cap = E; S.reduced = reduced;
  if (!poa_graph_body<true>(P, g, S, ncap_true, PoaAux{}, PoaScratch{}, false)) {
    if (lane == 0) { const uint32_t k = atomicAdd(P.fb_count, 1u); P.fb_list[k] = g; }
  }
}

// global-memory instantiation over a list of graphs: those too large for the LDS kernel, then what the LDS kernel left behind
#ifndef OTG_POA_WPEU
#define OTG_POA_WPEU 4      // waves per SIMD the global-memory kernel is compiled for (measured on configs[1], stage ms at 3 / 4 / 5 / 6: 44.8 / 39.5 / 43.4 / 50.2)
#endif
__global__ __launch_bounds__(64, OTG_POA_WPEU) void poa_graph_wave_kernel(PoaDev P, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count_ptr,
                                                               uint32_t count_imm)
{
  __shared__ uint32_t s_ops[64 * POA_WK], s_stage[64];
  __shared__ float s_hw[128];
  __shared__ int32_t s_pred[128];
  const uint32_t n = count_ptr ? *count_ptr : count_imm;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
    const uint32_t g = list[k];
    if (g >= P.n_graphs) continue;
    const uint64_t no = P.wnode_off[g], eo = P.wedge_off[g];
    PoaStore<false> S;
    S.nodes = P.nodes + no; S.edges = P.edges + eo;
    S.nbase = P.node_base + no; S.isend = P.is_end + no; S.bbc = P.bb_cnt + no; S.queue = P.queue + no;
    S.starts = P.start_list + P.wstart_off[g];
    S.node_cap = (uint32_t)(P.node_off[g + 1] - P.node_off[g]); S.edge_cap = (uint32_t)(P.edge_off[g + 1] - P.edge_off[g]); S.reduced = false;
    PoaAux X;
    X.anext = P.anext + eo;
    { const uint64_t ao = P.v2 ? P.wanch_off[g] : 0ull; X.ahead = P.ahead + ao; X.atail = P.atail + ao; X.acnt = P.acnt + ao; }
    PoaScratch L;
    L.ops = (volatile lds_u32*)&s_ops[0]; L.stage = (volatile lds_u32*)&s_stage[0]; L.hw = (volatile lds_f32*)&s_hw[0]; L.pred = (volatile lds_i32*)&s_pred[0];
    poa_graph_body<false>(P, g, S, S.node_cap, X, L, P.v2 != 0);
  }
}

} // namespace

// h_graphs: host copy (layout sizes); d_* device copies.  Outputs (device): consensus g starts at
// SLOT_P17 + node_off[g] + out_start[g] with length d_out_len[g]; out_start is SLOT_P29 (uint32 per graph),
// per-graph status SLOT_P28 (0 ok; 1 node / 2 edge capacity; 3 reference-UB index; 4 cycle).
int otg_launch_poa(otg_ctx* ctx, const uint8_t* d_seq_arena, const uint8_t* d_cig_arena,
                   const otg_poa_member* d_members, uint32_t n_members, const otg_poa_graph* d_graphs,
                   const otg_poa_graph* h_graphs, uint32_t n_graphs, uint32_t* d_out_len,
                   std::vector<uint64_t>& node_off)
{
  if (n_graphs == 0) return OTG_OK;
  // counting pre-pass
  uint32_t* d_alt = (uint32_t*)otg_slot(ctx, SLOT_P20, (size_t)(n_members + 1) * sizeof(uint32_t));
  uint32_t* d_nonm = (uint32_t*)otg_slot(ctx, SLOT_P21, (size_t)(n_members + 1) * sizeof(uint32_t));
  if (!d_alt || !d_nonm) return OTG_ERR_HIP;
  std::vector<uint32_t> h_alt(n_members), h_nonm(n_members);
  if (n_members) {
    hipLaunchKernelGGL(poa_count_kernel, dim3((n_members + 255) / 256), dim3(256), 0, ctx->stream, d_cig_arena, d_members, n_members, d_alt, d_nonm);
    HIP_TRY(ctx, hipMemcpyAsync(h_alt.data(), d_alt, (size_t)n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h_nonm.data(), d_nonm, (size_t)n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<uint64_t> edge_off(n_graphs + 1), start_off(n_graphs + 1);      // capacities as running sums
  node_off.assign(n_graphs + 1, 0);
  node_off[0] = edge_off[0] = start_off[0] = 0;
  for (uint32_t g = 0; g < n_graphs; ++g) {
    uint64_t alt = 0, mrun = 0;
    for (uint32_t m = 0; m < h_graphs[g].n_members; ++m) { alt += h_alt[h_graphs[g].first_member + m]; mrun += h_nonm[h_graphs[g].first_member + m]; }
    node_off[g + 1] = node_off[g] + (((uint64_t)h_graphs[g].backbone_len + alt + 2 + 3) & ~3ull);
    edge_off[g + 1] = edge_off[g] + ((alt + mrun + 2 + 1) & ~1ull);
    start_off[g + 1] = start_off[g] + h_graphs[g].n_members + 2;
  }
  const uint64_t NN = node_off[n_graphs];
  PoaDev P;
  P.seq_arena = d_seq_arena; P.cig_arena = d_cig_arena; P.members = d_members; P.graphs = d_graphs; P.n_graphs = n_graphs;
  static const bool poa_v1 = getenv("OTG_POA_V1") != nullptr;       // first generation only (serial threading, Kahn sweep)
  P.v2 = poa_v1 ? 0 : 1;
  for (uint32_t g = 0; g < n_graphs && P.v2; ++g)      // node ids share a word with a base in the edge records
    if (node_off[g + 1] - node_off[g] >= (1ull << 24) - 2) P.v2 = 0;
  // Graph slots without backbone and members (the pipeline keeps one slot per read) produce nothing: zero their outputs here and
  // leave them out of the launches.  The others are split by size: LDS bytes per graph block = the optimistic need (same formula
  // as poa_graph_lds_kernel) of 97 % of them; graphs within it go to the LDS kernel, the rest to the global-memory kernel.  Above
  // 16 KB per block too few graphs would be resident per CU to beat the 32 latency-bound waves of the global-memory kernel.
  static const bool no_lds = getenv("OTG_POA_NO_LDS") != nullptr;
  uint32_t lds_bytes = 0, n_lds = 0, n_glob = 0;
  std::vector<uint32_t> order;
  {
    std::vector<uint32_t> est(n_graphs, 0u), live;
    live.reserve(n_graphs);
    for (uint32_t g = 0; g < n_graphs; ++g) {
      if (h_graphs[g].backbone_len == 0 && h_graphs[g].n_members == 0) continue;
      const double B = (double)h_graphs[g].backbone_len, ncap = (double)(node_off[g + 1] - node_off[g]), ecap = (double)(edge_off[g + 1] - edge_off[g]);
      const double need = (double)POA_LDS_NODE_B * (B + 0.5 * (ncap - B) + 4.0) + (double)POA_LDS_EDGE_B * (0.5 * ecap + 4.0) + (double)((2u * (h_graphs[g].n_members + 2u) + 7u) & ~7u) + 32.0 + 64.0;
      est[g] = need > 4.0e9 ? 0xffffffffu : (uint32_t)need;
      live.push_back(g);
    }
    if (!live.empty() && !no_lds) {
      std::vector<uint32_t> e2(live.size());
      for (size_t i = 0; i < live.size(); ++i) e2[i] = est[live[i]];
      const size_t k = (size_t)((double)(live.size() - 1) * 0.97);
      std::nth_element(e2.begin(), e2.begin() + k, e2.end());
      if (e2[k] <= 16u * 1024u) lds_bytes = std::max<uint32_t>((e2[k] + 511u) & ~511u, 4096u);
    }
    // longest graphs first within each launch (work ~ op-string bytes = nodes + edges capacity computed above)
    auto heavier = [&](uint32_t a, uint32_t b) {
      const uint64_t wa = (edge_off[a + 1] - edge_off[a]) + (node_off[a + 1] - node_off[a]), wb = (edge_off[b + 1] - edge_off[b]) + (node_off[b + 1] - node_off[b]);
      return wa != wb ? wa > wb : a < b;
    };
    order.reserve(live.size());
    for (uint32_t g : live) if (lds_bytes && est[g] <= lds_bytes) order.push_back(g);
    n_lds = (uint32_t)order.size();
    for (uint32_t g : live) if (!(lds_bytes && est[g] <= lds_bytes)) order.push_back(g);
    n_glob = (uint32_t)order.size() - n_lds;
    std::sort(order.begin(), order.begin() + n_lds, heavier);
    std::sort(order.begin() + n_lds, order.end(), heavier);
  }
  // The graph images are sized for the worst case (every X / I op a new node), ~0.5 MB per allele of a 3 kb locus: a batch of 100 000 regions
  // would ask for 150 GB at once.  So the launch lists are cut into pieces whose images fit a fixed budget; the pieces run one after the
  // other on the stream and reuse the same work arrays (a piece of 24 GB holds tens of thousands of graphs, several times what the device
  // keeps in flight).  Only the consensus arena and the per-graph outputs span the whole batch.
  static const size_t piece_budget_cfg = getenv("OTG_POA_PIECE_MB") ? (size_t)atoll(getenv("OTG_POA_PIECE_MB")) << 20 : (size_t)24 << 30;
  size_t piece_budget = piece_budget_cfg;
  {   // ... and never more than half of what the device can still give (what the work arrays hold already counts as available)
    size_t free_b = 0, total_b = 0, held = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    for (int sl : {SLOT_P3, SLOT_P4, SLOT_P6, SLOT_P7, SLOT_P8, SLOT_P9, SLOT_P10, SLOT_P11, SLOT_P12, SLOT_P13, SLOT_P16}) held += ctx->pool[sl].cap;
    piece_budget = std::min(piece_budget, std::max<size_t>((size_t)1 << 30, (free_b + held) / 2));
  }
  struct Piece { uint32_t k0, k1; bool lds; };
  std::vector<Piece> pieces;
  std::vector<uint64_t> wnode(n_graphs + 1, 0), wedge(n_graphs + 1, 0), wstart(n_graphs + 1, 0), wanch(n_graphs + 1, 0);
  uint64_t maxN = 0, maxE = 0, maxS = 0, maxA = 0;
  {
    auto cut = [&](uint32_t k0, uint32_t k1, bool lds) {
      uint32_t p0 = k0;
      uint64_t cn = 0, ce = 0, cs = 0, ca = 0;
      for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t g = order[k];
        const uint64_t nc = node_off[g + 1] - node_off[g], ec = edge_off[g + 1] - edge_off[g], sc = start_off[g + 1] - start_off[g];
        const uint64_t ac = ((uint64_t)h_graphs[g].backbone_len + 2 + 3) & ~3ull;
        const uint64_t bytes = (cn + nc) * (sizeof(NodeG) + 10) + (ce + ec) * (sizeof(EdgeG) + 4) + (cs + sc) * 4 + (ca + ac) * 12;
        if (k > p0 && bytes > piece_budget) { pieces.push_back({p0, k, lds}); p0 = k; cn = ce = cs = ca = 0; }
        wnode[g] = cn; wedge[g] = ce; wstart[g] = cs; wanch[g] = ca;
        cn += nc; ce += ec; cs += sc; ca += ac;
        maxN = std::max(maxN, cn); maxE = std::max(maxE, ce); maxS = std::max(maxS, cs); maxA = std::max(maxA, ca);
      }
      if (k1 > p0) pieces.push_back({p0, k1, lds});
    };
    cut(n_lds, n_lds + n_glob, false);
    cut(0, n_lds, true);
  }
  uint64_t* d_node_off = (uint64_t*)otg_slot(ctx, SLOT_P0, (size_t)(n_graphs + 1) * sizeof(uint64_t));
  uint64_t* d_edge_off = (uint64_t*)otg_slot(ctx, SLOT_P1, (size_t)(n_graphs + 1) * sizeof(uint64_t));
  uint64_t* d_woff = (uint64_t*)otg_slot(ctx, SLOT_P2, (size_t)4 * (n_graphs + 1) * sizeof(uint64_t));
  P.node_base = (uint8_t*)otg_slot(ctx, SLOT_P3, maxN);
  P.is_end = (uint8_t*)otg_slot(ctx, SLOT_P4, maxN);
  P.nodes = (NodeG*)otg_slot(ctx, SLOT_P6, maxN * sizeof(NodeG));
  P.bb_cnt = (uint32_t*)otg_slot(ctx, SLOT_P9, maxN * 4);
  P.queue = (uint32_t*)otg_slot(ctx, SLOT_P12, maxN * 4);
  P.edges = (EdgeG*)otg_slot(ctx, SLOT_P13, maxE * sizeof(EdgeG));
  P.start_list = (uint32_t*)otg_slot(ctx, SLOT_P16, maxS * 4);
  P.out_arena = (uint8_t*)otg_slot(ctx, SLOT_P17, NN);
  P.out_start = (uint32_t*)otg_slot(ctx, SLOT_P29, (size_t)n_graphs * 4);
  P.status = (int32_t*)otg_slot(ctx, SLOT_P28, (size_t)n_graphs * 4);
  P.anext = nullptr; P.ahead = nullptr; P.atail = nullptr; P.acnt = nullptr;
  if (P.v2) {
    P.anext = (int32_t*)otg_slot(ctx, SLOT_P7, maxE * 4);
    P.ahead = (int32_t*)otg_slot(ctx, SLOT_P8, maxA * 4);
    P.atail = (int32_t*)otg_slot(ctx, SLOT_P10, maxA * 4);
    P.acnt = (uint32_t*)otg_slot(ctx, SLOT_P11, maxA * 4);
    if (!P.anext || !P.ahead || !P.atail || !P.acnt) return OTG_ERR_HIP;
  }
  uint32_t* d_order = (uint32_t*)otg_slot(ctx, SLOT_P18, (size_t)(n_graphs + 1) * sizeof(uint32_t));
  if (!d_node_off || !d_edge_off || !d_woff || !P.node_base || !P.is_end || !P.nodes || !P.bb_cnt || !P.queue ||
      !P.edges || !P.start_list || !P.out_arena || !P.out_start || !P.status || !d_order)
    return OTG_ERR_HIP;
  P.node_off = d_node_off; P.edge_off = d_edge_off;
  P.wnode_off = d_woff; P.wedge_off = d_woff + (n_graphs + 1); P.wstart_off = d_woff + 2 * (size_t)(n_graphs + 1); P.wanch_off = d_woff + 3 * (size_t)(n_graphs + 1);
  P.out_off = d_node_off;     // consensus g is written into [node_off[g], node_off[g+1]) of out_arena
  P.out_len = d_out_len;
  P.order = d_order;
  if (!order.empty()) HIP_TRY(ctx, hipMemcpyAsync(d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_out_len, 0, (size_t)n_graphs * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(P.out_start, 0, (size_t)n_graphs * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(P.status, 0, (size_t)n_graphs * sizeof(int32_t), ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_node_off, node_off.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_edge_off, edge_off.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_woff, wnode.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_woff + (n_graphs + 1), wedge.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_woff + 2 * (size_t)(n_graphs + 1), wstart.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_woff + 3 * (size_t)(n_graphs + 1), wanch.data(), (size_t)(n_graphs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host vectors go out of scope after return
  static const bool profile = getenv("OTG_POA_PROFILE") != nullptr;
  P.prof = nullptr; P.prof_graph = nullptr;
  if (profile) {
    P.prof_graph = (unsigned long long*)otg_slot(ctx, SLOT_P23, (size_t)n_graphs * 9 * sizeof(unsigned long long));
    if (!P.prof_graph) return OTG_ERR_HIP;
    HIP_TRY(ctx, hipMemsetAsync(P.prof_graph, 0, (size_t)n_graphs * 9 * sizeof(unsigned long long), ctx->stream));
    P.prof = (unsigned long long*)otg_slot(ctx, SLOT_P19, 32 * sizeof(unsigned long long));
    if (!P.prof) return OTG_ERR_HIP;
    HIP_TRY(ctx, hipMemsetAsync(P.prof, 0, 32 * sizeof(unsigned long long), ctx->stream));
  }
  P.fb_list = (uint32_t*)otg_slot(ctx, SLOT_P22, (size_t)(n_graphs + 1) * sizeof(uint32_t));
  if (!P.fb_list) return OTG_ERR_HIP;
  P.fb_count = P.fb_list + n_graphs;
  for (const Piece& pc : pieces) {
    const uint32_t n = pc.k1 - pc.k0;
    // one single-wave block per graph: the dispatcher refills a wave slot as soon as its graph is done
    if (!pc.lds) hipLaunchKernelGGL(poa_graph_wave_kernel, dim3(n), dim3(64), 0, ctx->stream, P, P.order + pc.k0, (const uint32_t*)nullptr, n);
    else {
      HIP_TRY(ctx, hipMemsetAsync(P.fb_count, 0, sizeof(uint32_t), ctx->stream));
      hipLaunchKernelGGL(poa_graph_lds_kernel, dim3(n), dim3(64), lds_bytes, ctx->stream, P, P.order + pc.k0, lds_bytes);
      const uint32_t fg = n < (uint32_t)ctx->n_cu * 32 ? n : (uint32_t)ctx->n_cu * 32;
      hipLaunchKernelGGL(poa_graph_wave_kernel, dim3(fg), dim3(64), 0, ctx->stream, P, (const uint32_t*)P.fb_list, (const uint32_t*)P.fb_count, 0u);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  if (profile) {
    unsigned long long h[32];
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(h, P.prof, sizeof(h), hipMemcpyDeviceToHost));
    const double n = h[3] ? (double)h[3] : 1.0;     // wall_clock64: 100 MHz
    fprintf(stderr, "[otg] poa profile: %u LDS + %u global launches; %llu graphs done (%llu in LDS, %u bytes each), per graph: insert %.1f us, sweep %.1f us, emit %.1f us; nodes %.0f (backbone %.0f), edges %.0f\n",
            n_lds, n_glob, h[3], h[7], lds_bytes, h[0] / n / 100.0, h[1] / n / 100.0, h[2] / n / 100.0, h[4] / n, h[6] / n, h[5] / n);
    fprintf(stderr, "[otg] poa profile: threading chunks per graph: %.1f wide (%.0f ops each), %.1f narrow one-lane-per-run, %.1f serial; slowest graph: insert %.1f us, whole %.1f us\n",
            h[8] / n, h[8] ? (double)h[9] / (double)h[8] : 0.0, h[10] / n, h[11] / n, h[12] / 100.0, h[13] / 100.0);
    { const double nw = h[8] ? (double)h[8] : 1.0;
      fprintf(stderr, "[otg] poa profile: a wide chunk, us: layout %.2f, bases + marks + staging %.2f, subtree walk %.2f, ids %.2f, chains %.2f; narrow + serial code per graph %.1f us\n",
              h[27] / nw / 100.0, h[28] / nw / 100.0, h[29] / nw / 100.0, h[30] / nw / 100.0, h[31] / nw / 100.0, h[20] / n / 100.0); }
    {
      std::vector<unsigned long long> pg((size_t)n_graphs * 9);
      HIP_TRY(ctx, hipMemcpy(pg.data(), P.prof_graph, pg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<uint32_t> idx(n_graphs);
      for (uint32_t g = 0; g < n_graphs; ++g) idx[g] = g;
      const size_t top = std::min<size_t>(5, n_graphs);
      std::partial_sort(idx.begin(), idx.begin() + top, idx.end(), [&](uint32_t a, uint32_t b) { return pg[9 * (size_t)a] > pg[9 * (size_t)b]; });
      for (size_t t = 0; t < top; ++t) {
        const unsigned long long* q = &pg[9 * (size_t)idx[t]];
        fprintf(stderr, "[otg] poa profile: slow graph %u: %llu members on a backbone of %llu, threading %.1f us = %llu wide + %llu narrow + %llu serial chunks; %.1f us in the narrow + serial code (%llu steps taken as a retraced chain)\n",
                idx[t], q[5] & 0xffffffffull, q[5] >> 32, q[0] / 100.0, q[1], q[2], q[3], (q[4] & ((1ull << 40) - 1ull)) / 100.0, q[4] >> 40);
        fprintf(stderr, "[otg] poa profile:   its serial code: %llu windows without an 'M', %llu member heads, %llu others; ops taken one by one: %llu M, %llu X, %llu I, %llu D\n",
                q[8] & 0xfffffull, (q[8] >> 20) & 0xfffffull, q[8] >> 40, q[6] & 0xffffffffull, q[6] >> 32, q[7] & 0xffffffffull, q[7] >> 32);
      }
    }
  }
  return OTG_OK;
}
