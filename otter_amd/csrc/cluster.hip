// cluster.hip — per-region KDE cut-height selection + average-linkage clustering + coverage repair (gfx950).
//
// Replaces otter_hclust (reference: src/otterclust.cpp:118-320) with everything it calls:
//   otter_find_clustering_dist (src/otterclust.cpp:20-116), KDE::f / KDE::maximas (src/ankde.cpp:8-62),
//   hclust_fast(AVERAGE) = NN_chain_core (include/hclust-cpp/fastcluster_dm.hpp:563-766) +
//   generate_R_dendrogram<false> (fastcluster_R_dm.hpp:68-115), cutree_cdist / cutree_k (fastcluster.cpp:33-105).
//
// One 256-thread workgroup per region.  FP64 throughout, compiled with -ffp-contract=off; every sum keeps
// the reference's order (one thread per KDE grid point looping over the distances in index order;
// normalisation and extremum scan sequential).  exp() is glibc's algorithm restated (table + degree-5
// polynomial) in the variant — FMA or not — that the host libm uses, so densities are bit-identical to the
// reference running on this host.  The integer-valued outputs (labels, ic, fc) are decided by
// the same comparisons on the same bits.
#include "otg_common.hpp"
#include <cmath>

#define OTG_EXP_TAB_QUAL __device__ __constant__ const
#include "exp_table.inc"

namespace {

constexpr int NMAX = 256;      // max valid reads per region handled on chip (reference default max_cov = 200)
constexpr int GMAX = 512;      // max KDE grid points
constexpr int DLDS = 2016;     // dist working copy kept in LDS when n <= 64

__device__ __constant__ double c_grid[GMAX];

struct ClusterArgs {
  int max_alleles;
  int bandwidth_length, min_cov_fraction2_l;
  double bandwidth_short, bandwidth_long, max_error, min_cov_fraction, min_cov_fraction2_f;
  double inv_sqrt_2pi, inv_h_short, inv_h_long;
  int n_grid, radius, exp_fma;
  double dinterval;
};

__device__ __forceinline__ uint64_t d2u(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

// glibc 2.28+ exp(), N = 128 (sysdeps/ieee754/dbl-64/e_exp.c), restated.  FMA=true mirrors the x86-64
// ifunc-selected FMA build (gcc contracts r, tmp and the final scale+scale*tmp; the subnormal path is not
// contracted) — verified bit-for-bit against libm on 4e7 arguments on the build host.
template <bool FMA>
__device__ double otg_exp(double x)
{
  const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double Shift = 0x1.8p52;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  uint32_t abstop = (uint32_t)(d2u(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
    if (abstop >= 0x409u) {
      if (d2u(x) == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      return (d2u(x) >> 63) ? 0.0 : u2d(0x7ff0000000000000ull);
    }
    abstop = 0;
  }
  const double z = InvLn2N * x;
  double kd = z + Shift;
  const uint64_t ki = d2u(kd);
  kd -= Shift;
  double r;
  if (FMA) r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
  else r = x + kd * NegLn2hiN + kd * NegLn2loN;
  const uint64_t idx = 2 * (ki % 128), top = ki << 45;
  const double tail = u2d(OTG_EXP_TAB[idx]);
  uint64_t sbits = OTG_EXP_TAB[idx + 1] + top;
  const double r2 = r * r;
  double tmp;
  if (FMA) tmp = fma(r2 * r2, fma(r, C5, C4), fma(r2, fma(r, C3, C2), tail + r));
  else tmp = tail + r + r2 * (C2 + r * C3) + r2 * r2 * (C4 + r * C5);
  if (abstop == 0) {
    double scale, y;
    if ((ki & 0x80000000ull) == 0) {
      sbits -= 1009ull << 52; scale = u2d(sbits);
      y = 0x1p1009 * (FMA ? fma(scale, tmp, scale) : scale + scale * tmp);
      return y;
    }
    sbits += 1022ull << 52; scale = u2d(sbits);
    const double st = scale * tmp;
    y = scale + st;
    if (y < 1.0) {
      double hi, lo;
      lo = scale - y + st; hi = 1.0 + y; lo = 1.0 - hi + y + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  const double scale = u2d(sbits);
  return FMA ? fma(scale, tmp, scale) : scale + scale * tmp;
}

__device__ __forceinline__ size_t didx(int N, int r, int c) { return (size_t)((((long long)(2 * N - 3 - r)) * r) >> 1) + c - 1; } // r < c
__device__ __forceinline__ double dget(const double* D, int N, int i, int j) { return i < j ? D[didx(N, i, j)] : D[didx(N, j, i)]; }

// ---- sequential pieces (run by thread 0) ---------------------------------------------------------------

// libstdc++ std::sort on an int array with the reference's non-strict-weak comparator
// (src/otterclust.cpp:61-66): introsort without the heapsort fallback (returns false if it would be needed).
struct MaxCmp {
  const int* mi; const double* mv;
  __device__ bool operator()(int a, int b) const {
    double diff = mv[a] - mv[b];
    diff = diff > 0 ? diff : -diff;
    if (diff <= 0.01) return mi[a] < mi[b];
    return mv[a] > mv[b];
  }
};

__device__ void ins_sort_guarded(int* f, int first, int last, const MaxCmp& cmp)
{
  for (int i = first + 1; i < last; ++i) {
    if (cmp(f[i], f[first])) {
      int val = f[i];
      for (int q = i; q > first; --q) f[q] = f[q - 1];
      f[first] = val;
    } else {
      int val = f[i], l = i, nx = i - 1;
      while (cmp(val, f[nx])) { f[l] = f[nx]; l = nx; --nx; }
      f[l] = val;
    }
  }
}

__device__ bool std_sort_emul(int* f, int n, const MaxCmp& cmp)
{
  if (n <= 1) return true;
  if (n > 16) {
    int lg = 31 - __clz(n);
    int st_first[64], st_last[64], st_depth[64];
    int sp = 0;
    st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg; sp = 1;
    while (sp > 0) {
      --sp;
      int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
      while (last - first > 16) {
        if (depth == 0) return false;
        --depth;
        int mid = first + (last - first) / 2;
        int a = first + 1, b = mid, c = last - 1, res = first;
        auto sw = [&](int x, int y) { int t = f[x]; f[x] = f[y]; f[y] = t; };
        if (cmp(f[a], f[b])) { if (cmp(f[b], f[c])) sw(res, b); else if (cmp(f[a], f[c])) sw(res, c); else sw(res, a); }
        else if (cmp(f[a], f[c])) sw(res, a);
        else if (cmp(f[b], f[c])) sw(res, c);
        else sw(res, b);
        int lo = first + 1, hi = last;
        for (;;) {
          while (cmp(f[lo], f[first])) ++lo;
          --hi;
          while (cmp(f[first], f[hi])) --hi;
          if (!(lo < hi)) break;
          sw(lo, hi);
          ++lo;
        }
        // recurse on [lo, last), iterate on [first, lo)
        if (sp >= 63) return false;
        st_first[sp] = lo; st_last[sp] = last; st_depth[sp] = depth; ++sp;
        last = lo;
      }
    }
    ins_sort_guarded(f, 0, 16, cmp);
    for (int i = 16; i < n; ++i) {
      int val = f[i], l = i, nx = i - 1;
      while (cmp(val, f[nx])) { f[l] = f[nx]; l = nx; --nx; }
      f[l] = val;
    }
    return true;
  }
  ins_sort_guarded(f, 0, n, cmp);
  return true;
}

// The recursion order of libstdc++ is depth-first on the RIGHT part first ([cut,last) recursive call,
// then the loop continues on [first,cut)); partitions are disjoint, so processing order does not change
// the result.

struct Lds {
  double dens[GMAX];
  double sums[GMAX];
  double members[NMAX];
  double zdist[NMAX];
  double height[NMAX];
  double dwork[DLDS];
  int maxi[GMAX / 2 + 2], mini[GMAX / 2 + 2];
  double maxv[GMAX / 2 + 2], minv[GMAX / 2 + 2];
  int sorted[GMAX / 2 + 2];
  int nn_chain[NMAX], succ[NMAX + 1], pred[NMAX + 1];
  int z1[NMAX], z2[NMAX], zrank[NMAX];
  int parent[2 * NMAX];
  int merge[2 * NMAX];
  int labels[NMAX], labels2[NMAX];
  int ct_up[NMAX + 1], ct_own[NMAX], ct_first[NMAX + 1], ct_lab[NMAX + 1];     // cutree_wave scratch
  int cnt[NMAX], maxsz[NMAX], req[NMAX], remap[NMAX];
  double total;
  int n_max, n_min, err;
  int do_hclust;
  int cut_k, recut, ic, fc;
  double b0, b1, bc, dist_final, bandwidth;
};

// ---- nn_chain_average below restates, operation for operation, the NN-chain average-linkage core of the vendored hclust-cpp / fastcluster
// the reference links (include/hclust-cpp/fastcluster_dm.hpp:563-766): merge order, tie-breaks and the floating-point evaluation order of
// `s*a + t*b` decide the labels, so the order of operations is kept.  That code carries this notice (BSD 2-clause, include/hclust-cpp/LICENSE):
//
//   fastcluster: Fast hierarchical clustering routines for R and Python.  Copyright (c) 2011 Daniel Müllner <http://danifold.net>.
//   C++ standalone version (hclust-cpp): Copyright Christoph Dalitz, 2020; Daniel Müllner, 2011.  All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
//   conditions are met: (1) redistributions of source code must retain the above copyright notice, this list of conditions and the
//   following disclaimer; (2) redistributions in binary form must reproduce the above copyright notice, this list of conditions and the
//   following disclaimer in the documentation and/or other materials provided with the distribution.
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT
//   LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL THE
//   COPYRIGHT HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
//   (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION)
//   HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE)
//   ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Tree cut into `nclust` clusters by ONE WAVE, with the numbering the reference's cutree_k produces (include/hclust-cpp/fastcluster.cpp:33-81:
// clusters are numbered in the order in which their first observation appears, :69-80).  Own formulation: the first K = n - nclust rows of the
// R-style merge matrix are a forest over the merge steps (a row names its two children: -j = observation j, +m = the cluster made by step m),
//   1. every step scatters itself as the parent of its children (`up` for steps, `own` for observations) — one pass, all lanes;
//   2. pointer jumping turns `up` into the root of every step's tree (a parent always has a larger step number: <= 8 rounds for 256 steps);
//   3. every tree learns its lowest observation (LDS atomic min), observations that are that lowest one — or singletons — are the
//      first appearances; a ballot + prefix count over them in index order hands out the labels; the rest copy their tree's label.
// Call from all 64 lanes of a wave with uniform arguments.
__device__ void cutree_wave(int n, const int* merge, int nclust, int* labels, int* up, int* own, int* firstobs, int* clab, int lane)
{
  if (nclust > n || nclust < 2) { for (int j = lane; j < n; j += 64) labels[j] = 0; wave_sync(); return; }
  const int K = n - nclust;
  for (int k = lane; k <= K; k += 64) { up[k] = 0; firstobs[k] = n; }
  for (int j = lane; j < n; j += 64) own[j] = 0;
  wave_sync();
  for (int k = lane + 1; k <= K; k += 64) {
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int m = merge[(k - 1) + side * (n - 1)];
      if (m < 0) own[-m - 1] = k; else up[m] = k;
    }
  }
  wave_sync();
  for (int d = 1; d < K; d <<= 1) {
    for (int k = lane + 1; k <= K; k += 64) { const int u = up[k]; if (u) { const int uu = up[u]; if (uu) up[k] = uu; } }
    wave_sync();
  }
  for (int j = lane; j < n; j += 64) {
    const int o = own[j];
    if (o) { const int c = up[o] ? up[o] : o; own[j] = c; atomicMin(&firstobs[c], j); }
  }
  wave_sync();
  int base = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const int c = j < n ? own[j] : 0;
    const bool first = j < n && (c == 0 || firstobs[c] == j);
    const unsigned long long fm = __ballot(first);
    const int rank = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
    if (first) { if (c) clab[c] = rank; else labels[j] = rank; }
    base += __builtin_popcountll(fm);
  }
  wave_sync();
  for (int j = lane; j < n; j += 64) { const int c = own[j]; if (c) labels[j] = clab[c]; }
  wave_sync();
}

// NN-chain average linkage run by ONE WAVE (64 lanes, uniform control flow; called by wave 0 of the block).  The chain logic is the
// reference's (restart when the chain tip is <= 3, merged cluster keeps the larger index, sizes as doubles); the three O(V) loops of every
// step — nearest neighbour of the chain tip, and the distance update `s*a + t*b` — run across the lanes: lane l owns the indices l, l + 64,
// ...  A nearest-neighbour search is a min-reduction over (distance, index) with ties to the LOWEST index, which is what the reference's
// ascending strict-'<' scan returns (fastcluster_dm.hpp:612-640); the incumbent keeps its place on a tie, as there.  Same doubles, same
// comparisons, same update expression per element => same merges.  L.pred doubles as the active flag here.
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
template <class S>
__device__ void nn_chain_average(int N, double* D, S& L, int lane)
{
#define D_(r_, c_) (D[didx(N, (r_), (c_))])
  constexpr double INF = 1.0e300;
  int* act = L.pred;                                  // 1 = still a cluster of its own
  for (int i = lane; i < N; i += 64) { act[i] = 1; L.members[i] = 1.0; }
  wave_sync();
  // lowest-index nearest neighbour of node `a` among the active nodes i with i > lo_excl, i != a; (INF, N) when there is none
  auto nearest = [&](int a, int lo_excl, double& dmin, int& imin) {
    double bd = INF; int bi = N;
    for (int i = lane; i < N; i += 64) {
      if (i > lo_excl && i != a && act[i]) {
        const double d = i < a ? D_(i, a) : D_(a, i);
        if (d < bd) { bd = d; bi = i; }                 // ascending per lane: keeps the lane's lowest index
      }
    }
    dmin = wave_min_f64(bd);
    imin = wave_min_i32(bd == dmin ? bi : N);
  };
  int start = 0, tip = 0, idx1 = 0, idx2 = 0;
  double mn = 0;
  for (int j = 0; j < N - 1; ++j) {
    if (tip <= 3) {
      if (lane == 0) L.nn_chain[0] = start;
      idx1 = start;
      tip = 1;
      nearest(idx1, idx1, mn, idx2);                    // first strict minimum over the active nodes after `start`
    } else {
      tip -= 3;
      idx1 = L.nn_chain[tip - 1];
      idx2 = L.nn_chain[tip];
      mn = idx1 < idx2 ? D_(idx1, idx2) : D_(idx2, idx1);
    }
    do {
      if (lane == 0) L.nn_chain[tip] = idx2;
      wave_sync();
      double dm; int im;
      nearest(idx2, -1, dm, im);                        // over every active node but idx2; the incumbent idx1 stays on a tie
      if (dm < mn) { mn = dm; idx1 = im; }
      idx2 = idx1;
      idx1 = L.nn_chain[tip++];
    } while (idx2 != L.nn_chain[tip - 2]);
    if (lane == 0) { L.z1[j] = idx1; L.z2[j] = idx2; L.zdist[j] = mn; }
    if (idx1 > idx2) { const int t = idx1; idx1 = idx2; idx2 = t; }
    const double size1 = L.members[idx1], size2 = L.members[idx2];
    wave_sync();
    if (lane == 0) { L.members[idx2] = size2 + size1; act[idx1] = 0; }
    wave_sync();
    if (idx1 == start) { int c = N; for (int i = lane; i < N; i += 64) if (act[i] && i < c) c = i; start = wave_min_i32(c); }
    const double s = size1 / (size1 + size2), t = size2 / (size1 + size2);
    for (int i = lane; i < N; i += 64) {
      if (!act[i] || i == idx2) continue;
      if (i < idx1) D_(i, idx2) = s * D_(i, idx1) + t * D_(i, idx2);
      else if (i < idx2) D_(i, idx2) = s * D_(idx1, i) + t * D_(i, idx2);
      else D_(idx2, i) = s * D_(idx1, i) + t * D_(idx2, i);
    }
    wave_sync();
  }
#undef D_
}

// hclust_fast(AVERAGE): NN-chain (wave 0) + generate_R_dendrogram<false> (stable sort by height = rank by
// (dist, position), computed in parallel; union-find relabel by thread 0).  Block-cooperative: call from all
// threads.  Leaves L.merge (R convention, column-major) and L.height.
template <class S>
__device__ void hclust_to_merge(int n, double* D, S& L, int tid)
{
  if (tid < 64) nn_chain_average(n, D, L, tid);
  __syncthreads();
  for (int i = tid; i < n - 1; i += blockDim.x) {
    int rank = 0;
    const double di = L.zdist[i];
    for (int j = 0; j < n - 1; ++j) { const double dj = L.zdist[j]; if (dj < di || (dj == di && j < i)) ++rank; }
    L.zrank[rank] = i;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < 2 * n - 1; ++i) L.parent[i] = 0;
    int nextparent = n;
    for (int k = 0; k < n - 1; ++k) {
      const int src = L.zrank[k];
      int a = L.z1[src], b = L.z2[src];
      // union_find::Find with path compression (fastcluster_dm.hpp:366-383)
      for (int w = 0; w < 2; ++w) {
        int idx = w ? b : a;
        if (L.parent[idx] != 0) {
          int p = idx;
          idx = L.parent[idx];
          if (L.parent[idx] != 0) {
            do { idx = L.parent[idx]; } while (L.parent[idx] != 0);
            do { int tmp = L.parent[p]; L.parent[p] = idx; p = tmp; } while (L.parent[p] != idx);
          }
        }
        if (w) b = idx; else a = idx;
      }
      L.parent[a] = L.parent[b] = nextparent++;
      if (a > b) { int t = a; a = b; b = t; }
      L.merge[k] = (a < n) ? -a - 1 : a - n + 1;
      L.merge[k + n - 1] = (b < n) ? -b - 1 : b - n + 1;
      L.height[k] = L.zdist[src];
    }
  }
  __syncthreads();
}

template <bool FMA>
__global__ __launch_bounds__(256) void cluster_kernel(
    ClusterArgs A, const double* __restrict__ dist, const uint64_t* __restrict__ dist_off,
    const uint32_t* __restrict__ read_len, const uint64_t* __restrict__ len_off, const uint32_t* __restrict__ n_valid,
    uint32_t n_regions, double* __restrict__ gwork,
    int32_t* __restrict__ labels_out, int32_t* __restrict__ ic_out, int32_t* __restrict__ fc_out,
    double* __restrict__ bounds_out, int32_t* __restrict__ err_out)
{
  __shared__ Lds L;
  const int tid = threadIdx.x;
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
    const int n = (int)n_valid[r];
    const double* dv = dist + dist_off[r];
    const uint32_t* lens = read_len + len_off[r];
    int32_t* lab = labels_out + len_off[r];
    const size_t npairs = (size_t)n * (size_t)(n - 1) / 2;
    __syncthreads();
    if (tid == 0) {
      L.err = 0; L.do_hclust = 0; L.b0 = L.b1 = L.bc = __longlong_as_double(0x7ff8000000000000ll);
    }
    __syncthreads();
    // ---- trivial cases (src/otterclust.cpp:121-156)
    if (n <= 2 || A.max_alleles == 1 || n > NMAX) {
      if (tid == 0) {
        int ic = 0, fc = 0;
        if (n > NMAX) { L.err = 10; }
        else if (n == 1) { lab[0] = 0; ic = fc = 1; }
        else if (n == 2) {
          lab[0] = 0; lab[1] = 0; ic = fc = 1;
          if (A.max_alleles != 1 && !(dv[0] <= A.max_error)) { lab[1] = 1; ic = fc = 2; }
        } else if (n > 2) { for (int i = 0; i < n; ++i) lab[i] = 0; ic = fc = 1; }
        ic_out[r] = ic; fc_out[r] = fc;
        if (bounds_out) { bounds_out[3 * r] = L.b0; bounds_out[3 * r + 1] = L.b1; bounds_out[3 * r + 2] = L.bc; }
        if (err_out) err_out[r] = L.err;
      }
      continue;
    }
    // ---- bandwidth (:162-168)
    if (tid == 0) {
      double bw = A.bandwidth_short;
      for (int i = 0; i < n; ++i) if ((int)lens[i] >= A.bandwidth_length) { bw = A.bandwidth_long; break; }
      L.bandwidth = bw;
    }
    __syncthreads();
    const double h = L.bandwidth;
    const double inv_h = (h == A.bandwidth_long && h != A.bandwidth_short) ? A.inv_h_long : (h == A.bandwidth_short ? A.inv_h_short : 1 / h);
    // ---- KDE::f on the grid (src/ankde.cpp:8-23; src/otterclust.cpp:26-28): one thread per grid point
    for (int p = tid; p < A.n_grid; p += blockDim.x) {
      const double x = c_grid[p];
      double total = 0.0;
      for (size_t q = 0; q < npairs; ++q) {
        const double y = x - dv[q];
        const double zq = y / h;
        const double e = otg_exp<FMA>(-(zq * zq / 2));
        const double kk = A.inv_sqrt_2pi * e;
        total += inv_h * kk;
      }
      L.dens[p] = total / (double)npairs;
    }
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int p = 0; p < A.n_grid; ++p) t += L.dens[p]; L.total = t; }   // :29-30
    __syncthreads();
    for (int p = tid; p < A.n_grid; p += blockDim.x) L.dens[p] = L.dens[p] / L.total;                     // :31-34
    __syncthreads();
    // ---- KDE::maximas (src/ankde.cpp:25-62): windowed sums in parallel, extremum scan sequential
    for (int i = tid; i < A.n_grid; i += blockDim.x) {
      double sum = 0.0;
      sum += L.dens[i];
      for (int j = 1; j < A.radius && (i - j) >= 0; ++j) sum += L.dens[i - j];
      for (int j = 1; j < A.radius && (i + j) < A.n_grid; ++j) sum += L.dens[i + j];
      L.sums[i] = sum;
    }
    __syncthreads();
    if (tid == 0) {
      bool find_maxima = true;
      double last_sum = 0.0;
      int last_sum_i = 1, nmx = 0, nmn = 0;
      for (int i = 1; i < A.n_grid - 1; ++i) {
        const double sum = L.sums[i];
        if (find_maxima) { if (sum < last_sum) { find_maxima = false; L.maxi[nmx] = last_sum_i; L.maxv[nmx] = last_sum; ++nmx; } }
        else { if (sum > last_sum) { find_maxima = true; L.mini[nmn] = last_sum_i; L.minv[nmn] = last_sum; ++nmn; } }
        last_sum = sum; last_sum_i = i;
      }
      if (find_maxima) { L.maxi[nmx] = last_sum_i; L.maxv[nmx] = last_sum; ++nmx; }
      L.n_max = nmx; L.n_min = nmn;
      // ---- decision bound (src/otterclust.cpp:39-115)
      const double di = A.dinterval;
      if (nmx == 0) L.err = 1;
      else if (nmx == 1) { L.b0 = L.maxi[0] * di; L.b1 = L.maxi[0] * di; L.bc = -1.0; }
      else if (nmn == 0) L.err = 2;
      else if (nmx == 2) { L.b0 = L.maxi[0] * di; L.b1 = L.maxi[1] * di; L.bc = L.mini[0] * di; }
      else {
        int ns = nmx;
        for (int i = 0; i < ns; ++i) L.sorted[i] = i;
        MaxCmp cmp{L.maxi, L.maxv};
        if (!std_sort_emul(L.sorted, ns, cmp)) L.err = 5;
        int last_i = 0, acc_i = 1;
        while (acc_i < ns) {
          int index_diff = acc_i > last_i ? acc_i - last_i : last_i - acc_i;
          double f_diff = L.maxv[L.sorted[acc_i]] - L.maxv[L.sorted[last_i]];
          f_diff = f_diff < 0 ? -f_diff : f_diff;
          if (index_diff == 1 && f_diff <= 0.01) {
            for (int q = acc_i; q + 1 < ns; ++q) L.sorted[q] = L.sorted[q + 1];
            --ns;
            last_i = acc_i;
          }
          ++acc_i;
        }
        if (ns < 2) { L.b0 = L.maxi[0] * di; L.b1 = L.maxi[1] * di; L.bc = L.mini[0] * di; }
        else {
          int m1 = L.sorted[0], m2 = L.sorted[1];
          if (m1 > m2) { int t = m1; m1 = m2; m2 = t; }
          int boundary_i = m2 - 1;
          if (boundary_i < 0 || boundary_i >= nmn) L.err = 3;
          else {
            if (m2 - m1 > 1 && m2 - 2 >= 0 && (L.maxi[m2] * di - L.mini[boundary_i] * di <= 0.01)) {
              boundary_i = m2 - 2;
              if (boundary_i < 0 || boundary_i >= nmn) L.err = 4;
            }
            if (!L.err) { L.b0 = L.maxi[m1] * di; L.b1 = L.maxi[m2] * di; L.bc = L.mini[m1 + (m2 - m1) / 2] * di; }
          }
        }
      }
      if (!L.err) {
        if (L.b1 - L.b0 <= A.max_error) L.do_hclust = 0;   // :172-176
        else { L.do_hclust = 1; L.dist_final = (L.b1 == L.bandwidth) ? L.b1 : L.bc + 0.0025; }   // :184
      }
    }
    __syncthreads();
    if (L.err || !L.do_hclust) {
      for (int i = tid; i < n; i += blockDim.x) lab[i] = L.err ? -1 : 0;
      if (tid == 0) {
        ic_out[r] = L.err ? 0 : 1; fc_out[r] = L.err ? 0 : 1;
        if (bounds_out) { bounds_out[3 * r] = L.b0; bounds_out[3 * r + 1] = L.b1; bounds_out[3 * r + 2] = L.bc; }
        if (err_out) err_out[r] = L.err;
      }
      continue;
    }
    // ---- hclust_fast(AVERAGE) on a working copy (:181-182)
    double* D = (npairs <= (size_t)DLDS) ? L.dwork : gwork + dist_off[r];
    for (size_t q = tid; q < npairs; q += blockDim.x) D[q] = dv[q];
    __syncthreads();
    hclust_to_merge(n, D, L, tid);
    if (tid == 0) {
      // cutree_cdist (:185; fastcluster.cpp:95-105): cut below the first merge whose height reaches dist_final
      int kc;
      for (kc = 0; kc < (n - 1); kc++) if (L.height[kc] >= L.dist_final) break;
      L.cut_k = n - kc;
    }
    __syncthreads();
    if (tid < 64) cutree_wave(n, L.merge, L.cut_k, L.labels, L.ct_up, L.ct_own, L.ct_first, L.ct_lab, tid);
    __syncthreads();
    if (tid == 0) {
      int total_alleles = 0;
      for (int i = 0; i < n; ++i) if (L.labels[i] > total_alleles) total_alleles = L.labels[i];
      ++total_alleles;
      int ic = total_alleles, fc = 0, recut = 0;
      const int min_cov1 = (int)(n * A.min_cov_fraction + 0.5);
      const int min_cov2 = (int)(n * A.min_cov_fraction2_f + 0.5);
      if (A.max_alleles != 0) {
        for (int l = 0; l < total_alleles; ++l) { L.cnt[l] = 0; L.maxsz[l] = 0; }
        for (int i = 0; i < n; ++i) { ++L.cnt[L.labels[i]]; if ((int)lens[i] > L.maxsz[L.labels[i]]) L.maxsz[L.labels[i]] = (int)lens[i]; }
        for (int l = 0; l < total_alleles; ++l) L.req[l] = (L.maxsz[l] < A.min_cov_fraction2_l) ? min_cov1 : min_cov2;
        bool only_single = true;
        for (int l = 0; l < total_alleles; ++l) if (L.cnt[l] >= L.req[l]) { only_single = false; break; }
        int seeds = 0;
        for (int l = 0; l < total_alleles; ++l) if (!(L.cnt[l] < L.req[l])) ++seeds;
        if (only_single || seeds == 0 || seeds > A.max_alleles) {
          recut = 1; fc = A.max_alleles;              // cutree_k(max_alleles) replaces the labels (:200-212, :243-249)
        } else {
          int sj = 0;
          for (int l = 0; l < total_alleles; ++l) L.remap[l] = (L.cnt[l] < L.req[l]) ? -1 : sj++;
          for (int i = 0; i < n; ++i) L.labels[i] = L.remap[L.labels[i]];
          for (int i = 0; i < n; ++i) {
            if (L.labels[i] == -1) {
              int closest_j = 0;
              double min_dist = 100000.0;
              for (int j = 0; j < n; ++j) {
                if (i != j && L.labels[j] != -1) {
                  const double jd = dget(dv, n, i, j);
                  if (jd < min_dist) { closest_j = j; min_dist = jd; }
                }
              }
              L.labels[i] = L.labels[closest_j];
            }
          }
          fc = seeds;
        }
      }
      L.ic = ic; L.fc = fc; L.recut = recut;
    }
    __syncthreads();
    if (L.recut) {
      if (tid < 64) cutree_wave(n, L.merge, A.max_alleles, L.labels, L.ct_up, L.ct_own, L.ct_first, L.ct_lab, tid);
      __syncthreads();
    }
    for (int i = tid; i < n; i += blockDim.x) lab[i] = L.labels[i];
    if (tid == 0) {
      ic_out[r] = L.ic; fc_out[r] = L.fc;
      if (bounds_out) { bounds_out[3 * r] = L.b0; bounds_out[3 * r + 1] = L.b1; bounds_out[3 * r + 2] = L.bc; }
      if (err_out) err_out[r] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// anallele_cluster (reference: src/otterclust.cpp:463-527) for `otter genotype`: length-ratio matrix
// (:322-327,367-382) and 3-mer-usage cosine matrix rounded to 3 decimals (:384-420; KUSAGE src/anseqs.cpp:111-147,
// seq2kcounts :149-166), two average-linkage cuts (cluter_to_e :329-349), genotype = distinct (gt_l, gt_k) pairs in
// first-seen order, representative = medoid under the LENGTH matrix (:517-524).  One 256-thread workgroup per region, every
// phase spread over the block; what the reference sums in a fixed order (the 65-term norm and dot products, the Hill-Shannon sum, a
// medoid's row sum) is summed in that order by ONE thread per output element, and different elements run in parallel:
//   1. 3-mer counts: one WAVE per allele (alleles dealt round-robin to the four waves), lanes stride the sequence (coalesced byte
//      loads), every lane counts into its own column of an LDS histogram [bin][lane] (16-bit, conflict-free), a rotated column walk
//      sums the 64 columns per bin;
//   2. frequencies, norm, Hill-Shannon diversity: one thread per allele;
//   3. the two matrices: the A(A-1)/2 pairs dealt to the threads (flat pair index);
//   4. the two clusterings one after the other: NN-chain by wave 0 on an LDS working copy of the matrix (regions of up to 102 alleles), rank
//      sort of the merges by all threads, union-find relabel by one thread, tree cut by wave 0;
//   5. genotypes = first appearances of (gt_l, gt_k): one thread per allele looks for an earlier allele with its pair, ballot + prefix
//      count numbers the first appearances; medoids: one thread per allele sums its row over its genotype (ascending, as the reference),
//      one thread per genotype takes the first strict minimum.
constexpr int GT_DLDS = 5152;      // 102 alleles: BASELINE configs[3] has 101
struct HcScratch {
  double members[NMAX], zdist[NMAX], height[NMAX];
  int nn_chain[NMAX], pred[NMAX + 1];
  int z1[NMAX], z2[NMAX], zrank[NMAX];
  int parent[2 * NMAX], merge[2 * NMAX];
  int labels[NMAX];
  int ct_up[NMAX + 1], ct_own[NMAX], ct_first[NMAX + 1], ct_lab[NMAX + 1];
  int cut_k;
};
struct GLds {
  union {
    uint16_t hist[4][65 * 64];      // phase 1: per wave, [bin][lane]
    HcScratch hc;                   // phase 4
  } u;
  int lab_l[NMAX], lab_k[NMAX], gtlab[NMAX], firstof[NMAX], nfirst[8];
  double rowsum[NMAX];
  double dmat[GT_DLDS];             // working copy of a condensed matrix while the NN-chain runs on it (regions of up to 102 alleles)
};

// union-find relabel of the NN-chain output into R's merge matrix (generate_R_dendrogram<false>, fastcluster_R_dm.hpp:68-115), one thread
template <class S>
__device__ void dendrogram_relabel(int n, S& L)
{
  for (int i = 0; i < 2 * n - 1; ++i) L.parent[i] = 0;
  int nextparent = n;
  for (int k = 0; k < n - 1; ++k) {
    const int src = L.zrank[k];
    int a = L.z1[src], b = L.z2[src];
    for (int w = 0; w < 2; ++w) {      // union_find::Find with path compression (fastcluster_dm.hpp:366-383)
      int idx = w ? b : a;
      if (L.parent[idx] != 0) {
        int p = idx;
        idx = L.parent[idx];
        if (L.parent[idx] != 0) {
          do { idx = L.parent[idx]; } while (L.parent[idx] != 0);
          do { int tmp = L.parent[p]; L.parent[p] = idx; p = tmp; } while (L.parent[p] != idx);
        }
      }
      if (w) b = idx; else a = idx;
    }
    L.parent[a] = L.parent[b] = nextparent++;
    if (a > b) { int t = a; a = b; b = t; }
    L.merge[k] = (a < n) ? -a - 1 : a - n + 1;
    L.merge[k + n - 1] = (b < n) ? -b - 1 : b - n + 1;
    L.height[k] = L.zdist[src];
  }
}

__global__ __launch_bounds__(256) void genotype_kernel(
    double max_error_l, double max_error_c, const uint8_t* __restrict__ arena, const uint64_t* __restrict__ seq_off,
    const uint32_t* __restrict__ seq_len, const uint32_t* __restrict__ first_allele, const uint32_t* __restrict__ n_alleles,
    uint32_t n_regions, const uint64_t* __restrict__ pair_off, double* __restrict__ g_dl, double* __restrict__ g_dk,
    double* __restrict__ g_work, double* __restrict__ g_kvec, double* __restrict__ g_vnorm,
    int32_t* __restrict__ gt, int32_t* __restrict__ gt_l, int32_t* __restrict__ gt_k, double* __restrict__ hsd,
    int32_t* __restrict__ n_gt, int32_t* __restrict__ reps, int32_t* __restrict__ err_out)
{
  __shared__ GLds L;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
    const int A = (int)n_alleles[r];
    const uint32_t f = first_allele[r];
    __syncthreads();
    if (A == 0) { if (tid == 0) { n_gt[r] = 0; err_out[r] = 0; } continue; }
    if (A > NMAX) { if (tid == 0) { n_gt[r] = 0; err_out[r] = 10; } continue; }
    double* dl = g_dl + pair_off[r]; double* dk = g_dk + pair_off[r]; double* wk = g_work + pair_off[r];
    double* kv = g_kvec + (size_t)f * 65; double* vn = g_vnorm + f;
    const size_t npairs = (size_t)A * (A - 1) / 2;
    const bool in_lds = npairs <= (size_t)GT_DLDS;
    // ---- 1. 3-mer counts (seq2kcounts, src/anseqs.cpp:149-166): 64 bins + one bin for 3-mers holding a byte outside ACGT (either case)
    {
      uint16_t* H = &L.u.hist[wv][0];
      for (int a = wv; a < A; a += 4) {
        for (int q = lane; q < 65 * 32; q += 64) ((uint32_t*)H)[q] = 0u;
        wave_sync();
        const uint8_t* s = arena + seq_off[f + a];
        const int n = (int)seq_len[f + a];
        // tiles of 1024 bases: a lane takes 16 consecutive positions + the two bases after them from ONE 16-byte load and one 2-byte load
        // (the arena ends in 64 bytes of slack), i.e. 16 three-mers per round trip instead of one
        for (int t0 = 0; t0 + 3 <= n; t0 += 1024) {
          const int j0 = t0 + 16 * lane;
          if (j0 + 3 <= n) {
            uint8_t b[18];
            __builtin_memcpy(b, s + j0, 16);
            __builtin_memcpy(b + 16, s + j0 + 16, 2);
            int code[18];
#pragma unroll
            for (int q = 0; q < 18; ++q) {
              const uint8_t ch = b[q];
              code[q] = (ch == 'A' || ch == 'a') ? 0 : (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : 4;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if (j0 + q + 3 <= n) {
                const bool ok = code[q] != 4 && code[q + 1] != 4 && code[q + 2] != 4;
                const int idx = 16 * (code[q] & 3) + 4 * (code[q + 1] & 3) + (code[q + 2] & 3);
                H[(ok ? idx : 64) * 64 + lane] += 1;        // the lane's own column: no atomics, no bank conflicts (two lanes share a dword)
              }
            }
          }
        }
        wave_sync();
        // column sums: lane q walks the 64 columns of bin q starting at its own index (rotated: distinct banks across the lanes)
        {
          uint32_t c0 = 0, c64 = 0;
          for (int j = 0; j < 64; ++j) c0 += H[lane * 64 + ((j + lane) & 63)];
          c64 = H[64 * 64 + lane];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) c64 += __shfl_xor(c64, off);
          double* v = kv + (size_t)a * 65;
          v[lane] = (double)c0;
          if (lane == 0) v[64] = (double)c64;
        }
        wave_sync();
      }
    }
    __syncthreads();
    // ---- 2. KUSAGE per allele (src/anseqs.cpp:111-147): frequencies, norm, Hill-Shannon diversity — sequential over the 65 bins
    for (int a = tid; a < A; a += blockDim.x) {
      double* v = kv + (size_t)a * 65;
      int total_counts = 0;
      for (int q = 0; q < 65; ++q) total_counts += v[q];            // int += double (src/anseqs.cpp:113-114)
      double norm = 0;
      for (int q = 0; q < 65; ++q) { const double value = v[q] / total_counts; v[q] = value; norm += value * value; }
      vn[a] = sqrt(norm);
      double acc = 0;                                                // hsdiv (:135-147), FP tolerance field
      for (int q = 0; q < 65; ++q) if (v[q] > 0) acc += (v[q] * log(v[q]));
      acc = -1 * acc;
      hsd[f + a] = pow(2.718281828459045235360287471352662498, acc);
    }
    __syncthreads();
    if (A == 1) { if (tid == 0) { gt[f] = gt_l[f] = gt_k[f] = 0; reps[f] = 0; n_gt[r] = 1; err_out[r] = 0; } continue; }
    // ---- 3. the two matrices, one pair per thread and pass
    for (size_t p = tid; p < npairs; p += blockDim.x) {
      // row i of the condensed index p: off(i) = i (2A - i - 1) / 2 <= p < off(i + 1)
      int i = (int)((2.0 * A - 1.0 - sqrt((2.0 * A - 1.0) * (2.0 * A - 1.0) - 8.0 * (double)p)) * 0.5);
      if (i < 0) i = 0;
      if (i > A - 2) i = A - 2;
      while (i > 0 && (size_t)i * (2 * A - i - 1) / 2 > p) --i;
      while ((size_t)(i + 1) * (2 * A - i - 2) / 2 <= p) ++i;
      const int j = (int)(p - (size_t)i * (2 * A - i - 1) / 2) + i + 1;
      const uint32_t x = seq_len[f + i], y = seq_len[f + j];
      const bool xs = x < y;
      double d = xs ? (double)(y - x) : (double)(x - y);
      d = xs ? d / y : d / x;
      dl[p] = d;
      if (in_lds) L.dmat[p] = d; else wk[p] = d;
      const double* vi = kv + (size_t)i * 65; const double* vj = kv + (size_t)j * 65;
      double dot = 0;
      for (int q = 0; q < 65; ++q) dot += vi[q] * vj[q];
      const double cs = dot / (vn[i] * vn[j]);
      const bool nan_norm = (vn[i] != vn[i]) || (vn[j] != vn[j]);
      dk[p] = 1.0 - (nan_norm ? 0 : (round(cs * 1000.0) / 1000.0));
    }
    __syncthreads();
    // ---- 4. the two clusterings, one after the other on one scratch: NN-chain on wave 0 (for regions of up to 102 alleles on the LDS working copy:
    // every nearest-neighbour scan and update at LDS latency), rank sort of the merges by all threads, union-find relabel by one thread, tree
    // cut by wave 0
    auto cluster_one = [&](double* D, double cut, int* labels_out) {
      HcScratch& S = L.u.hc;
      if (wv == 0) nn_chain_average(A, D, S, lane);
      __syncthreads();
      for (int i = tid; i < A - 1; i += blockDim.x) {        // stable sort by height = rank by (distance, position) (fastcluster_R_dm.hpp:74)
        int rank = 0;
        const double di = S.zdist[i];
        for (int j = 0; j < A - 1; ++j) { const double dj = S.zdist[j]; if (dj < di || (dj == di && j < i)) ++rank; }
        S.zrank[rank] = i;
      }
      __syncthreads();
      if (tid == 0) {
        dendrogram_relabel(A, S);
        int kc;                                              // cutree_cdist: below the first merge whose height reaches the threshold
        for (kc = 0; kc < (A - 1); kc++) if (S.height[kc] >= cut) break;
        S.cut_k = A - kc;
      }
      __syncthreads();
      if (wv == 0) cutree_wave(A, S.merge, S.cut_k, S.labels, S.ct_up, S.ct_own, S.ct_first, S.ct_lab, lane);
      __syncthreads();
      for (int a2 = tid; a2 < A; a2 += blockDim.x) labels_out[a2] = S.labels[a2];
      __syncthreads();
    };
    cluster_one(in_lds ? L.dmat : wk, max_error_l, L.lab_l);
    if (in_lds) { for (size_t q = tid; q < npairs; q += blockDim.x) L.dmat[q] = dk[q]; __syncthreads(); }
    cluster_one(in_lds ? L.dmat : dk, max_error_c, L.lab_k);
    // ---- 5. genotypes: distinct (gt_l, gt_k) pairs numbered by first appearance (:500-516)
    const int* lab_l = L.lab_l; const int* lab_k = L.lab_k;
    for (int a = tid; a < A; a += blockDim.x) {
      gt_l[f + a] = lab_l[a]; gt_k[f + a] = lab_k[a];
      int fo = a;
      for (int b2 = 0; b2 < a; ++b2) if (lab_l[b2] == lab_l[a] && lab_k[b2] == lab_k[a]) { fo = b2; break; }
      L.firstof[a] = fo;
    }
    __syncthreads();
    {   // number the first appearances in index order: per 64-chunk ballot + prefix count, chunk totals through LDS
      const bool isf = tid < A && L.firstof[tid < A ? tid : 0] == tid;
      const unsigned long long fm = __ballot(isf);
      if (lane == 0) L.nfirst[wv] = __builtin_popcountll(fm);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wv; ++w) base += L.nfirst[w];
      if (isf) L.gtlab[tid] = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
      __syncthreads();
    }
    const int ng = L.nfirst[0] + L.nfirst[1] + L.nfirst[2] + L.nfirst[3];
    for (int a = tid; a < A; a += blockDim.x) { const int g2 = L.gtlab[L.firstof[a]]; gt[f + a] = g2; reps[f + a] = -1; }
    __syncthreads();
    // medoid of every genotype under the length matrix (:517-524): row sums over the genotype in ascending order, first strict minimum
    for (int a = tid; a < A; a += blockDim.x) {
      const int fa = L.firstof[a];
      double sm = 0.0;
      for (int b2 = 0; b2 < A; ++b2) if (b2 != a && L.firstof[b2] == fa) sm += dget(dl, A, a, b2);
      L.rowsum[a] = sm;
    }
    __syncthreads();
    for (int a = tid; a < A; a += blockDim.x) {
      if (L.firstof[a] != a) continue;                        // one thread per genotype: the one of its first allele
      int min_i = -1; double min_sum = 100000000.0;
      for (int b2 = a; b2 < A; ++b2) {
        if (L.firstof[b2] != a) continue;
        if (min_i < 0) min_i = b2;
        if (L.rowsum[b2] < min_sum) { min_i = b2; min_sum = L.rowsum[b2]; }
      }
      reps[f + L.gtlab[a]] = min_i;
    }
    if (tid == 0) { n_gt[r] = ng; err_out[r] = 0; }
  }
}

} // namespace

int otg_launch_cluster(otg_ctx* ctx, const otg_params* P, const double* d_dist, const uint64_t* d_dist_off,
                       const uint32_t* d_read_len, const uint64_t* d_len_off, const uint32_t* d_n_valid,
                       uint32_t n_regions, int32_t* d_labels, int32_t* d_ic, int32_t* d_fc, double* d_bounds,
                       int32_t* d_err)
{
  if (n_regions == 0) return OTG_OK;
  ClusterArgs A;
  A.max_alleles = P->max_alleles;
  A.bandwidth_length = P->bandwidth_length;
  A.min_cov_fraction2_l = P->min_cov_fraction2_l;
  A.bandwidth_short = P->bandwidth_short; A.bandwidth_long = P->bandwidth_long;
  A.max_error = P->max_error; A.min_cov_fraction = P->min_cov_fraction; A.min_cov_fraction2_f = P->min_cov_fraction2_f;
  {
    // constants the reference evaluates at run time (src/ankde.cpp:10,15), in the same operation order
    volatile double pi = 3.14159265358979323846;
    volatile double two_pi = 2 * pi;
    A.inv_sqrt_2pi = 1 / std::sqrt(two_pi);
    volatile double hs = P->bandwidth_short, hl = P->bandwidth_long;
    A.inv_h_short = 1 / hs; A.inv_h_long = 1 / hl;
  }
  const double error_intervals = 0.0025;
  int radius = int(P->max_error / error_intervals);     // src/otterclust.cpp:160-161
  A.radius = radius < 1 ? 1 : radius;
  A.dinterval = error_intervals;
  A.exp_fma = ctx->exp_variant;
  static double grid[GMAX];
  int ng = 0;
  for (volatile double x = 0.0; x <= 1.0; x += error_intervals) {     // src/otterclust.cpp:26
    if (ng >= GMAX) return otg_fail(ctx, OTG_ERR_ARG, "KDE grid too large");
    grid[ng++] = x;
  }
  A.n_grid = ng;
  HIP_TRY(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_grid), grid, sizeof(double) * ng, 0, hipMemcpyHostToDevice, ctx->stream));
  // working copy for regions whose matrix does not fit LDS: same layout as d_dist
  // (size unknown here -> callers guarantee SLOT_AUX9 holds at least as many doubles as d_dist)
  double* gwork = (double*)ctx->pool[SLOT_AUX9].p;
  if (!gwork) return otg_fail(ctx, OTG_ERR_ARG, "cluster workspace (SLOT_AUX9) not allocated");
  uint32_t grid_dim = n_regions < (uint32_t)ctx->n_cu * 8 ? n_regions : (uint32_t)ctx->n_cu * 8;
  if (A.exp_fma)
    hipLaunchKernelGGL((cluster_kernel<true>), dim3(grid_dim), dim3(256), 0, ctx->stream, A, d_dist, d_dist_off, d_read_len, d_len_off,
                       d_n_valid, n_regions, gwork, d_labels, d_ic, d_fc, d_bounds, d_err);
  else
    hipLaunchKernelGGL((cluster_kernel<false>), dim3(grid_dim), dim3(256), 0, ctx->stream, A, d_dist, d_dist_off, d_read_len, d_len_off,
                       d_n_valid, n_regions, gwork, d_labels, d_ic, d_fc, d_bounds, d_err);
  HIP_TRY(ctx, hipGetLastError());
  return OTG_OK;
}

int otg_launch_genotype(otg_ctx* ctx, const otg_params* P, const uint8_t* d_arena, const uint64_t* d_seq_off,
                        const uint32_t* d_seq_len, const uint32_t* d_first, const uint32_t* d_n, uint32_t n_regions,
                        const uint64_t* d_pair_off, uint64_t n_pairs_total, uint64_t n_alleles_total,
                        int32_t* d_gt, int32_t* d_gtl, int32_t* d_gtk, double* d_hsd, int32_t* d_ngt, int32_t* d_reps,
                        int32_t* d_err)
{
  if (n_regions == 0) return OTG_OK;
  double* g_dl = (double*)otg_slot(ctx, SLOT_P20, (n_pairs_total + 1) * 8);
  double* g_dk = (double*)otg_slot(ctx, SLOT_P21, (n_pairs_total + 1) * 8);
  double* g_wk = (double*)otg_slot(ctx, SLOT_P22, (n_pairs_total + 1) * 8);
  double* g_kv = (double*)otg_slot(ctx, SLOT_P23, (n_alleles_total + 1) * 65 * 8);
  double* g_vn = (double*)otg_slot(ctx, SLOT_P24, (n_alleles_total + 1) * 8);
  if (!g_dl || !g_dk || !g_wk || !g_kv || !g_vn) return OTG_ERR_HIP;
  uint32_t grid_dim = n_regions < (uint32_t)ctx->n_cu * 8 ? n_regions : (uint32_t)ctx->n_cu * 8;
  hipLaunchKernelGGL(genotype_kernel, dim3(grid_dim), dim3(256), 0, ctx->stream, P->gt_max_error, P->gt_max_cosdis, d_arena, d_seq_off,
                     d_seq_len, d_first, d_n, n_regions, d_pair_off, g_dl, g_dk, g_wk, g_kv, g_vn, d_gt, d_gtl, d_gtk, d_hsd, d_ngt, d_reps, d_err);
  HIP_TRY(ctx, hipGetLastError());
  return OTG_OK;
}
