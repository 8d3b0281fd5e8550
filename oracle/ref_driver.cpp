/*
 * ref_driver.cpp — thin extern "C" driver around the REFERENCE'S OWN, UNMODIFIED sources, compiled
 * from where they lie under /root/reference (never copied into this repo):
 *     src/andistmat.cpp, src/ankde.cpp, include/hclust-cpp/fastcluster.cpp, src/anppoa.hpp (header-only)
 * Output goes to oracle/_ref/libotter_ref.so (git-ignored, travels to the GPU box).
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/otter_oracle.cpp and (optionally) as cpu_baseline
 * kind "reference" for the consensus step.  Files of the reference that include the absent
 * WFA2-lib header (analignments.cpp, otterclust.cpp, assemble.cpp) are NOT buildable here and are
 * not part of this library (no stand-in header is written).
 */
#include "andistmat.hpp"
#include "ankde.hpp"
#include "fastcluster.h"
#include "anppoa.hpp"
#include "../include/otter_gpu.h"

#include <cstring>
#include <string>
#include <vector>

extern "C" {

uint32_t ref_medoid(uint32_t n, const double* dist, const uint32_t* ind, uint32_t n_ind)
{
  DistMatrix dm(n);
  memcpy(dm.values.data(), dist, dm.values.size() * sizeof(double));
  std::vector<uint32_t> v(ind, ind + n_ind);
  return dm.get_medoid(v);
}

/* set_dist/get_dist round trip: fills an n x n matrix via set_dist(i,j,val[i*n+j]) for i<j and returns values */
void ref_distmatrix_layout(uint32_t n, const double* full, double* condensed_out)
{
  DistMatrix dm(n);
  for (uint32_t i = 0; i < n; ++i) for (uint32_t j = i + 1; j < n; ++j) dm.set_dist(i, j, full[(size_t)i * n + j]);
  memcpy(condensed_out, dm.values.data(), dm.values.size() * sizeof(double));
}

double ref_kde_f(double h, const double* values, uint64_t n, double x)
{
  KDE kde(h);
  kde.values.assign(values, values + n);
  return kde.f(x);
}

int ref_kde_maximas(int radius, const double* dens, int n, int* max_i, double* max_v, int* n_max, int* min_i, double* min_v, int* n_min)
{
  KDE kde(0.01);
  std::vector<double> d(dens, dens + n);
  std::vector<std::pair<int, double>> mx, mn;
  kde.maximas(radius, d, mx, mn);
  for (size_t i = 0; i < mx.size(); ++i) { max_i[i] = mx[i].first; max_v[i] = mx[i].second; }
  for (size_t i = 0; i < mn.size(); ++i) { min_i[i] = mn[i].first; min_v[i] = mn[i].second; }
  *n_max = mx.size(); *n_min = mn.size();
  return 0;
}

int ref_hclust_average(int n, const double* dist, int* merge, double* height)
{
  std::vector<double> cpy(dist, dist + (size_t)n * (n - 1) / 2);
  return hclust_fast(n, cpy.data(), HCLUST_METHOD_AVERAGE, merge, height);
}
void ref_cutree_k(int n, const int* merge, int nclust, int* labels) { cutree_k(n, merge, nclust, labels); }
void ref_cutree_cdist(int n, const int* merge, double* height, double cdist, int* labels) { cutree_cdist(n, merge, height, cdist, labels); }

int ref_poa_consensus_batch(const uint8_t* seq_arena, uint64_t, const uint8_t* cig_arena, uint64_t,
                            const otg_poa_member* members, uint32_t, const otg_poa_graph* graphs, uint32_t n_graphs,
                            uint64_t* out_off, uint32_t* out_len, uint8_t* out_arena, uint64_t cap, uint64_t* used)
{
  uint64_t pos = 0; int rc = 0;
  for (uint32_t g = 0; g < n_graphs; ++g) {
    const otg_poa_graph& G = graphs[g];
    PPOA poa;
    std::string bb((const char*)seq_arena + G.backbone_off, G.backbone_len);
    poa.init(bb);
    for (uint32_t m = 0; m < G.n_members; ++m) {
      const otg_poa_member& mm = members[G.first_member + m];
      std::string seq((const char*)seq_arena + mm.seq_off, mm.seq_len);
      std::string cig((const char*)cig_arena + mm.cigar_off, mm.cigar_len);
      bool l = mm.spanning_l, r = mm.spanning_r;
      poa.insert_alignment(seq, cig, l, r);
    }
    poa.adjust_weights(G.c, G.t);
    std::string cons;
    poa.consensus(cons);
    out_off[g] = pos; out_len[g] = cons.size();
    if (pos + cons.size() <= cap) memcpy(out_arena + pos, cons.data(), cons.size()); else rc = OTG_ERR_CAPACITY;
    pos += cons.size();
  }
  if (used) *used = pos;
  return rc;
}

/* One allele graph through the reference's own PPOA (quadratic heaviest path, src/anppoa.hpp:254-288): the consensus hook of the
 * oracle pipeline (oto_set_poa_hook) — bench.py's cpu_baseline "reference_consensus" (BASELINE.md §3 baseline A).  Thread-safe: the
 * PPOA instance is local.  Returns the consensus length (the caller maps an empty string to "N" as rapid_consensus does). */
int ref_poa_consensus_one(const char* backbone, int backbone_len, int n_members, const char* const* seqs, const int* seq_lens,
                          const char* const* cigars, const int* cig_lens, const uint8_t* spl, const uint8_t* spr, float c, float t,
                          char* out, int out_cap)
{
  PPOA poa;
  std::string bb(backbone, backbone_len);
  poa.init(bb);
  for (int m = 0; m < n_members; ++m) {
    std::string seq(seqs[m], seq_lens[m]);
    std::string cig(cigars[m], cig_lens[m]);
    bool l = spl[m], r = spr[m];
    poa.insert_alignment(seq, cig, l, r);
  }
  poa.adjust_weights(c, t);
  std::string cons;
  poa.consensus(cons);
  if ((int)cons.size() > out_cap) return -1;
  memcpy(out, cons.data(), cons.size());
  return (int)cons.size();
}

} /* extern "C" */
