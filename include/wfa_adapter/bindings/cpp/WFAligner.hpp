/*
 * bindings/cpp/WFAligner.hpp — operator-level adapter: the six `wfa::` symbols holstegelab/otter links against WFA2-lib
 * (the undefined symbols of the compiled reference objects), header-only, over the C-ABI of libotter_gpu.so.
 *
 *   wfa::WFAlignerEdit(AlignmentScope, MemoryModel)                         constructed src/assemble.cpp:49
 *   wfa::WFAlignerGapAffine(int mismatch, int gap_opening, int gap_extension, AlignmentScope, MemoryModel)     src/assemble.cpp:50
 *   AlignmentStatus alignEnd2End(std::string& pattern, std::string& text)                    src/analignments.cpp:25,70,268-273
 *   AlignmentStatus alignEndsFree(std::string& pattern, int pattern_begin_free, int pattern_end_free,
 *                                 std::string& text, int text_begin_free, int text_end_free)  src/analignments.cpp:31,88-97,274-279
 *   int getAlignmentScore()        edit: +distance; gap-affine with match 0: -penalty (WFA2-lib's convention)      src/analignments.cpp:71,91,97
 *   std::string getAlignmentCigar()      one op per column over M X I D, free end gaps explicit                    src/analignments.cpp:37,280
 *
 * A build of otter that puts THIS directory where its Makefile expects WFA2-lib (`-I$(WFADIR)`, Makefile:4) and links `-lotter_gpu` instead
 * of `-lwfacpp` (Makefile:5) runs its alignments on the MI355X — one alignment per call, i.e. at kernel-launch latency: a correctness drop-in
 * for the operator boundary.  Throughput needs the region-level boundary (otg_assemble_*, include/otter_gpu.h), which batches all
 * alignments of thousands of regions per launch.  Enumerators and their values are those of WFA2-lib v2.3.x (recovered from the debug
 * bundle test/ppoa_test.dSYM).  Heuristics: setHeuristicNone() (the default of THIS adapter: exact alignment) and
 * setHeuristicWFadaptive(min_wavefront_length, max_distance_threshold, steps_between_cutoffs) — WFAligner.hpp:107-122 of WFA2-lib.  otter
 * calls neither (src/assemble.cpp:49-50), i.e. it runs WFA2-lib's own default; a build that wants the adaptive default of WFA2-lib v2.3
 * defines OTG_ADAPTER_DEFAULT_WFADAPTIVE (aligners then start in wfadaptive(10, 50, 1)).  The banded / x-drop / z-drop strategies are not
 * provided (nothing in otter reaches them).
 *
 * One otg_ctx per aligner object, created on first use; without a HIP device every align call returns StatusOOM (there is no CPU fallback)
 * and strError() says why.  Not thread-safe per object — as the reference uses it: one aligner pair per worker thread.
 */
#ifndef OTTER_GPU_WFA_ADAPTER_HPP
#define OTTER_GPU_WFA_ADAPTER_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "otter_gpu.h"

namespace wfa {

class WFAligner {
 public:
  enum MemoryModel { MemoryHigh, MemoryMed, MemoryLow, MemoryUltralow };
  enum AlignmentScope { Score, Alignment };
  enum AlignmentStatus { StatusSuccessful = 0, StatusUnfeasible = -1, StatusMaxScoreReached = -2, StatusOOM = -3 };

  AlignmentStatus alignEnd2End(std::string& pattern, std::string& text) { return run(pattern, text, 0, 0, 0, 0, 0); }
  AlignmentStatus alignEndsFree(std::string& pattern, const int pattern_begin_free, const int pattern_end_free,
                                std::string& text, const int text_begin_free, const int text_end_free)
  {
    return run(pattern, text, 1, pattern_begin_free, pattern_end_free, text_begin_free, text_end_free);
  }
  // Heuristic of the following align calls (WFAligner.hpp:107-122 of WFA2-lib)
  void setHeuristicNone() { heur_ = OTG_HEURISTIC_NONE; heur_dirty_ = true; }
  void setHeuristicWFadaptive(const int min_wavefront_length, const int max_distance_threshold, const int steps_between_cutoffs = 1)
  {
    heur_ = OTG_HEURISTIC_WFADAPTIVE; heur_p_[0] = min_wavefront_length; heur_p_[1] = max_distance_threshold; heur_p_[2] = steps_between_cutoffs;
    heur_dirty_ = true;
  }
  int getAlignmentScore() { return score_; }
  int getAlignmentStatus() { return (int)status_; }
  std::string getAlignmentCigar() { return cigar_; }
  const char* strError() const { return error_.c_str(); }

  virtual ~WFAligner() { if (ctx_) otg_destroy(ctx_); }
  WFAligner(const WFAligner&) = delete;
  WFAligner& operator=(const WFAligner&) = delete;

 protected:
  WFAligner(const AlignmentScope scope, const MemoryModel) : scope_(scope) {}
  // device ordinal of the contexts the adapter creates (default 0): set OTG_ADAPTER_DEVICE-style policies in the host, not here
  virtual bool affine() const = 0;
  int x_ = 0, o_ = 0, e_ = 0;

 private:
  AlignmentStatus run(const std::string& pattern, const std::string& text, int endsfree, int pbf, int pef, int tbf, int tef)
  {
    score_ = 0; cigar_.clear(); status_ = StatusOOM;
    if (!ctx_ && otg_create(0, &ctx_) != OTG_OK) { const char* m = otg_last_error(nullptr); error_ = m ? m : "otg_create failed"; ctx_ = nullptr; return status_; }
    if (heur_dirty_) {
      if (otg_set_heuristic(ctx_, heur_, heur_p_[0], heur_p_[1], heur_p_[2]) != OTG_OK) { const char* m = otg_last_error(ctx_); error_ = m ? m : "otg_set_heuristic failed"; return status_; }
      heur_dirty_ = false;
    }
    // one arena: pattern, text, 64 bytes of slack (the kernels' 8-byte probes may read past an end)
    arena_.assign(pattern.size() + text.size() + 64, 0);
    if (!pattern.empty()) arena_.replace(0, pattern.size(), pattern);
    if (!text.empty()) arena_.replace(pattern.size(), text.size(), text);
    otg_align_task t;
    t.pattern_off = 0; t.text_off = pattern.size(); t.pattern_len = (uint32_t)pattern.size(); t.text_len = (uint32_t)text.size();
    t.pattern_begin_free = pbf; t.pattern_end_free = pef; t.text_begin_free = tbf; t.text_end_free = tef; t.endsfree = endsfree; t._pad = 0;
    int32_t sc = 0;
    int rc;
    if (!affine()) {
      rc = otg_edit_distance_batch(ctx_, (const uint8_t*)arena_.data(), arena_.size(), &t, 1, &sc, nullptr);
      if (rc == OTG_OK) score_ = sc;                       // edit distance: reported positive
    } else {
      uint64_t off = 0, used = 0; uint32_t len = 0;
      ops_.resize(pattern.size() + text.size() + 64);
      rc = otg_affine_align_batch(ctx_, (const uint8_t*)arena_.data(), arena_.size(), &t, 1, x_, o_, e_, &sc, &off, &len,
                                  (uint8_t*)&ops_[0], ops_.size(), &used, nullptr);
      if (rc == OTG_OK) { score_ = -sc; if (scope_ == Alignment) cigar_.assign(ops_.data() + off, len); }      // match score 0: WFA2-lib reports -penalty
    }
    if (rc != OTG_OK) { const char* m = otg_last_error(ctx_); error_ = m ? m : "alignment failed"; return status_; }
    status_ = StatusSuccessful;
    return status_;
  }
  otg_ctx* ctx_ = nullptr;
#ifdef OTG_ADAPTER_DEFAULT_WFADAPTIVE
  int heur_ = OTG_HEURISTIC_WFADAPTIVE;
#else
  int heur_ = OTG_HEURISTIC_NONE;
#endif
  int heur_p_[3] = {10, 50, 1};
  bool heur_dirty_ = true;
  AlignmentScope scope_;
  AlignmentStatus status_ = StatusSuccessful;
  int score_ = 0;
  std::string cigar_, arena_, ops_, error_;
};

class WFAlignerEdit : public WFAligner {
 public:
  WFAlignerEdit(const AlignmentScope alignmentScope, const MemoryModel memoryModel = MemoryHigh) : WFAligner(alignmentScope, memoryModel) {}
 protected:
  bool affine() const override { return false; }
};

class WFAlignerGapAffine : public WFAligner {
 public:
  WFAlignerGapAffine(const int mismatch, const int gap_opening, const int gap_extension, const AlignmentScope alignmentScope,
                     const MemoryModel memoryModel = MemoryHigh) : WFAligner(alignmentScope, memoryModel)
  {
    x_ = mismatch; o_ = gap_opening; e_ = gap_extension;
  }
 protected:
  bool affine() const override { return true; }
};

}  // namespace wfa

#endif
