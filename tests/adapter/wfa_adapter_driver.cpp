// Calls the operator-level adapter (include/wfa_adapter/bindings/cpp/WFAligner.hpp) the way the reference's call sites do
// (src/assemble.cpp:49-50; src/analignments.cpp:25-37, 70-71, 88-97, 268-280) on pairs read from stdin, one per line:
//   <pattern> <text> <endsfree 0|1> <pbf> <pef> <tbf> <tef>      ("-" = empty sequence)
// and prints per pair:  <edit status> <edit score> <affine status> <affine score> <op string or ->
// With arguments `wfadaptive <min_wavefront_length> <max_distance_threshold> <steps>` both aligners first get setHeuristicWFadaptive (as a host
// that wants WFA2-lib's adaptive default would), with `none` setHeuristicNone.
// Built by tests/test_wfa_adapter.py with g++ against libotter_gpu.so; the test compares every line with the CPU oracle.
#include "bindings/cpp/WFAligner.hpp"

#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char** argv)
{
  wfa::WFAlignerEdit aligner(wfa::WFAligner::Score, wfa::WFAligner::MemoryMed);
  wfa::WFAlignerGapAffine aligner2(4, 6, 2, wfa::WFAligner::Alignment, wfa::WFAligner::MemoryMed);
  if (argc >= 5 && std::string(argv[1]) == "wfadaptive") {
    aligner.setHeuristicWFadaptive(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    aligner2.setHeuristicWFadaptive(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  } else if (argc >= 2 && std::string(argv[1]) == "none") {
    aligner.setHeuristicNone(); aligner2.setHeuristicNone();
  }
  std::string p, t;
  int ef, pbf, pef, tbf, tef;
  while (std::cin >> p >> t >> ef >> pbf >> pef >> tbf >> tef) {
    if (p == "-") p.clear();
    if (t == "-") t.clear();
    int st1, st2;
    if (ef) { st1 = aligner.alignEndsFree(p, pbf, pef, t, tbf, tef); st2 = aligner2.alignEndsFree(p, pbf, pef, t, tbf, tef); }
    else { st1 = aligner.alignEnd2End(p, t); st2 = aligner2.alignEnd2End(p, t); }
    if (st1 != 0 || st2 != 0) { std::cerr << "adapter: " << aligner.strError() << " / " << aligner2.strError() << "\n"; return 3; }
    const std::string cigar = aligner2.getAlignmentCigar();
    std::cout << st1 << " " << aligner.getAlignmentScore() << " " << st2 << " " << aligner2.getAlignmentScore() << " " << (cigar.empty() ? "-" : cigar) << "\n";
  }
  return 0;
}
