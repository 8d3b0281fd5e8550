// otter_assemble — `otter assemble` on MI355X through the C-ABI alone (include/otter_gpu.h): the reference's command line
// (src/command_assemble.cpp:20-45) over otg_assemble_files, the dispatcher of libotter_gpu.so.  Host C++ only: BED / BAM / FASTA in,
// SAM or FASTA records on stdout, in BED order.
//   otter_assemble -b regions.bed -R sample [-r ref.fa] [--fasta] [--reads-only] [--haps] [-p] [-l] [-o L[,R]] [-a N] [-m Q] [-q RQ] [-c COV]
//                  [-F f] [-A len,f] [-e err] [-h bw[,len,bw]] [-f flank] [-s sim] [-t threads] [--batch N] [--gpus 0,1,..]
//                  [--wfa-heuristic none|wfadaptive[:min_wavefront_length,max_distance_threshold,steps]] <BAM>
// The last option has no counterpart in the reference: its aligners run whatever WFA2-lib's default heuristic is (src/assemble.cpp:49-50 never
// calls setHeuristic*); here the default is exact alignment and `wfadaptive` (= 10,50,1) reproduces a WFA2-lib whose default is the adaptive one.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/otter_gpu.h"

static int write_stdout(void*, const char* data, uint64_t len) { return fwrite(data, 1, (size_t)len, stdout) == (size_t)len ? 0 : 1; }

static std::vector<std::string> split(const std::string& s, char c)
{
  std::vector<std::string> out; size_t a = 0;
  for (;;) { const size_t b = s.find(c, a); out.push_back(s.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
  return out;
}

int main(int argc, char** argv)
{
  otg_assemble_job job; memset(&job, 0, sizeof job);
  otg_params_default(&job.params);
  job.ingest.offset_l = 1; job.ingest.offset_r = 0; job.ingest.threads = 1;       // --offset 1,0 and -t 1: the reference's defaults
  std::string bed, rg, ref, bam;
  std::vector<int32_t> devs;
  bool have_rg = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "[ERROR] %s needs a value\n", a.c_str()); exit(1); } return argv[++i]; };
    if (a == "-b" || a == "--bed") bed = val();
    else if (a == "-R" || a == "--sample-name") { rg = val(); have_rg = true; }
    else if (a == "-r" || a == "--reference") ref = val();
    else if (a == "--fasta") job.is_fasta = 1;
    else if (a == "--haps") job.params.ignore_haps = 0;
    else if (a == "--reads-only") job.reads_only = 1;
    else if (a == "-p" || a == "--non-primary") job.ingest.nonprimary = 1;
    else if (a == "-l" || a == "--omit-nonspanning") job.ingest.omit_nonspanning = 1;
    else if (a == "-o" || a == "--offset") { auto v = split(val(), ','); job.ingest.offset_l = atoi(v[0].c_str()); job.ingest.offset_r = v.size() > 1 ? atoi(v[1].c_str()) : job.ingest.offset_l; }
    else if (a == "-a" || a == "--max-alleles") job.params.max_alleles = atoi(val().c_str());
    else if (a == "-m" || a == "--mapq") job.ingest.mapq = atoi(val().c_str());
    else if (a == "-q" || a == "--read-quality") job.ingest.read_quality = atof(val().c_str());
    else if (a == "-c" || a == "--max-cov") job.params.max_cov = atoi(val().c_str());
    else if (a == "-F" || a == "--cov-fraction") job.params.min_cov_fraction = atof(val().c_str());
    else if (a == "-A" || a == "--cov-fraction-large") { auto v = split(val(), ','); if (v.size() == 2) { job.params.min_cov_fraction2_l = atoi(v[0].c_str()); job.params.min_cov_fraction2_f = atof(v[1].c_str()); } }
    else if (a == "-e" || a == "--max-error") job.params.max_error = atof(val().c_str());
    else if (a == "-h" || a == "--bandwidth") { auto v = split(val(), ','); job.params.bandwidth_short = atof(v[0].c_str());
      if (v.size() == 3) { job.params.bandwidth_length = atoi(v[1].c_str()); job.params.bandwidth_long = atof(v[2].c_str()); } else job.params.bandwidth_long = job.params.bandwidth_short; }
    else if (a == "-f" || a == "--flank-size") job.params.flank = atoi(val().c_str());
    else if (a == "-s" || a == "--min-sim") job.params.min_sim = atof(val().c_str());
    else if (a == "-t" || a == "--threads") job.ingest.threads = atoi(val().c_str());
    else if (a == "--batch") job.batch_regions = (uint32_t)atoi(val().c_str());
    else if (a == "--gpus") { for (auto& d : split(val(), ',')) devs.push_back(atoi(d.c_str())); }
    else if (a == "--wfa-heuristic") {
      const std::string h = val();
      if (h == "none") job.params.heuristic = OTG_HEURISTIC_NONE;
      else if (h.rfind("wfadaptive", 0) == 0) {
        job.params.heuristic = OTG_HEURISTIC_WFADAPTIVE;
        if (h.size() > 10 && h[10] == ':') { auto v = split(h.substr(11), ','); if (v.size() != 3) { fprintf(stderr, "[ERROR] --wfa-heuristic wfadaptive:<min_wavefront_length>,<max_distance_threshold>,<steps>\n"); return 1; }
          job.params.heur_min_wavefront_length = atoi(v[0].c_str()); job.params.heur_max_distance_threshold = atoi(v[1].c_str()); job.params.heur_steps_between_cutoffs = atoi(v[2].c_str()); }
      } else { fprintf(stderr, "[ERROR] --wfa-heuristic none | wfadaptive[:a,b,c]\n"); return 1; }
    }
    else if (a.size() && a[0] == '-') { fprintf(stderr, "[ERROR] unknown option %s\n", a.c_str()); return 1; }
    else bam = a;
  }
  if (bam.empty() || bed.empty() || !have_rg) { fprintf(stderr, "usage: otter_assemble -b <BED> -R <sample> [options] <BAM>   ('--bed' and '--sample-name' are required)\n"); return 1; }
  job.bam_path = bam.c_str(); job.bed_path = bed.c_str(); job.fasta_path = ref.empty() ? nullptr : ref.c_str(); job.read_group = rg.c_str();
  job.n_devices = (int32_t)devs.size(); job.devices = devs.empty() ? nullptr : devs.data();
  otg_job_stats st;
  const int rc = otg_assemble_files(&job, write_stdout, nullptr, &st);
  fflush(stdout);
  if (rc != OTG_OK) { fprintf(stderr, "[ERROR] otter_assemble failed (%d): %s\n", rc, otg_last_error(nullptr)); return 1; }
  fprintf(stderr, "otter_assemble: %llu regions (%llu with alleles), %llu reads, %llu alleles, %.1f MB out; %.3f s wall = %.0f regions/s on %u GPU(s); stage busy ms: ingest %.0f, hot path %.0f, emit %.0f\n",
          (unsigned long long)st.n_regions, (unsigned long long)st.n_regions_ok, (unsigned long long)st.n_reads, (unsigned long long)st.n_alleles, st.output_bytes / 1e6,
          st.ms_total / 1e3, st.ms_total > 0 ? st.n_regions / (st.ms_total / 1e3) : 0.0, st.n_devices, st.ms_ingest, st.ms_hot_path, st.ms_emit);
  return 0;
}
