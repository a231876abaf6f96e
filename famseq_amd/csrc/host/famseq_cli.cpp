// famseq_cli.cpp — FamSeq-compatible command line over libfamseq_hip's C ABI.
//
// Keeps the reference's CLI and file surface for the `-method 1` path:
//   FamSeq vcf -vcfFile f -pedFile p -output o [-v] [-a] [-d] [-o] [-l loc] [-method 1]
//              [-mRate r] [-genoProbN a b c] [-genoProbK a b c] [-genoProbXN a c]
//              [-genoProbXK a c] [-LRC x]
//   FamSeq LK  -lkFile f -pedFile p -output o [-lkType n|log10|ln|PS] [...]
// Reference behaviour being reproduced (all cites /root/reference/src):
//   flag parsing + defaults + messages   checkInput.cpp:149-578, 671-1067; FamSeq.cpp:28-156
//   readPed / setFam                     file.cpp:24-62, 1888-1968
//   VCF driver                           file.cpp:108-1004  (header rewrite :143-196, column
//                                        mapping :200-232, location filter :235-360, site rules
//                                        :362-473, Known/chrType :476-486, missing samples
//                                        :488-518, PL/GL -> likelihood :584-592/:821-828,
//                                        failure output :607-620, Phred text :684-765/:922-1001)
//   LK driver                            file.cpp:1640-1886
// Differences by design: ONE pass over the input (the reference reads the file twice), sites are
// queued and evaluated in batches on the GPU (famseq_bn_batch) with the output order preserved,
// and only -method 1 exists here (methods 2/3 are other algorithms, out of scope).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sstream>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "famseq_hip.h"
#include "fmt_g6.h"

namespace {

using std::string;
using std::vector;

// ---- small text helpers ------------------------------------------------------------------

// normal.cpp:15-53 with trim=false: empty tokens are kept, including a trailing one.
vector<string> split_keep(const string &s, char sep) {
  vector<string> out;
  if (s.empty()) return out;
  size_t head = 0, tail;
  while ((tail = s.find(sep, head)) != string::npos) {
    out.emplace_back(s, head, tail - head);
    head = tail + 1;
  }
  out.emplace_back(s, head);
  return out;
}

// the same split without copies: views into `s` (valid while `s` lives)
void split_views(std::string_view s, char sep, vector<std::string_view> &out) {
  out.clear();
  if (s.empty()) return;
  size_t head = 0, tail;
  while ((tail = s.find(sep, head)) != std::string_view::npos) {
    out.push_back(s.substr(head, tail - head));
    head = tail + 1;
  }
  out.push_back(s.substr(head));
}

// Output text of one formatting thread: appended through a raw pointer (no per-piece capacity checks beyond room()),
// kept from batch to batch.  Numbers go through famseq_fmt::g6 — printf's %g exactly, a third of to_chars' cost.
struct TextBuf {
  vector<char> v;
  size_t n = 0;
  char *room(size_t need) {
    if (v.size() < n + need) v.resize(std::max(v.size() * 2, n + need + (size_t(1) << 16)));
    return v.data() + n;
  }
  void put(const char *s, size_t len) {
    std::memcpy(room(len), s, len);
    n += len;
  }
  void put(std::string_view s) { put(s.data(), s.size()); }
  void ch(char c) { *room(1) = c, ++n; }
  void num(double x) {
    char *p = room(32);
    n += size_t(famseq_fmt::g6(p, x) - p);
  }
  // "a,b,c:d,e,f:" — the GPP and FPP triples of one sample
  void triples(const double *g, const double *f) {
    char *p = room(6 * 32 + 8), *q = p;
    q = famseq_fmt::g6(q, g[0]), *q++ = ',';
    q = famseq_fmt::g6(q, g[1]), *q++ = ',';
    q = famseq_fmt::g6(q, g[2]), *q++ = ':';
    q = famseq_fmt::g6(q, f[0]), *q++ = ',';
    q = famseq_fmt::g6(q, f[1]), *q++ = ',';
    q = famseq_fmt::g6(q, f[2]), *q++ = ':';
    n += size_t(q - p);
  }
  // the same as the device wrote it (famseq_bn_call_text_batch): "a,b,c:d,e,f:0/1\t", FAMSEQ_TEXT_STRIDE bytes, the count in the last
  void record(const char *r) {
    char *p = room(FAMSEQ_TEXT_STRIDE);
    std::memcpy(p, r, FAMSEQ_TEXT_STRIDE);
    n += size_t((unsigned char)r[FAMSEQ_TEXT_STRIDE - 1]);
  }
};

// Where the numbers become text: on the device (default — one more streaming kernel behind the call kernel, a record per
// sample comes back: famseq_bn_call_text_batch) or on the host's cores (FAMSEQ_HOST_FORMAT=1: famseq_fmt::g6, sixty calls per
// ten-member site, 0.74 of the 0.88 s loop per 3 M sites on 16 threads).  Both print the same bytes (csrc/g6_core.h).
bool device_text() {
  const char *e = std::getenv("FAMSEQ_HOST_FORMAT");
  return !(e && std::atoi(e) != 0);
}

// pow(10, -|x|/10) for a PL/GL field (file.cpp:588-590).  Integer fields (the usual PL) go
// through a table filled with the same libm pow call, so the value is identical.
struct PlTable {
  vector<double> lut;
  PlTable() : lut(4096) {
    for (size_t k = 0; k < lut.size(); ++k) lut[k] = std::pow(10.0, -std::fabs(double(k)) / 10.0);
  }
  // the likelihood of a packed integer PL: what operator() returned when it packed it
  double value(uint16_t v) const { return v < lut.size() ? lut[v] : std::pow(10.0, -std::fabs(double(v)) / 10.0); }
  // *packed receives the integer PL (clamped to 65534: anything >= 3240 is exactly 0 anyway) or
  // is left untouched and *integral cleared when the field is not a plain non-negative integer.
  double operator()(const char *b, const char *e, uint16_t *packed, bool *integral) const {
    unsigned long v = 0;
    const char *p = b;
    while (p < e && *p >= '0' && *p <= '9' && v < 100000000ul) v = v * 10 + unsigned(*p++ - '0');
    if (p == e && p > b) {
      *packed = uint16_t(v < 65534 ? v : 65534);
      return v < lut.size() ? lut[v] : std::pow(10.0, -std::fabs(double(v)) / 10.0);
    }
    *integral = false;
    const double x = std::atof(string(b, e).c_str());
    return std::pow(10.0, -std::fabs(x) / 10.0);
  }
};

// Text formatting is the slowest part of the pipeline (printf-style %g per number); the records
// of a flushed batch are independent, so they are formatted on all host cores and written in order.
template <class F>
void parallel_for(size_t n, F f) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("FAMSEQ_THREADS")) t = (unsigned)std::atoi(e);
  t = std::max(1u, std::min(t, 32u));
  if (n < 2048 || t == 1) {
    for (size_t i = 0; i < n; ++i) f(i);
    return;
  }
  vector<std::thread> pool;
  for (unsigned k = 0; k < t; ++k)
    pool.emplace_back([=] {
      for (size_t i = n * k / t, e = n * (k + 1) / t; i < e; ++i) f(i);
    });
  for (std::thread &th : pool) th.join();
}

// ---- options -----------------------------------------------------------------------------

struct Options {
  bool lk_mode = false;
  bool pack_mode = false;  // `FamSeq pack`: vcf -> packed binary PL file (no GPU)
  bool tune_mode = false;  // `FamSeq tune`: time this pedigree's kernel variants on this GPU, keep the winners' indices in the kernel cache
  bool pl_mode = false;    // `FamSeq PL`: packed binary PL file -> calls
  bool unpack_mode = false;  // `FamSeq unpack`: packed PL file + packed result file -> the text `FamSeq PL` writes
  bool bin_output = false;   // `FamSeq PL -binOutput`: results as a packed result file, not as text
  string pl_file, po_file;
  vector<string> vcf_files;
  string lk_file, ped_file, out_file, loc_file;
  bool var_only = false, all_line = false, diff_only = false, pos_order = false;
  int method = 1, lk_type = 1;
  double mrate = 1e-7, lrc = 1;
  vector<double> gN, gK, gXN, gXK;
  int num_burn = -999, num_rep = -999;
};

// returns 0 good, 1 warnings, -1 stop (checkInput.h:345-349)
int parse_options(int argc, char **argv, Options &o) {
  int rv = 0;
  auto missing = [&](int i) { return i == argc || argv[i][0] == '-'; };
  for (int i = 2; i < argc; i++) {
    if (argv[i][0] != '-') {
      std::cout << "Cannot recognize parameter: \"" << argv[i] << "\" in the command." << std::endl;
      rv = 1;
      continue;
    }
    const string opt(argv[i] + 1);
    auto need_file = [&](string &dst, const char *what) {
      i++;
      if (missing(i)) {
        std::cout << "The " << what << " hasn't been set. Please check input." << std::endl;
        return false;
      }
      dst = argv[i];
      return true;
    };
    auto probs = [&](vector<double> &dst, int n, const char *msg) {
      vector<double> tmp(3, 0);
      for (int j = 0; j < n; j++) {
        i++;
        if (missing(i)) {
          std::cout << msg << std::endl;
          i--;
          rv = 1;
          return;
        }
        tmp[n == 3 ? j : 2 * j] = std::atof(argv[i]);  // X priors: {a, 0, c} (checkInput.cpp:340-372)
      }
      dst = tmp;
    };
    if (!o.lk_mode && opt == "vcfFile") {
      bool first = true;
      while (true) {
        i++;
        if (first && missing(i)) {
          std::cout << "The vcf file hasn't been set. Please check input." << std::endl;
          return -1;
        }
        first = false;
        if (missing(i)) {
          i--;
          break;
        }
        o.vcf_files.push_back(argv[i]);
      }
    } else if (o.pl_mode && opt == "plFile") {
      if (!need_file(o.pl_file, "packed PL file")) return -1;
    } else if (o.unpack_mode && opt == "poFile") {
      if (!need_file(o.po_file, "packed result file")) return -1;
    } else if (o.pl_mode && !o.unpack_mode && opt == "binOutput") {
      o.bin_output = true;
    } else if (o.lk_mode && opt == "lkFile") {
      if (!need_file(o.lk_file, "likelihood file")) return -1;
    } else if (opt == "pedFile") {
      if (!need_file(o.ped_file, "ped file")) return -1;
    } else if (!o.lk_mode && opt == "l") {
      if (!need_file(o.loc_file, "location file")) return -1;
    } else if (opt == "output") {
      if (!need_file(o.out_file, "output file")) return -1;
    } else if (!o.lk_mode && opt == "v") {
      o.var_only = true;
    } else if (!o.lk_mode && opt == "a") {
      o.all_line = true;
    } else if (!o.lk_mode && opt == "d") {
      o.diff_only = true;
    } else if (!o.lk_mode && opt == "o") {
      o.pos_order = true;
    } else if (opt == "method") {
      i++;
      if (missing(i)) {
        std::cout << "Method hasn't been set. The default method (BN) will be used." << std::endl;
        i--;
        rv = 1;
      } else {
        o.method = std::atoi(argv[i]);
      }
    } else if (opt == "mRate") {
      i++;
      if (missing(i)) {
        std::cout << "Mutation rate hasn't been set. The default (1e-7) will be used." << std::endl;
        i--;
        rv = 1;
      } else {
        o.mrate = std::atof(argv[i]);
      }
    } else if (o.lk_mode && opt == "lkType") {
      i++;
      if (missing(i)) {
        std::cout << "Likelihood type hasn't been set. The default normal (n) will be used." << std::endl;
        i--;
        rv = 1;
      } else {
        const string t(argv[i]);
        if (t == "n") o.lk_type = 1;
        else if (t == "log10") o.lk_type = 2;
        else if (t == "ln") o.lk_type = 3;
        else if (t == "PS") o.lk_type = 4;
        else {
          std::cout << "Cannot recognize the likelihood type: " << t << ". The default normal (n) will be used." << std::endl;
          o.lk_type = 1;
          rv = 1;
        }
      }
    } else if (opt == "genoProbN") {
      probs(o.gN, 3, "genoProbN hasn't been set. The default (0.9985,0.001,0.0005) will be used.");
    } else if (opt == "genoProbK") {
      probs(o.gK, 3, "genoProbK hasn't been set. The default (0.45,0.1,0.45) will be used.");
    } else if (opt == "genoProbXN") {
      probs(o.gXN, 2, "genoProbN hasn't been set. The default (0.999,0.001) will be used.");
    } else if (opt == "genoProbXK") {
      probs(o.gXK, 2, "genoProbN hasn't been set. The default (0.5,0.5) will be used.");
    } else if (opt == "numBurnIn" || opt == "numRep") {
      i++;
      if (missing(i)) {
        std::cout << "Number of " << (opt == "numRep" ? "MCMC repeat" : "burn in")
                  << " times hasn't been set. The default will be used." << std::endl;
        i--;
        rv = 1;
      } else {
        (opt == "numRep" ? o.num_rep : o.num_burn) = std::atoi(argv[i]);
      }
    } else if (opt == "LRC") {
      i++;
      if (missing(i)) {
        std::cerr << "Likelihood ratio criteria is not set. The default will be used." << std::endl;
        i--;
        rv = 1;
      } else {
        o.lrc = std::atof(argv[i]);
      }
    } else {
      std::cout << "Cannot recognize option: \"" << opt << "\" in the command." << std::endl;
      rv = 1;
    }
  }
  if (!o.lk_mode && o.vcf_files.empty()) {
    std::cout << "The name of vcf file must be set. Please input the vcf file name." << std::endl;
    return -1;
  }
  if (o.pl_mode && o.pl_file.empty()) {
    std::cout << "The name of packed PL file must be set. Please input the file name (-plFile)." << std::endl;
    return -1;
  }
  if (o.unpack_mode && o.po_file.empty()) {
    std::cout << "The name of packed result file must be set. Please input the file name (-poFile)." << std::endl;
    return -1;
  }
  if (o.lk_mode && !o.pl_mode && o.lk_file.empty()) {
    std::cout << "The name of likelihood file must be set. Please input the likelihood file name." << std::endl;
    return -1;
  }
  if (o.ped_file.empty()) {
    std::cout << "The name of ped file must be set. Please input the ped file name." << std::endl;
    return -1;
  }
  if (o.out_file.empty() && !o.tune_mode) {
    std::cout << "The name of output file must be set. Please input the output file name." << std::endl;
    return -1;
  }
  if (o.method < 1 || o.method > 3) {
    std::cout << "Method could only be 1 or 3. The default method (BN) will be used." << std::endl;
    o.method = 1;
    rv = 1;
  }
  if (o.mrate < 0 || o.mrate > 0.5) {
    std::cout << "Mutation rate is set out of range. The default (1e-7) will be used." << std::endl;
    o.mrate = 1e-7;
    rv = 1;
  }
  if (o.var_only && o.all_line) {
    std::cout << "varOnly is setted, allLine is blocked." << std::endl;
    o.all_line = false;
    rv = 1;
  }
  if (o.lrc < 0) {
    std::cerr << "Likelihood ration criteria is not set correctly. The default will be used." << std::endl;
    o.lrc = 1;
    rv = 1;
  }
  if (o.diff_only) {  // file.cpp:124-128
    o.all_line = false;
    o.var_only = true;
  }
  return rv;
}

// ---- pedigree ------------------------------------------------------------------------------

struct Ped {
  vector<int32_t> id, mid, fid, gender;
  vector<string> name;
  int n() const { return (int)id.size(); }
};

bool read_ped(const string &path, Ped &p) {  // file.cpp:24-62
  std::ifstream fin(path.c_str());
  if (!fin.is_open()) {
    std::cout << "Cannot open " << path << std::endl;
    return false;
  }
  string line;
  std::getline(fin, line);
  while (std::getline(fin, line)) {
    if (line.size() < 2) break;
    std::istringstream in(line);
    int a = 0, b = 0, c = 0, d = 0;
    string nm;
    in >> a >> b >> c >> d >> nm;
    p.id.push_back(a);
    p.mid.push_back(b);
    p.fid.push_back(c);
    p.gender.push_back(d);
    p.name.push_back(nm);
  }
  return true;
}

// ---- batched caller: queue of output records, flushed through the GPU in order -------------

struct Record {
  string text;     // literal line (site < 0); unused for computed lines
  long site = -1;  // index into the pending batch, -1 for literal records
  string raw;      // the input line: computed lines print pieces of it (and the failure warning all of it)
  uint32_t prefix_len = 0;  // columns 1-9 of raw, without the tab after FORMAT (vcf mode)
  const char *head = nullptr;  // LK mode: a fixed first column instead
  uint32_t n_fmt = 0;       // FORMAT keys: a missing sample prints that many "NA:"
  struct Sample {
    uint32_t off, len;  // the sample's field in raw
    bool missing;       // shorter than 5 characters (file.cpp:565): printed as NA per FORMAT key
  };
  vector<Sample> samples;  // sequenced samples, in output column order
};

double now_s();

// n items in at most `parts` contiguous ranges, one thread each: f(lo, hi, part).  Returns the number of parts.
template <class F>
int parallel_ranges(size_t n, F f) {
  unsigned t = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("FAMSEQ_THREADS")) t = (unsigned)std::atoi(e);
  t = std::max(1u, std::min(t, 32u));
  if (n < 2048) t = 1;
  vector<std::thread> pool;
  for (unsigned k = 1; k < t; ++k) pool.emplace_back([=] { f(n * k / t, n * (k + 1) / t, (int)k); });
  f(0, n / t, 0);
  for (std::thread &th : pool) th.join();
  return (int)t;
}

// Queue of output records between the LK driver's line loop and the GPU (the vcf driver has its own block
// pipeline, run_vcf).  Records are kept in input order; a full batch is handed to a flusher thread — GPU call
// (famseq_bn_call_batch: posterior, Phred scaling and genotype call all on the device), formatting on all
// cores, one write per formatting thread — while the driver's thread goes on filling the next batch.
class BatchCaller {
 public:
  double t_gpu = 0, t_format = 0, t_write = 0, t_stall = 0;  // FAMSEQ_TIMING: where the flusher spends its time; driver waiting for it
  BatchCaller(famseq_ctx *ctx, int n_members, const vector<int> &seq_members, std::ostream &out, size_t cap)
      : ctx_(ctx), n_(n_members), seq_(seq_members.begin(), seq_members.end()), out_(out), cap_(cap) {}
  ~BatchCaller() { wait(); }

  void literal(string line) {
    Record r;
    r.text = std::move(line);
    cur_.q.push_back(std::move(r));
  }
  // lk: N x 3 in PED order; pl: n_seq x 3 packed integer PLs in column order, or NULL when some
  // field of the site is not a plain integer (the batch then goes through the fp64 input)
  bool site(Record &&r, const vector<double> &lk, const uint16_t *pl, uint8_t flags) {
    r.site = (long)cur_.flags.size();
    cur_.lk.insert(cur_.lk.end(), lk.begin(), lk.end());
    if (pl && cur_.packed_ok) cur_.pl.insert(cur_.pl.end(), pl, pl + 3 * seq_.size());
    else cur_.packed_ok = false;
    cur_.flags.push_back(flags);
    cur_.q.push_back(std::move(r));
    return cur_.flags.size() < cap_ || submit();
  }
  // Everything queued so far is computed and written when this returns.
  bool flush() {
    const bool ok = submit();
    wait();
    return ok && ok_;
  }

 private:
  struct Batch {
    vector<Record> q;
    vector<double> lk;
    vector<uint16_t> pl;
    vector<uint8_t> flags;
    bool packed_ok = true;
    void clear() {
      q.clear();
      lk.clear();
      pl.clear();
      flags.clear();
      packed_ok = true;
    }
  };

  void wait() {
    if (worker_.joinable()) {
      const double t0 = now_s();
      worker_.join();
      t_stall += now_s() - t0;
    }
  }
  // hand the current batch to the flusher (after the previous one is through) and start a fresh one
  bool submit() {
    wait();
    if (!ok_) return false;
    std::swap(cur_, fly_);
    cur_.clear();
    worker_ = std::thread([this] { ok_ = write_out(fly_); });
    return true;
  }

  bool write_out(Batch &b) {
    const int64_t s = (int64_t)b.flags.size();
    const size_t k = seq_.size();
    gpp_.resize(size_t(s) * k * 3);
    fpp_.resize(size_t(s) * k * 3);
    fgt_.resize(size_t(s) * k);
    status_.resize(b.flags.size());
    const double t0 = now_s();
    if (s > 0) {
      const bool packed = b.packed_ok && !b.pl.empty();
      const int rc = famseq_bn_call_batch(ctx_, s, packed ? nullptr : b.lk.data(), packed ? b.pl.data() : nullptr,
                                          b.flags.data(), seq_.data(), (int32_t)k, gpp_.data(), fpp_.data(), fgt_.data(),
                                          status_.data());
      if (rc != 0) {
        std::cerr << "famseq_bn_call_batch failed (" << rc << "): " << famseq_last_error(ctx_) << std::endl;
        return false;
      }
    }
    const double t1 = now_s();
    t_gpu += t1 - t0;
    // every formatting thread appends its contiguous range of records to one buffer: T writes per batch
    if (text_.size() < 64) text_.resize(64);
    const int parts = parallel_ranges(b.q.size(), [&](size_t lo, size_t hi, int part) {
      TextBuf &line = text_[part];
      line.n = 0;
      for (size_t i = lo; i < hi; ++i) {
        const Record &r = b.q[i];
        if (r.site < 0) {
          line.put(r.text);
        } else {
          if (r.head) {
            line.put(r.head, std::strlen(r.head));
          } else {
            line.put(r.raw.data(), r.prefix_len);  // columns 1-8 + FORMAT
            line.put(":GPP:FPP:FGT\t", 13);
          }
          if (status_[r.site] & 3) {  // file.cpp:607-620
            for (const Record::Sample &sm : r.samples) {
              line.put(r.raw.data() + sm.off, sm.len);
              line.put(":NA:NA:NA\t", 10);
            }
          } else {
            for (size_t j = 0; j < k; ++j) {
              const int gt = fgt_[size_t(r.site) * k + j];
              const Record::Sample &sm = r.samples[j];
              if (sm.missing) {
                for (uint32_t q = 0; q < r.n_fmt; ++q) line.put("NA:", 3);
              } else {
                line.put(r.raw.data() + sm.off, sm.len);
                line.ch(':');
              }
              line.triples(&gpp_[(size_t(r.site) * k + j) * 3], &fpp_[(size_t(r.site) * k + j) * 3]);
              line.put(gt == 0 ? "0/0\t" : (gt == 1 ? "0/1\t" : "1/1\t"), 4);
            }
          }
        }
        line.ch('\n');
      }
    });
    const double t2 = now_s();
    t_format += t2 - t1;
    for (const Record &r : b.q)  // the reference's warning on stdout, in input order (file.cpp:607-612)
      if (r.site >= 0 && (status_[r.site] & 3))
        std::cout << "Warning: this variant hasn't been calculated: " << std::endl << r.raw << std::endl;
    for (int part = 0; part < parts; ++part) out_.write(text_[part].v.data(), (std::streamsize)text_[part].n);
    t_write += now_s() - t2;
    return true;
  }

  famseq_ctx *ctx_;
  int n_;
  vector<int32_t> seq_;  // PED index of each sequenced output column, in input column order
  std::ostream &out_;
  size_t cap_;
  Batch cur_, fly_;  // being filled by the driver / being written out by the flusher
  std::thread worker_;
  bool ok_ = true;
  vector<double> gpp_, fpp_;
  vector<int8_t> fgt_;
  vector<uint8_t> status_;
  vector<TextBuf> text_;
};

size_t batch_capacity() {
  const char *e = std::getenv("FAMSEQ_BATCH");
  const long v = e ? std::atol(e) : 0;
  return v > 0 ? size_t(v) : size_t(1) << 16;
}

// ---- model + ctx ---------------------------------------------------------------------------

// The size-independent model of the ABI (famseq_pedigree) with the arrays it points into: the reference's drivers
// take a PED file of any length (readPed, file.cpp:24-62), and so does its -method 2.
struct CliModel {
  famseq_pedigree p;
  vector<int32_t> mo, fa;
  vector<uint8_t> seq;
  int init(const Ped &ped, const vector<uint8_t> &sequenced, double mrate, double lrc) {
    mo.assign(ped.n(), -1);
    fa.assign(ped.n(), -1);
    seq = sequenced;
    return famseq_pedigree_init(&p, ped.n(), ped.id.data(), ped.mid.data(), ped.fid.data(), ped.gender.data(), seq.data(), mrate,
                                lrc, mo.data(), fa.data());
  }
};

famseq_ctx *make_ctx(const Options &o, const Ped &ped, const vector<uint8_t> &sequenced, CliModel &m) {
  const int rc = m.init(ped, sequenced, o.mrate, o.lrc);
  if (rc == FAMSEQ_E_PED_HALF) std::cout << "This is not a fulfill family. Please check the ped file." << std::endl;
  if (rc == FAMSEQ_E_PED_SEX) std::cerr << "A mother is not a female or a father is not a male in the ped file." << std::endl;
  if (rc != 0) {
    std::cout << "Cannot initiate family. Please check ped file." << std::endl;
    return nullptr;
  }
  auto put = [](double *dst, const vector<double> &src) {
    if (src.size() == 3) std::copy(src.begin(), src.end(), dst);
  };
  put(m.p.genoProbN, o.gN);
  put(m.p.genoProbK, o.gK);
  put(m.p.genoProbXN, o.gXN);
  put(m.p.genoProbXK, o.gXK);
  const char *dev = std::getenv("FAMSEQ_DEVICE");
  char err[512] = {0};
  if (ped.n() > FAMSEQ_MAX_MEMBERS && o.method != 2) {  // the reference's own advice for -method 1 is "family size less than seven"
    std::cout << "The pedigree has " << ped.n() << " members: -method 1 enumerates 3^N joint genotypes and serves up to "
              << FAMSEQ_MAX_MEMBERS << ". Use -method 2 for this pedigree." << std::endl;
    return nullptr;
  }
  famseq_ctx *ctx = famseq_create_pedigree(&m.p, dev ? std::atoi(dev) : 0, err, sizeof err);
  if (!ctx) {
    std::cerr << "Cannot create the GPU context: " << err << std::endl;
    return nullptr;
  }
  // -method 2 (the reference's Elston-Stewart peeling, family.cpp:1126-1403) computes the same
  // marginals exactly: here that is the sum-product engine (loops are handled by conditioning on up
  // to three members).
  if (o.method == 2 && famseq_set_option(ctx, "engine", FAMSEQ_ENGINE_ELIM) != 0) {
    std::cout << "-method 2 cannot serve this pedigree: " << famseq_last_error(ctx) << std::endl
              << "Use -method 1 for this pedigree." << std::endl;
    famseq_destroy(ctx);
    return nullptr;
  }
  return ctx;
}

void put_triple(std::ostream &o, const double *p) { o << p[0] << ":" << p[1] << ":" << p[2]; }

// ---- packed PL file (row N4 of SURVEY.md 8(f)): the text-free feed for the GPU ---------------
//   bytes 0-7   "FSPL0001"
//   8-11        uint32 n_seq            sequenced samples (columns)
//   12-15       uint32 record_bytes     = 1 + 6*n_seq
//   16-23       uint64 n_sites          (0 while being written; patched on close)
//   24-...      n_seq x char[32]        sample names, NUL padded
//   then n_sites records: uint8 flags (bit0 Known, bit1 chrX); uint16 pl[n_seq][3] little endian,
//   integer PL clamped to 65534, 0xFFFF x3 = sample missing at the site.
// Only sites the vcf driver would compute are written (same site rules, file.cpp:362-555).
const char kPlMagic[9] = "FSPL0001";

struct PackWriter {
  std::ofstream out;
  uint32_t n_seq = 0;
  uint64_t n_sites = 0, skipped = 0;
  bool open(const string &path, const vector<string> &names) {
    out.open(path.c_str(), std::ios::binary);
    if (!out.is_open()) return false;
    n_seq = (uint32_t)names.size();
    const uint32_t rec = 1 + 6 * n_seq;
    const uint64_t zero = 0;
    out.write(kPlMagic, 8);
    out.write(reinterpret_cast<const char *>(&n_seq), 4);
    out.write(reinterpret_cast<const char *>(&rec), 4);
    out.write(reinterpret_cast<const char *>(&zero), 8);
    for (const string &n : names) {
      char buf[32] = {0};
      std::strncpy(buf, n.c_str(), 31);
      out.write(buf, 32);
    }
    return bool(out);
  }
  void add(uint8_t flags, const uint16_t *pl) {
    out.write(reinterpret_cast<const char *>(&flags), 1);
    out.write(reinterpret_cast<const char *>(pl), 6 * n_seq);
    ++n_sites;
  }
  bool close() {
    out.seekp(16);
    out.write(reinterpret_cast<const char *>(&n_sites), 8);
    out.close();
    return !out.fail();
  }
};

// Packed result file (`FamSeq PL -binOutput`), little-endian: "FSPO0001", n_seq u32, reserved u32,
// n_sites u64, n_seq x 32-byte names; then blocks of  n u64 | status[n] u8 | gpp[n][n_seq][3] f64 |
// fpp[n][n_seq][3] f64 | fgt[n][n_seq] i8  — exactly what famseq_bn_call_batch hands back, so that a
// run is bounded by the host link and the disk, not by printing 6 n_seq numbers per site.
const char kPoMagic[8] = {'F', 'S', 'P', 'O', '0', '0', '0', '1'};

// One batch of the packed-PL driver.  The arrays the GPU call reads and writes live in pinned host
// memory (famseq_alloc_pinned: both directions of the host link at their full rate).
struct PlBatch {
  size_t cap = 0, k = 0, n = 0;
  uint8_t *flags = nullptr, *status = nullptr;
  uint16_t *pl = nullptr;
  double *gpp = nullptr, *fpp = nullptr;
  int8_t *fgt = nullptr;
  char *text = nullptr;  // as_text: the device's records instead of gpp / fpp / fgt
  bool alloc(size_t cap_, size_t k_, bool as_text) {
    cap = cap_;
    k = k_;
    flags = static_cast<uint8_t *>(famseq_alloc_pinned(cap));
    status = static_cast<uint8_t *>(famseq_alloc_pinned(cap));
    pl = static_cast<uint16_t *>(famseq_alloc_pinned(cap * k * 6));
    if (as_text) {
      text = static_cast<char *>(famseq_alloc_pinned(cap * k * FAMSEQ_TEXT_STRIDE));
      return flags && status && pl && text;
    }
    gpp = static_cast<double *>(famseq_alloc_pinned(cap * k * 24));
    fpp = static_cast<double *>(famseq_alloc_pinned(cap * k * 24));
    fgt = static_cast<int8_t *>(famseq_alloc_pinned(cap * k));
    return flags && status && pl && gpp && fpp && fgt;
  }
  void release() {
    for (void *q : {(void *)flags, (void *)status, (void *)pl, (void *)gpp, (void *)fpp, (void *)fgt, (void *)text}) famseq_free_pinned(q);
  }
  // the GPU call of this batch's first n sites
  int call(famseq_ctx *ctx, size_t n_sites, const double *lk, const int32_t *seq_members) {
    if (text)
      return famseq_bn_call_text_batch(ctx, (int64_t)n_sites, lk, lk ? nullptr : pl, flags, seq_members, (int32_t)k, text, status);
    return famseq_bn_call_batch(ctx, (int64_t)n_sites, lk, lk ? nullptr : pl, flags, seq_members, (int32_t)k, gpp, fpp, fgt, status);
  }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Write `bytes` at `off` of fd from `parts` threads (the copy into the page cache is the slow part of
// storing 490 bytes per site; it parallelises, a single stream writer does not).
bool pwrite_parallel(int fd, const void *data, size_t bytes, off_t off, int parts) {
  if (bytes < (size_t(1) << 20)) parts = 1;
  std::vector<std::thread> pool;
  std::vector<char> good(parts, 1);
  for (int t = 0; t < parts; ++t)
    pool.emplace_back([&, t] {
      size_t lo = bytes * t / parts, hi = bytes * (t + 1) / parts;
      const char *p = static_cast<const char *>(data);
      while (lo < hi) {
        const ssize_t w = ::pwrite(fd, p + lo, hi - lo, off + (off_t)lo);
        if (w <= 0) {
          good[t] = 0;
          return;
        }
        lo += (size_t)w;
      }
    });
  for (std::thread &th : pool) th.join();
  for (char g : good)
    if (!g) return false;
  return true;
}

// FIFO hand-off between the stages of the packed-PL pipeline (reader -> GPU -> writer -> reader).
// -1 = the producer is done.
class Channel {
 public:
  void put(int v) {
    {
      std::lock_guard<std::mutex> g(mu_);
      q_.push_back(v);
    }
    cv_.notify_one();
  }
  int take() {
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return !q_.empty(); });
    const int v = q_.front();
    q_.pop_front();
    return v;
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<int> q_;
};

bool run_pl(const Options &o, const Ped &ped) {
  std::ifstream fin(o.pl_file.c_str(), std::ios::binary);
  if (!fin.is_open()) {
    std::cout << "Cannot open " << o.pl_file << std::endl;
    return false;
  }
  char magic[8];
  uint32_t n_seq = 0, rec = 0;
  uint64_t n_sites = 0;
  fin.read(magic, 8);
  fin.read(reinterpret_cast<char *>(&n_seq), 4);
  fin.read(reinterpret_cast<char *>(&rec), 4);
  fin.read(reinterpret_cast<char *>(&n_sites), 8);
  if (!fin || std::memcmp(magic, kPlMagic, 8) != 0 || n_seq == 0 || n_seq > 4096 || rec != 1 + 6 * n_seq) {
    std::cout << o.pl_file << " is not a packed PL file." << std::endl;
    return false;
  }
  vector<string> names(n_seq);
  for (uint32_t i = 0; i < n_seq; i++) {
    char buf[33] = {0};
    fin.read(buf, 32);
    names[i] = buf;
  }
  // columns -> PED members (file.cpp:1671-1689); columns that are not in the PED are ignored
  vector<int> col_member(n_seq, -1);
  vector<uint8_t> sequenced(ped.n(), 0);
  for (uint32_t i = 0; i < n_seq; i++)
    for (int j = 0; j < ped.n(); j++)
      if (names[i] == ped.name[j]) {
        col_member[i] = j;
        sequenced[j] = 1;
        break;
      }
  vector<int32_t> seq_members;
  vector<uint32_t> seq_cols;
  for (uint32_t i = 0; i < n_seq; i++)
    if (col_member[i] >= 0) {
      seq_cols.push_back(i);
      seq_members.push_back(col_member[i]);
    }
  if (seq_members.empty()) {
    std::cout << "No sample of " << o.pl_file << " is in the ped file." << std::endl;
    return false;
  }
  const size_t k = seq_members.size();

  // unpack mode: the results come from a packed result file instead of the GPU
  std::ifstream fpo;
  if (o.unpack_mode) {
    fpo.open(o.po_file.c_str(), std::ios::binary);
    char pm[8];
    uint32_t pk = 0, reserved = 0;
    uint64_t pn = 0;
    fpo.read(pm, 8);
    fpo.read(reinterpret_cast<char *>(&pk), 4);
    fpo.read(reinterpret_cast<char *>(&reserved), 4);
    fpo.read(reinterpret_cast<char *>(&pn), 8);
    if (!fpo || std::memcmp(pm, kPoMagic, 8) != 0 || pk != k) {
      std::cout << o.po_file << " is not a packed result file for these samples." << std::endl;
      return false;
    }
    for (size_t j = 0; j < k; j++) {
      char buf[33] = {0};
      fpo.read(buf, 32);
      if (names[seq_cols[j]] != buf) {
        std::cout << o.po_file << " was written for other samples than " << o.pl_file << " holds." << std::endl;
        return false;
      }
    }
  }

  CliModel m;
  famseq_ctx *ctx = nullptr;
  if (o.unpack_mode) {  // only the header needs the model's priors
    if (m.init(ped, sequenced, o.mrate, o.lrc) != 0) {
      std::cout << "Cannot initiate family. Please check ped file." << std::endl;
      return false;
    }
    if (o.gN.size() == 3) std::copy(o.gN.begin(), o.gN.end(), m.p.genoProbN);
  } else {
    const double t0 = now_s();
    ctx = make_ctx(o, ped, sequenced, m);
    if (!ctx) return false;
    if (std::getenv("FAMSEQ_TIMING")) std::cerr << "FamSeq PL: context (HIP start-up, plan, kernels) " << now_s() - t0 << " s" << std::endl;
  }
  std::ofstream fout(o.out_file.c_str(), o.bin_output ? std::ios::binary : std::ios::out);
  uint64_t written = 0;
  if (o.bin_output) {
    const uint32_t k32 = (uint32_t)k, reserved = 0;
    fout.write(kPoMagic, 8);
    fout.write(reinterpret_cast<const char *>(&k32), 4);
    fout.write(reinterpret_cast<const char *>(&reserved), 4);
    fout.write(reinterpret_cast<const char *>(&written), 8);
    for (uint32_t c : seq_cols) {
      char buf[32] = {0};
      std::strncpy(buf, names[c].c_str(), 31);
      fout.write(buf, 32);
    }
  } else {
    fout << "##FORMAT=<ID=GPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
            "calculated by individual-base Method\">" << std::endl;
    fout << "##FORMAT=<ID=FPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
            "calculated by FamSeqPro\">" << std::endl;
    fout << "##FORMAT=<ID=FGT,Number=1,Type=String,Description=\"Genotype called by FamSeqPro\">" << std::endl;
    fout << "##FS mutation rate=" << o.mrate << " " << std::endl;
    fout << "##FS genotype frequency in pupulation: "; put_triple(fout, m.p.genoProbN); fout << std::endl;
    fout << "#FORMAT\t";
    for (uint32_t c : seq_cols) fout << names[c] << '\t';
    fout << std::endl;
  }

  // Text of one batch (what `FamSeq PL` prints per site), formatted on all cores: one buffer per thread, written in order.
  vector<TextBuf> text(64);
  auto put_uint = [](TextBuf &t, unsigned v) {
    char *p = t.room(8);
    t.n += size_t(std::to_chars(p, p + 8, v).ptr - p);
  };
  auto format_and_write = [&](size_t n, const uint16_t *pl, const uint8_t *status, const double *gpp, const double *fpp, const int8_t *fgt,
                              const char *rec) {
    const int parts = parallel_ranges(n, [&](size_t lo, size_t hi, int part) {
      TextBuf &line = text[part];
      line.n = 0;
      for (size_t s = lo; s < hi; ++s) {
        line.put("PL:GPP:FPP:FGT\t", 15);
        for (size_t j = 0; j < k; j++) {
          const uint16_t *p = &pl[(s * k + j) * 3];
          if (p[0] == 0xFFFF && p[1] == 0xFFFF && p[2] == 0xFFFF) {
            line.put("NA", 2);
          } else {
            put_uint(line, p[0]), line.ch(',');
            put_uint(line, p[1]), line.ch(',');
            put_uint(line, p[2]);
          }
          if (status[s] & 3) {
            line.put(":NA:NA:NA\t", 10);
            continue;
          }
          line.ch(':');
          if (rec) {
            line.record(rec + (s * k + j) * FAMSEQ_TEXT_STRIDE);
            continue;
          }
          line.triples(&gpp[(s * k + j) * 3], &fpp[(s * k + j) * 3]);
          const int gt = fgt[s * k + j];
          line.put(gt == 0 ? "0/0\t" : (gt == 1 ? "0/1\t" : "1/1\t"), 4);
        }
        line.ch('\n');
      }
    });
    for (int part = 0; part < parts; ++part) fout.write(text[part].v.data(), (std::streamsize)text[part].n);
  };

  if (!o.unpack_mode) {
    // `FamSeq PL`: read, GPU and write run concurrently — a reader thread de-interleaves batch b + 1 out
    // of the memory-mapped file while the GPU call of batch b runs on this thread and a writer thread
    // stores (or prints) batch b - 1.  Three batches of pinned buffers go round; order is kept by the
    // FIFOs.  (The serial loop this replaces spent two thirds of its time with the GPU idle.)
    const long header = (long)fin.tellg();
    fin.close();
    const int fd = ::open(o.pl_file.c_str(), O_RDONLY);
    struct stat st;
    if (fd < 0 || ::fstat(fd, &st) != 0) {
      std::cout << "Cannot open " << o.pl_file << std::endl;
      return false;
    }
    const size_t body = (size_t)st.st_size > (size_t)header ? (size_t)st.st_size - (size_t)header : 0;
    const size_t total = body / rec;  // whole records only, like the stream reader
    const char *map = nullptr;
    if (total) {
      void *mp = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (mp == MAP_FAILED) {
        std::cout << "Cannot map " << o.pl_file << std::endl;
        ::close(fd);
        return false;
      }
      ::madvise(mp, (size_t)st.st_size, MADV_SEQUENTIAL);
      map = static_cast<const char *>(mp);
    }
    const size_t cap = std::min(batch_capacity(), std::max<size_t>(total, 1));
    constexpr int kBatches = 3;
    PlBatch bt[kBatches];
    bool ok = true;
    const double t_alloc0 = now_s();
    for (PlBatch &b : bt) ok = b.alloc(cap, k, !o.bin_output && device_text()) && ok;
    const double t_alloc = now_s() - t_alloc0;
    if (!ok) std::cerr << "cannot allocate pinned host buffers" << std::endl;
    double t_read = 0, t_gpu = 0, t_write = 0;
    const double t_start = now_s();
    Channel to_reader, to_gpu, to_writer;
    for (int i = 0; i < kBatches; i++) to_reader.put(i);
    std::thread reader([&] {
      for (size_t at = 0; at < total && ok;) {
        const int i = to_reader.take();
        const double t0 = now_s();
        PlBatch &b = bt[i];
        b.n = std::min(cap, total - at);
        const char *raw = map + header + at * rec;
        parallel_for(b.n, [&](size_t s) {  // de-interleave: flags[] and the PED-matched columns of pl[]
          const char *r = raw + s * rec;
          b.flags[s] = uint8_t(r[0]);
          for (size_t j = 0; j < k; j++) std::memcpy(&b.pl[(s * k + j) * 3], r + 1 + 6 * seq_cols[j], 6);
        });
        at += b.n;
        t_read += now_s() - t0;
        to_gpu.put(i);
      }
      to_gpu.put(-1);
    });
    bool write_ok = true;
    // packed results go out through pwrite on a descriptor of their own (positions are known: the
    // header, then fixed-layout blocks), several threads per block
    int ofd = -1;
    off_t opos = 0;
    if (o.bin_output) {
      fout.flush();
      opos = (off_t)fout.tellp();
      ofd = ::open(o.out_file.c_str(), O_WRONLY);
      if (ofd < 0) write_ok = false;
    }
    std::thread writer([&] {
      for (;;) {
        const int i = to_writer.take();
        if (i < 0) break;
        const double t0 = now_s();
        PlBatch &b = bt[i];
        if (o.bin_output) {
          const uint64_t bn = b.n;
          const size_t wide = b.n * k * 24;
          bool g = write_ok && ::pwrite(ofd, &bn, 8, opos) == 8 && pwrite_parallel(ofd, b.status, b.n, opos + 8, 1);
          g = g && pwrite_parallel(ofd, b.gpp, wide, opos + 8 + (off_t)b.n, 4);
          g = g && pwrite_parallel(ofd, b.fpp, wide, opos + 8 + (off_t)(b.n + wide), 4);
          g = g && pwrite_parallel(ofd, b.fgt, b.n * k, opos + 8 + (off_t)(b.n + 2 * wide), 1);
          opos += 8 + (off_t)(b.n + 2 * wide + b.n * k);
          write_ok = g;
          written += b.n;
        } else {
          format_and_write(b.n, b.pl, b.status, b.gpp, b.fpp, b.fgt, b.text);
          write_ok = write_ok && !fout.fail();
        }
        t_write += now_s() - t0;
        to_reader.put(i);
      }
    });
    for (;;) {  // the GPU stage, on the thread that owns the context
      const int i = to_gpu.take();
      if (i < 0) break;
      PlBatch &b = bt[i];
      if (ok) {
        const double t0 = now_s();
        const int rc = b.call(ctx, b.n, nullptr, seq_members.data());
        t_gpu += now_s() - t0;
        if (rc != 0) {
          std::cerr << "famseq_bn_call_batch failed (" << rc << "): " << famseq_last_error(ctx) << std::endl;
          ok = false;
        }
      }
      if (ok) to_writer.put(i);
      else to_reader.put(i);  // keep the reader from waiting on a batch that will never be written
    }
    to_writer.put(-1);
    reader.join();
    writer.join();
    for (PlBatch &b : bt) b.release();
    if (map) ::munmap(const_cast<char *>(map), (size_t)st.st_size);
    ::close(fd);
    if (o.bin_output) {
      if (ofd >= 0) {
        write_ok = write_ok && ::pwrite(ofd, &written, 8, 16) == 8;
        ::close(ofd);
      }
    }
    fout.close();
    if (std::getenv("FAMSEQ_TIMING"))
      std::cerr << "FamSeq PL: " << total << " sites, pipeline " << now_s() - t_start << " s (busy: reader " << t_read << ", GPU calls "
                << t_gpu << ", writer " << t_write << "); pinned buffers " << t_alloc << " s" << std::endl;
    famseq_destroy(ctx);
    return ok && write_ok && !fout.fail();
  }

  // `FamSeq unpack`: the result file's blocks set the pace; no GPU, a plain loop
  size_t cap = batch_capacity();
  vector<char> raw(cap * rec);
  vector<uint8_t> status(cap);
  vector<uint16_t> pl(cap * k * 3);
  vector<double> gpp(cap * k * 3), fpp(cap * k * 3);
  vector<int8_t> fgt(cap * k);
  bool ok = true;
  while (ok) {
    uint64_t bn = 0;
    fpo.read(reinterpret_cast<char *>(&bn), 8);
    if (!fpo || bn == 0) break;
    if (bn > cap) {
      cap = (size_t)bn;
      raw.resize(cap * rec); status.resize(cap); pl.resize(cap * k * 3);
      gpp.resize(cap * k * 3); fpp.resize(cap * k * 3); fgt.resize(cap * k);
    }
    const size_t n = (size_t)bn;
    fin.read(raw.data(), (std::streamsize)(n * rec));
    fpo.read(reinterpret_cast<char *>(status.data()), (std::streamsize)n);
    fpo.read(reinterpret_cast<char *>(gpp.data()), (std::streamsize)(n * k * 24));
    fpo.read(reinterpret_cast<char *>(fpp.data()), (std::streamsize)(n * k * 24));
    fpo.read(reinterpret_cast<char *>(fgt.data()), (std::streamsize)(n * k));
    if (!fpo || size_t(fin.gcount()) != n * rec) {
      std::cout << "The packed files end early or do not belong together." << std::endl;
      ok = false;
      break;
    }
    parallel_for(n, [&](size_t s) {  // the PED-matched columns of pl[]
      const char *r = raw.data() + s * rec;
      for (size_t j = 0; j < k; j++) std::memcpy(&pl[(s * k + j) * 3], r + 1 + 6 * seq_cols[j], 6);
    });
    format_and_write(n, pl.data(), status.data(), gpp.data(), fpp.data(), fgt.data(), nullptr);
  }
  fout.close();
  return ok && !fout.fail();
}

// ---- VCF driver ----------------------------------------------------------------------------

int chrom_number(const string &c) {  // file.cpp:321-343
  if (c == "X" || c == "chrX") return 23;
  if (c == "Y" || c == "chrY") return 24;
  if (c == "MT") return 25;
  return std::atoi(c.compare(0, 3, "chr") == 0 ? c.c_str() + 3 : c.c_str());
}

// The input as std::getline would cut it — lines up to the next '\n', a last line without one counts — over the
// memory-mapped file: no copy per line, and what the output echoes of a line (columns 1-9, the sample fields) is
// pointed at in place until the batch is written.  What cannot be mapped (a pipe) is read whole.
struct LineSource {
  const char *cur = nullptr, *end = nullptr;
  void *map = nullptr;
  size_t map_len = 0;
  string owned;
  bool open(const string &path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void *mp = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (mp != MAP_FAILED) {
        ::madvise(mp, (size_t)st.st_size, MADV_SEQUENTIAL);
        map = mp, map_len = (size_t)st.st_size;
        cur = static_cast<const char *>(mp), end = cur + map_len;
        ::close(fd);
        return true;
      }
    }
    // Not a regular file (a pipe, /dev/stdin) or one that cannot be mapped: read whole into memory — the parsed items
    // point into the input text until their block is written, which a bounded read-ahead buffer would have to guarantee
    // block by block.  A piped VCF therefore needs memory of its size; give large inputs as files (the usage text says so).
    char buf[1 << 16];
    ssize_t r;
    while ((r = ::read(fd, buf, sizeof buf)) > 0) owned.append(buf, size_t(r));
    ::close(fd);
    cur = owned.data(), end = cur + owned.size();
    return true;
  }
  bool next(std::string_view &line) {
    if (cur >= end) return false;
    const char *nl = static_cast<const char *>(std::memchr(cur, '\n', size_t(end - cur)));
    line = std::string_view(cur, size_t((nl ? nl : end) - cur));
    cur = nl ? nl + 1 : end;
    return true;
  }
  ~LineSource() {
    if (map) ::munmap(map, map_len);
  }
};

// run `f(t)` on threads 0..n-1 (the caller's thread takes 0)
template <class F>
void on_threads(int n, F f) {
  vector<std::thread> pool;
  for (int t = 1; t < n; ++t) pool.emplace_back([=] { f(t); });
  f(0);
  for (std::thread &th : pool) th.join();
}

bool run_vcf(const Options &o, const Ped &ped) {
  LineSource fin;
  if (!fin.open(o.vcf_files[0])) {
    std::cout << "Cannot open " << o.vcf_files[0] << std::endl;
    return false;
  }
  CliModel m;
  // defaults needed for the header before the ctx exists
  {
    vector<uint8_t> all(ped.n(), 1);
    if (m.init(ped, all, o.mrate, o.lrc) != 0) {
      std::cout << "Cannot initiate family. Please check ped file." << std::endl << "Cannot set family." << std::endl;
      return false;
    }
    if (o.gN.size() == 3) std::copy(o.gN.begin(), o.gN.end(), m.p.genoProbN);
    if (o.gK.size() == 3) std::copy(o.gK.begin(), o.gK.end(), m.p.genoProbK);
    if (o.gXN.size() == 3) std::copy(o.gXN.begin(), o.gXN.end(), m.p.genoProbXN);
    if (o.gXK.size() == 3) std::copy(o.gXK.begin(), o.gXK.end(), m.p.genoProbXK);
  }
  // pack mode writes a binary file: the text header goes nowhere
  std::ofstream fout;
  if (!o.pack_mode) fout.open(o.out_file.c_str());

  // ---- header (file.cpp:143-196): lines are echoed with a one-line lag
  auto format_tags = [&](bool fallback) {
    fout << "##FORMAT=<ID=GPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
         << (fallback ? "calbulated by Single Method" : "calculated by individual-based Method") << "\">" << std::endl;
    fout << "##FORMAT=<ID=FPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
            "calculated by FamSeqPro\">" << std::endl;
    fout << "##FORMAT=<ID=FGT,Number=1,Type=String,Description=\"Genotype called by FamSeqPro\">" << std::endl;
  };
  auto fs_info = [&] {
    fout << "##FS mutation rate=" << o.mrate << " " << std::endl;
    fout << "##FS genotype frequency in pupulation (Rare): "; put_triple(fout, m.p.genoProbN); fout << std::endl;
    fout << "##FS genotype frequency in population (Common): "; put_triple(fout, m.p.genoProbK); fout << std::endl;
    fout << "##FS genotype frequency for chromosome X of male in population (Rare): "; put_triple(fout, m.p.genoProbXN); fout << std::endl;
    fout << "##FS genotype frequency for chromosome X of male in population (Common): "; put_triple(fout, m.p.genoProbXK); fout << std::endl;
  };
  string title;
  std::string_view line;
  bool need_tags = true, need_info = true, have_line = false;
  auto after_hashes = [](std::string_view l, size_t n) { return l.size() > 2 ? l.substr(2, n) : std::string_view(); };
  while (fin.next(line)) {
    if (!line.empty() && line[0] == '#') {
      if (title.size() > 2) {
        fout << title << std::endl;
        if (title.compare(2, 6, "FORMAT") == 0 && after_hashes(line, 4) == "INFO" && need_tags) {
          format_tags(false);
          need_tags = false;
        }
        if (after_hashes(line, 6) == "contig" && need_info) {
          fs_info();
          need_info = false;
        }
      }
      title.assign(line);
      continue;
    }
    have_line = true;
    break;
  }
  if (need_tags) format_tags(true);
  if (need_info) fs_info();

  // ---- column mapping (file.cpp:200-232)
  const vector<string> head = split_keep(title, '\t');
  if (head.size() < 9) {
    std::cerr << "The vcf file has no #CHROM header line." << std::endl;
    return false;
  }
  const size_t ncol = head.size() - 9;
  vector<int> v2p(ncol, -1);
  vector<uint8_t> sequenced(ped.n(), 0);
  for (size_t i = 0; i < ncol; i++)
    for (int j = 0; j < ped.n(); j++)
      if (head[9 + i] == ped.name[j]) {
        v2p[i] = j;
        sequenced[j] = 1;
        break;
      }
  vector<int> seq_cols;
  vector<int32_t> seq_members;
  for (size_t i = 0; i < ncol; i++)
    if (v2p[i] >= 0) {
      seq_cols.push_back((int)i);
      seq_members.push_back(v2p[i]);
    }
  for (int i = 0; i < 9; i++) fout << head[i] << '\t';
  for (int c : seq_cols) fout << head[9 + c] << '\t';
  fout << std::endl;

  // ---- location filter (file.cpp:235-298)
  vector<vector<int>> loc(25);
  const bool use_loc = !o.loc_file.empty();
  if (use_loc) {
    std::ifstream fl(o.loc_file.c_str());
    if (!fl.is_open()) {
      std::cout << "Cannot open " << o.loc_file << std::endl;
      return false;
    }
    string l;
    while (std::getline(fl, l)) {
      if (l.size() < 2) break;
      const vector<string> t = split_keep(l, '\t');
      if (t.size() < 2) continue;
      int c = t[0] == "X" ? 23 : t[0] == "Y" ? 24 : t[0] == "MT" ? 25 : std::atoi(t[0].c_str());
      const int p = std::atoi(t[1].c_str());
      if (c < 1 || c > 25 || p == 0) continue;
      loc[c - 1].push_back(p);
    }
    for (auto &v : loc) std::sort(v.begin(), v.end());
  }

  PackWriter packer;
  famseq_ctx *ctx = nullptr;
  if (o.pack_mode) {
    vector<string> names;
    for (int c : seq_cols) names.push_back(head[9 + c]);
    if (names.empty() || !packer.open(o.out_file, names)) {
      std::cout << "Cannot write " << o.out_file << " (or no sample of the vcf file is in the ped file)." << std::endl;
      return false;
    }
  } else if (seq_cols.empty()) {  // nothing to compute and nothing to print per sample: say so instead of walking the file
    std::cout << "No sample of the vcf file is in the ped file (the names in the ped file's fifth column must match the vcf header)." << std::endl;
    return false;
  }
  // HIP start-up (runtime initialisation, context, streams, code objects: about a quarter of a second) runs on its own
  // thread while this one cuts and parses the first block(s); the flusher thread, the first to need the context, waits for it.
  // Until it is there the blocks are parsed into ordinary memory (pinned buffers need the runtime) and copied over — 6 bytes
  // per sample and site — once it is.
  std::atomic<bool> hip_up{false};
  std::future<famseq_ctx *> ctx_future;
  if (!o.pack_mode)
    ctx_future = std::async(std::launch::async, [&] {
      famseq_ctx *c = make_ctx(o, ped, sequenced, m);
      hip_up = true;
      return c;
    });
  const PlTable pl;
  const size_t n_seq = seq_cols.size(), N3 = size_t(3) * ped.n();
  const bool as_text = device_text();

  // What one parsing thread makes of its contiguous range of a block's lines, in input order.  The same thread
  // formats the same range once the block's sites are back from the GPU.
  struct Item {
    const char *raw;       // site: the input line (mapped; columns 1-9 and the sample fields are printed from it)
    uint32_t raw_len;
    uint32_t prefix_len;   // site: columns 1-9 without the tab after FORMAT
    uint32_t n_fmt;        // site: FORMAT keys (a missing sample prints that many "NA:")
    int32_t site;          // the line's index in the block = its site in the batch; -1: an echoed line, text = echo[echo_off, +echo_len)
    uint32_t echo_off, echo_len;
    uint32_t smp;          // site: its samples are samples[smp, +n_seq)
  };
  struct Part {
    vector<Item> items;
    vector<char> echo;
    vector<Record::Sample> samples;      // n_seq per site
    vector<uint32_t> explicit_sites;     // sites with a PL/GL field that is not a plain integer ...
    vector<double> explicit_lk;          // ... and their N x 3 likelihoods (the batch then goes in as fp64)
    bool any_failed = false;
    void clear() {
      items.clear(), echo.clear(), samples.clear(), explicit_sites.clear(), explicit_lk.clear();
      any_failed = false;
    }
  };

  // One input line -> an Item (or nothing).  Pure function of the line (no shared state).  Line q of a block is site q of
  // the batch: a site's flag byte and packed PLs are written where the GPU call will read them (flags_q, pl16: the
  // caller has set them to "all samples missing", which is what a line that is no site stays — computed and ignored;
  // compacting the sites first cost a pass over the batch, and the GPU idles nine tenths of the loop).
  auto parse_line = [&](std::string_view line, size_t q, Part &out, uint8_t *flags_q, uint16_t *pl16) {
    if (line[0] == '#') return;
    thread_local vector<std::string_view> t, fmt, sub;
    thread_local vector<double> lkrow;
    split_views(line, '\t', t);
    if (t.size() < 9 + ncol) return;  // malformed line (the reference would read out of bounds)
    auto echo = [&] {
      if (o.pack_mode) return;
      const size_t at = out.echo.size();
      for (int i = 0; i < 9; i++) {
        out.echo.insert(out.echo.end(), t[i].begin(), t[i].end());
        out.echo.push_back('\t');
      }
      for (int c : seq_cols) {
        out.echo.insert(out.echo.end(), t[9 + c].begin(), t[9 + c].end());
        out.echo.push_back('\t');
      }
      out.items.push_back(Item{nullptr, 0, 0, 0, -1, uint32_t(at), uint32_t(out.echo.size() - at), 0});
    };
    if (use_loc) {
      const int c = chrom_number(string(t[0]));
      const int p = std::atoi(string(t[1]).c_str());
      if (c < 1 || c > 25 || p == 0) return;
      if (!std::binary_search(loc[c - 1].begin(), loc[c - 1].end(), p)) return;
    }
    // site rules, in the reference's order (file.cpp:362-473)
    if (t[3] == "." || t[3] == "-" || t[3].size() != 1 || t[4].size() != 1) {
      if (o.all_line) echo();
      return;
    }
    if (o.var_only && (t[4] == "." || t[4] == "-")) return;
    if (t[0] == "Y" || t[0] == "chrY" || t[0] == "MT") {
      if (o.all_line) echo();
      return;
    }
    const bool is_x = t[0] == "X" || t[0] == "chrX" || t[0] == "CHRX";
    const string chrom(t[0]);
    const int cn = std::atoi(chrom.compare(0, 3, "chr") == 0 ? chrom.c_str() + 3 : chrom.c_str());
    if (!((0 < cn && cn < 23) || is_x)) {
      if (o.all_line) echo();
      return;
    }
    const uint8_t flags = uint8_t((t[2] != "." ? FAMSEQ_FLAG_KNOWN : 0) | (is_x ? FAMSEQ_FLAG_CHRX : 0));
    split_views(t[8], ':', fmt);
    size_t n_miss = 0;
    for (int c : seq_cols) n_miss += t[9 + c].size() < 5;
    if (n_miss == n_seq) {
      if (o.all_line) echo();
      return;
    }
    int i_pl = -1;
    for (size_t k = 0; k < fmt.size(); k++)
      if (fmt[k] == "PL" || fmt[k] == "GL") i_pl = (int)k;
    if (i_pl < 0) {  // echoed unchanged even without -a (file.cpp:541-555, :770-784)
      echo();
      return;
    }
    const char *base = line.data();
    out.items.push_back(Item{base, uint32_t(line.size()), uint32_t(t[8].data() + t[8].size() - base), uint32_t(fmt.size()),
                             int32_t(q), 0, 0, uint32_t(out.samples.size())});
    *flags_q = flags;
    lkrow.assign(N3, 1.0);
    bool integral = true;
    size_t col = 0;
    for (int c : seq_cols) {
      const size_t this_col = col++;
      const std::string_view f = t[9 + c];
      out.samples.push_back({uint32_t(f.data() - base), uint32_t(f.size()), f.size() < 5});
      if (f.size() < 5) continue;  // missing sample: flat likelihood (file.cpp:927-933)
      split_views(f, ':', sub);
      if (sub.size() != fmt.size()) continue;  // row stays {1,1,1} (file.cpp:573-578)
      const std::string_view pls = sub[i_pl];
      const char *b = pls.data(), *end = b + pls.size();
      for (int g = 0; g < 3; g++) {
        const char *e = static_cast<const char *>(std::memchr(b, ',', size_t(end - b)));
        if (!e) e = end;
        lkrow[size_t(3) * v2p[c] + g] = pl(b, e, &pl16[3 * this_col + g], &integral);
        b = e < end ? e + 1 : end;
      }
    }
    if (!integral) {
      out.explicit_sites.push_back(uint32_t(q));
      out.explicit_lk.insert(out.explicit_lk.end(), lkrow.begin(), lkrow.end());
    }
  };

  // ---- the pipeline.  A block of input lines is ONE batch: this thread cuts it into lines, has it parsed on all
  // cores (each thread its contiguous range, into its own Part and into its range of the batch's pinned arrays)
  // and hands the block to the flusher thread — GPU call (famseq_bn_call_batch: posterior, Phred scaling and
  // genotype call on the device), formatting (each thread the range it parsed), one write per thread — while it goes on
  // with the next block in the other slot.  (Before: lines copied one by one out of an ifstream, parsed results moved
  // one by one into the batch by this thread, 0.94 of the loop's 1.0 s per 1 M ten-member sites.)
  const size_t block = std::min<size_t>(batch_capacity(), size_t(1) << 16);
  unsigned n_threads = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("FAMSEQ_THREADS")) n_threads = (unsigned)std::atoi(e);
  n_threads = std::max(1u, std::min(n_threads, 32u));
  // A block is cut into n_threads parts; how many threads walk them while parsing and while formatting can be set apart
  // (formatting is the loop's critical path and runs beside the next block's parsing: tools/cli_throughput.py)
  unsigned parse_threads = n_threads, format_threads = n_threads;
  if (const char *e = std::getenv("FAMSEQ_PARSE_THREADS")) parse_threads = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("FAMSEQ_FORMAT_THREADS")) format_threads = std::max(1, std::atoi(e));
  auto on_parts = [](int n_parts, unsigned threads, const std::function<void(int)> &f) {
    const int nt = (int)std::min<unsigned>(threads, (unsigned)n_parts);
    on_threads(nt, [&](int j) {
      for (int t = j; t < n_parts; t += nt) f(t);
    });
  };
  struct Slot {
    vector<std::string_view> lines;
    vector<Part> parts;
    int n_parts = 0;
    size_t n_sites = 0;  // sites of the batch: the block's lines, or (large pedigrees) its sites moved together
    bool packed = true;
    bool staged = false;        // parsed into pk_pl / pk_flags because the HIP runtime was not up yet: the flusher copies them over
    vector<uint16_t> pk_pl;     // pack mode (and staged blocks): the block's arrays live here, nothing is pinned
    vector<uint8_t> pk_flags;
    PlBatch io;          // pinned: pl + flags in, gpp / fpp / fgt / status out
    vector<double> lk;   // fp64 input, only for a block with a non-integer PL/GL field
    vector<TextBuf> text;
  } slots[4];  // one being cut and parsed, one at the GPU, one being formatted, one being written
  bool ok = true;
  for (Slot &sl : slots) sl.parts.resize(n_threads), sl.text.resize(n_threads);

  double t_lines = 0, t_parse = 0, t_gather = 0, t_stall = 0, t_gpu = 0, t_format = 0, t_write = 0, t_ctx_wait = 0;
  int compact_from = 12;  // members from which a block's sites are moved together before the GPU call (below)
  if (const char *e = std::getenv("FAMSEQ_COMPACT_FROM")) compact_from = std::atoi(e);  // test aid
  Channel to_flusher, to_formatter, to_writer, to_driver;
  std::atomic<bool> flush_ok{true};
  std::thread flusher([&] {
    for (;;) {
      const int i = to_flusher.take();
      if (i < 0) break;
      Slot &sl = slots[i];
      if (!ctx && ctx_future.valid()) {  // the first block: the context has been coming up meanwhile
        const double tw = now_s();
        ctx = ctx_future.get();
        t_ctx_wait = now_s() - tw;
        if (!ctx) flush_ok = false;
      }
      if (sl.staged && flush_ok) {  // parsed before the runtime was up: into the pinned arrays now
        const size_t nl_ = sl.pk_flags.size();
        if (sl.io.cap < nl_) {
          sl.io.release();
          sl.io = PlBatch();
          if (!sl.io.alloc(nl_, std::max<size_t>(n_seq, 1), as_text)) {
            std::cerr << "cannot allocate pinned host buffers" << std::endl;
            flush_ok = false;
          }
        }
        if (flush_ok) {
          std::memcpy(sl.io.pl, sl.pk_pl.data(), nl_ * 6 * n_seq);
          std::memcpy(sl.io.flags, sl.pk_flags.data(), nl_);
        }
      }
      const double t0 = now_s();
      if (sl.n_sites > 0 && flush_ok) {
        const int rc = sl.io.call(ctx, sl.n_sites, sl.packed ? nullptr : sl.lk.data(), seq_members.data());
        if (rc != 0) {
          std::cerr << "famseq_bn_call_batch failed (" << rc << "): " << famseq_last_error(ctx) << std::endl;
          flush_ok = false;
        }
      }
      t_gpu += now_s() - t0;
      to_formatter.put(i);
    }
    to_formatter.put(-1);
  });
  // ... the results are turned into text on their own thread (and its helpers) while the next block is at the GPU ...
  std::thread formatter([&] {
    for (;;) {
      const int i = to_formatter.take();
      if (i < 0) break;
      Slot &sl = slots[i];
      const size_t k = n_seq;
      const double t1 = now_s();
      if (flush_ok) {
        on_parts(sl.n_parts, format_threads, [&](int t) {
          Part &pt = sl.parts[t];
          TextBuf &out = sl.text[t];
          out.n = 0;
          for (const Item &it : pt.items) {
            if (it.site < 0) {
              out.put(pt.echo.data() + it.echo_off, it.echo_len);
            } else {
              const size_t s = size_t(it.site);
              const Record::Sample *sm = &pt.samples[it.smp];
              out.put(it.raw, it.prefix_len);  // columns 1-8 + FORMAT
              out.put(":GPP:FPP:FGT\t", 13);
              if (sl.io.status[s] & 3) {  // file.cpp:607-620
                pt.any_failed = true;
                for (size_t j = 0; j < k; ++j) {
                  out.put(it.raw + sm[j].off, sm[j].len);
                  out.put(":NA:NA:NA\t", 10);
                }
              } else {
                for (size_t j = 0; j < k; ++j) {
                  if (sm[j].missing) {
                    for (uint32_t q = 0; q < it.n_fmt; ++q) out.put("NA:", 3);
                  } else {
                    out.put(it.raw + sm[j].off, sm[j].len);
                    out.ch(':');
                  }
                  if (sl.io.text) {
                    out.record(sl.io.text + (s * k + j) * FAMSEQ_TEXT_STRIDE);
                    continue;
                  }
                  out.triples(&sl.io.gpp[(s * k + j) * 3], &sl.io.fpp[(s * k + j) * 3]);
                  const int gt = sl.io.fgt[s * k + j];
                  out.put(gt == 0 ? "0/0\t" : (gt == 1 ? "0/1\t" : "1/1\t"), 4);
                }
              }
            }
            out.ch('\n');
          }
        });
        t_format += now_s() - t1;
      }
      to_writer.put(i);
    }
    to_writer.put(-1);
  });
  // ... and the text goes out on a third thread, so that block k + 1 is at the GPU and being formatted while block k is
  // written (formatting 0.50 s + writing 0.25 s per 3 M ten-member sites were one thread's work in round 2)
  std::thread writer([&] {
    for (;;) {
      const int i = to_writer.take();
      if (i < 0) break;
      Slot &sl = slots[i];
      if (flush_ok) {
        const double t2 = now_s();
        for (int t = 0; t < sl.n_parts; ++t) {
          const Part &pt = sl.parts[t];
          if (pt.any_failed)  // the reference's warning on stdout, in input order (file.cpp:607-612)
            for (const Item &it : pt.items)
              if (it.site >= 0 && (sl.io.status[it.site] & 3))
                std::cout << "Warning: this variant hasn't been calculated: " << std::endl << std::string_view(it.raw, it.raw_len) << std::endl;
        }
        // (one pwrite per thread at its own offset was tried: 0.33 s per 1.47 GB as well — the page cache, not the stream)
        for (int t = 0; t < sl.n_parts; ++t) fout.write(sl.text[t].v.data(), (std::streamsize)sl.text[t].n);
        if (fout.fail()) {
          std::cerr << "cannot write " << o.out_file << std::endl;
          flush_ok = false;
        }
        t_write += now_s() - t2;
      }
      to_driver.put(i);
    }
  });

  const double t_begin = now_s();
  to_driver.put(0), to_driver.put(1), to_driver.put(2), to_driver.put(3);
  bool more = have_line;
  while (more && ok && flush_ok) {
    double t0 = now_s();
    const int i = to_driver.take();  // a slot whose previous block is written
    Slot &sl = slots[i];
    double t1 = now_s();
    t_stall += t1 - t0;
    sl.lines.clear();
    while (more && sl.lines.size() < block) {  // `line` holds the next unread line
      if (line.size() < 2) {                   // the reference stops at the first empty line
        more = false;
        break;
      }
      sl.lines.push_back(line);
      more = fin.next(line);
    }
    const size_t nl = sl.lines.size();
    if (nl == 0) {
      to_driver.put(i);
      break;
    }
    sl.n_parts = nl < 2048 ? 1 : (int)n_threads;
    double t2 = now_s();
    t_lines += t2 - t1;
    sl.staged = !o.pack_mode && !hip_up;
    if (o.pack_mode || sl.staged) {
      sl.pk_pl.resize(nl * 3 * n_seq), sl.pk_flags.resize(nl);
    } else if (sl.io.cap < nl) {  // pinned, sized by the first block (a short file does not pay for 65,536 sites)
      sl.io.release();
      sl.io = PlBatch();
      if (!sl.io.alloc(nl, std::max<size_t>(n_seq, 1), as_text)) {
        std::cerr << "cannot allocate pinned host buffers" << std::endl;
        ok = false;
        to_driver.put(i);
        break;
      }
    }
    uint16_t *const pl_arr = o.pack_mode || sl.staged ? sl.pk_pl.data() : sl.io.pl;
    uint8_t *const flags_arr = o.pack_mode || sl.staged ? sl.pk_flags.data() : sl.io.flags;
    on_parts(sl.n_parts, parse_threads, [&](int t) {
      Part &pt = sl.parts[t];
      pt.clear();
      const size_t lo = nl * t / sl.n_parts, hi = nl * (t + 1) / sl.n_parts;
      std::memset(flags_arr + lo, 0, hi - lo);
      std::memset(pl_arr + lo * 3 * n_seq, 0xFF, (hi - lo) * 6 * n_seq);  // 0xFFFF x3 = missing sample
      for (size_t q = lo; q < hi; ++q) parse_line(sl.lines[q], q, pt, flags_arr + q, pl_arr + q * 3 * n_seq);
    });
    double t3 = now_s();
    t_parse += t3 - t2;
    bool packed = true;
    size_t real = 0;
    for (int t = 0; t < sl.n_parts; ++t) {
      packed = packed && sl.parts[t].explicit_sites.empty();
      real += sl.parts[t].samples.size();
    }
    sl.n_sites = real ? nl : 0, sl.packed = packed;  // a block without a single site (or without a sequenced sample) calls nothing
    if (o.pack_mode) {  // records of the integer sites, in input order; nothing goes to the GPU
      for (int t = 0; t < sl.n_parts; ++t) {
        const Part &pt = sl.parts[t];
        size_t x = 0;
        for (const Item &it : pt.items) {
          if (it.site < 0) continue;
          if (x < pt.explicit_sites.size() && pt.explicit_sites[x] == uint32_t(it.site)) {
            ++x, packer.skipped++;
            continue;
          }
          packer.add(flags_arr[it.site], pl_arr + size_t(it.site) * 3 * n_seq);
        }
      }
      t_gather += now_s() - t3;
      to_driver.put(i);
      continue;
    }
    // A line that is no site costs the GPU a site's work: nothing next to the text work for the usual pedigree, but
    // 3^N configurations each.  From twelve members on (0.5 M configurations) the sites are moved together first —
    // in place and in order, one thread: 60 bytes per site.
    size_t n_real = 0;
    for (int t = 0; t < sl.n_parts; ++t)
      for (const Item &it : sl.parts[t].items) n_real += it.site >= 0;
    if (n_real > 0 && n_real < nl && ped.n() >= compact_from) {
      size_t d = 0;
      for (int t = 0; t < sl.n_parts; ++t) {
        Part &pt = sl.parts[t];
        size_t x = 0;
        for (Item &it : pt.items) {
          if (it.site < 0) continue;
          const size_t q = size_t(it.site);
          if (d != q) {
            flags_arr[d] = flags_arr[q];
            std::memmove(pl_arr + d * 3 * n_seq, pl_arr + q * 3 * n_seq, 6 * n_seq);
          }
          if (x < pt.explicit_sites.size() && pt.explicit_sites[x] == q) pt.explicit_sites[x++] = uint32_t(d);
          it.site = int32_t(d++);
        }
      }
      sl.n_sites = d;
    }
    if (!packed) {
      // fp64 rows: the table's value for every integer field (what the parser computed for it), the parsed rows of the others
      const size_t n = sl.n_sites;
      sl.lk.resize(n * N3);
      on_threads(sl.n_parts, [&](int t) {
        const size_t lo = n * t / sl.n_parts, hi = n * (t + 1) / sl.n_parts;
        double *rows = sl.lk.data();
        std::fill(rows + lo * N3, rows + hi * N3, 1.0);
        for (size_t q = lo; q < hi; ++q)
          for (size_t j = 0; j < n_seq; ++j) {
            const uint16_t *p = &pl_arr[(q * n_seq + j) * 3];
            if (p[0] == 0xFFFF && p[1] == 0xFFFF && p[2] == 0xFFFF) continue;
            for (int g = 0; g < 3; ++g) rows[q * N3 + size_t(3) * seq_members[j] + g] = pl.value(p[g]);
          }
      });
      for (int t = 0; t < sl.n_parts; ++t) {
        const Part &pt = sl.parts[t];
        for (size_t x = 0; x < pt.explicit_sites.size(); ++x)
          std::copy(&pt.explicit_lk[x * N3], &pt.explicit_lk[(x + 1) * N3], sl.lk.data() + size_t(pt.explicit_sites[x]) * N3);
      }
    }
    t_gather += now_s() - t3;
    to_flusher.put(i);
  }
  to_flusher.put(-1);
  flusher.join();
  formatter.join();
  writer.join();
  if (!ctx && ctx_future.valid()) ctx = ctx_future.get();  // a file without a single data line: nobody asked for it yet
  ok = ok && flush_ok && (o.pack_mode || ctx);
  if (std::getenv("FAMSEQ_TIMING") && !o.pack_mode)
    std::cerr << "FamSeq vcf: loop " << now_s() - t_begin << " s; this thread: cutting lines " << t_lines << ", parsing " << t_parse
              << ", fp64 rows / pack records " << t_gather << ", waiting for the flusher " << t_stall << "; flusher thread: GPU calls " << t_gpu
              << ", formatting " << t_format << ", writing " << t_write << ", waiting for the context " << t_ctx_wait << std::endl;
  if (!o.pack_mode)
    for (Slot &sl : slots) sl.io.release();
  if (o.pack_mode) {
    std::cout << packer.n_sites << " sites packed";
    if (packer.skipped) std::cout << ", " << packer.skipped << " skipped (PL/GL field is not a plain integer)";
    std::cout << std::endl;
    return packer.close();
  }
  fout.close();
  famseq_destroy(ctx);
  return ok && !fout.fail();
}

// ---- LK driver (file.cpp:1640-1886) ----------------------------------------------------------

bool run_lk(const Options &o, const Ped &ped) {
  std::ifstream fin(o.lk_file.c_str());
  if (!fin.is_open()) {
    std::cout << "Cannot open " << o.lk_file << std::endl;
    return false;
  }
  string title, line;
  std::getline(fin, title);
  const vector<string> head = split_keep(title, '\t');
  vector<int> v2p(head.size(), -1);
  vector<uint8_t> sequenced(ped.n(), 0);
  for (size_t i = 0; i < head.size(); i++)
    for (int j = 0; j < ped.n(); j++)
      if (head[i] == ped.name[j]) {
        v2p[i] = j;
        sequenced[j] = 1;
        break;
      }
  CliModel m;
  famseq_ctx *ctx = make_ctx(o, ped, sequenced, m);
  if (!ctx) return false;
  std::ofstream fout(o.out_file.c_str());
  fout << "##FORMAT=<ID=GPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
          "calculated by individual-base Method\">" << std::endl;
  fout << "##FORMAT=<ID=FPP,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled for posterior probability "
          "calculated by FamSeqPro\">" << std::endl;
  fout << "##FORMAT=<ID=FGT,Number=1,Type=String,Description=\"Genotype called by FamSeqPro\">" << std::endl;
  fout << "##FS mutation rate=" << o.mrate << " " << std::endl;
  fout << "##FS genotype frequency in pupulation: "; put_triple(fout, m.p.genoProbN); fout << std::endl;
  vector<int> seq_cols, seq_members;
  fout << "#FORMAT\t";
  for (size_t i = 0; i < head.size(); i++)
    if (v2p[i] >= 0) {
      seq_cols.push_back((int)i);
      seq_members.push_back(v2p[i]);
      fout << head[i] << '\t';
    }
  fout << std::endl;
  BatchCaller caller(ctx, ped.n(), seq_members, fout, batch_capacity());
  vector<double> lk(size_t(3) * ped.n());
  bool ok = true;
  while (ok && std::getline(fin, line)) {
    if (line.size() < 2) break;
    vector<std::string_view> t;
    split_views(line, '\t', t);
    if (!seq_cols.empty() && t.size() <= size_t(seq_cols.back())) continue;  // short row
    std::fill(lk.begin(), lk.end(), 1.0);
    Record r;
    r.head = "LK:GPP:FPP:FGT\t";
    for (int c : seq_cols) {
      r.samples.push_back({uint32_t(t[c].data() - line.data()), uint32_t(t[c].size()), false});
      const vector<string> v = split_keep(string(t[c]), ',');
      for (int g = 0; g < 3 && g < (int)v.size(); g++) {
        double x = std::atof(v[g].c_str());
        if (o.lk_type == 2) x = std::pow(10.0, x);
        else if (o.lk_type == 3) x = std::exp(x);
        else if (o.lk_type == 4) x = std::pow(10.0, -x / 10.0);
        lk[size_t(3) * v2p[c] + g] = x;
      }
    }
    r.raw = line;
    ok = caller.site(std::move(r), lk, nullptr, 0);  // calPostProbBN() defaults: Known=false, chrType=0 (file.cpp:1751)
  }
  ok = ok && caller.flush();
  famseq_destroy(ctx);
  return ok;
}

// ---- `FamSeq tune -vcfFile f -pedFile p`: the generated kernels depend on the pedigree and on which of its members the
// vcf file has columns for; time their variants on this GPU (famseq_set_option "tune") and leave the picks in the cache.
bool run_tune(const Options &o, const Ped &ped) {
  LineSource fin;
  if (!fin.open(o.vcf_files[0])) {
    std::cout << "Cannot open " << o.vcf_files[0] << std::endl;
    return false;
  }
  std::string_view line;
  string title;
  while (fin.next(line) && !line.empty() && line[0] == '#') title.assign(line);
  const vector<string> head = split_keep(title, '\t');
  vector<uint8_t> sequenced(ped.n(), 0);
  for (size_t i = 9; i < head.size(); i++)
    for (int j = 0; j < ped.n(); j++)
      if (head[i] == ped.name[j]) sequenced[j] = 1;
  CliModel m;
  famseq_ctx *ctx = make_ctx(o, ped, sequenced, m);
  if (!ctx) return false;
  const int rc = famseq_set_option(ctx, "tune", 1);
  if (rc != 0) std::cout << "Tuning failed: " << famseq_last_error(ctx) << std::endl;
  else {
    const string plan = famseq_plan_json(ctx);
    const size_t k = plan.find("\"tune\":\"");
    std::cout << (k == string::npos ? plan : plan.substr(k + 8, plan.size() - k - 10)) << std::endl;
  }
  famseq_destroy(ctx);
  return rc == 0;
}

void usage_top() {
  std::cout << std::endl
            << "Program: FamSeq (Sequence calling using pedigree information), MI355X build of the -method 1 path"
            << std::endl << "Usage:\tFamSeq <input type> [options]" << std::endl << std::endl
            << "Input type: \tvcf\t\tinput vcf file" << std::endl << "\t\tLK\t\tinput likelihood file" << std::endl
            << std::endl << "Type FamSeq -h for help." << std::endl << std::endl;
}

void help() {
  std::cout << "Usage:\tFamSeq <input type> [options]" << std::endl << std::endl
            << "FamSeq vcf [options] for vcf input, FamSeq LK [options] for likelihood-only input." << std::endl
            << std::endl << "Options:" << std::endl << std::endl
            << "-vcfFile\tThe name of input vcf file." << std::endl
            << "-lkFile\t\tThe name of input likelihood only format file." << std::endl
            << "-lkType\t\tn:normal(default); log10: log10 scaled; ln: ln scaled; PS: phred scaled." << std::endl
            << "-pedFile\tThe name of the file storing the pedigree information." << std::endl
            << "-output\t\tThe name of output file" << std::endl
            << "-method\t\t1(default): Bayesian network enumeration; 2: exact sum-product." << std::endl
            << "-mRate\t\tMutation rate. The default value is 1e-7" << std::endl
            << "-v\t\tOnly record the position at which the genotype is not RR in the output file." << std::endl
            << "-a\t\tRecord all the position in the output file." << std::endl
            << "-l\t\tLocation file (chromosome<TAB>position): only these positions are processed." << std::endl
            << "-genoProbN\tPr(G) for autosome, variant not in dbSNP. Default 0.9985 0.001 0.0005." << std::endl
            << "-genoProbK\tPr(G) for autosome, variant in dbSNP. Default 0.45 0.1 0.45." << std::endl
            << "-genoProbXN\tPr(G) for chromosome X of males, not in dbSNP. Default 0.999 0.001." << std::endl
            << "-genoProbXK\tPr(G) for chromosome X of males, in dbSNP. Default 0.5 0.5." << std::endl
            << "-LRC\t\tLikelihood ratio criterion for the single-sample shortcut. Default 1." << std::endl
            << "pack\t\tFamSeq pack -vcfFile f -pedFile p -output f.fspl: write the computable sites as packed integer PLs." << std::endl
            << "PL\t\tFamSeq PL -plFile f.fspl -pedFile p -output o [-binOutput]: call variants from a packed PL file" << std::endl
            << "\t\t(-binOutput: write a packed result file instead of text)." << std::endl
            << "unpack\t\tFamSeq unpack -plFile f.fspl -poFile r.fspo -pedFile p -output o: the text of a packed result file." << std::endl
            << "tune\t\tFamSeq tune -vcfFile f -pedFile p: time this pedigree's kernel variants on this GPU once and keep the" << std::endl
            << "\t\tfaster ones' indices in the kernel cache (later runs start from them)." << std::endl
            << "Environment: FAMSEQ_DEVICE (GPU index, default 0), FAMSEQ_BATCH (sites per GPU batch)." << std::endl;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc == 1) {
    usage_top();
    return -1;
  }
  const string mode(argv[1]);
  if (mode == "-h") {
    help();
    return 0;
  }
  if (mode != "vcf" && mode != "LK" && mode != "pack" && mode != "PL" && mode != "unpack" && mode != "tune") {
    std::cout << "Cannot recognize the input type: \"" << argv[1] << "\"." << std::endl
              << "The input type can only be vcf or LK (or pack / PL / unpack for the packed binary formats, or tune)" << std::endl << std::endl
              << "Type FamSeq -h for help." << std::endl;
    return -1;
  }
  if (argc == 2) {
    std::cout << std::endl << "Usage:\tFamSeq " << mode << " [options]" << std::endl << std::endl
              << "Type FamSeq -h for help." << std::endl;
    return -1;
  }
  Options o;
  o.lk_mode = mode == "LK" || mode == "PL" || mode == "unpack";
  o.pl_mode = mode == "PL" || mode == "unpack";
  o.unpack_mode = mode == "unpack";
  o.pack_mode = mode == "pack";
  o.tune_mode = mode == "tune";
  const int rc = parse_options(argc, argv, o);
  if (rc < 0) return -1;
  if (rc > 0)
    std::cout << "There are some improper parameters in the command line. Some parameters are set to default."
              << std::endl;
  if (o.method == 3) {
    std::cout << "This build implements -method 1 (Bayesian network) and -method 2 (exact sum-product, "
                 "the result of Elston-Stewart peeling); -method 3 (MCMC) is not part of it." << std::endl;
    return -1;
  }
  Ped ped;
  if (!read_ped(o.ped_file, ped)) {
    std::cout << "Cannot read Ped file: " << o.ped_file << "." << std::endl << "Cannot set family." << std::endl;
    return -1;
  }
  const double t0 = now_s();
  if (o.tune_mode) return run_tune(o, ped) ? 0 : -1;
  const bool ok = o.pl_mode ? run_pl(o, ped) : (o.lk_mode ? run_lk(o, ped) : run_vcf(o, ped));
  if (std::getenv("FAMSEQ_TIMING")) std::cerr << "FamSeq " << mode << ": " << now_s() - t0 << " s in the driver" << std::endl;
  // Every output file has been closed by its driver.  Leave without the static destructors: unloading the HIP runtime takes
  // about a tenth of a second, which is a tenth of what a three-million-site file takes altogether.
#if defined(__SANITIZE_ADDRESS__)
  return ok ? 0 : -1;  // (the sanitizer build checks for leaks at exit)
#else
  std::cout.flush();
  std::cerr.flush();
  std::fflush(nullptr);
  std::_Exit(ok ? 0 : 255);
#endif
}
