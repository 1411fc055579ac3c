// Optional per-launch timing with HIP events recorded on the launch stream (used by bench.py to measure each
// kernel's average duration live; off by default and never active under graph capture).
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace ms {

static bool g_timing = false;
static std::vector<std::string> g_skip;   // ms_debug_set_skip: label substrings whose launches are dropped (timing ablations)
struct TimingRec { std::string label; hipEvent_t a, b; double flops, bytes; };
static std::vector<TimingRec> g_recs;
static std::mutex g_mu;

bool timing_enabled() { return g_timing; }

TimingScope::TimingScope(hipStream_t s, double flops, double bytes, const char* fmt, ...) : idx_(-1), s_(s), skip_(false) {
  if (!g_timing && g_skip.empty()) return;
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  for (auto& pat : g_skip)
    if (strstr(buf, pat.c_str())) { skip_ = true; return; }
  if (!g_timing) return;
  TimingRec r;
  r.label = buf; r.flops = flops; r.bytes = bytes;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(r);
  idx_ = (int)g_recs.size() - 1;
}

TimingScope::~TimingScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[idx_].b, s_);
}

}  // namespace ms

extern "C" int ms_debug_set_skip(const char* patterns) {
  std::lock_guard<std::mutex> lk(ms::g_mu);
  ms::g_skip.clear();
  if (!patterns) return 0;
  std::string p(patterns);
  size_t i = 0;
  while (i <= p.size()) {
    size_t j = p.find(';', i);
    if (j == std::string::npos) j = p.size();
    if (j > i) ms::g_skip.push_back(p.substr(i, j - i));
    i = j + 1;
  }
  return (int)ms::g_skip.size();
}

extern "C" int ms_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(ms::g_mu);
  for (auto& r : ms::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  ms::g_recs.clear();
  ms::g_timing = on != 0;
  return 0;
}

// Writes lines "label\tcount\ttotal_ms\tflops_per_launch\tbytes_per_launch\n"; returns the bytes needed
// (call again with a bigger buffer if the return value >= cap).
extern "C" size_t ms_timing_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(ms::g_mu);
  struct Agg { long count = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : ms::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    Agg& a = agg[r.label];
    a.count++; a.ms += t; a.flops = r.flops; a.bytes = r.bytes;
  }
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.0f\t%.0f\n", kv.first.c_str(), kv.second.count, kv.second.ms,
             kv.second.flops, kv.second.bytes);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}
