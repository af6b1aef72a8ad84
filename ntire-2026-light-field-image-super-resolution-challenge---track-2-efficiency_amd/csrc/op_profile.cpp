// Operator-level timing hooks of liblfsr_hip.so: when switched on (lfsr_op_profile(1)) every instrumented entry point brackets its launches with a
// hipEvent pair on the stream it launches on and the pairs are aggregated per (operator, tag a, tag b) by lfsr_op_profile_read.  This is what bench.py
// uses to put the dominant kernel's in-run duration of EPIT / LFT / the training step into its JSON line (the reference times whole forwards only:
// check_efficiency_official.py:306-330).  Switched off (the default) a hook is one relaxed atomic load.
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "lfsr_internal.h"

std::atomic<int> g_lfsr_op_profile_on{0};

namespace {
struct Rec { const char* op; int a, b; hipEvent_t e0, e1; };
std::mutex g_mu;
std::vector<Rec> g_recs;
int g_gen = 0;      // bumped whenever g_recs is emptied: a timer that began in an older generation finds its record gone and records nothing
std::vector<hipEvent_t> g_free;

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

LfsrOpTimer::LfsrOpTimer(const char* op, int a, int b, hipStream_t st) : slot_(-1), gen_(0), st_(st) {
  if (!g_lfsr_op_profile_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{op, a, b, get_event(), get_event()};
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, st);
  slot_ = (int)g_recs.size();
  gen_ = g_gen;
  g_recs.push_back(r);
}

// A read or a switch on another thread between the two ends of a timer empties the table (and recycles the events): the generation says so and the
// timer then records nothing rather than stamping another operator's record.  Hooked times of operators that run on two streams at once (the
// branch backward of the training step) overlap in wall time: their sum exceeds the step's (bench.py reports shares of hooked time, not of the step).
LfsrOpTimer::~LfsrOpTimer() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (gen_ == g_gen && slot_ < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot_].e1, st_);
}

extern "C" {

int lfsr_op_profile(int enable) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) { g_free.push_back(r.e0); g_free.push_back(r.e1); }
  g_recs.clear();
  ++g_gen;
  g_lfsr_op_profile_on.store(enable ? 1 : 0);
  return LFSR_OK;
}

// text: one line "op a b total_ms launches" per (op, a, b); returns the number of bytes the full text needs (incl. the terminator), or < 0
long long lfsr_op_profile_read(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::tuple<std::string, int, int>, std::pair<double, long long>> agg;
  for (auto& r : g_recs) {
    hipError_t e = hipEventSynchronize(r.e1);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (e != hipSuccess) return LFSR_HIP_ERR(e);
    auto& v = agg[std::make_tuple(std::string(r.op), r.a, r.b)];
    v.first += ms; v.second += 1;
    g_free.push_back(r.e0); g_free.push_back(r.e1);
  }
  g_recs.clear();
  ++g_gen;
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s %d %d %.6f %lld\n", std::get<0>(kv.first).c_str(), std::get<1>(kv.first), std::get<2>(kv.first), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && cap > 0) {
    size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (long long)out.size() + 1;
}

}  // extern "C"
