// prof.cpp — per-launch HIP-event timing, aggregated by kernel name.  Events are recorded on the stream the kernel is
// launched on; elapsed times are read only in prof_end() (one synchronisation), so the timed stream is never stalled.
#include "prof.h"

#include <map>
#include <sstream>
#include <vector>

namespace sdmi {
namespace {
struct Rec {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_enabled() { return g_on; }
void prof_begin() { g_recs.clear(); g_on = true; }
void prof_mark_start(const char* name, double flops, double bytes, hipStream_t s) {
    Rec r{name, flops, bytes, take_event(), take_event()};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_mark_stop(hipStream_t s) {
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, s);
}
std::string prof_end() {
    g_on = false;
    (void)hipDeviceSynchronize();
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        Agg& a = agg[r.name];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
    std::ostringstream os;
    os.precision(10);
    os << "{\"kernels\": [";
    bool first = true;
    for (auto& kv : agg) {
        if (!first) os << ", ";
        first = false;
        os << "{\"name\": \"" << kv.first << "\", \"launches\": " << kv.second.n << ", \"ms\": " << kv.second.ms
           << ", \"flops\": " << kv.second.flops << ", \"bytes\": " << kv.second.bytes << "}";
    }
    os << "]}";
    return os.str();
}
}  // namespace sdmi
