// prof.h — optional per-launch HIP-event profiler (used by bench.py's roofline block; off by default).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace sdmi {
bool prof_enabled();
void prof_begin();
// Aggregated JSON: {"kernels": [{"name":..., "launches":n, "ms":total, "flops":total, "bytes":total}, ...]}
std::string prof_end();
void prof_mark_start(const char* name, double flops, double bytes, hipStream_t s);
void prof_mark_stop(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), on(prof_enabled()) {
        if (on) prof_mark_start(name, flops, bytes, s);
    }
    ~ProfScope() {
        if (on) prof_mark_stop(s);
    }
};
}  // namespace sdmi
