// Optional per-kernel-family HIP-event timing on the launch stream (used by bench.py for the
// roofline line).  Disabled by default: zero cost unless ocrl_prof_enable() selects a family.
#include <vector>

#include "kernels.h"

namespace {
struct Rec { hipEvent_t a, b; int tag; };
unsigned g_mask = 0;
std::vector<Rec> g_recs;
size_t g_used = 0;
}  // namespace

int prof_begin(int tag, hipStream_t st) {
    if (!((g_mask >> tag) & 1u)) return -1;
    if (g_used == g_recs.size()) {
        Rec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
        g_recs.push_back(r);
    }
    Rec& r = g_recs[g_used];
    r.tag = tag;
    hipEventRecord(r.a, st);
    return (int)g_used++;
}
void prof_end(int idx, hipStream_t st) {
    if (idx >= 0) hipEventRecord(g_recs[idx].b, st);
}

extern "C" {
int ocrl_prof_enable(unsigned tag_mask) {
    g_mask = tag_mask;
    g_used = 0;
    return 0;
}
// Synchronises the device.  ms[t] = total time of family t, count[t] = launches, t < ntags.
int ocrl_prof_collect(double* ms, long long* count, int ntags) {
    if (hipDeviceSynchronize() != hipSuccess) { ocrl_set_error("ocrl_prof_collect: device error"); return 1; }
    for (int t = 0; t < ntags; ++t) { ms[t] = 0.0; count[t] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        float e = 0.f;
        if (hipEventElapsedTime(&e, g_recs[i].a, g_recs[i].b) != hipSuccess) continue;
        if (g_recs[i].tag < ntags) { ms[g_recs[i].tag] += e; count[g_recs[i].tag] += 1; }
    }
    g_used = 0;
    return 0;
}
}
