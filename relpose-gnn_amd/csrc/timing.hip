// Per-launch HIP-event timing of the hot kernels, recorded on the launch stream itself (bench.py's roofline
// numbers come from here), plus the thread-local last-error string.
#include <mutex>
#include <string>
#include <vector>

#include "rpg_common.h"

namespace {

struct Slot {
    hipEvent_t beg, end;
    int klass;
    double work, executed;
};
constexpr int MAX_SLOTS = 16384;
std::mutex g_mu;
bool g_enabled = false;
std::vector<Slot> g_slots;     // event pairs, created lazily and reused
int g_used = 0;
thread_local std::string t_last_error;

}  // namespace

namespace rpg {

void set_last_error(const char* where, hipError_t e) {
    t_last_error = std::string(where) + ": " + hipGetErrorString(e);
}

int timing_begin(int klass, hipStream_t s) {
    if (!g_enabled) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used >= MAX_SLOTS) return -1;
    if (g_used == (int)g_slots.size()) {
        Slot n{};
        if (hipEventCreate(&n.beg) != hipSuccess || hipEventCreate(&n.end) != hipSuccess) return -1;
        g_slots.push_back(n);
    }
    Slot& sl = g_slots[g_used];
    sl.klass = klass;
    sl.work = 0.0;
    sl.executed = 0.0;
    (void)hipEventRecord(sl.beg, s);
    return g_used++;
}

void timing_end(int slot, double work, hipStream_t s, double executed) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_slots[slot].work = work;
    g_slots[slot].executed = executed > 0.0 ? executed : work;
    (void)hipEventRecord(g_slots[slot].end, s);
}

}  // namespace rpg

extern "C" int rpg_abi_version(void) { return RPG_ABI_VERSION; }

extern "C" const char* rpg_last_error(void) { return t_last_error.c_str(); }

extern "C" int rpg_timing_enable(int enable) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_enabled = enable != 0;
    return RPG_OK;
}

extern "C" int rpg_timing_read(double* ms, long long* launches, double* work) {
    return rpg_timing_read_ex(ms, launches, work, nullptr);
}

extern "C" int rpg_timing_read_ex(double* ms, long long* launches, double* work, double* executed) {
    if (!ms || !launches || !work) return RPG_ERR_BAD_ARG;
    if (hipDeviceSynchronize() != hipSuccess) {
        rpg::set_last_error("timing_read", hipGetLastError());
        return RPG_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (int k = 0; k < RPG_TIMER_COUNT; ++k) {
        ms[k] = 0.0; launches[k] = 0; work[k] = 0.0;
        if (executed) executed[k] = 0.0;
    }
    for (int i = 0; i < g_used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_slots[i].beg, g_slots[i].end) != hipSuccess) continue;
        const int k = g_slots[i].klass;
        if (k < 0 || k >= RPG_TIMER_COUNT) continue;
        ms[k] += t;
        launches[k] += 1;
        work[k] += g_slots[i].work;
        if (executed) executed[k] += g_slots[i].executed;
    }
    g_used = 0;
    return RPG_OK;
}
