// Error plumbing + trivial entry points of libarvae_hip.so.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "diag.h"

namespace arvae {

thread_local char g_last_error[512] = "";
thread_local hipStream_t g_cur_stream = nullptr;

// ---- opt-in kernel timeline (bench.py): one HIP event on the launch stream after every kernel; a
// kernel's duration is the time since the previous event on that stream.  Off by default: zero cost.
struct ProfEvent {
    const char *name;
    hipEvent_t ev;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfEvent> g_prof;

static void prof_mark(const char *name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, g_cur_stream);
    g_prof.push_back(ProfEvent{name, ev});
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

bool profiling_active() { return g_prof_on; }

// closes the interval since the previous mark under the name "(gap)", so that the next kernel's interval starts at
// its own launch and the per-kernel times of arvae_profile_end() agree with a kernel trace.  An entry point that launches
// several kernels before its check_launch() (a weight gradient and its slice sum) marks the gap ONCE, in front of the first:
// all of them are timed under the label check_launch() gives (round 4: the second launch's gap mark used to take the first
// kernel's time with it -- Morpho-MNIST's weight gradient showed as its 8 us reduction).
static thread_local bool g_prof_pending = false;
void prof_gap() {
    if (g_prof_on && !g_prof_pending) {
        prof_mark("(gap)");
        g_prof_pending = true;
    }
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    g_prof_pending = false;
    if (e != hipSuccess) return fail(ARVAE_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    if (g_prof_on) prof_mark(what);
    return ARVAE_OK;
}

int device_cu_count() {
    static const int n = [] {                   // (thread-safe initialisation; the forward and the autograd thread both come here)
        int dev = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

}  // namespace arvae

extern "C" int arvae_abi_version(void) { return ARVAE_ABI_VERSION; }

extern "C" const char *arvae_last_error_string(void) { return arvae::g_last_error; }

extern "C" int arvae_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int arvae_profile_begin(arvae_stream_t stream) {
    {
        std::lock_guard<std::mutex> lk(arvae::g_prof_mu);
        for (auto &p : arvae::g_prof) (void)hipEventDestroy(p.ev);
        arvae::g_prof.clear();
        arvae::g_prof_on = true;
    }
    (void)arvae::as_stream(stream);
    arvae::prof_mark("(begin)");
    return ARVAE_OK;
}

// Stops recording, waits for the last event and writes one line per kernel name:
//   "<name>\t<launches>\t<total milliseconds>\n".  Returns the number of bytes needed (including the NUL).
extern "C" int64_t arvae_profile_end(char *out, int64_t cap) {
    std::vector<arvae::ProfEvent> ev;
    {
        std::lock_guard<std::mutex> lk(arvae::g_prof_mu);
        arvae::g_prof_on = false;
        ev.swap(arvae::g_prof);
    }
    std::map<std::string, std::pair<long, double>> agg;
    std::vector<std::string> order;
    if (!ev.empty()) (void)hipEventSynchronize(ev.back().ev);
    for (size_t i = 1; i < ev.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i - 1].ev, ev[i].ev) != hipSuccess) ms = 0.f;
        auto it = agg.find(ev[i].name);
        if (it == agg.end()) {
            order.push_back(ev[i].name);
            agg[ev[i].name] = {1, ms};
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
    }
    for (auto &p : ev) (void)hipEventDestroy(p.ev);
    std::string text;
    char line[256];
    for (auto &name : order) {
        snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", name.c_str(), agg[name].first, agg[name].second);
        text += line;
    }
    if (out != nullptr && cap > 0) {
        const size_t n = text.size() < (size_t)cap - 1 ? text.size() : (size_t)cap - 1;
        memcpy(out, text.data(), n);
        out[n] = 0;
    }
    return (int64_t)text.size() + 1;
}
