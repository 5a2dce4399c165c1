// Error reporting and ABI bookkeeping shared by every entry point.
#include <cxxabi.h>
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "common.h"
#include <stdlib.h>

namespace cnuda {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

bool raise_dynamic_lds(const void* host_function, size_t lds, size_t limit) {
    if (lds <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("raise_dynamic_lds: hipGetDevice failed"); return false; }
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({host_function, dev})) return true;
    const hipError_t e = hipFuncSetAttribute(host_function, hipFuncAttributeMaxDynamicSharedMemorySize, (int)limit);
    if (e != hipSuccess) {
        set_error("raise_dynamic_lds: hipFuncSetAttribute(%zu bytes) on device %d: %s", limit, dev, hipGetErrorString(e));
        return false;
    }
    done.insert({host_function, dev});
    return true;
}
}  // namespace cnuda

namespace cnuda {
namespace {
struct ProfRecord { hipEvent_t start, stop; int tag; char name[kProfNameLen]; };
std::vector<ProfRecord> g_prof_pool;   // pre-created event pairs
size_t g_prof_used = 0;
int g_prof_armed = -1;                 // tag for the next main-kernel launch, -1 = off
int g_prof_group = -1;                 // tag shared by every scope of a multi-kernel entry point
}  // namespace

ProfScope::ProfScope(hipStream_t st, int sub) : st_(st), rec_(-1) {
    const int tag = g_prof_armed >= 0 ? g_prof_armed : g_prof_group;
    if (tag < 0 || g_prof_used >= g_prof_pool.size()) return;
    rec_ = (int)g_prof_used++;
    g_prof_pool[rec_].tag = tag | (sub << 24);
    g_prof_pool[rec_].name[0] = 0;
    g_prof_armed = -1;
    (void)hipEventRecord(g_prof_pool[rec_].start, st_);
}
ProfScope::~ProfScope() {
    if (rec_ >= 0) (void)hipEventRecord(g_prof_pool[rec_].stop, st_);
}
void ProfScope::name(const char* fmt, ...) {
    if (rec_ < 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_prof_pool[rec_].name, kProfNameLen, fmt, ap);
    va_end(ap);
}
ProfGroup::ProfGroup() {
    g_prof_group = g_prof_armed;
    g_prof_armed = -1;
}
ProfGroup::~ProfGroup() { g_prof_group = -1; }
bool prof_active() { return !g_prof_pool.empty(); }
}  // namespace cnuda

extern "C" int cnuda_prof_enable(int max_records) {
    using namespace cnuda;
    for (auto& r : g_prof_pool) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    g_prof_pool.clear();
    g_prof_used = 0;
    g_prof_armed = -1;
    g_prof_group = -1;
    for (int i = 0; i < max_records; ++i) {
        ProfRecord r;
        r.tag = -1;
        r.name[0] = 0;
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
            set_error("cnuda_prof_enable: hipEventCreate failed");
            return (int)hipGetLastError();
        }
        g_prof_pool.push_back(r);
    }
    return 0;
}
extern "C" int cnuda_prof_arm(int tag) {
    cnuda::g_prof_armed = tag;
    return 0;
}
extern "C" int cnuda_prof_collect(int* tags, float* ms, char* names, int cap) {
    using namespace cnuda;
    int n = 0;
    for (size_t i = 0; i < g_prof_used && n < cap; ++i) {
        if (hipEventSynchronize(g_prof_pool[i].stop) != hipSuccess) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof_pool[i].start, g_prof_pool[i].stop) != hipSuccess) break;
        tags[n] = g_prof_pool[i].tag;
        ms[n] = t;
        if (names) memcpy(names + (size_t)n * kProfNameLen, g_prof_pool[i].name, kProfNameLen);
        ++n;
    }
    g_prof_used = 0;
    return n;
}
extern "C" int cnuda_prof_name_len(void) { return cnuda::kProfNameLen; }

namespace cnuda {
namespace {
int mode_from_env() {
    const char* e = getenv("CNUDA_MATRIX_MODE");
    return (e && e[0] == '1') ? 1 : 0;
}
int g_matrix_mode = mode_from_env();
}  // namespace
int matrix_mode() { return g_matrix_mode; }
bool wave_specialised() {
    static const bool on = !(getenv("CNUDA_WS") && getenv("CNUDA_WS")[0] == '0');
    return on;
}
}  // namespace cnuda
extern "C" int cnuda_set_matrix_mode(int mode) {
    if (mode != 0 && mode != 1) {
        cnuda::set_error("cnuda_set_matrix_mode: mode must be 0 (f32 MFMA) or 1 (split bf16 MFMA), got %d", mode);
        return CNUDA_ERR_INVALID_ARGUMENT;
    }
    cnuda::g_matrix_mode = mode;
    return 0;
}
extern "C" int cnuda_get_matrix_mode(void) { return cnuda::g_matrix_mode; }

namespace cnuda {
bool g_launch_log_on = false;
namespace {
struct LaunchCount { const void* fn; unsigned long long n; };
std::vector<LaunchCount> g_launch_log;      // one entry per distinct host function, in first-launch order
}
void launch_log(const void* fn) {
    for (auto& e : g_launch_log)
        if (e.fn == fn) { ++e.n; return; }
    g_launch_log.push_back({fn, 1});
}
}  // namespace cnuda
extern "C" int cnuda_launch_log_enable(int on) {
    if (on && !cnuda::g_launch_log_on) cnuda::g_launch_log.clear();
    cnuda::g_launch_log_on = on != 0;
    return 0;
}
extern "C" int cnuda_launch_log_collect(char* names, size_t cap) {
    using namespace cnuda;
    size_t used = 0;
    int n = 0;
    for (const auto& e : g_launch_log) {
        const char* mangled = hipKernelNameRefByPtr(e.fn, nullptr);
        if (!mangled) mangled = "?";
        int status = 1;
        char* plain = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
        const char* nm = (status == 0 && plain) ? plain : mangled;
        char count[32];
        const int clen = snprintf(count, sizeof(count), "\t%llu\n", e.n);
        const size_t len = strlen(nm);
        const bool fits = used + len + (size_t)clen + 1 <= cap;
        if (fits) {
            memcpy(names + used, nm, len);
            memcpy(names + used + len, count, (size_t)clen);
            used += len + (size_t)clen;
            ++n;
        }
        free(plain);
        if (!fits) {
            set_error("cnuda_launch_log_collect: %zu bytes do not hold the names of %zu kernels", cap, g_launch_log.size());
            return CNUDA_ERR_INVALID_ARGUMENT;
        }
    }
    if (cap) names[used] = 0;
    return n;
}

extern "C" int cnuda_abi_version(void) { return CNUDA_ABI_VERSION; }
extern "C" const char* cnuda_last_error(void) { return cnuda::g_error; }
