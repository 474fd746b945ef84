// trace.hpp -- roctx ranges around the library's launches (tracing): `rocprofv3 --marker-trace` shows every
// runStft / runDecayColour / ingest as a named range on the host timeline.  The marker library is bound with dlopen at first use
// (rocprofv3's librocprofiler-sdk-roctx, else roctracer's libroctx64) -- libsgz.so does not link it, and without it a range is two
// calls through null checks.
#pragma once
#include <dlfcn.h>

namespace sgz {

struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};

inline const RoctxApi &roctx()
{
    static const RoctxApi api;                 // (initialised once, thread-safe)
    return api;
}

// a named range for the lifetime of the object
struct TraceRange {
    explicit TraceRange(const char *name) { if (roctx().push) (void)roctx().push(name); }
    ~TraceRange() { if (roctx().pop) (void)roctx().pop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

}  // namespace sgz
