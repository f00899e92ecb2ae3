// dev_env.hpp — the measurement switches of DESIGN.md section 7 (STN_FFN, STN_XATTN, STN_GEMM_CFG, ...) are honoured only when the master
// switch STN_DEV_SWITCHES=1 is set beside them: one stray STN_* variable in a deployment's environment cannot move the engine off its default
// configuration.  A switch found without the master is named once on stderr and ignored.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace stn {

inline const char* dev_env(const char* name) {
    const char* v = getenv(name);
    if (!v) return nullptr;
    const char* m = getenv("STN_DEV_SWITCHES");
    if (m && !strcmp(m, "1")) return v;
    static std::mutex mu;
    static std::set<std::string> told;
    std::lock_guard<std::mutex> lk(mu);
    if (told.insert(name).second) fprintf(stderr, "libstn: %s=%s ignored (measurement switches need STN_DEV_SWITCHES=1)\n", name, v);
    return nullptr;
}

}  // namespace stn
