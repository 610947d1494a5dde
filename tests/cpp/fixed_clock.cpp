// LD_PRELOAD shim for tests: std::chrono::system_clock::now() (= high_resolution_clock in libstdc++) returns FRIES_FIXED_CLOCK_NS
// nanoseconds since the epoch while that variable is set.  The reference's drivers seed their mt19937 with
// (unsigned int) high_resolution_clock::now().time_since_epoch().count() (FRIES_bin/frisys_mol.cpp:104-106); this is how a test gives
// the unmodified driver source the seed of a golden trajectory.
#include <chrono>
#include <cstdlib>
#include <time.h>
namespace std { namespace chrono { inline namespace _V2 {
system_clock::time_point system_clock::now() noexcept {
    const char *s = getenv("FRIES_FIXED_CLOCK_NS");
    if (s) return time_point(duration(strtoll(s, nullptr, 10)));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return time_point(duration((long long)ts.tv_sec * 1000000000ll + ts.tv_nsec));
}
}}}
