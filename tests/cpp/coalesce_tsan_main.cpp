// ThreadSanitizer harness of the class API's combiner (host/coalesce.cpp) for the CPU: the file is compiled by itself with
// -fsanitize=thread (tests/test_coalesce_protocol.py), the few symbols it needs from the rest of the library are defined here, and
// mcg::co::selftest drives its protocol with the stand-in device (no GPU, no HIP call is ever made: libamdhip64 is only linked).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "montecarlooptionspricer_amd/csrc/mcg_internal.hpp"
#include "montecarlooptionspricer_amd/csrc/coalesce.hpp"
#include "montecarlooptionspricer_amd/host/coalesce_host.hpp"
namespace mcg { Stats g_stats; thread_local char g_err[256];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int fail(int status, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return status; }
namespace co { int execute_round(mcg_ctx*, RoundBuffers&, double*, Request**, int) { return 1; } } }
extern "C" const char* mcg_last_error() { return mcg::g_err; }
extern "C" int mcg_init(mcg_ctx**, int) { return 1; }
int main(int argc, char** argv) {
    int w = mcg::co::selftest(argc > 1 ? atoi(argv[1]) : 16, argc > 2 ? atoi(argv[2]) : 300);
    printf("wrong %d rounds %lld calls %lld\n", w, (long long)mcg::g_stats.coalesced_rounds.load(), (long long)mcg::g_stats.coalesced_calls.load());
    return w != 0;
}
