// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// C-callable harness around the reference's DRIVER translation unit, /root/reference/src/core/PredictionGen.cpp, compiled
// in place (nothing copied) into oracle/_ref/libmcref_driver.so by oracle/Makefile.  The one function wanted from it,
// compute20DayVolAndMomentum (:313-347), is `static` in a file that also defines main(): the file is therefore read
// through #include with `main` spelled differently, and everything the driver's main would pull in (the Eigen-based
// pricers this image cannot build) is dropped again by the linker -- the TU is compiled with hidden visibility and
// -ffunction-sections, the library linked with --gc-sections, so only what ref_row_features reaches survives and no
// pricer symbol stays undefined.  No stand-in header, library or stub is involved.
// Purpose: golden vectors for mcg_row_features (oracle/gen_golden.py -> tests/golden/features.npz).
#include <cstddef>
#include <utility>
#include <vector>

#define main mcg_reference_driver_main
#include "core/../../src/core/PredictionGen.cpp"
#undef main

extern "C" __attribute__((visibility("default"))) void ref_row_features(const double* hist, size_t n, double* out2) {
    const std::vector<double> h(hist, hist + n);
    const std::pair<double, double> p = compute20DayVolAndMomentum(h);  // PredictionGen.cpp:313-347
    out2[0] = p.first;
    out2[1] = p.second;
}
