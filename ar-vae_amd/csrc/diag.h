// Diagnostic switches of the library.  The PRODUCT build reads no environment variable: diag_env() is the constant "not set",
// every switch below is dead code the compiler removes, and there is one code path per kernel family -- the one the parity
// suite covers.  A DIAGNOSTIC build (-DARVAE_DIAG: ar-vae_amd/libarvae_hip_diag.so, built next to the product library and
// selected with ARVAE_LIB; tools/build_diag.sh for one-file variants) reads them, once per switch, for same-box A/B runs and
// for the two tests that hold the default paths to their alternatives (tests/test_hip_parity.py:
// test_paired_launches_match_the_separate_launches, test_latent_block_experiment_matches_the_per_layer_path).
// DESIGN.md section 5 lists the switches.
#pragma once
#include <stdlib.h>

namespace arvae {

#ifdef ARVAE_DIAG
inline const char *diag_env(const char *name) { return getenv(name); }
constexpr bool kDiagBuild = true;
#else
constexpr const char *diag_env(const char *) { return nullptr; }
constexpr bool kDiagBuild = false;
#endif
inline int diag_int(const char *name, int otherwise = 0) {
    const char *v = diag_env(name);
    return v != nullptr ? atoi(v) : otherwise;
}

// compute units of the current device (queried once per process; 256 when the query fails)
int device_cu_count();

}  // namespace arvae
