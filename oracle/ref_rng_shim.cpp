// C shim around the REFERENCE's own `rng` class (src/singlet.cpp:6-114), compiled from the reference tree where it
// lies: oracle/make_ref.sh cuts the class out of /root/reference/src/singlet.cpp at build time into the git-ignored
// oracle/_ref/rng_class.inc (never committed) and compiles this file against it -> oracle/_ref/librng_ref.so.
// Test infrastructure only: it pins the integer part of the oracle (hash, mask draws) to the reference's code.
// The class is dependency-free C++ (<cstdint>, <cmath> for std::floor); nothing is stubbed.
#include <cstdint>
#include <cmath>
#include <cstddef>

#include "_ref/rng_class.inc"

extern "C" {
// rng(state).rand(i, j), src/singlet.cpp:46-63
__attribute__((visibility("default"))) void ref_rand2(uint64_t state, const uint64_t* i, const uint64_t* j, int64_t n, uint64_t* out) {
    rng s(state);
    for (int64_t q = 0; q < n; ++q) out[q] = s.rand(i[q], j[q]);
}
// rng(state).rand(i), src/singlet.cpp:30-44
__attribute__((visibility("default"))) void ref_rand1(uint64_t state, const uint64_t* i, int64_t n, uint64_t* out) {
    rng s(state);
    for (int64_t q = 0; q < n; ++q) out[q] = s.rand(i[q]);
}
// rng(state).draw(i, j, inv_density) over a grid i = i0 .. i0+ni-1 (slow), j = j0 .. j0+nj-1 (fast): src/singlet.cpp:91-95,
// called as seed.draw(cell, gene, inv_density) at :450, :485, :553, :590
__attribute__((visibility("default"))) void ref_draw_grid(uint64_t state, uint64_t inv_density, uint64_t i0, int64_t ni, uint64_t j0, int64_t nj,
                                                          uint8_t* out) {
    rng s(state);
    for (int64_t a = 0; a < ni; ++a)
        for (int64_t b = 0; b < nj; ++b) out[a * nj + b] = s.draw(i0 + (uint64_t)a, j0 + (uint64_t)b, inv_density) ? 1 : 0;
}
}
