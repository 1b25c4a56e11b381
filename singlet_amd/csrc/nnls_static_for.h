// compile-time loop shared by the NNLS kernels
#pragma once
#include <utility>
#include <type_traits>

// compile-time loop: guarantees that b[] / x[] are only ever indexed by constants
// (so they live in VGPRs) regardless of the optimiser's unroll thresholds.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

