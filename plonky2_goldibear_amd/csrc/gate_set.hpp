// The gate set of a circuit as the kernels and the host code pass it around (CommonCircuitData.gates + selectors_info).
#pragma once
#include "../../include/goldibear_gpu.h"

namespace gbk {
namespace gates {

constexpr unsigned MAX_GATES = 24;  // DefaultGateSerializer knows 20 gate types (util/serialization/gate_serialization.rs:143-165)
constexpr unsigned UNUSED_SELECTOR = 0xFFFFFFFFu;  // gates/selectors.rs:13
constexpr unsigned POSEIDON_NUM_CONSTRAINTS = 12 * 7 + 22 + 12 + 1 + 4;
constexpr unsigned POSEIDON2_BB_CONSTRAINTS_PER_OP = 1 + 8 + 16 * 7 + 13 + 16;  // 150
constexpr unsigned POSEIDON2_BB_WIRES_PER_OP = 33 + 8 + 16 * 7 + 13;           // 166

constexpr unsigned MAX_INTERPOLATION_BITS = 4;   // CosetInterpolationGate subgroup_bits (= a FRI arity_bits, <= 4 here)
constexpr unsigned MAX_RANDOM_ACCESS_BITS = 6;   // (2 + 2^bits) routed wires per copy

struct GateSet {
    unsigned num_gates, num_selectors;
    gb_gate g[MAX_GATES];
    // two_adic_subgroup(4) and 1 / 2^b (b = 0..4) in the field's device form (u32 values for BabyBear), for
    // CosetInterpolationGate: its domain is every (16 >> b)-th entry and its barycentric weights are x_i / 2^b
    unsigned long long subgroup16[16], inv_pow2[MAX_INTERPOLATION_BITS + 1];
};

}  // namespace gates
}  // namespace gbk
