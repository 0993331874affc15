// The gate set of a circuit as the kernels and the host code pass it around (CommonCircuitData.gates + selectors_info).
#pragma once
#include "../../include/goldibear_gpu.h"

namespace gbk {
namespace gates {

constexpr unsigned MAX_GATES = 16;
constexpr unsigned UNUSED_SELECTOR = 0xFFFFFFFFu;  // gates/selectors.rs:13
constexpr unsigned POSEIDON_NUM_CONSTRAINTS = 12 * 7 + 22 + 12 + 1 + 4;
constexpr unsigned POSEIDON2_BB_CONSTRAINTS_PER_OP = 1 + 8 + 16 * 7 + 13 + 16;  // 150
constexpr unsigned POSEIDON2_BB_WIRES_PER_OP = 33 + 8 + 16 * 7 + 13;           // 166

struct GateSet {
    unsigned num_gates, num_selectors;
    gb_gate g[MAX_GATES];
};

}  // namespace gates
}  // namespace gbk
