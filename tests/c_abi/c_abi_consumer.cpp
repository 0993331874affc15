// A compiled consumer of the C ABI (the position a Rust FFI shim would be in): only include/goldibear_gpu.h and
// libgoldibear_gpu.so, no Python, no torch.  Builds the reference's bench-form dummy circuit (2^k rows, Goldilocks,
// standard_recursion_config_gl with 2 challenges; plonk/circuit_builder.rs:1110-1312 for the PublicInputGate /
// ConstantGate rows, selector, k_is and sigma columns), commits, proves, verifies, and checks the batch accessors.
//   g++ -O2 -std=c++17 -I include tests/c_abi/c_abi_consumer.cpp -L plonky2_goldibear_amd/lib -lgoldibear_gpu \
//       -Wl,-rpath,$PWD/plonky2_goldibear_amd/lib -o /tmp/c_abi_consumer && /tmp/c_abi_consumer 8
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "goldibear_gpu.h"

typedef unsigned long long u64;
static const u64 P = 0xFFFFFFFF00000001ULL;
static u64 mulmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % P); }
static u64 powmod(u64 b, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mulmod(r, b); b = mulmod(b, b); e >>= 1; } return r; }

#define CHECK(call)                                                                          \
    do {                                                                                     \
        gb_status st_ = (call);                                                              \
        if (st_ != GB_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, st_, gb_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const unsigned k = argc > 1 ? (unsigned)std::atoi(argv[1]) : 8;
    const size_t n = (size_t)1 << k;
    const unsigned routed = 80, wires = 135, nconst = 2;
    gb_ctx* ctx = nullptr;
    if (gb_ctx_create(0, &ctx) != GB_OK) { std::fprintf(stderr, "gb_ctx_create failed\n"); return 1; }

    // constants || sigmas values: selector, 2 constant columns, 80 sigma columns (gate indices: Noop 0, Constant 1, PI 2)
    const size_t pi_row = (n >> 1) + 1, const_row = pi_row + 1;
    std::vector<u64> cs((1 + nconst + routed) * n, 0), k_is(routed), sub(n);
    cs[pi_row] = 2;
    cs[const_row] = 1;
    for (unsigned j = 0; j < routed; j++) k_is[j] = powmod(7, j);
    const u64 w = powmod(1753635133440165772ULL, (u64)1 << (32 - k));
    sub[0] = 1;
    for (size_t i = 1; i < n; i++) sub[i] = mulmod(sub[i - 1], w);
    u64* sig = cs.data() + (1 + nconst) * n;
    for (unsigned j = 0; j < routed; j++)
        for (size_t i = 0; i < n; i++) sig[j * n + i] = mulmod(k_is[j], sub[i]);
    const size_t cls[5][2] = {{pi_row, 0}, {pi_row, 1}, {pi_row, 2}, {pi_row, 3}, {const_row, 0}};  // one copy class
    for (int t = 0; t < 5; t++) {
        const size_t* nx = cls[(t + 1) % 5];
        sig[cls[t][1] * n + cls[t][0]] = mulmod(k_is[nx[1]], sub[nx[0]]);
    }

    gb_circuit_config cfg{};
    cfg.field = GB_GOLDILOCKS; cfg.degree_bits = k; cfg.num_wires = wires; cfg.num_routed_wires = routed; cfg.num_constants = nconst;
    cfg.num_challenges = 2; cfg.max_quotient_degree_factor = 8; cfg.rate_bits = 3; cfg.cap_height = 4; cfg.proof_of_work_bits = 16;
    cfg.num_query_rounds = 28; cfg.arity_bits = 4; cfg.final_poly_bits = 5; cfg.num_selectors = 1; cfg.gate_constant = 1; cfg.gate_pi = 2;
    gb_circuit* circuit = nullptr;
    CHECK(gb_circuit_create(ctx, &cfg, cs.data(), k_is.data(), GB_INPUT_HOST, &circuit));
    u64 cap[16 * 4], digest[4];
    CHECK(gb_circuit_verifier_data(circuit, cap, digest));

    // MatrixWitness: zeros except the random wires of the PublicInputGate row
    std::vector<u64> wit((size_t)wires * n, 0);
    u64 x = 0x9E3779B97F4A7C15ULL;
    for (unsigned c = 4; c < wires; c++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; wit[c * n + pi_row] = x % P; }
    std::vector<uint8_t> proof(4 << 20);
    size_t len = 0;
    CHECK(gb_prove(circuit, wit.data(), GB_INPUT_HOST, nullptr, 0, proof.data(), proof.size(), &len));
    CHECK(gb_verify(circuit, proof.data(), len));
    proof[100] ^= 1;  // a cap word of the wires commitment
    if (gb_verify(circuit, proof.data(), len) != GB_ERR_VERIFY) { std::fprintf(stderr, "tampered proof was not rejected\n"); return 1; }
    proof[100] ^= 1;
    {   // MatrixWitness.wire_values as the reference holds it (iop/witness.rs:277-279): one separately allocated column per wire,
        // handed over as a pointer table - gb_prove_cols; the same bytes.  Then the same columns in page-locked memory of the library.
        std::vector<std::vector<u64>> cols(wires);
        std::vector<const void*> ptrs(wires);
        for (unsigned c = 0; c < wires; c++) { cols[c].assign(wit.begin() + (size_t)c * n, wit.begin() + (size_t)(c + 1) * n); ptrs[c] = cols[c].data(); }
        std::vector<uint8_t> p2(4 << 20);
        size_t len2 = 0;
        CHECK(gb_prove_cols(circuit, ptrs.data(), GB_INPUT_HOST, nullptr, 0, p2.data(), p2.size(), &len2));
        if (len2 != len || std::memcmp(p2.data(), proof.data(), len) != 0) { std::fprintf(stderr, "gb_prove_cols differs from gb_prove\n"); return 1; }
        void* pinned = nullptr;
        CHECK(gb_host_alloc(ctx, (size_t)wires * n * sizeof(u64), &pinned));
        std::memcpy(pinned, wit.data(), (size_t)wires * n * sizeof(u64));
        for (unsigned c = 0; c < wires; c++) ptrs[c] = static_cast<u64*>(pinned) + (size_t)c * n;
        CHECK(gb_prove_cols(circuit, ptrs.data(), GB_INPUT_HOST, nullptr, 0, p2.data(), p2.size(), &len2));
        if (len2 != len || std::memcmp(p2.data(), proof.data(), len) != 0) { std::fprintf(stderr, "gb_prove_cols (page-locked) differs\n"); return 1; }
        CHECK(gb_host_free(ctx, pinned));
        CHECK(gb_ctx_set_option(ctx, "copy_threads", 2));
        if (gb_ctx_set_option(ctx, "no_such_option", 1) != GB_ERR_INVALID) { std::fprintf(stderr, "unknown option accepted\n"); return 1; }
        if (gb_prove(circuit, wit.data(), 0x200, nullptr, 0, p2.data(), p2.size(), &len2) != GB_ERR_INVALID) { std::fprintf(stderr, "unknown flag bit accepted\n"); return 1; }
    }

    // the verifier's side without the prover's circuit object: CommonCircuitData (config + gate table) and VerifierOnlyCircuitData
    // (cap, digest) only - no device involved (ctx = NULL) - and the compressed form of the proof
    {
        const gb_gate gates[3] = {{GB_GATE_NOOP, 0, 0, 0, 3, 0, 0}, {GB_GATE_CONSTANT, nconst, 0, 0, 3, 0, 0}, {GB_GATE_PUBLIC_INPUT, 4, 0, 0, 3, 0, 0}};
        gb_circuit* verifier = nullptr;
        if (gb_verifier_create(nullptr, &cfg, gates, 3, k_is.data(), cap, digest, &verifier) != GB_OK) {
            std::fprintf(stderr, "gb_verifier_create: %s\n", gb_last_error(nullptr));
            return 1;
        }
        CHECK(gb_verify(verifier, proof.data(), len));
        std::vector<uint8_t> small(len), back(len + 64);
        size_t slen = 0, blen = 0;
        CHECK(gb_proof_compress(verifier, proof.data(), len, small.data(), small.size(), &slen));
        if (slen >= len) { std::fprintf(stderr, "compressed proof is not smaller\n"); return 1; }
        CHECK(gb_verify_compressed(verifier, small.data(), slen));
        CHECK(gb_proof_decompress(verifier, small.data(), slen, back.data(), back.size(), &blen));
        if (blen != len || std::memcmp(back.data(), proof.data(), len) != 0) { std::fprintf(stderr, "decompress(compress(proof)) != proof\n"); return 1; }
        size_t need = 0;
        if (gb_proof_compress(verifier, proof.data(), len, small.data(), 16, &need) != GB_ERR_BUFFER_TOO_SMALL || need != slen) {
            std::fprintf(stderr, "short output buffer was not reported\n");
            return 1;
        }
        CHECK(gb_circuit_free(verifier));
    }

    // the PolynomialBatch boundary on its own: commit the witness, read the cap (must equal the proof's first cap), a leaf + path
    gb_batch* batch = nullptr;
    CHECK(gb_commit_values(ctx, GB_GOLDILOCKS, wit.data(), wires, k, 3, 4, nullptr, GB_INPUT_HOST, &batch));
    u64 wcap[16 * 4];
    CHECK(gb_batch_cap(batch, wcap));
    if (std::memcmp(wcap, proof.data(), sizeof wcap) != 0) { std::fprintf(stderr, "wires cap differs from the proof's\n"); return 1; }
    std::vector<u64> row(wires), sibs((k + 3) * 4);
    uint32_t nsib = 0;
    CHECK(gb_batch_leaf(batch, 5, row.data(), sibs.data(), &nsib));
    if (nsib != k + 3 - 4) { std::fprintf(stderr, "unexpected Merkle path length %u\n", nsib); return 1; }
    CHECK(gb_batch_free(batch));
    CHECK(gb_circuit_free(circuit));
    CHECK(gb_ctx_destroy(ctx));
    std::printf("c_abi_consumer ok: 2^%u rows, proof %zu bytes, digest %016llx...\n", k, len, digest[0]);
    return 0;
}
