// Mutation harness for the host-side parsers of libgoldibear_gpu (csrc/verifier_host.inc, csrc/compress_host.inc), linked
// against a HOST-ONLY build of the library compiled with -fsanitize=address,undefined (tests/sanitize/build.py).  CPU only:
// gb_verifier_create touches no device.  Includes nothing but the public header.
//
//   fuzz_host_parsers <case-file> <iterations> <seed>
//   fuzz_host_parsers <case-file> alloc-fail <max-points>     (round 6: the "never unwinds" guards under a failing allocator - the
//       global operator new of this binary throws std::bad_alloc on the N-th call while armed; every entry point is run with
//       N = 1, 2, ... until a call gets through without reaching N: each interrupted call must come back as GB_ERR_OOM - not
//       abort, not unwind into this file, not trip a sanitizer - and the object must still work afterwards)
// case file (written by tests/test_sanitized_parsers.py from the reference's regression fixture or an oracle proof):
//   gb_circuit_config | u32 num_gates | gb_gate[num_gates] | k_is[num_routed_wires] | cap[2^cap_height][H] | digest[H] |
//   u64 proof_len | proof bytes            (elements: u64 Goldilocks / u32 BabyBear)
// Every mutated input must come back as GB_OK, GB_ERR_INVALID or GB_ERR_VERIFY - anything else, or a sanitizer report,
// fails the run.  Prints one summary line.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <new>

#include "goldibear_gpu.h"

// ---- the failing allocator (alloc-fail mode; the sanitizers' own operator new is replaced for the whole binary, malloc stays theirs)
static long g_alloc_count = 0, g_alloc_fail_at = -1;
static void* counted_alloc(size_t n, bool may_throw) {
    if (g_alloc_fail_at >= 0 && ++g_alloc_count == g_alloc_fail_at) {
        if (may_throw) throw std::bad_alloc();
        return nullptr;
    }
    void* p = std::malloc(n ? n : 1);
    if (!p && may_throw) throw std::bad_alloc();
    return p;
}
void* operator new(size_t n) { return counted_alloc(n, true); }
void* operator new[](size_t n) { return counted_alloc(n, true); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { return counted_alloc(n, false); }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { return counted_alloc(n, false); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, size_t) noexcept { std::free(p); }
void operator delete[](void* p, size_t) noexcept { std::free(p); }

// run `call` with the N-th allocation failing, N = 1, 2, ... (at most max_points values of N, spread over the call's allocations);
// returns the number of interrupted calls, all of which answered GB_ERR_OOM
template <class Fn>
static long alloc_fail_sweep(const char* what, long max_points, gb_status expect, Fn&& call) {
    g_alloc_fail_at = 0; g_alloc_count = 0;          // count only
    gb_status s = call();
    const long total = g_alloc_count;
    g_alloc_fail_at = -1;
    if (s != expect) { std::fprintf(stderr, "%s: %d instead of %d before any failure was injected\n", what, s, expect); std::exit(1); }
    const long stride = total > max_points ? (total + max_points - 1) / max_points : 1;
    long hit = 0;
    for (long n = 1; n <= total; n += (n < 48 ? 1 : stride)) {
        g_alloc_count = 0; g_alloc_fail_at = n;
        s = call();
        const bool reached = g_alloc_count >= n;
        g_alloc_fail_at = -1;
        if (reached) {
            hit++;
            if (s != GB_ERR_OOM) {
                std::fprintf(stderr, "%s: allocation %ld of %ld failed and the call answered %d (%s), not GB_ERR_OOM\n", what, n, total, s, gb_last_error(nullptr));
                std::exit(1);
            }
        } else if (s != expect) {
            std::fprintf(stderr, "%s: %d instead of %d with no failure reached\n", what, s, expect);
            std::exit(1);
        }
    }
    s = call();                                      // and the object still works
    if (s != expect) { std::fprintf(stderr, "%s: %d after the sweep\n", what, s); std::exit(1); }
    std::printf("alloc-fail %s: %ld allocations per call, %ld interrupted calls -> GB_ERR_OOM\n", what, total, hit);
    return hit;
}

static uint64_t rng_state;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static std::vector<uint8_t> mutate(const std::vector<uint8_t>& src) {
    std::vector<uint8_t> m = src;
    switch (rnd() % 8) {
        case 0:  // truncate
            m.resize(rnd() % (m.size() + 1));
            break;
        case 1: {  // flip one bit
            if (m.empty()) break;
            m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
            break;
        }
        case 2: {  // overwrite a byte (path-length bytes, counts)
            if (m.empty()) break;
            m[rnd() % m.size()] = (uint8_t)rnd();
            break;
        }
        case 3: {  // overwrite an aligned 8-byte word with an extreme value
            if (m.size() < 8) break;
            const uint64_t vals[4] = {0, ~0ULL, 0xFFFFFFFF00000001ULL, 1ULL << 63};
            const size_t off = (rnd() % (m.size() / 8)) * 8;
            std::memcpy(&m[off], &vals[rnd() % 4], 8);
            break;
        }
        case 4: {  // append junk
            const size_t n = 1 + rnd() % 64;
            for (size_t i = 0; i < n; i++) m.push_back((uint8_t)rnd());
            break;
        }
        case 5: {  // cut a run out of the middle
            if (m.size() < 2) break;
            const size_t a = rnd() % m.size(), n = 1 + rnd() % std::min<size_t>(64, m.size() - a);
            m.erase(m.begin() + a, m.begin() + a + n);
            break;
        }
        case 6: {  // a burst of flips
            for (int i = 0; i < 16 && !m.empty(); i++) m[rnd() % m.size()] ^= (uint8_t)rnd();
            break;
        }
        default: {  // duplicate a run (shifts everything behind it)
            if (m.empty()) break;
            const size_t a = rnd() % m.size(), n = 1 + rnd() % std::min<size_t>(32, m.size() - a);
            std::vector<uint8_t> run(m.begin() + a, m.begin() + a + n);
            m.insert(m.begin() + a, run.begin(), run.end());
            break;
        }
    }
    return m;
}

static bool acceptable(gb_status s) { return s == GB_OK || s == GB_ERR_INVALID || s == GB_ERR_VERIFY || s == GB_ERR_BUFFER_TOO_SMALL; }

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s case-file iterations seed\n", argv[0]);
        return 2;
    }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    std::fclose(f);
    const bool alloc_fail = std::strcmp(argv[2], "alloc-fail") == 0;
    const long iterations = alloc_fail ? 0 : std::atol(argv[2]);
    rng_state = std::strtoull(argv[3], nullptr, 0);

    size_t off = 0;
    auto take = [&](void* dst, size_t n) {
        if (off + n > file.size()) { std::fprintf(stderr, "short case file\n"); std::exit(2); }
        std::memcpy(dst, &file[off], n);
        off += n;
    };
    gb_circuit_config cfg;
    take(&cfg, sizeof cfg);
    uint32_t num_gates;
    take(&num_gates, 4);
    std::vector<gb_gate> gates(num_gates);
    take(gates.data(), num_gates * sizeof(gb_gate));
    const size_t es = cfg.field == GB_GOLDILOCKS ? 8 : 4, H = cfg.field == GB_GOLDILOCKS ? 4 : 8;
    std::vector<uint8_t> k_is(cfg.num_routed_wires * es), cap((H << cfg.cap_height) * es), digest(H * es);
    take(k_is.data(), k_is.size());
    take(cap.data(), cap.size());
    take(digest.data(), digest.size());
    uint64_t proof_len;
    take(&proof_len, 8);
    std::vector<uint8_t> proof(proof_len);
    take(proof.data(), proof_len);

    gb_circuit* c = nullptr;
    gb_status s = gb_verifier_create(nullptr, &cfg, gates.data(), num_gates, k_is.data(), cap.data(), digest.data(), &c);
    if (s != GB_OK) { std::fprintf(stderr, "gb_verifier_create: %d %s\n", s, gb_last_error(nullptr)); return 1; }
    s = gb_verify(c, proof.data(), proof.size());
    if (s != GB_OK) { std::fprintf(stderr, "the unmutated proof does not verify: %d %s\n", s, gb_last_error(nullptr)); return 1; }
    std::vector<uint8_t> out(2 * proof.size() + (1 << 16));
    size_t n = 0;
    s = gb_proof_compress(c, proof.data(), proof.size(), out.data(), out.size(), &n);
    if (s != GB_OK) { std::fprintf(stderr, "compress: %d %s\n", s, gb_last_error(nullptr)); return 1; }
    std::vector<uint8_t> compressed(out.begin(), out.begin() + n);
    s = gb_proof_decompress(c, compressed.data(), compressed.size(), out.data(), out.size(), &n);
    if (s != GB_OK || n != proof.size() || std::memcmp(out.data(), proof.data(), n)) { std::fprintf(stderr, "round trip failed\n"); return 1; }
    if (gb_verify_compressed(c, compressed.data(), compressed.size()) != GB_OK) { std::fprintf(stderr, "verify_compressed failed\n"); return 1; }

    if (alloc_fail) {
        const long pts = std::atol(argv[3]);
        long hit = 0;
        hit += alloc_fail_sweep("gb_verifier_create", pts, GB_OK, [&] {
            gb_circuit* c2 = nullptr;
            const gb_status st = gb_verifier_create(nullptr, &cfg, gates.data(), num_gates, k_is.data(), cap.data(), digest.data(), &c2);
            if (st == GB_OK) gb_circuit_free(c2);
            else if (c2) { std::fprintf(stderr, "gb_verifier_create failed and left an object behind\n"); std::exit(1); }
            return st;
        });
        hit += alloc_fail_sweep("gb_verify", pts, GB_OK, [&] { return gb_verify(c, proof.data(), proof.size()); });
        hit += alloc_fail_sweep("gb_proof_compress", pts, GB_OK, [&] { return gb_proof_compress(c, proof.data(), proof.size(), out.data(), out.size(), &n); });
        hit += alloc_fail_sweep("gb_proof_decompress", pts, GB_OK, [&] { return gb_proof_decompress(c, compressed.data(), compressed.size(), out.data(), out.size(), &n); });
        hit += alloc_fail_sweep("gb_verify_compressed", pts, GB_OK, [&] { return gb_verify_compressed(c, compressed.data(), compressed.size()); });
        std::vector<uint8_t> bad(proof);
        bad[bad.size() / 2] ^= 1;     // a rejected proof: the error path allocates its message
        const gb_status rej = gb_verify(c, bad.data(), bad.size());
        hit += alloc_fail_sweep("gb_verify (rejected proof)", pts, rej, [&] { return gb_verify(c, bad.data(), bad.size()); });
        gb_circuit_free(c);
        std::printf("alloc-fail ok: %ld interrupted calls\n", hit);
        return 0;
    }
    long counts[3][4] = {};  // [entry point][ok, invalid, verify, too small]
    auto tally = [&](int e, gb_status st, const char* what) {
        if (!acceptable(st)) {
            std::fprintf(stderr, "%s returned %d (%s) on a mutated input\n", what, st, gb_last_error(nullptr));
            std::exit(1);
        }
        counts[e][st == GB_OK ? 0 : st == GB_ERR_INVALID ? 1 : st == GB_ERR_VERIFY ? 2 : 3]++;
    };
    for (long it = 0; it < iterations; it++) {
        const std::vector<uint8_t> m = mutate(proof);
        const gb_status vs = gb_verify(c, m.data(), m.size());
        tally(0, vs, "gb_verify");
        if (vs == GB_OK && m != proof) {   // a mutation may be the identity (a byte overwritten with itself); nothing else may verify
            std::fprintf(stderr, "gb_verify ACCEPTED a mutated proof (iteration %ld)\n", it);
            return 1;
        }
        if (it % 4 == 0) tally(1, gb_proof_compress(c, m.data(), m.size(), out.data(), out.size(), &n), "gb_proof_compress");
        const std::vector<uint8_t> mc = mutate(compressed);
        const gb_status cs = gb_verify_compressed(c, mc.data(), mc.size());
        tally(2, cs, "gb_verify_compressed");
        if (cs == GB_OK && mc != compressed) {
            std::fprintf(stderr, "gb_verify_compressed ACCEPTED a mutated proof (iteration %ld)\n", it);
            return 1;
        }
        if (it % 4 == 1) tally(2, gb_proof_decompress(c, mc.data(), mc.size(), out.data(), it % 8 == 1 ? 16 : out.size(), &n), "gb_proof_decompress");
    }
    // empty and tiny inputs
    for (size_t len = 0; len < 24; len++) {
        tally(0, gb_verify(c, proof.data(), len), "gb_verify");
        tally(2, gb_verify_compressed(c, compressed.data(), len), "gb_verify_compressed");
    }
    gb_circuit_free(c);
    std::printf("fuzz ok: %ld iterations; verify ok/invalid/verify = %ld/%ld/%ld; compress = %ld/%ld/%ld; compressed = %ld/%ld/%ld/%ld\n",
                iterations, counts[0][0], counts[0][1], counts[0][2], counts[1][0], counts[1][1], counts[1][2], counts[2][0],
                counts[2][1], counts[2][2], counts[2][3]);
    return 0;
}
