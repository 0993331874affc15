// C-ABI implementation (include/goldibear_gpu.h): contexts, device-resident PolynomialBatch
// handles, twiddle/coset table caches and the reference's timing scopes.
#include "../../include/goldibear_gpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "bb_field.hpp"
#include "gl_field.hpp"
#include "kernels.hpp"

using gbk::u32;
using gbk::u64;

namespace {

thread_local std::string g_null_ctx_error;

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ScopeAcc {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
};

struct GlTableSet {
    gbk::GlNttTables t{};
    std::vector<void*> owned;
};
struct GlCosetSet {
    gbk::GlCosetTables t{};
    std::vector<void*> owned;
};
struct BbTableSet {
    gbk::BbNttTables t{};
    std::vector<void*> owned;
};
struct BbCosetSet {
    gbk::BbCosetTables t{};
    std::vector<void*> owned;
};

struct Stager;   // page-locked staging ring + copy threads for pageable host columns (below)

}  // namespace

struct gb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host -> device uploads that overlap kernels on `stream` (commit() of host input)
    std::string err;
    bool profiling = false;
    std::map<std::string, ScopeAcc> scopes;
    u64* tw4096_fwd = nullptr;
    u64* tw4096_inv = nullptr;
    u64 *tw4096_fwd_m = nullptr, *tw4096_inv_m = nullptr;  // times R (Montgomery form)
    std::map<u32, GlTableSet> gl_tables;                    // by log_n
    std::map<std::tuple<u32, u32, u64, int>, GlCosetSet> gl_cosets;  // by (log_n, rate_bits, shift, inverse)
    u32 *bb_tw4096_fwd = nullptr, *bb_tw4096_inv = nullptr;
    std::map<u32, BbTableSet> bb_tables;
    std::map<std::tuple<u32, u32, u32, int>, BbCosetSet> bb_cosets;  // by (log_n, rate_bits, shift (Montgomery), inverse)
    std::multimap<size_t, void*> pool;                      // freed batch blocks by size (stream-ordered reuse)
    DeviceBuf scratch;                                      // grow-only workspace
    DeviceBuf small;                                        // small gather staging (rows, siblings)
    DeviceBuf big_work[3];                                  // log_n > 22: work buffers of the outer radix steps (ntt_outer.hpp), one per nesting level
    hipEvent_t upload_mark = nullptr;                       // recorded on `stream` after the last host -> device copy of a commit
    bool upload_marked = false;
    std::vector<gb_circuit*> circuits;                      // live circuits of this context: what they keep for gb_prove_retry is
                                                            // released by gb_ctx_trim and when an allocation would fail
    Stager* stager = nullptr;                               // made on first use by a pageable host input
    // the prover's small transfers - uniform tables up, caps / openings / query rows down - go through page-locked memory of the
    // context (stage_up / ReadBack below): from pageable memory every one of them is a blocking, staged copy of 70-90 us
    char *xfer_up = nullptr, *xfer_down = nullptr;
    size_t xfer_up_off = 0;
    bool xfer_failed = false;
    // gb_ctx_set_option
    int copy_threads = 4;                                   // "copy_threads": -1 = no staging ring (hipMemcpyAsync straight from pageable memory)
    bool retry_verify = false;                              // "retry_verify": gb_prove_retry compares the whole matrix with the kept copy
};
static void drop_all_retry_state(gb_ctx* ctx);              // prover_host.inc

struct gb_batch {
    gb_ctx* ctx = nullptr;
    size_t coeffs_bytes = 0, lde_bytes = 0, levels_bytes = 0;
    u32 field = 0, log_n = 0, rate_bits = 0, cap_height = 0, nsalt = 0;
    size_t ncols = 0;
    u64* coeffs = nullptr;  // [ncols][n]
    u64* lde = nullptr;     // [ncols + nsalt][N], leaf order
    u64* levels = nullptr;  // level-major digests, level k at hash offset 2N - (2N >> k); last level = cap
};

namespace {

gb_status fail(gb_ctx* ctx, gb_status code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_null_ctx_error = msg;
    return code;
}

// ---- "never unwinds" (include/goldibear_gpu.h): EVERY extern "C" definition of this library is a function-try-block that ends in
// GB_CATCH / GB_CATCH_CIRCUIT (tests/test_abi_guards.py parses csrc/ and fails on one that does not) - a Rust caller's
// `panic = unwind` never crosses extern "C" either.  std::bad_alloc -> GB_ERR_OOM, any other std::exception -> GB_ERR_INVALID,
// anything else -> GB_ERR_HIP; device blocks and retry state are released by the guards of the frames the exception unwinds
// through (TmpAlloc, the batch / pool guards of commit() and prove()).  Writing the message must not throw either.
gb_status unwound(gb_ctx* ctx, const char* fn) noexcept {
    gb_status code = GB_ERR_HIP;
    std::string& dst = ctx ? ctx->err : g_null_ctx_error;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            code = GB_ERR_OOM;
            dst = std::string(fn) + ": host allocation failed";
        } catch (const std::exception& e) {
            code = GB_ERR_INVALID;
            dst = std::string(fn) + ": " + e.what();
        } catch (...) {
            dst = std::string(fn) + ": unknown exception";
        }
    } catch (...) {
        dst.clear();   // no memory for the message either: the status code says it
    }
    return code;
}
#define GB_CATCH(CTX) catch (...) { return unwound((CTX), __func__); }

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? GB_ERR_OOM : GB_ERR_HIP,                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

struct Scope {
    gb_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    hipStream_t on;
    Scope(gb_ctx* c, const char* n, hipStream_t stream = nullptr) : ctx(c), name(n), on(stream ? stream : c->stream) {
        if (!ctx->profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        hipEventRecord(a, on);
    }
    ~Scope() {
        if (!a) return;
        hipEventRecord(b, on);
        ctx->scopes[name].spans.emplace_back(a, b);
    }
};

// Host inputs of the public commit entry points stay the caller's only until the call returns (the reference moves its Vecs in):
// commit() marks the last host -> device copy it enqueues on the main stream, and gb_commit_* wait for that mark and for the copy
// stream before returning - the kernels behind the copies keep running asynchronously.
void mark_upload(gb_ctx* ctx) {
    if (!ctx->upload_mark && hipEventCreateWithFlags(&ctx->upload_mark, hipEventDisableTiming) != hipSuccess) {
        ctx->upload_mark = nullptr;
        (void)hipStreamSynchronize(ctx->stream);   // no event to wait on later: wait here
        return;
    }
    if (hipEventRecord(ctx->upload_mark, ctx->stream) == hipSuccess) ctx->upload_marked = true;
    else (void)hipStreamSynchronize(ctx->stream);
}
gb_status wait_uploads(gb_ctx* ctx) {
    hipError_t e = hipStreamSynchronize(ctx->copy_stream);
    if (e == hipSuccess && ctx->upload_marked) e = hipEventSynchronize(ctx->upload_mark);
    ctx->upload_marked = false;
    return e == hipSuccess ? GB_OK : fail(ctx, GB_ERR_HIP, "waiting for the upload of the host input failed");
}
// Every exit of a commit from HOST input: on success wait for the uploads (the kernels behind them keep running); after an error
// nothing of the call is worth keeping, so wait for both streams - copies from the caller's buffer and into pool blocks that
// commit()'s cleanup has already handed back may still be in flight - and keep the error that was reported first.  A batch
// whose uploads cannot be confirmed is freed, not returned.
gb_status finish_host_commit(gb_ctx* ctx, gb_status s, gb_batch** out) {
    if (!ctx) return s;
    if (s == GB_OK) {
        s = wait_uploads(ctx);
        if (s != GB_OK && out && *out) { gb_batch_free(*out); *out = nullptr; }
        return s;
    }
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->upload_marked = false;
    return s;
}

// ---- small transfers through page-locked memory (round 6).  A 2^12-row proof spent 1.1 of its 4.3 ms in 13 host round trips of
// ~80 us each (profiles/r06_recursion_shape.txt): hipMemcpyAsync from a std::vector is a synchronous, runtime-staged copy.
static constexpr size_t XFER_UP_BYTES = (size_t)4 << 20, XFER_DOWN_BYTES = (size_t)1 << 20;
bool xfer_ready(gb_ctx* ctx) {
    if (ctx->xfer_up) return true;
    if (ctx->xfer_failed) return false;
    void *u = nullptr, *d = nullptr;
    if (hipHostMalloc(&u, XFER_UP_BYTES, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&d, XFER_DOWN_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (u) (void)hipHostFree(u);
        ctx->xfer_failed = true;   // no page-locked memory to be had: the plain copies still work
        return false;
    }
    ctx->xfer_up = static_cast<char*>(u);
    ctx->xfer_down = static_cast<char*>(d);
    return true;
}
// `bytes` of page-locked memory to fill and hand to ONE hipMemcpyAsync on ctx->stream; it is not written again before the stream
// has passed that copy (the arena is a ring; wrapping around waits for the stream).  nullptr: too big or no arena - copy directly.
void* stage_up(gb_ctx* ctx, size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes > XFER_UP_BYTES / 4 || !xfer_ready(ctx)) return nullptr;
    if (ctx->xfer_up_off + bytes > XFER_UP_BYTES) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
        ctx->xfer_up_off = 0;
    }
    void* p = ctx->xfer_up + ctx->xfer_up_off;
    ctx->xfer_up_off += bytes;
    return p;
}
// device -> host copies that end in one wait: add() enqueues each into the page-locked buffer (or straight into the destination when
// it does not fit), finish() waits for the stream and hands the bytes out
struct ReadBack {
    gb_ctx* ctx;
    struct Item { void* dst; const char* src; size_t bytes; };
    Item items[48];
    unsigned n = 0;
    size_t off = 0;
    explicit ReadBack(gb_ctx* c) : ctx(c) {}
    gb_status add(void* host_dst, const void* dev, size_t bytes) {
        if (!bytes) return GB_OK;
        const size_t padded = (bytes + 63) & ~(size_t)63;
        if (n < 48 && off + padded <= XFER_DOWN_BYTES && xfer_ready(ctx)) {
            char* p = ctx->xfer_down + off;
            HIP_TRY(ctx, hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
            items[n++] = Item{host_dst, p, bytes};
            off += padded;
            return GB_OK;
        }
        HIP_TRY(ctx, hipMemcpyAsync(host_dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return GB_OK;
    }
    gb_status finish() {   // (polling an event before blocking was tried: no gain - the wake-up is not what these waits cost)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (unsigned i = 0; i < n; i++) std::memcpy(items[i].dst, items[i].src, items[i].bytes);
        n = 0; off = 0;
        return GB_OK;
    }
};
gb_status read_back(gb_ctx* ctx, void* host_dst, const void* dev, size_t bytes) {
    ReadBack rb(ctx);
    if (gb_status s = rb.add(host_dst, dev, bytes)) return s;
    return rb.finish();
}

gb_status ensure(gb_ctx* ctx, DeviceBuf& buf, size_t bytes) {
    if (buf.bytes >= bytes) return GB_OK;
    if (buf.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(buf.p));
        buf.p = nullptr;
        buf.bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&buf.p, bytes));
    buf.bytes = bytes;
    return GB_OK;
}

// Batch storage comes from a per-context pool: a prover commits batches of the same few shapes
// over and over, and hipMalloc/hipFree of multi-GiB blocks would otherwise dominate.  All work of
// a context is on one stream, so handing a freed block to a later commit is stream-ordered.
hipError_t pool_alloc(gb_ctx* ctx, size_t bytes, void** out) {
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end()) {
        *out = it->second;
        ctx->pool.erase(it);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        drop_all_retry_state(ctx);   // what failed attempts left for gb_prove_retry goes first (it lands in the pool)
        for (auto& kv : ctx->pool) (void)hipFree(kv.second);
        ctx->pool.clear();
        e = hipMalloc(out, bytes);
    }
    return e;
}
void pool_free(gb_ctx* ctx, void* p, size_t bytes) noexcept {   // (called from destructors of guards: must not throw)
    if (!p) return;
    try {
        ctx->pool.emplace(bytes, p);
    } catch (...) {   // no memory for the map node: hand the block back to HIP instead (hipFree waits for the device)
        (void)hipFree(p);
    }
}

gb_status upload(gb_ctx* ctx, const std::vector<u64>& host, u64** dev, std::vector<void*>* owned) {
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, host.size() * sizeof(u64)));
    HIP_TRY(ctx, hipMemcpy(p, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
    *dev = static_cast<u64*>(p);
    if (owned) owned->push_back(p);
    return GB_OK;
}

std::vector<u64> powers(u64 base, size_t count) {
    std::vector<u64> v(count);
    u64 x = 1;
    for (size_t i = 0; i < count; i++) {
        v[i] = x;
        x = gl::mul(x, base);
    }
    return v;
}

u32 bitrev32(u32 x, u32 bits) {
    u32 r = 0;
    for (u32 i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

std::vector<u64> times_r(const std::vector<u64>& v) {   // Montgomery form of a table: x R mod p, R = 2^64 mod p
    std::vector<u64> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = gl::to_mont_slow(v[i]);
    return o;
}

gb_status gl_tables_for(gb_ctx* ctx, u32 log_n, const gbk::GlNttTables** out) {
    auto it = ctx->gl_tables.find(log_n);
    if (it != ctx->gl_tables.end()) { *out = &it->second.t; return GB_OK; }
    if (!ctx->tw4096_fwd) {
        u64 w = gl::two_adic_generator(12);
        // [0, 4096): w_4096^j.  [4096, 8192): the same powers in the order k_gl_lde_pb16's first stage reads them,
        // T[s][m] = w_4096^(brev4(s) m), s < 16, m < 256 - a wave's 64 lanes then read 512 contiguous bytes per slot instead of
        // gathering at stride brev4(s) (up to 60 cache lines per load instruction).
        std::vector<u64> fwd = powers(w, 4096);
        fwd.resize(8192);
        for (u32 sl = 0; sl < 16; sl++)
            for (u32 m = 0; m < 256; m++) fwd[4096 + sl * 256 + m] = fwd[(bitrev32(sl, 4) * m) & 4095];
        gb_status s = upload(ctx, fwd, &ctx->tw4096_fwd, nullptr);
        if (s) return s;
        const std::vector<u64> inv = powers(gl::inv(w), 4096);
        s = upload(ctx, inv, &ctx->tw4096_inv, nullptr);
        if (s) return s;
        if ((s = upload(ctx, times_r(fwd), &ctx->tw4096_fwd_m, nullptr))) return s;
        if ((s = upload(ctx, times_r(inv), &ctx->tw4096_inv_m, nullptr))) return s;
    }
    GlTableSet set;
    set.t.log_n = log_n;
    set.t.sub = nullptr;
    set.t.tw16k_inv_m = nullptr;
    set.t.outer_bits = gbk::ntt_outer_bits(log_n);
    if (set.t.outer_bits) {   // the outer radix step (ntt_outer.hpp) runs smaller transforms: their tables first (std::map nodes do not move)
        const gbk::GlNttTables* sub;
        if (gb_status s2 = gl_tables_for(ctx, log_n - set.t.outer_bits, &sub)) return s2;
        set.t.sub = sub;
    } else if (log_n > 20) {  // 2^21 / 2^22 rows: the middle inverse pass is a radix-32 / radix-64 DFT with twiddles w_{2^14}^-j
        u64* t16;
        if (gb_status s2 = upload(ctx, times_r(powers(gl::inv(gl::two_adic_generator(14)), (size_t)1 << 14)), &t16, &set.owned)) return s2;
        set.t.tw16k_inv_m = t16;
    }
    set.t.tw4096_fwd = ctx->tw4096_fwd;
    set.t.tw4096_inv = ctx->tw4096_inv;
    set.t.tw4096_fwd_m = ctx->tw4096_fwd_m;
    set.t.tw4096_inv_m = ctx->tw4096_inv_m;
    u64 w = gl::two_adic_generator(log_n), wi = gl::inv(w);
    size_t n = (size_t)1 << log_n;
    size_t nhi = n > 1024 ? n / 1024 : 1;
    u64 *lo_f, *hi_f, *lo_i, *hi_i;
    gb_status s;
    const std::vector<u64> vlo_f = powers(w, 1024), vhi_f = powers(gl::pow(w, 1024), nhi), vlo_i = powers(wi, 1024),
                           vhi_i = powers(gl::pow(wi, 1024), nhi);
    if ((s = upload(ctx, vlo_f, &lo_f, &set.owned))) return s;
    if ((s = upload(ctx, vhi_f, &hi_f, &set.owned))) return s;
    if ((s = upload(ctx, vlo_i, &lo_i, &set.owned))) return s;
    if ((s = upload(ctx, vhi_i, &hi_i, &set.owned))) return s;
    set.t.tw_lo_fwd = lo_f; set.t.tw_hi_fwd = hi_f; set.t.tw_lo_inv = lo_i; set.t.tw_hi_inv = hi_i;
    set.t.n_inv = gl::inv((u64)n % gl::P);
    set.t.tw_top_fwd = nullptr;
    if (set.t.outer_bits) {
        const u32 log_m = log_n - set.t.outer_bits;
        std::vector<u64> top(256);
        for (u32 j = 0; j < 256; j++) top[j] = gl::pow(w, (u64)bitrev32(j, 8) << (log_m - 8));
        u64* t;
        if ((s = upload(ctx, top, &t, &set.owned))) return s;
        set.t.tw_top_fwd = t;
    }
    {
        u64 *a, *b, *c, *d;
        if ((s = upload(ctx, times_r(vlo_f), &a, &set.owned))) return s;
        if ((s = upload(ctx, times_r(vhi_f), &b, &set.owned))) return s;
        if ((s = upload(ctx, times_r(vlo_i), &c, &set.owned))) return s;
        if ((s = upload(ctx, times_r(vhi_i), &d, &set.owned))) return s;
        set.t.tw_lo_fwd_m = a; set.t.tw_hi_fwd_m = b; set.t.tw_lo_inv_m = c; set.t.tw_hi_inv_m = d;
        set.t.n_inv_m = gl::to_mont_slow(set.t.n_inv);
    }
    auto res = ctx->gl_tables.emplace(log_n, std::move(set));
    *out = &res.first->second.t;
    return GB_OK;
}

// coset c (block c of n leaves) has shift s_c = shift * w_N^bitrev_r(c): leaf j = c*n + jl is the LDE
// point shift * w_N^bitrev_logN(j) (fri/oracle.rs:109 + polynomial/mod.rs:282-295; shift = 7, or
// 7^(arity^l) for FRI layer l, fri/prover.rs:122-123).  inverse: tables of s_c^-1 instead.
// log_n > 22: the work buffers of the outer radix steps (de-interleaved coefficients + sub-LDEs of a group of columns), one per nesting
// level, grown on demand before every use: the coset tables hold the ADDRESS of the context's pointer, not the pointer
inline u32 big_work_level(u32 log_n) { return (log_n - gbk::NTT_NATIVE_LOG - 1) / gbk::NTT_OUTER_MAX_BITS; }
gb_status ensure_big_work(gb_ctx* ctx, u32 log_n, u32 rate_bits, size_t es) {
    for (; log_n > gbk::NTT_NATIVE_LOG; log_n -= gbk::ntt_outer_bits(log_n)) {
        const size_t n = (size_t)1 << log_n, per_col = (n + (n << rate_bits)) * es;
        const size_t cols = std::min<size_t>(8, std::max<size_t>(1, ((size_t)8 << 30) / per_col));   // groups of up to 8 columns within 8 GiB
        if (gb_status s = ensure(ctx, ctx->big_work[big_work_level(log_n)], cols * per_col)) return s;
    }
    return GB_OK;
}
gb_status gl_cosets_for(gb_ctx* ctx, u32 log_n, u32 rate_bits, u64 shift, bool inverse, const gbk::GlCosetTables** out) {
    auto key = std::make_tuple(log_n, rate_bits, shift, inverse ? 1 : 0);
    auto it = ctx->gl_cosets.find(key);
    if (it != ctx->gl_cosets.end()) {
        if (!inverse)   // (released by gb_ctx_trim)
            if (gb_status sw = ensure_big_work(ctx, log_n, rate_bits, sizeof(u64))) return sw;
        *out = &it->second.t;
        return GB_OK;
    }
    size_t n = (size_t)1 << log_n;
    size_t nlo = n < 4096 ? n : 4096, nhi = n > 4096 ? n / 4096 : 1;
    u32 nc = 1u << rate_bits;
    u64 wN = gl::two_adic_generator(log_n + rate_bits);
    std::vector<u64> lo(nc * nlo), hi(nc * nhi);
    for (u32 c = 0; c < nc; c++) {
        u64 s = gl::mul(shift, gl::pow(wN, bitrev32(c, rate_bits)));
        if (inverse) s = gl::inv(s);
        std::vector<u64> pl = powers(s, nlo), ph = powers(gl::pow(s, 4096), nhi);
        std::memcpy(&lo[c * nlo], pl.data(), nlo * sizeof(u64));
        std::memcpy(&hi[c * nhi], ph.data(), nhi * sizeof(u64));
    }
    GlCosetSet set;
    set.t.rate_bits = rate_bits;
    set.t.sub = nullptr; set.t.work = nullptr; set.t.work_bytes = nullptr;
    // inverse tables are read as power tables only (the quotient's coset_ifft); the LDE kernels never see them
    if (const u32 K = gbk::ntt_outer_bits(log_n); K && !inverse) {   // sub-transforms of n / R rows on the shift s_c^R, R = 2^K (ntt_outer.hpp)
        const gbk::GlCosetTables* sub;
        if (gb_status s2 = gl_cosets_for(ctx, log_n - K, rate_bits, gl::pow(shift, (u64)1 << K), false, &sub)) return s2;
        set.t.sub = sub;
        if (gb_status sw = ensure_big_work(ctx, log_n, rate_bits, sizeof(u64))) return sw;
        DeviceBuf& wb = ctx->big_work[big_work_level(log_n)];
        set.t.work = &wb.p; set.t.work_bytes = &wb.bytes;
    }
    u64 *dlo, *dhi;
    gb_status s;
    if ((s = upload(ctx, lo, &dlo, &set.owned))) return s;
    if ((s = upload(ctx, hi, &dhi, &set.owned))) return s;
    set.t.pow_lo = dlo; set.t.pow_hi = dhi;
    {
        u64 *a, *b;
        if ((s = upload(ctx, times_r(lo), &a, &set.owned))) return s;
        if ((s = upload(ctx, times_r(hi), &b, &set.owned))) return s;
        set.t.pow_lo_m = a; set.t.pow_hi_m = b;
    }
    auto res = ctx->gl_cosets.emplace(key, std::move(set));
    *out = &res.first->second.t;
    return GB_OK;
}

// ---- BabyBear tables (Montgomery form) ----
gb_status upload32(gb_ctx* ctx, const std::vector<u32>& host, u32** dev, std::vector<void*>* owned) {
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, host.size() * sizeof(u32)));
    HIP_TRY(ctx, hipMemcpy(p, host.data(), host.size() * sizeof(u32), hipMemcpyHostToDevice));
    *dev = static_cast<u32*>(p);
    if (owned) owned->push_back(p);
    return GB_OK;
}
std::vector<u32> bb_powers(u32 base_mont, size_t count) {
    std::vector<u32> v(count);
    u32 x = bb::R1;
    for (size_t i = 0; i < count; i++) {
        v[i] = x;
        x = bb::mul(x, base_mont);
    }
    return v;
}
gb_status bb_tables_for(gb_ctx* ctx, u32 log_n, const gbk::BbNttTables** out) {
    auto it = ctx->bb_tables.find(log_n);
    if (it != ctx->bb_tables.end()) { *out = &it->second.t; return GB_OK; }
    gb_status s;
    if (!ctx->bb_tw4096_fwd) {
        u32 w = bb::two_adic_generator(12);
        if ((s = upload32(ctx, bb_powers(w, 4096), &ctx->bb_tw4096_fwd, nullptr))) return s;
        if ((s = upload32(ctx, bb_powers(bb::inv(w), 4096), &ctx->bb_tw4096_inv, nullptr))) return s;
    }
    BbTableSet set;
    set.t.log_n = log_n;
    set.t.sub = set.t.wide = nullptr;
    set.t.tw16k_inv = nullptr;
    set.t.outer_bits = gbk::ntt_outer_bits(log_n);
    if (set.t.outer_bits) {
        const gbk::BbNttTables* sub;
        if (gb_status s2 = bb_tables_for(ctx, log_n - set.t.outer_bits, &sub)) return s2;
        set.t.sub = sub;
    } else if (log_n > 20) {
        if (log_n == 22) {   // the strided LDE pass of 2^22 rows works with the 2^20-row twiddles (k_bb_lde_pa16x2w)
            const gbk::BbNttTables* w20;
            if (gb_status s2 = bb_tables_for(ctx, 20, &w20)) return s2;
            set.t.wide = w20;
        }
        u32* t16;
        if (gb_status s2 = upload32(ctx, bb_powers(bb::inv(bb::two_adic_generator(14)), (size_t)1 << 14), &t16, &set.owned)) return s2;
        set.t.tw16k_inv = t16;
    }
    set.t.tw4096_fwd = ctx->bb_tw4096_fwd;
    set.t.tw4096_inv = ctx->bb_tw4096_inv;
    u32 w = bb::two_adic_generator(log_n), wi = bb::inv(w);
    size_t n = (size_t)1 << log_n, nhi = n > 1024 ? n / 1024 : 1;
    u32 *lo_f, *hi_f, *lo_i, *hi_i;
    if ((s = upload32(ctx, bb_powers(w, 1024), &lo_f, &set.owned))) return s;
    if ((s = upload32(ctx, bb_powers(bb::pow(w, 1024), nhi), &hi_f, &set.owned))) return s;
    if ((s = upload32(ctx, bb_powers(wi, 1024), &lo_i, &set.owned))) return s;
    if ((s = upload32(ctx, bb_powers(bb::pow(wi, 1024), nhi), &hi_i, &set.owned))) return s;
    set.t.tw_lo_fwd = lo_f; set.t.tw_hi_fwd = hi_f; set.t.tw_lo_inv = lo_i; set.t.tw_hi_inv = hi_i;
    set.t.n_inv = bb::inv(bb::to_mont((u32)(n % bb::P)));
    set.t.tw_top_fwd = nullptr;
    if (set.t.outer_bits) {
        const u32 log_m = log_n - set.t.outer_bits;
        std::vector<u32> top(256);
        for (u32 j = 0; j < 256; j++) top[j] = bb::pow(w, (u64)bitrev32(j, 8) << (log_m - 8));
        u32* t;
        if ((s = upload32(ctx, top, &t, &set.owned))) return s;
        set.t.tw_top_fwd = t;
    }
    auto res = ctx->bb_tables.emplace(log_n, std::move(set));
    *out = &res.first->second.t;
    return GB_OK;
}
gb_status bb_cosets_for(gb_ctx* ctx, u32 log_n, u32 rate_bits, u32 shift_mont, bool inverse, const gbk::BbCosetTables** out) {
    auto key = std::make_tuple(log_n, rate_bits, shift_mont, inverse ? 1 : 0);
    auto it = ctx->bb_cosets.find(key);
    if (it != ctx->bb_cosets.end()) {
        if (!inverse)
            if (gb_status sw = ensure_big_work(ctx, log_n, rate_bits, sizeof(u32))) return sw;
        *out = &it->second.t;
        return GB_OK;
    }
    size_t n = (size_t)1 << log_n;
    size_t nlo = n < 4096 ? n : 4096, nhi = n > 4096 ? n / 4096 : 1;
    u32 nc = 1u << rate_bits;
    u32 wN = bb::two_adic_generator(log_n + rate_bits);
    std::vector<u32> lo(nc * nlo), hi(nc * nhi);
    for (u32 c = 0; c < nc; c++) {
        u32 s = bb::mul(shift_mont, bb::pow(wN, bitrev32(c, rate_bits)));
        if (inverse) s = bb::inv(s);
        std::vector<u32> pl = bb_powers(s, nlo), ph = bb_powers(bb::pow(s, 4096), nhi);
        std::memcpy(&lo[c * nlo], pl.data(), nlo * sizeof(u32));
        std::memcpy(&hi[c * nhi], ph.data(), nhi * sizeof(u32));
    }
    BbCosetSet set;
    set.t.rate_bits = rate_bits;
    set.t.sub = set.t.fine = nullptr; set.t.work = nullptr; set.t.work_bytes = nullptr;
    if (const u32 K = gbk::ntt_outer_bits(log_n); K && !inverse) {
        const gbk::BbCosetTables* sub;
        if (gb_status s2 = bb_cosets_for(ctx, log_n - K, rate_bits, bb::pow(shift_mont, (u64)1 << K), false, &sub)) return s2;
        set.t.sub = sub;
        if (gb_status sw = ensure_big_work(ctx, log_n, rate_bits, sizeof(u32))) return sw;
        DeviceBuf& wb = ctx->big_work[big_work_level(log_n)];
        set.t.work = &wb.p; set.t.work_bytes = &wb.bytes;
    } else if (log_n == 22 && !inverse) {   // the finer cosets of 2^20 rows (k_bb_lde_pa16x2w)
        const gbk::BbCosetTables* fine;
        if (gb_status s2 = bb_cosets_for(ctx, 20, rate_bits + 2, shift_mont, false, &fine)) return s2;
        set.t.fine = fine;
    }
    u32 *dlo, *dhi;
    gb_status s;
    if ((s = upload32(ctx, lo, &dlo, &set.owned))) return s;
    if ((s = upload32(ctx, hi, &dhi, &set.owned))) return s;
    set.t.pow_lo = dlo; set.t.pow_hi = dhi;
    auto res = ctx->bb_cosets.emplace(key, std::move(set));
    *out = &res.first->second.t;
    return GB_OK;
}

// internal flag (not part of the C ABI): device input already in the field's device form (BabyBear: Montgomery)
static constexpr uint32_t GB_INPUT_DEVICE_FORM = 0x100;
static constexpr uint32_t GB_PUBLIC_INPUT_FLAGS = GB_INPUT_DEVICE | GB_INPUT_P3_REPR;   // what a caller may pass

// A matrix of input columns as the caller holds it: one contiguous [ncols][n] block (`base`), or ncols separately allocated
// columns (`ptrs`) - the reference's Vec<PolynomialValues<F>> (fri/oracle.rs:68-75) and MatrixWitness.wire_values: Vec<Vec<F>>
// (iop/witness.rs:277-279).
struct ColSrc {
    const void* base = nullptr;
    const void* const* ptrs = nullptr;
    ColSrc() {}
    ColSrc(const void* b) : base(b) {}   // NOLINT: the contiguous form converts implicitly
    static ColSrc columns(const void* const* p) { ColSrc s; s.ptrs = p; return s; }
    const char* col(size_t c, size_t col_bytes) const {
        return ptrs ? static_cast<const char*>(ptrs[c]) : static_cast<const char*>(base) + c * col_bytes;
    }
    explicit operator bool() const { return base || ptrs; }
};

// hipHostMalloc / hipHostRegister memory (gb_host_alloc, gb_host_register): the copy engine reads it directly
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
bool all_pinned(const ColSrc& src, size_t ncols, size_t col_bytes) {
    if (!src.ptrs) return is_pinned_host(src.base) && is_pinned_host(static_cast<const char*>(src.base) + ncols * col_bytes - 1);
    for (size_t c = 0; c < ncols; c++)
        if (!src.ptrs[c] || !is_pinned_host(src.ptrs[c])) return false;
    return true;
}

// ---- pageable host columns ------------------------------------------------------------------------------------------------------
// hipMemcpyAsync from pageable memory is a synchronous copy that the runtime stages itself, on the calling thread.  A context
// instead owns a page-locked ring (SLOTS slots) and a few copy threads: the columns of an upload chunk are copied into a slot in
// pieces (column 0 first), and the thread that drives the commit issues the host -> device copy of every column from the ring as
// soon as its last piece has landed - while the copy engine moves the slot before and the GPU transforms and hashes the chunk
// before.  The caller's columns are read by the time the call returns; the ring is the library's.
struct Stager {
    static constexpr unsigned SLOTS = 4;
    static constexpr size_t PIECE = (size_t)1 << 20, SLOT_MIN = (size_t)64 << 20;
    struct Piece { const char* src; char* dst; size_t bytes; u32 col; };
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    u64 generation = 0;      // guarded by m
    unsigned parked = 0;     // workers waiting for a job (guarded by m)
    bool stop = false;
    std::vector<Piece> pieces;                       // the job in flight: written only while every worker is parked
    std::atomic<size_t> next{0};
    std::unique_ptr<std::atomic<u32>[]> remaining;   // pieces left per column of the job
    size_t remaining_cap = 0;
    char* ring = nullptr;
    size_t slot_bytes = 0;
    unsigned slot = 0;
    hipEvent_t slot_done[SLOTS] = {};
    bool slot_used[SLOTS] = {};

    explicit Stager(unsigned nthreads) {
        try {   // the C ABI never unwinds: a thread the system refuses is one copy thread less (none: the calling thread copies)
            for (unsigned i = 0; i < nthreads; i++) workers.emplace_back([this] { work(); });
        } catch (...) {
        }
    }
    ~Stager() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
        for (unsigned i = 0; i < SLOTS; i++)
            if (slot_done[i]) { (void)hipEventSynchronize(slot_done[i]); (void)hipEventDestroy(slot_done[i]); }
        if (ring) (void)hipHostFree(ring);
    }
    void work() {
        std::unique_lock<std::mutex> lk(m);
        u64 seen = 0;
        for (;;) {
            parked++;
            cv_done.notify_all();
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            parked--;
            if (stop) return;
            seen = generation;
            lk.unlock();
            run_pieces();
            lk.lock();
        }
    }
    void run_pieces() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pieces.size()) return;
            const Piece& p = pieces[i];
            std::memcpy(p.dst, p.src, p.bytes);
            if (remaining[p.col].fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(m);
                cv_done.notify_all();
            }
        }
    }
    bool reserve(size_t need, hipStream_t copy_stream) {   // slots of at least `need` bytes
        if (slot_bytes >= need) return true;
        if (ring) {
            if (hipStreamSynchronize(copy_stream) != hipSuccess) return false;
            (void)hipHostFree(ring);
            ring = nullptr; slot_bytes = 0;
            std::fill(slot_used, slot_used + SLOTS, false);
        }
        const size_t sb = std::max(need, SLOT_MIN);
        void* p = nullptr;
        if (hipHostMalloc(&p, sb * SLOTS, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
        ring = static_cast<char*>(p);
        slot_bytes = sb;
        return true;
    }
    // columns [c0, c0 + cc) of `src` -> dst (device, contiguous [cc][col_bytes]) through the ring, on `copy_stream`
    bool upload(const ColSrc& src, size_t c0, size_t cc, size_t col_bytes, char* dst, hipStream_t copy_stream) {
        if (!reserve(col_bytes, copy_stream)) return false;
        const size_t per_slot = std::max<size_t>(1, slot_bytes / col_bytes);
        for (size_t b0 = 0; b0 < cc; b0 += per_slot) {
            const size_t bc = std::min(per_slot, cc - b0);
            const unsigned s = slot;
            slot = (slot + 1) % SLOTS;
            if (slot_used[s] && hipEventSynchronize(slot_done[s]) != hipSuccess) return false;   // the copy engine has left the slot
            char* base = ring + (size_t)s * slot_bytes;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_done.wait(lk, [&] { return parked == workers.size(); });   // nobody is still inside the job before
                pieces.clear();
                if (remaining_cap < bc) { remaining.reset(new std::atomic<u32>[bc]); remaining_cap = bc; }
                for (size_t c = 0; c < bc; c++) {
                    const char* sp = src.col(c0 + b0 + c, col_bytes);
                    u32 np = 0;
                    for (size_t o = 0; o < col_bytes; o += PIECE, np++)
                        pieces.push_back(Piece{sp + o, base + c * col_bytes + o, std::min(PIECE, col_bytes - o), (u32)c});
                    remaining[c].store(np, std::memory_order_relaxed);
                }
                next.store(0, std::memory_order_relaxed);
                generation++;
            }
            cv_work.notify_all();
            if (workers.empty()) run_pieces();   // "copy_threads" = 0: the calling thread copies
            auto drain = [&] {   // an error return must not leave a worker reading the caller's columns
                std::unique_lock<std::mutex> lk(m);
                cv_done.wait(lk, [&] { return next.load(std::memory_order_relaxed) >= pieces.size() && parked == workers.size(); });
                return false;
            };
            for (size_t c = 0; c < bc;) {
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv_done.wait(lk, [&] { return remaining[c].load(std::memory_order_acquire) == 0; });
                }
                size_t e = c + 1;   // every further column that is staged already goes into the same copy
                while (e < bc && remaining[e].load(std::memory_order_acquire) == 0) e++;
                if (hipMemcpyAsync(dst + (b0 + c) * col_bytes, base + c * col_bytes, (e - c) * col_bytes, hipMemcpyHostToDevice,
                                   copy_stream) != hipSuccess)
                    return drain();
                c = e;
            }
            // no event to guard the slot with: the copies out of it are in flight, so wait for them here before anybody may write
            // into the slot again (the caller falls back to plain copies for the rest)
            if (!slot_done[s] && hipEventCreateWithFlags(&slot_done[s], hipEventDisableTiming) != hipSuccess) {
                slot_done[s] = nullptr;
                (void)hipStreamSynchronize(copy_stream);
                return false;
            }
            if (hipEventRecord(slot_done[s], copy_stream) != hipSuccess) {
                (void)hipStreamSynchronize(copy_stream);
                return false;
            }
            slot_used[s] = true;
        }
        return true;
    }
};

// Host columns [c0, c0 + cc) -> dst (device, contiguous) on the context's copy stream.  Page-locked sources go straight to the copy
// engine (one copy for a contiguous block, one per column otherwise); pageable ones through the staging ring, small ones excepted.
bool upload_columns(gb_ctx* ctx, const ColSrc& src, bool pinned, size_t c0, size_t cc, size_t col_bytes, void* dst) {
    char* d = static_cast<char*>(dst);
    const bool ring = !pinned && ctx->copy_threads >= 0 && cc * col_bytes >= ((size_t)2 << 20);
    if (ring) {
        bool done = false;
        try {   // (std::bad_alloc from the piece list: thrown before any worker has been handed the job)
            if (!ctx->stager) ctx->stager = new (std::nothrow) Stager((unsigned)ctx->copy_threads);
            done = ctx->stager && ctx->stager->upload(src, c0, cc, col_bytes, d, ctx->copy_stream);
        } catch (...) {
            done = false;
        }
        if (done) return true;
        (void)hipGetLastError();   // no ring (out of page-locked memory?): the runtime's own pageable path still works
    }
    if (!src.ptrs) return hipMemcpyAsync(d, src.col(c0, col_bytes), cc * col_bytes, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess;
    for (size_t c = 0; c < cc; c++)
        if (hipMemcpyAsync(d + c * col_bytes, src.col(c0 + c, col_bytes), col_bytes, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess)
            return false;
    return true;
}
// the same on an arbitrary stream for inputs that are not chunked (small batches, coefficients): plain copies
bool copy_columns(const ColSrc& src, size_t ncols, size_t col_bytes, void* dst, hipMemcpyKind kind, hipStream_t st) {
    char* d = static_cast<char*>(dst);
    if (!src.ptrs) return hipMemcpyAsync(d, src.base, ncols * col_bytes, kind, st) == hipSuccess;
    for (size_t c = 0; c < ncols; c++)
        if (hipMemcpyAsync(d + c * col_bytes, src.ptrs[c], col_bytes, kind, st) != hipSuccess) return false;
    return true;
}

inline size_t hout(u32 field) { return field == GB_GOLDILOCKS ? 4 : 8; }
inline size_t esize(u32 field) { return field == GB_GOLDILOCKS ? 8 : 4; }
inline size_t level_offset(u64 N, u32 k) { return (size_t)(2 * N - ((2 * N) >> k)); }

struct EventList {  // events of one enqueue sequence, destroyed together (destruction is deferred until they have completed)
    std::vector<hipEvent_t> v;
    hipEvent_t make(bool& ok) {
        hipEvent_t e = nullptr;
        if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) v.push_back(e);
        else ok = false;
        return e;
    }
    ~EventList() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
};

// from_values of HOST input (2^12 rows and up): the upload runs in column chunks on the context's copy stream while the inverse NTT
// and LDE of the chunk before it run on the main stream.  values_dev (optional): a device buffer [ncols][n] that receives the input
// values in the field's device form, so that the caller gets them on the device without a second transfer (prove() needs the
// routed wires there for the permutation argument).
// What a segmented host-input commit can leave behind for a caller that may have to redo its LAST column segment (the prover's
// InvZeroPermArg retry re-draws one wire of the last column): the sponge state every leaf had after columns [0, start).
struct SegKeep {
    void* state = nullptr;   // pool block; the taker returns it with pool_free(ctx, state, bytes)
    size_t bytes = 0;
    u32 start = 0;           // first column of the last segment
};
// Upload chunks of a host batch, in columns: 4, 4, 8, then `full`.  The GPU's per-column work (transform + leaf hashing) is slower
// than PCIe delivers columns, so after a short ramp the upload is hidden - what is not hidden is the wait for the FIRST columns:
// with 4 + 4 the first 8-column hashing segment starts after ~1.5 ms of a 2^20-row Goldilocks witness (4 + 12: ~3.5 ms).
static inline size_t first_chunks(size_t c0, size_t full) { return c0 < 8 ? 4 : c0 < 16 ? 8 : full; }

gb_status commit(gb_ctx* ctx, uint32_t field, ColSrc cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                 uint32_t cap_height, const void* salts, uint32_t flags, bool is_coeffs, gb_batch** out, void* values_dev = nullptr,
                 SegKeep* keep = nullptr, size_t* values_mont_cols = nullptr) {
    // values_mont_cols (BabyBear, with values_dev): in - how many leading columns of values_dev the caller reads as device-form
    // VALUES afterwards; out - how many leading columns of values_dev ARE in device (Montgomery) form, the rest being canonical as
    // uploaded (the inverse transform takes canonical input where its radix-16 kernels cover the shape; otherwise all of them)
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    if (!out) return fail(ctx, GB_ERR_INVALID, "null out");
    *out = nullptr;
    if (field != GB_GOLDILOCKS && field != GB_BABYBEAR) return fail(ctx, GB_ERR_INVALID, "unknown field tag");
    if (ncols == 0) return fail(ctx, GB_ERR_INVALID, "from_values/from_coeffs needs at least one polynomial (oracle.rs:101)");
    if (!cols) return fail(ctx, GB_ERR_INVALID, "null cols");
    if (cols.ptrs)
        for (size_t c = 0; c < ncols; c++)
            if (!cols.ptrs[c]) return fail(ctx, GB_ERR_INVALID, "null column pointer");
    // GB_INPUT_P3_REPR: the elements are the in-memory words of the reference's field types (p3-goldilocks: any u64 representative;
    // p3-baby-bear / p3-monty-31: x 2^32 mod p) - a host-memory convention
    const bool p3 = (flags & GB_INPUT_P3_REPR) != 0;
    if (p3 && (flags & GB_INPUT_DEVICE)) return fail(ctx, GB_ERR_INVALID, "GB_INPUT_P3_REPR describes host memory; device inputs are canonical");
    if (log_n + rate_bits > (field == GB_GOLDILOCKS ? 32u : 27u))
        return fail(ctx, GB_ERR_INVALID, "LDE size exceeds the field's two-adicity (32 Goldilocks / 27 BabyBear)");
    if (cap_height > log_n + rate_bits)
        return fail(ctx, GB_ERR_INVALID, "cap_height should be at most log2(leaves.len()) (merkle_tree.rs:154-157)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const u32 log_N = log_n + rate_bits;
    const u32 nsalt = salts ? GB_SALT_SIZE : 0;
    const size_t width = ncols + nsalt;
    const bool dev_in = (flags & GB_INPUT_DEVICE) != 0;

    gb_batch* b = new (std::nothrow) gb_batch();
    if (!b) return fail(ctx, GB_ERR_OOM, "host allocation failed");
    b->ctx = ctx; b->field = field; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->nsalt = nsalt; b->ncols = ncols;
    auto cleanup = [&](gb_status s) {
        gb_batch_free(b);
        return s;
    };

    void* p = nullptr;
    const size_t es = esize(field);
    b->coeffs_bytes = ncols * n * es;
    b->lde_bytes = width * N * es;
    b->levels_bytes = 2 * N * 4 * sizeof(u64);  // 32-byte digests for both hashers
    if (pool_alloc(ctx, b->coeffs_bytes, &p) != hipSuccess) return cleanup(fail(ctx, GB_ERR_OOM, "hipMalloc coeffs"));
    b->coeffs = (u64*)p;
    if (pool_alloc(ctx, b->lde_bytes, &p) != hipSuccess) return cleanup(fail(ctx, GB_ERR_OOM, "hipMalloc lde"));
    b->lde = (u64*)p;
    if (pool_alloc(ctx, b->levels_bytes, &p) != hipSuccess) return cleanup(fail(ctx, GB_ERR_OOM, "hipMalloc digests"));
    b->levels = (u64*)p;

    gb_status s;
    hipStream_t st = ctx->stream;
    // separately allocated DEVICE columns: gathered into one block, then the contiguous path
    struct GatherGuard {
        gb_ctx* ctx; void* p = nullptr; size_t bytes = 0;
        ~GatherGuard() { if (p) pool_free(ctx, p, bytes); }   // stream-ordered reuse: every reader is enqueued before the next taker
    } gather{ctx};
    if (dev_in && cols.ptrs) {
        gather.bytes = ncols * n * es;
        if (pool_alloc(ctx, gather.bytes, &gather.p) != hipSuccess) { gather.p = nullptr; return cleanup(fail(ctx, GB_ERR_OOM, "hipMalloc column gather")); }
        if (!copy_columns(cols, ncols, n * es, gather.p, hipMemcpyDeviceToDevice, st)) return cleanup(fail(ctx, GB_ERR_HIP, "copy of input columns failed"));
        cols = ColSrc(gather.p);
    }
    const bool pinned = !dev_in && all_pinned(cols, ncols, n * es);
    // Host input of a big batch: the leaf sponges run in segments of SEG columns as the columns arrive (chunked upload below), so
    // that the hashing - most of a commitment's time - overlaps the PCIe transfer instead of waiting for its end; the sponge state
    // waits in `seg_state` between segments (kernels_merkle.hip / kernels_bb.hip: k_*_merkle_leaves).
    // Segment sizes: 8, 24, 32, then 64 columns, the last segment being what is left after the last multiple of eight (at most eight
    // columns more).  The first segment is short so that the hashing starts as soon as eight columns have been transformed (~1.5 ms of
    // a 2^20-row Goldilocks witness) instead of idling until more have crossed PCIe; from then on the GPU is behind the upload, and
    // segments are long because every one of them costs a kernel's ramp and tail (~0.3 ms at 2^23 leaves: T(k absorptions) = 0.3 +
    // 2.41 k ms, measured) and a round trip of the sponge state through HBM (0.27 GB); the last one is short because it is what an
    // InvZeroPermArg retry hashes again (gb_prove_retry).  135 columns: 8, 24, 32, 64, 7 (rounds 2-3: 8, 8, 16, 32, 32, 32, 7).
    constexpr u32 SEG = 32;
    auto seg_size = [ncols](u32 done) -> u32 {
        if (done < 8) return 8u;
        if (done < 32) return 24u;
        if (done < 64) return 32u;
        const u32 left = (u32)ncols - done;                 // leave the last (ragged or whole) group of eight to the last segment
        return std::max<u32>(8u, std::min<u32>(64u, left > 1 ? 8u * ((left - 1) / 8) : 8u));
    };
    const bool segmented = !dev_in && !is_coeffs && log_N >= 19 && ncols > SEG;
    const bool keep_split = keep && dev_in && !is_coeffs && log_N >= 19 && ncols > SEG;   // same split for device input, on request
    u32 seg_done = 0;
    void* seg_state = nullptr;
    // the parked sponge state: the capacity words, plus the rate words a ragged LAST absorption leaves alone when it is a segment
    // of its own (kernels_merkle.hip / kernels_bb.hip index the rows compactly)
    u32 last_seg_start = 0;
    for (u32 sz = seg_size(0); last_seg_start + sz < ncols; sz = seg_size(last_seg_start)) last_seg_start += sz;
    const u32 kf_final = std::min<u32>(8u, (u32)(width - last_seg_start));
    const size_t seg_rows = (field == GB_GOLDILOCKS ? 4 : 8) + (8 - kf_final);
    const size_t seg_state_bytes = seg_rows * N * (field == GB_GOLDILOCKS ? sizeof(u64) : sizeof(u32));
    struct SegGuard {
        gb_ctx* ctx; void** p; size_t bytes;
        ~SegGuard() { if (*p) pool_free(ctx, *p, bytes); }
    } seg_guard{ctx, &seg_state, seg_state_bytes};
    auto hash_ready_segments = [&](size_t cols_ready) -> bool {   // every full segment that is not the last one
        for (u32 sz = seg_size(seg_done); segmented && seg_done + sz < ncols && seg_done + sz <= cols_ready; sz = seg_size(seg_done)) {
            if (!seg_state && pool_alloc(ctx, seg_state_bytes, &seg_state) != hipSuccess) { seg_state = nullptr; return false; }
            Scope sm(ctx, "build Merkle tree");
            Scope sl(ctx, "hash leaves");
            const u32 next_cols = (u32)(width - (seg_done + sz));
            if (field == GB_GOLDILOCKS)
                gbk::gl_merkle_leaves_segment(b->lde, N, seg_done, seg_done + sz, N, (u64*)seg_state, false, next_cols, b->levels, st);
            else
                gbk::bb_merkle_leaves_segment((const u32*)b->lde, N, seg_done, seg_done + sz, N, (u32*)seg_state, false, next_cols,
                                              (u32*)b->levels, st);
            seg_done += sz;
        }
        return true;
    };
    if (field == GB_BABYBEAR) {
        // same flow over u32 Montgomery words; inputs are converted on the way in
        const gbk::BbNttTables* bt;
        const gbk::BbCosetTables* bc;
        if ((s = bb_tables_for(ctx, log_n, &bt))) return cleanup(s);
        if ((s = bb_cosets_for(ctx, log_n, rate_bits, bb::to_mont(bb::GENERATOR), false, &bc))) return cleanup(s);
        u32* coeffs = (u32*)b->coeffs;
        u32* lde = (u32*)b->lde;
        const size_t in_bytes = ncols * n * 4, scr_bytes = std::max(in_bytes, (size_t)nsalt * N * 4);
        if ((s = ensure(ctx, ctx->scratch, 2 * scr_bytes))) return cleanup(s);
        u32* scr = (u32*)ctx->scratch.p;
        const u32* in_dev = static_cast<const u32*>(cols.base);
        const bool staged = !dev_in && !is_coeffs && log_n >= 12 && ncols >= 4;
        if (staged) {
            // column chunks: H2D straight into values_dev / the coefficient block (copy stream) -> Montgomery form in place ->
            // inverse NTT -> LDE -> the leaf-sponge segments that have all their columns
            const size_t per = 16;
            u32* vals = values_dev ? static_cast<u32*>(values_dev) : coeffs;  // without a taker the values are transformed in place
            u32* ntt_scr = scr + scr_bytes / 4;
            EventList evs;
            bool ok = true;
            hipEvent_t e0 = evs.make(ok);                           // values_dev / coeffs may be a pool block still in use on `st`
            ok = ok && hipEventRecord(e0, st) == hipSuccess && hipStreamWaitEvent(ctx->copy_stream, e0, 0) == hipSuccess;
            for (size_t c0 = 0, cc = 0; c0 < ncols && ok; c0 += cc) {
                cc = std::min(c0 < 32 ? 2 * first_chunks(c0 / 2, per / 2) : per, ncols - c0);   // 4-byte words: the same bytes per chunk as Goldilocks' 4, 4, 8
                hipEvent_t copied = evs.make(ok);
                ok = ok && upload_columns(ctx, cols, pinned, c0, cc, n * 4, vals + c0 * n) &&
                     hipEventRecord(copied, ctx->copy_stream) == hipSuccess && hipStreamWaitEvent(st, copied, 0) == hipSuccess;
                if (!ok) break;
                const size_t want_mont = !values_dev ? 0 : values_mont_cols ? *values_mont_cols : ncols;
                bool direct;
                if (p3) {   // Montgomery words as they are in the host's memory: nothing to convert, but nothing to trust either
                    gbk::bb_reduce_words(vals + c0 * n, cc * n, st);
                    Scope sc(ctx, "IFFT");
                    gbk::bb_intt_columns(vals + c0 * n, coeffs + c0 * n, ntt_scr, cc, *bt, st);
                    if (values_mont_cols) *values_mont_cols = ncols;
                    direct = true;
                } else {
                    Scope sc(ctx, "IFFT");
                    direct = gbk::bb_intt_columns_canonical(vals + c0 * n, coeffs + c0 * n, ntt_scr, cc, want_mont > c0 ? want_mont - c0 : 0, *bt, st);
                }
                if (!direct) {
                    gbk::bb_to_mont(vals + c0 * n, vals + c0 * n, cc * n, st);
                    { Scope sc(ctx, "IFFT"); gbk::bb_intt_columns(vals + c0 * n, coeffs + c0 * n, ntt_scr, cc, *bt, st); }
                    if (values_mont_cols) *values_mont_cols = ncols;
                }
                { Scope sc(ctx, "FFT + blinding"); gbk::bb_lde_columns(coeffs + c0 * n, lde + c0 * N, cc, *bt, *bc, st); }
                if (!hash_ready_segments(c0 + cc)) return cleanup(fail(ctx, GB_ERR_OOM, "sponge state"));
            }
            if (!ok) return cleanup(fail(ctx, GB_ERR_HIP, "chunked upload of the input columns failed"));
        } else if (!dev_in) {
            if (!copy_columns(cols, ncols, n * 4, scr, hipMemcpyHostToDevice, st))
                return cleanup(fail(ctx, GB_ERR_HIP, "copy of input columns failed"));
            mark_upload(ctx);
            in_dev = scr;
        }
        bool direct_intt = false;
        if (staged) {
        } else if ((flags & GB_INPUT_DEVICE_FORM) || p3) {  // already Montgomery words on the device (prover-internal; a host's p3 words)
            if (hipMemcpyAsync(coeffs, in_dev, in_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return cleanup(fail(ctx, GB_ERR_HIP, "copy of input columns failed"));
            if (p3) gbk::bb_reduce_words(coeffs, ncols * n, st);   // a host's words (the library's own device form is in range)
        } else if (!is_coeffs && log_n >= 16 && log_n <= gbk::NTT_NATIVE_LOG) {
            // canonical values on the device (a resident witness, a small host batch): the inverse transform takes them as they are
            Scope sc(ctx, "IFFT");
            (void)gbk::bb_intt_columns_canonical(const_cast<u32*>(in_dev), coeffs, scr + scr_bytes / 4, ncols, 0, *bt, st);
            direct_intt = true;
        } else {
            gbk::bb_to_mont(in_dev, coeffs, ncols * n, st);
        }
        if (!is_coeffs && !staged && !direct_intt) {
            Scope sc(ctx, "IFFT");
            gbk::bb_intt_columns(coeffs, coeffs, scr + scr_bytes / 4, ncols, *bt, st);
        }
        {
            Scope sc(ctx, "FFT + blinding");
            if (!staged) gbk::bb_lde_columns(coeffs, lde, ncols, *bt, *bc, st);
            if (nsalt) {
                const u32* sdev = static_cast<const u32*>(salts);
                if (!dev_in) {
                    if (hipMemcpyAsync(scr, salts, (size_t)nsalt * N * 4, hipMemcpyHostToDevice, st) != hipSuccess)
                        return cleanup(fail(ctx, GB_ERR_HIP, "copy of salts failed"));
                    mark_upload(ctx);
                    if (p3) { gbk::bb_reduce_words(scr, (size_t)nsalt * N, st); gbk::bb_from_mont(scr, scr, (size_t)nsalt * N, st); }   // the salt columns are F::rand_vec words too
                    sdev = scr;
                }
                gbk::bb_bitrev_copy_to_mont(sdev, lde + ncols * N, log_N, nsalt, st);
            }
        }
        {
            Scope sc(ctx, "build Merkle tree");
            u32* lv = (u32*)b->levels;
            {
                Scope sl(ctx, "hash leaves");
                if (!seg_done && keep_split) {   // device input whose last column segment may have to be redone: hash in two segments
                    if (pool_alloc(ctx, seg_state_bytes, &seg_state) != hipSuccess) { seg_state = nullptr; return cleanup(fail(ctx, GB_ERR_OOM, "sponge state")); }
                    gbk::bb_merkle_leaves_segment(lde, N, 0, last_seg_start, N, (u32*)seg_state, false, (u32)(width - last_seg_start), lv, st);
                    seg_done = last_seg_start;
                }
                if (seg_done) gbk::bb_merkle_leaves_segment(lde, N, seg_done, (u32)width, N, (u32*)seg_state, true, 0, lv, st);
                else gbk::bb_merkle_leaves(lde, N, (u32)width, N, lv, st);
            }
            for (u32 k = 0; k < log_N - cap_height; k++)
                gbk::bb_merkle_level(lv + 8 * level_offset(N, k), lv + 8 * level_offset(N, k + 1), N >> (k + 1), st);
        }
        if (hipGetLastError() != hipSuccess) return cleanup(fail(ctx, GB_ERR_HIP, "kernel launch failed"));
        if (keep && seg_done && seg_state) { *keep = SegKeep{seg_state, seg_state_bytes, seg_done}; seg_state = nullptr; }
        *out = b;
        return GB_OK;
    }
    const gbk::GlNttTables* tabs;
    const gbk::GlCosetTables* cos;
    if ((s = gl_tables_for(ctx, log_n, &tabs))) return cleanup(s);
    if ((s = gl_cosets_for(ctx, log_n, rate_bits, gl::GENERATOR, false, &cos))) return cleanup(s);

    const u64* src = static_cast<const u64*>(cols.base);
    const bool staged = !dev_in && !is_coeffs && log_n >= 12;
    if (staged) {
        // column chunks: H2D into values_dev on the copy stream, then (main stream, behind an event) inverse NTT and LDE of the chunk
        const size_t CH = 16;
        u64* vals = values_dev ? static_cast<u64*>(values_dev) : b->coeffs;  // without a taker the values are transformed in place
        if ((s = ensure(ctx, ctx->scratch, CH * n * sizeof(u64)))) return cleanup(s);
        EventList evs;
        bool ok = true;
        hipEvent_t e0 = evs.make(ok);                               // values_dev may be a pool block still in use on `st`
        ok = ok && hipEventRecord(e0, st) == hipSuccess && hipStreamWaitEvent(ctx->copy_stream, e0, 0) == hipSuccess;
        for (size_t c0 = 0, cc = 0; c0 < ncols && ok; c0 += cc) {
            cc = std::min(first_chunks(c0, CH), ncols - c0);
            hipEvent_t copied = evs.make(ok);
            ok = ok && upload_columns(ctx, cols, pinned, c0, cc, n * sizeof(u64), vals + c0 * n) &&
                 hipEventRecord(copied, ctx->copy_stream) == hipSuccess && hipStreamWaitEvent(st, copied, 0) == hipSuccess;
            if (!ok) break;
            if (p3) gbk::gl_canonicalize(vals + c0 * n, cc * n, st);   // p3-goldilocks keeps any u64 representative in memory
            { Scope sc(ctx, "IFFT"); gbk::gl_intt_columns(vals + c0 * n, b->coeffs + c0 * n, (u64*)ctx->scratch.p, cc, *tabs, st); }
            { Scope sc(ctx, "FFT + blinding"); gbk::gl_lde_columns(b->coeffs + c0 * n, b->lde + c0 * N, cc, *tabs, *cos, st); }
            if (!hash_ready_segments(c0 + cc)) return cleanup(fail(ctx, GB_ERR_OOM, "sponge state"));
        }
        if (!ok) return cleanup(fail(ctx, GB_ERR_HIP, "chunked upload of the input columns failed"));
    } else if (!dev_in || is_coeffs) {
        // host input, or coefficients the batch must own a copy of
        if (!copy_columns(cols, ncols, n * sizeof(u64), b->coeffs, dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st))
            return cleanup(fail(ctx, GB_ERR_HIP, "copy of input columns failed"));
        if (!dev_in) mark_upload(ctx);
        if (p3) gbk::gl_canonicalize(b->coeffs, ncols * n, st);
        src = b->coeffs;
    }
    if (!is_coeffs && !staged) {
        if ((s = ensure(ctx, ctx->scratch, ncols * n * sizeof(u64)))) return cleanup(s);
        Scope sc(ctx, "IFFT");
        gbk::gl_intt_columns(src, b->coeffs, (u64*)ctx->scratch.p, ncols, *tabs, st);
    }
    {
        Scope sc(ctx, "FFT + blinding");
        if (!staged) gbk::gl_lde_columns(b->coeffs, b->lde, ncols, *tabs, *cos, st);
        if (nsalt) {
            // salt columns arrive in LDE-point order (like lde_values' extra columns, oracle.rs:144-148)
            // and are stored, like everything else, in leaf order: leaf j <- point bitrev(j)
            const u64* sdev = static_cast<const u64*>(salts);
            if (!dev_in) {
                if ((s = ensure(ctx, ctx->scratch, nsalt * N * sizeof(u64)))) return cleanup(s);
                if (hipMemcpyAsync(ctx->scratch.p, salts, nsalt * N * sizeof(u64), hipMemcpyHostToDevice, st) != hipSuccess)
                    return cleanup(fail(ctx, GB_ERR_HIP, "copy of salts failed"));
                mark_upload(ctx);
                if (p3) gbk::gl_canonicalize((u64*)ctx->scratch.p, (size_t)nsalt * N, st);
                sdev = (const u64*)ctx->scratch.p;
            }
            gbk::u64_bitrev_copy(sdev, b->lde + ncols * N, log_N, nsalt, st);
        }
    }
    {
        Scope sc(ctx, "build Merkle tree");
        {
            Scope sl(ctx, "hash leaves");
            if (!seg_done && keep_split) {   // device input whose last column segment may have to be redone: hash in two segments
                if (pool_alloc(ctx, seg_state_bytes, &seg_state) != hipSuccess) { seg_state = nullptr; return cleanup(fail(ctx, GB_ERR_OOM, "sponge state")); }
                gbk::gl_merkle_leaves_segment(b->lde, N, 0, last_seg_start, N, (u64*)seg_state, false, (u32)(width - last_seg_start), b->levels, st);
                seg_done = last_seg_start;
            }
            if (seg_done) gbk::gl_merkle_leaves_segment(b->lde, N, seg_done, (u32)width, N, (u64*)seg_state, true, 0, b->levels, st);
            else gbk::gl_merkle_leaves(b->lde, N, (u32)width, N, b->levels, st);
        }
        for (u32 k = 0; k < log_N - cap_height; k++)
            gbk::gl_merkle_level(b->levels + 4 * level_offset(N, k), b->levels + 4 * level_offset(N, k + 1), N >> (k + 1), st);
    }
    if (hipGetLastError() != hipSuccess) return cleanup(fail(ctx, GB_ERR_HIP, "kernel launch failed"));
    if (keep && seg_done && seg_state) { *keep = SegKeep{seg_state, seg_state_bytes, seg_done}; seg_state = nullptr; }
    *out = b;
    return GB_OK;
}

}  // namespace

extern "C" {

gb_status gb_ctx_create(int device, gb_ctx** out) try {
    if (!out) return fail(nullptr, GB_ERR_INVALID, "null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(nullptr, GB_ERR_HIP, "no HIP device available");
    if (device < 0 || device >= count) return fail(nullptr, GB_ERR_INVALID, "device index out of range");
    gb_ctx* ctx = new (std::nothrow) gb_ctx();
    if (!ctx) return fail(nullptr, GB_ERR_OOM, "host allocation failed");
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, GB_ERR_HIP, "hipSetDevice/hipStreamCreate failed");
    }
    *out = ctx;
    return GB_OK;
} GB_CATCH(nullptr)

gb_status gb_ctx_destroy(gb_ctx* ctx) try {
    if (!ctx) return GB_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    gb_ctx_scope_reset(ctx);
    for (auto& kv : ctx->gl_tables) for (void* p : kv.second.owned) hipFree(p);
    for (auto& kv : ctx->gl_cosets) for (void* p : kv.second.owned) hipFree(p);
    for (auto& kv : ctx->bb_tables) for (void* p : kv.second.owned) hipFree(p);
    for (auto& kv : ctx->bb_cosets) for (void* p : kv.second.owned) hipFree(p);
    if (ctx->bb_tw4096_fwd) hipFree(ctx->bb_tw4096_fwd);
    if (ctx->bb_tw4096_inv) hipFree(ctx->bb_tw4096_inv);
    if (ctx->tw4096_fwd) hipFree(ctx->tw4096_fwd);
    if (ctx->tw4096_inv) hipFree(ctx->tw4096_inv);
    if (ctx->tw4096_fwd_m) hipFree(ctx->tw4096_fwd_m);
    if (ctx->tw4096_inv_m) hipFree(ctx->tw4096_inv_m);
    if (ctx->scratch.p) hipFree(ctx->scratch.p);
    if (ctx->small.p) hipFree(ctx->small.p);
    for (DeviceBuf& wb : ctx->big_work) if (wb.p) hipFree(wb.p);
    if (ctx->upload_mark) hipEventDestroy(ctx->upload_mark);
    for (auto& kv : ctx->pool) hipFree(kv.second);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    delete ctx->stager;   // joins the copy threads, frees the page-locked ring
    if (ctx->xfer_up) (void)hipHostFree(ctx->xfer_up);
    if (ctx->xfer_down) (void)hipHostFree(ctx->xfer_down);
    hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    return GB_OK;
} GB_CATCH(nullptr)   // (the object may be gone: the message goes to the thread's own slot)

const char* gb_last_error(const gb_ctx* ctx) try {
    return ctx ? ctx->err.c_str() : g_null_ctx_error.c_str();
} catch (...) {
    return "";
}

gb_status gb_ctx_synchronize(gb_ctx* ctx) try {
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_ctx_trim(gb_ctx* ctx) try {
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drop_all_retry_state(ctx);
    for (auto& kv : ctx->pool) (void)hipFree(kv.second);
    ctx->pool.clear();
    for (DeviceBuf& wb : ctx->big_work) {   // the coset tables hold these fields' addresses; the next lookup grows them again
        if (wb.p) (void)hipFree(wb.p);
        wb.p = nullptr; wb.bytes = 0;
    }
    if (ctx->stager) {   // the page-locked staging ring and its copy threads come back with the next pageable input
        HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
        delete ctx->stager;
        ctx->stager = nullptr;
    }
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_ctx_stream(gb_ctx* ctx, void** stream_out) try {
    if (!ctx || !stream_out) return fail(ctx, GB_ERR_INVALID, "null argument");
    *stream_out = (void*)ctx->stream;
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_ctx_set_profiling(gb_ctx* ctx, int32_t on) try {
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    ctx->profiling = on != 0;
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_ctx_scope_ms(gb_ctx* ctx, const char* scope, double* ms_out, uint64_t* count_out) try {
    if (!ctx || !scope || !ms_out) return fail(ctx, GB_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0;
    uint64_t cnt = 0;
    auto it = ctx->scopes.find(scope);
    if (it != ctx->scopes.end()) {
        for (auto& sp : it->second.spans) {
            float ms = 0;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, sp.first, sp.second));
            total += ms;
            cnt++;
        }
    }
    *ms_out = total;
    if (count_out) *count_out = cnt;
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_ctx_scope_reset(gb_ctx* ctx) try {
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->scopes)
        for (auto& sp : kv.second.spans) {
            hipEventDestroy(sp.first);
            hipEventDestroy(sp.second);
        }
    ctx->scopes.clear();
    return GB_OK;
} GB_CATCH(ctx)

static gb_status commit_entry(gb_ctx* ctx, uint32_t field, ColSrc cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, const void* salts, uint32_t flags, bool is_coeffs, gb_batch** out) {
    if (ctx && (flags & ~GB_PUBLIC_INPUT_FLAGS)) {
        if (out) *out = nullptr;
        return fail(ctx, GB_ERR_INVALID, "unknown bits in flags");
    }
    gb_status s = commit(ctx, field, cols, ncols, log_n, rate_bits, cap_height, salts, flags, is_coeffs, out);
    if (!(flags & GB_INPUT_DEVICE)) s = finish_host_commit(ctx, s, out);   // `cols` / `salts` are the caller's again on return
    return s;
}

gb_status gb_commit_values(gb_ctx* ctx, uint32_t field, const void* cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                           uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out) try {
    return commit_entry(ctx, field, cols, ncols, log_n, rate_bits, cap_height, salts, flags, false, out);
} GB_CATCH(ctx)

gb_status gb_commit_coeffs(gb_ctx* ctx, uint32_t field, const void* cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                           uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out) try {
    return commit_entry(ctx, field, cols, ncols, log_n, rate_bits, cap_height, salts, flags, true, out);
} GB_CATCH(ctx)

// Vec<PolynomialValues<F>> / Vec<PolynomialCoeffs<F>> as the reference holds them: ncols separately allocated columns
gb_status gb_commit_values_cols(gb_ctx* ctx, uint32_t field, const void* const* cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                                uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out) try {
    return commit_entry(ctx, field, ColSrc::columns(cols), ncols, log_n, rate_bits, cap_height, salts, flags, false, out);
} GB_CATCH(ctx)

gb_status gb_commit_coeffs_cols(gb_ctx* ctx, uint32_t field, const void* const* cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                                uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out) try {
    return commit_entry(ctx, field, ColSrc::columns(cols), ncols, log_n, rate_bits, cap_height, salts, flags, true, out);
} GB_CATCH(ctx)

// ---- page-locked host memory for a host that builds its columns where the copy engine can read them
gb_status gb_host_alloc(gb_ctx* ctx, size_t bytes, void** out) try {
    if (!ctx || !out) return fail(ctx, GB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (!bytes) return fail(ctx, GB_ERR_INVALID, "zero-sized allocation");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GB_ERR_OOM, "hipHostMalloc failed (page-locked memory is limited by RLIMIT_MEMLOCK and physical memory)");
    }
    *out = p;
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_host_free(gb_ctx* ctx, void* p) try {
    if (!ctx) return fail(nullptr, GB_ERR_INVALID, "null ctx");
    if (!p) return GB_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostFree(p));
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_host_register(gb_ctx* ctx, void* p, size_t bytes) try {
    if (!ctx || !p || !bytes) return fail(ctx, GB_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GB_ERR_HIP, "hipHostRegister failed (already registered, or RLIMIT_MEMLOCK)");
    }
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_host_unregister(gb_ctx* ctx, void* p) try {
    if (!ctx || !p) return fail(ctx, GB_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));   // no upload of this context is still reading it
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GB_ERR_INVALID, "hipHostUnregister failed (not a registered range)");
    }
    return GB_OK;
} GB_CATCH(ctx)

// Tuning and debugging switches (none changes a result), all of them per context.
gb_status gb_ctx_set_option(gb_ctx* ctx, const char* key, int64_t value) try {
    if (!ctx || !key) return fail(ctx, GB_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "copy_threads") {
        if (value < -1 || value > 64) return fail(ctx, GB_ERR_INVALID, "copy_threads must be -1 (no staging ring), 0 (calling thread) .. 64");
        if (ctx->stager && (int)value != ctx->copy_threads) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
            delete ctx->stager;
            ctx->stager = nullptr;
        }
        ctx->copy_threads = (int)value;
    } else if (k == "retry_verify") {
        ctx->retry_verify = value != 0;
    } else {
        return fail(ctx, GB_ERR_INVALID, "unknown option: " + k);
    }
    return GB_OK;
} GB_CATCH(ctx)

gb_status gb_batch_free(gb_batch* b) try {
    if (!b) return GB_OK;
    if (b->ctx) {
        pool_free(b->ctx, b->coeffs, b->coeffs_bytes);
        pool_free(b->ctx, b->lde, b->lde_bytes);
        pool_free(b->ctx, b->levels, b->levels_bytes);
    }
    delete b;
    return GB_OK;
} GB_CATCH(nullptr)   // (the object may be gone: the message goes to the thread's own slot)

gb_status gb_batch_info(const gb_batch* b, uint32_t* field, size_t* ncols, uint32_t* degree_log, uint32_t* rate_bits,
                        uint32_t* cap_height, uint32_t* blinding) try {
    if (!b) return fail(nullptr, GB_ERR_INVALID, "null batch");
    if (field) *field = b->field;
    if (ncols) *ncols = b->ncols;
    if (degree_log) *degree_log = b->log_n;
    if (rate_bits) *rate_bits = b->rate_bits;
    if (cap_height) *cap_height = b->cap_height;
    if (blinding) *blinding = b->nsalt ? 1 : 0;
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_cap(gb_batch* b, void* out) try {
    if (!b || !out) return fail(b ? b->ctx : nullptr, GB_ERR_INVALID, "null argument");
    gb_ctx* ctx = b->ctx;
    const u64 N = (u64)1 << (b->log_n + b->rate_bits);
    const u64* cap = b->levels + 4 * level_offset(N, b->log_n + b->rate_bits - b->cap_height);  // 32 B per digest
    return read_back(ctx, out, cap, ((size_t)1 << b->cap_height) * 32);
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_coeffs(gb_batch* b, size_t col, void* out) try {
    if (!b || !out) return fail(b ? b->ctx : nullptr, GB_ERR_INVALID, "null argument");
    gb_ctx* ctx = b->ctx;
    if (col >= b->ncols) return fail(ctx, GB_ERR_INVALID, "polynomial index out of range");
    const size_t n = (size_t)1 << b->log_n;
    if (b->field == GB_BABYBEAR) {
        gb_status s = ensure(ctx, ctx->scratch, n * 4);
        if (s) return s;
        gbk::bb_from_mont((const u32*)b->coeffs + col * n, (u32*)ctx->scratch.p, n, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GB_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out, b->coeffs + col * n, n * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

static gb_status read_row(gb_batch* b, u64 leaf, u32 width, void* out) {
    gb_ctx* ctx = b->ctx;
    const u64 N = (u64)1 << (b->log_n + b->rate_bits);
    gb_status s = ensure(ctx, ctx->small, 64 * 1024);
    if (s) return s;
    if (width * sizeof(u64) > ctx->small.bytes) return fail(ctx, GB_ERR_UNSUPPORTED, "row too wide");
    if (b->field == GB_BABYBEAR)
        gbk::bb_gather_row((const u32*)b->lde, N, width, leaf, (u32*)ctx->small.p, ctx->stream);
    else
        gbk::gl_gather_row(b->lde, N, width, leaf, (u64*)ctx->small.p, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->small.p, width * esize(b->field), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
}

gb_status gb_batch_lde_values(gb_batch* b, uint64_t index, uint64_t step, void* out) try {
    if (!b || !out) return fail(b ? b->ctx : nullptr, GB_ERR_INVALID, "null argument");
    const u32 bits = b->log_n + b->rate_bits;
    const u64 N = (u64)1 << bits;
    u64 i = index * step;
    if (i >= N) return fail(b->ctx, GB_ERR_INVALID, "index * step out of range (oracle.rs:154-156 would index out of bounds)");
    u64 leaf = 0;
    for (u32 k = 0; k < bits; k++) leaf |= ((i >> k) & 1ull) << (bits - 1 - k);
    return read_row(b, leaf, (u32)b->ncols, out);
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_leaf(gb_batch* b, uint64_t leaf_index, void* row, void* siblings, uint32_t* nsib) try {
    if (!b) return fail(nullptr, GB_ERR_INVALID, "null batch");
    gb_ctx* ctx = b->ctx;
    const u32 bits = b->log_n + b->rate_bits;
    if (leaf_index >> bits) return fail(ctx, GB_ERR_INVALID, "leaf_index out of range (merkle_tree.rs:191)");
    if (row) {
        gb_status s = read_row(b, leaf_index, (u32)(b->ncols + b->nsalt), row);
        if (s) return s;
    }
    const u32 layers = bits - b->cap_height;
    if (nsib) *nsib = layers;
    if (siblings && layers) {
        gb_status s = ensure(ctx, ctx->small, 64 * 1024);
        if (s) return s;
        gbk::gl_gather_siblings(b->levels, bits, b->cap_height, leaf_index, (u64*)ctx->small.p, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(siblings, ctx->small.p, (size_t)layers * 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_digests(gb_batch* b, void* out) try {
    if (!b || !out) return fail(b ? b->ctx : nullptr, GB_ERR_INVALID, "null argument");
    gb_ctx* ctx = b->ctx;
    const u32 bits = b->log_n + b->rate_bits;
    const u64 N = (u64)1 << bits;
    const size_t total = 2 * (N - ((u64)1 << b->cap_height));
    if (!total) return GB_OK;
    const size_t bytes = total * 32;
    gb_status s = ensure(ctx, ctx->scratch, bytes);
    if (s) return s;
    gbk::gl_digests_to_reference_layout(b->levels, (u64*)ctx->scratch.p, bits, b->cap_height, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_leaves(gb_batch* b, void* out) try {
    if (!b || !out) return fail(b ? b->ctx : nullptr, GB_ERR_INVALID, "null argument");
    gb_ctx* ctx = b->ctx;
    const u64 N = (u64)1 << (b->log_n + b->rate_bits);
    const u32 width = (u32)(b->ncols + b->nsalt);
    const size_t bytes = (size_t)N * width * esize(b->field);
    gb_status s = ensure(ctx, ctx->scratch, bytes);
    if (s) return s;
    if (b->field == GB_BABYBEAR)
        gbk::bb_transpose_to_rows((const u32*)b->lde, N, width, N, (u32*)ctx->scratch.p, ctx->stream);
    else
        gbk::u64_transpose_to_rows(b->lde, N, width, N, (u64*)ctx->scratch.p, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_batch_device_ptrs(gb_batch* b, void** coeffs, void** lde, void** digest_levels) try {
    if (!b) return fail(nullptr, GB_ERR_INVALID, "null batch");
    if (coeffs) *coeffs = b->coeffs;
    if (lde) *lde = b->lde;
    if (digest_levels) *digest_levels = b->levels;
    return GB_OK;
} GB_CATCH(b ? b->ctx : nullptr)

gb_status gb_permute(gb_ctx* ctx, uint32_t field, const void* in, void* out, uint64_t count) try {
    if (!ctx || !in || !out) return fail(ctx, GB_ERR_INVALID, "null argument");
    if (field != GB_GOLDILOCKS && field != GB_BABYBEAR) return fail(ctx, GB_ERR_INVALID, "unknown field tag");
    if (!count) return GB_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (field == GB_BABYBEAR) {
        const size_t bytes = count * 16 * sizeof(u32);
        gb_status s = ensure(ctx, ctx->scratch, 2 * bytes);
        if (s) return s;
        u32* din = (u32*)ctx->scratch.p;
        u32* dout = din + count * 16;
        HIP_TRY(ctx, hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream));
        gbk::bb_poseidon2_permute(din, dout, count, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GB_OK;
    }
    const size_t bytes = count * 12 * sizeof(u64);
    gb_status s = ensure(ctx, ctx->scratch, 2 * bytes);
    if (s) return s;
    u64* din = (u64*)ctx->scratch.p;
    u64* dout = din + count * 12;
    HIP_TRY(ctx, hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    gbk::gl_poseidon_permute(din, dout, count, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GB_OK;
} GB_CATCH(ctx)

}  // extern "C"

#include "prover_host.inc"
#include "verifier_host.inc"
#include "compress_host.inc"
