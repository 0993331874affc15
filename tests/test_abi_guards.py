""""Never unwinds" by construction (include/goldibear_gpu.h; SURVEY.md 8(b): `extern "C"`, status codes, never unwind - a Rust
caller's panic never crosses extern "C" either): every function the two headers declare is DEFINED in csrc/ as a function-try-block
whose handler is the library's catch-all (GB_CATCH / GB_CATCH_CIRCUIT -> unwound(): std::bad_alloc -> GB_ERR_OOM, other
std::exception -> GB_ERR_INVALID, anything else -> GB_ERR_HIP).  This test parses the sources and fails on a definition that is not;
tests/test_sanitized_parsers.py drives the host-only entry points through an operator new that fails on the N-th call.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc")


def _declared():
    names = set()
    for h in ("goldibear_gpu.h", "goldibear_gpu_test_hooks.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        names |= set(re.findall(r"\b(gb_[a-z0-9_]+)\s*\(", src))
    return names


def _definitions():
    """name -> (file, text from the definition's first line to its closing line at column 0)"""
    defs = {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".inc")):
            continue
        lines = open(os.path.join(CSRC, f)).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r"^(?:gb_status|const char\*) (gb_\w+)\(", ln)
            if not m or ln.rstrip().endswith(";"):
                continue
            j = i
            while not lines[j].startswith("}"):
                j += 1
            end = j
            while not lines[end].startswith("}") or "catch" in lines[end] and lines[end].rstrip().endswith("{"):
                end += 1   # (a spelled-out handler: up to its own closing brace)
            assert m.group(1) not in defs, "two definitions of " + m.group(1)
            defs[m.group(1)] = (f, "\n".join(lines[i:end + 1]))
    return defs


def test_every_exported_definition_is_a_guarded_function_try_block():
    declared, defs = _declared(), _definitions()
    assert len(declared) >= 55
    assert declared <= set(defs), "declared but not defined at column 0 of csrc/: %s" % sorted(declared - set(defs))
    assert set(defs) <= declared, "extern \"C\"-style definitions the headers do not declare: %s" % sorted(set(defs) - declared)
    for name, (f, text) in sorted(defs.items()):
        sig_end = re.search(r"\)\s*try\s*\{\s*$", text, flags=re.M)
        assert sig_end, "%s (%s) is not a function-try-block" % (name, f)
        last = text.split("\n")[-1]
        if name == "gb_last_error":      # returns a string, not a status: its own catch-all
            assert "catch (...)" in text and 'return "";' in text
            continue
        assert re.match(r"^\} GB_CATCH(_CIRCUIT)?\(.*\)", last), "%s (%s) does not end in the library's catch-all: %r" % (name, f, last)


def test_the_catch_all_maps_exceptions_to_status_codes_and_cannot_throw():
    api = open(os.path.join(CSRC, "api.hip")).read()
    body = api[api.index("gb_status unwound(gb_ctx* ctx, const char* fn) noexcept {"):api.index("#define GB_CATCH(CTX)")]
    assert "std::bad_alloc" in body and "GB_ERR_OOM" in body and "GB_ERR_INVALID" in body and "GB_ERR_HIP" in body
    assert body.count("catch (...)") >= 2      # the unknown exception, and a failure to write the message
    assert "catch (...) { return unwound((CTX), __func__); }" in api
    inc = open(os.path.join(CSRC, "prover_host.inc")).read()
    assert "catch (...) { return unwound_circuit((C), __func__); }" in inc and "unwound_circuit(gb_circuit* c, const char* fn) noexcept" in inc
    # blocks go back to the pool from destructors: that path must not throw
    assert "void pool_free(gb_ctx* ctx, void* p, size_t bytes) noexcept" in api
