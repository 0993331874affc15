"""bench.py quotes two figures from profiles/ (HBM traffic of the NTT passes, VALU instructions per permutation): each summary
carries the hash of the library sources it was measured on, and the line says "stale" when the tree has moved on."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_quoted_profiles_carry_a_source_hash_and_a_stale_flag():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_sha16
    b = _bench()
    now = csrc_sha16()
    assert len(now) == 16 and now == csrc_sha16()
    for pattern in ("r*_poseidon_valu_goldilocks.json", "r*_poseidon_valu_babybear.json", "r*_ntt_traffic_pmc_goldilocks.json",
                    "r*_ntt_traffic_pmc_babybear.json"):
        j = b._latest_profile(pattern)
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[-1]
        assert j is not None and j["profile_file"] == os.path.relpath(newest, ROOT), pattern     # the newest round's file is the one quoted
        raw = json.load(open(os.path.join(ROOT, j["profile_file"])))
        assert "csrc_sha16" in raw, "%s was summarised without the hash of the sources it was measured on" % j["profile_file"]
        assert j["stale"] == (raw["csrc_sha16"] != now)
    assert b._latest_profile("r*_no_such_summary.json") is None


def test_a_source_change_makes_the_quote_stale(tmp_path, monkeypatch):
    b = _bench()
    j = b._latest_profile("r*_poseidon_valu_goldilocks.json")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import csrc_hash
    monkeypatch.setattr(csrc_hash, "csrc_sha16", lambda: "0" * 16)
    assert b._latest_profile("r*_poseidon_valu_goldilocks.json")["stale"] is True
    assert j["stale"] in (True, False)
