"""The committed records reproduce from the committed rocprofv3 summaries (scripts/verify_records.py; no GPU):
every fraction a record states is F x units / the CSV's average duration / peak."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_round6_records_reproduce_from_the_committed_csvs():
    import verify_records as vr
    if not os.path.exists(os.path.join(ROOT, "profiles", "r06_bench_n1.json")):
        pytest.skip("no round-6 records committed yet")
    rep, _ = vr.verify("r06")
    bad = [(n, d) for n, ok, d in rep.items if not ok]
    assert not bad, bad
    assert len(rep.items) >= 10
