"""CPU: the documents' generated parts are what their generators produce from the committed files -- the ctypes binding
block of INTEGRATION.md (tools/gen_bindings.py; also checked in test_abi_cpu.py) and every round-4 table of
profiles/README.md (tools/profile_tables.py: no cell of those tables is typed by hand)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_profiles_readme_tables_are_the_generated_ones():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_tables.py"), "r04"], capture_output=True, text=True,
                         check=True).stdout
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    parts = [p for p in out.split("## ") if p.strip()]
    assert {p.split("\n", 1)[0].strip() for p in parts} >= {"bench", "kernels", "points", "stats", "profiled"}
    for p in parts:
        name, body = p.split("\n", 1)
        body = body.strip()
        if not body:
            continue
        a, b = f"<!-- tables:r04:{name.strip()} -->", f"<!-- /tables:r04:{name.strip()} -->"
        if a not in readme:
            continue                                  # (a table this round's README does not carry: gather)
        have = readme[readme.index(a) + len(a): readme.index(b)].strip()
        assert have == body, f"profiles/README.md table `{name.strip()}` is stale: run tools/profile_tables.py r04 --write"


def test_latest_traffic_belongs_to_the_committed_kernel_sources():
    """bench.py attaches roofline.traffic only while profiles/latest_traffic.json carries the digest of vae_amd/csrc/: the
    committed PMC passes are of the committed kernels."""
    import json
    sys.path.insert(0, ROOT)
    from vae_amd.build import sources_digest
    import pytest
    t = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
    assert t["fwd"]["hbm_bytes_per_launch"] > 0 and t["bwd_adam"]["hbm_bytes_per_launch"] > 0
    if t["_csrc_sha1"] != sources_digest():       # (kernels edited since the last tools/profile_round.sh: bench.py then reports traffic null)
        pytest.skip("profiles/latest_traffic.json is of older kernel sources: re-run tools/profile_round.sh + collect_profiles.py")
