"""tools/compare_reference_dump.py — what a maintainer with a Julia install runs to pin the oracle against the real package
(per-segment records and, separately, SURVEY §9 items 1–4).  Julia is not available here: the test feeds the tool a dump in
the same CSV formats written from the oracle itself, so at least every file is parsed and every comparison is exercised."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compare_reference_dump_self_test(orc):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_reference_dump.py"), "--self-test"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    for key in ("element ids equal: True", "§9.1", "§9.2", "§9.3", "§9.4", "ALL PINNED"):
        assert key in r.stdout, r.stdout
