"""Record the declaration hash of include/recalgo.h for its CURRENT RECALGO_ABI_VERSION in include/recalgo.abi.
Run after bumping the version for a signature change (tests/test_abi.py compares).  Re-recording an existing version is
only legitimate while that version has not left the development tree (no library of it exists anywhere else)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_abi import HEADER, declaration_hash  # noqa: E402

version = int(re.search(r"#define RECALGO_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
path = os.path.join(ROOT, "include", "recalgo.abi")
lines = [ln for ln in open(path).read().splitlines() if ln.strip()]
lines = [ln for ln in lines if ln.startswith("#") or int(ln.split()[0]) != version]
lines.append(f"{version} {declaration_hash()}")
open(path, "w").write("\n".join(lines) + "\n")
print(lines[-1])
