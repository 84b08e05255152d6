import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Strict-bound table (tests/util.py STRICT_LOG): per parity assertion, how many elements lie outside
    SURVEY.md §8c's per-element |a-b| <= 1e-5*max(|a|,|b|,eps) — printed and written to gpurun_out/."""
    from tests import util
    log = util.STRICT_LOG
    if util.GUARD_TRIPS:
        terminalreporter.write_line("strict guard trips (test | tensor | n | outside | fp32 oracle outside | limit):")
        for t in util.GUARD_TRIPS:
            terminalreporter.write_line("GUARD " + t)
    if not log:
        return
    n_assert = len(log)
    n_clean = sum(1 for r in log if r["strict_fail"] == 0)
    tot = sum(r["n"] for r in log)
    bad = sum(r["strict_fail"] for r in log)
    lines = [f"# strict 1e-5*max(|a|,|b|,{util.STRICT_EPS:g}) accounting: {n_assert} parity assertions, "
             f"{n_clean} with zero elements outside the strict bound; {bad} of {tot} elements outside in total", "",
             "| test | tensor | elements | outside strict | worst err/tol | fp32-oracle outside strict |", "|---|---|---:|---:|---:|---:|"]
    for r in log:
        if r["strict_fail"] or r["ref32_strict_fail"]:
            lines.append(f"| {r['test']} | {r['what']} | {r['n']} | {r['strict_fail']} | {r['worst']:.3g} | "
                         f"{'' if r['ref32_strict_fail'] is None else r['ref32_strict_fail']} |")
    terminalreporter.write_line(lines[0])
    for l in lines[2:42]:
        terminalreporter.write_line(l)
    if len(lines) > 42:
        terminalreporter.write_line(f"... {len(lines) - 42} more rows in gpurun_out/strict_parity.md")
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "strict_parity.md"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
