import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- the plain 1e-5 bar, one table per session (tests/helpers.py: plain_bar / log_plain_bar) ------------------------------------------
def pytest_sessionstart(session):
    from helpers import PLAIN_LOG
    try:
        os.remove(PLAIN_LOG)
    except OSError:
        pass


def plain_bar_table(path):
    """The records parity tests filed during this session, summed per (fixture family, test, reference): the share of rows whose
    modulated velocity meets north_star's plain 1e-5 bar, and the rows that miss it (nothing else is admitted)."""
    import json
    agg = {}
    with open(path) as f:
        for line in f:
            r = json.loads(line)
            a = agg.setdefault((r["family"], r["what"], r["against"]), dict(rows=0, plain=0, envelope=0, mask=0, worst=0.0))
            for k in ("rows", "plain", "envelope", "mask"):
                a[k] += r[k]
            a["worst"] = max(a["worst"], r["worst"])
    out = ["modulated velocity |u - u_ref| <= 1e-5 x max|u_ref|, every row, nothing else admitted (helpers.plain_bar)",
           f"{'fixture family':14s} {'test':40s} {'against':10s} {'rows':>8s} {'plain 1e-5':>11s} {'rows missed':>12s} {'worst row':>10s}"]
    for (fam, what, ag), a in sorted(agg.items()):
        n = max(a["rows"], 1)
        out.append(f"{fam:14s} {what:40s} {ag:10s} {a['rows']:8d} {100.0 * a['plain'] / n:10.3f}% {a['rows'] - a['plain']:12d} {a['worst']:10.2e}")
    return "\n".join(out)


def pytest_terminal_summary(terminalreporter):
    from helpers import PLAIN_LOG
    if os.path.exists(PLAIN_LOG):
        table = plain_bar_table(PLAIN_LOG)
        terminalreporter.write_line("")
        terminalreporter.write_line(table)
        try:
            with open(os.path.join(os.path.dirname(PLAIN_LOG), "parity_plain_bar.txt"), "w") as f:
                f.write(table + "\n")
        except OSError:
            pass
