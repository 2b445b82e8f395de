"""DESIGN.md is the live design: short enough to be read as a brief (<= 400 lines of <= 120 columns); the history lives in EXPERIMENTS.md."""
import os
import re

from helpers import ROOT


def test_design_md_stays_a_brief():
    lines = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().split("\n")
    assert len(lines) <= 400, len(lines)
    long = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not long, long[:5]


def test_docs_point_at_files_that_exist():
    """Every profiles/ and tools/ path DESIGN.md, README.md and EXPERIMENTS.md's round-5 part name exists in the tree."""
    missing = []
    for doc in ("DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for m in re.finditer(r"`((?:profiles|tools|tests|examples|include|oracle)/[A-Za-z0-9_./-]+\.(?:txt|json|py|sh|hip|h|c|md|npz))`", text):
            if not os.path.exists(os.path.join(ROOT, m.group(1))):
                missing.append((doc, m.group(1)))
    assert not missing, missing
