"""Housekeeping with teeth: every INET_* name that appears in the package, the header, bench.py or the driver entry
must be listed in DESIGN.md section 4's switch table, and there may not be more than 40 of them (round 5 had 74)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = re.compile(r"\bINET_[A-Z][A-Z0-9_]*[A-Z0-9]\b")


def _names_in(paths):
    found = {}
    for path in paths:
        with open(path, errors="replace") as f:
            for n in NAME.findall(f.read()):
                found.setdefault(n, os.path.relpath(path, ROOT))
    return found


def _sources():
    out = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for base, exts in (("inpaintnet_amd", (".py", ".hip", ".h")), ("include", (".h",))):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            out += [os.path.join(d, f) for f in files if f.endswith(exts)]
    return out


def _design_table():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    start = text.index("**Switches.**")
    end = text.index("## 5.", start)
    return set(NAME.findall(text[start:end]))


def test_every_switch_is_documented_and_there_are_few():
    used = _names_in(_sources())
    listed = _design_table()
    missing = {n: where for n, where in used.items() if n not in listed}
    assert not missing, f"INET_* names used in the sources but absent from DESIGN.md section 4: {missing}"
    assert len(used) <= 40, sorted(used)


def test_the_table_lists_nothing_that_is_gone():
    used = _names_in(_sources())
    stale = sorted(n for n in _design_table() if n not in used)
    assert not stale, f"DESIGN.md section 4 lists switches no source reads any more: {stale}"
