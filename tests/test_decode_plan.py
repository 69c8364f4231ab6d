"""The launch planner of the register-resident decode (csrc/decode_b1.hip make_plan / place_role), checked on the host for every call
size and vocabulary class: which workgroup id plays which role is pure arithmetic, and a mistake in it shows up on a GPU as a hang
until the bounded spins run out -- here it shows up as ok == 0.  (Plans per size: the comment above decode_b1_team_rows.)"""
import ctypes as C

import pytest

from inpaintnet_amd import _lib


@pytest.fixture(scope="module")
def L():
    _lib.build(verbose=False)
    return _lib.lib()


def plan(L, B, V, Z=256):
    out = (C.c_int * 8)()
    rc = L.inet_decode_b1_plan(B, V, Z, out)
    keys = ("teams", "team_rows", "rgroups", "crit", "placed", "grid", "live", "ok")
    return rc, dict(zip(keys, list(out)))


@pytest.mark.parametrize("V", [20, 32, 48, 64, 80, 100, 128])
@pytest.mark.parametrize("B", list(range(1, 17)))
def test_every_role_once_critical_roles_on_one_residue(L, B, V):
    rc, p = plan(L, B, V)
    assert rc == 0, (B, V)
    assert p["ok"] == 1, (B, V, p)
    assert p["placed"] == 1 and p["grid"] <= 256 and p["live"] <= 256, (B, V, p)
    assert p["teams"] * p["team_rows"] >= B


def test_the_plans_per_call_size(L):
    """The table in csrc/decode_b1.hip, at V = 48 (merged build for one-row teams) and V = 100 (not merged)."""
    for V, merged in ((48, True), (100, False)):
        got = {B: plan(L, B, V)[1] for B in range(1, 17)}
        assert got[1]["teams"] == 1 and got[1]["live"] == (128 if merged else 129)          # 16 CB + 16 TA + 16 TBh + 80 | + C
        assert got[1]["crit"] == (32 if merged else 17)
        for B in (2, 3):                                    # whole one-row teams + the beat path
            assert (got[B]["teams"], got[B]["team_rows"], got[B]["rgroups"]) == (B, 1, 0), (V, B, got[B])
            assert got[B]["live"] == B * (48 if merged else 49) + 80
        assert got[4]["team_rows"] == 1 and got[4]["rgroups"] == 2                              # four measures: shared groups
        assert got[4]["crit"] == (32 if merged else 17)                                         # merged: CB + TA per team, TBh pairs
        assert got[4]["live"] == (4 * 32 + 2 * 16 + 80 if merged else 4 * 17 + 2 * 32 + 80)
        for B in (5, 6):                                    # one-row critical teams + groups of three rows + the beat path
            assert (got[B]["team_rows"], got[B]["rgroups"], got[B]["crit"]) == (1, 2, 16 if merged else 17), (V, B, got[B])
        for B in (7, 8, 9, 10):                             # whole two-row teams
            assert (got[B]["team_rows"], got[B]["rgroups"]) == (2, 0) and got[B]["live"] == got[B]["teams"] * 49
        for B in range(11, 17):                             # two-row critical teams + groups of six rows
            assert got[B]["team_rows"] == 2 and got[B]["rgroups"] == (2 * ((B + 1) // 2) + 5) // 6 and got[B]["crit"] == 17
            assert got[B]["live"] == got[B]["teams"] * 17 + got[B]["rgroups"] * 32


def test_calls_the_launch_does_not_take(L):
    assert plan(L, 17, 48)[0] == -1 and plan(L, 0, 48)[0] == -1 and plan(L, 4, 129)[0] == -1
