"""tools/pmc_slots.py: the counter budget of one rocprofv3 --pmc pass (profiles/r03_pmc_counter_budget.txt)."""
import os
import subprocess
import sys

from conftest import ROOT

TOOL = os.path.join(ROOT, "tools", "pmc_slots.py")


def _run(*args):
    return subprocess.run([sys.executable, TOOL, *args], capture_output=True, text=True)


def test_two_tcc_derived_counters_are_refused_and_split():
    res = _run("FETCH_SIZE", "WRITE_SIZE")
    assert res.returncode == 2 and "5 of 4" in res.stderr
    res = _run("--split", "FETCH_SIZE", "WRITE_SIZE")
    assert res.returncode == 0 and res.stdout.split("\n")[:2] == ["FETCH_SIZE", "WRITE_SIZE"]


def test_passes_of_the_profile_scripts_fit():
    for script in ("profile.sh", "pmc_extra.sh", "profile_dense.sh"):
        text = open(os.path.join(ROOT, "tools", script)).read()
        loop = [l for l in text.splitlines() if l.startswith("for PASS in")]
        assert loop, script
        for line in loop:
            for quoted in line.split('"')[1::2]:
                if quoted.strip() and "$" not in quoted:
                    assert _run(*quoted.split()).returncode == 0, (script, quoted)
        assert "pmc_slots.py" in text


def test_nine_sq_counters_do_not_fit():
    assert _run(*[f"SQ_X{i}" for i in range(9)]).returncode == 2
    assert _run(*[f"SQ_X{i}" for i in range(8)], "TCC_HIT_sum", "TCC_MISS_sum", "WRITE_SIZE").returncode == 0
