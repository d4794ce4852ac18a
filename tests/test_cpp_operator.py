"""The C++ host mirror of csql::TableExpression (include/evql_host.hpp) driven
like the reference's ResultCursor drives an operator; output compared, in the
reference's golden-file format, with the oracle."""
import os
import subprocess

import pytest

from eventql_amd import capi as K
from eventql_amd.plan import Plan, col, count, sum_
import oracle_lib as O
import tables as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "group_by_scan_test.cc")


def _compile(tmp_path):
    exe = str(tmp_path / "group_by_scan_test")
    libdir = os.path.join(ROOT, "eventql_amd")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
        "-L" + libdir, "-levql_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_adapter_compiles_against_the_c_abi(built, tmp_path):
    exe = _compile(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_operator_matches_oracle(built, tmp_path):
    exe = _compile(tmp_path)
    img, _ = T.mixed_table(300_000)
    path = str(tmp_path / "t.cst")
    open(path, "wb").write(img)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    plan = Plan(T.MIXED_SCHEMA, select=[col("k"), sum_(col("a")), count(1)],
                group_by=[col("k")], where=col("a") > 30000)
    exp = O.oracle_run(path, plan)
    lines = ["k;sum(a);count(1)"] + ["%d;%d;%d" % r for r in sorted(exp.rows())]
    assert out.stdout.strip().split("\n") == lines
    assert "heartbeats=2" in out.stderr
    # the same query as two partial operators (row ranges) + GroupByMerge
    out = subprocess.run([exe, path, "merge"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().split("\n") == lines
