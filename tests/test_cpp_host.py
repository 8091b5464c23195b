"""The C++ host mirror (naive_query_engine_amd/host/naive_db.hpp): builds on CPU (header compiles, links
against the C ABI), runs the reference's own tests on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_physical_plan")


def build_exe():
    src = os.path.join(ROOT, "tests", "cpp", "test_physical_plan.cpp")
    libdir = os.path.join(ROOT, "naive_query_engine_amd")
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", src, "-o", EXE, f"-L{libdir}", "-lnqe_hip", f"-Wl,-rpath,{libdir}",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_host_mirror_compiles_and_links():
    exe = build_exe()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_tests_through_cpp_host_mirror():
    exe = build_exe()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "15/15 tests passed" in out.stdout


def test_exchange_fabric_test_compiles_and_links():
    """tests/cpp/test_exchange_fabric.cpp (run on the GPU by tests/test_gpu_parallel.py) builds against the C ABI here"""
    from tests.test_gpu_parallel import build_fabric_exe

    assert os.path.exists(build_fabric_exe())
