"""Compile hygiene of the three header-only adapters (cube_slam_wu_amd/adapters): the build image has no Eigen / OpenCV / g2o, so they
are type-checked against declaration-only stand-ins (tests/adapter_stubs/README.md) with g++ -fsyntax-only.  Catches misspelt members,
wrong signatures, and a g2o::Solver subclass that is still abstract.  Nothing is linked or run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapters_parse_and_type_check():
    stubs = os.path.join(ROOT, "tests", "adapter_stubs")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function", "-I", stubs, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "cube_slam_wu_amd", "adapters"), os.path.join(stubs, "compile_adapters.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
