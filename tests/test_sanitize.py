"""The oracle and the host-side scene code (camera, shapes, meshes — everything of the library that runs without a
device) under AddressSanitizer + UndefinedBehaviorSanitizer: tests/sanitize/sanitize_main.cpp is built as a plain
program with -fsanitize=address,undefined and run on the reference's mesh, malformed mesh files and small renders in
both arithmetics.  Any report ends the program with a non-zero status."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_oracle_and_host_code_are_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(HERE, "sanitize"), "run"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sanitize_main: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
