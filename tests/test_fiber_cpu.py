"""The cooperative fibers that run the K verifier loops of a lock-step batch on one host thread (zkcnn_amd/csrc/host/fiber.hpp; the reference runs one
loop per process, src/verifier.cpp:118-373): round-robin order, exceptions reach the driver, and the portable ucontext fallback (every host but x86-64;
round-4 advisor finding) behaves like the hand-written switch."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [[], ["-DZKFIBER_UCONTEXT"]], ids=["x86-64 switch", "ucontext fallback"])
def test_fibers_round_robin_and_exceptions(tmp_path, flags):
    exe = str(tmp_path / "fiber_check")
    subprocess.run(["g++", "-std=c++14", "-O2", *flags, "-I", os.path.join(ROOT, "zkcnn_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "native", "fiber_check.cpp"), "-o", exe], check=True)
    for mb in ("1", "8"):
        out = subprocess.run([exe], capture_output=True, text=True, check=True, env=dict(os.environ, ZKCNN_FIBER_STACK_MB=mb)).stdout
        assert out.split() == "0 10 20 1 11 21 2 12 22 caught 1".split()
