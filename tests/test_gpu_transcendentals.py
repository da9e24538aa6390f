"""How accurate are the hardware's transcendental instructions?  The worst-case error bounds of the engine's approximate forms
(tools/filter_error_bound.py, tests/test_filter_error_bound.py) charge v_rcp / v_rsq / v_sqrt with one ulp -- what the ISA guide
says.  tests/ulp_sweep.hip measures it on the device this suite runs on, EXHAUSTIVELY: every positive normal float through the three
fp32 instructions (2.1e9 operands each), every positive normal half through the three fp16 ones, and the two instructions the block
bound uses (v_exp_f32 over the exponents it can meet, v_sin_f32 over [2^-20, 1/6] revolutions) against the cushions `stats_bound`
(filter.hip) gives them."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_transcendental_instructions_are_as_accurate_as_the_bounds_assume(tmp_path):
    exe = str(tmp_path / "ulp_sweep")
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "ulp_sweep.hip")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    print(out)
    worst = {ln.split()[0]: float(ln.split()[1]) for ln in out.splitlines() if ln.strip()}
    # tools/filter_error_bound.py: ULP = 2^-23 (single) / 2^-10 (half) RELATIVE, i.e. one ulp of a result at the bottom of its binade
    for name in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f16", "v_rsq_f16", "v_sqrt_f16"):
        assert worst[name] <= 1.0, (name, worst[name])
    # stats_bound: e_max = exp2(..) is multiplied into sin(..) whose cushion is 1.001 x + 1e-5: two ulp of exp2 are 2.4e-7 relative
    assert worst["v_exp_f32"] <= 2.0, worst
    # ... and v_sin_f32's absolute error must stay below the additive 1e-5 of that cushion (168 x 2^-24)
    assert worst["v_sin_f32_abs_2^-24"] <= 100.0, worst


def test_sdwa_joined_half_transcendentals_equal_the_builtins(tmp_path):
    """colour_diff_fp16.hpp h2_rsq / h2_rcp / h2_sqrt (the packed-fp16 form of the optional pre-tier) join the two halves of a packed transcendental with the SDWA destination select
    (inline asm: the compiler's hazard recognizer does not look inside, so the block carries its own wait states before, between and
    after the two instructions).  tools/experiments/archive/r4_sdwa_test.hip runs them beside the builtins over 65 536 operand pairs, alone
    and chained into the arithmetic the pre-tier's form puts around them: every result must be the builtin's, bit for bit."""
    exe = str(tmp_path / "sdwa_test")
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "mosaicmagnifique_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tools", "experiments", "archive", "r4_sdwa_test.hip")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    print(out)
    assert "mismatches rsq 0 rcp 0 sqrt 0 chained 0 of 65536" in out, out
