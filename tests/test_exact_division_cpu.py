"""CPU rehearsal of the kernel's exact division by a per-ray constant (csrc/exact_div.h):
the same five operations with fmaf() against `/`, 20M operand pairs incl. structured
significands (all ones, powers of two, single bits)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_five_operation_division_equals_true_division(tmp_path):
    exe = str(tmp_path / "divcheck")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "native_divcheck.c"), "-lm"], check=True)
    for seed in (1, 77):
        out = subprocess.run([exe, "10000000", str(seed)], check=True, capture_output=True, text=True)
        assert int(out.stdout.strip()) == 0, out.stderr
