"""Regenerates the oracle regression fixtures under tests/golden/ (run from the repo root:
python tests/golden/make_golden.py).  Inputs are the reference's own test files copied verbatim
into tests/golden/data/ (data, not source).  Outputs come from oracle/tho_oracle."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
D = os.path.join(HERE, "data")
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
BIN = os.path.join(ROOT, "oracle", "tho_oracle")
JOBS = {
    "test_fq_tab": ["-f", "2", "-t", "4", os.path.join(D, "test.fq")],
    "test_fq_unit": ["-u", "-f", "2", "-t", "4", os.path.join(D, "test.fq")],
    "full_length_fastq": ["-f", "3", "-t", "1", "-5", os.path.join(D, "5prime.fa"), "-3", os.path.join(D, "3prime.fa"),
                          os.path.join(D, "full_length.fa")],
}
for name, args in JOBS.items():
    out = subprocess.run([BIN, *args], check=True, capture_output=True, text=True).stdout
    open(os.path.join(HERE, "oracle_%s.txt" % name), "w").write(out)
    print(name, len(out.splitlines()), "lines")
