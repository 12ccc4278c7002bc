"""Randomised cross-checks (small editions of tools/stress_forms.py and tools/stress_ingest.py): on random inputs -- read
lengths 48-250, errors, repeats, duplicates, masks; messy FASTA / FASTQ files -- the source-side form of the transitive
reduction must equal the per-target replay, and the GPU input stage must equal the host one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reduction_forms_agree_on_random_inputs():
    out = _run("stress_forms.py", 80, 77000)
    assert "mismatches 0" in out and "source-side used" in out


def test_gpu_input_stage_equals_host_on_random_files():
    out = _run("stress_ingest.py", 40, 500)
    assert "mismatches 0" in out
