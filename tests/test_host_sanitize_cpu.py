"""The host-side input stages of the product (alga_amd/host/ingest.cpp + host_capi.cpp: FASTA / FASTQ parsing, trimming, N / STR
filters, duplicate and prefix-read removal, packing) under AddressSanitizer + UndefinedBehaviorSanitizer, and under ThreadSanitizer (the parser and the packer run on `threads` host threads), on the CPU -- the GPU pool
runs no sanitizers, so this is where the host code gets them: the golden fixtures' inputs (must parse) and 240 generated messy or
broken files (may be rejected; no out-of-bounds access, leak of a failed call's buffers or undefined arithmetic either way)."""
import os
import shutil
import subprocess

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("sanitizers", ["address,undefined", "thread"])
def test_host_ingest_under_sanitizers(golden_dir, tmp_path, sanitizers):
    exe = str(tmp_path / "ingest_asan")
    src = [os.path.join(ROOT, "tests", "sanitize", "ingest_asan.cpp"), os.path.join(ROOT, "alga_amd", "host", "ingest.cpp"),
           os.path.join(ROOT, "alga_amd", "host", "host_capi.cpp")]
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizers, "-fno-sanitize-recover=undefined", "-pthread", "-o", exe] + src,
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and ("cannot find" in cc.stderr or "unrecognized" in cc.stderr):
        pytest.skip("this toolchain has no sanitizer runtime: " + cc.stderr.splitlines()[-1])
    assert cc.returncode == 0, cc.stderr[-2000:]
    args, fixtures = [], []
    try:
        for name in O.FIXTURES:
            fx = O.Fixture(golden_dir, name)
            fixtures.append(fx)
            f1, f2 = fx.inputs()
            args.append(f1 + (":" + f2 if f2 else ""))
        scratch = tmp_path / "scratch"
        scratch.mkdir()
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0")
        run = subprocess.run([exe, str(scratch)] + args, capture_output=True, text=True, env=env, timeout=600)
    finally:
        for fx in fixtures:
            fx.cleanup()
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-4000:])
    assert "0 unexpected failures" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr and "LeakSanitizer" not in run.stderr
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
