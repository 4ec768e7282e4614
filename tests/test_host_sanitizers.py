"""CPU: the host side of the product that runs without a device -- seed index build (sparse, anchored and dense passes),
match-mask tables, the gatherv protocol, journal + FASTA / VCF ingestion -- and the CPU oracle, under AddressSanitizer and
UndefinedBehaviorSanitizer ("host code under ASan/UBSan").  CPU builds only: `make -C libspm_amd/csrc asan`,
`make -C oracle asan`, `make -C tests/cpp asan`; never on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if out and os.path.isabs(out) and os.path.exists(out) else None


def _env():
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    return env


def test_index_tables_protocol_and_oracle_under_sanitizers():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "libspm_amd", "csrc"), "-s", "asan"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = _env()
    env["LD_PRELOAD"] = asan   # the interpreter is not instrumented: the runtime has to come first
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], capture_output=True, text=True,
                       env=env, timeout=900)
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    assert p.returncode == 0, (p.returncode, p.stderr[-4000:])
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and len(res["cases"]) >= 20
    dense = [c for c in res["cases"] if c["case"].startswith("dense pass")]
    assert dense and all(c["passes"] == 1 for c in dense)


def test_journal_and_ingestion_under_sanitizers():
    """tests/cpp/jst_cases --cpu (journal invariants and edits; reference + VCF -> the fixture haplotypes) built with
    -fsanitize=address,undefined."""
    if _libasan() is None:
        pytest.skip("gcc's libasan.so not found")
    cpp = os.path.join(ROOT, "tests", "cpp")
    if not os.path.exists(os.path.join(ROOT, "libspm_amd", "libspm_hip.so")):
        pytest.skip("libspm_hip.so not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    subprocess.check_call(["make", "-C", cpp, "-s", "asan"])
    p = subprocess.run([os.path.join(cpp, "jst_cases_asan"), "--cpu"], capture_output=True, text=True, env=_env(), timeout=900)
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "0 failures" in p.stdout
