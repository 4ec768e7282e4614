"""The C++ mirror of the reference's header-only API (include/libspm/) with the reference's own test cases."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "reference_cases")


def test_mirror_headers_compile_with_reference_warning_flags():
    """-std=c++20 -pedantic -Wall -Wextra -Werror (the reference's test flags, test/jstmap_test.cmake:44); concept
    checks (spm::window_matcher / restorable_matcher) are static parts of that program.  Built by build()."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_reference_cases_through_cpp_api():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


JST = os.path.join(ROOT, "tests", "cpp", "jst_cases")


def test_journal_and_vcf_ingestion_reproduce_reference_haplotypes():
    """CPU part of tests/cpp/jst_cases.cpp: journal invariants/edits, and reference + VCF -> all 100 haplotypes of
    the reference's fixture FASTAs (SNP-only and SNP+indel/SV sets)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    r = subprocess.run([JST, "--cpu"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("100/100 haplotypes reproduce the fixture FASTA") == 2


@pytest.mark.gpu
def test_journaled_sequence_tree_search_equals_per_haplotype_scans():
    r = subprocess.run([JST], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
