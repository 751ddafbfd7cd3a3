"""CPU: the edge-file reader / writers and the contig reader of the CLI's host side (megahit_amd/csrc/host/formats.cpp),
driven by a small C++ program built here with g++ (no GPU, no libmhx): tests/host/formats_check.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "megahit_amd", "csrc", "host")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def test_edge_files_and_contigs_round_trip(tmp_path):
    exe = os.path.join(str(tmp_path), "formats_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", HOST, os.path.join(ROOT, "tests", "host", "formats_check.cpp"),
                    os.path.join(HOST, "formats.cpp"), "-lz", "-o", exe], check=True)
    work = os.path.join(str(tmp_path), "w")
    os.makedirs(work)
    p = subprocess.run([exe, work], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr
