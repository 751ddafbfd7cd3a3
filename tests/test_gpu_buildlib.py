"""GPU: SURVEY.md section 8f N3 — `mhx_core buildlib` (FASTA/FASTQ parsed and packed on the GPU, mhx_fastx_to_records)
writes byte-identical .bin / .lib_info to the reference's buildlib (oracle/_ref/ref_core: sequence_lib.cpp:8-91,
fastx_reader.cpp:28-71, kseq.h:193-247) — plain FASTA (multi-line, last line unterminated, empty records), four-line
FASTQ (also gzip'ed), paired and interleaved libraries, N-trimming, lower case, foreign letters — and hands the shapes
it declines (CRLF, multi-line FASTQ, junk, malformed records, unequal mates) to the sequential parser."""
import os
import subprocess

import numpy as np
import pytest

import buildlib_util as bu
import golden_util as gu
from megahit_amd import canon, synth
from test_buildlib import run_both

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(gu.REF_CORE), reason="needs oracle/_ref/ref_core")]


def test_gpu_buildlib_matches_reference(tmp_path):
    cases = bu.make_cases(str(tmp_path), seed=2)
    for name, (lib_text, seq_only) in cases.items():
        got = run_both(lib_text, str(tmp_path), name, None)
        assert got["mhx"] == got["ref"], name
        libf = os.path.join(str(tmp_path), name + ".lib")
        p = subprocess.run([gu.MHX_CORE, "buildlib", libf, os.path.join(str(tmp_path), name + "_again")], stderr=subprocess.PIPE, text=True)
        assert ("using the sequential parser" in p.stderr) == seq_only, (name, p.stderr[-400:])


def test_gpu_buildlib_then_count_equals_reference_pipeline(tmp_path):
    """FASTA -> buildlib -> count through mhx_core only = the same through the reference only"""
    d = str(tmp_path)
    reads = synth.gen_pe_reads(3000, 8000, read_len=100, frag=250, err=0.01, seed=4)
    synth.write_fasta(os.path.join(d, "r1.fa"), reads[0::2])
    synth.write_fasta(os.path.join(d, "r2.fa"), reads[1::2])
    with open(os.path.join(d, "lib"), "w") as f:
        f.write("synthetic pe\npe %s %s\n" % (os.path.join(d, "r1.fa"), os.path.join(d, "r2.fa")))
    dig = {}
    for tag, exe in (("ref", gu.REF_CORE), ("mhx", gu.MHX_CORE)):
        lib = os.path.join(d, "lib_" + tag)
        subprocess.run([exe, "buildlib", os.path.join(d, "lib"), lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([exe, "count", "-k", "21", "-m", "2", "--host_mem", "2e9", "--num_cpu_threads", "3", "--read_lib_file", lib,
                        "--output_prefix", os.path.join(d, "cnt_" + tag)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dig[tag] = (canon.digest_file(lib + ".bin"), canon.digest_edges(os.path.join(d, "cnt_" + tag)))
    assert dig["mhx"] == dig["ref"]


def test_fastx_to_records_through_the_c_abi(engine):
    text = b">a\nACGTNNACGT\n>b\nnnnn\n>c\nacgtacgtacgtacgtacgt\n"
    r, rec = engine.fastx_to_records(text)
    assert (r.status, r.n_reads, r.n_bases, r.max_len) == (0, 3, 4 + 1 + 20, 20)
    assert list(rec[:2]) == [4, 0x1B000000]           # ACGT
    assert list(rec[2:4]) == [1, 0]                    # all N -> one fake 'A'
    assert rec[4] == 20 and rec[5] == 0x1B1B1B1B and rec[6] == 0x1B000000
    r, rec = engine.fastx_to_records(b">a\r\nACGT\r\n")
    assert r.status == 1 and rec is None
