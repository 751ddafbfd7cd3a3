"""GPU: REFERENCE code consumes what mhx_core writes (VERDICT r1 "What's weak" #2).

(a) the reference's `assemble` (SdBG loader: src/sdbg/sdbg_raw_content.cpp:18-96 with its file-offset assertion ON,
    rank/select construction, graph traversal) loads mhx_core's 1-file and 3-file .sdbg* and writes the same contigs as
    from the reference's own .sdbg*;
(b) the reference's `seq2sdbg` (EdgeReader, src/sequence/io/edge/edge_reader.h:105-138) reads mhx_core's .edges*;
(c) the reference's unmodified orchestrator (src/megahit --test) with megahit_core -> mhx_core reproduces the final
    contigs of the all-reference run, 2-pass (count + seq2sdbg --need_mercy) and --kmin-1pass (read2sdbg --need_mercy)."""
import os
import subprocess

import numpy as np
import pytest

import consume_util as cu
import golden_util as gu
from megahit_amd import canon, synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(cu.REF_FULL), reason="oracle/_ref/ref_megahit_core not built")]


def call(binary, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stderr


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("consume"))
    reads = synth.gen_pe_reads(20000, 60000, read_len=100, frag=300, err=0.01, seed=21)
    synth.write_read_lib(os.path.join(d, "reads"), [reads])
    return d


def assemble(sdbg_prefix, out_prefix):
    call(cu.REF_FULL, ["assemble", "-s", sdbg_prefix, "-o", out_prefix, "-t", "1", "--min_standalone", "200", "--prune_level", "2",
                       "--merge_len", "20", "--merge_similar", "0.95", "--cleaning_rounds", "5", "--disconnect_ratio", "0.1",
                       "--low_local_ratio", "0.2", "--min_depth", "2", "--bubble_level", "2", "--max_tip_len", "-1", "--careful_bubble"])
    return {ext: canon.digest_file(out_prefix + ext) for ext in (".contigs.fa", ".addi.fa", ".bubble_seq.fa")}


@pytest.mark.parametrize("mercy", [False, True], ids=["plain", "need_mercy"])
def test_reference_assemble_loads_our_sdbg(lib, tmp_path, mercy):
    common = ["-k", "21", "-m", "2", "--host_mem", "4e9", "--num_cpu_threads", "3", "--read_lib_file", os.path.join(lib, "reads")]
    flag = ["--need_mercy"] if mercy else []
    ref = str(tmp_path / "ref")
    call(gu.REF_CORE, ["read2sdbg"] + common + flag + ["--output_prefix", ref])
    want = assemble(ref, str(tmp_path / "ref_asm"))
    assert os.path.getsize(str(tmp_path / "ref_asm.contigs.fa")) > 10000
    one = str(tmp_path / "one")
    call(gu.MHX_CORE, ["read2sdbg"] + common + flag + ["--output_prefix", one])
    assert assemble(one, str(tmp_path / "one_asm")) == want
    three = str(tmp_path / "three")
    call(gu.MHX_CORE, ["read2sdbg"] + common + flag + ["--output_prefix", three], env={"MHX_NUM_OUT_FILES": "3"})
    assert os.path.exists(three + ".sdbg.2")
    assert assemble(three, str(tmp_path / "three_asm")) == want


@pytest.mark.parametrize("n_files", [1, 3])
def test_reference_seq2sdbg_reads_our_edges(lib, tmp_path, n_files):
    common = ["--host_mem", "4e9", "--num_cpu_threads", "3"]
    cnt_ref, cnt_mhx = str(tmp_path / "cref"), str(tmp_path / "cmhx")
    args = ["count", "-k", "21", "-m", "2", "--read_lib_file", os.path.join(lib, "reads")] + common
    call(gu.REF_CORE, args + ["--output_prefix", cnt_ref])
    call(gu.MHX_CORE, args + ["--output_prefix", cnt_mhx], env={"MHX_NUM_OUT_FILES": str(n_files)})
    assert canon.digest_edges(cnt_mhx) == canon.digest_edges(cnt_ref)
    outs = {}
    for name, cnt in (("ref", cnt_ref), ("mhx", cnt_mhx)):
        out = str(tmp_path / ("s_" + name))
        call(cu.REF_FULL, ["seq2sdbg", "-k", "21", "--kmer_from", "0", "--input_prefix", cnt, "--output_prefix", out, "--need_mercy"] + common)
        outs[name] = canon.digest_sdbg(out)
    assert outs["mhx"] == outs["ref"]


@pytest.mark.skipif(not os.path.exists(os.path.join(cu.HARNESS, "bin_mhx", "megahit")), reason="oracle/_ref/harness not staged")
@pytest.mark.parametrize("extra", [(), ("--kmin-1pass",), ("--no-mercy",)], ids=["2pass", "kmin-1pass", "no-mercy"])
def test_unmodified_orchestrator_with_mhx_core(tmp_path, extra):
    want_summary, want = cu.run_orchestrator("bin", str(tmp_path / "ref"), extra)
    got_summary, got = cu.run_orchestrator("bin_mhx", str(tmp_path / "mhx"), extra, env={"MHX_REF_CORE": cu.REF_FULL})
    if not extra:
        assert want_summary.startswith("2 contigs, total 1788 bp")
    assert got_summary == want_summary
    assert got == want
    log = open(str(tmp_path / "mhx" / "log")).read()
    assert "GPU" in log or "mhx" in log.lower(), "the run did not go through mhx_core"
