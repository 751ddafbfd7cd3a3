"""CPU, only where the reference sources are present (this container): the oracle binary against oracle/_ref/ref_core
(= the reference's own sources compiled in place, oracle/Makefile) on FRESH seeded inputs, beyond the committed golden
digests.  Skipped on the GPU box, where /root/reference and ref_core's build do not exist... the prebuilt binary does
travel, so the test also runs wherever oracle/_ref/ref_core is found."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
from megahit_amd import canon, synth

pytestmark = pytest.mark.skipif(not os.path.exists(gu.REF_CORE), reason="oracle/_ref/ref_core not built (no reference sources here)")


def _lib(tmp, seed, ragged):
    rng = np.random.default_rng(seed)
    reads = synth.gen_pe_reads(1200, 5000, read_len=90, frag=220, err=0.015, seed=seed)
    if ragged:
        blocks = [[r[: int(rng.integers(0, 91))] for r in reads]]
    else:
        blocks = [reads]
    prefix = os.path.join(tmp, "reads")
    synth.write_read_lib(prefix, blocks)
    return prefix


def _run(binary, args):
    subprocess.run([binary] + args, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


COMMON = ["--host_mem", "2e9", "--num_cpu_threads", "3"]


@pytest.mark.parametrize("k,m,ragged,seed", [(21, 2, False, 5), (25, 3, True, 6), (33, 2, True, 7)])
def test_count_and_seq2sdbg_with_mercy(k, m, ragged, seed, tmp_path):
    gu.ensure_oracle()
    lib = _lib(str(tmp_path), seed, ragged)
    out = {}
    for name, binary in (("ref", gu.REF_CORE), ("orc", gu.ORACLE_CORE)):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        cnt, sd = os.path.join(d, "cnt"), os.path.join(d, "sdbg")
        _run(binary, ["count", "-k", str(k), "-m", str(m), "--read_lib_file", lib, "--output_prefix", cnt] + COMMON)
        _run(binary, ["seq2sdbg", "-k", str(k), "--kmer_from", "0", "--input_prefix", cnt, "--output_prefix", sd, "--need_mercy"] + COMMON)
        out[name] = (canon.digest_edges(cnt), canon.digest_file(cnt + ".cand"), canon.digest_file(cnt + ".counting"), canon.digest_sdbg(sd))
    assert out["ref"] == out["orc"]


@pytest.mark.parametrize("k,m,mercy,ragged,seed", [(21, 2, False, False, 8), (21, 2, True, True, 9), (29, 3, True, True, 10), (27, 1, False, True, 11)])
def test_read2sdbg(k, m, mercy, ragged, seed, tmp_path):
    gu.ensure_oracle()
    lib = _lib(str(tmp_path), seed, ragged)
    out = {}
    for name, binary in (("ref", gu.REF_CORE), ("orc", gu.ORACLE_CORE)):
        d = os.path.join(str(tmp_path), name)
        os.makedirs(d)
        o = os.path.join(d, "out")
        _run(binary, ["read2sdbg", "-k", str(k), "-m", str(m), "--read_lib_file", lib, "--output_prefix", o] + (["--need_mercy"] if mercy else []) + COMMON)
        out[name] = canon.digest_sdbg(o)
    assert out["ref"] == out["orc"]  # with mercy: the oracle's kmsort restatement reproduces the reference's tie order (H1)
