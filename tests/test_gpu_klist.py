"""GPU, BASELINE configs[3] at single-GPU-shard size: the iterative k-list 21,29,39,59,79,99,119 on 12.5 M synthetic
150 bp PE reads (genome with planted repeats, so that every k of the list has work: tools/make_klist_golden.py).

k = 21: `mhx_core count` + `seq2sdbg --need_mercy` on the read library (regenerated here, .bin md5 checked) against the
digests of the reference's own run.  k >= 29: `mhx_core seq2sdbg` on the inputs the REFERENCE's assemble / local / iterate
produced at that size (contig files + the unsorted edge file, packed under oracle/_ref/klist/ by the golden tool) against
the digest of the SdBG the reference's seq2sdbg built from the same files (tests/golden/klist.json).  Item widths 2, 3, 4,
5, 6, 7, 9 words.  Reference: src/sorting/seq_to_sdbg.cpp:359-528,530-789."""
import gzip
import json
import os
import shutil
import subprocess
import time

import pytest

import golden_util as gu
from megahit_amd import canon, synth

pytestmark = pytest.mark.gpu

KLIST_JSON = os.path.join(gu.GOLD, "klist.json")
PACK = os.path.join(gu.ROOT, "oracle", "_ref", "klist")
if os.path.exists(KLIST_JSON):
    with open(KLIST_JSON) as f:
        KL = json.load(f)
else:
    KL = None

needs_golden = pytest.mark.skipif(KL is None, reason="tests/golden/klist.json not generated")


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.perf_counter()
    p = subprocess.run([gu.MHX_CORE] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return time.perf_counter() - t0, p.stderr


def sdbg_counts(prefix):
    _hdr, rows = canon.read_sdbg_info(prefix)
    live = [r for r in rows if r[0] != canon.NULL_ID]
    return sum(r[3] for r in live), sum(r[4] for r in live), sum(r[5] for r in live)


def check_sdbg(prefix, want):
    assert sdbg_counts(prefix) == (want["n_sdbg"], want["n_tips"], want["n_large"])
    assert canon.digest_sdbg(prefix) == want["digest"]


def unpack(k, dst):
    src = os.path.join(PACK, "k%d" % k)
    os.makedirs(dst, exist_ok=True)
    for name in os.listdir(src):
        with gzip.open(os.path.join(src, name), "rb") as fi, open(os.path.join(dst, name[:-3]), "wb") as fo:
            shutil.copyfileobj(fi, fo, 1 << 22)


def cli_args(k, d, out):
    args = []
    it = iter(KL["cases"]["k%d" % k]["args"])
    for a in it:
        args.append(a)
        if a in ("--contig", "--bubble", "--addi_contig", "--local_contig", "--input_prefix"):
            args.append(os.path.join(d, next(it)))
        elif a in ("-k", "--kmer_from"):
            args.append(next(it))
    return args + ["--host_mem", "64e9", "--num_cpu_threads", "8", "--output_prefix", out]


@needs_golden
@pytest.mark.skipif(not os.path.isdir(PACK), reason="oracle/_ref/klist not packed (tools/make_klist_golden.py)")
@pytest.mark.parametrize("k", [29, 39, 59, 79, 99, 119])
@pytest.mark.parametrize("hybrid", ["1", "0"], ids=["prefix+finish", "lsd-passes"])
def test_klist_seq2sdbg_on_reference_produced_inputs(tmp_path, k, hybrid):
    if hybrid == "0" and k not in (29, 119):
        pytest.skip("the plain LSD plan is checked at the narrowest and the widest key")
    d = str(tmp_path / "in")
    unpack(k, d)
    out = str(tmp_path / "out")
    run(cli_args(k, d, out), env={"MHX_SORT_HYBRID": hybrid})
    check_sdbg(out, KL["cases"]["k%d" % k])


@needs_golden
def test_klist_k21_count_then_seq2sdbg_need_mercy(tmp_path):
    d = str(tmp_path)
    _genome, blocks = synth.gen_shard_library(KL["reads"], KL["genome_seed"], KL["read_seed0"], repeat_families=KL["repeat_families"])
    synth.write_read_lib(os.path.join(d, "reads"), blocks)
    del blocks
    assert canon.digest_file(os.path.join(d, "reads.bin")) == KL["lib_bin_md5"], "the generator is not deterministic across boxes"
    cnt = os.path.join(d, "21")
    run(["count", "-k", "21", "-m", "2", "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file", os.path.join(d, "reads"), "--output_prefix", cnt])
    want = KL["k21"]["count"]
    hdr, _rows = canon.read_edges_info(cnt)
    assert hdr["num_edges"] == want["n_edges"]
    assert canon.digest_file(cnt + ".counting") == want["counting_md5"]
    assert canon.digest_file(cnt + ".cand") == want["cand_md5"]
    assert canon.digest_edges(cnt) == want["digest"]
    # as the orchestrator calls it: input and output prefix are the same (src/megahit:939-966)
    run(["seq2sdbg", "-k", "21", "--kmer_from", "0", "--host_mem", "64e9", "--num_cpu_threads", "8", "--input_prefix", cnt, "--need_mercy", "--output_prefix", cnt])
    check_sdbg(cnt, KL["k21"]["seq2sdbg_need_mercy"])
