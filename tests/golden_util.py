"""Run a `megahit_core`-compatible binary on the golden inputs and digest its outputs (test infra)."""
import hashlib
import json
import os
import subprocess

from megahit_amd import canon

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ORACLE_CORE = os.path.join(ROOT, "oracle", "oracle_core")
REF_CORE = os.path.join(ROOT, "oracle", "_ref", "ref_core")
MHX_CORE = os.path.join(ROOT, "megahit_amd", "mhx_core")


def cases():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)["cases"]


def case_id(ent):
    c = ent["case"]
    return "-".join("%s%s" % (k[0], v) for k, v in sorted(c.items()))


def ensure_oracle():
    if not os.path.exists(ORACLE_CORE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)


def run_case(binary, ent, workdir, extra=()):
    """Runs the case's sub-program (and, for seq2sdbg-from-count, the count that feeds it) with `binary`;
    returns a dict of digests comparable with the golden entry."""
    c = ent["case"]
    common = ["--host_mem", "2e9", "--num_cpu_threads", "3"]
    out = os.path.join(workdir, "out")

    def call(args):
        subprocess.run([binary] + args, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    got = {}
    if c["prog"] == "count":
        call(["count", "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(GOLD, c["lib"]), "--output_prefix", out] + common)
        got["edges"] = canon.digest_edges(out)
        got["cand"] = canon.digest_file(out + ".cand")
        got["counting"] = canon.digest_file(out + ".counting")
        got["n_edges"] = int(canon.canonical_edges(out)[1].shape[0])
    elif c["prog"] == "read2sdbg":
        call(["read2sdbg", "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(GOLD, c["lib"]), "--output_prefix", out] +
             (["--need_mercy"] if c.get("mercy") else []) + common + list(extra))
        got["sdbg"] = canon.digest_sdbg(out)
        got["n_sdbg"] = int(sum(b[1] for b in canon.canonical_sdbg(out)[1]))
        if os.path.exists(out + ".counting"):
            got["counting"] = canon.digest_file(out + ".counting")
        if os.path.exists(out + ".mercy_cand.0"):
            got["mercy_cand_kmsort"] = hashlib.md5(canon.sorted_mercy_cand(out).tobytes()).hexdigest()
    elif c["prog"] == "seq2sdbg" and c["input"] == "count":
        cnt = os.path.join(workdir, "cnt")
        call(["count", "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(GOLD, c["lib"]), "--output_prefix", cnt] + common)
        call(["seq2sdbg", "-k", str(c["k"]), "--kmer_from", "0", "--input_prefix", cnt, "--output_prefix", out] +
             (["--need_mercy"] if c.get("mercy") else []) + common)
        got["sdbg"] = canon.digest_sdbg(out)
        got["n_sdbg"] = int(sum(b[1] for b in canon.canonical_sdbg(out)[1]))
    else:
        d = os.path.join(GOLD, "ctg_k%d" % c["k"])
        args = ["seq2sdbg", "-k", str(c["k"]), "--kmer_from", str(c["k_from"]), "--contig", d + ".contigs.fa", "--bubble", d + ".bubble.fa"]
        if c["input"] == "contigs":
            args += ["--input_prefix", d, "--addi_contig", d + ".addi.fa", "--local_contig", d + ".local.fa"]
        call(args + ["--output_prefix", out] + common)
        got["sdbg"] = canon.digest_sdbg(out)
        got["n_sdbg"] = int(sum(b[1] for b in canon.canonical_sdbg(out)[1]))
    return got
