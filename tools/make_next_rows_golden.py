"""Known answers of the SURVEY section 8f rows, from the reference's own code, for tests that must not need the reference at
run time (tests/golden/next_rows.json + tests/golden/iterate_*/):
   N4  per golden SdBG case: the graph the reference builds (oracle/_ref/ref_core), then sdbg_pruning::RemoveTips through
       oracle/_ref/ref_sdbg_dump with max_tip_len = 2k -> number of tips removed, sha256 of the invalid bit vector before
       and after
   N2  three small iterate inputs (contigs, bubbles, reads: tests/test_oracle_iterate.make_case) -> sha256 of the sorted
       edge records of oracle/_ref/ref_megahit_core iterate
Run here (needs /root/reference built into oracle/_ref):   python tools/make_next_rows_golden.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import consume_util as cu  # noqa: E402
import golden_util as gu  # noqa: E402
import test_gpu_sdbg_index as ti  # noqa: E402
import test_oracle_iterate as toi  # noqa: E402
from megahit_amd import canon  # noqa: E402

ITER_CASES = [(21, 8, 1), (22, 6, 5), (39, 20, 4)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {"tips": [], "iterate": []}
    for ent in ti._tip_cases():
        with tempfile.TemporaryDirectory() as d:
            gu.run_case(gu.REF_CORE, ent, d)
            k = ent["case"]["k"]
            dump = os.path.join(d, "ref.dump")
            subprocess.run([ti.REF_DUMP, os.path.join(d, "out"), dump, str(2 * k)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            want = ti.read_dump(dump)
            out["tips"].append({"case": ent["case"], "max_tip_len": 2 * k, "n_items": int(want["meta"][0]), "tips_removed": int(want["tips_removed"][0]),
                                "invalid_before": sha(want["invalid"]), "invalid_after": sha(want["invalid_after_tips"])})
    for k, step, seed in ITER_CASES:
        name = "iterate_k%d_s%d" % (k, step)
        d = os.path.join(gu.GOLD, name)
        os.makedirs(d, exist_ok=True)
        toi.make_case(d, k, seed)
        with tempfile.TemporaryDirectory() as t:
            subprocess.run([cu.REF_FULL, "iterate", "-c", os.path.join(d, "c.fa"), "-b", os.path.join(d, "b.fa"), "-t", "3", "-k", str(k), "-s", str(step),
                            "-o", os.path.join(t, "ref"), "-r", os.path.join(d, "reads.bin")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            hdr, edges, _ = canon.canonical_edges(os.path.join(t, "ref"))
            edges = np.ascontiguousarray(edges)
            edges = edges[np.lexsort(edges.T[::-1])]
        out["iterate"].append({"dir": name, "k": k, "step": step, "header": hdr, "n_edges": int(edges.shape[0]), "sorted_edges": sha(edges)})
    with open(os.path.join(gu.GOLD, "next_rows.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("%d tip cases, %d iterate cases -> tests/golden/next_rows.json" % (len(out["tips"]), len(out["iterate"])))


if __name__ == "__main__":
    main()
