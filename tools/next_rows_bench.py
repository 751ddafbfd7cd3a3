"""SURVEY.md section 8f rows at the size of the headline benchmark (10 M synthetic 150 bp reads, k = 21), each against the
reference's own code on the host cores of the same box:
   N1  SDBG::LoadFromFile (oracle/_ref/ref_sdbg_dump)           vs  mhx_sdbg_build_index on the device   -> every array equal
   N4  sdbg_pruning::RemoveTips (the same dumper, max_tip_len 2k) vs  mhx_sdbg_remove_tips                 -> count + bitmap equal
   N2  megahit_core iterate (oracle/_ref/ref_megahit_core)        vs  mhx_core iterate                     -> equal edge sets
The contigs of N2 are cut from the genome the reads were drawn from (pieces of 100..500 bases, half of them reverse-
complemented, consecutive pieces overlapping by k bases as unitigs do, one junction in ten a gap instead), written as the assembler writes them.
   python tools/next_rows_bench.py [reads] > profiles/r03_next_rows.json"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import make_fullsize_golden as mfg  # noqa: E402
from megahit_amd import canon, lib  # noqa: E402

MHX = os.path.join(ROOT, "megahit_amd", "mhx_core")
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_sdbg_dump")
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "ref_megahit_core")
K, M, STEP = 21, 2, 8


def kernel_table(stats, top=8):
    """per-kernel HIP-event ms and algorithmic bytes -> the heaviest kernels with their achieved algorithmic GB/s"""
    rows = sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:top]
    return {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "algo_bytes": v["bytes"],
                "GBs": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None, "frac_of_8TBs": round(v["bytes"] / v["ms"] / 1e6 / 8000.0, 4) if v["ms"] else None}
            for k, v in rows}


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=e)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        raise SystemExit("command failed: " + " ".join(cmd))
    return dt, p.stderr


def write_contigs(path_ctg, path_bub, n_reads):
    G = int(n_reads * 2.5)
    genome = np.random.default_rng(1).integers(0, 4, size=G, dtype=np.uint8)  # the genome of mfg.gen_library
    rng = np.random.default_rng(99)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 0
    with open(path_ctg, "wb") as f:
        pos = 0
        while pos + 200 < G:
            ln = int(rng.integers(100, 501))
            piece = genome[pos:pos + ln]
            if rng.random() < 0.5:
                piece = (3 - piece[::-1]).astype(np.uint8)
            f.write(b">k%d_%d flag=0 multi=30.0000 len=%d\n" % (K, n, piece.size))
            f.write(lut[piece].tobytes() + b"\n")
            n += 1
            pos += ln - K if rng.random() < 0.9 else ln + int(rng.integers(0, 7))  # unitigs overlap by k bases; some gaps
    with open(path_bub, "wb") as f:
        for i in range(500):
            a = int(rng.integers(0, G - 100))
            piece = genome[a:a + int(rng.integers(30, 90))].copy()
            piece[piece.size // 2] ^= 1
            f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n" % (K, n + i, piece.size))
            f.write(lut[piece].tobytes() + b"\n")
    return n


def sorted_edges(prefix):
    hdr, edges, _ = canon.canonical_edges(prefix)
    edges = np.ascontiguousarray(edges)
    return hdr, edges[np.lexsort(edges.T[::-1])] if edges.size else edges


def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    import test_gpu_sdbg_index as ti
    out = {"reads": n_reads, "k": K}
    with tempfile.TemporaryDirectory(prefix="mhx_next_") as d:
        mfg.gen_library(os.path.join(d, "reads"), n_reads)
        run([MHX, "read2sdbg", "-k", str(K), "-m", str(M), "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file",
             os.path.join(d, "reads"), "--output_prefix", os.path.join(d, "g")])
        # ---- N1 + N4
        dump = os.path.join(d, "ref.dump")
        _dt, log = run([REF_DUMP, os.path.join(d, "g"), dump, str(2 * K)])
        t_ref_load = float(re.search(r"LoadFromFile ([0-9.]+) s", log).group(1))
        t_ref_tips = float(re.search(r"RemoveTips ([0-9.]+) s", log).group(1))
        want = ti.read_dump(dump)
        os.remove(dump)
        eng = lib.Engine(0)
        t0 = time.perf_counter()
        k = ti.load_files_into(eng, os.path.join(d, "g"))
        t_files = time.perf_counter() - t0
        eng.sdbg_build_index(k)  # warm-up (allocations)
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        info = eng.sdbg_build_index(k)
        eng.synchronize()
        t_index = time.perf_counter() - t0
        k_index = kernel_table(eng.profile_get())
        ti.check_index(eng, k, want)  # every array against the reference's (raises on a difference)
        info = eng.sdbg_build_index(k)
        eng.synchronize()
        eng.profile_reset()
        t0 = time.perf_counter()
        n_tips = eng.sdbg_remove_tips(info, 2 * K)
        eng.synchronize()
        t_tips = time.perf_counter() - t0
        k_tips = kernel_table(eng.profile_get())
        eng.profile(False)
        tips_equal = n_tips == int(want["tips_removed"][0]) and np.array_equal(eng.fetch(lib.BUF_SDBG_INVALID, np.uint64), want["invalid_after_tips"])
        out["N1_sdbg_index"] = {"records": int(info.n_items), "reference_LoadFromFile_s": t_ref_load, "mhx_build_index_s": round(t_index, 4),
                                "mhx_read_files_and_upload_s": round(t_files, 3), "all_arrays_equal": True, "kernels": k_index}
        out["N4_remove_tips"] = {"max_tip_len": 2 * K, "tips_removed": n_tips, "reference_RemoveTips_s": t_ref_tips, "mhx_remove_tips_s": round(t_tips, 4),
                                 "count_and_bitmap_equal": bool(tips_equal), "kernels": k_tips}
        del eng, want
        # ---- N2
        n_ctg = write_contigs(os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), n_reads)
        common = ["iterate", "-c", os.path.join(d, "c.fa"), "-b", os.path.join(d, "b.fa"), "-t", "16", "-k", str(K), "-s", str(STEP),
                  "-r", os.path.join(d, "reads.bin")]
        t_ref, _ = run([REF_FULL] + common + ["-o", os.path.join(d, "it_ref")])
        t_mhx, log = run([MHX] + common + ["-o", os.path.join(d, "it_mhx")])
        prof = os.path.join(d, "it_prof.json")
        t_mhx2, log = run([MHX] + common + ["-o", os.path.join(d, "it_mhx")], env={"MHX_PROFILE": "1", "MHX_PROFILE_JSON": prof})
        with open(prof) as f:
            k_iter = kernel_table(json.load(f)["kernels"])
        hr, er = sorted_edges(os.path.join(d, "it_ref"))
        hm, em = sorted_edges(os.path.join(d, "it_mhx"))
        out["N2_iterate"] = {"contigs": n_ctg, "step": STEP, "edges": int(er.shape[0]), "reference_wall_s_16_threads": round(t_ref, 3),
                             "mhx_core_wall_s": round(min(t_mhx, t_mhx2), 3), "edge_sets_equal": bool(hm == hr and em.shape == er.shape and np.array_equal(em, er)),
                             "kernels": k_iter}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
