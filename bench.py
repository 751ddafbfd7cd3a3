#!/usr/bin/env python
"""Benchmark of the SdBG-construction hot path on MI355X (contract: see the task's bench.py section).

A "step" is one pass of `read2sdbg` (stage 1 + stage 2, k=21, min count 2, no mercy) over one batch of
synthetic 150 bp paired-end reads that is already resident in HBM (BASELINE.json configs[1]: 10 M reads
per GPU).  metric = M (k+1)-mer edge occurrences sorted+counted per second, E = sum(max(0, len-k)).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--no-cpu-baseline] [--no-e2e]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 21
MIN_COUNT = 2
READ_LEN = 150
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_reads(n_reads, rank, world):
    """Reads of this rank: PE fragments from ONE genome shared by all ranks (G = 2.5 bp per read in the
    whole job, i.e. ~60x coverage as in SURVEY.md §8d), per-rank read seed.  Returned reversed+packed.
    Rank 0 of a 1-GPU run holds exactly the library of tools/make_fullsize_golden.py (read seeds 1001+i), for which
    tests/golden/fullsize.json holds the reference's answers."""
    import numpy as np
    from megahit_amd import synth
    total_reads = n_reads * world
    G = max(5000, int(total_reads * 2.5))
    # blocks of 1 M pairs made by spawned worker processes (a few per rank), stored reversed, as the reference loads them;
    # every chunk is a whole number of words only if n*2*150 % 16 == 0: true for n multiple of 8
    jobs = synth.pe_jobs("reversed", n_reads, G, 1, 1000 * (rank + 1) + 1, read_len=READ_LEN)
    procs = max(1, min(len(jobs), (os.cpu_count() or 1) // max(1, world)))
    return np.concatenate(list(synth.map_pe_blocks(jobs, procs)))


def _latest_pmc_file():
    """the counter measurement of the latest round under profiles/ (rNN_pmc_traffic.json; tools/evidence_short.sh writes it)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return os.path.relpath(found[-1], ROOT) if found else "profiles/r03_pmc_traffic.json"


PMC_FILE = _latest_pmc_file()


def lib_built_from_current_sources():
    """libmhx.so is at least as new as every kernel source (make's own criterion).  A library REBUILT from the very sources a
    counter measurement was taken on (same build_id) need not be byte-identical to the measured one; a library OLDER than the
    sources is a stale build, and the measurement of the sources says nothing about it."""
    import glob
    try:
        so = os.path.getmtime(os.path.join(ROOT, "megahit_amd", "libmhx.so"))
        src = os.path.join(ROOT, "megahit_amd", "csrc")
        return all(os.path.getmtime(f) <= so for f in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")))
    except OSError:
        return False


def pmc_traffic(kernel_name, path=None):
    """HBM bytes per launch of `kernel_name` measured with rocprofv3 PMC passes on THIS workload and THIS build
    (profiles/r03_pmc_traffic.json, produced by tools/gpu_evidence.sh -> tools/pmc_to_json.py with the gfx950
    FETCH_SIZE x2 correction).  Counters cannot be collected from inside the timed run, so the committed measurement is
    reported — but only when it was taken on the same kernel sources (megahit_amd/buildid.py: sha256 of the sources) and the
    library that runs is the measured one (sha256 of libmhx.so) or was built from those sources afterwards: otherwise null."""
    path = path or os.path.join(ROOT, PMC_FILE)
    try:
        from megahit_amd.buildid import build_id, lib_id
        with open(path) as f:
            doc = json.load(f)
        if doc.get("build_id") != build_id():
            return None, "%s was measured on other kernel sources (build_id %s, running %s)" % (PMC_FILE, doc.get("build_id"), build_id())
        if doc.get("lib_id") is not None and doc.get("lib_id") != lib_id() and not lib_built_from_current_sources():
            return None, "%s was measured with another build of libmhx.so (lib_id %s, running %s)" % (PMC_FILE, doc.get("lib_id"), lib_id())
        kernels = doc["kernels"]
    except Exception:
        return None, None
    # profile name -> kernel symbol prefix
    # (k_s1_stream<false, ...> is the 1/64 sampling launch "s1_sample"; the group-by itself emits the aggregated items: <true, ...>)
    table = {"s1_groups": ("k_s1_stream<true", "k_s1_stream<", "k_s1_seg<", "k_tile_groups<3"), "count_groups": ("k_count_seg<",),
             "s1_extract": ("k_s1_extract_fast<", "k_s1_extract_fixed<", "k_s1_extract<"), "s1_digit_hist": ("k_s1_digit_hist", "k_s1_extract_fast<4, false"),
             "count_extract": ("k_count_extract<",),
             "radix_scatter_12B_gen": ("k_radix_onesweep_u<3, 8, 3, S1Gen", "k_radix_onesweep<3, 8, 3, S1Gen"),
             "s1_sample": ("k_s1_stream<false",), "s1_skm_groups": ("k_s1_skm<",), "s1_skm_make": ("k_skm_make<",), "s1_skm_bounds": ("k_skm_bounds",),
             "count_skm_groups": ("k_count_skm<",), "count_skm_make": ("k_skm_make<",)}
    prefixes = list(table.get(kernel_name, ()))
    for stem, names in (("radix_scatter_", ("k_radix_onesweep", "k_radix_scatter")), ("radix_hist_all_", ("k_radix_hist_all",)),
                        ("radix_hist_", ("k_radix_hist",))):
        if not prefixes and kernel_name.startswith(stem) and kernel_name.endswith("B") and kernel_name[len(stem):-1].isdigit():
            w = int(kernel_name[len(stem):-1]) // 4
            # (the chained-scan kernel that LOADS its records: SrcArray; the generated first pass has a name of its own)
            prefixes = ["%s_u<%d, 8, 3, SrcArray" % (names[0], w), "%s_u<%d, 8, 2, SrcArray" % (names[0], w),
                        "%s<%d, 8, 3, SrcArray" % (names[0], w), "%s<%d, 8, 2, SrcArray" % (names[0], w)] + \
                       ["%s<%d," % (nm, w) for nm in names] + ["%s<%d>" % (nm, w) for nm in names]
            break
    for prefix in prefixes:
        for k, v in kernels.items():
            if k.startswith(prefix):
                return v["hbm_bytes"], PMC_FILE + ":" + k
    return None, None


def pmc_traffic_per_step(path=None):
    """HBM bytes of ALL kernels of one step, from the same counter file (its command runs 3 + 1 steps): sum of launches x bytes per launch / 4;
    None unless the file was measured on the running build"""
    path = path or os.path.join(ROOT, PMC_FILE)
    try:
        from megahit_amd.buildid import build_id
        with open(path) as f:
            doc = json.load(f)
        if doc.get("build_id") != build_id():
            return None
        steps = 4 if "--steps 3 --warmup 1" in doc.get("command", "") else None
        if not steps:
            return None
        return int(sum(v["hbm_bytes"] * v["launches"] for k, v in doc["kernels"].items() if not k.startswith("at::")) / steps)
    except Exception:
        return None


def tuned_defaults():
    """name -> value of megahit_amd/mhx_tuning.conf (the tuned defaults libmhx reads at mhx_create), {} when absent"""
    out = {}
    if os.environ.get("MHX_NO_TUNING"):
        return out
    try:
        with open(os.environ.get("MHX_TUNING_FILE") or os.path.join(ROOT, "megahit_amd", "mhx_tuning.conf")) as f:
            for line in f:
                parts = line.split("#")[0].replace("=", " ").split()
                if len(parts) == 2:
                    out[parts[0]] = int(parts[1])
    except OSError:
        pass
    return out


def copy_bandwidth(torch):
    """GB/s (read + write) of a plain device-to-device copy of 2 GiB in this run: what a streaming kernel reaches on
    this part, quoted beside the datasheet peak."""
    try:
        n = 1 << 31
        a = torch.empty(n, dtype=torch.uint8, device="cuda")
        b = torch.empty(n, dtype=torch.uint8, device="cuda")
        a.zero_()
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        del a, b
        return round(2 * n / ms / 1e6, 1)
    except Exception:
        return None


def cpu_baseline(sample_reads, threads=None):
    """The reference's own CPU path (oracle/_ref/ref_core = reference sources compiled in place) on a bounded sample
    of the same workload, timed at several OpenMP thread counts (the reference does NOT get faster with every core:
    SURVEY.md section 6); the best is reported with the thread count that gave it.  Falls back to the C port
    (oracle_core, 1 thread).  The full-size figure measured once per round on the GPU box's host
    (tools/gpu_evidence.sh -> profiles/r02_cpu_fullsize.json) is quoted beside it when present."""
    from megahit_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    port = os.path.join(ROOT, "oracle", "oracle_core")
    cores = os.cpu_count() or 1
    G = max(5000, int(sample_reads * 2.5))
    reads = synth.gen_pe_reads(sample_reads // 2, G, read_len=READ_LEN, frag=400, err=0.005, seed=77)
    E = reads.shape[0] * (READ_LEN - K)
    tried = {}
    with tempfile.TemporaryDirectory(prefix="mhx_cpu_") as d:
        synth.write_read_lib(os.path.join(d, "reads"), [reads])
        if os.path.exists(ref):
            kind = "reference"
            for t in (threads or sorted({min(8, cores), min(32, cores), min(64, cores), cores})):  # up to nproc, once
                cmd = [ref, "read2sdbg", "-k", str(K), "-m", str(MIN_COUNT), "--host_mem", "32e9", "--num_cpu_threads", str(t),
                       "--read_lib_file", os.path.join(d, "reads"), "--output_prefix", os.path.join(d, "out")]
                t0 = time.perf_counter()
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                tried[t] = time.perf_counter() - t0
        else:
            if not os.path.exists(port):
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
            kind = "port"
            cmd = [port, "read2sdbg", "-k", str(K), "-m", str(MIN_COUNT), "--read_lib_file", os.path.join(d, "reads"),
                   "--output_prefix", os.path.join(d, "out")]
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            tried[1] = time.perf_counter() - t0
    best = min(tried, key=tried.get)
    out = {"value": round(E / tried[best] / 1e6, 3), "unit": "M edges/s", "cores": best, "kind": kind, "host_cores": cores,
           "threads_tried": {str(t): round(E / dt / 1e6, 3) for t, dt in tried.items()},
           "sample": "read2sdbg k=%d m=%d on %d synthetic %d bp reads (%.1f M edges), best wall %.1f s incl. file I/O"
                     % (K, MIN_COUNT, reads.shape[0], READ_LEN, E / 1e6, tried[best])}
    try:
        import glob
        with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_cpu_fullsize.json")))[-1]) as f:
            out["full_size"] = json.load(f)
    except Exception:
        pass
    return out


def cpu_full_size(n_reads, threads=8):
    """--cpu-full: the reference's read2sdbg on the whole workload of this run (the library of tools/make_fullsize_golden.py),
    timed here and now on this host, digest compared with the committed known answer"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_fullsize_golden as mfg
    from megahit_amd import canon
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    with tempfile.TemporaryDirectory(prefix="mhx_cpufull_") as d:
        mfg.gen_library(os.path.join(d, "reads"), n_reads)
        cmd = [ref, "read2sdbg", "-k", str(K), "-m", str(MIN_COUNT), "--host_mem", "64e9", "--num_cpu_threads", str(threads),
               "--read_lib_file", os.path.join(d, "reads"), "--output_prefix", os.path.join(d, "out")]
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        digest = canon.digest_sdbg(os.path.join(d, "out"))
    return {"wall_s": round(dt, 1), "threads": threads, "M_edges_per_s": round(n_reads * (READ_LEN - K) / dt / 1e6, 2), "digest": digest,
            "host_cores": os.cpu_count()}


def end_to_end(n_reads):
    """Files in -> files out through the drop-in CLI on the same 10 M-read library (.bin/.lib_info), every process started
    right behind the previous one — no pauses — as the reference's orchestrator starts its sub-programs (src/megahit:771-847):
      read2sdbg (the headline sub-program), twice: cold and warm page cache;
      the orchestrator's default k_min route, back to back: count, then seq2sdbg --need_mercy on count's outputs.
    Wall time = the caller's clock around each process (device memory released before the process returns)."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_fullsize_golden as mfg
    from megahit_amd import canon
    mhx = os.path.join(ROOT, "megahit_amd", "mhx_core")
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
            full = json.load(f)
    except Exception:
        full = None
    known = full is not None and n_reads == full["reads"]

    def call(args):
        t0 = time.perf_counter()
        p = subprocess.run([mhx] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env={k_: v_ for k_, v_ in os.environ.items() if k_ != "MHX_SERVER"} if mhx.endswith("mhx_core") else None)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-500:])
        phases = [(m.group(1).strip(), float(m.group(2))) for m in re.finditer(r"INFO\s+(.*?)\.? Time elapsed: ([0-9.]+)", p.stderr)]
        return dt, {name[:40]: round(sec, 3) for name, sec in phases}

    with tempfile.TemporaryDirectory(prefix="mhx_e2e_") as d:
        mfg.gen_library(os.path.join(d, "reads"), n_reads)
        common = ["-k", str(K), "-m", str(MIN_COUNT), "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file", os.path.join(d, "reads")]
        # Under the reference's name first (megahit_amd/megahit_core -> mhx_core): what an unmodified orchestrator gets.  No
        # environment variable: the first call starts the resident server (one process keeps the handle and its device buffers,
        # INTEGRATION.md), the others find it.  "cold" = the first three sub-programs (server start, handle creation, every
        # buffer allocated for the first time), "steady" = the same three again.
        served = None
        outs = {}
        drop_in = os.path.join(ROOT, "megahit_amd", "megahit_core")
        if not os.path.exists(drop_in):
            os.symlink("mhx_core", drop_in)
        sock = subprocess.run([mhx, "--default-socket"], stdout=subprocess.PIPE, text=True).stdout.strip()
        try:
            os.environ.pop("MHX_SERVER", None)
            mhx_plain, mhx = mhx, drop_in
            served = {"how": "megahit_core <sub-program> ..., no environment variable (resident server by default under this name)"}
            for label in ("cold", "steady"):
                t0 = time.perf_counter()
                s_r2s, ph_r2s = call(["read2sdbg"] + common + ["--output_prefix", os.path.join(d, "sv")])
                s_r2s2, _ = call(["read2sdbg"] + common + ["--output_prefix", os.path.join(d, "sv")])
                s_cnt, ph_sc = call(["count"] + common + ["--output_prefix", os.path.join(d, "scnt")])
                s_s2s, ph_ss = call(["seq2sdbg", "-k", str(K), "--kmer_from", "0", "--host_mem", "64e9", "--num_cpu_threads", "8", "--input_prefix",
                                     os.path.join(d, "scnt"), "--need_mercy", "--output_prefix", os.path.join(d, "ss2m")])
                served[label] = {"read2sdbg_s": round(s_r2s, 3), "read2sdbg_again_s": round(s_r2s2, 3), "count_s": round(s_cnt, 3),
                                 "seq2sdbg_need_mercy_s": round(s_s2s, 3), "default_route_back_to_back_s": round(s_cnt + s_s2s, 3),
                                 "four_sub_programs_back_to_back_s": round(time.perf_counter() - t0, 3),
                                 "phases_read2sdbg_s": ph_r2s, "phases_count_s": ph_sc, "phases_seq2sdbg_s": ph_ss}
            outs = {"sv": canon.digest_sdbg(os.path.join(d, "sv")), "ss2m": canon.digest_sdbg(os.path.join(d, "ss2m"))}
        except Exception as ex:
            served = {"error": str(ex)[-300:]}
        finally:
            mhx = mhx_plain
            try:  # (the server would leave by itself after two idle minutes; the runs below want the device to themselves)
                subprocess.run([mhx, "--serve-stop", sock], timeout=30, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                for _ in range(300):
                    if not os.path.exists(sock):
                        break
                    time.sleep(0.05)
            except Exception:
                pass
        t_all = time.perf_counter()
        r2s = [call(["read2sdbg"] + common + ["--output_prefix", os.path.join(d, "out")]) for _ in range(2)]
        t_cnt, ph_cnt = call(["count"] + common + ["--output_prefix", os.path.join(d, "cnt")])
        t_s2s, ph_s2s = call(["seq2sdbg", "-k", str(K), "--kmer_from", "0", "--host_mem", "64e9", "--num_cpu_threads", "8", "--input_prefix",
                              os.path.join(d, "cnt"), "--need_mercy", "--output_prefix", os.path.join(d, "s2m")])
        t_all = time.perf_counter() - t_all
        digest = canon.digest_sdbg(os.path.join(d, "out"))
        digest_route = canon.digest_sdbg(os.path.join(d, "s2m"))
        if served and "error" not in served:
            served["digests_equal_process_runs"] = outs.get("sv") == digest and outs.get("ss2m") == digest_route
    dt, phases = r2s[1]
    if served and "error" not in served:  # the default way in: under the reference's name
        wall, wall_first = served["steady"]["read2sdbg_again_s"], served["cold"]["read2sdbg_s"]
        what = ("megahit_core read2sdbg (the reference's name for mhx_core: resident server by default, no environment variable): .bin read + H2D + "
                "GPU stages + D2H + .sdbg/.sdbg_info/.counting written; the second of two back to back, steady state; wall_s_first_run = the very "
                "first call, which starts the server")
    else:
        wall, wall_first = dt, r2s[0][0]
        what = "mhx_core read2sdbg as a process of its own (the drop-in name failed: see served)"
    out = {"wall_s": round(wall, 3), "wall_s_first_run": round(wall_first, 3), "M_edges_per_s": round(n_reads * (READ_LEN - K) / wall / 1e6, 1),
           "phases_s": phases, "digest": digest, "what": what,
           "process_runs": {"what": "mhx_core read2sdbg as a process of its own, twice back to back: device initialisation, every buffer allocated "
                                    "(the driver scrubs the memory the process before gave back) and released again",
                            "read2sdbg_s": [round(r2s[0][0], 3), round(dt, 3)]},
           "default_route": {"count_s": round(t_cnt, 3), "seq2sdbg_need_mercy_s": round(t_s2s, 3), "back_to_back_s": round(t_cnt + t_s2s, 3),
                             "phases_count_s": ph_cnt, "phases_seq2sdbg_s": ph_s2s, "digest": digest_route},
           "four_processes_back_to_back_s": round(t_all, 3), "served": served}
    if known:
        out["bit_identical_to_reference"] = digest == full["cases"]["read2sdbg"]["digest"]
        out["reference_wall_s_8_threads_build_container"] = full["cases"]["read2sdbg"]["wall_s"]
        out["default_route"]["bit_identical_to_reference"] = digest_route == full["cases"]["seq2sdbg_need_mercy"]["digest"]
        out["default_route"]["reference_wall_s_8_threads_build_container"] = round(full["cases"]["count"]["wall_s"] + full["cases"]["seq2sdbg_need_mercy"]["wall_s"], 1)
    return out


def output_parity(eng, engine, n_reads, world, res):
    """After the timed region: digest the outputs the LAST step left in HBM and compare them with the reference's
    known answer for this exact workload (tests/golden/fullsize.json: oracle/_ref/ref_core on the same 10 M reads,
    tools/make_fullsize_golden.py).  None when no known answer exists for the configuration."""
    import numpy as np
    from megahit_amd import canon, lib
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
            full = json.load(f)
    except Exception:
        return None
    if world != 1 or n_reads != full["reads"] or K != full["k"] or MIN_COUNT != full["m"]:
        return None
    if engine == "read2sdbg":
        want = full["cases"]["read2sdbg"]
        got = canon.digest_sdbg_buffers(K, eng.fetch(lib.BUF_SDBG_BYTES, np.uint8), eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
                                        eng.fetch(lib.BUF_BUCKET_TIPS, np.uint64), eng.fetch(lib.BUF_BUCKET_LARGE, np.uint64),
                                        eng.fetch(lib.BUF_BUCKET_OFFSET, np.uint64))
        n_got, n_want = int(res[1].n_sdbg), want["n_sdbg"]
    elif engine == "count":
        import hashlib
        want = full["cases"]["count"]
        edges = eng.fetch(lib.BUF_EDGES, np.uint32)
        counts = eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64).astype(np.int64)
        wpe = int(res[0].words_per_edge)
        h = hashlib.md5()
        h.update(("k%d w%d n%d|" % (K, wpe, edges.size // wpe)).encode())
        h.update(counts.tobytes())
        h.update(edges.tobytes())
        got = h.hexdigest()
        n_got, n_want = int(res[0].n_edges), want["n_edges"]
    else:
        return None
    return {"checked": got == want["digest"] and n_got == n_want, "digest": got, "reference_digest": want["digest"],
            "records": n_got, "reference_records": n_want,
            "reference": "oracle/_ref/ref_core (reference sources) on the same reads: tests/golden/fullsize.json"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=float, default=10e6, help="reads per GPU (BASELINE configs[1]: 10 M)")
    ap.add_argument("--cpu-sample-reads", type=float, default=1e6)
    ap.add_argument("--no-e2e", action="store_true", help="skip the files-in -> files-out run of the CLI after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full", dest="cpu_full", action="store_false", help="do not time the reference's CPU path on the FULL workload "
                    "in this very run (~2 min of host time at 10 M reads; the full-size figure quoted is then the one committed under profiles/)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path (RCCL collectives) even with one rank")
    ap.add_argument("--engine", choices=["read2sdbg", "count", "seq2sdbg"], default="read2sdbg",
                    help="sub-program to time; read2sdbg is BASELINE.json's metric, the others are reported beside it (1 GPU)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result of rank 0: libraries that chat on stdout (RCCL's version banner,
    # gloo's connection notes) are sent to stderr by pointing fd 1 there for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # called the way `--gpus 1` is called, without a launcher: become the launcher — one rank per GPU over RCCL,
            # rendezvous on the loopback address (the container's hostname may not resolve), this process's stdout (the
            # one JSON line of rank 0) passed through
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            os.dup2(real_stdout, 1)
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                      "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import numpy as np
    import torch
    from megahit_amd import lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libmhx has no CPU fallback)")
    if world > torch.cuda.device_count():
        raise SystemExit("--gpus %d: this node shows %d GPUs (one rank per GPU: RCCL refuses two ranks on one device)" % (world, torch.cuda.device_count()))
    n_reads = int(args.reads) // 16 * 16
    # The reference's CPU path on the WHOLE workload of this run (8 of this host's cores, ~95 s at 10 M reads) runs at the very
    # END, alone: started beside the read synthesis, the CLI runs and the GPU driver threads (round 4) it was timed under
    # host contention, which flattered the GPU/CPU ratio.
    cpu_full_wanted = rank == 0 and world == 1 and args.cpu_full and not args.no_cpu_baseline and args.engine == "read2sdbg" and \
        os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_core"))
    # Files in -> files out through the CLI, measured FIRST: mhx_core is a process of its own and is meant to find the GPU
    # as a caller finds it (measured after the timed steps, next to this process's ~100 GB of freshly released HBM, its
    # allocations alone took 0.2 s longer).  Reported in the JSON line at the end.
    e2e_result = None
    if world == 1 and not args.no_e2e and args.engine == "read2sdbg":
        try:
            e2e_result = end_to_end(n_reads)
        except Exception as ex:
            e2e_result = {"error": str(ex)}
        log("[rank 0] end to end: %s" % json.dumps(e2e_result)[:300])
    torch.cuda.set_device(local_rank)
    t0 = time.time()
    packed = make_reads(n_reads, rank, world)
    log("[rank %d] generated %d reads in %.1f s" % (rank, n_reads, time.time() - t0))

    eng = lib.Engine(local_rank)
    eng.load_sequences(packed, n_reads, READ_LEN, None)  # H2D happens here, outside the timed region
    E = n_reads * (READ_LEN - K)

    if args.engine != "read2sdbg" and world > 1:
        raise SystemExit("--engine %s is a single-GPU report" % args.engine)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # The data plane is the C++ multi-GPU driver of libmhx (include/mhx.h mhx_comm_* / mhx_dist_*: RCCL called
        # directly, grouped ncclSend/ncclRecv on the engine's stream).  torch.distributed is the control plane only: it
        # carries the RCCL unique id from rank 0 to the others, the barriers around the timed region and the max of the
        # per-rank times (gloo: no second RCCL communicator).
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("gloo")
        ids = [lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        log("[rank %d] mhx_comm_init_rank, world %d" % (rank, world))
        comm = lib.Comm.rccl(eng, ids[0], rank, world)
        comm.setup(1, K, MIN_COUNT)  # bucket ranges balanced by the all-reduced stage-1 bucket histogram
        log("[rank %d] communicator up, partition set" % rank)

        def step():
            r1, r2, _nm = comm.read2sdbg(K, MIN_COUNT)
            return r1, r2

        def barrier():
            eng.synchronize()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    elif args.engine == "count":
        def step():
            return eng.count(K, MIN_COUNT), None
    elif args.engine == "seq2sdbg":
        # the reference's 2-pass route: count, then seq2sdbg over the solid (k+1)-mer edges (untimed: count + reload)
        rc = eng.count(K, MIN_COUNT)
        edges = eng.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, rc.words_per_edge)
        mult = (edges[:, -1] & 0xFFFF).astype(np.uint16)
        n_edges = edges.shape[0]
        from megahit_amd import synth
        chars = np.zeros((n_edges, K + 1), dtype=np.uint8)
        for i in range(K + 1):
            chars[:, i] = (edges[:, i // 16] >> (30 - 2 * (i % 16))) & 3
        eng.load_sequences(synth.pack_reads_concat(chars[: n_edges // 16 * 16]), n_edges // 16 * 16, K + 1, None)
        eng.load_multiplicity(mult[: n_edges // 16 * 16])
        E = n_edges // 16 * 16  # units: input edges

        def step():
            return None, eng.seq2sdbg(K)
    else:
        def step():
            r1 = eng.read2sdbg_s1(K, MIN_COUNT)
            r2 = eng.read2sdbg_s2(K, MIN_COUNT)
            return r1, r2

    if not use_dist:
        def barrier():
            eng.synchronize()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    barrier()
    eng.profile(True)
    eng.profile_reset()
    if use_dist:
        comm.bytes_sent(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    sent_per_step = comm.bytes_sent() / max(1, args.steps) if use_dist else None  # payload this rank handed to OTHER ranks (what crosses xGMI)
    stats = eng.profile_get()
    eng.profile(False)

    parity = output_parity(eng, args.engine, n_reads, world, res) if rank == 0 and world == 1 else None

    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = E * world * args.steps / dt / 1e6
        # dominant kernel by total time
        name, ks = max(stats.items(), key=lambda kv: kv[1]["ms"])
        per_launch_bytes = ks["bytes"] / ks["launches"]
        per_launch_ms = ks["ms"] / ks["launches"]
        achieved = per_launch_bytes / per_launch_ms / 1e6  # GB/s
        traffic, traffic_src = pmc_traffic(name) if n_reads == 10000000 and args.engine == "read2sdbg" and not use_dist else (None, None)
        copy_gbs = copy_bandwidth(torch)
        roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "copy_kernel_GBs": copy_gbs, "frac_of_copy_kernel": round(achieved / copy_gbs, 4) if copy_gbs else None,
                "launches_per_step": ks["launches"] // max(1, args.steps), "avg_launch_ms": round(per_launch_ms, 4),
                "algo_bytes_per_launch": per_launch_bytes,
                # every kernel of a step together (counter file of the running build, else null): what the step moves through HBM
                "traffic_per_step_all_kernels": pmc_traffic_per_step() if n_reads == 10000000 and args.engine == "read2sdbg" and not use_dist else None,
                # the contract's "bound" knows hbm | mfma; the group-by over super-k-mer records is neither: it reads 16 bytes per 3.5 windows and
                # spends its time on LDS compare-and-swaps and the vector instructions that form the keys (profiles/rNN_pmc_sq.json)
                "note": ("dominant kernel is the LDS group-by of stage 1 on super-k-mer records: bound by LDS atomics and vector issue, not by HBM — its "
                         "fraction of the HBM peak says how few bytes it needs, not how well it runs; the HBM-bound kernels of the step are the sort passes "
                         "(top_kernels)") if name in ("s1_skm_groups", "count_skm_groups") else None,
                "kernel_ms_per_step": {k2: round(v["ms"] / args.steps, 3) for k2, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])},
                # the same figure for the three kernels with the largest time per step (they are within a few per cent of each
                # other, and which of them leads changes from box to box): algorithmic GB/s of one launch and its fraction of the peak
                "top_kernels": [{"kernel": k2, "ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] // max(1, args.steps),
                                 "achieved": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else None,
                                 "frac": round(v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS, 4) if v["ms"] > 0 else None,
                                 "traffic": pmc_traffic(k2)[0] if n_reads == 10000000 and args.engine == "read2sdbg" and not use_dist else None}
                                for k2, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:3]]}
        metric = {"read2sdbg": "M (k+1)-mer edges sorted+counted/sec, sdbg_build k=21",
                  "count": "M (k+1)-mer edges sorted+counted/sec, count k=21 (beside the headline metric)",
                  "seq2sdbg": "M input (k+1)-mer edges/sec, seq2sdbg k=21 (beside the headline metric)"}[args.engine]
        workload = {"read2sdbg": "read2sdbg (S1+S2) k=21 m=2 no mercy", "count": "count k=21 m=2",
                    "seq2sdbg": "seq2sdbg k=21 over the solid edges of count"}[args.engine]
        out = {"metric": metric, "value": round(value, 2), "unit": "M edges/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": workload + ", %d synthetic 150 bp PE reads per GPU "
                                      "(BASELINE configs[1]), inputs resident in HBM, outputs left in HBM" % n_reads,
                          "reads_per_gpu": n_reads, "edges_per_gpu": E, "k": K, "min_count": MIN_COUNT,
                          "s1_plan": eng.last_s1_plan() if args.engine == "read2sdbg" else None,
                          "parallelism": "1 GPU" if not use_dist else
                          "lv1 buckets over %d GPUs (C++ driver, RCCL ncclSend/ncclRecv all-to-all, marks routed to the read owners)" % world},
               "roofline": roof,
               "parity_checked": bool(parity["checked"]) if parity else None, "parity": parity}
        if use_dist:
            out["config"]["rank0_bytes_sent_to_other_ranks_per_step"] = int(sent_per_step)
        try:
            from megahit_amd.buildid import build_id, lib_id
            out["build_id"], out["lib_id"] = build_id(), lib_id()
        except Exception:
            pass
        out["tuning"] = tuned_defaults()  # megahit_amd/mhx_tuning.conf: the knob defaults this run (and the CLI) used
        if use_dist:
            out["config"]["rank0_s1_items"] = int(res[0].n_items)
            out["config"]["rank0_s2_items"] = int(res[1].n_items)
            out["config"]["rank0_sdbg_records"] = int(res[1].n_sdbg)
        if not use_dist:
            r1, r2 = res
            if args.engine == "read2sdbg":
                out["config"]["s1_items"] = int(r1.n_items)
            if args.engine == "count":
                out["config"]["items"] = int(r1.n_items)
                out["config"]["solid_edges"] = int(r1.n_edges)
            if r2 is not None:
                out["config"]["s2_items" if args.engine == "read2sdbg" else "items"] = int(r2.n_items)
                out["config"]["sdbg_records"] = int(r2.n_sdbg)
            if not args.no_cpu_baseline and args.engine == "read2sdbg":
                try:
                    out["cpu_baseline"] = cpu_baseline(int(args.cpu_sample_reads) // 2 * 2)
                    if cpu_full_wanted:
                        eng.close()  # (nothing of this process competes with it: HBM released, no GPU work in flight)
                        load_before = os.getloadavg()[0]
                        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_fullsize.py"), "--threads", "8", "--reads", str(n_reads)],
                                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
                        full_now = json.loads(proc.stdout.decode())
                        full_now["how"] = "tools/cpu_fullsize.py --threads 8, run alone after the GPU work of this script (host load average before it: %.1f)" % load_before
                        out["cpu_baseline"]["full_size_this_run"] = full_now
                        # the headline CPU figure is the reference on the WHOLE workload, same host, same run (digest
                        # checked against the known answer) — the bounded 1 M-read sample stays beside it
                        want_digest = None
                        try:
                            with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as gf:
                                g = json.load(gf)
                            want_digest = g["cases"]["read2sdbg"]["digest"] if g["reads"] == n_reads else None
                        except Exception:
                            pass
                        run8 = full_now.get("runs", {}).get(str(full_now.get("best_threads", 8)))
                        if run8 and full_now.get("best_M_edges_per_s") and (want_digest is None or run8.get("digest") == want_digest):
                            cb = out["cpu_baseline"]
                            cb["bounded_sample"] = {"value": cb["value"], "cores": cb["cores"], "sample": cb["sample"], "threads_tried": cb.get("threads_tried")}
                            cb["value"] = full_now["best_M_edges_per_s"]
                            cb["cores"] = int(full_now["best_threads"])
                            cb["digest_equals_reference_golden"] = want_digest is not None
                            cb["sample"] = ("the WHOLE workload: the reference's read2sdbg k=%d m=%d on the %d reads of this run, %d threads (its best on this host: "
                                            "bounded_sample.threads_tried), %.1f s wall incl. file I/O, run alone after the GPU work"
                                            % (K, MIN_COUNT, n_reads, cb["cores"], full_now["best_wall_s"]))
                except Exception as ex:  # the baseline is reporting only; never lose the GPU number
                    out["cpu_baseline"] = {"value": None, "error": str(ex)}
            if e2e_result is not None:
                out["e2e"] = e2e_result
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        import torch.distributed as dist
        comm.close()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
