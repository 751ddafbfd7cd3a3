"""GPU: SURVEY.md section 8f N1 — the device-resident SdBG hand-over (mhx_sdbg_build_index) against the REFERENCE's own
loader.  oracle/_ref/ref_sdbg_dump = SDBG::LoadFromFile (LoadSdbgRawContent + kmlib::RankAndSelect construction,
reference src/sdbg/sdbg.h:26-61, sdbg_raw_content.cpp:18-96, kmlib/kmrns.h:118-175) compiled from the reference
sources, dumping every array it builds; the device buffers must equal them word for word."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu
from megahit_amd import canon, lib

REF_DUMP = os.path.join(gu.ROOT, "oracle", "_ref", "ref_sdbg_dump")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="oracle/_ref/ref_sdbg_dump not built")]

DT = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            hdr = f.read(48)
            if len(hdr) < 48:
                break
            name = hdr[:32].split(b"\0", 1)[0].decode()
            elem, n = struct.unpack("<QQ", hdr[32:])
            out[name] = np.frombuffer(f.read(elem * n), dtype=DT[elem])
    return out


def load_files_into(engine, prefix):
    """.sdbg_info + .sdbg.* -> the handle's current SdBG (bucket byte ranges back to back, bucket-id order)"""
    hdr, buckets = canon.canonical_sdbg(prefix)
    off = np.zeros(65536, dtype=np.uint64)
    items, tips, large = off.copy(), off.copy(), off.copy()
    parts, pos = [], 0
    for bid, ni, nt, nl, b in buckets:
        off[bid], items[bid], tips[bid], large[bid] = pos, ni, nt, nl
        parts.append(b)
        pos += len(b)
    data = np.frombuffer(b"".join(parts), dtype=np.uint8) if parts else np.zeros(0, dtype=np.uint8)
    engine.sdbg_load_bytes(data, off, items, tips, large)
    return hdr["k"]


def check_index(engine, k, want):
    info = engine.sdbg_build_index(k)
    meta = want["meta"]
    assert (info.n_items, info.n_tips, info.n_large, info.k, info.words_per_tip_label, info.use_full_mul) == tuple(int(x) for x in meta)
    for name, buf, dt in (("w", lib.BUF_SDBG_W, np.uint64), ("last", lib.BUF_SDBG_LAST, np.uint64), ("tip", lib.BUF_SDBG_TIP, np.uint64),
                          ("invalid", lib.BUF_SDBG_INVALID, np.uint64), ("mul", lib.BUF_SDBG_MUL, np.uint16),
                          ("small_mul", lib.BUF_SDBG_SMALL_MUL, np.uint8), ("tip_labels", lib.BUF_SDBG_TIP_LABELS, np.uint32),
                          ("prefix_lkt", lib.BUF_SDBG_PREFIX_LKT, np.uint64)):
        got = engine.fetch(buf, dt)
        assert got.size == want[name].size, name
        assert np.array_equal(got, want[name]), name
    assert np.array_equal(np.array(list(info.f), dtype=np.int64).view(np.uint64), want["f"])
    assert np.array_equal(np.array(list(info.rank_f), dtype=np.int64).view(np.uint64), want["rank_f"])
    # rank/select over W: nine characters
    l2 = engine.fetch(lib.BUF_SDBG_RS_W_L2, np.uint64).reshape(9, -1)
    l1 = engine.fetch(lib.BUF_SDBG_RS_W_L1, np.uint16).reshape(9, -1)
    sel = engine.fetch(lib.BUF_SDBG_RS_W_SEL, np.uint32)
    assert l2.shape[1] == info.num_l2_w and l1.shape[1] == info.num_l1_w
    assert np.array_equal(np.array(list(info.w_char_count), dtype=np.uint64), want["rsw_count"])
    for c in range(9):
        assert np.array_equal(l2[c], want["rsw_l2_%d" % c]), c
        assert np.array_equal(l1[c], want["rsw_l1_%d" % c]), c
        assert np.array_equal(sel[info.w_sel_offset[c]:info.w_sel_offset[c + 1]], want["rsw_sel_%d" % c]), c
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_RS_LAST_L2, np.uint64), want["rslast_l2_1"])
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_RS_LAST_L1, np.uint16), want["rslast_l1_1"])
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_RS_LAST_SEL, np.uint32), want["rslast_sel_1"])
    assert info.ones_in_last == int(want["rslast_count"][0]) and info.ones_in_tip == int(want["rstip_count"][0])
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_RS_TIP_L2, np.uint64), want["rstip_l2_1"])
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_RS_TIP_L1, np.uint16), want["rstip_l1_1"])
    return info


def _cases():
    seen, out = set(), []
    for e in gu.cases():
        c = e["case"]
        key = (c["prog"], c.get("lib"), c["k"], bool(c.get("mercy")), c.get("input"))
        if c["prog"] in ("read2sdbg", "seq2sdbg") and key not in seen:
            seen.add(key)
            out.append(e)
    return out


@pytest.mark.parametrize("ent", _cases(), ids=gu.case_id)
def test_index_equals_reference_loader(engine, ent, tmp_path):
    gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    prefix = os.path.join(str(tmp_path), "out")
    dump = os.path.join(str(tmp_path), "ref.dump")
    subprocess.run([REF_DUMP, prefix, dump], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = read_dump(dump)
    k = load_files_into(engine, prefix)
    info = check_index(engine, k, want)
    assert info.n_items == ent["n_sdbg"]


def test_index_straight_from_stage2_buffers(engine, tmp_path):
    """no files in between: stage 2's byte stream in HBM -> index; the reference loader reads the CLI's files of the same run"""
    import oracle_binding as ob
    from megahit_amd import synth
    from test_gpu_count import load
    reads = synth.gen_pe_reads(4000, 20000, read_len=100, frag=250, err=0.01, seed=5)
    synth.write_read_lib(os.path.join(str(tmp_path), "reads"), [reads])
    prefix = os.path.join(str(tmp_path), "out")
    subprocess.run([gu.MHX_CORE, "read2sdbg", "-k", "21", "-m", "2", "--host_mem", "2e9", "--num_cpu_threads", "3", "--read_lib_file",
                    os.path.join(str(tmp_path), "reads"), "--output_prefix", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dump = os.path.join(str(tmp_path), "ref.dump")
    subprocess.run([REF_DUMP, prefix, dump], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    pkg = ob.Package([r for r in reads], reverse=True)
    load(engine, pkg)
    engine.read2sdbg_s1(21, 2)
    engine.read2sdbg_s2(21, 2)
    check_index(engine, 21, read_dump(dump))


def _tip_cases():
    out = [e for e in _cases() if e["case"]["prog"] == "read2sdbg"][:6]
    out += [e for e in _cases() if e["case"]["prog"] == "seq2sdbg"][:4]
    seen = {(e["case"]["prog"], e["case"]["k"]) for e in out}
    for e in _cases():  # one graph per further k: multi-word tip labels, max_tip_len up to 238
        key = (e["case"]["prog"], e["case"]["k"])
        if key not in seen:
            seen.add(key)
            out.append(e)
    return out


@pytest.mark.parametrize("ent", _tip_cases(), ids=gu.case_id)
def test_tip_trimming_equals_reference(engine, ent, tmp_path):
    """mhx_sdbg_remove_tips on the device-resident graph = sdbg_pruning::RemoveTips
    (assembly/sdbg_pruning.cpp:61-179) of the reference, run by the dumper on the same files with max_tip_len = 2k
    (main_assemble.cpp:143-156) — same number of tips, same invalid bit vector."""
    gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    prefix = os.path.join(str(tmp_path), "out")
    dump = os.path.join(str(tmp_path), "ref.dump")
    k = ent["case"]["k"]
    subprocess.run([REF_DUMP, prefix, dump, str(2 * k)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = read_dump(dump)
    load_files_into(engine, prefix)
    info = engine.sdbg_build_index(k)
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_INVALID, np.uint64), want["invalid"])
    n = engine.sdbg_remove_tips(info, 2 * k)
    assert n == int(want["tips_removed"][0])
    assert np.array_equal(engine.fetch(lib.BUF_SDBG_INVALID, np.uint64), want["invalid_after_tips"])
