"""CPU-only tests of the host logic of the product: the TAPE3 reader of the HIP library (C++, no GPU needed)
against the oracle's loader and against hand-built files - block skip / stop rules, coupling records,
malformed files."""
import os
import struct

import numpy as np
import pytest

from common import Golden, golden_names
from monortm_amd import api, synth, tape3
from oracle.pyoracle import Oracle


@pytest.mark.parametrize("name", golden_names())
def test_reader_agrees_with_oracle_loader(name, workdir):
    g = Golden(name, workdir)
    v1, v2 = g.profiles[0].wn[0], g.profiles[0].wn[-1]
    nphys, nent, ncpl = api.tape3_probe(g.tape3, v1, v2)
    orc = Oracle(g.tape3, v1, v2)
    rec = tape3.read_tape3(g.tape3)
    for m in range(1, 40):
        # NBLM(m) of the reference = physical lines + coupling records of the kept blocks
        assert orc.nlines(m) >= nphys[m]
        assert nent[m] <= orc.nlines(m)
    assert nphys[0] == nphys[1:].sum() and nphys[0] <= rec.n_physical
    assert ncpl[0] == ncpl[1:].sum()


def test_block_skip_and_stop(workdir):
    """Blocks entirely below v1-25 are skipped, reading stops after the first block whose last record lies above
    v2+25 (src/lnfl_mod.f90:116,:161-165): lines outside the window survive only inside kept blocks."""
    n = 900
    rec = synth.synthetic_lines(n, seed=12, vlo=0.05, vhi=54.9)
    path = os.path.join(workdir, "TAPE3_blocks")
    tape3.write_tape3(path, rec, split_blocks_at=[150, 300, 450, 600, 750])
    vnu = rec.vnu
    bounds = [0, 150, 300, 450, 600, 750, n]
    for v1, v2, expect_skip, expect_stop in ((40.0, 40.5, True, False), (3.0, 4.0, False, True), (35.0, 36.0, True, False)):
        keep = np.zeros(n, bool)
        skipped = stopped = False
        for a, b in zip(bounds[:-1], bounds[1:]):
            if vnu[a:b].max() < max(0.0, v1 - 25.0):
                skipped = True
                continue
            keep[a:b] = True
            if vnu[b - 1] > v2 + 25.0:
                stopped = b < n
                break
        assert skipped == expect_skip and stopped == expect_stop
        nphys, nent, _ = api.tape3_probe(path, v1, v2)
        assert nphys[0] == keep.sum() and nent[0] == keep.sum()
        assert 0 < keep.sum() < n
        for m in np.unique(rec.mol % 100):
            assert nphys[m] == np.count_nonzero(keep & (rec.mol % 100 == m))
    # whole file for a window that touches everything
    assert api.tape3_probe(path, 20.0, 35.0)[0][0] == n


def test_coupling_records_are_paired_not_counted(workdir):
    rec = synth.synthetic_lines(200, seed=5, lc_frac=1.0)  # every O2 line carries an IFLG=-1 record
    path = os.path.join(workdir, "TAPE3_lc")
    tape3.write_tape3(path, rec)
    nphys, nent, ncpl = api.tape3_probe(path, 0.3, 30.0)
    n_o2 = int(np.count_nonzero((rec.iflg >= 0) & (rec.mol % 100 == 7)))
    assert nphys[7] == n_o2 and nent[7] == n_o2 and ncpl[7] == n_o2 and ncpl[0] == n_o2
    assert nphys[0] == rec.n_physical


def test_real_like_file_layout(workdir):
    """The line file of the real_like fixture as a real LNFL product lays it out: second header record
    announced by '^' in HLINID(7)(8:8) (src/lnfl_mod.f90:258-262), 23 blocks with short ones in the middle, isotopologues > 1,
    molecules beyond NMOL - and one coupling record as the FIRST record of a block.  The parser must count what the oracle's
    loader (pinned to the compiled reference by the fixture's outputs) counts, and file the slot-1 record where the
    reference's rule puts it: molecule 4 in the "dbl" build (its entry count = its lines + 1)."""
    g = Golden("real_like", workdir)
    raw = open(g.tape3, "rb").read()
    assert raw[4 + 6 * 8 + 7: 4 + 6 * 8 + 8] == b"^"
    # walk the records: header, second header (64 + 64 INTEGER*4 + 4096 REAL*4), then (panel header, block) pairs
    pos, sizes = 0, []
    while pos < len(raw):
        (m,) = struct.unpack_from("<i", raw, pos)
        sizes.append(m)
        pos += m + 8
    assert sizes[0] == 1664 and sizes[1] == 4 * (64 + 64 + 4096)
    nrecs = [struct.unpack_from("<ddii", raw, sum(x + 8 for x in sizes[:k]) + 4)[2] for k in range(2, len(sizes), 2)]
    assert len(nrecs) >= 20 and min(nrecs[1:-1]) < 250 and nrecs.count(250) >= 15, nrecs
    rec = tape3.read_tape3(g.tape3)
    assert len(rec) == sum(nrecs)
    starts = np.cumsum([0] + nrecs[:-1])
    slot1 = [int(s) for s in starts if rec.iflg[s] < 0]
    assert len(slot1) == 1 and rec.iflg[slot1[0] - 1] == 1 and rec.mol[slot1[0] - 1] % 100 == 7   # its line ends the block before
    blk = list(starts).index(slot1[0])
    assert nrecs[blk] == 250
    e250 = float(rec.epp[slot1[0] + 249])
    assert synth.slot1_owner(e250, 8) == 4 and synth.slot1_owner(e250, 4) == 24
    phys = rec.iflg >= 0
    iso = (rec.mol[phys] % 1000) // 100
    assert set(np.unique(iso)) >= {1, 2, 3, 4, 5} and (rec.mol[phys] % 100 > 7).sum() > 100
    v1, v2 = 0.3, 54.9     # a window that keeps every block (the fixture's own channels end at 7.6 cm-1: reading stops at 32.6)
    nphys, nent, ncpl = api.tape3_probe(g.tape3, v1, v2)
    orc = Oracle(g.tape3, v1, v2)
    for m in range(1, 40):
        want = int(np.count_nonzero(phys & (rec.mol % 100 == m)))
        assert nphys[m] == want, (m, nphys[m], want)
        # NBLM(m) of the reference: lines + coupling records filed under m
        lc_m = int(np.count_nonzero((rec.iflg < 0)[1:] & (rec.mol[:-1] % 100 == m) & (rec.iflg[:-1] > 0)))   # pairs inside one block
        if m == 7:
            lc_m -= 1      # the slot-1 record is NOT filed under O2 ...
        if m == 4:
            lc_m += 1      # ... but under N2O ("dbl": MOD(bits of REAL*8(epp(250)), 100) = 4)
        assert orc.nlines(m) == want + lc_m, (m, orc.nlines(m), want, lc_m)
    orc.close()


def test_slot1_coupling_record_without_a_valid_owner_is_refused(workdir):
    """The same layout with a 250th lower-state energy whose bits name no molecule: the reference would index NBLM(0) / ISO(0,.)
    (src/lnfl_mod.f90:65-67) - memory outside its tables.  Refused with a format error that says why."""
    rec = synth.synthetic_lines(600, seed=8, lc_frac=1.0)
    i = int(np.flatnonzero(rec.iflg < 0)[5])
    assert i + 250 < len(rec)
    x = np.float32(rec.epp[i + 249])
    while synth.slot1_owner(float(x), 8) != 0:
        x = np.nextafter(x, np.float32(np.inf))
    rec.epp[i + 249] = x
    if rec.iflg[i + 249] < 0:      # (the 250th record may be a coupling record: its EPP field is G(250 K) - any REAL*4 will do)
        pass
    path = os.path.join(workdir, "TAPE3_slot1_bad")
    tape3.write_tape3(path, rec, split_blocks_at=[i, i + 250])
    with pytest.raises(api.MonoRTMError) as e:
        api.tape3_probe(path, 0.3, 30.0)
    assert e.value.code == 2 and "first record of a block" in str(e.value)
    with pytest.raises(Exception):
        Oracle(path, 0.3, 30.0)


def test_malformed_files(workdir):
    missing = os.path.join(workdir, "nope")
    with pytest.raises(api.MonoRTMError) as e:
        api.tape3_probe(missing, 1, 2)
    assert e.value.code == 1
    # header without the isotope tag 'I' (reference: STOP ' PRLNHD - NO ISOTOPE INFO ON LINFIL ', lnfl_mod.f90:297-302)
    good = os.path.join(workdir, "TAPE3_good")
    tape3.write_tape3(good, synth.synthetic_lines(10))
    b = bytearray(open(good, "rb").read())
    b[4 + 9 * 8 + 7] = ord("X")
    bad = os.path.join(workdir, "TAPE3_noI")
    open(bad, "wb").write(bytes(b))
    with pytest.raises(api.MonoRTMError) as e:
        api.tape3_probe(bad, 1, 2)
    assert e.value.code == 2
    # truncated block
    trunc = os.path.join(workdir, "TAPE3_trunc")
    open(trunc, "wb").write(open(good, "rb").read()[:-100])
    with pytest.raises(api.MonoRTMError) as e:
        api.tape3_probe(trunc, 1, 2)
    assert e.value.code == 2
    # header only: a valid, empty line list
    empty = os.path.join(workdir, "TAPE3_empty")
    open(empty, "wb").write(open(good, "rb").read()[: 1664 + 8])
    assert api.tape3_probe(empty, 1, 2)[0][0] == 0
    # unknown coupling flag (reference: 'LC flag not recognized', lnfl_mod.f90:60-62)
    rec = synth.synthetic_lines(10)
    rec.iflg[3] = -2
    rec.iflg[4] = -7
    flg = os.path.join(workdir, "TAPE3_flag")
    tape3.write_tape3(flg, rec)
    with pytest.raises(api.MonoRTMError) as e:
        api.tape3_probe(flg, 0.3, 30)
    assert e.value.code == 2


def test_host_parser_under_sanitizers(workdir):
    """The host TAPE3 reader (plain C++, no HIP) compiled with AddressSanitizer + UBSan parses every golden line file, a
    truncated copy of each, and files of random bytes without a memory error (errors must come back as status codes)."""
    import glob
    import subprocess

    from common import GOLDEN_DIR, ROOT

    src = os.path.join(ROOT, "monortm_amd", "csrc", "line_table.cpp")
    main = os.path.join(workdir, "asan_main.cpp")
    with open(main, "w") as f:
        f.write('#include "line_table.hpp"\n#include <cstdio>\n#include <cstdlib>\n'
                "int main(int argc, char **argv) {\n"
                "    for (int i = 1; i + 2 < argc; i += 3) {\n"
                "        monortm::LineTable t; std::string err;\n"
                "        int rc = monortm::load_tape3(argv[i], atof(argv[i + 1]), atof(argv[i + 2]), t, err);\n"
                '        printf("%d %zu\\n", rc, t.size());\n'
                "    }\n    return 0;\n}\n")
    exe = os.path.join(workdir, "asan_parser")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "monortm_amd", "csrc"), main, src, "-o", exe])
    args = []
    rng = np.random.default_rng(5)
    for k, p in enumerate(sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))):
        z = np.load(p)
        raw = z["tape3"].tobytes()
        full = os.path.join(workdir, f"asan_t3_{k}")
        open(full, "wb").write(raw)
        cut = os.path.join(workdir, f"asan_t3_{k}_cut")
        open(cut, "wb").write(raw[: int(rng.integers(8, len(raw)))])
        args += [full, "0.5", "60.0", cut, "0.5", "60.0", full, "900.0", "60000.0"]
    junk = os.path.join(workdir, "asan_junk")
    open(junk, "wb").write(rng.integers(0, 256, 5000, dtype=np.uint8).tobytes())
    args += [junk, "1.0", "2.0", os.path.join(workdir, "asan_missing"), "1.0", "2.0"]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert len(r.stdout.splitlines()) == len(args) // 3
