"""cstable format: the product's writer/parser, the oracle's readers and the
reference's own cstable library (oracle/_ref, when built) must agree.

Value formulas follow the reference's own round-trip test,
src/eventql/io/cstable/cstable_test.cc:587-755 (i, i%2==0, i*1.1, i+12,
"x{i}x", i*5, i*8 over 131072 rows, all seven encodings)."""
import hashlib
import os

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
import oracle_lib as O
import tables as T

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


def _formula_columns(n):
    i = np.arange(n, dtype=np.uint64)
    return dict(
        leb=i, boolean=(i % 2 == 0).astype(np.uint64), flt=i.astype(np.float64) * 1.1,
        u32=i + np.uint64(12), s=[b"x%dx" % k for k in range(n)], u64=i * np.uint64(5),
        bp=i * np.uint64(8))


_FORMULA_SPECS = [
    dict(name="leb", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
    dict(name="boolean", logical_type=K.COL_BOOLEAN, storage_type=K.ENC_BOOLEAN_BITPACKED),
    dict(name="flt", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
    dict(name="u32", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_PLAIN),
    dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
    dict(name="u64", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
    dict(name="bp", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED),
]


def _write_formula_table(path, n):
    c = _formula_columns(n)
    w = E.Writer(_FORMULA_SPECS)
    for spec in _FORMULA_SPECS:
        w.put(spec["name"], c[spec["name"]])
    w.commit(n)
    w.write_file(path)
    w.close()
    return c


def _check_reader(path, which, c, n):
    r = O.TableReader(path, which)
    try:
        assert r.num_rows == n
        for name in ("leb", "boolean", "u32", "u64", "bp"):
            _, _, pr, v = r.read(name, n, "uint")
            assert pr.all() and (v == c[name]).all(), (which, name)
        _, _, pr, v = r.read("flt", n, "float")
        assert (v == c["flt"]).all()
        _, _, pr, v = r.read("s", n, "string")
        assert v == c["s"]
    finally:
        r.close()


@pytest.mark.parametrize("n", [131072, 131072 + 1, 300000, 1, 127, 128, 129])
def test_writer_read_back_by_oracle(tmp_path, built, n):
    path = str(tmp_path / "t.cst")
    c = _write_formula_table(path, n)
    _check_reader(path, "orc", c, n)


@needs_ref
@pytest.mark.parametrize("n", [131072 + 1, 300000, 129])
def test_writer_read_back_by_reference_reader(tmp_path, built, n):
    """files produced by the product's writer are read identically by the
    reference's CSTableReader (cstable_reader.cc:133-200)"""
    path = str(tmp_path / "t.cst")
    c = _write_formula_table(path, n)
    _check_reader(path, "ref", c, n)


def _ref_schema(specs):
    return [dict(name=s["name"], type=s["logical_type"], encoding=s["storage_type"],
                 repeated=0, optional=1 if s.get("dlevel_max", 0) else 0, parent=-1)
            for s in specs]


@needs_ref
@pytest.mark.parametrize("n", [1000, 131072 + 1, 200000])
def test_reference_writer_read_back_by_oracle(tmp_path, built, n):
    """files produced by the REFERENCE writer decode identically through the
    oracle's restatement of the readers"""
    path = str(tmp_path / "r.cst")
    c = _formula_columns(n)
    kinds = dict(leb="uint", boolean="uint", flt="float", u32="uint", s="string", u64="uint",
                 bp="uint")
    O.ref_write_table(path, _ref_schema(_FORMULA_SPECS),
                      [(s["name"], kinds[s["name"]], c[s["name"]], None, None, None)
                       for s in _FORMULA_SPECS], n)
    _check_reader(path, "orc", c, n)
    _check_reader(path, "ref", c, n)


@needs_ref
def test_byte_identical_with_reference_writer(tmp_path, built):
    """same schema order, same column-at-a-time append order => identical bytes
    (page allocation order, index, metablock checksum)"""
    n = 140000
    c = _formula_columns(n)
    ref_path = str(tmp_path / "ref.cst")
    kinds = dict(leb="uint", boolean="uint", flt="float", u32="uint", s="string", u64="uint",
                 bp="uint")
    O.ref_write_table(ref_path, _ref_schema(_FORMULA_SPECS),
                      [(s["name"], kinds[s["name"]], c[s["name"]], None, None, None)
                       for s in _FORMULA_SPECS], n)
    # the reference orders header columns by its schema map; mirror that order
    rr = O.TableReader(ref_path, "ref")
    hdr = rr.columns()
    rr.close()
    by_name = {s["name"]: s for s in _FORMULA_SPECS}
    specs = []
    for h in hdr:
        s = dict(by_name[h["name"]])
        s["column_id"] = h["column_id"]
        specs.append(s)
    w = E.Writer(specs)
    for s in _FORMULA_SPECS:  # append order = the order used with the reference writer
        w.put(s["name"], c[s["name"]])
    w.commit(n)
    mine = w.image()
    w.close()
    ref = open(ref_path, "rb").read()
    assert len(mine) == len(ref)
    assert hashlib.sha1(mine).hexdigest() == hashlib.sha1(ref).hexdigest()


def test_nullable_and_narrow_bitpacking(tmp_path, built):
    img, c = T.mixed_table(300_000)
    path = str(tmp_path / "m.cst")
    open(path, "wb").write(img)
    n = 300_000
    readers = ["orc"] + (["ref"] if O.have_ref() else [])
    for which in readers:
        r = O.TableReader(path, which)
        for name in ("k", "a", "b", "p", "f", "k10", "t", "w"):
            _, _, pr, v = r.read(name, n, "uint")
            assert pr.all() and (v == c[name]).all(), (which, name)
        for name in ("n", "nb"):
            _, dl, pr, v = r.read(name, n, "uint")
            m = c[name + "_present"] == 1
            assert (pr == c[name + "_present"]).all()
            assert (v[m] == c[name][m]).all() and (v[~m] == 0).all()
            assert (dl == c[name + "_present"]).all()
        _, _, pr, v = r.read("nv", n, "float")
        m = c["nv_present"] == 1
        assert (pr == c["nv_present"]).all() and (v[m] == c["nv"][m]).all() and (v[~m] == 0).all()
        _, _, pr, v = r.read("ns", n, "string")
        assert (pr == c["ns_present"]).all()
        assert all(v[i] == (c["ns"][i] if c["ns_present"][i] else b"") for i in range(0, n, 37))
        r.close()


def test_v010_fixture_oracle_matches_reference_fixture(built):
    """test/sql_testdata/testtbl.cst (cstable v0.1.0): header + the `time`
    column against test/sql/00001_*.result.txt"""
    path = os.path.join(T.GOLDEN, "testtbl.cst")
    r = O.TableReader(path, "orc")
    assert r.num_rows == 213
    cols = {c["name"]: c for c in r.columns()}
    assert len(cols) == 63
    assert cols["time"]["rlevel_max"] == 0 and cols["time"]["dlevel_max"] == 1
    assert cols["event.search_query.time"]["rlevel_max"] == 1
    assert cols["event.search_query.time"]["dlevel_max"] == 3
    assert cols["event.search_query.result_items.position"]["rlevel_max"] == 2
    assert cols["event.search_query.result_items.position"]["dlevel_max"] == 4
    _, _, pr, v = r.read("time", 213, "uint")
    exp = open(os.path.join(
        T.GOLDEN, "00001_test_column_reference_with_table_name_prefix.result.txt")
    ).read().split("\n")
    assert exp[0] == "testtable.time"
    assert [str(x) for x in v] == [x for x in exp[1:] if x]
    r.close()


@needs_ref
def test_v010_fixture_oracle_equals_reference_reader(built):
    path = os.path.join(T.GOLDEN, "testtbl.cst")
    ro, rr = O.TableReader(path, "orc"), O.TableReader(path, "ref")
    L = O.oracle()
    for c in ro.columns():
        nvals = L.orc_table_column_num_values(ro.h, c["name"].encode())
        kind = {K.COL_STRING: "string", K.COL_FLOAT: "float"}.get(c["logical_type"], "uint")
        a = ro.read(c["name"], nvals, kind)
        b = rr.read(c["name"], nvals, kind)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), c["name"]
        if kind == "string":
            assert a[3] == b[3], c["name"]
        else:
            assert (a[3] == b[3]).all(), c["name"]
    ro.close()
    rr.close()


def test_product_upgrade_of_v010_fixture_keeps_every_triple(built, tmp_path):
    """evql_cstable_upgrade (what evql_table_open_* apply to v0.1.0 files): all 63
    columns of the reference fixture keep their (rlevel, dlevel, value) streams,
    storage types and level maxima, read back by the oracle (and, when built, by
    the reference's own v0.2.0 reader)"""
    path = os.path.join(T.GOLDEN, "testtbl.cst")
    v2 = E.upgrade_image(open(path, "rb").read())
    assert v2[:6] == b"\x23\x17\x23\x17\x02\x00"
    p2 = str(tmp_path / "testtbl_v2.cst")
    open(p2, "wb").write(v2)
    readers = [O.TableReader(p2, "orc")] + ([O.TableReader(p2, "ref")] if O.have_ref() else [])
    r1 = O.TableReader(path, "orc")
    L = O.oracle()
    for r2 in readers:
        assert r2.num_rows == r1.num_rows == 213
        c2 = {c["name"]: c for c in r2.columns()}
        for c in r1.columns():
            d = c2[c["name"]]
            for f in ("logical_type", "storage_type", "rlevel_max", "dlevel_max"):
                assert d[f] == c[f], (c["name"], f)
            nvals = L.orc_table_column_num_values(r1.h, c["name"].encode())
            kind = {K.COL_STRING: "string", K.COL_FLOAT: "float"}.get(c["logical_type"], "uint")
            a = r1.read(c["name"], nvals, kind)
            b = r2.read(c["name"], nvals, kind)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), c["name"]
            if kind == "string":
                assert a[3] == b[3], c["name"]
            else:
                assert (a[3] == b[3]).all(), c["name"]
        r2.close()
    r1.close()
    with pytest.raises(E.EvqlError):
        E.upgrade_image(v2)  # already v0.2.0
    with pytest.raises(E.EvqlError):
        E.upgrade_image(open(path, "rb").read()[:4000])  # truncated body


def test_sha1_known_answers(built):
    # FIPS 180 / RFC 3174 vectors
    assert O.sha1(b"abc").hex() == "a9993e364706816aba3e25717850c26c9cd0d89d"
    assert O.sha1(b"").hex() == "da39a3ee5e6b4b0d3255bfef95601890afd80709"
    assert O.sha1(b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq").hex() == \
        "84983e441c3bd26ebaae4aa1f95129e5e54670f1"
    data = bytes(range(256)) * 9
    assert O.sha1(data).hex() == hashlib.sha1(data).hexdigest()
    if O.have_ref():
        assert O.sha1(data, "ref") == O.sha1(data)


def test_corrupt_files_rejected(tmp_path, built):
    L = O.oracle()
    assert not L.orc_table_open_image(b"nope" * 200, 800)
    img, _ = T.mixed_table(300_000)
    bad = bytearray(img[:4096])
    bad[14 + 48] ^= 0xFF  # break the only valid metablock's checksum input
    assert not L.orc_table_open_image(bytes(bad), len(bad))


def test_product_parser_rejects_truncated_and_corrupt_images(built):
    """parse_cstable must not trust the file.  Truncations at every
    structural boundary, wrapped u64 offsets, wrong page geometry and too few pages
    for num_rows all answer EVQL_EIO (the reference: "end of column reached" /
    "invalid file"); nothing may crash -- this also runs under the ASan build
    (EVQL_LIB=eventql_amd/libevql_asan.so)."""
    import struct
    import numpy as np
    import eventql_amd as E
    n = 140_000  # > one bit-packed page (131072 values), > two PLAIN64 pages
    w = E.Writer([
        dict(name="u64", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="bp", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED,
             bitpack_max_value=1023),
        dict(name="nl", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128,
             dlevel_max=1),
        dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN)])
    i = np.arange(n, dtype=np.uint64)
    w.put("u64", i)
    w.put("bp", i % np.uint64(1000))
    w.put("nl", i, present=(i % np.uint64(3) != 0).astype(np.uint8))
    w.put("s", [b"x%dx" % k for k in range(n)])
    w.commit(n)
    img = bytes(w.image())
    w.close()
    assert E.inspect_image(img) == (n, 4)

    def rejected(b):
        with pytest.raises(E.EvqlError) as ei:
            E.inspect_image(b)
        assert ei.value.code == K.EVQL_EIO, ei.value
        return ei.value.msg

    # truncations: inside the header, the metablocks, the page area, the index
    for cut in (0, 5, 13, 40, 300, 511, 512, 4096, len(img) // 2, len(img) - 1):
        rejected(img[:cut])
    # metablock: txid / num_rows / index_offset / index_size live in the first 28 bytes
    # of a 48-byte block sealed by SHA1: flipping any of them invalidates the block
    mb = 14
    flipped = bytearray(img)
    flipped[mb + 8] ^= 0xff       # num_rows, slot 0
    flipped[mb + 48 + 8] ^= 0xff  # num_rows, slot 1
    assert "metablock" in rejected(bytes(flipped))

    def reseal(b, txid, nrows, ioff, isize):
        import hashlib
        body = struct.pack("<QQQI", txid, nrows, ioff, isize)
        blk = body + hashlib.sha1(body).digest()
        b = bytearray(b)
        slot = mb + (txid % 2) * 48
        b[slot:slot + 48] = blk
        return bytes(b)

    txid, nrows, ioff, isize = struct.unpack_from("<QQQI", img, mb + 48 * 0)
    if nrows != n:  # the valid block is the other slot
        txid, nrows, ioff, isize = struct.unpack_from("<QQQI", img, mb + 48)
    assert nrows == n
    # crafted (validly sealed) metablocks
    assert "index" in rejected(reseal(img, txid + 2, n, (1 << 64) - 8, 64))     # offset + size wraps
    assert "index" in rejected(reseal(img, txid + 2, n, len(img) - 4, 64))     # runs past the end
    assert "end of column" in rejected(reseal(img, txid + 2, n * 50, ioff, isize))  # pages too few
    assert E.inspect_image(reseal(img, txid + 2, n - 5, ioff, isize)) == (n - 5, 4)
    # index entries: a page whose offset + size wraps / leaves the file, a page of the
    # wrong size for its encoding
    idx = bytearray(img[ioff:ioff + isize])

    def varuint(v):
        out = bytearray()
        while True:
            x = v & 0x7f
            v >>= 7
            out.append(x | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def index_with(entries):
        body = varuint(len(entries)) + b"".join(
            varuint(k) + varuint(c) + varuint(o) + varuint(sz) for k, c, o, sz in entries)
        b = img[:ioff] + body
        return reseal(b, txid + 2, n, ioff, len(body))

    def read_index():
        pos, out = 0, []

        def rd():
            nonlocal pos
            v = s = 0
            while True:
                x = idx[pos]
                pos += 1
                v |= (x & 0x7f) << s
                s += 7
                if not x & 0x80:
                    return v
        cnt = rd()
        for _ in range(cnt):
            out.append((rd(), rd(), rd(), rd()))
        return out

    entries = read_index()
    assert E.inspect_image(index_with(entries)) == (n, 4)
    e0 = list(entries)
    e0[0] = (e0[0][0], e0[0][1], (1 << 64) - 16, e0[0][3])
    assert "page out of bounds" in rejected(index_with(e0))
    e1 = list(entries)
    e1[0] = (e1[0][0], e1[0][1], e1[0][2], e1[0][3] - 8)
    assert "page" in rejected(index_with(e1))
    # dropping the last data page of the PLAIN64 column leaves too few values
    last_u64 = max(k for k, e in enumerate(entries) if e[0] == 1 and e[1] == entries[0][1]
                   and entries[0][0] == 1) if entries[0][0] == 1 else None
    if last_u64 is not None:
        e2 = [e for k, e in enumerate(entries) if k != last_u64]
        assert "end of column" in rejected(index_with(e2))
    # random single-byte corruption of header and index must never crash
    rng = np.random.default_rng(5)
    for _ in range(300):
        b = bytearray(img[:ioff + isize])
        pos = int(rng.integers(0, 600)) if rng.random() < 0.5 else int(ioff + rng.integers(0, isize))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            E.inspect_image(bytes(b))
        except E.EvqlError as e:
            assert e.code == K.EVQL_EIO
