"""Partitions of an evqld table for the LSM / PartitionCursor tests: chains of cstable
files as db/partition_arena.cc:41-45 and db/compaction_strategy.cc:62-65 write them
(payload columns + __lsm_id / __lsm_is_update / __lsm_skip / __lsm_version /
__lsm_sequence), built with the product's host writer (pinned byte-identical to the
reference's CSTableWriter in test_format_cpu.py).

A partition is a list of files OLDEST first -- the order of PartitionState::lsm_tables
("Last is most recent", db/partition_state.proto) -- each with its LSMTableRef flags
has_skiplist / has_updates.  PartitionCursor scans them newest first."""
import functools
import hashlib

import numpy as np

import eventql_amd as E
from eventql_amd import capi as K

LSM_SCHEMA = dict(rid=K.T_UINT64, k=K.T_UINT64, a=K.T_UINT64, n=K.T_UINT64, v=K.T_FLOAT64,
                  s=K.T_STRING)

# name -> (seed, [(rows, has_skiplist, has_updates)] oldest first)
PARTITIONS = {
    # updates in newer files shadow older rows; the middle file also has a skiplist
    "basic": (101, [(3000, 0, 1), (2000, 1, 1), (2500, 0, 1)]),
    # no skiplists, has_updates = false everywhere: no file gets a filter, although the
    # rows do carry is_update bits (partition_cursor.cc:153-155: the flags decide)
    "quiet": (102, [(1500, 0, 0), (1000, 0, 0)]),
    # five files incl. a one-row file; an unfiltered newest file whose updates are NOT
    # remembered, then filtered ones
    "mixed": (103, [(1200, 0, 1), (1, 1, 1), (800, 0, 0), (1500, 1, 0), (900, 0, 0)]),
    "single_skip": (104, [(4000, 1, 1)]),
    # oldest file, no skiplist, nothing remembered: scanned whole (:149-151)
    "single_plain": (105, [(4000, 0, 1)]),
    "newest_silent": (106, [(2000, 0, 1), (1500, 0, 1), (1000, 0, 0)]),
    # several 512 KiB pages per stream (20-byte ids: 24,966 per page)
    "big": (107, [(150_000, 0, 1), (100_000, 1, 1), (70_001, 0, 1)]),
}


def lsm_id(who):
    return hashlib.sha1(b"id%d" % who).digest()


def _file_image(rng, file_index, nrows, has_skiplist, id_space):
    who = rng.integers(0, id_space, nrows)
    upd = (rng.random(nrows) < 0.3).astype(np.uint64)
    skip = (rng.random(nrows) < 0.1).astype(np.uint64)
    i = np.arange(nrows, dtype=np.uint64)
    cols = dict(
        rid=np.uint64(file_index) * np.uint64(10_000_000) + i,
        k=(who % 40).astype(np.uint64),
        a=rng.integers(0, 65536, nrows).astype(np.uint64),
        n=rng.integers(0, 1 << 34, nrows).astype(np.uint64),
        n_present=(rng.random(nrows) >= 0.2).astype(np.uint8),
        v=rng.integers(0, 1 << 20, nrows).astype(np.float64) / 64.0,
        s=[b"g%d" % (w % 13) if w % 7 else b"" for w in who],
        ids=[lsm_id(int(w)) for w in who], upd=upd, skip=skip, who=who)
    specs = [
        dict(name="rid", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128),
        dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="n", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128,
             dlevel_max=1),
        dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
        dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
        dict(name="__lsm_is_update", logical_type=K.COL_BOOLEAN,
             storage_type=K.ENC_BOOLEAN_BITPACKED)]
    if has_skiplist:
        specs.append(dict(name="__lsm_skip", logical_type=K.COL_BOOLEAN,
                          storage_type=K.ENC_BOOLEAN_BITPACKED))
    specs += [
        dict(name="__lsm_id", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
        dict(name="__lsm_version", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT64_LEB128),
        dict(name="__lsm_sequence", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT64_LEB128)]
    w = E.Writer(specs)
    w.put("rid", cols["rid"])
    w.put("k", cols["k"])
    w.put("a", cols["a"])
    w.put("n", cols["n"], present=cols["n_present"])
    w.put("v", cols["v"])
    w.put("s", cols["s"])
    w.put("__lsm_is_update", upd)
    if has_skiplist:
        w.put("__lsm_skip", skip)
    w.put("__lsm_id", cols["ids"])
    w.put("__lsm_version", i + np.uint64(1))
    w.put("__lsm_sequence", i + np.uint64(file_index * 1_000_000 + 1))
    w.commit(nrows)
    img = w.image()
    w.close()
    return img, cols


@functools.lru_cache(maxsize=None)
def partition(name):
    """[(file name, image bytes, has_skiplist, has_updates, columns dict)], oldest first"""
    seed, files = PARTITIONS[name]
    rng = np.random.default_rng(seed)
    total = sum(f[0] for f in files)
    id_space = max(4, total // 2)
    out = []
    for fi, (nrows, skl, upd) in enumerate(files):
        img, cols = _file_image(rng, fi, nrows, bool(skl), id_space)
        out.append(("%s_%02d" % (name, fi), img, bool(skl), bool(upd), cols))
    return out


def model_filters(files_oldest_first):
    """PartitionCursor::openNextTable (partition_cursor.cc:134-195) in python, an
    independent cross-check of the C restatement: list of bool arrays in SCAN order
    (newest first); None where the file gets no filter"""
    seen = set()
    out = []
    n = len(files_oldest_first)
    for k, (_, _, skl, upd, c) in enumerate(reversed(files_oldest_first)):
        tblidx = n - 1 - k
        needs = True
        if not skl and tblidx == 0 and not seen:
            needs = False
        if not skl and not upd and not seen:
            needs = False
        if not needs:
            out.append(None)
            continue
        f = np.zeros(len(c["ids"]), bool)
        for i, (d, u, s) in enumerate(zip(c["ids"], c["upd"], c["skip"])):
            if (skl and s) or d in seen:
                continue
            if u:
                seen.add(d)
            f[i] = True
        out.append(f)
    return out
