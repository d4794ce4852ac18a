"""nested (REPEATED RECORD) tables for the Dremel-scan tests"""
import functools
import os

import numpy as np

import eventql_amd as E
from eventql_amd import capi as K
import oracle_lib as O
import tables as T

NESTED_COLS = [
    ("time", K.COL_UNSIGNED_INT, K.ENC_UINT64_LEB128),
    ("event.search_query.time", K.COL_UNSIGNED_INT, K.ENC_UINT64_LEB128),
    ("event.search_query.num_result_items", K.COL_UNSIGNED_INT, K.ENC_UINT32_BITPACKED),
    ("event.search_query.result_items.position", K.COL_UNSIGNED_INT, K.ENC_UINT32_PLAIN),
    ("event.search_query.result_items.clicked", K.COL_BOOLEAN, K.ENC_BOOLEAN_BITPACKED),
]

NESTED_SCHEMA = {
    "time": K.T_UINT64,
    "event.search_query.time": K.T_UINT64,
    "event.search_query.num_result_items": K.T_UINT64,
    "event.search_query.result_items.position": K.T_UINT64,
    "event.search_query.result_items.clicked": K.T_BOOL,
}


@functools.lru_cache(maxsize=1)
def testtbl_v2():
    """the reference's v0.1.0 fixture re-encoded as cstable v0.2.0: identical
    (r, d, value) triples of five of its columns, written by the product's writer"""
    src = os.path.join(T.GOLDEN, "testtbl.cst")
    r = O.TableReader(src, "orc")
    info = {c["name"]: c for c in r.columns()}
    L = O.oracle()
    specs, data = [], {}
    for name, lt, enc in NESTED_COLS:
        ci = info[name]
        specs.append(dict(name=name, logical_type=lt, storage_type=enc,
                          rlevel_max=ci["rlevel_max"], dlevel_max=ci["dlevel_max"]))
        nvals = L.orc_table_column_num_values(r.h, name.encode())
        data[name] = r.read(name, nvals, "uint")
    r.close()
    w = E.Writer(specs)
    for name, _, _ in NESTED_COLS:
        rl, dl, pr, v = data[name]
        w.put(name, v, rlvl=rl, dlvl=dl)
    w.commit(213)
    img = w.image()
    w.close()
    return img


ITEMS_SCHEMA = {"id": K.T_UINT64, "items.position": K.T_UINT64, "items.price": K.T_UINT64,
                "score": K.T_FLOAT64}


@functools.lru_cache(maxsize=2)
def items_table(nrec=100_000, seed=3):
    """config-5 shape: REPEATED RECORD items{position, price} (rlevel_max 1,
    dlevel_max 2), geometric 0..8 items per record, plus two top-level columns"""
    rng = np.random.default_rng(seed)
    cnt = np.minimum(rng.geometric(0.35, nrec) - 1, 8)
    # slots: a record with no items still has one (r=0, d=0) slot
    slots = np.maximum(cnt, 1)
    total = int(slots.sum())
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]])
    rl = np.ones(total, np.uint64)
    rl[starts] = 0
    rec_of_slot = np.repeat(np.arange(nrec), slots)
    dl = np.where(cnt[rec_of_slot] > 0, 2, 0).astype(np.uint64)
    pos = (np.arange(total) - starts[rec_of_slot] + 1).astype(np.uint64)
    price = rng.integers(1, 100000, total).astype(np.uint64)
    ids = np.arange(nrec, dtype=np.uint64) * np.uint64(7)
    score = rng.random(nrec) * 100.0
    w = E.Writer([
        dict(name="id", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="items.position", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT32_BITPACKED, rlevel_max=1, dlevel_max=2,
             bitpack_max_value=15),
        dict(name="items.price", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT64_LEB128, rlevel_max=1, dlevel_max=2),
        dict(name="score", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754,
             dlevel_max=1)])
    w.put("id", ids)
    w.put("items.position", pos, rlvl=rl, dlvl=dl)
    w.put("items.price", price, rlvl=rl, dlvl=dl)
    w.put("score", score, present=(np.arange(nrec) % 9 != 0).astype(np.uint8))
    w.commit(nrec)
    img = w.image()
    w.close()
    defined = dl == 2
    return img, dict(cnt=cnt, total=total, n_items=int(defined.sum()),
                     sum_price=int(price[defined].sum()), sum_pos=int(pos[defined].sum()))
