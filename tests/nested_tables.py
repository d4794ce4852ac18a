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
    # string fields at every repetition depth
    ("session_id", K.COL_STRING, K.ENC_STRING_PLAIN),
    ("event.search_query.query_string", K.COL_STRING, K.ENC_STRING_PLAIN),
    ("event.search_query.result_items.item_id", K.COL_STRING, K.ENC_STRING_PLAIN),
]

NESTED_SCHEMA = {
    "time": K.T_UINT64,
    "event.search_query.time": K.T_UINT64,
    "event.search_query.num_result_items": K.T_UINT64,
    "event.search_query.result_items.position": K.T_UINT64,
    "event.search_query.result_items.clicked": K.T_BOOL,
    "session_id": K.T_STRING,
    "event.search_query.query_string": K.T_STRING,
    "event.search_query.result_items.item_id": K.T_STRING,
}


# columns of testtbl.cst from SIBLING repeated groups (event.cart_items / event.page_view /
# event.search_query are repeated side by side, result_items sits below search_query):
# CSTableScan zips them level by level (CSTableScan.cc:187-541)
SIBLING_SCHEMA = dict(NESTED_SCHEMA, **{
    "event.cart_items.quantity": K.T_UINT64,
    "event.cart_items.price_cents": K.T_UINT64,
    "event.cart_items.item_id": K.T_STRING,
    "event.page_view.time": K.T_UINT64,
    "event.page_view.item_id": K.T_STRING,
    "event.search_query.page": K.T_UINT64,
})


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
        data[name] = r.read(name, nvals, "string" if lt == K.COL_STRING else "uint")
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
    """config-5 shape, see eventql_amd.synth.items_table_image"""
    from eventql_amd import synth
    return synth.items_table_image(nrec, seed)
