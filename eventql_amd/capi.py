"""ctypes declarations of the C ABI in include/evql_gpu.h.

Only struct layouts and constants live here; loading libevql_mi355x.so is done
in eventql_amd/__init__.py.
"""
import ctypes as C

# evql_status
EVQL_OK = 0
EVQL_EIO = -1
EVQL_EARG = -2
EVQL_ERUNTIME = -3
EVQL_ENOTSUP = -4
EVQL_EDEVICE = -5
EVQL_ENOMEM = -6

# evql_stype (csql::SType, sql/svalue.h:41-49)
T_NIL, T_UINT64, T_INT64, T_FLOAT64, T_BOOL, T_STRING, T_TIMESTAMP64 = range(7)
STAG_NULL = 1

# cstable column types / encodings (io/cstable/cstable.h:113-131)
COL_SUBRECORD, COL_BOOLEAN, COL_UNSIGNED_INT, COL_SIGNED_INT, COL_STRING, \
    COL_FLOAT, COL_DATETIME = range(7)
ENC_BOOLEAN_BITPACKED = 1
ENC_UINT32_BITPACKED = 10
ENC_UINT32_PLAIN = 11
ENC_UINT64_PLAIN = 12
ENC_UINT64_LEB128 = 13
ENC_FLOAT_IEEE754 = 14
ENC_STRING_PLAIN = 100

# opcodes (runtime/vm.h:44-52)
X_CALL_PURE, X_CALL_INSTANCE, X_LITERAL, X_INPUT, X_JUMP, X_CJUMP, X_RETURN = \
    range(1, 8)

# type slots / function families
TS_UINT64, TS_INT64, TS_FLOAT64, TS_BOOL, TS_STRING, TS_TIMESTAMP64, TS_NIL = \
    range(7)
(FAM_LOGICAL_AND, FAM_LOGICAL_OR, FAM_NEG, FAM_CMP, FAM_EQ, FAM_NEQ, FAM_LT,
 FAM_LTE, FAM_GT, FAM_GTE, FAM_ADD, FAM_SUB, FAM_MUL, FAM_DIV, FAM_MOD, FAM_POW,
 FAM_TO_NIL, FAM_TO_INT64, FAM_TO_TIMESTAMP64, FAM_TO_STRING, FAM_CONCAT, FAM_LCASE, FAM_UCASE,
 FAM_SUBSTRING, FAM_LTRIM, FAM_RTRIM, FAM_STARTSWITH, FAM_ENDSWITH) = range(1, 29)


def FN(family, type_slot):
    return family * 16 + type_slot


(AGG_NONE, AGG_COUNT, AGG_SUM_UINT64, AGG_SUM_INT64, AGG_SUM_FLOAT64,
 AGG_MIN_UINT64, AGG_MAX_UINT64, AGG_MIN_INT64, AGG_MAX_INT64, AGG_MIN_FLOAT64,
 AGG_MAX_FLOAT64, AGG_MEAN_UINT64, AGG_MEAN_INT64, AGG_MEAN_FLOAT64,
 AGG_COUNT_DISTINCT_UINT64) = range(15)

INSTANCE_ACCUMULATE = 1
INSTANCE_GET = 2

MODE_FINAL, MODE_PARTIAL = 0, 1
SCAN_FLAT, SCAN_NESTED, SCAN_NESTED_WITHIN_RECORD = 0, 1, 2


class Instr(C.Structure):
    _fields_ = [("op", C.c_uint32), ("argt", C.c_uint32), ("arg0", C.c_int64)]


class Program(C.Structure):
    _fields_ = [
        ("code", C.POINTER(Instr)),
        ("code_len", C.c_uint32),
        ("method_call", C.c_uint32),
        ("method_accumulate", C.c_uint32),
        ("return_type", C.c_uint32),
        ("aggregate_fn", C.c_uint32),
        ("static_storage", C.POINTER(C.c_uint8)),
        ("static_storage_len", C.c_size_t),
    ]


class PlanDesc(C.Structure):
    _fields_ = [
        ("scan_columns", C.POINTER(C.c_char_p)),
        ("scan_column_types", C.POINTER(C.c_uint32)),
        ("n_scan_columns", C.c_uint32),
        ("where", C.POINTER(Program)),
        ("scan_select", C.POINTER(Program)),
        ("n_scan_select", C.c_uint32),
        ("group_exprs", C.POINTER(Program)),
        ("n_group", C.c_uint32),
        ("select_exprs", C.POINTER(Program)),
        ("n_select", C.c_uint32),
        ("row_filter_bits", C.POINTER(C.c_uint8)),
        ("row_filter_len", C.c_uint64),
        ("group_mode", C.c_uint32),
        ("scan_mode", C.c_uint32),
        ("groups_hint", C.c_uint64),
        ("row_begin", C.c_uint64),
        ("row_end", C.c_uint64),
        ("float_sum_mode", C.c_uint32),
        ("float_sum_bound", C.c_double),
    ]


class ColumnInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 256),
        ("logical_type", C.c_int32),
        ("storage_type", C.c_int32),
        ("column_id", C.c_uint64),
        ("rlevel_max", C.c_uint32),
        ("dlevel_max", C.c_uint32),
        ("n_data_pages", C.c_uint32),
        ("n_rlevel_pages", C.c_uint32),
        ("n_dlevel_pages", C.c_uint32),
        ("payload_bytes", C.c_uint64),
    ]


class SynthSpec(C.Structure):
    _fields_ = [
        ("num_rows", C.c_uint64),
        ("seed", C.c_uint64),
        ("k_mod", C.c_uint64),
        ("u_mod", C.c_uint64),
        ("columns", C.c_uint32),
        ("k_bits", C.c_uint32),
    ]


class ColumnSpec(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("logical_type", C.c_int32),
        ("storage_type", C.c_int32),
        ("column_id", C.c_uint64),
        ("rlevel_max", C.c_uint32),
        ("dlevel_max", C.c_uint32),
        ("bitpack_max_value", C.c_uint32),
    ]


class DeviceColumn(C.Structure):
    _fields_ = [("values", C.c_void_p), ("nulls", C.c_void_p), ("bytes", C.c_void_p),
                ("rlevels", C.c_void_p), ("dlevels", C.c_void_p), ("num_slots", C.c_uint64)]


class ColumnBuf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_size_t)]


class SortSpec(C.Structure):
    _fields_ = [("expr", Program), ("descending", C.c_uint32)]


class QueryStats(C.Structure):
    _fields_ = [
        ("rows_scanned", C.c_uint64),
        ("rows_passed", C.c_uint64),
        ("num_groups", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("n_kernel_launches", C.c_uint32),
        ("used_lds_table", C.c_uint32),
        ("estimated_groups", C.c_uint64),
    ]


class PartialView(C.Structure):
    _fields_ = [
        ("device_words", C.c_void_p),
        ("capacity", C.c_uint64),
        ("words_per_group", C.c_uint32),
        ("num_groups", C.c_uint64),
    ]


HEARTBEAT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class ExchangeStats(C.Structure):
    _fields_ = [("groups_sent", C.c_uint64), ("groups_received", C.c_uint64),
                ("bytes_sent", C.c_uint64), ("export_ms", C.c_double),
                ("transfer_ms", C.c_double), ("merge_ms", C.c_double),
                ("merge_buckets", C.c_uint64)]


EXCHANGE_GATHER_ALL = 0
EXCHANGE_BY_OWNER = 1
LSM_HAS_SKIPLIST, LSM_HAS_UPDATES = 1, 2

# evql_transport_t callbacks
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64,
                            C.POINTER(C.c_uint64))
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p,
                            C.POINTER(C.c_uint64), C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather_u64", ALL_GATHER_FN),
                ("all_to_all_words", ALL_TO_ALL_FN), ("name", C.c_char_p)]

PAGE_ORDER_COLUMNS, PAGE_ORDER_ROWS = 0, 1
FLOAT_SUM_FAST = 0
FLOAT_SUM_EXACT = 1
