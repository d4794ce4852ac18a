// codegen.h -- plan analysis and HIP source generation for the fused
// scan -> filter -> GROUP BY kernel ("csql::VM bytecode lowered to fused HIP
// kernels", BASELINE.json north_star).
#pragma once
#include <string>
#include <vector>
#include "cstable_format.h"
#include "plan_ir.h"

namespace evql {

// how the fused kernel reads one scan column
struct ColAccess {
  enum Mode { PLAIN64 = 0, PLAIN32 = 1, BITPACKED = 2, SOA = 3 };
  std::string name;
  uint32_t stype = EVQL_T_NIL;  // evql_stype seen by the VM
  Mode mode = PLAIN64;
  uint32_t bits = 0;          // BITPACKED width
  bool has_tags = false;      // SOA with a tag byte array (nullable column)
  bool bool_normalize = false;  // BOOLEAN column: value = (raw > 0)
  bool from_uint_to_float = false;  // FLOAT64 stype over a uint column: (double) u
  bool string_hash = false;   // STRING column materialised as hash64
  bool string_bytes = false;  // ... and compared bytewise on the device (strpos array)
  bool packed = false;        // BITPACKED over the runtime's re-encoded copy (LEB128)
  bool dict_code = false;     // PLAIN32 over the table's dictionary codes of a STRING column
  int layout_index = -1;      // index into TableLayout::columns
};

// one 8-byte aggregate state word
struct StateWord {
  int op;  // EVQL_OP_* (evql_device.h)
};

struct AggPlan {
  uint32_t fn = EVQL_AGG_NONE;
  ExprPtr arg;         // over scan columns; null for count
  int first_word = 0;  // index into state words
  int nwords = 0;
  // count_distinct: index of the (group, value) pair set in EvqlArgs::pairset; the
  // state word counts the pairs this aggregate inserted first
  int distinct_index = -1;
  // EVQL_FLOAT_SUM_EXACT: the sum is kept as two integer words (high part, low 31
  // bits) of multiples of a quantum; index into EvqlArgs::fscale / fbound
  int exact_index = -1;
  // partitioned path: the argument is an unsigned value known to stay below 2^32 - 1
  // (column statistics, runtime.cc choose_tuple_widths); it travels as 32 bits
  bool narrow_arg = false;
};

static const int kMaxExactSums = 4;

static const int kMaxDistinct = 4;

enum KeyMode {
  KEY_NONE = 0,    // no GROUP BY: one global group, register accumulators
  KEY_EXACT = 1,   // one fixed-width key: identity = value bits (+ NULL slot)
  KEY_HASHED = 2   // several keys / string keys: identity = 64-bit tuple hash
};

struct KernelPlan {
  std::vector<ColAccess> cols;
  ExprPtr where;                // over scan columns, may be null
  std::vector<ExprPtr> group;   // over scan columns
  std::vector<AggPlan> aggs;    // one per aggregate select expression
  std::vector<StateWord> states;
  KeyMode key_mode = KEY_NONE;
  bool need_first_row = false;
  bool has_row_filter = false;
  int n_distinct = 0;  // count_distinct aggregates (one HBM pair set each)
  int n_exact = 0;     // exact float sums
  // slot layout: word 0 identity, [second identity word], [first_row], states.
  // Hashed keys (several keys / strings) are identified by TWO independent
  // 64-bit hashes of the key tuple -- like the reference, which identifies a
  // group by a hash of its tuple bytes (SHA1, groupby.cc:129-135).
  bool has_ident2() const { return key_mode == KEY_HASHED; }
  int first_row_word() const { return 1 + (has_ident2() ? 1 : 0); }
  int words_per_slot() const { return state_word_base() + int(states.size()); }
  int state_word_base() const { return first_row_word() + (need_first_row ? 1 : 0); }
  // launch shape
  int block = 256;
  int unroll = 4;
  int lds_slots = 0;  // 0 => aggregate straight into the HBM table
  // entries of the lane-private accumulator cache in front of the LDS table (1, or 4 for
  // plans with 2 .. 4 expected groups), codegen_kernels.inc evql_update
  int lane_cache = 1;
  // high cardinality: radix-partition the passing rows into 2^part_bits buckets
  // of tuples, then aggregate each bucket in LDS (codegen_kernels.inc)
  bool partitioned = false;
  int part_bits = 12;
  // two-level partitioning without the count pass: tuples go into slack-allocated coarse
  // buckets (capacity EvqlPartArgs::coarse_cap each), the fine-bucket sizes are counted
  // while they are scattered; false = exact offsets from evql_part_count first
  bool part_fused = false;
  // partition tuples are arrays of 32-bit words: the identity of a single unsigned key
  // and the first-row index travel as 32 bits where the table's statistics bound them
  bool narrow_ident = false;
  bool narrow_first_row = false;
  // nested scans whose WHERE reads columns of different repetition depth: an extra
  // kernel (evql_where_rows) writes the predicate of every row, from which the runtime
  // reproduces the reference's reset of parent values behind a rejected row
  bool where_rows_kernel = false;
  int tile_rows() const { return block * 2 * unroll; }
};

// launch shape for `hint` expected groups (planner.cc)
void choose_launch_shape(KernelPlan* kp, uint64_t hint);
uint64_t lds_table_max_slots(const KernelPlan& kp);
extern const uint64_t kPartitionAboveSlots;  // partitioned path for hint > this * LDS slots
bool partitioned_path_possible(const KernelPlan& kp);

// the generated translation unit (device library excluded)
std::string generate_kernel_source(const KernelPlan& kp);
// 32-bit words of one partition tuple (identity, [identity 2], [row], update words)
int partition_tuple_u32_words(const KernelPlan& kp);

// the embedded text of evql_device.h
const char* device_library_source();

}  // namespace evql
