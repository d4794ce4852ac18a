/*
 * gpu_partition.h -- evqld's partitions as sources of the GPU operator.
 *
 * REFERENCE-SIDE ADAPTER (see gpu_bridge.h).  On a data node the scan under a (partial)
 * GROUP BY is eventql::PartitionCursor (server/sql/table_scan.cc:116-144,
 * server/sql/partition_cursor.cc): head arena, compacting arena, then the partition's
 * LSM files newest first, each under the row filter openNextTable builds.  The GPU
 * operator takes the same PartitionSnapshot:
 *
 *   chainFromSnapshot()   PartitionSnapshot -> the file chain GpuTableRegistry scans
 *   partitionResolver()   GpuTableRegistry::Resolver over a function that finds the
 *                         snapshot of a scan's table name; in evqld:
 *
 *     tables->setResolver(evql_adapter::partitionResolver(
 *         [pmap](const std::string& table_name) -> RefPtr<eventql::PartitionSnapshot> {
 *           auto ref = eventql::TSDBTableRef::parse(table_name);       // tsdb://localhost/<t>/<p>
 *           if (ref.partition_key.isEmpty()) return nullptr;           // (not a partition scan)
 *           auto p = pmap->findPartition(ns, ref.table_key, ref.partition_key.get());
 *           return p.isEmpty() ? nullptr : p.get()->getSnapshot();     // table_scan.cc:119-142
 *         }));
 *
 * Arenas: a snapshot whose head or compacting arena holds records is NOT lowered -- the
 * resolver answers false and the CPU PartitionCursor runs (the arenas are in-memory
 * cstables that change with every insert; partition_cursor.cc:87-131).  The C ABI itself
 * takes arenas (evql_lsm_chain_add with a skiplist) for hosts that want to upload them.
 */
#pragma once
#include <functional>
#include <string>
#include <vector>
#include <eventql/db/partition_snapshot.h>
#include <eventql/util/io/fileutil.h>
#include <eventql/util/stringutil.h>
#include "gpu_group_by_scan.h"

namespace evql_adapter {

/* false: the snapshot cannot be scanned from its files alone (records in an arena) */
inline bool chainFromSnapshot(const eventql::PartitionSnapshot& snap,
                              std::vector<ChainFile>* oldest_first, std::string* version_tag) {
  if (snap.head_arena.get() && snap.head_arena->getCSTableFile()) return false;
  if (snap.compacting_arena.get() && snap.compacting_arena->getCSTableFile()) return false;
  oldest_first->clear();
  for (const auto& tbl : snap.state.lsm_tables()) { /* "Last is most recent" */
    ChainFile f;
    f.file = FileUtil::joinPaths(snap.base_path, tbl.filename() + ".cst"); /* :140-142 */
    f.has_skiplist = tbl.has_skiplist();
    f.has_updates = tbl.has_updates();
    oldest_first->push_back(f);
  }
  if (version_tag) {
    /* the ingredients of TableScan's cache key (server/sql/table_provider.cc:229-238) */
    *version_tag = StringUtil::format("$0~$1~$2~$3", snap.state.tsdb_namespace(),
                                      snap.state.table_key(), snap.key.toString(),
                                      snap.state.lsm_sequence());
  }
  return !oldest_first->empty();
}

typedef std::function<RefPtr<eventql::PartitionSnapshot>(const std::string& table_name)>
    SnapshotLookup;

inline GpuTableRegistry::Resolver partitionResolver(SnapshotLookup find_snapshot) {
  return [find_snapshot](const std::string& table_name, std::vector<ChainFile>* files,
                         ScanKind* kind, std::string* version_tag) -> bool {
    RefPtr<eventql::PartitionSnapshot> snap = find_snapshot(table_name);
    if (!snap.get()) return false;
    /* PartitionCursor builds FastCSTableScan for NO_AGGREGATION statements and CSTableScan
     * otherwise (:42-50, 197-213); evql_query_create_chain lowers the former */
    *kind = ScanKind::FAST;
    return chainFromSnapshot(*snap.get(), files, version_tag);
  };
}

}  // namespace evql_adapter
