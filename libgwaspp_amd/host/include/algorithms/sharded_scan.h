// algorithms/sharded_scan.h -- the case/control scan of algorithms/maf_func.cpp:292-316 (inline_cc_maf's loop)
// sharded over the GPUs of a node from C++: one process per GPU, every process holds the SNP rows
// gw_shard_range() gives its rank (the loop touches row i and the shared masks only, so rows shard freely), scans
// them with the HIP kernel and the 96-byte per-SNP records are gathered to rank 0 over RCCL / xGMI
// (include/gwas_scan.h, "multi-GPU").  The reference is single-process; this is the north-star's
// "SNP columns shard embarrassingly across the 8 GPUs of one node with an RCCL gather of per-SNP statistics".
#pragma once
#include <string>
#include <vector>

#include "genetics/genetic_data.h"

namespace libgwaspp {
namespace algorithms {

struct ShardInfo {
    int rank, world, device;
    // rendezvous: rank 0 writes the communicator id here (gw_comm_new_id), the others wait for it
    std::string id_file;
    ShardInfo() : rank(0), world(1), device(0) {}
};

// [begin, end) of `rank`'s SNPs out of n_total
void shardRange(unsigned long n_total, int rank, int world, unsigned long &begin, unsigned long &end);

// `shard` holds this rank's rows (row r = global SNP begin + r) and the job's CaseControlSet.  On rank 0 counts
// (n_total x 8: case aa ab bb xx, control aa ab bb xx) and stats (n_total x 8, gw_scan_case_control's order) come
// back in global SNP order; on the other ranks they come back empty.  Errors abort, like the reference's asserts.
void shardedCaseControlScan(genetics::GeneticData *shard, const ShardInfo &si, unsigned long n_total,
                            std::vector<uint> &counts, std::vector<double> &stats);

}  // namespace algorithms
}  // namespace libgwaspp
