// algorithms/sharded_scan.h -- the case/control scan of algorithms/maf_func.cpp:292-316 (inline_cc_maf's loop)
// sharded over the GPUs of a node from C++: one process per GPU, every process holds the SNP rows
// gw_shard_range() gives its rank (the loop touches row i and the shared masks only, so rows shard freely), scans
// them with the HIP kernel and the 96-byte per-SNP records are gathered to rank 0 over RCCL / xGMI
// (include/gwas_scan.h, "multi-GPU").  The reference is single-process; this is the north-star's
// "SNP columns shard embarrassingly across the 8 GPUs of one node with an RCCL gather of per-SNP statistics".
#pragma once
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "genetics/genetic_data.h"

namespace libgwaspp {
namespace algorithms {

struct ShardInfo {
    int rank, world, device;
    // rendezvous: rank 0 writes the communicator id here (gw_comm_new_id), the others wait for it.  The launcher
    // gives every job a FRESH path or a fresh `nonce` (any number the ranks of one job share and no earlier job
    // used -- a start time, the launcher's pid): the file carries the nonce, rank 0 removes whatever a crashed
    // earlier job left at the path before it makes the new id, and a peer ignores a file with another nonce
    // (it would otherwise hand ncclCommInitRank a dead job's id and wait for ever).
    std::string id_file;
    unsigned long long nonce;
    ShardInfo() : rank(0), world(1), device(0), nonce(0) {}
};

// [begin, end) of `rank`'s SNPs out of n_total
void shardRange(unsigned long n_total, int rank, int world, unsigned long &begin, unsigned long &end);

// `shard` holds this rank's rows (row r = global SNP begin + r) and the job's CaseControlSet.  On rank 0 counts
// (n_total x 8: case aa ab bb xx, control aa ab bb xx) and stats (n_total x 8, gw_scan_case_control's order) come
// back in global SNP order; on the other ranks they come back empty.  Errors abort, like the reference's asserts.
void shardedCaseControlScan(genetics::GeneticData *shard, const ShardInfo &si, unsigned long n_total,
                            std::vector<uint> &counts, std::vector<double> &stats);


// computeBoost's O(M^2) pre-screen (algorithms/epistasis_func.cpp:393-486) sharded over the GPUs of a node: the
// pair triangle is cut into row blocks dealt to the ranks in snake order (gw_shard_pair_blocks), every rank needs
// every row, so the rows `root` holds (the rank that ingested the .tped) are first replicated into the same-sized
// tables of the others with one RCCL broadcast over xGMI (gw_table_replicate); each rank appends the hits of its
// blocks to a fixed-capacity list in HBM (gw_boost_screen_device) and the lists are all-gathered
// (gw_comm_allgather).  Every rank comes back with the same `passing`: all pairs with measure > threshold, sorted
// by (i, j) -- the order computeBoost's loop visits them in.  *n_pairs_out = pairs evaluated by all ranks.
typedef std::pair<std::pair<uint, uint>, double> ShardedInteractionPair;
void shardedBoostScreen(genetics::GeneticData *gd, const ShardInfo &si, double threshold,
                        std::vector<ShardedInteractionPair> &passing, unsigned long long *n_pairs_out = nullptr,
                        int root = 0);
// computeBoost (algorithms/epistasis_func.cpp:349-506) with the pre-screen sharded as above; rank 0 then runs stage 2
// (computeGTest) on the gathered hits and prints the reference's report (without its wall-clock line).
void computeBoostSharded(genetics::GeneticData *gd, const ShardInfo &si, std::ostream *out, int root = 0);

}  // namespace algorithms
}  // namespace libgwaspp
