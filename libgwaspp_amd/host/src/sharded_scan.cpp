// sharded_scan.cpp -- see algorithms/sharded_scan.h.
#include "algorithms/sharded_scan.h"

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

#include "genetics/genotype/hip_genotype_table.h"
#include "gwas_scan.h"

namespace libgwaspp {
namespace algorithms {

using namespace libgwaspp::genetics;

namespace {

void die_unless_ok(int rc, const char *what) {
    if (rc == GW_OK) return;
    std::fprintf(stderr, "shardedCaseControlScan: %s failed (%d): %s\n", what, rc, gw_last_error());
    std::abort();
}

// rank 0 publishes the communicator id through a file (written under another name, then renamed: a reader never
// sees half of it); the others poll for it
void rendezvous(const ShardInfo &si, unsigned char (&id)[GW_COMM_ID_BYTES]) {
    if (si.rank == 0) {
        die_unless_ok(gw_comm_new_id(id), "gw_comm_new_id");
        if (si.world == 1) return;
        const std::string tmp = si.id_file + ".tmp";
        {
            std::ofstream f(tmp.c_str(), std::ios::binary | std::ios::trunc);
            f.write(reinterpret_cast<const char *>(id), sizeof id);
            if (!f) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); std::abort(); }
        }
        if (std::rename(tmp.c_str(), si.id_file.c_str()) != 0) { std::perror("rename"); std::abort(); }
        return;
    }
    for (int waited_ms = 0;; waited_ms += 50) {
        std::ifstream f(si.id_file.c_str(), std::ios::binary);
        if (f.read(reinterpret_cast<char *>(id), sizeof id) && f.gcount() == (std::streamsize)sizeof id) return;
        if (waited_ms > 120000) {
            std::fprintf(stderr, "rank %d: no communicator id at %s after 120 s\n", si.rank, si.id_file.c_str());
            std::abort();
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

}  // namespace

void shardRange(unsigned long n_total, int rank, int world, unsigned long &begin, unsigned long &end) {
    uint64_t b = 0, e = 0;
    die_unless_ok(gw_shard_range(n_total, rank, world, &b, &e), "gw_shard_range");
    begin = b;
    end = e;
}

void shardedCaseControlScan(GeneticData *shard, const ShardInfo &si, unsigned long n_total, std::vector<uint> &counts,
                            std::vector<double> &stats) {
    counts.clear();
    stats.clear();
    HipGenotypeTable *ht = dynamic_cast<HipGenotypeTable *>(shard->getGenotypeTable());
    if (!ht || !shard->getCaseControlSet()) {
        std::fprintf(stderr, "shardedCaseControlScan: needs a HipGenotypeTable and a CaseControlSet\n");
        std::abort();
    }
    unsigned long begin = 0, end = 0;
    shardRange(n_total, si.rank, si.world, begin, end);
    const unsigned long n_local = end - begin;
    if ((unsigned long)shard->getGenotypedMarkersCount() != n_local) {
        std::fprintf(stderr, "shardedCaseControlScan: rank %d holds %d rows, its shard of %lu SNPs has %lu\n", si.rank,
                     shard->getGenotypedMarkersCount(), n_total, n_local);
        std::abort();
    }
    ht->selectCaseControl(*shard->getCaseControlSet());  // installs the masks (replicated on every rank)
    gw_table *t = ht->handle();

    unsigned char id[GW_COMM_ID_BYTES];
    rendezvous(si, id);
    gw_comm *comm = nullptr;
    die_unless_ok(gw_comm_create(si.device, si.rank, si.world, id, &comm), "gw_comm_create");

    // equal slices for the gather: the largest shard's size; counts block then stats block, as one buffer
    const unsigned long cap = (n_total + (unsigned long)si.world - 1ul) / (unsigned long)si.world;
    const size_t slice = (size_t)cap * (GW_N_COUNTS * sizeof(uint32_t) + GW_N_STATS * sizeof(double));
    void *d_send = nullptr, *d_recv = nullptr, *stream = nullptr;
    die_unless_ok(gw_device_malloc(si.device, slice, &d_send), "gw_device_malloc");
    if (si.rank == 0) die_unless_ok(gw_device_malloc(si.device, slice * (size_t)si.world, &d_recv), "gw_device_malloc");
    die_unless_ok(gw_table_stream(t, &stream), "gw_table_stream");
    char *d_stats = static_cast<char *>(d_send) + (size_t)cap * GW_N_COUNTS * sizeof(uint32_t);
    if (n_local)
        die_unless_ok(gw_scan_case_control_device(t, 0, n_local, d_send, d_stats, stream), "gw_scan_case_control_device");
    die_unless_ok(gw_comm_gather(comm, d_send, slice, d_recv, 0, stream), "gw_comm_gather");  // behind the scan, same stream
    die_unless_ok(gw_comm_synchronize(comm, stream), "gw_comm_synchronize");

    if (si.rank == 0) {
        std::vector<char> host(slice * (size_t)si.world);
        die_unless_ok(gw_device_download(si.device, host.data(), d_recv, host.size(), stream), "gw_device_download");
        counts.resize((size_t)n_total * GW_N_COUNTS);
        stats.resize((size_t)n_total * GW_N_STATS);
        for (int p = 0; p < si.world; ++p) {
            unsigned long b = 0, e = 0;
            shardRange(n_total, p, si.world, b, e);
            const char *src = host.data() + (size_t)p * slice;
            std::memcpy(&counts[(size_t)b * GW_N_COUNTS], src, (size_t)(e - b) * GW_N_COUNTS * sizeof(uint32_t));
            std::memcpy(&stats[(size_t)b * GW_N_STATS], src + (size_t)cap * GW_N_COUNTS * sizeof(uint32_t),
                        (size_t)(e - b) * GW_N_STATS * sizeof(double));
        }
    }
    die_unless_ok(gw_device_free(si.device, d_send), "gw_device_free");
    die_unless_ok(gw_device_free(si.device, d_recv), "gw_device_free");
    die_unless_ok(gw_comm_destroy(comm), "gw_comm_destroy");
    if (si.rank == 0 && si.world > 1) (void)unlink(si.id_file.c_str());
}

}  // namespace algorithms
}  // namespace libgwaspp
