// sharded_scan.cpp -- see algorithms/sharded_scan.h.
#include "algorithms/sharded_scan.h"

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

#include "algorithms/epistasis_func.h"
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

// Rank 0 publishes the communicator id through a file: 8 bytes "GWCOMMID", the job's nonce, the 128-byte id --
// written under another name, then renamed, so a reader never sees half of it.  The others poll for a file that
// carries THEIR job's nonce; anything else at the path (a crashed earlier job's file, which rank 0 also removes
// before it makes the new id) is ignored.
const char kMagic[8] = {'G', 'W', 'C', 'O', 'M', 'M', 'I', 'D'};

void rendezvous(const ShardInfo &si, unsigned char (&id)[GW_COMM_ID_BYTES]) {
    if (si.rank == 0) {
        if (si.world > 1) (void)unlink(si.id_file.c_str());  // a leftover of a job that did not finish
        die_unless_ok(gw_comm_new_id(id), "gw_comm_new_id");
        if (si.world == 1) return;
        const std::string tmp = si.id_file + ".tmp";
        {
            std::ofstream f(tmp.c_str(), std::ios::binary | std::ios::trunc);
            f.write(kMagic, sizeof kMagic);
            f.write(reinterpret_cast<const char *>(&si.nonce), sizeof si.nonce);
            f.write(reinterpret_cast<const char *>(id), sizeof id);
            if (!f) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); std::abort(); }
        }
        if (std::rename(tmp.c_str(), si.id_file.c_str()) != 0) { std::perror("rename"); std::abort(); }
        return;
    }
    for (int waited_ms = 0;; waited_ms += 50) {
        std::ifstream f(si.id_file.c_str(), std::ios::binary);
        char magic[8];
        unsigned long long nonce = 0;
        if (f.read(magic, sizeof magic) && f.read(reinterpret_cast<char *>(&nonce), sizeof nonce) &&
            f.read(reinterpret_cast<char *>(id), sizeof id) && std::memcmp(magic, kMagic, sizeof magic) == 0 &&
            nonce == si.nonce)
            return;
        if (waited_ms > 120000) {
            std::fprintf(stderr, "rank %d: no communicator id with nonce %llu at %s after 120 s\n", si.rank, si.nonce,
                         si.id_file.c_str());
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


void shardedBoostScreen(GeneticData *gd, const ShardInfo &si, double threshold, std::vector<ShardedInteractionPair> &passing,
                        unsigned long long *n_pairs_out, int root) {
    passing.clear();
    HipGenotypeTable *ht = dynamic_cast<HipGenotypeTable *>(gd->getGenotypeTable());
    if (!ht || !gd->getCaseControlSet()) {
        std::fprintf(stderr, "shardedBoostScreen: needs a HipGenotypeTable and a CaseControlSet\n");
        std::abort();
    }
    const uint64_t M = (uint64_t)gd->getGenotypedMarkersCount();
    unsigned char id[GW_COMM_ID_BYTES];
    rendezvous(si, id);
    gw_comm *comm = nullptr;
    die_unless_ok(gw_comm_create(si.device, si.rank, si.world, id, &comm), "gw_comm_create");

    ht->replicateFrom(comm, root);                      // every rank needs every row
    ht->selectCaseControl(*gd->getCaseControlSet());    // masks come from the .tfam every rank read
    gw_table *t = ht->handle();
    void *stream = nullptr;
    die_unless_ok(gw_table_stream(t, &stream), "gw_table_stream");

    size_t n_blocks = 0;
    die_unless_ok(gw_shard_pair_blocks(M, 2048, si.rank, si.world, nullptr, nullptr, 0, &n_blocks), "gw_shard_pair_blocks");
    std::vector<uint64_t> a_begin(n_blocks ? n_blocks : 1), n_a(n_blocks ? n_blocks : 1);
    die_unless_ok(gw_shard_pair_blocks(M, 2048, si.rank, si.world, a_begin.data(), n_a.data(), n_blocks, &n_blocks),
                  "gw_shard_pair_blocks");

    uint64_t capacity = 1u << 16;
    int64_t prefilter_before = 1;
    die_unless_ok(gw_get_option(t, GW_OPT_BOOST_PREFILTER, &prefilter_before), "gw_get_option");
    std::vector<char> host;
    unsigned long long n_pairs = 0;
    for (;;) {
        const size_t list_bytes = GW_BOOST_LIST_HEADER_BYTES + (size_t)capacity * sizeof(gw_boost_hit);
        void *d_list = nullptr, *d_all = nullptr;
        die_unless_ok(gw_device_malloc(si.device, list_bytes, &d_list), "gw_device_malloc");
        die_unless_ok(gw_device_malloc(si.device, list_bytes * (size_t)si.world, &d_all), "gw_device_malloc");
        die_unless_ok(gw_boost_list_init(t, d_list, stream), "gw_boost_list_init");
        for (size_t k = 0; k < n_blocks; ++k)
            die_unless_ok(gw_boost_screen_device(t, a_begin[k], n_a[k], a_begin[k], M - a_begin[k], threshold, d_list,
                                                 capacity, stream),
                          "gw_boost_screen_device");
        die_unless_ok(gw_comm_allgather(comm, d_list, list_bytes, d_all, stream), "gw_comm_allgather");  // behind the screens
        die_unless_ok(gw_comm_synchronize(comm, stream), "gw_comm_synchronize");
        host.resize(list_bytes * (size_t)si.world);
        die_unless_ok(gw_device_download(si.device, host.data(), d_all, host.size(), stream), "gw_device_download");
        die_unless_ok(gw_device_free(si.device, d_list), "gw_device_free");
        die_unless_ok(gw_device_free(si.device, d_all), "gw_device_free");
        // every rank sees every header, so all of them take the same decision below
        uint64_t most = 0, outside_model = 0;
        n_pairs = 0;
        int64_t prefilter_now = 1;
        die_unless_ok(gw_get_option(t, GW_OPT_BOOST_PREFILTER, &prefilter_now), "gw_get_option");
        for (int p = 0; p < si.world; ++p) {
            uint64_t head[4];
            std::memcpy(head, host.data() + (size_t)p * list_bytes, sizeof head);
            most = std::max(most, head[0]);
            n_pairs += head[1];
            if (prefilter_now) outside_model += head[2];
        }
        if (outside_model) {  // the single-precision bound's error model failed somewhere: every pair in FP64
            die_unless_ok(gw_set_option(t, GW_OPT_BOOST_PREFILTER, 0), "gw_set_option");
            continue;
        }
        if (most > capacity) {  // a list was full: again with room for the longest
            capacity = most;
            continue;
        }
        for (int p = 0; p < si.world; ++p) {
            const char *base = host.data() + (size_t)p * list_bytes;
            uint64_t n = 0;
            std::memcpy(&n, base, sizeof n);
            const gw_boost_hit *h = reinterpret_cast<const gw_boost_hit *>(base + GW_BOOST_LIST_HEADER_BYTES);
            for (uint64_t k = 0; k < n; ++k)
                passing.push_back(std::make_pair(std::make_pair((uint)h[k].snp_a, (uint)h[k].snp_b), h[k].measure));
        }
        break;
    }
    die_unless_ok(gw_set_option(t, GW_OPT_BOOST_PREFILTER, prefilter_before), "gw_set_option");
    std::sort(passing.begin(), passing.end(), [](const ShardedInteractionPair &x, const ShardedInteractionPair &y) {
        return x.first < y.first;
    });
    if (n_pairs_out) *n_pairs_out = n_pairs;
    die_unless_ok(gw_comm_destroy(comm), "gw_comm_destroy");
    if (si.rank == 0 && si.world > 1) (void)unlink(si.id_file.c_str());
}

void computeBoostSharded(GeneticData *gd, const ShardInfo &si, std::ostream *out, int root) {
    std::vector<ShardedInteractionPair> passing;
    unsigned long long n_pairs = 0;
    shardedBoostScreen(gd, si, 30.0, passing, &n_pairs, root);   // threshold: epistasis_func.cpp:390
    if (si.rank != 0) return;
    CaseControlSet &ccs = *gd->getCaseControlSet();
    const int nIndivids = (int)(ccs.getCaseCount() + ccs.getControlCount());
    GenoTable &gt = *gd->getGenotypeTable();
    marginal_information *pMargins = nullptr;
    int nMarkerCount = 0;
    computeMargins(gt, nIndivids, pMargins, nMarkerCount);
    *out << "Pre-screening " << nMarkerCount << " SNP interactions" << std::endl;
    boostStage2AndReport(gt, pMargins, nIndivids, passing, out);
    delete[] pMargins;
}

}  // namespace algorithms
}  // namespace libgwaspp
