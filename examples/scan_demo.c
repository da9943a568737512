/* scan_demo.c -- the C-ABI of include/gwas_scan.h from plain C99: build a small table from
 * genotype text, install a case/control split, scan, print counts and statistics.
 *
 *   gcc -std=c99 -Iinclude examples/scan_demo.c -Llibgwaspp_amd -lgwscan -Wl,-rpath,$PWD/libgwaspp_amd -o scan_demo
 *
 * Needs an MI355X at run time (there is no CPU fallback); `scan_demo --host-only` exercises the
 * GPU-free entry points and is what tests/test_capi_host.py runs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gwas_scan.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != GW_OK) {                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gw_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t n = 8;
    const char *rows[3] = {
        "AA\tAC\tCC\tAA\t00\tAC\tAA\tCC\t", /* one missing call */
        "GG\tGG\tGT\tTT\tGT\tGG\tGT\tGG\t",
        "CC\tCC\tCC\tCC\tCC\tCC\tCC\tCC\t"  /* monomorphic */
    };
    const int32_t cases[4] = {0, 2, 4, 6}, ctrls[4] = {1, 3, 5, 7};
    uint16_t packed[64], case_mask[8], ctrl_mask[8];
    uint32_t n_cases = 0, n_ctrls = 0;
    const uint32_t blocks = gw_stream_blocks(n);
    int r, k;

    printf("%s\n", gw_version());
    /* host-only part: the packer and the mask builder */
    CHECK(gw_pack_genotype_row(rows[0], rows[0] + strlen(rows[0]), '\t', n, GW_LAYOUT_STREAM, packed));
    printf("row 0 header 0x%04x plane0 0x%04x plane1 0x%04x\n", packed[0], packed[1], packed[1 + blocks]);
    CHECK(gw_build_stream_mask(cases, 4, n, case_mask, &n_cases));
    CHECK(gw_build_stream_mask(ctrls, 4, n, ctrl_mask, &n_ctrls));
    printf("case mask 0x%04x (%u)  control mask 0x%04x (%u)\n", case_mask[0], n_cases, ctrl_mask[0], n_ctrls);
    if (argc > 1 && strcmp(argv[1], "--host-only") == 0) return 0;

    {
        gw_table *t = NULL;
        uint32_t counts[3][GW_N_COUNTS];
        double stats[3][GW_N_STATS];
        CHECK(gw_table_create(0, n, 3, &t));
        for (r = 0; r < 3; ++r) CHECK(gw_table_add_genotype_row(t, (uint64_t)r, rows[r], rows[r] + strlen(rows[r]), '\t'));
        CHECK(gw_table_set_case_control(t, case_mask, ctrl_mask, blocks, n_cases, n_ctrls));
        CHECK(gw_scan_case_control(t, 0, 3, &counts[0][0], &stats[0][0]));
        for (r = 0; r < 3; ++r) {
            printf("SNP %d  cases aa/ab/bb/xx %u %u %u %u  controls %u %u %u %u  ", r, counts[r][0], counts[r][1],
                   counts[r][2], counts[r][3], counts[r][4], counts[r][5], counts[r][6], counts[r][7]);
            for (k = 0; k < GW_N_STATS; ++k) printf(" %.6g", stats[r][k]);
            printf("\n");
        }
        CHECK(gw_table_destroy(t));
    }
    return 0;
}
