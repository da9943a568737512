#!/bin/bash
# Regenerates tests/golden/perl/* from the reference's own golden-by-construction
# generator, scripts/perl/genotype_set_builder.pl (reference file:lines 68-207).
#
# The generator writes a random .tped/.tfam pair *and* the per-SNP genotype
# counts it expects ("xx aa ab bb" per line; aa = "A A", bb = "C C"):
#   plain mode        -> <name>.expected.dist           (25 % missing calls)
#   --case-control    -> <name>.expected.{case,control}.dist (no missing calls)
# perl 5.34 has no PERL_RAND_SEED, so the files are NOT reproducible bit for
# bit: they were generated once by this script in the build container and are
# committed as data.  Re-running replaces them with an equally valid set.
#
# Only runs where /root/reference exists (never on the GPU box).
set -euo pipefail
REF=${REF:-/root/reference}
GEN="$REF/scripts/perl/genotype_set_builder.pl"
OUT="$(cd "$(dirname "$0")" && pwd)/perl"
mkdir -p "$OUT"
cd "$OUT"
rm -f ./*.tped ./*.tfam ./*.dist
# word/block/16-byte-unit boundary sizes, 4 markers each
for n in 1 7 8 9 15 16 17 31 32 33 63 64 65 127 128 129 255 256 257; do
  perl "$GEN" --markers 4 --individs $n --tplink plain_m4_n$n >/dev/null
  perl "$GEN" --markers 4 --individs $n --case-control --tplink cc_m4_n$n >/dev/null
done
# the shape SURVEY.md section 4 records (6 markers x 70 individuals), and larger rows
perl "$GEN" --markers 6  --individs 70   --case-control --tplink cc_m6_n70 >/dev/null
perl "$GEN" --markers 40 --individs 70   --tplink plain_m40_n70 >/dev/null
perl "$GEN" --markers 24 --individs 1000 --tplink plain_m24_n1000 >/dev/null
perl "$GEN" --markers 24 --individs 1000 --case-control --tplink cc_m24_n1000 >/dev/null
perl "$GEN" --markers 12 --individs 4097 --tplink plain_m12_n4097 >/dev/null
perl "$GEN" --markers 12 --individs 4097 --case-control --tplink cc_m12_n4097 >/dev/null
ls | wc -l
