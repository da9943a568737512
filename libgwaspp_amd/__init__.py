"""libgwaspp_amd -- MI355X-native per-SNP association scan behind libgwaspp's
table API.

The product is the C-ABI shared library ``libgwscan.so`` (include/gwas_scan.h,
sources under libgwaspp_amd/csrc) and the C++ mirror of the reference's
GenoTable interface (libgwaspp_amd/host).  This Python package is plumbing for
tests and bench.py: a thin ctypes view of the C-ABI.  It has no CPU fallback:
importing works anywhere, but every compute call needs the HIP library and a
GPU and fails loudly otherwise.
"""
from .capi import (GwError, GwTable, GwComm, comm_new_id, shard_range, shard_pair_blocks, parse_boost_list, DeviceBuffer, build_library, device_count, library, library_path,  # noqa: F401
                   pack_genotype_row, build_stream_mask, stream_blocks, block_blocks,
                   LAYOUT_STREAM, LAYOUT_BLOCK, N_COUNTS, N_STATS)
