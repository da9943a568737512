"""ctypes view of the C-ABI in include/gwas_scan.h (libgwaspp_amd/libgwscan.so).

Plumbing only: argument marshalling with numpy, no arithmetic.  A process that
also uses torch must `import torch` BEFORE the first call here: torch's wheel
bundles its own libamdhip64 (same SONAME), and whichever HIP runtime is loaded
first serves both; loading ROCm's first leaves torch without a device.  There is no
fallback path -- if the shared library is missing or no GPU is present the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GW_LIBGWSCAN: an alternative build of the same library (kernel experiments, tools/build_variant.sh)
_LIB = os.environ.get("GW_LIBGWSCAN") or os.path.join(_HERE, "libgwscan.so")

LAYOUT_STREAM = 5
LAYOUT_BLOCK = 3
N_COUNTS = 8
N_STATS = 8

OPT_UNROLL, OPT_MASK_IN_LDS, OPT_BLOCKS_PER_CU, OPT_NONTEMPORAL, OPT_BLOCK_THREADS, OPT_PIPELINE, OPT_COMPLEMENT = 1, 2, 3, 4, 5, 6, 7
OPT_BOOST_PREFILTER = 8
OPT_BOOST_MATRIX_CORES = 9
OPT_SHORT_ROWS = 10
OPT_COLUMN_SEGMENT = 11
OPT_BOOST_PREFILTER_TRIPS = 12
OPT_RESERVE_CUS = 14
OPT_DEFER_RECORDS = 15
OPT_BOOST_STAGGER_US = 16   # retired (round 4)
OPT_BOOST_LUT = 17          # retired (round 4)
OPT_BOOST_FP4 = 18
OPT_BOOST_FIRST_LEVEL = 19
OPT_BOOST_CANDIDATES = 20
OPT_BOOST_ROWS_PER_LAUNCH = 21


class GwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gw status {code}: {msg}")
        self.code = code


class Geometry(C.Structure):
    _fields_ = [("n_samples", C.c_uint32), ("n_snps", C.c_uint64), ("stream_blocks", C.c_uint32),
                ("block_blocks", C.c_uint32), ("stream_row_bytes", C.c_uint32),
                ("block_row_bytes", C.c_uint32), ("device_units", C.c_uint32),
                ("device_plane_bytes", C.c_uint32), ("device_row_bytes", C.c_uint64),
                ("device_table_bytes", C.c_uint64), ("algorithmic_bytes_per_snp", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int),
                ("lds_bytes_per_cu", C.c_int), ("hbm_bytes", C.c_uint64), ("hbm_free_bytes", C.c_uint64)]


class MarginalInformation(C.Structure):
    """gw_marginal_information == the reference's marginal_information (common_genotype.h:101-106)."""
    _fields_ = [("margins", C.c_uint32 * 4), ("cases", C.c_uint32 * 4), ("controls", C.c_uint32 * 4),
                ("marginal_entropy", C.c_double), ("marginal_entropy_y", C.c_double),
                ("pbc", C.c_double * 8), ("pca", C.c_double * 8)]


MARGINAL_DTYPE = np.dtype([("margins", np.uint32, 4), ("cases", np.uint32, 4), ("controls", np.uint32, 4),
                           ("marginal_entropy", np.float64), ("marginal_entropy_y", np.float64),
                           ("pbc", np.float64, 8), ("pca", np.float64, 8)])
assert MARGINAL_DTYPE.itemsize == C.sizeof(MarginalInformation) == 192


def library_path() -> str:
    return _LIB


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libgwscan.so (in-tree)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_LIB):
        os.remove(_LIB)
    subprocess.run(["make", "-C", src_dir, "-s"], check=True)
    return _LIB


_lib = None

# every extern "C" symbol include/gwas_scan.h declares
EXPORTS = [
    "gw_last_error", "gw_version", "gw_device_count", "gw_device_info_get", "gw_table_create",
    "gw_table_destroy", "gw_table_geometry", "gw_table_add_genotype_row", "gw_table_add_genotype_rows", "gw_table_flush",
    "gw_table_upload_rows", "gw_table_download_rows", "gw_table_get_headers", "gw_table_simulate", "gw_table_simulate_exact",
    "gw_table_simulate_phenotype", "gw_table_set_case_control", "gw_table_get_case_control",
    "gw_scan_case_control", "gw_scan_case_control_device", "gw_scan_distribution",
    "gw_scan_distribution_device", "gw_scan_margins", "gw_pair_contingency", "gw_pair_case_control_contingency",
    "gw_boost_screen", "gw_boost_measures", "gw_table_download_selected", "gw_boost_list_init",
    "gw_boost_screen_device", "gw_boost_gtest", "gw_boost_gtest_device", "gw_last_boost_kernel", "gw_shard_pair_blocks", "gw_table_replicate",
    "gw_comm_allgather", "gw_comm_broadcast", "gw_device_memset",
    "gw_timer_start", "gw_timer_stop", "gw_stream_read_probe", "gw_synchronize", "gw_last_scan_kernel",
    "gw_shard_range", "gw_comm_new_id", "gw_comm_create", "gw_comm_destroy", "gw_comm_gather", "gw_comm_synchronize",
    "gw_table_stream", "gw_device_malloc", "gw_device_free", "gw_device_download", "gw_set_option",
    "gw_get_option", "gw_pack_genotype_row", "gw_header_assign", "gw_build_stream_mask", "gw_stream_blocks",
    "gw_block_blocks",
]


def library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise GwError(-2, f"{_LIB} is not built (run __graft_entry__.build() / make -C libgwaspp_amd/csrc); "
                          "there is no CPU fallback")
    L = C.CDLL(_LIB)
    vp, u16p, u32p, f64p = C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L.gw_last_error.restype = C.c_char_p
    L.gw_version.restype = C.c_char_p
    L.gw_device_count.argtypes = [C.POINTER(C.c_int)]
    L.gw_device_info_get.argtypes = [C.c_int, C.POINTER(DeviceInfo)]
    L.gw_table_create.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.POINTER(vp)]
    L.gw_table_destroy.argtypes = [vp]
    L.gw_table_geometry.argtypes = [vp, C.POINTER(Geometry)]
    L.gw_table_add_genotype_row.argtypes = [vp, C.c_uint64, vp, vp, C.c_char]
    L.gw_table_add_genotype_rows.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, C.c_char, C.c_int]
    L.gw_table_flush.argtypes = [vp]
    L.gw_table_upload_rows.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_size_t, C.c_int]
    L.gw_table_download_rows.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_size_t, C.c_int]
    L.gw_table_get_headers.argtypes = [vp, C.c_uint64, C.c_uint64, u16p]
    L.gw_table_simulate.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32]
    L.gw_table_simulate_exact.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32]
    L.gw_table_simulate_phenotype.argtypes = [vp, C.c_uint64]
    L.gw_table_set_case_control.argtypes = [vp, u16p, u16p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.gw_table_get_case_control.argtypes = [vp, u16p, u16p, C.c_uint32, u32p, u32p]
    L.gw_scan_case_control.argtypes = [vp, C.c_uint64, C.c_uint64, u32p, f64p]
    L.gw_scan_case_control_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp]
    L.gw_scan_distribution.argtypes = [vp, C.c_uint64, C.c_uint64, u32p]
    L.gw_scan_distribution_device.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp]
    L.gw_scan_margins.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    L.gw_pair_contingency.argtypes = [vp, u32p, C.c_uint64, u32p]
    L.gw_pair_case_control_contingency.argtypes = [vp, u32p, C.c_uint64, u32p, u32p]
    L.gw_boost_screen.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, vp, C.c_uint64,
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), f64p, f64p]
    L.gw_boost_measures.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, f64p]
    L.gw_table_download_selected.argtypes = [vp, C.c_uint64, C.c_uint64, u16p, u32p, u32p]
    L.gw_boost_list_init.argtypes = [vp, vp, vp]
    L.gw_boost_screen_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, vp, C.c_uint64, vp]
    L.gw_boost_gtest.argtypes = [vp, vp, C.c_uint64, f64p, f64p, C.POINTER(C.c_uint64)]
    L.gw_boost_gtest_device.argtypes = [vp, vp, C.c_uint64, vp, vp, vp]
    L.gw_last_boost_kernel.argtypes = [vp]
    L.gw_last_boost_kernel.restype = C.c_char_p
    L.gw_shard_pair_blocks.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.c_size_t, C.POINTER(C.c_size_t)]
    L.gw_table_replicate.argtypes = [vp, vp, C.c_int, vp]
    L.gw_comm_allgather.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.gw_comm_broadcast.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.gw_device_memset.argtypes = [C.c_int, vp, C.c_int, C.c_size_t, vp]
    L.gw_timer_start.argtypes = [vp, vp]
    L.gw_timer_stop.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.gw_stream_read_probe.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.gw_synchronize.argtypes = [vp]
    L.gw_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gw_comm_new_id.argtypes = [vp]
    L.gw_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.gw_comm_destroy.argtypes = [vp]
    L.gw_comm_gather.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
    L.gw_comm_synchronize.argtypes = [vp, vp]
    L.gw_table_stream.argtypes = [vp, C.POINTER(vp)]
    L.gw_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.gw_device_free.argtypes = [C.c_int, vp]
    L.gw_device_download.argtypes = [C.c_int, vp, vp, C.c_size_t, vp]
    L.gw_last_scan_kernel.argtypes = [vp]
    L.gw_last_scan_kernel.restype = C.c_char_p
    L.gw_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.gw_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.gw_pack_genotype_row.argtypes = [vp, vp, C.c_char, C.c_uint32, C.c_int, u16p]
    L.gw_header_assign.argtypes = [u16p, C.c_uint32, C.POINTER(C.c_int)]
    L.gw_build_stream_mask.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.c_uint32, u16p, u32p]
    L.gw_stream_blocks.argtypes = [C.c_uint32]
    L.gw_stream_blocks.restype = C.c_uint32
    L.gw_block_blocks.argtypes = [C.c_uint32]
    L.gw_block_blocks.restype = C.c_uint32
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise GwError(rc, library().gw_last_error().decode())


def _p16(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def device_count() -> int:
    n = C.c_int(0)
    rc = library().gw_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def device_info(device: int = 0) -> DeviceInfo:
    d = DeviceInfo()
    _check(library().gw_device_info_get(device, C.byref(d)))
    return d


def stream_blocks(n: int) -> int:
    return library().gw_stream_blocks(n)


def block_blocks(n: int) -> int:
    return library().gw_block_blocks(n)


def pack_genotype_row(text: bytes, n_samples: int, layout: int = LAYOUT_STREAM) -> np.ndarray:
    """Host packer behind gw_table_add_genotype_row (no GPU needed)."""
    nb = 2 * stream_blocks(n_samples) + 1 if layout == LAYOUT_STREAM else block_blocks(n_samples) + 1
    row = np.zeros(nb, dtype=np.uint16)
    buf = C.create_string_buffer(text, len(text))
    base = C.addressof(buf)
    _check(library().gw_pack_genotype_row(base, base + len(text), b"\t", n_samples, layout, _p16(row)))
    return row


def build_stream_mask(idx, n_samples: int):
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int32))
    mask = np.zeros(stream_blocks(n_samples), dtype=np.uint16)
    cnt = C.c_uint32(0)
    _check(library().gw_build_stream_mask(idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size, n_samples,
                                          _p16(mask), C.byref(cnt)))
    return mask, cnt.value


def shard_range(n_total: int, rank: int, world: int):
    b, e = C.c_uint64(0), C.c_uint64(0)
    _check(library().gw_shard_range(n_total, rank, world, C.byref(b), C.byref(e)))
    return b.value, e.value


def shard_pair_blocks(n_snps: int, rank: int, world: int, block: int = 2048):
    """[(a_begin, n_a), ...]: the rank's row blocks of the pair triangle (gw_shard_pair_blocks; adjacent blocks merged)."""
    L = library()
    n = C.c_size_t(0)
    _check(L.gw_shard_pair_blocks(n_snps, block, rank, world, None, None, 0, C.byref(n)))
    a, na = (C.c_uint64 * max(1, n.value))(), (C.c_uint64 * max(1, n.value))()
    _check(L.gw_shard_pair_blocks(n_snps, block, rank, world, a, na, n.value, C.byref(n)))
    return [(int(a[k]), int(na[k])) for k in range(n.value)]


COMM_ID_BYTES = 128
BOOST_LIST_HEADER_BYTES = 32
BOOST_HIT_DTYPE = np.dtype([("snp_a", np.uint32), ("snp_b", np.uint32), ("measure", np.float64)])


def parse_boost_list(raw: np.ndarray, capacity: int):
    """(hits, n_hits, n_pairs, aux0) of one device hit list downloaded as bytes (gw_boost_screen_device's layout)."""
    head = raw[:BOOST_LIST_HEADER_BYTES].view(np.uint64)
    n_hits, n_pairs, aux0 = int(head[0]), int(head[1]), int(head[2])
    hits = raw[BOOST_LIST_HEADER_BYTES:BOOST_LIST_HEADER_BYTES + 16 * capacity].view(BOOST_HIT_DTYPE)[:min(n_hits, capacity)]
    return hits, n_hits, n_pairs, aux0


class DeviceBuffer:
    """A plain device allocation through the C-ABI (gw_device_malloc), for callers that have no torch."""

    def __init__(self, nbytes: int, device: int = 0):
        self._L = library()
        self.device, self.nbytes = device, nbytes
        p = C.c_void_p()
        _check(self._L.gw_device_malloc(device, nbytes, C.byref(p)))
        self.ptr = p.value or 0

    def download(self, stream: int | None = None) -> np.ndarray:
        out = np.zeros(self.nbytes, dtype=np.uint8)
        _check(self._L.gw_device_download(self.device, out.ctypes.data, self.ptr, self.nbytes, stream))
        return out

    def memset(self, value: int = 0, stream: int | None = None):
        _check(self._L.gw_device_memset(self.device, self.ptr, value, self.nbytes, stream))

    def close(self):
        if self.ptr:
            self._L.gw_device_free(self.device, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_new_id() -> bytes:
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(library().gw_comm_new_id(buf))
    return buf.raw


class GwComm:
    """The C-ABI's RCCL communicator (gw_comm): gather of equally sized device buffers to a root."""

    def __init__(self, device: int, rank: int, world: int, comm_id: bytes):
        self._L = library()
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        _check(self._L.gw_comm_create(device, rank, world, C.create_string_buffer(comm_id, COMM_ID_BYTES), C.byref(self._h)))

    def gather(self, d_send: int, nbytes: int, d_recv: int | None, root: int = 0, stream: int | None = None):
        _check(self._L.gw_comm_gather(self._h, d_send, nbytes, d_recv, root, stream))

    def allgather(self, d_send: int, nbytes: int, d_recv: int, stream: int | None = None):
        _check(self._L.gw_comm_allgather(self._h, d_send, nbytes, d_recv, stream))

    def broadcast(self, d_buf: int, nbytes: int, root: int = 0, stream: int | None = None):
        _check(self._L.gw_comm_broadcast(self._h, d_buf, nbytes, root, stream))

    def synchronize(self, stream: int | None = None):
        _check(self._L.gw_comm_synchronize(self._h, stream))

    def close(self):
        if self._h:
            self._L.gw_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GwTable:
    """One genotype table resident in HBM (opaque gw_table handle)."""

    def __init__(self, n_samples: int, n_snps: int, device: int = 0):
        self._L = library()
        self._h = C.c_void_p()
        _check(self._L.gw_table_create(device, n_samples, n_snps, C.byref(self._h)))
        self.geometry = Geometry()
        _check(self._L.gw_table_geometry(self._h, C.byref(self.geometry)))
        self.n_samples, self.n_snps = n_samples, n_snps

    def close(self):
        if self._h:
            self._L.gw_table_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- ingest
    def add_genotype_row(self, snp: int, text: bytes, delim: bytes = b"\t"):
        buf = C.create_string_buffer(text, len(text))
        base = C.addressof(buf)
        _check(self._L.gw_table_add_genotype_row(self._h, snp, base, base + len(text), delim))

    def add_genotype_rows(self, snp_begin: int, texts, delim: bytes = b"\t", n_threads: int = 0):
        """Batch form: texts = list of bytes, one genotype row each, packed by n_threads host threads."""
        n = len(texts)
        # the rows are read in place: `texts` keeps the bytes objects (and their buffers) alive for the call
        addr = [C.cast(C.c_char_p(t), C.c_void_p).value or 0 for t in texts]
        begins = (C.c_void_p * n)(*addr)
        ends = (C.c_void_p * n)(*[a + len(t) for a, t in zip(addr, texts)])
        _check(self._L.gw_table_add_genotype_rows(self._h, snp_begin, n, begins, ends, delim, n_threads))

    def flush(self):
        _check(self._L.gw_table_flush(self._h))

    def upload_rows(self, snp_begin: int, rows: np.ndarray, layout: int = LAYOUT_STREAM):
        rows = np.ascontiguousarray(rows, dtype=np.uint16)
        assert rows.ndim == 2
        _check(self._L.gw_table_upload_rows(self._h, snp_begin, rows.shape[0], rows.ctypes.data,
                                            rows.shape[1] * 2, layout))

    def download_rows(self, snp_begin: int, n_rows: int, layout: int = LAYOUT_STREAM, pad_blocks: int = 0) -> np.ndarray:
        """rows in a reference packing; pad_blocks > 0 asks for a pitched destination (extra uint16 per row)."""
        g = self.geometry
        nb = (g.stream_row_bytes if layout == LAYOUT_STREAM else g.block_row_bytes) // 2 + pad_blocks
        rows = np.zeros((n_rows, nb), dtype=np.uint16)
        _check(self._L.gw_table_download_rows(self._h, snp_begin, n_rows, rows.ctypes.data, nb * 2, layout))
        return rows

    def headers(self, snp_begin: int, n_rows: int) -> np.ndarray:
        h = np.zeros(n_rows, dtype=np.uint16)
        _check(self._L.gw_table_get_headers(self._h, snp_begin, n_rows, _p16(h)))
        return h

    def simulate(self, seed: int, snp_global_begin: int = 0, missing_q16: int = 0, exact: bool = False):
        """exact=True: the reference simulator's exact minor-allele count per SNP (gw_table_simulate_exact)"""
        f = self._L.gw_table_simulate_exact if exact else self._L.gw_table_simulate
        _check(f(self._h, seed, snp_global_begin, missing_q16))

    def simulate_phenotype(self, seed: int):
        _check(self._L.gw_table_simulate_phenotype(self._h, seed))

    # ---- masks
    def set_case_control(self, case_stream, ctrl_stream, case_count: int, ctrl_count: int):
        cs = np.ascontiguousarray(case_stream, dtype=np.uint16)
        ct = np.ascontiguousarray(ctrl_stream, dtype=np.uint16)
        _check(self._L.gw_table_set_case_control(self._h, _p16(cs), _p16(ct), cs.size, case_count, ctrl_count))

    def get_case_control(self):
        nb = self.geometry.stream_blocks
        cs, ct = np.zeros(nb, np.uint16), np.zeros(nb, np.uint16)
        nc, nt = C.c_uint32(), C.c_uint32()
        _check(self._L.gw_table_get_case_control(self._h, _p16(cs), _p16(ct), nb, C.byref(nc), C.byref(nt)))
        return cs, ct, nc.value, nt.value

    # ---- scans (host results)
    def scan_case_control(self, snp_begin: int = 0, n_snps: int | None = None, stats: bool = True):
        n = self.n_snps - snp_begin if n_snps is None else n_snps
        counts = np.zeros((n, N_COUNTS), dtype=np.uint32)
        st = np.zeros((n, N_STATS), dtype=np.float64) if stats else None
        _check(self._L.gw_scan_case_control(self._h, snp_begin, n, _p32(counts),
                                            st.ctypes.data_as(C.POINTER(C.c_double)) if stats else None))
        return counts, st

    def scan_distribution(self, snp_begin: int = 0, n_snps: int | None = None):
        n = self.n_snps - snp_begin if n_snps is None else n_snps
        dist = np.zeros((n, 4), dtype=np.uint32)
        _check(self._L.gw_scan_distribution(self._h, snp_begin, n, _p32(dist)))
        return dist

    def scan_margins(self, snp_begin: int = 0, n_snps: int | None = None) -> np.ndarray:
        """marginal_information records (structured array, MARGINAL_DTYPE) of SNPs [snp_begin, +n)."""
        n = self.n_snps - snp_begin if n_snps is None else n_snps
        out = np.zeros(n, dtype=MARGINAL_DTYPE)
        _check(self._L.gw_scan_margins(self._h, snp_begin, n, out.ctypes.data))
        return out

    # ---- scans (device results; pointers are raw device addresses)
    def scan_case_control_device(self, snp_begin: int, n_snps: int, d_counts: int, d_stats: int | None,
                                 stream: int | None = None):
        _check(self._L.gw_scan_case_control_device(self._h, snp_begin, n_snps, d_counts, d_stats, stream))

    def scan_distribution_device(self, snp_begin: int, n_snps: int, d_dist: int, stream: int | None = None):
        _check(self._L.gw_scan_distribution_device(self._h, snp_begin, n_snps, d_dist, stream))

    # ---- pairs
    def pair_contingency(self, pairs) -> np.ndarray:
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros((pairs.shape[0], 16), dtype=np.uint32)
        _check(self._L.gw_pair_contingency(self._h, _p32(pairs), pairs.shape[0], _p32(out)))
        return out

    def pair_case_control_contingency(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        a = np.zeros((pairs.shape[0], 16), dtype=np.uint32)
        b = np.zeros((pairs.shape[0], 16), dtype=np.uint32)
        _check(self._L.gw_pair_case_control_contingency(self._h, _p32(pairs), pairs.shape[0], _p32(a), _p32(b)))
        return a, b

    # ---- BOOST / KSA pre-screen
    def boost_screen(self, a_begin: int, n_a: int, b_begin: int, n_b: int, threshold: float = 30.0,
                     max_hits: int = 1 << 20, extremes: bool = True):
        """-> (hits structured array [snp_a, snp_b, measure], n_hits, n_pairs, max, min);
        extremes=False passes NULL for max/min (they come back as None): the screen may then use its
        single-precision bound first (GW_OPT_BOOST_PREFILTER)."""
        dt = np.dtype([("snp_a", np.uint32), ("snp_b", np.uint32), ("measure", np.float64)])
        hits = np.zeros(max_hits, dtype=dt)
        nh, npairs = C.c_uint64(0), C.c_uint64(0)
        vmax, vmin = C.c_double(0), C.c_double(0)
        _check(self._L.gw_boost_screen(self._h, a_begin, n_a, b_begin, n_b, threshold, hits.ctypes.data, max_hits,
                                       C.byref(nh), C.byref(npairs), C.byref(vmax) if extremes else None,
                                       C.byref(vmin) if extremes else None))
        return (hits[:min(nh.value, max_hits)], nh.value, npairs.value, vmax.value if extremes else None,
                vmin.value if extremes else None)

    def boost_list_init(self, d_list: int, stream: int | None = None):
        _check(self._L.gw_boost_list_init(self._h, d_list, stream))

    def boost_screen_device(self, a_begin: int, n_a: int, b_begin: int, n_b: int, d_list: int, capacity: int,
                            threshold: float = 30.0, stream: int | None = None):
        """Appends the block's hits to the device list at d_list (BOOST_LIST_HEADER_BYTES + 16 * capacity bytes);
        enqueued on `stream`, not synchronised."""
        _check(self._L.gw_boost_screen_device(self._h, a_begin, n_a, b_begin, n_b, threshold, d_list, capacity, stream))

    def boost_gtest(self, pairs):
        """BOOST stage 2 (computeGTest) for the pairs of a hit array (dtype BOOST_HIT_DTYPE) or an (n, 2) integer array
        -> (measure[n], z[n], n_unsettled)."""
        if pairs.dtype != BOOST_HIT_DTYPE:
            ij = np.asarray(pairs)
            pairs = np.zeros(ij.shape[0], dtype=BOOST_HIT_DTYPE)
            pairs["snp_a"], pairs["snp_b"] = ij[:, 0], ij[:, 1]
        pairs = np.ascontiguousarray(pairs)
        n = pairs.shape[0]
        measure, z = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.float64)
        unsettled = C.c_uint64(0)
        _check(self._L.gw_boost_gtest(self._h, pairs.ctypes.data, n, measure.ctypes.data_as(C.POINTER(C.c_double)),
                                      z.ctypes.data_as(C.POINTER(C.c_double)), C.byref(unsettled)))
        return measure, z, unsettled.value

    def boost_gtest_device(self, d_list: int, capacity: int, d_measure_z: int, d_status: int | None = None,
                           stream: int | None = None):
        """... for a hit list in HBM (gw_boost_screen_device's layout); {measure, z} per record go to d_measure_z."""
        _check(self._L.gw_boost_gtest_device(self._h, d_list, capacity, d_measure_z, d_status, stream))

    def last_boost_kernel(self) -> str:
        return self._L.gw_last_boost_kernel(self._h).decode()

    def replicate(self, comm: "GwComm", root: int = 0, stream: int | None = None):
        _check(self._L.gw_table_replicate(self._h, comm._h, root, stream))

    def boost_measures(self, a_begin: int, n_a: int, b_begin: int, n_b: int) -> np.ndarray:
        out = np.zeros((n_a, n_b), dtype=np.float64)
        _check(self._L.gw_boost_measures(self._h, a_begin, n_a, b_begin, n_b,
                                         out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def download_selected(self, snp_begin: int, n_rows: int):
        cb, tb = C.c_uint32(0), C.c_uint32(0)
        _check(self._L.gw_table_download_selected(self._h, snp_begin, 0, None, C.byref(cb), C.byref(tb)))
        rows = np.zeros((n_rows, 2 * (cb.value + tb.value)), dtype=np.uint16)
        _check(self._L.gw_table_download_selected(self._h, snp_begin, n_rows, _p16(rows), C.byref(cb), C.byref(tb)))
        return rows, cb.value, tb.value

    # ---- measurement
    def timer_start(self, stream: int | None = None):
        _check(self._L.gw_timer_start(self._h, stream))

    def timer_stop(self, stream: int | None = None) -> float:
        ms = C.c_float(0)
        _check(self._L.gw_timer_stop(self._h, stream, C.byref(ms)))
        return ms.value

    def stream_read_probe(self, stream: int | None = None) -> int:
        b = C.c_uint64(0)
        _check(self._L.gw_stream_read_probe(self._h, stream, C.byref(b)))
        return b.value

    def synchronize(self):
        _check(self._L.gw_synchronize(self._h))

    def stream(self) -> int:
        s = C.c_void_p()
        _check(self._L.gw_table_stream(self._h, C.byref(s)))
        return s.value or 0

    def last_scan_kernel(self) -> str:
        return self._L.gw_last_scan_kernel(self._h).decode()

    def set_option(self, option: int, value: int):
        _check(self._L.gw_set_option(self._h, option, value))

    def get_option(self, option: int) -> int:
        v = C.c_int64(0)
        _check(self._L.gw_get_option(self._h, option, C.byref(v)))
        return v.value
