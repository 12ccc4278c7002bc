"""alga_amd -- MI355X (gfx950) overlap-graph engine for the ALGA assembler.

The product is the C-ABI shared library `alga_amd/lib/libalga_amd.so` (include/alga_amd.h) built from
the hand-written HIP kernels under alga_amd/csrc/.  This Python package is a thin ctypes binding used by
the tests, bench.py and the multi-GPU driver; torch supplies device memory, streams and
torch.distributed only.  There is no CPU fallback: importing works anywhere, computing needs the GPU.
"""
from .engine import (Engine, MultiEngine, AlgaError, PrefSufParams, load_library, library_path, pack_reads, derive_params, ingest_files, parse_files,  # noqa: F401
                     EDGE_DTYPE)
