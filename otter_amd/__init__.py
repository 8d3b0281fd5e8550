"""otter_amd — MI355X-native drop-in for otter's per-region alignment / clustering / consensus hot path.

The product is libotter_gpu.so (hand-written HIP for gfx950 behind the C-ABI of include/otter_gpu.h);
this package is the thin host-side mirror used by tests and bench.py.  No CPU fallback exists."""
from . import abi, synth  # noqa: F401
from ._lib import Context, Comm, OtterGpuError, device_count, load, LIB_PATH, EXPORTS, emit_alleles, emit_sam_header, Bam  # noqa: F401
from ._lib import Fasta, parse_bed_file, bed_tuples, emit_reads, assemble_files, assemble_files_release, assemble_batch_plan, genotype_files, wgat  # noqa: F401
from ._lib import emit_vcf_header, emit_vcf_lines, emit_genotype_lengths, genotype_blocks  # noqa: F401
from .abi import default_params  # noqa: F401
