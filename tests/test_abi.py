"""CPU: the C-ABI library loads, exports every symbol include/otter_gpu.h declares, the ctypes/numpy mirrors
have the C layout, and — on a box without a GPU — the product fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest
import otter_amd
from otter_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "otter_gpu.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(otg_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = otter_amd.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert getattr(lib, n) is not None, n
    assert sorted(otter_amd.EXPORTS) == names


def test_struct_layouts_match_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "otter_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(otg_params),sizeof(otg_align_task),sizeof(otg_read),sizeof(otg_region),sizeof(otg_allele),'
                   'sizeof(otg_region_result),sizeof(otg_poa_member),sizeof(otg_poa_graph),sizeof(otg_run_stats),'
                   'offsetof(otg_allele,se),offsetof(otg_read,ccoord_second));'
                   'printf("%zu %zu %zu %zu\\n",sizeof(otg_bed),sizeof(otg_read_meta),sizeof(otg_ingest_opts),offsetof(otg_ingest_opts,read_quality));return 0;}\n')
    exe = tmp_path / "sz"
    # the boundary is a plain C header: strict C99, no warnings (a type used before its declaration would only warn)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(abi.otg_params), abi.align_task_dt.itemsize, abi.read_dt.itemsize, abi.region_dt.itemsize, abi.allele_dt.itemsize,
           abi.region_result_dt.itemsize, abi.poa_member_dt.itemsize, abi.poa_graph_dt.itemsize, abi.run_stats_dt.itemsize,
           abi.allele_dt.fields["se"][1], abi.read_dt.fields["ccoord_second"][1],
           abi.bed_dt.itemsize, abi.read_meta_dt.itemsize, abi.ingest_opts_dt.itemsize, abi.ingest_opts_dt.fields["read_quality"][1]]
    assert got == exp


def test_default_params_match_reference_cli():
    lib = otter_amd.load()
    p = abi.otg_params()
    lib.otg_params_default(C.byref(p))
    q = abi.default_params()
    for name, _ in abi.otg_params._fields_:
        assert getattr(p, name) == getattr(q, name), name
    assert (p.max_alleles, p.max_cov, p.flank, p.mismatch, p.gap_open, p.gap_ext) == (2, 200, 100, 4, 6, 2)
    assert (p.max_error, p.bandwidth_short, p.bandwidth_long, p.min_sim) == (0.01, 0.01, 0.015, 0.9)


def test_no_silent_cpu_fallback():
    if otter_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(otter_amd.OtterGpuError) as e:
        otter_amd.Context(0)
    assert "no HIP device" in str(e.value) and "no CPU fallback" in str(e.value)
    lib = otter_amd.load()
    lib.otg_edit_distance_batch.restype = C.c_int
    assert lib.otg_edit_distance_batch(None, None, C.c_uint64(0), None, C.c_uint32(1), None, None) == abi.OTG_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "otter_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".inc")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_lib" not in txt and "libotter_oracle" not in txt and "otter_oracle" not in txt, os.path.join(dp, f)
