"""The host-side readers of the library (BAM / BAI ingest, allele ingest, otg_wgat, BED parser, FASTA index + fetch) built with
-fsanitize=address,undefined from the same sources and run over corrupted inputs: records mutated INSIDE the BGZF blocks (the blocks
still inflate, so the garbage reaches the record parser), truncated files, mutated BAI / FAI / BED files.  Every call must come back (OTG_OK
or an error code) with no sanitizer report.  CPU only; needs hipcc (host-only compile) — skipped when it is absent."""
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    if not (os.path.exists(HIPCC) and os.path.exists(CLANGXX)):
        pytest.skip("hipcc / clang++ not available")
    d = str(tmp_path_factory.mktemp("san"))
    flags = ["--cuda-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    objs, procs = [], []
    for f in ("ingest", "bedfa", "emit", "otg_api"):
        o = os.path.join(d, f + ".o"); objs.append(o)
        procs.append(subprocess.Popen([HIPCC] + flags + ["-fPIC", "-c", os.path.join(ROOT, "otter_amd", "csrc", f + ".hip"), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    o = os.path.join(d, "fuzz.o"); objs.append(o)
    procs.append(subprocess.Popen([HIPCC] + flags + ["-x", "hip", "-c", os.path.join(ROOT, "tools", "fuzz_host_io.cpp"), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()[-2000:]
    exe = os.path.join(d, "fuzz_host_io")
    # the objects reference the kernel launchers of the other translation units; the harness never calls them
    subprocess.check_call([CLANGXX, "-fsanitize=address,undefined"] + objs + ["-o", exe, "-lz", "-pthread", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                                                                               "-Wl,--unresolved-symbols=ignore-all"])
    return exe


def _bgzf_blocks(raw):
    out, p = [], 0
    while p + 18 <= len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        out.append(zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15))
        p += bsize
    return out


def _bgzf_write(path, blocks):
    with open(path, "wb") as f:
        for data in blocks:
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = co.compress(data) + co.flush()
            f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25))
            f.write(comp)
            f.write(struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def _bed_for(bam_path, out_path, rng):
    """Regions on the BAM's own targets (so that the readers do reach its records): windows where alignments are, and one region per whole target."""
    import otter_amd
    b = otter_amd.Bam(bam_path)
    tg = b.targets()
    b.close()
    with open(out_path, "w") as f:
        for name, ln in tg:
            f.write("%s\t0\t%d\n" % (name, min(ln, 2_000_000)))
            for _ in range(25):
                s0 = int(rng.integers(0, max(1, min(ln, 500_000) - 10)))
                f.write("%s\t%d\t%d\n" % (name, s0, s0 + int(rng.integers(1, 3000))))
    return out_path


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe] + list(args), capture_output=True, timeout=300, env=env)
    err = p.stderr.decode(errors="replace")
    assert "AddressSanitizer" not in err and "runtime error" not in err and p.returncode == 0, (p.returncode, err[-3000:])
    return p.stdout.decode()


def test_sanitized_readers_on_clean_inputs(fuzz_bin):
    rng = np.random.default_rng(1)
    for which in ("ingest_small.bam", "wgat_small.bam", "genotype_small.bam"):
        bed = _bed_for(os.path.join(GOLD, which), os.path.join(os.path.dirname(fuzz_bin), which + ".bed"), rng)
        out = _run(fuzz_bin, os.path.join(GOLD, which), bed)
        assert "done" in out and " reads 0," not in out, out               # the readers did reach the records


@pytest.mark.parametrize("which", ["ingest_small.bam", "wgat_small.bam", "genotype_small.bam"])
def test_sanitized_readers_on_mutated_records(fuzz_bin, tmp_path, which):
    rng = np.random.default_rng(len(which))
    raw = open(os.path.join(GOLD, which), "rb").read()
    blocks = _bgzf_blocks(raw)
    bed = _bed_for(os.path.join(GOLD, which), str(tmp_path / "regions.bed"), rng)
    for trial in range(14):
        mut = [bytearray(b) for b in blocks]
        for _ in range(int(rng.integers(1, 30))):
            bi = int(rng.integers(1 if len(mut) > 2 else 0, len(mut)))      # mostly past the header block
            if len(mut[bi]) == 0:
                continue
            pos = int(rng.integers(0, len(mut[bi])))
            kind = trial % 4
            if kind == 0:
                mut[bi][pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                mut[bi][pos] = int(rng.integers(0, 256))
            elif kind == 2:                                              # a plausible but wrong 32-bit field
                mut[bi][pos:pos + 4] = struct.pack("<I", int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000, 65536, 1 << 29])))[:max(0, min(4, len(mut[bi]) - pos))]
            else:
                del mut[bi][pos:pos + int(rng.integers(1, 40))]
        p = str(tmp_path / ("m%d.bam" % trial))
        _bgzf_write(p, [bytes(b) for b in mut])
        shutil.copy(os.path.join(GOLD, which) + ".bai", p + ".bai")
        if trial % 5 == 4:                                                # and a damaged index
            bai = bytearray(open(p + ".bai", "rb").read())
            for _ in range(6):
                bai[int(rng.integers(4, len(bai)))] = int(rng.integers(0, 256))
            open(p + ".bai", "wb").write(bytes(bai))
        _run(fuzz_bin, p, bed)
    # truncations of the compressed file
    for trial in range(6):
        p = str(tmp_path / ("t%d.bam" % trial))
        open(p, "wb").write(raw[:int(rng.integers(30, len(raw)))])
        shutil.copy(os.path.join(GOLD, which) + ".bai", p + ".bai")
        _run(fuzz_bin, p, bed)


def test_sanitized_readers_on_mutated_text_inputs(fuzz_bin, tmp_path):
    rng = np.random.default_rng(5)
    bam = os.path.join(GOLD, "ingest_small.bam")
    bed_raw = open(_bed_for(bam, str(tmp_path / "clean.bed"), rng), "rb").read()
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as f:
        for c in ("chr1", "chrBig"):
            f.write(">%s some text\n" % c)
            for _ in range(40):
                f.write("".join("ACGTNacgt"[int(x)] for x in rng.integers(0, 9, 60)) + "\n")
    _run(fuzz_bin, bam, str(tmp_path / "clean.bed"), fa)                       # writes r.fa.fai
    fai_raw = open(fa + ".fai", "rb").read()
    for trial in range(12):
        b = bytearray(bed_raw)
        for _ in range(8):
            b[int(rng.integers(0, len(b)))] = int(rng.choice(list(b"\t\n:-09azAZ#\x00\xff ")))
        bp = str(tmp_path / ("b%d.bed" % trial)); open(bp, "wb").write(bytes(b))
        fi = bytearray(fai_raw)
        for _ in range(4):
            fi[int(rng.integers(0, len(fi)))] = int(rng.choice(list(b"\t\n09-a\x00")))
        open(fa + ".fai", "wb").write(bytes(fi))
        _run(fuzz_bin, bam, bp, fa)
