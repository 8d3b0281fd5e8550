"""Writes tests/golden/wgat_small.{bam,bam.bai,bed,sam.txt,fa.txt}: a small synthetic whole-genome-alignment BAM and the text the REFERENCE's
own wgat() (oracle/_ref/libotter_ref_io.so, built from /root/reference/src/wgat.cpp where it lies) prints for it.  Run in the build container."""
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_wgat
tmp = tempfile.mkdtemp()
bam, bed = test_wgat._make_wga(tmp, 11, n_contigs=12, n_beds=60)
g = os.path.join(ROOT, "tests", "golden")
shutil.copy(bam, os.path.join(g, "wgat_small.bam")); shutil.copy(bam + ".bai", os.path.join(g, "wgat_small.bam.bai")); shutil.copy(bed, os.path.join(g, "wgat_small.bed"))
for fasta, name in ((False, "wgat_small.sam.txt"), (True, "wgat_small.fa.txt")):
    open(os.path.join(g, name), "wb").write(test_wgat._ref_wgat(os.path.join(g, "wgat_small.bam"), os.path.join(g, "wgat_small.bed"), "asm1", fasta, 1, 0))
print("ok", os.path.getsize(os.path.join(g, "wgat_small.bam")))
