"""Per-region digests of hot-path results (test infrastructure, shared by scripts/make_golden_digests.py — which fills them from the
CPU oracle in the build container — and tests/test_gpu_digests.py, which recomputes them from the GPU results on the same inputs).

A digest holds, per region: status, ic, fc, number of allele records and a hash of the region's final per-read labels; per allele
record: region, label, length, scov / acov / tcov, ic, PS / HP, the bit pattern of the float `se` and the first 16 bytes of the
SHA-256 of the sequence.  Everything is integer data, so equality is bit-exactness (north_star asks 1e-6 on `se`; the oracle and the
device agree to the bit, and the test says so if that ever stops being true)."""
import hashlib

import numpy as np

REGION_FIELDS = ("status", "ic", "fc", "n_alleles")
ALLELE_FIELDS = ("region", "label", "seq_len", "scov", "acov", "tcov", "ic", "ps", "hp")


def _sha16(b):
    return np.frombuffer(hashlib.sha256(b).digest()[:16], dtype=np.uint8)


def digest(res, batch, lo, hi):
    """Digest of regions [lo, hi) of a result dict (Context.assemble_collect / oracle_lib.assemble_batch layout: `regions` indexed by
    batch region, `alleles` with absolute .region)."""
    reg = res["regions"]
    first = batch["regions"]["first_read"].astype(np.int64)
    nrd = batch["regions"]["n_reads"].astype(np.int64)
    out = {k: np.asarray(reg[k][lo:hi], dtype=np.int64) for k in REGION_FIELDS}
    out["labels_sha"] = np.stack([_sha16(np.ascontiguousarray(res["labels"][first[r]:first[r] + nrd[r]], dtype=np.int32).tobytes())
                                  for r in range(lo, hi)]) if hi > lo else np.zeros((0, 16), np.uint8)
    rows, shas, sebits = [], [], []
    for r in range(lo, hi):
        g = reg[r]
        if int(g["n_alleles"]) == 0:
            continue
        for a in res["alleles"][int(g["first_allele"]):int(g["first_allele"]) + int(g["n_alleles"])]:
            assert int(a["region"]) == r, (int(a["region"]), r)
            rows.append([int(a[k]) for k in ALLELE_FIELDS])
            sebits.append(np.float32(a["se"]).view(np.uint32))
            shas.append(_sha16(res["seqs"][int(a["seq_off"]):int(a["seq_off"]) + int(a["seq_len"])].tobytes()))
    out["alleles"] = np.array(rows, dtype=np.int64).reshape(-1, len(ALLELE_FIELDS))
    out["se_bits"] = np.array(sebits, dtype=np.uint32)
    out["seq_sha"] = np.stack(shas) if shas else np.zeros((0, 16), np.uint8)
    return out


def compare(got, want, what=""):
    """Raises AssertionError naming the first region that differs; returns the number of regions and allele records compared."""
    for k in REGION_FIELDS:
        bad = np.nonzero(got[k] != want[k])[0]
        assert bad.size == 0, "%s: region field %s differs in %d regions, first at region %d (got %d, oracle %d)" % (
            what, k, bad.size, int(bad[0]), int(got[k][bad[0]]), int(want[k][bad[0]]))
    bad = np.nonzero((got["labels_sha"] != want["labels_sha"]).any(axis=1))[0]
    assert bad.size == 0, "%s: final read labels differ in %d regions, first at region %d" % (what, bad.size, int(bad[0]))
    assert got["alleles"].shape == want["alleles"].shape, "%s: %d allele records, oracle %d" % (what, len(got["alleles"]), len(want["alleles"]))
    bad = np.nonzero((got["alleles"] != want["alleles"]).any(axis=1))[0]
    assert bad.size == 0, "%s: allele record fields differ in %d records, first: region %d got %s oracle %s" % (
        what, bad.size, int(want["alleles"][bad[0]][0]), got["alleles"][bad[0]].tolist(), want["alleles"][bad[0]].tolist())
    bad = np.nonzero((got["seq_sha"] != want["seq_sha"]).any(axis=1))[0]
    assert bad.size == 0, "%s: allele sequences differ in %d records, first in region %d" % (what, bad.size, int(want["alleles"][bad[0]][0]))
    if not np.array_equal(got["se_bits"], want["se_bits"]):
        g, w = got["se_bits"].view(np.float32), want["se_bits"].view(np.float32)
        assert np.allclose(g, w, atol=1e-6, rtol=0), "%s: se differs by more than 1e-6" % what
    return len(want["status"]), len(want["alleles"])
