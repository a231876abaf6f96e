"""Packed PL files (.fspl): the text-free site feed.

Layout (little endian; written by `FamSeq pack`, read by `FamSeq PL` and by this module):
  0   "FSPL0001"            8   uint32 n_seq        12  uint32 record_bytes = 1 + 6*n_seq
  16  uint64 n_sites        24  n_seq x char[32] sample names (NUL padded)
  then n_sites records:  uint8 flags (bit0 Known, bit1 chrX), uint16 pl[n_seq][3]
PL is the integer Phred-scaled likelihood clamped to 65534; 0xFFFF x3 = sample missing.
"""
import numpy as np

MAGIC = b"FSPL0001"


def record_dtype(n_seq):
    return np.dtype([("flags", "u1"), ("pl", "<u2", (n_seq, 3))])


def write_plfile(path, names, flags, pl):
    """names[n_seq]; flags uint8 [S]; pl uint16 [S, n_seq, 3]."""
    pl = np.ascontiguousarray(pl, dtype="<u2")
    s, k = pl.shape[0], pl.shape[1]
    assert k == len(names) and pl.shape[2] == 3 and len(flags) == s
    rec = np.zeros(s, dtype=record_dtype(k))
    rec["flags"] = flags
    rec["pl"] = pl
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([k, 1 + 6 * k], "<u4").tobytes())
        f.write(np.array([s], "<u8").tobytes())
        for n in names:
            f.write(n.encode()[:31].ljust(32, b"\0"))
        f.write(rec.tobytes())


def read_plfile(path, mmap=True):
    """-> (names, flags[S] uint8, pl[S, n_seq, 3] uint16).  With mmap=True the arrays are views
    of a memory map (records are interleaved, so they are strided views)."""
    with open(path, "rb") as f:
        head = f.read(24)
        if head[:8] != MAGIC:
            raise ValueError("%s is not a packed PL file" % path)
        k, rec = np.frombuffer(head[8:16], "<u4")
        n = int(np.frombuffer(head[16:24], "<u8")[0])
        if rec != 1 + 6 * k:
            raise ValueError("bad record size in %s" % path)
        names = [f.read(32).rstrip(b"\0").decode() for _ in range(k)]
    off = 24 + 32 * int(k)
    dt = record_dtype(int(k))
    if mmap:
        data = np.memmap(path, dtype=dt, mode="r", offset=off, shape=(n,))
    else:
        data = np.fromfile(path, dtype=dt, offset=off, count=n)
    return names, data["flags"], data["pl"]
