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


RESULT_MAGIC = b"FSPO0001"


def read_results(path):
    """Packed result file written by `FamSeq PL -binOutput`:
      0 "FSPO0001"  8 uint32 n_seq  12 uint32 reserved  16 uint64 n_sites  24 n_seq x char[32] names
      then blocks:  uint64 n | status[n] u8 | gpp[n][n_seq][3] f64 | fpp[n][n_seq][3] f64 | fgt[n][n_seq] i8
    -> dict(names, status[S], gpp[S,n_seq,3], fpp[S,n_seq,3], fgt[S,n_seq])."""
    with open(path, "rb") as f:
        head = f.read(24)
        if head[:8] != RESULT_MAGIC:
            raise ValueError("%s is not a packed result file" % path)
        k = int(np.frombuffer(head[8:12], "<u4")[0])
        total = int(np.frombuffer(head[16:24], "<u8")[0])
        names = [f.read(32).rstrip(b"\0").decode() for _ in range(k)]
        st, gpp, fpp, fgt = [], [], [], []
        while True:
            b = f.read(8)
            if len(b) < 8:
                break
            n = int(np.frombuffer(b, "<u8")[0])
            st.append(np.frombuffer(f.read(n), "u1"))
            gpp.append(np.frombuffer(f.read(n * k * 24), "<f8").reshape(n, k, 3))
            fpp.append(np.frombuffer(f.read(n * k * 24), "<f8").reshape(n, k, 3))
            fgt.append(np.frombuffer(f.read(n * k), "i1").reshape(n, k))
    out = dict(names=names, status=np.concatenate(st) if st else np.zeros(0, "u1"),
               gpp=np.concatenate(gpp) if gpp else np.zeros((0, k, 3)), fpp=np.concatenate(fpp) if fpp else np.zeros((0, k, 3)),
               fgt=np.concatenate(fgt) if fgt else np.zeros((0, k), "i1"))
    if out["status"].shape[0] != total:
        raise ValueError("%s: header says %d sites, blocks hold %d" % (path, total, out["status"].shape[0]))
    return out


def write_results(path, names, status, gpp, fpp, fgt, block=65536):
    """The writer side of read_results (what `FamSeq PL -binOutput` produces), in blocks of `block` sites."""
    k, s = len(names), len(status)
    gpp = np.ascontiguousarray(gpp, "<f8").reshape(s, k, 3)
    fpp = np.ascontiguousarray(fpp, "<f8").reshape(s, k, 3)
    fgt = np.ascontiguousarray(fgt, "i1").reshape(s, k)
    status = np.ascontiguousarray(status, "u1")
    with open(path, "wb") as f:
        f.write(RESULT_MAGIC)
        f.write(np.array([k, 0], "<u4").tobytes())
        f.write(np.array([s], "<u8").tobytes())
        for n in names:
            f.write(n.encode()[:31].ljust(32, b"\0"))
        for lo in range(0, s, block):
            hi = min(s, lo + block)
            f.write(np.array([hi - lo], "<u8").tobytes())
            f.write(status[lo:hi].tobytes())
            f.write(gpp[lo:hi].tobytes())
            f.write(fpp[lo:hi].tobytes())
            f.write(fgt[lo:hi].tobytes())
