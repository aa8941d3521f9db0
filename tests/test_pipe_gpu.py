"""End to end through the C ABI only: tests/c/pamd_pipe.c (plain C, the HIP runtime for buffers and copies,
no Python / torch in the data path) turns two FASTQ files into a FASTA / FASTQ file; the bytes must equal the
oracle chain's (reader -> assembler -> output.c formatting)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from pandaseq_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGOS = ["simple_bayesian", "pear", "rdp_mle", "uparse", "flash", "ea_util", "stitch"]


@pytest.fixture(scope="module")
def pipe(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pipe") / "pamd_pipe")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c", "pamd_pipe.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _oracle_text(ft, rt, algo, fmt, stride=456, **kw):
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt, stride=stride)
    res, nt, p, ctr = po.assemble_batch(po.make_config(algo, **kw), pf, pr, pfl, prl)
    parts = []
    for i in range(st.n_pairs):
        if res["status"][i] == 0:
            m = int(res["seq_len"][i])
            parts.append(po.output_fasta(ids[i], float(res["quality"][i]), nt[i, :m]) if fmt == "fasta"
                         else po.output_fastq(ids[i], float(res["quality"][i]), nt[i, :m], p[i, :m]))
    return b"".join(parts), ctr, st


def _run(pipe, tmp_path, ft, rt, algo, fmt, chunk, extra=()):
    f, r, o = tmp_path / "f.fastq", tmp_path / "r.fastq", tmp_path / "out.txt"
    f.write_bytes(ft)
    r.write_bytes(rt)
    out = subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO[algo]), fmt, str(chunk), *extra], timeout=300)
    stat = dict(line.split("\t")[1:3] for line in out.decode().splitlines() if line.startswith("STAT"))
    stop = [line.split("\t")[1:] for line in out.decode().splitlines() if line.startswith("STOP")][0]
    return o.read_bytes(), {k: int(v) for k, v in stat.items()}, (int(stop[0]), int(stop[1]))


@pytest.mark.parametrize("algo", ALGOS)
def test_files_to_fasta_every_scorer(pipe, tmp_path, algo):
    ft, rt = util.random_fastq(2500, 41)
    # chunks of 700 pairs: records straddle every chunk boundary and the tails are carried over
    got, stat, stop = _run(pipe, tmp_path, ft, rt, algo, "fasta", 700)
    want, ctr, st = _oracle_text(ft, rt, algo, "fasta")
    assert got == want
    assert stat["READS"] == int(ctr["count"]) and stat["OK"] == int(ctr["ok"]) and stat["NOALGN"] == int(ctr["noalgn"])
    assert stat["LOWQ"] == int(ctr["lowq"]) and stat["BADR"] == int(ctr["badr"]) and stop == (0, 0)


def test_files_to_fastq_config1(pipe, tmp_path):
    spec = synth.config_spec(1)
    f, r, fl, rl = synth.generate_host(spec, 0, 6000)
    ft, rt = synth.render_fastq(f, r, fl, rl, tag="TTAGGC")
    got, stat, stop = _run(pipe, tmp_path, ft, rt, "simple_bayesian", "fastq", 2048)
    want, ctr, st = _oracle_text(ft, rt, "simple_bayesian", "fastq")
    assert got == want and stat["OK"] == int(ctr["ok"]) and len(want) > 1_000_000


@pytest.mark.parametrize("after", [0, 1])
def test_files_with_primers(pipe, tmp_path, after):
    spec = synth.config_spec(2)
    f, r, fl, rl = synth.generate_host(spec, 0, 3000)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    got, stat, stop = _run(pipe, tmp_path, ft, rt, "pear", "fasta", 1000, extra=(str(after), spec.fprimer, spec.rprimer))
    want, ctr, st = _oracle_text(ft, rt, "pear", "fasta", fprimer=synth.nt_codes(spec.fprimer),
                                 rprimer=synth.nt_codes(spec.rprimer, complement=True), primers_after=bool(after))
    assert got == want
    assert stat["NOFP"] == int(ctr["nofp"]) and stat["NORP"] == int(ctr["norp"]) and stat["OK"] == int(ctr["ok"])


def test_reader_stops_where_the_reference_stops(pipe, tmp_path):
    """A bad record in the middle of the second chunk: everything before it is written, then the reader's code."""
    ft, rt = util.random_fastq(1500, 43)
    recs = ft.split(b"\n@")  # not exact for '@' in qualities, so rebuild from the oracle's view instead
    pf, pr, pfl, prl, ids, st = po.fastq_read(ft, rt)
    assert st.stop_code == 0
    # corrupt record 1100's forward sequence line with a digit
    lines = ft.split(b"\n")
    k = 4 * 1100 + 1
    lines[k] = (b"1" + lines[k][1:]) if lines[k] else b"1"
    bad = b"\n".join(lines)
    got, stat, stop = _run(pipe, tmp_path, bad, rt, "simple_bayesian", "fasta", 400)
    want, ctr, st2 = _oracle_text(bad, rt, "simple_bayesian", "fasta")
    assert po.FQ_STOP[st2.stop_code] in ("BADNT", "NOQUAL") and st2.n_records == 1100
    assert got == want and stop == (st2.stop_code, 1100)


def test_pump_with_unequal_files(pipe, tmp_path):
    """What pamd_pipe_run() gathers on the device, chunk after chunk, when the two files do not move in step: forward reads of
    ~300 bases against reverse reads of ~60 (the forward text is consumed four times as fast as the reverse: the reverse side's
    unconsumed tail grows with every block until the gathers balance it), small chunks, many blocks; then a forward file with
    more records than the reverse one (the reader stops where the shorter file ends, fastq.c:123,130); then empty files."""
    rng = np.random.default_rng(11)
    n = 9000
    fo, ro = [], []
    for i in range(n):
        h = f"@M1:7:FC:1:{1101 + i % 40}:{i}:{7 * i % 9973}"
        lf, lr = int(rng.integers(280, 301)), int(rng.integers(40, 80))
        for hh, ln, dst in ((h + " 1:N:0:ACGT", lf, fo), (h + " 2:N:0:ACGT", lr, ro)):
            seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ln).tobytes()
            dst.append(hh.encode() + b"\n" + seq + b"\n+\n" + bytes([int(q) for q in rng.integers(35, 74, size=ln)]) + b"\n")
    ft, rt = b"".join(fo), b"".join(ro)
    for chunk in (300, 2500):
        got, stat, stop = _run(pipe, tmp_path, ft, rt, "simple_bayesian", "fasta", chunk)
        want, ctr, st = _oracle_text(ft, rt, "simple_bayesian", "fasta")
        assert got == want and stat["READS"] == int(ctr["count"]) == n and stop == (0, 0), chunk
    # 500 forward records without a mate: nothing of them is assembled, nothing fails
    cut = b"".join(ro[:n - 500])
    got, stat, stop = _run(pipe, tmp_path, ft, cut, "simple_bayesian", "fasta", 1000)
    want, ctr, st = _oracle_text(ft, cut, "simple_bayesian", "fasta")
    assert got == want and stat["READS"] == n - 500 == int(ctr["count"]) and stop == (0, 0)
    # ... and a reverse file that ends inside a record
    part = cut[:-30]
    got, stat, stop = _run(pipe, tmp_path, ft, part, "simple_bayesian", "fasta", 1000)
    want, ctr, st = _oracle_text(ft, part, "simple_bayesian", "fasta")
    assert got == want and stat["READS"] == int(ctr["count"]) == n - 501
    got, stat, stop = _run(pipe, tmp_path, b"", b"", "simple_bayesian", "fasta", 1000)
    assert got == b"" and stat["READS"] == 0 and stop == (0, 0)
    got, stat, stop = _run(pipe, tmp_path, ft, b"", "pear", "fastq", 1000)
    assert got == b"" and stat["READS"] == 0


def test_pump_fails_cleanly(pipe, tmp_path):
    """pamd_pipe_run() on files that are not there, or an output that cannot be made: an error code and a message, no threads
    left behind (the client exits non-zero, promptly)."""
    ft, rt = util.random_fastq(200, 3)
    f, r = tmp_path / "f.fastq", tmp_path / "r.fastq"
    f.write_bytes(ft)
    r.write_bytes(rt)
    for args in ((str(tmp_path / "missing.fastq"), str(r), str(tmp_path / "o.txt")), (str(f), str(tmp_path / "missing.fastq"), str(tmp_path / "o.txt")),
                 (str(f), str(r), str(tmp_path / "no_such_dir" / "o.txt"))):
        out = subprocess.run([pipe, *args, "0", "fasta", "1000"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 3 and "cannot" in out.stderr, (args, out.returncode, out.stderr[-300:])
    out = subprocess.run([pipe, str(f), str(r), str(tmp_path / "o.txt"), "0", "fasta", "4000000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "3 000 000" in out.stderr     # more pairs per chunk than a text chunk below 4 GiB can hold
    # an input that fails in mid-stream -- gzip text with a damaged block, read by ONE thread that nobody else would ever wake for: the
    # failure is recorded, every waiter is notified under its own lock, and the run returns the I/O error instead of hanging
    import gzip
    big_f, big_r = util.random_fastq(40_000, 5)
    def damaged(text):
        z = bytearray(gzip.compress(text, 1))
        for k in range(len(z) // 2, len(z) // 2 + 64):
            z[k] ^= 0x5A
        return bytes(z)
    (tmp_path / "f.gz").write_bytes(gzip.compress(big_f, 1))
    (tmp_path / "r.gz").write_bytes(gzip.compress(big_r, 1))
    (tmp_path / "f_bad.gz").write_bytes(damaged(big_f))
    (tmp_path / "r_bad.gz").write_bytes(damaged(big_r))
    for args in ((str(tmp_path / "f_bad.gz"), str(tmp_path / "r.gz")), (str(tmp_path / "f.gz"), str(tmp_path / "r_bad.gz"))):
        out = subprocess.run([pipe, *args, str(tmp_path / "o.txt"), "0", "fasta", "1000"], capture_output=True, text=True, timeout=120)
        # (zlib reports the damage -- an I/O error, exit 3 -- unless what it inflated before noticing already stopped the reader at a record)
        assert (out.returncode == 3 and "gzread" in out.stderr) or (out.returncode == 0 and "STOP\t0\t" not in out.stdout), (out.returncode, out.stderr[-300:])


# ---- the reference's own file-level API: panda_assembler_open() + panda_assembler_next() ------------------
@pytest.fixture(scope="module")
def open_client(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("open") / "test_open_fastq")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "test_open_fastq.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "pandaseq_amd"), "-lpandaseq_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "pandaseq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _run_open(exe, tmp_path, ft, rt, algo, gz=False):
    import bz2
    import gzip
    f, r, o = tmp_path / ("f.fastq.gz" if gz else "f.fastq"), tmp_path / ("r.fastq.gz" if gz else "r.fastq"), tmp_path / "out.fasta"
    pack = {False: lambda t: t, True: lambda t: gzip.compress(t, 1), "bz2": bz2.compress}[gz]    # (the file's name says nothing: fileio.c:70-96 goes by its first two bytes)
    f.write_bytes(pack(ft))
    r.write_bytes(pack(rt))
    out = subprocess.check_output([exe, str(f), str(r), str(o), algo], timeout=600).decode()
    stat = {line.split("\t")[1]: int(line.split("\t")[2]) for line in out.splitlines() if line.startswith("STAT")}
    stop = int([line.split("\t")[1] for line in out.splitlines() if line.startswith("STOP")][0])
    return o.read_bytes(), stat, stop


@pytest.mark.parametrize("algo,gz", [("simple_bayesian", False), ("pear", True), ("uparse", False), ("rdp_mle", "bz2")])
def test_reference_file_api(open_client, tmp_path, algo, gz):
    """panda_assembler_open(forward, reverse, ...) on plain, gzip and bzip2 files, then the pool.c:76 loop: the FASTQ text is
    parsed on the device a chunk at a time behind the per-pair PandaNextSeq contract, names included."""
    ft, rt = util.random_fastq(3000, 47)
    got, stat, stop = _run_open(open_client, tmp_path, ft, rt, algo, gz)
    want, ctr, st = _oracle_text(ft, rt, algo, "fasta")
    assert got == want
    assert stat["READS"] == int(ctr["count"]) and stat["OK"] == int(ctr["ok"]) and stat["NOALGN"] == int(ctr["noalgn"])
    assert stat["LOWQ"] == int(ctr["lowq"]) and stat["BADR"] == int(ctr["badr"]) and stop == 0


def test_reference_file_api_many_chunks_and_stop(open_client, tmp_path):
    """More pairs than one refill holds (65 536), and a record the reader must stop at."""
    spec = synth.config_spec(0)
    f, r, fl, rl = synth.generate_host(spec, 0, 150_000)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    lines = rt.split(b"\n")
    lines[4 * 140_000 + 3] = lines[4 * 140_000 + 3][:-1]  # one quality too few in the reverse read of record 140000
    rt_bad = b"\n".join(lines)
    got, stat, stop = _run_open(open_client, tmp_path, ft, rt_bad, "simple_bayesian")
    want, ctr, st = _oracle_text(ft, rt_bad, "simple_bayesian", "fasta", stride=152)
    assert st.n_records == 140_000 and po.FQ_STOP[st.stop_code] == "NOQUAL"
    assert got == want and stat["READS"] == 140_000 and stop == st.stop_code


def test_reference_file_api_with_reference_headers(built, tmp_path):
    """The same client compiled against the reference's OWN pandaseq.h (oracle/_ref/refopen)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "refopen")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time: make -C oracle ref)")
    ft, rt = util.random_fastq(1200, 49)
    got, stat, stop = _run_open(exe, tmp_path, ft, rt, "rdp_mle")
    want, ctr, st = _oracle_text(ft, rt, "rdp_mle", "fasta")
    assert got == want and stat["OK"] == int(ctr["ok"])


@pytest.mark.parametrize("which", ["both", "forward", "reverse"])
def test_gzip_inputs(pipe, tmp_path, which):
    """fileio.c:86-96 opens every input through zlib; the pump does when a file begins with the gzip magic: one reader inflating in
    order, the size known only at the end, blocks of 64 KiB (chunks of 700 pairs), several gzip members in one file (zlib reads
    them on, as `cat a.gz b.gz` needs)."""
    import gzip
    ft, rt = util.random_fastq(9000, 43)
    want, ctr, st = _oracle_text(ft, rt, "simple_bayesian", "fasta")
    f, r, o = tmp_path / "f.fastq.gz", tmp_path / "r.fastq", tmp_path / "out.txt"
    half = ft.index(b"\n@", len(ft) // 2) + 1
    f.write_bytes((gzip.compress(ft[:half]) + gzip.compress(ft[half:])) if which != "reverse" else ft)
    r.write_bytes(gzip.compress(rt, 1) if which != "forward" else rt)
    out = subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["simple_bayesian"]), "fasta", "700"], timeout=300).decode()
    assert o.read_bytes() == want
    stat = {k: int(v) for k, v in (line.split("\t")[1:3] for line in out.splitlines() if line.startswith("STAT"))}
    assert stat["READS"] == int(ctr["count"]) and stat["OK"] == int(ctr["ok"])


def test_inputs_from_fifos(pipe, tmp_path):
    """Inputs that are not regular files (`-f <(bzcat x.bz2)`): read in order, no size to go by."""
    import threading
    ft, rt = util.random_fastq(6000, 44)
    want, ctr, st = _oracle_text(ft, rt, "pear", "fastq")
    f, r, o = tmp_path / "f.fifo", tmp_path / "r.fifo", tmp_path / "out.txt"
    os.mkfifo(f)
    os.mkfifo(r)

    def feed(path, text):
        with open(path, "wb") as h:
            for at in range(0, len(text), 50_000):
                h.write(text[at:at + 50_000])
    feeders = [threading.Thread(target=feed, args=(f, ft)), threading.Thread(target=feed, args=(r, rt))]
    for t in feeders:
        t.start()
    subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["pear"]), "fastq", "1000"], timeout=300)
    for t in feeders:
        t.join()
    assert o.read_bytes() == want


def test_truncated_gzip_is_the_end_of_the_input(pipe, tmp_path):
    """A gzip stream cut short: zlib hands out what it could inflate and then reports the end of the file, and fileio.c:31-46 takes
    that for the end of the input (gzread() < 1, gzeof() true) -- so does the pump: the pairs before the cut, then a record
    without its end."""
    import gzip
    import zlib
    ft, rt = util.random_fastq(4000, 45)
    f, r, o = tmp_path / "f.fastq.gz", tmp_path / "r.fastq.gz", tmp_path / "out.txt"
    cut = gzip.compress(ft)[:-2000]
    f.write_bytes(cut)
    r.write_bytes(gzip.compress(rt))
    seen = zlib.decompressobj(31).decompress(cut)
    assert 0 < len(seen) < len(ft)
    want, ctr, st = _oracle_text(seen, rt, "simple_bayesian", "fasta")
    subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["simple_bayesian"]), "fasta", "700"], timeout=300)
    assert o.read_bytes() == want and len(want) > 100_000


@pytest.mark.parametrize("streams", [1, 2])
def test_bzip2_inputs(pipe, tmp_path, streams):
    """fileio.c:75-85: an input that begins with "BZ" is read through libbz2 (bound at run time here).  Two streams in one file
    (what pbzip2 writes): BZ2_bzRead() stops at the end of the first and fileio.c:48-58 takes the next call's error for the end
    of the input -- the pairs of the first stream, and nothing of the second, here as there."""
    import bz2
    ft, rt = util.random_fastq(5000, 46)
    half = ft.index(b"\n@", len(ft) // 2) + 1
    f, r, o = tmp_path / "f.fastq.bz2", tmp_path / "r.fastq.bz2", tmp_path / "out.txt"
    f.write_bytes(bz2.compress(ft) if streams == 1 else bz2.compress(ft[:half]) + bz2.compress(ft[half:]))
    r.write_bytes(bz2.compress(rt))
    want, ctr, st = _oracle_text(ft if streams == 1 else ft[:half], rt, "simple_bayesian", "fasta")
    subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["simple_bayesian"]), "fasta", "700"], timeout=300)
    assert o.read_bytes() == want and len(want) > 100_000


def test_output_into_a_pipe(pipe, tmp_path):
    """An output that cannot seek (`... /dev/stdout | bzip2 > merged.bz2`): one writer, the pieces in order, plain write()."""
    import threading
    ft, rt = util.random_fastq(6000, 48)
    want, ctr, st = _oracle_text(ft, rt, "simple_bayesian", "fastq")
    f, r, o = tmp_path / "f.fastq", tmp_path / "r.fastq", tmp_path / "out.fifo"
    f.write_bytes(ft)
    r.write_bytes(rt)
    os.mkfifo(o)
    got = []
    reader = threading.Thread(target=lambda: got.append(open(o, "rb").read()))
    reader.start()
    subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["simple_bayesian"]), "fastq", "700"], env=dict(os.environ, PAMD_PIPE_WRITERS="3"), timeout=300)
    reader.join()
    assert got[0] == want and len(want) > 1_000_000


def test_reader_stops_early_in_streamed_inputs(pipe, tmp_path):
    """The reader gives up at a record far from the end of gzip inputs much longer than what has been read: what came before is
    written, the streaming readers are wound down (no further block is inflated for nobody) and the run ends."""
    import gzip
    ft, rt = util.random_fastq(40_000, 50)
    lines = ft.split(b"\n")
    k = 4 * 3100 + 1
    lines[k] = b"1" + lines[k][1:]
    bad = b"\n".join(lines)
    f, r, o = tmp_path / "f.fastq.gz", tmp_path / "r.fastq.gz", tmp_path / "out.txt"
    f.write_bytes(gzip.compress(bad, 1))
    r.write_bytes(gzip.compress(rt, 1))
    out = subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO["simple_bayesian"]), "fasta", "500"], timeout=120).decode()
    stop = [line.split("\t")[1:] for line in out.splitlines() if line.startswith("STOP")][0]
    want, ctr, st = _oracle_text(bad, rt, "simple_bayesian", "fasta")
    assert st.n_records == 3100 and (int(stop[0]), int(stop[1])) == (st.stop_code, 3100)
    assert o.read_bytes() == want


# ---- pamd_pipe_run_multi(): the same run dealt to N devices (pool.c:110-181's shape, a device per worker) ------------------
def _run_full(pipe, tmp_path, ft, rt, algo, fmt, chunk, devices=None, extra=(), out_name="out.txt"):
    """The client's output file and everything it prints but the timings and the number of chunks."""
    f, r, o = tmp_path / "f.fastq", tmp_path / "r.fastq", tmp_path / out_name
    f.write_bytes(ft)
    r.write_bytes(rt)
    env = dict(os.environ)
    env.pop("PAMD_PIPE_DEVICES", None)
    if devices is not None:
        env["PAMD_PIPE_DEVICES"] = devices
    out = subprocess.check_output([pipe, str(f), str(r), str(o), str(po.ALGO[algo]), fmt, str(chunk), *extra], timeout=300, env=env).decode()
    lines = [line for line in out.splitlines() if not line.startswith("PIPE")]
    pipe_line = [line for line in out.splitlines() if line.startswith("PIPE")][0].split("\t")
    return o.read_bytes(), lines, dict(zip(pipe_line[1::2], pipe_line[2::2]))


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
@pytest.mark.parametrize("algo,fmt", [("simple_bayesian", "fasta"), ("pear", "fastq")])
def test_multi_device_run_equals_the_single_device_run(pipe, tmp_path, devices, algo, fmt):
    """Output bytes, STAT lines, the reference-format STAT block and all 916 counters of pamd_pipe_run_multi() over devices
    {0}, {0, 0}, {0, 0, 0} (workers that share the one GPU of this pool, each with its own assembler, streams and buffers)
    equal pamd_pipe_run()'s -- and the oracle chain's.  Chunks of 700 of 5 000 pairs: eight chunks dealt round robin."""
    ft, rt = util.random_fastq(5000, 61)
    want_bytes, want_lines, _ = _run_full(pipe, tmp_path, ft, rt, algo, fmt, 700)
    got_bytes, got_lines, info = _run_full(pipe, tmp_path, ft, rt, algo, fmt, 700, devices=devices, out_name="multi.txt")
    assert got_bytes == want_bytes and got_lines == want_lines
    assert int(info["chunks"]) == 8 and int(info["pairs"]) == 5000
    oracle_bytes, ctr, st = _oracle_text(ft, rt, algo, fmt)
    assert got_bytes == oracle_bytes
    block = [line for line in got_lines if line.startswith("pamd\tSTAT")]
    assert block[0] == "pamd\tSTAT\tREADS\t%d" % int(ctr["count"]) and block[-1].startswith("pamd\tSTAT\tOVERLAPS\t")
    assert [int(x) for x in block[-1].split("\t")[3].split(" ")] == [int(x) for x in ctr["overlapcount"][:int(ctr["longest_overlap"]) + 1]]


def test_multi_device_run_with_primers_and_unequal_records(pipe, tmp_path):
    """config 2 (primers before and after assembly), then records of very different sizes in the two files (the chunk boundaries
    are found per file by counting lines) and a forward file with more records than the reverse one."""
    spec = synth.config_spec(2)
    f, r, fl, rl = synth.generate_host(spec, 0, 3000)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    for after in (0, 1):
        extra = (str(after), spec.fprimer, spec.rprimer)
        want = _run_full(pipe, tmp_path, ft, rt, "pear", "fasta", 1000, extra=extra)
        got = _run_full(pipe, tmp_path, ft, rt, "pear", "fasta", 1000, devices="0,0", extra=extra, out_name="multi.txt")
        assert got[0] == want[0] and got[1] == want[1] and len(got[0]) > 100_000
    rng = np.random.default_rng(12)
    n = 6000
    fo, ro = [], []
    for i in range(n):
        h = f"@M1:7:FC:1:{1101 + i % 40}:{i}:{7 * i % 9973}"
        lf, lr = int(rng.integers(280, 301)), int(rng.integers(40, 80))
        for hh, ln, dst in ((h + " 1:N:0:ACGT", lf, fo), (h + " 2:N:0:ACGT", lr, ro)):
            seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ln).tobytes()
            dst.append(hh.encode() + b"\n" + seq + b"\n+\n" + bytes([int(q) for q in rng.integers(35, 74, size=ln)]) + b"\n")
    ft, rt = b"".join(fo), b"".join(ro)
    for rev in (rt, b"".join(ro[:n - 500]), b"".join(ro[:n - 500])[:-30], b""):
        want = _run_full(pipe, tmp_path, ft, rev, "simple_bayesian", "fasta", 900)
        got = _run_full(pipe, tmp_path, ft, rev, "simple_bayesian", "fasta", 900, devices="0,0,0", out_name="multi.txt")
        assert got[0] == want[0] and got[1] == want[1], len(rev)
    # both files end inside a record: the reader runs out of lines in read_seq(): PANDA_CODE_PREMATURE_EOF (8), either way
    for devices in (None, "0,0"):
        got = _run_full(pipe, tmp_path, ft[:-40], rt[:-25], "simple_bayesian", "fasta", 900, devices=devices)
        assert "STOP\t8\t%d" % (n - 1) in got[1], got[1][:9]


def test_multi_device_run_stops_where_the_reader_stops(pipe, tmp_path):
    """A record the reader gives up at, in the third of eight chunks: the chunks dealt beyond it (some of them assembled before the
    bad one was parsed) leave nothing behind -- neither text nor counts."""
    ft, rt = util.random_fastq(4000, 67)
    lines = ft.split(b"\n")
    k = 4 * 1300 + 1
    lines[k] = (b"1" + lines[k][1:]) if lines[k] else b"1"
    bad = b"\n".join(lines)
    want = _run_full(pipe, tmp_path, bad, rt, "simple_bayesian", "fasta", 500)
    for devices in ("0,0", "0,0,0"):
        got = _run_full(pipe, tmp_path, bad, rt, "simple_bayesian", "fasta", 500, devices=devices, out_name="multi.txt")
        assert got[0] == want[0] and got[1] == want[1], devices
    assert [line for line in want[1] if line.startswith("STOP")][0].split("\t")[2] == "1300"
    oracle_bytes, ctr, st = _oracle_text(bad, rt, "simple_bayesian", "fasta")
    assert want[0] == oracle_bytes and st.n_records == 1300


def test_multi_device_run_into_a_pipe_and_from_gzip(pipe, tmp_path):
    """An output that cannot seek takes the pieces of all devices in offset order from one writer; inputs that are read in order
    (gzip) go to the first device's pamd_pipe_run()."""
    import gzip
    ft, rt = util.random_fastq(3000, 71)
    want = _run_full(pipe, tmp_path, ft, rt, "simple_bayesian", "fastq", 400)
    f, r = tmp_path / "f.fastq", tmp_path / "r.fastq"
    env = dict(os.environ, PAMD_PIPE_DEVICES="0,0")
    fifo = tmp_path / "out.fifo"
    os.mkfifo(fifo)
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    reader.start()
    subprocess.check_call([pipe, str(f), str(r), str(fifo), "0", "fastq", "400"], timeout=300, env=env, stdout=subprocess.DEVNULL)
    reader.join()
    assert got[0] == want[0]
    # ... and one that is no file at all
    subprocess.check_call([pipe, str(f), str(r), "/dev/null", "0", "fastq", "400"], timeout=300, env=env, stdout=subprocess.DEVNULL)
    (tmp_path / "f.gz").write_bytes(gzip.compress(ft, 1))
    (tmp_path / "r.gz").write_bytes(gzip.compress(rt, 1))
    subprocess.check_call([pipe, str(tmp_path / "f.gz"), str(tmp_path / "r.gz"), str(tmp_path / "gz.txt"), "0", "fastq", "400"], timeout=300, env=env, stdout=subprocess.DEVNULL)
    assert (tmp_path / "gz.txt").read_bytes() == want[0]


def test_multi_device_run_fails_cleanly(pipe, tmp_path):
    ft, rt = util.random_fastq(200, 3)
    f, r = tmp_path / "f.fastq", tmp_path / "r.fastq"
    f.write_bytes(ft)
    r.write_bytes(rt)
    env = dict(os.environ, PAMD_PIPE_DEVICES="0,0")
    for args in ((str(tmp_path / "missing.fastq"), str(r), str(tmp_path / "o.txt")), (str(f), str(r), str(tmp_path / "no_such_dir" / "o.txt"))):
        out = subprocess.run([pipe, *args, "0", "fasta", "1000"], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 3 and "cannot" in out.stderr, (args, out.returncode, out.stderr[-300:])
    out = subprocess.run([pipe, str(f), str(r), str(tmp_path / "o.txt"), "0", "fasta", "1000"], capture_output=True, text=True, timeout=120, env=dict(os.environ, PAMD_PIPE_DEVICES="0,99"))
    assert out.returncode == 3, out.stderr[-300:]    # no device 99
