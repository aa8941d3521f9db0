#!/usr/bin/env python3
"""Throughput of the two data-format rows either side of the hot path, GPU only:
FASTQ text -> batch rows, and accepted pairs -> FASTA / FASTQ text, with the text resident in HBM.

    python tools/bench_io.py [--pairs 4000000] [--config 1] [--steps 3] [--out profiles/rNN_io_bench.json]

One JSON object: per stage the pairs/s, the kernels' HIP-event time, the algorithmic bytes per pair
and the fraction of the 8 TB/s HBM roofline they amount to; the end-to-end text -> text rate of the
three stages back to back.  `python bench.py --io` runs the same legs and adds the CPU baseline (the
oracle's reader and formatter on a bounded sample of the same text).

Algorithmic bytes (the minimum a pass must move):
  ingest   both texts once + one row of 2 bytes/base per read + lengths (4 B) + the parsed header (48 B)
  format   the result record (48 B) + the parsed header (48 B) + the header characters copied + 1 B
           (FASTA) or 9 B (FASTQ: nt + f64 log-p) per merged base in, + the text out
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0


def add_args(ap):
    ap.add_argument("--pairs", type=int, default=4_000_000, help="pairs per text chunk (a chunk must stay below 4 GiB)")
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)


def run(args):
    """The GPU legs.  Returns (result dict, context for a CPU baseline leg: bench.py --io adds that one, because
    only bench.py's cpu_baseline leg and tests/ may call the oracle)."""
    import torch

    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_next
    import pandaseq_amd as pa
    from pandaseq_amd import synth, textio

    dev = torch.device("cuda:0")
    spec = synth.config_spec(args.config)
    L = spec.read_len
    stride = spec.row_stride()
    # the text: a 250k-pair rendering of the config's synthetic pairs, repeated (names repeat; parsing does not care)
    base_n = min(args.pairs, 250_000)
    f, r, fl, rl = bench_next.synth_pairs(spec, base_n)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    reps = max(1, args.pairs // base_n)
    n = base_n * reps
    ftd = textio.text_to_device(ft).repeat(reps)
    rtd = textio.text_to_device(rt).repeat(reps)
    assert ftd.numel() < (1 << 32) and rtd.numel() < (1 << 32), "chunk too large: lower --pairs"
    parser = textio.FastqParser(0)
    formatter = textio.Formatter(0)
    rows = dict(
        fwd=torch.empty((n, stride, 2), dtype=torch.uint8, device=dev), rev=torch.empty((n, stride, 2), dtype=torch.uint8, device=dev),
        flen=torch.zeros(n, dtype=torch.int16, device=dev), rlen=torch.zeros(n, dtype=torch.int16, device=dev),
        ids=torch.zeros((n, 48), dtype=torch.uint8, device=dev))
    asm, _ = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=False)
    outs = pa.alloc_outputs(n, n * 420 + (1 << 27), dev)
    text_out = torch.empty(n * (2 * (2 * L) + 96), dtype=torch.uint8, device=dev)

    def ingest():
        t0 = time.perf_counter()
        _, st = parser.parse(ftd, rtd, n, stride, out=rows)
        wall = time.perf_counter() - t0
        assert st.n_good == n and st.stop_code == 0, st.as_dict()
        k = parser.last_kernel_ms()
        return wall, k["index_ms"], k["parse_ms"]

    def assemble():
        b = pa.DeviceBatch(n=n, fwd=rows["fwd"], rev=rows["rev"], flen=rows["flen"], rlen=rows["rlen"], **outs)
        b.cursor.zero_()
        b.counters.zero_()
        t0 = time.perf_counter()
        asm.assemble_device(b)
        torch.cuda.synchronize()
        return b, time.perf_counter() - t0, asm.last_kernel_ms()

    def fmt(b, which):
        t0 = time.perf_counter()
        _, nbytes, fix = formatter.format(b, rows["ids"], ftd, which, out=text_out)
        return time.perf_counter() - t0, formatter.last_kernel_ms(), nbytes, len(fix)

    # warm up every stage once, then time
    ingest()
    b, _, _ = assemble()
    fmt(b, textio.OUT_FASTA)
    fmt(b, textio.OUT_FASTQ)
    ing = [ingest() for _ in range(args.steps)]
    asms = [assemble() for _ in range(args.steps)]
    b = asms[-1][0]
    fas = [fmt(b, textio.OUT_FASTA) for _ in range(args.steps)]
    fqs = [fmt(b, textio.OUT_FASTQ) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ingest()
        bb, _, _ = assemble()
        fmt(bb, textio.OUT_FASTA)
    e2e = (time.perf_counter() - t0) / args.steps

    res = b.results_numpy()
    okm = res["status"] == 0
    merged = int(res["seq_len"][okm].sum())
    n_ok = int(okm.sum())
    text_in = ftd.numel() + rtd.numel()
    ingest_bytes = text_in + 2 * 2 * L * n + 4 * n + 48 * n
    best = lambda xs, i: min(x[i] for x in xs)  # noqa: E731
    idx_ms, par_ms = best(ing, 1), best(ing, 2)
    fa_ms, fa_bytes = best(fas, 1), fas[0][2]
    fq_ms, fq_bytes = best(fqs, 1), fqs[0][2]
    hdr_chars = fa_bytes - merged - 2 * n_ok - 10 * n_ok  # id characters copied from the forward text (approx.)
    fa_algo = 96 * n + hdr_chars + merged + fa_bytes
    fq_algo = 96 * n + hdr_chars + 9 * merged + fq_bytes
    out = {
        "workload": f"{n} synthetic 2x{L} bp pairs as CASAVA 1.7 FASTQ text in HBM ({text_in / 1e9:.2f} GB of text), config {args.config}",
        "ingest": {
            "pairs_per_s": n / ((idx_ms + par_ms) / 1e3), "wall_pairs_per_s": n / best(ing, 0),
            "index_ms": idx_ms, "parse_ms": par_ms, "bytes_per_pair": ingest_bytes / n,
            "roofline": {"bound": "hbm", "achieved": ingest_bytes / ((idx_ms + par_ms) / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ingest_bytes / ((idx_ms + par_ms) / 1e3) / 1e9 / HBM_PEAK_GBPS},
        },
        "assemble": {"pairs_per_s": n / (best(asms, 2) / 1e3), "kernel_ms": best(asms, 2)},
        "format_fasta": {
            "pairs_per_s": n / (fa_ms / 1e3), "kernel_ms": fa_ms, "text_bytes": fa_bytes, "fixups": fas[0][3], "bytes_per_pair": fa_algo / n,
            "roofline": {"bound": "hbm", "achieved": fa_algo / (fa_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fa_algo / (fa_ms / 1e3) / 1e9 / HBM_PEAK_GBPS},
        },
        "format_fastq": {
            "pairs_per_s": n / (fq_ms / 1e3), "kernel_ms": fq_ms, "text_bytes": fq_bytes, "fixups": fqs[0][3], "bytes_per_pair": fq_algo / n,
            "roofline": {"bound": "hbm", "achieved": fq_algo / (fq_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fq_algo / (fq_ms / 1e3) / 1e9 / HBM_PEAK_GBPS},
        },
        "text_to_text": {"pairs_per_s": n / e2e, "ms": e2e * 1e3, "stages": "FASTQ parse -> assemble (fast sums) -> FASTA format, host wall clock, text resident in HBM"},
    }
    ctx = dict(ft=ft, rt=rt, base_n=base_n, stride=stride, rows=rows, res=res, okm=okm, batch=b, n=n)
    return out, ctx


def main():
    ap = argparse.ArgumentParser()
    add_args(ap)
    args = ap.parse_args()
    out, _ = run(args)
    s = json.dumps(out)
    print(s)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
