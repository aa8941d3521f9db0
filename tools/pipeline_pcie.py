#!/usr/bin/env python3
"""PCIe-inclusive text -> text rate (development aid, GPU only; never bench.py's `value`).

FASTQ text sits in pinned host memory (what a reader thread leaves after decompression), goes to HBM in
chunks, is parsed, assembled and formatted on the device, and the FASTA text comes back to pinned host
memory.  Two input and two output buffers: chunk k+1 is copied in and chunk k-1's text copied out while
chunk k is computed (copy streams + events; the parse / format calls synchronise their own stream only).

    python tools/pipeline_pcie.py [--pairs-per-chunk 1000000] [--chunks 8] [--config 1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs-per-chunk", type=int, default=1_000_000)
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_next
    import pandaseq_amd as pa
    from pandaseq_amd import synth, textio

    dev = torch.device("cuda:0")
    spec = synth.config_spec(args.config)
    stride = spec.row_stride()
    base_n = min(args.pairs_per_chunk, 250_000)
    f, r, fl, rl = bench_next.synth_pairs(spec, base_n)
    ft, rt = synth.render_fastq(f, r, fl, rl)
    reps = max(1, args.pairs_per_chunk // base_n)
    n = base_n * reps
    # the "file": one chunk's worth of text, pinned; every chunk re-reads it (content does not matter for the rate)
    host_f = torch.frombuffer(bytearray(ft), dtype=torch.uint8).repeat(reps).pin_memory()
    host_r = torch.frombuffer(bytearray(rt), dtype=torch.uint8).repeat(reps).pin_memory()
    d_f = [torch.empty_like(host_f, device=dev) for _ in range(2)]
    d_r = [torch.empty_like(host_r, device=dev) for _ in range(2)]
    out_cap = n * (2 * spec.read_len + 96)
    d_out = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    h_out = [torch.empty(out_cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
    rows = dict(
        fwd=torch.empty((n, stride, 2), dtype=torch.uint8, device=dev), rev=torch.empty((n, stride, 2), dtype=torch.uint8, device=dev),
        flen=torch.zeros(n, dtype=torch.int16, device=dev), rlen=torch.zeros(n, dtype=torch.int16, device=dev),
        ids=torch.zeros((n, 48), dtype=torch.uint8, device=dev))
    outs = pa.alloc_outputs(n, n * 420 + (1 << 27), dev)
    parser, formatter = textio.FastqParser(0), textio.Formatter(0)
    asm, _ = bench.setup_assembler(pa, synth, args.config, 0, exact_sums=False)
    s_in, s_out, s_cmp = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    def copy_in(k):
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_cmp[k % 2])  # the previous user of this buffer pair has been formatted
            d_f[k % 2].copy_(host_f, non_blocking=True)
            d_r[k % 2].copy_(host_r, non_blocking=True)
            ev_in[k % 2].record(s_in)

    def run(chunks):
        total_ok = 0
        total_bytes = 0
        copy_in(0)
        for k in range(chunks):
            if k + 1 < chunks:
                copy_in(k + 1)
            s_cmp.wait_event(ev_in[k % 2])
            _, st = parser.parse(d_f[k % 2], d_r[k % 2], n, stride, out=rows)
            assert st.n_good == n and st.stop_code == 0
            b = pa.DeviceBatch(n=n, fwd=rows["fwd"], rev=rows["rev"], flen=rows["flen"], rlen=rows["rlen"], **outs)
            b.cursor.zero_()
            b.counters.zero_()
            asm.assemble_device(b)
            s_cmp.wait_event(ev_out[k % 2])  # this output buffer's previous copy-out is done
            _, nbytes, fix = formatter.format(b, rows["ids"], d_f[k % 2], textio.OUT_FASTA, out=d_out[k % 2])
            ev_cmp[k % 2].record(s_cmp)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_cmp[k % 2])
                h_out[k % 2][:nbytes].copy_(d_out[k % 2][:nbytes], non_blocking=True)
                ev_out[k % 2].record(s_out)
            total_bytes += nbytes
            total_ok += 1
        torch.cuda.synchronize()
        return total_bytes

    run(2)  # warm up
    t0 = time.perf_counter()
    out_bytes = run(args.chunks)
    dt = time.perf_counter() - t0
    pairs = n * args.chunks
    text_in = (host_f.numel() + host_r.numel()) * args.chunks
    res = {
        "what": "pinned host FASTQ text -> H2D -> parse -> assemble (fast sums) -> FASTA -> D2H -> pinned host text, double-buffered",
        "pairs_per_chunk": n, "chunks": args.chunks, "pairs_per_s": pairs / dt, "seconds": dt,
        "h2d_GBps": text_in / dt / 1e9, "d2h_GBps": out_bytes / dt / 1e9,
        "text_in_bytes_per_pair": text_in / pairs, "text_out_bytes_per_pair": out_bytes / pairs,
        "first_fasta_line": bytes(h_out[(args.chunks - 1) % 2][:60].numpy().tobytes()).split(b"\n")[0].decode(),
    }
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
