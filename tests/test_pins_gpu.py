"""What can be pinned on the reference's own text, held against the DEVICE: candidate sets (through overlaps_examined)
against k-mers emitted by the reference's FOREACH_KMER macros (oracle/ref_macros.c -> oracle/_ref/librefmacros.so, which
travels to the GPU box), and the six header vectors of the reference's check_parser.c through the device parser."""
import json
import os

import numpy as np
import pytest

import pins
from util import adversarial_pairs

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fast_path", [True, False])
def test_device_candidate_sets_against_reference_kmer_macros(built, fast_path):
    """every pair's overlaps_examined from the kernels (lean + general, and general alone)
    equals the count rebuilt from the reference macros' k-mer emissions (tests/pins.py)."""
    import torch

    import pandaseq_amd as pa
    from pandaseq_amd import synth
    if pins.lib() is None:
        pytest.skip("oracle/_ref/librefmacros.so not built")
    spec = synth.config_spec(1)
    fwd, rev, flen, rlen = adversarial_pairs(spec, 2500, seed=17)
    # tandem repeats and unrelated reads: many candidates, k-mers that occur dozens of times
    rng = np.random.default_rng(2)
    L = spec.read_len
    for i in range(0, 300, 2):
        unit = rng.choice(np.array([1, 2, 4, 8], np.uint8), size=int(rng.integers(1, 13)))
        rep = np.tile(unit, L // len(unit) + 5)
        fwd[i, :L, 0] = rep[:L]
        rev[i, :L, 0] = rep[3:3 + L]
    asm = pa.Assembler(0)
    asm.set_exact_sums(False)
    asm.set_fast_path(fast_path)
    b = pa.upload_batch(fwd, rev, flen, rlen, device="cuda:0", seq_stride=0)
    asm.assemble_device(b)
    torch.cuda.synchronize()
    res = b.results_numpy()
    checked = many = 0
    for i in range(len(res)):
        ex = pins.examined(fwd[i], rev[i], flen[i], rlen[i])
        if ex is None:
            assert res["status"][i] in (1, 4), i
            continue
        assert int(res["examined"][i]) == ex, (i, int(res["examined"][i]), ex)
        checked += 1
        many += ex > 4
    assert checked > 1800 and many > 150


def test_check_parser_vectors_through_the_device_parser(built):
    """The reference's own header-parser test data (check_parser.c:29-36, tests/golden/seqid_check_parser.json) fed to
    fq_ids_kernel: every field, the format and the mate number must come out as the reference's test expects."""
    from pandaseq_amd import textio
    with open(os.path.join(GOLDEN, "seqid_check_parser.json")) as fh:
        g = json.load(fh)
    assert len(g["cases"]) == 6

    def mate_of(h, d):   # the same header with the other mate number
        if h.endswith("/1") or h.endswith("/2"):
            return h[:-1] + str(d)
        head, _, tail = h.rpartition(" ")
        return head + " " + str(d) + tail[1:]

    fwd_text = b""
    rev_text = b""
    for c in g["cases"]:
        f = c["header"] if c["dir"] == 1 else mate_of(c["header"], 1)
        r = c["header"] if c["dir"] == 2 else mate_of(c["header"], 2)
        fwd_text += ("@" + f + "\nACGTACGTAC\n+\nIIIIIIIIII\n").encode()
        rev_text += ("@" + r + "\nACGTACGTAC\n+\nIIIIIIIIII\n").encode()
    parser = textio.FastqParser(0, 33, 2)   # PANDA_TAG_OPTIONAL, as check_parser.c passes it
    out, st = parser.parse(textio.text_to_device(fwd_text), textio.text_to_device(rev_text), 8, 16)
    assert int(st.n_good) == 6 and textio.FQ_STOP[st.stop_code] == "END"
    ids = out["ids"].cpu().numpy().view(textio.SEQID_DTYPE).reshape(-1)[:6]
    fmt_names = ["UNKNOWN", "SRA", "CASAVA_1_4", "CASAVA_1_7", "EBI_SRA", "CASAVA_CONVERTED"]
    for rec, c in zip(ids, g["cases"]):
        assert int(rec["flag"]) == 0
        assert list(textio.seqid_fields(rec, fwd_text)) == c["id"], c["header"]
        assert fmt_names[int(rec["format"])] == c["format"], c["header"]
    # the reverse file's headers are parsed by the same kernel and must pair up: a mate-1 header on both sides does not
    bad_rev = rev_text.replace(b" 2:N:0:1", b" 1:N:0:1", 1)
    out2, st2 = parser.parse(textio.text_to_device(fwd_text), textio.text_to_device(bad_rev), 8, 16)
    assert int(st2.n_good) == 0
    parser.close()
