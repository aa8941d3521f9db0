"""The library's two link modes against the reference's sources.

standalone   libpandaseq_amd.so          -- self-contained: every panda_* name it uses it also defines
reflink      libpandaseq_amd_reflink.so  -- built with -DPAMD_WITH_REFERENCE_OBJECTS, as a PANDAseq maintainer links it INTO
             libpandaseq after dropping assembler.c, assembler_support.c and module.c from Makefile.am:102-140: it must define
             exactly what those three files define (so nothing is defined twice and nothing a retained file calls is missing) and
             may leave undefined only names the RETAINED files define.

What each reference file defines is tests/golden/ref_exports.json (tools/ref_exports.py: names per file -- the reference cannot
be compiled here, every .c includes autoconf's config.h); where /root/reference exists the fixture is re-derived and compared.
The last test runs the reflink library for real, without a GPU: the per-pair client linked against it with the algorithm
objects coming from ANOTHER library, as they come from the reference's algo*.c in a maintainer's link."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import per_pair as pp

REPLACED = ("assembler.c", "assembler_support.c", "module.c")
# internal (non-panda_) names of the replaced files that only the replaced files themselves call (assembler.c:259,339,
# assembler_support.c:168; module_init has no caller at all): nothing that stays in the link needs them
INTERNAL_TO_REPLACED = {"assemble_seq", "module_checkseq", "module_precheckseq", "module_destroy", "module_init"}
# declared by the public headers, defined by no file of the reference (pandaseq-assembler.h:96, pandaseq-module.h:94): ours in every build
DECLARED_ONLY = {"panda_assembler_open", "panda_module_get_args"}


def _fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_exports.json")))


def _nm(lib):
    out = subprocess.check_output(["nm", "-D", lib], text=True)
    defined, undefined = set(), set()
    for line in out.splitlines():
        parts = line.split()
        if len(parts) == 2 and parts[0] in ("U", "w"):
            undefined.add(parts[1].split("@")[0])
        elif len(parts) == 3 and parts[1] in "TDBRV":
            defined.add(parts[2].split("@")[0])
    return defined, undefined


def _ours(names):
    """the names that belong to the reference's interface: panda_* and module.c's module_show_all"""
    return {n for n in names if n.startswith("panda_") or n.startswith("module_")}


@pytest.fixture(scope="module")
def reflink(built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pandaseq_amd", "csrc"), "reflink"])
    return os.path.join(ROOT, "pandaseq_amd", "libpandaseq_amd_reflink.so")


def test_fixture_is_what_the_reference_sources_say():
    if not os.path.isdir("/root/reference"):
        pytest.skip("/root/reference is not on this machine: the committed fixture stands")
    import ref_exports
    fx = _fixture()
    assert ref_exports.exports("/root/reference") == fx["defines"]
    assert ref_exports.declared_public("/root/reference") == fx["declared_public"]


def test_reflink_defines_exactly_the_replaced_files(reflink):
    fx = _fixture()["defines"]
    replaced = set().union(*(fx[f] for f in REPLACED)) - INTERNAL_TO_REPLACED
    retained = set().union(*(v for f, v in fx.items() if f not in REPLACED))
    defined, undefined = _nm(reflink)
    assert _ours(defined) == replaced | DECLARED_ONLY, (sorted(_ours(defined) - replaced - DECLARED_ONLY), sorted(replaced - _ours(defined)))
    assert not (defined & retained), "defined twice in a maintainer's link: %s" % sorted(defined & retained)
    need = _ours(undefined)
    assert need <= retained, "undefined and not provided by a retained file: %s" % sorted(need - retained)
    # what it asks of the reference is the algorithm objects (identity + parameters, algo.c:44-48, pandaseq-algorithm.h:117,170),
    # the reader constructor its own panda_assembler_new_fastq_reader() wraps, and panda_result_phred for min_phred behind a host module
    assert {"panda_algorithm_is_a", "panda_algorithm_simple_bayes_class", "panda_algorithm_pear_class", "panda_algorithm_rdp_mle_class",
            "panda_algorithm_simple_bayes_get_error_estimation", "panda_algorithm_pear_get_random_base_log_p", "panda_create_fastq_reader"} <= need
    by_file = {f for f, v in fx.items() if need & set(v)}
    assert by_file <= {"algo.c", "algo_simple_bayes.c", "algo_pear.c", "algo_rdp_mle.c", "algo_uparse.c", "algo_flash.c", "algo_ea_util.c", "algo_stitch.c",
                       "nt.c", "fastq.c", "fileio.c"}, by_file


def test_standalone_is_self_contained(built):
    defined, undefined = _nm(os.path.join(ROOT, "pandaseq_amd", "libpandaseq_amd.so"))
    assert not _ours(undefined), sorted(_ours(undefined))
    fx = _fixture()["defines"]
    replaced = set().union(*(fx[f] for f in REPLACED)) - INTERNAL_TO_REPLACED
    assert replaced - {"module_show_all"} <= defined, sorted(replaced - defined)   # (args.c's helper: only a link with args.c needs it)
    # every panda_* name it exports is one the reference declares in a public header or defines in a library source (the class
    # objects and the registry array are data the headers reach through macros; fileio.c:139 defines panda_assembler_open_fastq
    # where the header says panda_assembler_open): no invented panda_* names
    known = set(_fixture()["declared_public"]) | set().union(*fx.values())
    extra = {n for n in defined if n.startswith("panda_")} - known
    assert not extra, sorted(extra)


def test_only_two_sources_know_the_flag():
    src = os.path.join(ROOT, "pandaseq_amd", "csrc")
    users = sorted(f for f in os.listdir(src) if f.endswith((".cpp", ".hip", ".h")) and "PAMD_WITH_REFERENCE_OBJECTS" in open(os.path.join(src, f)).read())
    assert users == ["pamd_compat.cpp", "pamd_compat_fastq.cpp"]    # = REFLINK_SRCS of the Makefile


@pytest.mark.parametrize("algo,options,kw", [
    ("simple_bayesian", [], {}),
    ("pear", ["-o", "9", "-t", "0.9"], dict(minoverlap=9, threshold=0.9)),
    ("rdp_mle", ["-m", "min_overlapbits:15"], dict(modules=[("min_overlapbits", 15)])),
])
def test_reflink_runs_with_algorithm_objects_from_another_library(reflink, tmp_path, algo, options, kw):
    """The per-pair client linked `-lpandaseq_amd_reflink -lpandaseq_amd`: the assembler and the modules are the reflink
    library's (first in the search order), the algorithm objects -- which it does not define -- are found in the second
    library, the way the reference's algo*.c provide them in a maintainer's libpandaseq.  Host path, no GPU needed."""
    from oracle import pyoracle as po
    from pandaseq_amd import synth
    libdir = os.path.join(ROOT, "pandaseq_amd")
    exe = str(tmp_path / "per_pair_reflink")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_per_pair.c"), "-o", exe,
                           "-L", libdir, "-Wl,--no-as-needed", "-lpandaseq_amd_reflink", "-lpandaseq_amd", "-ldl", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    fwd, rev, flen, rlen = synth.generate_numpy(synth.config_spec(1), 7000, 300)
    if algo == "pear":
        rlen = np.minimum(rlen, flen)
    pairs, out = tmp_path / "pairs.bin", tmp_path / "out.bin"
    pp.write_pairs(pairs, fwd, rev, flen, rlen)
    env = dict(os.environ, LD_DEBUG="bindings", LD_DEBUG_OUTPUT=str(tmp_path / "ld"))
    records, counters = pp.run_client(exe, pairs, out, algo, "host", options, env=env)
    modules = kw.pop("modules", ())
    o = po.assemble_batch(po.make_config(algo, modules=modules, **kw), fwd, rev, flen, rlen)
    pp.compare_with_oracle(records, counters, *o, modules=bool(modules))
    assert int(o[3]["ok"]) > 100
    # who served what: the loader's own record of the bindings
    log = "".join(open(os.path.join(str(tmp_path), f)).read() for f in os.listdir(str(tmp_path)) if f.startswith("ld."))
    def bound_to(symbol):
        return {line.split(" to ")[1].split(" ")[0].split("[")[0] for line in log.splitlines() if "`%s'" % symbol in line and " to " in line}
    assert all(t.endswith("libpandaseq_amd_reflink.so") for t in bound_to("panda_assembler_assemble")) and bound_to("panda_assembler_assemble")
    assert all(t.endswith("libpandaseq_amd.so") for t in bound_to("panda_algorithm_is_a")) and bound_to("panda_algorithm_is_a")
