"""ORACLE — TEST INFRASTRUCTURE ONLY.

Search-scheme tables that drive the oracle.  They are READ FROM THE REFERENCE'S OWN DATA: tests/golden/search_schemes/
is a verbatim copy of /root/reference/search_schemes (data, not code; tests/test_oracle_golden.py checks the copy
byte for byte where the reference is present, and every file in it is parsed by the reference's own
SearchScheme::readScheme in the golden vectors).  Only what exists nowhere as data is written out here, each with
the reference lines it restates:

* the two k = 1 tables where a hard-coded class differs from the data directory of the same name
  (OptimalKianfar, src/searchstrategy.h:3028-3030; MinUSearchStrategy, :3286-3288),
* the k-mer cut-offs of the hard-coded classes (src/searchstrategy.h:2907, :3009, :3091, :3193: 100; base class 20,
  :222; CustomSearchStrategy 50, :2308),
* how `-S columba` (DynamicColumbaStrategy, :3666-3736) and `-c <dir>` (DynamicCustomStrategy, :3744-3776) assemble
  their alternatives: scheme, mirror image (MultipleSchemes ctor :2468-2477), then the "middle" schemes.

The product library carries its own, independently typed tables (columba_amd/csrc/host_schemes.hpp);
tests/test_strategy_tables.py compares the two through the C-ABI.
"""
import os

SCHEME_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                          "search_schemes")


def _vec(tok):
    return [int(x) for x in tok[1:-1].split(",")]


def read_scheme_file(path):
    """SearchScheme::readScheme (src/search.h:684-711): one search "{pi} {L} {U}" per non-empty line"""
    out = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            assert len(t) == 3, (path, line)
            out.append((_vec(t[0]), _vec(t[1]), _vec(t[2])))
    return out


def mirror(scheme):
    """SearchScheme::mirrorPiStrings (src/search.h:488-493, :745-753)"""
    return [([len(pi) - 1 - p for p in pi], lo, up) for pi, lo, up in scheme]


def load_custom_dir(name, kmer_cutoff=50, dynamic=False, max_k=13):
    """`-c <dir> -nD` (CustomSearchStrategy::getSearchSchemeFromFolder, src/searchstrategy.cpp:1990-2117) or, with
    dynamic=True, `-c <dir>` (DynamicCustomStrategy: scheme + mirror image, base-class partitioning)."""
    base = name if os.path.isabs(name) else os.path.join(SCHEME_DIR, name)
    spec = {"kmer_cutoff": kmer_cutoff, "schemes": {}, "partition_params": {}}
    for k in range(1, max_k + 1):
        p = os.path.join(base, str(k), "searches.txt")
        if not os.path.exists(p):
            continue
        sch = read_scheme_file(p)
        spec["schemes"][k] = [sch, mirror(sch)] if dynamic else [sch]
        if dynamic:
            continue
        pp = {}
        ps = os.path.join(base, str(k), "static_partitioning.txt")
        if os.path.exists(ps):
            pp["begins"] = [float(x) for x in open(ps).readline().split()]
        pd = os.path.join(base, str(k), "dynamic_partitioning.txt")
        if os.path.exists(pd):
            with open(pd) as f:
                pp["seeding"] = [float(x) for x in f.readline().split()]
                pp["weights"] = [int(x) for x in f.readline().split()]
        if pp:
            spec["partition_params"][k] = pp
    if not spec["partition_params"]:
        del spec["partition_params"]
    return spec


def load_multiple_dir(name, max_k=13):
    """`-d <dir>` (MultipleSchemesStrategy::readSchemes, src/searchstrategy.h:2624-2660): <k>/scheme<i>.txt"""
    base = name if os.path.isabs(name) else os.path.join(SCHEME_DIR, name)
    spec = {"kmer_cutoff": 20, "schemes": {}}
    for k in range(1, max_k + 1):
        alts = []
        i = 1
        while os.path.exists(os.path.join(base, str(k), f"scheme{i}.txt")):
            alts.append(read_scheme_file(os.path.join(base, str(k), f"scheme{i}.txt")))
            i += 1
        if alts:
            spec["schemes"][k] = alts
    return spec


def _restrict(spec, ks):
    out = dict(spec)
    out["schemes"] = {k: v for k, v in spec["schemes"].items() if k in ks}
    if "partition_params" in spec:
        out["partition_params"] = {k: v for k, v in spec["partition_params"].items() if k in ks}
    return out


MULTIPLE_OPT = load_multiple_dir("multiple_opt")
# the hard-coded classes support 1..4 errors (getMaxSupportedDistance) with cut-off 100
KUCH1 = _restrict(load_custom_dir("kuch_k+1", 100), range(1, 5))
KUCH2 = _restrict(load_custom_dir("kuch_k+2", 100), range(1, 5))
KIANFAR = _restrict(load_custom_dir("kianfar", 100), range(1, 5))
KIANFAR["schemes"][1] = [[([0, 1], [0, 0], [0, 1]), ([1, 0], [0, 1], [0, 1])]]  # searchstrategy.h:3028-3030
O1STAR = _restrict(load_custom_dir("01star0", 100), range(1, 5))
# PigeonHoleSearchStrategy (searchstrategy.h:3221-3274): 1..4 errors, base-class partitioning and cut-off
PIGEON = {"kmer_cutoff": 20, "schemes": _restrict(load_custom_dir("pigeon"), range(1, 5))["schemes"]}
# MinUSearchStrategy (searchstrategy.h:3284-3389) = search_schemes/multiple_opt/individual_schemes/scheme1 except k = 1
MINU = {"kmer_cutoff": 20,
        "schemes": load_custom_dir(os.path.join("multiple_opt", "individual_schemes", "scheme1"))["schemes"]}
MINU["schemes"][1] = [[([0, 1], [0, 0], [0, 1]), ([1, 0], [0, 0], [0, 1])]]  # searchstrategy.h:3286-3288


def _columba():
    # DynamicColumbaStrategy::createDynamicColumbaStrategy (searchstrategy.h:3720-3735): minU up to 7 errors, the greedy schemes
    # of ColumbaSearchStrategy for 8 .. 13 (:3417-3658 = the data files search_schemes/pigeon_adapted/<k>)
    spec = {"kmer_cutoff": 20, "schemes": {}}
    for k, (sch,) in MINU["schemes"].items():
        spec["schemes"][k] = [sch, mirror(sch)]
    greedy = load_custom_dir("pigeon_adapted")["schemes"]
    for k in range(8, 14):
        (sch,) = greedy[k]
        spec["schemes"][k] = [sch, mirror(sch)]
    mid = {2: MULTIPLE_OPT["schemes"][2][1], 4: MULTIPLE_OPT["schemes"][4][1], 6: MULTIPLE_OPT["schemes"][6][1]}
    # getMidSearch2 / 4 / 6 (searchstrategy.h:3669-3706) — checked against these data files in the tests
    spec["schemes"][2].append(mid[2])
    spec["schemes"][4].append(mid[4])
    spec["schemes"][6] += [mid[6], mirror(mid[6])]
    return spec


COLUMBA = _columba()

# NaiveBackTrackingStrategy (searchstrategy.h:2785-2820): one part for every k (the search it lists is never run: every read
# goes to approxMatchesNaive, searchstrategy.cpp:148-152, :442-459)
NAIVE = {"kmer_cutoff": 20, "schemes": {k: [[([0], [0], [k])]] for k in range(1, 14)}}

BY_NAME = {"multiple_opt": MULTIPLE_OPT, "kuch1": KUCH1, "kuch2": KUCH2, "kianfar": KIANFAR, "01*0": O1STAR,
           "pigeon": PIGEON, "minU": MINU, "columba": COLUMBA, "naive": NAIVE}
